"""align_loci_stream on the many-loci table: python tools/loci_stream_probe.py depth [steps]   (GPU_MAX_HW_QUEUES from the environment)"""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import indelpost_amd as ip
from indelpost_amd import synth
from indelpost_amd.batch import align_loci_stream
depth = int(sys.argv[1]); steps = int(sys.argv[2]) if len(sys.argv) > 2 else 8
jobs = synth.config5_jobs(n_loci=12500)
per = jobs.n_jobs // 12500
loci = [jobs.shard(k * per, (k + 1) * per) for k in range(12500)]
warm, t0 = 3, None
marks = []
for k, parts in enumerate(align_loci_stream((loci for _ in range(warm + steps)), 3, 2, depth=depth)):
    marks.append(time.perf_counter())
    if k == warm - 1:
        t0 = time.perf_counter()
dt = (marks[-1] - t0) / steps
print(json.dumps({"depth": depth, "queues": os.environ.get("GPU_MAX_HW_QUEUES"), "ms_per_list": round(dt * 1e3, 2), "M_aln_s": round(jobs.n_jobs / dt / 1e6, 2),
                  "gaps_ms": [round((b - a) * 1e3, 1) for a, b in zip(marks, marks[1:])]}))
