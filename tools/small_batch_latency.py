"""Fixed cost of one batched call: upload / run / sync / download for small job tables."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import indelpost_amd as ip
from indelpost_amd import synth
g = ip.GpuAligner(0, 3, 2)
for n in (1, 100, 1000, 10000, 100000, 1000000):
    jobs = synth.config2_jobs(n)
    g.upload(jobs); g.run(); g.sync(); g.download()
    T = {"upload": 0, "run": 0, "sync": 0, "download": 0}
    R = 20 if n <= 100000 else 4
    for _ in range(R):
        t0 = time.perf_counter(); g.upload(jobs)
        t1 = time.perf_counter(); g.run()
        t2 = time.perf_counter(); g.sync()
        t3 = time.perf_counter(); g.download()
        t4 = time.perf_counter()
        T["upload"] += t1 - t0; T["run"] += t2 - t1; T["sync"] += t3 - t2; T["download"] += t4 - t3
    tot = sum(T.values()) / R
    print("n=%6d: " % n + "  ".join("%s %.3f ms" % (k, v / R * 1e3) for k, v in T.items()) + "  | total %.3f ms = %.2f us/job, gpu %.3f ms" % (tot * 1e3, tot * 1e6 / n, g.last_run_ms()))

# the same end to end through the 4-stream aligner (what sswpy/localn use), 1 M jobs
jobs = synth.config2_jobs(1000000)
m = ip.MultiStreamAligner(0, 3, 2, streams=4)
m.align(jobs)
t0 = time.perf_counter()
for _ in range(4):
    res = m.align(jobs)
dt = (time.perf_counter() - t0) / 4
print("MultiStreamAligner.align, 1M jobs, host buffers in -> records + CIGAR pool out: %.1f ms = %.1f M aln/s" % (dt * 1e3, 1.0 / dt))
