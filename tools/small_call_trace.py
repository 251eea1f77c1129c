"""One small batched call, repeated, for a rocprofv3 --kernel-trace timeline:
    rocprofv3 --kernel-trace --output-format csv -d gpurun_out/sc -- python3 tools/small_call_trace.py [n_jobs] [routing] [runs]
    python3 tools/small_call_trace.py --report gpurun_out/sc/*/*_kernel_trace.csv     (prints the LAST run: kernel, start, duration, gap)"""
import sys, os, csv, re
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def report(path):
    rows = list(csv.DictReader(open(path)))
    ks = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows)
    inits = [i for i, k in enumerate(ks) if k[2].startswith("k_init")]
    if len(inits) < 2:
        print("fewer than two runs in the trace"); return
    a, b = inits[-2], inits[-1]
    run = ks[a:b]
    t0 = run[0][0]
    prev_end = t0
    busy = 0
    for s, e, n in run:
        m = re.match(r"void (k_\w+)<([^>]*)>", n)
        nm = "%s<%s>" % (m.group(1), m.group(2).replace(" ", "")) if m else n.split("(")[0]
        print("%8.1f us  +%7.1f us  gap %6.1f us  %s" % ((s - t0) / 1e3, (e - s) / 1e3, (s - prev_end) / 1e3, nm[:90]))
        busy += e - s
        prev_end = max(prev_end, e)
    print("run: %d launches, first start to last end %.1f us, kernels busy %.1f us" % (len(run), (prev_end - t0) / 1e3, busy / 1e3))


if len(sys.argv) > 2 and sys.argv[1] == "--report":
    report(sys.argv[2]); sys.exit(0)
import indelpost_amd as ip
from indelpost_amd import synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
g = ip.GpuAligner(0, 3, 2)
if len(sys.argv) > 2:
    g.set_routing(int(sys.argv[2]))
runs = int(sys.argv[3]) if len(sys.argv) > 3 else 6
jobs = synth.config2_jobs(n)
g.upload(jobs)
for _ in range(runs):
    g.run(); g.sync()
print("n=%d gpu %.3f ms" % (n, g.last_run_ms()))
