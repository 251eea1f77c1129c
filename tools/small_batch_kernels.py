"""Per-kernel time of one small batch (profiling events around every launch): python tools/small_batch_kernels.py [n_jobs] [routing]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import indelpost_amd as ip
from indelpost_amd import synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
g = ip.GpuAligner(0, 3, 2)
if len(sys.argv) > 2:
    g.set_routing(int(sys.argv[2]))
jobs = synth.config2_jobs(n)
g.upload(jobs); g.run(); g.sync()
g.set_profiling(True)
for _ in range(10): g.run(); g.sync()
kt = g.kernel_times()
tot = 0
for k, v in sorted(kt.items(), key=lambda kv: -kv[1][0]):
    if v[1]: print("%-26s %7.3f ms/run  (%d launches/run)" % (k, v[0] / 10, v[1] // 10)); tot += v[0] / 10
print("sum of kernels %.3f ms, run %.3f ms; traceback routing %s" % (tot, g.last_run_ms(), g.traceback_routing()))
