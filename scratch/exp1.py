import sys, os, json, time, numpy as np
sys.path.insert(0,'.')
import indelpost_amd as ip
from indelpost_amd import synth
jobs=synth.config2_jobs(1000000)
g=ip.GpuAligner(0,3,2); g.upload(jobs)
for it in range(2): g.run()
g.sync(); g.set_profiling(True)
for it in range(3): g.run()
g.sync()
kt=g.kernel_times()
print(os.environ.get("IPX_DEBUG_MAXCOLS"), {k:round(v[0]/3,2) for k,v in kt.items() if v[0]/3>0.3}, "step", round(g.last_run_ms(),2))
