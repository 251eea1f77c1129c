"""Traceback routing of one batch: jobs per first band width 1..7 and jobs handed to the one-wave-per-job kernel (GpuAligner.traceback_routing)."""
import sys; sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import numpy as np
import indelpost_amd as ip
from indelpost_amd import synth
jobs=synth.config2_jobs(200000)
g=ip.GpuAligner(0,3,2)
g.upload(jobs); g.run(); g.sync()
print("2b", g.traceback_routing())
jobs=synth.config5_jobs(20000)
g.upload(jobs); g.run(); g.sync()
print("5", g.traceback_routing())
jobs=synth.config4_jobs(n_windows=250)
g.upload(jobs); g.run(); g.sync()
print("4 (249k jobs)", g.traceback_routing())
g.set_profiling(1)
for _ in range(5): g.run(); g.sync()
kt=g.kernel_times()
print({k:(round(v[0]/5,3),v[1]//5) for k,v in kt.items() if k.startswith("traceback") or k.startswith("prove") or "exact" in k or "check" in k or "low2" in k or k.startswith("plan")})
