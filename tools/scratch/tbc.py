import sys; sys.path.insert(0,'/root/repo')
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
