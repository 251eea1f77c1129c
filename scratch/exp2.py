import sys, os
sys.path.insert(0,'.')
import indelpost_amd as ip
from indelpost_amd import synth
jobs=synth.config2_jobs(1000000)
g=ip.GpuAligner(0,3,2); g.upload(jobs)
g.run(); g.sync()
g.run(); g.sync()
print("ok")
