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
