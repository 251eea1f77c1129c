import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import indelpost_amd as ip
from indelpost_amd import synth
jobs = synth.config2_jobs(1000000)
for prof in (False, True, False, True):
    g = ip.MultiStreamAligner(0, 3, 2, streams=4)
    g.upload(jobs)
    for _ in range(2): g.run()
    g.sync()
    g.set_profiling(prof)
    t0 = time.perf_counter()
    for _ in range(10): g.run()
    g.sync()
    dt = (time.perf_counter() - t0) / 10
    print("profiling=%s: %.2f ms/step = %.2f M aln/s" % (prof, dt * 1e3, 1.0 / dt))
    g.close()
