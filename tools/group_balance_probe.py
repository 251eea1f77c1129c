"""config 4 grouped by read length: how well do the four slices balance?  GPU ms per slice and step, for a few work models of shard_bounds"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import indelpost_amd as ip
from indelpost_amd import synth, batch

jobs = synth.config4_jobs()
orig = batch.shard_bounds

def model(name):
    def bounds(n, k, jb=None):
        if jb is None or k <= 1 or name == "product":
            return orig(n, k, jb)
        a, b, pw = name
        L = np.diff(jb.read_off).astype(np.float64)
        Wd = (jb.ref_off[1:] - jb.ref_off[:-1])[jb.ref_id].astype(np.float64)
        w = (L + a) ** pw * (Wd + b)
        c = np.cumsum(w)
        return [0] + [int(np.searchsorted(c, c[-1] * i / k)) for i in range(1, k)] + [n]
    return bounds

for name in ("product", (40, 60, 0.85), (40, 60, 0.7), (80, 60, 0.7), (40, 200, 0.7), (40, 60, 0.5)):
    batch.shard_bounds = model(name)
    g = ip.MultiStreamAligner(0, 3, 2, streams=4)
    g.upload(jobs)
    for _ in range(5): g.run()
    g.sync()
    t0 = time.perf_counter()
    for _ in range(20): g.run()
    g.sync()
    dt = (time.perf_counter() - t0) / 20
    per = [round(p.last_run_ms(), 2) for p in g._active]
    print(name, "%.2f M aln/s" % (jobs.n_jobs / dt / 1e6), "jobs per slice", [s.n_jobs for s in g._slices], "last run ms per slice", per)
    g.close()
