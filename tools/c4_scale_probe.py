#!/usr/bin/env python3
"""Scratch probe: throughput of the config-4 shape as a function of table size and stream count (launch granularity)."""
import sys, time
sys.path.insert(0, ".")
import indelpost_amd as ip
from indelpost_amd import synth

for nwin, streams in ((1000, 4), (2000, 4), (4000, 4), (4000, 2), (4000, 1), (1000, 1)):
    jobs = synth.config4_jobs(n_windows=nwin)
    g = ip.MultiStreamAligner(0, 3, 2, streams=streams)
    g.upload(jobs)
    for _ in range(2):
        g.run()
    g.sync()
    t0 = time.perf_counter()
    K = 4
    for _ in range(K):
        g.run()
    g.sync()
    dt = (time.perf_counter() - t0) / K
    print("windows %5d jobs %8d streams %d: %.2f ms/step = %.1f M aln/s" % (nwin, jobs.n_jobs, streams, dt * 1e3, jobs.n_jobs / dt / 1e6), flush=True)
    g.close()
