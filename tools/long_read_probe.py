"""Reads beyond the register-resident kernels: k_dp_wide (one wavefront per read) against k_dp_long (the transcribed loops, ROUTE_NO_WIDE).
GPU time of one call per configuration, kernel times by class, and the records of both routes compared.
    python tools/long_read_probe.py [n_reads] [read_len] [window_len]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import indelpost_amd as ip
from indelpost_amd.batch import JobTable, ROUTE_NO_WIDE

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4
L = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
W = int(sys.argv[3]) if len(sys.argv) > 3 else 20000
rng = np.random.default_rng(11)
refs = [rng.integers(0, 4, W).astype(np.int8) for _ in range(min(n, 8))]
reads, rid = [], []
for i in range(n):
    w = refs[i % len(refs)]
    st = int(rng.integers(0, W - L - 10))
    r = w[st:st + L].copy()
    m = rng.random(L) < 0.02
    r[m] = rng.integers(0, 4, int(m.sum()))
    if i % 2:
        r = np.concatenate([r[:L // 2], r[L // 2 + 11:]])
    reads.append(r); rid.append(i % len(refs))
jobs = JobTable.from_sequences(reads, refs, rid, [3] * n, [1] * n, encoded=True)
out = {}
for name, routing in (("k_dp_wide", 0), ("k_dp_long", ROUTE_NO_WIDE)):
    g = ip.GpuAligner(0, 3, 2)
    g.set_routing(routing)
    g.set_profiling(True)
    g.upload(jobs)
    best = 1e9
    for rep in range(3):
        g.run(); g.sync()
        best = min(best, g.last_run_ms())
    res = g.download()
    kt = g.kernel_times()
    top = sorted(((t / 3.0, k) for k, (t, c) in kt.items()), reverse=True)[:6]
    print("%s: %d reads of %d bp vs %d bp: %.2f ms of GPU time per call; kernel ms per call %s" % (name, n, L, W, best, [(k, round(ms, 2)) for ms, k in top]))
    out[name] = res
    g.close()
a, b = out["k_dp_wide"], out["k_dp_long"]
# (the digest covers every public field and every job's CIGAR; cigar_off, the place in the pool, depends on the order the jobs finished in)
print("digests equal:", a.digest() == b.digest(), " mean score1 %.1f, %d of %d jobs with a CIGAR" % (float(a.records["score1"].mean()), int((a.records["cigar_len"] > 0).sum()), n))
