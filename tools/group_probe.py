"""Does a mixed-length batch run faster when its jobs are grouped by read length before the streams cut it?  (config 4 shape)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import indelpost_amd as ip
from indelpost_amd import synth
from indelpost_amd.batch import JobTable

jobs = synth.config4_jobs()
lens = np.diff(jobs.read_off)
order = np.argsort(lens, kind="stable")
# grouped table (numpy, slow: a probe)
new_len = lens[order]
new_off = np.zeros(len(order) + 1, np.int64); np.cumsum(new_len, out=new_off[1:])
reads = np.empty(int(new_off[-1]), np.int8)
for L in np.unique(lens):
    idx = np.flatnonzero(new_len == L)
    src = jobs.read_off[order[idx]]
    reads[(new_off[idx][:, None] + np.arange(L)).ravel()] = jobs.reads[(src[:, None] + np.arange(L)).ravel()]
grouped = JobTable(reads, new_off, jobs.refs, jobs.ref_off, jobs.ref_id[order], jobs.gap_open[order], jobs.gap_ext[order])

def rate(table, by_cells, steps=20, warm=5):
    g = ip.MultiStreamAligner(0, 3, 2, streams=4)
    g.balance_by_cells = by_cells
    g.upload(table)
    for _ in range(warm): g.run()
    g.sync()
    t0 = time.perf_counter()
    for _ in range(steps): g.run()
    g.sync()
    dt = (time.perf_counter() - t0) / steps
    res = g.download()
    g.close()
    return table.n_jobs / dt / 1e6, res

r0, res0 = rate(jobs, False)
r1, res1 = rate(jobs, True)
r2, res2 = rate(grouped, False)
r3, res3 = rate(grouped, True)
back = np.empty_like(res3.records); back[order] = res3.records
same = all((back[f] == res0.records[f]).all() for f in ("score1", "score2", "ref_begin1", "ref_end1", "read_begin1", "read_end1", "ref_end2", "cigar_len", "flag"))
print("as given: %.2f (by count) %.2f (by cells) | grouped by length: %.2f (by count) %.2f (by cells) M aln/s; records equal after un-grouping: %s" % (r0, r1, r2, r3, same))
