"""Where does the wall time of the batched Python driver go?  align_pileup on N reads (2N jobs)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import indelpost_amd as ip
from indelpost_amd import synth, localn
from indelpost_amd.batch import dna_score_matrix

N = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
jobs = synth.config2_jobs(N)
letters = np.array(list("ACGTN"))
ref = "".join(letters[jobs.refs])
reads = ["".join(letters[jobs.reads[i * 150:(i + 1) * 150]]) for i in range(N)]
mut = ref[:140] + ref[146:]          # a 6 bp deletion
for rep in range(3):
    t0 = time.perf_counter()
    jt = localn.realign_pileup_jobs(reads, mut, ref, 3, 1)
    t1 = time.perf_counter()
    g = localn._gpu(0)
    g.set_scoring(matrix=dna_score_matrix(3, 2), flag=1, score_size=2)
    res = g.align(jt)
    t2 = time.perf_counter()
    out = [(localn._alignment_from(res, 2 * k), localn._alignment_from(res, 2 * k + 1)) for k in range(N)]
    t3 = time.perf_counter()
    pairs = localn.align_pileup(reads, mut, ref, 3, 2, 3, 1)
    t4 = time.perf_counter()
    print("N=%d reads (%d jobs): build jobs %.1f ms, upload+run+download %.1f ms, tuples %.1f ms | align_pileup total %.1f ms = %.2f us/job"
          % (N, 2 * N, (t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3, (t4 - t3) * 1e3, (t4 - t3) * 1e6 / (2 * N)))
assert pairs == out
