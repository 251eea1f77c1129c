"""PortAligner: the GpuAligner interface on top of the oracle's restatement of ssw.c (TEST INFRASTRUCTURE).

Lets the CPU-only suite replay the driver fixtures (tests/golden/driver_cases.json) through indelpost_amd's host code:
the drivers build their job tables and read their results back exactly as they do on the GPU, only the alignments come
from the oracle.  Never imported by the package."""
import numpy as np

from indelpost_amd._lib import RESULT_DTYPE
from indelpost_amd.batch import BatchResult, dna_score_matrix


class PortAligner:
    def __init__(self, oracle_mod, device=0):
        self._O = oracle_mod
        self._port = oracle_mod.Backend("port")
        self.matrix = dna_score_matrix(2, 2)
        self.n_calls = 0
        self.n_jobs = 0

    def set_scoring(self, match_score=2, mismatch_penalty=2, matrix=None, flag=1, filters=0, filterd=0, score_size=2):
        self.matrix = dna_score_matrix(match_score, mismatch_penalty) if matrix is None else np.ascontiguousarray(matrix, np.int8)

    def set_routing(self, flags):
        pass

    def align(self, jobs):
        n = jobs.n_jobs
        self.n_calls += 1
        self.n_jobs += n
        rec = np.zeros(n, RESULT_DTYPE)
        pool = []
        for i in range(n):
            rid = int(jobs.ref_id[i])
            e = self._port.align(jobs.reads[jobs.read_off[i]:jobs.read_off[i + 1]], jobs.refs[jobs.ref_off[rid]:jobs.ref_off[rid + 1]],
                                 self.matrix, int(jobs.gap_open[i]), int(jobs.gap_ext[i]))
            r = rec[i]
            r["score1"], r["score2"] = e["score1"], e["score2"]
            for f in ("ref_begin1", "ref_end1", "read_begin1", "read_end1", "ref_end2", "flag"):
                r[f] = e[f]
            if e["cigar"] is not None:
                r["cigar_off"], r["cigar_len"] = len(pool), len(e["cigar"])
                pool.extend(int(c) for c in e["cigar"])
        return BatchResult(rec, np.array(pool, np.uint32))
