"""Golden digests for bench.py's workloads (TEST INFRASTRUCTURE, build container only).

Runs the reference's own ssw.c (oracle/_ref/libssw_ref.so, compiled unmodified from /root/reference by
oracle/Makefile) over the exact job tables bench.py times -- config 2a, 2b (first 100 000 reads, the SURVEY 8c
checksum dataset, plus the full million), the config-4 and config-5 shapes -- and writes, per workload, the
record digest of indelpost_amd.batch.record_digest (xxHash64 over every public field and the per-job CIGAR hash) together with plain sums, to tests/golden/bench_digests.json.

    python oracle/gen_bench_digests.py          # about 1-2 minutes per million jobs on 8 cores
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle import oracle as O                                   # noqa: E402
from indelpost_amd import synth                                  # noqa: E402
from indelpost_amd.batch import record_digest                    # noqa: E402


def workloads():
    yield "2a", (1, 1), synth.config2_jobs(1_000_000, 3, 1)
    yield "2b", (3, 2), synth.config2_jobs(1_000_000, 3, 1)
    yield "4", (3, 2), synth.config4_jobs()
    yield "5", (3, 2), synth.config5_jobs()
    yield "4x1250", (3, 2), synth.config4_jobs(n_windows=1250)      # chunk 0 of bench.py --sharded (1250 windows per GPU)
    # BASELINE's full sizes on one GPU (bench.py --full-configs): configs[3] = 10 tables of 996 k jobs, configs[4] = 8 tables of 1.2 M jobs,
    # table k generated from seed SEED + 977 k (k = 0 is the table above)
    for k in range(1, 10):
        yield "4#%d" % k, (3, 2), synth.config4_jobs(seed=synth.SEED + 977 * k)
    for k in range(1, 8):
        yield "5#%d" % k, (3, 2), synth.config5_jobs(seed=synth.SEED + 977 * k)


def main():
    O.build()
    assert O.have_reference(), "needs oracle/_ref/libssw_ref.so (run `make -C oracle`)"
    be = O.Backend("reference")
    cores = len(os.sched_getaffinity(0))
    path = os.path.join(ROOT, "tests", "golden", "bench_digests.json")
    out = {"generator": "oracle/gen_bench_digests.py", "checker": "reference ssw.c (oracle/_ref/libssw_ref.so)", "workloads": {}}
    only_missing = "--missing" in sys.argv                      # keep what the file already has (the reference has not changed)
    if only_missing and os.path.exists(path):
        out = json.load(open(path))
    for name, scoring, jobs in workloads():
        if only_missing and name in out["workloads"]:
            continue
        rec = O.cpu_batch_results(be, jobs, O.dna_matrix(*scoring), cores)
        assert (rec["is_null"] == 0).all()
        out["workloads"][name] = {
            "scoring": list(scoring), "n_jobs": int(jobs.n_jobs), "digest": record_digest(rec, rec["cigar_hash"]),
            "sum_score1": int(rec["score1"].astype(np.int64).sum()), "sum_score2": int(rec["score2"].astype(np.int64).sum()),
            "sum_cigar_len": int(rec["cigar_len"].astype(np.int64).sum()), "flag_nonzero": int((rec["flag"] != 0).sum()),
            "flag1": int((rec["flag"] == 1).sum())}
        print(name, out["workloads"][name], flush=True)
    with open(path, "w") as f:
        json.dump(out, f, indent=1)
        f.write("\n")


if __name__ == "__main__":
    main()
