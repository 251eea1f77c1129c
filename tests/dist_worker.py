"""Worker of tests/test_bench_dist.py: one rank of a 2-process gloo job (CPU).  Each rank aligns its
contiguous shard of a job table (emulated kernels), rank 0 gathers and writes the merged records."""
import os
import sys

import numpy as np
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from indelpost_amd.batch import JobTable, merge_results, shard_bounds  # noqa: E402
from tests.emu_backend import EmuAligner  # noqa: E402


def make_jobs():
    rng = np.random.default_rng(77)
    refs = [rng.integers(0, 4, n).astype(np.int8) for n in (180, 240)]
    reads, rid = [], []
    for i in range(14):
        k = i % 2
        st = int(rng.integers(0, 100))
        r = refs[k][st:st + 50 + i].copy()
        r[rng.integers(0, len(r))] ^= 1
        reads.append(r)
        rid.append(k)
    return JobTable.from_sequences(reads, refs, rid, 3, 1, encoded=True)


def main(out_path):
    dist.init_process_group(backend="gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    jobs = make_jobs()
    b = shard_bounds(jobs.n_jobs, world)
    part = EmuAligner(rank, 3, 2).align(jobs.shard(b[rank], b[rank + 1]))
    dist.barrier()
    gathered = [None] * world
    dist.all_gather_object(gathered, (part.records.tobytes(), part.cigar_pool.tobytes()))
    if rank == 0:
        from indelpost_amd._lib import RESULT_DTYPE
        from indelpost_amd.batch import BatchResult
        parts = [BatchResult(np.frombuffer(r, RESULT_DTYPE).copy(), np.frombuffer(p, np.uint32).copy())
                 for r, p in gathered]
        merged = merge_results(parts)
        whole = EmuAligner(0, 3, 2).align(jobs)
        ok = all(merged.as_dict(i) == whole.as_dict(i) for i in range(jobs.n_jobs))
        with open(out_path, "w") as f:
            f.write("OK %d\n" % jobs.n_jobs if ok else "MISMATCH\n")
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main(sys.argv[1])
