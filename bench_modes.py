"""bench.py's secondary measurement modes (kept out of the headline path).

run_end_to_end  PCIe-inclusive rate on one GPU: host job table in, result records and CIGAR pool out, per step.
run_sharded     ONE job table (config-4 shape) cut over the ranks with JobTable.shard, each rank aligning its shard on
                its GPU and rank 0 gathering every record inside the timed region -- the real shape of BASELINE
                configs[3] "sharded over 8 GPUs" (no collective on the data path, a host-side gather).
"""
import json
import os
import time

import numpy as np


def run_end_to_end(ip, jobs, scoring, dev, streams, steps, depth=None):
    """A stream of job tables through ONE GPU, host memory to host memory.  `depth` aligners take the batches in turn: while
    one computes, another's results are downloaded and a third's next batch is cut, copied in and launched, so the GPU does
    not wait for the host or the PCIe link (each aligner: `streams` contexts, its own device buffers)."""
    depth = depth or int(os.environ.get("IPX_E2E_DEPTH", "3"))
    ring = [ip.MultiStreamAligner(dev, scoring[0], scoring[1], streams=streams) for _ in range(depth)]
    try:
        pinned = all([g.pin_host(jobs) for g in ring])           # input arrays shared, one pinned output pair per aligner
        for g in ring:                                           # warm-up: allocations, launch sizing
            g.align(jobs)
        t0 = time.perf_counter()
        res = None
        for k in range(steps + depth - 1):
            if k < steps:
                ring[k % depth].submit(jobs)                     # batch k is on its way ...
            if k >= depth - 1:
                res = ring[(k - depth + 1) % depth].collect()    # ... while batch k-depth+1 comes back
        dt = (time.perf_counter() - t0) / steps
        return {"value": round(jobs.n_jobs / dt / 1e6, 4), "unit": "million alignments/s", "ms_per_step": round(dt * 1e3, 3),
                "steps": steps, "aligners_in_rotation": depth, "pinned_host_buffers": bool(pinned),
                "sum_score1": int(res.records["score1"].astype(np.int64).sum()), "digest": res.digest(),
                "bytes_in_per_step": int(jobs.reads.nbytes + jobs.read_off.nbytes + jobs.ref_id.nbytes + jobs.gap_open.nbytes
                                         + jobs.gap_ext.nbytes + jobs.refs.nbytes),
                "bytes_out_per_step": int(res.records.nbytes + 4 * int(res.records["cigar_len"].astype(np.int64).sum())),
                "note": "per batch: host job table -> H2D -> pipeline -> D2H of every record and CIGAR; the aligners take batches "
                        "in turn so that transfers and host work of one batch overlap the kernels of the others; never the headline value"}
    finally:
        for g in ring:
            g.close()


def run_sharded(ip, args, dist, rank, world, my_devices, aligner_cls):
    """strong-scaling form: the table is fixed, the ranks split it"""
    from bench import WORKLOADS, make_jobs, check_results
    from indelpost_amd.batch import BatchResult, merge_results, shard_bounds
    from indelpost_amd._lib import RESULT_DTYPE
    name = "4"
    scoring, desc = WORKLOADS[name]
    jobs = make_jobs(name, args.reads_per_gpu)                   # every rank generates the same table (deterministic)
    if args.backend == "emu":                                    # CPU rehearsal: a few jobs are enough
        jobs = jobs.shard(0, 48)
    n_parts = world if world > 1 else len(my_devices)
    b = shard_bounds(jobs.n_jobs, n_parts)
    mine = [rank] if world > 1 else list(range(n_parts))
    aligners = [ip.MultiStreamAligner(dev, scoring[0], scoring[1], streams=args.streams, aligner_cls=aligner_cls)
                for dev in my_devices]

    def step():
        parts = {}
        shards = {k: jobs.shard(b[k], b[k + 1]) for k in mine}   # host packing is part of the job
        for g, k in zip(aligners, mine):
            g.upload(shards[k])
            g.run()
        for g, k in zip(aligners, mine):
            g.sync()
            parts[k] = g.download()
        if dist is not None:                                     # host-side gather on rank 0 (no data-path collective on the GPUs)
            payload = (parts[rank].records.tobytes(), np.ascontiguousarray(parts[rank].cigar_pool).tobytes())
            gathered = [None] * world if rank == 0 else None
            dist.gather_object(payload, gathered, dst=0)
            if rank != 0:
                return None
            ordered = [BatchResult(np.frombuffer(r, RESULT_DTYPE), np.frombuffer(p, np.uint32)) for r, p in gathered]
        else:
            ordered = [parts[k] for k in range(n_parts)]
        return merge_results(ordered)

    for _ in range(max(1, args.warmup)):
        step()
    if dist is not None:
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        res = step()
    if dist is not None:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        import torch
        t = torch.tensor([elapsed], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t[0])
    for g in aligners:
        g.close()
    if rank == 0:
        chk = check_results(name, res, jobs, 0 if args.backend == "hip" else 1)
        out = {"metric": "million read-alignments/sec (150 bp x 300 bp, affine gap)", "mode": "sharded",
               "value": round(jobs.n_jobs * args.steps / elapsed / 1e6, 6), "unit": "million alignments/s",
               "n_gpus": args.gpus, "steps": args.steps, "warmup": args.warmup,
               "ms_per_step": round(elapsed / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "strong",
               "vs_baseline": None, "dtype": "int16 (packed pairs; the 8-bit pass is computed in 16-bit containers)",
               "data": "synthetic",
               "config": {"workload": desc.format(n=jobs.n_jobs) + "; ONE table sharded over the ranks (JobTable.shard), "
                                      "upload + run + download per shard and the host gather on rank 0 inside the timed region",
                          "jobs_total": jobs.n_jobs, "streams_per_gpu": args.streams, "backend": args.backend}}
        out.update(chk)
        print(json.dumps(out))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
