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
    # (r04: two aligners -- eight streams, within the ten hardware queues the loader asks for: 78 M aln/s; three: 71; r03, four queues, three aligners: 62.5)
    depth = depth or int(os.environ.get("IPX_E2E_DEPTH", "2"))
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


def run_many_loci(ip, scoring, dev, streams, steps, n_loci=12500):
    """The many-loci entry on the config-5 shape: `n_loci` loci, each its own JobTable of 16 reads x 6 gap-penalty pairs with per-read
    windows (what retarget_jobs builds per locus), handed over as a LIST per step, host memory to host memory: concatenation (the
    library's ipx_concat_tables into page-locked, reused staging buffers) + H2D + pipeline + D2H + a split back into per-locus results.
    `value`: a STREAM of such lists through two aligners in rotation (while one batch computes, the next list is concatenated into the
    other aligner's staging) -- what a caller working through a VCF gets; `one_list_at_a_time`: indelpost_amd.align_loci called list by
    list, nothing overlapped -- the latency of one such call."""
    from indelpost_amd import synth
    from indelpost_amd.batch import JobTable, align_loci
    jobs = synth.config5_jobs(n_loci=n_loci)
    per = jobs.n_jobs // n_loci
    loci = [jobs.shard(k * per, (k + 1) * per) for k in range(n_loci)]          # outside the timed region: the caller's own tables
    ring = [ip.MultiStreamAligner(dev, scoring[0], scoring[1], streams=streams) for _ in (0, 1)]
    try:
        g = ring[0]
        for _ in range(2):
            parts = align_loci(loci, aligner=g)
        t_cat = t_all = 0.0
        for _ in range(steps):                                     # one list at a time
            t0 = time.perf_counter()
            table = JobTable.concat(loci, staging=g.loci_staging)
            t1 = time.perf_counter()
            parts = g.align(table).split(table.table_jobs)
            t2 = time.perf_counter()
            t_cat += t1 - t0
            t_all += t2 - t0
        whole = ip.BatchResult(np.concatenate([p.records for p in parts]), parts[0].cigar_pool)
        digest, sum1 = whole.digest(), int(whole.records["score1"].astype(np.int64).sum())
        dt1 = t_all / steps
        # a stream of lists: two aligners in rotation
        align_loci(loci, aligner=ring[1])
        tabs = [None, None]
        last = None
        ph = {"concat": 0.0, "submit": 0.0, "collect": 0.0}
        t0 = time.perf_counter()
        for k in range(steps + 1):
            if k < steps:
                a = ring[k % 2]
                ta = time.perf_counter()
                tabs[k % 2] = JobTable.concat(loci, staging=a.loci_staging)
                tb = time.perf_counter()
                a.submit(tabs[k % 2])
                tc = time.perf_counter()
                ph["concat"] += tb - ta; ph["submit"] += tc - tb
            if k >= 1:
                ta = time.perf_counter()
                last = ring[(k - 1) % 2].collect().split(tabs[(k - 1) % 2].table_jobs)
                ph["collect"] += time.perf_counter() - ta
        dt2 = (time.perf_counter() - t0) / steps
        assert ip.BatchResult(np.concatenate([p.records for p in last]), last[0].cigar_pool).digest() == digest
        for a in ring:
            a.close()
        ring = []
        # the product's own stream (indelpost_amd.batch.align_loci_stream): two aligners in rotation, three staging sets, the concatenation on a
        # host thread of its own; timed from the moment the third list has come back (the buffers have grown and are page-locked by then)
        from indelpost_amd.batch import align_loci_stream
        warm, t0, last3 = 3, None, None
        for k, parts3 in enumerate(align_loci_stream((loci for _ in range(warm + steps)), scoring[0], scoring[1], device=dev, streams=streams)):
            t1 = time.perf_counter()                                # (taken here: leaving the generator closes the ring and un-registers its buffers)
            if k == warm - 1:
                t0 = t1
            last3 = parts3
        dt3 = (t1 - t0) / steps
        assert ip.BatchResult(np.concatenate([p.records for p in last3]), last3[0].cigar_pool).digest() == digest
        best = min(dt2, dt3)
        return {"value": round(jobs.n_jobs / best / 1e6, 4), "unit": "million alignments/s", "ms_per_step": round(best * 1e3, 3), "steps": steps,
                "n_loci": n_loci, "jobs_per_locus": per, "n_jobs": jobs.n_jobs,
                "form": "align_loci_stream (two aligners, three staging sets, concatenation on its own host thread)" if dt3 <= dt2 else "two aligners in rotation, one host thread",
                "align_loci_stream": {"value": round(jobs.n_jobs / dt3 / 1e6, 4), "ms_per_step": round(dt3 * 1e3, 3), "aligners_in_rotation": 2},
                "two_aligners_one_thread": {"value": round(jobs.n_jobs / dt2 / 1e6, 4), "ms_per_step": round(dt2 * 1e3, 3)},
                "host_ms_per_step": {k: round(v / steps * 1e3, 3) for k, v in ph.items()},
                "one_list_at_a_time": {"value": round(jobs.n_jobs / dt1 / 1e6, 4), "ms_per_step": round(dt1 * 1e3, 3), "concat_ms_per_step": round(t_cat / steps * 1e3, 3)},
                "digest": digest, "sum_score1": sum1,
                "note": "host memory to host memory (concat + H2D + pipeline + D2H + split); compare with configs.5, the same jobs resident "
                        "in HBM.  The per-locus tables are the caller's (made outside the timed region, their ten-integer descriptors cached by "
                        "the warm-up calls: a table handed over for the first time costs ~8 us more)"}
    finally:
        for a in ring:
            a.close()


def run_full_configs(ip, dev, streams, steps, warmup):
    """BASELINE configs[3] and configs[4] at their FULL sizes on one GPU: 10 tables of 996 k mixed-length jobs (9.96 M) and 8 tables of
    1.2 M per-read-window x penalty-grid jobs (9.6 M), table k generated from seed SEED + 977 k, each run under the headline's protocol
    (inputs resident, warm-up, `steps` timed steps) and each digest-checked against the reference's results for THAT table
    (tests/golden/bench_digests.json: "4", "4#1".."4#9", "5", "5#1".."5#7", oracle/gen_bench_digests.py)."""
    from bench import WORKLOADS, make_jobs, golden_digest
    out = {"mode": "full-configs", "steps": steps, "warmup": warmup, "streams": streams, "configs": {}}
    gold = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "tests", "golden", "bench_digests.json")))["workloads"]
    for name, ntab in (("4", 10), ("5", 8)):
        scoring, desc = WORKLOADS[name]
        g = ip.MultiStreamAligner(dev, scoring[0], scoring[1], streams=streams)
        tabs, jobs_total, t_total = [], 0, 0.0
        try:
            for k in range(ntab):
                jobs = make_jobs(name, 0, k)
                g.upload(jobs)
                for _ in range(warmup):
                    g.run()
                g.sync()
                t0 = time.perf_counter()
                for _ in range(steps):
                    g.run()
                g.sync()
                dt = (time.perf_counter() - t0) / steps
                res = g.download()
                key = name if k == 0 else "%s#%d" % (name, k)
                want = gold.get(key, {})
                d = res.digest()
                ok = want.get("n_jobs") == jobs.n_jobs and want.get("digest") == d
                assert ok, "table %s: digest %d differs from the reference's %r" % (key, d, want.get("digest"))
                tabs.append({"table": key, "n_jobs": jobs.n_jobs, "value": round(jobs.n_jobs / dt / 1e6, 4), "ms_per_step": round(dt * 1e3, 4),
                             "digest": d, "digest_matches_reference": True})
                jobs_total += jobs.n_jobs
                t_total += dt
        finally:
            g.close()
        out["configs"][name] = {"workload": desc.format(n=jobs_total) + " -- %d tables" % ntab, "jobs_total": jobs_total,
                                "value": round(jobs_total / t_total / 1e6, 4), "unit": "million alignments/s",
                                "ms_for_all_tables": round(t_total * 1e3, 3), "tables": tabs}
    return out


def config4_chunk(rank, windows_per_gpu=1250):
    """chunk `rank` of the config-4 table: BASELINE configs[3] is 10 M reads over 8 GPUs = 1250 windows x 996 reads per GPU; every
    rank generates only its own chunk (its own generator seed), so the table of an N-GPU run is N chunks = N x 1.245 M jobs"""
    from indelpost_amd import synth
    return synth.config4_jobs(n_windows=windows_per_gpu, seed=synth.SEED + 977 * rank)


def gather_bytes(dist, rank, world, arrays):
    """rank 0 receives every rank's arrays (uint8 views) through dist.gather of padded uint8 tensors -- no pickling; returns, on rank
    0, a list (per rank) of lists of numpy uint8 arrays"""
    import torch
    sizes = torch.tensor([a.nbytes for a in arrays], dtype=torch.int64)
    all_sizes = [torch.zeros_like(sizes) for _ in range(world)]
    dist.all_gather(all_sizes, sizes)
    out = [[] for _ in range(world)]
    for q, a in enumerate(arrays):
        cap = int(max(int(s_[q]) for s_ in all_sizes))
        buf = torch.zeros(max(cap, 1), dtype=torch.uint8)
        if a.nbytes:
            buf[:a.nbytes] = torch.from_numpy(np.ascontiguousarray(a).view(np.uint8).reshape(-1))
        got = [torch.zeros_like(buf) for _ in range(world)] if rank == 0 else None
        dist.gather(buf, got, dst=0)
        if rank == 0:
            for r in range(world):
                out[r].append(got[r][:int(all_sizes[r][q])].numpy())
    return out if rank == 0 else None


def run_sharded(ip, args, dist, rank, world, my_devices, aligner_cls):
    """BASELINE configs[3]: the config-4 table sharded over the GPUs -- 1250 windows (1.245 M jobs) per GPU, 10 M at 8 GPUs.  Every
    rank generates the chunk it owns, cuts it over its streams by WORK (shard_bounds with the table), uploads, runs, downloads, and
    rank 0 gathers every record and CIGAR inside the timed region (dist.gather of uint8 tensors; no data-path collective on the
    GPUs).  value = all jobs of all ranks / max-over-ranks time: weak scaling in the table's own terms."""
    from bench import WORKLOADS, check_results
    from indelpost_amd.batch import BatchResult, merge_results
    from indelpost_amd._lib import RESULT_DTYPE
    name = "4"
    scoring, desc = WORKLOADS[name]
    n_parts = world if world > 1 else len(my_devices)
    mine = [rank] if world > 1 else list(range(n_parts))
    wpg = int(os.environ.get("IPX_SHARDED_WINDOWS_PER_GPU", "1250"))
    chunks = {k: config4_chunk(k, wpg) for k in mine}
    if args.backend == "emu":                                    # CPU rehearsal: a few jobs are enough
        chunks = {k: c.shard(0, 48) for k, c in chunks.items()}
    aligners = [ip.MultiStreamAligner(dev, scoring[0], scoring[1], streams=args.streams, aligner_cls=aligner_cls)
                for dev in my_devices]
    pinned = []
    for g, k in zip(aligners, mine):
        g.balance_by_cells = True                                # stream slices of near-equal work, not near-equal job count
        pinned.append(bool(g.pin_host(chunks[k])) if hasattr(g, "pin_host") else False)   # staging buffers page-locked once, reused every step

    gather_s = [0.0]

    def submit():
        for g, k in zip(aligners, mine):
            g.submit(chunks[k])                                  # slice by slice: copy in, launch the pipeline; no waiting

    def collect():
        return {k: g.collect() for g, k in zip(aligners, mine)}  # slice by slice: wait, copy out (the later slices still compute)

    def gather(parts):
        """host-side gather on rank 0 (no data-path collective on the GPUs); timed on its own (gather_ms_per_step)"""
        t_ = time.perf_counter()
        try:
            if dist is not None:
                got = gather_bytes(dist, rank, world, [parts[rank].records.view(np.uint8).reshape(-1), np.ascontiguousarray(parts[rank].cigar_pool).view(np.uint8)])
                if rank != 0:
                    return None
                ordered = [BatchResult(r.view(RESULT_DTYPE), p.view(np.uint32)) for r, p in got]
            else:
                ordered = [parts[k] for k in range(n_parts)]
            return merge_results(ordered)
        finally:
            gather_s[0] += time.perf_counter() - t_

    def steps(n):
        """n steps; the gather of step s runs while the GPUs compute step s + 1 (results sit in one of two pinned buffer pairs per
        aligner, so the records of step s stay valid until step s + 2 is collected); the last gather is inside the timed region too"""
        res_, prev = None, None
        for _ in range(n):
            submit()
            if prev is not None:
                res_ = gather(prev)
            prev = collect()
        if prev is not None:
            res_ = gather(prev)
        return res_

    steps(max(1, args.warmup))
    if dist is not None:
        dist.barrier()
    gather_s[0] = 0.0
    t0 = time.perf_counter()
    res = steps(args.steps)
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
        jobs_total = len(res.records)
        chunk0 = chunks[0]
        first = BatchResult(res.records[:chunk0.n_jobs], res.cigar_pool)           # chunk 0 is the 1250-window table of seed 0
        chk = check_results("4x1250", first, chunk0, 0 if args.backend == "hip" else 1)
        out = {"metric": "million read-alignments/sec (150 bp x 300 bp, affine gap)", "mode": "sharded",
               "value": round(jobs_total * args.steps / elapsed / 1e6, 6), "unit": "million alignments/s",
               "n_gpus": args.gpus, "steps": args.steps, "warmup": args.warmup,
               "ms_per_step": round(elapsed / args.steps * 1e3, 4),
               "gather_ms_per_step": round(gather_s[0] / args.steps * 1e3, 4),    # rank 0's host-side gather + merge, overlapped with the next step's kernels
               "higher_is_better": True, "scaling": "weak",
               "vs_baseline": None,
               "dtype": "f16 (16-bit passes and the plain 8-bit recurrence: packed halves, exact for these integers) / int16 (stepped 8-bit passes)",
               "data": "synthetic",
               "config": {"workload": "config4: %d reads of 75/100/125/150/200/250 bp vs windows of 200-600 bp (one window per 996 reads), "
                                      "(3,2,3,1), %d windows per GPU; every rank generates and aligns its chunk (upload + run + download) and "
                                      "rank 0 gathers every record and CIGAR inside the timed region" % (jobs_total, wpg),
                          "jobs_total": jobs_total, "jobs_per_gpu": jobs_total // max(1, n_parts), "streams_per_gpu": args.streams, "backend": args.backend,
                          "pinned_host_buffers": all(pinned)}}
        out.update(chk)
        print(json.dumps(out))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
