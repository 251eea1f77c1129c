#!/usr/bin/env python3
"""bench.py -- million read-alignments/s of the striped-SW realignment path on MI355X.

Workload (BASELINE.json metric / configs[1], concretely SURVEY.md 8d "config 2b"): per GPU, 1M
synthetic 150 bp reads vs one 300 bp window, indelPost default scoring (match 3, mismatch 2,
gap_open 3, gap_ext 1), flag=1 -> every alignment returns score1/score2, the five coordinates and
the CIGAR.  A step = one full pass of the pipeline (8-bit forward, 16-bit forward rescore, reverse,
banded traceback) over that batch with the inputs already resident in HBM.

  python bench.py --gpus N --steps K --warmup W
  N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...
         (one process per GPU; torch.distributed/gloo is used only for the barrier and the
         max-over-ranks reduction; the reads are sharded, there is no data-path collective)

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

READS_PER_GPU = 1_000_000
READ_LEN, WINDOW_LEN = 150, 300
SCORING = (3, 2, 3, 1)
HBM_PEAK_GBS = 8000.0            # MI355X_MICROARCH.md: 8.0 TB/s spec
VALU_CLOCK_GHZ = 2.4             # peak engine clock; the DP kernels sustain ~2.3 (GRBM_GUI_ACTIVE / 8 / duration)
PK16_CYCLES_PER_INST = 3.13      # v_pk_{add,sub,max}_i16, v_perm_b32, DPP moves at 4 waves/SIMD (profiles/r01_valu_issue_ubench.txt)
# algorithmic HBM bytes per alignment: read codes + 8 B offsets in, 28 B fixed record
# + 4 B per CIGAR op out; the shared window is amortised over the batch
ALG_BYTES_FIXED = READ_LEN + 8 + 28


def cpu_baseline(jobs, scoring, budget_s=12.0):
    """Reference per-read loop on the host cores (oracle/_ref = the reference's ssw.c when built,
    else this repo's scalar port), on a bounded sample of the same workload."""
    from oracle import oracle as O
    O.build()
    kind = "reference" if O.have_reference() else "port"
    be = O.Backend(kind)
    mat = O.dna_matrix(scoring[0], scoring[1])
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)

    def run(n):
        return be.cpu_baseline(jobs.reads[:n * READ_LEN], jobs.read_off[:n + 1], jobs.refs, jobs.ref_off,
                               jobs.ref_id[:n], jobs.gap_open[:n], jobs.gap_ext[:n], mat, cores)
    n0 = min(jobs.n_jobs, 4000 * cores if kind == "reference" else 300 * cores)
    sec, _, _ = run(n0)
    rate = n0 / max(sec, 1e-9)
    n = int(min(jobs.n_jobs, max(n0, rate * budget_s)))
    reps = max(1, int(round(rate * budget_s / n)))        # the sample is bounded by the batch: repeat it to fill the budget
    tot, chk = 0.0, 0
    for _ in range(reps):
        sec, chk, _ = run(n)
        tot += sec
    return {"value": round(n * reps / tot / 1e6, 6), "unit": "million alignments/s", "cores": cores, "kind": kind,
            "sample": "first %d reads of the same workload x %d repeats, %d threads, %.1f s of CPU wall time; "
                      "sum(score1)=%d per repeat" % (n, reps, cores, tot, chk)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--reads-per-gpu", type=int, default=READS_PER_GPU)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--streams", type=int, default=4, help="HIP streams (independent slices of the batch) per GPU")
    ap.add_argument("--all-kernel-times", action="store_true", help="HIP events around every launch, not only the DP kernels (costs ~1.5 %%)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend="gloo", rank=rank, world_size=world)

    import indelpost_amd as ip
    from indelpost_amd import synth

    n_gpus = args.gpus
    # one process per GPU under torchrun; a plain `python bench.py --gpus N` drives N GPUs from threads
    my_devices = [local_rank] if world > 1 else list(range(n_gpus))
    if os.environ.get("IPX_BENCH_FORCE_DEVICE"):           # rehearsal of the N>1 path on a one-GPU box
        my_devices = [int(os.environ["IPX_BENCH_FORCE_DEVICE"])] * len(my_devices)
    n = args.reads_per_gpu
    aligners, tables = [], []
    for k, dev in enumerate(my_devices):
        shard = rank if world > 1 else k
        # weak scaling: every GPU gets its own n reads (different generator seed per shard)
        jobs = synth.config2_jobs(n, SCORING[2], SCORING[3], READ_LEN, WINDOW_LEN, seed=synth.SEED + 977 * shard)
        g = ip.MultiStreamAligner(dev, SCORING[0], SCORING[1], streams=args.streams)
        g.upload(jobs)                      # inputs resident in HBM before the timed region
        aligners.append(g)
        tables.append(jobs)

    def barrier():
        for g in aligners:
            g.sync()
        if dist is not None:
            dist.barrier()

    def step():
        for g in aligners:
            g.run()                          # asynchronous: all GPUs of this process overlap

    for _ in range(args.warmup):
        step()
    barrier()
    for g in aligners:
        # HIP events on the ctx streams: around the striped DP kernels only (level 2) unless --all-kernel-times
        g.set_profiling(0 if os.environ.get("IPX_BENCH_NO_PROFILE") else (1 if args.all_kernel_times else 2))
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        import torch
        t = torch.tensor([elapsed], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t[0])

    # correctness guard on the timed data: the 100k-read checksum of the reference
    res = aligners[0].download()
    chk = int(res.records["score1"][:100000].astype(np.int64).sum()) if n >= 100000 else None
    mean_cigar = float(res.records["cigar_len"].mean())
    if rank == 0 and n >= 100000 and SCORING == (3, 2, 3, 1):
        assert chk == 43222928, "bench output differs from the reference checksum: %r" % chk

    if rank == 0:
        total = n * n_gpus * args.steps
        value = total / elapsed / 1e6
        kt = aligners[0].kernel_times()
        per_step = {k: v[0] / args.steps for k, v in kt.items() if v[1] > 0}
        dom = max((k for k in per_step if k.startswith("dp_") or k.startswith("traceback")), key=lambda k: per_step[k])
        dom_ms = kt[dom][0] / kt[dom][1]                      # average duration of one launch
        per_launch = n / max(1, min(args.streams, n // 50000))   # alignments one launch of the dominant kernel processes
        alg_bytes = (ALG_BYTES_FIXED + 4.0 * mean_cigar) * per_launch + WINDOW_LEN
        achieved = alg_bytes / (dom_ms * 1e-3) / 1e9
        # HBM bytes per launch of the dominant kernel from the committed rocprofv3 PMC passes
        # (profiles/summarize_pmc.py; FETCH_SIZE x2 on gfx950 + WRITE_SIZE, MI355X_MICROARCH.md HBM section)
        traffic = None
        pmc = os.path.join(ROOT, "profiles", "pmc_latest.json")
        if os.path.exists(pmc):
            try:
                key = dom.replace("dp_word_first", "dp_word_fwd")
                traffic = json.load(open(pmc)).get(key, {}).get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
        # VALU issue, the bound that actually binds this path: wave-instructions of the dominant
        # kernel from the committed rocprofv3 SQ_INSTS_VALU pass, over its live duration, against the issue rate of
        # packed 16-bit VALU measured by tools/ubench_valu.hip (1 wave-instruction per 3.13 cycles per SIMD)
        valu = None
        sq = os.path.join(ROOT, "profiles", "sq_latest.json")
        if os.path.exists(sq):
            try:
                per_read = json.load(open(sq))["kernels"][dom.replace("dp_word_first", "dp_word_fwd")]["valu_insts_per_launch"] / 1e6
                rate = per_read * per_launch / (dom_ms * 1e-3) / 1e9
                ceil_ = 1024 * VALU_CLOCK_GHZ / PK16_CYCLES_PER_INST
                valu = {"achieved": round(rate, 1), "peak": round(ceil_, 1), "unit": "G wave-instr/s", "frac": round(rate / ceil_, 4),
                        "insts_per_alignment": round(per_read, 1),
                        "note": "streams overlap, so per-launch durations are stretched; single-stream figure in DESIGN.md"}
            except Exception:
                valu = None
        out = {
            "metric": "million read-alignments/sec (150 bp x 300 bp, affine gap)",
            "value": round(value, 4), "unit": "million alignments/s", "n_gpus": n_gpus, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "int16 (packed pairs; the 8-bit pass is computed in 16-bit containers)",
            "data": "synthetic",
            "config": {"workload": "config2b: %d synthetic %d bp reads vs one %d bp window per GPU, scoring "
                                   "(match,mismatch,gap_open,gap_ext)=%s, flag=1 (scores+coords+CIGAR), "
                                   "100%% of reads rescored in 16 bit" % (n, READ_LEN, WINDOW_LEN, SCORING),
                       "reads_per_gpu": n, "read_len": READ_LEN, "window_len": WINDOW_LEN,
                       "sharding": "reads split over GPUs, host gather, no collective",
                       "streams_per_gpu": args.streams},
            "roofline": {"bound": "hbm", "achieved": round(achieved, 3), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 6), "traffic": traffic, "kernel": dom,
                         "kernel_ms_per_launch": round(dom_ms, 4),
                         "alg_bytes_per_launch": int(alg_bytes),
                         "note": "integer-VALU-bound DP: HBM fraction is expected << 1 (SURVEY 8d)", "valu_issue": valu},
            "kernel_ms_per_step": {k: round(v, 4) for k, v in sorted(per_step.items())},   # summed over the streams (they overlap)
            "gpu_event_ms_per_step": round(aligners[0].last_run_ms(), 4),
            "mean_cigar_ops": round(mean_cigar, 3),
            "checksum_first_100k": chk,
        }
        if n_gpus == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(tables[0], SCORING)
            out["speedup_vs_cpu_baseline"] = round(value / out["cpu_baseline"]["value"], 2)
        print(json.dumps(out))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    for g in aligners:
        g.close()


if __name__ == "__main__":
    main()
