"""Large randomized parity run on the GPU box: every field of every job against the CPU checker
(the reference's own ssw.c from oracle/_ref when it travelled with the tree, else the port), using
all host cores.  Far more adversarial inputs than the per-case tests can afford."""
import json
import os

import numpy as np
import pytest

from indelpost_amd import batch as R
from indelpost_amd.batch import JobTable

pytestmark = pytest.mark.gpu


def _make_jobs(rng, n_jobs, n_refs, alpha_mix=True, fast_gaps_only=False):
    refs = []
    for k in range(n_refs):
        alpha = 2 if (alpha_mix and k % 4 == 0) else 4
        w = rng.integers(0, alpha, int(rng.integers(6, 620))).astype(np.int8)
        if k % 11 == 0:
            w[rng.integers(0, len(w), max(1, len(w) // 12))] = 4
        refs.append(w)
    reads, rid, go, ge = [], [], [], []
    gaps = [(3, 1), (3, 0), (5, 1), (5, 0), (4, 1), (4, 0), (1, 1), (1, 0), (0, 0), (0, 1), (1, 2), (2, 2), (6, 3),
            (10, 1), (2, 1), (255, 1)]
    if fast_gaps_only:          # gap_open > gap_ext everywhere: the batch takes the kernels that have no stepped lazy-F loop
        gaps = [g for g in gaps if g[0] > g[1]]
    for i in range(n_jobs):
        k = int(rng.integers(0, n_refs))
        w = refs[k]
        L = int(rng.integers(1, 257))
        kind = i % 5
        if kind == 0:
            r = rng.integers(0, 4, L).astype(np.int8)
        else:
            st = int(rng.integers(0, len(w)))
            err = (0.0, 0.02, 0.06, 0.2)[i % 4]
            out, q = [], st
            while len(out) < L:
                if q >= len(w):
                    out.append(int(rng.integers(0, 4))); continue
                u = rng.random()
                if u < err: out.append(int(rng.integers(0, 4))); q += 1
                elif u < err * 1.6: q += int(rng.integers(1, 9))
                elif u < err * 2.2: out.extend(rng.integers(0, 4, int(rng.integers(1, 9))).tolist())
                else: out.append(int(w[q])); q += 1
            r = np.array(out[:L], np.int8)
        if i % 13 == 0:
            r[rng.integers(0, L, max(1, L // 10))] = 4
        g = gaps[int(rng.integers(0, len(gaps)))]
        reads.append(r); rid.append(k)
        if i % 17 == 0:         # indelPost's mutant-contig jobs: gap_open = len(read) (narrowed to uint8 like the reference does)
            go.append(min(max(L, g[1] + 1), 255) if fast_gaps_only else L)
        else:
            go.append(g[0])
        ge.append(g[1])
    return JobTable.from_sequences(reads, refs, rid, go, ge, encoded=True)


# Routing switches (indelpost_amd.batch.ROUTE_*) repeat a scoring with speed-only decisions turned off (no upper-bound
# stage, 8-bit pass before the 16-bit one, LDS-staged profile, ...): same results required.  "fast_gaps": gap_open >
# gap_ext everywhere, so that every job takes the kernels that have no stepped lazy-F loop.
@pytest.mark.parametrize("scoring,knobs", [((3, 2), ()), ((1, 1), ()), ((2, 2), ()), ((1, 3), ()), ((5, 4), ()), ((2, 4), ()),
                                           ((1, 1), (R.ROUTE_NO_BRACKET, R.ROUTE_NO_PERM_PROFILE)),
                                           ((3, 2), (R.ROUTE_NO_WORD_FIRST, R.ROUTE_NO_BRACKET, R.ROUTE_NO_PERM_PROFILE)),
                                           ((3, 2), (R.ROUTE_NO_WORD_FIRST, R.ROUTE_NO_MC_LDS, R.ROUTE_TB_NO_FUSE)),
                                           ((3, 2), ("fast_gaps",)), ((2, 2), ("fast_gaps",)), ((1, 1), ("fast_gaps",)),
                                           ((3, 2), (R.ROUTE_NO_F16,)), ((3, 2), (R.ROUTE_NO_SKEW,)), ((3, 2), (R.ROUTE_NO_VL2, "fast_gaps")), ((3, 2), ("fast_gaps", "vl2")), ((1, 1), ("fast_gaps", "vl2")), ((5, 4), (R.ROUTE_NO_SKEW, "fast_gaps")), ((7, 8), ("fast_gaps",)), ((6, 5), ()), ((8, 1), ("fast_gaps",)),
                                           ((3, 2), (R.ROUTE_NO_PLAIN_FIRST, "fast_gaps")), ((1, 1), (R.ROUTE_NO_PLAIN_FIRST,)), ((3, 2), (R.ROUTE_NO_CLASS_MERGE,)),
                                           ((1, 1), (R.ROUTE_NO_CLASS_MERGE, "fast_gaps")), ((2, 2), (R.ROUTE_NO_VL2, "fast_gaps")),
                                           ((3, 2), (R.ROUTE_NO_EXACT_DIRECT, "fast_gaps")), ((1, 1), (R.ROUTE_NO_EXACT_DIRECT,)), ((2, 2), (R.ROUTE_NO_TIERS,)),
                                           ((3, 2), (R.ROUTE_TB_NO_UNGAPPED,)), ((1, 3), (R.ROUTE_TB_NO_UNGAPPED, "fast_gaps")),
                                           # r04: the anti-diagonal traceback tiers off; small batches (every traceback in a tier: "small" = 6 000 jobs)
                                           ((3, 2), (R.ROUTE_TB_NO_DIAG,)), ((2, 2), (R.ROUTE_TB_NO_DIAG, "fast_gaps")),
                                           ((3, 2), ("small",)), ((1, 1), ("small",)), ((2, 4), ("small", "fast_gaps")), ((3, 2), ("small", R.ROUTE_TB_NO_DIAG)),
                                           # r04: the latency tier of the wavefront passes (32 lanes per read) forced on a whole batch / off for a small one; tiny batches
                                           ((3, 2), (R.ROUTE_FORCE_LAT, "fast_gaps")), ((1, 1), (R.ROUTE_FORCE_LAT, "fast_gaps")), ((2, 2), (R.ROUTE_FORCE_LAT, "fast_gaps")),
                                           ((5, 4), (R.ROUTE_FORCE_LAT, "fast_gaps")), ((3, 2), ("small", "fast_gaps")), ((3, 2), ("small", "fast_gaps", R.ROUTE_NO_LAT)),
                                           ((3, 2), ("tiny", "fast_gaps")), ((1, 1), ("tiny",)), ((3, 2), (R.ROUTE_FORCE_LAT, R.ROUTE_NO_LAT_PROOF, "fast_gaps")),
                                           ((2, 2), ("tiny", "fast_gaps")),
                                           # r04 (second half): the lane-per-job tracebacks one launch per band width again (default: one launch, blocks shared out
                                           # on the device), the latency-bound kernels at the default wave priority
                                           ((3, 2), (R.ROUTE_TB_PER_WIDTH,)), ((1, 3), (R.ROUTE_TB_PER_WIDTH, R.ROUTE_NO_SETPRIO, "fast_gaps")), ((2, 2), (R.ROUTE_NO_SETPRIO,)),
                                           # r04: the 16-bit reverse pass as a band (k_dp_band_rev) whatever the size of the class -- in a batch of 30 000 no class has the
                                           # 6 000 tiles that switch it on by itself; and switched off
                                           ((3, 2), (R.ROUTE_FORCE_BAND_REV,)), ((3, 2), (R.ROUTE_FORCE_BAND_REV, "fast_gaps")), ((5, 4), (R.ROUTE_FORCE_BAND_REV, "fast_gaps")),
                                           ((7, 8), (R.ROUTE_FORCE_BAND_REV,)), ((2, 4), (R.ROUTE_FORCE_BAND_REV, "fast_gaps")), ((3, 2), (R.ROUTE_NO_BAND_REV, "fast_gaps")),
                                           # ... and of the plain recurrence in the 8-bit dialect
                                           ((1, 1), (R.ROUTE_FORCE_BAND_REV, "fast_gaps")), ((2, 2), (R.ROUTE_FORCE_BAND_REV,)), ((1, 3), (R.ROUTE_FORCE_BAND_REV, "fast_gaps"))])
def test_gpu_stress_vs_cpu_checker(gpu, oracle_mod, scoring, knobs, capfd):
    from oracle.oracle import cpu_batch_results, fnv1a_ops
    fast = "fast_gaps" in knobs
    routing = sum(k for k in knobs if isinstance(k, int))
    rng = np.random.default_rng(1000 + 7 * scoring[0] + scoring[1] + 100 * len(knobs) + (5000 if fast else 0) + (77 if "small" in knobs else 0) + (99 if "tiny" in knobs else 0)
                                + 100000 * int(os.environ.get("IPX_STRESS_SEED", "0")))
    n = 6000 if "small" in knobs else 900 if "tiny" in knobs else int(os.environ.get("IPX_STRESS_JOBS", "30000"))
    jobs = _make_jobs(rng, n, 97, fast_gaps_only=fast)
    be = oracle_mod.Backend("reference" if oracle_mod.have_reference() else "port")
    mat = oracle_mod.dna_matrix(*scoring)
    cores = min(16, len(os.sched_getaffinity(0)))
    exp = cpu_batch_results(be, jobs, mat, cores)
    # The one class of jobs whose (flag, CIGAR) the reference leaves undefined, BY RULE: its traceback reads a direction
    # cell that no band iteration has written (banded_sw mallocs `direction`, ssw.c:610, 679-719).  The restatement keeps
    # that buffer zeroed and written codes are 1..5, so it takes the reference's own error exit (flag 1, no CIGAR)
    # exactly when such a cell is read: U = {restatement reports flag 1}.
    port = cpu_batch_results(oracle_mod.Backend("port"), jobs, mat, cores)
    capfd.readouterr()
    undefined = port["flag"] == 1
    gpu.set_scoring(*scoring)
    gpu.set_routing(routing)
    try:
        res = gpu.align(jobs)
    finally:
        gpu.set_routing(0)
    rec = res.records
    assert (exp["is_null"] == 0).all()
    hashes = np.array([fnv1a_ops(res.cigar_ops(i)) if rec["cigar_len"][i] else 2166136261 for i in range(n)], np.uint32)
    # the GPU follows the rule exactly: flag 1 and no CIGAR on U, nowhere else
    assert np.array_equal(rec["flag"] == 1, undefined), "GPU traceback failures differ from the rule: %s" % np.flatnonzero((rec["flag"] == 1) != undefined)[:5]
    assert (rec["cigar_len"][undefined] == 0).all()
    # the reference can only fail where the rule says it reads an unwritten cell
    ref_fail = exp["flag"] == 1
    assert not (ref_fail & ~undefined).any()
    # DP fields: every job, no exception
    bad = np.zeros(n, bool)
    for f in ("score1", "score2", "ref_begin1", "ref_end1", "read_begin1", "read_end1", "ref_end2"):
        bad |= rec[f] != exp[f]
    # flag and CIGAR: every job outside U
    d = ~undefined
    bad |= d & ((rec["flag"] != exp["flag"]) | (rec["cigar_len"] != exp["cigar_len"]) | (hashes != exp["cigar_hash"]))
    counts = {"scoring": list(scoring), "routing": routing, "fast_gaps_only": fast, "jobs": n, "undefined_by_rule": int(undefined.sum()),
              "reference_reports_flag1": int((ref_fail & undefined).sum()),
              "reference_returns_cigar_from_unwritten_cells": int((undefined & ~ref_fail).sum()),
              "gpu_reports_flag1": int((rec["flag"] == 1).sum()), "flag2_jobs": int((exp["flag"] == 2).sum()),
              "differences_outside_rule": int(bad.sum())}
    print("stress counts:", counts)
    out_dir = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    if os.path.isdir(out_dir):
        with open(os.path.join(out_dir, "stress_undefined_traceback.jsonl"), "a") as f:
            f.write(json.dumps(counts) + "\n")
    assert int(undefined.sum()) <= max(8, n // 200), "implausibly many undefined tracebacks: %d" % int(undefined.sum())
    first = np.flatnonzero(bad)
    assert len(first) == 0, "scoring %s: %d jobs differ, first %d: gpu %s cpu %s" % (
        scoring, len(first), first[0], rec[first[0]], exp[first[0]])


def _windows_reads(n_windows, reads_per_window, rls, wl_lo, wl_hi):
    """SURVEY 8d generator (xorshift64 windows + error-model reads) for several windows / read lengths."""
    from indelpost_amd import _lib, synth
    L = _lib.lib()
    st = synth.SEED
    rng = np.random.default_rng(5)
    refs, reads, rid = [], [], []
    for w in range(n_windows):
        wl = int(rng.integers(wl_lo, wl_hi + 1))
        ref = np.zeros(wl, np.int8)
        st = L.ipx_synth_window(st, ref.ctypes.data, wl)
        refs.append(ref)
        per = max(1, reads_per_window // len(rls))
        for rl in rls:
            rl = min(rl, wl)
            buf = np.zeros(per * rl, np.int8)
            st = L.ipx_synth_reads(st, ref.ctypes.data, wl, buf.ctypes.data, per, rl)
            reads.extend(buf.reshape(per, rl))
            rid.extend([w] * per)
    return reads, refs, rid


def _check_all(gpu, oracle_mod, jobs, scoring, capfd):
    from oracle.oracle import cpu_batch_results, fnv1a_ops
    be = oracle_mod.Backend("reference" if oracle_mod.have_reference() else "port")
    exp = cpu_batch_results(be, jobs, oracle_mod.dna_matrix(*scoring), len(os.sched_getaffinity(0)))
    capfd.readouterr()
    gpu.set_scoring(*scoring)
    res = gpu.align(jobs)
    rec = res.records
    for f in ("score1", "score2", "ref_begin1", "ref_end1", "read_begin1", "read_end1", "ref_end2", "cigar_len", "flag"):
        assert (rec[f] == exp[f]).all(), f
    n = jobs.n_jobs
    hashes = np.array([fnv1a_ops(res.cigar_ops(i)) if rec["cigar_len"][i] else 2166136261 for i in range(n)], np.uint32)
    assert (hashes == exp["cigar_hash"]).all()
    return res


def test_gpu_config4_shape_mixed_lengths_and_windows(gpu, oracle_mod, capfd):
    """BASELINE configs[3] in miniature (1 GPU): read lengths {75,100,125,150,200,250}, windows of 200-600 bp,
    one window per ~1000 reads, defaults (3,2,3,1): short reads stay in the 8-bit pass, long ones are rescored."""
    reads, refs, rid = _windows_reads(40, 996, [75, 100, 125, 150, 200, 250], 200, 600)
    jobs = JobTable.from_sequences(reads, refs, rid, 3, 1, encoded=True)
    res = _check_all(gpu, oracle_mod, jobs, (3, 2), capfd)
    assert set(res.records["mode"].tolist()) == {0, 1}


def test_gpu_config5_shape_penalty_grid_per_read_windows(gpu, oracle_mod, capfd):
    """BASELINE configs[4] at the job-list level: per-read 300 bp windows x the gap-penalty grid of
    varaln.pyx:1127-1143 (retarget / grid_search, pileup.pyx:639-648), CIGAR bit-exact."""
    reads, refs, rid = _windows_reads(6000, 1, [150], 300, 300)
    grid = [(3, 1), (3, 0), (5, 1), (5, 0), (4, 1), (4, 0)]
    R, I, GO, GE = [], [], [], []
    for r, w in zip(reads, rid):
        for go, ge in grid:
            R.append(r); I.append(w); GO.append(go); GE.append(ge)
    jobs = JobTable.from_sequences(R, refs, I, GO, GE, encoded=True)
    _check_all(gpu, oracle_mod, jobs, (3, 2), capfd)


def test_gpu_long_reads_and_long_windows(gpu, oracle_mod, capfd):
    """Reads up to 512 bp (long-read instantiation, segLen 33..64) and windows up to 4096 bp."""
    rng = np.random.default_rng(4242)
    refs = [rng.integers(0, 4, n).astype(np.int8) for n in (700, 1500, 4096)]
    reads, rid, go, ge = [], [], [], []
    for i in range(600):
        k = i % 3
        w = refs[k]
        L = int(rng.integers(257, 513))
        st = int(rng.integers(0, len(w) - 100))
        r = np.resize(w[st:], L).copy()
        m = rng.random(L) < (0.0, 0.03, 0.1)[i % 3]
        r[m] = rng.integers(0, 4, int(m.sum()))
        if i % 5 == 0:
            cut = int(rng.integers(20, L - 20))
            r = np.concatenate([r[:cut], r[cut + int(rng.integers(1, 30)):]])
        reads.append(r); rid.append(k)
        g = [(3, 1), (5, 0), (1, 1), (4, 1), (2, 2)][i % 5]
        go.append(g[0]); ge.append(g[1])
    jobs = JobTable.from_sequences(reads, refs, rid, go, ge, encoded=True)
    for scoring in ((3, 2), (1, 1)):
        _check_all(gpu, oracle_mod, jobs, scoring, capfd)


def test_gpu_reads_of_kilobases_vs_windows_of_tens_of_kilobases(gpu, oracle_mod, capfd):
    """r03: reads up to 4 096 bp (the limit was 512) take k_dp_long -- the reference's loops transcribed, striped columns in global
    memory -- and the wave-per-job traceback with its band rows in global memory; here 513-4 000 bp reads against 3-20 kb windows,
    indels of up to 40 bp, fast and slow gap penalties, two scorings, ordinary short reads in the same batch, every field and every
    CIGAR against the compiled reference"""
    rng = np.random.default_rng(2000)
    refs = [rng.integers(0, 4, n).astype(np.int8) for n in (3000, 8000, 20000)]
    reads, rid, go, ge = [], [], [], []
    for i in range(48):
        k = i % 3
        w = refs[k]
        L = int(rng.choice([513, 640, 1000, 1025, 2000, 3000, 4000])) if i % 4 else int(rng.integers(50, 512))
        L = min(L, len(w) - 60)
        st = int(rng.integers(0, len(w) - L - 50))
        r = w[st:st + L].copy()
        m = rng.random(L) < (0.0, 0.01, 0.05)[i % 3]
        r[m] = rng.integers(0, 4, int(m.sum()))
        if i % 5 == 0:
            cut = int(rng.integers(20, L - 20))
            r = np.concatenate([r[:cut], r[cut + int(rng.integers(1, 40)):]])
        if i % 5 == 1:
            cut = int(rng.integers(20, L - 20))
            r = np.concatenate([r[:cut], rng.integers(0, 4, int(rng.integers(1, 30))).astype(np.int8), r[cut:]])[:4096]
        if i % 11 == 7:
            r = rng.integers(0, 4, len(r)).astype(np.int8)                # noise: a long read that stays in the 8-bit pass
        reads.append(r); rid.append(k)
        g = [(3, 1), (5, 0), (1, 1), (4, 1), (2, 2), (3, 0)][i % 6]
        go.append(g[0]); ge.append(g[1])
    jobs = JobTable.from_sequences(reads, refs, rid, go, ge, encoded=True)
    for scoring in ((3, 2), (1, 1)):
        res = _check_all(gpu, oracle_mod, jobs, scoring, capfd)
    assert int(np.diff(jobs.read_off).max()) >= 4000 and (res.records["read_end1"] > 2000).any()


def test_gpu_kilobase_reads_one_wavefront_per_read(gpu, oracle_mod, capfd):
    """r04, k_dp_wide: the 16-bit passes of reads from 505 bp under gap_open > gap_ext as ONE wavefront per read (64 lanes x 16 / 32 / 48 / 64
    rows, 32-bit cells).  Every bucket, 505-4 096 bp against windows of 70-12 000 bp, insertions, deletions, N, noise, three scorings, six
    gap-penalty pairs -- every field and CIGAR against the compiled reference; then the same table through the transcribed loops
    (ROUTE_NO_WIDE): equal digests, and the wavefront form at least 20 times faster in the 16-bit passes (r03 verdict, task 8)."""
    from indelpost_amd.batch import ROUTE_NO_WIDE
    rng = np.random.default_rng(50505)
    refs = [rng.integers(0, 4, n).astype(np.int8) for n in (900, 70, 5000, 12000)]
    reads, rid, go, ge = [], [], [], []
    lens = [505, 511, 512, 520, 1000, 1024, 1025, 1030, 1700, 2047, 2048, 2049, 2100, 3000, 3072, 3073, 3080, 4000, 4095, 4096]
    for i in range(80):
        ln = lens[i % len(lens)]
        k = (0, 2, 3, 2, 1, 3)[i % 6]
        w = refs[k]
        src = np.resize(w[int(rng.integers(0, max(1, len(w) - 60))):], ln).copy() if i % 13 != 12 else rng.integers(0, 4, ln).astype(np.int8)
        m = rng.random(ln) < (0.0, 0.01, 0.04)[i % 3]
        src[m] = rng.integers(0, 5, int(m.sum()))
        if i % 4 == 1:
            cut = int(rng.integers(20, ln - 60))
            src = np.concatenate([src[:cut], src[cut + int(rng.integers(1, 40)):]])
        if i % 4 == 2:
            cut = int(rng.integers(20, ln - 20))
            src = np.concatenate([src[:cut], rng.integers(0, 4, int(rng.integers(1, 30))).astype(np.int8), src[cut:]])[:4096]
        reads.append(src); rid.append(k)
        g = [(3, 1), (5, 0), (4, 1), (6, 2), (2, 1), (9, 3)][i % 6]
        go.append(g[0]); ge.append(g[1])
    jobs = JobTable.from_sequences(reads, refs, rid, go, ge, encoded=True)
    for scoring in ((3, 2), (1, 3), (7, 4)):
        gpu.set_routing(0)
        res = _check_all(gpu, oracle_mod, jobs, scoring, capfd)
    gpu.set_profiling(True)
    res = gpu.align(jobs)
    wide = {k: v for k, v in gpu.kernel_times().items() if k.endswith("_kb_wavefront")}
    assert {"dp_word_first_kb_wavefront", "dp_word_rev_kb_wavefront"} <= set(wide)
    assert not [k for k in gpu.kernel_times() if k.startswith("dp_word") and k.endswith("_kb_loops")]      # (no job with gap_open <= gap_ext here)
    gpu.set_profiling(False); gpu.set_profiling(True)
    gpu.set_routing(ROUTE_NO_WIDE)
    res2 = gpu.align(jobs)
    loops = {k: v for k, v in gpu.kernel_times().items() if k.startswith("dp_word") and k.endswith("_kb_loops")}
    gpu.set_profiling(False)
    gpu.set_routing(0)
    assert res2.digest() == res.digest()
    t_wide, t_loops = sum(v[0] for v in wide.values()), sum(v[0] for v in loops.values())
    print("16-bit passes of 80 reads of 505-4096 bp: wavefront per read %.2f ms, transcribed loops %.2f ms" % (t_wide, t_loops))
    assert t_loops > 20 * t_wide


def test_gpu_windows_of_tens_of_kilobases(gpu, oracle_mod, capfd):
    """windows up to 32 000 bp (r03; the limit was 4 096): short reads against 20 kb windows, among them reads that bridge a deletion of
    more than a kilobase under gap extension 0 -- their traceback band grows past what LDS holds and lives in the global scratch"""
    rng = np.random.default_rng(777)
    refs = [rng.integers(0, 4, n).astype(np.int8) for n in (20000, 31999, 5000)]
    reads, rid, go, ge = [], [], [], []
    for i in range(240):
        k = i % 3
        w = refs[k]
        L = int(rng.integers(60, 400))
        st = int(rng.integers(0, len(w) - 3000))
        if i % 6 == 0:                                            # two halves 1.1-2.5 kb apart: one alignment with a long deletion when gap_ext = 0
            gap = int(rng.integers(1100, 2500))
            r = np.concatenate([w[st:st + L // 2], w[st + L // 2 + gap:st + gap + L]]).copy()
        else:
            r = w[st:st + L].copy()
        m = rng.random(len(r)) < 0.02
        r[m] = rng.integers(0, 4, int(m.sum()))
        reads.append(r); rid.append(k)
        g = [(3, 0), (5, 0), (3, 1), (4, 0)][i % 4]
        go.append(g[0]); ge.append(g[1])
    jobs = JobTable.from_sequences(reads, refs, rid, go, ge, encoded=True)
    res = _check_all(gpu, oracle_mod, jobs, (3, 2), capfd)
    span = res.records["ref_end1"] - res.records["ref_begin1"]
    assert (span > 1100).sum() >= 20                              # the long deletions were bridged


def test_gpu_limits_are_refused_loudly(gpu):
    import indelpost_amd as ip
    too_long_read = JobTable.from_sequences([np.zeros(4097, np.int8)], [np.zeros(100, np.int8)], [0], 3, 1, encoded=True)
    with pytest.raises(ip.IpxError):
        gpu.align(too_long_read)
    too_long_ref = JobTable.from_sequences([np.zeros(50, np.int8)], [np.zeros(32001, np.int8)], [0], 3, 1, encoded=True)
    with pytest.raises(ip.IpxError):
        gpu.align(too_long_ref)
