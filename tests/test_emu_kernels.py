"""Product kernel source + launch sequence under the lock-step wave emulator vs the oracle (CPU).

These tests exercise indelpost_amd/csrc/ipx_kernels.h and ipx_pipeline.h (compiled with
-DIPX_CPU_EMU) -- the DP passes, planner, traceback and the host-side job-table code -- without a
GPU.  The GPU parity tests (test_gpu_parity.py) run the same cases through the real C ABI.
"""
import numpy as np
import pytest

from indelpost_amd import batch as R
from indelpost_amd.batch import JobTable
from tests.conftest import codes


def _compare(res, jobs_py, port, mat):
    for i, (rd, rf, go, ge) in enumerate(jobs_py):
        exp = port.align(rd, rf, mat, go, ge)
        assert res.as_dict(i) == exp, "job %d (read %d bp, window %d bp)" % (i, len(rd), len(rf))


def test_emu_golden_subset(emu, oracle_mod, port, golden_c):
    """Golden vectors from the reference, grouped by scoring so each group is one batch."""
    groups = {}
    for k, c in enumerate(golden_c[:160]):
        groups.setdefault((c["match"], c["mismatch"]), []).append(c)
    for (ms, mm), cs in groups.items():
        a = emu(0, ms, mm)
        jobs = JobTable.from_sequences([codes(c["read"]) for c in cs], [codes(c["ref"]) for c in cs],
                                       np.arange(len(cs), dtype=np.int32), [c["gap_open"] for c in cs],
                                       [c["gap_ext"] for c in cs], encoded=True)
        res = a.align(jobs)
        assert a.status == 0
        for i, c in enumerate(cs):
            assert res.as_dict(i) == c["expect"], "scoring %s case %d" % ((ms, mm), i)


def test_emu_config1_like(emu, oracle_mod, port, hip_lib):
    """BASELINE config 1 shape (150 bp reads, one 300 bp window, defaults), reduced to 48 reads."""
    from indelpost_amd import synth
    jobs = synth.config2_jobs(48)
    a = emu(0, 3, 2)
    res = a.align(jobs)
    mat = oracle_mod.dna_matrix(3, 2)
    ref = jobs.refs
    _compare(res, [(jobs.reads[i * 150:(i + 1) * 150], ref, 3, 1) for i in range(48)], port, mat)
    assert (res.records["mode"] == 1).all()        # every 150 bp read overflows the 8-bit pass


def test_emu_flags_and_score_size(emu, oracle_mod, port):
    rng = np.random.default_rng(3)
    w = rng.integers(0, 4, 120).astype(np.int8)
    reads = [w[10:70].copy(), w[30:110].copy(), rng.integers(0, 4, 50).astype(np.int8)]
    reads[0][5] ^= 1
    mat = oracle_mod.dna_matrix(2, 2)
    jobs = JobTable.from_sequences(reads, [w], [0, 0, 0], 3, 1, encoded=True)
    for flag, filters, filterd, ss in [(0, 0, 0, 2), (1, 0, 0, 2), (2, 100, 0, 2), (2, 500, 0, 2), (4, 0, 65, 2),
                                       (8, 0, 0, 2), (1, 0, 0, 1), (1, 0, 0, 0)]:
        a = emu(0, 2, 2)
        a.set_scoring(matrix=mat, flag=flag, filters=filters, filterd=filterd, score_size=ss)
        res = a.align(jobs)
        for i, r in enumerate(reads):
            exp = port.align(r, w, mat, 3, 1, flag=flag, filters=filters, filterd=filterd, score_size=ss)
            if exp is None:
                assert res.records[i]["mode"] == 2      # reference returns NULL (ssw.c:848-851)
            else:
                assert res.as_dict(i) == exp, (flag, filters, filterd, ss, i)


def test_emu_band_escalation_and_long_cigar(emu, oracle_mod, port):
    """Reads whose traceback band outgrows the tier-0 scratch (large indels) and zero gap penalties."""
    rng = np.random.default_rng(11)
    w = rng.integers(0, 4, 400).astype(np.int8)
    r1 = np.concatenate([w[20:80], w[140:200]])             # 60 bp deletion
    r2 = np.concatenate([w[50:90], rng.integers(0, 4, 40).astype(np.int8), w[90:130]])   # 40 bp insertion
    r3 = w[100:220].copy()
    mat = oracle_mod.dna_matrix(3, 2)
    reads = [r1, r2, r3, r1, r2]
    go = [1, 1, 0, 3, 0]
    ge = [0, 0, 0, 1, 0]
    jobs = JobTable.from_sequences(reads, [w], [0] * 5, go, ge, encoded=True)
    a = emu(0, 3, 2)
    res = a.align(jobs)
    assert a.status == 0
    _compare(res, [(reads[i], w, go[i], ge[i]) for i in range(5)], port, mat)


def test_emu_degenerate(emu, oracle_mod, port):
    ref = oracle_mod.encode("ACGTACGTTTGACCAGT")
    reads = [np.zeros(0, np.int8), np.full(4, 4, np.int8), oracle_mod.encode("A"), oracle_mod.encode("GGGGGGGG"),
             oracle_mod.encode("ACGTACGTCCCTTGACCAGT")]
    jobs = JobTable.from_sequences(reads, [ref, np.zeros(0, np.int8)], [0, 0, 0, 0, 1], 3, 1, encoded=True)
    a = emu(0, 3, 2)
    res = a.align(jobs)
    mat = oracle_mod.dna_matrix(3, 2)
    for i in range(5):
        exp = port.align(reads[i], ref if i < 4 else np.zeros(0, np.int8), mat, 3, 1)
        assert res.as_dict(i) == exp, i


def test_emu_mixed_lengths_many_windows(emu, oracle_mod, port):
    """Config-4 shape in miniature: read lengths 75..250, windows 200..600, one tile mixes windows."""
    rng = np.random.default_rng(5)
    refs = [rng.integers(0, 4, int(n)).astype(np.int8) for n in (200, 333, 600)]
    reads, rid = [], []
    for L in (75, 100, 125, 150, 200, 250) * 3:
        k = int(rng.integers(0, 3))
        st = int(rng.integers(0, max(1, len(refs[k]) - L)))
        r = np.resize(refs[k][st:], L).copy()
        m = rng.random(L) < 0.03
        r[m] = rng.integers(0, 4, int(m.sum()))
        reads.append(r)
        rid.append(k)
    jobs = JobTable.from_sequences(reads, refs, rid, 3, 1, encoded=True)
    a = emu(0, 3, 2)
    res = a.align(jobs)
    mat = oracle_mod.dna_matrix(3, 2)
    _compare(res, [(reads[i], refs[rid[i]], 3, 1) for i in range(len(reads))], port, mat)
    modes = set(res.records["mode"].tolist())
    assert modes == {0, 1}      # short reads stay in the 8-bit pass, long ones take the 16-bit pass


def test_emu_long_reads_generic_kernel(emu, oracle_mod, port):
    """Reads of 260..500 bp need segLen 33..63: the branch-guarded long-read instantiation."""
    rng = np.random.default_rng(8)
    w = rng.integers(0, 4, 700).astype(np.int8)
    reads = []
    for L in (260, 333, 400, 500):
        st = int(rng.integers(0, 150))
        r = w[st:st + L].copy()
        m = rng.random(L) < 0.04
        r[m] = rng.integers(0, 4, int(m.sum()))
        reads.append(np.concatenate([r[:L // 2], r[L // 2 + 3:], rng.integers(0, 4, 3).astype(np.int8)]))
    jobs = JobTable.from_sequences(reads, [w], [0] * len(reads), [3, 5, 3, 4], [1, 0, 1, 1], encoded=True)
    for scoring in ((3, 2), (1, 1)):
        a = emu(0, *scoring)
        res = a.align(jobs)
        assert a.status == 0
        mat = oracle_mod.dna_matrix(*scoring)
        for i, r in enumerate(reads):
            assert res.as_dict(i) == port.align(r, w, mat, int(jobs.gap_open[i]), int(jobs.gap_ext[i])), (scoring, i)


def test_emu_explicit_mask_len(emu, oracle_mod, port):
    rng = np.random.default_rng(12)
    w = rng.integers(0, 4, 260).astype(np.int8)
    w[130:260] = w[0:130]
    reads = [w[20:100].copy() for _ in range(4)]
    masks = [15, 40, 5, 14]
    jobs = JobTable.from_sequences(reads, [w], [0] * 4, 3, 1, encoded=True)
    jobs.mask_len = np.asarray(masks, np.int32)
    res = emu(0, 2, 2).align(jobs)
    mat = oracle_mod.dna_matrix(2, 2)
    for i in range(4):
        assert res.as_dict(i) == port.align(reads[i], w, mat, 3, 1, mask_len=masks[i]), i


@pytest.mark.parametrize("knobs", [(R.ROUTE_NO_BRACKET,), (R.ROUTE_NO_PERM_PROFILE, R.ROUTE_TB_NO_FUSE, R.ROUTE_TB_NO_WAVE_PER_JOB), (R.ROUTE_NO_MC_LDS, R.ROUTE_TB_NO_FUSE),
                                   (R.ROUTE_NO_WORD_FIRST, R.ROUTE_NO_BRACKET, R.ROUTE_NO_PERM_PROFILE), (R.ROUTE_NO_PLAIN_FIRST,),
                                   (R.ROUTE_NO_CLASS_MERGE,), (R.ROUTE_NO_PLAIN_FIRST, R.ROUTE_NO_CLASS_MERGE, R.ROUTE_NO_VL2), (R.ROUTE_NO_EXACT_DIRECT,),
                                   (R.ROUTE_TB_NO_WAVE_PER_JOB,), (R.ROUTE_TB_NO_WAVE_PER_JOB, R.ROUTE_TB_NO_FUSE, R.ROUTE_NO_MC_LDS), (R.ROUTE_TB_NO_UNGAPPED,),
                                   (R.ROUTE_TB_NO_UNGAPPED, R.ROUTE_TB_NO_WAVE_PER_JOB)])
def test_emu_routing_knobs_off(emu, golden_c, knobs):
    """The speed-only routing decisions (the upper-bound stage, 16-bit pass first, register-selector profile, column
    maxima in LDS, fused traceback launch) must not change any result: golden vectors with each turned off."""
    groups = {}
    for c in golden_c[160:320]:
        groups.setdefault((c["match"], c["mismatch"]), []).append(c)
    for (ms, mm), cs in groups.items():
        a = emu(0, ms, mm)
        a.set_routing(sum(knobs))
        jobs = JobTable.from_sequences([codes(c["read"]) for c in cs], [codes(c["ref"]) for c in cs],
                                       np.arange(len(cs), dtype=np.int32), [c["gap_open"] for c in cs],
                                       [c["gap_ext"] for c in cs], encoded=True)
        res = a.align(jobs)
        assert a.status == 0
        for i, c in enumerate(cs):
            assert res.as_dict(i) == c["expect"], "knobs %s scoring %s case %d" % (knobs, (ms, mm), i)


# kernel classes of ipx_pipeline.h (timing keys = class * 256 + job class)
K_BYTE_LOW, K_BYTE_CHECK, K_BYTE_HIGH, K_BYTE_EXACT, K_WORD_FIRST, K_WORD_FWD, K_BYTE_REV, K_WORD_REV = range(2, 10)
SLOW_BASE = 65
K_PACK = 12
K_BYTE_PLAIN, K_BYTE_LOW2, K_BYTE_REV_PLAIN, K_PROVE_PLAIN = 14, 15, 16, 17


def _launched(a, kclass):
    return sorted(k % 256 for k in a.launches if k // 256 == kclass)


def test_emu_bracket_certifies_or_steps(emu, oracle_mod, port):
    """8-bit-resident reads whose scores run far above 128 (75 bp at match 3): the lower-bound stage cannot settle them,
    the upper-bound stage certifies most, the stepped pass takes the rest -- and with the bracket switched off the
    stepped pass takes all of them.  Same records either way, equal to the oracle."""
    rng = np.random.default_rng(21)
    w = rng.integers(0, 4, 260).astype(np.int8)
    lowc = np.resize(np.array([0, 1, 0, 0, 1, 1], np.int8), 200)          # low complexity: many equal-scoring paths
    reads, refs_id = [], []
    for i in range(20):
        st = int(rng.integers(0, 170))
        r = w[st:st + 75].copy()
        if i % 3 == 0:
            r[int(rng.integers(5, 70))] ^= 2
        if i % 4 == 0:
            r = np.concatenate([r[:40], r[43:], w[st + 75:st + 78]])      # 3 bp deletion
        reads.append(r); refs_id.append(0)
    for i in range(8):
        st = int(rng.integers(0, 100))
        r = lowc[st:st + 70 + i].copy()
        r[int(rng.integers(0, len(r)))] ^= 1
        reads.append(r); refs_id.append(1)
    jobs = JobTable.from_sequences(reads, [w, lowc], refs_id, [3, 5, 4, 3] * 7, [1, 0, 1, 0] * 7, encoded=True)
    mat = oracle_mod.dna_matrix(3, 2)
    a = emu(0, 3, 2)
    res = a.align(jobs)
    assert a.status == 0
    for i, r in enumerate(reads):
        assert res.as_dict(i) == port.align(r, [w, lowc][refs_id[i]], mat, int(jobs.gap_open[i]), int(jobs.gap_ext[i])), i
    assert (res.records["mode"] == 0).all()                               # 8-bit semantics throughout
    # default (r03): the plain recurrence first (a wavefront at 2 x 5 segments, marker 2), then the proof kernel; only what the proof
    # leaves open takes the lower-bound stage, and only what that cannot settle is stepped.  The reverse pass likewise.
    assert _launched(a, K_BYTE_PLAIN) == [5] and _launched(a, K_BYTE_LOW) == [] and _launched(a, K_BYTE_HIGH) == [] and _launched(a, K_WORD_FIRST) == []
    assert a.launches.get(a.key(K_PROVE_PLAIN, 0)) == 1 and a.launches.get(a.key(K_PROVE_PLAIN, 1)) == 1
    n_first, n_low2, n_exact, n_rev, n_rev_plain = a.pass_jobs[1], a.pass_jobs[8], a.pass_jobs[4], a.pass_jobs[6], a.pass_jobs[9]
    # what the proof leaves open goes to the stepped pass at once (it steps only where a cut can happen); equal outputs keep the plain reverse pass
    assert n_first == 28 and n_low2 == 0 and 0 < n_exact < 14 and n_rev_plain >= 28 - n_exact and n_rev + n_rev_plain >= 28 and n_rev < 14
    # ... or, first half of r03, through the lower-bound stage first: same records
    q = emu(0, 3, 2)
    q.set_routing(R.ROUTE_NO_EXACT_DIRECT)
    res_q = q.align(jobs)
    assert q.status == 0 and all(res_q.as_dict(i) == res.as_dict(i) for i in range(jobs.n_jobs))
    assert q.pass_jobs[8] == n_exact and q.pass_jobs[4] <= q.pass_jobs[8]
    # the r02 order (lower bound, upper bound, stepped): same records
    o = emu(0, 3, 2)
    o.set_routing(R.ROUTE_NO_PLAIN_FIRST)
    res_o = o.align(jobs)
    assert o.status == 0 and all(res_o.as_dict(i) == res.as_dict(i) for i in range(jobs.n_jobs))
    assert _launched(o, K_BYTE_LOW) == [5] and _launched(o, K_BYTE_HIGH) == [5] and _launched(o, K_BYTE_PLAIN) == [] and _launched(o, K_WORD_FIRST) == []
    # both stages in halves: the lower bound column by column (marker 3), the upper bound as a wavefront at segLen 2 x 5 (marker 2)
    # (marker 4: the lower bound with two reference lanes per GPU lane, 16 reads per wave; marker 2: the upper bound as a wavefront at 2 x 5 segments)
    assert {k % 256 for k in o.launches if k // 256 == K_PACK and k % 256 >= 90} == {10 + 40 * 4 + 5, 10 + 40 * 2 + 10}
    # the same batch in the reference's 16-lane layout (8 reads per wave): same records
    c = emu(0, 3, 2)
    c.set_routing(R.ROUTE_NO_VL2 | R.ROUTE_NO_PLAIN_FIRST)
    res_c = c.align(jobs)
    assert {k % 256 for k in c.launches if k // 256 == K_PACK and k % 256 >= 90} == {10 + 40 * 3 + 5, 10 + 40 * 2 + 10}
    assert all(res_c.as_dict(i) == res.as_dict(i) for i in range(jobs.n_jobs))
    n_low, n_high, n_exact = o.pass_jobs[1], o.pass_jobs[3], o.pass_jobs[4]
    assert n_low == 28 and n_high >= 20 and n_exact < n_high          # most of what reaches the upper-bound stage is certified
    b = emu(0, 3, 2)
    b.set_routing(R.ROUTE_NO_BRACKET)
    res2 = b.align(jobs)
    assert _launched(b, K_BYTE_HIGH) == [] and _launched(b, K_BYTE_PLAIN) == [] and b.pass_jobs[3] == 0 and b.pass_jobs[4] == n_high
    for f in res.records.dtype.names:
        if f != "cigar_off":
            assert np.array_equal(res.records[f], res2.records[f]), f
    assert res.cigar_strings() == res2.cigar_strings()


def test_emu_slow_gap_jobs_are_a_class_of_their_own(emu, oracle_mod, port):
    """gap_open <= gap_ext (e.g. is_perfect_match's gap_open = gap_ext = len(read), varaln.pyx:1230) needs the stepped
    lazy-F loop: such jobs form their own planner classes, the other jobs of the batch keep the kernels without it."""
    rng = np.random.default_rng(23)
    w = rng.integers(0, 4, 300).astype(np.int8)
    reads, go, ge = [], [], []
    for i in range(24):
        L = (60, 150)[i % 2]
        st = int(rng.integers(0, 300 - L))
        r = w[st:st + L].copy()
        r[int(rng.integers(0, L))] ^= 3
        if i % 5 == 0:
            r = np.concatenate([r[:20], r[24:]])
        reads.append(r)
        g = [(3, 1), (L, L), (5, 0), (1, 2), (2, 2), (4, 1)][i % 6]       # (L, L) narrows to uint8 like the reference's arguments
        go.append(g[0]); ge.append(g[1])
    jobs = JobTable.from_sequences(reads, [w], [0] * len(reads), go, ge, encoded=True)
    mat = oracle_mod.dna_matrix(3, 2)
    a = emu(0, 3, 2)
    res = a.align(jobs)
    assert a.status == 0
    for i, r in enumerate(reads):
        assert res.as_dict(i) == port.align(r, w, mat, go[i], ge[i]), i
    wf = _launched(a, K_WORD_FIRST)
    assert any(c < SLOW_BASE for c in wf) and any(c >= SLOW_BASE for c in wf)     # 150 bp reads: fast and slow classes side by side
    # 60 bp reads: the fast-gap ones take the plain recurrence first, the slow-gap ones the stepped pass at once (r03) ...
    assert _launched(a, K_BYTE_PLAIN) and all(c < SLOW_BASE for c in _launched(a, K_BYTE_PLAIN))
    assert any(c >= SLOW_BASE for c in _launched(a, K_BYTE_EXACT)) and _launched(a, K_BYTE_LOW) == []
    # ... and in the r02 order both take the lower-bound stage, in classes of their own
    o = emu(0, 3, 2)
    o.set_routing(R.ROUTE_NO_PLAIN_FIRST)
    res_o = o.align(jobs)
    assert o.status == 0 and all(res_o.as_dict(i) == res.as_dict(i) for i in range(len(reads)))
    low = _launched(o, K_BYTE_LOW)
    assert any(c < SLOW_BASE for c in low) and any(c >= SLOW_BASE for c in low)


def test_emu_band_doubling_stays_in_the_lane_per_job_kernels(emu, oracle_mod, port):
    """Compensating indels (net length change 0..2, path wandering 2..4 diagonals away): the first band fails (max < score,
    ssw.c:669) and doubles once or twice.  With one launch per band width the doubled run is served by the wider
    lane-per-job kernel; with everything fused (small batch, default) by the general kernel: same CIGARs, equal to the oracle."""
    rng = np.random.default_rng(77)
    w = rng.integers(0, 4, 320).astype(np.int8)
    reads = []
    for i in range(12):
        st = int(rng.integers(0, 120))
        r = w[st:st + 150].copy()
        a, b2 = 40 + i, 100 + i
        k = 1 + i % 4                                             # insert k bases at a, delete k (or k-1) bases at b2
        r = np.concatenate([r[:a], rng.integers(0, 4, k).astype(np.int8), r[a:b2], r[b2 + k - (i % 2):]])
        reads.append(r)
    jobs = JobTable.from_sequences(reads, [w], [0] * len(reads), [3, 2, 1, 3] * 3, [1, 0, 0, 0] * 3, encoded=True)
    mat = oracle_mod.dna_matrix(3, 2)
    exp = [port.align(r, w, mat, int(jobs.gap_open[i]), int(jobs.gap_ext[i])) for i, r in enumerate(reads)]
    assert sum(1 for e in exp if e["cigar"] and sum(1 for c in e["cigar"] if c & 15) >= 2) >= 6      # both indels recovered
    for routing in (R.ROUTE_TB_NO_FUSE | R.ROUTE_TB_NO_WAVE_PER_JOB, R.ROUTE_TB_NO_WAVE_PER_JOB, 0):
        a = emu(0, 3, 2)
        a.set_routing(routing)
        res = a.align(jobs)
        assert a.status == 0
        for i in range(len(reads)):
            assert res.as_dict(i) == exp[i], (routing, i)


def test_emu_half_precision_and_wavefront_forms_of_the_16_bit_passes(emu, oracle_mod, port):
    """k_dp_pass F16 / k_dp_skew: where every matrix entry is a half with a zero low byte and no score can pass 2047, the
    16-bit passes run in packed half precision (exact for these integers), as a wavefront over the SSE lanes (k_dp_skew) or
    column by column (k_dp_pass F16).  Same records as the integer form and as the oracle; matrices the form cannot
    represent (9, 11) and reads that could score above 2047 fall back to the integer form by themselves."""
    rng = np.random.default_rng(77)
    w = rng.integers(0, 4, 330).astype(np.int8)
    w2 = rng.integers(0, 4, 97).astype(np.int8)                           # a second, much shorter window in the same tiles
    reads, rid = [], []
    for i in range(40):
        win = w2 if i % 5 == 4 else w
        ln = int(rng.choice([40, 75, 101, 150, 151, 256] if win is w else [30, 60, 90]))
        st = int(rng.integers(0, len(win) - ln))
        r = win[st:st + ln].copy()
        for k in np.flatnonzero(rng.random(ln) < 0.03):
            r[k] = rng.integers(0, 5)                                     # some N among them
        if i % 4 == 1:
            r = np.concatenate([r[:ln // 2], r[ln // 2 + 4:]])            # deletion
        if i % 4 == 2:
            r = np.concatenate([r[:ln // 3], rng.integers(0, 4, 5).astype(np.int8), r[ln // 3:]])   # insertion
        reads.append(r)
        rid.append(1 if win is w2 else 0)
    gos, ges = [3, 5, 4, 12, 2] * 8, [1, 0, 2, 3, 1] * 8
    jobs = JobTable.from_sequences(reads, [w, w2], rid, gos, ges, encoded=True)
    for ms, mm, applies in ((3, 2, True), (5, 4, True), (7, 8, True), (1, 1, True), (9, 2, False), (3, 11, False), (8, 7, "short")):
        mat = oracle_mod.dna_matrix(ms, mm)
        out = []
        for routing in (0, R.ROUTE_NO_SKEW, R.ROUTE_NO_F16):
            a = emu(0, ms, mm)
            a.set_routing(routing)
            res = a.align(jobs)
            assert a.status == 0
            col = {k % 256 - 10 for k in a.launches if k // 256 == K_PACK and 10 <= k % 256 < 50}      # (markers of tests/emu: note_f16)
            wav = {k % 256 - 50 for k in a.launches if k // 256 == K_PACK and 50 <= k % 256 < 90}
            half = wav if routing == 0 else col
            assert not (col if routing == 0 else wav)
            if routing == R.ROUTE_NO_F16 or applies is False:
                assert not half
            elif applies is True:
                assert half >= {19, 32} or ms == 1                        # (match 1: no read leaves the 8-bit pass)
            else:
                assert half and max(half) <= 2047 // ms // 8              # 8 x segLen x max(mat) <= 2047
            out.append(res)
        assert all(out[0].as_dict(i) == out[1].as_dict(i) == out[2].as_dict(i) for i in range(len(reads)))
        _compare(out[0], [(r, (w, w2)[k], go, ge) for r, k, go, ge in zip(reads, rid, gos, ges)], port, mat)


def test_emu_rare_classes_ride_along_and_tiers_share_a_launch(emu, oracle_mod, port):
    """r03: (1) a segLen class with few reads is listed under the next populated class and served by that class's wavefront
    kernel with its rows shifted down (IpxBatch::cls_map, k_dp_skew ROW SHIFT) -- forward and reverse, 16-bit and plain 8-bit;
    (2) the classes of the stepped 8-bit passes share ONE launch (k_dp_pass_tier; the wavefront passes' tier launches of r03 are
    gone in r04).  Same records with either switched off, equal to
    the oracle; empty reads and reads that end before their window does are in the mix (their reverse prefix is much shorter
    than the read: a class of its own, decided on the device)."""
    rng = np.random.default_rng(99)
    wins = [rng.integers(0, 4, n).astype(np.int8) for n in (333, 210, 97)]
    reads, rid, go, ge = [], [], [], []
    lens = [150] * 14 + [100] * 10 + [104] * 2 + [131] * 2 + [75] * 8 + [61] * 2 + [33] * 2 + [250] * 6 + [201, 1, 0, 7]
    for i, ln in enumerate(lens):
        k = i % 3
        w = wins[k]
        ln = min(ln, len(w))
        st = int(rng.integers(0, len(w) - ln + 1))
        r = w[st:st + ln].copy()
        for q in np.flatnonzero(rng.random(ln) < 0.03):
            r[q] = rng.integers(0, 5)
        if i % 4 == 1 and ln > 40:
            r = np.concatenate([r[:ln // 2], r[ln // 2 + 3:]])
        if i % 4 == 2 and ln > 40:
            r = np.concatenate([r[:ln // 3], rng.integers(0, 4, 2).astype(np.int8), r[ln // 3:]])
        if i % 7 == 3 and ln > 60:
            r = np.concatenate([r[:ln // 2], rng.integers(0, 4, ln - ln // 2).astype(np.int8)])      # second half random: the alignment ends early
        reads.append(r); rid.append(k)
        g = [(3, 1), (5, 0), (4, 1), (3, 0)][i % 4]
        go.append(g[0]); ge.append(g[1])
    jobs = JobTable.from_sequences(reads, wins, rid, go, ge, encoded=True)
    for ms, mm in ((3, 2), (1, 1)):
        mat = oracle_mod.dna_matrix(ms, mm)
        out = {}
        for routing in (0, R.ROUTE_NO_TIERS, R.ROUTE_NO_CLASS_MERGE, R.ROUTE_NO_CLASS_MERGE | R.ROUTE_NO_PLAIN_FIRST):
            a = emu(0, ms, mm)
            a.set_routing(routing)
            out[routing] = (a, a.align(jobs))
            assert a.status == 0
        base = out[0][1]
        for routing, (a, res) in out.items():
            assert all(res.as_dict(i) == base.as_dict(i) for i in range(jobs.n_jobs)), routing
        _compare(base, [(r, wins[k], o_, e_) for r, k, o_, e_ in zip(reads, rid, go, ge)], port, mat)
        a0, a1, a2 = out[0][0], out[R.ROUTE_NO_TIERS][0], out[R.ROUTE_NO_CLASS_MERGE][0]
        tier = lambda a, kc: [k % 256 for k in a.launches if k // 256 == kc and k % 256 >= 150 and k % 256 < 160]
        if ms == 3:
            # 16-bit forward first: classes 13 (100 + the 104s), 17 (131), 19, 32 (250 + 201) ... rare ones merged away; 13..19 share tier 1
            assert not tier(a0, K_WORD_FIRST) and not tier(a1, K_WORD_FIRST)
            assert len(_launched(a1, K_WORD_FIRST)) < len(_launched(a2, K_WORD_FIRST))        # merged classes: fewer launches
            assert not [c for c in _launched(a1, K_WORD_REV) if c == 140]                   # no branch-guarded sweep launch left for the reverse pass
        else:
            assert not tier(a0, K_BYTE_PLAIN) and not tier(a1, K_BYTE_PLAIN)
            assert (base.records["mode"] == 0).all()
            # the stepped 8-bit passes (what the proofs leave open, slow gaps aside): ONE launch for the classes 1..16 (k_dp_pass_tier,
            # timing sub-key 158) instead of one per class; stand-alone launches again with the tiers switched off
            assert 158 in tier(a0, K_BYTE_EXACT) and 158 in tier(a0, K_BYTE_REV)
            assert not tier(a1, K_BYTE_EXACT) and len(_launched(a1, K_BYTE_EXACT)) > len(_launched(a0, K_BYTE_EXACT))


def test_emu_long_reads_take_the_long_read_kernel(emu, oracle_mod, port):
    """reads beyond the register-resident kernels (more than 64 striped segments: over 512 bp in the 16-bit passes, over 1 024 bp in
    the 8-bit ones) take k_dp_long, a transcription of the reference's loops with the striped columns in global memory -- forward,
    reverse, 8-bit (a low-scoring long read never leaves it) and 16-bit, fast and slow gaps, next to ordinary reads in the same batch"""
    rng = np.random.default_rng(4096)
    w = rng.integers(0, 4, 700).astype(np.int8)
    reads, go, ge = [], [], []
    for i, ln in enumerate([530, 600, 150, 1100, 640, 75, 513]):
        st = int(rng.integers(0, max(1, len(w) - ln))) if ln < len(w) else 0
        r = np.resize(w[st:], ln).copy()
        m = rng.random(ln) < (0.02 if i != 3 else 0.75)                   # the 1 100 bp read is mostly noise: it stays in the 8-bit pass
        r[m] = rng.integers(0, 4, int(m.sum()))
        if i % 2 == 0:
            r = np.concatenate([r[:ln // 3], r[ln // 3 + 5:]])
        reads.append(r)
        g = [(3, 1), (5, 0), (4, 1), (3, 1), (2, 2), (3, 1), (1, 0)][i]
        go.append(g[0]); ge.append(g[1])
    jobs = JobTable.from_sequences(reads, [w], [0] * len(reads), go, ge, encoded=True)
    for ms, mm in ((3, 2), (1, 1)):
        a = emu(0, ms, mm)
        res = a.align(jobs)
        assert a.status == 0
        _compare(res, [(r, w, o_, e_) for r, o_, e_ in zip(reads, go, ge)], port, oracle_mod.dna_matrix(ms, mm))
        assert any(k % 256 == 141 for k in a.launches)                    # IPX_SUB_LONG: the long-read kernel ran
        assert any(k % 256 == 142 for k in a.launches)                    # IPX_SUB_WIDE: ... and, for the 16-bit passes under fast gaps, k_dp_wide


def test_emu_long_reads_one_wavefront_per_read(emu, oracle_mod, port):
    """r04, k_dp_wide<16 / 32 / 48 / 64>: the 16-bit passes of reads from 505 bp under gap_open > gap_ext as one wavefront per read --
    64 lanes x S consecutive rows, 32-bit cells, the window staged in LDS in processing order.  Reads of every bucket (505..4 000 bp), with
    an insertion, a deletion, N, a read longer than its window, a window of 70 columns (shorter than the wavefront's 63 steps of lead-in),
    a read that matches nowhere; forward and reverse, every field and CIGAR against the oracle; the same jobs through the transcribed loops
    (ROUTE_NO_WIDE) give the same records."""
    from indelpost_amd.batch import ROUTE_NO_WIDE
    rng = np.random.default_rng(505)
    refs = [rng.integers(0, 4, n).astype(np.int8) for n in (900, 70, 2300)]
    reads, rid, go, ge = [], [], [], []
    for i, ln in enumerate([505, 512, 1000, 1024, 1030, 1700, 2047, 2100, 3000, 3080, 4000, 4096, 640, 777]):
        k = (0, 2, 0, 2, 1, 2, 0, 2, 0, 2, 2, 0, 1, 0)[i]
        w = refs[k]
        src = np.resize(w[int(rng.integers(0, 40)):], ln).copy() if i != 12 else rng.integers(0, 4, ln).astype(np.int8)
        m = rng.random(ln) < 0.03
        src[m] = rng.integers(0, 5, int(m.sum()))                         # (5 letters: a few N)
        if i % 3 == 0:
            src = np.concatenate([src[:ln // 2], src[ln // 2 + 7:]])
        if i % 3 == 1:
            src = np.concatenate([src[:ln // 3], rng.integers(0, 4, 9).astype(np.int8), src[ln // 3:]])[:4096]
        reads.append(src); rid.append(k)
        g = [(3, 1), (5, 0), (4, 1), (6, 2), (2, 1)][i % 5]
        go.append(g[0]); ge.append(g[1])
    for (ms, mm), pick in (((3, 2), range(len(reads))), ((1, 3), (0, 3, 4, 7, 9, 11))):      # (the second scoring: one read of every bucket)
        jobs = JobTable.from_sequences([reads[i] for i in pick], refs, [rid[i] for i in pick], [go[i] for i in pick], [ge[i] for i in pick], encoded=True)
        a = emu(0, ms, mm)
        res = a.align(jobs)
        assert a.status == 0
        _compare(res, [(reads[i], refs[rid[i]], go[i], ge[i]) for i in pick], port, oracle_mod.dna_matrix(ms, mm))
        assert sum(n for k, n in a.launches.items() if k % 256 == 142) >= 8     # four buckets, forward and reverse
        assert not any(k % 256 == 141 and k // 256 in (K_WORD_FIRST, K_WORD_FWD, K_WORD_REV) for k in a.launches)
    # (the transcribed loops are slow in the emulator -- ballots per lazy-F step: three of the shorter jobs, one scoring)
    few = [0, 1, 4]
    sub = JobTable.from_sequences([reads[i] for i in few], refs, [rid[i] for i in few], [go[i] for i in few], [ge[i] for i in few], encoded=True)
    a, b = emu(0, 3, 2), emu(0, 3, 2)
    b.set_routing(ROUTE_NO_WIDE)
    res, res2 = a.align(sub), b.align(sub)
    assert a.status == 0 and b.status == 0 and not any(k % 256 == 142 for k in b.launches) and any(k % 256 == 142 for k in a.launches)
    assert res2.records.tobytes() == res.records.tobytes() and res2.cigar_pool.tobytes() == res.cigar_pool.tobytes()


def test_emu_reverse_pass_as_a_band(emu, oracle_mod, port):
    """r04, k_rev_split / k_dp_band_rev: the 16-bit reverse pass of a job whose score budget bounds its drift from the end cell's diagonal by
    the class's band half-width (segments + 1: 20 for 150 bp) is computed inside that band, one lane per pair of reads, block of rows after block of rows; the other jobs take the full
    wavefront kernel through a list of their own.  Clean reads, reads with junk in front of / behind the aligned part (the begin cell is not
    row 0, the end cell not the last base), short and long indels (inside and beyond the band), noise, N, gap_ext 0 and 2, read lengths across
    several classes and windows shorter than the reads; every field and CIGAR against the oracle; the same records with the band switched off."""
    rng = np.random.default_rng(2020)
    w = rng.integers(0, 4, 420).astype(np.int8)
    w2 = rng.integers(0, 4, 110).astype(np.int8)
    reads, rid, go, ge = [], [], [], []
    for i in range(150):
        win = w2 if i % 7 == 6 else w
        ln = int(rng.choice([66, 70, 100, 104, 128, 150, 152, 200, 250, 256])) if win is w else int(rng.choice([70, 90, 100]))
        st = int(rng.integers(0, max(1, len(win) - ln)))
        r = np.resize(win[st:], ln).copy()
        kind = i % 10
        if kind in (1, 2):
            m = rng.random(ln) < (0.01, 0.06)[kind - 1]
            r[m] = rng.integers(0, 5, int(m.sum()))
        if kind == 3:
            r = np.concatenate([r[:ln // 2], r[ln // 2 + int(rng.integers(1, 9)):]])
        if kind == 4:
            r = np.concatenate([r[:ln // 3], rng.integers(0, 4, int(rng.integers(1, 9))).astype(np.int8), r[ln // 3:]])
        if kind == 5:
            r = np.concatenate([r[:ln // 2], r[ln // 2 + int(rng.integers(22, 40)):]])                # a deletion beyond the band
        if kind == 6:
            r = np.concatenate([rng.integers(0, 4, int(rng.integers(5, 30))).astype(np.int8), r[10:]])  # junk in front: the alignment begins inside the read
        if kind == 7:
            r = np.concatenate([r[:-12], rng.integers(0, 4, int(rng.integers(5, 25))).astype(np.int8)])  # junk behind
        if kind == 8:
            r = rng.integers(0, 4, ln).astype(np.int8)                                               # noise
        reads.append(r[:256]); rid.append(1 if win is w2 else 0)
        g = [(3, 1), (5, 2), (4, 1), (3, 0), (6, 1), (2, 1)][i % 6]
        go.append(g[0]); ge.append(g[1])
    jobs = JobTable.from_sequences(reads, [w, w2], rid, go, ge, encoded=True)
    for ms, mm in ((3, 2), (5, 4), (1, 1)):
        a = emu(0, ms, mm)
        res = a.align(jobs)
        assert a.status == 0
        _compare(res, [(r, (w, w2)[k], o_, e_) for r, k, o_, e_ in zip(reads, rid, go, ge)], port, oracle_mod.dna_matrix(ms, mm))
        # (match 1: every read stays in the 8-bit passes -- the plain recurrence in the 8-bit dialect and ITS reverse pass as a band)
        kc = K_BYTE_REV_PLAIN if ms == 1 else K_WORD_REV
        band = sum(n for k, n in a.launches.items() if k // 256 == kc and k % 256 == 143)
        assert band >= 4                                                      # several classes took the band kernel ...
        assert any(n for k, n in a.launches.items() if k // 256 == kc and 4 <= k % 256 <= 32)      # ... and the full kernel its leftovers
        o = emu(0, ms, mm)
        o.set_routing(R.ROUTE_NO_BAND_REV)
        res2 = o.align(jobs)
        assert o.status == 0 and not any(k % 256 == 143 for k in o.launches)
        assert res2.records.tobytes() == res.records.tobytes() and res2.cigar_pool.tobytes() == res.cigar_pool.tobytes()
    assert (res.records["read_begin1"] > 0).sum() >= 5                                     # alignments that begin inside the read were among them


def test_emu_anti_diagonal_traceback_tiers(emu, oracle_mod, port):
    """r04, k_tb_diag<16 / 32 / 64>: banded_sw as an anti-diagonal wavefront.  Jobs whose first band |refLen - readLen| + 1 runs from 1 to
    ~90 (deletions and insertions of 1..90 bp bridged under cheap gap extension), compensating indels that make a band double inside
    a tier and across tiers, windows narrower than the band arrays (the reference's `h_b[edge] = 0` wipes a cell inside the band there),
    reads with N, short rectangles; every field and CIGAR against the oracle -- as a small batch (every job takes a tier), behind the
    lane-per-job kernels (large-batch routing), and with the tiers switched off."""
    rng = np.random.default_rng(404)
    w = rng.integers(0, 4, 520).astype(np.int8)
    w2 = rng.integers(0, 2, 90).astype(np.int8)                     # low complexity: ties, wandering paths
    reads, rid, go, ge = [], [], [], []
    gaps = [(3, 0), (1, 0), (3, 1), (2, 1), (5, 0), (4, 1), (1, 1), (0, 0)]
    for i in range(72):
        st = int(rng.integers(0, 200))
        L = int(rng.integers(60, 200))
        r = w[st:st + L].copy()
        kind = i % 6
        d = (1, 3, 9, 14, 17, 30, 33, 47, 62, 70, 90, 5, 22, 12)[(i // 6 + 5 * (i % 6)) % 14]
        if kind in (0, 1):                                         # deletion of d bp: first band d + 1
            cut = int(rng.integers(20, L - 20))
            r = np.concatenate([w[st:st + cut], w[st + cut + d:st + L + d]])
        elif kind == 2:                                            # insertion of d bp
            cut = int(rng.integers(20, L - 20))
            r = np.concatenate([r[:cut], rng.integers(0, 4, min(d, 60)).astype(np.int8), r[cut:]])
        elif kind == 3:                                            # insertion of k, deletion of k further on: the band doubles
            k = 1 + i % 11
            a, b2 = 25, L - 30
            r = np.concatenate([r[:a], rng.integers(0, 4, k).astype(np.int8), r[a:b2], r[b2 + k:]])
        elif kind == 4:
            r[rng.integers(0, len(r), 4)] = 4
        if i % 9 == 0:
            r = r[:int(rng.integers(3, 12))]                         # rectangles of a few cells
        g = gaps[i % len(gaps)]
        reads.append(r); rid.append(0); go.append(g[0]); ge.append(g[1])
    for i in range(16):                                            # low-complexity window, 2-letter reads
        st = int(rng.integers(0, 30))
        r = w2[st:st + int(rng.integers(8, 55))].copy()
        if i % 2:
            r = np.concatenate([r[:5], r[5 + i % 7:]])
        g = gaps[(i + 3) % len(gaps)]
        reads.append(r); rid.append(1); go.append(g[0]); ge.append(g[1])
    jobs = JobTable.from_sequences(reads, [w, w2], rid, go, ge, encoded=True)
    mat = oracle_mod.dna_matrix(3, 2)
    refs = [w, w2]
    exp = [port.align(r, refs[rid[i]], mat, go[i], ge[i]) for i, r in enumerate(reads)]
    first_bands = [abs((e["ref_end1"] - e["ref_begin1"]) - (e["read_end1"] - e["read_begin1"])) + 1 for e in exp if e["cigar"]]
    assert sum(1 for f in first_bands if 8 <= f <= 15) >= 3 and sum(1 for f in first_bands if 16 <= f <= 31) >= 3
    assert sum(1 for f in first_bands if 32 <= f <= 63) >= 3 and sum(1 for f in first_bands if f >= 64) >= 2
    seen = {}
    for routing in (0, R.ROUTE_TB_NO_WAVE_PER_JOB, R.ROUTE_TB_NO_WAVE_PER_JOB | R.ROUTE_TB_NO_FUSE, R.ROUTE_TB_NO_DIAG, R.ROUTE_TB_NO_UNGAPPED):
        a = emu(0, 3, 2)
        a.set_routing(routing)
        res = a.align(jobs)
        assert a.status == 0
        for i in range(len(reads)):
            assert res.as_dict(i) == exp[i], (routing, i, res.as_dict(i), exp[i])
        seen[routing] = a.tb_jobs
    coop, d16, d32, d64 = seen[0]
    assert d16 == 0 and d32 >= 60 and d64 >= 4 and 2 <= coop <= 12, seen      # small batch: every job starts at 32 lanes at least; only bands past 63 reach k_tb_coop
    big = seen[R.ROUTE_TB_NO_WAVE_PER_JOB]
    assert big[2] > sum(1 for f in first_bands if 16 <= f <= 31) or big[3] > sum(1 for f in first_bands if 32 <= f <= 63), (seen, first_bands)   # some reach a tier by doubling out of the one below
    assert 5 <= seen[R.ROUTE_TB_NO_WAVE_PER_JOB][1] < 40                       # behind the lane-per-job kernels the first tier sees wide and doubled bands only
    assert seen[R.ROUTE_TB_NO_DIAG][1:] == [0, 0, 0]
    # a tiny batch: one job per wave
    tiny = JobTable.from_sequences(reads[:56], refs, rid[:56], go[:56], ge[:56], encoded=True)
    a = emu(0, 3, 2)
    res = a.align(tiny)
    assert a.status == 0 and all(res.as_dict(i) == exp[i] for i in range(56))
    assert a.tb_jobs[1] == 0 and a.tb_jobs[2] == 0 and a.tb_jobs[3] >= 40, a.tb_jobs

def test_emu_latency_tier_32_lanes_per_read(emu, oracle_mod, port):
    """r04, k_dp_skew<S, REV, BH, 32>: the wavefront passes of a small batch at 32 lanes per read (four reads per wave, every class of a pass
    in one launch, a whole-wavefront lane shift, windows staged in LDS).  Reads of 1..256 bp against windows shorter and longer than the reads, with N, with
    indels, empty reads; gap penalties with gap_open > gap_ext only (anything else keeps the batch off the tier); scorings that put reads
    in the 16-bit passes (3,2), in the plain 8-bit flow (1,1) and in both (2,2); every field and CIGAR against the oracle, and the same
    batch with the tier forced off.  (ROUTE_FORCE_LAT: the emulator build never takes the tier by itself.)"""
    rng = np.random.default_rng(3232)
    refs = [rng.integers(0, 4, int(n)).astype(np.int8) for n in (40, 160, 300, 333, 610)]
    refs[3][rng.integers(0, 333, 20)] = 4
    reads, rid, go, ge = [], [], [], []
    gaps = [(3, 1), (3, 0), (5, 1), (4, 0), (2, 1), (10, 1), (255, 1)]
    for i in range(64):
        k = int(rng.integers(0, len(refs)))
        w = refs[k]
        L = int(rng.integers(1, 257)) if i % 7 else (1, 7, 8, 9, 31, 32, 33, 64, 65, 128, 255, 256, 0)[(i // 7) % 13]
        st = int(rng.integers(0, max(1, len(w) - 10)))
        r = np.resize(w[st:], L).copy() if L else np.zeros(0, np.int8)
        if L > 20 and i % 3 == 0:
            cut = int(rng.integers(5, L - 5))
            r = np.concatenate([r[:cut], r[cut + 1 + i % 4:]])
        if L > 20 and i % 3 == 1:
            cut = int(rng.integers(5, L - 5))
            r = np.concatenate([r[:cut], rng.integers(0, 4, 1 + i % 3).astype(np.int8), r[cut:]])[:256]
        m = rng.random(len(r)) < (0.0, 0.03, 0.12)[i % 3]
        r[m] = rng.integers(0, 4, int(m.sum()))
        if i % 11 == 0 and len(r):
            r[rng.integers(0, len(r), max(1, len(r) // 9))] = 4
        if i % 13 == 5:
            r = rng.integers(0, 4, len(r)).astype(np.int8)
        g = gaps[i % len(gaps)]
        reads.append(r); rid.append(k); go.append(g[0]); ge.append(g[1])
    jobs = JobTable.from_sequences(reads, refs, rid, go, ge, encoded=True)
    K_WORD_FIRST, K_WORD_REV, K_BYTE_PLAIN = 6, 9, 14          # IPX_K_* of csrc/ipx_pipeline.h
    for scoring in ((3, 2), (1, 1), (2, 2)):
        mat = oracle_mod.dna_matrix(*scoring)
        exp = [port.align(r, refs[rid[i]], mat, go[i], ge[i]) for i, r in enumerate(reads)]
        got = {}
        for routing in ((R.ROUTE_FORCE_LAT, R.ROUTE_FORCE_LAT | R.ROUTE_NO_LAT_PROOF, R.ROUTE_FORCE_LAT | R.ROUTE_NO_PLAIN_FIRST) if scoring == (3, 2) else (R.ROUTE_FORCE_LAT,)):
            a = emu(0, *scoring)
            a.set_routing(routing)
            res = a.align(jobs)
            assert a.status == 0
            for i in range(len(reads)):
                assert res.as_dict(i) == exp[i], (scoring, routing, i, len(reads[i]), res.as_dict(i), exp[i])
            got[routing] = {kc: sorted(c for c in range(140) if a.launches.get(a.key(kc, c))) for kc in (K_WORD_FIRST, K_WORD_REV, K_BYTE_PLAIN)}
        lat = got[R.ROUTE_FORCE_LAT]
        assert all(len(v) <= 1 for v in lat.values()), lat                       # one launch per pass on the tier ...
    # a job with gap_open <= gap_ext keeps the whole batch off the tier (its stepped kernels have tiles of their own size)
    jobs2 = JobTable.from_sequences(reads[:20], refs, rid[:20], [3] * 19 + [1], [1] * 19 + [1], encoded=True)
    a = emu(0, 3, 2)
    a.set_routing(R.ROUTE_FORCE_LAT)
    res = a.align(jobs2)
    mat = oracle_mod.dna_matrix(3, 2)
    assert a.status == 0 and all(res.as_dict(i) == port.align(reads[i], refs[rid[i]], mat, 3 if i < 19 else 1, 1) for i in range(20))


def test_emu_latency_tier_speculation_is_caught_by_the_guard(emu, oracle_mod, port):
    """r04: the latency tier leaves out passes the previous run found empty; k_tb_list notices a job left behind in one (status bit
    IPX_STATUS_RERUN) and the run is repeated with every pass.  ROUTE_TEST_SKIP_ALL predicts EVERY dynamic pass empty: the first
    attempt leaves every record without its reverse pass (and the 8-bit reads without their proofs), nothing of it may reach the
    traceback kernels, and the repeated run must give the oracle's answers."""
    rng = np.random.default_rng(515)
    w = rng.integers(0, 4, 260).astype(np.int8)
    reads = []
    for i in range(24):
        L = (150, 60, 100, 33)[i % 4]
        st = int(rng.integers(0, 260 - L))
        r = w[st:st + L].copy()
        r[rng.integers(0, L, 2)] ^= 1
        if i % 3 == 0:
            r = np.concatenate([r[:L // 2], r[L // 2 + 2:]])
        reads.append(r)
    jobs = JobTable.from_sequences(reads, [w], [0] * len(reads), 3, 1, encoded=True)
    mat = oracle_mod.dna_matrix(3, 2)
    exp = [port.align(r, w, mat, 3, 1) for r in reads]
    K_PACK = 12
    for routing, rerun in ((R.ROUTE_FORCE_LAT | R.ROUTE_TEST_SKIP_ALL, True), (R.ROUTE_FORCE_LAT, False), (R.ROUTE_FORCE_LAT | R.ROUTE_TEST_SKIP_ALL | R.ROUTE_NO_SPECULATE, False)):
        a = emu(0, 3, 2)
        a.set_routing(routing)
        res = a.align(jobs)
        assert a.status == 0
        assert bool(a.launches.get(a.key(K_PACK, 5))) == rerun, (routing, a.launches.get(a.key(K_PACK, 5)))
        for i in range(len(reads)):
            assert res.as_dict(i) == exp[i], (routing, i)
