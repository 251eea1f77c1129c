"""What k_dp_skew rests on ("the 16-bit passes as a wavefront"): the reference's 16-bit pass never feeds an H that
its lazy-F loop raised back into E (ssw.c:507-518), yet its H matrix equals the plain Gotoh matrix whenever gap_open > gap_ext, and
every output of the pass is a function of H.  Checked here on the CPU: a plain, unstriped Gotoh matrix (no lazy-F, no segments) with the
reference's output rules (first column holding the maximum, smallest row there, best column maximum outside the mask and its first
column, ssw.c:521-581) against the compiled reference / the restatement run in 16-bit mode, and the same matrix read backwards against
the reference's begin positions.  One thing the striping does leave in the matrix: the read is padded to 8 x segLen rows, the
padding rows score 0 against every letter and their H enters the column maxima (an alignment that ends on the last base
keeps its score for up to 7 more columns there: found by this test) -- so the plain matrix is taken over the padded read."""
import numpy as np
import pytest


def gotoh(read, ref, mat, go, ge):
    """H[i][j] of local alignment with affine gaps: E, F take H of the cell they leave minus gap_open, or extend by gap_ext."""
    n, m = len(read), len(ref)
    H = np.zeros((n, m), np.int64)
    E = np.zeros((n, m), np.int64)                      # gap along the window (arrives from the left)
    F = np.zeros((n, m), np.int64)                      # gap along the read (arrives from above)
    for i in range(n):
        for j in range(m):
            e = max(E[i][j - 1] - ge, H[i][j - 1] - go, 0) if j > 0 else 0
            f = max(F[i - 1][j] - ge, H[i - 1][j] - go, 0) if i > 0 else 0
            d = (H[i - 1][j - 1] if i > 0 and j > 0 else 0) + int(mat[int(ref[j]) * 5 + int(read[i])])
            E[i][j], F[i][j] = e, f
            H[i][j] = max(d, e, f, 0)
    return H


def outputs(H, mask_len):
    n, m = H.shape
    colmax = H.max(axis=0)
    best = int(colmax.max())
    if best == 0:
        return None
    ref_end = int(np.argmax(colmax == best))
    read_end = int(np.argmax(H[:, ref_end] == best))
    lo, hi = max(ref_end - mask_len, 0), min(ref_end + mask_len, m)
    s2, e2 = 0, 0
    for c in list(range(0, lo)) + list(range(hi, m)):
        if colmax[c] > s2:
            s2, e2 = int(colmax[c]), c
    return best, ref_end, read_end, s2, e2


@pytest.mark.parametrize("scoring", [(3, 2), (1, 1), (5, 4), (2, 7)])
def test_plain_gotoh_matrix_gives_the_16_bit_pass_outputs(oracle_mod, port, scoring):
    rng = np.random.default_rng(100 + scoring[0])
    mat = oracle_mod.dna_matrix(*scoring)
    be = oracle_mod.Backend("reference") if oracle_mod.have_reference() else port
    n_checked = 0
    for case in range(150):
        m = int(rng.integers(20, 70))
        w = rng.integers(0, 4, m).astype(np.int8)
        if case % 3 == 0:
            w = np.resize(np.array([0, 1, 0, 0, 1], np.int8), m)           # low complexity: ties everywhere
        L = int(rng.integers(8, 40))
        st = int(rng.integers(0, max(1, m - L)))
        r = w[st:st + L].copy()
        for k in np.flatnonzero(rng.random(len(r)) < 0.08):
            r[k] = rng.integers(0, 5)
        if case % 2:
            cut = int(rng.integers(2, max(3, len(r) - 2)))
            r = np.concatenate([r[:cut], rng.integers(0, 4, int(rng.integers(1, 5))).astype(np.int8), r[cut:]])   # insertion
        if case % 5 == 0 and len(r) > 12:
            cut = int(rng.integers(2, len(r) - 6))
            r = np.concatenate([r[:cut], r[cut + int(rng.integers(1, 5)):]])                                      # deletion
        go, ge = [(3, 1), (5, 0), (4, 2), (12, 3), (2, 1)][case % 5]
        pad = lambda q: np.concatenate([q, np.full(-len(q) % 8, 4, np.int8)])   # rows up to 8 x segLen: letter N scores 0 (ssw.c:398-402)
        H = gotoh(pad(r), w, mat, go, ge)
        mask_len = max(15, len(r) // 2)
        out = outputs(H, mask_len)
        if out is not None:
            out = (out[0], out[1], min(out[2], len(r) - 1)) + out[3:]            # end_read starts at readLen - 1 (ssw.c:426)
        e = be.align(r, w, mat, go, ge, flag=1, score_size=1)              # 16-bit pass only (ssw.c:853-855)
        if out is None:
            assert e["score1"] == 0
            continue
        assert (e["score1"], e["ref_end1"], e["read_end1"]) == out[:3], (case, out, e)
        assert (e["score2"], e["ref_end2"]) == (out[3], out[4]), (case, out, e)
        # the reverse pass: the same matrix of the reversed prefixes; first column (from the end) holding score1, smallest row there
        rr, wr = r[:e["read_end1"] + 1][::-1], w[:e["ref_end1"] + 1][::-1]
        Hr = gotoh(pad(rr), wr, mat, go, ge)
        cm = Hr.max(axis=0)
        c = int(np.argmax(cm == out[0]))
        assert cm.max() == out[0]                                           # the score to reach is the maximum of this matrix
        assert (e["ref_begin1"], e["read_begin1"]) == (e["ref_end1"] - c, e["read_end1"] - min(int(np.argmax(Hr[:, c] == out[0])), len(rr) - 1)), case
        n_checked += 1
    assert n_checked > 100
