"""Deterministic synthetic workloads: xorshift64 window + error-model reads.

The generator itself is C (ipx_synth_* in libindelpost_hip.so, host code) so that a million reads
take milliseconds; these wrappers only shape numpy arrays.
"""
import numpy as np

from . import _lib
from .batch import JobTable

SEED = 88172645463325252


def window_and_reads(n_reads, read_len=150, window_len=300, seed=SEED):
    L = _lib.lib()
    ref = np.zeros(window_len, np.int8)
    st = L.ipx_synth_window(seed, ref.ctypes.data, window_len)
    reads = np.zeros(n_reads * read_len, np.int8)
    st = L.ipx_synth_reads(st, ref.ctypes.data, window_len, reads.ctypes.data, n_reads, read_len)
    return ref, reads.reshape(n_reads, read_len), st


def config2_jobs(n_reads, gap_open=3, gap_ext=1, read_len=150, window_len=300, seed=SEED):
    """BASELINE.json configs[1]: n synthetic 150 bp reads vs one 300 bp window."""
    ref, reads, _ = window_and_reads(n_reads, read_len, window_len, seed)
    read_off = np.arange(n_reads + 1, dtype=np.int64) * read_len
    return JobTable(reads.reshape(-1), read_off, ref, np.array([0, window_len], np.int64),
                    np.zeros(n_reads, np.int32), gap_open, gap_ext)


def mixed_jobs(n_windows, per, read_lens, wl_lo, wl_hi, gap_open=3, gap_ext=1, seed=SEED):
    """Many windows, several read lengths (one call into the C generator): `per` reads of every length in
    `read_lens` for each of `n_windows` windows whose length is drawn from [wl_lo, wl_hi]."""
    L = _lib.lib()
    rls = np.ascontiguousarray(read_lens, np.int32)
    nj = n_windows * per * len(rls)
    refs = np.zeros(n_windows * wl_hi, np.int8)
    ref_off = np.zeros(n_windows + 1, np.int64)
    reads = np.zeros(n_windows * per * int(rls.sum()), np.int8)
    read_off = np.zeros(nj + 1, np.int64)
    ref_id = np.zeros(nj, np.int32)
    import ctypes as C
    st = C.c_uint64(seed)
    n = L.ipx_synth_mixed(C.byref(st), n_windows, wl_lo, wl_hi, rls.ctypes.data, len(rls), per, refs.ctypes.data,
                          ref_off.ctypes.data, reads.ctypes.data, read_off.ctypes.data, ref_id.ctypes.data)
    if n != nj:
        raise RuntimeError("ipx_synth_mixed returned %d" % n)
    return JobTable(reads[:read_off[-1]], read_off, refs[:ref_off[-1]], ref_off, ref_id, gap_open, gap_ext)


def config4_jobs(n_windows=1000, reads_per_window=996, seed=SEED):
    """BASELINE.json configs[3] shape (SURVEY.md 8d "Config 4"): read lengths {75,100,125,150,200,250}, windows of
    200-600 bp, one window per ~1000 reads, indelPost defaults (3,2,3,1): short reads stay in the 8-bit pass, long
    ones are rescored in 16 bit.  n_windows=1000 gives 996 000 jobs (a tenth of the full 10 M)."""
    lens = [75, 100, 125, 150, 200, 250]
    return mixed_jobs(n_windows, reads_per_window // len(lens), lens, 200, 600, 3, 1, seed)


PENALTY_GRID = [(3, 1), (3, 0), (5, 1), (5, 0), (4, 1), (4, 0)]     # varaln.pyx:1127-1143


def config5_jobs(n_loci=12500, reads_per_locus=16, seed=SEED):
    """BASELINE.json configs[4] shape (SURVEY.md 8d "Config 5"): per-read 300 bp windows (retarget, pileup.pyx:639-648)
    x the gap-penalty grid of grid_search (varaln.pyx:1127-1143); job 6k+g = read k under grid pair g.
    12 500 loci x 16 reads x 6 pairs = 1.2 M jobs (an eighth of the full 9.6 M)."""
    base = mixed_jobs(n_loci * reads_per_locus, 1, [150], 300, 300, 3, 1, seed)
    g = len(PENALTY_GRID)
    n = base.n_jobs
    reads = np.repeat(base.reads.reshape(n, 150), g, axis=0).reshape(-1)
    read_off = np.arange(n * g + 1, dtype=np.int64) * 150
    go = np.tile(np.array([p[0] for p in PENALTY_GRID], np.uint8), n)
    ge = np.tile(np.array([p[1] for p in PENALTY_GRID], np.uint8), n)
    return JobTable(reads, read_off, base.refs, base.ref_off, np.repeat(base.ref_id, g), go, ge)
