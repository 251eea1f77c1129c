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
