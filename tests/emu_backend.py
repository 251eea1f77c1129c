"""EmuAligner: the GpuAligner interface on top of tests/emu/libipx_emu.so (TEST INFRASTRUCTURE).

The emulator library is the product kernel source (indelpost_amd/csrc/ipx_kernels.h and the launch
sequence ipx_pipeline.h) compiled with g++ -DIPX_CPU_EMU and executed by a lock-step 64-fiber wave
emulator.  It lets the CPU-only suite check kernel logic, the planner and the host code paths
against the oracle.  It is never imported by the package.
"""
import ctypes as C
import os
import subprocess
import threading

import numpy as np

from indelpost_amd._lib import RESULT_DTYPE
from indelpost_amd.batch import BatchResult, dna_score_matrix

HERE = os.path.dirname(os.path.abspath(__file__))
EMU_DIR = os.path.join(HERE, "emu")
EMU_LIB = os.path.join(EMU_DIR, "libipx_emu.so")
CSRC = os.path.join(os.path.dirname(HERE), "indelpost_amd", "csrc")


def build_emu(force=False):
    srcs = [os.path.join(EMU_DIR, "emu.cpp")] + [os.path.join(CSRC, h) for h in
                                                   ("ipx_simt.h", "ipx_types.h", "ipx_kernels.h", "ipx_pipeline.h")]
    if not force and os.path.exists(EMU_LIB) and all(os.path.getmtime(s) <= os.path.getmtime(EMU_LIB) for s in srcs):
        return EMU_LIB
    # (-g doubles the compile time of this one big translation unit: only on request, IPX_EMU_DEBUG=1)
    subprocess.check_call(["g++", "-O1"] + (["-g"] if os.environ.get("IPX_EMU_DEBUG") else []) + ["-std=c++17", "-DIPX_CPU_EMU", "-DIPX_PROVE_CHUNK_MIN_JOBS=0", "-DIPX_MERGE_BELOW=6", "-DIPX_MERGE_TINY=2", "-DIPX_LAT_MAX_JOBS=0", "-DIPX_BAND_MIN_TILES=0", "-DIPX_TB_SMALL_DIAG=2048", "-DIPX_TB_TINY_DIAG=60", "-fPIC", "-shared",
                           "-U_FORTIFY_SOURCE", "-Wall", "-Wno-unused-function", "-o", EMU_LIB, srcs[0]])
    return EMU_LIB


_EMU_LOCK = threading.Lock()   # the emulator keeps global fiber state: one batch at a time


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


class EmuAligner:
    def __init__(self, device=0, match_score=2, mismatch_penalty=2, matrix=None):
        self._L = C.CDLL(build_emu())
        self.device = device
        self.routing = 0
        self.launches = {}
        self.set_scoring(match_score, mismatch_penalty, matrix)

    def set_routing(self, flags):
        """speed-only routing switches (IPX_ROUTE_* of ipx_types.h / indelpost_amd.batch.ROUTE_*)"""
        self.routing = int(flags)

    def key(self, kclass, sub):
        return int(self._L.emu_key(kclass, sub))

    def set_scoring(self, match_score=2, mismatch_penalty=2, matrix=None, flag=1, filters=0, filterd=0, score_size=2):
        self.matrix = dna_score_matrix(match_score, mismatch_penalty) if matrix is None else np.ascontiguousarray(matrix, np.int8)
        self.flag, self.filters, self.filterd, self.score_size = flag, filters, filterd, score_size

    def align(self, jobs):
        n = jobs.n_jobs
        rec = np.zeros(n, RESULT_DTYPE)
        cap = 2 * int(jobs.read_off[-1]) + 64 * n + 64
        pool = np.zeros(cap, np.uint32)
        st = C.c_uint32(0)
        launches = np.zeros(int(self._L.emu_num_keys()), np.int32)
        pass_jobs = np.zeros(16, np.uint32)
        reads = np.concatenate([jobs.reads, np.zeros(8, np.int8)])
        refs = np.concatenate([jobs.refs, np.zeros(8, np.int8)])
        with _EMU_LOCK:
            self._L.emu_align_batch(_p(reads), _p(jobs.read_off), _p(refs), _p(jobs.ref_off), _p(jobs.ref_id),
                                    _p(jobs.gap_open), _p(jobs.gap_ext), _p(jobs.mask_len), _p(self.matrix),
                                    C.c_int64(n), C.c_int32(jobs.n_refs), self.flag, self.filters, self.filterd,
                                    self.score_size, self.routing, _p(rec), _p(pool), C.c_uint32(cap), C.byref(st), _p(launches), _p(pass_jobs))
        self.status = st.value
        self.launches = {int(k): int(launches[k]) for k in np.flatnonzero(launches)}
        self.pass_jobs = pass_jobs[:10].tolist()    # jobs per pass (IPX_PASS_* order of csrc/ipx_types.h)
        self.tb_jobs = pass_jobs[11:15].tolist()    # jobs listed for k_tb_coop and for k_tb_diag<16 / 32 / 64> (hand-overs included)
        used = int((rec["cigar_off"].astype(np.int64) + rec["cigar_len"]).max()) if n else 0
        return BatchResult(rec, pool[:used])

    # staged interface of GpuAligner (the emulator runs at sync time)
    def upload(self, jobs):
        self._jobs = jobs

    def run(self):
        self._res = None

    def sync(self):
        self._res = self.align(self._jobs)

    def download(self, cigar_ops_per_job=16):
        return self._res

    # asynchronous-download interface of GpuAligner (stub: the copy happens at once), for the bookkeeping tests of MultiStreamAligner
    _n_jobs = property(lambda self: self._jobs.n_jobs)

    def set_async_io(self, on=True):
        pass

    def download_async_into(self, rec, pool):
        used = len(self._res.cigar_pool)
        if used > len(pool):
            return -used
        rec[:] = self._res.records
        pool[:used] = self._res.cigar_pool
        return used

    def wait(self):
        pass

    def set_profiling(self, on):
        pass

    def kernel_times(self):
        return {}

    def kernel_units(self):
        return {}

    def last_run_ms(self):
        return 0.0

    def close(self):
        pass
