"""The C ABI: the library loads here (no GPU), exports every symbol include/indelpost_hip.h
declares, and refuses loudly to run without a device (no CPU fallback)."""
import ctypes as C
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_functions():
    txt = open(os.path.join(ROOT, "include", "indelpost_hip.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    names = set()
    for m in re.finditer(r"^[A-Za-z_][\w\s\*]*?\b(\w+)\s*\(", txt, flags=re.M):
        if "static" in m.group(0) or "typedef" in m.group(0):
            continue
        names.add(m.group(1))
    return names


def test_header_symbols_are_exported(hip_lib):
    from indelpost_amd import _lib
    declared = _declared_functions()
    assert {"ssw_init", "ssw_align", "align_destroy", "init_destroy", "ipx_align_batch"} <= declared
    assert declared == set(_lib.EXPORTS)
    for name in declared:
        assert hasattr(hip_lib, name), name


def test_result_record_layout():
    from indelpost_amd._lib import RESULT_DTYPE
    assert RESULT_DTYPE.itemsize == 32
    assert [RESULT_DTYPE.fields[n][1] for n in ("score1", "score2", "ref_begin1", "ref_end1", "read_begin1",
                                               "read_end1", "ref_end2", "cigar_off", "cigar_len", "flag", "mode")] == \
        [0, 2, 4, 8, 12, 16, 20, 24, 28, 30, 31]


@pytest.mark.skipif(os.path.exists("/dev/kfd"), reason="a GPU is present")
def test_no_gpu_is_a_loud_failure(hip_lib):
    import indelpost_amd as ip
    assert ip.device_count() == 0
    with pytest.raises(ip.IpxError) as e:
        ip.GpuAligner(0)
    assert "no CPU fallback" in str(e.value)
    a = ip.SSW(3, 2)
    a.setReference("ACGTACGT")
    a.setRead("ACGT")
    with pytest.raises(ip.IpxError):
        a.align()
    # the reference-compatible entry point reports NULL like ssw_align does on error (ssw.c:848-859)
    hip_lib.ssw_init.restype = C.c_void_p
    hip_lib.ssw_init.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_int8]
    hip_lib.ssw_align.restype = C.c_void_p
    hip_lib.ssw_align.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_uint8, C.c_uint8, C.c_uint8, C.c_uint16,
                                  C.c_int32, C.c_int32]
    hip_lib.init_destroy.argtypes = [C.c_void_p]
    rd = np.array([0, 1, 2, 3], np.int8)
    mat = ip.dna_score_matrix(2, 2)
    p = hip_lib.ssw_init(rd.ctypes.data, 4, mat.ctypes.data, 5, 2)
    assert p
    assert hip_lib.ssw_align(p, rd.ctypes.data, 4, 3, 1, 1, 0, 0, 15) is None
    hip_lib.init_destroy(p)


def test_synth_generator_is_deterministic(hip_lib):
    from indelpost_amd import synth
    ref, reads, st = synth.window_and_reads(16)
    ref2, reads2, st2 = synth.window_and_reads(16)
    assert (ref == ref2).all() and (reads == reads2).all() and st == st2
    assert ref.tolist()[:8] == [2, 0, 3, 1, 0, 0, 0, 3] and reads.min() >= 0 and reads.max() <= 3
