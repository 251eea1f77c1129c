"""The reference's own Cython binding compiles against include/indelpost_hip.h and links to libindelpost_hip.so
(INTEGRATION.md section 2): build container only -- it needs /root/reference and Cython, and nothing it builds travels."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF_PYX = "/root/reference/indelpost/sswpy.pyx"
OUT = os.path.join(ROOT, "oracle", "_ref", "linkproof")


@pytest.mark.skipif(not os.path.exists(REF_PYX), reason="reference sources not present (GPU box)")
def test_reference_sswpy_links_to_the_hip_library(hip_lib):
    pytest.importorskip("Cython")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "oracle", "build_linkproof.py")], capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-3000:]
    mods = [f for f in os.listdir(os.path.join(OUT, "refbind")) if f.startswith("sswpy") and f.endswith(".so")]
    assert mods, "extension module not built"
    so = os.path.join(OUT, "refbind", mods[0])
    # the four entry points (and the cigar table) are UNDEFINED in the binding and come from our library
    und = subprocess.run(["nm", "-D", "--undefined-only", so], capture_output=True, text=True).stdout
    for sym in ("ssw_init", "ssw_align", "align_destroy", "init_destroy"):
        assert (" U " + sym) in und, sym
    needed = subprocess.run(["readelf", "-d", so], capture_output=True, text=True).stdout
    assert "libindelpost_hip.so" in needed
    # and it runs: the reference's SSW class drives our ssw_init / ssw_align (no GPU here -> NULL -> the binding's ValueError)
    code = ("import sys; sys.path.insert(0, %r); from refbind.sswpy import SSW; a = SSW(3, 2); a.setReference('ACGTACGTTTGACCAGT'); "
            "a.setRead('ACGTAGTTTGACCAGT')\n"
            "try:\n    r = a.align(3, 1); print('ALIGNED', tuple(r))\n"
            "except ValueError as e:\n    print('VALUEERROR', e)\n" % OUT)
    q = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert q.returncode == 0, q.stderr[-2000:]
    if os.path.exists("/dev/kfd"):
        assert "ALIGNED ('5M1D11M', 45, 3, 0, 16, 0, 15)" in q.stdout
    else:
        assert "VALUEERROR" in q.stdout and "no CPU fallback" in q.stderr
