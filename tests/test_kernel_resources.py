"""Register budgets of the kernels whose occupancy the measured numbers rest on, read from the built code objects (no GPU needed).
A kernel that crosses a waves-per-SIMD step (128 / 168 registers) or starts to spill loses 10-20 % without any test failing --
r03 saw it twice (k_prove_plain 164 -> 248 registers: 2a 75 -> 61.5 M aln/s); this is the tripwire."""
import glob
import os
import sys
import tempfile

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))

BUDGET = {  # kernel name prefix: (max registers, max spilled registers)
    "k_dp_skew<19, false, 0, 8>": (168, 0),        # headline forward pass: three waves per SIMD
    "k_dp_skew<19, true, 0, 8>": (168, 0),
    "k_dp_skew<13, false, 0, 8>": (128, 0),        # four waves
    "k_dp_skew<20, false, 2, 8>": (168, 0),        # plain 8-bit recurrence of 150 bp reads
    "k_dp_skew<25, false, 0, 8>": (168, 40),       # three waves at the price of a few spills
    "k_dp_skew<32, false, 0, 8>": (256, 0),
    "k_dp_skew<5, false, 0, 32>": (128, 0),     # r04 latency tier: 150 bp reads at 32 lanes per read
    "k_dp_skew<5, true, 0, 32>": (128, 0),
    "k_tb_diag<16>": (128, 0),                  # r04 anti-diagonal traceback tiers
    "k_tb_diag<32>": (128, 0),
    "k_tb_diag<64>": (128, 0),
    "k_prove_overflow_diag": (64, 0),
    "k_prove_plain<false>": (168, 0),
    "k_prove_plain<true>": (128, 0),
    "k_prove_overflow": (128, 0),
    "k_tb_coop": (128, 0),
    "k_tb_fast<2>": (128, 0),
    "k_dp_wide<16, false>": (256, 0),           # r04 reads beyond 504 bp, one wavefront per read: two waves per SIMD for up to 1 024 rows ...
    "k_dp_wide<32, false>": (256, 0),           # ... one wave, no accumulation registers, up to 2 048 ...
    "k_dp_wide<64, false>": (512, 400),         # ... and the whole register file of a SIMD at 4 096 (what does not fit the 256 architectural registers lives in accumulation registers, not in memory)
}


def test_register_budgets_of_the_hot_kernels():
    import kernel_regs
    objs = sorted(glob.glob(os.path.join(ROOT, "indelpost_amd", "csrc", "build", "*.o")))
    if not objs or not os.path.exists(kernel_regs.LLVM + "clang-offload-bundler"):
        pytest.skip("no built objects / no ROCm tools here")
    import subprocess
    rows = []
    with tempfile.TemporaryDirectory() as tmp:
        for o in objs:
            rows += kernel_regs.kernels_of(o, tmp)
    names = subprocess.run(["c++filt"], input="\n".join(r.get(".name", "?") for r in rows), capture_output=True, text=True).stdout.split("\n")
    seen, private = {}, {}
    for r, n in zip(rows, names):
        n = n.replace("void ", "").split("(")[0]
        seen[n] = (int(r.get(".vgpr_count", 0)), int(r.get(".vgpr_spill_count", 0)))
        private[n] = int(r.get(".private_segment_fixed_size", 0))
    for k, (vmax, smax) in BUDGET.items():
        assert k in seen, "kernel %s not found in the built objects" % k
        v, sp = seen[k]
        if k.startswith("k_dp_wide"):
            assert private.get(k, 0) == 0, "%s keeps %d bytes per lane in private memory" % (k, private[k])
        assert v <= vmax and sp <= smax, "%s: %d registers, %d spilled (budget %d / %d)" % (k, v, sp, vmax, smax)


def test_called_tier_bodies_find_the_descriptors_where_they_read_them():
    """k_dp_pass_tier's called bodies read IpxBatch / IpxPlan from the kernel-argument segment at offset 0 and align_up(sizeof(IpxBatch), 8)
    (csrc/ipx_kernels.h IPX_CALLEE_DESC_LOCALS): the code object's argument table must say the same"""
    import subprocess
    import kernel_regs
    obj = os.path.join(ROOT, "indelpost_amd", "csrc", "build", "ipx_dp_v.o")
    if not os.path.exists(obj) or not os.path.exists(kernel_regs.LLVM + "clang-offload-bundler"):
        pytest.skip("no built objects / no ROCm tools here")
    with tempfile.TemporaryDirectory() as tmp:
        fat, co = os.path.join(tmp, "x.fatbin"), os.path.join(tmp, "x.co")
        subprocess.run([kernel_regs.LLVM + "llvm-objcopy", "--dump-section", ".hip_fatbin=" + fat, obj, os.path.join(tmp, "unused.o")], check=True)
        subprocess.run([kernel_regs.LLVM + "clang-offload-bundler", "--unbundle", "--type=o", "--input=" + fat,
                        "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--output=" + co], check=True, stderr=subprocess.DEVNULL)
        txt = subprocess.run([kernel_regs.LLVM + "llvm-readelf", "--notes", co], capture_output=True, text=True).stdout
    src = "#include <stdio.h>\n#include \"%s\"\nint main(){printf(\"%%zu %%zu\", sizeof(IpxBatch), sizeof(IpxPlan));}" % os.path.join(ROOT, "indelpost_amd", "csrc", "ipx_types.h")
    with tempfile.TemporaryDirectory() as tmp:
        open(os.path.join(tmp, "s.cpp"), "w").write(src)
        subprocess.run(["g++", "-o", os.path.join(tmp, "s"), os.path.join(tmp, "s.cpp")], check=True)
        sb, sp = (int(x) for x in subprocess.run([os.path.join(tmp, "s")], capture_output=True, text=True).stdout.split())
    import re
    blocks = txt.split(".args:")[1:]
    assert blocks
    for blk in blocks:
        offs = [(int(o), int(z)) for o, z in re.findall(r"\.offset:\s+(\d+)\s+\.size:\s+(\d+)", blk)[:2]]
        assert offs == [(0, sb), ((sb + 7) & ~7, sp)], offs
