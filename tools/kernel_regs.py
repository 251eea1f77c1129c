"""Registers, spills and private memory of every kernel in the built objects (indelpost_amd/csrc/build/*.o): unbundles the gfx950
code object of each translation unit and reads its metadata notes.  Usage: python tools/kernel_regs.py [substring-of-kernel-name]"""
import glob
import os
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin/"
HERE = os.path.dirname(os.path.abspath(__file__))


def kernels_of(obj, tmp):
    out = os.path.join(tmp, os.path.basename(obj) + ".co")
    fat = os.path.join(tmp, os.path.basename(obj) + ".fatbin")
    subprocess.run([LLVM + "llvm-objcopy", "--dump-section", ".hip_fatbin=" + fat, obj, os.path.join(tmp, "unused.o")], check=True)
    subprocess.run([LLVM + "clang-offload-bundler", "--unbundle", "--type=o", "--input=" + fat,
                    "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--output=" + out], check=True, stderr=subprocess.DEVNULL)
    txt = subprocess.run([LLVM + "llvm-readelf", "--notes", out], capture_output=True, text=True).stdout
    cur, rows = {}, []
    for line in txt.split("\n"):
        line = line.strip().lstrip("- ").strip()
        for k in (".name", ".vgpr_count", ".vgpr_spill_count", ".sgpr_count", ".private_segment_fixed_size", ".agpr_count"):
            if line.startswith(k + ":"):
                cur[k] = line.split(":", 1)[1].strip()
        if line.startswith(".wavefront_size:"):
            rows.append(dict(cur))
            cur = {}
    return rows


def main():
    want = sys.argv[1] if len(sys.argv) > 1 else ""
    rows = []
    with tempfile.TemporaryDirectory() as tmp:
        for obj in sorted(glob.glob(os.path.join(HERE, "..", "indelpost_amd", "csrc", "build", "*.o"))):
            rows += kernels_of(obj, tmp)
    names = subprocess.run(["c++filt"], input="\n".join(r.get(".name", "?") for r in rows), capture_output=True, text=True).stdout.split("\n")
    print("%-80s %5s %5s %6s %7s" % ("kernel", "vgpr", "agpr", "spill", "private"))
    for r, n in sorted(zip(rows, names), key=lambda x: x[1]):
        n = n.replace("void ", "").split("(")[0]
        if want in n:
            print("%-80s %5s %5s %6s %7s" % (n[:80], r.get(".vgpr_count"), r.get(".agpr_count", "0"), r.get(".vgpr_spill_count"), r.get(".private_segment_fixed_size")))


if __name__ == "__main__":
    main()
