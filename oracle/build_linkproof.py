"""Compile/link proof of the C-level drop-in (TEST INFRASTRUCTURE, build container only).

Cythonizes the reference's OWN binding, /root/reference/indelpost/sswpy.pyx, from where it lies, against
include/indelpost_hip.h through a one-line `ssw.h` shim, and links the extension to indelpost_amd/libindelpost_hip.so
instead of ssw.c -- the patch INTEGRATION.md section 2 describes.  Every output (shim, generated C, objects, the
extension module) goes under oracle/_ref/linkproof/ (git-ignored, gpurun-ignored: it never travels).

    python oracle/build_linkproof.py        ->  oracle/_ref/linkproof/refbind/sswpy.*.so
"""
import os
import sys

from setuptools import Extension, setup
from Cython.Build import cythonize

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
REF = os.environ.get("IPX_REFERENCE", "/root/reference/indelpost")
OUT = os.path.join(HERE, "_ref", "linkproof")

if __name__ == "__main__":
    if not os.path.exists(os.path.join(REF, "sswpy.pyx")):
        sys.exit("reference sources not present at %s" % REF)
    os.makedirs(os.path.join(OUT, "shim"), exist_ok=True)
    os.makedirs(os.path.join(OUT, "refbind"), exist_ok=True)
    with open(os.path.join(OUT, "shim", "ssw.h"), "w") as f:       # the one-line shim of INTEGRATION.md section 2
        f.write('#include "indelpost_hip.h"\n')
    open(os.path.join(OUT, "refbind", "__init__.py"), "a").close()
    lib_dir = os.path.join(ROOT, "indelpost_amd")
    ext = Extension("refbind.sswpy", [os.path.join(REF, "sswpy.pyx")],
                    include_dirs=[os.path.join(OUT, "shim"), os.path.join(ROOT, "include")],   # NOT the reference's directory: its ssw.h must not be found
                    libraries=["indelpost_hip"], library_dirs=[lib_dir], runtime_library_dirs=[lib_dir],
                    extra_link_args=["-Wl,-rpath-link,/opt/rocm/lib"],
                    extra_compile_args=["-Wno-unused-function"])
    os.chdir(OUT)
    setup(name="refbind", script_args=["build_ext", "--build-lib", OUT, "--build-temp", os.path.join(OUT, "build")],
          ext_modules=cythonize([ext], language_level=3, build_dir=os.path.join(OUT, "cy")))
