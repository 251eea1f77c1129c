"""Build the reference's own Python binding (sswpy.pyx + ssw.c) for golden-vector generation.

TEST INFRASTRUCTURE, build-container only.  Sources are compiled from where they lie under
/root/reference (nothing is copied into the repo); every output -- the cythonized C file, objects
and the extension module -- goes under oracle/_ref/ (git-ignored).  The module is used only by
oracle/gen_golden.py, here; it is never imported by tests, smoke() or bench.py.
"""
import os
import sys

from setuptools import Extension, setup
from Cython.Build import cythonize

HERE = os.path.dirname(os.path.abspath(__file__))
REF = os.environ.get("IPX_REFERENCE", "/root/reference/indelpost")
OUT = os.path.join(HERE, "_ref")

if __name__ == "__main__":
    if not os.path.exists(os.path.join(REF, "sswpy.pyx")):
        sys.exit("reference sources not present at %s" % REF)
    os.makedirs(OUT, exist_ok=True)
    ext = Extension("ref_sswpy_pkg.sswpy",
                    [os.path.join(REF, "sswpy.pyx"), os.path.join(REF, "ssw.c")],
                    include_dirs=[REF], extra_compile_args=["-Wno-unused-function"])
    os.makedirs(os.path.join(OUT, "ref_sswpy_pkg"), exist_ok=True)
    open(os.path.join(OUT, "ref_sswpy_pkg", "__init__.py"), "a").close()
    os.chdir(OUT)
    setup(name="ref_sswpy", script_args=["build_ext", "--build-lib", OUT, "--build-temp",
                                         os.path.join(OUT, "build")],
          ext_modules=cythonize([ext], language_level=3, build_dir=os.path.join(OUT, "cy")))
