"""bench.py on another build of the library (an experiment's variant under tests/variants/): python tools/bench_with_lib.py <lib.so> <bench.py arguments>"""
import os
import runpy
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from indelpost_amd import _lib

_lib.LIB_PATH = os.path.abspath(sys.argv[1])
_lib.needs_build = lambda: False
sys.argv = ["bench.py"] + sys.argv[2:]
runpy.run_path(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bench.py"), run_name="__main__")
