"""many-loci record only (bench_modes.run_many_loci), optionally without the threaded submit: python tools/loci_quick.py [steps] [threaded 0/1]"""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import indelpost_amd as ip
from indelpost_amd import batch
import bench_modes
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 8
if len(sys.argv) > 2 and sys.argv[2] == "0":
    _init = batch.MultiStreamAligner.__init__
    def init(self, *a, **k):
        _init(self, *a, **k)
        self.threaded_submit = False
    batch.MultiStreamAligner.__init__ = init
bench_modes.run_many_loci(ip, (3, 2), 0, 4, 2)
rec = bench_modes.run_many_loci(ip, (3, 2), 0, 4, steps)
print(json.dumps({k: rec[k] for k in ("value", "ms_per_step", "form", "align_loci_stream", "two_aligners_one_thread", "host_ms_per_step", "one_list_at_a_time")}))
