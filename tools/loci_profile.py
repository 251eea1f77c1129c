"""Where does the host time of the many-loci stream go?  cProfile over bench_modes.run_many_loci's loop (12 500 loci x 96 jobs)."""
import cProfile
import os
import pstats
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import indelpost_amd as ip
import bench_modes

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 6
bench_modes.run_many_loci(ip, (3, 2), 0, 4, 2)                  # warm
pr = cProfile.Profile()
pr.enable()
rec = bench_modes.run_many_loci(ip, (3, 2), 0, 4, steps)
pr.disable()
print({k: rec[k] for k in ("value", "ms_per_step", "host_ms_per_step", "one_list_at_a_time")})
pstats.Stats(pr).sort_stats("tottime").print_stats(28)
