#!/usr/bin/env python3
"""Sustained-clock evidence (run on the GPU box through gpurun; writes a small summary, deletes the raw rocprofv3 output):

    python3 tools/sustained_clock.py <workload> <steps> <outdir under gpurun_out>

1. bench.py --workload W --steps 20          (the usual short run)
2. bench.py --workload W --steps <steps>     (>= 5 s timed)
3. rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace -- python3 bench.py --workload W --steps <steps/4>: per launch of the dominant
   kernel, effective clock = GRBM_GUI_ACTIVE / 8 XCDs / launch duration (MI355X_MICROARCH.md, DVFS give-back), early vs late launches.
"""
import csv
import glob
import json
import os
import shutil
import subprocess
import sys

w, steps, out = sys.argv[1], int(sys.argv[2]), os.path.join("gpurun_out", sys.argv[3])
os.makedirs(out, exist_ok=True)
base = ["python3", "bench.py", "--workload", w, "--no-subconfigs", "--no-cpu-baseline", "--no-single-stream", "--warmup", "3"]
env = dict(os.environ, TMPDIR="/tmp")


def bench(n):
    p = subprocess.run(base + ["--steps", str(n)], capture_output=True, text=True, env=env)
    line = [l for l in p.stdout.splitlines() if l.startswith("{")][-1]
    d = json.loads(line)
    return {"steps": n, "ms_per_step": d["ms_per_step"], "value": d["value"], "timed_s": round(d["ms_per_step"] * n / 1e3, 2)}


res = {"workload": w, "short": bench(20), "long": bench(steps)}
res["long_vs_short_ms_per_step"] = round(res["long"]["ms_per_step"] / res["short"]["ms_per_step"], 4)
raw = os.path.join(out, "raw_" + w)
psteps = max(40, steps // 4)
subprocess.run(["rocprofv3", "--pmc", "GRBM_GUI_ACTIVE", "--kernel-trace", "--output-format", "csv", "-d", raw, "--"] + base + ["--steps", str(psteps)],
               capture_output=True, text=True, env=env)
rows = []
for f in glob.glob(os.path.join(raw, "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == "GRBM_GUI_ACTIVE":
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], float(r["Counter_Value"])))
rows.sort()
by = {}
for a, b, n, v in rows:
    by.setdefault(n, []).append((a, b, v))
# dominant kernel by total time; only launches that did real work (>= half the kernel's longest launch)
dom = max(by, key=lambda n: sum(b - a for a, b, _ in by[n]))
L = [(a, b, v) for a, b, v in by[dom] if (b - a) >= 0.5 * max(y - x for x, y, _ in by[dom])]
clk = [v / 8.0 / (b - a) for a, b, v in L]            # cycles per ns = GHz
q = max(1, len(clk) // 4)
res["profiled"] = {"steps": psteps, "dominant_kernel": dom.split("(")[0], "launches": len(clk),
                   "effective_clock_ghz_first_quarter": round(sum(clk[:q]) / q, 3), "effective_clock_ghz_last_quarter": round(sum(clk[-q:]) / q, 3),
                   "effective_clock_ghz_mean": round(sum(clk) / len(clk), 3),
                   "launch_ms_first_quarter": round(sum(b - a for a, b, _ in L[:q]) / q / 1e6, 4),
                   "launch_ms_last_quarter": round(sum(b - a for a, b, _ in L[-q:]) / q / 1e6, 4)}
shutil.rmtree(raw, ignore_errors=True)
json.dump(res, open(os.path.join(out, "sustained_%s.json" % w), "w"), indent=1)
print(json.dumps(res))
