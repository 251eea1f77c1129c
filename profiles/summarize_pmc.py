#!/usr/bin/env python3
"""Turn rocprofv3 output directories into the summaries committed under profiles/.

  python profiles/summarize_pmc.py <kernel-trace-dir> <fetch-pmc-dir> <write-pmc-dir> <tag>

* <tag>_kernel_stats.csv  : copy of rocprofv3's kernel_stats.csv (--kernel-trace --stats)
* <tag>_pmc.md            : per kernel FETCH_SIZE / WRITE_SIZE per launch
* pmc_latest.json         : {bench kernel class: {"hbm_bytes_per_launch": ...}} read by bench.py

HBM bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024: on gfx950 FETCH_SIZE tallies 64 B per 128 B request
(MI355X_MICROARCH.md, HBM section) -- exact for wide streaming reads, an upper bound for our narrow
loads; WRITE_SIZE is exact.
"""
import collections
import csv
import glob
import json
import os
import re
import shutil
import sys


def load(d, counter):
    """per kernel: [launches counted, sum] over the launches that did real work (>= 20 % of the kernel's
    largest launch: the same instantiation is also launched for passes that turn out empty)"""
    vals = collections.defaultdict(list)
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter:
                vals[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    agg = {}
    for k, v in vals.items():
        top = max(v)
        real = [x for x in v if x >= 0.2 * top] or v
        agg[k] = [len(real), sum(real)]
    return agg


def bench_name(k):
    m = re.match(r"void k_dp_pass<(\d+), (\d+), (true|false), (true|false), (true|false)(?:, (?:true|false))?>", k)
    if m:
        w, s, rev, exact, low = int(m.group(1)), int(m.group(2)), m.group(3) == "true", m.group(4) == "true", m.group(5) == "true"
        base = "dp_%s_%s" % ("byte" if w == 16 else "word", "rev" if rev else "fwd")
        return "%s_%s" % (base, ("s%d" % s) if exact else "long")
    m = re.match(r"void k_tb_fast<(\d+)>", k)
    if m:
        return "traceback_fast_bw%s" % m.group(1)
    for a, b in (("k_traceback", "traceback_tier0"), ("k_tb_coop", "traceback_tier1"), ("k_plan", "plan"),
                 ("k_tb_list", "tb_list"), ("k_prove_overflow", "prove_overflow"), ("k_init", "init")):
        if k.startswith(a):
            return b
    return k.split("(")[0]


def main(kt, fd, wd, tag):
    here = os.path.dirname(os.path.abspath(__file__))
    for f in glob.glob(os.path.join(kt, "**", "*kernel_stats.csv"), recursive=True):
        shutil.copy(f, os.path.join(here, "%s_kernel_stats.csv" % tag))
    f, w = load(fd, "FETCH_SIZE"), load(wd, "WRITE_SIZE")
    out, rows = {}, []
    for k in sorted(f, key=lambda k: -f[k][1]):
        n, v = f[k]
        wn, wv = w.get(k, [0, 0.0])
        fk, wk = v / n, (wv / wn if wn else 0.0)
        name = bench_name(k)
        rows.append((k, name, n, fk, wk))
        e = out.setdefault(name, {"hbm_bytes_per_launch": 0, "launches": 0})
        if n > e["launches"]:
            e.update(hbm_bytes_per_launch=int((2 * fk + wk) * 1024), fetch_kb_raw=round(fk, 1), write_kb=round(wk, 1), launches=n)
    with open(os.path.join(here, "%s_pmc.md" % tag), "w") as o:
        o.write("# %s: rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes), per launch\n\n" % tag)
        o.write("| kernel | bench class | launches | FETCH_SIZE KB (raw) | WRITE_SIZE KB | HBM bytes = (2F+W)*1024 |\n|---|---|---|---|---|---|\n")
        for k, name, n, fk, wk in rows[:24]:
            o.write("| `%s` | %s | %d | %.1f | %.1f | %.0f |\n" % (k[:64], name, n, fk, wk, (2 * fk + wk) * 1024))
    json.dump(out, open(os.path.join(here, "pmc_latest.json"), "w"), indent=1, sort_keys=True)
    print("wrote %s_pmc.md, pmc_latest.json (%d kernels)" % (tag, len(out)))


SQ = ["SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_WAVE_CYCLES", "SQ_ACTIVE_INST_ANY", "SQ_WAIT_ANY",
      "SQ_WAIT_INST_ANY", "SQ_BUSY_CYCLES"]


def sq_table(sd, tag, note):
    """<tag>_sq_counters.md from a `--pmc SQ_*` pass (per launch that did real work)"""
    here = os.path.dirname(os.path.abspath(__file__))
    cols = {c: load(sd, c) for c in SQ}
    ks = sorted(cols["SQ_INSTS_VALU"], key=lambda k: -cols["SQ_INSTS_VALU"][k][1])
    with open(os.path.join(here, "%s_sq_counters.md" % tag), "w") as o:
        o.write("# %s: rocprofv3 --pmc SQ_* per launch (%s)\n\n" % (tag, note))
        o.write("SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_ANY count quad-cycles (MI355X_MICROARCH.md).\n\n")
        o.write("| kernel | launches | " + " | ".join(c[3:] for c in SQ) + " |\n|---|---|" + "---|" * len(SQ) + "\n")
        for k in ks[:14]:
            n = cols["SQ_INSTS_VALU"][k][0]
            o.write("| `%s` | %d | %s |\n" % (k[:60], n, " | ".join("%.3g" % (cols[c].get(k, [1, 0])[1] / max(1, cols[c].get(k, [1, 0])[0])) for c in SQ)))
    # VALU wave-instructions per launch of each bench kernel class (single-stream pass: a launch = the whole batch)
    out = {}
    for k in ks:
        n, v = cols["SQ_INSTS_VALU"][k]
        e = out.setdefault(bench_name(k), {"valu_insts_per_launch": 0})
        e["valu_insts_per_launch"] = max(e["valu_insts_per_launch"], int(v / max(1, n)))
    json.dump({"note": note, "kernels": out}, open(os.path.join(here, "sq_latest.json"), "w"), indent=1, sort_keys=True)
    print("wrote %s_sq_counters.md, sq_latest.json" % tag)


if __name__ == "__main__":
    main(*sys.argv[1:5])
    if len(sys.argv) > 5:
        sq_table(sys.argv[5], sys.argv[4], sys.argv[6] if len(sys.argv) > 6 else "bench.py --steps 2 --warmup 1 --streams 1: one stream, so a launch = 1 M reads")
