#!/usr/bin/env python3
"""Turn the rocprofv3 output of tools/profile_round.sh into the summaries committed under profiles/.

  python profiles/summarize_pmc.py <gpurun_out/rNN dir> <tag, e.g. r02> <workload> [<workload> ...]

per workload w (2b, 2a, 4, 5):
* <tag>_<w>_kernel_stats.csv     : rocprofv3's kernel_stats.csv (--kernel-trace --stats), 4 streams
* <tag>_<w>_kernel_stats_1stream.csv : the same with one stream (a launch = the whole batch)
* <tag>_<w>_pmc.md               : per kernel FETCH_SIZE / WRITE_SIZE per launch (separate passes)
* <tag>_<w>_sq_counters.md       : SQ_* counters per launch, one stream, with VALU instructions per tile column
* pmc_latest.json / sq_latest.json : {workload: {bench kernel class: ...}} read by bench.py (roofline.traffic / valu_issue)

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

HERE = os.path.dirname(os.path.abspath(__file__))


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
    """rocprof kernel symbol -> kernel class name of bench.py (ipx_kernel_class_name)"""
    m = re.match(r"void k_dp_pass<(\d+), (\d+), (true|false), (true|false), (\d+)((?:, (?:true|false))*)>", k)
    if m:
        w, s, rev, exact, stage = int(m.group(1)), int(m.group(2)), m.group(3) == "true", m.group(4) == "true", int(m.group(5))
        flags = [x == "true" for x in re.findall(r"true|false", m.group(6))]        # selector profile, half precision, two lanes per GPU lane
        if len(flags) >= 3 and flags[2]:                                           # the 8-bit lower-bound stage in the 8-lane layout
            return "dp_byte_low_s%d" % (s // 2)
        if w == 16:
            base = "dp_byte_rev" if rev else ("dp_byte_exact", "dp_byte_low", "dp_byte_high")[stage]
        else:
            base = "dp_word_rev" if rev else "dp_word_fwd"
        return "%s_%s" % (base, ("s%d" % s) if exact else "long")
    m = re.match(r"void k_dp_skew<(\d+), (true|false)(?:, (\d+|true|false))?(?:, (\d+))?>", k)     # (r04: a fourth argument, lanes per read: 8, or 32 = the latency tier)
    if m:
        bh = {"true": 1, "false": 0, None: 0}.get(m.group(3), None)
        bh = int(m.group(3)) if bh is None else bh
        rev = m.group(2) == "true"
        if bh == 1:                                                # the 8-bit upper-bound stage at segLen8 = S / 2 (bracket flow)
            return "dp_byte_high_s%d" % (int(m.group(1)) // 2)
        if bh == 2:                                                # the plain recurrence in the 8-bit dialect (plain-first flow)
            return "%s_s%d" % ("dp_byte_rev_plain" if rev else "dp_byte_plain", int(m.group(1)) // 2)
        return "%s_s%s" % ("dp_word_rev" if rev else "dp_word_fwd", m.group(1))
    m = re.match(r"void k_tb_fast<(\d+)>", k)
    if m:
        return "traceback_fast_bw%s" % m.group(1)
    m = re.match(r"void k_tb_diag<(\d+)>", k)
    if m:
        return "traceback_diag%s" % m.group(1)
    for a, b in (("k_tb_coop", "traceback_coop"), ("k_plan", "plan"), ("k_tb_list", "tb_list"),
                 ("k_prove_overflow", "prove_overflow"), ("k_init", "init"), ("void k_prove_plain<false>", "prove_plain_fwd"),
                 ("void k_prove_plain<true>", "prove_plain_rev")):
        if k.startswith(a):
            return b
    return k.split("(")[0]


SOURCE_TAG = {}          # filled by __main__: tag and the fingerprint of the kernel sources the profile was taken on


def merge_json(name, workload, data):
    path = os.path.join(HERE, name)
    try:
        cur = json.load(open(path))
    except (OSError, ValueError):
        cur = {}
    data = dict(data)
    data.update(SOURCE_TAG)   # bench.py quotes these figures only while the kernel sources still have this fingerprint
    cur[workload] = data
    json.dump(cur, open(path, "w"), indent=1, sort_keys=True)


def bench_line(path):
    try:
        for line in open(path):
            if line.startswith("{"):
                return json.loads(line)
    except OSError:
        pass
    return {}


def one(root, tag, w):
    d = os.path.join(root, w)
    for sub, suffix in (("kt", ""), ("kt1", "_1stream")):
        for f in glob.glob(os.path.join(d, sub, "**", "*kernel_stats.csv"), recursive=True):
            shutil.copy(f, os.path.join(HERE, "%s_%s_kernel_stats%s.csv" % (tag, w, suffix)))
    f, wr = load(os.path.join(d, "fetch"), "FETCH_SIZE"), load(os.path.join(d, "write"), "WRITE_SIZE")
    out, rows = {}, []
    for k in sorted(f, key=lambda k: -f[k][1]):
        n, v = f[k]
        wn, wv = wr.get(k, [0, 0.0])
        fk, wk = v / n, (wv / wn if wn else 0.0)
        name = bench_name(k)
        rows.append((k, name, n, fk, wk))
        e = out.setdefault(name, {"hbm_bytes_per_launch": 0, "launches": 0})
        if n > e["launches"]:
            e.update(hbm_bytes_per_launch=int((2 * fk + wk) * 1024), fetch_kb_raw=round(fk, 1), write_kb=round(wk, 1), launches=n)
    with open(os.path.join(HERE, "%s_%s_pmc.md" % (tag, w)), "w") as o:
        o.write("# %s, workload %s: rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, 4 streams), per launch\n\n" % (tag, w))
        o.write("| kernel | bench class | launches | FETCH_SIZE KB (raw) | WRITE_SIZE KB | HBM bytes = (2F+W)*1024 |\n|---|---|---|---|---|---|\n")
        for k, name, n, fk, wk in rows[:24]:
            o.write("| `%s` | %s | %d | %.1f | %.1f | %.0f |\n" % (k[:64], name, n, fk, wk, (2 * fk + wk) * 1024))
    merge_json("pmc_latest.json", w, out)

    SQ = ["SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_WAVE_CYCLES", "SQ_ACTIVE_INST_ANY", "SQ_WAIT_ANY",
          "SQ_WAIT_INST_ANY", "SQ_BUSY_CYCLES"]
    cols = {c: load(os.path.join(d, "sq"), c) for c in SQ}
    line = bench_line(os.path.join(root, "%s.sq.log" % w))
    per_launch = dict((line.get("roofline") or {}).get("dp_alignments_per_launch") or {})
    ks = sorted(cols["SQ_INSTS_VALU"], key=lambda k: -cols["SQ_INSTS_VALU"][k][1])
    with open(os.path.join(HERE, "%s_%s_sq_counters.md" % (tag, w)), "w") as o:
        o.write("# %s, workload %s: rocprofv3 --pmc SQ_* per launch (bench.py --workload %s --streams 1: a launch = the whole batch)\n\n" % (tag, w, w))
        o.write("SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_ANY count quad-cycles (MI355X_MICROARCH.md).  `aln/launch` = alignments one launch\n"
                "processed (planner tile counts); `VALU/aln` = wave-instructions per alignment.\n\n")
        o.write("| kernel | bench class | launches | aln/launch | VALU/aln | " + " | ".join(c[3:] for c in SQ) + " |\n|---|---|---|---|---|" + "---|" * len(SQ) + "\n")
        for k in ks[:16]:
            n = cols["SQ_INSTS_VALU"][k][0]
            name = bench_name(k)
            alias = [a for a in per_launch if a == name or a.replace("dp_word_first", "dp_word_fwd").replace("dp_byte_check", "dp_byte_low").replace("dp_byte_low2", "dp_byte_low") == name]
            u = max([per_launch[a] for a in alias], default=0)
            valu = cols["SQ_INSTS_VALU"][k][1] / max(1, n)
            o.write("| `%s` | %s | %d | %s | %s | %s |\n" % (k[:60], name, n, u or "", ("%.0f" % (valu / u)) if u else "",
                                                        " | ".join("%.3g" % (cols[c].get(k, [1, 0])[1] / max(1, cols[c].get(k, [1, 0])[0])) for c in SQ)))
    out = {"kernels": {}, "alignments_per_launch": {}}
    for k in ks:
        n, v = cols["SQ_INSTS_VALU"][k]
        name = bench_name(k)
        e = out["kernels"].setdefault(name, {"valu_insts_per_launch": 0})
        if int(v / max(1, n)) > e["valu_insts_per_launch"]:
            e["valu_insts_per_launch"] = int(v / max(1, n))
            alias = [a for a in per_launch if a.replace("dp_word_first", "dp_word_fwd").replace("dp_byte_check", "dp_byte_low").replace("dp_byte_low2", "dp_byte_low") == name]
            if alias:
                out["alignments_per_launch"][name] = max(per_launch[a] for a in alias)
    out["note"] = "bench.py --workload %s --steps 2 --warmup 1 --streams 1" % w
    merge_json("sq_latest.json", w, out)
    print("workload %s: wrote %s_%s_{kernel_stats,pmc,sq_counters}, merged pmc_latest.json / sq_latest.json" % (w, tag, w))


if __name__ == "__main__":
    root, tag = sys.argv[1], sys.argv[2]
    try:                                                   # written on the GPU box by tools/profile_round.sh, from the sources that were profiled
        sha = open(os.path.join(root, "kernel_source_sha16.txt")).read().strip()
    except OSError:
        sys.path.insert(0, os.path.dirname(HERE))
        import bench
        sha = bench.kernel_source_sha16()
    SOURCE_TAG.update({"_tag": tag, "_kernel_source_sha16": sha})
    for w in sys.argv[3:]:
        one(root, tag, w)
    bl = bench_line(os.path.join(root, "bench.log"))
    if bl:
        json.dump(bl, open(os.path.join(HERE, "%s_bench_line.json" % tag), "w"))
        print("wrote %s_bench_line.json" % tag)
