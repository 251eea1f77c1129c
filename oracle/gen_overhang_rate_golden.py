"""Golden vectors for check_overhangs' DEFAULT splice-rate threshold (TEST INFRASTRUCTURE, build container only).

pileup.pyx:435 declares `check_overhangs(pileup, splice_rate=0.2)` and its only caller (varaln.pyx:260) passes no value, so the
default decides which loci get the overhang filter.  The driver scenarios of gen_driver_golden.py happen to have junctional support
well above 20 % or none at all; these cases sweep the support rate through the threshold on minimal read dicts (the function touches
`intron_pattern`, `is_covering`, `covering_subread`, `aln_start`, `aln_end` only).  As in the other generators the reference's
function TEXT is read from /root/reference at generation time and executed as it stands; only inputs and outputs are written.

    python oracle/gen_overhang_rate_golden.py        -> tests/golden/overhang_rate_cases.json
"""
import json
import os
import random
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from gen_decoder_golden import function_text, REF            # noqa: E402
from gen_variant_golden import strip_cython                  # noqa: E402


def load():
    ns = {}
    for fname, names in (("utilities.pyx", ["most_common"]), ("pileup.pyx", ["check_overhangs", "is_junctional", "is_overhang"])):
        for n in names:
            exec(compile(strip_cython(function_text(os.path.join(REF, fname), n)), "<%s:%s>" % (fname, n), "exec"), ns)
    return ns


def make_pileup(rng, n_reads, n_spliced, n_other, n_overhang, intron):
    """n_spliced reads carry `intron`, n_other a different intron, n_overhang unspliced reads hang over the intron's left edge,
    the rest are unspliced covering reads elsewhere; a few non-covering unspliced reads (not junctional) ride along"""
    a, b = intron
    reads = []
    for k in range(n_reads):
        if k < n_spliced:
            r = {"intron_pattern": (a, b), "is_covering": True, "covering_subread": (a - 60, a - 1), "aln_start": a - 60, "aln_end": b + 40}
        elif k < n_spliced + n_other:
            r = {"intron_pattern": (a + 7, b + 300), "is_covering": True, "covering_subread": (a - 50, a + 6), "aln_start": a - 50, "aln_end": b + 350}
        elif k < n_spliced + n_other + n_overhang:
            s = a - rng.randrange(20, 90)
            r = {"intron_pattern": (0, 0), "is_covering": True, "covering_subread": (s, a + rng.randrange(2, 12)), "aln_start": s, "aln_end": a + 12}
        else:
            s = b + rng.randrange(5, 50)
            r = {"intron_pattern": (0, 0), "is_covering": True, "covering_subread": (s, s + 100), "aln_start": s, "aln_end": s + 100}
        reads.append(r)
    for _ in range(rng.randrange(0, 4)):
        reads.append({"intron_pattern": (0, 0), "is_covering": False, "covering_subread": None, "aln_start": a - 300, "aln_end": a - 200})
    rng.shuffle(reads)
    return reads


def main():
    F = load()
    rng = random.Random(20261005)
    cases = []
    for n_reads in (20, 40, 100):
        for pct in (0, 5, 10, 12, 15, 18, 19, 20, 21, 25, 30, 50):
            for n_over in (0, 3):
                n_sp = (n_reads * pct) // 100
                pl = make_pileup(rng, n_reads, n_sp, 1 if pct >= 15 else 0, n_over, (5000, 5400))
                ans = F["check_overhangs"](pl)
                junctional = [r for r in pl if F["is_junctional"](r)]
                out = None if ans is None else {"intron": list(ans[0]), "overhangs": [i for i, r in enumerate(pl) if any(r is o for o in ans[1])]}
                cases.append({"pileup": [[list(r["intron_pattern"]), r["is_covering"], list(r["covering_subread"]) if r["covering_subread"] else None,
                                          r["aln_start"], r["aln_end"]] for r in pl],
                              "support": round(sum(1 for r in junctional if r["intron_pattern"] == (5000, 5400)) / max(1, len(junctional)), 4),
                              "expected": out})
    between = sum(1 for c in cases if 0.1 <= c["support"] < 0.2)
    path = os.path.join(ROOT, "tests", "golden", "overhang_rate_cases.json")
    json.dump({"cases": cases}, open(path, "w"), separators=(",", ":"))
    print("%d cases (%d with support in [0.1, 0.2), %d answered None) -> %s" % (len(cases), between, sum(1 for c in cases if c["expected"] is None), path))


if __name__ == "__main__":
    main()
