"""Golden vectors for the read filters in front of the realignment (TEST INFRASTRUCTURE, build container only).

findall_mismatches (localn.pyx:71-136) and is_worth_realn (localn.pyx:139-220) decide which reads of a pileup are realigned at all.
The driver scenarios of gen_driver_golden.py exercise them on well-formed simulator reads; these cases go through the CIGAR shapes
the index arithmetic of utilities.pyx:429-503 (`split`) treats in its own way -- a soft clip behind a hard clip (not trimmed), an
insertion right behind the clip, =/X/P tokens, introns, reference bases in lower case, aln_end one off -- and through targets whose
shiftable span reaches over a read's end.  As in the other generators the reference's function TEXT is read from /root/reference at
generation time and executed as it stands; only inputs and outputs are written.

    python oracle/gen_realn_filter_golden.py        -> tests/golden/realn_filter_cases.json
"""
import json
import os
import random
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from gen_decoder_golden import function_text, REF            # noqa: E402
from gen_variant_golden import strip_cython, _IntNumpy       # noqa: E402  (_IntNumpy: `split`'s np.zeros as integer arrays, see there)

TOKEN = re.compile(r"[0-9]+[MIDNSHPX=]")


def load():
    ns = {"cigar_ptrn": TOKEN, "np": _IntNumpy}
    for fname, names in (("utilities.pyx", ["get_mapped_subreads", "split"]), ("localn.pyx", ["findall_mismatches", "is_worth_realn"])):
        for n in names:
            exec(compile(strip_cython(function_text(os.path.join(REF, fname), n)), "<%s:%s>" % (fname, n), "exec"), ns)
    return ns


class Pos:
    def __init__(self, pos):
        self.pos = pos


class Target:
    """what is_worth_realn touches of a Variant: pos, ref, is_ins, generate_equivalents() -> objects with .pos"""

    def __init__(self, pos, ref, alt, shifts):
        self.pos, self.ref, self.alt, self.shifts, self.is_ins = pos, ref, alt, shifts, len(alt) > len(ref)

    def generate_equivalents(self):
        return [Pos(p) for p in self.shifts]


def random_read(rng, odd):
    length = rng.randint(20, 160)
    ops, left = [], length
    if odd and rng.random() < 0.15:
        ops.append((rng.randint(1, 5), "H"))
    if rng.random() < 0.3:
        n = rng.randint(1, min(20, left - 5))
        ops.append((n, "S"))
        left -= n
    tail = 0
    if rng.random() < 0.3 and left > 10:
        tail = rng.randint(1, min(20, left - 5))
        left -= tail
    first = True
    while left > 0:
        kind = rng.random()
        if (first and not (odd and kind > 0.9)) or kind < 0.5:
            n = rng.randint(1, left)
            ops.append((n, rng.choice("MMMM=X") if odd else "M"))
            left -= n
        elif kind < 0.65 or (first and odd):
            n = rng.randint(1, min(8, left))
            ops.append((n, "I"))
            left -= n
        elif kind < 0.8:
            ops.append((rng.randint(1, 8), "D"))
        elif kind < 0.9:
            ops.append((rng.randint(20, 300), "N"))
        elif odd:
            ops.append((rng.randint(1, 3), "P"))
        first = False
    if tail:
        ops.append((tail, "S"))
    if odd and rng.random() < 0.1:
        ops.append((3, "H"))
    cigar = "".join("%d%s" % o for o in ops)
    toks = TOKEN.findall(cigar)
    seq = "".join(rng.choice("ACGT") for _ in range(length))
    quals = [rng.randint(2, 40) for _ in range(length)]
    aln_start = rng.randint(1000, 1100)
    start_offset = int(toks[0][:-1]) if toks[0].endswith("S") else 0
    end_offset = int(toks[-1][:-1]) if toks[-1].endswith("S") else 0
    ref, at = [], 0
    for t in toks:
        n, op = int(t[:-1]), t[-1]
        if op in "M=X":
            for _ in range(n):
                b = seq[at] if at < length else "A"
                ref.append(b if rng.random() > 0.04 else rng.choice("ACGTacgtN"))
                at += 1
        elif op == "D":
            ref += [rng.choice("ACGT") for _ in range(n)]
        elif op in "IS":
            at += n
    ref = "".join(ref)
    span = sum(int(t[:-1]) for t in toks if t[-1] in "MDN=X")
    aln_end = aln_start + span - 1 + rng.choice([0, 0, 0, 1])
    cover = rng.choice([None, None, (aln_start + rng.randint(-5, 30), aln_start + rng.randint(31, 200))])
    return {"read_seq": seq, "read_qual": quals, "ref_seq": ref, "cigar_string": cigar, "cigar_list": toks, "aln_start": aln_start,
            "aln_end": aln_end, "start_offset": start_offset, "end_offset": end_offset, "is_reference_seq": seq == ref or rng.random() < 0.05,
            "covering_subread": cover, "I": [[aln_start + rng.randint(0, 150), "x"]] if rng.random() < 0.2 else [],
            "D": [[aln_start + rng.randint(0, 150), "y"]] if rng.random() < 0.2 else []}


def attempt(f, *args):
    try:
        return f(*args)
    except Exception as e:                                   # the reference's own failure (an empty left part: IndexError) is an output too
        return {"raises": type(e).__name__}


def main():
    F = load()
    rng = random.Random(20261006)
    reads = [random_read(rng, odd=k % 2 == 1) for k in range(700)]
    targets = []
    for _ in range(12):
        pos = rng.randint(990, 1250)
        if rng.random() < 0.5:
            ref, alt = "A" + "".join(rng.choice("ACGT") for _ in range(rng.randint(1, 6))), "A"
        else:
            ref, alt = "A", "A" + "".join(rng.choice("ACGT") for _ in range(rng.randint(1, 6)))
        shifts = sorted({pos, pos + rng.randint(0, 12), pos + rng.randint(0, 5)})
        targets.append({"pos": pos, "ref": ref, "alt": alt, "shifts": shifts, "qual_lim": rng.choice([23, 23, 10, 35])})
    out = []
    for r in reads:
        rec = {"read": r, "mismatches": {}, "worth": []}
        for trim in (0, 3):
            c = json.loads(json.dumps(r))
            res = attempt(F["findall_mismatches"], c, trim)
            rec["mismatches"][str(trim)] = res if isinstance(res, dict) and "raises" in res else [list(m) for m in c["mismatches"]]
        c = json.loads(json.dumps(r))
        if not isinstance(attempt(F["findall_mismatches"], c), dict) or "mismatches" in c:
            for t in targets:
                v = attempt(F["is_worth_realn"], c, Target(t["pos"], t["ref"], t["alt"], t["shifts"]), t["qual_lim"])
                rec["worth"].append(v if isinstance(v, dict) else bool(v))
        out.append(rec)
    path = os.path.join(ROOT, "tests", "golden", "realn_filter_cases.json")
    with open(path, "w") as f:
        json.dump({"targets": targets, "reads": out}, f, separators=(",", ":"))
    n_raise = sum(isinstance(r["mismatches"]["0"], dict) for r in out)
    n_mm = sum(len(r["mismatches"]["0"]) for r in out if not isinstance(r["mismatches"]["0"], dict))
    n_true = sum(v is True for r in out for v in r["worth"])
    print("%d reads (%d raise in findall_mismatches), %d mismatches, %d targets, %d of %d is_worth_realn verdicts True -> %s (%d KB)"
          % (len(out), n_raise, n_mm, len(targets), n_true, sum(len(r["worth"]) for r in out), path, os.path.getsize(path) // 1024))


if __name__ == "__main__":
    main()
