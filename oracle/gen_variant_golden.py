"""Golden vectors for indelpost_amd.variant.Variant and the pileup front-end helpers (TEST INFRASTRUCTURE, build container only).

variant.pyx / pileup.pyx / utilities.pyx cannot be compiled here (they cimport pysam).  Like oracle/gen_decoder_golden.py
this script reads their TEXT from /root/reference at generation time and executes it with the Cython declarations dropped
mechanically -- `cdef class` -> `class`, typed parameters -> names, `cdef T x = e` -> `x = e`, bare declarations removed;
every statement stays as it is -- against an in-memory FASTA duck type, and records inputs and outputs as data
(tests/golden/variant_cases.json).  No reference text is written anywhere.

    python oracle/gen_variant_golden.py
"""
import array
import json
import os
import re
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from gen_decoder_golden import function_text, REF            # noqa: E402

TYPES = r"(?:str|int|bint|list|tuple|dict|object|double|float|Variant|FastaFile|VariantFile|AlignmentFile|AlignedSegment|UnsplicedLocalReference|array\.array)"
LET = "ACGT"


def strip_cython(src):
    out = []
    for line in src.split("\n"):
        if re.match(r"^\s*(from .* cimport|cimport |from \.|#cython)", line):
            continue
        line = re.sub(r"^(\s*)cdef class ", r"\1class ", line)
        line = re.sub(r"^(\s*)(?:cpdef|cdef) (?:%s )?(\w+\s*\()" % TYPES, r"\1def \2", line) if re.match(r"^\s*(?:cpdef|cdef) (?:%s )?\w+\s*\(" % TYPES, line) else line
        if re.match(r"^\s*def ", line) or re.match(r"^\s+%s \w+,?\s*$" % TYPES, line):
            line = re.sub(r"(?<=[(,])\s*%s\s+(?=\w)" % TYPES, " ", line).replace("( ", "(")
            line = re.sub(r"^(\s+)%s (\w+,?)\s*$" % TYPES, r"\1\2", line)
        m = re.match(r"^(\s+)cdef (?:double \[:\]|%s) (.*)$" % TYPES, line)
        if m:
            decl = m.group(2)
            if "=" not in decl:
                line = ""
            else:
                parts, depth, cur = [], 0, ""
                for ch in decl:                                    # split the declaration list at top-level commas
                    depth += ch in "([{"
                    depth -= ch in ")]}"
                    if ch == "," and depth == 0:
                        parts.append(cur)
                        cur = ""
                    else:
                        cur += ch
                parts.append(cur)
                line = m.group(1) + "; ".join(x.strip() for x in parts if "=" in x)
        out.append(line)
    return "\n".join(out)


class Fasta:
    def __init__(self, seqs):
        self.seqs, self.references, self.filename = seqs, list(seqs), None

    def fetch(self, chrom, start, end):
        return self.seqs[chrom][max(0, start):max(0, end)]

    def get_reference_length(self, chrom):
        return len(self.seqs[chrom])


def s(codes):
    return "".join(LET[int(c)] for c in codes)


class _IntNumpy:
    """utilities.split keeps its CIGAR moves in `double [:]` views but reads them back into `cdef int` variables: with the
    declarations dropped the truncation would be lost, so its np.zeros hands out integer arrays (the values are lengths)"""
    @staticmethod
    def zeros(shape):
        return np.zeros(shape, dtype=np.int64)


def float_split():
    """utilities.split executed with numpy's own float64 np.zeros and its index arithmetic forced back to int where Cython's `cdef int`
    declarations would do it (int(...) around the slice bounds): used only to CHECK that the integer stand-in above changes nothing"""
    ns = {"re": re, "np": np, "cigar_ptrn": re.compile(r"[0-9]+[MIDNSHPX=]")}
    src = strip_cython(function_text(os.path.join(REF, "utilities.pyx"), "split"))
    exec(compile(src, "<utilities:split/float>", "exec"), ns)
    f = ns["split"]

    class IntIndex:                                                   # sequence wrapper: float slice bounds truncate like a C int
        def __init__(self, d):
            self.d = d

        def __getitem__(self, k):
            if isinstance(k, slice):
                t = lambda v: None if v is None else int(v)
                return IntIndex(self.d[slice(t(k.start), t(k.stop), t(k.step))])
            return self.d[int(k)]

        def __len__(self):
            return len(self.d)

    def run(data, *a, **kw):
        lt, rt = f(IntIndex(data), *a, **kw)
        return lt.d, rt.d
    return run


def load():
    ns = {"re": re, "np": _IntNumpy, "array": array, "cigar_ptrn": re.compile(r"[0-9]+[MIDNSHPX=]")}
    for n in ("to_flat_list", "to_minimal_repeat_unit", "repeat_counter", "count_lowqual_non_ref_bases", "get_mapped_subreads",
              "get_spliced_subreads", "get_end_pos", "locate_indels", "split_cigar", "split"):
        exec(compile(strip_cython(function_text(os.path.join(REF, "utilities.pyx"), n)), "<utilities:%s>" % n, "exec"), ns)
    exec(compile(strip_cython(open(os.path.join(REF, "variant.pyx")).read()), "<variant.pyx>", "exec"), ns)
    for n in ("is_end_dirty", "parse_spliced_read", "leftalign_cigar", "leftalign_indel_read"):
        exec(compile(strip_cython(function_text(os.path.join(REF, "pileup.pyx"), n)), "<pileup:%s>" % n, "exec"), ns)
    return ns


def vtuple(v):
    return [v.chrom, v.pos, v.ref, v.alt]


FLOAT_SPLIT = None
N_FLOAT_CHECKED = [0]


def main():
    global FLOAT_SPLIT
    F = load()
    FLOAT_SPLIT = float_split()
    Variant = F["Variant"]
    rng = np.random.default_rng(424242)
    g = list(s(rng.integers(0, 4, 3000)))
    for k in range(40):                                           # tandem repeats and homopolymers to shift indels through
        p = int(rng.integers(350, 2600))
        unit = s(rng.integers(0, 4, int(rng.integers(1, 4))))
        rep = unit * int(rng.integers(3, 9))
        g[p:p + len(rep)] = list(rep)
    genome = "".join(g)
    fa = Fasta({"chr1": genome})
    out = {"generator": "oracle/gen_variant_golden.py", "genome": genome, "variants": [], "equal": [], "helpers": {}}
    made = []
    for k in range(260):
        pos = int(rng.integers(320, 2650))
        kind = k % 6
        base = genome[pos - 1]
        if kind in (0, 1):                                        # insertion (sometimes a copy of what follows: shiftable)
            n = int(rng.integers(1, 7))
            ins = genome[pos:pos + n] if kind == 1 else s(rng.integers(0, 4, n))
            ref, alt = base, base + ins
        elif kind in (2, 3):                                      # deletion
            n = int(rng.integers(1, 9))
            ref, alt = genome[pos - 1:pos + n], base
        elif kind == 4:                                           # complex / MNV
            ref = genome[pos - 1:pos + int(rng.integers(1, 5))]
            alt = base + s(rng.integers(0, 4, int(rng.integers(1, 5))))
            if ref == alt:
                continue
        elif rng.random() < 0.5:                                  # padded representation of an insertion
            ref, alt = genome[pos - 1:pos + 2], genome[pos - 1:pos + 2] + s(rng.integers(0, 4, 2))
        else:                                                     # SNV
            ref, alt = base, LET[(LET.index(base) + 1) % 4]
        try:
            v = Variant("chr1", pos, ref, alt, fa)
        except ValueError:
            continue
        made.append(v)
        rec = {"in": ["chr1", pos, ref, alt], "type": v.variant_type, "indel_seq": v.indel_seq, "normalized": vtuple(v.normalize()),
               "is_leftaligned": bool(v.is_leftaligned), "is_normalized": bool(v.is_normalized),
               "non_complex": bool(v.is_non_complex_indel()), "equivalents": [vtuple(e) for e in v.generate_equivalents()],
               "left_flank": v.left_flank(), "right_flank": v.right_flank(), "left_flank_n20": v.left_flank(20, True),
               "count_repeats": v.count_repeats(), "count_repeats_raw": v.count_repeats(False)}
        rec["private_equivalents"] = [vtuple(e) for e in v._generate_equivalents_private()]
        rec["indel_seq_I"], rec["indel_seq_D"] = v._get_indel_seq("I"), v._get_indel_seq("D")
        r = v._reduce_complex_indel("D" if len(ref) > len(alt) else "I")
        rec["reduced"] = vtuple(r) if r is not None else None
        out["variants"].append(rec)
    for _ in range(300):
        a, b = made[int(rng.integers(0, len(made)))], made[int(rng.integers(0, len(made)))]
        if rng.random() < 0.5:                                    # an equivalent representation of a
            eq = a.generate_equivalents()
            b = eq[int(rng.integers(0, len(eq)))]
        out["equal"].append({"a": vtuple(a), "b": vtuple(b), "eq": bool(a == b), "same_hash": hash(a) == hash(b)})

    H = out["helpers"]
    def rand_cigar():
        toks = [str(int(rng.integers(5, 40))) + "M"]
        for _ in range(int(rng.integers(0, 5))):
            toks.append(str(int(rng.integers(1, 9 if rng.random() < 0.8 else 300))) + "IDN"[int(rng.integers(0, 3))])
            toks.append(str(int(rng.integers(5, 40))) + "M")
        if rng.random() < 0.3:
            toks = [str(int(rng.integers(1, 9))) + "S"] + toks
        if rng.random() < 0.3:
            toks.append(str(int(rng.integers(1, 9))) + "S")
        return toks
    def spans(toks):
        q = sum(int(t[:-1]) for t in toks if t[-1] in "MIS=X")
        r = sum(int(t[:-1]) for t in toks if t[-1] in "MDN=X")
        return q, r
    for name in ("mapped_subreads", "spliced_subreads", "locate_indels", "end_pos", "split", "parse_spliced_read", "is_end_dirty",
                 "lowqual", "leftalign_cigar", "leftalign_indel_read"):
        H[name] = []
    for k in range(200):
        toks = rand_cigar()
        cs = "".join(toks)
        q, r = spans(toks)
        aln_start = int(rng.integers(400, 900))
        off0 = int(toks[0][:-1]) if toks[0][-1] == "S" else 0
        off1 = int(toks[-1][:-1]) if toks[-1][-1] == "S" else 0
        read_start, aln_end = aln_start - off0, aln_start + r - 1
        read_end = aln_end + off1
        H["mapped_subreads"].append({"cigar": cs, "start": aln_start, "end": aln_end, "expect": [list(x) for x in F["get_mapped_subreads"](cs, aln_start, aln_end)]})
        H["spliced_subreads"].append({"cigar": cs, "start": read_start, "end": read_end, "expect": [list(x) for x in F["get_spliced_subreads"](cs, read_start, read_end)]})
        ins, dels = F["locate_indels"](cs, read_start)
        H["locate_indels"].append({"cigar": cs, "start": read_start, "expect": [[list(x) for x in ins], [list(x) for x in dels]]})
        flank = "A" * int(rng.integers(1, max(2, q - 1)))
        try:
            H["end_pos"].append({"start": read_start, "flank_len": len(flank), "cigar": cs, "expect": int(F["get_end_pos"](read_start, flank, cs))})
        except IndexError:
            pass
        read = s(rng.integers(0, 4, q))
        quals = array.array("B", [int(x) for x in rng.integers(2, 41, q)])
        tp = read_start + int(rng.integers(0, r + off0 + off1 + 2))
        for data, sp, isref, tag in ((read, read_start, False, "seq"), (quals, read_start, False, "qual"), (genome[aln_start - 1:aln_start - 1 + r], aln_start, True, "ref")):
            if tag == "ref" and "N" in cs:
                continue
            for rev in (False, True):
                lt, rt = F["split"](data, cs, tp, sp, is_for_ref=isref, reverse=rev)
                try:                                              # the stand-in is value-preserving on every committed input
                    flt, frt = FLOAT_SPLIT(data, cs, tp, sp, is_for_ref=isref, reverse=rev)
                    assert (list(flt), list(frt)) == (list(lt), list(rt)), ("float path differs", cs, tp, sp, isref, rev)
                    N_FLOAT_CHECKED[0] += 1
                except (TypeError, IndexError):
                    pass                                          # (float slice bounds that plain Python cannot index with: not comparable)
                H["split"].append({"data": list(data) if tag == "qual" else data, "kind": tag, "cigar": cs, "target_pos": tp, "string_pos": sp,
                                   "is_for_ref": isref, "reverse": rev, "expect": [list(lt) if tag == "qual" else lt, list(rt) if tag == "qual" else rt]})
        pos = read_start + int(rng.integers(-5, r + 10))
        rpos = pos + int(rng.integers(0, 6))
        res = F["parse_spliced_read"](cs, read_start, read_end, pos, rpos)
        H["parse_spliced_read"].append({"cigar": cs, "read_start": read_start, "read_end": read_end, "pos": pos, "rpos": rpos,
                                        "expect": [bool(res[0]), list(res[1]) if res[1] else None, bool(res[2]), list(res[3]), list(res[4])]})
        thr = int(rng.integers(10, 30))
        H["is_end_dirty"].append({"quals": list(quals), "thresh": thr, "pos": pos, "read_start": read_start, "read_end": read_end, "cigar": cs,
                                  "expect": bool(F["is_end_dirty"](quals, thr, pos, read_start, read_end, cs))})
        if "N" not in cs:
            ref_seq = genome[aln_start - 1:aln_start - 1 + r]
            H["lowqual"].append({"read": read, "ref": ref_seq, "quals": list(quals), "cigar_list": toks, "thresh": thr,
                                 "expect": int(F["count_lowqual_non_ref_bases"](read, ref_seq, quals, toks, thr))})
            for (p_, n_), t_ in [(x, "I") for x in ins] + [(x, "D") for x in dels]:
                res = F["leftalign_indel_read"]("chr1", p_, n_, t_, cs, read_start, aln_start, read, ref_seq, quals, fa)
                H["leftalign_indel_read"].append({"args": ["chr1", p_, n_, t_, cs, read_start, aln_start, read, ref_seq, list(quals)],
                                                  "expect": [res[0], res[1], res[2], res[3], res[4], res[5], list(res[6]), list(res[7]), vtuple(res[8])]})
    for v in made[:120]:                                          # leftalign_cigar: move a read's gap to the normalised position
        if not v.is_non_complex_indel():
            continue
        n = len(v.indel_seq)
        st = v.pos - int(rng.integers(10, 40))
        cs = "%dM%d%s%dM" % (v.pos - st + 1, n, v.variant_type, int(rng.integers(10, 40)))
        cp = Variant(v.chrom, v.pos, v.ref, v.alt, fa, skip_validation=True)
        H["leftalign_cigar"].append({"cigar": cs, "variant": vtuple(v), "read_start": st, "expect": F["leftalign_cigar"](cs, cp, st)})

    with open(os.path.join(ROOT, "tests", "golden", "variant_cases.json"), "w") as f:
        json.dump(out, f)
        f.write("\n")
    print(len(out["variants"]), len(out["equal"]), {k: len(v) for k, v in H.items()}, "split cases also run through the float path:", N_FLOAT_CHECKED[0])


if __name__ == "__main__":
    main()
