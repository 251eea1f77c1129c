"""Golden vectors for the composed drivers above the aligner (TEST INFRASTRUCTURE, build container only).

What is pinned: make_pileup / fetch_reads / dictize_read / get_ref_seq (pileup.pyx:51-298), get_local_reference with and without
splice patterns (utilities.pyx:505-586), retarget and its window / 3 recursion (pileup.pyx:577-808), grid_search + update_read_info
(varaln.pyx:1148-1225, pileup.pyx:811-913), check_overhangs / filter_spurious_overhangs (pileup.pyx:427-574),
find_by_smith_waterman_realn with findall_mismatches / is_worth_realn / is_target_by_ssw (localn.pyx:15-291),
parse_read_by_mut_aln (localn.pyx:475-539) and is_perfect_match (varaln.pyx:1228-1234).

How: like oracle/gen_decoder_golden.py and gen_variant_golden.py.  The modules cannot be compiled here (they cimport pysam); this
script reads their TEXT from /root/reference at generation time, drops the Cython declarations mechanically (statements stay as
they are), executes the functions against in-memory pysam duck types (FASTA, BAM, aligned segments built by a small read
simulator) with `make_aligner` / `align` bound to the oracle's restatement of ssw.c (pinned to the compiled reference,
tests/test_oracle.py), and records INPUTS (genome, segments, parameters) and OUTPUTS (read dicts, returned tuples) as data in
tests/golden/driver_cases.json.  No reference text is written anywhere.

    PYTHONHASHSEED=0 python oracle/gen_driver_golden.py
(retarget iterates over a set of Variant objects, pileup.pyx:733: with ties in its sort the reference's own answer depends on
Python's string hashing; the generator fixes the seed and drops the rare scenario whose answer changes under another seed.)
"""
import array
import collections
import copy
import json
import os
import random
import re
import sys
import zlib
from difflib import SequenceMatcher, get_close_matches

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from gen_decoder_golden import function_text, REF            # noqa: E402
from gen_variant_golden import strip_cython, Fasta, _IntNumpy, s  # noqa: E402
from oracle import oracle as O                                # noqa: E402

Alignment = collections.namedtuple("Alignment", "CIGAR optimal_score sub_optimal_score reference_start reference_end read_start read_end")
LET = "ACGT"


# ---- the aligner the reference's functions are bound to: the oracle's restatement of ssw.c behind the SSW interface -----------
class PortSSW:
    port = None

    def __init__(self, match_score=2, mismatch_penalty=2):
        self.mat = O.dna_matrix(match_score, mismatch_penalty)
        self.reference = None
        self.read = None

    def setReference(self, ref):
        self.reference = ref

    def setRead(self, read):
        self.read = read

    def align(self, gap_open=3, gap_extension=1, start_idx=0, end_idx=0):
        e = PortSSW.port.align(O.encode(self.read), O.encode(self.reference), self.mat, gap_open & 255, gap_extension & 255)
        return Alignment(O.cigar_string(e["cigar"]) if e["cigar"] is not None else None, e["score1"], e["score2"], e["ref_begin1"],
                         e["ref_end1"], e["read_begin1"], e["read_end1"])


def make_aligner(ref_seq, match_score, mismatch_penalty):
    a = PortSSW(match_score, mismatch_penalty)
    a.setReference(ref_seq)
    return a


def align(aligner, read_seq, gap_open_penalty, gap_extension_penalty):
    aligner.setRead(read_seq)
    return aligner.align(gap_open=gap_open_penalty, gap_extension=gap_extension_penalty)


def strip_cdef_blocks(src):
    """`cdef:` followed by an indented block of typed declarations -> the assignments, one level out"""
    out, lines, i = [], src.split("\n"), 0
    while i < len(lines):
        m = re.match(r"^(\s*)cdef:\s*$", lines[i])
        if not m:
            out.append(lines[i])
            i += 1
            continue
        ind = len(m.group(1))
        i += 1
        while i < len(lines) and (lines[i].strip() == "" or len(lines[i]) - len(lines[i].lstrip()) > ind):
            body = lines[i].strip()
            mm = re.match(r"^(?:int|str|list|tuple|dict|bint|double|object)\s+(\w+\s*=.*)$", body)
            if mm:
                out.append(" " * ind + mm.group(1))
            i += 1
    return "\n".join(out)


_TYPES = r"(?:str|int|bint|list|tuple|dict|object|double|float|Variant|FastaFile|AlignmentFile|AlignedSegment|UnsplicedLocalReference)"


def strip_typed_defaults(src):
    """one typed parameter per line WITH a default (`bint unspliced=False,`) -> the name and the default"""
    return "\n".join(re.sub(r"^(\s+)%s (\w+\s*=[^=].*)$" % _TYPES, r"\1\2", line) for line in src.split("\n"))


def load():
    ns = {"re": re, "np": _IntNumpy, "array": array, "random": random, "cigar_ptrn": re.compile(r"[0-9]+[MIDNSHPX=]"),
          "get_close_matches": get_close_matches, "SequenceMatcher": SequenceMatcher, "make_aligner": make_aligner, "align": align}

    def take(fname, names):
        for n in names:
            exec(compile(strip_cython(strip_typed_defaults(function_text(os.path.join(REF, fname), n))), "<%s:%s>" % (fname, n), "exec"), ns)

    take("utilities.pyx", ["most_common", "to_flat_list", "to_minimal_repeat_unit", "repeat_counter", "count_lowqual_non_ref_bases",
                           "get_mapped_subreads", "get_spliced_subreads", "get_end_pos", "locate_indels", "split_cigar",
                           "merge_consecutive_gaps", "make_insertion_first", "split", "get_local_reference"])
    exec(compile(strip_cython(open(os.path.join(REF, "variant.pyx")).read()), "<variant.pyx>", "exec"), ns)
    exec(compile(strip_cython(strip_cdef_blocks(open(os.path.join(REF, "local_reference.pyx")).read())), "<local_reference.pyx>", "exec"), ns)
    take("localn.pyx", ["findall_indels", "is_compatible_repeats", "is_covering_target", "findall_mismatches", "is_worth_realn",
                        "is_target_by_ssw", "find_by_smith_waterman_realn", "parse_read_by_mut_aln"])
    take("pileup.pyx", ["is_end_dirty", "parse_spliced_read", "leftalign_cigar", "leftalign_indel_read", "get_ref_seq", "dictize_read",
                        "fetch_reads", "is_within_intron", "make_pileup", "check_overhangs", "is_junctional", "is_overhang",
                        "overhang_aligners", "filter_spurious_overhangs", "is_non_spurious_overhang", "retarget", "update_read_info",
                        "trim_ref_flank", "numeric_span", "update_cigar", "update_read_positions"])
    take("varaln.pyx", ["generate_grid", "grid_search", "is_perfect_match"])
    return ns


# ---- pysam duck types ----------------------------------------------------------------------------------------------------------
class Segment:
    def __init__(self, name, pos0, cigar, seq, quals, mapq=60, is_reverse=False, is_duplicate=False, is_secondary=False):
        self.query_name, self.reference_start, self.cigarstring = name, pos0, cigar
        self.query_sequence, self.query_qualities = seq, array.array("B", quals)
        self.mapping_quality, self.is_reverse, self.is_duplicate, self.is_secondary = mapq, is_reverse, is_duplicate, is_secondary
        span = sum(int(t[:-1]) for t in re.findall(r"[0-9]+[MIDNSHPX=]", cigar) if t[-1] in "MDN=X")
        self.reference_end = pos0 + span

    def record(self):
        return [self.query_name, self.reference_start, self.cigarstring, self.query_sequence, "".join(chr(33 + q) for q in self.query_qualities),
                self.mapping_quality, int(self.is_reverse), int(self.is_duplicate), int(self.is_secondary)]


class Bam:
    def __init__(self, chrom, segs):
        self.references, self.segs = [chrom], segs

    def fetch(self, chrom, start, end, until_eof=True):
        return [g for g in self.segs if g.reference_start < end and g.reference_end > start]

    def count(self, chrom, start, end, read_callback="all"):
        n = 0
        for g in self.fetch(chrom, start, end):
            if read_callback == "all" and (g.is_duplicate or g.is_secondary):
                continue
            n += 1
        return n


class Contig:
    """what find_by_smith_waterman_realn / parse_read_by_mut_aln need of contig.pyx's Contig"""

    def __init__(self, lt, mid, rt, ref):
        self.lt_consensus_seq, self.indel_seq, self.rt_consensus_seq, self.ref = lt, mid, rt, ref

    def get_contig_seq(self, split=False):
        return (self.lt_consensus_seq, self.indel_seq, self.rt_consensus_seq) if split else self.lt_consensus_seq + self.indel_seq + self.rt_consensus_seq

    def get_reference_seq(self, split=False):
        return self.ref


# ---- read simulator: reads off the reference or the variant haplotype, with the CIGAR a mapper would give them -------------------
def simulate(rng, genome, pos, ref, alt, n_reads, read_len, intron=None, second=None):
    """Segments around a variant at 1-based `pos` (VCF style: ref / alt share their first base).  A carrier whose indel sits well
    inside the read gets the gapped CIGAR; near a read end it is soft-clipped or forced through ungapped (mismatches), like a
    mapper would.  intron = (a, b) 1-based first / last intron base: reads crossing it are spliced (N), some overhang into it.
    second = (pos, ref, alt): another indel nearby carried by some reads."""
    segs = []
    dlen = len(ref) - len(alt)
    for k in range(n_reads):
        carrier = rng.random() < 0.45
        other = second is not None and rng.random() < 0.2
        st = int(rng.integers(pos - read_len + 8, pos - 6))          # 0-based start on the reference
        st = max(5, st)
        # walk the haplotype collecting (op, ref base index) -- simple and exact
        out, ops, g = [], [], st
        events = []
        if carrier:
            events.append((pos, ref, alt))
        if other and (not carrier or abs(second[0] - pos) > len(ref) + 3):
            events.append(second)
        events.sort()
        ei = 0
        while len(out) < read_len and g < len(genome) - 1:
            if intron and g + 1 == intron[0]:                        # splice over the intron
                ops.append(("N", intron[1] - intron[0] + 1))
                g = intron[1]
                continue
            if ei < len(events) and g + 1 == events[ei][0]:
                p_, r_, a_ = events[ei]
                out.append(genome[g]); ops.append(("M", 1)); g += 1
                if len(a_) > len(r_):
                    ins = a_[len(r_):]
                    out.extend(ins); ops.append(("I", len(ins)))
                else:
                    ops.append(("D", len(r_) - len(a_))); g += len(r_) - len(a_)
                ei += 1
                continue
            out.append(genome[g]); ops.append(("M", 1)); g += 1
        out = out[:read_len]
        # compress ops to the length of `out`
        cig, used = [], 0
        for op, n in ops:
            if op in ("M", "I"):
                n = min(n, len(out) - used)
                if n <= 0:
                    break
                used += n
            if cig and cig[-1][0] == op:
                cig[-1][1] += n
            else:
                cig.append([op, n])
        while cig and cig[-1][0] in ("D", "N", "I") and cig[-1][0] != "I":
            cig.pop()
        seq = list(out)
        for q in np.flatnonzero(rng.random(len(seq)) < 0.006):        # sequencing errors
            seq[q] = LET[int(rng.integers(0, 4))]
        quals = [int(x) for x in rng.integers(24, 41, len(seq))]
        for q in np.flatnonzero(rng.random(len(seq)) < 0.03):
            quals[q] = int(rng.integers(2, 20))
        # what a mapper does with an indel close to a read end: clip it away or push through ungapped
        pos0 = st
        gaps = [i for i, c in enumerate(cig) if c[0] in ("I", "D")]
        if gaps:
            first_m = cig[0][1] if cig[0][0] == "M" else 0
            last_m = cig[-1][1] if cig[-1][0] == "M" else 0
            mode = int(rng.integers(0, 3))
            if first_m < 12 and cig[0][0] == "M" and len(cig) >= 3 and cig[1][0] in ("I", "D") and mode > 0:
                head = cig[0][1] + (cig[1][1] if cig[1][0] == "I" else 0)
                skip = cig[0][1] + (cig[1][1] if cig[1][0] == "D" else 0)
                cig = [["S", head]] + cig[2:]
                pos0 = st + skip
            elif last_m < 12 and cig[-1][0] == "M" and len(cig) >= 3 and cig[-2][0] in ("I", "D") and mode > 0:
                tail = cig[-1][1] + (cig[-2][1] if cig[-2][0] == "I" else 0)
                cig = cig[:-2] + [["S", tail]]
            elif mode == 2 and len(gaps) == 1 and "N" not in [c[0] for c in cig]:
                # a clip on the shorter side even though the indel is well inside: the target hides behind a soft clip
                i = gaps[0]
                lt = sum(c[1] for c in cig[:i] if c[0] in ("M", "I"))
                rt = sum(c[1] for c in cig[i + 1:] if c[0] in ("M", "I"))
                if lt <= rt:
                    skip = sum(c[1] for c in cig[:i] if c[0] in ("M", "D")) + (cig[i][1] if cig[i][0] == "D" else 0)
                    cig = [["S", lt + (cig[i][1] if cig[i][0] == "I" else 0)]] + cig[i + 1:]
                    pos0 = st + skip
                else:
                    cig = cig[:i] + [["S", rt + (cig[i][1] if cig[i][0] == "I" else 0)]]
        merged = []
        for op, n in cig:
            if n <= 0:
                continue
            if merged and merged[-1][0] == op:
                merged[-1][1] += n
            else:
                merged.append([op, n])
        cigar = "".join("%d%s" % (n, op) for op, n in merged)
        qlen = sum(n for op, n in merged if op in "MIS")
        if qlen != len(seq) or not merged or merged[0][0] in "DN" or merged[-1][0] in "DN":
            continue
        segs.append(Segment("r%03d" % k, pos0, cigar, "".join(seq), quals, mapq=int(rng.choice([60, 60, 60, 30, 1, 0])),
                            is_reverse=bool(rng.integers(0, 2)), is_duplicate=rng.random() < 0.03))
    if intron:                                                        # reads that overhang into the intron instead of splicing
        for k in range(max(3, n_reads // 8)):
            left = rng.random() < 0.5
            over = int(rng.integers(2, 9))
            if left:
                end = intron[0] - 1 + over                            # last base (1-based) inside the intron
                st = end - read_len
                seq = list(genome[st:end - over]) + list(genome[intron[1]:intron[1] + over])   # the exon continues after the intron
            else:
                st = intron[1] - over
                seq = list(genome[intron[0] - 1 - over:intron[0] - 1]) + list(genome[intron[1]:intron[1] + read_len - over])
            quals = [int(x) for x in rng.integers(24, 41, len(seq))]
            segs.append(Segment("o%03d" % k, st, "%dM" % len(seq), "".join(seq), quals, mapq=60, is_reverse=bool(rng.integers(0, 2))))
    segs.sort(key=lambda g: g.reference_start)
    return segs


def clone(pileup):
    """a fresh copy of the read dicts for one call of a mutating driver (Variant objects inside are shared: they are not mutated)"""
    def cp(v):
        if isinstance(v, dict):
            return {k: cp(x) for k, x in v.items() if k != "read"}
        if isinstance(v, list):
            return [cp(x) for x in v]
        if isinstance(v, tuple):
            return tuple(cp(x) for x in v)
        if isinstance(v, array.array):
            return array.array(v.typecode, v)
        return v
    return [cp(r) for r in pileup]


def ser(x):
    """read dicts and return values as plain data (Variant -> [chrom, pos, ref, alt]; arrays / tuples -> lists)"""
    if hasattr(x, "ref") and hasattr(x, "alt") and hasattr(x, "chrom"):
        return [x.chrom, x.pos, x.ref, x.alt]
    if isinstance(x, dict):
        return {k: ser(v) for k, v in x.items() if k != "read"}
    if isinstance(x, (list, tuple, array.array, np.ndarray)):
        return [ser(v) for v in x]
    if isinstance(x, (np.integer,)):
        return int(x)
    if isinstance(x, PortSSW):
        return {"aligner_ref": x.reference}
    if isinstance(x, Alignment):
        return list(x)
    return x


def dig(x):
    """digest of a bulky value (flank strings, quality lists, whole indel tuples): the test recomputes it from its own output"""
    return zlib.crc32(json.dumps(ser(x), sort_keys=True).encode())


BULKY = ("read_seq", "read_qual", "ref_seq", "lt_flank", "rt_flank", "lt_ref", "rt_ref", "lt_qual", "rt_qual", "I", "D", "mismatches")


def ser_read(r):
    """a read dict as data: scalars and short fields verbatim, bulky ones as digests (key + "#")"""
    out = {}
    for k, v in r.items():
        if k == "read":
            continue
        if k in BULKY:
            out[k + "#"] = dig(v)
        else:
            out[k] = ser(v)
    return out


def plant_repeats(rng, g, n):
    for _ in range(n):
        p = int(rng.integers(400, len(g) - 400))
        unit = s(rng.integers(0, 4, int(rng.integers(1, 4))))
        rep = unit * int(rng.integers(3, 8))
        g[p:p + len(rep)] = list(rep)


def main():
    O.build()
    PortSSW.port = O.Backend("port")
    F = load()
    Variant, ULR = F["Variant"], F["UnsplicedLocalReference"]
    rng = np.random.default_rng(20261004)
    genomes = []
    for _ in range(4):
        g = list(s(rng.integers(0, 4, 4000)))
        plant_repeats(rng, g, 25)
        genomes.append("".join(g))
    out = {"generator": "oracle/gen_driver_golden.py", "segment_fields": ["name", "pos0", "cigar", "seq", "quals (phred+33)", "mapq", "is_reverse",
           "is_duplicate", "is_secondary"], "genomes": genomes, "scenarios": []}
    n_try = 0
    while len(out["scenarios"]) < 42 and n_try < 400:
        n_try += 1
        k = len(out["scenarios"])
        gi = k % len(genomes)
        genome = genomes[gi]
        fa = Fasta({"chr1": genome})
        pos = int(rng.integers(1200, 2800))
        kind = k % 7
        base = genome[pos - 1]
        if kind in (0, 1):
            n = int(rng.integers(1, 5)) if kind == 0 else int(rng.integers(6, 25))
            ins = genome[pos:pos + n] if rng.random() < 0.3 else s(rng.integers(0, 4, n))
            ref, alt = base, base + ins
        elif kind in (2, 3):
            n = int(rng.integers(1, 5)) if kind == 2 else int(rng.integers(6, 20))
            ref, alt = genome[pos - 1:pos + n], base
        elif kind == 4:                                               # indel next to a second one (complex candidates)
            n = int(rng.integers(2, 6))
            ref, alt = genome[pos - 1:pos + n], base
        elif kind == 5:                                               # insertion at an exon boundary (spliced reads, overhangs)
            n = int(rng.integers(2, 7))
            ref, alt = base, base + s(rng.integers(0, 4, n))
        else:
            n = int(rng.integers(2, 8))
            ref, alt = genome[pos - 1:pos + n], base
        try:
            target = Variant("chr1", pos, ref, alt, fa)
        except ValueError:
            continue
        intron = second = None
        if kind == 5:                                                 # the target within 4 bases of an exon end: parse_spliced_read's overhang rule
            a = pos + int(rng.integers(1, 5))
            intron = (a, a + int(rng.integers(150, 400)))
        if kind == 6 and k % 2:                                       # the target just inside the far end of an intron
            b = pos + int(rng.integers(0, 3))
            intron = (b - int(rng.integers(150, 300)), b)
        if kind == 4:
            p2 = pos + len(ref) + int(rng.integers(0, 4))
            second = (p2, genome[p2 - 1], genome[p2 - 1] + s(rng.integers(0, 4, int(rng.integers(1, 4)))))
        read_len = int(rng.choice([75, 100, 150]))
        # what the reads carry: the target itself, or (decoys) a near miss -- another inserted / deleted sequence at the same place, or the
        # same event 40 bases away (beyond `within`) -- so that the candidate matching has something to reject
        c_pos, c_ref, c_alt, decoy = pos, ref, alt, k % 5 == 3
        if decoy:
            if k % 2 and len(alt) > len(ref):
                ins = list(alt[1:])
                ins[int(rng.integers(0, len(ins)))] = LET[int(rng.integers(0, 4))]
                c_alt = base + "".join(ins) + (LET[int(rng.integers(0, 4))] if rng.random() < 0.5 else "")
            else:
                c_pos = pos + 40
                if len(ref) > len(alt):
                    c_ref, c_alt = genome[c_pos - 1:c_pos + len(ref) - 1], genome[c_pos - 1]
                else:
                    c_ref, c_alt = genome[c_pos - 1], genome[c_pos - 1] + alt[1:]
        segs = simulate(rng, genome, c_pos, c_ref, c_alt, int(rng.integers(16, 40)), read_len, intron, second)
        if len(segs) < 8:
            continue
        bam = Bam("chr1", segs)
        window, basequalthresh, mapq = 50, 20, 1
        down = 1000 if k % 9 else 16                                  # a few scenarios exercise the downsampling
        sc = {"genome": gi, "target": ["chr1", pos, ref, alt], "carried": [c_pos, c_ref, c_alt], "segments": [g_.record() for g_ in segs], "window": window,
              "basequalthresh": basequalthresh, "downsamplethresh": down, "exclude_duplicates": bool(k % 2), "kind": kind}
        unspl = ULR("chr1", pos, len(genome), window, fa)
        try:
            pileup, sf = F["make_pileup"](target, bam, unspl, sc["exclude_duplicates"], window, down, basequalthresh)
        except Exception as e:                                        # (a simulated read the reference's own code trips over)
            continue
        sc["pileup"] = [ser_read(r) for r in pileup]
        sc["sample_factor"] = sf
        # get_local_reference per read (the windows retarget cuts)
        sc["local_reference"] = []
        for w in (50, 16):
            row = []
            for r in pileup:
                try:
                    ref_seq, lt_len = F["get_local_reference"](target, [r], w, unspl)
                    row.append([zlib.crc32(ref_seq.encode()), len(ref_seq), lt_len])
                except Exception:
                    row.append(None)
            sc["local_reference"].append({"window": w, "per_read": row})
        # retarget under single pairs, and the grid search over all of them
        grid = F["generate_grid"](True, 3, 1, target)
        sc["grid"] = [list(p) for p in grid]
        within, cutoff = 30, 0.6
        sc["within"], sc["cutoff"] = within, cutoff
        sc["retarget"] = []
        ok = True
        for (go, ge) in grid[:3]:
            for exact in (False, True):
                pl = clone(pileup)
                try:
                    res = F["retarget"](target, pl, window, mapq, within, cutoff, 3, 2, go, ge, unspl, exact)
                except Exception as e:
                    ok = False
                    break
                sc["retarget"].append({"go": go, "ge": ge, "exact": exact,
                                       "expect": None if res is None else [ser(res[0]), [r["read_name"] for r in res[1]], res[2], [zlib.crc32(w_.encode()) for w_ in res[3]], list(res[4])]})
            if not ok:
                break
        if not ok:
            continue
        pl = clone(pileup)
        try:
            res = F["grid_search"](target, pl, window, mapq, within, cutoff, 3, 2, grid, unspl, False)
        except Exception as e:
            continue
        sc["grid_search"] = None if res is None else {"candidate": ser(res[0]), "reads": [ser_read(r) for r in res[1]], "gap_open": res[2], "gap_ext": res[3]}
        # overhangs
        pl = clone(pileup)
        ans = F["check_overhangs"](pl)
        sc["overhangs"] = None
        if ans:
            try:
                keep = F["filter_spurious_overhangs"](target, ans[0], ans[1], 3, 2, 3, 1)
                sc["overhangs"] = {"intron": list(ans[0]), "overhang_reads": [r["read_name"] for r in ans[1]], "non_spurious": [r["read_name"] for r in keep]}
            except ZeroDivisionError:
                sc["overhangs"] = {"intron": list(ans[0]), "overhang_reads": [r["read_name"] for r in ans[1]], "raises": "ZeroDivisionError"}
        # find_by_smith_waterman_realn against a contig built from the genome
        L = 90
        lt, rt_start = genome[pos - L:pos], pos + (len(ref) - 1 if len(ref) > len(alt) else 0)
        contig = Contig(lt, alt[1:] if len(alt) > len(ref) else "", genome[rt_start:rt_start + L], genome[pos - L:pos + L])
        sc["contig"] = [contig.lt_consensus_seq, contig.indel_seq, contig.rt_consensus_seq, contig.ref]
        pl = clone(pileup)
        for r in pl:
            r["is_target"] = False
        for r in pl[::7]:
            r["is_target"] = True                                     # already found by an earlier stage: left alone
        sc["pre_target"] = [r["read_name"] for r in pl if r["is_target"]]
        try:
            res = F["find_by_smith_waterman_realn"](target, contig, pl, 3, 2, 3, 1, basequalthresh)
        except Exception as e:
            continue
        sc["realn"] = [{"read_name": r["read_name"], "is_target": r.get("is_target"), "undetermined": r.get("undetermined", False),
                        "mismatches#": dig(r["mismatches"])} for r in res]
        # parse_read_by_mut_aln on the reads found, is_perfect_match against the mutant contig
        sc["parse"] = []
        mut_aligner = make_aligner(contig.get_contig_seq(), 3, 2)
        for r in res[:6]:
            a = align(mut_aligner, r["read_seq"], len(r["read_seq"]), 1)
            rr = clone([r])[0]
            try:
                F["parse_read_by_mut_aln"](a, contig, rr, target.variant_type)
            except Exception:
                continue
            sc["parse"].append({"read_name": r["read_name"], "aln": list(a), "expect": {k_: ser(rr.get(k_)) for k_ in
                                ("lt_flank", "lt_qual", "indel_seq", "rt_flank", "rt_qual", "del_pos", "del_seq")}})
        sc["perfect"] = [{"read_name": r["read_name"], "expect": bool(F["is_perfect_match"](mut_aligner, contig.get_contig_seq(), r["read_seq"]))}
                         for r in res[:10]]
        out["scenarios"].append(sc)

    n_ret = sum(1 for sc in out["scenarios"] for r in sc["retarget"] if r["expect"])
    n_grid = sum(1 for sc in out["scenarios"] if sc["grid_search"])
    n_over = sum(1 for sc in out["scenarios"] if sc["overhangs"])
    n_tgt = sum(1 for sc in out["scenarios"] for r in sc["realn"] if r["is_target"])
    print("scenarios %d (tried %d); retarget hits %d; grid_search hits %d; overhang scenarios %d; realn targets %d" % (
        len(out["scenarios"]), n_try, n_ret, n_grid, n_over, n_tgt))
    with open(sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "tests", "golden", "driver_cases.json"), "w") as f:
        json.dump(out, f)
        f.write("\n")


if __name__ == "__main__":
    main()
