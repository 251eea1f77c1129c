"""Replay of tests/golden/driver_cases.json (vectors from the reference's own function text, oracle/gen_driver_golden.py) through
indelpost_amd's composed drivers.  Used by tests/test_drivers.py (CPU: alignments from the oracle) and tests/test_gpu_driver.py
(GPU: alignments from libindelpost_hip.so)."""
import array
import json
import os
import zlib

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "driver_cases.json")
BULKY = ("read_seq", "read_qual", "ref_seq", "lt_flank", "rt_flank", "lt_ref", "rt_ref", "lt_qual", "rt_qual", "I", "D", "mismatches")


def load():
    with open(GOLDEN) as f:
        return json.load(f)


class Fasta:
    def __init__(self, seqs):
        self.seqs, self.references, self.filename = seqs, list(seqs), None

    def fetch(self, chrom, start, end):
        return self.seqs[chrom][max(0, start):max(0, end)]

    def get_reference_length(self, chrom):
        return len(self.seqs[chrom])


class Segment:
    def __init__(self, rec):
        (self.query_name, self.reference_start, self.cigarstring, self.query_sequence, quals, self.mapping_quality, rev, dup, sec) = rec
        self.query_qualities = array.array("B", [ord(c) - 33 for c in quals])
        self.is_reverse, self.is_duplicate, self.is_secondary = bool(rev), bool(dup), bool(sec)
        import re
        span = sum(int(t[:-1]) for t in re.findall(r"[0-9]+[MIDNSHPX=]", self.cigarstring) if t[-1] in "MDN=X")
        self.reference_end = self.reference_start + span


class Bam:
    def __init__(self, chrom, segs):
        self.references, self.segs = [chrom], segs

    def fetch(self, chrom, start, end, until_eof=True):
        return [g for g in self.segs if g.reference_start < end and g.reference_end > start]

    def count(self, chrom, start, end, read_callback="all"):
        return sum(1 for g in self.fetch(chrom, start, end) if not (read_callback == "all" and (g.is_duplicate or g.is_secondary)))


class Contig:
    def __init__(self, lt, mid, rt, ref):
        self.lt_consensus_seq, self.indel_seq, self.rt_consensus_seq, self.ref = lt, mid, rt, ref

    def get_contig_seq(self, split=False):
        return (self.lt_consensus_seq, self.indel_seq, self.rt_consensus_seq) if split else self.lt_consensus_seq + self.indel_seq + self.rt_consensus_seq

    def get_reference_seq(self, split=False):
        return self.ref


def ser(x):
    if hasattr(x, "ref") and hasattr(x, "alt") and hasattr(x, "chrom"):
        return [x.chrom, x.pos, x.ref, x.alt]
    if isinstance(x, dict):
        return {k: ser(v) for k, v in x.items() if k != "read"}
    if isinstance(x, (list, tuple, array.array, np.ndarray)):
        return [ser(v) for v in x]
    if isinstance(x, np.integer):
        return int(x)
    return x


def dig(x):
    return zlib.crc32(json.dumps(ser(x), sort_keys=True).encode())


def ser_read(r):
    out = {}
    for k, v in r.items():
        if k == "read":
            continue
        if k in BULKY:
            out[k + "#"] = dig(v)
        else:
            out[k] = ser(v)
    return out


def clone(pileup):
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


def same_variant(v, expect):
    return [v.chrom, v.pos, v.ref, v.alt] == expect


def replay(sc, genomes, parts=("pileup", "local_reference", "retarget", "grid_search", "overhangs", "realn", "parse", "perfect")):
    """run one scenario through the package; raises AssertionError naming the first difference.  Returns the number of checks made."""
    from indelpost_amd import localn, pileup as P, retarget as RT, varaln
    from indelpost_amd.variant import Variant
    genome = genomes[sc["genome"]]
    fa = Fasta({"chr1": genome})
    chrom, pos, ref, alt = sc["target"]
    target = Variant(chrom, pos, ref, alt, fa)
    window, bq = sc["window"], sc["basequalthresh"]
    unspl = RT.UnsplicedLocalReference(chrom, pos, len(genome), window, fa)
    bam = Bam("chr1", [Segment(r) for r in sc["segments"]])
    pile, sf = P.make_pileup(target, bam, unspl, sc["exclude_duplicates"], window, sc["downsamplethresh"], bq)
    n = 0
    if "pileup" in parts:
        assert sf == sc["sample_factor"]
        got = [ser_read(r) for r in pile]
        assert len(got) == len(sc["pileup"]), "pileup size %d, reference %d" % (len(got), len(sc["pileup"]))
        for g, e in zip(got, sc["pileup"]):
            assert g == e, "dictize_read %s: %s" % (e["read_name"], {k: (g.get(k), e[k]) for k in e if g.get(k) != e[k]})
            n += 1
    if "local_reference" in parts:
        for row in sc["local_reference"]:
            for r, e in zip(pile, row["per_read"]):
                try:
                    ref_seq, lt_len = RT.get_local_reference(target, [r], row["window"], unspl)
                    g = [zlib.crc32(ref_seq.encode()), len(ref_seq), lt_len]
                except Exception:
                    g = None
                assert g == e, "get_local_reference window %d read %s: %s vs %s" % (row["window"], r["read_name"], g, e)
                n += 1
    mapq = 1
    if "retarget" in parts:
        for case in sc["retarget"]:
            res = P.retarget(target, clone(pile), window, mapq, sc["within"], sc["cutoff"], 3, 2, case["go"], case["ge"], unspl, case["exact"])
            e = case["expect"]
            if e is None:
                assert res is None, "retarget (%d,%d,exact=%s): %s, reference None" % (case["go"], case["ge"], case["exact"], ser(res[0]))
            else:
                assert res is not None, "retarget (%d,%d,exact=%s): None, reference %s" % (case["go"], case["ge"], case["exact"], e[0])
                g = [ser(res[0]), [r["read_name"] for r in res[1]], res[2], [zlib.crc32(w.encode()) for w in res[3]], list(res[4])]
                assert g == e, "retarget (%d,%d,exact=%s): %s vs %s" % (case["go"], case["ge"], case["exact"], g[:3], e[:3])
                assert [a.reference for a in res[5]] == list(res[3])          # the aligner objects handed back carry those windows
            n += 1
    if "grid_search" in parts:
        res = varaln.grid_search(target, clone(pile), window, mapq, sc["within"], sc["cutoff"], 3, 2, [tuple(p) for p in sc["grid"]], unspl, False)
        e = sc["grid_search"]
        if e is None:
            assert res is None
        else:
            assert res is not None and ser(res[0]) == e["candidate"] and (res[2], res[3]) == (e["gap_open"], e["gap_ext"]), \
                "grid_search: %s vs %s" % (res and (ser(res[0]), res[2], res[3]), (e["candidate"], e["gap_open"], e["gap_ext"]))
            got = [ser_read(r) for r in res[1]]
            assert len(got) == len(e["reads"])
            for g, x in zip(got, e["reads"]):
                assert g == x, "update_read_info %s: %s" % (x["read_name"], {k: (g.get(k), x[k]) for k in x if g.get(k) != x[k]})
        n += 1
    if "overhangs" in parts:
        pl = clone(pile)
        ans = P.check_overhangs(pl)
        e = sc["overhangs"]
        if e is None:
            assert ans is None
        else:
            assert ans is not None and list(ans[0]) == e["intron"] and [r["read_name"] for r in ans[1]] == e["overhang_reads"]
            if "raises" in e:
                import pytest
                with pytest.raises(ZeroDivisionError):
                    P.filter_spurious_overhangs(target, ans[0], ans[1], 3, 2, 3, 1)
            else:
                keep = P.filter_spurious_overhangs(target, ans[0], ans[1], 3, 2, 3, 1)
                assert [r["read_name"] for r in keep] == e["non_spurious"]
        n += 1
    contig = Contig(*sc["contig"])
    if "realn" in parts or "parse" in parts or "perfect" in parts:
        pl = clone(pile)
        for r in pl:
            r["is_target"] = r["read_name"] in sc["pre_target"]
        res = localn.find_by_smith_waterman_realn(target, contig, pl, 3, 2, 3, 1, bq)
        if "realn" in parts:
            got = [{"read_name": r["read_name"], "is_target": r.get("is_target"), "undetermined": r.get("undetermined", False),
                    "mismatches#": dig(r["mismatches"])} for r in res]
            assert got == sc["realn"], [(g, e) for g, e in zip(got, sc["realn"]) if g != e][:3]
            n += len(got)
        by_name = {r["read_name"]: r for r in res}
        if "parse" in parts:
            from indelpost_amd.sswpy import Alignment
            for case in sc["parse"]:
                rr = clone([by_name[case["read_name"]]])[0]
                localn.parse_read_by_mut_aln(Alignment(*case["aln"]), contig, rr, target.variant_type)
                g = {k: ser(rr.get(k)) for k in case["expect"]}
                assert g == case["expect"], "parse_read_by_mut_aln %s" % case["read_name"]
                n += 1
        if "perfect" in parts:
            mut = localn.make_aligner(contig.get_contig_seq(), 3, 2)
            for case in sc["perfect"]:
                assert bool(varaln.is_perfect_match(mut, contig.get_contig_seq(), by_name[case["read_name"]]["read_seq"])) == case["expect"]
                n += 1
    return n
