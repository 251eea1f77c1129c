"""Host-side decoders and the target classification against vectors produced by the reference's
OWN function bodies (oracle/gen_decoder_golden.py executes the text of merge_consecutive_gaps / make_insertion_first /
to_minimal_repeat_unit / findall_indels / is_compatible_repeats / is_covering_target / generate_grid as it stands in
/root/reference and records inputs and outputs), plus hand-derived known answers on the CIGAR KATs of SURVEY.md 8c."""
import json
import os

import pytest

from indelpost_amd import cigar as C
from indelpost_amd import localn, retarget
from indelpost_amd.sswpy import Alignment

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "decoder_cases.json")


@pytest.fixture(scope="module")
def gold():
    with open(GOLD) as f:
        return json.load(f)


def test_cigar_rewrites(gold):
    for c in gold["cigar"]:
        toks = C.cigar_ptrn.findall(c["cigar"])
        assert C.merge_consecutive_gaps(toks) == c["merged"], c["cigar"]
        assert C.make_insertion_first(c["cigar"]) == c["insertion_first"], c["cigar"]
    assert C.make_insertion_first("10M2D3I5M") == "10M3I2D5M" and C.make_insertion_first("10M3I2D5M") == "10M3I2D5M"
    assert C.merge_consecutive_gaps(["5M", "2I", "3D"]) == ["5M", "2I", "3D"]        # the reference's end-of-list quirk
    assert C.merge_consecutive_gaps(["5M", "2I", "3D", "1I"]) == ["5M", "2I3D", "1I"]


def test_minimal_repeat_unit(gold):
    for c in gold["repeat_unit"]:
        assert C.to_minimal_repeat_unit(c["seq"]) == c["unit"], c["seq"]
    assert C.to_minimal_repeat_unit("ATATAT") == "AT" and C.to_minimal_repeat_unit("ATA") == "ATA" and C.to_minimal_repeat_unit("") == ""


def test_findall_indels_golden(gold):
    n_indels = 0
    for c in gold["findall_indels"]:
        got = C.findall_indels(Alignment(*c["aln"]), c["genome_aln_pos"], c["ref_seq"], c["read_seq"], report_snvs=c["report_snvs"],
                               basequals=c["basequals"])
        if c["report_snvs"]:
            assert [list(got[0]), list(got[1])] == c["expect"]
            n_indels += len(got[0])
        else:
            assert got == c["expect"]
            n_indels += len(got)
    assert n_indels > 100


def test_findall_indels_known_answers():
    """SURVEY.md 8c KATs: reference SSW(3,2), ref ACGTACGTTTGACCAGT."""
    ref = "ACGTACGTTTGACCAGT"
    d = C.findall_indels(Alignment("5M1D11M", 45, 3, 0, 16, 0, 15), 101, ref, "ACGTAGTTTGACCAGT")
    assert len(d) == 1 and d[0]["indel_type"] == "D" and d[0]["pos"] == 105 and d[0]["del_seq"] == "C"
    assert d[0]["lt_ref"] == "ACGTA" and d[0]["rt_ref"] == "GTTTGACCAGT" and d[0]["lt_flank"] == "ACGTA" and d[0]["rt_flank"] == "GTTTGACCAGT"
    assert (d[0]["ref_idx"], d[0]["read_idx"], d[0]["lt_clipped"], d[0]["rt_clipped"]) == (5, 5, "", "")
    i = C.findall_indels(Alignment("8M3I9M", 46, 3, 0, 16, 0, 19), 101, ref, "ACGTACGTCCCTTGACCAGT")
    assert len(i) == 1 and i[0]["indel_type"] == "I" and i[0]["pos"] == 108 and i[0]["indel_seq"] == "CCC" and i[0]["rt_flank"] == "TTGACCAGT"
    # window slice KAT: align(3,1,start_idx=4,end_idx=12) -> '2M3I6M', coordinates relative to the slice
    s = C.findall_indels(Alignment("2M3I6M", 19, 0, 0, 7, 0, 10), 105, ref[4:12], "ACGTAGTTTGACCAGT"[:11])
    assert s[0]["indel_type"] == "I" and s[0]["pos"] == 106 and s[0]["read_idx"] == 2
    # the adversarial KAT of SURVEY 8c: 7 gaps, insertions and deletions both
    cg = "2M1D12M1D5M1D17M1I3M1D12M3I3M2I3M2D4M"
    read = "CCATGCTCACTCCAACCCGGCCCTGAGTCCGAGGAGAGGGGGCTTCAGAGTATTGGGTATGTACCTGGACTGGCA"
    w = "ATCACAGTCTACACTGCTCACTCCAACCCCGGCCCCTGAGTCCGAGGAGAGGGTGCTTCAGAGTATGTATACCACTGGGTAGGATACGGCGGAGGGCACGTCAATACGGTTCAATGCCCT"
    a = C.findall_indels(Alignment(cg, 49, 37, 11, 77, 1, 67), 1 + 11, w, read)   # window starts at genome position 1
    assert [x["indel_type"] for x in a] == ["D", "D", "D", "I", "D", "I", "I", "D"]
    assert sum(len(x["indel_seq"]) for x in a) == 6 and sum(len(x.get("del_seq", "")) for x in a) == 6
    assert all(x["lt_clipped"] == "C" for x in a) and a[0]["pos"] == 13 and a[-1]["rt_clipped"] == read[68:]


def test_compatible_repeats_and_covering_target(gold):
    for c in gold["compatible_repeats"]:
        assert localn.is_compatible_repeats(c["seq"], c["unit"], c["n"], c["is_left"]) == c["expect"], c
    seen = set()
    for c in gold["covering_target"]:
        assert int(localn.is_covering_target(*c["args"])) == c["expect"], c["args"]
        seen.add(c["expect"])
    assert seen == {0, 1, -1}


def test_generate_grid(gold):
    for c in gold["grid"]:
        assert [list(x) for x in retarget.generate_grid(c["auto"], c["gap_open"], c["gap_ext"], c["indel_len"])] == c["expect"], c


def test_local_reference_windows():
    """UnsplicedLocalReference / get_local_reference (local_reference.pyx, utilities.pyx:505-586): hand-derived."""
    class Fasta:
        def __init__(self, seq):
            self.seq = seq

        def fetch(self, chrom, start, end):
            return self.seq[start:end]

        def get_reference_length(self, chrom):
            return len(self.seq)

    class Target:
        pass
    g = "".join("ACGT"[(i * 7 + i // 5) % 4] for i in range(3000))
    t = Target()
    t.chrom, t.pos, t.reference = "1", 1500, Fasta(g)
    u = retarget.UnsplicedLocalReference("1", 1500, len(g), 50, t.reference)
    assert u.local_ref_start == 1000 and u.unspliced_local_reference == g[1000:2000]
    w, lt = retarget.get_local_reference(t, [{"splice_pattern": ("", "")}], 50, u)
    assert w == g[1350:1650] and lt == 150                                     # pos +- 3 windows; 6 x window bases
    t2 = Target()
    t2.chrom, t2.pos, t2.reference = "1", 40, t.reference
    u2 = retarget.UnsplicedLocalReference("1", 40, len(g), 50, t.reference)
    w2, lt2 = retarget.get_local_reference(t2, [], 50, u2)
    assert w2 == g[0:190] and lt2 == 40                                        # clipped at the chromosome start
    # one intron 1601-1700 on the right of the target: exons stitched, 2 windows beyond the outer boundaries
    t3 = Target()
    t3.chrom, t3.pos, t3.reference = "1", 1550, t.reference
    w3, lt3 = retarget.get_local_reference(t3, [{"splice_pattern": ("", "1601-1700")}, {"splice_pattern": ("", "")}], 50, u)
    assert w3 == g[1501:1600] + g[1700:1800] and lt3 == 99 - (1600 - 1550)    # window bases up to and including the target position
    spans = retarget.get_local_reference(t3, [{"splice_pattern": ("", "1601-1700")}], 50, u, splice_pattern_only=True)
    assert spans == ((1501, 1600), (1701, 1800))
    assert retarget.most_common(["b", "a", "b", "a", "c"]) == "a"


def test_indel_candidates_filters():
    """the per-alignment part of retarget (pileup.pyx:650-711): filters and ref/alt construction, hand-derived"""
    ref = "ACGTACGTTTGACCAGT" * 6
    read = ref[3:40] + ref[43:90]                                              # 3 bp deletion after window index 39
    aln = Alignment("37M3D47M", 200, 0, 3, 89, 0, 83)
    c, cx = retarget.indel_candidates(aln, read, ref, 1000, "D", 1003, 1090, 50)
    assert c == [(1039, ref[39:43], ref[39])] and cx is False                  # pos = base left of the deletion, ref = that base + deleted bases
    assert retarget.indel_candidates(aln, read, ref, 1000, "I", 1003, 1090, 50) == ([], False)
    assert retarget.indel_candidates(aln._replace(CIGAR="84M"), read, ref, 1000, "D", 1003, 1090, 50) == ([], False)      # no gap
    assert retarget.indel_candidates(aln._replace(read_end=40), read, ref, 1000, "D", 1003, 1090, 50) == ([], False)      # < 70 % aligned
    assert retarget.indel_candidates(aln._replace(CIGAR=None), read, ref, 1000, "D", 1003, 1090, 50) == ([], False)
    # an indel close to a read end only counts when it is the target itself
    near = retarget.indel_candidates(aln, read, ref, 1000, "D", 1037, 1090, 50)
    assert near == ([], False)
    assert retarget.indel_candidates(aln, read, ref, 1000, "D", 1037, 1090, 50, is_target=lambda p, r, a: p == 1039)[0] == c
    # complex: insertion and deletion at one position are reported as one ref>alt
    aln2 = Alignment("37M2I3D45M", 180, 0, 3, 87, 0, 83)
    read2 = ref[3:40] + "GG" + ref[43:88]
    c2, cx2 = retarget.indel_candidates(aln2, read2, ref, 1000, "D", 1003, 1090, 50)
    assert cx2 is True and c2 == [(1039, ref[39:43], ref[39] + "GG")]


def test_cigar_surgery_golden(gold):
    """split_cigar / trim_ref_flank / update_cigar / update_read_positions (utilities.pyx:330-357, pileup.pyx:913-1048)
    against the reference's own function bodies, incl. splice junctions re-inserted as N and soft clips"""
    from indelpost_amd import pileup as P
    for c in gold["split_cigar"]:
        got = P.split_cigar(c["cigar"], c["target_pos"], c["start"])
        assert (None if got is None else [list(got[0]), list(got[1])]) == c["expect"], c
    for c in gold["trim_ref_flank"]:
        assert P.trim_ref_flank(c["flank"], c["cigar"], c["left"]) == c["expect"], c
    n_with_n = 0
    for c in gold["update_cigar"]:
        got = P.update_cigar("", list(c["realn_cigar"]), c["start_pos"], tuple(c["splice"]), c["clipped"], c["left"])
        assert got == c["expect"], c
        n_with_n += any(t.endswith("N") for t in got)
    assert n_with_n > 40
    for c in gold["read_positions"]:
        read = {"lt_cigar": c["lt_cigar"], "rt_cigar": c["rt_cigar"]}
        P.update_read_positions(read, c["target_pos"])
        assert {k: read[k] for k in c["expect"]} == c["expect"], c


def test_update_read_info_realn_known_answer():
    """the realignment branch of update_read_info (pileup.pyx:847-911) on the 5M1D11M KAT of SURVEY 8c, hand-derived"""
    from indelpost_amd import pileup as P
    ref, seq = "ACGTACGTTTGACCAGT", "ACGTAGTTTGACCAGT"
    read = {"read_seq": seq, "read_qual": list(range(16)), "cigar_string": "16M", "read_start": 1001, "splice_pattern": ("", "")}
    aln = Alignment("5M1D11M", 45, 3, 0, 16, 0, 15)
    out = P.update_read_info_realn(read, aln, ref, 1001, 1005, "", False, lambda p, r, a: (p, r, a) == (1005, "AC", "A"))
    assert out["cigar_updated"] and out["is_target"] and out["cigar_string"] == "5M1D11M" and out["cigar_list"] == ["5M", "1D", "11M"]
    assert out["lt_flank"] == "ACGTA" and out["rt_flank"] == "GTTTGACCAGT" and out["indel_seq"] == ""
    assert out["lt_ref"] == "ACGTA" and out["rt_ref"] == "GTTTGACCAGT" and out["lt_qual"] == [0, 1, 2, 3, 4] and out["rt_qual"] == list(range(5, 16))
    assert (out["read_start"], out["read_end"], out["aln_start"], out["aln_end"], out["start_offset"], out["end_offset"]) == (1001, 1017, 1001, 1017, 0, 0)
    fresh = {"read_seq": seq, "read_qual": list(range(16)), "cigar_string": "16M", "read_start": 1001, "splice_pattern": ("", "")}
    miss = P.update_read_info_realn(fresh, aln, ref, 1001, 1005, "", False, lambda p, r, a: False)
    assert miss["cigar_updated"] is False and "lt_cigar" not in miss
