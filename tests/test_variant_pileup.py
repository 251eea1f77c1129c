"""indelpost_amd.variant.Variant and the pileup front-end helpers against vectors produced by executing the
reference's own class / function text (oracle/gen_variant_golden.py), plus the pysam-free BAM / FASTA reader and make_pileup on
a synthetic locus (hand-derived: no pysam in the build container)."""
import array
import json
import os

import numpy as np
import pytest

import indelpost_amd as ip
from indelpost_amd import bamio, pileup as P
from indelpost_amd.variant import Variant

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "variant_cases.json")


@pytest.fixture(scope="module")
def gold():
    with open(GOLD) as f:
        return json.load(f)


@pytest.fixture(scope="module")
def fasta(gold):
    return bamio.FastaFile({"chr1": gold["genome"]})


def vt(v):
    return [v.chrom, v.pos, v.ref, v.alt]


def test_variant_golden(gold, fasta):
    for c in gold["variants"]:
        v = Variant(*c["in"], fasta)
        assert v.variant_type == c["type"] and v.indel_seq == c["indel_seq"], c["in"]
        assert vt(v.normalize()) == c["normalized"], c["in"]
        assert bool(v.is_leftaligned) == c["is_leftaligned"] and bool(v.is_normalized) == c["is_normalized"], c["in"]
        assert bool(v.is_non_complex_indel()) == c["non_complex"], c["in"]
        assert [vt(e) for e in v.generate_equivalents()] == c["equivalents"], c["in"]
        assert [vt(e) for e in v._generate_equivalents_private()] == c["private_equivalents"], c["in"]
        assert v.left_flank() == c["left_flank"] and v.right_flank() == c["right_flank"] and v.left_flank(20, True) == c["left_flank_n20"]
        assert v.count_repeats() == c["count_repeats"] and v.count_repeats(False) == c["count_repeats_raw"], c["in"]
        assert v._get_indel_seq("I") == c["indel_seq_I"] and v._get_indel_seq("D") == c["indel_seq_D"]
        r = v._reduce_complex_indel("D" if len(c["in"][2]) > len(c["in"][3]) else "I")
        assert (vt(r) if r is not None else None) == c["reduced"], c["in"]
    n_eq = 0
    for c in gold["equal"]:
        a, b = Variant(*c["a"], fasta, skip_validation=True), Variant(*c["b"], fasta, skip_validation=True)
        assert (a == b) == c["eq"] and (hash(a) == hash(b)) == c["same_hash"], c
        n_eq += c["eq"]
    assert 100 < n_eq < 290


def test_variant_validation(fasta):
    with pytest.raises(ValueError):
        Variant("chr1", 100, "A", "A", fasta)
    with pytest.raises(ValueError):
        Variant("chr1", 100, "", "A", fasta)
    with pytest.raises(ValueError):
        Variant("chr1", 10 ** 7, "A", "AT", fasta)
    assert Variant("1", 100, "A", "AT", fasta).chrom == "chr1"                  # name formatted after the FASTA's convention
    v = Variant("chr1", 100, "AZ", "A", fasta)
    assert v.ref == "AN"                                                        # letters outside ACGTN become N
    nv = ip.NullVariant("chr1", 7, fasta)
    assert not nv and nv != Variant("chr1", 7, "A", "AT", fasta, skip_validation=True)
    with pytest.raises(NotImplementedError):
        v.query_vcf(None)


def test_pileup_helper_golden(gold, fasta):
    H = gold["helpers"]
    for c in H["mapped_subreads"]:
        assert [list(x) for x in P.get_mapped_subreads(c["cigar"], c["start"], c["end"])] == c["expect"], c
    for c in H["spliced_subreads"]:
        assert [list(x) for x in P.get_spliced_subreads(c["cigar"], c["start"], c["end"])] == c["expect"], c
    for c in H["locate_indels"]:
        ins, dels = P.locate_indels(c["cigar"], c["start"])
        assert [[list(x) for x in ins], [list(x) for x in dels]] == c["expect"], c
    for c in H["end_pos"]:
        assert P.get_end_pos(c["start"], "A" * c["flank_len"], c["cigar"]) == c["expect"], c
    for c in H["split"]:
        data = array.array("B", c["data"]) if c["kind"] == "qual" else c["data"]
        lt, rt = P.split(data, c["cigar"], c["target_pos"], c["string_pos"], c["is_for_ref"], c["reverse"])
        assert [list(lt) if c["kind"] == "qual" else lt, list(rt) if c["kind"] == "qual" else rt] == c["expect"], c
    for c in H["parse_spliced_read"]:
        r = P.parse_spliced_read(c["cigar"], c["read_start"], c["read_end"], c["pos"], c["rpos"])
        assert [bool(r[0]), list(r[1]) if r[1] else None, bool(r[2]), list(r[3]), list(r[4])] == c["expect"], c
    for c in H["is_end_dirty"]:
        assert bool(P.is_end_dirty(array.array("B", c["quals"]), c["thresh"], c["pos"], c["read_start"], c["read_end"], c["cigar"])) == c["expect"], c
    for c in H["lowqual"]:
        assert P.count_lowqual_non_ref_bases(c["read"], c["ref"], array.array("B", c["quals"]), c["cigar_list"], c["thresh"]) == c["expect"], c
    for c in H["leftalign_cigar"]:
        assert P.leftalign_cigar(c["cigar"], Variant(*c["variant"], fasta, skip_validation=True), c["read_start"]) == c["expect"], c
    for c in H["leftalign_indel_read"]:
        a = list(c["args"])
        a[9] = array.array("B", a[9])
        r = P.leftalign_indel_read(*a, fasta)
        assert [r[0], r[1], r[2], r[3], r[4], r[5], list(r[6]), list(r[7]), vt(r[8])] == c["expect"], c["args"][:7]


def _locus(tmp_path, n_reads=60, dup_every=7):
    rng = np.random.default_rng(8)
    genome = "".join("ACGT"[int(c)] for c in rng.integers(0, 4, 10000))
    pos = 5000                                                    # 1-based position of the base left of a 6 bp deletion
    segs = []
    for i in range(n_reads):
        st = int(rng.integers(pos - 140, pos - 10))              # 0-based start
        carrier = i % 3 == 0
        if carrier:
            seq, cig = genome[st:pos] + genome[pos + 6:st + 156], "%dM6D%dM" % (pos - st, 150 - (pos - st))
        else:
            seq, cig = genome[st:st + 150], "150M"
        if i % 11 == 0:                                           # soft-clipped start
            seq, cig = "GGGGG" + seq[5:], "5S" + cig.replace("%dM" % (pos - st if carrier else 150), "%dM" % ((pos - st if carrier else 150) - 5), 1)
            st += 5
        flag = (0x10 if i % 2 else 0) | (0x400 if i % dup_every == 0 else 0)
        segs.append(bamio.AlignedSegment("r%d" % i, flag, "chr1", st, 60 if i % 5 else 0, cig, seq, [30 + (i % 5)] * len(seq)))
    segs.sort(key=lambda s: s.reference_start)
    bam_path = str(tmp_path / "locus.bam")
    bamio.write_bam(bam_path, [("chr1", len(genome))], segs)
    fa_path = str(tmp_path / "locus.fa")
    with open(fa_path, "w") as f:
        f.write(">chr1 synthetic\n" + "\n".join(genome[i:i + 60] for i in range(0, len(genome), 60)) + "\n")
    return genome, pos, segs, bam_path, fa_path


def test_bam_and_fasta_reader_round_trip(tmp_path):
    genome, pos, segs, bam_path, fa_path = _locus(tmp_path)
    fa = bamio.FastaFile(fa_path)
    assert fa.references == ["chr1"] and fa.get_reference_length("chr1") == 10000 and fa.fetch("chr1", 4990, 5010) == genome[4990:5010]
    bam = bamio.AlignmentFile(bam_path)
    assert bam.references == ["chr1"] and bam.lengths == [10000]
    got = list(bam.fetch())
    assert len(got) == len(segs)
    for a, b in zip(got, segs):
        assert (a.query_name, a.flag, a.reference_start, a.mapping_quality, a.cigarstring, a.query_sequence, list(a.query_qualities)) == \
            (b.query_name, b.flag, b.reference_start, b.mapping_quality, b.cigarstring, b.query_sequence, list(b.query_qualities))
        assert a.reference_end == b.reference_end and a.is_reverse == b.is_reverse and a.is_duplicate == b.is_duplicate
    inside = [s for s in segs if s.reference_start < 5001 and s.reference_end > 4999]
    assert [s.query_name for s in bam.fetch("chr1", 4999, 5001, until_eof=True)] == [s.query_name for s in inside]
    assert bam.count("chr1", 4999, 5000, read_callback="nofilter") == sum(1 for s in segs if s.reference_start < 5000 < s.reference_end + 1 and s.reference_end > 4999)
    assert bam.count("chr1", 4999, 5000, read_callback="all") == sum(1 for s in segs if s.reference_start < 5000 and s.reference_end > 4999 and not s.is_duplicate)


def test_make_pileup_on_a_synthetic_locus(tmp_path):
    """make_pileup / dictize_read (pileup.pyx:51-266) on a BAM written and read back without pysam: hand-derived expectations"""
    genome, pos, segs, bam_path, fa_path = _locus(tmp_path)
    fa, bam = bamio.FastaFile(fa_path), bamio.AlignmentFile(bam_path)
    target = Variant("chr1", pos, genome[pos - 1:pos + 6], genome[pos - 1], fa)
    u = ip.UnsplicedLocalReference("chr1", target.pos, fa.get_reference_length("chr1"), 50, fa)
    pile, factor = ip.make_pileup(target, bam, u, True, 50, 1000, 20)
    assert factor == 1.0
    expect = [s for s in segs if not s.is_duplicate and s.reference_start < pos + 50 and s.reference_end > pos - 1 - 50]
    assert [r["read_name"] for r in pile] == [s.query_name for s in expect] and len(pile) > 30
    for r, s in zip(pile, expect):
        assert r["read"] is not None and r["read_seq"] == s.query_sequence and r["mapq"] == s.mapping_quality
        assert r["aln_start"] == s.reference_start + 1 and r["aln_end"] == s.reference_end
        clip = 5 if s.cigarstring.startswith("5S") else 0
        assert r["start_offset"] == clip and r["read_start"] == r["aln_start"] - clip and r["end_offset"] == 0 and r["read_end"] == r["aln_end"]
        carrier = "6D" in s.cigarstring
        assert len(r["D"]) == (1 if carrier else 0) and r["I"] == []
        assert r["ref_seq"] == genome[s.reference_start:s.reference_end]
        assert r["is_reference_seq"] == (not carrier and clip == 0)
        assert r["is_covering"] and r["covering_subread"] == (r["read_start"], r["read_end"]) and not r["is_spliced"]
        assert r["splice_pattern"] == ("", "") and r["intron_pattern"] == (0, 0) and not r["is_dirty"] and r["low_qual_base_num"] == 0
        if carrier:
            d = r["D"][0]
            assert d[0] == pos and d[2] == genome[pos:pos + 6] and d[8] == target      # same variant after normalisation
            assert d[1] + d[3] == r["read_seq"] and d[4][-1] == genome[pos - 1] and d[5] == genome[pos + 6:s.reference_end]
    # downsampling: depth above the threshold -> random.sample with seed 123, factor = reads before / after
    pile2, factor2 = ip.make_pileup(target, bam, u, False, 50, 20, 20)
    assert len(pile2) == int(len([s for s in segs if s.reference_start < pos + 50 and s.reference_end > pos - 51]) * (20 / bam.count("chr1", pos - 1, pos)))
    assert factor2 > 1.5 and all(r["read_name"] for r in pile2)
