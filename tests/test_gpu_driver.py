"""The batched callers of the aligner on a real MI355X: retarget x grid_search job tables with
per-read windows, the overhang filter's two alignments per read, is_perfect_match, and the end-to-end target
classification.  Every SSW job is checked bit for bit against the oracle (the restatement pinned to the reference's
ssw.c); the decoders and the classification on top of the alignments are pinned in tests/test_decoders.py."""
import numpy as np
import pytest

import indelpost_amd as ip
from indelpost_amd import retarget
from indelpost_amd.sswpy import Alignment

pytestmark = pytest.mark.gpu

LET = "ACGT"


def _s(a):
    return "".join(LET[int(c)] for c in a)


def _oracle_alignment(oracle_mod, port, read, ref, mat, go, ge):
    e = port.align(oracle_mod.encode(read), oracle_mod.encode(ref), mat, go, ge)
    return Alignment(oracle_mod.cigar_string(e["cigar"]), e["score1"], e["score2"], e["ref_begin1"], e["ref_end1"],
                     e["read_begin1"], e["read_end1"])


class _Fasta:
    def __init__(self, seq):
        self.seq = seq

    def fetch(self, chrom, start, end):
        return self.seq[start:end]

    def get_reference_length(self, chrom):
        return len(self.seq)


class _Target:
    pass


def _locus(rng, n_reads=120, ins=False):
    """10 kb genome, one planted indel at 5000 (6 bp deletion or 5 bp insertion), reads tiling +-150 bp, 40 % carriers"""
    genome = _s(rng.integers(0, 4, 10000))
    pos = 5000
    if ins:
        alt = genome[:pos] + "GATTA" + genome[pos:]
        indel_seq = "GATTA"
    else:
        alt = genome[:pos] + genome[pos + 6:]
        indel_seq = genome[pos:pos + 6]
    reads, starts = [], []
    for i in range(n_reads):
        st = int(rng.integers(pos - 140, pos - 10))
        src = alt if rng.random() < 0.4 else genome
        r = list(src[st:st + 150])
        for k in np.flatnonzero(rng.random(150) < 0.01):
            r[k] = LET[int(rng.integers(0, 4))]
        reads.append("".join(r))
        starts.append(st)
    return genome, pos, indel_seq, reads, starts


def test_gpu_retarget_grid_equals_per_call_loop(gpu, oracle_mod, port):
    """retarget x grid_search (pileup.pyx:639-648 x varaln.pyx:1163-1178) as one job table: per-read windows cut with
    get_local_reference, all six penalty pairs; every alignment equals the reference's per-call result."""
    rng = np.random.default_rng(31)
    genome, pos, indel_seq, reads, starts = _locus(rng)
    t = _Target()
    t.chrom, t.reference = "1", _Fasta(genome)
    grid = retarget.generate_grid(True, 3, 1, len(indel_seq))
    assert grid == [(3, 1), (3, 0), (5, 1), (5, 0), (4, 1), (4, 0)]
    wins, ref_starts = [], []
    for st in starts:                                   # each read gets ITS window: here centred on its own mid-point
        t.pos = st + 75
        u = retarget.UnsplicedLocalReference("1", t.pos, len(genome), 50, t.reference)
        w, lt = retarget.get_local_reference(t, [{"splice_pattern": ("", "")}], 50, u)
        wins.append(w)
        ref_starts.append(t.pos + 1 - lt)                # pileup.pyx:648
    jobs = retarget.retarget_jobs(reads, wins, grid)
    assert jobs.n_jobs == len(reads) * 6 and (np.diff(jobs.ref_off) == 300).all()
    alns = retarget.grid_align(reads, wins, grid, 3, 2)
    mat = oracle_mod.dna_matrix(3, 2)
    n_cand = 0
    for g, (go, ge) in enumerate(grid):
        assert len(alns[g]) == len(reads)
        for k, (r, w) in enumerate(zip(reads, wins)):
            assert alns[g][k] == _oracle_alignment(oracle_mod, port, r, w, mat, go, ge), (g, k)
            c, _ = retarget.indel_candidates(alns[g][k], r, w, ref_starts[k], "D", starts[k] + 1, starts[k] + 150, 50)
            for p, ref, alt in c:                        # every candidate is a real deletion of the window at that position
                assert len(ref) > len(alt) == 1 and genome[p - 1:p - 1 + len(ref)] == ref
            n_cand += sum(1 for p, ref, alt in c if p == pos and len(ref) == 7)
    assert n_cand > 50                                   # the planted 6 bp deletion is found again and again


def test_gpu_find_targets_end_to_end(gpu, oracle_mod, port):
    """find_by_smith_waterman_realn for one locus (localn.pyx:15-68): two alignments per realignable read in one batch,
    then the classification; the alignments equal the oracle's, the verdicts equal is_covering_target applied to them."""
    for ins in (False, True):
        rng = np.random.default_rng(41 + ins)
        genome, pos, indel_seq, reads, starts = _locus(rng, 150, ins)
        lt, rt = genome[pos - 150:pos], (genome[pos:pos + 150] if ins else genome[pos + 6:pos + 156])
        mid = indel_seq if ins else ""
        ref_ref = genome[pos - 150:pos + 156]
        mask = rng.random(len(reads)) < 0.8              # the caller's alignment-independent filters (localn.pyx:244-249)
        is_t, und, pairs = ip.find_targets_by_ssw(reads, mask, indel_seq, 0, lt, mid, rt, ref_ref, 3, 2, 3, 1)
        mat = oracle_mod.dna_matrix(3, 2)
        for k, r in enumerate(reads):
            if not mask[k]:
                assert pairs[k] is None and not is_t[k] and not und[k]
                continue
            ra = _oracle_alignment(oracle_mod, port, r, ref_ref, mat, 3, 1)
            ma = _oracle_alignment(oracle_mod, port, r, lt + mid + rt, mat, len(r), 1)
            assert pairs[k] == (ra, ma), k
            want = 0
            if ma.optimal_score > ra.optimal_score:
                want = ip.is_covering_target("", r, indel_seq, lt, mid, rt, ma.CIGAR, len(r), ma.reference_start, ma.reference_end,
                                             ma.read_start, ma.read_end, 0)
            assert (bool(is_t[k]), bool(und[k])) == (want == 1, want == -1), k
        assert 20 < int(is_t.sum()) < 100                # about 40 % of the masked reads carry the indel


def test_gpu_overhang_and_perfect_match_jobs(gpu, oracle_mod, port):
    """the other live SSW call sites as batch jobs: the overhang filter (2 per read, pileup.pyx:540-545) and
    is_perfect_match (gap_open = gap_ext = len(read), varaln.pyx:1228-1234 -- the gap_open <= gap_ext regime)."""
    rng = np.random.default_rng(51)
    genome = _s(rng.integers(0, 4, 3000))
    g_ref = genome[1400:1600]                            # target.pos +- 100
    j_ref = genome[1000:1100] + genome[1500:1600]        # exon-exon junction: intron 1100..1500 spliced out
    reads = []
    for i in range(60):
        if i % 2:
            st = int(rng.integers(1010, 1090))           # spliced read: spans the junction
            src = genome[st:1100] + genome[1500:1500 + 100 - (1100 - st)]
        else:
            st = int(rng.integers(1400, 1500))
            src = genome[st:st + 100]
            if i % 4 == 0:
                src = src[:50] + src[53:]                # a genomic read with a 3 bp deletion: one gap, not ruled out by alignment alone
        r = list(src)
        for k in np.flatnonzero(rng.random(len(r)) < 0.02):
            r[k] = LET[int(rng.integers(0, 4))]
        reads.append("".join(r))
    verdicts, pairs = retarget.overhang_alignment_verdicts(reads, g_ref, j_ref, 3, 2, 3, 1)
    mat = oracle_mod.dna_matrix(3, 2)
    for k, r in enumerate(reads):
        ga = _oracle_alignment(oracle_mod, port, r, g_ref, mat, 3, 1)
        ja = _oracle_alignment(oracle_mod, port, r, j_ref, mat, 3, 1)
        assert pairs[k] == (ga, ja), k
        if ga.optimal_score <= ja.optimal_score:
            assert verdicts[k] is False                  # pileup.pyx:548-549
    assert sum(v is False for v in verdicts) >= 25 and any(v is None for v in verdicts)
    contig = genome[1400:1700]
    pm_reads = [contig[20:120], contig[50:150][:40] + "A" + contig[50:150][41:], contig[100:260], genome[100:180]]
    got = retarget.perfect_match_batch(pm_reads, contig, 3, 2)
    for r, ok in zip(pm_reads, got):
        a = _oracle_alignment(oracle_mod, port, r, contig, mat, len(r), len(r))
        assert ok == (contig[a.reference_start:a.reference_end] == r[a.read_start:a.read_end])
    assert got[0] is True and got[2] is True


def test_gpu_update_reads_batch(gpu, oracle_mod, port):
    """update_read_info's realignment branch for a whole grid_search response (pileup.pyx:847-911, varaln.pyx:1200-1216):
    one GPU batch, then the CIGAR surgery; reads carrying the planted deletion come back with it in their BAM CIGAR."""
    rng = np.random.default_rng(61)
    genome, pos, indel_seq, reads, starts = _locus(rng, 80)
    wins = [genome[st + 75 - 150:st + 75 + 150] for st in starts]
    ref_starts = [st + 75 - 150 + 1 for st in starts]                        # 1-based genome position of window index 0
    dicts = [{"read_seq": r, "read_qual": [30] * len(r), "cigar_string": "%dM" % len(r), "read_start": st + 1, "splice_pattern": ("", "")}
             for r, st in zip(reads, starts)]
    alt_genome = genome[:pos] + genome[pos + 6:]
    # the caller's `Variant(...) == candidate` (normalised equality): same haplotype after applying (p, ref, alt)
    same = lambda p, ref, alt: len(ref) == 7 and genome[:p - 1] + alt + genome[p - 1 + len(ref):] == alt_genome
    # the candidate's normalised (left-aligned) position
    cpos = pos
    while genome[cpos - 1] == genome[cpos + 5]:
        cpos -= 1
    out = ip.update_reads_batch(dicts, wins, ref_starts, cpos, indel_seq, False, same, 3, 2, 3, 1)
    mat = oracle_mod.dna_matrix(3, 2)
    n_upd = 0
    for d, r, w in zip(out, reads, wins):
        a = _oracle_alignment(oracle_mod, port, r, w, mat, 3, 1)
        if d["cigar_updated"]:
            n_upd += 1
            assert "6D" in d["cigar_string"] and "6D" in a.CIGAR and d["is_target"] and d["lt_flank"] + d["rt_flank"] in r
            assert d["read_end"] - d["read_start"] + 1 == sum(int(t[:-1]) for t in d["cigar_list"] if t[-1] != "I")
        else:
            assert a.CIGAR is None or "6D" not in a.CIGAR or d["cigar_string"] == "%dM" % len(r)
    assert n_upd > 15


def test_gpu_bam_locus_to_target_reads(gpu, oracle_mod, port, tmp_path):
    """BASELINE configs[2] in miniature, files first: a BAM and a FASTA on disk -> make_pileup (pileup.pyx:51-110) with the
    pysam-free reader -> the reads without the target in their CIGAR are realigned in one GPU batch (localn.pyx:15-68).
    Half the carriers are written to the BAM with the deletion soft-clipped away, as a mapper would leave them: the
    realignment must find them again."""
    from tests.test_variant_pileup import _locus as bam_locus
    from indelpost_amd import bamio
    from indelpost_amd.variant import Variant
    genome, pos, segs, bam_path, fa_path = bam_locus(tmp_path, n_reads=90)
    hidden = set()
    for i, s in enumerate(segs):                         # hide the deletion of every other carrier behind a soft clip
        if "6D" in s.cigarstring and "S" not in s.cigarstring and i % 2:
            lt = s.cigartuples[0][1]
            if lt >= 30:
                s.cigartuples = [(0, lt), (4, 150 - lt)]
                hidden.add(s.query_name)
    bamio.write_bam(bam_path, [("chr1", len(genome))], segs)
    assert len(hidden) > 5
    fa, bam = bamio.FastaFile(fa_path), bamio.AlignmentFile(bam_path)
    target = Variant("chr1", pos, genome[pos - 1:pos + 6], genome[pos - 1], fa)
    u = ip.UnsplicedLocalReference("chr1", target.pos, len(genome), 50, fa)
    pile, _ = ip.make_pileup(target, bam, u, True, 50, 1000, 20)
    by_cigar = np.array([any(d[8] == target for d in r["D"]) for r in pile])
    mask = ~by_cigar
    lt, rt, ref_ref = genome[pos - 150:pos], genome[pos + 6:pos + 156], genome[pos - 150:pos + 156]
    reads = [r["read_seq"] for r in pile]
    is_t, und, pairs = ip.find_targets_by_ssw(reads, mask, target.indel_seq, target.count_repeats(), lt, "", rt, ref_ref, 3, 2, 3, 1)
    mat = oracle_mod.dna_matrix(3, 2)
    for k, r in enumerate(reads):
        if mask[k]:
            assert pairs[k] == (_oracle_alignment(oracle_mod, port, r, ref_ref, mat, 3, 1),
                                _oracle_alignment(oracle_mod, port, r, lt + rt, mat, len(r), 1)), k
    found = {pile[k]["read_name"] for k in np.flatnonzero(is_t)}
    assert found == {n for n in hidden if n in {r["read_name"] for r in pile}} and not und.any()


def test_gpu_composed_drivers_replay_reference_vectors():
    """retarget / grid_search / update_read_info, the overhang filter, find_by_smith_waterman_realn and the pileup front-end as
    ONE stack on the GPU: the vectors of tests/golden/driver_cases.json (the reference's own function text, run against duck-typed
    BAM / FASTA with the oracle as aligner: oracle/gen_driver_golden.py) replayed through libindelpost_hip.so -- whole return
    values and whole read dicts compared."""
    from tests import driver_replay as DR
    cases = DR.load()
    n = sum(DR.replay(sc, cases["genomes"]) for sc in cases["scenarios"])
    assert n > 4000
