"""The aligner helpers and the batched-align driver of ``indelpost.localn`` on MI355X.

Mirrors (all citations into /root/reference/indelpost/localn.pyx):
  make_aligner, align                    :464-472   (same names, same arguments)
  realign_pileup_jobs / align_pileup     the SSW part of find_by_smith_waterman_realn :15-68 and
                                         is_target_by_ssw :223-259 -- two alignments per read
                                         (reference contig with (go, ge); mutant contig with
                                         gap_open=len(read)), collapsed into one GPU batch.
  is_covering_target, is_compatible_repeats, classify_realigned / find_targets_by_ssw
                                         the classification that follows the two alignments
                                         (is_target_by_ssw :257-291, is_covering_target :293-430,
                                         is_compatible_repeats :433-459): score compare over whole result
                                         columns, string checks only for the reads that pass it.
What stays with the caller is everything that needs the BAM read (the filters of :244-249: already
target, reference-identical, mapq, is_worth_realn): they do not depend on the alignments and arrive
here as a mask.
"""
import numpy as np

from .batch import JobTable, encode_dna
from .cigar import cigar_ptrn, to_minimal_repeat_unit
from .sswpy import SSW, _alignment_from, _gpu, alignments_from


def make_aligner(ref_seq, match_score, mismatch_penalty):
    """localn.pyx:464-467"""
    aligner = SSW(match_score=match_score, mismatch_penalty=mismatch_penalty)
    aligner.setReference(ref_seq)
    return aligner


def align(aligner, read_seq, gap_open_penalty, gap_extension_penalty):
    """localn.pyx:470-472"""
    aligner.setRead(read_seq)
    return aligner.align(gap_open=gap_open_penalty, gap_extension=gap_extension_penalty)


def realign_pileup_jobs(read_seqs, mut_ref, ref_ref, gap_open_penalty, gap_extension_penalty):
    """Job table of the two alignments per read that is_target_by_ssw issues (localn.pyx:253-255):
    job 2k   = read k vs the reference contig, (gap_open, gap_ext);
    job 2k+1 = read k vs the mutant contig, gap_open = len(read) (forced ungapped), gap_ext."""
    n = len(read_seqs)
    raw = [s.encode("utf8") if isinstance(s, str) else bytes(s) for s in read_seqs]
    lens = np.fromiter((len(b) for b in raw), np.int64, n)
    # every read twice, back to back (job 2k, 2k+1); one table lookup for the whole pileup
    reads = encode_dna(b"".join([b for b in raw for _ in (0, 1)]))
    read_off = np.zeros(2 * n + 1, np.int64)
    np.cumsum(np.repeat(lens, 2), out=read_off[1:])
    rid = np.tile(np.array([0, 1], np.int32), n)
    go = np.empty(2 * n, np.int64); go[0::2] = gap_open_penalty; go[1::2] = lens
    ge = np.full(2 * n, gap_extension_penalty, np.int64)
    refs = [encode_dna(ref_ref.encode("utf8") if isinstance(ref_ref, str) else ref_ref),
            encode_dna(mut_ref.encode("utf8") if isinstance(mut_ref, str) else mut_ref)]
    ref_off = np.array([0, len(refs[0]), len(refs[0]) + len(refs[1])], np.int64)
    return JobTable(reads, read_off, np.concatenate(refs), ref_off, rid, go, ge)


def align_pileup(read_seqs, mut_ref, ref_ref, match_score, mismatch_penalty, gap_open_penalty,
                 gap_extension_penalty, device=0):
    """Batched form of the per-read loop of find_by_smith_waterman_realn (localn.pyx:47-66).

    Returns a list of (ref_aln, mut_aln) Alignment pairs, one per read, identical to
        ref_aln = align(ref_aligner, read, go, ge); mut_aln = align(mut_aligner, read, len(read), ge)
    """
    if not read_seqs:
        return []
    jobs = realign_pileup_jobs(read_seqs, mut_ref, ref_ref, gap_open_penalty, gap_extension_penalty)
    g = _gpu(device)
    from .batch import dna_score_matrix
    g.set_scoring(matrix=dna_score_matrix(match_score, mismatch_penalty), flag=1, score_size=2)
    res = g.align(jobs)
    alns = alignments_from(res)
    return list(zip(alns[0::2], alns[1::2]))


def is_compatible_repeats(seq, repeat_unit, expected_n_repeats, is_left):
    """localn.pyx:433-459: walk whole repeat units off the indel-facing end of a flank (the left flank is read
    right to left).  Incompatible when the flank is nothing but repeats, or holds some but not the expected number."""
    if is_left:
        seq, repeat_unit = seq[::-1], repeat_unit[::-1]
    n, cnt = len(repeat_unit), 0
    while seq and seq[:n] == repeat_unit:
        seq = seq[n:]
        cnt += 1
    if not seq:
        return False
    return not (cnt and cnt != expected_n_repeats)


def is_covering_target(readname, read_seq, indel_seq, mut_ref_lt, mut_ref_mid, mut_ref_rt, mut_aln_cigar, read_seq_len,
                       ref_aln_start, ref_aln_end, read_aln_start, read_aln_end, n_repeats):
    """localn.pyx:293-430.  Does the read's UNGAPPED alignment to the mutant contig (lt + mid + rt; mid = inserted
    sequence, empty for a deletion) really span the indel?  1 yes, 0 no, -1 undetermined (flank repeats disagree)."""
    toks = cigar_ptrn.findall(mut_aln_cigar)
    if len(toks) > 1:
        return 0
    unit = to_minimal_repeat_unit(indel_seq)
    lt_len, mid_len = len(mut_ref_lt), len(mut_ref_mid)
    consumed = read_aln_end - read_aln_start + 1
    from_read_start, to_read_end = read_aln_start == 0, read_aln_end == read_seq_len - 1
    if ref_aln_end < lt_len or lt_len + mid_len <= ref_aln_start:       # the alignment misses the indel altogether
        return 0
    if mid_len:                                                        # insertion
        if ref_aln_start < lt_len:
            lt_use = lt_len - ref_aln_start
            if consumed > lt_use + mid_len:                            # through the insertion and out the other side
                a = read_aln_start + lt_use
                lt_ok = is_compatible_repeats(read_seq[read_aln_start:a], unit, n_repeats, is_left=True)
                rt_ok = is_compatible_repeats(read_seq[a + mid_len:read_aln_end + 1], unit, n_repeats, is_left=False)
                return 1 if (lt_ok and rt_ok) else -1
            if to_read_end:
                return 1
            rt_use = consumed - lt_use                                  # entered from the left, stopped inside the insertion
            return 1 if mut_ref_mid[:rt_use] == read_seq[-rt_use:] else 0
        if from_read_start:                                             # no left flank aligned at all
            return 1
        lt_use = lt_len + mid_len - ref_aln_start                       # entered from the right, stopped inside the insertion
        return 1 if mut_ref_mid[-lt_use:] == read_seq[:lt_use] else 0
    lt_use = lt_len - ref_aln_start                                     # deletion
    rt_use = consumed - lt_use
    a = read_aln_start + lt_use
    if not (is_compatible_repeats(read_seq[read_aln_start:a], unit, n_repeats, is_left=True)
            and is_compatible_repeats(read_seq[a:read_aln_end], unit, n_repeats, is_left=False)):
        return -1
    if lt_use <= rt_use:
        return 1 if (from_read_start or lt_use > 2) else 0
    return 1 if (to_read_end or rt_use > 2) else 0


def classify_realigned(read_seqs, pairs, indel_seq, mut_ref_lt, mut_ref_mid, mut_ref_rt, n_repeats, read_names=None):
    """The verdict of is_target_by_ssw (localn.pyx:257-291) for every realigned read at once.

    pairs[k] = (ref_aln, mut_aln) as align_pileup returns them.  Returns (is_target, undetermined): two bool arrays.
    The score compare runs over the whole columns; only the reads whose mutant-contig score beats the reference-contig
    score go through the string checks of is_covering_target."""
    n = len(pairs)
    is_target, undetermined = np.zeros(n, bool), np.zeros(n, bool)
    if n == 0:
        return is_target, undetermined
    ref_score = np.fromiter((p[0].optimal_score for p in pairs), np.int64, n)
    mut_score = np.fromiter((p[1].optimal_score for p in pairs), np.int64, n)
    for k in np.flatnonzero(mut_score > ref_score):                    # localn.pyx:257
        m = pairs[k][1]
        seq = read_seqs[k]
        v = is_covering_target(read_names[k] if read_names else "", seq, indel_seq, mut_ref_lt, mut_ref_mid, mut_ref_rt, m.CIGAR,
                               len(seq), m.reference_start, m.reference_end, m.read_start, m.read_end, n_repeats)
        if v == 1:
            is_target[k] = True
        elif v == -1:
            undetermined[k] = True
    return is_target, undetermined


def find_targets_by_ssw(read_seqs, realign_mask, indel_seq, n_repeats, mut_ref_lt, mut_ref_mid, mut_ref_rt, ref_ref,
                        match_score, mismatch_penalty, gap_open_penalty, gap_extension_penalty, device=0):
    """find_by_smith_waterman_realn (localn.pyx:15-68) end to end for one locus: the reads selected by `realign_mask`
    (the alignment-independent filters of localn.pyx:244-249, decided by the caller) are aligned to the reference and
    the mutant contig in ONE GPU batch and classified.  Returns (is_target, undetermined, pairs): bool arrays over ALL
    reads (False where the mask is False) and the Alignment pairs (None where not aligned)."""
    n = len(read_seqs)
    mask = np.asarray(realign_mask, bool)
    idx = np.flatnonzero(mask)
    sel = [read_seqs[i] for i in idx]
    pairs = align_pileup(sel, mut_ref_lt + mut_ref_mid + mut_ref_rt, ref_ref, match_score, mismatch_penalty,
                         gap_open_penalty, gap_extension_penalty, device)
    t, u = classify_realigned(sel, pairs, indel_seq, mut_ref_lt, mut_ref_mid, mut_ref_rt, n_repeats)
    is_target, undetermined = np.zeros(n, bool), np.zeros(n, bool)
    is_target[idx], undetermined[idx] = t, u
    out = [None] * n
    for i, p in zip(idx, pairs):
        out[i] = p
    return is_target, undetermined, out
