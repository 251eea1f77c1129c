"""The aligner helpers and the batched-align driver of ``indelpost.localn`` on MI355X.

Mirrors (all citations into /root/reference/indelpost/localn.pyx):
  make_aligner, align                    :464-472   (same names, same arguments)
  realign_pileup_jobs / align_pileup     the SSW part of find_by_smith_waterman_realn :15-68 and
                                         is_target_by_ssw :223-259 -- two alignments per read
                                         (reference contig with (go, ge); mutant contig with
                                         gap_open=len(read)), collapsed into one GPU batch.
The string post-processing of is_target_by_ssw (is_covering_target etc., :268-459) consumes only
the returned Alignment tuples and is not part of this path.
"""
import numpy as np

from .batch import JobTable, encode_dna
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
