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
from .sswpy import SSW, _alignment_from, _gpu


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
    reads, rid, go, ge = [], [], [], []
    for s in read_seqs:
        e = encode_dna(s.encode("utf8") if isinstance(s, str) else s)
        reads += [e, e]
        rid += [0, 1]
        go += [gap_open_penalty, len(s)]
        ge += [gap_extension_penalty, gap_extension_penalty]
    refs = [encode_dna(ref_ref.encode("utf8") if isinstance(ref_ref, str) else ref_ref),
            encode_dna(mut_ref.encode("utf8") if isinstance(mut_ref, str) else mut_ref)]
    return JobTable.from_sequences(reads, refs, np.asarray(rid, np.int32), np.asarray(go, np.int64),
                                   np.asarray(ge, np.int64), encoded=True)


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
    return [(_alignment_from(res, 2 * k), _alignment_from(res, 2 * k + 1)) for k in range(len(read_seqs))]
