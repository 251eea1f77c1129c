"""The alignment-driven searches of ``indelpost.varaln`` with the reference's signatures.

Mirrors (citations into /root/reference/indelpost/varaln.pyx):
  generate_grid      :1122-1145  the (gap_open, gap_ext) pairs to try
  grid_search        :1148-1225  retarget under every pair, the best response's reads updated
  is_perfect_match   :1228-1234
grid_search is where the reference spends its alignments: retarget once per pair (up to 7), each aligning every
non-reference read against its own window, then update_read_info re-aligning the winning reads.  Here all pairs are ONE
GPU batch per recursion level (pileup.retarget_many) and the winning reads reuse the alignments already made.
Parity: pinned by vectors from the reference's own function text (oracle/gen_driver_golden.py).
"""
from .pileup import _RetargetSearch, run_retarget_searches, update_read_info
from .retarget import generate_grid as _grid_of_len


def generate_grid(auto_adjust_extension_penalty, gap_open_penalty, gap_extension_penalty, target):
    """varaln.pyx:1122-1145 (the reference takes the target Variant; retarget.generate_grid takes len(target.indel_seq))"""
    return _grid_of_len(auto_adjust_extension_penalty, gap_open_penalty, gap_extension_penalty, len(target.indel_seq))


def _finish_grid_search(search, grid):
    """varaln.pyx:1180-1225 with the responses of every pair in hand: the best response's reads updated"""
    responses, used = search.results, search.used
    best, best_score = None, None
    for h, res in enumerate(responses):
        if not res:
            continue
        score = res[2] * len(res[1]) if res[2] == 1.0 else res[2]      # an exact match counts once per supporting read
        if best is None or score > best_score:                          # (first pair wins ties, as scores.index(max(scores)) does)
            best, best_score = h, score
    if best is None:
        return None
    candidate, reads, _, ref_seqs, ref_starts, aligners = responses[best]
    gap_open_penalty, gap_extension_penalty = grid[best]
    updated = [update_read_info(read, candidate, False, gap_open_penalty, gap_extension_penalty, aligner, ref_seq, ref_start,
                                aln=used[best].get(id(read)))
               for read, aligner, ref_seq, ref_start in zip(reads, aligners, ref_seqs, ref_starts)]
    return candidate, updated, gap_open_penalty, gap_extension_penalty


def grid_search(target, pileup, window, mapq_thresh, within, retarget_cutoff, match_score, mismatch_penalty, grid, unspl_loc_ref,
                exact_match_for_shiftable, device=0):
    """varaln.pyx:1148-1225, same arguments, same return value: None, or (candidate, updated reads, gap_open, gap_ext) of the
    pair whose response scores best."""
    return grid_search_many([(target, pileup, window, mapq_thresh, within, retarget_cutoff, match_score, mismatch_penalty, grid, unspl_loc_ref,
                              exact_match_for_shiftable)], device)[0]


def grid_search_many(requests, device=0):
    """grid_search for MANY loci at once: requests = argument tuples of grid_search (same scoring everywhere).  Real use is one
    VariantAlignment per VCF row (docs/examples.rst:247-266), a few hundred to a few thousand alignments per locus -- far too few
    to fill a GPU, whose fixed cost per call (~1.5 ms) is a thousand alignments' worth.  Here every recursion level of every
    locus' search shares ONE batch: the per-call cost is paid once per level, not once per locus.  Returns one result per request,
    exactly what grid_search returns for it."""
    searches = [_RetargetSearch(t, pl, w, mq, wi, cut, ms, mm, grid, unspl, exact) for (t, pl, w, mq, wi, cut, ms, mm, grid, unspl, exact) in requests]
    run_retarget_searches(searches, device)
    return [_finish_grid_search(s_, req[8]) for s_, req in zip(searches, requests)]


def is_perfect_match(aligner, contig_seq, read_seq):
    """varaln.pyx:1228-1234: the ungapped alignment (gap_open = gap_ext = len(read)) covers identical stretches
    (end coordinates exclusive, as the reference slices them)"""
    aligner.setRead(read_seq)
    a = aligner.align(gap_open=len(read_seq), gap_extension=len(read_seq))
    return contig_seq[a.reference_start:a.reference_end] == read_seq[a.read_start:a.read_end]
