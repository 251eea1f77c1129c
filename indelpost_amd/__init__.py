"""indelpost_amd -- MI355X-native drop-in for indelPost's Smith-Waterman realignment path.

Scope: the striped SSW aligner and its batched driver.  The names the
reference exports from ``indelpost/__init__.py`` that lie outside that path (``Variant``,
``VariantAlignment``, ...) are present as shells that say so when used.
"""
from .sswpy import SSW, Alignment, force_align, format_force_align          # noqa: F401
from .localn import (make_aligner, align, align_pileup, realign_pileup_jobs, classify_realigned, find_targets_by_ssw,  # noqa: F401
                     is_covering_target, is_compatible_repeats)
from .pileup import (split_cigar, trim_ref_flank, update_cigar, update_read_positions, update_read_info_realn,  # noqa: F401
                     update_reads_batch)
from .cigar import findall_indels, make_insertion_first, merge_consecutive_gaps, to_minimal_repeat_unit  # noqa: F401
from .retarget import (generate_grid, retarget_jobs, grid_align, indel_candidates, get_local_reference,  # noqa: F401
                       UnsplicedLocalReference, overhang_jobs, overhang_alignment_verdicts, perfect_match_batch)
from .batch import (GpuAligner, MultiStreamAligner, JobTable, BatchResult, IpxError, align_sharded, device_count,  # noqa: F401
                    dna_score_matrix, encode_dna, cigar_to_string)

__version__ = "0.1.0"


from .variant import Variant, NullVariant                                   # noqa: F401,E402
from . import bamio                                                         # noqa: F401,E402
from .pileup import (make_pileup, dictize_read, fetch_reads, parse_spliced_read, retarget_many, update_read_info,   # noqa: F401,E402
                     check_overhangs, filter_spurious_overhangs)
from .localn import (find_by_smith_waterman_realn, find_by_smith_waterman_realn_many, is_target_by_ssw, is_worth_realn,   # noqa: F401,E402
                     findall_mismatches, parse_read_by_mut_aln)
from .varaln import grid_search, grid_search_many, is_perfect_match                                                       # noqa: F401,E402
# (retarget itself lives where the reference has it: indelpost_amd.pileup.retarget -- the name indelpost_amd.retarget is the module)

_WHY = ("%s (%s) is outside the hot path this package replaces (SURVEY.md section 8: contig / consensus construction, "
        "phasing, the orchestration state machine); use the reference implementation for it and plug this package in at "
        "make_aligner()/align() or, better, at the batched drivers align_pileup / find_targets_by_ssw / grid_align.")


class VariantAlignment:
    """API shell with the reference's constructor signature (indelpost/varaln.pyx:102-120)."""

    def __init__(self, target, bam, window=50, exclude_duplicates=True, retarget_search_window=30,
                 retarget_similarity_cutoff=0.7, exact_match_for_shiftable=True, mapping_quality_threshold=1,
                 downsample_threshold=1000, base_quality_threshold=20, match_score=3, mismatch_penalty=2,
                 gap_open_penalty=3, gap_extension_penalty=1, auto_adjust_extension_penalty=True, no_realignment=False):
        raise NotImplementedError(_WHY % ("VariantAlignment", "indelpost/varaln.pyx:41"))


class Contig:
    """API shell with the reference's constructor signature (indelpost/contig.pyx:22)."""

    def __init__(self, target, pileup, unspl_loc_ref, basequalthresh, mapqthresh, low_consensus_thresh=0.7, donwsample_lim=100):
        raise NotImplementedError(_WHY % ("Contig", "indelpost/contig.pyx:19"))


class FailedContig:
    """API shell with the reference's constructor signature (indelpost/contig.pyx:354)."""

    def __init__(self):
        raise NotImplementedError(_WHY % ("FailedContig", "indelpost/contig.pyx:339"))
