"""indelpost_amd -- MI355X-native drop-in for indelPost's Smith-Waterman realignment path.

Scope: the striped SSW aligner and its batched driver.  The names the
reference exports from ``indelpost/__init__.py`` that lie outside that path (``Variant``,
``VariantAlignment``, ...) are present as shells that say so when used.
"""
from .sswpy import SSW, Alignment, force_align, format_force_align          # noqa: F401
from .localn import (make_aligner, align, align_pileup, realign_pileup_jobs, classify_realigned, find_targets_by_ssw,  # noqa: F401
                     is_covering_target, is_compatible_repeats)
from .cigar import findall_indels, make_insertion_first, merge_consecutive_gaps, to_minimal_repeat_unit  # noqa: F401
from .retarget import (generate_grid, retarget_jobs, grid_align, indel_candidates, get_local_reference,  # noqa: F401
                       UnsplicedLocalReference, overhang_jobs, overhang_alignment_verdicts, perfect_match_batch)
from .batch import (GpuAligner, MultiStreamAligner, JobTable, BatchResult, IpxError, align_sharded, device_count,  # noqa: F401
                    dna_score_matrix, encode_dna, cigar_to_string)

__version__ = "0.1.0"


def _out_of_scope(name, where):
    class _Shell:
        __doc__ = ("%s (%s) is outside the hot path this package replaces; use the reference "
                   "implementation for it and plug this package in at make_aligner()/align()." % (name, where))

        def __init__(self, *a, **k):
            raise NotImplementedError(self.__doc__)
    _Shell.__name__ = name
    return _Shell


Variant = _out_of_scope("Variant", "indelpost/variant.pyx:62")
NullVariant = _out_of_scope("NullVariant", "indelpost/variant.pyx:9")
VariantAlignment = _out_of_scope("VariantAlignment", "indelpost/varaln.pyx:41")
Contig = _out_of_scope("Contig", "indelpost/contig.pyx:19")
FailedContig = _out_of_scope("FailedContig", "indelpost/contig.pyx:338")
