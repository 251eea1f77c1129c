"""Drop-in for ``indelpost.sswpy`` (the Cython binding of ssw.c) running on MI355X.

Same surface as the reference module (indelpost/sswpy.pyx): the stateful ``SSW`` class
(``setRead`` / ``setReference`` / ``align``), the ``Alignment`` named tuple, ``force_align`` and
``format_force_align`` -- plus ``SSW.align_batch`` for many reads against the aligner's reference,
which is what the GPU is for.  Alignments run in libindelpost_hip.so; there is no CPU path.
"""
from typing import NamedTuple, Optional, Union

import numpy as np

from .batch import MultiStreamAligner, JobTable, cigar_to_string, dna_score_matrix, encode_dna

Alignment = NamedTuple("Alignment", [          # sswpy.pyx:85-94
    ("CIGAR", Optional[str]),
    ("optimal_score", int),
    ("sub_optimal_score", int),
    ("reference_start", int),
    ("reference_end", int),
    ("read_start", int),
    ("read_end", int),
])

STR_T = Union[str, bytes]

_shared_gpu = {}


def _gpu(device=0):
    """One multi-stream aligner per device, shared by every SSW object (created on first use).
    Small batches (a single SSW.align) use one stream; large ones are cut into 4 concurrent slices."""
    g = _shared_gpu.get(device)
    if g is None:
        g = MultiStreamAligner(device, streams=4)
        _shared_gpu[device] = g
    return g


def _as_bytes(s):
    if isinstance(s, bytes):
        return s
    if isinstance(s, str):
        return s.encode("utf8")
    raise TypeError("expected str or bytes")


def _alignment_from(res, i):
    r = res.records[i]
    if int(r["mode"]) == 2:      # reference: ssw_align returned NULL (sswpy.pyx:220-223)
        raise ValueError("Problem Running alignment, see stdout")
    return Alignment(res.cigar_string(i), int(r["score1"]), int(r["score2"]), int(r["ref_begin1"]),
                     int(r["ref_end1"]), int(r["read_begin1"]), int(r["read_end1"]))


def alignments_from(res):
    """All Alignment tuples of a batch at once (same content as _alignment_from(res, i) for every i): the
    record columns are converted with one numpy call each and the CIGAR strings by the library."""
    rec = res.records
    if len(rec) and bool((rec["mode"] == 2).any()):     # reference: ssw_align returned NULL (sswpy.pyx:220-223)
        raise ValueError("Problem Running alignment, see stdout")
    cols = [res.cigar_strings()] + [rec[f].tolist() for f in ("score1", "score2", "ref_begin1", "ref_end1", "read_begin1", "read_end1")]
    return list(map(Alignment._make, zip(*cols)))


class SSW:
    """Mirror of ``cdef class SSW`` (sswpy.pyx:99-337)."""

    def __init__(self, match_score: int = 2, mismatch_penalty: int = 2, device: int = 0):
        self.score_matrix = dna_score_matrix(match_score, mismatch_penalty)   # sswpy.pyx:127-130
        self.read = None
        self.reference = None
        self._read_arr = None
        self._ref_arr = None
        self._device = device

    # -- sswpy.pyx:149-178
    def setRead(self, read: STR_T):
        self._read_arr = encode_dna(_as_bytes(read))
        self.read = read

    # -- sswpy.pyx:180-197
    def setReference(self, reference: STR_T):
        self._ref_arr = encode_dna(_as_bytes(reference))
        self.reference = reference

    def _window(self, start_idx, end_idx):
        """Index checks and slicing of SSW.align (sswpy.pyx:263-280)."""
        ref_length = 0 if self._ref_arr is None else len(self._ref_arr)
        if start_idx < 0 or end_idx < 0:
            raise ValueError("negative indexing not supported")
        if end_idx > ref_length or start_idx > ref_length:
            raise ValueError("start_idx: {} or end_idx: {} can't be greater than ref_length: {}".format(
                start_idx, end_idx, ref_length))
        end_final = ref_length if end_idx == 0 else end_idx
        if self.reference is None:
            raise ValueError("call setReference first")
        search_length = end_final - start_idx
        if search_length < 0:
            search_length = 0    # reference passes a negative refLen to C (no columns are visited)
        return self._ref_arr[start_idx:start_idx + search_length]

    def _run(self, reads, gap_open, gap_extension, window):
        g = _gpu(self._device)
        g.set_scoring(matrix=self.score_matrix, flag=1, filters=0, filterd=0, score_size=2)  # sswpy.pyx:172-177, 219
        n = len(reads)
        jobs = JobTable.from_sequences(reads, [window], np.zeros(n, np.int32), gap_open, gap_extension,
                                       encoded=True)
        return g.align(jobs)

    # -- sswpy.pyx:227-304
    def align(self, gap_open: int = 3, gap_extension: int = 1, start_idx: int = 0, end_idx: int = 0) -> Alignment:
        window = self._window(start_idx, end_idx)
        if self._read_arr is None:
            raise ValueError("Must set profile first")        # sswpy.pyx:220-221
        res = self._run([self._read_arr], int(gap_open), int(gap_extension), window)
        return _alignment_from(res, 0)

    def align_batch(self, reads, gap_open=3, gap_extension=1, start_idx: int = 0, end_idx: int = 0):
        """Align many reads against this aligner's reference in one GPU batch.

        ``gap_open`` / ``gap_extension`` may be scalars or one value per read.  Equivalent to
        ``[self.setRead(r) or self.align(go, ge, start_idx, end_idx) for r in reads]``.
        """
        window = self._window(start_idx, end_idx)
        enc = [encode_dna(_as_bytes(r)) for r in reads]
        if not enc:
            return []
        res = self._run(enc, gap_open, gap_extension, window)
        return alignments_from(res)


def force_align(read: STR_T, reference: STR_T, force_overhang: bool = False, aligner: SSW = None) -> Alignment:
    """sswpy.pyx:339-367: forbid gaps by raising gap_open to len(read)."""
    a = SSW() if aligner is None else aligner
    a.setRead(read)
    a.setReference(reference)
    res = a.align(gap_open=len(read))
    if res.optimal_score < 4:
        raise ValueError("No solution found")
    if force_overhang:
        if res.reference_start != 0 or res.reference_end != len(reference) - 1:
            raise ValueError("Read does not align to one overhang")
    return res


def format_force_align(read: STR_T, reference: STR_T, alignment: Alignment, do_print: bool = False):
    """sswpy.pyx:370-395."""
    def _s(x):
        return x.decode("utf8") if isinstance(x, bytes) else x
    start_ref, start_read = alignment.reference_start, alignment.read_start
    buffer_ref = buffer_read = ""
    if start_ref < start_read:
        buffer_ref = " " * (start_read - start_ref)
    else:
        buffer_read = " " * (start_ref - start_read)
    ref_out = buffer_ref + _s(reference)
    read_out = buffer_read + _s(read)
    if do_print:
        print(ref_out)
        print(read_out)
    return ref_out, read_out
