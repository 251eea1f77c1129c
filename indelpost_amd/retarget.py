"""The multi-window / multi-penalty realignment of indelPost as ONE GPU batch.

In the reference, ``retarget`` aligns every non-reference read against ITS OWN local window, one SSW call per read
(pileup.pyx:639-648), and ``grid_search`` repeats the whole thing for each of up to seven (gap_open, gap_ext) pairs
(varaln.pyx:1122-1225); the overhang filter adds two alignments per read (pileup.pyx:527-548) and
``is_perfect_match`` one more with gap_open = gap_ext = len(read) (varaln.pyx:1228-1234).  Through the C-level
drop-in each of those calls would be a GPU round trip of its own (~1.5 ms against 78 us on a CPU core); here they are
rows of one job table (read k x penalty pair g -> job k * G + g, per-read windows) and one call into
libindelpost_hip.so.

What is mirrored (citations into /root/reference/indelpost/):
  generate_grid                      varaln.pyx:1122-1145
  UnsplicedLocalReference            local_reference.pyx:4-33
  get_local_reference (unspliced)    utilities.pyx:505-586 (the splice-pattern branch needs the pileup's splice patterns)
  the alignment loop of retarget     pileup.pyx:639-648   -> retarget_jobs / grid_align
  candidate extraction of retarget   pileup.pyx:650-711   -> indel_candidates (plain tuples instead of Variant objects)
  is_non_spurious_overhang, SSW part pileup.pyx:527-548   -> overhang_jobs / overhang_alignment_verdicts
  is_perfect_match                   varaln.pyx:1228-1234 -> perfect_match_batch
The composed functions with the reference's signatures -- retarget (candidate matching with Variant objects, the window / 3
recursion), update_read_info, filter_spurious_overhangs -- are in pileup.py, grid_search in varaln.py; they are built on the
job tables and helpers here.
"""
import numpy as np

from .batch import JobTable, dna_score_matrix, encode_dna
from .cigar import findall_indels, gap_count, make_insertion_first
from .sswpy import _gpu, alignments_from


def generate_grid(auto_adjust_extension_penalty, gap_open_penalty, gap_extension_penalty, indel_len):
    """The (gap_open, gap_ext) pairs grid_search tries, in its order (varaln.pyx:1122-1145).  `indel_len` is
    len(target.indel_seq): long indels try (3, 0) before (3, 1)."""
    if not auto_adjust_extension_penalty:
        return [(gap_open_penalty, gap_extension_penalty)]
    std = [(3, 1), (3, 0)] if indel_len < 20 else [(3, 0), (3, 1)]
    std += [(5, 1), (5, 0), (4, 1), (4, 0)]
    if (gap_open_penalty, gap_extension_penalty) != (3, 1):
        return [(gap_open_penalty, gap_extension_penalty)] + std
    return std


class UnsplicedLocalReference:
    """The cached stretch of genome around a target (pos +- 10 windows) that per-read windows are cut from
    (local_reference.pyx:4-33).  `reference` is a pysam.FastaFile duck type: fetch(chrom, start, end)."""

    def __init__(self, chrom, pos, ref_len, window, reference):
        self.chrom, self.pos, self.ref_len, self.window = chrom, pos, ref_len, window
        self.local_ref_start = max(0, pos - window * 10)
        self.unspliced_local_reference = reference.fetch(chrom, self.local_ref_start, min(pos + window * 10, ref_len))

    def fetch_ref_seq(self, target_pos, window):
        self.left_len = target_pos - max(0, target_pos - window * 3)
        return self.get_ref_seq(max(0, target_pos - window * 3), min(target_pos + window * 3, self.ref_len))

    def get_ref_seq(self, start, end):
        k = start - self.local_ref_start
        return self.unspliced_local_reference[k:k + (end - start)]


def most_common(lst):
    """the most frequent element, ties going to the smallest (utilities.pyx:19-22)"""
    return max(sorted(set(lst)), key=lst.count)


def get_local_reference(target, pileup, window, unspl_loc_ref, unspliced=False, splice_pattern_only=False):
    """(window string, left_len) for the reads in `pileup` around `target` (utilities.pyx:505-586).

    Unspliced reads: target.pos +- 3 windows cut from the cached stretch.  Reads carrying splice patterns
    (read["splice_pattern"] = (left "a-b:c-d", right "e-f"), 1-based exon/intron boundaries): the most common left and
    right patterns are stitched into an exon-only window reaching 2 windows beyond the outermost boundaries.  `target`
    needs .chrom, .pos and .reference (fetch, get_reference_length): the pysam duck types of SURVEY.md 8b."""
    chrom, pos, reference = target.chrom, target.pos, target.reference
    patterns = None if unspliced else [r["splice_pattern"] for r in pileup if r["splice_pattern"] != ("", "")]
    spans = []
    if patterns:
        ref_len = reference.get_reference_length(chrom)
        bounds = []
        for side in (0, 1):
            side_patterns = [p[side] for p in patterns if p[side]]
            if side_patterns:
                for span in most_common(side_patterns).split(":"):
                    bounds += [int(x) for x in span.split("-")]
        last = len(bounds) - 1
        local, left_len, placed = "", 0, False
        rt_end = 0
        for i, x in enumerate(bounds):
            if i == 0:
                lt_end = max(0, x - window * 2)
                local += reference.fetch(chrom, lt_end, x - 1)
                rt_end = x - 1
                spans.append((x + 1, rt_end) if x + 1 < rt_end else (lt_end, rt_end))
            elif i % 2 == 1 and i != last:
                rt_end = bounds[i + 1] - 1
                local += reference.fetch(chrom, x, rt_end)
                spans.append((x + 1, rt_end))
            elif i % 2 == 0:
                pass
            elif i == last:
                rt_end = min(x + window * 2, ref_len)
                local += reference.fetch(chrom, x, rt_end)
                spans.append((x + 1, rt_end))
            if pos <= rt_end and not placed:
                left_len = len(local) - (rt_end - pos)
                placed = True
    else:
        local = unspl_loc_ref.fetch_ref_seq(pos, window)
        left_len = pos - max(0, pos - window * 3)
    if splice_pattern_only:
        return tuple(spans)
    return local, left_len


def retarget_jobs(read_seqs, ref_seqs, grid):
    """Job table of retarget x grid_search: job k * len(grid) + g = read k against ITS window ref_seqs[k] under
    grid[g] (pileup.pyx:639-648 inside the loop of varaln.pyx:1163-1178).  Reads sharing a window string share the
    window in the table."""
    n, G = len(read_seqs), len(grid)
    raw = [s.encode("utf8") if isinstance(s, str) else bytes(s) for s in read_seqs]
    lens = np.fromiter((len(b) for b in raw), np.int64, n)
    reads = encode_dna(b"".join(b for b in raw for _ in range(G)))
    read_off = np.zeros(n * G + 1, np.int64)
    np.cumsum(np.repeat(lens, G), out=read_off[1:])
    uniq, wid = {}, np.empty(n, np.int32)
    for k, w in enumerate(ref_seqs):
        wid[k] = uniq.setdefault(w, len(uniq))
    wins = [encode_dna(w) for w in uniq]
    ref_off = np.zeros(len(wins) + 1, np.int64)
    if wins:
        np.cumsum([len(w) for w in wins], out=ref_off[1:])
    refs = np.concatenate(wins) if wins and ref_off[-1] else np.zeros(0, np.int8)
    go = np.tile(np.array([g[0] for g in grid], np.int64), n)
    ge = np.tile(np.array([g[1] for g in grid], np.int64), n)
    return JobTable(reads, read_off, refs, ref_off, np.repeat(wid, G), go, ge)


def grid_align(read_seqs, ref_seqs, grid, match_score, mismatch_penalty, device=0):
    """All alignments of a grid search in one GPU call.  Returns alns[g][k] = the Alignment tuple that
    ``align(make_aligner(ref_seqs[k], match, mismatch), read_seqs[k], *grid[g])`` returns in the reference."""
    if not read_seqs:
        return [[] for _ in grid]
    g = _gpu(device)
    g.set_scoring(matrix=dna_score_matrix(match_score, mismatch_penalty), flag=1, score_size=2)
    alns = alignments_from(g.align(retarget_jobs(read_seqs, ref_seqs, grid)))
    G = len(grid)
    return [alns[k::G] for k in range(G)]


def indel_candidates(aln, read_seq, ref_seq, ref_start, target_type, read_start, read_end, window, is_target=None):
    """What retarget derives from ONE read's alignment (pileup.pyx:650-711), without Variant objects.

    Returns (candidates, is_complex): candidates = [(pos, ref, alt)] of the indels of the target's type found in the
    alignment (complex ins+del pairs at one position reported as one ref>alt), after the reference's filters: a usable
    CIGAR with 1..5 gaps, more than 70 % of the read (or of the 6-window span) aligned, and the read-end rule -- an indel
    within max(len(read)/30, 3) bases of a read end only counts when it IS the target (`is_target(pos, ref, alt)`, the
    caller's Variant equality; default: never) or when the read carries a complex position elsewhere."""
    if not aln.CIGAR:
        return [], False
    aligned_frac = (aln.read_end - aln.read_start) / min(len(read_seq), window * 6)
    if not (0 < gap_count(aln.CIGAR) < 6 and aligned_frac > 0.7):
        return [], False
    indels = findall_indels(aln, ref_start + aln.reference_start, ref_seq, read_seq)
    positions = [d["pos"] for d in indels]
    complex_positions = {p for p in positions if positions.count(p) == 2}
    out = []
    end_thresh = max(len(read_seq) / 30, 3)
    for d in indels:
        if d["indel_type"] != target_type:
            continue
        if d["pos"] in complex_positions:
            dl = [x for x in indels if x["pos"] == d["pos"] and x["indel_type"] == "D"][0]
            ins = [x for x in indels if x["pos"] == d["pos"] and x["indel_type"] == "I"][0]
            ref, alt = dl["lt_ref"][-1] + dl["del_seq"], ins["lt_ref"][-1] + ins["indel_seq"]
        elif target_type == "I":
            ref = d["lt_ref"][-1]
            alt = ref + d["indel_seq"]
        else:
            alt = d["lt_ref"][-1]
            ref = alt + d["del_seq"]
        pos = d["pos"]
        if pos - read_start <= end_thresh or read_end - pos <= end_thresh:
            same = bool(is_target(pos, ref, alt)) if is_target else False
            if not (same or (complex_positions and pos not in complex_positions)):
                continue
        out.append((pos, ref, alt))
    return out, bool(complex_positions)


def overhang_jobs(read_seqs, genome_ref, junction_ref, gap_open_penalty, gap_extension_penalty):
    """Two jobs per overhanging read: vs the unspliced genome window and vs the exon-exon junction window
    (pileup.pyx:540-545; windows from overhang_aligners, pileup.pyx:477-492): job 2k genome, 2k+1 junction."""
    n = len(read_seqs)
    raw = [s.encode("utf8") if isinstance(s, str) else bytes(s) for s in read_seqs]
    lens = np.fromiter((len(b) for b in raw), np.int64, n)
    reads = encode_dna(b"".join(b for b in raw for _ in (0, 1)))
    read_off = np.zeros(2 * n + 1, np.int64)
    np.cumsum(np.repeat(lens, 2), out=read_off[1:])
    wins = [encode_dna(genome_ref), encode_dna(junction_ref)]
    ref_off = np.array([0, len(wins[0]), len(wins[0]) + len(wins[1])], np.int64)
    return JobTable(reads, read_off, np.concatenate(wins), ref_off, np.tile(np.array([0, 1], np.int32), n),
                    gap_open_penalty, gap_extension_penalty)


def overhang_alignment_verdicts(read_seqs, genome_ref, junction_ref, match_score, mismatch_penalty, gap_open_penalty,
                                gap_extension_penalty, device=0):
    """The alignment-only part of is_non_spurious_overhang (pileup.pyx:527-563) for a whole set of overhangs at once.
    Per read: False = spurious by its alignments alone; None = the alignments do not rule it out and the caller goes on
    with the reference's read-level checks (indels inside the intron, is_worth_realn; pileup.pyx:565-574).  Also
    returns the (genome, junction) Alignment pairs."""
    if not read_seqs:
        return [], []
    g = _gpu(device)
    g.set_scoring(matrix=dna_score_matrix(match_score, mismatch_penalty), flag=1, score_size=2)
    alns = alignments_from(g.align(overhang_jobs(read_seqs, genome_ref, junction_ref, gap_open_penalty, gap_extension_penalty)))
    pairs = list(zip(alns[0::2], alns[1::2]))
    verdicts = []
    for seq, (ga, ja) in zip(read_seqs, pairs):
        gs, js = ga.optimal_score, ja.optimal_score
        v = None
        if gs <= js:
            v = False
        else:
            gaps = gap_count(make_insertion_first(ga.CIGAR))
            if gaps > 3:
                v = False
            elif 1 < gaps <= 3:
                if gs / js < 1.2 or gs < match_score * 50:       # (js == 0 raises ZeroDivisionError, as in the reference)
                    v = False
            elif gaps == 0 and (ga.read_end - ga.read_start + 1) / len(seq) > 0.98:
                v = False
        verdicts.append(v)
    return verdicts, pairs


def perfect_match_batch(read_seqs, contig_seq, match_score, mismatch_penalty, device=0):
    """is_perfect_match (varaln.pyx:1228-1234) for many reads against one contig: gap_open = gap_ext = len(read)
    (narrowed to uint8 at the C boundary like the reference's arguments, ssw.h:129-130); the aligned stretches -- end
    coordinates EXCLUSIVE, as the reference slices them -- must be identical strings."""
    if not read_seqs:
        return []
    lens = [len(s) for s in read_seqs]
    g = _gpu(device)
    g.set_scoring(matrix=dna_score_matrix(match_score, mismatch_penalty), flag=1, score_size=2)
    jobs = JobTable.from_sequences(list(read_seqs), [contig_seq], np.zeros(len(read_seqs), np.int32), lens, lens)
    out = []
    for seq, a in zip(read_seqs, alignments_from(g.align(jobs))):
        out.append(contig_seq[a.reference_start:a.reference_end] == seq[a.read_start:a.read_end])
    return out


def align_many(read_seqs, ref_seqs, gap_opens, gap_exts, match_score, mismatch_penalty, device=0):
    """One GPU batch of arbitrary (read, window, gap_open, gap_ext) jobs -> Alignment tuples in job order.  Reads and windows
    that repeat (the same read under several penalty pairs, reads sharing a window) are stored once per distinct window."""
    n = len(read_seqs)
    if n == 0:
        return []
    raw = [s.encode("utf8") if isinstance(s, str) else bytes(s) for s in read_seqs]
    lens = np.fromiter((len(b) for b in raw), np.int64, n)
    reads = encode_dna(b"".join(raw))
    read_off = np.zeros(n + 1, np.int64)
    np.cumsum(lens, out=read_off[1:])
    uniq, wid = {}, np.empty(n, np.int32)
    for k, w in enumerate(ref_seqs):
        wid[k] = uniq.setdefault(w, len(uniq))
    wins = [encode_dna(w) for w in uniq]
    ref_off = np.zeros(len(wins) + 1, np.int64)
    np.cumsum([len(w) for w in wins], out=ref_off[1:])
    refs = np.concatenate(wins) if ref_off[-1] else np.zeros(0, np.int8)
    g = _gpu(device)
    g.set_scoring(matrix=dna_score_matrix(match_score, mismatch_penalty), flag=1, score_size=2)
    return alignments_from(g.align(JobTable(reads, read_off, refs, ref_off, wid, np.asarray(gap_opens, np.int64), np.asarray(gap_exts, np.int64))))
