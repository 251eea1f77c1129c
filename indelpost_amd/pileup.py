"""What indelPost does to a read once its realignment carries the target indel: the second consumer of the alignments.

Mirrors (citations into /root/reference/indelpost/):
  split_cigar                             utilities.pyx:330-357
  trim_ref_flank, update_cigar,
  numeric_span, update_read_positions     pileup.pyx:913-1024
  update_read_info, realignment branch    pileup.pyx:847-911  -> update_read_info_realn / update_reads_batch
The reference realigns one read per call (`align(aligner, read["read_seq"], ...)`, pileup.pyx:849) for every read of
the best grid_search response (varaln.pyx:1200-1216); here those alignments are one GPU batch (retarget.grid_align with
the chosen penalty pair) and this module turns each into the read-dict surgery the reference performs: flanks and
qualities left and right of the indel, window flanks trimmed to the aligned part, the read's BAM CIGAR rewritten from the
realignment (soft clips, splice junctions re-inserted as N), positions and clip offsets updated.
update_read_info (further down) is the whole function with the reference's signature, gapped-alignment branch included.
Parity: pinned by vectors from the reference's own function bodies -- split_cigar, trim_ref_flank, update_cigar and
update_read_positions by oracle/gen_decoder_golden.py, update_read_info as a whole (through grid_search) by
oracle/gen_driver_golden.py.
"""
import numpy as np

from .cigar import cigar_ptrn, findall_indels, make_insertion_first

_NO_REF_MOVE = ("I", "H", "P")


def split_cigar(cigarstring, target_pos, start):
    """CIGAR tokens left of / right of genome position target_pos for a read starting at `start` (1-based): the token
    that reaches target_pos is cut there (utilities.pyx:330-357).  Returns None when the CIGAR ends before target_pos,
    as the reference falls off its loop."""
    toks = cigar_ptrn.findall(cigarstring)
    pos = start - 1
    for k, tok in enumerate(toks):
        op, n = tok[-1], int(tok[:-1])
        if op not in _NO_REF_MOVE:
            pos += n
        if target_pos <= pos:
            over = pos - target_pos
            rest = toks[k + 1:]
            if over:
                rest = [str(over) + op] + rest
            return toks[:k] + [str(n - over) + op], rest
    return None


def trim_ref_flank(ref_flank, flank_cigar, left):
    """the part of a window flank the realigned flank CIGAR spans on the reference (pileup.pyx:913-921)"""
    span = sum(int(c[:-1]) for c in flank_cigar if c[-1] != "I")
    return ref_flank[-span:] if left else ref_flank[:span]


def numeric_span(spl_span):
    return [int(x) for x in spl_span.split("-")]


def update_cigar(orig_cigar_string, realn_cigar, start_pos, splice_prtn, clipped_bases, left):
    """One flank of the read's new BAM CIGAR from the realignment's flank CIGAR (pileup.pyx:924-1024): unaligned read
    ends become soft clips; the realignment was made against an exon-only window, so the introns of the read's splice
    pattern (splice_prtn = (left "a-b:c-d", right "e-f")) are put back as N where an M run (or an insertion sitting
    exactly on a junction) crosses them.  Left flank: starts at start_pos (+ the clip).  Right flank: realn_cigar[0] is
    the indel itself; the walk starts behind it.  (orig_cigar_string is not used by the reference either.)"""
    pattern = splice_prtn[0] if left else splice_prtn[1]
    spans = [numeric_span(x) for x in pattern.split(":")] if pattern else []
    clip = [str(len(clipped_bases)) + "S"] if len(clipped_bases) else []
    out = []
    if left:
        out += clip
        pos = start_pos + len(clipped_bases)
        body = list(realn_cigar)
    else:
        head = realn_cigar[0]
        pos = start_pos + 1 if head[-1] == "I" else start_pos + int(head[:-1]) + 1
        body = list(realn_cigar[1:])
    for tok in body:
        op, n = tok[-1], int(tok[:-1])
        if op == "M":
            if not spans:
                out.append(str(n) + "M")
                pos += n
                continue
            pending = list(spans)
            last = len(pending) - 1
            for i, (a, b) in enumerate(pending):
                intron = b - a + 1
                if a <= pos + n:
                    if i != last:
                        m = a - pos
                        out += ([str(m) + "M"] if m else []) + [str(intron) + "N"]
                        pos += m + intron
                        n -= m
                    else:
                        m1 = a - pos
                        m2 = n - m1
                        if m2:
                            out += ([str(m1) + "M"] if m1 else []) + [str(intron) + "N", str(m2) + "M"]
                        else:
                            out += [str(n) + "M", str(intron) + "N"]
                        pos += intron + n
                    spans = spans[1:]
                else:
                    out.append(str(n) + "M")
                    pos += n - 1                     # (the reference's own "hotfix -1", pileup.pyx:986-987)
                    break
        elif op == "I":
            if spans and spans[0][0] == pos:
                out += [str(n) + "I", str(spans[0][1] - spans[0][0] + 1) + "N"]
                pos += spans[0][1] - spans[0][0] + 1
                spans = spans[1:]
            else:
                out.append(str(n) + "I")
                pos += 1
        elif op == "D":
            out.append(str(n) + "D")
            pos += n
    return out if left else [head] + out + clip


def update_read_positions(read, target_pos):
    """read_start / read_end / clip offsets / aln_start / aln_end from the new flank CIGARs (pileup.pyx:1032-1048)"""
    left = sum(int(c[:-1]) for c in read["lt_cigar"] if c[-1] != "I")
    right = sum(int(c[:-1]) for c in read["rt_cigar"] if c[-1] != "I")
    read["read_start"] = target_pos - left + 1
    read["read_end"] = target_pos + right
    first, last = read["lt_cigar"][0], read["rt_cigar"][-1]
    read["start_offset"] = int(first[:-1]) if "S" in first else 0
    read["end_offset"] = int(last[:-1]) if "S" in last else 0
    read["aln_start"] = read["read_start"] + read["start_offset"]
    read["aln_end"] = read["read_end"] - read["end_offset"]


def update_read_info_realn(read, aln, ref_seq, ref_start, candidate_pos, candidate_indel_seq, candidate_is_ins, is_candidate):
    """The realignment branch of update_read_info (pileup.pyx:847-911) with the alignment already made.

    read: the reference's read dict (needs read_seq, read_qual, cigar_string, read_start, splice_pattern); aln: its
    Alignment against ref_seq (the read's local window, genome position ref_start at window index 0);
    is_candidate(pos, ref, alt): the caller's `Variant(...) == candidate`.  Updates and returns the dict exactly as the
    reference does: cigar_updated False when the candidate is not among the alignment's indels."""
    genome_aln_pos = ref_start + aln.reference_start
    indels = findall_indels(aln, genome_aln_pos, ref_seq, read["read_seq"], basequals=read["read_qual"])
    hit = None
    for d in indels:
        if not d.get("del_seq", False):
            ref = d["lt_ref"][-1]
            alt = ref + d["indel_seq"]
        else:
            alt = d["lt_ref"][-1]
            ref = alt + d["del_seq"]
        if is_candidate(d["pos"], ref, alt):
            hit = d
            break
    if hit is None:
        read["cigar_updated"] = False
        return read
    read["lt_flank"] = hit["lt_flank"]
    read["indel_seq"] = candidate_indel_seq if candidate_is_ins else ""
    read["rt_flank"] = hit["rt_flank"]
    read["lt_qual"] = hit["lt_qual"]
    read["rt_qual"] = hit["rt_qual"]
    lt_c, rt_c = split_cigar(make_insertion_first(aln.CIGAR), hit["pos"], genome_aln_pos)
    read["lt_ref"] = trim_ref_flank(hit["lt_ref"], lt_c, left=True)
    read["rt_ref"] = trim_ref_flank(hit["rt_ref"], rt_c, left=False)
    read["lt_cigar"] = update_cigar(read["cigar_string"], lt_c, read["read_start"], read["splice_pattern"], hit["lt_clipped"], left=True)
    read["rt_cigar"] = update_cigar(read["cigar_string"], rt_c, candidate_pos, read["splice_pattern"], hit["rt_clipped"], left=False)
    read["cigar_list"] = read["lt_cigar"] + read["rt_cigar"]
    read["cigar_string"] = "".join(read["cigar_list"])
    read["cigar_updated"] = True
    update_read_positions(read, hit["pos"])
    read["is_target"] = True
    return read


def update_reads_batch(reads, ref_seqs, ref_starts, candidate_pos, candidate_indel_seq, candidate_is_ins, is_candidate,
                       match_score, mismatch_penalty, gap_open_penalty, gap_extension_penalty, device=0):
    """update_read_info(..., is_gapped_aln=False) for every read of a grid_search response (varaln.pyx:1200-1216):
    all realignments in one GPU call, then the per-read surgery."""
    from .retarget import grid_align
    alns = grid_align([r["read_seq"] for r in reads], ref_seqs, [(gap_open_penalty, gap_extension_penalty)], match_score,
                      mismatch_penalty, device)[0]
    return [update_read_info_realn(r, a, w, s, candidate_pos, candidate_indel_seq, candidate_is_ins, is_candidate)
            for r, a, w, s in zip(reads, alns, ref_seqs, ref_starts)]


# =====================================================================================================================
# The pileup front-end: BAM reads around a target -> the read dicts everything above consumes.
# Mirrors pileup.pyx:51-434 (make_pileup, fetch_reads, dictize_read, get_ref_seq, leftalign_indel_read, is_end_dirty,
# leftalign_cigar, parse_spliced_read, is_within_intron) and the helpers of utilities.pyx they call (:187-330, :429-503).
# `bam` / `reference` are pysam duck types (indelpost_amd.bamio provides pysam-free ones); `target` is a Variant.
# Parity: pinned.  The pure helpers by vectors from the reference's function bodies (oracle/gen_variant_golden.py, "helpers");
# make_pileup / fetch_reads / dictize_read / get_ref_seq as a whole by oracle/gen_driver_golden.py: the reference's own function
# text run against duck-typed BAM segments, every read dict compared field by field (tests/test_drivers.py).
# =====================================================================================================================
import random

_READ_ONLY = ("I", "S")
_ALIGNED = ("M", "=", "X")


def to_flat_list(lst_of_lst):
    return [x for lst in lst_of_lst for x in lst]


def count_lowqual_non_ref_bases(read_seq, ref_seq, quals, cigar_list, basequalthresh):
    """mismatching aligned bases and inserted / clipped bases below the quality threshold (utilities.pyx:187-218)"""
    i = j = cnt = 0
    for tok in cigar_list:
        op, n = tok[-1], int(tok[:-1])
        if op in _ALIGNED:
            for _ in range(n):
                if read_seq[i] != ref_seq[j] and quals[i] < basequalthresh:
                    cnt += 1
                i += 1
                j += 1
        elif op in _READ_ONLY:
            cnt += sum(1 for q in quals[i:i + n] if q < basequalthresh)
            i += n
        elif op == "D":
            j += n
    return cnt


def get_mapped_subreads(cigarstring, aln_start_pos, aln_end_pos):
    """genome intervals (inclusive) of the M/=/X runs (utilities.pyx:221-240)"""
    res, pos = [], aln_start_pos
    for tok in cigar_ptrn.findall(cigarstring):
        op, n = tok[-1], int(tok[:-1])
        if op in ("M", "X", "="):
            res.append((pos, pos + n - 1))
            pos += n
        elif op not in ("I", "S", "H", "P"):
            pos += n
    return res


def get_spliced_subreads(cigarstring, read_start_pos, read_end_pos):
    """the exon pieces of a read, split at N (soft clips included in the outer pieces) (utilities.pyx:243-278)"""
    if "N" not in cigarstring:
        return [(read_start_pos, read_end_pos)]
    pos, marks, prev = read_start_pos, [read_start_pos], "A"
    for tok in cigar_ptrn.findall(cigarstring):
        op, n = tok[-1], int(tok[:-1])
        if op == "N":
            marks.append(pos - 1)
        elif prev == "N":
            marks.append(pos)
        if op not in ("I", "H", "P"):
            pos += n
        prev = op
    if prev != "N":
        marks.append(read_end_pos)
    return [marks[i:i + 2] for i in range(0, len(marks), 2)]


def get_end_pos(read_start_pos, lt_flank, cigarstring):
    """genome position reached after consuming len(lt_flank) read bases (utilities.pyx:281-304)"""
    pos, left, toks, i = read_start_pos - 1, len(lt_flank), cigar_ptrn.findall(cigarstring), 0
    while left > 0:
        op, n = toks[i][-1], int(toks[i][:-1])
        if op in ("D", "N"):
            pos += n
        elif op == "I":
            left -= n
        elif op not in ("H", "P"):
            left -= n
            pos += n
        i += 1
    return pos + left


def locate_indels(cigarstring, aln_start_pos):
    """([(pos, len)] insertions, [(pos, len)] deletions): pos = genome position of the base left of the event
    (utilities.pyx:307-328)"""
    pos, ins, dels = aln_start_pos - 1, [], []
    for tok in cigar_ptrn.findall(cigarstring):
        op, n = tok[-1], int(tok[:-1])
        if op == "I":
            ins.append((pos, n))
        elif op == "D":
            dels.append((pos, n))
            pos += n
        elif op not in ("H", "P"):
            pos += n
    return ins, dels


def split(data, cigarstring, target_pos, string_pos, is_for_ref, reverse):
    """cut a read-side (or, is_for_ref, reference-side) sequence / quality array at genome position target_pos, walking
    the CIGAR from string_pos (utilities.pyx:429-503).  Slices use the reference's index arithmetic unchanged, negative
    indices included."""
    moves = []
    for tok in cigar_ptrn.findall(cigarstring):
        op, n = tok[-1], int(tok[:-1])
        if op == "N":
            moves.append((0, n))
        elif op == "I":
            moves.append((0 if is_for_ref else n, 0))
        elif op == "D":
            moves.append((n if is_for_ref else 0, n))
        elif op in ("H", "P"):
            moves.append((0, 0))
        else:
            moves.append((n, n))
    j = 0
    if reverse:
        string_pos += 1
        data = data[::-1]
        for d, g in moves[::-1]:
            if not target_pos < string_pos:
                break
            string_pos -= g
            j += d
        k = j + string_pos - (target_pos + 1)
        return data[k:][::-1], data[:k][::-1]
    string_pos -= 1
    for d, g in moves:
        if not string_pos < target_pos:
            break
        string_pos += g
        j += d
    k = j + target_pos - string_pos
    return data[:k], data[k:]


def get_ref_seq(chrom, aln_start, aln_end, cigar_string, cigar_list, reference, unspl_loc_ref):
    """the reference bases under the read's M and D runs (introns skipped) (pileup.pyx:269-298)"""
    pos = aln_start - 1
    if "N" not in cigar_string:
        return unspl_loc_ref.get_ref_seq(pos, aln_end)
    out = ""
    for tok in cigar_list:
        op, n = tok[-1], int(tok[:-1])
        if op in ("M", "D"):
            out += reference.fetch(chrom, pos, pos + n)
            pos += n
        elif op not in ("I", "S", "H", "P"):
            pos += n
    return out


def leftalign_indel_read(chrom, pos, indel_len, indel_type, cigar_string, read_start, aln_start, read_seq, ref_seq, read_qual,
                         reference):
    """one indel of a read's own CIGAR as (pos, lt_flank, indel_seq, rt_flank, lt_ref, rt_ref, lt_qual, rt_qual, Variant)
    (pileup.pyx:301-335)"""
    from .variant import Variant
    lt_flank, rt_flank = split(read_seq, cigar_string, pos, read_start, is_for_ref=False, reverse=False)
    lt_ref, rt_ref = split(ref_seq, cigar_string, pos, aln_start, is_for_ref=True, reverse=False)
    lt_qual, rt_qual = split(read_qual, cigar_string, pos, read_start, is_for_ref=False, reverse=False)
    pad = reference.fetch(chrom, pos - 1, pos) if ("N" in cigar_string or not lt_ref) else lt_ref[-1]
    if indel_type == "I":
        indel_seq = rt_flank[:indel_len]
        rt_flank, rt_qual = rt_flank[indel_len:], rt_qual[indel_len:]
        var = Variant(chrom, pos, pad, pad + indel_seq, reference, skip_validation=True)
    else:
        indel_seq = rt_ref[:indel_len]
        rt_ref = rt_ref[indel_len:]
        var = Variant(chrom, pos, pad + indel_seq, pad, reference, skip_validation=True)
    return pos, lt_flank, indel_seq, rt_flank, lt_ref, rt_ref, lt_qual, rt_qual, var


def is_end_dirty(read_qual, basequalthresh, pos, read_start, read_end, cigar_string):
    """low base quality among the three bases of the read end nearer to the target (pileup.pyx:338-356)"""
    to_left, to_right = pos - read_start, read_end - pos
    lefty = True if to_left < 0 else False if to_right < 0 else to_left <= to_right
    if cigar_string.count("N") > 1:
        return False
    return min(read_qual[:3] if lefty else read_qual[-3:]) < basequalthresh


def leftalign_cigar(cigarstring, target, read_start):
    """move the target's gap in a read's CIGAR to its normalised position (pileup.pyx:359-377)"""
    target.normalize(inplace=True)
    lt, rt = split_cigar(cigarstring, target.pos, read_start)
    if len(rt) < 3 or "M" not in rt[0] or "M" not in rt[2]:
        return cigarstring
    return "".join(lt) + rt[1] + str(int(rt[0][:-1]) + int(rt[2][:-1])) + "M" + "".join(rt[3:])


def parse_spliced_read(cigar_string, read_start, read_end, pos, rpos):
    """(is_covering, covering_subread, is_spliced, splice_pattern, intron_pattern) of a read relative to the target at
    pos (right-most equivalent position rpos) (pileup.pyx:380-432)"""
    pieces = get_spliced_subreads(cigar_string, read_start, read_end)
    covering, sub = False, None
    for p in pieces:
        if p[0] <= pos <= p[1]:
            covering, sub = True, tuple(p)
        elif p[0] <= rpos <= p[1]:
            covering, sub = True, tuple(p)
            pos = rpos
    if len(pieces) <= 1:
        return covering, sub, False, ("", ""), (0, 0)
    bounds = to_flat_list(pieces)[1:-1]
    lt, rt, intron = [], [], (0, 0)
    for i in range(0, len(bounds), 2):
        a, b = bounds[i] + 1, bounds[i + 1] - 1
        if b < pos:
            lt.append("%d-%d" % (a, b))
        elif pos < a - 1:
            rt.append("%d-%d" % (a, b))
        if a - 4 <= pos <= b:                       # overhang reads
            intron = (a, b)
    return covering, sub, True, (":".join(lt), ":".join(rt)), intron


def is_within_intron(read, pos, window):
    a, b = read["intron_pattern"]
    return (a, b) != (0, 0) and a < pos - window and pos + window < b


def fetch_reads(chrom, pos, bam, ref_len, window, exclude_duplicates):
    """pileup.pyx:126-153 (including its quirk: with exclude_duplicates a read starting at reference position 0 is dropped)"""
    pos -= 1
    segs = bam.fetch(chrom, max(0, pos - window), min(pos + 1 + window, ref_len), until_eof=True)
    if exclude_duplicates:
        return [s for s in segs if not s.is_duplicate and not s.is_secondary and s.cigarstring and s.reference_start]
    return [s for s in segs if not s.is_secondary and s.cigarstring]


def dictize_read(read, chrom, pos, rpos, reference, unspl_loc_ref, basequalthresh):
    """one BAM segment as the dict the rest of indelPost works on (pileup.pyx:156-266)"""
    cigar_string = read.cigarstring
    cigar_list = cigar_ptrn.findall(cigar_string)
    aln_start = read.reference_start + 1
    start_offset = int(cigar_list[0][:-1]) if cigar_list[0].endswith("S") else 0
    read_start = aln_start - start_offset
    aln_end = read.reference_end                       # (0-based exclusive = 1-based inclusive: not incremented)
    if aln_end is None:
        aln_end = aln_start + sum(int(c[:-1]) for c in cigar_list if c[-1] in ("M", "N", "D", "=", "X"))
    end_offset = int(cigar_list[-1][:-1]) if cigar_list[-1].endswith("S") else 0
    read_end = aln_end + end_offset
    read_seq, read_qual = read.query_sequence, read.query_qualities
    ref_seq = get_ref_seq(chrom, aln_start, aln_end, cigar_string, cigar_list, reference, unspl_loc_ref)
    d = {"read": read, "read_seq": read_seq, "read_qual": read_qual, "ref_seq": ref_seq, "is_reverse": read.is_reverse,
         "read_name": read.query_name, "mapq": read.mapping_quality, "start_offset": start_offset, "aln_start": aln_start,
         "read_start": read_start, "end_offset": end_offset, "aln_end": aln_end, "read_end": read_end,
         "cigar_string": cigar_string, "cigar_list": cigar_list, "is_reference_seq": read_seq == ref_seq, "I": [], "D": []}
    d["low_qual_base_num"] = count_lowqual_non_ref_bases(read_seq, ref_seq, read_qual, cigar_list, basequalthresh)
    d["is_end_dirty"] = is_end_dirty(read_qual, basequalthresh, pos, read_start, read_end, cigar_string)
    d["is_dirty"] = sum(q <= basequalthresh for q in read_qual) / len(read_seq) > 0.15
    insertions, deletions = locate_indels(cigar_string, read_start)
    for key, events in (("I", insertions), ("D", deletions)):
        for ev_pos, ev_len in events:
            d[key].append(leftalign_indel_read(chrom, ev_pos, ev_len, key, cigar_string, read_start, aln_start, read_seq, ref_seq,
                                               read_qual, reference))
    (d["is_covering"], d["covering_subread"], d["is_spliced"], d["splice_pattern"], d["intron_pattern"]) = parse_spliced_read(
        cigar_string, read_start, read_end, pos, rpos)
    return d


def make_pileup(target, bam, unspl_loc_ref, exclude_duplicates, window, downsamplethresh, basequalthresh):
    """(pileup, sample_factor): the reads within `window` of the target as dicts, downsampled like the reference does
    (random.seed(123), at most downsamplethresh deep, never below half of it) (pileup.pyx:51-111)"""
    chrom, pos, reference = target.chrom, target.pos, target.reference
    rpos = max(v.pos for v in target.generate_equivalents())
    ref_len = reference.get_reference_length(chrom)
    if chrom in bam.references:
        bam_chrom = chrom
    else:
        bam_chrom = chrom.replace("chr", "") if chrom.startswith("chr") else "chr" + chrom
    segs = fetch_reads(bam_chrom, pos, bam, ref_len, window, exclude_duplicates)
    depth = bam.count(bam_chrom, pos - 1, pos, read_callback="all" if exclude_duplicates else "nofilter")
    n_reads, sample_factor = len(segs), 1.0
    if depth > downsamplethresh:
        random.seed(123)
        n_sample = int(n_reads * (downsamplethresh / depth))
        if n_sample >= downsamplethresh / 2 > 0:
            segs = random.sample(segs, n_sample)
            sample_factor = n_reads / len(segs)
    pileup = [dictize_read(s, chrom, pos, rpos, reference, unspl_loc_ref, basequalthresh) for s in segs]
    return [r for r in pileup if not is_within_intron(r, pos, window)], sample_factor


# =====================================================================================================================
# retarget / update_read_info / the overhang filter with the reference's signatures (pileup.pyx:495-913), composed from the
# pieces above and in retarget.py.  The alignments are batched: retarget_many() runs the read pre-filter, the per-read
# windows, every alignment of every (gap_open, gap_ext) pair still in play as ONE GPU call per recursion level (the
# reference's window / 3 recursion for insertions, pileup.pyx:715-730 and 795-808, becomes one more batch per level, not one
# more call per read), and the candidate selection (equivalents, difflib best match, `within`, complex reduction,
# pileup.pyx:732-793).  retarget() is the one-pair case.  Parity: pinned by vectors from the reference's own function text
# (oracle/gen_driver_golden.py).
# =====================================================================================================================
from difflib import SequenceMatcher, get_close_matches


def _retarget_reads(target, pileup, mapq4retarget):
    """the reads retarget realigns (pileup.pyx:592-634)"""
    if target.is_ins:
        non_refs = [r for r in pileup if not r["is_reference_seq"] and r["is_covering"] and r["mapq"] > mapq4retarget]
    else:
        non_refs = [r for r in pileup if not r["is_reference_seq"] and r["mapq"] > mapq4retarget]
    if not non_refs:
        return non_refs
    clean = [r for r in non_refs if r["low_qual_base_num"] < 6 and not r["is_dirty"] and not r["is_end_dirty"] and r.get("is_worth_realn", True)]
    return clean if clean else [r for r in non_refs if not r["is_dirty"]]


def _retarget_select(target, non_refs, alns, ref_seqs, ref_starts, make, window, within, retargetcutoff, require_exact_for_shiftable):
    """everything retarget does AFTER its alignments (pileup.pyx:650-808).  Returns ("done", result-or-None) or
    ("shrink", window // 3): the reference's recursion with a smaller window.  make(ref_seq) builds the aligner object the
    reference hands back per candidate read (only the reads of the final answer get one)."""
    from .variant import Variant
    target_type = target.variant_type
    cutoff = 1.0 if len(target.indel_seq) < 3 else retargetcutoff
    complex_flags, cand, cand_reads, cand_refs, cand_starts = [], [], [], [], []
    for read, aln, ref_seq, ref_start in zip(non_refs, alns, ref_seqs, ref_starts):
        if not aln.CIGAR:
            continue
        seq = read["read_seq"]
        aligned_frac = (aln.read_end - aln.read_start) / min(len(seq), window * 6)
        gaps = aln.CIGAR.count("I") + aln.CIGAR.count("D")
        if not (0 < gaps < 6 and aligned_frac > 0.7):
            continue
        indels = findall_indels(aln, ref_start + aln.reference_start, ref_seq, seq)
        positions = [d["pos"] for d in indels]
        complex_positions = set(p for p in positions if positions.count(p) == 2)
        if complex_positions:
            complex_flags.append(1)
        end_thresh = max(len(seq) / 30, 3)
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
            var = Variant(target.chrom, d["pos"], ref, alt, target.reference, skip_validation=True)
            at_end = var.pos - read["read_start"] <= end_thresh or read["read_end"] - var.pos <= end_thresh
            if at_end and not (var == target or (complex_positions and var.pos not in complex_positions)):
                continue                                             # a non-target indel at a read end is not considered
            cand.append(var); cand_reads.append(read); cand_refs.append(ref_seq); cand_starts.append(ref_start)
    shrink = ("shrink", int(window / 3)) if (target.is_ins and window > 3) else ("done", None)
    if not cand:
        return shrink                                                # (a long window may align an insertion as a deletion)
    if len(target.indel_seq) <= 3 and not sum(complex_flags) and target not in cand:
        return "done", None
    u_cand = to_flat_list([v._generate_equivalents_private() for v in set(cand)])
    u_cand.sort(key=lambda x: abs(x.pos - target.pos))
    seqs = [v._get_indel_seq(how=target_type) for v in u_cand]
    best = get_close_matches(target.indel_seq, seqs, n=1, cutoff=cutoff)
    if not best:
        return shrink
    ratio = SequenceMatcher(None, target.indel_seq, best[0]).ratio()
    hit = u_cand[seqs.index(best[0])]
    if require_exact_for_shiftable and (len(hit.generate_equivalents()) > 1 or len(target.generate_equivalents()) > 1) and hit != target:
        return "done", None
    if not abs(target.pos - hit.pos) < within:
        return "done", None
    try:
        k = cand.index(hit)                                          # the original representation: no normalisation here
    except ValueError:
        hit.pos = hit.pos - len(hit.ref)
        k = cand.index(hit)
    candidate = cand[k]
    idx = [i for i, v in enumerate(cand) if v == candidate]
    if candidate.is_non_complex_indel():                             # can it be the del / ins component of a complex indel?
        for cplx in [v for v in set(cand) if not v.is_non_complex_indel()]:
            reduced = cplx._reduce_complex_indel(to=target_type)
            if candidate == reduced:
                idx = [i for i, v in enumerate(cand) if v == cplx]
                candidate = reduced
                break
    else:
        candidate = candidate._reduce_complex_indel(to=target_type)
    pick = lambda lst: [lst[i] for i in idx]
    return "done", (candidate, pick(cand_reads), ratio, pick(cand_refs), pick(cand_starts), [make(w) for w in pick(cand_refs)])


class _RetargetSearch:
    """One locus' retarget under a set of (gap_open, gap_ext) pairs, as a resumable search: jobs() lists the alignments the
    current recursion level needs, feed() takes them and moves every pair to its answer or to the next (smaller) window.
    retarget_many drives one search; grid_search_many drives the searches of many loci level by level, so that ALL their
    alignments of a level are one GPU batch."""

    def __init__(self, target, pileup, window, mapq4retarget, within, retargetcutoff, match_score, mismatch_penalty, gap_pairs, unspl_loc_ref,
                 require_exact_for_shiftable):
        self.target, self.within, self.cutoff, self.exact = target, within, retargetcutoff, require_exact_for_shiftable
        self.match_score, self.mismatch_penalty, self.gap_pairs, self.unspl = match_score, mismatch_penalty, list(gap_pairs), unspl_loc_ref
        self.non_refs = _retarget_reads(target, pileup, mapq4retarget)
        self.results, self.used = [None] * len(self.gap_pairs), [{} for _ in self.gap_pairs]
        self.level = {g: window for g in range(len(self.gap_pairs))} if self.non_refs else {}
        self._geom, self._order = {}, []

    @property
    def done(self):
        return not self.level

    def jobs(self):
        """(read_seqs, window strings, gap_opens, gap_exts) of the current level, pair-major"""
        from .retarget import get_local_reference
        self._geom = {}
        for w in set(self.level.values()):
            refs, starts = [], []
            for read in self.non_refs:
                ref_seq, lt_len = get_local_reference(self.target, [read], w, self.unspl)
                refs.append(ref_seq)
                starts.append(self.target.pos + 1 - lt_len)
            self._geom[w] = (refs, starts)
        self._order = sorted(self.level)
        R, W, GO, GE = [], [], [], []
        for g in self._order:
            for read, ref_seq in zip(self.non_refs, self._geom[self.level[g]][0]):
                R.append(read["read_seq"]); W.append(ref_seq); GO.append(self.gap_pairs[g][0]); GE.append(self.gap_pairs[g][1])
        return R, W, GO, GE

    def feed(self, alns):
        from .localn import make_aligner
        n, nxt = len(self.non_refs), {}
        make = lambda ref_seq: make_aligner(ref_seq, self.match_score, self.mismatch_penalty)
        for q, g in enumerate(self._order):
            w = self.level[g]
            refs, starts = self._geom[w]
            mine = alns[q * n:(q + 1) * n]
            verdict, val = _retarget_select(self.target, self.non_refs, mine, refs, starts, make, w, self.within, self.cutoff, self.exact)
            if verdict == "shrink":
                nxt[g] = val
            else:
                self.results[g] = val
                self.used[g] = {id(r): a for r, a in zip(self.non_refs, mine)}
        self.level = nxt


def run_retarget_searches(searches, device=0):
    """drive any number of _RetargetSearch objects to completion: one GPU batch per recursion level for ALL of them.  (Searches
    must share the scoring: match / mismatch of the first one is used for the batch.)"""
    from .retarget import align_many
    live = [s_ for s_ in searches if not s_.done]
    while live:
        R, W, GO, GE, cuts = [], [], [], [], []
        for s_ in live:
            r, w, go, ge = s_.jobs()
            R += r; W += w; GO += go; GE += ge
            cuts.append(len(r))
        alns = align_many(R, W, GO, GE, live[0].match_score, live[0].mismatch_penalty, device)
        at = 0
        for s_, k in zip(live, cuts):
            s_.feed(alns[at:at + k])
            at += k
        live = [s_ for s_ in live if not s_.done]


def retarget_many(target, pileup, window, mapq4retarget, within, retargetcutoff, match_score, mismatch_penalty, gap_pairs, unspl_loc_ref,
                  require_exact_for_shiftable, device=0):
    """retarget (pileup.pyx:577-808) for every (gap_open, gap_ext) of gap_pairs at once.  Returns (results, alignments):
    results[g] = what retarget(..., gap_pairs[g][0], gap_pairs[g][1], ...) returns (None, or the 6-tuple with SSW aligner
    objects last); alignments[g] = {id(read): Alignment} of the level that produced results[g] (update_read_info makes the same
    alignment again in the reference; the caller can reuse it)."""
    s_ = _RetargetSearch(target, pileup, window, mapq4retarget, within, retargetcutoff, match_score, mismatch_penalty, gap_pairs, unspl_loc_ref,
                         require_exact_for_shiftable)
    run_retarget_searches([s_], device)
    return s_.results, s_.used


def retarget(target, pileup, window, mapq4retarget, within, retargetcutoff, match_score, mismatch_penalty, gap_open_penalty,
             gap_extension_penalty, unspl_loc_ref, require_exact_for_shiftable, device=0):
    """pileup.pyx:577-808, same arguments, same return value: None, or (candidate, candidate_reads, match ratio,
    candidate_ref_seqs, candidate_ref_starts, candidate_aligners)"""
    return retarget_many(target, pileup, window, mapq4retarget, within, retargetcutoff, match_score, mismatch_penalty,
                         [(gap_open_penalty, gap_extension_penalty)], unspl_loc_ref, require_exact_for_shiftable, device)[0][0]


def update_read_info(read, candidate, is_gapped_aln=True, gap_open_penalty=3, gap_extension_penalty=1, aligner=None, ref_seq=None,
                     ref_start=None, aln=None):
    """pileup.pyx:811-913, same arguments (+ aln: the read's alignment when the caller already has it -- grid_search does).
    Gapped branch: the read carries the candidate in its own CIGAR; realignment branch: see update_read_info_realn."""
    if is_gapped_aln:
        parsed = leftalign_indel_read(candidate.chrom, candidate.pos, len(candidate.indel_seq), candidate.variant_type, read["cigar_string"],
                                      read["read_start"], read["aln_start"], read["read_seq"], read["ref_seq"], read["read_qual"],
                                      candidate.reference)
        read["lt_flank"] = parsed[1]
        read["indel_seq"] = parsed[2] if candidate.is_ins else ""
        read["rt_flank"], read["lt_ref"], read["rt_ref"], read["lt_qual"], read["rt_qual"] = parsed[3], parsed[4], parsed[5], parsed[6], parsed[7]
        read["lt_cigar"], read["rt_cigar"] = split_cigar(read["cigar_string"], candidate.pos, read["read_start"])
        read["is_target"] = True
        return read
    if aln is None:
        from .localn import align
        aln = align(aligner, read["read_seq"], gap_open_penalty, gap_extension_penalty)
    from .variant import Variant
    same = lambda pos, ref, alt: candidate == Variant(candidate.chrom, pos, ref, alt, candidate.reference, skip_validation=True)
    return update_read_info_realn(read, aln, ref_seq, ref_start, candidate.pos, candidate.indel_seq, candidate.is_ins, same)


def is_junctional(read):
    return read["is_covering"] if read["intron_pattern"] == (0, 0) else True


def check_overhangs(pileup, splice_rate=0.2):
    """(intron, overhanging reads) when the locus sits at a well-supported exon boundary (pileup.pyx:427-451)"""
    from .retarget import most_common
    intron_ptrns = [r["intron_pattern"] for r in pileup if is_junctional(r)]
    introns = [p for p in intron_ptrns if p != (0, 0)]
    if not introns:
        return None
    intron = most_common(introns)
    if intron_ptrns.count(intron) / len(intron_ptrns) < splice_rate:
        return None
    overhangs = [r for r in pileup if is_overhang(r, intron[0], intron[1])]
    return (intron, overhangs) if overhangs else None


def is_overhang(read, intron_start, intron_end):
    """the covering piece of the read hangs over one edge of the intron (pileup.pyx:461-474)"""
    sub = read["covering_subread"]
    if not sub:
        return False
    lt, rt = max(sub[0], read["aln_start"]), min(sub[1], read["aln_end"])
    return (lt < intron_start and rt < intron_end) or (intron_start < lt and intron_end < rt)


def overhang_windows(target, intron):
    """(genome window, exon-exon junction window) of overhang_aligners (pileup.pyx:477-492)"""
    genome_ref = target.reference.fetch(target.chrom, target.pos - 100, target.pos + 100)
    lt_exon_end, rt_exon_start = intron[0] - 1, intron[1]
    junction_ref = target.reference.fetch(target.chrom, lt_exon_end - 100, lt_exon_end) + target.reference.fetch(target.chrom, rt_exon_start, rt_exon_start + 100)
    return genome_ref, junction_ref


def filter_spurious_overhangs(target, intron, overhangs, match_score, mismatch_penalty, gap_open_penalty, gap_extension_penalty, device=0):
    """pileup.pyx:495-525 / 527-574, same arguments: the overhanging reads that are not spurious.  Both alignments of every
    non-reference overhang are one GPU batch; the read-level checks follow per read as in the reference."""
    from .localn import findall_mismatches_pileup, worth_realn_mask
    from .retarget import overhang_alignment_verdicts
    genome_ref, junction_ref = overhang_windows(target, intron)
    todo = [r for r in overhangs if not r["is_reference_seq"]]
    verdicts, _ = overhang_alignment_verdicts([r["read_seq"] for r in todo], genome_ref, junction_ref, match_score, mismatch_penalty,
                                              gap_open_penalty, gap_extension_penalty, device)
    lt_exon_end, rt_exon_start = intron[0] - 1, intron[1]
    keep, ask = set(), []
    for k, (read, v) in enumerate(zip(todo, verdicts)):
        if v is False:
            continue
        # (the reference's expression: `read["D"] and read["I"]` is read["I"] when the read has deletions, else the empty list;
        #  the list of booleans is only tested for being non-empty)
        if [lt_exon_end < var[-1].pos < rt_exon_start for var in (read["D"] and read["I"])]:
            keep.add(k)
        else:
            ask.append(k)
    if ask:                                                # the read-level checks of every read still open, as arrays over those reads
        sub = findall_mismatches_pileup([todo[k] for k in ask])
        keep.update(np.asarray(ask)[worth_realn_mask(sub, target)].tolist())
    out = [todo[k] for k in sorted(keep)]
    return out
