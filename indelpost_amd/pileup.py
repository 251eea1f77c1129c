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
The gapped-alignment branch (leftalign_indel_read, pileup.pyx:822-846) needs the FASTA and is the caller's.
Parity: pinned by vectors from the reference's own function bodies (oracle/gen_decoder_golden.py) for split_cigar,
trim_ref_flank, update_cigar and update_read_positions; update_read_info_realn on top of them by hand-derived cases.
"""
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
