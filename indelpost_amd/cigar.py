"""CIGAR decoding on the host: what indelPost does with the alignments the GPU returns.

Mirrors (citations into /root/reference/indelpost/):
  merge_consecutive_gaps, make_insertion_first   utilities.pyx:360-401
  findall_indels                                 localn.pyx:542-621
  to_minimal_repeat_unit                         utilities.pyx:150-166
Same names, arguments and return shapes (lists of dicts with the reference's keys), so the callers of the
reference (retarget, pileup.pyx:650-711; update_read_info, pileup.pyx:849-880) read the same values.  Their
reference modules cimport pysam and cannot be imported in the build container; parity is PINNED all the same: the
reference's own function bodies, read as text and executed, produced the vectors of tests/golden/decoder_cases.json
(oracle/gen_decoder_golden.py), and tests/test_decoders.py replays them through these functions.
"""
import re

cigar_ptrn = re.compile(r"[0-9]+[MIDNSHPX=]")      # utilities.pyx / localn.pyx:11

_GAP = ("I", "D")


def cigar_tokens(cigarstring):
    """'5M1D11M' -> [(5, 'M'), (1, 'D'), (11, 'M')]"""
    return [(int(t[:-1]), t[-1]) for t in cigar_ptrn.findall(cigarstring)]


def merge_consecutive_gaps(cigar_lst):
    """Runs of adjacent I/D tokens become ONE list element ('2I' '3D' -> '2I3D'); other tokens stay as they are
    (utilities.pyx:360-381).  One quirk of the reference is kept: its look-ahead stops one token short when a run of
    gaps reaches the END of the list, so the last token of such a run stays an element of its own."""
    toks = list(cigar_lst)
    out, k, n = [], 0, len(toks)
    while k < n:
        if toks[k][-1] not in _GAP:
            out.append(toks[k])
            k += 1
            continue
        e = k
        while e < n and toks[e][-1] in _GAP:
            e += 1
        if e == n and e - k >= 2:                  # run reaches the end: the reference joins all but its last token
            out.append("".join(toks[k:e - 1]))
            out.append(toks[e - 1])
        else:
            out.append("".join(toks[k:e]))
        k = e
    return out


def make_insertion_first(cigarstring):
    """Inside every run of adjacent gaps that holds both kinds, and whose FIRST token is a deletion, the tokens are
    reversed -- for the two-token runs SSW produces that puts the insertion first (utilities.pyx:384-401)."""
    parts = []
    for c in merge_consecutive_gaps(cigar_ptrn.findall(cigarstring)):
        if "I" in c and "D" in c:
            toks = cigar_ptrn.findall(c)
            if toks[0][-1] == "D":
                toks = toks[::-1]
            c = "".join(toks)
        parts.append(c)
    return "".join(parts)


def findall_indels(ref_aln, genome_aln_pos, ref_seq, read_seq, report_snvs=False, basequals=None):
    """Every insertion / deletion of an alignment as the dict the reference builds (localn.pyx:542-621).

    ref_aln: Alignment tuple (CIGAR, ..., reference_start, reference_end, read_start, read_end);
    genome_aln_pos: 1-based genome position of the first aligned window base.  Per indel: pos (position of the base
    left of the event), lt_ref / rt_ref (window left / right of it), lt_flank / rt_flank (read likewise), indel_type,
    indel_seq, del_seq (deletions), ref_idx, read_idx, lt_clipped / rt_clipped (unaligned read ends), and lt_qual /
    rt_qual when basequals is given.  With report_snvs also the mismatching positions inside M runs."""
    pos = genome_aln_pos - 1
    ri, qi = ref_aln.reference_start, ref_aln.read_start
    head = read_seq[:qi]
    indels, snvs = [], []
    for n, op in cigar_tokens(make_insertion_first(ref_aln.CIGAR)):
        if op in _GAP:
            d = {"pos": pos, "lt_ref": ref_seq[:ri], "lt_flank": read_seq[:qi]}
            if basequals:
                d["lt_qual"] = basequals[:qi]
            d["indel_type"] = op
            if op == "I":
                d["indel_seq"] = read_seq[qi:qi + n]
                d["rt_ref"] = ref_seq[ri:]
                d["rt_flank"] = read_seq[qi + n:]
            else:
                d["indel_seq"] = ""
                d["del_seq"] = ref_seq[ri:ri + n]
                d["rt_ref"] = ref_seq[ri + n:]
                d["rt_flank"] = read_seq[qi:]
            d["ref_idx"], d["read_idx"] = ri, qi
            if basequals:
                d["rt_qual"] = basequals[qi + n:] if op == "I" else basequals[qi:]
            if op == "I":
                qi += n
            else:
                ri += n
                pos += n
            indels.append(d)
            continue
        if report_snvs:
            for i in range(n):
                a, b = ref_seq[ri + i:ri + i + 1], read_seq[qi + i:qi + i + 1]
                if a != b:
                    snvs.append({"pos": pos + i + 1, "ref": a, "alt": b})
        ri += n
        qi += n
        pos += n
    tail = read_seq[qi:]
    for d in indels:
        d["lt_clipped"], d["rt_clipped"] = head, tail
    return (indels, snvs) if report_snvs else indels


def to_minimal_repeat_unit(seq):
    """Shortest unit whose tandem repetition spells seq exactly ('ATAT' -> 'AT'), else seq (utilities.pyx:150-166)."""
    for j in range(1, len(seq) // 2 + 1):
        if len({seq[i:i + j] for i in range(0, len(seq), j)}) == 1:
            return seq[:j]
    return seq


def gap_count(cigarstring):
    """number of I plus D tokens, as `CIGAR.count("I") + CIGAR.count("D")` (pileup.pyx:663, 513)"""
    return cigarstring.count("I") + cigarstring.count("D")
