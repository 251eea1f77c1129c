"""The aligner helpers and the batched-align driver of ``indelpost.localn`` on MI355X.

Mirrors (all citations into /root/reference/indelpost/localn.pyx):
  make_aligner, align                    :464-472   (same names, same arguments)
  realign_pileup_jobs / align_pileup     the SSW part of find_by_smith_waterman_realn :15-68 and
                                         is_target_by_ssw :223-259 -- two alignments per read
                                         (reference contig with (go, ge); mutant contig with
                                         gap_open=len(read)), collapsed into one GPU batch.
  is_covering_target, is_compatible_repeats, classify_realigned / find_targets_by_ssw
                                         the classification that follows the two alignments
                                         (is_target_by_ssw :257-291, is_covering_target :293-430,
                                         is_compatible_repeats :433-459): score compare over whole result
                                         columns, string checks only for the reads that pass it.
find_targets_by_ssw takes the alignment-independent filters (:244-249) as a mask; find_by_smith_waterman_realn (further
down) is the whole function with the reference's signature, filters included, and find_by_smith_waterman_realn_many
runs it for many loci in one GPU batch.
"""
import numpy as np

from .batch import JobTable, encode_dna
from .cigar import cigar_ptrn, to_minimal_repeat_unit
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


def is_compatible_repeats(seq, repeat_unit, expected_n_repeats, is_left):
    """localn.pyx:433-459: walk whole repeat units off the indel-facing end of a flank (the left flank is read
    right to left).  Incompatible when the flank is nothing but repeats, or holds some but not the expected number."""
    if is_left:
        seq, repeat_unit = seq[::-1], repeat_unit[::-1]
    n, cnt = len(repeat_unit), 0
    while seq and seq[:n] == repeat_unit:
        seq = seq[n:]
        cnt += 1
    if not seq:
        return False
    return not (cnt and cnt != expected_n_repeats)


def is_covering_target(readname, read_seq, indel_seq, mut_ref_lt, mut_ref_mid, mut_ref_rt, mut_aln_cigar, read_seq_len,
                       ref_aln_start, ref_aln_end, read_aln_start, read_aln_end, n_repeats):
    """localn.pyx:293-430.  Does the read's UNGAPPED alignment to the mutant contig (lt + mid + rt; mid = inserted
    sequence, empty for a deletion) really span the indel?  1 yes, 0 no, -1 undetermined (flank repeats disagree)."""
    toks = cigar_ptrn.findall(mut_aln_cigar)
    if len(toks) > 1:
        return 0
    unit = to_minimal_repeat_unit(indel_seq)
    int(toks[0][:-1])                                                  # (mapped_len, :314: unused, but an empty CIGAR raises here as it does there)
    lt_len, mid_len = len(mut_ref_lt), len(mut_ref_mid)
    consumed = read_aln_end - read_aln_start + 1
    from_read_start, to_read_end = read_aln_start == 0, read_aln_end == read_seq_len - 1
    if ref_aln_end < lt_len or lt_len + mid_len <= ref_aln_start:       # the alignment misses the indel altogether
        return 0
    if mid_len:                                                        # insertion
        if ref_aln_start < lt_len:
            lt_use = lt_len - ref_aln_start
            if consumed > lt_use + mid_len:                            # through the insertion and out the other side
                a = read_aln_start + lt_use
                lt_ok = is_compatible_repeats(read_seq[read_aln_start:a], unit, n_repeats, is_left=True)
                rt_ok = is_compatible_repeats(read_seq[a + mid_len:read_aln_end + 1], unit, n_repeats, is_left=False)
                return 1 if (lt_ok and rt_ok) else -1
            if to_read_end:
                return 1
            rt_use = consumed - lt_use                                  # entered from the left, stopped inside the insertion
            return 1 if mut_ref_mid[:rt_use] == read_seq[-rt_use:] else 0
        if from_read_start:                                             # no left flank aligned at all
            return 1
        lt_use = lt_len + mid_len - ref_aln_start                       # entered from the right, stopped inside the insertion
        return 1 if mut_ref_mid[-lt_use:] == read_seq[:lt_use] else 0
    lt_use = lt_len - ref_aln_start                                     # deletion
    rt_use = consumed - lt_use
    a = read_aln_start + lt_use
    if not (is_compatible_repeats(read_seq[read_aln_start:a], unit, n_repeats, is_left=True)
            and is_compatible_repeats(read_seq[a:read_aln_end], unit, n_repeats, is_left=False)):
        return -1
    if lt_use <= rt_use:
        return 1 if (from_read_start or lt_use > 2) else 0
    return 1 if (to_read_end or rt_use > 2) else 0


def classify_realigned(read_seqs, pairs, indel_seq, mut_ref_lt, mut_ref_mid, mut_ref_rt, n_repeats, read_names=None):
    """The verdict of is_target_by_ssw (localn.pyx:257-291) for every realigned read at once.

    pairs[k] = (ref_aln, mut_aln) as align_pileup returns them.  Returns (is_target, undetermined): two bool arrays.
    The score compare runs over the whole columns; only the reads whose mutant-contig score beats the reference-contig
    score go through the string checks of is_covering_target."""
    n = len(pairs)
    is_target, undetermined = np.zeros(n, bool), np.zeros(n, bool)
    if n == 0:
        return is_target, undetermined
    ref_score = np.fromiter((p[0].optimal_score for p in pairs), np.int64, n)
    mut_score = np.fromiter((p[1].optimal_score for p in pairs), np.int64, n)
    for k in np.flatnonzero(mut_score > ref_score):                    # localn.pyx:257
        m = pairs[k][1]
        seq = read_seqs[k]
        v = is_covering_target(read_names[k] if read_names else "", seq, indel_seq, mut_ref_lt, mut_ref_mid, mut_ref_rt, m.CIGAR,
                               len(seq), m.reference_start, m.reference_end, m.read_start, m.read_end, n_repeats)
        if v == 1:
            is_target[k] = True
        elif v == -1:
            undetermined[k] = True
    return is_target, undetermined


def find_targets_by_ssw(read_seqs, realign_mask, indel_seq, n_repeats, mut_ref_lt, mut_ref_mid, mut_ref_rt, ref_ref,
                        match_score, mismatch_penalty, gap_open_penalty, gap_extension_penalty, device=0):
    """find_by_smith_waterman_realn (localn.pyx:15-68) end to end for one locus: the reads selected by `realign_mask`
    (the alignment-independent filters of localn.pyx:244-249, decided by the caller) are aligned to the reference and
    the mutant contig in ONE GPU batch and classified.  Returns (is_target, undetermined, pairs): bool arrays over ALL
    reads (False where the mask is False) and the Alignment pairs (None where not aligned)."""
    n = len(read_seqs)
    mask = np.asarray(realign_mask, bool)
    idx = np.flatnonzero(mask)
    sel = [read_seqs[i] for i in idx]
    pairs = align_pileup(sel, mut_ref_lt + mut_ref_mid + mut_ref_rt, ref_ref, match_score, mismatch_penalty,
                         gap_open_penalty, gap_extension_penalty, device)
    t, u = classify_realigned(sel, pairs, indel_seq, mut_ref_lt, mut_ref_mid, mut_ref_rt, n_repeats)
    is_target, undetermined = np.zeros(n, bool), np.zeros(n, bool)
    is_target[idx], undetermined[idx] = t, u
    out = [None] * n
    for i, p in zip(idx, pairs):
        out[i] = p
    return is_target, undetermined, out


# =====================================================================================================================
# find_by_smith_waterman_realn as ONE function with the reference's signature (localn.pyx:15-68): the alignment-independent
# filters (findall_mismatches :71-136, is_worth_realn :139-220), the two alignments per remaining read as one GPU batch, and
# the verdict written into the read dicts exactly as is_target_by_ssw does (:223-291).  parse_read_by_mut_aln (:475-539)
# is the decomposition of a read along its mutant-contig alignment (the reference defines it but calls it nowhere).
# Parity: pinned by vectors from the reference's own function text run against duck-typed reads / contigs with
# make_aligner / align bound to the oracle (oracle/gen_driver_golden.py, tests/golden/driver_cases.json).
# =====================================================================================================================
def findall_mismatches(read, end_trim=0):
    """read["mismatches"] = [(pos, ref base, read base, quality)] over the read's mapped blocks (localn.pyx:71-136)"""
    from .pileup import get_mapped_subreads, split
    if read["is_reference_seq"]:
        read["mismatches"] = []
        return read
    aln_start, aln_end = read["aln_start"], read["aln_end"]
    out = []
    for start, end in get_mapped_subreads(read["cigar_string"], aln_start, aln_end):
        span = end - start + 1
        cigarstring = read["cigar_string"]
        read_seq, quals = read["read_seq"], read["read_qual"]
        if "S" in cigarstring:                                       # soft clips are cut off both the CIGAR and the read
            cigarlst = read["cigar_list"]
            if "S" in cigarlst[0]:
                cigarlst = cigarlst[1:]
                read_seq, quals = read_seq[read["start_offset"]:], quals[read["start_offset"]:]
            if "S" in cigarlst[-1]:
                cigarlst = cigarlst[:-1]
                read_seq, quals = read_seq[:-read["end_offset"]], quals[:-read["end_offset"]]
            cigarstring = "".join(cigarlst)
        lt_seq, rt_seq = split(read_seq, cigarstring, start, aln_start, is_for_ref=False, reverse=False)
        lt_qual, rt_qual = split(quals, cigarstring, start, aln_start, is_for_ref=False, reverse=False)
        lt_ref, rt_ref = split(read["ref_seq"], cigarstring, start, aln_start, is_for_ref=True, reverse=False)
        mapped_seq = lt_seq[-1] + rt_seq[:span - 1]
        mapped_qual = [lt_qual[-1]] + list(rt_qual[:span - 1])
        mapped_ref = lt_ref[-1] + rt_ref[:span - 1]
        pos = start
        for r, a, q in zip(mapped_ref, mapped_seq, mapped_qual):
            if r != a and aln_start + end_trim <= pos <= aln_end - end_trim:
                out.append((pos, r.upper(), a, q))
            pos += 1
    read["mismatches"] = out
    return read


def is_worth_realn(read, target_indel, qual_lim=23):
    """could a realignment of this read show the target?  (localn.pyx:139-220: clipped at the locus, high-quality mismatches
    or indels over the target's span; not when the read ends inside the target's repeat and matches the reference there)"""
    if read["covering_subread"]:
        is_covered = True
        covering_start, covering_end = read["covering_subread"][0], read["covering_subread"][1]
    else:
        is_covered = False
        if target_indel.is_ins:
            return False
        covering_start = target_indel.pos
        covering_end = covering_start + len(target_indel.ref)
    to_left, to_right = target_indel.pos - read["aln_start"], read["aln_end"] - target_indel.pos
    is_lefty = True if to_left < 0 else False if to_right < 0 else to_left <= to_right
    start_cigar, end_cigar = read["cigar_list"][0], read["cigar_list"][-1]
    if is_lefty and covering_start < read["aln_start"] <= covering_end and int(start_cigar[:-1]) > 2:
        return True
    if not is_lefty and covering_start <= read["aln_end"] < covering_end and int(end_cigar[:-1]) > 2:
        return True
    mismatches = [m for m in read["mismatches"] if covering_start <= m[0] <= covering_end and m[3] > qual_lim]
    shiftable = [v.pos for v in target_indel.generate_equivalents()]
    lt_pos, rt_pos = min(shiftable), max(shiftable)
    if lt_pos < rt_pos:
        if is_lefty:
            if lt_pos < read["aln_start"]:
                k = rt_pos - read["aln_start"]
                if read["read_seq"][:k] == read["ref_seq"][:k]:
                    return False
        elif read["aln_end"] <= rt_pos:
            k = read["aln_end"] - lt_pos
            if read["read_seq"][-k:] == read["ref_seq"][-k:]:
                return False
    if mismatches:
        if is_lefty:
            at_end = abs(min(m[0] for m in mismatches) - read["aln_start"]) < 4
        else:
            at_end = abs(max(m[0] for m in mismatches) - read["aln_end"]) < 4
        return True if at_end else is_covered
    return bool([v for v in read["I"] + read["D"] if covering_start <= v[0] <= covering_end])


def _needs_realn(read, target_indel, mapq_lim):
    """the filters of is_target_by_ssw that do not depend on the alignments (localn.pyx:243-249): None = already a target (left
    as it is), False = not realigned (is_target set False), True = realign"""
    if read["is_target"]:
        return None
    if read["is_reference_seq"] or read["mapq"] <= mapq_lim or not is_worth_realn(read, target_indel):
        return False
    return True


def is_target_by_ssw(read, target_indel, contig, mut_ref_lt, mut_ref_mid, mut_ref_rt, mut_aligner, ref_aligner, match_score,
                     mismatch_penalty, gap_open_penalty, gap_extension_penalty, indel_type, basequalthresh, mapq_lim,
                     mapped_base_cnt_thresh=40, allow_mismatches=10):
    """one read, the reference's signature (localn.pyx:223-291): two single alignments through the aligner objects.  The batched
    form is find_by_smith_waterman_realn below; this one exists for callers that hold a single read."""
    need = _needs_realn(read, target_indel, mapq_lim)
    if need is None:
        return read
    if not need:
        read["is_target"] = False
        return read
    read_seq = read["read_seq"]
    ref_aln = align(ref_aligner, read_seq, gap_open_penalty, gap_extension_penalty)
    mut_aln = align(mut_aligner, read_seq, len(read_seq), gap_extension_penalty)
    return _apply_ssw_verdict(read, target_indel, mut_ref_lt, mut_ref_mid, mut_ref_rt, ref_aln, mut_aln)


def _apply_ssw_verdict(read, target_indel, mut_ref_lt, mut_ref_mid, mut_ref_rt, ref_aln, mut_aln):
    """localn.pyx:257-291 with the two alignments made"""
    if mut_aln.optimal_score <= ref_aln.optimal_score:
        read["is_target"] = False
        return read
    seq = read["read_seq"]
    v = is_covering_target(read["read_name"], seq, target_indel.indel_seq, mut_ref_lt, mut_ref_mid, mut_ref_rt, mut_aln.CIGAR, len(seq),
                           mut_aln.reference_start, mut_aln.reference_end, mut_aln.read_start, mut_aln.read_end,
                           target_indel.count_repeats())
    if v == 1:
        read["is_target"] = True
    elif v == -1:
        read["undetermined"] = True
    return read


def find_by_smith_waterman_realn(target_indel, contig, pileup, match_score, mismatch_penalty, gap_open_penalty, gap_extension_penalty,
                                 basequalthresh, mapq_lim=1, device=0):
    """localn.pyx:15-68, same arguments, same annotated pileup back: every read that passes the filters is aligned to the
    reference contig under (gap_open, gap_ext) and to the mutant contig with gap_open = len(read), all in ONE GPU batch, and
    gets is_target / undetermined as is_target_by_ssw would have set them.  `contig` needs get_contig_seq(split=True) and
    get_reference_seq() (contig.pyx)."""
    return find_by_smith_waterman_realn_many([(target_indel, contig, pileup, match_score, mismatch_penalty, gap_open_penalty,
                                               gap_extension_penalty, basequalthresh, mapq_lim)], device)[0]


def find_by_smith_waterman_realn_many(requests, device=0):
    """find_by_smith_waterman_realn for MANY loci in ONE GPU batch: requests = (target_indel, contig, pileup, match_score,
    mismatch_penalty, gap_open_penalty, gap_extension_penalty, basequalthresh[, mapq_lim]) per locus (same match / mismatch
    everywhere).  Returns the annotated pileups in request order."""
    from .retarget import align_many
    R, W, GO, GE, plans = [], [], [], [], []
    for req in requests:
        target_indel, contig, pileup, ms, mm, go, ge, _bq = req[:8]
        mapq_lim = req[8] if len(req) > 8 else 1
        lt, mid, rt = contig.get_contig_seq(split=True)
        ref_ref, mut_ref = contig.get_reference_seq(), lt + mid + rt
        pileup = [findall_mismatches(read) for read in pileup]
        todo = []
        for k, read in enumerate(pileup):
            need = _needs_realn(read, target_indel, mapq_lim)
            if need is False:
                read["is_target"] = False
            elif need:
                todo.append(k)
                seq = read["read_seq"]
                R += [seq, seq]; W += [ref_ref, mut_ref]; GO += [go, len(seq)]; GE += [ge, ge]      # localn.pyx:253-255
        plans.append((target_indel, pileup, todo, lt, mid, rt))
    alns = align_many(R, W, GO, GE, requests[0][3], requests[0][4], device) if R else []
    at, out = 0, []
    for target_indel, pileup, todo, lt, mid, rt in plans:
        for k in todo:
            _apply_ssw_verdict(pileup[k], target_indel, lt, mid, rt, alns[at], alns[at + 1])
            at += 2
        out.append(pileup)
    return out


def parse_read_by_mut_aln(mut_aln, contig, read, indel_type):
    """the read cut into left flank / indel / right flank along its alignment to the mutant contig (localn.pyx:475-539).
    `contig` needs lt_consensus_seq, indel_seq, rt_consensus_seq."""
    from .pileup import get_end_pos, split
    lt_len, indel_len = len(contig.lt_consensus_seq), len(contig.indel_seq)
    read_seq, read_qual = read["read_seq"], read["read_qual"]
    ref_start, ref_end = mut_aln.reference_start, mut_aln.reference_end
    aln_start, aln_end = mut_aln.read_start, mut_aln.read_end
    lt_flank = mid_seq = rt_flank = ""
    lt_qual, rt_qual = [], []
    if ref_start <= lt_len:
        lt_diff = lt_len - ref_start
        cut = aln_start + lt_diff
        lt_flank, lt_qual = read_seq[aln_start:cut], read_qual[aln_start:cut]
        if indel_type == "I":
            mid_seq = read_seq[cut:min(cut + indel_len, aln_end)]
        else:
            rt_flank, rt_qual = read_seq[cut:], read_qual[cut:]
            del_pos = get_end_pos(read["read_start"] + aln_start, lt_flank, read["cigar_string"])
            _, rt_ref = split(read["ref_seq"], read["cigar_string"], del_pos, read["aln_start"], is_for_ref=True, reverse=False)
            read["del_pos"] = del_pos
            read["del_seq"] = rt_ref[:indel_len]
    if lt_len + indel_len <= ref_end and indel_type == "I":
        rt_diff = ref_end - (lt_len + indel_len)
        rt_flank, rt_qual = read_seq[aln_end - rt_diff:aln_end], read_qual[aln_end - rt_diff:aln_end]
        mid_seq = read_seq[max(aln_start, aln_end - rt_diff - indel_len):aln_end - rt_diff]
    read["lt_flank"], read["lt_qual"], read["indel_seq"], read["rt_flank"], read["rt_qual"] = lt_flank, lt_qual, mid_seq, rt_flank, rt_qual
    return read
