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
from bisect import bisect_left

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
def _block_table(read):
    """The mapped blocks of one read, located the way localn.pyx:71-136 locates them.  Returns (seq, ref, front, blocks):
    the read bases with the soft clips the reference cuts off (:86-101: only when the FIRST / LAST CIGAR token is a clip),
    the number of bases cut in front, and per M/=/X run of the untrimmed CIGAR (utilities.pyx:221-240) the tuple
    (genome start, span, kr, kf): kr / kf are the cut points utilities.pyx:429-503 (`split`, forward) arrives at on the trimmed
    read and on ref_seq for that genome position -- the run's bases are data[kr - 1 : kr - 1 + span] when the read is well
    formed.  The cut points come from cumulative (genome, read, reference) advances per token and one bisection per run,
    not from three CIGAR walks per run."""
    cigar = read["cigar_string"]
    every = cigar_ptrn.findall(cigar)
    seq, front, toks = read["read_seq"], 0, every
    if "S" in cigar:
        toks = read["cigar_list"]
        if "S" in toks[0]:
            toks, front = toks[1:], read["start_offset"]
            seq = seq[front:]
        if "S" in toks[-1]:
            toks, seq = toks[:-1], seq[:-read["end_offset"]]
    at, nr, nf = read["aln_start"] - 1, 0, 0
    g_cum, r_cum, f_cum = [at], [0], [0]
    for tok in toks:
        op, n = tok[-1], int(tok[:-1])
        if op == "N":
            at += n
        elif op == "I":
            nr += n
        elif op == "D":
            at += n; nf += n
        elif op not in "HP":
            at += n; nr += n; nf += n
        g_cum.append(at); r_cum.append(nr); f_cum.append(nf)
    last = len(toks)
    blocks, pos = [], read["aln_start"]
    for tok in every:
        op, n = tok[-1], int(tok[:-1])
        if op in "M=X":
            c = min(bisect_left(g_cum, pos), last)          # tokens the forward walk consumes before it has reached `pos`
            over = g_cum[c] - pos
            blocks.append((pos, n, r_cum[c] - over, f_cum[c] - over))
            pos += n
        elif op not in "ISHP":
            pos += n
    return seq, read["ref_seq"], front, blocks


def _trimmed_quals(read):
    """read_qual under the same clip trimming as the bases (localn.pyx:90-99)"""
    quals, toks = read["read_qual"], read["cigar_list"]
    if "S" in read["cigar_string"]:
        if "S" in toks[0]:
            toks, quals = toks[1:], quals[read["start_offset"]:]
        if "S" in toks[-1]:
            quals = quals[:-read["end_offset"]]
    return quals


def _mismatches_by_slicing(read, seq, ref, blocks, end_trim):
    """a read whose cut points leave the sequences (hard clip in front of a soft clip, CIGAR and sequence lengths that disagree):
    Python's slice rules decide what the reference compares -- negative and overlong indices included, and an empty left part
    raises IndexError there as here (localn.pyx:120-122)"""
    quals, out = _trimmed_quals(read), []
    lo, hi = read["aln_start"] + end_trim, read["aln_end"] - end_trim
    for start, span, kr, kf in blocks:
        bases = seq[:kr][-1] + seq[kr:][:span - 1]
        q = [quals[:kr][-1]] + list(quals[kr:][:span - 1])
        under = ref[:kf][-1] + ref[kf:][:span - 1]
        out += [(start + i, r.upper(), a, qq) for i, (r, a, qq) in enumerate(zip(under, bases, q)) if r != a and lo <= start + i <= hi]
    return out


def findall_mismatches_pileup(pileup, end_trim=0):
    """read["mismatches"] = [(pos, reference base, read base, quality)] for every read of a pileup (localn.pyx:71-136 per read): the
    mapped runs of ALL reads are compared in one pass over two byte buffers (read bases, reference bases under them); Python
    objects are made for the differing bases only.  Returns the pileup."""
    rows, seqs, refs = [], [], []                                  # rows: (read index, genome start, span, offset in seqs, offset in refs)
    at_s = at_f = 0
    held = []                                                      # reads compared in the buffers: (index, kr - 1 of each block)
    for i, read in enumerate(pileup):
        if read["is_reference_seq"]:
            read["mismatches"] = []
            continue
        seq, ref, front, blocks = _block_table(read)
        ls, lf = len(seq), len(ref)
        plain = len(read["read_qual"]) == len(read["read_seq"]) and seq.isascii() and ref.isascii()
        for start, span, kr, kf in blocks:
            if not (1 <= kr and kr - 1 + span <= ls and 1 <= kf and kf - 1 + span <= lf):
                plain = False
        if not plain:
            read["mismatches"] = _mismatches_by_slicing(read, seq, ref, blocks, end_trim)
            continue
        read["mismatches"] = []
        for start, span, kr, kf in blocks:
            rows.append((i, start, span, at_s + kr - 1, at_f + kf - 1))
        held.append(i)
        seqs.append(seq); refs.append(ref)
        at_s += ls; at_f += lf
    if not rows:
        return pileup
    T = np.array(rows, np.int64)
    who, start, span, in_s, in_f = T.T
    first = np.zeros(len(T), np.int64)
    np.cumsum(span[:-1], out=first[1:])
    run = np.repeat(np.arange(len(T)), span)
    step = np.arange(int(span.sum())) - first[run]
    S = np.frombuffer("".join(seqs).encode("ascii"), np.uint8)
    F = np.frombuffer("".join(refs).encode("ascii"), np.uint8)
    a, r = S[in_s[run] + step], F[in_f[run] + step]
    pos = start[run] + step
    a0 = np.fromiter((pileup[i]["aln_start"] for i in held), np.int64, len(held))
    a1 = np.fromiter((pileup[i]["aln_end"] for i in held), np.int64, len(held))
    slot = np.searchsorted(np.asarray(held), who)[run]             # position of the run's read in `held`
    hit = np.flatnonzero((a != r) & (a0[slot] + end_trim <= pos) & (pos <= a1[slot] - end_trim))
    if not len(hit):
        return pileup
    in_read = (in_s[run] + step)[hit]                               # index into the concatenated trimmed reads
    base = np.zeros(len(held) + 1, np.int64)
    np.cumsum([len(x) for x in seqs], out=base[1:])
    quals = {}
    for h, p, rb, ab, ir in zip(slot[hit].tolist(), pos[hit].tolist(), r[hit].tolist(), a[hit].tolist(), in_read.tolist()):
        read = pileup[held[h]]
        q = quals.get(h)
        if q is None:
            q = quals[h] = _trimmed_quals(read)
        read["mismatches"].append((p, chr(rb).upper(), chr(ab), q[ir - int(base[h])]))
    return pileup


def findall_mismatches(read, end_trim=0):
    """localn.pyx:71-136, one read (same arguments, the read back)"""
    return findall_mismatches_pileup([read], end_trim)[0]


def worth_realn_mask(pileup, target_indel, qual_lim=23):
    """is_worth_realn (localn.pyx:139-220) for every read of a pileup as one bool array: could a realignment of the read show the
    target?  Clipped at the locus, or high-quality mismatches / indels over the target's span; not when the read ends inside
    the target's repeat and equals the reference there.  The interval tests run over arrays; the target's equivalents -- which
    the reference regenerates per read -- are made once; strings are compared only for reads that end inside the shiftable span.
    Needs read["mismatches"] (findall_mismatches_pileup)."""
    n = len(pileup)
    verdict = np.zeros(n, bool)
    if not n:
        return verdict
    col = lambda key: np.fromiter((r[key] for r in pileup), np.int64, n)
    a0, a1 = col("aln_start"), col("aln_end")
    cover = [r["covering_subread"] for r in pileup]
    covered = np.fromiter((bool(c) for c in cover), bool, n)
    at = target_indel.pos
    cs = np.fromiter((c[0] if c else at for c in cover), np.int64, n)
    ce = np.fromiter((c[1] if c else at + len(target_indel.ref) for c in cover), np.int64, n)
    open_ = covered.copy() if target_indel.is_ins else np.ones(n, bool)          # :146-149: no covering piece and an insertion -> no
    to_left, to_right = at - a0, a1 - at
    lefty = (to_left < 0) | ((to_right >= 0) & (to_left <= to_right))             # :154-159
    idx = np.flatnonzero(open_)
    edge = np.zeros(n, np.int64)
    for i in idx.tolist():                                                        # length of the CIGAR token at the read's near end
        toks = pileup[i]["cigar_list"]
        edge[i] = int((toks[0] if lefty[i] else toks[-1])[:-1])
    clipped = open_ & (edge > 2) & np.where(lefty, (cs < a0) & (a0 <= ce), (cs <= a1) & (a1 < ce))     # :163-169
    verdict[clipped] = True
    open_ &= ~clipped
    shifts = [v.pos for v in target_indel.generate_equivalents()]
    lo, hi = min(shifts), max(shifts)
    if lo < hi:                                                                   # :177-191
        for i in np.flatnonzero(open_ & lefty & (lo < a0)).tolist():
            k, r = hi - int(a0[i]), pileup[i]
            if r["read_seq"][:k] == r["ref_seq"][:k]:
                open_[i] = False
        for i in np.flatnonzero(open_ & ~lefty & (a1 <= hi)).tolist():
            k, r = int(a1[i]) - lo, pileup[i]
            if r["read_seq"][-k:] == r["ref_seq"][-k:]:
                open_[i] = False
    idx = np.flatnonzero(open_).tolist()
    who, where, q = [], [], []
    for i in idx:
        for m in pileup[i]["mismatches"]:
            who.append(i); where.append(m[0]); q.append(m[3])
    has = np.zeros(n, bool)
    if who:
        who, where, q = np.asarray(who), np.asarray(where, np.int64), np.asarray(q)
        keep = (cs[who] <= where) & (where <= ce[who]) & (q > qual_lim)            # :171-175
        who, where = who[keep], where[keep]
        has[who] = True
        far = np.iinfo(np.int64).max
        lt_most, rt_most = np.full(n, far), np.full(n, -far)
        np.minimum.at(lt_most, who, where)
        np.maximum.at(rt_most, who, where)
        w = np.flatnonzero(has)
        near_end = np.where(lefty[w], np.abs(lt_most[w] - a0[w]) < 4, np.abs(rt_most[w] - a1[w]) < 4)
        verdict[w] = near_end | covered[w]                                        # :194-208
    for i in idx:                                                                 # :210-217: an indel of the read's own over the span
        if not has[i]:
            r = pileup[i]
            lo_i, hi_i = cs[i], ce[i]
            verdict[i] = any(lo_i <= v[0] <= hi_i for v in r["I"] + r["D"])
    return verdict


def is_worth_realn(read, target_indel, qual_lim=23):
    """localn.pyx:139-220, one read"""
    return bool(worth_realn_mask([read], target_indel, qual_lim)[0])


def realn_plan(pileup, target_indel, mapq_lim):
    """the filters of is_target_by_ssw that do not depend on the alignments (localn.pyx:243-249), for a whole pileup: an int8 array
    with -1 = already a target (left as it is), 0 = not realigned (is_target becomes False), 1 = realign.  is_worth_realn is
    evaluated (worth_realn_mask) only for the reads the cheaper tests let through, as the reference's `or` chain does."""
    n = len(pileup)
    plan = np.zeros(n, np.int8)
    plan[[i for i, r in enumerate(pileup) if r["is_target"]]] = -1
    ask = [i for i, r in enumerate(pileup) if not (r["is_target"] or r["is_reference_seq"] or r["mapq"] <= mapq_lim)]
    if ask:
        worth = worth_realn_mask([pileup[i] for i in ask], target_indel)
        plan[np.asarray(ask)[worth]] = 1
    return plan


def is_target_by_ssw(read, target_indel, contig, mut_ref_lt, mut_ref_mid, mut_ref_rt, mut_aligner, ref_aligner, match_score,
                     mismatch_penalty, gap_open_penalty, gap_extension_penalty, indel_type, basequalthresh, mapq_lim,
                     mapped_base_cnt_thresh=40, allow_mismatches=10):
    """one read, the reference's signature (localn.pyx:223-291): two single alignments through the aligner objects.  The batched
    form is find_by_smith_waterman_realn below; this one exists for callers that hold a single read."""
    need = realn_plan([read], target_indel, mapq_lim)[0]
    if need < 0:
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
        pileup = findall_mismatches_pileup(list(pileup))
        plan = realn_plan(pileup, target_indel, mapq_lim)
        for k in np.flatnonzero(plan == 0).tolist():
            pileup[k]["is_target"] = False
        todo = np.flatnonzero(plan > 0).tolist()
        for k in todo:
            seq = pileup[k]["read_seq"]
            R += [seq, seq]; W += [ref_ref, mut_ref]; GO += [go, len(seq)]; GE += [ge, ge]          # localn.pyx:253-255
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
    """the read cut into left flank / indel / right flank along its alignment to the mutant contig (localn.pyx:475-539; the
    reference defines it and calls it nowhere).  `contig` needs lt_consensus_seq, indel_seq, rt_consensus_seq.  The three pieces
    are worked out as index ranges first and cut once."""
    from .pileup import get_end_pos, split
    flank, gap = len(contig.lt_consensus_seq), len(contig.indel_seq)
    lo, hi = mut_aln.read_start, mut_aln.read_end
    nothing = slice(0, 0)
    lt = mid = rt = nothing
    insertion = indel_type == "I"
    if mut_aln.reference_start <= flank:                      # the alignment starts in the left flank: the read reaches the indel from the left
        cut = lo + flank - mut_aln.reference_start
        lt = slice(lo, cut)
        if insertion:
            mid = slice(cut, min(cut + gap, hi))
        else:
            rt = slice(cut, None)
    if insertion and flank + gap <= mut_aln.reference_end:    # ... and ends in the right flank: the inserted bases are what lies before it
        back = hi - (mut_aln.reference_end - flank - gap)
        rt = slice(back, hi)
        mid = slice(max(lo, back - gap), back)
    seq, qual = read["read_seq"], read["read_qual"]
    if lt is not nothing and not insertion:                   # :507-520: where the deletion sits on the genome and what it removed
        where = get_end_pos(read["read_start"] + lo, seq[lt], read["cigar_string"])
        read["del_pos"] = where
        read["del_seq"] = split(read["ref_seq"], read["cigar_string"], where, read["aln_start"], is_for_ref=True, reverse=False)[1][:gap]
    read["lt_flank"], read["indel_seq"], read["rt_flank"] = seq[lt], seq[mid], seq[rt]
    read["lt_qual"] = qual[lt] if lt is not nothing else []
    read["rt_qual"] = qual[rt] if rt is not nothing else []
    return read
