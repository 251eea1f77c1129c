// ipx_kernels.h -- the hot path of indelPost's local realignment as CDNA4 kernels.
//
// Replaces (all citations into /root/reference/indelpost/):
//   k_dp_pass      <- sw_sse2_byte ssw.c:197-384 and sw_sse2_word ssw.c:410-586, forward and reverse,
//                     including qP_byte/qP_word profile construction (ssw.c:163-188, 386-408) and
//                     seq_reverse (ssw.c:774-785)
//   k_dp_skew      <- the same two functions where the result is the plain recurrence (16-bit passes, 8-bit upper-bound stage):
//                     a wavefront over the SSE lanes in packed half precision, no lazy-F
//   k_tb_fast<BW>, k_tb_coop <- banded_sw ssw.c:588-772 (lane per job for first bands 1..7, wave per job otherwise)
//   k_prove_overflow : nothing in the reference -- proves, from a lower bound, that the 8-bit pass (which the
//                     reference always runs first, ssw.c:842-847) overflows, so that it need not be run
//   k_plan_* / k_tb_list : the control flow of ssw_align (ssw.c:842-916), turned into device-side
//                     job lists so that a whole batch runs without a host round trip
//
// Mapping of the SSE2 algorithm onto a 64-lane wavefront ("sub-wave groups"):
//   * one SSE vector lane = one GPU lane.  The 8-bit pass (16 SSE lanes) uses one DPP row of 16
//     lanes per read, the 16-bit pass (8 SSE lanes) uses half a row.  _mm_slli_si128 becomes
//     `row_shr:1`, the horizontal max becomes 3-4 DPP butterfly steps.
//   * every 32-bit VGPR holds TWO DP cells (v_pk_*_i16/u16): the low half belongs to the even
//     alignment slot of the group, the high half to the odd slot.  A wave therefore carries 8 reads
//     in the 8-bit pass and 16 reads in the 16-bit pass.  8-bit cells are computed in 16-bit
//     containers (CDNA4 has no packed 8-bit saturating VALU); because the reference leaves the
//     8-bit pass at the first column that reaches 255-bias (ssw.c:327) no value ever saturates
//     before that column, so the un-saturated 16-bit result is bit-identical up to the exit.
//   * the striped column state (H, E, and the column saved at the best score) lives in registers:
//     segment j of the reference = register j, fully unrolled (one instantiation per segLen 0..32).
//   * the substitution profile: each striped row keeps a v_perm_b32 selector of its two read letters in
//     a register and a column's scores are looked up in an 8-byte table of the two window letters
//     (PERM; needs mat[.][N] == 0); otherwise an int8 profile [letter][j][lane][half] staged in LDS
//     (12 KB per wave at 150 bp).  Window letters are streamed four columns per dword from 4-byte
//     aligned, re-packed windows; per-column maxima stay in LDS when the window is short enough.
//   * the kernels are bound by VALU issue (every packed 16-bit op costs 4 cycles per wave-instruction per SIMD,
//     tools/ubench_issue.hip), not by latency or memory: what counts is the number of packed operations per cell
//     -- hence the half-precision three-operand maxima, the wavefront without lazy-F and the hand-scheduled stripe.
//   * lazy-F: a max-plus prefix scan over the lanes where that provably equals the reference's loop
//     (gap_open > gap_ext, no carry in signed-compare territory); otherwise the reference's
//     step-by-step loop with its data-dependent exit per read (a read that would `goto end` gets its
//     F zeroed, the wave leaves the loop when every read is out).
#pragma once
#include "ipx_simt.h"
#include "ipx_types.h"

// The library is built from several translation units (the ~400 instantiations of k_dp_pass take minutes to compile
// in one): csrc/ipx_dp_*.hip define IPX_DP_TEMPLATES_ONLY and instantiate families of k_dp_pass explicitly
// (IPX_DP_FAMILY at the end of this file); ipx_runtime.hip defines IPX_EXTERN_KERNELS, declares them extern and
// owns every other kernel.  The emulator build (tests/emu) is one translation unit with implicit instantiation.
#if defined(IPX_DP_TEMPLATES_ONLY)
#define IPX_AUX_KERNELS 0
#else
#define IPX_AUX_KERNELS 1
#endif

// ------------------------------------------------------------------------------------------------
// small helpers
// ------------------------------------------------------------------------------------------------
template <int W> IPX_DEV pk16 group_or(pk16 x)
{
    x |= xl_xor1(x);
    x |= xl_xor2(x);
    x |= xl_half_mirror(x);
    if (W >= 16) x |= xl_mirror(x);
    if (W >= 32) x |= xl_shfl(x, lane_id() ^ 16);      // (32 / 64 lanes per group: the latency tier of k_dp_skew; once per group of steps at most)
    if (W >= 64) x |= xl_shfl(x, lane_id() ^ 32);
    return x;
}
template <int W> IPX_DEV pk16 group_max(pk16 x)
{
    x = pk_max(x, xl_xor1(x));
    x = pk_max(x, xl_xor2(x));
    x = pk_max(x, xl_half_mirror(x));
    if (W >= 16) x = pk_max(x, xl_mirror(x));
    if (W >= 32) x = pk_max(x, xl_shfl(x, lane_id() ^ 16));
    if (W >= 64) x = pk_max(x, xl_shfl(x, lane_id() ^ 32));
    return x;
}
template <int W> IPX_DEV uint32_t group_umax(uint32_t x)
{
    uint32_t y;
    y = xl_xor1(x); x = x > y ? x : y;
    y = xl_xor2(x); x = x > y ? x : y;
    y = xl_half_mirror(x); x = x > y ? x : y;
    if (W >= 16) { y = xl_mirror(x); x = x > y ? x : y; }
    if (W >= 32) { y = xl_shfl(x, lane_id() ^ 16); x = x > y ? x : y; }
    if (W >= 64) { y = xl_shfl(x, lane_id() ^ 32); x = x > y ? x : y; }
    return x;
}
template <int W> IPX_DEV uint32_t group_umin(uint32_t x)
{
    uint32_t y;
    y = xl_xor1(x); x = x < y ? x : y;
    y = xl_xor2(x); x = x < y ? x : y;
    y = xl_half_mirror(x); x = x < y ? x : y;
    if (W >= 16) { y = xl_mirror(x); x = x < y ? x : y; }
    if (W >= 32) { y = xl_shfl(x, lane_id() ^ 16); x = x < y ? x : y; }
    if (W >= 64) { y = xl_shfl(x, lane_id() ^ 32); x = x < y ? x : y; }
    return x;
}
IPX_DEV uint32_t wave_umax(uint32_t x)
{
    for (int s = 1; s < 64; s <<= 1) { uint32_t y = xl_shfl(x, lane_id() ^ s); x = x > y ? x : y; }
    return x;
}
IPX_DEV int mask_len_of(const IpxBatch &b, int64_t job, int readLen)
{
    if (b.mask_len) return b.mask_len[job];
    int m = readLen / 2;
    return m < 15 ? 15 : m;                                   // sswpy.pyx:209-211
}
IPX_DEV bool rev_needed(const IpxBatch &b, unsigned score1)   // ssw.c:872
{
    return !(b.flag == 0 || (b.flag == 2 && score1 < b.filters));
}

#if IPX_AUX_KERNELS
// ------------------------------------------------------------------------------------------------
// k_pack_refs: copy every window to a 4-byte aligned, padded slot; codes outside 0..4 become 4 (N)
// ------------------------------------------------------------------------------------------------
IPX_KERNEL void k_pack_refs(const int8_t *IPX_RESTRICT refs, const int64_t *IPX_RESTRICT ref_off,
                            const int64_t *IPX_RESTRICT refp_off, int8_t *IPX_RESTRICT packed, int32_t n_refs)
{
    const int lane = lane_id();
    const int waves_per_block = IPX_BDIM / 64;
    for (int64_t r = (int64_t)IPX_BID * waves_per_block + IPX_TID / 64; r < n_refs;
         r += (int64_t)IPX_GDIM * waves_per_block) {
        const int64_t s = ref_off[r];
        const int len = (int)(ref_off[r + 1] - s);
        const int padded = ((len + 3) & ~3) + IPX_REF_PAD;
        int8_t *dst = packed + refp_off[r];
        for (int k = lane; k < padded; k += 64) {
            int8_t c = 4;
            if (k < len) { c = refs[s + k]; if ((uint8_t)c > 4) c = 4; }
            dst[k] = c;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// k_init: result records start as "nothing aligned yet" (ssw.c:831-836)
// ------------------------------------------------------------------------------------------------
// r04: block 0 also resets the small per-run tables -- CIGAR cursor, status word, traceback list counters, the dynamic passes' count and
// cursor rows (seeded with the jobs that START in the stepped pass) -- which were four fill / copy commands of their own on the stream,
// ~6 us each with nothing to overlap in a small call.
struct IpxRunReset {
    uint32_t *cursor;            // 1 word (null: nothing to reset, e.g. the launch that only prepares the static plans)
    uint32_t *status;            // 1 word, or null (the caller has cleared it)
    uint32_t *tb_n;              // IPX_TB_NCOUNTERS words
    uint32_t *dyn;               // dyn_words words: count + cursor rows of the dynamic passes
    int32_t dyn_words;
    uint32_t *exact_dst;         // IPX_NUM_CLASSES words <- exact_src (after dyn is zeroed: the row lies inside it), or null
    const uint32_t *exact_src;
};
IPX_KERNEL void k_init(IpxBatch b, IpxRunReset z)
{
    IPX_RAISE_PRIO(b);
    for (int64_t i = (int64_t)IPX_BID * IPX_BDIM + IPX_TID; i < b.n_jobs; i += (int64_t)IPX_GDIM * IPX_BDIM) {
        IpxResult r;
        r.score1 = 0; r.score2 = 0; r.ref_begin1 = -1; r.ref_end1 = 0; r.read_begin1 = -1; r.read_end1 = 0;
        r.ref_end2 = 0; r.cigar_off = 0; r.cigar_len = 0; r.flag = 0; r.mode = IPX_MODE_PENDING;
        b.res[i] = r;
    }
    if (IPX_BID == 0 && z.cursor) {
        if (IPX_TID == 0) { z.cursor[0] = 0; if (z.status) z.status[0] = 0; }
        for (int q = IPX_TID; q < 12; q += IPX_BDIM) z.tb_n[q] = 0;                 // (IPX_TB_NCOUNTERS)
        for (int q = IPX_TID; q < z.dyn_words; q += IPX_BDIM) z.dyn[q] = 0;
        IPX_SYNC();
        if (z.exact_dst) for (int q = IPX_TID; q < IPX_NUM_CLASSES; q += IPX_BDIM) z.exact_dst[q] = z.exact_src[q];
    }
}

#endif // IPX_AUX_KERNELS
// ------------------------------------------------------------------------------------------------
// planner: which jobs take part in a pass, and in which class
// ------------------------------------------------------------------------------------------------
// The pass a job's record says it takes next, as pass * 256 + class, or -1 when there is none.  class = number of
// striped segments of that pass (+ IPX_SLOW_BASE for a job with gap_open <= gap_ext, which needs a kernel with the
// reference's stepped lazy-F loop).  This ONE function is the authority: the kernels that write a record count it into
// the pass it names (plan_note), and the scatter kernel of a pass collects exactly the jobs whose record names it.
IPX_DEV int next_pass_key(const IpxBatch &b, const IpxResult &r, int readLen, bool slow)
{
    int pass, L = readLen, lanes = 16;
    switch (r.mode) {
    case IPX_MODE_PENDING:
        if (b.score_size == 1) { pass = IPX_PASS_WORD_FWD; lanes = 8; }                      // 16-bit profile only (ssw.c:853-855)
        else if (b.score_size == 2 && b.word_first_len > 0 && readLen >= b.word_first_len) { pass = IPX_PASS_WORD_FIRST; lanes = 8; }
        else if (b.plain_first) pass = (slow || readLen > b.plain_max_len) ? IPX_PASS_BYTE_EXACT : IPX_PASS_BYTE_FIRST;   // plain recurrence first, certified afterwards (gap_open <= gap_ext, or beyond the plain kernels' reach: stepped at once)
        else if (readLen < b.byte_safe_len && (!b.use_bracket || readLen < b.bracket_min_len)) pass = IPX_PASS_BYTE_EXACT;
        else pass = IPX_PASS_BYTE_FIRST;                                                     // ssw.c:842-843
        break;
    case IPX_MODE_NEED_BYTE_CHECK: pass = IPX_PASS_BYTE_CHECK; break;
    case IPX_MODE_NEED_BYTE_HIGH: pass = IPX_PASS_BYTE_HIGH; break;
    case IPX_MODE_NEED_BYTE_LOW:
    case IPX_MODE_NEED_BYTE_LOW_CMP: pass = IPX_PASS_BYTE_LOW2; break;
    case IPX_MODE_NEED_BYTE_EXACT:
    case IPX_MODE_NEED_BYTE_EXACT_P:
    case IPX_MODE_NEED_BYTE_EXACT_W: pass = IPX_PASS_BYTE_EXACT; break;
    case IPX_MODE_NEED_WORD: pass = IPX_PASS_WORD_FWD; lanes = 8; break;                     // ssw.c:844-847
    case IPX_MODE_BYTE:
        // (read_begin1 >= 0: a reverse pass has already located the begin position -- the plain reverse recurrence, certified)
        if (!rev_needed(b, r.score1) || r.read_begin1 >= 0) return -1;
        pass = IPX_PASS_BYTE_REV; L = r.read_end1 + 1;                                       // ssw.c:875-886
        break;
    case IPX_MODE_BYTE_PLAIN:
        if (!rev_needed(b, r.score1)) return -1;
        pass = IPX_PASS_BYTE_REV_PLAIN; L = r.read_end1 + 1;
        break;
    case IPX_MODE_WORD:
        if (!rev_needed(b, r.score1)) return -1;
        pass = IPX_PASS_WORD_REV; L = r.read_end1 + 1; lanes = 8;
        break;
    default: return -1;                                       // FAIL; WORD_UNPROVEN, NEED_*_PROOF (the proof kernels move those on)
    }
    if (L < 0) L = 0;
    int cls = (L + lanes - 1) / lanes;
    if (cls > IPX_MAX_SEG) cls = IPX_MAX_SEG;                 // 64 segments or more: the long-read kernel's class (k_dp_long; upload refuses reads beyond its reach)
    if (slow) cls += IPX_SLOW_BASE;
    if (b.cls_map) cls = b.cls_map[pass * IPX_NUM_CLASSES + cls];                            // (a rare class rides in a longer class's launch)
    return pass * 256 + cls;
}

// class of job i in `pass`, or -1 when the job does not take part
IPX_DEV int plan_class(const IpxBatch &b, int pass, int64_t i)
{
    const int readLen = (int)(b.read_off[i + 1] - b.read_off[i]);
    const int key = next_pass_key(b, b.res[i], readLen, b.gap_open[i] <= b.gap_ext[i]);
    return (key >= 0 && (key >> 8) == pass) ? (key & 255) : -1;
}

// Count a job into the pass its record names.  EVERY lane of the wave calls it (key -1: nothing to count); jobs of
// the wave with the same key share one atomic.
IPX_DEV void plan_note(const IpxBatch &b, int key)
{
    const int lane = lane_id();
    uint64_t todo = xl_ballot(key >= 0);
    while (todo) {                                         // one round per distinct key in the wave
        const int leader = __builtin_ffsll((long long)todo) - 1;
        const int k = (int)xl_shfl((uint32_t)key, leader);
        const uint64_t m = xl_ballot(key == k);
        if (lane == leader) (void)atomic_add_u32(&b.plan_counts[(k >> 8) * (2 * IPX_NUM_CLASSES) + (k & 255)], (uint32_t)__builtin_popcountll(m));
        todo &= ~m;
    }
}

// Slots in per-class lists for IPX_PLAN_ROUNDS jobs per lane at once: ONE atomic per distinct class for all 64 *
// IPX_PLAN_ROUNDS jobs of the wave (r02: with one wave per block and one round per atomic, a million jobs of a single class meant 15 600 atomics on
// one address -- 0.2 to 0.8 ms of pure serialisation per planner launch).  Slots go round by round, lane by lane.
#define IPX_PLAN_ROUNDS 8
IPX_DEV void wave_class_slots(uint32_t *counter, const int (&cls)[IPX_PLAN_ROUNDS], uint32_t (&slot)[IPX_PLAN_ROUNDS])
{
    const int lane = lane_id();
    uint64_t todo[IPX_PLAN_ROUNDS];
    bool any = false;
    IPX_UNROLL
    for (int r = 0; r < IPX_PLAN_ROUNDS; ++r) { todo[r] = xl_ballot(cls[r] >= 0); any = any || todo[r] != 0; slot[r] = 0; }
    while (any) {                                          // one iteration per distinct class in the wave's jobs
        int c = -1;
        IPX_UNROLL
        for (int r = IPX_PLAN_ROUNDS - 1; r >= 0; --r)
            if (todo[r]) c = (int)xl_shfl((uint32_t)cls[r], __builtin_ffsll((long long)todo[r]) - 1);
        uint64_t m[IPX_PLAN_ROUNDS];
        uint32_t total = 0;
        IPX_UNROLL
        for (int r = 0; r < IPX_PLAN_ROUNDS; ++r) { m[r] = xl_ballot(cls[r] == c); total += (uint32_t)__builtin_popcountll(m[r]); }
        uint32_t base = 0;
        if (lane == 0) base = atomic_add_u32(&counter[c], total);
        base = xl_first(base);
        any = false;
        IPX_UNROLL
        for (int r = 0; r < IPX_PLAN_ROUNDS; ++r) {
            if (cls[r] == c) slot[r] = base + (uint32_t)__builtin_popcountll(m[r] & ((1ull << lane) - 1ull));
            base += (uint32_t)__builtin_popcountll(m[r]);
            todo[r] &= ~m[r];
            any = any || todo[r] != 0;
        }
    }
}

// Planner kernels are ONE wave per block and a handful of registers: on a GPU busy with other streams' DP blocks (which
// fill the register files) a one-wave block is resident as soon as any single DP wave retires, where a 512-thread block
// had to wait for room on all four SIMDs of one CU at once (r01: k_plan_count was the top row of the 4-stream profile).
#define IPX_PLAN_BLOCK 64
#define IPX_PLAN_LDS (8 * (IPX_NUM_CLASSES + 2))

#if IPX_AUX_KERNELS
// count the jobs of `pass` from their records: used for the static passes (once per resident batch) and for a
// pass whose jobs come straight from k_init (16-bit only profiles)
IPX_KERNEL void k_plan_count(IpxBatch b, int pass)
{
    IPX_RAISE_PRIO(b);
    const int64_t chunk = (int64_t)IPX_BDIM * IPX_PLAN_ROUNDS, stride = (int64_t)IPX_GDIM * chunk;
    const int64_t iters = (b.n_jobs + stride - 1) / stride;
    uint32_t *row = b.plan_counts + (size_t)pass * (2 * IPX_NUM_CLASSES);
    for (int64_t q = 0; q < iters; ++q) {                  // every lane runs every iteration (wave-wide ballots)
        int cls[IPX_PLAN_ROUNDS];
        uint32_t slot[IPX_PLAN_ROUNDS];
        IPX_UNROLL
        for (int r = 0; r < IPX_PLAN_ROUNDS; ++r) {
            const int64_t i = q * stride + (int64_t)IPX_BID * chunk + (int64_t)r * IPX_BDIM + IPX_TID;
            cls[r] = i < b.n_jobs ? plan_class(b, pass, i) : -1;
        }
        wave_class_slots(row, cls, slot);                  // (the slots are not needed here, the sums are)
    }
}

// Scatter the jobs of `pass` into perm, grouped by class.  Every block first turns the class counts into slot and tile
// offsets for itself (exclusive scans over the IPX_NUM_CLASSES classes by one wave; NA = alignments per tile); block 0
// also publishes them for the DP kernels of the pass.
IPX_KERNEL void k_plan_scatter(IpxBatch b, IpxPlan p, int pass, int na)
{
    IPX_RAISE_PRIO(b);
    uint32_t *cls_off = (uint32_t *)IPX_LDS_BASE, *tile_off = cls_off + IPX_NUM_CLASSES + 1;
    const int lane = lane_id();
    {
        uint32_t c[3], s = 0, t = 0;
        IPX_UNROLL
        for (int k = 0; k < 3; ++k) {
            const int cl = 3 * lane + k;
            c[k] = cl < IPX_NUM_CLASSES ? p.count[cl] : 0u;
            s += c[k];
            t += (c[k] + (uint32_t)na - 1u) / (uint32_t)na;
        }
        uint32_t si = s, ti = t;                           // inclusive scans over the lanes
        for (int d = 1; d < 64; d <<= 1) {
            const uint32_t ys = xl_shfl(si, lane >= d ? lane - d : lane), yt = xl_shfl(ti, lane >= d ? lane - d : lane);
            if (lane >= d) { si += ys; ti += yt; }
        }
        uint32_t so = si - s, to = ti - t;
        IPX_UNROLL
        for (int k = 0; k < 3; ++k) {
            const int cl = 3 * lane + k;
            if (cl < IPX_NUM_CLASSES) {
                cls_off[cl] = so; tile_off[cl] = to;
                if (IPX_BID == 0) { p.cls_off[cl] = so; p.tile_off[cl] = to; if (p.stats) p.stats[cl] = (c[k] + (uint32_t)na - 1u) / (uint32_t)na; }
            }
            so += c[k];
            to += (c[k] + (uint32_t)na - 1u) / (uint32_t)na;
        }
        if (lane == 63) {
            cls_off[IPX_NUM_CLASSES] = si; tile_off[IPX_NUM_CLASSES] = ti;
            if (IPX_BID == 0) { p.cls_off[IPX_NUM_CLASSES] = si; p.tile_off[IPX_NUM_CLASSES] = ti; if (p.stats) p.stats[IPX_NUM_CLASSES] = ti; }
        }
    }
    IPX_SYNC();
    const int64_t chunk = (int64_t)IPX_BDIM * IPX_PLAN_ROUNDS, stride = (int64_t)IPX_GDIM * chunk;
    const int64_t iters = (b.n_jobs + stride - 1) / stride;
    for (int64_t q = 0; q < iters; ++q) {
        int cls[IPX_PLAN_ROUNDS];
        uint32_t pos[IPX_PLAN_ROUNDS];
        IPX_UNROLL
        for (int r = 0; r < IPX_PLAN_ROUNDS; ++r) {
            const int64_t i = q * stride + (int64_t)IPX_BID * chunk + (int64_t)r * IPX_BDIM + IPX_TID;
            cls[r] = i < b.n_jobs ? plan_class(b, pass, i) : -1;
        }
        wave_class_slots(p.cursor, cls, pos);
        IPX_UNROLL
        for (int r = 0; r < IPX_PLAN_ROUNDS; ++r) {
            const int64_t i = q * stride + (int64_t)IPX_BID * chunk + (int64_t)r * IPX_BDIM + IPX_TID;
            if (cls[r] >= 0) {
                // (a job the counting side missed would run past its class: refuse it instead)
                if (pos[r] >= cls_off[cls[r] + 1] - cls_off[cls[r]]) atomic_or_u32(b.status, IPX_STATUS_INTERNAL);
                else p.perm[cls_off[cls[r]] + pos[r]] = (uint32_t)i;
            }
        }
    }
}

#endif // IPX_AUX_KERNELS
#if !defined(IPX_CPU_EMU) && !defined(IPX_NO_STRIPE_ASM)
#define IPX_STRIPE_ASM 1
#else
#define IPX_STRIPE_ASM 0
#endif
// ------------------------------------------------------------------------------------------------
// One column of the striped recurrence in packed half precision (k_dp_pass F16, k_dp_skew): H, E updated in place, vF = F
// entering segment 0 on entry and F leaving the last segment on exit, cmx = maximum of the new H with the cmx passed in.
// go / ge hold -gapO / -gapE as halves, tab0 / tab1 the score tables of the two window letters, SEL the selectors.
// ------------------------------------------------------------------------------------------------
// MID > 0: the lane's segments are TWO of the reference's lanes (k_dp_pass VL2): F starts again from 0 at segment MID, and the F
// that left segment MID-1 is returned in *fmid.
template <int SMAX, int MID = 0>
IPX_DEV void dp_stripe_f16(pk16 (&H)[SMAX > 0 ? SMAX : 1], pk16 (&E)[SMAX > 0 ? SMAX : 1], const pk16 (&SEL)[SMAX > 0 ? SMAX : 1],
                           pk16 &vF, pk16 &cmx, pk16 vH, const uint32_t tab0, const uint32_t tab1, const pk16 go, const pk16 ge,
                           pk16 *fmid = nullptr)
{
#if !IPX_STRIPE_ASM
    // plain form (emulator)
    IPX_UNROLL
    for (int j = 0; j < SMAX; ++j) {
        if (MID > 0 && j == MID) { *fmid = vF; vF = 0; }
        const pk16 h = pkh_max3(pkh_add(vH, pk_perm(tab1, tab0, SEL[j])), E[j], vF);
        cmx = pkh_max(cmx, h);
        vH = H[j];
        H[j] = h;
        const pk16 tt = pkh_add(h, go);
        E[j] = pkh_max(pkh_add(E[j], ge), tt);                     // (no floor: see F16 at k_dp_pass)
        vF = pkh_max3(pkh_add(vF, ge), tt, 0u);
    }
#else
    // half-precision form, hand-scheduled like the integer stripe below: every operand is at least two
    // instructions away from the packed operation that produced it, H is updated in place.
    //   qj = diag + score of the segment about to be finished, em = its E - gapE   (prepared one block ahead)
    //   p1 = score of the next segment                                             (prepared one block ahead)
    if (SMAX > 0) {
        pk16 qj, em, p1 = 0, q2, emn, vFm, tt;
        qj = pkh_add(vH, pk_perm(tab1, tab0, SEL[0]));
        em = pkh_add(E[0], ge);
        if (SMAX > 1) p1 = pk_perm(tab1, tab0, SEL[SMAX > 1 ? 1 : 0]);
        asm volatile("s_nop 0");       // (the compiler does not see the packed reads inside the blocks: keep its last write a state away)
        // operands: 0 H[j], 1 E[j], 2 F, 3 p1, 4 column maximum | 5 q2, 6 em', 7 F', 8 tt (temporaries / next block's inputs) |
        //           9 qj, 10 em, 11 E[j+1], 12 selector of segment j+2, 13/14 score tables, 15 -gapO, 16 -gapE, 17 H[j-1]
#define IPX_H_OPS                                                                                                              \
            : "+v"(H[j]), "+v"(E[j]), "+v"(vF), "+v"(p1), "+v"(cmx), "=&v"(q2), "=&v"(emn), "=&v"(vFm), "=&v"(tt)          \
            : "v"(qj), "v"(em), "v"(E[j + 1 < SMAX ? j + 1 : 0]), "v"(SEL[j + 2 < SMAX ? j + 2 : 0]), "v"(tab0), "v"(tab1),  \
              "v"(go), "v"(ge), "v"(H[j > 0 ? j - 1 : 0])
#define IPX_H_CMX3 "v_pk_maximum3_f16 %4, %4, %17, %0\n\t"     /* column maximum: two segments at once (odd j) */
#define IPX_H_HEAD                                                                                      \
            "v_pk_add_f16 %5, %0, %3\n\t"               /* q2 = H[j](old) + score[j+1]          */ \
            "v_pk_add_f16 %7, %2, %16\n\t"              /* F' = F - gapE                        */ \
            "v_pk_maximum3_f16 %0, %9, %1, %2\n\t"      /* H[j] = max3(qj, E[j], F)             */
#define IPX_H_TAIL                                                                                      \
            "v_pk_maximum3_f16 %2, %7, %8, 0\n\t"       /* F = max3(F', tt, 0)                  */ \
            "v_pk_max_f16 %1, %10, %8"                   /* E[j] = max(em, tt)                   */
        IPX_UNROLL
        for (int j = 0; j < SMAX; ++j) {
            if (MID > 0 && j == MID) { *fmid = vF; vF = 0; asm volatile("s_nop 0"); }
            if (j + 2 < SMAX) {
                if (j & 1)
                    asm volatile(IPX_H_HEAD
                                 "v_perm_b32 %3, %14, %13, %12\n\t"          /* score[j+2], already a half */
                                 "v_pk_add_f16 %8, %0, %15\n\t"              /* tt = H[j] - gapO           */
                                 "v_pk_add_f16 %6, %11, %16\n\t"             /* em' = E[j+1] - gapE        */
                                 IPX_H_CMX3 IPX_H_TAIL IPX_H_OPS);
                else
                    asm volatile(IPX_H_HEAD
                                 "v_perm_b32 %3, %14, %13, %12\n\t"
                                 "v_pk_add_f16 %8, %0, %15\n\t"
                                 "v_pk_add_f16 %6, %11, %16\n\t"
                                 IPX_H_TAIL IPX_H_OPS);
            } else if (j + 1 < SMAX) {
                if (j & 1)
                    asm volatile(IPX_H_HEAD
                                 "v_pk_add_f16 %6, %11, %16\n\t"
                                 "v_pk_add_f16 %8, %0, %15\n\t"
                                 IPX_H_CMX3 IPX_H_TAIL IPX_H_OPS);
                else
                    asm volatile(IPX_H_HEAD
                                 "v_pk_add_f16 %6, %11, %16\n\t"
                                 "v_pk_add_f16 %8, %0, %15\n\t"
                                 "s_nop 0\n\t"
                                 IPX_H_TAIL IPX_H_OPS);
            } else {
                if (j & 1)
                    asm volatile("v_pk_maximum3_f16 %0, %9, %1, %2\n\t"
                                 "v_pk_add_f16 %7, %2, %16\n\t"
                                 "v_pk_add_f16 %8, %0, %15\n\t"
                                 IPX_H_CMX3 IPX_H_TAIL IPX_H_OPS);
                else
                    asm volatile("v_pk_maximum3_f16 %0, %9, %1, %2\n\t"
                                 "v_pk_add_f16 %7, %2, %16\n\t"
                                 "v_pk_add_f16 %8, %0, %15\n\t"
                                 "v_pk_max_f16 %4, %4, %0\n\t"
                                 IPX_H_TAIL IPX_H_OPS);
            }
            qj = q2; em = emn;
        }
#undef IPX_H_OPS
#undef IPX_H_CMX3
#undef IPX_H_HEAD
#undef IPX_H_TAIL
    }
#endif
}

// ------------------------------------------------------------------------------------------------
// k_dp_pass: one striped Smith-Waterman pass over a tile of 128/W reads per wavefront
//   W    = SSE lanes of the reference pass: 16 (8-bit semantics) or 8 (16-bit semantics)
//   SMAX = segLen capacity held in registers.  EXACT: the tile's segLen S == SMAX is a compile-time
//          constant, so the segment loop is straight-line code and the profile reads pipeline;
//          !EXACT (long reads): S <= SMAX is read per tile and every segment step is branch-guarded
//   REV  = reverse pass (reversed read prefix vs window prefix walked right to left, ssw.c:875-886)
// Block = one wavefront (64 threads); grid-stride over the tiles of classes [cls_lo, cls_hi].
// Dynamic LDS: !PERM profile 640*SMAX B | matrix 32 B;  PERM table 64 B | matrix 64 B | column maxima 4*G*maxcols B when they fit.
// Global: column maxima, 4*G*maxcols B per block, otherwise (forward only)
// ------------------------------------------------------------------------------------------------
//   STAGE (8-bit forward pass only; IPX_STAGE_EXACT everywhere else).  The reference's lazy-F loop leaves on a SIGNED
//          byte compare (ssw.c:311), so in a column where some carry of a read is >= 128+gapE it may stop before
//          every carry has been passed on; what it computes there lies between "no lazy-F at all" and "every carry
//          passed on to the end", both of which are closed forms:
//          LOW   in such a column that read's whole lazy-F step is skipped.  H then is a lower bound of the exact pass
//                (lazy-F only ever raises H, and the recurrences are monotone), so an overflow seen here is certain;
//                a read that skipped no column is exact.
//          HIGH  every carry is passed on (the closed form, whatever its size): an upper bound, cell by cell.
//          Every output of the pass -- best score, its first column, the smallest row holding it there, the best
//          column maximum outside the mask and its first column -- is monotone in the matrix in the sense that if
//          LOW and HIGH agree on it the exact pass, squeezed between them, agrees too.  So a read whose LOW and HIGH
//          outputs are all equal is CERTIFIED without stepping; the others go to the EXACT stage
//          (IPX_MODE_NEED_BYTE_EXACT), which steps through the reference's loop.
//   PERM = the query profile is not staged in LDS: every striped row keeps a v_perm_b32 selector of its
//          two read letters in a register, and a column's scores come from an 8-byte table
//          {mat[c0][A..T], mat[c1][A..T]} of the two window letters: one v_perm_b32 puts the selected
//          bytes in the high byte of each half and one packed arithmetic shift sign-extends them.
//          Read letter N and padding rows select the constant 0, so this needs mat[c][N] == 0 for every
//          c (true for indelPost's matrix, sswpy.pyx:306-336).  No LDS traffic in the column loop.
//   F16  = (16-bit passes, PERM, exact segLen) the same recurrence in packed HALF PRECISION.  A packed 16-bit VALU operation
//          costs four cycles per wavefront whatever it computes (tools/ubench_issue.hip), so only the NUMBER of operations
//          per cell counts, and gfx950's v_pk_maximum3_f16 takes three operands: H = max3(diag + score, E, F) is one
//          operation instead of two, F = max3(F - gapE, H - gapO, 0) carries the floor the integer form gets from its
//          saturating subtraction, two segments' H enter the column maximum at once, and v_perm_b32 delivers the score
//          AS a half (high byte from the table, low byte 0) so the sign-extending shift goes too: 8.5 operations per
//          segment instead of 11.  Integers up to 2048 and their sums are exact in half precision, so the results are
//          bit-identical as long as no score can exceed 2047 (host: IpxBatch::f16_max_len, and every matrix entry a half
//          with a zero low byte: 0..8, 10, 12, ...; device-guarded).  E is left without its floor at 0: max(E, 0) equals
//          the reference's E by induction, and F >= 0 keeps H >= 0.  Non-negative halves order like integers, so the
//          bookkeeping around the stripe (column maxima, best score, saved column) works on the bit patterns unchanged.
// With the profile out of LDS the register file alone sets the occupancy.  The request below only nudges
// the kernels that sit just above a waves-per-SIMD step (the reverse pass at segLen 17-19: 182 -> 170
// VGPRs, 2 -> 3 waves).  Pushing harder (4 waves at segLen 19) was measured: +2 % speed for register
// spills worth 3x the tile's input in memory traffic -- the DP is issue-bound, not latency-bound.
// (state = H, E, Hmax and the selectors = 4 registers per segment.)
IPX_HD constexpr int ipx_dp_perm_waves(int smax)
{
    const int w = 512 / (4 * smax + 90);
    return w < 1 ? 1 : w;
}
// (the body is a function of the block's rank among the blocks working on these classes: the kernel below passes its block id,
//  k_dp_pass_tier the rank it computes per class)
template <int W, int SMAX, bool REV, bool EXACT, int STAGE, bool PERM, bool F16, bool VL2>
IPX_DEV void dp_pass_body(const IpxBatch &b, const IpxPlan &p, int cls_lo, int cls_hi, int maxcols, int pass, uint64_t skip_fast, uint64_t skip_slow,
                          const uint32_t rank, const uint32_t nrank)
{
    constexpr bool LOW = STAGE == IPX_STAGE_LOW, HIGH = STAGE == IPX_STAGE_HIGH;
    static_assert(STAGE == IPX_STAGE_EXACT || ((W == 16 || VL2) && !REV), "the bracket stages exist for the 8-bit forward pass");
    static_assert(!VL2 || (W == 8 && F16 && STAGE == IPX_STAGE_LOW && SMAX % 2 == 0), "two reference lanes per GPU lane: the half-precision lower-bound stage");
    static_assert(!F16 || (PERM && EXACT && ((W == 8 && STAGE == IPX_STAGE_EXACT) || ((W == 16 || VL2) && STAGE == IPX_STAGE_LOW))),
                  "the half-precision form exists for the exact-segLen selector-profile kernels without a stepped loop: 16-bit passes, 8-bit lower-bound stage");
    constexpr int SA = SMAX > 0 ? SMAX : 1;            // array extent (segLen 0 = empty read)
    constexpr int G = 64 / W;
    constexpr int NA = 2 * G;
    constexpr bool BYTE = (W == 16) || VL2;            // 8-bit semantics (VL2: in the 8-lane layout, see below)
    // the reference's step-by-step lazy-F loop is needed for reads with gap_open <= gap_ext and, in the exact
    // 8-bit passes, for carries in signed-compare territory.  The selector-profile kernels of the 16-bit passes
    // and of the lower-bound stage leave it out (the host launches them only when no job has gap_open <= gap_ext):
    // without that loop the H registers of a column are defined once, in place, and no copies are needed.
    constexpr bool STEP = !(PERM && (W == 8 || LOW || HIGH));
    const int lane = lane_id();
    const int g = lane / W, l = lane % W;
    unsigned char *lds = IPX_LDS_BASE;
    int8_t *prof = (int8_t *)lds;                                  // [5][S][64][2]      (!PERM)
    const uint32_t *tab8 = (const uint32_t *)lds;                  // [5 window letters][4 int8]   (PERM)
    int8_t *matl = (int8_t *)(lds + (PERM ? 64 : 640 * SA));
    // this block's column maxima: in LDS when the launch reserved room for them (windows short enough to
    // keep the occupancy), else in its region of the global scratch
    const bool mc_lds = PERM && !REV && (pass & IPX_PASS_MC_LDS) != 0;
    pass &= 0xFF;
    uint32_t *maxcol = mc_lds ? (uint32_t *)(lds + 128) : b.maxcol_scratch + (size_t)IPX_BID * (size_t)(G * maxcols);

    if (lane < 25) matl[lane] = b.mat[lane];
    if (PERM && lane < 20) {
        const int v = b.mat[(lane >> 2) * 5 + (lane & 3)];
        ((int8_t *)lds)[lane] = F16 ? (int8_t)(ipx_f16_from_int(v) >> 8) : (int8_t)v;          // F16: the score as a half, high byte
    }
    IPX_SYNC();

    // Block b takes the b-th, (b + gridDim)-th, ... tile of the classes this launch OWNS.  An exact-segLen launch owns its one
    // class; the sweep launch owns what the exact launches left over (skip_fast / skip_slow) and walks the classes once,
    // counting only its own tiles -- a sweep grid is sized for those few tiles, so striding over ALL tiles of the pass and
    // skipping the foreign ones (as an earlier version did) cost each block thousands of class look-ups.
    int own_cls = cls_lo;                                             // (sweep) class reached by the walk
    uint32_t own_base = 0;                                            // (sweep) owned tiles in the classes before own_cls
    for (uint32_t want = rank;; want += nrank) {
        // ---- locate the tile: class (= segLen), first slot in perm, number of reads -------------
        int cls;
        uint32_t tile;
        if (EXACT) {
            cls = cls_lo;
            tile = p.tile_off[cls] + want;
            if (tile >= p.tile_off[cls + 1]) break;
        } else {
            for (; own_cls <= cls_hi; ++own_cls) {
                const int sg = own_cls >= IPX_SLOW_BASE ? own_cls - IPX_SLOW_BASE : own_cls;
                const bool foreign = sg == IPX_MAX_SEG ||                                                             // 64 segments or more: k_dp_long's
                                     (((own_cls >= IPX_SLOW_BASE ? skip_slow : skip_fast) >> sg) & 1ull);             // owned by an exact-segLen launch
                const uint32_t n = foreign ? 0u : p.tile_off[own_cls + 1] - p.tile_off[own_cls];
                if (want < own_base + n) break;
                own_base += n;
            }
            if (own_cls > cls_hi) break;
            cls = own_cls;
            tile = p.tile_off[cls] + (want - own_base);
        }
        const int seg = cls >= IPX_SLOW_BASE ? cls - IPX_SLOW_BASE : cls;      // class = segLen (+ IPX_SLOW_BASE: gap_open <= gap_ext)
        if (!EXACT && seg > SMAX) { if (lane == 0) atomic_or_u32(b.status, IPX_STATUS_INTERNAL); continue; }   // sweep kernel sized too small (host error)
        const int S = EXACT ? SMAX : (int)xl_first((uint32_t)seg);
        const uint32_t first = p.cls_off[cls] + (tile - p.tile_off[cls]) * NA;
        const uint32_t avail = p.cls_off[cls + 1] - first;
        const int cnt = avail < (uint32_t)NA ? (int)avail : NA;

        // ---- per-slot parameters (index 0 = low half, 1 = high half of every packed register) ----
        int64_t job[2];
        int L[2], ncol[2], tb[2], idx0[2], kmax[2], score1[2], rend1[2];
        const int8_t *rd[2];
        const uint32_t *refw[2];
        int gO[2], gE[2];
        IPX_UNROLL
        for (int h = 0; h < 2; ++h) {
            const int slot = 2 * g + h;
            job[h] = -1; L[h] = 0; ncol[h] = 0; tb[h] = 0; idx0[h] = 3; kmax[h] = 0; score1[h] = 0; rend1[h] = -1;
            rd[h] = b.reads; refw[h] = (const uint32_t *)b.refs_packed; gO[h] = 0; gE[h] = 0;
            if (slot < cnt) {
                const int64_t jb = (int64_t)p.perm[first + slot];
                const int rid = b.ref_id[jb];
                const int refLen = b.ref_len[rid];
                job[h] = jb;
                rd[h] = b.reads + b.read_off[jb];
                refw[h] = (const uint32_t *)(b.refs_packed + b.refp_off[rid]);
                kmax[h] = ((refLen + 3) >> 2) + 1;
                gO[h] = b.gap_open[jb];
                gE[h] = b.gap_ext[jb];
                if (!REV) {
                    L[h] = (int)(b.read_off[jb + 1] - b.read_off[jb]);
                    ncol[h] = refLen;
                } else {
                    const IpxResult r = b.res[jb];
                    L[h] = r.read_end1 + 1; if (L[h] < 0) L[h] = 0;
                    ncol[h] = r.ref_end1 + 1; if (ncol[h] < 0) ncol[h] = 0;
                    score1[h] = r.score1;
                    rend1[h] = r.read_end1;
                    if (ncol[h] > 0) { idx0[h] = r.ref_end1 | 3; tb[h] = idx0[h] - r.ref_end1; }
                }
            }
        }
        // F16: -gapO and -gapE as halves (the stripe adds them), the score to stop at as a half
        const pk16 go = F16 ? pk_make((int)ipx_f16_from_int(-gO[0]), (int)ipx_f16_from_int(-gO[1])) : pk_make(gO[0], gO[1]);
        const pk16 ge = F16 ? pk_make((int)ipx_f16_from_int(-gE[0]), (int)ipx_f16_from_int(-gE[1])) : pk_make(gE[0], gE[1]);
        const pk16 term = F16 ? pk_make((int)ipx_f16_from_uint((uint32_t)score1[0]), (int)ipx_f16_from_uint((uint32_t)score1[1])) : pk_make(score1[0], score1[1]);
        const pk16 capm1 = pk_splat(F16 ? (int)ipx_f16_from_uint((uint32_t)(255 - b.bias - 1)) : 255 - b.bias - 1);   // overflow when colmax >= 255-bias (ssw.c:327)
        const pk16 capv = pk_splat(F16 ? (int)ipx_f16_from_uint((uint32_t)(255 - b.bias)) : 255 - b.bias);         // (halves: compared and stored as bit patterns)
        // closed-form lazy-F (below) applies to a read when gap_open > gap_ext
#ifdef IPX_DEBUG_NOFAST
        const pk16 fast_static = 0;
#else
        const pk16 fast_static = (gO[0] > gE[0] ? 0x0000FFFFu : 0u) | (gO[1] > gE[1] ? 0xFFFF0000u : 0u);
#endif
        if (!STEP) {                                            // this variant has no step loop: refuse what would need it
            bool bad = (job[0] >= 0 && gO[0] <= gE[0]) || (job[1] >= 0 && gO[1] <= gE[1]);
            if (F16) bad = bad || L[0] > b.f16_max_len || L[1] > b.f16_max_len;     // ... or could leave the exact range of a half
            if (xl_any(bad) && lane == 0) atomic_or_u32(b.status, IPX_STATUS_INTERNAL);
        }
        const pk16 bigthr = F16 ? pk_make((int)ipx_f16_from_uint((uint32_t)(127 + gE[0])), (int)ipx_f16_from_uint((uint32_t)(127 + gE[1])))
                                : pk_make(127 + gE[0], 127 + gE[1]);  // F carry > 127+gapE: signed-byte compare territory
        // (exact 8-bit stage) a cell can see a cut only with its largest carry below cutw and its main-loop H below cuth: see the lazy-F step
        const pk16 cutw = pk_make(128 + 2 * gO[0] - gE[0], 128 + 2 * gO[1] - gE[1]), cuth = pk_make(128 + gO[0], 128 + gO[1]);
        pk16 D1, D2, D4, D8;                                    // decay of a carry across 1/2/4/8 whole segments
        {
            int d[2][4];
            IPX_UNROLL
            for (int h = 0; h < 2; ++h)
                for (int q = 0; q < 4; ++q) {
                    int v = (S * gE[h]) << q;
                    d[h][q] = v > 65535 ? 65535 : v;
                    if (F16) d[h][q] = (int)ipx_f16_from_int(-(v > 2047 ? 2047 : v));      // (a carry is <= 2047: minus 2047 ends it)
                }
            D1 = pk_make(d[0][0], d[1][0]); D2 = pk_make(d[0][1], d[1][1]);
            D4 = pk_make(d[0][2], d[1][2]); D8 = pk_make(d[0][3], d[1][3]);
        }
        pk16 Dh = 0;                                            // VL2: decay across ONE reference lane = half a GPU lane's segments
        if (VL2) {
            int v[2];
            IPX_UNROLL
            for (int h = 0; h < 2; ++h) { v[h] = (S / 2) * gE[h]; v[h] = (int)ipx_f16_from_int(-(v[h] > 2047 ? 2047 : v[h])); }
            Dh = pk_make(v[0], v[1]);
        }

        // ---- stage the tile's query profile in LDS: int8 [5 letters][S][64 lanes][2 halves] ------------
        pk16 SEL[PERM ? SA : 1];                                   // PERM: per striped row, byte selectors of the two read letters
        if (PERM) {
            // all letter loads of the tile are issued back to back (clamped addresses, no branches) ...
            int raw[2][SA];
            IPX_UNROLL
            for (int h = 0; h < 2; ++h) {
                IPX_UNROLL
                for (int j = 0; j < SMAX; ++j) {
                    if (j < S) {
                        const int r = j + l * S;                   // striped row (ssw.c:178-185)
                        int idx = r < L[h] ? r : L[h] - 1;
                        if (REV) idx = L[h] - 1 - idx;             // reverse pass: seq_reverse (ssw.c:774-785)
                        raw[h][j] = load_stream_i8(L[h] > 0 ? rd[h] + idx : (const int8_t *)b.read_off);
                    }
                }
            }
            // ... then turned into selectors
            IPX_UNROLL
            for (int j = 0; j < SMAX; ++j) {
                if (j < S) {
                    const int r = j + l * S;
                    uint32_t sel = 0;
                    IPX_UNROLL
                    for (int h = 0; h < 2; ++h) {
                        const unsigned base = (unsigned)raw[h][j];
                        uint32_t sh = 0x0c0cu;                     // padding row / letter N: constant 0
                        if (r < L[h] && base < 4u) sh = 0x000cu | ((base + 4u * h) << 8);   // high byte <- table byte 4*h + base
                        sel |= sh << (16 * h);
                    }
                    SEL[j] = sel;
                }
            }
        } else {
        IPX_SYNC();   // previous tile's finalisation reads are done
        IPX_UNROLL
        for (int h = 0; h < 2; ++h) {
            for (int j = 0; j < S; ++j) {
                const int r = j + l * S;                           // striped row (ssw.c:178-185)
                int base = -1;
                if (r < L[h]) {
                    base = load_stream_i8(REV ? rd[h] + (L[h] - 1 - r) : rd[h] + r);   // reverse pass: seq_reverse (ssw.c:774-785)
                    if ((unsigned)base > 4u) base = 4;
                }
                for (int c = 0; c < 5; ++c)
                    prof[((c * S + j) * 64 + lane) * 2 + h] = base >= 0 ? matl[c * 5 + base] : (int8_t)0;
            }
        }
        IPX_SYNC();
        }

        // ---- DP state -----------------------------------------------------------------------------
        pk16 H[SA], E[SA], HM[SA];
        IPX_UNROLL
        for (int j = 0; j < SMAX; ++j) { H[j] = 0; E[j] = 0; HM[j] = 0; }
        pk16 Hlast = 0, best = 0, done = 0, ovf = 0, dropped = 0;
        pk16 endref = BYTE ? 0xFFFFFFFFu : 0u;                      // ssw.c:220 / 427
        const pk16 icol0 = REV ? pk_make(idx0[0], idx0[1]) : 0u;

        const int T = (int)wave_umax((uint32_t)((tb[0] + ncol[0]) > (tb[1] + ncol[1]) ? (tb[0] + ncol[0]) : (tb[1] + ncol[1])));
        uint32_t cur[2], nxt[2];
        IPX_UNROLL
        for (int h = 0; h < 2; ++h) {
            const int k0 = REV ? (idx0[h] >> 2) : 0;
            int k1 = REV ? (idx0[h] >> 2) - 1 : 1;
            if (k1 < 0) k1 = 0;
            if (k1 > kmax[h]) k1 = kmax[h];
            cur[h] = load_global_u32(refw[h] + (k0 > kmax[h] ? kmax[h] : k0));
            nxt[h] = load_global_u32(refw[h] + k1);
        }

        // Columns go in groups of four = one dword of window letters.  All global-memory traffic of the
        // column loop sits at the edges of a group: at its top the previous group's four column maxima are stored, at its
        // end the dword fetched one group ahead is consumed and the next one requested -- so the only wait on
        // global memory is for operations issued a whole group (thousands of cycles) earlier.
        bool stop = false;
        int tdone = -1;                                             // last column processed
        pk16 cm4 = 0;                                               // lane l<4: maximum of column (group base + l)
        for (int t0 = 0; t0 < T && !stop; t0 += 4) {
          if (t0 > 0 && !REV && l < 4) { if (mc_lds) maxcol[(t0 - 4 + l) * G + g] = cm4; else store_global_u32(maxcol + ((t0 - 4 + l) * G + g), cm4); }
          const int tn = t0 + 4 < T ? t0 + 4 : T;
          IPX_NOUNROLL
          for (int t = t0; t < tn; ++t) {
            tdone = t;
            const uint32_t sh = (uint32_t)(REV ? 3 - (t & 3) : (t & 3)) * 8u;
            uint32_t c[2];
            pk16 act = 0;
            IPX_UNROLL
            for (int h = 0; h < 2; ++h) {
                const bool valid = (uint32_t)(t - tb[h]) < (uint32_t)ncol[h];
                c[h] = ubfe(cur[h], sh, 8);                           // always a valid letter (windows are sanitised and padded)
                if (valid) act |= (h ? 0xFFFF0000u : 0x0000FFFFu);
            }
            act &= ~done;
            const pk16 icol = REV ? pk_sub(icol0, pk_splat(t)) : pk_splat(t);
            // a read whose first column is t (reverse pass: up to 3 lead-in columns while the other reads of
            // the tile start) begins from an all-zero state, whatever the lead-in columns left behind
            if (REV && t > 0 && t <= 3) {
                const pk16 fresh = (tb[0] == t ? 0x0000FFFFu : 0u) | (tb[1] == t ? 0xFFFF0000u : 0u);
                if (xl_any(fresh != 0)) {
                    IPX_UNROLL
                    for (int j = 0; j < SMAX; ++j)
                        if (j < S) { H[j] &= ~fresh; E[j] &= ~fresh; }
                    Hlast &= ~fresh;
                }
            }
            const int8_t *pa0 = prof + ((int)c[0] * S * 64 + lane) * 2;
            const int8_t *pa1 = prof + ((int)c[1] * S * 64 + lane) * 2 + 1;
            uint32_t tab0 = 0, tab1 = 0;
            if (PERM) { tab0 = tab8[c[0]]; tab1 = tab8[c[1]]; }

            // -- striped inner loop (ssw.c:274-299 / 480-504) -----------------------------------------
            pk16 vH = xl_row_shr1(Hlast);                         // _mm_slli_si128(pvHStore[segLen-1], 1|2)
            if (W == 8 && l == 0) vH = 0;
            pk16 vF = 0, cmx = 0;
#define IPX_DP_STRIPE(PP)                                                                                    \
            IPX_UNROLL                                                                                       \
            for (int j = 0; j < SMAX; ++j) {                                                                 \
                if (j < S) {                                                                                 \
                    const pk16 pp = (PP);                                                                    \
                    pk16 h = pk_add_sat(vH, pp);                                                             \
                    pk16 e = E[j];                                                                           \
                    h = pk_max(h, e);                                                                        \
                    h = pk_max(h, vF);                                                                       \
                    cmx = pk_max(cmx, h);                                                                    \
                    vH = H[j];                                                                               \
                    H[j] = h;                                                                                \
                    if (j == S - 1) Hlast = h;                                                               \
                    const pk16 tt = pk_subus(h, go);                                                         \
                    e = pk_subus(e, ge);                                                                     \
                    E[j] = pk_max(e, tt);                                                                    \
                    vF = pk_subus(vF, ge);                                                                   \
                    vF = pk_max(vF, tt);                                                                     \
                }                                                                                            \
            }
            pk16 fmid = 0;                                          // VL2: F that left the lane's first reference lane
            if constexpr (F16) {
                dp_stripe_f16<SMAX, VL2 ? SMAX / 2 : 0>(H, E, SEL, vF, cmx, vH, tab0, tab1, go, ge, &fmid);
                Hlast = H[SA - 1];
            } else if (!PERM) {
                IPX_DP_STRIPE(pk_lo16_pair((uint32_t)(int)pa0[j * 128], (uint32_t)(int)pa1[j * 128]))
            } else if (!IPX_STRIPE_ASM || !EXACT) {
                IPX_DP_STRIPE(pk_sext_hi8(pk_perm(tab1, tab0, SEL[PERM ? j : 0])))
            } else {
#if IPX_STRIPE_ASM
                // The same recurrence, hand-scheduled (gfx950).  A packed 16-bit op that feeds the very next packed
                // op costs a wait state, and the compiler's schedule left ~50 of them per column plus a copy of
                // every H register (old H[j] is the next segment's diagonal, so old and new overlapped).  Here a
                // segment's three dependent ops (H -> H-gapO -> F) are interleaved with the independent work of the
                // next two segments (score lookup, diagonal add, E): no wait states, and H[j] is updated in
                // place because the diagonal add of segment j+1 reads the old H[j] first.
                //   hp = max(diag + score, E)   of the segment about to be finished     (prepared one block ahead)
                //   em = E - gapE               of the same segment
                //   p1 = score of the next segment                                       (prepared one block ahead)
                if (SMAX > 0) {
                    pk16 hp, em, p1 = 0, q, vFm, tt;
                    {
                        const pk16 p0 = pk_sext_hi8(pk_perm(tab1, tab0, SEL[0]));
                        hp = pk_max(pk_add_sat(vH, p0), E[0]);
                        em = pk_subus(E[0], ge);
                        if (SMAX > 1) p1 = pk_sext_hi8(pk_perm(tab1, tab0, SEL[SMAX > 1 ? 1 : 0]));
                    }
                    IPX_UNROLL
                    for (int j = 0; j < SMAX; ++j) {
                        if (j + 2 < SMAX) {
                            asm volatile(
                                "v_pk_add_i16 %7, %0, %6 clamp\n\t"            /* q  = H[j](old) + score[j+1]          */
                                "v_pk_sub_u16 %8, %2, %15 clamp\n\t"           /* F' = F - gapE                         */
                                "v_pk_max_i16 %0, %4, %2\n\t"                  /* H[j] = max(hp, F)                     */
                                "v_perm_b32 %6, %13, %12, %11\n\t"             /* score[j+2] lookup ...                 */
                                "v_pk_sub_u16 %9, %0, %14 clamp\n\t"           /* tt = H[j] - gapO                      */
                                "v_pk_ashrrev_i16 %6, 8, %6 op_sel_hi:[0,1]\n\t" /* ... sign-extended                  */
                                "v_pk_max_i16 %2, %8, %9\n\t"                  /* F = max(F', tt)                       */
                                "v_pk_max_i16 %1, %5, %9\n\t"                  /* E[j] = max(E[j]-gapE, tt)             */
                                "v_pk_max_i16 %3, %3, %0\n\t"                  /* column maximum                        */
                                "v_pk_max_i16 %4, %7, %10\n\t"                 /* hp = max(q, E[j+1])                   */
                                "v_pk_sub_u16 %5, %10, %15 clamp"               /* em = E[j+1] - gapE                    */
                                : "+v"(H[j]), "+v"(E[j]), "+v"(vF), "+v"(cmx), "+v"(hp), "+v"(em), "+v"(p1), "=&v"(q), "=&v"(vFm), "=&v"(tt)
                                : "v"(E[j + 1 < SMAX ? j + 1 : 0]), "v"(SEL[j + 2 < SMAX ? j + 2 : 0]), "v"(tab0), "v"(tab1), "v"(go), "v"(ge));
                        } else if (j + 1 < SMAX) {
                            asm volatile(
                                "v_pk_add_i16 %7, %0, %6 clamp\n\t"
                                "v_pk_sub_u16 %8, %2, %12 clamp\n\t"
                                "v_pk_max_i16 %0, %4, %2\n\t"
                                "v_pk_max_i16 %4, %7, %10\n\t"                 /* hp = max(q, E[j+1]) (fills the gap)   */
                                "v_pk_sub_u16 %9, %0, %11 clamp\n\t"
                                "v_pk_max_i16 %3, %3, %0\n\t"                  /* column maximum (fills the gap)        */
                                "v_pk_max_i16 %2, %8, %9\n\t"
                                "v_pk_max_i16 %1, %5, %9\n\t"
                                "v_pk_sub_u16 %5, %10, %12 clamp"
                                : "+v"(H[j]), "+v"(E[j]), "+v"(vF), "+v"(cmx), "+v"(hp), "+v"(em), "+v"(p1), "=&v"(q), "=&v"(vFm), "=&v"(tt)
                                : "v"(E[j + 1 < SMAX ? j + 1 : 0]), "v"(go), "v"(ge));
                        } else {
                            asm volatile(
                                "v_pk_sub_u16 %6, %2, %9 clamp\n\t"
                                "v_pk_max_i16 %0, %4, %2\n\t"
                                "s_nop 0\n\t"
                                "v_pk_sub_u16 %7, %0, %8 clamp\n\t"
                                "v_pk_max_i16 %3, %3, %0\n\t"
                                "v_pk_max_i16 %2, %6, %7\n\t"
                                "v_pk_max_i16 %1, %5, %7"
                                : "+v"(H[j]), "+v"(E[j]), "+v"(vF), "+v"(cmx), "+v"(hp), "+v"(em), "=&v"(vFm), "=&v"(tt)
                                : "v"(go), "v"(ge));
                        }
                    }
                    Hlast = H[SA - 1];
                }
#endif
            }
#undef IPX_DP_STRIPE

            // -- lazy-F (ssw.c:302-313 / 507-518) --------------------------------------------------------
            // The reference passes each lane's final F to the next lane, walks the segments applying
            // H = max(H, F), F -= gapE, and leaves as soon as no lane has F > H - gapO.  With
            // gap_open > gap_ext (and, in the 8-bit pass, no carry in signed-compare territory) leaving
            // early never changes H, so the whole loop equals one max-plus prefix scan over the lanes,
            //   C_l = max_k sat(Fend[l-1-k] - k*segLen*gapE),   H[l][j] = max(H[l][j], sat(C_l - j*gapE)),
            // done here with log2(W) DPP row shifts.  Reads outside that regime fall through to the
            // reference's step-by-step loop below with its per-read, data-dependent exit.
            if constexpr (VL2) {
                // Two reference lanes per GPU lane: the lane's first SMAX/2 segments are SSE lane 2l (its F left as fmid), the
                // rest SSE lane 2l+1 (F left as vF), so the rows sit exactly where the 8-lane layout puts them and only the
                // carries differ: into lane 2l comes what left lane 2l-1 (the GPU lane above), into lane 2l+1 what left lane
                // 2l (this GPU lane).  What leaves a GPU lane downwards when nothing comes in is inj = max(F_B, F_A - D);
                // the scan over the GPU lanes is the usual one with twice the decay.
                pk16 cA = fmid, cB = vF;
                const pk16 big = pk_nzmask(pk_subus(cA, bigthr)) | pk_nzmask(pk_subus(cB, bigthr));
                const pk16 anybig = group_or<W>(big);
                const pk16 drop = anybig & fast_static;              // LOW: no lazy-F at all in such a column (see STAGE)
                cA &= ~drop; cB &= ~drop; dropped |= drop;
                cA &= fast_static; cB &= fast_static;
                pk16 x = xl_row_shr1(pkh_max(cB, pkh_add(cA, Dh)));
                if (l == 0) x = 0;
                pk16 y;
                y = xl_row_shr<1>(x); if (l < 1) y = 0; x = pkh_max(x, pkh_add(y, D1));
                y = xl_row_shr<2>(x); if (l < 2) y = 0; x = pkh_max(x, pkh_add(y, D2));
                y = xl_row_shr<4>(x); if (l < 4) y = 0; x = pkh_max(x, pkh_add(y, D4));
                const pk16 xb = pkh_max(cA, pkh_add(x, Dh));         // carry into the second reference lane of this GPU lane
                cmx = pk_max(cmx, pk_max(x, xb));                    // (all non-negative: halves order like integers)
                pk16 a = x;
                IPX_UNROLL
                for (int j = 0; j < SMAX; ++j) {
                    if (j == SMAX / 2) a = xb;
                    H[j] = pkh_max(H[j], a);
                    a = pkh_add(a, ge);
                }
                Hlast = H[SA - 1];
                vF = 0;
            } else {
                pk16 fe = fast_static;
                pk16 recheck = 0;                                   // (exact 8-bit stage) reads with a big carry in this column
                if (BYTE) {
                    const pk16 big = pk_nzmask(pk_subus(vF, bigthr));
                    const pk16 anybig = group_or<W>(big);
                    // LOW: a read with such a carry in this column gets NO lazy-F at all here (H stays the
                    // main loop's value, which the reference's loop can only raise): dropping just the big
                    // carry would not be a lower bound, because the reference's early exit also cuts the
                    // small carries that the big one happened to dominate
                    if (LOW) { const pk16 drop = anybig & fast_static; vF &= ~drop; dropped |= drop; }
                    else if (HIGH) dropped |= anybig & fast_static;   // every carry is passed on; remember that the read had such a column
                    else if (F16) fe &= ~anybig;
                    else recheck = anybig & fast_static;
                }
                pk16 x = xl_row_shr1(vF & fe);
                if (W == 8 && l == 0) x = 0;
                {   // (unconditional: a carry-free column is rare, and a branch here costs a copy of every H register)
                    pk16 y;
                    if (F16) {                                  // (D1.. hold the negated decays as halves)
                        y = xl_row_shr<1>(x); if (W == 8 && l < 1) y = 0; x = pkh_max(x, pkh_add(y, D1));
                        y = xl_row_shr<2>(x); if (W == 8 && l < 2) y = 0; x = pkh_max(x, pkh_add(y, D2));
                        y = xl_row_shr<4>(x); if (W == 8 && l < 4) y = 0; x = pkh_max(x, pkh_add(y, D4));
                        if (W == 16) { y = xl_row_shr<8>(x); x = pkh_max(x, pkh_add(y, D8)); }
                    } else {
                    y = xl_row_shr<1>(x); if (W == 8 && l < 1) y = 0; x = pk_max(x, pk_subus(y, D1));
                    y = xl_row_shr<2>(x); if (W == 8 && l < 2) y = 0; x = pk_max(x, pk_subus(y, D2));
                    y = xl_row_shr<4>(x); if (W == 8 && l < 4) y = 0; x = pk_max(x, pk_subus(y, D4));
                    if (W == 16) { y = xl_row_shr<8>(x); x = pk_max(x, pk_subus(y, D8)); }
                    }
                    if constexpr (BYTE && !LOW && !HIGH && !F16) {
                        // Exact 8-bit stage, a read with a big carry in this column (r03).  The reference's loop can leave a carry
                        // unapplied -- a CUT -- only at a step where that carry's value f lies in [128 + gapE, 128 + gapO) (f - gapE is
                        // still a "negative" byte, H' - gapO with H' = max(H, f) no longer is: the signed compare of ssw.c:311 says
                        // "not greater" where the unsigned one says "greater") and nobody else votes.  Such a carry is within
                        // gapO - gapE of the largest carry a arriving at that cell, so the cell has a in [128 + gapE, 128 + 2 gapO - gapE)
                        // and a main-loop H below 128 + gapO.  No such cell in the column: every exit of the loop is one the unsigned
                        // compare would have taken too, and the loop equals the closed form whatever the carries' size (checked column
                        // by column against the stepped loop on 1.6 x 10^7 columns before it went in: 2-3 % of columns have such a
                        // cell where 7-26 % have a big carry).  Only the reads WITH such a cell step through the reference's loop.
                        if (xl_any(recheck != 0)) {
                            pk16 a = x, cc = 0;
                            IPX_UNROLL
                            for (int j = 0; j < SMAX; ++j) {
                                if (j < S) {
                                    // (a >= 128 + gapE) and (a < 128 + 2 gapO - gapE) and (H[j] < 128 + gapO): all three differences non-zero
                                    cc |= pk_minu(pk_minu(pk_subus(a, bigthr), pk_subus(cutw, a)), pk_subus(cuth, H[j]));
                                    a = pk_subus(a, ge);
                                }
                            }
                            const pk16 cutcap = pk_nzmask(group_or<W>(cc)) & recheck;
                            fe &= ~cutcap;                          // these reads keep their carries for the stepped loop below
                            x &= ~cutcap;
                        }
                    }
                    cmx = pk_max(cmx, x);                       // (both non-negative: halves order like integers)
                    pk16 a = x;
                    if (F16 && IPX_STRIPE_ASM) {
#if IPX_STRIPE_ASM
                        // the integer apply below with the half-precision operations (ge = -gapE)
                        pk16 a2;
                        IPX_UNROLL
                        for (int j = 0; j < SMAX; j += 4) {
                            if (j + 3 < SMAX)
                                asm volatile("v_pk_add_f16 %5, %4, %6\n\tv_pk_max_f16 %0, %0, %4\n\t"
                                             "v_pk_add_f16 %4, %5, %6\n\tv_pk_max_f16 %1, %1, %5\n\t"
                                             "v_pk_add_f16 %5, %4, %6\n\tv_pk_max_f16 %2, %2, %4\n\t"
                                             "v_pk_add_f16 %4, %5, %6\n\tv_pk_max_f16 %3, %3, %5"
                                             : "+v"(H[j]), "+v"(H[j + 1 < SMAX ? j + 1 : 0]), "+v"(H[j + 2 < SMAX ? j + 2 : 0]),
                                               "+v"(H[j + 3 < SMAX ? j + 3 : 0]), "+v"(a), "=&v"(a2)
                                             : "v"(ge));
                            else if (j + 2 < SMAX)
                                asm volatile("v_pk_add_f16 %4, %3, %5\n\tv_pk_max_f16 %0, %0, %3\n\t"
                                             "v_pk_add_f16 %3, %4, %5\n\tv_pk_max_f16 %1, %1, %4\n\t"
                                             "s_nop 0\n\tv_pk_max_f16 %2, %2, %3"
                                             : "+v"(H[j]), "+v"(H[j + 1 < SMAX ? j + 1 : 0]), "+v"(H[j + 2 < SMAX ? j + 2 : 0]), "+v"(a), "=&v"(a2)
                                             : "v"(ge));
                            else if (j + 1 < SMAX)
                                asm volatile("v_pk_add_f16 %3, %2, %4\n\tv_pk_max_f16 %0, %0, %2\n\t"
                                             "s_nop 0\n\tv_pk_max_f16 %1, %1, %3"
                                             : "+v"(H[j]), "+v"(H[j + 1 < SMAX ? j + 1 : 0]), "+v"(a), "=&v"(a2)
                                             : "v"(ge));
                            else if (j < SMAX)
                                asm volatile("v_pk_max_f16 %0, %0, %1" : "+v"(H[j]) : "v"(a));
                        }
                        Hlast = H[SA - 1];
#endif
                    } else if (F16) {
                        IPX_UNROLL
                        for (int j = 0; j < SMAX; ++j) {
                            H[j] = pkh_max(H[j], a);
                            a = pkh_add(a, ge);
                        }
                        Hlast = H[SA - 1];
                    } else if (PERM && EXACT && IPX_STRIPE_ASM) {
#if IPX_STRIPE_ASM
                        // H[j] = max(H[j], a), a -= gapE, hand-scheduled like the stripe: two alternating carry
                        // registers keep every operand two instructions away from the op that produced it,
                        // and H is updated in place
                        pk16 a2;
                        IPX_UNROLL
                        for (int j = 0; j < SMAX; j += 4) {
                            if (j + 3 < SMAX)
                                asm volatile("v_pk_sub_u16 %5, %4, %6 clamp\n\tv_pk_max_i16 %0, %0, %4\n\t"
                                             "v_pk_sub_u16 %4, %5, %6 clamp\n\tv_pk_max_i16 %1, %1, %5\n\t"
                                             "v_pk_sub_u16 %5, %4, %6 clamp\n\tv_pk_max_i16 %2, %2, %4\n\t"
                                             "v_pk_sub_u16 %4, %5, %6 clamp\n\tv_pk_max_i16 %3, %3, %5"
                                             : "+v"(H[j]), "+v"(H[j + 1 < SMAX ? j + 1 : 0]), "+v"(H[j + 2 < SMAX ? j + 2 : 0]),
                                               "+v"(H[j + 3 < SMAX ? j + 3 : 0]), "+v"(a), "=&v"(a2)
                                             : "v"(ge));
                            else if (j + 2 < SMAX)
                                asm volatile("v_pk_sub_u16 %4, %3, %5 clamp\n\tv_pk_max_i16 %0, %0, %3\n\t"
                                             "v_pk_sub_u16 %3, %4, %5 clamp\n\tv_pk_max_i16 %1, %1, %4\n\t"
                                             "s_nop 0\n\tv_pk_max_i16 %2, %2, %3"
                                             : "+v"(H[j]), "+v"(H[j + 1 < SMAX ? j + 1 : 0]), "+v"(H[j + 2 < SMAX ? j + 2 : 0]), "+v"(a), "=&v"(a2)
                                             : "v"(ge));
                            else if (j + 1 < SMAX)
                                asm volatile("v_pk_sub_u16 %3, %2, %4 clamp\n\tv_pk_max_i16 %0, %0, %2\n\t"
                                             "s_nop 0\n\tv_pk_max_i16 %1, %1, %3"
                                             : "+v"(H[j]), "+v"(H[j + 1 < SMAX ? j + 1 : 0]), "+v"(a), "=&v"(a2)
                                             : "v"(ge));
                            else if (j < SMAX)
                                asm volatile("v_pk_max_i16 %0, %0, %1" : "+v"(H[j]) : "v"(a));
                        }
                        Hlast = H[SA - 1];
#endif
                    } else {
                    IPX_UNROLL
                    for (int j = 0; j < SMAX; ++j) {
                        if (j < S) {
                            H[j] = pk_max(H[j], a);
                            a = pk_subus(a, ge);
                            if (j == S - 1) Hlast = H[j];
                        }
                    }
                    }
                }
                vF &= ~fe;
            }
            // The reference's loop for the remaining reads, with its data-dependent exit (ssw.c:302-313).
            // 8-bit pass, up to 16 segments: one ROUND (all segments of one lane shift) at a time.  Inside a round step j
            // touches only segment j, so every step's exit vote -- F_{j+1} > H'_j - gapO as signed bytes, H'_j = max(H_j, F_j) --
            // depends on the round's incoming F alone: all votes are computed at once, each lane packs them into a bit per
            // step, ONE lane-group OR per round (instead of one per step) tells for every step whether anybody voted, and the
            // round is applied up to and including the first step nobody voted in.  (r02: the stepped form below cost ~20
            // instructions and a wave-wide branch per step, ~22 steps per stepping column of config 2a.)
            constexpr bool ROUNDS = STEP && BYTE && SMAX >= 1 && SMAX <= 16;
            if (ROUNDS) {
                const pk16 K = 0x00800080u, ONE = 0x00010001u;
                for (int k = 0; k < W; ++k) {
                    vF = xl_row_shr1(vF);
                    if (!xl_any(vF != 0)) break;
                    pk16 f = vF, acc = 0;
                    IPX_UNROLL
                    for (int j = 0; j < SMAX; ++j) {
                        if (j < S) {
                            const pk16 hn = pk_max(H[j], f);
                            f = pk_subus(f, ge);
                            const pk16 d = pk_subus(f ^ K, pk_subus(hn, go) ^ K);     // != 0: this lane votes to go on after step j
                            acc |= pk_minu(d, ONE) << j;                              // (j < 16: the halves stay apart)
                        }
                    }
                    const pk16 V = group_or<W>(acc);                                  // bit j (per half): somebody voted at step j
                    pk16 Mj = V ^ pk_add(V, ONE);                                     // bits 0..z, z = first step without a vote: the steps applied
                    f = vF;
                    IPX_UNROLL
                    for (int j = 0; j < SMAX; ++j) {
                        if (j < S) {
                            const pk16 h = pk_max(H[j], pk_mul(f, Mj & ONE));
                            cmx = pk_max(cmx, h);
                            H[j] = h;
                            if (j == S - 1) Hlast = h;
                            f = pk_subus(f, ge);
                            Mj = pk_shr1(Mj);
                        }
                    }
                    // a read goes on to the next round iff every step of this one had a vote; the others carry F = 0
                    const pk16 full = pk_splat(S < 16 ? (1 << (S & 15)) - 1 : 0xFFFF);
                    vF = f & ~pk_nzmask(~V & full);
                }
            }
            // step-by-step form: the 16-bit pass (jobs with gap_open <= gap_ext) and 8-bit reads of more than 256 bp
            for (int k = 0; STEP && !ROUNDS && k < W; ++k) {
                vF = xl_row_shr1(vF);
                if (W == 8 && l == 0) vF = 0;
                if (!xl_any(vF != 0)) break;
                bool fin = false;
                IPX_UNROLL
                for (int j = 0; j < SMAX; ++j) {
                    if (j < S && !fin) {
                        const pk16 h = pk_max(H[j], vF);
                        cmx = pk_max(cmx, h);
                        H[j] = h;
                        if (j == S - 1) Hlast = h;
                        const pk16 h2 = pk_subus(h, go);
                        vF = pk_subus(vF, ge);
                        // exit test: no lane with F > H-gapO; the 8-bit pass compares SIGNED bytes (ssw.c:311)
                        const pk16 d = BYTE ? pk_subus(vF ^ 0x00800080u, h2 ^ 0x00800080u) : pk_subus(vF, h2);
                        if (!xl_any(d != 0)) fin = true;
                        else vF &= pk_nzmask(group_or<W>(d));      // reads that left the loop carry F = 0
                    }
                }
                if (fin) break;
            }

            // -- column maximum, best score bookkeeping (ssw.c:316-337 / 521-539) ----------------------
            const pk16 cmA = group_max<W>(cmx);
            const pk16 om = BYTE ? (pk_nzmask(pk_subus(cmA, capm1)) & act) : 0u;  // 8-bit overflow: leave before recording
            const pk16 a2 = act & ~om;
            const pk16 nb = pk_max(best, cmA);
            const pk16 m = pk_nzmask(pk_sub(nb, best)) & a2;                         // strictly better and still running
            best = pk_select(m, nb, best);
            // on overflow the reference has already stored the saturated maximum (255-bias) in `max` when it
            // leaves, but not the column: the end-of-pass search then compares the OLD column with it
            if (BYTE) best = pk_select(om, capv, best);
            endref = pk_select(m, icol, endref);
            if (xl_any(m != 0)) {
                IPX_UNROLL
                for (int j = 0; j < SMAX; ++j)
                    if (j < S) HM[j] = pk_select(m, H[j], HM[j]);                    // pvHmax (ssw.c:331 / 533)
            }
            if (!REV) cm4 = (l == (t & 3)) ? cmA : cm4;
            ovf |= om;
            done |= om;
            if (REV) done |= (~pk_nzmask(cmA ^ term)) & a2;                          // maxColumn[i] == terminate
            // -- anything left to do? ---------------------------------------------------------------
            bool alive = false;
            IPX_UNROLL
            for (int h = 0; h < 2; ++h)
                alive = alive || ((t + 1 < tb[h] + ncol[h]) && ((done >> (16 * h)) & 0xFFFFu) == 0);
            if (!xl_any(alive)) { stop = true; break; }
          }
          // the next group's letters have arrived (requested a group ago); request those of the group after it -- at the END of the
          // body and unconditionally (see dp_skew_tile: requested at the top under "not the first group", every group waited for the
          // request it had just issued)
          IPX_UNROLL
          for (int h = 0; h < 2; ++h) cur[h] = nxt[h];
          IPX_VMEM_FENCE();
          IPX_UNROLL
          for (int h = 0; h < 2; ++h) {
              int k = REV ? (idx0[h] >> 2) - ((t0 >> 2) + 2) : (t0 >> 2) + 2;
              if (k < 0) k = 0;
              if (k > kmax[h]) k = kmax[h];
              nxt[h] = load_global_u32(refw[h] + k);
          }
        }
        if (!REV && tdone >= 0 && l <= (tdone & 3)) {                                                             // last group
            if (mc_lds) maxcol[((tdone & ~3) + l) * G + g] = cm4; else store_global_u32(maxcol + (((tdone & ~3) + l) * G + g), cm4);
        }

        // ---- finalisation ---------------------------------------------------------------------------
        IPX_SYNC();   // column maxima written by lane 0 of each group (global scratch, same wave) are visible to the group
        // The per-slot facts needed from here on are looked up AGAIN rather than kept alive across the column
        // loop: kept alive they are spilled (one scratch dword per lane each), which costs more memory
        // traffic per tile than the tile's whole input.
        IPX_COMPILER_FENCE();
        IPX_UNROLL
        for (int h = 0; h < 2; ++h) {
            const int slot = 2 * g + h;
            job[h] = -1; L[h] = 0; ncol[h] = 0; score1[h] = 0; rend1[h] = -1;
            if (slot < cnt) {
                const int64_t jb = (int64_t)p.perm[first + slot];
                job[h] = jb;
                if (!REV) {
                    L[h] = (int)(b.read_off[jb + 1] - b.read_off[jb]);
                    ncol[h] = b.ref_len[b.ref_id[jb]];
                } else {
                    const IpxResult r = b.res[jb];
                    L[h] = r.read_end1 + 1; if (L[h] < 0) L[h] = 0;
                    score1[h] = r.score1;
                    rend1[h] = r.read_end1;
                }
            }
        }
        IPX_UNROLL
        for (int h = 0; h < 2; ++h) {
            // end position on the read: smallest striped row holding `best` in the saved column (ssw.c:340-349)
            const unsigned bh = (best >> (16 * h)) & 0xFFFFu;
            uint32_t rmin = 0x7FFFFFFFu;
            IPX_UNROLL
            for (int j = SMAX - 1; j >= 0; --j)
                if (j < S && (((HM[j] >> (16 * h)) & 0xFFFFu) == bh)) rmin = (uint32_t)(j + l * S);
            rmin = group_umin<W>(rmin);
            int end_read = L[h] - 1;
            if ((int)rmin < end_read) end_read = (int)rmin;
            const int eref = (int)(int16_t)((endref >> (16 * h)) & 0xFFFFu);
            const bool overflow = ((ovf >> (16 * h)) & 0xFFFFu) != 0;

            if (!REV) {
                // second best outside the mask window (ssw.c:366-379 / 568-581)
                const int refLen = ncol[h];
                const int maskLen = job[h] >= 0 ? mask_len_of(b, job[h], L[h]) : 15;
                int edgeL = eref - maskLen; if (edgeL < 0) edgeL = 0;
                int edgeR = eref + maskLen; if (edgeR > refLen) edgeR = refLen;
                if (BYTE) edgeR += 1;
                uint32_t key2 = 0xFFFFu;                              // (score2 = 0, ref_end2 = 0)
                for (int col = l; col < refLen; col += W) {
                    if (col < edgeL || col >= edgeR) {
                        const uint32_t v = (maxcol[col * G + g] >> (16 * h)) & 0xFFFFu;
                        const uint32_t kk = (v << 16) | (0xFFFFu - (uint32_t)col);
                        if (v > (key2 >> 16)) key2 = kk;
                    }
                }
                key2 = group_umax<W>(key2);
                const bool lost = (LOW || HIGH) && ((group_or<W>(dropped) >> (16 * h)) & 0xFFFFu) != 0;   // (all lanes take part)
                int key = -1;                                         // pass the job takes next (plan_note below)
                if (l == 0 && job[h] >= 0) {
                    IpxResult r = b.res[job[h]];
                    // a 16-bit result may already sit in the record (IPX_PASS_WORD_FIRST): an overflowing 8-bit pass keeps it
                    const bool has_word = r.mode == IPX_MODE_NEED_BYTE_CHECK || r.mode == IPX_MODE_NEED_BYTE_EXACT_W;
                    const int s2 = maskLen >= 15 ? (int)(F16 ? ipx_f16_to_uint(key2 >> 16) : (key2 >> 16)) : 0;  // ssw.c:864-870
                    const int e2 = maskLen >= 15 ? (int)(0xFFFFu - (key2 & 0xFFFFu)) : -1;
                    // plain-first flow: this is the lower-bound stage AFTER the plain recurrence (IPX_PASS_BYTE_LOW2), whose outputs
                    // the record may hold for comparison (proof failed) or not (the plain recurrence reached the overflow threshold)
                    const bool cmp_plain = r.mode == IPX_MODE_NEED_BYTE_LOW_CMP, after_plain = cmp_plain || r.mode == IPX_MODE_NEED_BYTE_LOW;
                    if (HIGH) {
                        // the record holds the lower-bound stage's outputs: equal outputs certify them (see STAGE above);
                        // a read the upper-bound stage saw no big carry in is exact by itself
                        const bool same = !overflow && r.score1 == (uint16_t)bh && r.ref_end1 == eref && r.read_end1 == end_read &&
                                          r.score2 == (uint16_t)s2 && r.ref_end2 == e2;
                        if (same || (!overflow && !lost)) {
                            r.mode = IPX_MODE_BYTE;
                            r.score1 = (uint16_t)bh; r.ref_end1 = eref; r.read_end1 = end_read; r.read_begin1 = -1;
                            r.score2 = (uint16_t)s2; r.ref_end2 = e2;
                        } else r.mode = IPX_MODE_NEED_BYTE_EXACT;
                    } else if (BYTE && overflow) {
                        if (has_word) r.mode = IPX_MODE_WORD;
                        else if (b.score_size == 2) { r.mode = IPX_MODE_NEED_WORD; r.score1 = 255; }            // -> 16-bit pass (ssw.c:844-847)
                        else { r.mode = IPX_MODE_FAIL; r.score1 = 255; }                                       // ssw.c:848-851
                    } else if (r.mode == IPX_MODE_NEED_BYTE_EXACT_P) {
                        // the stepped pass after a failed proof: the record holds the plain recurrence's outputs -- equal ones keep the
                        // read on the plain reverse pass
                        const unsigned sc = F16 ? ipx_f16_to_uint(bh) : bh;
                        const bool same = r.score1 == (uint16_t)sc && r.ref_end1 == eref && r.read_end1 == end_read && r.score2 == (uint16_t)s2 && r.ref_end2 == e2;
                        r.mode = (same && rev_needed(b, sc)) ? IPX_MODE_BYTE_PLAIN : IPX_MODE_BYTE;
                        r.score1 = (uint16_t)sc; r.ref_end1 = eref; r.read_end1 = end_read; r.read_begin1 = -1;
                        r.score2 = (uint16_t)s2; r.ref_end2 = e2;
                    } else if (after_plain) {
                        const unsigned sc = F16 ? ipx_f16_to_uint(bh) : bh;
                        const bool same = cmp_plain && r.score1 == (uint16_t)sc && r.ref_end1 == eref && r.read_end1 == end_read &&
                                          r.score2 == (uint16_t)s2 && r.ref_end2 == e2;
                        if (same) r.mode = rev_needed(b, sc) ? IPX_MODE_BYTE_PLAIN : IPX_MODE_BYTE;   // squeezed between equal bounds: the reference's
                        else if (lost) r.mode = IPX_MODE_NEED_BYTE_EXACT;                             // bounds differ: the stepped pass decides
                        else {                                                                         // nothing was dropped: this IS the exact pass
                            r.mode = IPX_MODE_BYTE;
                            r.score1 = (uint16_t)sc; r.ref_end1 = eref; r.read_end1 = end_read; r.read_begin1 = -1;
                            r.score2 = (uint16_t)s2; r.ref_end2 = e2;
                        }
                    } else if (lost && (has_word || !b.use_bracket || L[h] < b.bracket_min_len)) {
                        r.mode = has_word ? IPX_MODE_NEED_BYTE_EXACT_W : IPX_MODE_NEED_BYTE_EXACT;             // lower bound only: exact 8-bit pass decides
                    } else {
                        r.mode = lost ? IPX_MODE_NEED_BYTE_HIGH                                                // lower-bound outputs kept for the upper-bound stage
                                      : BYTE ? IPX_MODE_BYTE : (pass == IPX_PASS_WORD_FIRST ? IPX_MODE_WORD_UNPROVEN : IPX_MODE_WORD);
                        r.score1 = (uint16_t)(F16 ? ipx_f16_to_uint(bh) : bh);
                        r.ref_end1 = eref;
                        r.read_end1 = end_read;
                        r.read_begin1 = -1;
                        r.score2 = (uint16_t)s2; r.ref_end2 = e2;
                    }
                    b.res[job[h]] = r;
                    key = next_pass_key(b, r, L[h], b.gap_open[job[h]] <= b.gap_ext[job[h]]);
                }
                plan_note(b, key);
            } else {
                if (l == 0 && job[h] >= 0) {
                    IpxResult r = b.res[job[h]];
                    const unsigned best_rev = (BYTE && overflow) ? 255u : (F16 ? ipx_f16_to_uint(bh) : bh);
                    r.ref_begin1 = eref;                                                                       // ssw.c:885
                    r.read_begin1 = rend1[h] - end_read;                                                       // ssw.c:886
                    if ((unsigned)score1[h] > best_rev) r.flag = 2;                                            // ssw.c:888-891
                    b.res[job[h]] = r;
                }
            }
        }
    }
}
template <int W, int SMAX, bool REV, bool EXACT, int STAGE, bool PERM = false, bool F16 = false, bool VL2 = false>
IPX_KERNEL_WAVE_OCC(((PERM && REV) || VL2) ? ipx_dp_perm_waves(SMAX) : 1) void k_dp_pass(IpxBatch b, IpxPlan p, int cls_lo, int cls_hi, int maxcols, int pass, uint64_t skip_fast, uint64_t skip_slow)
{
    dp_pass_body<W, SMAX, REV, EXACT, STAGE, PERM, F16, VL2>(b, p, cls_lo, cls_hi, maxcols, pass, skip_fast, skip_slow, (uint32_t)IPX_BID, (uint32_t)IPX_GDIM);
}

// a function body as a CALL (tier kernels): inlined, the bodies of a tier share one register allocation and the longest one's spills
// land in all of them; called, each keeps the allocation of its stand-alone kernel (the batch and plan descriptors then travel by
// reference, i.e. through private memory)
#if defined(IPX_CPU_EMU)
#define IPX_NOINLINE_DEV static
// (emulator: plain references)
#define IPX_CALLEE_DESC_PARAMS const IpxBatch &b, const IpxPlan &p,
#define IPX_CALLEE_DESC_ARGS b, p,
#define IPX_CALLEE_DESC_LOCALS
#else
#define IPX_NOINLINE_DEV static __device__ __attribute__((noinline))
// r04: the called bodies do NOT receive the batch and plan descriptors.  Passed by reference, the caller had to materialise both structs
// in private memory -- 288-368 bytes per LANE, written by every wave of every tier launch before its first tile (a launch that found
// nothing still wrote a quarter of a gigabyte) and read back by the callee.  Every kernel that calls these bodies takes (IpxBatch b,
// IpxPlan p, ...) as its first two arguments, so the callee reads them where they already are: the kernel-argument segment, through a
// wave-uniform pointer (scalar loads, no private memory).
#define IPX_KERNARG_AS __attribute__((address_space(4)))
// (the KERNEL takes the address of its argument segment and hands it down: the intrinsic is a kernel's own, in a called function it
//  returned null on the box -- a memory fault in the first r04 attempt; made wave-uniform again in the callee so that the loads are scalar)
#define IPX_CALLEE_DESC_PARAMS const IPX_KERNARG_AS char *ka_in_,
#define IPX_CALLEE_DESC_ARGS ((const IPX_KERNARG_AS char *)__builtin_amdgcn_kernarg_segment_ptr()),
#define IPX_CALLEE_DESC_LOCALS                                                                                                      \
    const uint64_t kau_ = (uint64_t)ka_in_;                                                                                          \
    const IPX_KERNARG_AS char *ka_ = (const IPX_KERNARG_AS char *)(((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(kau_ >> 32)) << 32) | \
                                                                    (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)kau_));       \
    const IpxBatch &b = *(const IpxBatch *)ka_;                        /* (the compiler infers the constant address space back) */    \
    const IpxPlan &p = *(const IpxPlan *)(ka_ + ((sizeof(IpxBatch) + alignof(IpxPlan) - 1) & ~(alignof(IpxPlan) - 1)));
#endif

// k_dp_pass_tier (r03): the STEPPED 8-bit passes (exact stage, forward and reverse; selector profile, fast gaps) of the classes
// SLO..SHI in ONE launch.  What reaches these passes is what the proofs leave open: a few thousand reads of one class, eight of
// another, none of most -- and every class had a launch of its own, each lasting as long as ONE tile takes (0.2-0.7 ms: hundreds of
// columns, a hundred-odd dependent instructions each), one after the other on the stream.  Here block b takes the b-th tile of
// the concatenated tile lists of the launch's classes (grid = the number of tiles the previous run of the context saw), so the
// tiles of all classes are in flight together and the launch lasts as long as its longest tile.
#define IPX_PASS_TIER_LO 0                      // classes the launch covers (8-bit segLen: reads of up to 256 bp; 0 = empty reads)
#define IPX_PASS_TIER_HI 16
template <int W, int S, bool REV, int STAGE>
IPX_NOINLINE_DEV void dp_pass_body_call(IPX_CALLEE_DESC_PARAMS int maxcols, int pass, uint32_t rank, uint32_t nrank)
{
    IPX_CALLEE_DESC_LOCALS
    dp_pass_body<W, S, REV, true, STAGE, true, false, false>(b, p, S, S, maxcols, pass, (uint64_t)0, (uint64_t)0, rank, nrank);
}
template <int W, int SLO, int SHI, bool REV, int STAGE>
IPX_DEV void dp_pass_tier_walk(uint32_t set_mask, const IpxBatch &b, const IpxPlan &p, int maxcols, int pass, uint32_t base)
{
    if constexpr (SLO <= SHI) {
        if ((set_mask >> SLO) & 1u) {
            const uint32_t n = p.tile_off[SLO + 1] - p.tile_off[SLO], nb = (uint32_t)IPX_GDIM;
            const uint32_t rank = ((uint32_t)IPX_BID + nb - base % nb) % nb;      // this block's rank for the class: its tiles follow the classes before it
            if (rank < n) dp_pass_body_call<W, SLO, REV, STAGE>(IPX_CALLEE_DESC_ARGS maxcols, pass, rank, nb);
            base += n;
        }
        dp_pass_tier_walk<W, SLO + 1, SHI, REV, STAGE>(set_mask, b, p, maxcols, pass, base);
    }
}
template <int W, int SLO, int SHI, bool REV, int STAGE>
IPX_KERNEL_WAVE_OCC(REV ? ipx_dp_perm_waves(SHI) : 1) void k_dp_pass_tier(IpxBatch b, IpxPlan p, uint32_t set_mask, int maxcols, int pass)
{
    IPX_RAISE_PRIO(b);
    dp_pass_tier_walk<W, SLO, SHI, REV, STAGE>(set_mask, b, p, maxcols, pass, 0u);
}

// ------------------------------------------------------------------------------------------------
// k_dp_skew: the 16-bit passes (exact segLen, selector profile, half precision: see k_dp_pass PERM / F16) as a WAVEFRONT
// over the SSE lanes.  In the striped layout lane l of a read holds rows l*S .. l*S+S-1, and the only values that cross from
// lane l-1 to lane l inside a column are the F leaving its last row and (one column earlier) the H of that row.  k_dp_pass runs
// all lanes on the same column, so it has to guess F = 0 at the lane boundaries and repair the column afterwards (lazy-F: a
// max-plus scan over the lanes and one max per segment, ~50 operations per column).  Here lane l works on column t - l at
// step t: what it needs from lane l-1 was computed one step (F, column maximum so far) or two steps (diagonal H) earlier and
// arrives through one DPP row shift each.  No lazy-F, no group reduction for the column maximum (a running maximum travels
// down the lanes with the column; lane 7 stores it), 7 extra steps per tile.
//   Exactness: what this computes is the plain Gotoh recurrence -- E also sees an H that F raised across a lane boundary,
//   which the reference's lazy-F loop never feeds back into E (ssw.c:507-518).  The H matrix is the same nevertheless (the
//   corner "gap down, then gap right" costs what "gap right, then gap down" costs, and the reference computes the latter
//   exactly), with gap_open > gap_ext as for every kernel without a stepped loop; every output of the pass is a function
//   of H.  Pinned like every other kernel: bit for bit against the reference (tests/test_gpu_stress.py, the bench digests).
//   Columns outside a read's window (before its start, the wavefront's lead-in and drain, other reads' longer windows) are
//   given window letter 5, whose scores are -2048: H stays 0 there until real columns arrive and stays below the best after.
//   Best score: every LANE tracks (best over its rows, first column with it, its H there); the end of the pass combines:
//   maximum, then first column, then smallest row -- the reference's order (ssw.c:521-539, 545-556).
// Same LDS layout, job tables and finalisation as k_dp_pass.
// ------------------------------------------------------------------------------------------------
template <int W> IPX_DEV pk16 group_minu(pk16 x)
{
    x = pk_minu(x, xl_xor1(x));
    x = pk_minu(x, xl_xor2(x));
    x = pk_minu(x, xl_half_mirror(x));
    if (W >= 16) x = pk_minu(x, xl_mirror(x));
    if (W >= 32) x = pk_minu(x, xl_shfl(x, lane_id() ^ 16));
    if (W >= 64) x = pk_minu(x, xl_shfl(x, lane_id() ^ 32));
    return x;
}
//   BH = 1: the 8-bit forward pass's UPPER-BOUND stage (k_dp_pass HIGH) for reads of up to 8*SMAX bp: what that stage computes --
//        every carry passed on -- is this same plain recurrence, whatever the striping (8 lanes here, 16 in the reference's 8-bit
//        pass: row numbers, not lanes, enter the outputs), so it runs here at 16 reads per wave and without lazy-F.  The
//        finalisation speaks the 8-bit pass's dialect (end_ref starts at -1, the second-best scan reaches one column further,
//        ssw.c:220, 374) and compares with the lower-bound stage's outputs in the record, as k_dp_pass HIGH does.  `cls` is the
//        8-bit class the pass's job list is bucketed by; its tiles hold 16 jobs.
//   BH = 2: the same plain recurrence in the 8-bit dialect run FIRST (IPX_PASS_BYTE_FIRST of the plain-first flow, forward) and as the
//        8-bit REVERSE pass of reads whose forward result equals the plain recurrence's (IPX_PASS_BYTE_REV_PLAIN).  The reference's
//        8-bit matrix lies below the plain one cell by cell (its lazy-F loop only ever stops early, ssw.c:302-313) and equals it in
//        every column processed before the first value >= 128 appears (no carry can reach the signed compare, ssw.c:311).  So:
//        a best score < 128 is final as it stands; otherwise the outputs go into the record and k_prove_plain certifies them from
//        below (a banded lower bound through the best cell) -- or hands the read to the lower-bound / stepped kernels.
//   ROW SHIFT (r03): the kernel has 8*SMAX rows; a read whose padded row count in the reference (8*segLen, 16*segLen8 in the 8-bit
//        dialect) is SMALLER is served all the same: its rows are shifted DOWN by the difference.  The rows above it select the
//        constant score 0 and, starting from H = E = F = 0 with nothing but zeros arriving from above, stay 0 for ever -- they are
//        the matrix's row -1 repeated -- so the read's own rows compute exactly what they would in a kernel of their own size, and
//        the row numbers in the outputs are the kernel's minus the shift.  This lets the planner list a rare segLen class under the
//        next populated one (IpxBatch::cls_map) instead of giving ~50 waves a launch to themselves on a 1 024-SIMD chip.
// waves per SIMD to ask for: segLen 20..25 sits just above the three-wave register budget (170) -- a few spilled values cost less
// than the third wave brings (r02: config 4's 200 bp class); longer reads run at two waves
IPX_HD constexpr int ipx_skew_waves(int smax, bool rev) { return (smax >= 20 && smax <= 25) ? 3 : (rev ? ipx_dp_perm_waves(smax) : 1); }
// one tile of k_dp_skew: the 16 reads p.perm[first .. first + cnt)
// lane l <- lane l-1: inside the 16-lane DPP row for 8 lanes per read (what crosses into a group's first lane is masked by the caller),
// across the whole wavefront for the latency tier's 32 / 64 lanes per read
template <int W> IPX_DEV uint32_t skew_shr1(uint32_t v) { return W <= 16 ? xl_row_shr1(v) : xl_wave_shr1(v); }
template <int SMAX, bool REV, int BH, int W = 8>
IPX_DEV void dp_skew_tile(const IpxBatch &b, const IpxPlan &p, const uint32_t first, const int cnt, const int pass, uint32_t *maxcol, const bool mc_lds,
                          unsigned char *lds, const uint32_t nz, uint32_t *winlds = nullptr, const int winstride = 0)
{
    // LATENCY TIER (W >= 32): the tile's windows are staged in LDS (winlds: winstride words per read).  The throughput form fetches one dword of
    // window letters per group of four steps from global memory, one group ahead: four steps of ~200 instructions hide that, four steps of
    // 50-70 do not -- the first W = 64 kernel spent 330 ns per step waiting for it (121 us per pass where W = 32 took 71).
    constexpr bool WINLDS = W >= 32;
    static_assert(!(BH == 1 && REV), "the upper-bound stage of the bracket is a forward pass");
    constexpr int SA = SMAX > 0 ? SMAX : 1;
    static_assert(W == 8 || W == 32, "8 lanes per read (16 reads per wave), or the latency tier's 32");
    constexpr int G = 64 / W, S = SMAX;
    const int lane = lane_id();
    const int g = lane / W, l = lane % W;
    // ---- per-slot parameters (index 0 = low half, 1 = high half of every packed register) ----
    int64_t job[2];
    int L[2], ncol[2], tb[2], idx0[2], kmax[2], score1[2], rend1[2];
    const int8_t *rd[2];
    const uint32_t *refw[2];
    int gO[2], gE[2];
    IPX_UNROLL
    for (int h = 0; h < 2; ++h) {
        const int slot = 2 * g + h;
        job[h] = -1; L[h] = 0; ncol[h] = 0; tb[h] = 0; idx0[h] = 3; kmax[h] = 0; score1[h] = 0; rend1[h] = -1;
        rd[h] = b.reads; refw[h] = (const uint32_t *)b.refs_packed; gO[h] = 1; gE[h] = 0;
        if (slot < cnt) {
            const int64_t jb = (int64_t)p.perm[first + slot];
            const int rid = b.ref_id[jb];
            const int refLen = b.ref_len[rid];
            job[h] = jb;
            rd[h] = b.reads + b.read_off[jb];
            refw[h] = (const uint32_t *)(b.refs_packed + b.refp_off[rid]);
            kmax[h] = ((refLen + 3) >> 2) + 1;
            gO[h] = b.gap_open[jb];
            gE[h] = b.gap_ext[jb];
            if (!REV) {
                L[h] = (int)(b.read_off[jb + 1] - b.read_off[jb]);
                ncol[h] = refLen;
            } else {
                const IpxResult r = b.res[jb];
                L[h] = r.read_end1 + 1; if (L[h] < 0) L[h] = 0;
                ncol[h] = r.ref_end1 + 1; if (ncol[h] < 0) ncol[h] = 0;
                score1[h] = r.score1;
                rend1[h] = r.read_end1;
                if (ncol[h] > 0) { idx0[h] = r.ref_end1 | 3; tb[h] = idx0[h] - r.ref_end1; }
            }
        }
    }
    // rows the read is shifted down by: the kernel's row count minus the reference's padded row count for this read (ROW SHIFT above)
    int dl[2];
    IPX_UNROLL
    for (int h = 0; h < 2; ++h) dl[h] = W * S - (BH ? 16 * ((L[h] + 15) >> 4) : 8 * ((L[h] + 7) >> 3));
    {   // this kernel has no stepped lazy-F and computes in halves: refuse what would need more (host-side routing error)
        const bool bad = (job[0] >= 0 && (gO[0] <= gE[0] || L[0] > b.f16_max_len || dl[0] < 0)) ||
                         (job[1] >= 0 && (gO[1] <= gE[1] || L[1] > b.f16_max_len || dl[1] < 0));
        if (xl_any(bad) && lane == 0) atomic_or_u32(b.status, IPX_STATUS_INTERNAL);
    }
    const pk16 go = pk_make((int)ipx_f16_from_int(-gO[0]), (int)ipx_f16_from_int(-gO[1]));     // -gapO, -gapE as halves
    const pk16 ge = pk_make((int)ipx_f16_from_int(-gE[0]), (int)ipx_f16_from_int(-gE[1]));
    const pk16 term = pk_make((int)ipx_f16_from_uint((uint32_t)score1[0]), (int)ipx_f16_from_uint((uint32_t)score1[1]));

    // ---- selectors of the striped rows' read letters (k_dp_pass PERM) ----
    pk16 SEL[SA];
    {
        int raw[2][SA];
        IPX_UNROLL
        for (int h = 0; h < 2; ++h) {
            IPX_UNROLL
            for (int j = 0; j < SMAX; ++j) {
                const int r = j + l * S - dl[h];               // the read's row (ssw.c:178-185) held by this lane's segment j
                int idx = r < L[h] ? r : L[h] - 1;
                if (idx < 0) idx = 0;
                if (REV) idx = L[h] - 1 - idx;                 // reverse pass: seq_reverse (ssw.c:774-785)
                raw[h][j] = load_stream_i8(L[h] > 0 ? rd[h] + idx : (const int8_t *)b.read_off);
            }
        }
        IPX_UNROLL
        for (int j = 0; j < SMAX; ++j) {
            uint32_t sel = 0;
            IPX_UNROLL
            for (int h = 0; h < 2; ++h) {
                const int r = j + l * S - dl[h];
                const unsigned base = (unsigned)raw[h][j];
                uint32_t sh = 0x0c0cu;                         // rows above the read, padding rows, letter N: constant 0
                if (r >= 0 && r < L[h] && base < 4u) sh = 0x000cu | ((base + 4u * h) << 8);   // high byte <- table byte 4*h + base
                sel |= sh << (16 * h);
            }
            SEL[j] = sel;
        }
    }

    // ---- DP state -----------------------------------------------------------------------------
    pk16 H[SA], E[SA], HM[SA];
    IPX_UNROLL
    for (int j = 0; j < SA; ++j) { H[j] = 0; E[j] = 0; HM[j] = 0; }
    pk16 Hl_cur = 0, Hl_old = 0;           // this lane's last row after the previous step / the step before
    pk16 vFend = 0;                        // F leaving this lane's last row, previous step
    pk16 pm = 0;                           // maximum of this lane's column over the lanes up to this one
    pk16 lbest = 0, lcol = 0;              // per lane: best H of its rows, first column with it (counted in processing order)
    uint32_t let = 0x1414u;                // window letters of this lane's column, TIMES FOUR (= byte offsets into the score table; low byte:
                                           //   low half's read), 5 = no column
    uint32_t lsel0 = l == 0 ? 0x0c0c0100u : 0x0c0c0504u, lsel1 = l == 0 ? 0x0c0c0302u : 0x0c0c0504u;   // (see the step loop)
    IPX_KEEP_VGPR(lsel0);
    IPX_KEEP_VGPR(lsel1);
    pk16 ccol = pk_make(-l, -l);           // this lane's column, counted in processing order: t - l
    pk16 seen = 0;                         // (reverse) this read was seen to have reached its score

    const int T = (int)xl_first(wave_umax((uint32_t)((tb[0] + ncol[0]) > (tb[1] + ncol[1]) ? (tb[0] + ncol[0]) : (tb[1] + ncol[1]))));   // (a scalar: the step loop's bounds are uniform)
    // the last lane is W-1 columns behind the first; steps come in groups of four, the last group's surplus steps process columns past
    // every window (letter 5: nothing improves, and the column maxima they store land in the 4 spare columns of the scratch)
    const int TT = T > 0 ? (T + (W - 1) + 3) & ~3 : 0;
    uint32_t pairA = 0x14141414u, pairB = 0x14141414u;         // letters (x 4) of the first lane's next four columns: bytes (half 0, half 1) x 2 each
    uint32_t cur[2], nxt[2];
    IPX_UNROLL
    for (int h = 0; h < 2; ++h) {
        const int k0 = REV ? (idx0[h] >> 2) : 0;
        int k1 = REV ? (idx0[h] >> 2) - 1 : 1;
        if (k1 < 0) k1 = 0;
        if (k1 > kmax[h]) k1 = kmax[h];
        if (WINLDS) {
            uint32_t *wl = winlds + (size_t)(2 * g + h) * (size_t)winstride;
            for (int q = l; q <= kmax[h] && q < winstride; q += W) wl[q] = load_global_u32(refw[h] + q);
        } else {
            cur[h] = load_global_u32(refw[h] + (k0 > kmax[h] ? kmax[h] : k0));
            nxt[h] = load_global_u32(refw[h] + k1);
        }
    }
    if (WINLDS) {
        IPX_SYNC();                                                 // the staged windows are visible to every lane of their group
        IPX_UNROLL
        for (int h = 0; h < 2; ++h) {
            const uint32_t *wl = winlds + (size_t)(2 * g + h) * (size_t)winstride;
            const int k0 = REV ? (idx0[h] >> 2) : 0;
            int k1 = REV ? (idx0[h] >> 2) - 1 : 1;
            if (k1 < 0) k1 = 0;
            if (k1 > kmax[h]) k1 = kmax[h];
            cur[h] = wl[k0 > kmax[h] ? kmax[h] : k0];
            nxt[h] = wl[k1];
        }
    }

    int tend = -1;                                              // (reverse) step at which every read has passed its last column
    for (int t0 = 0; t0 < TT && (tend < 0 || t0 < tend); t0 += 4) {
        if (REV && tend < 0) {
            // A lane that holds the score the pass must reach got it in a column at or after the read's first such column,
            // W-1 steps at most before every lane has been through that column: seen here, the read needs 7 more steps.
            seen |= group_or<W>(~pk_nzmask(lbest ^ term));
            const bool pending = (job[0] >= 0 && (seen & 0xFFFFu) == 0) || (job[1] >= 0 && (seen >> 16) == 0);
            if (!xl_any(pending)) tend = t0 + (W - 1);
        }
        {   // the four columns of this group as the first lane will see them: letter x 4 (always a valid letter: windows are
            // sanitised and padded), 5 x 4 outside the read's window; interleaved so that one v_perm_b32 per step picks a column
            uint32_t x[2];
            bool inside = true;                                  // all four columns inside both windows (the usual group)
            IPX_UNROLL
            for (int h = 0; h < 2; ++h) {
                x[h] = cur[h] << 2;
                inside = inside && t0 >= tb[h] && t0 + 4 <= tb[h] + ncol[h];
            }
            if (xl_any(!inside)) {
                IPX_UNROLL
                for (int h = 0; h < 2; ++h) {
                    uint32_t vm = 0;
                    IPX_UNROLL
                    for (int k = 0; k < 4; ++k)
                        if ((uint32_t)(t0 + k - tb[h]) < (uint32_t)ncol[h]) vm |= 0xFFu << (8 * (REV ? 3 - k : k));
                    x[h] = (x[h] & vm) | (0x14141414u & ~vm);
                }
            }
            pairA = pk_perm(x[1], x[0], REV ? 0x06020703u : 0x05010400u);
            pairB = pk_perm(x[1], x[0], REV ? 0x04000501u : 0x07030602u);
        }
        // the four steps of the group, unrolled: which pair and which bytes of it a step takes are compile-time, and the
        // hand-over registers (Hl_old / Hl_cur) rotate by renaming
        // The score-table words of a step are looked up ONE STEP AHEAD (inside a group): the letters move down the lanes independently of
        // the DP, so the two LDS reads of step u + 1 are issued before the stripe of step u and have arrived when it is done.
        // -- the first lane's column is t; lane l takes over lane l-1's letters of the step before
        // (one v_perm_b32 with a per-lane selector: the first lane picks its two bytes of the pair, the others the lane above's letters)
        let = pk_perm(skew_shr1<W>(let), pairA, lsel0);
        uint32_t tabn0 = *(const uint32_t *)(lds + (let & 0xFFu)), tabn1 = *(const uint32_t *)(lds + ((let >> 8) & 0xFFu));
        IPX_UNROLL
        for (int u = 0; u < 4; ++u) {
            const int t = t0 + u;
            const uint32_t tab0 = tabn0, tab1 = tabn1;
            if (u < 3) {
                let = pk_perm(skew_shr1<W>(let), ((u + 1) & 2) ? pairB : pairA, ((u + 1) & 1) ? lsel1 : lsel0);
                tabn0 = *(const uint32_t *)(lds + (let & 0xFFu));
                tabn1 = *(const uint32_t *)(lds + ((let >> 8) & 0xFFu));
            }
            // -- what the lane above passes on: the diagonal H (two steps old), F and the column maximum so far (one step old)
            const pk16 vH = skew_shr1<W>(Hl_old) & nz;
            pk16 vF = skew_shr1<W>(vFend) & nz;
            // The column maximum starts from what the lanes above found in this column, so that after the stripe it is the
            // maximum over the lanes up to this one.  The lane's best below therefore also covers the rows ABOVE its own in the
            // columns it has processed: harmless -- the lane that owns such a row records the same value at the same column, so
            // neither the best, nor its first column, nor (smallest row wins) the row found in the snapshots changes.
            pk16 cmx = skew_shr1<W>(pm) & nz;
            dp_stripe_f16<SMAX>(H, E, SEL, vF, cmx, vH, tab0, tab1, go, ge);
            vFend = vF;
            Hl_old = Hl_cur;
            Hl_cur = H[SA - 1];
            pm = cmx;
            if (!REV && l == W - 1 && t >= W - 1) {              // column t-7 is complete
                // (both addresses spelled out -- LDS behind the table, or this block's scratch: a step then costs no address arithmetic)
                const uint32_t idx = (uint32_t)((t - (W - 1)) * G + g);
                if (mc_lds) ((uint32_t *)(IPX_LDS_BASE + 128))[idx] = pm; else store_global_u32(maxcol + idx, pm);
            }
            // -- this lane's best (ssw.c:521-539, per lane; non-negative halves order like integers)
            const pk16 nb = pk_max(lbest, cmx);
            const pk16 dif = nb ^ lbest;                         // a half that is not 0: strictly better
            lbest = nb;
            if (!REV) {
                // (forward: some lane of the wave improves in nearly every step -- no test, the selects cost less than the branch)
                const pk16 m = pk_nzmask_pos(dif);
                lcol = pk_select(m, ccol, lcol);
                IPX_UNROLL
                for (int j = 0; j < SMAX; ++j) HM[j] = pk_select(m, H[j], HM[j]);
            } else if (xl_any(dif != 0)) {
                // Reverse pass: the score to reach is the maximum of this matrix (it is the forward optimum, and every local
                // alignment inside the prefix rectangle is one of the forward matrix), so the one improvement whose column and
                // H values are ever looked at is the one that reaches it: no bookkeeping for the others.
                const pk16 m = pk_nzmask_pos(dif) & ~pk_nzmask(nb ^ term);
                if (xl_any(m != 0)) {
                    lcol = pk_select(m, ccol, lcol);
                    IPX_UNROLL
                    for (int j = 0; j < SMAX; ++j) HM[j] = pk_select(m, H[j], HM[j]);
                }
            }
            ccol = pk_add(ccol, 0x00010001u);
        }
        // the next group's letters have arrived (requested a group ago); request those of the group after it.  (At the END of the body,
        // unconditionally: requested at the top under "not the first group", the compiler copied the loaded word into the loop's
        // register at once and every group waited for its own request.)
        IPX_UNROLL
        for (int h = 0; h < 2; ++h) cur[h] = nxt[h];
        if (!WINLDS) IPX_VMEM_FENCE();
        IPX_UNROLL
        for (int h = 0; h < 2; ++h) {
            // (unsigned, one minimum: the word index costs the vector ALU a v_min and the 64-bit address add, nothing else)
            int kr = (idx0[h] >> 2) - ((t0 >> 2) + 2);
            if (kr < 0) kr = 0;
            const uint32_t k = REV ? (uint32_t)kr : (uint32_t)((t0 >> 2) + 2);
            const uint32_t kk = k < (uint32_t)kmax[h] ? k : (uint32_t)kmax[h];
            if (WINLDS) nxt[h] = (winlds + (size_t)(2 * g + h) * (size_t)winstride)[kk < (uint32_t)winstride ? kk : (uint32_t)winstride - 1u];
            else nxt[h] = load_global_u32(refw[h] + kk);
        }
    }

    // ---- finalisation ---------------------------------------------------------------------------
    IPX_SYNC();   // column maxima written by the last lane of each group are visible to the group
    IPX_COMPILER_FENCE();
    IPX_UNROLL
    for (int h = 0; h < 2; ++h) {
        const int slot = 2 * g + h;
        job[h] = -1; L[h] = 0; ncol[h] = 0; score1[h] = 0; rend1[h] = -1; idx0[h] = 3;
        if (slot < cnt) {
            const int64_t jb = (int64_t)p.perm[first + slot];
            job[h] = jb;
            if (!REV) {
                L[h] = (int)(b.read_off[jb + 1] - b.read_off[jb]);
                ncol[h] = b.ref_len[b.ref_id[jb]];
            } else {
                const IpxResult r = b.res[jb];
                L[h] = r.read_end1 + 1; if (L[h] < 0) L[h] = 0;
                score1[h] = r.score1;
                rend1[h] = r.read_end1;
                if (r.ref_end1 + 1 > 0) idx0[h] = r.ref_end1 | 3;
            }
        }
    }
    const pk16 bestA = group_max<W>(lbest);
    const pk16 isb = ~pk_nzmask(lbest ^ bestA);                                  // lanes holding the read's best ...
    const pk16 cminA = group_minu<W>(pk_select(isb, lcol, 0x7FFF7FFFu));          // ... the first column any of them has it in ...
    const pk16 isc = isb & ~pk_nzmask(lcol ^ cminA);                             // ... and the lanes that have it there
    IPX_UNROLL
    for (int h = 0; h < 2; ++h) {
        // end position on the read: smallest striped row holding `best` in that column (ssw.c:545-556), counted from the read's
        // first row (ROW SHIFT: the rows above it hold 0, which is the best only when nothing scored -- row 0 then, as in the reference)
        const unsigned bh = (bestA >> (16 * h)) & 0xFFFFu;
        const bool mine = ((isc >> (16 * h)) & 0xFFFFu) != 0;
        const int shift = W * S - (BH ? 16 * ((L[h] + 15) >> 4) : 8 * ((L[h] + 7) >> 3));
        uint32_t rmin = 0x7FFFFFFFu;
        IPX_UNROLL
        for (int j = SMAX - 1; j >= 0; --j)
            if (mine && (((HM[j] >> (16 * h)) & 0xFFFFu) == bh)) rmin = (uint32_t)(j + l * S);
        rmin = group_umin<W>(rmin);
        int rrow = (int)rmin - shift;
        if (rrow < 0) rrow = 0;
        int end_read = L[h] - 1;
        if (rrow < end_read) end_read = rrow;
        const int cfirst = (int)((cminA >> (16 * h)) & 0xFFFFu);
        const int eref = bh == 0 ? (BH ? -1 : 0) : (REV ? idx0[h] - cfirst : cfirst);   // (never improved: the initial 0 / -1, ssw.c:427 / 220)
        const unsigned bv = ipx_f16_to_uint(bh);
        if (REV && l == 0 && job[h] >= 0 && bv != (unsigned)score1[h]) atomic_or_u32(b.status, IPX_STATUS_INTERNAL);   // (cannot happen: see the column loop)

        if (!REV) {
            // second best outside the mask window (ssw.c:568-581)
            const int refLen = ncol[h];
            const int maskLen = job[h] >= 0 ? mask_len_of(b, job[h], L[h]) : 15;
            int edgeL = eref - maskLen; if (edgeL < 0) edgeL = 0;
            int edgeR = eref + maskLen; if (edgeR > refLen) edgeR = refLen;
            if (BH) edgeR += 1;                                   // ssw.c:374
            uint32_t key2 = 0xFFFFu;                              // (score2 = 0, ref_end2 = 0)
            uint32_t cbig = 0x7FFFFFFFu;                          // (BH = 2) first column holding a value >= 128
            for (int col = l; col < refLen; col += W) {
                const uint32_t v = (maxcol[col * G + g] >> (16 * h)) & 0xFFFFu;
                if (BH == 2 && v >= 0x5800u && cbig == 0x7FFFFFFFu) cbig = (uint32_t)col;      // (0x5800 = 128.0; non-negative halves order like integers)
                if (col < edgeL || col >= edgeR) {
                    const uint32_t kk = (v << 16) | (0xFFFFu - (uint32_t)col);
                    if (v > (key2 >> 16)) key2 = kk;
                }
            }
            key2 = group_umax<W>(key2);
            if (BH == 2) cbig = group_umin<W>(cbig);
            int key = -1;                                         // pass the job takes next (plan_note below)
            if (l == 0 && job[h] >= 0) {
                IpxResult r = b.res[job[h]];
                const uint16_t s2 = (uint16_t)(maskLen >= 15 ? ipx_f16_to_uint(key2 >> 16) : 0u);         // ssw.c:864-870
                const int e2 = maskLen >= 15 ? (int)(0xFFFFu - (key2 & 0xFFFFu)) : -1;
                if (BH == 1) {
                    // the record holds the lower-bound stage's outputs: equal outputs certify them (k_dp_pass STAGE); an upper
                    // bound that reaches the overflow threshold certifies nothing
                    const bool same = bv < (unsigned)(255 - b.bias) && r.score1 == (uint16_t)bv && r.ref_end1 == eref && r.read_end1 == end_read &&
                                      r.score2 == s2 && r.ref_end2 == e2;
                    r.mode = same ? IPX_MODE_BYTE : IPX_MODE_NEED_BYTE_EXACT;
                    if (same) r.read_begin1 = -1;
                } else {
                    r.score1 = (uint16_t)bv;
                    r.ref_end1 = eref;
                    r.read_end1 = end_read;
                    r.read_begin1 = -1;
                    r.score2 = s2;
                    r.ref_end2 = e2;
                    if (BH == 2) {
                        // plain recurrence first (see BH = 2 above): below 128 nothing can reach the signed compare and the
                        // result is the reference's; at the overflow threshold it says nothing; in between it is the candidate
                        // k_prove_plain certifies -- the second-best column needs no proof when it was processed before any
                        // value >= 128 existed (or is the initial 0)
                        if (bv >= (unsigned)(255 - b.bias)) r.mode = b.exact_direct ? IPX_MODE_NEED_BYTE_EXACT : IPX_MODE_NEED_BYTE_LOW;
                        else if (bv < 128u) r.mode = rev_needed(b, bv) ? IPX_MODE_BYTE_PLAIN : IPX_MODE_BYTE;
                        else r.mode = (s2 == 0 || (e2 >= 0 && (uint32_t)e2 < cbig)) ? IPX_MODE_NEED_FWD_PROOF : IPX_MODE_NEED_FWD_PROOF2;
                    } else r.mode = pass == IPX_PASS_WORD_FIRST ? IPX_MODE_WORD_UNPROVEN : IPX_MODE_WORD;
                }
                b.res[job[h]] = r;
                key = next_pass_key(b, r, L[h], false);
            }
            plan_note(b, key);
        } else {
            if (l == 0 && job[h] >= 0) {
                IpxResult r = b.res[job[h]];
                r.ref_begin1 = eref;                                                                       // ssw.c:885
                r.read_begin1 = rend1[h] - end_read;                                                       // ssw.c:886
                if ((unsigned)score1[h] > bv) r.flag = 2;                                                  // ssw.c:888-891
                // (BH = 2) the plain reverse recurrence: final below 128, otherwise k_prove_plain certifies the begin cell
                if (BH == 2) r.mode = score1[h] < 128 ? IPX_MODE_BYTE : IPX_MODE_NEED_REV_PROOF;
                b.res[job[h]] = r;
            }
        }
    }
}

// prologue shared by the wavefront kernels: the score table in LDS (offset 0: [6 window letters][4 read letters], the high byte of each score
// as a half, looked up by byte offset = letter x 4), this block's column maxima, the lane-0 mask
#define IPX_SKEW_PROLOGUE                                                                                                             \
    constexpr int G = 64 / W, NA = 2 * G;                                                                                             \
    const int lane = lane_id();                                                                                                       \
    const int l = lane % W;                                                                                                           \
    unsigned char *lds = IPX_LDS_BASE;                                                                                                \
    const bool mc_lds = !REV && (pass & IPX_PASS_MC_LDS) != 0;                                                                        \
    pass &= 0xFF;                                                                                                                     \
    uint32_t *maxcol = mc_lds ? (uint32_t *)(lds + 128) : b.maxcol_scratch + (size_t)IPX_BID * (size_t)(G * maxcols);                 \
    if (lane < 24) {                                                                                                                  \
        const int v = lane < 20 ? b.mat[(lane >> 2) * 5 + (lane & 3)] : -2048;                                                        \
        ((int8_t *)lds)[lane] = (int8_t)((v == -2048 ? 0xE800u : ipx_f16_from_int(v)) >> 8);                                          \
    }                                                                                                                                 \
    IPX_SYNC();                                                                                                                       \
    uint32_t nz = l == 0 ? 0u : 0xFFFFFFFFu;       /* what arrives from the lane above: nothing, in the first lane */                 \
    IPX_KEEP_VGPR(nz);                             /* (kept a register: v_and_b32 costs 2 cycles, the v_cndmask_b32 on a lane mask the compiler prefers 4) */

// W (r04): lanes per read.  8 = the throughput form (16 reads per wave, 8 * SMAX rows).  32 = the LATENCY TIER for small batches (up to
// IPX_LAT_MAX_JOBS jobs: one locus of indelPost is a few hundred to two thousand alignments, varaln.pyx:112, localn.pyx:47-66): the chip has
// more SIMDs than such a batch has 16-read tiles, so a pass lasts as long as ONE wave needs for its tile -- columns x instructions per step.
// With 32 lanes per read a lane holds a quarter of the segments (32 * SMAX rows, SMAX = ceil(rows / 32)): a step is ~9.5 * SMAX + 22
// instructions instead of ~9.5 * 4 * SMAX + 22, for 31 instead of 7 steps of wavefront lead-in: 150 bp against 300 bp, 332 x 70 instead of
// 307 x 205 instructions per tile.  Four reads per wave; the lane-to-lane hand-over is a whole-wavefront DPP shift (wave_shr:1) instead
// of a row shift; the per-read reductions of the finalisation add two ds_bpermute steps.  Same recurrence, same outputs (ROW SHIFT serves
// any read whose padded row count fits).
// (16 lanes per read for the LONG classes of a big batch -- 26..32 segments at 8 lanes are 4 x 32 state registers, two waves per SIMD, 77 % of
//  the issue ceiling -- was built and measured in r04: 13..16 segments, ~145 registers, three waves, no spills; the forward kernel of 250 bp
//  reads took 10.4 instead of 12.2 ms per million, and the BATCH got slower (250 bp alone: 35.4 against 36.8 M aln/s; config 4: 54.6-55.2
//  against 55.0-55.6): three resident DP waves per SIMD leave the other streams' latency-bound kernels even less room.  Taken out again.)
template <int SMAX, bool REV, int BH = 0, int W = 8>
IPX_KERNEL_WAVE_OCC(W == 8 ? ipx_skew_waves(SMAX, REV) : 1) void k_dp_skew(IpxBatch b, IpxPlan p, int cls, int maxcols, int pass)
{
    IPX_SKEW_PROLOGUE
    for (uint32_t want = (uint32_t)IPX_BID;; want += (uint32_t)IPX_GDIM) {
        if (p.tile_off[cls] + want >= p.tile_off[cls + 1]) break;
        const uint32_t first = p.cls_off[cls] + want * NA;
        const uint32_t avail = p.cls_off[cls + 1] - first;
        if (W >= 32) {
            // (latency tier) window letters of the tile's reads staged behind the score table and the column maxima: winstride words per read
            const int winstride = (maxcols + 3) / 4 + 4;
            uint32_t *winlds = (uint32_t *)(lds + 128 + (mc_lds ? G * maxcols * 4 : 0));
            IPX_SYNC();                                             // (the previous tile's finalisation has read its column maxima and windows)
            dp_skew_tile<SMAX, REV, BH, W>(b, p, first, avail < (uint32_t)NA ? (int)avail : NA, pass, maxcol, mc_lds, lds, nz, winlds, winstride);
        } else
        dp_skew_tile<SMAX, REV, BH, W>(b, p, first, avail < (uint32_t)NA ? (int)avail : NA, pass, maxcol, mc_lds, lds, nz);
    }
}

// (r03 had k_dp_skew_tier here: the tile bodies of one occupancy tier's classes as CALLS from one launch.  r04 took it out: every call
//  spilled the callee-saved half of the body's registers to scratch -- 288-368 bytes per lane, a quarter of a gigabyte per launch, what the
//  r03 profile showed as WRITE_SIZE -- and with the launches sized from the previous run's tile counts the per-class launches are as fast:
//  config 4 55.8 with the tiers, 56.3 without; config 5 72.7 / 73.5.)
// ------------------------------------------------------------------------------------------------
// k_dp_band_rev<S> (r04): the 16-bit REVERSE pass as a band, one LANE per pair of reads.
//
// The reverse pass (ssw.c:875-886) exists to find where score1 is reached walking back from (read_end1, ref_end1): its outputs are the
// first column (in its own, reversed order) whose maximum equals score1 and the smallest row holding it there.  With i / j counting rows /
// columns from the end cell, a local alignment that starts there, covers at most `rows` read bases and drifts d diagonals away from i = j
// contains gaps of d letters at least and scores at most max(mat) * rows - (gap_open + (d - 1) * gap_ext).  So with
//     budget = max(mat) * rows - score1,     d_max = budget < gap_open ? 0 : (budget - gap_open) / gap_ext + 1        (gap_ext >= 1)
// no cell with |i - j| > d_max holds score1, and no optimal path to a cell that does leaves that band (each of its prefixes is under the
// same budget).  A recurrence restricted to |i - j| <= D >= d_max (everything outside reads 0) is cell by cell <= the full one and equal
// along every optimal path: its column maxima reach score1 in the same column, in the same rows -- the same two outputs.  (score1 IS the
// maximum of the full reverse matrix: the forward optimum lies inside the prefix rectangle.  Needs what k_dp_skew needs: gap_open >
// gap_ext, scores exact in halves.)  Jobs whose budget asks for more than D = S + 1 (ipx_band_d) keep the full wavefront kernel (k_rev_split makes the
// two lists).
// Layout: lanes run in lockstep, so a band pays only where a lane walks its OWN column window.  A lane holds TWO reads (the halves of its
// packed registers) and takes their rows in eight blocks of S, one after the other: block q = rows qS .. qS+S-1 over the columns
// qS - D .. qS + S - 1 + D (+ what rounds the count up to four), with the striped recurrence of k_dp_skew (dp_stripe_f16: F runs down the
// block's rows in registers).  What block q + 1 needs of block q -- H and F of its last row, one value each per column of the overlap --
// waits in a ring in LDS ([column][lane]: two 8-byte accesses per step, never a bank conflict); no lane talks to another, there is no
// wavefront skew and no lead-in.  8 x (S + 2 D + 1) steps per 128 reads instead of (columns + 7) per 16: at S = 19, D = 20 a reverse pass
// is ~770 vector instructions per alignment instead of 2 149.  Window letters: two aligned dwords per half and group of four columns,
// one group ahead, cut to the four letters with v_alignbyte_b32; columns outside the window get the letter that scores -2048.
// Dynamic LDS: 128 B score table | ring of (WW - S + 1) x 64 lanes x 8 B
// ------------------------------------------------------------------------------------------------
// half-width of the band of class S: errors, and with them the budget, grow with the read's length (2b, 150 bp, D = 20: 72 % of the jobs eligible)
#ifndef IPX_BAND_DX
#define IPX_BAND_DX 1
#endif
IPX_HD constexpr int ipx_band_d(int S) { return S + IPX_BAND_DX; }
IPX_HD constexpr int ipx_band_steps(int S) { return (S + 2 * ipx_band_d(S) + 1 + 3) & ~3; }
static inline int ipx_band_lds_bytes(int S) { return 128 + (ipx_band_steps(S) - S + 1) * 64 * 8; }
IPX_DEV bool band_rev_ok(const IpxBatch &b, const IpxResult &r, int gO, int gE, int D)
{
    const int rows = r.read_end1 + 1;
    if (gE < 1 || gO <= gE || rows < 1 || r.score1 == 0 || r.ref_end1 < 0) return false;
    const int budget = b.max_match * rows - (int)r.score1;
    const int dmax = budget < gO ? 0 : (budget - gO) / gE + 1;
    return dmax <= D;
}
#if IPX_AUX_KERNELS
// the jobs of class `cls` of the reverse pass's list, sorted into those the band serves (listA) and the others (listB); cnt[0..1]: their numbers
IPX_KERNEL_WAVE void k_rev_split(IpxBatch b, IpxPlan p, int cls, int D, uint32_t *listA, uint32_t *listB, uint32_t *cnt)
{
    IPX_RAISE_PRIO(b);
    constexpr int R = 8;                                            // jobs per lane and round of the wave: ONE pair of atomics per 512 jobs
    const int lane = lane_id();
    const uint32_t lo = p.cls_off[cls], n = p.cls_off[cls + 1] - lo;
    const uint64_t below = (1ull << lane) - 1ull;
    for (uint32_t base = (uint32_t)IPX_BID * (64u * R); base < n; base += (uint32_t)IPX_GDIM * (64u * R)) {      // (uniform)
        uint32_t job[R], pa[R], pb[R];
        bool a[R], in[R];
        uint32_t na = 0, nb = 0;
        IPX_UNROLL
        for (int k = 0; k < R; ++k) {
            const uint32_t at = base + (uint32_t)(k * 64 + lane);
            in[k] = at < n; a[k] = false; job[k] = 0;
            if (in[k]) {
                job[k] = p.perm[lo + at];
                a[k] = band_rev_ok(b, b.res[job[k]], b.gap_open[job[k]], b.gap_ext[job[k]], D);
            }
            const uint64_t ma = xl_ballot(in[k] && a[k]), mb = xl_ballot(in[k] && !a[k]);
            pa[k] = na + (uint32_t)__builtin_popcountll(ma & below); pb[k] = nb + (uint32_t)__builtin_popcountll(mb & below);
            na += (uint32_t)__builtin_popcountll(ma); nb += (uint32_t)__builtin_popcountll(mb);
        }
        uint32_t ba = 0, bb = 0;
        if (lane == 0) {
            if (na) ba = atomic_add_u32(&cnt[0], na);
            if (nb) bb = atomic_add_u32(&cnt[1], nb);
        }
        ba = xl_first(ba); bb = xl_first(bb);
        IPX_UNROLL
        for (int k = 0; k < R; ++k) {
            if (in[k] && a[k]) listA[ba + pa[k]] = job[k];
            if (in[k] && !a[k]) listB[bb + pb[k]] = job[k];
        }
    }
}
#endif // IPX_AUX_KERNELS
// BH = 2: the plain recurrence in the 8-bit dialect (IPX_PASS_BYTE_REV_PLAIN: the reverse pass of reads whose 8-bit forward result equals the plain
// recurrence's, k_dp_skew BH = 2) -- the same band, S = 2 x the 8-bit class; afterwards the begin cell is final below 128 and k_prove_plain's otherwise
template <int S, int BH = 0>
IPX_KERNEL_WAVE_OCC(2) void k_dp_band_rev(IpxBatch b, const uint32_t *listA, const uint32_t *cnt, uint32_t *cls_off_b, uint32_t *tile_off_b, int cls)
{
    constexpr int D = ipx_band_d(S), WW = ipx_band_steps(S), NR = WW - S + 1;
    const int lane = lane_id();
    unsigned char *lds = IPX_LDS_BASE;
    if (lane < 24) {
        const int v = lane < 20 ? b.mat[(lane >> 2) * 5 + (lane & 3)] : -2048;
        ((int8_t *)lds)[lane] = (int8_t)((v == -2048 ? 0xE800u : ipx_f16_from_int(v)) >> 8);
    }
    if (IPX_BID == 0 && lane == 0) {
        // the full kernel's launch that follows walks listB as class `cls` of a plan of its own: 16-job tiles
        cls_off_b[cls] = 0; cls_off_b[cls + 1] = cnt[1];
        tile_off_b[cls] = 0; tile_off_b[cls + 1] = (cnt[1] + 15u) / 16u;
    }
    IPX_SYNC();
    uint2 *ring = (uint2 *)(lds + 128) + lane;                                   // [column of the overlap][lane]
    const uint32_t nA = cnt[0];
    for (uint32_t base = (uint32_t)IPX_BID * 128u; base < nA; base += (uint32_t)IPX_GDIM * 128u) {    // (uniform)
        int64_t job[2];
        int L[2], e1[2], kmax[2], score1[2], gO[2], gE[2];
        const int8_t *rd[2];
        const uint32_t *refw[2];
        IPX_UNROLL
        for (int h = 0; h < 2; ++h) {
            const uint32_t slot = base + 2u * (uint32_t)lane + (uint32_t)h;
            job[h] = -1; L[h] = 0; e1[h] = -1; kmax[h] = 0; score1[h] = 0; gO[h] = 1; gE[h] = 0;
            rd[h] = b.reads; refw[h] = (const uint32_t *)b.refs_packed;
            if (slot < nA) {
                const int64_t jb = (int64_t)listA[slot];
                const IpxResult r = b.res[jb];
                const int rid = b.ref_id[jb];
                job[h] = jb;
                L[h] = r.read_end1 + 1;
                e1[h] = r.ref_end1;
                score1[h] = r.score1;
                gO[h] = b.gap_open[jb]; gE[h] = b.gap_ext[jb];
                rd[h] = b.reads + b.read_off[jb];
                refw[h] = (const uint32_t *)(b.refs_packed + b.refp_off[rid]);
                kmax[h] = ((b.ref_len[rid] + 3) >> 2) + 1;
                if (!band_rev_ok(b, r, gO[h], gE[h], D) || L[h] > 8 * S || L[h] > b.f16_max_len) { atomic_or_u32(b.status, IPX_STATUS_INTERNAL); job[h] = -1; L[h] = 0; e1[h] = -1; }
            }
        }
        const pk16 go = pk_make((int)ipx_f16_from_int(-gO[0]), (int)ipx_f16_from_int(-gO[1]));
        const pk16 ge = pk_make((int)ipx_f16_from_int(-gE[0]), (int)ipx_f16_from_int(-gE[1]));
        const pk16 term = pk_make((int)ipx_f16_from_uint((uint32_t)score1[0]), (int)ipx_f16_from_uint((uint32_t)score1[1]));
        const pk16 live = pk_make(job[0] >= 0 ? -1 : 0, job[1] >= 0 ? -1 : 0);
        int bcol[2] = {0x7FFFFFFF, 0x7FFFFFFF}, brow[2] = {0, 0};
        // four letters (x 4, bytes in column order) of the group of columns starting at j0, for half h; 0x14 where there is no column
        auto group_words = [&](int h, int j0, uint32_t &w0, uint32_t &w1) {
            const int a0 = e1[h] - j0 - 3;
            int wi = a0 >> 2;
            const int top = kmax[h] > 0 ? kmax[h] - 1 : 0;          // (a half without a job: word 0 of the packed windows)
            const int i0 = wi < 0 ? 0 : (wi > top ? top : wi), i1 = wi + 1 < 0 ? 0 : (wi + 1 > top ? top : wi + 1);
            w0 = load_global_u32(refw[h] + i0);
            w1 = load_global_u32(refw[h] + i1);
        };
        auto group_letters = [&](int h, int j0, uint32_t w0, uint32_t w1) -> uint32_t {
            const int a0 = e1[h] - j0 - 3;
            const uint32_t four = xl_alignbyte(w1, w0, (uint32_t)a0 & 3u);     // byte q = letter at window position a0 + q = column j0 + 3 - q
            uint32_t x = pk_perm(0u, four << 2, 0x00010203u);                   // ... x 4, byte k = column j0 + k
            uint32_t vm = 0;
            IPX_UNROLL
            for (int k = 0; k < 4; ++k)
                if ((uint32_t)(j0 + k) <= (uint32_t)e1[h] && e1[h] >= 0) vm |= 0xFFu << (8 * k);
            return (x & vm) | (0x14141414u & ~vm);
        };
        for (int q = 0; q < 8; ++q) {
            {   // rows of this block in no read of the wave: done (padded row counts: 8 * ceil(L / 8))
                const bool more = (job[0] >= 0 && (BH ? 16 * ((L[0] + 15) >> 4) : 8 * ((L[0] + 7) >> 3)) > q * S) ||
                                  (job[1] >= 0 && (BH ? 16 * ((L[1] + 15) >> 4) : 8 * ((L[1] + 7) >> 3)) > q * S);
                if (!xl_any(more)) break;
            }
            pk16 SEL[S], H[S], E[S];
            {
                int raw[2][S];
                IPX_UNROLL
                for (int h = 0; h < 2; ++h) {
                    IPX_UNROLL
                    for (int j = 0; j < S; ++j) {
                        const int r = q * S + j;
                        int idx = L[h] - 1 - (r < L[h] ? r : L[h] - 1);
                        if (idx < 0) idx = 0;
                        raw[h][j] = load_stream_i8(L[h] > 0 ? rd[h] + idx : (const int8_t *)b.read_off);
                    }
                }
                IPX_UNROLL
                for (int j = 0; j < S; ++j) {
                    uint32_t sel = 0;
                    IPX_UNROLL
                    for (int h = 0; h < 2; ++h) {
                        const unsigned bs = (unsigned)raw[h][j];
                        uint32_t sh = 0x0c0cu;
                        if (q * S + j < L[h] && bs < 4u) sh = 0x000cu | ((bs + 4u * h) << 8);
                        sel |= sh << (16 * h);
                    }
                    SEL[j] = sel; H[j] = 0; E[j] = 0;
                }
            }
            const int jbase = q * S - D;                                         // first column of the block
            pk16 hit = 0;                                                        // halves that have found score1 in this block
            int hcol[2] = {0, 0}, hrow[2] = {0, 0};
            uint32_t cw[2][2], nw[2][2];
            IPX_UNROLL
            for (int h = 0; h < 2; ++h) { group_words(h, jbase, cw[h][0], cw[h][1]); group_words(h, jbase + 4, nw[h][0], nw[h][1]); }
            // ring entries u and u + 1 (H of the block above's last row one column back; F leaving that row in this column), fetched a step ahead
            uint2 r0 = q > 0 ? ring[0] : uint2{0u, 0u}, r1 = q > 0 ? ring[64] : uint2{0u, 0u};
            uint32_t pairA, pairB, tabn0, tabn1;
            {
                const uint32_t x0 = group_letters(0, jbase, cw[0][0], cw[0][1]), x1 = group_letters(1, jbase, cw[1][0], cw[1][1]);
                pairA = pk_perm(x1, x0, 0x05010400u); pairB = pk_perm(x1, x0, 0x07030602u);    // (half 0, half 1) of columns 0,1 | 2,3
                tabn0 = *(const uint32_t *)(lds + (pairA & 0xFFu)); tabn1 = *(const uint32_t *)(lds + ((pairA >> 8) & 0xFFu));
            }
            for (int u0 = 0; u0 < WW; u0 += 4) {
                // the next group's letters (their words arrived a group ago), and the request for the group after it
                IPX_UNROLL
                for (int h = 0; h < 2; ++h) { cw[h][0] = nw[h][0]; cw[h][1] = nw[h][1]; group_words(h, jbase + u0 + 8, nw[h][0], nw[h][1]); }
                const uint32_t y0 = group_letters(0, jbase + u0 + 4, cw[0][0], cw[0][1]), y1 = group_letters(1, jbase + u0 + 4, cw[1][0], cw[1][1]);
                const uint32_t nextA = pk_perm(y1, y0, 0x05010400u), nextB = pk_perm(y1, y0, 0x07030602u);
                IPX_UNROLL
                for (int k = 0; k < 4; ++k) {
                    const int u = u0 + k;
                    const uint32_t tab0 = tabn0, tab1 = tabn1;
                    {   // the score-table words of the next step, looked up before this step's stripe
                        const uint32_t letn = k == 3 ? nextA : (((k + 1) & 2) ? pairB : pairA) >> (((k + 1) & 1) ? 16 : 0);
                        tabn0 = *(const uint32_t *)(lds + (letn & 0xFFu)); tabn1 = *(const uint32_t *)(lds + ((letn >> 8) & 0xFFu));
                    }
                    const pk16 vH = (q > 0 && u < NR) ? r0.x : 0u;
                    pk16 vF = (q > 0 && u + 1 < NR) ? r1.y : 0u, cmx = 0;
                    r0 = r1;
                    if (q > 0 && u + 2 < NR) r1 = ring[(u + 2) * 64];
                    dp_stripe_f16<S>(H, E, SEL, vF, cmx, vH, tab0, tab1, go, ge);
                    if (u >= S - 1) ring[(u - (S - 1)) * 64] = uint2{H[S - 1], vF};
                    const pk16 fresh = ~pk_nzmask(cmx ^ term) & live & ~hit;      // halves that reach score1 here for the first time in this block
                    if (xl_any(fresh != 0)) {
                        IPX_UNROLL
                        for (int h = 0; h < 2; ++h)
                            if ((fresh >> (16 * h)) & 0xFFFFu) {
                                const unsigned th = (term >> (16 * h)) & 0xFFFFu;
                                int jm = S - 1;
                                IPX_UNROLL
                                for (int j = S - 1; j >= 0; --j) if (((H[j] >> (16 * h)) & 0xFFFFu) == th) jm = j;
                                hcol[h] = jbase + u; hrow[h] = q * S + jm;
                            }
                        hit |= fresh;
                    }
                }
                pairA = nextA; pairB = nextB;
            }
            IPX_UNROLL
            for (int h = 0; h < 2; ++h)
                if (((hit >> (16 * h)) & 0xFFFFu) && hcol[h] < bcol[h]) { bcol[h] = hcol[h]; brow[h] = hrow[h]; }   // (the same column in a later block: a larger row)
        }
        IPX_UNROLL
        for (int h = 0; h < 2; ++h)
            if (job[h] >= 0) {
                if (bcol[h] == 0x7FFFFFFF) { atomic_or_u32(b.status, IPX_STATUS_INTERNAL); continue; }   // (cannot happen: the band holds every cell with score1)
                IpxResult r = b.res[job[h]];
                int end_read = L[h] - 1;
                if (brow[h] < end_read) end_read = brow[h];
                r.ref_begin1 = e1[h] - bcol[h];                                                            // ssw.c:885
                r.read_begin1 = r.read_end1 - end_read;                                                    // ssw.c:886
                if (BH == 2) r.mode = score1[h] < 128 ? IPX_MODE_BYTE : IPX_MODE_NEED_REV_PROOF;           // (as k_dp_skew BH = 2)
                b.res[job[h]] = r;
            }
    }
}

// ------------------------------------------------------------------------------------------------
// k_dp_wide<S, REV> (r04): the 16-bit passes (sw_sse2_word, ssw.c:410-586) of reads of 64 striped segments or more -- from 505 bp, up to
// IPX_LONG_MAX_READ -- with gap_open > gap_ext, as ONE wavefront per read: the plain Gotoh recurrence as a wavefront over the lanes
// (k_dp_skew: lane l owns the S consecutive rows l*S .. l*S+S-1 and works on column t - l at step t; what it needs from the lane above
// arrives through whole-wavefront DPP shifts, one or two steps old; no lazy-F, see "Exactness" at k_dp_skew), on all 64 lanes, 64 * S rows,
// S = 16 / 32 / 48 / 64 chosen per read (ROW SHIFT: the rows above the read stay 0).  Scores of reads this long leave the range halves
// hold exactly (2 047), so the cells are unpacked 32-bit integers: H, E and the column saved at the lane's best in registers (3 * S + S
// selectors: a whole SIMD's register file for one wave at S = 64), no saturation needed while readLen * max(mat) <= 32 767 (checked on the
// host: the reference's _mm_adds_epi16 then never saturates either).  The score of a cell is ONE v_perm_b32: the scores of the lane's
// window letter against A, C, G, T sit in bytes 1 and 3 of two table words, and the selector of a row picks its byte and the SIGN selectors
// (8..11: bit 15 / 31 of either source replicated) for the three bytes above it; rows above or below the read and the letter N select the
// constant 0 (the selector profile's condition, mat[.][N] = 0).  The window is staged in LDS once per read, in processing order, with
// IPX_WIDE_PAD letters "5" (no column: scores -128, nothing is recorded there) on either side, so that lane l simply reads byte t - l.
// k_dp_long -- the reference's loops transcribed, state in global memory -- keeps the jobs with gap_open <= gap_ext and the 8-bit passes
// of reads beyond 1 008 bp: 0.3 s per tile of four 2 kb reads against 20 kb there, ~10 ms per read here.
// ------------------------------------------------------------------------------------------------
#define IPX_WIDE_PAD 64
#define IPX_WIDE_MAX_SCORE 32767                 // readLen * max(mat) beyond it: the reference's 16-bit adds could saturate -- k_dp_long's business
static inline int ipx_wide_lds_bytes(int maxcols) { return 64 + ((maxcols + 2 * IPX_WIDE_PAD + 8 + 3) & ~3); }
IPX_HD constexpr int ipx_wide_bucket(int rows) { return rows <= 64 * 16 ? 16 : (rows <= 64 * 32 ? 32 : (rows <= 64 * 48 ? 48 : 64)); }
IPX_DEV int imax(int a, int b) { return a > b ? a : b; }
template <int S, bool REV>
IPX_KERNEL_WAVE_OCC(S <= 16 ? 2 : 1) void k_dp_wide(IpxBatch b, IpxPlan p, int maxcols, int pass)
{
    const int l = lane_id();
    unsigned char *lds = IPX_LDS_BASE;
    uint32_t *tab = (uint32_t *)lds;                                   // [6 window letters][2]: scores against A, C | G, T in bytes 1 and 3
    unsigned char *win = lds + 64;
    if (l < 12) {
        const int c = l >> 1, a = 2 * (l & 1);
        uint32_t w = 0x80008000u;                                      // letter 5 (no column): -128 against everything
        if (c < 5) w = ((uint32_t)(uint8_t)b.mat[c * 5 + a] << 8) | ((uint32_t)(uint8_t)b.mat[c * 5 + a + 1] << 24);
        tab[l] = w;
    }
    uint32_t *maxcol = b.maxcol_scratch + (size_t)IPX_BID * (size_t)(8 * maxcols);
    const int cls = IPX_MAX_SEG;
    const uint32_t njobs = p.cls_off[cls + 1] - p.cls_off[cls];
    for (uint32_t want = (uint32_t)IPX_BID; want < njobs; want += (uint32_t)IPX_GDIM) {
        const int64_t job = (int64_t)p.perm[p.cls_off[cls] + want];
        const int rid = b.ref_id[job];
        const int8_t *rd = b.reads + b.read_off[job];
        const int8_t *rf = b.refs_packed + b.refp_off[rid];
        const int gO = b.gap_open[job], gE = b.gap_ext[job];
        int L, ncol, score1 = 0, rend1 = -1;
        if (!REV) { L = (int)(b.read_off[job + 1] - b.read_off[job]); ncol = b.ref_len[rid]; }
        else {
            const IpxResult r = b.res[job];
            L = r.read_end1 + 1; if (L < 0) L = 0;
            ncol = r.ref_end1 + 1; if (ncol < 0) ncol = 0;
            score1 = r.score1; rend1 = r.read_end1;
        }
        const int rows = 8 * ((L + 7) >> 3);                           // the reference's padded row count (ssw.c:396-398)
        if (ipx_wide_bucket(rows) != S) continue;                      // (another instantiation's read; uniform over the wave)
        if (gO <= gE || rows > 64 * S || ncol > maxcols) { if (l == 0) atomic_or_u32(b.status, IPX_STATUS_INTERNAL); continue; }   // (host-side routing error)
        const int dl = 64 * S - rows;                                  // rows the read is shifted down by
        IPX_SYNC();
        for (int k = l; k < ncol + 2 * IPX_WIDE_PAD + 8; k += 64) {
            const int c = k - IPX_WIDE_PAD;
            win[k] = (uint8_t)((c >= 0 && c < ncol) ? rf[REV ? ncol - 1 - c : c] : 5);
        }
        uint32_t SEL[S];
        IPX_UNROLL
        for (int j = 0; j < S; ++j) {
            const int r = j + l * S - dl;
            uint32_t sel = 0x0c0c0c0cu;
            if (r >= 0 && r < L) {
                const unsigned a = (unsigned)(int)load_stream_i8(rd + (REV ? L - 1 - r : r));
                if (a < 4u) sel = a == 0 ? 0x08080801u : (a == 1 ? 0x09090903u : (a == 2 ? 0x0a0a0a05u : 0x0b0b0b07u));
            }
            SEL[j] = sel;
        }
        int H[S], E[S], HM[S];
        IPX_UNROLL
        for (int j = 0; j < S; ++j) { H[j] = 0; E[j] = 0; HM[j] = 0; }
        int Hl_cur = 0, Hl_old = 0, vFend = 0, pm = 0, lbest = 0, lcol = 0;
        IPX_SYNC();
        int let = win[IPX_WIDE_PAD - l], letn = win[IPX_WIDE_PAD + 1 - l];
        uint32_t X = tab[2 * let], Y = tab[2 * let + 1];
        const int TT = ncol > 0 ? ncol + 63 : 0;
        int tend = -1;
        for (int t = 0; t < TT && (tend < 0 || t <= tend); ++t) {
            const uint32_t Xc = X, Yc = Y;
            const bool valid = let < 5;
            // (the next step's table words and the letter after it: requested before the stripe, arrived after it)
            let = letn;
            X = tab[2 * let]; Y = tab[2 * let + 1];
            letn = win[IPX_WIDE_PAD + t + 2 - l];
            int vH = (int)xl_wave_shr1((uint32_t)Hl_old);
            int vF = (int)xl_wave_shr1((uint32_t)vFend);
            int cmx = (int)xl_wave_shr1((uint32_t)pm);
            IPX_UNROLL
            for (int j = 0; j < S; ++j) {
                int h = vH + (int)pk_perm(Yc, Xc, SEL[j]);
                h = imax(imax(h, E[j]), vF);
                vH = H[j];
                H[j] = h;
                cmx = imax(cmx, h);
                const int tt = h - gO;
                E[j] = imax(imax(E[j] - gE, tt), 0);
                vF = imax(imax(vF - gE, tt), 0);
            }
            vFend = vF;
            Hl_old = Hl_cur;
            Hl_cur = H[S - 1];
            pm = cmx;
            if (!REV && l == 63 && t >= 63) store_global_u32(maxcol + (t - 63), (uint32_t)pm);   // column t-63 is complete
            const bool better = valid && cmx > lbest;
            if (better) lbest = cmx;
            const bool rec = REV ? (better && cmx == score1) : better;
            if (xl_any(rec)) {
                if (rec) lcol = t - l;
                IPX_UNROLL
                for (int j = 0; j < S; ++j) HM[j] = rec ? H[j] : HM[j];
            }
            if (REV && tend < 0 && xl_any(lbest == score1 && lbest > 0)) tend = t + 63;   // every lane has been through that column 63 steps on
        }
        // ---- finalisation (dp_skew_tile's, one read) ----
        IPX_SYNC();
        IPX_COMPILER_FENCE();
        const uint32_t bestA = group_umax<64>((uint32_t)lbest);
        const bool isb = (uint32_t)lbest == bestA;
        const uint32_t cminA = group_umin<64>(isb ? (uint32_t)lcol : 0x7FFFFFFFu);
        const bool mine = isb && (uint32_t)lcol == cminA;
        uint32_t rmin = 0x7FFFFFFFu;
        IPX_UNROLL
        for (int j = S - 1; j >= 0; --j)
            if (mine && (uint32_t)HM[j] == bestA) rmin = (uint32_t)(j + l * S);
        rmin = group_umin<64>(rmin);
        int rrow = (int)rmin - dl;
        if (rrow < 0) rrow = 0;
        int end_read = L - 1;
        if (rrow < end_read) end_read = rrow;
        const int eref = bestA == 0 ? 0 : (REV ? ncol - 1 - (int)cminA : (int)cminA);
        if (REV && l == 0 && bestA != (uint32_t)score1) atomic_or_u32(b.status, IPX_STATUS_INTERNAL);   // (cannot happen: the forward optimum lies inside the prefix rectangle)
        int key = -1;
        if (!REV) {
            const int maskLen = mask_len_of(b, job, L);
            int edgeL = eref - maskLen; if (edgeL < 0) edgeL = 0;
            int edgeR = eref + maskLen; if (edgeR > ncol) edgeR = ncol;
            uint32_t v2 = 0, c2 = 0;                                   // second best outside the mask (ssw.c:568-581): first strict maximum in scan order
            for (int col = l; col < ncol; col += 64)
                if (col < edgeL || col >= edgeR) {
                    const uint32_t v = load_global_u32(maxcol + col);
                    if (v > v2) { v2 = v; c2 = (uint32_t)col; }
                }
            const uint32_t v2A = group_umax<64>(v2);
            const uint32_t c2A = group_umin<64>(v2 == v2A ? c2 : 0x7FFFFFFFu);
            if (l == 0) {
                IpxResult r = b.res[job];
                r.score1 = (uint16_t)bestA;
                r.ref_end1 = eref;
                r.read_end1 = end_read;
                r.read_begin1 = -1;
                r.score2 = (uint16_t)(maskLen >= 15 ? v2A : 0u);                                   // ssw.c:864-870
                r.ref_end2 = maskLen >= 15 ? (int)c2A : -1;
                r.mode = pass == IPX_PASS_WORD_FIRST ? IPX_MODE_WORD_UNPROVEN : IPX_MODE_WORD;
                b.res[job] = r;
                key = next_pass_key(b, r, L, false);
            }
        } else if (l == 0) {
            IpxResult r = b.res[job];
            r.ref_begin1 = eref;                                                                   // ssw.c:885
            r.read_begin1 = rend1 - end_read;                                                      // ssw.c:886
            if ((uint32_t)score1 > bestA) r.flag = 2;                                              // ssw.c:888-891
            b.res[job] = r;
        }
        if (!REV) plan_note(b, key);
    }
}

#if IPX_AUX_KERNELS
// ------------------------------------------------------------------------------------------------
// k_dp_long<W, REV>: sw_sse2_byte (W = 16, ssw.c:197-384) / sw_sse2_word (W = 8, ssw.c:410-586) for reads too long for the
// register-resident kernels (more than IPX_MAX_SEG = 64 striped segments: over 512 bp in the 16-bit pass, over 1 024 bp in the
// 8-bit pass; r03: up to IPX_LONG_MAX_READ = 4 096 bp).  Correctness first, speed not at all: a plain transcription of the
// reference's loop structure.  One read per 16-lane DPP row (4 per wave), one SSE lane per GPU lane (the 16-bit pass uses 8 lanes
// of its row), one unpacked 32-bit value per lane; the striped columns H, E, the column saved at the best score and the read
// letters of a row's segments live in the block's region of a global scratch, [segment][lane] (coalesced); the lazy-F loop is the
// reference's, step by step, every read of the wave at its own (round, segment); column maxima go to the block's column-maxima
// scratch.  Serves the planner's class IPX_MAX_SEG ("64 segments or more", fast- and slow-gap alike) of any pass; the jobs of a
// tile (the pass's tile size) are taken four at a time.  Finalisation and mode logic are k_dp_pass's for the exact stage.
// ------------------------------------------------------------------------------------------------
#define IPX_LONG_MAX_READ 4096
static inline size_t ipx_long_state_bytes(int max_read_len)   // per block: H, E, Hmax (int32) + letters (int8), [segment][64 lanes]
{
    const size_t S = (size_t)((max_read_len + 7) / 8) + 1;
    return S * 64 * 13 + 64;
}
IPX_DEV int group16_or_i(int x)
{
    uint32_t v = (uint32_t)x;
    v |= xl_xor1(v); v |= xl_xor2(v); v |= xl_half_mirror(v); v |= xl_mirror(v);
    return (int)v;
}
IPX_DEV int group16_max_i(int x)
{
    int y;
    y = (int)xl_xor1((uint32_t)x); x = x > y ? x : y;
    y = (int)xl_xor2((uint32_t)x); x = x > y ? x : y;
    y = (int)xl_half_mirror((uint32_t)x); x = x > y ? x : y;
    y = (int)xl_mirror((uint32_t)x); x = x > y ? x : y;
    return x;
}
template <int W, bool REV>
IPX_KERNEL_WAVE void k_dp_long(IpxBatch b, IpxPlan p, int na, int maxcols, int pass, unsigned char *state, int64_t state_stride, int halves)
{
    constexpr bool BYTE = W == 16;
    const int lane = lane_id();
    const int slot = lane >> 4, l = lane & 15;
    const bool lane_on = l < W;                                        // the 16-bit pass uses 8 lanes of its row
    int8_t *matl = (int8_t *)IPX_LDS_BASE;
    if (lane < 25) matl[lane] = b.mat[lane];
    IPX_SYNC();
    unsigned char *st = state + (int64_t)IPX_BID * state_stride;
    const int64_t cap = (state_stride - 64) / (64 * 13);               // segments the region holds
    int32_t *Hs = (int32_t *)st, *Es = Hs + cap * 64, *HMs = Es + cap * 64;
    int8_t *LET = (int8_t *)(HMs + cap * 64);
    uint32_t *maxcol = b.maxcol_scratch + (size_t)IPX_BID * (size_t)(8 * maxcols);   // [column][4 slots]
    const int bias = b.bias;

    for (int half = 0; half < 2; ++half) {
        if (!((halves >> half) & 1)) continue;                         // (bit 0: the fast-gap class, bit 1: the slow-gap class; r04: k_dp_wide may have the first)
        const int cls = IPX_MAX_SEG + half * IPX_SLOW_BASE;
        const uint32_t ntile = p.tile_off[cls + 1] - p.tile_off[cls];
        const uint32_t nsub = (uint32_t)((na + 3) / 4);
        for (uint32_t want = (uint32_t)IPX_BID; want < ntile * nsub; want += (uint32_t)IPX_GDIM) {
            const uint32_t tile = want / nsub, sub = want % nsub;
            const uint32_t first = p.cls_off[cls] + tile * (uint32_t)na + sub * 4u;
            const uint32_t tile_end = p.cls_off[cls] + (tile + 1u) * (uint32_t)na;
            uint32_t end = p.cls_off[cls + 1] < tile_end ? p.cls_off[cls + 1] : tile_end;
            const int cnt = first >= end ? 0 : (end - first < 4u ? (int)(end - first) : 4);
            if (cnt == 0) continue;
            // ---- this row's read ----
            int64_t job = -1;
            int L = 0, ncol = 0, gO = 0, gE = 0, score1 = 0, rend1 = -1, S = 0;
            const int8_t *rd = b.reads, *rf = b.refs_packed;
            if (slot < cnt) {
                job = (int64_t)p.perm[first + slot];
                const int rid = b.ref_id[job];
                rd = b.reads + b.read_off[job];
                rf = b.refs_packed + b.refp_off[rid];
                gO = b.gap_open[job]; gE = b.gap_ext[job];
                if (!REV) { L = (int)(b.read_off[job + 1] - b.read_off[job]); ncol = b.ref_len[rid]; }
                else {
                    const IpxResult r = b.res[job];
                    L = r.read_end1 + 1; if (L < 0) L = 0;
                    ncol = r.ref_end1 + 1; if (ncol < 0) ncol = 0;
                    score1 = r.score1; rend1 = r.read_end1;
                }
                S = (L + W - 1) / W;
            }
            if ((int64_t)S > cap) { if (l == 0) atomic_or_u32(b.status, IPX_STATUS_READ_TOO_LONG); S = 0; L = 0; ncol = 0; job = -1; }
            const int Smax = (int)wave_umax((uint32_t)S);
            const int T = (int)wave_umax((uint32_t)ncol);
            for (int j = 0; j < Smax; ++j) {                           // state and letters of the row's segments
                const int r = j + l * S;
                int a = 5;                                             // 5 = padding row (scores 0 against everything, ssw.c:174-176 / 396-398)
                if (lane_on && j < S && r < L) { a = rd[REV ? L - 1 - r : r]; if ((unsigned)a > 4u) a = 4; }
                LET[j * 64 + lane] = (int8_t)a;
                Hs[j * 64 + lane] = 0; Es[j * 64 + lane] = 0; HMs[j * 64 + lane] = 0;
            }
            for (int c = l; c < T; c += 16) maxcol[c * 4 + slot] = 0;   // maxColumn is calloc'ed (ssw.c:224 / 431)
            IPX_SYNC();
            int best = 0, endref = BYTE ? -1 : 0, Hlast = 0;
            bool done = ncol == 0, ovf = false;
            for (int t = 0; t < T; ++t) {
                if (!xl_any(!done && t < ncol)) break;
                const bool act = !done && t < ncol;
                const int ri = REV ? ncol - 1 - t : t;                 // reference walks the window right to left in the reverse pass (ssw.c:253-257)
                int c = act ? rf[ri] : 0;
                if ((unsigned)c > 4u) c = 4;
                // -- main loop (ssw.c:274-299 / 480-504)
                int vH = (int)xl_row_shr1((uint32_t)Hlast);
                if (l == 0) vH = 0;
                int vF = 0, cmx = 0;
                for (int j = 0; j < Smax; ++j) {
                    if (act && lane_on && j < S) {
                        const int a = LET[j * 64 + lane];
                        const int pp = a < 5 ? matl[c * 5 + a] : 0;
                        int h;
                        if (BYTE) { h = vH + pp + bias; if (h > 255) h = 255; h = h > bias ? h - bias : 0; }
                        else { h = vH + pp; if (h > 32767) h = 32767; if (h < -32768) h = -32768; }
                        int e = Es[j * 64 + lane];
                        if (h < e) h = e;
                        if (h < vF) h = vF;
                        if (cmx < h) cmx = h;
                        vH = Hs[j * 64 + lane];
                        Hs[j * 64 + lane] = h;
                        if (j == S - 1) Hlast = h;
                        const int tt = h > gO ? h - gO : 0;
                        e = e > gE ? e - gE : 0;
                        Es[j * 64 + lane] = e > tt ? e : tt;
                        vF = vF > gE ? vF - gE : 0;
                        if (vF < tt) vF = tt;
                    }
                }
                // -- lazy-F (ssw.c:302-313 / 507-518), step by step; every row of the wave at its own (round, segment)
                {
                    bool fin = !act || S == 0;
                    int kk = 0, jj = 0;
                    while (xl_any(!fin)) {
                        const int sh = (int)xl_row_shr1((uint32_t)vF);
                        if (!fin && jj == 0) vF = l == 0 ? 0 : sh;     // _mm_slli_si128 at the start of a round
                        int vote = 0;
                        if (!fin && lane_on) {
                            int h = Hs[jj * 64 + lane];
                            if (h < vF) h = vF;
                            if (cmx < h) cmx = h;
                            Hs[jj * 64 + lane] = h;
                            if (jj == S - 1) Hlast = h;
                            const int hh = h > gO ? h - gO : 0;
                            vF = vF > gE ? vF - gE : 0;
                            vote = BYTE ? ((int8_t)(uint8_t)vF > (int8_t)(uint8_t)hh) : ((int16_t)(uint16_t)vF > (int16_t)(uint16_t)hh);
                        }
                        const int anyv = group16_or_i(vote);
                        if (!fin) {
                            if (!anyv) fin = true;                     // goto end
                            else if (++jj == S) { jj = 0; if (++kk == W) fin = true; }
                        }
                    }
                }
                // -- column maximum, best score (ssw.c:316-337 / 521-539)
                const int cmA = group16_max_i(lane_on ? cmx : 0);
                if (act) {
                    bool leave = false;
                    if (cmA > best) {
                        best = cmA;
                        if (BYTE && best + bias >= 255) { ovf = true; leave = true; }        // break before recording (ssw.c:327)
                        else {
                            endref = ri;
                            for (int j = 0; j < S; ++j) if (lane_on) HMs[j * 64 + lane] = Hs[j * 64 + lane];
                        }
                    }
                    if (!leave) {
                        if (l == 0) maxcol[ri * 4 + slot] = (uint32_t)cmA;
                        if (REV && cmA == score1) leave = true;                              // maxColumn[i] == terminate
                    }
                    if (leave) done = true;
                }
            }
            IPX_SYNC();
            // ---- finalisation (k_dp_pass, exact stage) ----
            uint32_t rmin = 0x7FFFFFFFu;
            for (int j = S - 1; j >= 0; --j)
                if (lane_on && HMs[j * 64 + lane] == best) rmin = (uint32_t)(j + l * S);
            rmin = group_umin<16>(rmin);
            int end_read = L - 1;
            if ((int)rmin < end_read) end_read = (int)rmin;
            int key = -1;
            if (!REV) {
                const int maskLen = job >= 0 ? mask_len_of(b, job, L) : 15;
                int edgeL = endref - maskLen; if (edgeL < 0) edgeL = 0;
                int edgeR = endref + maskLen; if (edgeR > ncol) edgeR = ncol;
                if (BYTE) edgeR += 1;
                uint64_t key2 = 0xFFFFFFFFull;                                             // (score2 = 0, ref_end2 = 0)
                for (int col = l; col < ncol; col += 16)
                    if (col < edgeL || col >= edgeR) {
                        const uint64_t v = maxcol[col * 4 + slot];
                        if (v > (key2 >> 32)) key2 = (v << 32) | (uint64_t)(0xFFFFFFFFu - (uint32_t)col);
                    }
                {   // first strict maximum in scan order = largest value, then smallest column
                    uint32_t hi = (uint32_t)(key2 >> 32), lo = (uint32_t)key2;
                    const uint32_t him = group_umax<16>(hi);
                    lo = group_umax<16>(hi == him ? lo : 0u);
                    key2 = ((uint64_t)him << 32) | lo;
                }
                if (l == 0 && job >= 0) {
                    IpxResult r = b.res[job];
                    const bool has_word = r.mode == IPX_MODE_NEED_BYTE_CHECK || r.mode == IPX_MODE_NEED_BYTE_EXACT_W;
                    const int s2 = maskLen >= 15 ? (int)(key2 >> 32) : 0;
                    const int e2 = maskLen >= 15 ? (int)(0xFFFFFFFFu - (uint32_t)key2) : -1;
                    if (BYTE && ovf) {
                        if (has_word) r.mode = IPX_MODE_WORD;
                        else if (b.score_size == 2) { r.mode = IPX_MODE_NEED_WORD; r.score1 = 255; }
                        else { r.mode = IPX_MODE_FAIL; r.score1 = 255; }
                    } else {
                        r.mode = BYTE ? IPX_MODE_BYTE : ((pass & 0xFF) == IPX_PASS_WORD_FIRST ? IPX_MODE_WORD_UNPROVEN : IPX_MODE_WORD);
                        r.score1 = (uint16_t)best; r.ref_end1 = endref; r.read_end1 = end_read; r.read_begin1 = -1;
                        r.score2 = (uint16_t)s2; r.ref_end2 = e2;
                    }
                    b.res[job] = r;
                    key = next_pass_key(b, r, L, gO <= gE);
                }
            } else if (l == 0 && job >= 0) {
                IpxResult r = b.res[job];
                const unsigned best_rev = (BYTE && ovf) ? 255u : (unsigned)best;
                r.ref_begin1 = endref;                                                     // ssw.c:885
                r.read_begin1 = rend1 - end_read;                                          // ssw.c:886
                if ((unsigned)score1 > best_rev) r.flag = 2;                               // ssw.c:888-891
                b.res[job] = r;
            }
            if (!REV) plan_note(b, key);
            IPX_SYNC();
        }
    }
}

// ------------------------------------------------------------------------------------------------
// k_prove_overflow: the reference always runs the 8-bit pass first and only then the 16-bit pass
// (ssw.c:842-847).  For reads that will almost surely overflow we run the 16-bit pass first
// (IPX_PASS_WORD_FIRST) and then try to PROVE, without the 8-bit pass, that it overflows: every cell of
// the 8-bit matrix is >= the ungapped local score of its diagonal (H = max(diag + P, E, F) >= diag + P
// and >= 0, up to the first saturation), so if the best ungapped run on the diagonal through the
// 16-bit end point reaches 255-bias, the 8-bit pass is certain to leave with 255 and the 16-bit
// result is the answer.  One lane per read, O(readLen).
//
// Reads that fail the ungapped test (an indel splits their matches) get a GAPPED lower bound over a
// band of IPX_PROVE_BAND diagonals around the end diagonal, built only from moves whose value the 8-bit
// pass is certain to contain whatever its lazy-F loop does (ssw.c:302-313 leaves early on a signed
// compare, so a vertical gap that crosses a segment boundary may be lost there):
//   diagonal steps; horizontal gaps (E is updated from the main loop's H, ssw.c:286-291); vertical gaps
//   that stay inside one segment of the striped layout, i.e. never INTO a row r with r % segLen == 0
//   (inside a segment F comes from the main loop itself, ssw.c:294-296).
// Cells outside the band or the window count as 0, which only lowers the bound.  If the bound reaches
// 255-bias the 8-bit pass overflows.  Still-unproven reads take the 8-bit pass as usual.
// ------------------------------------------------------------------------------------------------
#define IPX_PROVE_BAND 15
#define IPX_PROVE_BAND_NARROW 5                 // r03: a band of 5 diagonals first (most reads with an indel stay within 2 of the end diagonal), 15 for the rest
#define IPX_PROVE_CHUNK 4                       // blocks of 64 consecutive jobs whose still-unproven reads share the band rounds
// LDS: 64 B score columns | narrow-band queue (64 * IPX_PROVE_CHUNK + 64 jobs) | wide-band queue (128 jobs) | read staging (lds_cap bytes)
#define IPX_PROVE_QA (64 * IPX_PROVE_CHUNK + 64)
#define IPX_PROVE_QB 128
static inline int ipx_prove_lds_bytes(int lds_cap) { return 64 + 4 * (IPX_PROVE_QA + IPX_PROVE_QB) + lds_cap; }

// ungapped test: best run on the diagonal through the 16-bit end point (rd = the read's letters)
IPX_DEV bool prove_ungapped(const IpxBatch &b, const IpxResult &r, const int8_t *rd, const int8_t *rf)
{
    const int m = r.ref_end1 < r.read_end1 ? r.ref_end1 : r.read_end1;   // cells on the diagonal up to the end point
    const int cap = 255 - b.bias;
    int u = 0;
    for (int k = m; k >= 0; --k) {
        int a = rd[r.read_end1 - k], c = rf[r.ref_end1 - k];
        if ((unsigned)a > 4u) a = 4;
        u += b.mat[c * 5 + a];
        if (u < 0) u = 0;
        if (u >= cap) return true;
    }
    return false;
}
// gapped, banded lower bound (see above).  r03: a vertical gap step INTO the first row of a segment is safe as well -- it is the
// first, unconditional step of the reference's lazy-F loop (ssw.c:303-308) -- but what it leaves there is a final H only: the
// next column's diagonal sees it, the row's E and the F chain below it do not (see k_prove_plain, which shares these moves).
template <int BW>
IPX_DEV bool prove_band(const IpxBatch &b, const IpxResult &r, const int8_t *rd, const int8_t *rf, int Lr, int refLen, int go, int ge,
                        const uint64_t *coltab)
{
    constexpr int HB = BW / 2;
    const int cap = 255 - b.bias;
    const int S8 = (Lr + 15) >> 4;                         // segLen of the 8-bit pass (ssw.c:166)
    const int d0 = r.ref_end1 - r.read_end1;               // column - row of the end diagonal
    bool proven = false;
    int Hm[BW], Hf[BW], F[BW];                             // previous row (main-loop H, final H, F entering the row): cell k is column (row + d0 - HB + k)
    IPX_UNROLL
    for (int k = 0; k < BW; ++k) { Hm[k] = 0; Hf[k] = 0; F[k] = 0; }
    uint64_t win = 0;                                      // 4 bits per band cell: window letter, 7 = outside the window
    IPX_UNROLL
    for (int k = 0; k < BW; ++k) {
        const int c = d0 - HB + k;
        const uint64_t cl = (c >= 0 && c < refLen) ? (uint64_t)(uint8_t)rf[c] : 7u;
        win |= cl << (4 * k);
    }
    int seg = 0;                                           // row % segLen
    int a_next = r.read_end1 >= 0 ? rd[0] : 4;             // (rd may be HBM: the next row's letter is requested a whole row ahead)
    for (int rr = 0; rr <= r.read_end1 && !proven; ++rr) {
        int a = a_next;
        a_next = rd[rr < r.read_end1 ? rr + 1 : rr];
        if ((unsigned)a > 4u) a = 4;
        const uint64_t row = coltab[a];
        const bool cross = seg == 0;                       // first row of a segment: a vertical gap enters through lazy-F's first step only
        int e = 0, hleft = 0;
        IPX_UNROLL
        for (int k = 0; k < BW; ++k) {
            const unsigned cl = (unsigned)(win >> (4 * k)) & 7u;
            int f = 0;
            if (k + 1 < BW) {
                const int f1 = F[k + 1] - ge, f2 = Hm[k + 1] - go;
                f = f1 > f2 ? f1 : f2;
                if (f < 0) f = 0;
            }
            {
                const int e1 = e - ge, e2 = hleft - go;
                e = e1 > e2 ? e1 : e2;
                if (e < 0) e = 0;
            }
            // (a cell outside the window, letter 7, scores 0 -- bytes 5..7 of a table entry -- and needs no special case: left of the
            //  window everything a cell sees is 0 and stays 0; right of it whatever a cell holds never comes back, moves go right and down)
            int hm = Hf[k] + (int)(int8_t)(row >> (8 * cl));
            if (hm < e) hm = e;
            if (hm < 0) hm = 0;
            int hf, fk;
            if (!cross) { if (hm < f) hm = f; hf = hm; fk = f; }
            else { hf = hm > f ? hm : f; fk = 0; }
            if (hf >= cap && cl <= 4u) proven = true;
            Hm[k] = hm; Hf[k] = hf; F[k] = fk; hleft = hm;
        }
        const int cn = rr + 1 + d0 + HB;                   // column entering the band on the next row
        const uint64_t cl = (cn >= 0 && cn < refLen) ? (uint64_t)(uint8_t)rf[cn] : 7u;
        win = (win >> 4) | (cl << (4 * (BW - 1)));
        if (++seg == S8) seg = 0;
    }
    return proven;
}

IPX_KERNEL_WAVE void k_prove_overflow(IpxBatch b, int lds_cap, int chunk_blocks)
{
    IPX_RAISE_PRIO(b);
    // 64 consecutive jobs per round: their reads are contiguous in HBM, so the wave copies them into LDS with coalesced loads
    // and every lane then walks its own read there (the per-lane backwards byte walk straight from HBM fetched ~6 KB per
    // read).  The ungapped test settles about half of the reads.  The others are QUEUED in LDS and the bands run on full waves of
    // them: a narrow band first, the wide one for what it leaves open.  Both queues carry over from chunk to chunk of the wave's
    // walk (r03: a band round runs as soon as 64 jobs wait for it, and once at the end for the remainder; before, every chunk of
    // 256 jobs ran its own rounds, the wide one with a dozen lanes busy).
    const int lane = lane_id();
    uint64_t *coltab = (uint64_t *)IPX_LDS_BASE;                   // [read letter a] -> bytes mat[c][a], c = 0..4
    uint32_t *qa = (uint32_t *)(IPX_LDS_BASE + 64);                // jobs waiting for the narrow band
    uint32_t *qb = qa + IPX_PROVE_QA;                              // jobs waiting for the wide band
    int8_t *stage = (int8_t *)(qb + IPX_PROVE_QB);
    if (lane < 5) {
        uint64_t t = 0;
        for (int c = 0; c < 5; ++c) t |= (uint64_t)(uint8_t)b.mat[c * 5 + lane] << (8 * c);
        coltab[lane] = t;
    }
    const int64_t nb = (b.n_jobs + 63) / 64;
    const int CB = chunk_blocks < 1 ? 1 : chunk_blocks > IPX_PROVE_CHUNK ? IPX_PROVE_CHUNK : chunk_blocks;   // (small batches: 1, for latency)
    const int64_t nchunk = (nb + CB - 1) / CB;
    uint32_t na = 0, nw = 0;                                       // queued jobs (the same in every lane)
    // one band round over cnt <= 64 queue entries starting at q[first]; wide = false: what the narrow band leaves open goes to qb
    auto band_round = [&](const uint32_t *q, uint32_t first, uint32_t cnt, bool wide) {
        IPX_SYNC();                                                // queue entries written; the staging area is free
        const bool mine = (uint32_t)lane < cnt;
        const int64_t i = mine ? (int64_t)q[first + lane] : 0;
        int key = -1;
        bool open = false;
        if (mine) {
            IpxResult r = b.res[i];
            const int Lr = (int)(b.read_off[i + 1] - b.read_off[i]);
            const int8_t *rd = b.reads + b.read_off[i];           // straight from HBM: one letter per band row, requested a row ahead
            const int rid = b.ref_id[i];
            const int8_t *rf = b.refs_packed + b.refp_off[rid];
            const bool proven = wide ? prove_band<IPX_PROVE_BAND>(b, r, rd, rf, Lr, b.ref_len[rid], b.gap_open[i], b.gap_ext[i], coltab)
                                     : prove_band<IPX_PROVE_BAND_NARROW>(b, r, rd, rf, Lr, b.ref_len[rid], b.gap_open[i], b.gap_ext[i], coltab);
            if (!proven && !wide) open = true;
            else {
                // not proven: the 8-bit pass decides.  A read that could score well beyond the threshold (1.5 x) most likely does overflow:
                // the lower-bound stage at 16 reads per wave sees that soonest; a borderline read most likely does not: the stepped pass at once
                const bool likely = 2 * Lr * b.max_match >= 3 * (255 - b.bias);
                r.mode = proven ? IPX_MODE_WORD : ((b.exact_direct && !likely) ? IPX_MODE_NEED_BYTE_EXACT_W : IPX_MODE_NEED_BYTE_CHECK);
                b.res[i] = r;
                key = next_pass_key(b, r, Lr, b.gap_open[i] <= b.gap_ext[i]);
            }
        }
        plan_note(b, key);
        if (!wide) {
            const uint64_t om = xl_ballot(open);
            if (open) qb[nw + (uint32_t)__builtin_popcountll(om & ((1ull << lane) - 1ull))] = (uint32_t)i;
            nw += (uint32_t)__builtin_popcountll(om);
        }
    };
    // run the rounds that are full (all = false) or everything that waits (all = true: the wave's last chunk is done)
    auto drain = [&](bool all) {
        for (;;) {                                                 // (one call site per band, see k_prove_plain)
            if (nw >= 64u || (all && na == 0u && nw > 0u)) { const uint32_t cnt = nw < 64u ? nw : 64u; band_round(qb, nw - cnt, cnt, true); nw -= cnt; continue; }
            if (na >= 64u || (all && na > 0u)) { const uint32_t cnt = na < 64u ? na : 64u; band_round(qa, na - cnt, cnt, false); na -= cnt; continue; }
            break;
        }
    };
    for (int64_t chunk = IPX_BID; chunk < nchunk; chunk += IPX_GDIM) {
        const int64_t cbase = chunk * CB * 64;
        for (int sb = 0; sb < CB; ++sb) {
            const int64_t i0 = cbase + (int64_t)sb * 64, i = i0 + lane;
            if (i0 >= b.n_jobs) break;
            const int64_t i1 = i0 + 64 < b.n_jobs ? i0 + 64 : b.n_jobs;
            const int64_t lo = b.read_off[i0], hi = b.read_off[i1];
            const bool staged = hi - lo <= (int64_t)lds_cap;
            IPX_SYNC();
            if (staged) for (int64_t q = lo + lane; q < hi; q += 64) stage[q - lo] = load_stream_i8(b.reads + q);
            IPX_SYNC();
            int key = -1;                                          // pass the job takes next (plan_note: all lanes)
            bool open = false;
            if (i < b.n_jobs) {
                IpxResult r = b.res[i];
                if (r.mode == IPX_MODE_WORD_UNPROVEN) {
                    const int8_t *rd = staged ? stage + (b.read_off[i] - lo) : b.reads + b.read_off[i];
                    const int8_t *rf = b.refs_packed + b.refp_off[b.ref_id[i]];
                    if (prove_ungapped(b, r, rd, rf)) {
                        r.mode = IPX_MODE_WORD;
                        b.res[i] = r;
                        key = next_pass_key(b, r, (int)(b.read_off[i + 1] - b.read_off[i]), b.gap_open[i] <= b.gap_ext[i]);
                    } else open = true;
                }
            }
            plan_note(b, key);
            const uint64_t om = xl_ballot(open);
            // (a small batch -- one block of 64 jobs per chunk -- takes the wide band at once: two rounds in a row would only add latency)
            if (CB == 1) {
                if (open) qb[nw + (uint32_t)__builtin_popcountll(om & ((1ull << lane) - 1ull))] = (uint32_t)i;
                nw += (uint32_t)__builtin_popcountll(om);
            } else {
                if (open) qa[na + (uint32_t)__builtin_popcountll(om & ((1ull << lane) - 1ull))] = (uint32_t)i;
                na += (uint32_t)__builtin_popcountll(om);
            }
        }
        drain(false);
    }
    drain(true);
}

// ------------------------------------------------------------------------------------------------
// k_prove_overflow_diag (r04, latency tier): the same proof for a SMALL batch, EIGHT LANES PER READ.  k_prove_overflow gives a read one
// lane, which walks up to 150 band rows of 15 cells one after the other (and the ungapped diagonal before that): 0.13 ms however few
// reads there are -- a quarter of a 1 000-job call.  Here the band of IPX_PROVE_BAND = 15 diagonals is an anti-diagonal wavefront (as in
// k_tb_diag: lane kl owns band diagonals 2 kl and 2 kl + 1, a cell's left and upper neighbours are one step old, the diagonal one two):
// 2 x rows steps of a few dozen instructions.  The cells are prove_band<15>'s, move for move (the ungapped test is the band's middle
// diagonal alone and needs no pass of its own); any sound proof gives the same results, it only decides which reads skip the 8-bit pass.
// Dynamic LDS: 64 B score table | per group of 8 lanes: row score words 8 x 256 | band window letters 272
// ------------------------------------------------------------------------------------------------
#define IPX_PROVED_ROWS 256
static inline int ipx_proved_lds_bytes() { return 64 + 8 * (8 * IPX_PROVED_ROWS + IPX_PROVED_ROWS + 16); }
IPX_KERNEL_WAVE void k_prove_overflow_diag(IpxBatch b)
{
    IPX_RAISE_PRIO(b);
    constexpr int BW = IPX_PROVE_BAND, HB = BW / 2, ROWS = IPX_PROVED_ROWS, GB = 8 * ROWS + ROWS + 16;
    static_assert(BW == 15, "eight lanes hold fifteen diagonals");
    const int lane = lane_id(), kl = lane & 7, g = lane >> 3;
    unsigned char *lds = IPX_LDS_BASE;
    uint64_t *coltab = (uint64_t *)lds;                             // [read letter a] -> bytes mat[c][a], c = 0..4 (bytes 5..7 = 0: a column outside the window scores 0)
    uint64_t *srow = (uint64_t *)(lds + 64 + g * GB);               // [row] -> coltab[read letter]
    int8_t *sref = (int8_t *)(srow + ROWS);                         // [band column t = column - (d0 - HB)] -> window letter x 8; 56 = outside the window
    if (lane < 5) {
        uint64_t t = 0;
        for (int c = 0; c < 5; ++c) t |= (uint64_t)(uint8_t)b.mat[c * 5 + lane] << (8 * c);
        coltab[lane] = t;
    }
    IPX_SYNC();
    const int cap = 255 - b.bias;
    const uint32_t lowmask = kl == 0 ? 0u : 0xFFFFFFFFu;
    for (int64_t base = (int64_t)IPX_BID * 8; base < b.n_jobs; base += (int64_t)IPX_GDIM * 8) {      // (uniform)
        const int64_t i = base + g;
        IpxResult r;
        bool mine = false;
        int Lr = 0, go = 1, ge = 0, rows = 0, S8 = 1;
        if (i < b.n_jobs) {
            r = b.res[i];
            Lr = (int)(b.read_off[i + 1] - b.read_off[i]);
            mine = r.mode == IPX_MODE_WORD_UNPROVEN;
            if (mine) {
                go = b.gap_open[i]; ge = b.gap_ext[i];
                rows = r.read_end1 + 1;
                S8 = (Lr + 15) >> 4;                                  // segLen of the 8-bit pass (ssw.c:166)
                if (S8 < 1) S8 = 1;
            }
        }
        const bool fits = rows >= 1 && rows <= ROWS;                  // (longer: left to the 8-bit pass; the tier takes reads of up to 256 bp)
        IPX_SYNC();                                                   // the previous job's letters are no longer read
        if (mine && fits) {
            const int rid = b.ref_id[i];
            const int8_t *rd = b.reads + b.read_off[i];
            const int8_t *rf = b.refs_packed + b.refp_off[rid];
            const int refLen = b.ref_len[rid];
            const int cb = r.ref_end1 - r.read_end1 - HB;             // window column of band column 0
            for (int q = kl; q < rows; q += 8) { const int a = rd[q]; srow[q] = coltab[(unsigned)a > 4u ? 4 : a]; }
            for (int q = kl; q < rows + 16; q += 8) { const int c = cb + q; sref[q] = (int8_t)((c >= 0 && c < refLen) ? rf[c] * 8 : 56); }
        }
        IPX_SYNC();
        const bool run = mine && fits;
        const int TH = (int)xl_first(wave_umax(run ? (uint32_t)(rows + 7) : 0u));       // hi = row + kl
        int Hme = 0, Hfe = 0, Fe = 0, Ee = 0, Hmo = 0, Hfo = 0, Fo = 0, Eo = 0;
        bool proven = false;
        int seg = (S8 - kl % S8) % S8;                                // row % segLen of this lane's row, hi = 0: row = -kl
        auto row8 = [&](int rr) -> uint64_t { return srow[rr < 0 ? 0 : rr > ROWS - 1 ? ROWS - 1 : rr]; };
        auto ref8 = [&](int t) -> int { return (int)sref[t < 0 ? 0 : t > ROWS + 15 ? ROWS + 15 : t] & 56; };
        uint64_t mrow_n = row8(-kl);
        int rc0 = ref8(-kl + 2 * kl), rc1 = ref8(-kl + 2 * kl + 1);    // band column of (row, diagonal q) = row + q
        for (int hi = 0; hi < TH; ++hi) {
            if ((hi & 15) == 15) {                                    // every read of the wave proven (or done): nothing left to find
                const uint32_t pg = group_or<8>(proven ? 1u : 0u);
                if (!xl_any(run && pg == 0u && hi < rows + 7)) break;
            }
            const int rr = hi - kl;
            const uint64_t mrow = mrow_n;
            mrow_n = row8(rr + 1);
            const int rcn = ref8(rr + 2 * kl + 2);                    // the odd cell's band column in the next step
            const bool rowok = run && rr >= 0 && rr < rows;
            const bool cross = seg == 0;                              // first row of a segment: a vertical gap enters through lazy-F's first step only
            // ---- even step: diagonal 2 kl ----
            {
                const int Hl = (int)(xl_row_shr1((uint32_t)Hmo) & lowmask), El = (int)(xl_row_shr1((uint32_t)Eo) & lowmask);
                int f = Fo - ge; { const int f2 = Hmo - go; f = f > f2 ? f : f2; f = f > 0 ? f : 0; }
                int e = El - ge; { const int e2 = Hl - go; e = e > e2 ? e : e2; e = e > 0 ? e : 0; }
                int hm = Hfe + (int)(int8_t)(mrow >> rc0);
                hm = hm > e ? hm : e; hm = hm > 0 ? hm : 0;
                int hf, fk;
                if (!cross) { hm = hm > f ? hm : f; hf = hm; fk = f; }
                else { hf = hm > f ? hm : f; fk = 0; }
                if (rowok && hf >= cap && rc0 != 56) proven = true;
                Hme = rowok ? hm : 0; Hfe = rowok ? hf : 0; Fe = rowok ? fk : 0; Ee = rowok ? e : 0;
            }
            // ---- odd step: diagonal 2 kl + 1 (the eighth lane has none) ----
            {
                const int Fu = (int)xl_row_shl1((uint32_t)Fe), Hu = (int)xl_row_shl1((uint32_t)Hme);
                const bool ok = rowok && kl < 7;
                int f = Fu - ge; { const int f2 = Hu - go; f = f > f2 ? f : f2; f = f > 0 ? f : 0; }
                int e = Ee - ge; { const int e2 = Hme - go; e = e > e2 ? e : e2; e = e > 0 ? e : 0; }
                int hm = Hfo + (int)(int8_t)(mrow >> rc1);
                hm = hm > e ? hm : e; hm = hm > 0 ? hm : 0;
                int hf, fk;
                if (!cross) { hm = hm > f ? hm : f; hf = hm; fk = f; }
                else { hf = hm > f ? hm : f; fk = 0; }
                if (ok && hf >= cap && rc1 != 56) proven = true;
                Hmo = ok ? hm : 0; Hfo = ok ? hf : 0; Fo = ok ? fk : 0; Eo = ok ? e : 0;
            }
            rc0 = rc1; rc1 = rcn;
            if (++seg == S8) seg = 0;
        }
        const bool pj = group_or<8>(proven ? 1u : 0u) != 0u;
        int key = -1;
        if (mine && kl == 0) {
            // not proven: the 8-bit pass decides (k_prove_overflow's rule for which of its stages comes first)
            const bool likely = 2 * Lr * b.max_match >= 3 * (255 - b.bias);
            r.mode = pj ? IPX_MODE_WORD : ((b.exact_direct && !likely) ? IPX_MODE_NEED_BYTE_EXACT_W : IPX_MODE_NEED_BYTE_CHECK);
            b.res[i] = r;
            key = next_pass_key(b, r, Lr, b.gap_open[i] <= b.gap_ext[i]);
        }
        plan_note(b, key);
    }
}
// ------------------------------------------------------------------------------------------------
// k_prove_plain<REV>: certify, from BELOW, an output of the plain recurrence that k_dp_skew (BH = 2) has put into the record.
//
// The reference's 8-bit matrix H8 (sw_sse2_byte, ssw.c:197-384) lies below the plain recurrence's matrix cell by cell, because
// its lazy-F loop can only stop early (ssw.c:311).  It lies ABOVE every score that is reachable with moves the main loop
// (ssw.c:274-299) and the first, unconditional step of the lazy-F loop (k = 0, j = 0, ssw.c:303-308: applied before any exit
// test) are certain to make -- the "safe moves":
//   diagonal steps (the diagonal is read from the final column, pvHStore);
//   horizontal gaps (E is fed by the main loop's H, ssw.c:286-291);
//   vertical gaps INSIDE a segment of the striped layout (F of the main loop, ssw.c:294-296);
//   a vertical gap step INTO the first row of a segment (row % segLen8 == 0) from the row above: that is lazy-F's first step.
//     The value it leaves there is a FINAL H only: the next column's diagonal sees it, this row's E and the F chain below it
//     do not (lazy-F corrects neither, ssw.c:302).
// (Checked cell by cell against the reference's loop on 30 000 adversarial inputs, 9 x 10^8 cells, before it went in; the
// stress tests pin the whole route.)  So if the safe-move score of the cell the plain recurrence names equals the plain value,
// H8 has that value there too, and -- H8 being nowhere larger than the plain matrix -- the output built on that cell is the
// reference's: the best score, its first column and smallest row (forward), the begin cell (reverse).  The second-best column
// (forward, NEED_FWD_PROOF2: a column at or after the first value >= 128) is certified the same way from a witness in that
// column: a band cell, or the horizontal gap that leaves the band towards it -- typically the tail of the best cell carried
// through the padding rows.  One lane per read: the ungapped diagonal first (exact for every read without an indel), then a
// band of IPX_PROVE_BAND diagonals around the target's diagonal for the reads still open, queued so that waves stay full.
// A read whose proof fails takes the lower-bound stage (forward: IPX_PASS_BYTE_LOW2, compared with the record) or the stepped
// reverse pass (IPX_PASS_BYTE_REV).
// ------------------------------------------------------------------------------------------------
struct IpxProveTarget {
    int Lp, ncols;            // rows (letters of the pass's read) and columns (of its window) of the pass
    int r1, c1, v1;           // the cell to certify and the value it must reach
    int e2, s2;               // second-best column and its value; e2 < 0: nothing to certify there
};
// letter of row rr / column c in PASS coordinates (reverse pass: reversed read prefix, window walked right to left)
template <bool REV> IPX_DEV int prove_read_letter(const int8_t *rd, int Lp, int rr) { int a = rd[REV ? Lp - 1 - rr : rr]; return (unsigned)a > 4u ? 4 : a; }
template <bool REV> IPX_DEV unsigned prove_ref_letter(const int8_t *rf, int ncols, int c)
{
    return (c >= 0 && c < ncols) ? (unsigned)(uint8_t)rf[REV ? ncols - 1 - c : c] : 7u;
}

// ungapped: the target's diagonal alone, and the second-best witness through the padding rows and one horizontal gap
template <bool REV>
IPX_DEV bool prove_plain_ungapped(const IpxBatch &b, const IpxProveTarget &t, const int8_t *rd, const int8_t *rf, int go, int ge)
{
    const int m = t.r1 < t.c1 ? t.r1 : t.c1;
    int u = 0;
    for (int k = m; k >= 0; --k) {
        const int a = prove_read_letter<REV>(rd, t.Lp, t.r1 - k);
        const unsigned c = prove_ref_letter<REV>(rf, t.ncols, t.c1 - k);
        u += b.mat[c * 5 + a];
        if (u < 0) u = 0;
    }
    if (u < t.v1) return false;
    if (t.e2 < 0) return true;
    // rows r1+1 .. rows-1 carry the value down the diagonal (padding rows score 0, ssw.c:174-176); only past the read's last letter
    const int rows = 16 * ((t.Lp + 15) >> 4);
    int pad = t.r1 == t.Lp - 1 ? rows - 1 - t.r1 : 0;
    if (t.c1 + pad > t.ncols - 1) pad = t.ncols - 1 - t.c1;
    const int cs = t.c1 + pad;
    return t.e2 > cs && u - go - (t.e2 - cs - 1) * ge >= t.s2;
}

// (wide band, forward) TAIL THEN DESCENT.  The second-best column -- the first one right of the mask -- usually holds the tail of the best
// alignment, a horizontal gap from one of its last cells, and the largest value in that column is the one whose tail ran along one of the
// read's LAST rows and then came DOWN the diagonal through the rows below: the letters the alignment left unaligned, which out there
// may match by chance, and the padding rows, which cost nothing (emulator, 3 000 reads of config 2a: every proof the band left open
// was such a column, 1-2 points above the band's own witnesses; this one closes two thirds of them).  wrow[q * 64] (LDS, this lane's
// column): best hm + c * gapE over the band cells, at least IPX_PROVE_EXT_COLS columns left of e2, of read row Lp - IPX_PROVE_EXT + q.
// Tail along row r0 up to column cs, then down the diagonal to (R, e2): every step is a safe move -- the horizontal gap is fed by the
// main loop's H of the band cell, each diagonal step reads the final H of the column before -- so the sum is a lower bound of the
// reference's H at (R, e2).  A call (rare: only where the band's witnesses fall short), so its registers are not the band loop's.
#define IPX_PROVE_EXT 8
#define IPX_PROVE_EXT_COLS 24
IPX_NOINLINE_DEV int prove_plain_tail_descent(const IpxProveTarget &t, const int8_t *rd, const int8_t *rf, int go, int ge, const uint64_t *coltab,
                                              const int *wrow, int rows)
{
    int best = -1;
    for (int q = 0; q < IPX_PROVE_EXT; ++q) {
        const int r0 = t.Lp - IPX_PROVE_EXT + q, w = wrow[q * 64];
        if (r0 < 0 || w < 0) continue;
        for (int R = r0 + 1; R < rows; ++R) {
            const int cs = t.e2 - (R - r0);
            const int real = (R < t.Lp ? R : t.Lp - 1) - r0;                         // rows of the descent that carry read letters
            int v = w - go - (cs - 1) * ge;
            for (int k = 1; k <= real; ++k)
                v += (int)(int8_t)(coltab[prove_read_letter<false>(rd, t.Lp, r0 + k)] >> (8 * prove_ref_letter<false>(rf, t.ncols, cs + k)));
            if (v > best) best = v;
        }
    }
    return best;
}

template <bool REV, int BW>
IPX_DEV bool prove_plain_band(const IpxBatch &b, const IpxProveTarget &t, const int8_t *rd, const int8_t *rf, int go, int ge, const uint64_t *coltab,
                              int *wrow = nullptr)
{
    constexpr int HB = BW / 2;
    constexpr bool EXTW = !REV && BW == IPX_PROVE_BAND;
    const int S8 = (t.Lp + 15) >> 4;                       // segLen of the 8-bit pass (ssw.c:221)
    const int rows = 16 * S8;
    const int d0 = t.c1 - t.r1;
    const int last = t.e2 >= 0 ? rows - 1 : t.r1;          // the second-best witness may sit in the padding rows
    int Hm[BW], Hf[BW], F[BW];                             // previous row: main-loop H, final H, F entering the row; cell k = column row + d0 - HB + k
    IPX_UNROLL
    for (int k = 0; k < BW; ++k) { Hm[k] = 0; Hf[k] = 0; F[k] = 0; }
    uint64_t win = 0;                                      // 4 bits per band cell: window letter, 7 = outside the window
    IPX_UNROLL
    for (int k = 0; k < BW; ++k) win |= (uint64_t)prove_ref_letter<REV>(rf, t.ncols, d0 - HB + k) << (4 * k);
    int seg = 0, got = -1, w2 = -1, wb = -(1 << 28);
    int a_next = t.Lp > 0 ? prove_read_letter<REV>(rd, t.Lp, 0) : 4;
    for (int rr = 0; rr <= last; ++rr) {
        const int a = a_next;
        a_next = rr + 1 < t.Lp ? prove_read_letter<REV>(rd, t.Lp, rr + 1) : 4;
        const uint64_t row = rr < t.Lp ? coltab[a] : 0ull;  // padding rows score 0 against every letter
        const bool cross = seg == 0;
        int e = 0, hleft = 0, rowbest = -(1 << 28);
        IPX_UNROLL
        for (int k = 0; k < BW; ++k) {
            const unsigned cl = (unsigned)(win >> (4 * k)) & 7u;
            const int c = rr + d0 - HB + k;
            int f = 0;                                     // F entering this row in column c: from the cell above = cell k+1 of the previous row
            if (k + 1 < BW) {
                const int f1 = F[k + 1] - ge, f2 = Hm[k + 1] - go;
                f = f1 > f2 ? f1 : f2;
                if (f < 0) f = 0;
            }
            {
                const int e1 = e - ge, e2 = hleft - go;
                e = e1 > e2 ? e1 : e2;
                if (e < 0) e = 0;
            }
            int hm = Hf[k] + (int)(int8_t)(row >> (8 * cl));        // (outside the window: score 0, no special case -- see prove_band)
            if (hm < e) hm = e;
            if (hm < 0) hm = 0;
            int hf, fk;
            if (!cross) { if (hm < f) hm = f; hf = hm; fk = f; }      // inside a segment: the main loop's own F
            else { hf = hm > f ? hm : f; fk = 0; }                   // first row of a segment: lazy-F's first step, a final value only
            Hm[k] = hm; Hf[k] = hf; F[k] = fk; hleft = hm;
            if (t.e2 >= 0 && cl <= 4u) {
                if (c == t.e2 && hf > w2) w2 = hf;
                if (c < t.e2 && hm + c * ge > wb) wb = hm + c * ge;
                if (EXTW && c < t.e2 - IPX_PROVE_EXT_COLS && hm + c * ge > rowbest) rowbest = hm + c * ge;
            }
        }
        if (EXTW && wrow && rr >= t.Lp - IPX_PROVE_EXT && rr < t.Lp) wrow[(rr - (t.Lp - IPX_PROVE_EXT)) * 64] = rowbest;   // (see prove_plain_tail_descent)
        if (rr == t.r1) got = Hf[HB];
        win = (win >> 4) | ((uint64_t)prove_ref_letter<REV>(rf, t.ncols, rr + 1 + d0 + HB) << (4 * (BW - 1)));
        if (++seg == S8) seg = 0;
    }
    if (got < t.v1) return false;
    if (t.e2 < 0) return true;
    if (wb - go - (t.e2 - 1) * ge > w2) w2 = wb - go - (t.e2 - 1) * ge;
    if (EXTW && wrow && w2 < t.s2 && rows - (t.Lp - IPX_PROVE_EXT) <= IPX_PROVE_EXT_COLS) {
        const int v = prove_plain_tail_descent(t, rd, rf, go, ge, coltab, wrow, rows);
        if (v > w2) w2 = v;
    }
    return w2 >= t.s2;
}

template <bool REV>
IPX_KERNEL_WAVE void k_prove_plain(IpxBatch b, int chunk_blocks)
{
    IPX_RAISE_PRIO(b);
    const int lane = lane_id();
    uint64_t *coltab = (uint64_t *)IPX_LDS_BASE;                   // [read letter a] -> bytes mat[c][a], c = 0..4
    uint32_t *qa = (uint32_t *)(IPX_LDS_BASE + 64);                // jobs waiting for the narrow band
    uint32_t *qb = qa + IPX_PROVE_QA;                              // jobs waiting for the wide band
    int *wrow = (int *)(qb + IPX_PROVE_QB) + lane;                 // [IPX_PROVE_EXT][64]: prove_plain_tail_descent
    if (lane < 5) {
        uint64_t tt = 0;
        for (int c = 0; c < 5; ++c) tt |= (uint64_t)(uint8_t)b.mat[c * 5 + lane] << (8 * c);
        coltab[lane] = tt;
    }
    IPX_SYNC();
    const int64_t nb = (b.n_jobs + 63) / 64;
    const int CB = chunk_blocks < 1 ? 1 : chunk_blocks > IPX_PROVE_CHUNK ? IPX_PROVE_CHUNK : chunk_blocks;
    const int64_t nchunk = (nb + CB - 1) / CB;
    uint32_t na = 0, nw = 0;                                       // queued jobs (the same in every lane); both queues carry over
                                                                   // from chunk to chunk of the wave's walk (see k_prove_overflow)
    // stage 0: the job's own lane, ungapped test; stage 1: narrow band; stage 2: wide band (its verdict stands).  i < 0: no job.
    auto step = [&](int64_t i, int stage) {
        int key = -1;
        bool open = false;
        if (i >= 0) {
            IpxResult r = b.res[i];
            const bool mine = REV ? r.mode == IPX_MODE_NEED_REV_PROOF : (r.mode == IPX_MODE_NEED_FWD_PROOF || r.mode == IPX_MODE_NEED_FWD_PROOF2);
            if (mine) {
                const int rid = b.ref_id[i];
                const int8_t *rd = b.reads + b.read_off[i];
                const int8_t *rf = b.refs_packed + b.refp_off[rid];
                const int Lr = (int)(b.read_off[i + 1] - b.read_off[i]);
                IpxProveTarget t;
                if (!REV) {
                    t.Lp = Lr; t.ncols = b.ref_len[rid]; t.r1 = r.read_end1; t.c1 = r.ref_end1; t.v1 = r.score1;
                    t.e2 = r.mode == IPX_MODE_NEED_FWD_PROOF2 ? r.ref_end2 : -1; t.s2 = r.score2;
                } else {
                    t.Lp = r.read_end1 + 1; t.ncols = r.ref_end1 + 1; t.r1 = r.read_end1 - r.read_begin1; t.c1 = r.ref_end1 - r.ref_begin1;
                    t.v1 = r.score1; t.e2 = -1; t.s2 = 0;
                }
                const bool sane = t.r1 >= 0 && t.r1 < t.Lp && t.c1 >= 0 && t.c1 < t.ncols;
                bool ok = false, decided = true;
                if (sane) {
                    if (stage == 0) { ok = prove_plain_ungapped<REV>(b, t, rd, rf, b.gap_open[i], b.gap_ext[i]); decided = ok; }
                    else if (stage == 1) { ok = prove_plain_band<REV, IPX_PROVE_BAND_NARROW>(b, t, rd, rf, b.gap_open[i], b.gap_ext[i], coltab); decided = ok; }
                    else {
                        if (!REV) { IPX_UNROLL for (int q = 0; q < IPX_PROVE_EXT; ++q) wrow[q * 64] = -1; }   // (rows the read does not have)
                        ok = prove_plain_band<REV, IPX_PROVE_BAND>(b, t, rd, rf, b.gap_open[i], b.gap_ext[i], coltab, REV ? nullptr : wrow);
                    }
                }
                if (!decided) open = true;
                else {
                    if (!REV) r.mode = ok ? (rev_needed(b, r.score1) ? IPX_MODE_BYTE_PLAIN : IPX_MODE_BYTE) : (b.exact_direct ? IPX_MODE_NEED_BYTE_EXACT_P : IPX_MODE_NEED_BYTE_LOW_CMP);
                    else {
                        // certified: final (and score1 is the plain optimum: k_tb_list may use that); otherwise the stepped reverse pass decides
                        r.mode = (ok && (7 & b.flag) != 0) ? IPX_MODE_BYTE_OPT : IPX_MODE_BYTE;
                        if (!ok) { r.ref_begin1 = -1; r.read_begin1 = -1; }
                    }
                    b.res[i] = r;
                    key = next_pass_key(b, r, Lr, b.gap_open[i] <= b.gap_ext[i]);
                }
            }
        }
        plan_note(b, key);
        if (stage < 2) {                                           // still open: the next stage's queue (a small batch -- one block of 64
            const uint64_t om = xl_ballot(open);                   //  jobs per chunk -- goes from the ungapped test to the wide band at once)
            const uint32_t at = (uint32_t)__builtin_popcountll(om & ((1ull << lane) - 1ull)), cnt = (uint32_t)__builtin_popcountll(om);
            if (stage == 0 && CB > 1) { if (open) qa[na + at] = (uint32_t)i; na += cnt; }
            else { if (open) qb[nw + at] = (uint32_t)i; nw += cnt; }
        }
    };
    auto band_round = [&](const uint32_t *q, uint32_t first, uint32_t cnt, int stage) {
        IPX_SYNC();                                                // queue entries written
        step((uint32_t)lane < cnt ? (int64_t)q[first + lane] : (int64_t)-1, stage);
    };
    // full rounds first (the wide band's before the narrow band's, whose leftovers feed it); with `all` -- the wave's last chunk is done --
    // the remainders too: the narrow queue's, then the wide queue's.  (ONE call site per band: inlined three times, the wide band's code
    // took the kernel from 164 to 248 registers.)
    auto drain = [&](bool all) {
        for (;;) {
            if (nw >= 64u || (all && na == 0u && nw > 0u)) { const uint32_t cnt = nw < 64u ? nw : 64u; band_round(qb, nw - cnt, cnt, 2); nw -= cnt; continue; }
            if (na >= 64u || (all && na > 0u)) { const uint32_t cnt = na < 64u ? na : 64u; band_round(qa, na - cnt, cnt, 1); na -= cnt; continue; }
            break;
        }
    };
    for (int64_t chunk = IPX_BID; chunk < nchunk; chunk += IPX_GDIM) {
        const int64_t cbase = chunk * CB * 64;
        for (int sb = 0; sb < CB; ++sb) {
            const int64_t ii = cbase + (int64_t)sb * 64 + lane;
            step(ii < b.n_jobs ? ii : (int64_t)-1, 0);
        }
        drain(false);
    }
    drain(true);
}

// ------------------------------------------------------------------------------------------------
// traceback job list: jobs that get a CIGAR (ssw.c:894)
// ------------------------------------------------------------------------------------------------
IPX_DEV bool cigar_needed(const IpxBatch &b, const IpxResult &r)
{
    if (r.mode != IPX_MODE_BYTE && r.mode != IPX_MODE_WORD && r.mode != IPX_MODE_BYTE_OPT) return false;
    if (!rev_needed(b, r.score1)) return false;
    if ((7 & b.flag) == 0) return false;
    if ((2 & b.flag) != 0 && r.score1 < b.filters) return false;
    if ((4 & b.flag) != 0 && (r.ref_end1 - r.ref_begin1 > b.filterd || r.read_end1 - r.read_begin1 > b.filterd)) return false;
    return true;
}

// Jobs are bucketed by their first band width |refLen-readLen|+1 (ssw.c:899): widths 1..7 go to the
// register/LDS-resident kernel k_tb_fast<BW> (list BW-1 = lists + (BW-1)*n_jobs, counter BW-1), wider
// ones straight to the general kernel (list 7 = `esc`, counter 7).
#define IPX_TBF_MAXBW 7
#define IPX_TBD_ROWS 256          // (k_tb_diag) longest aligned read span the anti-diagonal tiers take (longer jobs: k_tb_coop)
#define IPX_TBD_CIG 64            // (k_tb_diag) CIGAR runs of a job held in LDS (more: k_tb_coop)
// r03, UNGAPPED alignments: when refLen == readLen (first band 1) and the scores on the rectangle's diagonal add up to score1, the
// CIGAR is one run of M and banded_sw's DP is not needed to know it.  Why: every value of banded_sw's matrix is the score of a local
// alignment inside the rectangle (floors at 0, ssw.c:655-656), hence <= score1, the optimum.  Were H at some diagonal cell larger than
// the diagonal's prefix sum there, continuing along the diagonal would end above score1; so H equals the prefix sum on the whole
// diagonal, the band of width 1 reaches score1 (no doubling, ssw.c:669), at every diagonal cell the gap candidates are <= the diagonal
// one and ties go to the diagonal (ssw.c:663), and the walk back from the corner (ssw.c:673-733) never leaves it: (n)M by the tail rule
// (ssw.c:734-751).  The job gets that CIGAR here and appears in no traceback list.  (A prefix sum cannot be negative either: dropping
// the prefix would beat the optimum.)
// The argument needs score1 to BE the optimum of the plain recurrence.  That holds for gap_open > gap_ext and a 16-bit result, an 8-bit
// result below 128 (no carry reaches the signed compare, ssw.c:311) or one certified against the plain recurrence (IPX_MODE_BYTE_OPT); an
// 8-bit score of 128 or more from the stepped passes may lie BELOW the optimum (the reference's loop can stop early), and so may any
// result under gap_open <= gap_ext (textbook 50, reference 49) -- the banded DP can then beat the diagonal through an
// insertion / deletion pair and the reference's walk leaves it: those jobs keep the traceback kernels.
IPX_DEV bool tb_ungapped(const IpxBatch &b, const IpxResult &r, int64_t i, int n, const uint64_t *coltab)
{
    const int8_t *rd = b.reads + b.read_off[i] + r.read_begin1;
    const int8_t *rf = b.refs_packed + b.refp_off[b.ref_id[i]] + r.ref_begin1;
    int u = 0;
    for (int k = 0; k < n; ++k) {
        int a = rd[k];
        if ((unsigned)a > 4u) a = 4;
        u += (int)(int8_t)(coltab[a] >> (8 * (unsigned)(uint8_t)rf[k]));
    }
    return u == (int)r.score1;
}
// r04: with the anti-diagonal tiers (IpxBatch::tb_diag, k_tb_diag) the classes are 0..6 = lane-per-job widths 1..7, 7 = k_tb_coop (`esc`),
// 8 / 9 / 10 = the tiers of 16 / 32 / 64 lanes per job (first band <= 15 / 31 / 63; lists 7 / 8 / 9 of `lists`, counters 8 / 9 / 10).  A SMALL batch
// (`all_general`) has more SIMDs than jobs: every job takes a tier -- a quarter of a wave at least -- instead of one lane.
#define IPX_TB_CLS_COOP 7
#define IPX_TB_CLS_DIAG 8
#define IPX_TB_NLISTS 10          // lists of n_jobs entries in IpxWorkspace::tb_list (7 widths + 3 tiers; k_tb_coop's list is IpxWorkspace::tb_esc)
#define IPX_TB_NCOUNTERS 12
IPX_KERNEL void k_tb_list(IpxBatch b, uint32_t *lists, uint32_t *counters, uint32_t *esc, int all_general, int ungapped)
{
    IPX_RAISE_PRIO(b);
    uint64_t *coltab = (uint64_t *)IPX_LDS_BASE;                   // [read letter a] -> bytes mat[c][a], c = 0..4
    if (IPX_TID < 5) {
        uint64_t t = 0;
        for (int c = 0; c < 5; ++c) t |= (uint64_t)(uint8_t)b.mat[c * 5 + IPX_TID] << (8 * c);
        coltab[IPX_TID] = t;
    }
    IPX_SYNC();
    const int64_t chunk = (int64_t)IPX_BDIM * IPX_PLAN_ROUNDS, stride = (int64_t)IPX_GDIM * chunk;
    const int64_t iters = (b.n_jobs + stride - 1) / stride;
    for (int64_t q = 0; q < iters; ++q) {                  // every lane runs every iteration (wave-wide ballots)
        int cls[IPX_PLAN_ROUNDS];
        uint32_t slot[IPX_PLAN_ROUNDS];
        IPX_UNROLL
        for (int k = 0; k < IPX_PLAN_ROUNDS; ++k) {
            const int64_t i = q * stride + (int64_t)IPX_BID * chunk + (int64_t)k * IPX_BDIM + IPX_TID;
            cls[k] = -1;
            if (i < b.n_jobs) {
                IpxResult r = b.res[i];
                const bool opt = r.mode == IPX_MODE_BYTE_OPT;
                if (opt) { r.mode = IPX_MODE_BYTE; ((volatile uint8_t *)&b.res[i].mode)[0] = IPX_MODE_BYTE; }
                // every record is final by now -- unless its next pass was predicted empty and not launched (latency tier): the host repeats the run
                // (such a record is not listed for the traceback either: its begin position is not there yet)
                const bool stuck = (r.mode != IPX_MODE_BYTE && r.mode != IPX_MODE_WORD && r.mode != IPX_MODE_FAIL) ||
                                   (r.mode != IPX_MODE_FAIL && rev_needed(b, r.score1) && r.read_begin1 < 0);       // (... or its reverse pass was not launched)
                if (stuck) atomic_or_u32(b.status, IPX_STATUS_RERUN);
                if (!stuck && cigar_needed(b, r)) {
                    const int refLen = r.ref_end1 - r.ref_begin1 + 1, readLen = r.read_end1 - r.read_begin1 + 1;
                    const bool exact_opt = b.gap_open[i] > b.gap_ext[i] && (r.mode == IPX_MODE_WORD || opt || r.score1 < 128);
                    const int bw = (refLen > readLen ? refLen - readLen : readLen - refLen) + 1;
                    if (b.tb_diag) {
                        b.tb_bw[i] = 0;                                                        // (every tier starts the job from its own first band)
                        // (a small batch starts in a WIDER tier than its band needs: more lanes per job = more widths of the doubling sequence
                        //  side by side -- all_general 1: at least 32 lanes per job, 2: 64, 3: 16)
                        const int lo = all_general == 2 ? 63 : all_general == 1 ? 31 : 15;      // (3: 16 lanes at least)
                        const int bt = bw > lo ? bw : lo;
                        const int tier = (readLen > IPX_TBD_ROWS || readLen < 1 || refLen < 1) ? IPX_TB_CLS_COOP
                                         : bt <= 15 ? IPX_TB_CLS_DIAG : bt <= 31 ? IPX_TB_CLS_DIAG + 1 : bt <= 63 ? IPX_TB_CLS_DIAG + 2 : IPX_TB_CLS_COOP;
                        cls[k] = (bw <= IPX_TBF_MAXBW && !all_general) ? bw - 1 : tier;
                    } else cls[k] = (bw <= IPX_TBF_MAXBW && !all_general) ? bw - 1 : IPX_TB_CLS_COOP;   // (all_general: a small batch, one wave per job)
                    if (ungapped && exact_opt && bw == 1 && readLen > 0 && r.ref_begin1 >= 0 && r.read_begin1 >= 0 && tb_ungapped(b, r, i, readLen, coltab)) {
                        const uint32_t off = atomic_add_u32(b.cigar_cursor, 1u);
                        if (off + 1u > b.cigar_cap) atomic_or_u32(b.status, IPX_STATUS_CIGAR_POOL);
                        else {
                            b.cigar_pool[off] = (uint32_t)readLen << 4;                        // (n)M
                            r.cigar_off = off;
                            r.cigar_len = 1;
                            b.res[i] = r;
                        }
                        cls[k] = -1;
                    }
                }
            }
        }
        wave_class_slots(counters, cls, slot);
        IPX_UNROLL
        for (int k = 0; k < IPX_PLAN_ROUNDS; ++k) {
            const int64_t i = q * stride + (int64_t)IPX_BID * chunk + (int64_t)k * IPX_BDIM + IPX_TID;
            if (cls[k] >= 0) {
                if (cls[k] < IPX_TBF_MAXBW) lists[(int64_t)cls[k] * b.n_jobs + slot[k]] = (uint32_t)i;
                else if (cls[k] == IPX_TB_CLS_COOP) esc[slot[k]] = (uint32_t)i;
                else lists[(int64_t)(cls[k] - 1) * b.n_jobs + slot[k]] = (uint32_t)i;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// k_tb_fast<BW>: banded_sw (ssw.c:588-772) for the common case -- first band BW = |refLen-readLen|+1
// in 1..7 and no band doubling.  One lane per job, all lanes of a launch share BW, so a DP row is
// straight-line code: the band rows h_b/e_b/h_c (2*BW+3 ints each) live in REGISTERS (slot = compile-
// time index; the one-column band shift of a row is a per-lane select between neighbouring slots),
// the window letters under the band slide through a 128-bit register pair, and the three direction
// planes of the row's 2*BW+1 cells are packed into ONE word (32 bit for BW <= 3, 64 bit for BW 4..7;
// 4 bits per cell: 0 = never written, else 1 + 4*Hsrc + 2*Fopen + Eopen) kept in LDS [row][lane].  Nothing of the traceback touches HBM except
// the job's own letters and the CIGAR.  Cells keep the reference's linear index width_d*i + (j-shift),
// so out-of-band reads alias exactly as in the reference.  Jobs that need a second band iteration
// (max < score, ssw.c:669), more rows than `rowcap` or more than 32 CIGAR runs go to `next`.
// Dynamic LDS: 64 B score table (per read letter, 5 window letters) | 32*64 CIGAR ops.   Global: rowcap*64 direction words (4 or 8 bytes) per block
// ------------------------------------------------------------------------------------------------
#define IPX_TBF_CIG 32
static inline int ipx_tbf_lds_bytes() { return 64 + IPX_TBF_CIG * 256; }
IPX_HD size_t ipx_tbf_scratch_bytes_per_block(int rowcap) { return (size_t)rowcap * 512; }   // sized for 64-bit words
template <int BW> struct IpxTbWord { typedef uint32_t type; };
template <> struct IpxTbWord<4> { typedef uint64_t type; };
template <> struct IpxTbWord<5> { typedef uint64_t type; };
template <> struct IpxTbWord<6> { typedef uint64_t type; };
template <> struct IpxTbWord<7> { typedef uint64_t type; };

// (body shared by the per-width kernels and the all-widths kernel below: `bid` of `gdim` blocks work on this list,
//  `scratch_bid` names the block's direction-word region)
// WIDEN: a job whose band has to double (max < score, ssw.c:669) is handed to the list of width 2*BW -- served by a LATER
// launch of this same lane-per-job kernel -- instead of the one-wave-per-job kernel, as long as 2*BW <= 7.  A list entry
// with bit 31 set is such a second (third) band iteration: its first band was narrower than BW.  The doubled run starts
// from fresh direction words, the reference from the words the narrower iterations left behind; the two differ only if
// the walk back reads a cell the current iteration has not written, so a retried job that meets such a cell goes to the
// general kernel (which keeps the reference's buffer across iterations) instead of reporting the failure itself.
template <int BW, bool WIDEN>
IPX_DEV void tb_fast_body(const IpxBatch &b, const uint32_t *list, const uint32_t *list_n, int rowcap,
                          unsigned char *dir_scratch, uint32_t *next, uint32_t *next_n, int bid, int gdim, int scratch_bid)
{
    uint32_t *lists_base = (uint32_t *)list - (int64_t)(BW - 1) * b.n_jobs;   // list BW-1 of the seven per-width lists
    uint32_t *counters = next_n - IPX_TBF_MAXBW;                              // their counters (the eighth is next_n)
    constexpr int WD = 2 * BW + 1, W = 2 * BW + 3;
    typedef typename IpxTbWord<BW>::type word_t;
    const int lane = lane_id();
    unsigned char *lds = IPX_LDS_BASE;
    uint64_t *coltab = (uint64_t *)lds;                                 // [read letter a] -> bytes mat[c*5 + a], c = 0..4
    uint32_t *cig = (uint32_t *)(lds + 64) + lane;                      // [k*64]
    word_t *dirw = (word_t *)(dir_scratch + (size_t)scratch_bid * ipx_tbf_scratch_bytes_per_block(rowcap)) + lane;   // [row*64]
    if (lane < 5) {
        uint64_t t = 0;
        for (int c = 0; c < 5; ++c) t |= (uint64_t)(uint8_t)b.mat[c * 5 + lane] << (8 * c);
        coltab[lane] = t;
    }
    IPX_SYNC();
    const uint32_t n = *list_n;

    for (int64_t base = (int64_t)bid * 64; base < (int64_t)n; base += (int64_t)gdim * 64) {
        const int64_t li = base + lane;
        if (li >= (int64_t)n) continue;
        const uint32_t ent = list[li];
        const int64_t jb = ent & 0x7FFFFFFFu;
        const bool retried = (ent >> 31) != 0;                        // a doubled band: the first band was narrower
        IpxResult r = b.res[jb];
        const int rid = b.ref_id[jb];
        const int fullRef = b.ref_len[rid];
        const int8_t *refp = b.refs_packed + b.refp_off[rid];
        const int8_t *readp = b.reads + b.read_off[jb] + r.read_begin1;
        const int rb = r.ref_begin1;
        const int refLen = r.ref_end1 - r.ref_begin1 + 1;             // ssw.c:897-899
        const int readLen = r.read_end1 - r.read_begin1 + 1;
        const int gapO = b.gap_open[jb], gapE = b.gap_ext[jb];
        const int score = r.score1;
        const int bw = (refLen > readLen ? refLen - readLen : readLen - refLen) + 1;
        const int len = refLen > readLen ? refLen : readLen;
        bool esc = (!retried && bw != BW) || readLen > rowcap;
        bool widen = false, todiag = false;
        int mx = 0;
        if (!esc) {
            int hb[W + 1], eb[W + 1], hc[W + 1];                        // fresh arrays (ssw.c:607-609, 627)
            IPX_UNROLL
            for (int q = 0; q < W + 1; ++q) { hb[q] = 0; eb[q] = 0; hc[q] = 0; }
            auto ref_at = [&](int j) -> uint64_t {
                const int ri = rb + j;
                int c = (j >= 0 && j < refLen && ri >= 0 && ri < fullRef) ? refp[ri] : 0;
                return (uint64_t)(uint32_t)(c & 0xFF);
            };
            // byte s of (win1:win0) = letter of column shift+s
            uint64_t win0 = 0, win1 = 0;
            IPX_UNROLL
            for (int q = 0; q <= BW; ++q) win0 |= ref_at(q) << (8 * q);  // row 0 covers columns 0..BW (BW <= 7)
            int rc_next = readLen > 0 ? readp[0] : 0;
            uint64_t c_next = ref_at(BW + 1);                           // column entering at row 1
            for (int i = 0; i < readLen; ++i) {
                const int x = i - BW > 0 ? i - BW : 0;                  // band shift of row i (= first column)
                const bool sh = i > BW;                                 // the band moved one column since row i-1
                int end = refLen - 1;
                if (i + BW < end) end = i + BW;
                const int nact = end - x + 1;                           // cells of this row
                const int edge = end + 1 < W - 1 ? end + 1 : W - 1;     // ssw.c:632
                hb[0] = 0; eb[0] = 0; hc[0] = 0;                        // ssw.c:633
                IPX_UNROLL
                for (int q = 1; q < W; ++q) if (q == edge) { hb[q] = 0; eb[q] = 0; }
                int rc = rc_next;
                if ((unsigned)rc > 4u) rc = 4;
                if (i > 0) {                                            // slide the window to row i
                    if (sh) { win0 = (win0 >> 8) | (win1 << 56); win1 >>= 8; }
                    const int pos = i + BW - x;                         // column i+BW (letter 0 beyond refLen)
                    if (pos < 8) win0 |= c_next << (8 * pos);
                    else win1 |= c_next << (8 * (pos - 8));
                }
                rc_next = i + 1 < readLen ? readp[i + 1] : 0;
                c_next = ref_at(i + 1 + BW);
                const uint64_t mrow = coltab[rc];                       // this row's scores against the 5 window letters: one LDS
                                                                        // read per row, the cells then only shift (mat[ref*5 + read])
                int f = 0, hleft = 0;
                word_t word = 0;
                IPX_UNROLL
                for (int s2 = 0; s2 < WD; ++s2) {
                    const bool act = s2 < nact;
                    const int u = s2 + 1;                               // set_u (ssw.c:92); e = u + sh, d = e - 1
                    const int hbe = sh ? hb[u + 1] : hb[u];
                    const int ebe = sh ? eb[u + 1] : eb[u];
                    const int hbd = sh ? hb[u] : hb[u - 1];
                    int t1 = i == 0 ? -gapO : hbe - gapO;               // ssw.c:644-648
                    int t2 = i == 0 ? -gapE : ebe - gapE;
                    const int ev = t1 > t2 ? t1 : t2;
                    const int de = t1 > t2 ? 1 : 0;
                    t1 = hleft - gapO;                                  // ssw.c:650-653
                    t2 = f - gapE;
                    const int fv = t1 > t2 ? t1 : t2;
                    const int df = t1 > t2 ? 1 : 0;
                    const int e1 = ev > 0 ? ev : 0;                     // ssw.c:655-664
                    const int f1 = fv > 0 ? fv : 0;
                    t1 = e1 > f1 ? e1 : f1;
                    const int rcode = (int)(((s2 < 8 ? win0 >> (8 * s2) : win1 >> (8 * (s2 - 8)))) & 0xFFu);
                    t2 = hbd + (int)(int8_t)(mrow >> (8 * rcode));
                    const int hv = t1 > t2 ? t1 : t2;
                    const int dh = t1 <= t2 ? 0 : (e1 > f1 ? 1 : 2);
                    if (act) {
                        eb[u] = ev; f = fv; hc[u] = hv; hleft = hv;
                        if (hv > mx) mx = hv;
                        word |= (word_t)(1 + dh * 4 + df * 2 + de) << (4 * s2);
                    }
                }
                dirw[i * 64] = word;
                IPX_UNROLL
                for (int q = 1; q <= WD; ++q) if (q <= nact) hb[q] = hc[q];   // ssw.c:666
            }
            if (mx < score && BW * 2 <= len) {                          // band would double (ssw.c:668-669)
                if (WIDEN && 2 * BW <= IPX_TBF_MAXBW) widen = true;
                else if (b.tb_diag && readLen <= IPX_TBD_ROWS) todiag = true;   // the band of 2 * BW: the anti-diagonal tiers (k_tb_diag) start there
                else esc = true;
            }
        }
        int lcnt = 0, e = 0, op = 0;
        bool fail = false;
        if (!esc && !widen && !todiag) {
            // ---- trace back (ssw.c:673-751) ----
            // the walk moves up one row at most per step: keep the words of rows i and i-1 in registers and
            // fetch row i-2 as soon as the walk moves (a cell index outside those two rows reads memory)
            int i = readLen - 1, j = refLen - 1, plane = 2, prev = 0;   // op: 0 M, 1 I, 2 D
            int wrow = i;                                               // row held in w0; w1 = row wrow-1
            word_t w0 = i >= 0 ? dirw[i * 64] : 0, w1 = i >= 1 ? dirw[(i - 1) * 64] : 0;
            while (i >= 0 && j > 0) {
                const int x = i - BW > 0 ? i - BW : 0;
                const int cell = WD * i + (j - x);
                int code = 0;
                if (cell >= 0) {
                    const int row = cell / WD, slot = cell - row * WD;
                    if (row < readLen) {
                        const word_t ww = row == wrow ? w0 : row == wrow - 1 ? w1 : dirw[row * 64];
                        const int v = (int)((ww >> (4 * slot)) & 15u);
                        if (v) {
                            const int de = 2 + ((v - 1) & 1), df = 4 + (((v - 1) >> 1) & 1), dh = (v - 1) >> 2;
                            code = plane == 0 ? de : plane == 1 ? df : (dh == 0 ? 1 : dh == 1 ? de : df);
                        }
                    }
                }
                const int iold = i;
                if (code == 1) { --i; --j; plane = 2; op = 0; }
                else if (code == 2) { --i; plane = 0; op = 1; }
                else if (code == 3) { --i; plane = 2; op = 1; }
                else if (code == 4) { --j; plane = 1; op = 2; }
                else if (code == 5) { --j; plane = 2; op = 2; }
                else { fail = true; break; }
                if (i != iold) { wrow = i; w0 = w1; w1 = i >= 1 ? dirw[(i - 1) * 64] : 0; }
                if (op == prev) ++e;
                else {
                    ++lcnt;
                    if (lcnt + 2 > IPX_TBF_CIG) { esc = true; break; }
                    cig[(lcnt - 1) * 64] = ((uint32_t)e << 4) | (uint32_t)prev;
                    prev = op;
                    e = 1;
                }
            }
        }
        if (widen) {
            constexpr int NB = 2 * BW <= IPX_TBF_MAXBW ? 2 * BW : IPX_TBF_MAXBW;
            lists_base[(int64_t)(NB - 1) * b.n_jobs + atomic_add_u32(&counters[NB - 1], 1u)] = (uint32_t)jb | 0x80000000u;
            continue;
        }
        if (todiag) {
            b.tb_bw[jb] = (uint16_t)(2 * BW);
            lists_base[(int64_t)(IPX_TB_CLS_DIAG - 1) * b.n_jobs + atomic_add_u32(&counters[IPX_TB_CLS_DIAG], 1u)] = (uint32_t)jb;
            continue;
        }
        if (fail && retried) esc = true;                               // (see WIDEN above)
        if (esc) { if (b.tb_diag) b.tb_bw[jb] = 0; next[atomic_add_u32(next_n, 1u)] = (uint32_t)jb; continue; }
        if (fail) { r.flag = 1; r.cigar_len = 0; b.res[jb] = r; continue; }                // ssw.c:911
        if (op == 0) { ++lcnt; cig[(lcnt - 1) * 64] = ((uint32_t)(e + 1) << 4); }          // ssw.c:734-751
        else { lcnt += 2; cig[(lcnt - 2) * 64] = ((uint32_t)e << 4) | (uint32_t)op; cig[(lcnt - 1) * 64] = (1u << 4); }
        const uint32_t off = atomic_add_u32(b.cigar_cursor, (uint32_t)lcnt);
        if (off + (uint32_t)lcnt > b.cigar_cap) { atomic_or_u32(b.status, IPX_STATUS_CIGAR_POOL); continue; }
        for (int k = 0; k < lcnt; ++k) b.cigar_pool[off + k] = cig[(lcnt - 1 - k) * 64];   // reverse (ssw.c:754-762)
        r.cigar_off = off;
        r.cigar_len = (uint16_t)lcnt;
        b.res[jb] = r;
    }
}

template <int BW>
IPX_KERNEL_WAVE void k_tb_fast(IpxBatch b, const uint32_t *list, const uint32_t *list_n, int rowcap,
                               unsigned char *dir_scratch, uint32_t *next, uint32_t *next_n)
{
    IPX_RAISE_PRIO(b);
    tb_fast_body<BW, true>(b, list, list_n, rowcap, dir_scratch, next, next_n, (int)IPX_BID, (int)IPX_GDIM, (int)IPX_BID);
}

// Several widths in ONE launch, `per` blocks each (block b serves width bw_first + b / per + 1).  The per-width launches
// are each a chain of dependent steps one lane deep: run one after the other they are most of the latency of a small
// call (1000 jobs: 1.85 of 3.0 ms), and in a big batch the rare wide bands (5..7: a few hundred jobs) still cost a full
// ~0.5 ms chain each (r02 profile); side by side they cost as much as the slowest.  Small batches fuse all seven widths
// (bw_first 0), big ones the widths 4..7 (bw_first 3) behind the per-width launches of 1..3.
IPX_KERNEL_WAVE void k_tb_fast_all(IpxBatch b, const uint32_t *lists, const uint32_t *counters, int rowcap,
                                   unsigned char *dir_scratch, uint32_t *next, uint32_t *next_n, int per, int bw_first)
{
    IPX_RAISE_PRIO(b);
    if (per == 0) {
        // r04, big batches: the seven widths in ONE launch, the blocks shared out by the lists' lengths (read here: k_tb_list has filled
        // them) -- virtual block v serves the v-th chunk of 64 jobs of the concatenated lists.  Four launches in a row (1, 2, 3, 4..7) were four
        // ramps of a thousand waves each through wave slots the other streams' wavefront kernels keep full (config 4, four streams: 2.75 ms
        // of a slice's 16.6 ms step).  A band that has to double goes to the anti-diagonal tiers (no later launch of this kernel to take it).
        uint32_t first[IPX_TBF_MAXBW + 1];
        first[0] = 0;
        for (int w = 0; w < IPX_TBF_MAXBW; ++w) first[w + 1] = first[w] + (counters[w] + 63u) / 64u;
        for (uint32_t v = (uint32_t)IPX_BID; v < first[IPX_TBF_MAXBW]; v += (uint32_t)IPX_GDIM) {
            int w = 0;
            while (v >= first[w + 1]) ++w;
            const int loc = (int)(v - first[w]), cnt_w = (int)(first[w + 1] - first[w]);
            const uint32_t *list = lists + (int64_t)w * b.n_jobs, *cnt = counters + w;
            IPX_SYNC();                                               // (the previous chunk's lanes are done with the score table and the CIGAR buffer)
            switch (w) {
            case 0: tb_fast_body<1, false>(b, list, cnt, rowcap, dir_scratch, next, next_n, loc, cnt_w, (int)IPX_BID); break;
            case 1: tb_fast_body<2, false>(b, list, cnt, rowcap, dir_scratch, next, next_n, loc, cnt_w, (int)IPX_BID); break;
            case 2: tb_fast_body<3, false>(b, list, cnt, rowcap, dir_scratch, next, next_n, loc, cnt_w, (int)IPX_BID); break;
            case 3: tb_fast_body<4, false>(b, list, cnt, rowcap, dir_scratch, next, next_n, loc, cnt_w, (int)IPX_BID); break;
            case 4: tb_fast_body<5, false>(b, list, cnt, rowcap, dir_scratch, next, next_n, loc, cnt_w, (int)IPX_BID); break;
            case 5: tb_fast_body<6, false>(b, list, cnt, rowcap, dir_scratch, next, next_n, loc, cnt_w, (int)IPX_BID); break;
            default: tb_fast_body<7, false>(b, list, cnt, rowcap, dir_scratch, next, next_n, loc, cnt_w, (int)IPX_BID); break;
            }
        }
        return;
    }
    const int bid = (int)IPX_BID, bw = bw_first + bid / per, loc = bid % per;
    const uint32_t *list = lists + (int64_t)bw * b.n_jobs, *cnt = counters + bw;
    switch (bw) {
    case 0: tb_fast_body<1, false>(b, list, cnt, rowcap, dir_scratch, next, next_n, loc, per, bid); break;
    case 1: tb_fast_body<2, false>(b, list, cnt, rowcap, dir_scratch, next, next_n, loc, per, bid); break;
    case 2: tb_fast_body<3, false>(b, list, cnt, rowcap, dir_scratch, next, next_n, loc, per, bid); break;
    case 3: tb_fast_body<4, false>(b, list, cnt, rowcap, dir_scratch, next, next_n, loc, per, bid); break;
    case 4: tb_fast_body<5, false>(b, list, cnt, rowcap, dir_scratch, next, next_n, loc, per, bid); break;
    case 5: tb_fast_body<6, false>(b, list, cnt, rowcap, dir_scratch, next, next_n, loc, per, bid); break;
    default: tb_fast_body<7, false>(b, list, cnt, rowcap, dir_scratch, next, next_n, loc, per, bid); break;
    }
}

// scratch of the one-wave-per-job kernel: direction bytes and CIGAR runs, one region per block
struct IpxTbScratch {
    uint8_t *dir;
    uint32_t *cig;
    int32_t *band;            // band rows of jobs too wide for LDS: 4 * arrcap ints per block
    int32_t arrcap, dircap, cigcap;
    int32_t arrcap_lds;       // band rows up to this many entries live in LDS
};

// ------------------------------------------------------------------------------------------------
// k_tb_coop: banded_sw (ssw.c:588-772), general form, for the jobs k_tb_fast hands over: first band > 7,
// band doubling (max < score, ssw.c:669: large indels, or a forward score the banded DP never reaches, so
// the band grows to the full rectangle), very long reads or CIGARs.  ONE WAVEFRONT
// PER JOB: the cells of a DP row are spread over the 64 lanes (one contiguous chunk per lane).  The only
// dependency inside a row is the horizontal-gap chain f_j = max(H_{j-1} - gapO, f_{j-1} - gapE); with
// A_j = max(E_j, 0, diag_j) ("H without F") it is the max-plus recurrence
//     f_{j+1} = max(A_j - gapO, f_j - min(gapO, gapE)),
// a prefix scan: each lane folds its chunk, the lane aggregates are scanned across the wave, and a
// second sweep produces H, the direction planes (with the reference's tie-breaks, which need the actual
// H_{j-1} and f_{j-1}; across a lane boundary they are handed over through LDS) and the new band row.
// Band rows live in LDS; direction bytes -- 1 byte per DP cell: bit7 written | bits3:2 H source (0 diag,
// 1 E, 2 F) | bit1 F came from open | bit0 E came from open, i.e. the reference's three direction planes
// of a cell, kept at the reference's linear cell index so that out-of-band reads and cells left over from
// an earlier, narrower band iteration alias exactly as in the reference; never-written cells read 0 = the
// reference's "Trace back error" path (it reads uninitialised heap there) -- sit in a per-block global
// scratch; the trace back itself is sequential (lane 0).
// Dynamic LDS: 64 B matrix | 128 ints hand-over | 4 * arrcap_lds ints (h_b, e_b, h_c, new e_b).  A job whose band rows need more
// (2 * max(aligned window span, aligned read span) + 8 entries: long reads, long deletions) keeps them in its block's region of the
// global scratch instead (r03: windows up to 32 000 bp, reads up to 4 096 bp).
// ------------------------------------------------------------------------------------------------
#define IPX_TBC_STAGE 1024        // letters of a job's read / window rectangle staged in LDS when each has at most this many (r03)
#define IPX_TBC_DIR_LDS 8192      // direction bytes of a job kept in LDS while there are at most this many (the walk back is a chain of dependent reads)
static inline int ipx_tbc_lds_bytes(int arrcap) { return 64 + 512 + 2 * IPX_TBC_STAGE + 16 * arrcap + IPX_TBC_DIR_LDS; }

IPX_KERNEL_WAVE void k_tb_coop(IpxBatch b, const uint32_t *list, const uint32_t *list_n, uint8_t *dir_scratch,
                               int64_t dircap, int arrcap_g, uint32_t *cig_scratch, int cigcap, int32_t *band_scratch, int arrcap_lds)
{
    IPX_RAISE_PRIO(b);
    const int lane = lane_id();
    unsigned char *lds = IPX_LDS_BASE;
    int8_t *matl = (int8_t *)lds;
    int32_t *lastH = (int32_t *)(lds + 64);
    int32_t *lastF = lastH + 64;
    // this job's letters, staged once: a row then reads LDS where it read HBM -- with one cell per lane and three barriers per row, a
    // row WAS the latency of its global loads (r03: a typical job 0.15 -> 0.06 ms, and every doubling of the band costs that again)
    int8_t *sread = (int8_t *)(lastF + 64), *sref = sread + IPX_TBC_STAGE;
    int32_t *band_lds = (int32_t *)(sref + IPX_TBC_STAGE);
    uint8_t *dir_lds = (uint8_t *)(band_lds + 4 * arrcap_lds);
    uint8_t *dir_glob = dir_scratch + (int64_t)IPX_BID * dircap;
    uint32_t *cig = cig_scratch + (int64_t)IPX_BID * cigcap;
    if (lane < 25) matl[lane] = b.mat[lane];
    IPX_SYNC();
    const uint32_t n = *list_n;
    const int NEG = -(1 << 29);

    for (uint32_t item = (uint32_t)IPX_BID; item < n; item += (uint32_t)IPX_GDIM) {
        const int64_t jb = list[item];
        IpxResult r = b.res[jb];
        const int rid = b.ref_id[jb];
        const int fullRef = b.ref_len[rid];
        const int8_t *refp = b.refs_packed + b.refp_off[rid];
        const int8_t *readp = b.reads + b.read_off[jb] + r.read_begin1;
        const int rb = r.ref_begin1;
        const int refLen = r.ref_end1 - r.ref_begin1 + 1;             // ssw.c:897-899
        const int readLen = r.read_end1 - r.read_begin1 + 1;
        const int gapO = b.gap_open[jb], gapE = b.gap_ext[jb];
        const int g = gapO < gapE ? gapO : gapE;
        const int score = r.score1;
        const int len = refLen > readLen ? refLen : readLen;
        // r04: a job handed over by an anti-diagonal tier (k_tb_diag) starts at the band that tier could not hold (IpxBatch::tb_bw): the
        // narrower iterations are known to end in "max < score", and all they leave behind are stale direction cells, which matter only if
        // the walk back reads a cell the last iteration has not written -- then (`redo`) the job is run again from its first band, as the
        // reference does it
        int start = b.tb_diag ? (int)b.tb_bw[jb] : 0;
        for (;;) {
        int bw = start ? start : (refLen > readLen ? refLen - readLen : readLen - refLen) + 1;
        int mx = 0, width = 0, width_d = 0, extent = 0;
        bool broken = false;
        // band rows: LDS when 2 * len + 8 entries fit there, else this block's region of the global scratch
        const int need = 2 * (len > 0 ? len : 1) + 8;
        const bool in_lds = need <= arrcap_lds;
        const int arrcap = in_lds ? arrcap_lds : arrcap_g;
        int32_t *hb = in_lds ? band_lds : band_scratch + (size_t)IPX_BID * 4u * (size_t)arrcap_g;
        int32_t *eb = hb + arrcap;
        int32_t *hc = eb + arrcap;
        int32_t *en = hc + arrcap;
        for (int q = lane; q < (need < arrcap ? need : arrcap); q += 64) { hb[q] = 0; eb[q] = 0; hc[q] = 0; en[q] = 0; }
        const bool staged = readLen <= IPX_TBC_STAGE && refLen <= IPX_TBC_STAGE;
        if (staged) {
            for (int q = lane; q < readLen; q += 64) { const int a = readp[q]; sread[q] = (int8_t)((unsigned)a > 4u ? 4 : a); }
            for (int q = lane; q < refLen; q += 64) { const int ri = rb + q; sref[q] = (ri >= 0 && ri < fullRef) ? refp[ri] : (int8_t)0; }
        }
        IPX_SYNC();

        uint8_t *dir = dir_lds;                                   // (moves to the global scratch when a doubled band outgrows the LDS region)
        do {
            width = bw * 2 + 3;
            width_d = bw * 2 + 1;
            const int64_t cells = (int64_t)width_d * (readLen > 0 ? readLen : 1);
            if (width + 1 > arrcap || cells > dircap) { broken = true; break; }
            if (dir == dir_lds && cells > IPX_TBC_DIR_LDS) {       // what the narrower iterations left behind goes along (ssw.c:610: one buffer)
                for (int q = lane; q < extent; q += 64) dir_glob[q] = dir_lds[q];
                dir = dir_glob;
                IPX_SYNC();
            }
            for (int64_t q = extent + lane; q < cells; q += 64) dir[q] = 0;
            if ((int)cells > extent) extent = (int)cells;
            for (int j = 1 + lane; j < width - 1; j += 64) hb[j] = 0;                       // ssw.c:627
            IPX_SYNC();
            for (int i = 0; i < readLen; ++i) {
                int beg = 0, end = refLen - 1;
                if (i - bw > beg) beg = i - bw;
                if (i + bw < end) end = i + bw;
                const int edge = end + 1 < width - 1 ? end + 1 : width - 1;                // ssw.c:632
                const int x = beg;                                                         // band shift of row i
                const int xp = i - 1 - bw > 0 ? i - 1 - bw : 0;                            // ... of row i-1
                if (lane == 0) { hb[0] = 0; eb[0] = 0; hb[edge] = 0; eb[edge] = 0; hc[0] = 0; }   // ssw.c:633
                IPX_SYNC();
                int rc = staged ? (int)sread[i] : (int)readp[i];
                if ((unsigned)rc > 4u) rc = 4;
                const int ncell = end - beg + 1;
                const int cp = (ncell + 63) / 64;                                          // cells per lane
                const int c0 = lane * cp < ncell ? lane * cp : ncell;
                const int c1 = c0 + cp < ncell ? c0 + cp : ncell;                          // this lane's chunk [c0, c1)
                // sweep 1: fold the chunk into (dec, agg): f_out = max(f_in - dec, agg)
                int agg = NEG;
                for (int c = c0; c < c1; ++c) {
                    const int j = beg + c, e = j - xp + 1;
                    const int t1 = i == 0 ? -gapO : hb[e] - gapO;
                    const int t2 = i == 0 ? -gapE : eb[e] - gapE;
                    const int ev = t1 > t2 ? t1 : t2;
                    const int ri = rb + j;
                    const int rcode = staged ? (int)sref[j] : ((ri >= 0 && ri < fullRef) ? refp[ri] : 0);
                    int a = hb[e - 1] + matl[rcode * 5 + rc];
                    if (ev > a) a = ev;
                    if (a < 0) a = 0;                                                      // A_j
                    agg = (agg - g > a - gapO) ? agg - g : a - gapO;                       // F carried out after this cell
                }
                int dec = (c1 - c0) * g;
                // inclusive scan of (dec, agg) over the lanes: DPP row shifts inside each 16-lane row, then the
                // three row totals (v_readlane) are folded into the rows behind them
                const int CAPD = 1 << 28;
#define IPX_SCAN_STEP(N)                                                                      \
                {                                                                             \
                    const int pd_ = (int)xl_row_shr<N>((uint32_t)dec), pa_ = (int)xl_row_shr<N>((uint32_t)agg); \
                    if ((lane & 15) >= N) {                                                   \
                        const int cand_ = pa_ - dec;                                          \
                        agg = cand_ > agg ? cand_ : agg;                                      \
                        dec = dec + pd_ > CAPD ? CAPD : dec + pd_;                            \
                    }                                                                         \
                }
                IPX_SCAN_STEP(1) IPX_SCAN_STEP(2) IPX_SCAN_STEP(4) IPX_SCAN_STEP(8)
#undef IPX_SCAN_STEP
                const int d0 = (int)xl_readlane<15>((uint32_t)dec), a0r = (int)xl_readlane<15>((uint32_t)agg);
                const int d1 = (int)xl_readlane<31>((uint32_t)dec), a1r = (int)xl_readlane<31>((uint32_t)agg);
                const int d2 = (int)xl_readlane<47>((uint32_t)dec), a2r = (int)xl_readlane<47>((uint32_t)agg);
                // prefix (everything before row q): P1 = T0, P2 = T0 then T1, P3 = P2 then T2
                const int p1d = d0, p1a = a0r;
                const int p2d = d0 + d1 > CAPD ? CAPD : d0 + d1, p2a = (a0r - d1 > a1r) ? a0r - d1 : a1r;
                const int p3d = p2d + d2 > CAPD ? CAPD : p2d + d2, p3a = (p2a - d2 > a2r) ? p2a - d2 : a2r;
                const int rowq = lane >> 4;
                const int ppd = rowq == 1 ? p1d : rowq == 2 ? p2d : p3d, ppa = rowq == 1 ? p1a : rowq == 2 ? p2a : p3a;
                // exclusive value for this lane = inclusive value of lane-1 (row-local, before the prefix is folded in)
                int pdx = (int)xl_row_shr<1>((uint32_t)dec), pax = (int)xl_row_shr<1>((uint32_t)agg);
                if ((lane & 15) == 0) { pdx = 0; pax = NEG; }                              // nothing inside the row before lane 16q
                if (rowq > 0) {                                                            // prefix first, then the row-local part
                    const int cand = ppa - pdx;
                    pax = cand > pax ? cand : pax;
                    pdx = pdx + ppd > CAPD ? CAPD : pdx + ppd;
                }
                // h_c[0] = 0 and f = 0 before the row's first cell -> the F entering cell 0 is max(-gapO, -gapE) = -g
                int fcur = -g;
                if (lane > 0) { const int a0 = -g - pdx; fcur = a0 > pax ? a0 : pax; }
                // sweep 2: H, directions, new band rows
                int hprev = 0, fprev = 0;                                                  // (H, F) of the cell to the left
                for (int c = c0; c < c1; ++c) {
                    const int j = beg + c, u = j - x + 1, e = j - xp + 1;
                    const int t1 = i == 0 ? -gapO : hb[e] - gapO;                          // ssw.c:644-648
                    const int t2 = i == 0 ? -gapE : eb[e] - gapE;
                    const int ev = t1 > t2 ? t1 : t2;
                    const int de = t1 > t2 ? 1 : 0;
                    int df = 0;
                    if (c > c0) df = (hprev - gapO > fprev - gapE) ? 1 : 0;                // ssw.c:650-653 (first cell: below)
                    const int e1 = ev > 0 ? ev : 0;                                        // ssw.c:655-664
                    const int f1 = fcur > 0 ? fcur : 0;
                    const int tt1 = e1 > f1 ? e1 : f1;
                    const int ri = rb + j;
                    const int rcode = staged ? (int)sref[j] : ((ri >= 0 && ri < fullRef) ? refp[ri] : 0);
                    const int tt2 = hb[e - 1] + matl[rcode * 5 + rc];
                    const int hv = tt1 > tt2 ? tt1 : tt2;
                    const int dh = tt1 <= tt2 ? 0 : (e1 > f1 ? 1 : 2);
                    en[u] = ev;
                    hc[u] = hv;
                    if (hv > mx) mx = hv;
                    dir[(int64_t)width_d * i + (j - x)] = (uint8_t)(0x80 | (dh << 2) | (df << 1) | de);
                    hprev = hv; fprev = fcur;
                    const int o1 = hv - gapO, o2 = fcur - gapE;
                    fcur = o1 > o2 ? o1 : o2;
                }
                lastH[lane] = hprev; lastF[lane] = fprev;                                  // hand-over to the next lane
                IPX_SYNC();
                if (c0 < c1) {                                                             // F direction of the chunk's first cell
                    int hl = 0, fl = 0;                                                    // before the row: h_c[0] = 0, f = 0
                    if (c0 > 0) { hl = lastH[lane - 1]; fl = lastF[lane - 1]; }
                    if (hl - gapO > fl - gapE) dir[(int64_t)width_d * i + c0] |= 2;
                }
                for (int uu = 1 + lane; uu <= ncell; uu += 64) { hb[uu] = hc[uu]; eb[uu] = en[uu]; }   // ssw.c:666; e_b as in place
                IPX_SYNC();
            }
            mx = (int)wave_umax((uint32_t)mx);                                             // every lane saw only its own cells
            bw *= 2;
        } while (mx < score && bw <= len);                                                // ssw.c:669
        if (broken) { if (lane == 0) atomic_or_u32(b.status, IPX_STATUS_TB_SCRATCH); break; }
        bw /= 2;

        // ---- trace back (ssw.c:673-751), lane 0 ----
        IPX_SYNC();
        int redo = 0;
        if (lane == 0) {
            int i = readLen - 1, j = refLen - 1, e = 0, lcnt = 0, plane = 2, op = 0, prev = 0;
            bool fail = false, full = false;
            while (i >= 0 && j > 0) {
                const int x = i - bw > 0 ? i - bw : 0;
                const int64_t cell = (int64_t)width_d * i + (j - x);
                int code = 0;
                if (cell >= 0 && cell < extent) {
                    const int v = dir[cell];
                    if (v & 0x80) {
                        const int de = 2 + (v & 1), df = 4 + ((v >> 1) & 1), dh = (v >> 2) & 3;
                        code = plane == 0 ? de : plane == 1 ? df : (dh == 0 ? 1 : dh == 1 ? de : df);
                    }
                }
                if (code == 1) { --i; --j; plane = 2; op = 0; }
                else if (code == 2) { --i; plane = 0; op = 1; }
                else if (code == 3) { --i; plane = 2; op = 1; }
                else if (code == 4) { --j; plane = 1; op = 2; }
                else if (code == 5) { --j; plane = 2; op = 2; }
                else { fail = true; break; }
                if (op == prev) ++e;
                else {
                    ++lcnt;
                    if (lcnt + 2 > cigcap) { full = true; break; }
                    cig[lcnt - 1] = ((uint32_t)e << 4) | (uint32_t)prev;
                    prev = op;
                    e = 1;
                }
            }
            if (full) atomic_or_u32(b.status, IPX_STATUS_TB_SCRATCH);
            else if (fail && start != 0) redo = 1;
            else if (fail) { r.flag = 1; r.cigar_len = 0; b.res[jb] = r; }                 // ssw.c:911
            else {
                if (op == 0) { ++lcnt; cig[lcnt - 1] = ((uint32_t)(e + 1) << 4); }         // ssw.c:734-751
                else { lcnt += 2; cig[lcnt - 2] = ((uint32_t)e << 4) | (uint32_t)op; cig[lcnt - 1] = (1u << 4); }
                const uint32_t off = atomic_add_u32(b.cigar_cursor, (uint32_t)lcnt);
                if (off + (uint32_t)lcnt > b.cigar_cap) atomic_or_u32(b.status, IPX_STATUS_CIGAR_POOL);
                else {
                    for (int k = 0; k < lcnt; ++k) b.cigar_pool[off + k] = cig[lcnt - 1 - k];   // reverse (ssw.c:754-762)
                    r.cigar_off = off;
                    r.cigar_len = (uint16_t)lcnt;
                    b.res[jb] = r;
                }
            }
        }
        redo = (int)xl_first((uint32_t)redo);
        IPX_SYNC();
        if (!redo) break;
        start = 0;
        }
    }
}
// ------------------------------------------------------------------------------------------------
// k_tb_diag<LG> (r04): banded_sw (ssw.c:588-772) as an ANTI-DIAGONAL WAVEFRONT, LG = 16 / 32 / 64 lanes per job (4 / 2 / 1 jobs per wave).
// The tiers between the lane-per-job kernels (first band <= 7, no help for a small batch: one lane walks a whole job) and the
// wave-per-job kernel (k_tb_coop: three barriers and a scan over 64 lanes per DP row, whatever the band): bands of half-width
// 8..15 / ..31 / ..63, and EVERY job of a small batch.
//   A band of half-width bw takes bw + 1 lanes: lane k owns the band diagonals q = 2k and 2k+1 (q = j - i + bw, 0..2bw).  A cell (i, q) sits
//   on anti-diagonal tau = 2i + q; at step tau lane k works on row i = tau / 2 - k of diagonal 2k + (tau & 1).  Everything a cell needs is
//   one or two steps old: the upper neighbour (i-1, q+1) and the left one (i, q-1) are on anti-diagonal tau - 1 -- in the lane's own
//   other diagonal, or in the neighbouring lane's (one DPP shift) -- the diagonal one (i-1, q) on tau - 2 in the lane's own registers.
//   So a step is straight-line code without any scan, barrier or memory dependency: 2 * rows + 2 * bw steps for a band, H / E / F of
//   the last two anti-diagonals in six registers per lane.
//   SEVERAL BAND WIDTHS AT ONCE.  The reference doubles the band until the banded maximum reaches the score (ssw.c:668-669), and a narrow
//   band has nothing to run in parallel but its bw + 1 anti-diagonal cells: run one after the other, the iterations bw, 2bw, 4bw.. are a
//   chain of 2 * rows dependent steps EACH.  The lanes a narrow band leaves idle take the next widths of the doubling sequence in the same
//   steps -- bands 1, 2, 4 are 2 + 3 + 5 lanes of a 16-lane group; 1, 2, 4, 8 are 19 of 32; 1..16 are 36 of 64 -- each sub-band with its
//   own lanes, registers and direction words; afterwards the first width whose maximum reaches the score (or that is the sequence's last,
//   2bw > len) is the reference's final iteration and its directions are walked.  Widths nobody needed cost nothing but idle lanes' work.
//   Out-of-band and out-of-rectangle neighbours read 0, as the reference's h_b / e_b / h_c arrays do (ssw.c:627, 633) -- which also gives
//   row 0 its -gapO / -gapE seeds (ssw.c:644-645) -- with the reference's one irregularity kept: `h_b[edge] = e_b[edge] = 0` (ssw.c:632-633)
//   wipes the upper neighbour of the LAST window column in rows 1..bw+1 when the window is no wider than the band arrays (refLen <= 2bw + 2)
//   although that neighbour is inside the band.  (The formulation "cell by cell with these neighbour rules" was checked against a literal
//   transcription of the reference's loops on 40 000 random rectangles before the kernel was written.)
//   Directions: one nibble per cell (k_tb_fast's code: 0 = never written, else 1 + 4 * Hsrc + 2 * Fopen + Eopen), eight steps = four rows
//   x two diagonals per 32-bit word, words in LDS [hi / 4][lane] with hi = row + lane-in-band: the walk back (first lane of the group,
//   ssw.c:673-751) finds cell (row, slot) -- the reference's linear cell index, so out-of-band reads alias as there -- at lane q / 2 of the
//   band, word (row + q / 2) / 4.
//   Only the FINAL iteration's cells exist here: a walk that meets an unwritten cell after an earlier, narrower iteration (in this kernel
//   or a previous tier) hands the job to k_tb_coop, which keeps the reference's one buffer across iterations; without an earlier
//   iteration an unwritten cell is the reference's "Trace back error" (flag 1), as everywhere.
//   A band that outgrows the tier (max < score and the next width > LG - 1) is handed to the next tier with the band to start from
//   (IpxBatch::tb_bw): the narrower iterations' only lasting effect is the stale cells the previous paragraph deals with.
// Dynamic LDS: 64 B score table | per group: row score words 8 * IPX_TBD_ROWS | window letters IPX_TBD_ROWS + LG | direction words | IPX_TBD_CIG runs
// ------------------------------------------------------------------------------------------------
IPX_HD constexpr int ipx_tbd_group_bytes(int lg) { return 8 * IPX_TBD_ROWS + (IPX_TBD_ROWS + lg) + ((IPX_TBD_ROWS + lg) / 4) * lg * 4 + IPX_TBD_CIG * 4; }
IPX_HD constexpr int ipx_tbd_lds_bytes(int lg) { return 64 + (64 / lg) * ipx_tbd_group_bytes(lg); }
#define IPX_TBD_MAXSUB 5          // band widths run side by side at most (1, 2, 4, 8, 16 in 36 of 64 lanes)

template <int LG> IPX_DEV uint32_t tbd_from_lower(uint32_t v) { return LG == 16 ? xl_row_shr1(v) : xl_wave_shr1(v); }      // lane k <- lane k-1
template <int LG> IPX_DEV uint32_t tbd_from_upper(uint32_t v) { return LG == 16 ? xl_row_shl1(v) : xl_wave_shl1(v); }      // lane k <- lane k+1
template <int LG> IPX_DEV uint32_t tbd_group_umax(uint32_t x)
{
    if (LG == 16) return group_umax<16>(x);
    for (int s = 1; s < LG; s <<= 1) { const uint32_t y = xl_shfl(x, lane_id() ^ s); x = x > y ? x : y; }
    return x;
}

template <int LG>
IPX_KERNEL_WAVE void k_tb_diag(IpxBatch b, const uint32_t *list, const uint32_t *list_n, uint32_t *next, uint32_t *next_n, uint32_t *coop, uint32_t *coop_n)
{
    IPX_RAISE_PRIO(b);
    constexpr int NG = 64 / LG, ROWS = IPX_TBD_ROWS, REFCAP = ROWS + LG, NW = (ROWS + LG) / 4, GB = ipx_tbd_group_bytes(LG);
    const int lane = lane_id(), k = lane % LG, grp = lane / LG;
    unsigned char *lds = IPX_LDS_BASE;
    uint64_t *coltab = (uint64_t *)lds;                             // [read letter a] -> bytes mat[c][a], c = 0..4
    uint64_t *srow = (uint64_t *)(lds + 64 + grp * GB);             // [row] -> the row's scores against the five window letters (coltab[read letter])
    int8_t *sref = (int8_t *)(srow + ROWS);                         // [column] -> window letter TIMES EIGHT (a shift count into the row's score word)
    uint32_t *dirw = (uint32_t *)(sref + REFCAP);                   // [hi / 4][lane of the group]
    uint32_t *cig = dirw + NW * LG;
    if (lane < 5) {
        uint64_t t = 0;
        for (int c = 0; c < 5; ++c) t |= (uint64_t)(uint8_t)b.mat[c * 5 + lane] << (8 * c);
        coltab[lane] = t;
    }
    IPX_SYNC();
    const uint32_t n = *list_n;

    for (uint32_t base = (uint32_t)IPX_BID * NG; base < n; base += (uint32_t)IPX_GDIM * NG) {      // (uniform: n > 0 inside)
        const uint32_t item = base + (uint32_t)grp;
        const bool has = item < n;
        const int64_t jb = has ? (int64_t)list[item] : (int64_t)list[base];
        IpxResult r = b.res[jb];
        const int rid = b.ref_id[jb];
        const int fullRef = b.ref_len[rid];
        const int8_t *refp = b.refs_packed + b.refp_off[rid];
        const int8_t *readp = b.reads + b.read_off[jb] + r.read_begin1;
        const int rb = r.ref_begin1;
        const int refLen = r.ref_end1 - r.ref_begin1 + 1;             // ssw.c:897-899
        const int readLen = r.read_end1 - r.read_begin1 + 1;
        const int gapO = b.gap_open[jb], gapE = b.gap_ext[jb];
        const int score = r.score1;
        const int len = refLen > readLen ? refLen : readLen;
        const int start = b.tb_bw[jb];                                // 0: the job's own first band; else an earlier tier went up to start / 2
        int bw = start ? start : (refLen > readLen ? refLen - readLen : readLen - refLen) + 1;
        // 0 = this tier's, 1 = hand over to the next tier (band too wide), 2 = k_tb_coop from its first band, 3 = no job in this group
        int route = !has ? 3 : (readLen < 1 || refLen < 1 || readLen > ROWS || refLen > REFCAP) ? 2 : 0;
        if (route == 0) {                                             // the job's letters, staged once (codes outside 0..4 -> N as everywhere)
            for (int q = k; q < readLen; q += LG) { const int a = readp[q]; srow[q] = coltab[(unsigned)a > 4u ? 4 : a]; }
            for (int q = k; q < refLen; q += LG) { const int ri = rb + q; sref[q] = (int8_t)(((ri >= 0 && ri < fullRef) ? refp[ri] : 0) * 8); }
        }
        IPX_SYNC();

        int fin_bw = 0, fin_base = 0;                                 // the final iteration: its band and the first lane of its sub-band
        bool settled = false, earlier = start != 0;                   // earlier: some narrower iteration ran before the final one
        for (;;) {                                                    // rounds of band iterations (ssw.c:624-669); groups that are done idle
            bool run = route == 0 && !settled;
            if (run && bw > LG - 1) { route = 1; run = false; }
            if (!xl_any(run)) break;
            // the widths of this round: bw, 2bw, .. while they fit into the group's lanes and the reference's loop could still reach them
            // (an iteration of width w runs only if w <= len, ssw.c:669)
            int nsub = 0, wlast = 0, used = 0;
            int sbw = 0, sbase = 0, sidx = -1;                        // this lane's sub-band: width, first lane, index
            if (run) {
                int w = bw;
                while (nsub < IPX_TBD_MAXSUB && used + w + 1 <= LG && (nsub == 0 || w <= len)) {
                    if (k >= used && k <= used + w) { sbw = w; sbase = used; sidx = nsub; }
                    used += w + 1; wlast = w; ++nsub;
                    w *= 2;
                }
            }
            const int kl = k - sbase;                                 // lane within its sub-band
            const bool mine = sidx >= 0;
            const int TH = (int)xl_first(wave_umax(run ? (uint32_t)(readLen + wlast) : 0u));     // hi = row + lane-in-band = 0 .. readLen - 1 + wlast
            // rows for which this lane's two cells exist: i >= 0, i < readLen, 0 <= j < refLen with j = i + 2 kl (+1) - sbw
            int lo0 = sbw - 2 * kl; if (lo0 < 0) lo0 = 0;
            int hi0 = refLen + sbw - 2 * kl; if (hi0 > readLen) hi0 = readLen;
            int lo1 = sbw - 2 * kl - 1; if (lo1 < 0) lo1 = 0;
            int hi1 = refLen + sbw - 2 * kl - 1; if (hi1 > readLen) hi1 = readLen;
            const uint32_t n0 = mine && hi0 > lo0 ? (uint32_t)(hi0 - lo0) : 0u;                  // even diagonal: rows lo0 .. hi0 - 1
            const uint32_t n1 = mine && kl < sbw && hi1 > lo1 ? (uint32_t)(hi1 - lo1) : 0u;      // odd diagonal (the band's last lane has none)
            lo0 += kl; lo1 += kl;                                     // ... as values of hi
            // the row whose last-column upper neighbour the reference wipes (see above), as a value of hi; -1: none
            int hq0 = -1, hq1 = -1;
            if (mine && refLen <= 2 * sbw + 2) {
                const int iq0 = refLen - 1 - 2 * kl + sbw, iq1 = iq0 - 1;                          // row in which the lane's even / odd cell is in the last column
                if (iq0 >= 1 && iq0 <= sbw + 1) hq0 = iq0 + kl;
                if (iq1 >= 1 && iq1 <= sbw + 1) hq1 = iq1 + kl;
            }
            const uint32_t lowmask = kl == 0 ? 0u : 0xFFFFFFFFu;      // nothing enters a band's first lane from below (the lane there is another band's)
            int He = 0, Ee = 0, Fe = 0, Ho = 0, Eo = 0, Fo = 0, mx = 0;
            uint32_t word = 0;
            // letters one step ahead: the row's score word and the window letter of the odd cell (= the even cell's of the next row)
            const int jbase = 2 * kl - sbw - kl;                      // j0 = hi + jbase
            // (loaded whether the cell exists or not, from a clamped index: the score of a cell that does not exist is never used)
            auto ref8 = [&](int j) -> int { return (int)sref[j < 0 ? 0 : j > REFCAP - 1 ? REFCAP - 1 : j] & 56; };
            auto row8 = [&](int i2) -> uint64_t { return srow[i2 < 0 ? 0 : i2 > ROWS - 1 ? ROWS - 1 : i2]; };
            uint64_t mrow_n = row8(-kl);
            int rc0 = ref8(jbase), rc1 = ref8(jbase + 1);
            for (int hi = 0; hi < TH; ++hi) {
                const uint64_t mrow = mrow_n;
                mrow_n = row8(hi + 1 - kl);                            // next step's letters (independent of the recurrence: the loads overlap it)
                const int rcn = ref8(hi + jbase + 2);
                // ---- even step: diagonal 2 kl ----
                {
                    const int Hl = (int)(tbd_from_lower<LG>((uint32_t)Ho) & lowmask), Fl = (int)(tbd_from_lower<LG>((uint32_t)Fo) & lowmask);
                    const bool valid = (uint32_t)(hi - lo0) < n0;
                    int Hu = Ho, Eu = Eo;
                    if (hi == hq0) { Hu = 0; Eu = 0; }
                    int t1 = Hu - gapO, t2 = Eu - gapE;                 // ssw.c:644-648 (row 0: the neighbours above are 0)
                    const int ev = t1 > t2 ? t1 : t2, de = t1 > t2 ? 1 : 0;
                    t1 = Hl - gapO; t2 = Fl - gapE;                     // ssw.c:650-653
                    const int fv = t1 > t2 ? t1 : t2, df = t1 > t2 ? 2 : 0;
                    const int e1 = ev > 0 ? ev : 0, f1 = fv > 0 ? fv : 0;   // ssw.c:655-664
                    t1 = e1 > f1 ? e1 : f1;
                    t2 = He + (int)(int8_t)(mrow >> rc0);
                    const int hv = t1 > t2 ? t1 : t2;
                    const int dh = t1 <= t2 ? 0 : (e1 > f1 ? 4 : 8);
                    He = valid ? hv : 0; Ee = valid ? ev : 0; Fe = valid ? fv : 0;
                    mx = He > mx ? He : mx;
                    word |= (valid ? (uint32_t)(1 + dh + df + de) : 0u) << (8 * (hi & 3));
                }
                // ---- odd step: diagonal 2 kl + 1 ----
                {
                    int Hu = (int)tbd_from_upper<LG>((uint32_t)He), Eu = (int)tbd_from_upper<LG>((uint32_t)Ee);
                    const bool valid = (uint32_t)(hi - lo1) < n1;
                    if (hi == hq1) { Hu = 0; Eu = 0; }
                    int t1 = Hu - gapO, t2 = Eu - gapE;
                    const int ev = t1 > t2 ? t1 : t2, de = t1 > t2 ? 1 : 0;
                    t1 = He - gapO; t2 = Fe - gapE;
                    const int fv = t1 > t2 ? t1 : t2, df = t1 > t2 ? 2 : 0;
                    const int e1 = ev > 0 ? ev : 0, f1 = fv > 0 ? fv : 0;
                    t1 = e1 > f1 ? e1 : f1;
                    t2 = Ho + (int)(int8_t)(mrow >> rc1);
                    const int hv = t1 > t2 ? t1 : t2;
                    const int dh = t1 <= t2 ? 0 : (e1 > f1 ? 4 : 8);
                    Ho = valid ? hv : 0; Eo = valid ? ev : 0; Fo = valid ? fv : 0;
                    mx = Ho > mx ? Ho : mx;
                    word |= (valid ? (uint32_t)(1 + dh + df + de) : 0u) << (8 * (hi & 3) + 4);
                }
                rc0 = rc1; rc1 = rcn;
                if ((hi & 3) == 3 || hi == TH - 1) { if (run) dirw[(hi >> 2) * LG + k] = word; word = 0; }
            }
            // the first width of the sequence whose maximum reaches the score, or the sequence's last one (ssw.c:668-669)
            int pick = -1;
            {
                int w = bw;
                for (int m = 0; m < IPX_TBD_MAXSUB; ++m) {
                    const int mxm = (int)tbd_group_umax<LG>(sidx == m ? (uint32_t)mx : 0u);
                    if (run && pick < 0 && m < nsub) {
                        if (mxm >= score || 2 * w > len) { pick = m; fin_bw = w; }
                        else { earlier = true; w *= 2; }
                    }
                }
                if (run) {
                    if (pick >= 0) { settled = true; fin_base = 0; int ww = bw; for (int m = 0; m < pick; ++m) { fin_base += ww + 1; ww *= 2; } }
                    else bw = w;                                      // every width of the round fell short: the next round starts at twice the last
                }
            }
        }
        IPX_SYNC();                                                   // direction words visible to the group's first lane

        // ---- trace back (ssw.c:673-751): the group's first lane ----
        if (k == 0 && route == 0) {
            const int fbw = fin_bw;
            const int WD = 2 * fbw + 1;
            const int nwords = (readLen + fbw + 3) >> 2;              // words a lane of the final band wrote
            const uint32_t *dw = dirw + fin_base;
            int i = readLen - 1, j = refLen - 1, e = 0, lcnt = 0, plane = 2, op = 0, prev = 0;
            int cidx = -1;                                            // the direction word held in cw (four rows of one lane: a run of matches reads LDS once in four steps)
            uint32_t cw = 0;
            bool fail = false, full = false;
            while (i >= 0 && j > 0) {
                // the reference's linear cell index width_d * i + (j - x): a column outside the row's band aliases into a neighbouring row
                int row = i, slot = j - (i - fbw > 0 ? i - fbw : 0);
                while (slot < 0) { slot += WD; --row; }
                while (slot >= WD) { slot -= WD; ++row; }
                int code = 0;
                if (row >= 0 && row < readLen) {
                    const int q = slot + (fbw - row > 0 ? fbw - row : 0);        // band diagonal of that slot in that row
                    const int lk = q >> 1, hh = row + lk;
                    if (q <= 2 * fbw && (hh >> 2) < nwords) {
                        const int widx = (hh >> 2) * LG + lk;
                        if (widx != cidx) { cw = dw[widx]; cidx = widx; }
                        // A RUN OF MATCHES in one go: the word holds the lane's cells of rows hh&~3 .. hh (same diagonal: a diagonal move keeps
                        // q), codes 1..4 = "H came from the diagonal".  While the walk is in the H plane and stays inside the rectangle
                        // (ssw.c:679: i >= 0 && j > 0 before every step) those steps are all case 1 (ssw.c:681-687).
                        if (plane == 2 && row == i) {
                            const int avail = (hh & 3) + 1;
                            const uint32_t t = (cw >> (4 * (q & 1))) & 0x0F0F0F0Fu;
                            uint32_t bad = (((t + 0x03030303u) & 0x0C0C0C0Cu) ^ 0x04040404u) << (8 * (4 - avail));   // byte != 0: not a diagonal code; this row's byte on top
                            int rn = bad ? (int)(__builtin_clz(bad) >> 3) : avail;
                            if (rn > avail) rn = avail;
                            if (rn > i + 1) rn = i + 1;
                            if (rn > j) rn = j;
                            if (rn > 1) {
                                i -= rn; j -= rn; op = 0;
                                if (prev == 0) e += rn;
                                else {
                                    ++lcnt;
                                    if (lcnt + 2 > IPX_TBD_CIG) { full = true; break; }
                                    cig[lcnt - 1] = ((uint32_t)e << 4) | (uint32_t)prev;
                                    prev = 0;
                                    e = rn;
                                }
                                continue;
                            }
                        }
                        const int v = (int)((cw >> (8 * (hh & 3) + 4 * (q & 1))) & 15u);
                        if (v) {
                            const int de = 2 + ((v - 1) & 1), df = 4 + (((v - 1) >> 1) & 1), dh = (v - 1) >> 2;
                            code = plane == 0 ? de : plane == 1 ? df : (dh == 0 ? 1 : dh == 1 ? de : df);
                        }
                    }
                }
                if (code == 1) { --i; --j; plane = 2; op = 0; }
                else if (code == 2) { --i; plane = 0; op = 1; }
                else if (code == 3) { --i; plane = 2; op = 1; }
                else if (code == 4) { --j; plane = 1; op = 2; }
                else if (code == 5) { --j; plane = 2; op = 2; }
                else { fail = true; break; }
                if (op == prev) ++e;
                else {
                    ++lcnt;
                    if (lcnt + 2 > IPX_TBD_CIG) { full = true; break; }
                    cig[lcnt - 1] = ((uint32_t)e << 4) | (uint32_t)prev;
                    prev = op;
                    e = 1;
                }
            }
            if (full || (fail && earlier)) route = 2;                                // (stale cells of narrower iterations could matter: k_tb_coop keeps them)
            else if (fail) { r.flag = 1; r.cigar_len = 0; b.res[jb] = r; }          // ssw.c:711-719, 911
            else {
                if (op == 0) { ++lcnt; cig[lcnt - 1] = ((uint32_t)(e + 1) << 4); }  // ssw.c:734-751
                else { lcnt += 2; cig[lcnt - 2] = ((uint32_t)e << 4) | (uint32_t)op; cig[lcnt - 1] = (1u << 4); }
                const uint32_t off = atomic_add_u32(b.cigar_cursor, (uint32_t)lcnt);
                if (off + (uint32_t)lcnt > b.cigar_cap) atomic_or_u32(b.status, IPX_STATUS_CIGAR_POOL);
                else {
                    for (int q = 0; q < lcnt; ++q) b.cigar_pool[off + q] = cig[lcnt - 1 - q];   // reverse (ssw.c:754-762)
                    r.cigar_off = off;
                    r.cigar_len = (uint16_t)lcnt;
                    b.res[jb] = r;
                }
            }
        }
        if (k == 0 && route == 1) { b.tb_bw[jb] = (uint16_t)bw; next[atomic_add_u32(next_n, 1u)] = (uint32_t)jb; }
        if (k == 0 && route == 2) { b.tb_bw[jb] = 0; coop[atomic_add_u32(coop_n, 1u)] = (uint32_t)jb; }
        IPX_SYNC();                                                   // the group's LDS is free for its next job
    }
}
#endif // IPX_AUX_KERNELS

// ------------------------------------------------------------------------------------------------
// explicit instantiation of k_dp_pass, one family = segLen 0..32 of (W, REV, STAGE, PERM)
// ------------------------------------------------------------------------------------------------
#if !defined(IPX_CPU_EMU)
#define IPX_DP_SIG (IpxBatch, IpxPlan, int, int, int, int, uint64_t, uint64_t)
#define IPX_DP_FAMILY(X, W, REV, STAGE, PERM)                                                                                   \
    X(W, 0, REV, true, STAGE, PERM) X(W, 1, REV, true, STAGE, PERM) X(W, 2, REV, true, STAGE, PERM) X(W, 3, REV, true, STAGE, PERM)       \
    X(W, 4, REV, true, STAGE, PERM) X(W, 5, REV, true, STAGE, PERM) X(W, 6, REV, true, STAGE, PERM) X(W, 7, REV, true, STAGE, PERM)       \
    X(W, 8, REV, true, STAGE, PERM) X(W, 9, REV, true, STAGE, PERM) X(W, 10, REV, true, STAGE, PERM) X(W, 11, REV, true, STAGE, PERM)     \
    X(W, 12, REV, true, STAGE, PERM) X(W, 13, REV, true, STAGE, PERM) X(W, 14, REV, true, STAGE, PERM) X(W, 15, REV, true, STAGE, PERM)   \
    X(W, 16, REV, true, STAGE, PERM) X(W, 17, REV, true, STAGE, PERM) X(W, 18, REV, true, STAGE, PERM) X(W, 19, REV, true, STAGE, PERM)   \
    X(W, 20, REV, true, STAGE, PERM) X(W, 21, REV, true, STAGE, PERM) X(W, 22, REV, true, STAGE, PERM) X(W, 23, REV, true, STAGE, PERM)   \
    X(W, 24, REV, true, STAGE, PERM) X(W, 25, REV, true, STAGE, PERM) X(W, 26, REV, true, STAGE, PERM) X(W, 27, REV, true, STAGE, PERM)   \
    X(W, 28, REV, true, STAGE, PERM) X(W, 29, REV, true, STAGE, PERM) X(W, 30, REV, true, STAGE, PERM) X(W, 31, REV, true, STAGE, PERM)   \
    X(W, 32, REV, true, STAGE, PERM)
#define IPX_DP_DEFINE(W, S, REV, EX, STAGE, PERM) template __global__ void k_dp_pass<W, S, REV, EX, STAGE, PERM> IPX_DP_SIG;
#define IPX_DP_EXTERN(W, S, REV, EX, STAGE, PERM) extern template __global__ void k_dp_pass<W, S, REV, EX, STAGE, PERM> IPX_DP_SIG;
// every family the pipeline launches (ipx_launch_dp): translation unit, then what it holds
#define IPX_DP_UNIT_A(X) IPX_DP_FAMILY(X, 16, false, IPX_STAGE_LOW, true) IPX_DP_FAMILY(X, 16, false, IPX_STAGE_LOW, false) X(16, IPX_MAX_SEG, false, false, IPX_STAGE_LOW, false)
#define IPX_DP_UNIT_B(X) IPX_DP_FAMILY(X, 16, false, IPX_STAGE_HIGH, true) IPX_DP_FAMILY(X, 16, false, IPX_STAGE_EXACT, true)
#define IPX_DP_UNIT_C(X) IPX_DP_FAMILY(X, 16, false, IPX_STAGE_EXACT, false) X(16, IPX_MAX_SEG, false, false, IPX_STAGE_EXACT, false)
#define IPX_DP_UNIT_D(X) IPX_DP_FAMILY(X, 16, true, IPX_STAGE_EXACT, true) IPX_DP_FAMILY(X, 16, true, IPX_STAGE_EXACT, false)              \
    X(16, 16, true, false, IPX_STAGE_EXACT, false) X(16, 32, true, false, IPX_STAGE_EXACT, false) X(16, IPX_MAX_SEG, true, false, IPX_STAGE_EXACT, false)
#define IPX_DP_UNIT_E(X) IPX_DP_FAMILY(X, 8, false, IPX_STAGE_EXACT, true)
#define IPX_DP_UNIT_F(X) IPX_DP_FAMILY(X, 8, false, IPX_STAGE_EXACT, false) X(8, IPX_MAX_SEG, false, false, IPX_STAGE_EXACT, false)
#define IPX_DP_UNIT_G(X) IPX_DP_FAMILY(X, 8, true, IPX_STAGE_EXACT, true)
#define IPX_DP_UNIT_H(X) IPX_DP_FAMILY(X, 8, true, IPX_STAGE_EXACT, false)                                                                 \
    X(8, 16, true, false, IPX_STAGE_EXACT, false) X(8, 32, true, false, IPX_STAGE_EXACT, false) X(8, IPX_MAX_SEG, true, false, IPX_STAGE_EXACT, false)
// the half-precision form of the 16-bit selector-profile passes (k_dp_pass F16)
#define IPX_DP_DEFINE_H(W, S, REV, EX, STAGE, PERM) template __global__ void k_dp_pass<W, S, REV, EX, STAGE, PERM, true> IPX_DP_SIG;
#define IPX_DP_EXTERN_H(W, S, REV, EX, STAGE, PERM) extern template __global__ void k_dp_pass<W, S, REV, EX, STAGE, PERM, true> IPX_DP_SIG;
#define IPX_DP_UNIT_I(X) IPX_DP_FAMILY(X, 8, false, IPX_STAGE_EXACT, true)
#define IPX_DP_UNIT_J(X) IPX_DP_FAMILY(X, 8, true, IPX_STAGE_EXACT, true)
#define IPX_DP_UNIT_M(X) IPX_DP_FAMILY(X, 16, false, IPX_STAGE_LOW, true)          // 8-bit lower-bound stage in halves
// ... and with two reference lanes per GPU lane (VL2): 8-bit classes 0..16 = 0, 2, .., 32 segments in the 8-lane layout
#define IPX_VL2_FAMILY(X) X(0) X(2) X(4) X(6) X(8) X(10) X(12) X(14) X(16) X(18) X(20) X(22) X(24) X(26) X(28) X(30) X(32)
#define IPX_VL2_DEFINE(S) template __global__ void k_dp_pass<8, S, false, true, IPX_STAGE_LOW, true, true, true> IPX_DP_SIG;
#define IPX_VL2_EXTERN(S) extern template __global__ void k_dp_pass<8, S, false, true, IPX_STAGE_LOW, true, true, true> IPX_DP_SIG;
// the wavefront form of the same passes (k_dp_skew)
#define IPX_SKEW_FAMILY(X, REV)                                                                                              \
    X(0, REV) X(1, REV) X(2, REV) X(3, REV) X(4, REV) X(5, REV) X(6, REV) X(7, REV) X(8, REV) X(9, REV) X(10, REV) X(11, REV)    \
    X(12, REV) X(13, REV) X(14, REV) X(15, REV) X(16, REV) X(17, REV) X(18, REV) X(19, REV) X(20, REV) X(21, REV) X(22, REV)     \
    X(23, REV) X(24, REV) X(25, REV) X(26, REV) X(27, REV) X(28, REV) X(29, REV) X(30, REV) X(31, REV) X(32, REV)
#define IPX_SKEW_DEFINE(S, REV) template __global__ void k_dp_skew<S, REV, 0>(IpxBatch, IpxPlan, int, int, int);
#define IPX_SKEW_EXTERN(S, REV) extern template __global__ void k_dp_skew<S, REV, 0>(IpxBatch, IpxPlan, int, int, int);
// the plain recurrence in the 8-bit dialect: 8-bit classes 1..16 = 2, 4, .., 32 segments of the 8-lane layout
#define IPX_SKEW_BH_FAMILY(X, REV, BH)                                                                                       \
    X(2, REV, BH) X(4, REV, BH) X(6, REV, BH) X(8, REV, BH) X(10, REV, BH) X(12, REV, BH) X(14, REV, BH) X(16, REV, BH)         \
    X(18, REV, BH) X(20, REV, BH) X(22, REV, BH) X(24, REV, BH) X(26, REV, BH) X(28, REV, BH) X(30, REV, BH) X(32, REV, BH)
#define IPX_SKEW_BH_DEFINE(S, REV, BH) template __global__ void k_dp_skew<S, REV, BH>(IpxBatch, IpxPlan, int, int, int);
#define IPX_SKEW_BH_EXTERN(S, REV, BH) extern template __global__ void k_dp_skew<S, REV, BH>(IpxBatch, IpxPlan, int, int, int);
// the latency tier (r04): 32 lanes per read, 1..8 segments = up to 256 rows; 16-bit passes (BH 0) and the plain recurrence in the 8-bit dialect (BH 2).
// (64 lanes per read -- two reads per wave, 1..4 segments -- was built and measured too: 50 instead of 70 instructions per step and the SAME
//  time per pass, 66 against 71 us: one wave per SIMD is bound by the dependent chain of a step, ~180 ns whatever its length.  Not kept.)
#define IPX_LAT_FAMILY(X, REV, BH) X(1, REV, BH, 32) X(2, REV, BH, 32) X(3, REV, BH, 32) X(4, REV, BH, 32) X(5, REV, BH, 32) X(6, REV, BH, 32) X(7, REV, BH, 32) X(8, REV, BH, 32)
#define IPX_LAT_DEFINE(S, REV, BH, W) template __global__ void k_dp_skew<S, REV, BH, W>(IpxBatch, IpxPlan, int, int, int);
#define IPX_LAT_EXTERN(S, REV, BH, W) extern template __global__ void k_dp_skew<S, REV, BH, W>(IpxBatch, IpxPlan, int, int, int);
#define IPX_DP_UNIT_X(X) IPX_LAT_FAMILY(X, false, 0) IPX_LAT_FAMILY(X, true, 0) IPX_LAT_FAMILY(X, false, 2) IPX_LAT_FAMILY(X, true, 2)
#define IPX_DP_UNIT_K(X) IPX_SKEW_FAMILY(X, false)
#define IPX_DP_UNIT_L(X) IPX_SKEW_FAMILY(X, true)
// the stepped 8-bit passes of classes 1..16 in one launch (k_dp_pass_tier)
#define IPX_PASS_TIER_DEFINE(REV) template __global__ void k_dp_pass_tier<16, IPX_PASS_TIER_LO, IPX_PASS_TIER_HI, REV, IPX_STAGE_EXACT>(IpxBatch, IpxPlan, uint32_t, int, int);
#define IPX_PASS_TIER_EXTERN(REV) extern template __global__ void k_dp_pass_tier<16, IPX_PASS_TIER_LO, IPX_PASS_TIER_HI, REV, IPX_STAGE_EXACT>(IpxBatch, IpxPlan, uint32_t, int, int);
#define IPX_BAND_FAMILY(X) X(8) X(9) X(10) X(11) X(12) X(13) X(14) X(15) X(16) X(17) X(18) X(19) X(20) X(21) X(22) X(23) X(24) X(25) X(26) X(27) X(28) X(29) X(30) X(31) X(32)
#define IPX_BAND_DEFINE(S) template __global__ void k_dp_band_rev<S, 0>(IpxBatch, const uint32_t *, const uint32_t *, uint32_t *, uint32_t *, int);
#define IPX_BAND_EXTERN(S) extern template __global__ void k_dp_band_rev<S, 0>(IpxBatch, const uint32_t *, const uint32_t *, uint32_t *, uint32_t *, int);
#define IPX_BAND8_FAMILY(X) X(8) X(10) X(12) X(14) X(16) X(18) X(20) X(22) X(24) X(26) X(28) X(30) X(32)
#define IPX_BAND8_DEFINE(S) template __global__ void k_dp_band_rev<S, 2>(IpxBatch, const uint32_t *, const uint32_t *, uint32_t *, uint32_t *, int);
#define IPX_BAND8_EXTERN(S) extern template __global__ void k_dp_band_rev<S, 2>(IpxBatch, const uint32_t *, const uint32_t *, uint32_t *, uint32_t *, int);
#define IPX_WIDE_FAMILY(X) X(16, false) X(16, true) X(32, false) X(32, true) X(48, false) X(48, true) X(64, false) X(64, true)
#define IPX_WIDE_DEFINE(S, REV) template __global__ void k_dp_wide<S, REV>(IpxBatch, IpxPlan, int, int);
#define IPX_WIDE_EXTERN(S, REV) extern template __global__ void k_dp_wide<S, REV>(IpxBatch, IpxPlan, int, int);
#if defined(IPX_EXTERN_KERNELS)
IPX_WIDE_FAMILY(IPX_WIDE_EXTERN)
IPX_BAND_FAMILY(IPX_BAND_EXTERN)
IPX_BAND8_FAMILY(IPX_BAND8_EXTERN)
IPX_DP_UNIT_X(IPX_LAT_EXTERN)
IPX_PASS_TIER_EXTERN(false) IPX_PASS_TIER_EXTERN(true)
IPX_DP_UNIT_K(IPX_SKEW_EXTERN) IPX_DP_UNIT_L(IPX_SKEW_EXTERN)
IPX_SKEW_BH_FAMILY(IPX_SKEW_BH_EXTERN, false, 1) IPX_SKEW_BH_FAMILY(IPX_SKEW_BH_EXTERN, false, 2) IPX_SKEW_BH_FAMILY(IPX_SKEW_BH_EXTERN, true, 2)
IPX_DP_UNIT_I(IPX_DP_EXTERN_H) IPX_DP_UNIT_J(IPX_DP_EXTERN_H) IPX_DP_UNIT_M(IPX_DP_EXTERN_H) IPX_VL2_FAMILY(IPX_VL2_EXTERN)
IPX_DP_UNIT_A(IPX_DP_EXTERN) IPX_DP_UNIT_B(IPX_DP_EXTERN) IPX_DP_UNIT_C(IPX_DP_EXTERN) IPX_DP_UNIT_D(IPX_DP_EXTERN)
IPX_DP_UNIT_E(IPX_DP_EXTERN) IPX_DP_UNIT_F(IPX_DP_EXTERN) IPX_DP_UNIT_G(IPX_DP_EXTERN) IPX_DP_UNIT_H(IPX_DP_EXTERN)
#endif
#endif
