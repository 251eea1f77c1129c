// ipx_pipeline.h -- the launch sequence of one alignment batch (the control flow of ssw_align,
// ssw.c:842-916, over a whole job table).  Written against a tiny launcher so the HIP runtime
// (ipx_runtime.hip) and the test-only wave emulator (tests/emu) run the identical sequence.
//
//   init -> [16-bit forward first + overflow proof, long reads]
//        -> [8-bit forward: plain recurrence + proof, else lower bound, else stepped   (r02 order: lower bound, upper bound, stepped)]
//        -> [16-bit forward for overflowed reads] -> [8-bit reverse: plain recurrence + proof, else stepped] -> [16-bit reverse]
//        -> traceback list -> banded traceback
//
// No host synchronisation happens between the stages: which job takes which branch is decided on the device.  The
// job lists of the two passes every job STARTS in depend only on host-known facts and are built once per resident
// batch (ipx_build_static_plans); every other pass is counted by the kernels that send jobs into it and needs one
// scatter launch (k_plan_scatter) on the stream.
#pragma once
#include <string.h>
#include "ipx_kernels.h"

struct IpxWorkspace {
    IpxPlan plan[IPX_NUM_PASSES];
    uint32_t *plan_tables;              // [IPX_NUM_PASSES][2][IPX_NUM_CLASSES]: count and cursor of every pass, the dynamic passes' part zeroed per run
    uint32_t *exact_starters;           // [IPX_NUM_CLASSES] jobs that START in the stepped 8-bit pass (host-known): seeds its count row every run
    uint32_t *tb_list, *tb_list_n;      // jobs that get a CIGAR: IPX_TB_NLISTS lists of n_jobs slots (first band 1..7; then the anti-diagonal tiers of 16 / 32 / 64
                                        //   lanes per job); IPX_TB_NCOUNTERS counters: [0..6] the widths, [7] = tb_esc_n, [8..10] the tiers
    uint32_t *tb_esc, *tb_esc_n;        // jobs the fast traceback hands to the general (one wave per job) kernel
    IpxTbScratch tb1;
    unsigned char *tbf_scratch;         // direction words of the fast traceback: ipx_tbf_scratch_bytes_per_block(rowcap) per block
    unsigned char *long_state;          // k_dp_long: striped columns of reads beyond the register kernels, long_stride bytes per block (null: no such read)
    int64_t long_stride;
    int long_blocks;
    int tbf_waves, tb1_waves;
    int tbd_waves;                      // blocks of a k_tb_diag launch
    // r04, banded reverse pass (k_rev_split / k_dp_band_rev): two job lists of n_jobs slots, their counters per class (2 words each), and the class / tile
    // offsets the full kernel's launch over the second list reads (null: no banded reverse pass)
    uint32_t *rev_listA, *rev_listB, *rev_cnt, *rev_cls_off, *rev_tile_off;
};
static inline size_t ipx_rev_words(int64_t n_jobs) { return 2 * (size_t)n_jobs + 2 * IPX_NUM_CLASSES + 2 * (IPX_NUM_CLASSES + 2) + 64; }

#ifndef IPX_PROVE_CHUNK_MIN_JOBS
#define IPX_PROVE_CHUNK_MIN_JOBS 100000   // (the emulator build sets 0 so that its small batches take the queued form)
#endif
#define IPX_TBF_ROWCAP 512  // rows of direction words per fast-traceback block
#ifndef IPX_TB_SMALL_DIAG
#define IPX_TB_TINY_DIAG 1024    // ... up to this many: one job per wave (64 lanes: five widths of the doubling sequence side by side)
#define IPX_TB_SMALL_DIAG 16384  // batches up to this many jobs: every traceback takes an anti-diagonal tier (4 jobs per wave at most) instead of a lane
#endif
#define IPX_MAX_EXACT 32     // segLen classes 0..32 have their own straight-line instantiation
#define IPX_MAX_READ_LEN IPX_LONG_MAX_READ   // longest read the library takes (beyond 8 * IPX_MAX_SEG = 512 bp: k_dp_long)

// what the host knows about a batch: which classes can occur in which pass
struct IpxDims {
    int max_read_len;                  // longest read of the batch
    int max_ref_len;                   // longest window of the batch
    uint32_t lenhist[2][IPX_MAX_READ_LEN + 1];   // reads per length; [1] = jobs with gap_open <= gap_ext ("slow": stepped lazy-F)
    // derived by ipx_dims_finish for the current scoring parameters: classes (segLen, + IPX_SLOW_BASE when slow) present
    uint8_t has8_low[IPX_NUM_CLASSES], has8_wf[IPX_NUM_CLASSES];     // 8-bit classes of the reads that start in the 8-bit / in the 16-bit-first pass
    uint8_t has16_low[IPX_NUM_CLASSES], has16_wf[IPX_NUM_CLASSES];   // 16-bit classes of the same two sets
    uint8_t any_wf, any_low;           // some read starts in the 16-bit-first pass / in the first 8-bit stage
    // derived by ipx_plan_classes: where the wavefront kernels (k_dp_skew) serve a pass, the class every class is LISTED under
    // (IpxBatch::cls_map) and the classes that get a launch
    uint8_t cls_map[IPX_NUM_PASSES][IPX_NUM_CLASSES];
    uint8_t set[IPX_NUM_PASSES][IPX_NUM_CLASSES];
    uint8_t word_sets;                 // the fast-gap classes of the three 16-bit passes are launched from `set` (wavefront kernels throughout)
    uint8_t plain_first;               // the 8-bit passes take the plain-first flow (IpxBatch::plain_first)
    uint8_t high_sets;                 // (bracket flow) the upper-bound stage runs as a wavefront, launched from `set`
    int word_from;                     // 16-bit fast-gap classes below this one are planned into `set`; from it on: class-by-class launches / k_dp_long
    int plain_max_len;                 // plain-first flow: reads up to this length take the plain kernels
    uint8_t lat;                       // latency tier (r04): 0, or the lanes per read (32) a small batch's wavefront passes run at -- 2 * 64 / lat reads per
                                       //   tile, every class of a pass in ONE launch (ipx_plan_classes)
};
#ifndef IPX_LAT_MAX_JOBS
#define IPX_LAT_MAX_JOBS 16384         // batches up to this size take the latency tier: 4 096 four-read tiles, four per SIMD (r04: 8 000 jobs 0.67 ms with the throughput kernels, 0.47 with these; 16 000: see DESIGN section 5)
#endif
// alignments per tile of the passes the wavefront kernels serve
static inline int ipx_skew_na(const IpxDims &d) { return d.lat ? 2 * (64 / d.lat) : 16; }

static inline void ipx_dims_add_read(IpxDims &d, int len, bool slow)
{
    if (len > d.max_read_len) d.max_read_len = len;
    if (len > IPX_MAX_READ_LEN) len = IPX_MAX_READ_LEN;       // longer reads are refused (upload / planner status bit)
    if (len < 0) len = 0;
    ++d.lenhist[slow ? 1 : 0][len];
}

// kernel classes for per-kernel timing (ipx_runtime.hip records HIP events around each launch)
enum {
    IPX_K_INIT = 0, IPX_K_PLAN, IPX_K_BYTE_LOW, IPX_K_BYTE_CHECK, IPX_K_BYTE_HIGH, IPX_K_BYTE_EXACT, IPX_K_WORD_FIRST, IPX_K_WORD_FWD,
    IPX_K_BYTE_REV, IPX_K_WORD_REV, IPX_K_TB_LIST, IPX_K_TRACEBACK, IPX_K_PACK, IPX_K_PROVE,
    IPX_K_BYTE_PLAIN, IPX_K_BYTE_LOW2, IPX_K_BYTE_REV_PLAIN, IPX_K_PROVE_PLAIN, IPX_K_NUM
};
static inline bool ipx_k_is_dp(int kc) { return (kc >= IPX_K_BYTE_LOW && kc <= IPX_K_WORD_REV) || (kc >= IPX_K_BYTE_PLAIN && kc <= IPX_K_BYTE_REV_PLAIN); }

// DP grid cap, in blocks (= waves) per CU.  Far more than are resident (<= 24): with several streams
// sharing the GPU, short-lived blocks let the kernels of different streams interleave and even out the
// tail (measured on config 2b, 4 streams: 12 -> 41.4, 24 -> 42.5, 48 -> 43.3, 96 -> 43.6 M aln/s).
// Each block owns a column-maxima scratch region, so the grid is also capped by IPX_DP_SCRATCH_BUDGET.
#define IPX_DP_WAVES_PER_CU 64
#define IPX_DP_SCRATCH_BUDGET ((size_t)1 << 30)
// column maxima of a forward selector-profile kernel stay in LDS while 16 waves per CU still fit (160 KB)
#define IPX_DP_MC_LDS_MAX 10112
static inline bool ipx_dp_mc_in_lds(int W, bool rev, int maxcols, bool perm, int routing)
{
    return perm && !rev && (64 / W) * maxcols * 4 <= IPX_DP_MC_LDS_MAX && !(routing & IPX_ROUTE_NO_MC_LDS);
}
static inline int ipx_dp_lds_bytes(int W, int SMAX, bool rev, int maxcols, bool perm, int routing)
{
    const int mc = ipx_dp_mc_in_lds(W, rev, maxcols, perm, routing) ? (64 / W) * maxcols * 4 : 0;
    // (r04 experiment, not kept: reserving 20 KB here caps the DP kernels at two waves per SIMD so that the other streams' latency-bound kernels
    //  find register room beside them -- config 4 55.4 -> 49.1 M aln/s, 2b 85 -> 72: the DP kernels need their three waves more)
    return (perm ? 64 : 640 * (SMAX > 0 ? SMAX : 1)) + 64 + mc;
}

// the register-selector profile (k_dp_pass PERM) needs a read letter N to score 0 against every window letter
static inline bool ipx_perm_profile_ok(const int8_t *mat, int routing)
{
    return !(routing & IPX_ROUTE_NO_PERM_PROFILE) && mat[4] == 0 && mat[9] == 0 && mat[14] == 0 && mat[19] == 0 && mat[24] == 0;
}

// The 8-bit lower-bound stage at 16 reads per wave (k_dp_pass VL2: two reference lanes per GPU lane, computed in halves): when the
// register-selector profile and exact halves apply to every 8-bit class of the batch, none of them is a slow-gap class (those
// need the stepped kernels, whose tiles hold 8 reads: a pass has ONE tile size) and reads are at most 256 bp.
static inline bool ipx_low2_ok(const IpxBatch &b, const IpxDims &d, int routing)
{
    if (!ipx_perm_profile_ok(b.mat, routing) || (routing & (IPX_ROUTE_NO_F16 | IPX_ROUTE_NO_VL2))) return false;
    int top = 0;
    for (int c = 0; c < IPX_NUM_CLASSES; ++c)
        if (d.has8_low[c] || d.has8_wf[c]) { if (c >= IPX_SLOW_BASE) return false; top = c; }
    return top >= 1 && top <= 16 && 16 * top <= b.f16_max_len;
}

// timing key of a launch: kernel class * 256 + sub (DP kernels: sub = class, IPX_SUB_GENERIC = long-read / sweep kernel)
#define IPX_KEY(kclass, sub) ((kclass) * 256 + (sub))
#define IPX_NUM_KEYS (IPX_K_NUM * 256)
#define IPX_SUB_GENERIC 140
#define IPX_SUB_LONG 141      // k_dp_long (reads of 64 segments or more)
#define IPX_SUB_WIDE 142      // k_dp_wide (r04: the 16-bit passes of those reads with gap_open > gap_ext)
#define IPX_SUB_BAND 143      // k_dp_band_rev (r04: the 16-bit reverse pass as a band)
#ifndef IPX_BAND_MIN_TILES
#define IPX_BAND_MIN_TILES 6000   // 16-job tiles a class needs (previous run's count) to take the banded reverse pass (the emulator build sets 0)
#endif
#define IPX_SUB_TIER 150      // timing sub-key of tier t: IPX_SUB_TIER + t (wavefront tiers 0..3; 8 = the stepped 8-bit tier, k_dp_pass_tier)

template <class BE, int W, bool REV, int STAGE>
static void ipx_launch_dp_class(BE &be, const IpxBatch &b, const IpxPlan &p, int cls, int maxcols, int kclass, int pass, int routing)
{
    const bool slow = cls >= IPX_SLOW_BASE;
    const int S = slow ? cls - IPX_SLOW_BASE : cls;
    // the selector-profile kernels of the 16-bit passes and of the 8-bit bracket stages have no stepped lazy-F loop
    // (k_dp_pass, STEP): jobs with gap_open <= gap_ext are a class of their own and take the LDS-profile kernels
    const bool perm = ipx_perm_profile_ok(b.mat, routing) && (!slow || (W == 16 && STAGE == IPX_STAGE_EXACT));
    constexpr int NP_STAGE = STAGE == IPX_STAGE_HIGH ? IPX_STAGE_EXACT : STAGE;   // (the upper-bound stage exists in selector-profile form only)
    // 16-bit passes: the half-precision form where every score of the class stays exact in a half (IpxBatch::f16_max_len)
    bool f16 = false;
    if constexpr (W == 8 && STAGE == IPX_STAGE_EXACT) f16 = perm && !slow && !(routing & IPX_ROUTE_NO_F16) && 8 * S <= b.f16_max_len;
    if constexpr (W == 16 && STAGE == IPX_STAGE_LOW && !REV) f16 = perm && !slow && !(routing & IPX_ROUTE_NO_F16) && 16 * S <= b.f16_max_len;
#define IPX_DP_CASE(N)                                                                                       \
    case N:                                                                                                  \
        if constexpr (W == 8 && STAGE == IPX_STAGE_EXACT) {                                                  \
            if (f16 && !(routing & IPX_ROUTE_NO_SKEW)) {                                                     \
                be.note_f16(1, N);                                                                         \
                be.launch(IPX_KEY(kclass, cls), k_dp_skew<N, REV>, be.dp_grid(pass, cls), 64,                \
                          ipx_dp_lds_bytes(W, N, REV, maxcols, true, routing), b, p, cls, maxcols,           \
                          pass | (ipx_dp_mc_in_lds(W, REV, maxcols, true, routing) ? IPX_PASS_MC_LDS : 0)); \
                break;                                                                                       \
            }                                                                                                \
            if (f16) {                                                                                       \
                be.note_f16(0, N);                                                                              \
                be.launch(IPX_KEY(kclass, cls), k_dp_pass<W, N, REV, true, STAGE, true, true>, be.dp_grid(pass, cls), 64, \
                          ipx_dp_lds_bytes(W, N, REV, maxcols, true, routing), b, p, cls, cls, maxcols,      \
                          pass | (ipx_dp_mc_in_lds(W, REV, maxcols, true, routing) ? IPX_PASS_MC_LDS : 0), (uint64_t)0, (uint64_t)0);  \
                break;                                                                                       \
            }                                                                                                \
        }                                                                                                    \
        if constexpr (W == 16 && STAGE == IPX_STAGE_LOW && !REV && (N) <= 16) {                              \
            if (f16 && (routing & IPX_ROUTE_INTERNAL_VL2)) {                                                 \
                be.note_f16(4, N);                                                                           \
                be.launch(IPX_KEY(kclass, cls), k_dp_pass<8, 2 * (N), false, true, IPX_STAGE_LOW, true, true, true>, be.dp_grid(pass, cls), 64, \
                          ipx_dp_lds_bytes(8, 2 * (N), false, maxcols, true, routing), b, p, cls, cls, maxcols, \
                          pass | (ipx_dp_mc_in_lds(8, false, maxcols, true, routing) ? IPX_PASS_MC_LDS : 0), (uint64_t)0, (uint64_t)0);  \
                break;                                                                                       \
            }                                                                                                \
        }                                                                                                    \
        if constexpr (W == 16 && STAGE == IPX_STAGE_LOW && !REV) {                                           \
            if (f16) {                                                                                       \
                be.note_f16(3, N);                                                                              \
                be.launch(IPX_KEY(kclass, cls), k_dp_pass<W, N, REV, true, STAGE, true, true>, be.dp_grid(pass, cls), 64, \
                          ipx_dp_lds_bytes(W, N, REV, maxcols, true, routing), b, p, cls, cls, maxcols,      \
                          pass | (ipx_dp_mc_in_lds(W, REV, maxcols, true, routing) ? IPX_PASS_MC_LDS : 0), (uint64_t)0, (uint64_t)0);  \
                break;                                                                                       \
            }                                                                                                \
        }                                                                                                    \
        if (perm)                                                                                            \
            be.launch(IPX_KEY(kclass, cls), k_dp_pass<W, N, REV, true, STAGE, true>, be.dp_grid(pass, cls), 64,     \
                      ipx_dp_lds_bytes(W, N, REV, maxcols, true, routing), b, p, cls, cls, maxcols,          \
                      pass | (ipx_dp_mc_in_lds(W, REV, maxcols, true, routing) ? IPX_PASS_MC_LDS : 0), (uint64_t)0, (uint64_t)0);  \
        else                                                                                                 \
            be.launch(IPX_KEY(kclass, cls), k_dp_pass<W, N, REV, true, NP_STAGE, false>, be.dp_grid(pass, cls), 64, \
                      ipx_dp_lds_bytes(W, N, REV, maxcols, false, routing), b, p, cls, cls, maxcols, pass,   \
                      (uint64_t)0, (uint64_t)0);                                                             \
        break;
    be.note_dp(IPX_KEY(kclass, cls), pass, cls, (W == 16 && STAGE == IPX_STAGE_LOW && !REV && f16 && S <= 16 && (routing & IPX_ROUTE_INTERNAL_VL2)) ? 16 : 128 / W);
    switch (S) {
        IPX_DP_CASE(0) IPX_DP_CASE(1) IPX_DP_CASE(2) IPX_DP_CASE(3) IPX_DP_CASE(4) IPX_DP_CASE(5) IPX_DP_CASE(6)
        IPX_DP_CASE(7) IPX_DP_CASE(8) IPX_DP_CASE(9) IPX_DP_CASE(10) IPX_DP_CASE(11) IPX_DP_CASE(12) IPX_DP_CASE(13)
        IPX_DP_CASE(14) IPX_DP_CASE(15) IPX_DP_CASE(16) IPX_DP_CASE(17) IPX_DP_CASE(18) IPX_DP_CASE(19) IPX_DP_CASE(20)
        IPX_DP_CASE(21) IPX_DP_CASE(22) IPX_DP_CASE(23) IPX_DP_CASE(24) IPX_DP_CASE(25) IPX_DP_CASE(26) IPX_DP_CASE(27)
        IPX_DP_CASE(28) IPX_DP_CASE(29) IPX_DP_CASE(30) IPX_DP_CASE(31) IPX_DP_CASE(32)
    default: break;
    }
#undef IPX_DP_CASE
}

// Forward passes launch exactly the classes that can occur among their reads (known on the host).
// Reverse passes align a read PREFIX whose length is only known on the device: almost always the
// prefix has the read's own class or the one below, so those get their exact-segLen launch and ONE
// branch-guarded launch sweeps up every other class (it skips the tiles the exact launches own).
// halves: bit 0 = the fast-gap classes, bit 1 = the slow-gap classes (gap_open <= gap_ext) of `has` are served here
// fast_from: the fast-gap classes below it are served elsewhere (the wavefront launches of ipx_launch_skew_set)
template <class BE, int W, bool REV, int STAGE>
static void ipx_launch_dp(BE &be, const IpxBatch &b, const IpxPlan &p, const IpxWorkspace &ws, const uint8_t *has, int maxcols, int kclass, int pass, int routing,
                          int halves = 3, int fast_from = 0, int na = 0)
{
    uint64_t exact[2] = {0, 0};                                   // classes with their own launch: [0] fast, [1] slow gaps
    bool rest = false, lng = false;                               // anything the exact launches do not cover?  any read of 64 segments or more?
    int lng_halves = 0;                                           // ... bit 0: among the fast-gap classes, bit 1: among the slow-gap ones
    int need = 0;                                                 // ... and its largest segLen
    for (int half = 0; half < 2; ++half) {
        if (!((halves >> half) & 1)) { exact[half] = ~0ull; continue; }       // (not ours: the sweep below skips these tiles too)
        const uint8_t *hs = has + half * IPX_SLOW_BASE;
        const int from = half == 0 ? fast_from : 0;
        if (from > 0) exact[half] |= (from >= 64 ? ~0ull : ((1ull << from) - 1ull));   // (served elsewhere: the sweep skips them)
        int top = -1;
        for (int c = 0; c <= IPX_MAX_SEG; ++c) if (hs[c]) top = c;
        if (top >= IPX_MAX_SEG) { lng = true; lng_halves |= 1 << half; }
        uint32_t tier = 0;                                        // classes that share the tier launch (k_dp_pass_tier) instead of having their own
        if constexpr (W == 16 && STAGE == IPX_STAGE_EXACT) {
            if (half == 0 && ipx_perm_profile_ok(b.mat, routing) && !(routing & IPX_ROUTE_NO_TIERS)) {
                int n = 0;
                // (reverse pass: EVERY class up to the longest read's -- the class of the aligned prefix is decided on the device, the bodies
                //  are all there, and no sweep launch is left for the classes below)
                for (int c = from > IPX_PASS_TIER_LO ? from : IPX_PASS_TIER_LO; c <= top && c <= IPX_PASS_TIER_HI; ++c)
                    if (REV || hs[c] != 0) { tier |= 1u << c; ++n; }
                if (n < 2) tier = 0;
            }
            if (tier) {
                const int key = IPX_KEY(kclass, IPX_SUB_TIER + 8);
                be.note_dp_set(key, pass, 0, tier, 8);
                be.launch(key, k_dp_pass_tier<16, IPX_PASS_TIER_LO, IPX_PASS_TIER_HI, REV, IPX_STAGE_EXACT>, be.dp_grid_set(pass, 0, tier), 64,
                          ipx_dp_lds_bytes(W, IPX_PASS_TIER_HI, REV, maxcols, true, routing), b, p, tier, maxcols,
                          pass | (ipx_dp_mc_in_lds(W, REV, maxcols, true, routing) ? IPX_PASS_MC_LDS : 0));
            }
        }
        for (int c = from; c <= top && c <= IPX_MAX_EXACT; ++c) {
            const bool own = REV ? (hs[c] || (c + 1 <= top && hs[c + 1])) : hs[c] != 0;
            if (c < 32 && ((tier >> c) & 1u)) { exact[half] |= 1ull << c; continue; }
            if (own) { exact[half] |= 1ull << c; ipx_launch_dp_class<BE, W, REV, STAGE>(be, b, p, c + half * IPX_SLOW_BASE, maxcols, kclass, pass, routing); }
        }
        for (int c = from; c <= top && c < IPX_MAX_SEG; ++c)
            if ((REV || hs[c]) && !((exact[half] >> c) & 1ull)) { rest = true; need = c > need ? c : need; }
    }
    if (lng && ws.long_state) {
        // reads of 64 segments or more (over 512 bp in the 16-bit passes, over 1 024 bp in the 8-bit passes): the long-read kernel
        const int tile = na > 0 ? na : 128 / W;
        int mine = lng_halves;                                    // bit 0: the fast-gap long class, bit 1: the slow-gap one
        if constexpr (W == 8 && STAGE == IPX_STAGE_EXACT) {
            // r04: the 16-bit passes of the fast-gap long class as one wavefront per read (k_dp_wide), where its unsaturated 32-bit cells
            // are the reference's: selector profile (mat[.][N] = 0) and no score beyond 32 767
            if ((mine & 1) && !(routing & IPX_ROUTE_NO_WIDE) && ipx_perm_profile_ok(b.mat, routing) &&
                (int64_t)IPX_LONG_MAX_READ * b.max_match <= IPX_WIDE_MAX_SCORE && maxcols <= 32768) {
                int64_t g = (int64_t)be.dp_grid(pass, IPX_MAX_SEG) * tile;
                if (g > be.dp_grid()) g = be.dp_grid();
                if (g < 1) g = 1;
                const int lds = ipx_wide_lds_bytes(maxcols);
                be.launch(IPX_KEY(kclass, IPX_SUB_WIDE), k_dp_wide<16, REV>, (int)g, 64, lds, b, p, maxcols, pass);
                be.launch(IPX_KEY(kclass, IPX_SUB_WIDE), k_dp_wide<32, REV>, (int)g, 64, lds, b, p, maxcols, pass);
                be.launch(IPX_KEY(kclass, IPX_SUB_WIDE), k_dp_wide<48, REV>, (int)g, 64, lds, b, p, maxcols, pass);
                be.launch(IPX_KEY(kclass, IPX_SUB_WIDE), k_dp_wide<64, REV>, (int)g, 64, lds, b, p, maxcols, pass);
                mine &= ~1;
            }
        }
        int grid = be.dp_grid();
        if (grid > ws.long_blocks) grid = ws.long_blocks;
        if (mine) be.launch(IPX_KEY(kclass, IPX_SUB_LONG), k_dp_long<W, REV>, grid, 64, 64, b, p, tile, maxcols, pass, ws.long_state, ws.long_stride, mine);
    }
    if (!rest) return;
    // the sweep kernel keeps segLen registers for its largest class: size it for the largest class it
    // really has to serve (reverse passes: usually only short prefixes are left over), because the 64-segment
    // version needs a whole SIMD's register file per wave and queues behind everything else on a busy GPU
    constexpr int SW = STAGE == IPX_STAGE_HIGH ? IPX_STAGE_EXACT : STAGE;   // (8-bit classes never exceed 32 segments: no sweep in the bracket stages)
    const int last = IPX_NUM_CLASSES - 1;
    if constexpr (REV) {
        if (need <= 16) {
            be.launch(IPX_KEY(kclass, IPX_SUB_GENERIC), k_dp_pass<W, 16, REV, false, SW>, be.sweep_grid(pass, exact[0], exact[1]), 64,
                      ipx_dp_lds_bytes(W, 16, REV, maxcols, false, routing), b, p, 0, last, maxcols, pass, exact[0], exact[1]);
            return;
        }
        if (need <= 32) {
            be.launch(IPX_KEY(kclass, IPX_SUB_GENERIC), k_dp_pass<W, 32, REV, false, SW>, be.sweep_grid(pass, exact[0], exact[1]), 64,
                      ipx_dp_lds_bytes(W, 32, REV, maxcols, false, routing), b, p, 0, last, maxcols, pass, exact[0], exact[1]);
            return;
        }
    }
    be.launch(IPX_KEY(kclass, IPX_SUB_GENERIC), k_dp_pass<W, IPX_MAX_SEG, REV, false, SW>, be.sweep_grid(pass, exact[0], exact[1]), 64,
              ipx_dp_lds_bytes(W, IPX_MAX_SEG, REV, maxcols, false, routing), b, p, 0, last, maxcols, pass, exact[0], exact[1]);
}

// Wavefront launches of a pass whose (fast-gap) classes were planned by ipx_plan_classes.  BH = 0: 16-bit passes, class c = c segments.
// BH = 1 / 2: the plain recurrence in the 8-bit dialect, class c (8-bit segLen) = 2c segments of the 8-lane layout.  A launch also serves
// the shorter classes the planner listed under its class (k_dp_skew, ROW SHIFT).  The classes of `set` that fall into one occupancy
// (r03's shared tier launches, k_dp_skew_tier, are gone: see ipx_kernels.h.)
template <class BE, bool REV, int BH>
static void ipx_launch_skew_set(BE &be, const IpxBatch &b, const IpxPlan &p, const uint8_t *set, int maxcols, int kclass, int pass, int routing, int lat = 0, const IpxWorkspace *ws_rev = nullptr)
{
    int banded_from = IPX_MAX_EXACT + 1;                          // classes from here on were served by the banded reverse pass
    uint8_t todo_extra[IPX_MAX_EXACT + 1] = {0};                  // ... except these (the full kernel after all)
    if (lat && BH != 1) {
        // latency tier: LW = 32 or 64 lanes per read, the kernel with ceil(rows / LW) segments serves the class (ipx_plan_classes listed every
        // class of the pass under one)
        const int LW = lat;
        // (+ the tile's windows staged in LDS: (maxcols + 3) / 4 + 4 words per read, 2 * 64 / LW reads: k_dp_skew W >= 32)
        const int lds = ipx_dp_lds_bytes(LW, 0, REV, maxcols, true, routing) + 2 * (64 / LW) * ((maxcols + 3) / 4 + 4) * 4;
        const int pflag = pass | (ipx_dp_mc_in_lds(LW, REV, maxcols, true, routing) ? IPX_PASS_MC_LDS : 0);
        for (int c = 0; c <= (BH ? 16 : IPX_MAX_EXACT); ++c) {
            if (!set[c]) continue;
            const int rows = BH ? 16 * c : 8 * c;
            int S = (rows + LW - 1) / LW;
            if (S < 1) S = 1;
            be.note_dp(IPX_KEY(kclass, c), pass, c, 2 * (64 / LW));
            be.note_f16(BH ? 2 : 1, BH ? 2 * c : c);
#define IPX_LAT_CASE(N, LWC) case N: if constexpr (BH != 1) be.launch(IPX_KEY(kclass, c), k_dp_skew<N, REV, BH, LWC>, be.dp_grid(pass, c), 64, lds, b, p, c, maxcols, pflag); break;
            switch (S) { IPX_LAT_CASE(1, 32) IPX_LAT_CASE(2, 32) IPX_LAT_CASE(3, 32) IPX_LAT_CASE(4, 32) IPX_LAT_CASE(5, 32) IPX_LAT_CASE(6, 32) IPX_LAT_CASE(7, 32) IPX_LAT_CASE(8, 32) default: break; }
#undef IPX_LAT_CASE
        }
        return;
    }
    const int lds = ipx_dp_lds_bytes(8, 0, REV, maxcols, true, routing);
    const int pflag = pass | (ipx_dp_mc_in_lds(8, REV, maxcols, true, routing) ? IPX_PASS_MC_LDS : 0);
    if constexpr (REV && BH != 1) {
        // r04: the reverse pass as a band where a job's score budget allows it (k_dp_band_rev; the others, through a list and a plan of their own, in
        // the full wavefront kernel as before).  BH = 2, the plain recurrence in the 8-bit dialect: class c = 2 c segments.
        if (ws_rev && ws_rev->rev_listA && !(routing & IPX_ROUTE_NO_BAND_REV)) {
            be.zero_u32(ws_rev->rev_cnt, 2 * IPX_NUM_CLASSES);
            IpxPlan pb = p;
            pb.perm = ws_rev->rev_listB; pb.cls_off = ws_rev->rev_cls_off; pb.tile_off = ws_rev->rev_tile_off;
            const int c_lo = BH ? 4 : 8, c_hi = BH ? 16 : IPX_MAX_EXACT;
            for (int c = c_lo; c <= c_hi; ++c) {
                if (!set[c]) continue;
                const int tiles = be.dp_grid(pass, c);                          // (sized from the previous run's tile count of the class)
                // a band wave holds 128 jobs and lives eight blocks of ~60 steps: a class that does not fill the chip with such waves (config 4: 41 k
                // jobs per class and stream, 0.2 waves per SIMD) only gets a longer chain of launches out of it -- 57.9 -> 56.6 M aln/s; 2b: 85.3 -> 94.4
                // (a context's first run has no tile counts yet and sizes every launch for the whole batch: the batch's own size bounds the class then)
                if ((tiles < IPX_BAND_MIN_TILES || b.n_jobs / 16 < IPX_BAND_MIN_TILES) && !(routing & IPX_ROUTE_FORCE_BAND_REV)) { todo_extra[c] = 1; continue; }
                int gs = tiles / 32 + 1, gb = tiles / 8 + 1;                   // 512 jobs per block of the split, 128 per block of the band
                uint32_t *cnt = ws_rev->rev_cnt + 2 * c;
                be.launch(IPX_KEY(IPX_K_PLAN, 6), k_rev_split, gs, 64, 0, b, p, c, ipx_band_d(BH ? 2 * c : c), ws_rev->rev_listA, ws_rev->rev_listB, cnt);
                be.note_dp(IPX_KEY(kclass, c), pass, c, 16);
                be.note_f16(BH ? 2 : 1, BH ? 2 * c : c);
#define IPX_BAND_CASE(C) case C: if constexpr (BH == 0 && (C) >= 8) { \
                               be.launch(IPX_KEY(kclass, IPX_SUB_BAND), k_dp_band_rev<(C), 0>, gb, 64, ipx_band_lds_bytes(C), b, (const uint32_t *)ws_rev->rev_listA, (const uint32_t *)cnt, ws_rev->rev_cls_off, ws_rev->rev_tile_off, c); \
                               be.launch(IPX_KEY(kclass, c), k_dp_skew<(C), true, 0>, tiles, 64, lds, b, pb, c, maxcols, pflag); \
                           } else if constexpr (BH == 2 && (C) <= 16) { \
                               be.launch(IPX_KEY(kclass, IPX_SUB_BAND), k_dp_band_rev<2 * (C), 2>, gb, 64, ipx_band_lds_bytes(2 * (C)), b, (const uint32_t *)ws_rev->rev_listA, (const uint32_t *)cnt, ws_rev->rev_cls_off, ws_rev->rev_tile_off, c); \
                               be.launch(IPX_KEY(kclass, c), k_dp_skew<2 * (C), true, 2>, tiles, 64, lds, b, pb, c, maxcols, pflag); \
                           } break;
                switch (c) {
                    IPX_BAND_CASE(4) IPX_BAND_CASE(5) IPX_BAND_CASE(6) IPX_BAND_CASE(7)
                    IPX_BAND_CASE(8) IPX_BAND_CASE(9) IPX_BAND_CASE(10) IPX_BAND_CASE(11) IPX_BAND_CASE(12) IPX_BAND_CASE(13) IPX_BAND_CASE(14) IPX_BAND_CASE(15)
                    IPX_BAND_CASE(16) IPX_BAND_CASE(17) IPX_BAND_CASE(18) IPX_BAND_CASE(19) IPX_BAND_CASE(20) IPX_BAND_CASE(21) IPX_BAND_CASE(22) IPX_BAND_CASE(23)
                    IPX_BAND_CASE(24) IPX_BAND_CASE(25) IPX_BAND_CASE(26) IPX_BAND_CASE(27) IPX_BAND_CASE(28) IPX_BAND_CASE(29) IPX_BAND_CASE(30) IPX_BAND_CASE(31)
                    IPX_BAND_CASE(32)
                default: break;
                }
#undef IPX_BAND_CASE
            }
            banded_from = c_lo;
        }
    }
    uint8_t todo[IPX_MAX_EXACT + 1];
    for (int c = 0; c <= IPX_MAX_EXACT; ++c) todo[c] = (c <= (BH ? 16 : IPX_MAX_EXACT) && (c < banded_from || todo_extra[c])) ? set[c] : 0;
    for (int c = 0; c <= (BH ? 16 : IPX_MAX_EXACT); ++c) {
        if (!todo[c]) continue;
        be.note_dp(IPX_KEY(kclass, c), pass, c, 16);
        be.note_f16(BH ? 2 : 1, BH ? 2 * c : c);
#define IPX_SK_CASE(C)                                                                                                         \
    case C:                                                                                                                    \
        if constexpr (BH != 0) {                                                                                               \
            if constexpr ((C) >= 1 && (C) <= 16)                                                                               \
                be.launch(IPX_KEY(kclass, c), k_dp_skew<2 * (C), REV, BH>, be.dp_grid(pass, c), 64, lds, b, p, c, maxcols, pflag); \
        } else be.launch(IPX_KEY(kclass, c), k_dp_skew<(C), REV, 0>, be.dp_grid(pass, c), 64, lds, b, p, c, maxcols, pflag);     \
        break;
        switch (c) {
            IPX_SK_CASE(0) IPX_SK_CASE(1) IPX_SK_CASE(2) IPX_SK_CASE(3) IPX_SK_CASE(4) IPX_SK_CASE(5) IPX_SK_CASE(6) IPX_SK_CASE(7) IPX_SK_CASE(8)
            IPX_SK_CASE(9) IPX_SK_CASE(10) IPX_SK_CASE(11) IPX_SK_CASE(12) IPX_SK_CASE(13) IPX_SK_CASE(14) IPX_SK_CASE(15) IPX_SK_CASE(16)
            IPX_SK_CASE(17) IPX_SK_CASE(18) IPX_SK_CASE(19) IPX_SK_CASE(20) IPX_SK_CASE(21) IPX_SK_CASE(22) IPX_SK_CASE(23) IPX_SK_CASE(24)
            IPX_SK_CASE(25) IPX_SK_CASE(26) IPX_SK_CASE(27) IPX_SK_CASE(28) IPX_SK_CASE(29) IPX_SK_CASE(30) IPX_SK_CASE(31) IPX_SK_CASE(32)
        default: break;
        }
#undef IPX_SK_CASE
    }
}

// one scatter launch turns the counts of a pass (left by the kernels that sent jobs into it) into its job list;
// count_first: the jobs come straight from their initial state, count them here
template <class BE>
static void ipx_plan_pass(BE &be, const IpxBatch &b, const IpxPlan &p, int pass, int na, bool count_first = false)
{
    if (count_first) be.launch(IPX_KEY(IPX_K_PLAN, 0), k_plan_count, be.plan_grid(b.n_jobs), IPX_PLAN_BLOCK, 0, b, pass);
    be.launch(IPX_KEY(IPX_K_PLAN, 1), k_plan_scatter, be.plan_grid(b.n_jobs), IPX_PLAN_BLOCK, IPX_PLAN_LDS, b, p, pass, na);
}

// count + cursor tables of pass `pass` inside IpxWorkspace::plan_tables
static inline uint32_t *ipx_plan_count_of(uint32_t *tables, int pass) { return tables + (size_t)pass * 2 * IPX_NUM_CLASSES; }
#define IPX_PLAN_TABLE_WORDS (IPX_NUM_PASSES * 2 * IPX_NUM_CLASSES)

// tile size (alignments per wave) of the first 8-bit stage
static inline int ipx_first_na(const IpxBatch &b, const IpxDims &d, int routing) { return d.plain_first ? ipx_skew_na(d) : ipx_low2_ok(b, d, routing) ? 16 : 8; }

// The job lists of the passes a job starts in (every record PENDING): they depend on the read lengths, the penalties
// and the scoring parameters only, so they are built when a batch (or the parameters) changed, not in every run.
template <class BE>
static void ipx_build_static_plans(BE &be, const IpxBatch &b, const IpxWorkspace &ws, const IpxDims &d, int routing)
{
    IpxRunReset nothing;
    memset(&nothing, 0, sizeof nothing);
    be.launch(IPX_KEY(IPX_K_INIT, 0), k_init, be.flat_grid(b.n_jobs), 256, 0, b, nothing);
    be.zero_u32(ipx_plan_count_of(ws.plan_tables, 0), IPX_FIRST_DYNAMIC_PASS * 2 * IPX_NUM_CLASSES);
    if (b.score_size == 2) ipx_plan_pass(be, b, ws.plan[IPX_PASS_WORD_FIRST], IPX_PASS_WORD_FIRST, ipx_skew_na(d), true);
    if (b.score_size != 1) {
        ipx_plan_pass(be, b, ws.plan[IPX_PASS_BYTE_FIRST], IPX_PASS_BYTE_FIRST, ipx_first_na(b, d, routing), true);
        // jobs that start in the stepped pass: counted now, their count seeds that pass's (dynamic) row in every run
        be.zero_u32(ws.plan[IPX_PASS_BYTE_EXACT].count, IPX_NUM_CLASSES);
        be.launch(IPX_KEY(IPX_K_PLAN, 0), k_plan_count, be.plan_grid(b.n_jobs), IPX_PLAN_BLOCK, 0, b, (int)IPX_PASS_BYTE_EXACT);
        be.copy_u32(ws.exact_starters, ws.plan[IPX_PASS_BYTE_EXACT].count, IPX_NUM_CLASSES);
    }
}

template <class BE>
static void ipx_run_pipeline(BE &be, const IpxBatch &b, const IpxWorkspace &ws, const IpxDims &d, int routing, bool reset_status = false, bool speculate = false)
{
    static_assert(IPX_TB_NCOUNTERS == 12, "k_init resets twelve traceback counters");
    // SPECULATION (r04, latency tier): which passes a job takes is decided on the device, so the host launches every pass -- and in a small
    // call the planner launch and the kernels of a pass that finds nothing cost 8 + 4.5 us of a 300 us chain, five times over.  A pass the
    // PREVIOUS run of the context found empty (be.pass_predicted_empty: planner tile counts read back at ipx_sync) is not launched; k_tb_list,
    // which sees every record, notices a job left behind in such a pass (IPX_STATUS_RERUN) and the caller repeats the run with every pass.
    // Only with the traceback in the pipeline (it carries the guard) and only where the caller can repeat (`speculate`).
    const bool spec = speculate && d.lat && (7 & b.flag) != 0 && !(routing & IPX_ROUTE_NO_SPECULATE);
    auto skip = [&](int pass) { return spec && ((routing & IPX_ROUTE_TEST_SKIP_ALL) || be.pass_predicted_empty(pass)); };
    const int maxcols = d.max_ref_len + 4;
    uint8_t has8_all[IPX_NUM_CLASSES], has16_all[IPX_NUM_CLASSES];
    for (int c = 0; c < IPX_NUM_CLASSES; ++c) { has8_all[c] = d.has8_low[c] | d.has8_wf[c]; has16_all[c] = d.has16_low[c] | d.has16_wf[c]; }
    {   // records back to "nothing aligned yet", and (block 0) the small per-run tables: one launch where there were five commands
        IpxRunReset z;
        memset(&z, 0, sizeof z);
        z.cursor = b.cigar_cursor;
        z.status = reset_status ? b.status : nullptr;
        z.tb_n = ws.tb_list_n;
        z.dyn = ipx_plan_count_of(ws.plan_tables, IPX_FIRST_DYNAMIC_PASS);
        z.dyn_words = (IPX_NUM_PASSES - IPX_FIRST_DYNAMIC_PASS) * 2 * IPX_NUM_CLASSES;
        if (b.score_size != 1) { z.exact_dst = ws.plan[IPX_PASS_BYTE_EXACT].count; z.exact_src = ws.exact_starters; }
        be.launch(IPX_KEY(IPX_K_INIT, 0), k_init, be.flat_grid(b.n_jobs), 256, 0, b, z);
    }

    const bool low2 = ipx_low2_ok(b, d, routing);                 // 8-bit lower-bound launches: 16 reads per wave (k_dp_pass VL2)
    if (low2) routing |= IPX_ROUTE_INTERNAL_VL2;
    const bool wf = b.score_size == 2 && d.any_wf;
    // the fast-gap classes of a 16-bit pass: from the planned set as wavefronts, or class by class (ipx_launch_dp); slow-gap classes always the latter
    const int word_from = d.word_sets ? d.word_from : 0;         // fast-gap 16-bit classes below this: wavefront launches from the planned sets
    const int prove_chunk = b.n_jobs >= IPX_PROVE_CHUNK_MIN_JOBS ? IPX_PROVE_CHUNK : 1;   // (small batches: shortest chain of rounds per wave)
    // grids of the proof kernels (a block = a wave walking chunks of 64 * prove_chunk jobs): the overflow proof one chunk per wave; the plain
    // proofs -- nearly every read takes a band there -- TWO once there are many, so that the band queues carry over and the rounds run on full
    // waves (measured on 2a, 250 k jobs per stream: one chunk per wave 73.2 M aln/s, two 75.1, four 68.9: fewer waves than that leave the
    // launch to its own latency)
    const int64_t prove_nchunk = (b.n_jobs + 64 * prove_chunk - 1) / (64 * prove_chunk);
    const int prove_cap = be.flat_grid((int64_t)1 << 40);         // (the backend's cap on such grids)
    const int prove_grid1 = (int)(prove_nchunk < prove_cap ? (prove_nchunk > 0 ? prove_nchunk : 1) : prove_cap);
    const int prove_grid2 = prove_nchunk >= 512 ? (prove_grid1 + 1) / 2 : prove_grid1;
    if (wf) {
        // long reads: 16-bit pass first, then try to prove the 8-bit overflow from the end diagonal
        if (d.word_sets) ipx_launch_skew_set<BE, false, 0>(be, b, ws.plan[IPX_PASS_WORD_FIRST], d.set[IPX_PASS_WORD_FIRST], maxcols, IPX_K_WORD_FIRST, IPX_PASS_WORD_FIRST, routing, d.lat);
        ipx_launch_dp<BE, 8, false, IPX_STAGE_EXACT>(be, b, ws.plan[IPX_PASS_WORD_FIRST], ws, d.has16_wf, maxcols, IPX_K_WORD_FIRST, IPX_PASS_WORD_FIRST, routing, 3, word_from, 16);
        int cap = 64 * d.max_read_len;                            // one wave's reads
        if (cap > 60 * 1024) cap = 60 * 1024;
        if (d.lat && !(routing & IPX_ROUTE_NO_LAT_PROOF)) {       // latency tier: eight lanes per read (every read of the batch is within its reach: <= 256 bp)
            const int64_t pg = (b.n_jobs + 7) / 8;
            be.launch(IPX_KEY(IPX_K_PROVE, 1), k_prove_overflow_diag, (int)(pg < prove_cap ? pg : prove_cap), 64, ipx_proved_lds_bytes(), b);
        } else
        be.launch(IPX_KEY(IPX_K_PROVE, 0), k_prove_overflow, prove_grid1, 64, ipx_prove_lds_bytes(cap), b, cap, prove_chunk);
    }
    if (b.score_size != 1) {                                     // 8-bit forward pass (ssw.c:842-843)
        if (d.plain_first) {
            // plain recurrence first (k_dp_skew BH = 2), certified from below by k_prove_plain; what neither settles takes the
            // lower-bound stage (compared with the plain outputs) and, failing that, the stepped pass.  (Jobs with gap_open <= gap_ext
            // start in the stepped pass: next_pass_key.)
            if (d.any_low) {
                ipx_launch_skew_set<BE, false, 2>(be, b, ws.plan[IPX_PASS_BYTE_FIRST], d.set[IPX_PASS_BYTE_FIRST], maxcols, IPX_K_BYTE_PLAIN, IPX_PASS_BYTE_FIRST, routing, d.lat);
                be.launch(IPX_KEY(IPX_K_PROVE_PLAIN, 0), k_prove_plain<false>, prove_grid2, 64, ipx_prove_lds_bytes(4 * 64 * IPX_PROVE_EXT), b, prove_chunk);
                if (!b.exact_direct && !skip(IPX_PASS_BYTE_LOW2)) {
                    ipx_plan_pass(be, b, ws.plan[IPX_PASS_BYTE_LOW2], IPX_PASS_BYTE_LOW2, low2 ? 16 : 8);
                    ipx_launch_dp<BE, 16, false, IPX_STAGE_LOW>(be, b, ws.plan[IPX_PASS_BYTE_LOW2], ws, d.has8_low, maxcols, IPX_K_BYTE_LOW2, IPX_PASS_BYTE_LOW2, routing, 1, 0, low2 ? 16 : 8);
                }
            }
            if (wf && !skip(IPX_PASS_BYTE_CHECK)) {                // word-first reads whose overflow could not be proven (and that most likely do overflow)
                ipx_plan_pass(be, b, ws.plan[IPX_PASS_BYTE_CHECK], IPX_PASS_BYTE_CHECK, low2 ? 16 : 8);
                ipx_launch_dp<BE, 16, false, IPX_STAGE_LOW>(be, b, ws.plan[IPX_PASS_BYTE_CHECK], ws, d.has8_wf, maxcols, IPX_K_BYTE_CHECK, IPX_PASS_BYTE_CHECK, routing, 3, 0, low2 ? 16 : 8);
            }
        } else {
        if (d.any_low)
            ipx_launch_dp<BE, 16, false, IPX_STAGE_LOW>(be, b, ws.plan[IPX_PASS_BYTE_FIRST], ws, d.has8_low, maxcols, IPX_K_BYTE_LOW, IPX_PASS_BYTE_FIRST, routing, 3, 0, ipx_first_na(b, d, routing));
        if (wf && !skip(IPX_PASS_BYTE_CHECK)) {                    // word-first reads whose overflow could not be proven
            ipx_plan_pass(be, b, ws.plan[IPX_PASS_BYTE_CHECK], IPX_PASS_BYTE_CHECK, low2 ? 16 : 8);
            ipx_launch_dp<BE, 16, false, IPX_STAGE_LOW>(be, b, ws.plan[IPX_PASS_BYTE_CHECK], ws, d.has8_wf, maxcols, IPX_K_BYTE_CHECK, IPX_PASS_BYTE_CHECK, routing, 3, 0, low2 ? 16 : 8);
        }
        if (b.use_bracket && d.any_low) {                        // upper-bound stage: certifies the lower-bound outputs or not
            uint8_t hs[IPX_NUM_CLASSES];                          // (fast-gap classes only: a slow-gap read is stepped, never bracketed)
            memset(hs, 0, sizeof hs);
            memcpy(hs, d.has8_low, IPX_SLOW_BASE);
            // "every carry passed on" is the plain recurrence: where halves are exact for these reads it runs as a wavefront at 16
            // reads per wave (k_dp_skew BH, read lengths up to 256); otherwise column by column in the 8-bit layout
            ipx_plan_pass(be, b, ws.plan[IPX_PASS_BYTE_HIGH], IPX_PASS_BYTE_HIGH, d.high_sets ? 16 : 8);
            if (d.high_sets) ipx_launch_skew_set<BE, false, 1>(be, b, ws.plan[IPX_PASS_BYTE_HIGH], d.set[IPX_PASS_BYTE_HIGH], maxcols, IPX_K_BYTE_HIGH, IPX_PASS_BYTE_HIGH, routing);
            else ipx_launch_dp<BE, 16, false, IPX_STAGE_HIGH>(be, b, ws.plan[IPX_PASS_BYTE_HIGH], ws, hs, maxcols, IPX_K_BYTE_HIGH, IPX_PASS_BYTE_HIGH, routing, 3, 0, 8);
        }
        }
        // reads the bounds left open: the reference's stepped lazy-F
        if (!skip(IPX_PASS_BYTE_EXACT)) {
        ipx_plan_pass(be, b, ws.plan[IPX_PASS_BYTE_EXACT], IPX_PASS_BYTE_EXACT, 8);
        ipx_launch_dp<BE, 16, false, IPX_STAGE_EXACT>(be, b, ws.plan[IPX_PASS_BYTE_EXACT], ws, has8_all, maxcols, IPX_K_BYTE_EXACT, IPX_PASS_BYTE_EXACT, routing, 3, 0, 8);
        }
    }
    if (b.score_size != 0 && !(b.score_size == 2 && skip(IPX_PASS_WORD_FWD))) {   // 16-bit forward pass (ssw.c:844-847, 853-855)
        ipx_plan_pass(be, b, ws.plan[IPX_PASS_WORD_FWD], IPX_PASS_WORD_FWD, ipx_skew_na(d), b.score_size == 1);
        if (d.word_sets) ipx_launch_skew_set<BE, false, 0>(be, b, ws.plan[IPX_PASS_WORD_FWD], d.set[IPX_PASS_WORD_FWD], maxcols, IPX_K_WORD_FWD, IPX_PASS_WORD_FWD, routing, d.lat);
        ipx_launch_dp<BE, 8, false, IPX_STAGE_EXACT>(be, b, ws.plan[IPX_PASS_WORD_FWD], ws, b.score_size == 1 ? has16_all : d.has16_low, maxcols,
                                                     IPX_K_WORD_FWD, IPX_PASS_WORD_FWD, routing, 3, word_from, 16);
    }
    if (b.flag != 0) {                                           // begin position (ssw.c:872-886)
        if (b.score_size != 1) {
            if (d.plain_first && !skip(IPX_PASS_BYTE_REV_PLAIN)) {
                // reads whose forward result equals the plain recurrence's: plain reverse recurrence, certified by proof
                ipx_plan_pass(be, b, ws.plan[IPX_PASS_BYTE_REV_PLAIN], IPX_PASS_BYTE_REV_PLAIN, ipx_skew_na(d));
                ipx_launch_skew_set<BE, true, 2>(be, b, ws.plan[IPX_PASS_BYTE_REV_PLAIN], d.set[IPX_PASS_BYTE_REV_PLAIN], maxcols, IPX_K_BYTE_REV_PLAIN, IPX_PASS_BYTE_REV_PLAIN, routing, d.lat, &ws);
                be.launch(IPX_KEY(IPX_K_PROVE_PLAIN, 1), k_prove_plain<true>, prove_grid2, 64, ipx_prove_lds_bytes(4 * 64 * IPX_PROVE_EXT), b, prove_chunk);
            }
            if (!skip(IPX_PASS_BYTE_REV)) {
            ipx_plan_pass(be, b, ws.plan[IPX_PASS_BYTE_REV], IPX_PASS_BYTE_REV, 8);
            ipx_launch_dp<BE, 16, true, IPX_STAGE_EXACT>(be, b, ws.plan[IPX_PASS_BYTE_REV], ws, has8_all, maxcols, IPX_K_BYTE_REV, IPX_PASS_BYTE_REV, routing, 3, 0, 8);
            }
        }
        if (b.score_size != 0 && !skip(IPX_PASS_WORD_REV)) {
            ipx_plan_pass(be, b, ws.plan[IPX_PASS_WORD_REV], IPX_PASS_WORD_REV, ipx_skew_na(d));
            if (d.word_sets) ipx_launch_skew_set<BE, true, 0>(be, b, ws.plan[IPX_PASS_WORD_REV], d.set[IPX_PASS_WORD_REV], maxcols, IPX_K_WORD_REV, IPX_PASS_WORD_REV, routing, d.lat, &ws);
            ipx_launch_dp<BE, 8, true, IPX_STAGE_EXACT>(be, b, ws.plan[IPX_PASS_WORD_REV], ws, has16_all, maxcols, IPX_K_WORD_REV, IPX_PASS_WORD_REV, routing, 3, word_from, 16);
        }
        if ((7 & b.flag) != 0) {                                 // CIGAR (ssw.c:894-916)
            // a SMALL batch has more SIMDs than jobs: one wave per job (k_tb_coop: a DP row spread over the lanes) then finishes a typical job in
            // a fifth of the time one lane needs for it (r03, 1000 jobs: 0.76 -> 0.42 ms of traceback), and the lane-per-job launch is skipped
            // (with the anti-diagonal tiers: 2 = tiny batch, one job per wave; 1 = small batch, every job at 32 lanes at least; 3 = at 16 lanes at
            //  least -- six one-wave blocks of these kernels are resident per CU, 1 536 in all: more waves than that take a second round)
            const int tb_all_general = (routing & IPX_ROUTE_TB_NO_WAVE_PER_JOB) ? 0 : !b.tb_diag ? (b.n_jobs <= 2048 ? 1 : 0)
                                       : b.n_jobs <= IPX_TB_TINY_DIAG ? 2 : b.n_jobs <= 3 * IPX_TB_TINY_DIAG ? 1 : b.n_jobs <= IPX_TB_SMALL_DIAG ? 3 : 0;
            be.launch(IPX_KEY(IPX_K_TB_LIST, 0), k_tb_list, be.plan_grid(b.n_jobs), IPX_PLAN_BLOCK, 64, b, ws.tb_list, ws.tb_list_n, ws.tb_esc, tb_all_general,
                      ((routing & IPX_ROUTE_TB_NO_UNGAPPED) || tb_all_general) ? 0 : 1);   // (a small batch has a wave for every job: the check would only add its own latency)
            // rows of direction words that fit in LDS next to the CIGAR buffer (longer jobs take the general kernel)
            const int want = d.max_read_len > 0 ? d.max_read_len : 1;
            const int rowcap = want < IPX_TBF_ROWCAP ? want : IPX_TBF_ROWCAP;
#define IPX_TBF_LAUNCH(BW)                                                                                        \
    be.launch(IPX_KEY(IPX_K_TRACEBACK, 1 + BW), k_tb_fast<BW>,                                                    \
              BW <= 3 ? ws.tbf_waves : (ws.tbf_waves + 7) / 8, 64, ipx_tbf_lds_bytes(), b,   /* wide first bands are rare */ \
              (const uint32_t *)(ws.tb_list + (int64_t)(BW - 1) * b.n_jobs), (const uint32_t *)(ws.tb_list_n + (BW - 1)),  \
              rowcap, ws.tbf_scratch, ws.tb_esc, ws.tb_esc_n);
            // small batch: all widths side by side in one launch (latency); large batch: one launch per width
            // (each width has its own register footprint and occupancy)
            const int64_t fuse_max = (routing & IPX_ROUTE_TB_NO_FUSE) ? 0 : 20000;
            const int per_want = (int)((b.n_jobs + 63) / 64) + 1, per_have = ws.tbf_waves / 7;
            if (tb_all_general) {
                // (nothing for the lane-per-job kernels)
            } else if (b.n_jobs <= fuse_max && per_have >= 1) {
                const int per = per_want < per_have ? per_want : per_have;
                be.launch(IPX_KEY(IPX_K_TRACEBACK, 9), k_tb_fast_all, 7 * per, 64, ipx_tbf_lds_bytes(), b, (const uint32_t *)ws.tb_list,
                          (const uint32_t *)ws.tb_list_n, rowcap, ws.tbf_scratch, ws.tb_esc, ws.tb_esc_n, per, 0);
            } else if (!(routing & (IPX_ROUTE_TB_NO_FUSE | IPX_ROUTE_TB_PER_WIDTH)) && b.tb_diag) {
                // r04, big batches: one launch for all seven widths, its blocks shared out by the lists' lengths on the device
                // (the grid: a block per chunk where the scratch allows -- 512 to 2 048 blocks walking several chunks each were no faster)
                be.launch(IPX_KEY(IPX_K_TRACEBACK, 9), k_tb_fast_all, ws.tbf_waves, 64, ipx_tbf_lds_bytes(), b, (const uint32_t *)ws.tb_list,
                          (const uint32_t *)ws.tb_list_n, rowcap, ws.tbf_scratch, ws.tb_esc, ws.tb_esc_n, 0, 0);
            } else {
                // widths 1..3 (nearly every job): their own launches.  (Direction words in LDS instead of the global scratch were
                // measured in r02 and dropped: 46 KB per one-wave block leaves 3 waves per CU, and the lane-per-job walk needs
                // many waves in flight more than it needs short loads -- 4.0 ms against 2.6 ms per million jobs.)
            IPX_TBF_LAUNCH(1) IPX_TBF_LAUNCH(2) IPX_TBF_LAUNCH(3)
                // widths 4..7 (rare): side by side in one launch
                const int per = (ws.tbf_waves + 7) / 8 > 0 ? (ws.tbf_waves + 7) / 8 : 1;
                if (!(routing & IPX_ROUTE_TB_NO_FUSE) && 4 * per <= ws.tbf_waves)   // (direction-word regions are indexed by block id)
                    be.launch(IPX_KEY(IPX_K_TRACEBACK, 10), k_tb_fast_all, 4 * per, 64, ipx_tbf_lds_bytes(), b, (const uint32_t *)ws.tb_list,
                              (const uint32_t *)ws.tb_list_n, rowcap, ws.tbf_scratch, ws.tb_esc, ws.tb_esc_n, per, 3);
                else {
            IPX_TBF_LAUNCH(4) IPX_TBF_LAUNCH(5) IPX_TBF_LAUNCH(6) IPX_TBF_LAUNCH(7)
                }
            }
#undef IPX_TBF_LAUNCH
            // anti-diagonal tiers (r04): first bands 8..63, bands the lane-per-job kernels saw double past 7, and every job of a small batch; a tier
            // hands a band it cannot hold to the next one
            if (b.tb_diag) {
                uint32_t *l16 = ws.tb_list + (int64_t)(IPX_TB_CLS_DIAG - 1) * b.n_jobs, *l32 = l16 + b.n_jobs, *l64 = l32 + b.n_jobs;
                uint32_t *c16 = ws.tb_list_n + IPX_TB_CLS_DIAG;
                be.launch(IPX_KEY(IPX_K_TRACEBACK, 16), k_tb_diag<16>, ws.tbd_waves, 64, ipx_tbd_lds_bytes(16), b, (const uint32_t *)l16, (const uint32_t *)c16, l32, c16 + 1, ws.tb_esc, ws.tb_esc_n);
                be.launch(IPX_KEY(IPX_K_TRACEBACK, 17), k_tb_diag<32>, ws.tbd_waves, 64, ipx_tbd_lds_bytes(32), b, (const uint32_t *)l32, (const uint32_t *)(c16 + 1), l64, c16 + 2, ws.tb_esc, ws.tb_esc_n);
                be.launch(IPX_KEY(IPX_K_TRACEBACK, 18), k_tb_diag<64>, ws.tbd_waves, 64, ipx_tbd_lds_bytes(64), b, (const uint32_t *)l64, (const uint32_t *)(c16 + 2), ws.tb_esc, ws.tb_esc_n, ws.tb_esc, ws.tb_esc_n);
            }
            // everything else: one wavefront per job
            be.launch(IPX_KEY(IPX_K_TRACEBACK, 1), k_tb_coop, ws.tb1_waves, 64, ipx_tbc_lds_bytes(ws.tb1.arrcap_lds), b,
                      (const uint32_t *)ws.tb_esc, (const uint32_t *)ws.tb_esc_n, ws.tb1.dir, (int64_t)ws.tb1.dircap,
                      ws.tb1.arrcap, ws.tb1.cig, ws.tb1.cigcap, ws.tb1.band, ws.tb1.arrcap_lds);
        }
    }
}


// reads at least this long are likely to overflow the 8-bit pass (their best possible score is >= 1.1x the overflow
// threshold: a 93 bp read at match 3 still overflows with five mismatches or an indel and two): they take the 16-bit pass
// first.  Any value is correct; it only moves work between passes.  (1.4x until the 16-bit passes became the cheapest
// kernels per read: a 100 bp read took the 8-bit lower-bound stage only to overflow there and be rescored.)
static inline int ipx_word_first_len(const int8_t *mat, int bias)
{
    int mx = 0;
    for (int k = 0; k < 25; ++k) if (mat[k] > mx) mx = mat[k];
    if (mx <= 0) return 0;
    const int cap = 255 - bias;
    return (cap * 11 / 10 + mx - 1) / mx;
}

// shortest read that could overflow the 8-bit pass: len * max(mat) >= 255 - bias
static inline int ipx_byte_safe_len(const int8_t *mat, int bias)
{
    int mx = 0;
    for (int k = 0; k < 25; ++k) if (mat[k] > mx) mx = mat[k];
    if (mx <= 0) return 0x7FFFFFFF;
    return (255 - bias + mx - 1) / mx;
}
// reads shorter than this start in the stepped 8-bit pass (next_pass_key, IPX_MODE_PENDING)
static inline int ipx_exact_start_len(int byte_safe_len, int bracket_min_len, bool use_bracket)
{
    return use_bracket ? (byte_safe_len < bracket_min_len ? byte_safe_len : bracket_min_len) : byte_safe_len;
}

// The bracket (lower + upper bound stage) costs two closed-form passes; the stepped pass costs one plus the stepping,
// which grows with the number of columns whose carries sit in signed-compare territory (>= 128).  A read that can only
// just get there has few such columns and is cheaper stepped.  r02, per tile column of 8 reads: config 2a (150 bp at
// match 1) 360 instructions stepped vs 194 (lower bound) + ~110 (upper bound as a wavefront at 16 reads per wave,
// k_dp_skew BH); 75 bp at match 3: 745 vs 140 + ~60.  With the upper bound that cheap the bracket wins from a best possible
// score of ~130 (it was ~160 while both stages ran column by column at 8 reads per wave).
// Any value is correct; it only moves work between passes.
static inline int ipx_bracket_min_len(const int8_t *mat)
{
    int mx = 0;
    for (int k = 0; k < 25; ++k) if (mat[k] > mx) mx = mat[k];
    if (mx <= 0) return 0x7FFFFFFF;
    return (130 + mx - 1) / mx;
}

// longest read that may take the half-precision form of the 16-bit passes (k_dp_pass F16): every matrix entry must be a half
// with a zero low byte (v_perm_b32 delivers only the high byte) and no score may exceed 2047, the last integer before halves
// step by 2.  0 = the form does not apply.
static inline int ipx_f16_max_len(const int8_t *mat)
{
    int mx = 0;
    for (int k = 0; k < 25; ++k) {
        if (ipx_f16_from_int(mat[k]) & 0xFFu) return 0;
        if (mat[k] > mx) mx = mat[k];
    }
    return mx > 0 ? 2047 / mx : 0;
}

// which classes can occur in which pass, for the current scoring parameters (host-known facts only)
static inline void ipx_dims_finish(IpxDims &d, int word_first_len, int score_size, int exact_start_len)
{
    memset(d.has8_low, 0, sizeof d.has8_low); memset(d.has8_wf, 0, sizeof d.has8_wf);
    memset(d.has16_low, 0, sizeof d.has16_low); memset(d.has16_wf, 0, sizeof d.has16_wf);
    d.any_wf = d.any_low = 0;
    for (int slow = 0; slow < 2; ++slow)
        for (int len = 0; len <= IPX_MAX_READ_LEN; ++len) {
            if (!d.lenhist[slow][len]) continue;
            int c8 = (len + 15) / 16, c16 = (len + 7) / 8;
            if (c8 > IPX_MAX_SEG) c8 = IPX_MAX_SEG;
            if (c16 > IPX_MAX_SEG) c16 = IPX_MAX_SEG;
            const bool wfirst = score_size == 2 && word_first_len > 0 && len >= word_first_len;
            (wfirst ? d.has8_wf : d.has8_low)[c8 + slow * IPX_SLOW_BASE] = 1;
            (wfirst ? d.has16_wf : d.has16_low)[c16 + slow * IPX_SLOW_BASE] = 1;
            if (wfirst) d.any_wf = 1; else if (len >= exact_start_len) d.any_low = 1;   // (shorter ones start in the stepped pass)
        }
}

// Where the wavefront kernels serve a pass: which class every class is listed under and which classes get a launch.
// k_dp_skew computes the plain recurrence, for which the striping is only a matter of row numbers: a kernel of S segments serves
// any read whose padded row count fits into its 8 x S rows (ROW SHIFT at k_dp_skew).  So
//   * a class with few reads does not get a launch of its own (~50 waves on a 1 024-SIMD chip cost what 20 000 cost -- r02, config 4:
//     2 % of the jobs took 20 % of the DP stream time) but rides in the next populated class's, at that class's cost per read;
//   * reverse passes, whose class (of the aligned PREFIX) is decided on the device, launch the classes c and c-1 of every forward
//     class c and list every other prefix class under the next of those: no branch-guarded sweep launch is left.
// n16wf / n16low / n8low: reads per class among the fast-gap reads (16-bit classes of the word-first reads and of the others; 8-bit
// classes of the others).  Any map that sends a class to one at least as long is correct; this only moves work between launches.
#ifndef IPX_MERGE_BELOW          // (the emulator build sets small values so that its small batches keep several classes)
#define IPX_MERGE_BELOW 8192     // reads: a class with fewer is merged into the next kept class ...
#define IPX_MERGE_TINY 1024      // ... unless that more than doubles its cost and it has at least this many reads
#endif
static inline void ipx_merge_classes(const uint32_t *n, int top, bool merge, uint8_t *map, uint8_t *set)
{
    int target = -1;
    for (int c = top; c >= 0; --c) {
        if (!n[c]) continue;
        const bool small = merge && n[c] < IPX_MERGE_BELOW && (2 * c >= target || n[c] < IPX_MERGE_TINY);
        if (target >= 0 && (small || c == 0)) map[c] = (uint8_t)target;
        else { target = c; set[c] = 1; map[c] = (uint8_t)c; }
    }
}
// reverse pass: launches for c and c-1 of every kept forward class, every other prefix class listed under the next launch
// (r04, second half: `below` = false lists class c-1 under c as well -- a prefix eight rows shorter costs the longer kernel 1/c more work and the
//  stream one launch, with its ramp and its tail, less per forward class)
static inline void ipx_reverse_classes(const uint8_t *kept_a, const uint8_t *kept_b, int top, uint8_t *map, uint8_t *set, const uint8_t *extra, bool below = true)
{
    for (int c = 1; c <= top; ++c)
        if (kept_a[c] || (kept_b && kept_b[c]) || (extra && extra[c])) { set[c] = 1; if (below && c > 1 && !(extra && extra[c] && !kept_a[c] && !(kept_b && kept_b[c]))) set[c - 1] = 1; }
    int target = -1;
    for (int c = top; c >= 0; --c) {
        if (set[c]) target = c;
        if (target >= 0) map[c] = (uint8_t)target;
    }
}
static inline void ipx_plan_classes(IpxDims &d, const IpxBatch &b, int routing)
{
    for (int ps = 0; ps < IPX_NUM_PASSES; ++ps)
        for (int c = 0; c < IPX_NUM_CLASSES; ++c) d.cls_map[ps][c] = (uint8_t)c;
    memset(d.set, 0, sizeof d.set);
    d.word_sets = d.plain_first = d.high_sets = d.lat = 0;
    d.word_from = 0; d.plain_max_len = 0;
    const bool skew_ok = ipx_perm_profile_ok(b.mat, routing) && !(routing & (IPX_ROUTE_NO_F16 | IPX_ROUTE_NO_SKEW)) && b.f16_max_len > 0;
    if (!skew_ok) return;
    const bool merge = !(routing & IPX_ROUTE_NO_CLASS_MERGE);
    // the wavefront kernels reach 32 segments (16 in the 8-bit dialect) and need every score of the class exact in a half; longer
    // reads of the same batch keep the class-by-class launches (k_dp_pass, k_dp_long) and their own classes
    const int fmax16 = b.f16_max_len / 8 < IPX_MAX_EXACT ? b.f16_max_len / 8 : IPX_MAX_EXACT;
    const int fmax8 = b.f16_max_len / 16 < 16 ? b.f16_max_len / 16 : 16;
    uint32_t n16wf[IPX_MAX_SEG + 1], n16low[IPX_MAX_SEG + 1], n8low[IPX_MAX_SEG + 1];
    memset(n16wf, 0, sizeof n16wf); memset(n16low, 0, sizeof n16low); memset(n8low, 0, sizeof n8low);
    int top16 = -1, top8 = -1, all16 = -1, all8 = -1;
    for (int len = 0; len <= IPX_MAX_READ_LEN; ++len) {
        const uint32_t n = d.lenhist[0][len];
        if (!n) continue;
        const int c8 = (len + 15) / 16, c16 = (len + 7) / 8;
        const bool wfirst = b.score_size == 2 && b.word_first_len > 0 && len >= b.word_first_len;
        if (c16 > all16) all16 = c16;
        if (c16 <= fmax16) { (wfirst ? n16wf : n16low)[c16] += n; if (c16 > top16) top16 = c16; }
        if (!wfirst && c8 > all8) all8 = c8;
        if (!wfirst && c8 <= fmax8) { n8low[c8] += n; if (c8 > top8) top8 = c8; }
    }
    // latency tier (r04): a small batch, no job with gap_open <= gap_ext (those need the stepped kernels, whose tiles hold 8 or 16 reads:
    // a pass has ONE tile size) and every 16-bit class within the wavefront kernels' reach.  Every class of a pass is then listed under the
    // pass's longest one: ONE launch per pass -- on a chip with more SIMDs than the batch has tiles a launch lasts as long as its longest
    // tile whatever its size, and two launches last twice that.
    bool lat = merge && !(routing & IPX_ROUTE_NO_LAT) && (b.n_jobs <= IPX_LAT_MAX_JOBS || (routing & IPX_ROUTE_FORCE_LAT)) && all16 <= fmax16 && (all8 < 0 || all8 <= 16);
    for (int len = 0; lat && len <= IPX_MAX_READ_LEN; ++len) if (d.lenhist[1][len]) lat = false;
    auto single_class = [](const uint32_t *n1, const uint32_t *n2, int top, uint8_t *map, uint8_t *set) {
        int t = -1;
        for (int c = top; c >= 0 && t < 0; --c) if (n1[c] || (n2 && n2[c])) t = c;
        if (t < 0) return;
        if (t < 1) t = 1;
        set[t] = 1;
        for (int c = 0; c <= t; ++c) map[c] = (uint8_t)t;
    };
    if (lat && b.score_size != 0 && top16 >= 1) {
        d.lat = 32;
        d.word_sets = 1;
        d.word_from = fmax16 + 1;
        single_class(n16wf, nullptr, top16, d.cls_map[IPX_PASS_WORD_FIRST], d.set[IPX_PASS_WORD_FIRST]);
        single_class(n16low, nullptr, top16, d.cls_map[IPX_PASS_WORD_FWD], d.set[IPX_PASS_WORD_FWD]);
        single_class(n16wf, n16low, top16, d.cls_map[IPX_PASS_WORD_REV], d.set[IPX_PASS_WORD_REV]);      // (a prefix is never longer than its read)
    } else if (merge && b.score_size != 0 && top16 >= 1) {
        d.word_sets = 1;
        d.word_from = fmax16 + 1;
        ipx_merge_classes(n16wf, top16, true, d.cls_map[IPX_PASS_WORD_FIRST], d.set[IPX_PASS_WORD_FIRST]);
        ipx_merge_classes(n16low, top16, true, d.cls_map[IPX_PASS_WORD_FWD], d.set[IPX_PASS_WORD_FWD]);
        // every prefix class up to fmax16 must land in a launch: a longer read's alignment may end anywhere
        uint8_t reach[IPX_NUM_CLASSES];
        memset(reach, 0, sizeof reach);
        reach[all16 < fmax16 ? (all16 > 0 ? all16 : 1) : fmax16] = 1;
        ipx_reverse_classes(d.set[IPX_PASS_WORD_FIRST], d.set[IPX_PASS_WORD_FWD], fmax16, d.cls_map[IPX_PASS_WORD_REV], d.set[IPX_PASS_WORD_REV], reach, (routing & IPX_ROUTE_REV_BELOW) != 0);
    }
    // 8-bit passes as the plain recurrence
    const bool plain_ok = b.score_size != 1 && top8 >= 1;
    if (d.lat && plain_ok && b.use_bracket && !(routing & IPX_ROUTE_NO_PLAIN_FIRST)) {
        d.plain_first = 1;
        d.plain_max_len = 16 * fmax8;
        single_class(n8low, nullptr, top8, d.cls_map[IPX_PASS_BYTE_FIRST], d.set[IPX_PASS_BYTE_FIRST]);
        single_class(n8low, nullptr, top8, d.cls_map[IPX_PASS_BYTE_REV_PLAIN], d.set[IPX_PASS_BYTE_REV_PLAIN]);
    } else if (plain_ok && b.use_bracket && !(routing & IPX_ROUTE_NO_PLAIN_FIRST)) {
        d.plain_first = 1;
        d.plain_max_len = 16 * fmax8;
        ipx_merge_classes(n8low, top8, merge, d.cls_map[IPX_PASS_BYTE_FIRST], d.set[IPX_PASS_BYTE_FIRST]);
        if (merge) ipx_reverse_classes(d.set[IPX_PASS_BYTE_FIRST], nullptr, top8, d.cls_map[IPX_PASS_BYTE_REV_PLAIN], d.set[IPX_PASS_BYTE_REV_PLAIN], nullptr, (routing & IPX_ROUTE_REV_BELOW) != 0);
        else { for (int c = 1; c <= top8; ++c) d.set[IPX_PASS_BYTE_REV_PLAIN][c] = 1; d.cls_map[IPX_PASS_BYTE_REV_PLAIN][0] = 1; }
    } else if (plain_ok && b.use_bracket && top8 == all8) {
        // bracket flow: the upper-bound stage as a wavefront (r02) when every 8-bit class of the batch is within its reach, its rare
        // classes merged as well.  (Only reads that reach the bracket length get there; the counts of all 8-bit starters are an upper
        // bound, which is all the merge rule needs.)
        d.high_sets = 1;
        ipx_merge_classes(n8low, top8, merge, d.cls_map[IPX_PASS_BYTE_HIGH], d.set[IPX_PASS_BYTE_HIGH]);
    }
}

// scratch sizing shared by both back-ends -------------------------------------------------------
#define IPX_TBC_ARRCAP_LDS 1024     // band rows of up to this many entries stay in LDS (16 KB per one-wave block; with the staged letters and the direction bytes 27 KB: five blocks per CU)
struct IpxTbSizing { int arrcap, dircap, cigcap, arrcap_lds; };
static inline IpxTbSizing ipx_tb1_sizing(const IpxDims &d)
{
    IpxTbSizing s;
    const int len = d.max_read_len > d.max_ref_len ? d.max_read_len : d.max_ref_len;
    s.arrcap = 2 * (len > 0 ? len : 1) + 8;                      // band_width <= len (ssw.c:669)
    s.dircap = (2 * (len > 0 ? len : 1) + 1) * (d.max_read_len > 0 ? d.max_read_len : 1);
    s.cigcap = d.max_read_len + d.max_ref_len + 8;
    s.arrcap_lds = s.arrcap < IPX_TBC_ARRCAP_LDS ? s.arrcap : IPX_TBC_ARRCAP_LDS;
    return s;
}
// the one-wave-per-job kernel needs direction bytes and CIGAR runs for ONE job per block
static inline size_t ipx_tbc_bytes_per_block(const IpxTbSizing &s)
{
    return (((size_t)s.dircap + 15) & ~(size_t)15) + 4ull * (size_t)s.cigcap + 16 + (s.arrcap > s.arrcap_lds ? 16ull * (size_t)s.arrcap : 0ull);
}
