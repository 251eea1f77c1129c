// ipx_pipeline.h -- the launch sequence of one alignment batch (the control flow of ssw_align,
// ssw.c:842-916, over a whole job table).  Written against a tiny launcher so the HIP runtime
// (ipx_runtime.hip) and the test-only wave emulator (tests/emu) run the identical sequence.
//
//   init -> [plan + 8-bit forward] -> [plan + 16-bit forward for overflowed reads]
//        -> [plan + 8-bit reverse] -> [plan + 16-bit reverse] -> traceback list -> banded traceback
//
// No host synchronisation happens between the stages: which job takes which branch is decided by
// the planner kernels on the device.
#pragma once
#include <stdlib.h>
#include "ipx_kernels.h"

struct IpxWorkspace {
    IpxPlan plan;
    uint32_t *tb_list, *tb_list_n;      // jobs that get a CIGAR: 7 lists (first band 1..7) of n_jobs slots; 8 counters, the 8th = tb_esc_n
    uint32_t *tb_esc, *tb_esc_n;        // jobs the fast traceback hands to the general (one wave per job) kernel
    IpxTbScratch tb1;
    unsigned char *tbf_scratch;         // direction words of the fast traceback: ipx_tbf_scratch_bytes_per_block(rowcap) per block
    int tbf_waves, tb1_waves;
};

#define IPX_TBF_ROWCAP 512  // rows of direction words per fast-traceback block
#define IPX_MAX_EXACT 32     // segLen classes 0..32 have their own straight-line instantiation

struct IpxDims {
    int max_read_len;                  // longest read of the batch
    int max_ref_len;                   // longest window of the batch
    uint8_t has8[IPX_NUM_CLASSES];     // segLen classes present among the reads, 8-bit pass
    uint8_t has16[IPX_NUM_CLASSES];    // ... 16-bit pass
    uint8_t any_slow_gap;              // some job has gap_open <= gap_ext
};

static inline void ipx_dims_add_read(IpxDims &d, int len)
{
    if (len > d.max_read_len) d.max_read_len = len;
    int c8 = (len + 15) / 16, c16 = (len + 7) / 8;
    if (c8 > IPX_MAX_SEG) c8 = IPX_MAX_SEG;       // longer reads are refused by the planner (status bit)
    if (c16 > IPX_MAX_SEG) c16 = IPX_MAX_SEG;
    d.has8[c8] = 1;
    d.has16[c16] = 1;
}

// kernel classes for per-kernel timing (ipx_runtime.hip records HIP events around each launch)
enum {
    IPX_K_INIT = 0, IPX_K_PLAN, IPX_K_BYTE_FWD, IPX_K_WORD_FWD, IPX_K_BYTE_REV, IPX_K_WORD_REV,
    IPX_K_TB_LIST, IPX_K_TRACEBACK, IPX_K_PACK, IPX_K_BYTE_FWD_X, IPX_K_WORD_FIRST, IPX_K_PROVE, IPX_K_NUM
};

// DP grid cap, in blocks (= waves) per CU.  Far more than are resident (<= 24): with several streams
// sharing the GPU, short-lived blocks let the kernels of different streams interleave and even out the
// tail (measured on config 2b, 4 streams: 12 -> 41.4, 24 -> 42.5, 48 -> 43.3, 96 -> 43.6 M aln/s).
// Each block owns a column-maxima scratch region, so the grid is also capped by IPX_DP_SCRATCH_BUDGET.
#define IPX_DP_WAVES_PER_CU 64
#define IPX_DP_SCRATCH_BUDGET ((size_t)1 << 30)
static inline int ipx_dp_grid_mult()
{
    static const int m = getenv("IPX_DP_GRID_MULT") ? atoi(getenv("IPX_DP_GRID_MULT")) : IPX_DP_WAVES_PER_CU;   // tuning experiments
    return m > 0 ? m : IPX_DP_WAVES_PER_CU;
}
// column maxima of a forward selector-profile kernel stay in LDS while 16 waves per CU still fit (160 KB)
#define IPX_DP_MC_LDS_MAX 10112
static inline bool ipx_dp_mc_in_lds(int W, bool rev, int maxcols, bool perm)
{
    return perm && !rev && (64 / W) * maxcols * 4 <= IPX_DP_MC_LDS_MAX && !getenv("IPX_NO_MC_LDS");
}
static inline int ipx_dp_lds_bytes(int W, int SMAX, bool rev, int maxcols, bool perm = false)
{
    static const int extra = getenv("IPX_DEBUG_EXTRA_LDS") ? atoi(getenv("IPX_DEBUG_EXTRA_LDS")) : 0;   // occupancy experiments
    const int mc = ipx_dp_mc_in_lds(W, rev, maxcols, perm) ? (64 / W) * maxcols * 4 : 0;
    return (perm ? 64 : 640 * (SMAX > 0 ? SMAX : 1)) + 64 + mc + extra;
}

// the register-selector profile (k_dp_pass PERM) needs a read letter N to score 0 against every window letter
static inline bool ipx_perm_profile_ok(const int8_t *mat)
{
    const bool off = getenv("IPX_NO_PERM_PROFILE") != nullptr;   // (looked up per launch: the tests flip it)
    return !off && mat[4] == 0 && mat[9] == 0 && mat[14] == 0 && mat[19] == 0 && mat[24] == 0;
}

// timing key of a launch: kernel class * 128 + sub (DP kernels: sub = segLen, 65 = long-read kernel)
#define IPX_KEY(kclass, sub) ((kclass) * 128 + (sub))
#define IPX_NUM_KEYS (IPX_K_NUM * 128)
#define IPX_SUB_GENERIC 65

template <class BE, int W, bool REV, bool LOW>
static void ipx_launch_dp_class(BE &be, const IpxBatch &b, const IpxPlan &p, int S, int maxcols, int kclass, int pass)
{
#define IPX_DP_CASE(N)                                                                                       \
    case N:                                                                                                  \
        if (perm)                                                                                            \
            be.launch(IPX_KEY(kclass, N), k_dp_pass<W, N, REV, true, LOW, true>, be.dp_grid(pass, N), 64,           \
                      ipx_dp_lds_bytes(W, N, REV, maxcols, true), b, p, N, N, maxcols,                       \
                      pass | (ipx_dp_mc_in_lds(W, REV, maxcols, true) ? IPX_PASS_MC_LDS : 0), (uint64_t)0);  \
        else                                                                                                 \
            be.launch(IPX_KEY(kclass, N), k_dp_pass<W, N, REV, true, LOW, false>, be.dp_grid(pass, N), 64,          \
                      ipx_dp_lds_bytes(W, N, REV, maxcols), b, p, N, N, maxcols, pass, (uint64_t)0);         \
        break;
    // the selector-profile kernels of the 16-bit passes and of the 8-bit lower-bound stage have no stepped lazy-F
    // loop (k_dp_pass, STEP): they are for batches in which every job has gap_open > gap_ext
    const bool perm = ipx_perm_profile_ok(b.mat) && ((W == 16 && !LOW) || !b.any_slow_gap);
    be.note_dp(IPX_KEY(kclass, S), pass, S, 128 / W);
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

// Forward passes launch exactly the segLen classes that occur among the reads (known on the host).
// Reverse passes align a read PREFIX whose length is only known on the device: almost always the
// prefix has the read's own class or the one below, so those get their exact-segLen launch and ONE
// branch-guarded launch sweeps up every other class (it skips the tiles the exact launches own).
template <class BE, int W, bool REV, bool LOW>
static void ipx_launch_dp(BE &be, const IpxBatch &b, const IpxPlan &p, const uint8_t *has, int maxcols, int kclass, int pass)
{
    int top = -1;
    for (int c = 0; c < IPX_NUM_CLASSES; ++c) if (has[c]) top = c;
    if (top < 0) return;
    uint64_t exact = 0;                                           // classes with their own launch
    for (int c = 0; c <= top && c <= IPX_MAX_EXACT; ++c) {
        const bool own = REV ? (has[c] || (c + 1 <= top && has[c + 1])) : has[c] != 0;
        if (own) { exact |= 1ull << c; ipx_launch_dp_class<BE, W, REV, LOW>(be, b, p, c, maxcols, kclass, pass); }
    }
    bool rest = false;                                            // anything the exact launches do not cover?
    for (int c = 0; c <= top; ++c)
        if ((REV || has[c]) && !(c < 64 && ((exact >> c) & 1ull))) rest = true;
    if (!rest) return;
    // the sweep kernel keeps segLen registers for its largest class: size it for the largest class it
    // really has to serve (reverse passes: usually only short prefixes are left over), because the 64-segment
    // version needs a whole SIMD's register file per wave and queues behind everything else on a busy GPU
    int need = 0;
    for (int c = 0; c <= top; ++c)
        if ((REV || has[c]) && !(c < 64 && ((exact >> c) & 1ull))) need = c;
    if (REV && need <= 16)
        be.launch(IPX_KEY(kclass, IPX_SUB_GENERIC), k_dp_pass<W, 16, REV, false, LOW>, be.sweep_grid(pass, exact, top), 64,
                  ipx_dp_lds_bytes(W, 16, REV, maxcols), b, p, 0, top, maxcols, pass, exact);
    else if (REV && need <= 32)
        be.launch(IPX_KEY(kclass, IPX_SUB_GENERIC), k_dp_pass<W, 32, REV, false, LOW>, be.sweep_grid(pass, exact, top), 64,
                  ipx_dp_lds_bytes(W, 32, REV, maxcols), b, p, 0, top, maxcols, pass, exact);
    else
        be.launch(IPX_KEY(kclass, IPX_SUB_GENERIC), k_dp_pass<W, IPX_MAX_SEG, REV, false, LOW>, be.sweep_grid(pass, exact, top), 64,
                  ipx_dp_lds_bytes(W, IPX_MAX_SEG, REV, maxcols), b, p, 0, top, maxcols, pass, exact);
}

template <class BE>
static void ipx_plan_pass(BE &be, const IpxBatch &b, const IpxPlan &p, int pass, int na)
{
    be.launch(IPX_KEY(IPX_K_PLAN, 0), k_plan_zero, 1, 128, 0, p);
    be.launch(IPX_KEY(IPX_K_PLAN, 0), k_plan_count, be.flat_grid(b.n_jobs) * 256 / IPX_PLAN_BLOCK + 1, IPX_PLAN_BLOCK, IPX_PLAN_LDS, b, p, pass);
    be.launch(IPX_KEY(IPX_K_PLAN, 0), k_plan_scan, 1, 64, 0, p, na, pass);
    be.launch(IPX_KEY(IPX_K_PLAN, 0), k_plan_scatter, be.flat_grid(b.n_jobs) * 256 / IPX_PLAN_BLOCK + 1, IPX_PLAN_BLOCK, IPX_PLAN_LDS, b, p, pass);
}

template <class BE>
static void ipx_run_pipeline(BE &be, const IpxBatch &b, const IpxWorkspace &ws, const IpxDims &d)
{
    int maxcols = d.max_ref_len + 4;
    if (getenv("IPX_DEBUG_MAXCOLS")) maxcols = atoi(getenv("IPX_DEBUG_MAXCOLS"));   // timing experiments only (breaks score2)
    be.launch(IPX_KEY(IPX_K_INIT, 0), k_init, be.flat_grid(b.n_jobs), 256, 0, b);
    be.zero_u32(b.cigar_cursor, 1);
    be.zero_u32(ws.tb_list_n, 8);

    if (b.score_size == 2 && b.word_first_len > 0 && d.max_read_len >= b.word_first_len) {
        // long reads: 16-bit pass first, then try to prove the 8-bit overflow from the end diagonal
        ipx_plan_pass(be, b, ws.plan, IPX_PASS_WORD_FIRST, 16);
        ipx_launch_dp<BE, 8, false, false>(be, b, ws.plan, d.has16, maxcols, IPX_K_WORD_FIRST, IPX_PASS_WORD_FIRST);
        {
            int cap = 64 * d.max_read_len;                        // one wave's reads
            if (cap > 60 * 1024) cap = 60 * 1024;
            be.launch(IPX_KEY(IPX_K_PROVE, 0), k_prove_overflow, be.flat_grid(b.n_jobs * 4), 64, cap + 64, b, cap);
        }
    }
    if (b.score_size != 1) {                                     // 8-bit forward pass (ssw.c:842-843)
        ipx_plan_pass(be, b, ws.plan, IPX_PASS_BYTE_FWD, 8);
        ipx_launch_dp<BE, 16, false, true>(be, b, ws.plan, d.has8, maxcols, IPX_K_BYTE_FWD, IPX_PASS_BYTE_FWD);
        // reads whose lower-bound stage was inconclusive: exact 8-bit pass
        ipx_plan_pass(be, b, ws.plan, IPX_PASS_BYTE_FWD_EXACT, 8);
        ipx_launch_dp<BE, 16, false, false>(be, b, ws.plan, d.has8, maxcols, IPX_K_BYTE_FWD_X, IPX_PASS_BYTE_FWD_EXACT);
    }
    if (b.score_size != 0) {                                     // 16-bit forward pass (ssw.c:844-847, 853-855)
        ipx_plan_pass(be, b, ws.plan, IPX_PASS_WORD_FWD, 16);
        ipx_launch_dp<BE, 8, false, false>(be, b, ws.plan, d.has16, maxcols, IPX_K_WORD_FWD, IPX_PASS_WORD_FWD);
    }
    if (b.flag != 0) {                                           // begin position (ssw.c:872-886)
        if (b.score_size != 1) {
            ipx_plan_pass(be, b, ws.plan, IPX_PASS_BYTE_REV, 8);
            ipx_launch_dp<BE, 16, true, false>(be, b, ws.plan, d.has8, maxcols, IPX_K_BYTE_REV, IPX_PASS_BYTE_REV);
        }
        if (b.score_size != 0) {
            ipx_plan_pass(be, b, ws.plan, IPX_PASS_WORD_REV, 16);
            ipx_launch_dp<BE, 8, true, false>(be, b, ws.plan, d.has16, maxcols, IPX_K_WORD_REV, IPX_PASS_WORD_REV);
        }
        if ((7 & b.flag) != 0) {                                 // CIGAR (ssw.c:894-916)
            be.launch(IPX_KEY(IPX_K_TB_LIST, 0), k_tb_list, be.flat_grid(b.n_jobs) * 256 / IPX_PLAN_BLOCK + 1, IPX_PLAN_BLOCK, IPX_PLAN_LDS, b, ws.tb_list, ws.tb_list_n, ws.tb_esc);
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
            const int64_t fuse_max = getenv("IPX_TBF_FUSE_MAX") ? atoll(getenv("IPX_TBF_FUSE_MAX")) : 20000;
            const int per_want = (int)((b.n_jobs + 63) / 64) + 1, per_have = ws.tbf_waves / 7;
            if (b.n_jobs <= fuse_max && per_have >= 1) {
                const int per = per_want < per_have ? per_want : per_have;
                be.launch(IPX_KEY(IPX_K_TRACEBACK, 9), k_tb_fast_all, 7 * per, 64, ipx_tbf_lds_bytes(), b, (const uint32_t *)ws.tb_list,
                          (const uint32_t *)ws.tb_list_n, rowcap, ws.tbf_scratch, ws.tb_esc, ws.tb_esc_n, per);
            } else {
            IPX_TBF_LAUNCH(1) IPX_TBF_LAUNCH(2) IPX_TBF_LAUNCH(3) IPX_TBF_LAUNCH(4)
            IPX_TBF_LAUNCH(5) IPX_TBF_LAUNCH(6) IPX_TBF_LAUNCH(7)
            }
#undef IPX_TBF_LAUNCH
            // everything else: one wavefront per job
            be.launch(IPX_KEY(IPX_K_TRACEBACK, 1), k_tb_coop, ws.tb1_waves, 64, ipx_tbc_lds_bytes(ws.tb1.arrcap), b,
                      (const uint32_t *)ws.tb_esc, (const uint32_t *)ws.tb_esc_n, ws.tb1.dir, (int64_t)ws.tb1.dircap,
                      ws.tb1.arrcap, ws.tb1.cig, ws.tb1.cigcap);
        }
    }
}


// reads at least this long are very likely to overflow the 8-bit pass (their best possible score is
// >= 1.4x the overflow threshold): they take the 16-bit pass first.  Any value is correct; it only
// moves work between passes.
static inline int ipx_word_first_len(const int8_t *mat, int bias)
{
    int mx = 0;
    for (int k = 0; k < 25; ++k) if (mat[k] > mx) mx = mat[k];
    if (mx <= 0) return 0;
    const int cap = 255 - bias;
    return (cap * 14 / 10 + mx - 1) / mx;
}

// shortest read that could overflow the 8-bit pass: len * max(mat) >= 255 - bias
static inline int ipx_byte_safe_len(const int8_t *mat, int bias)
{
    int mx = 0;
    for (int k = 0; k < 25; ++k) if (mat[k] > mx) mx = mat[k];
    if (mx <= 0) return 0x7FFFFFFF;
    return (255 - bias + mx - 1) / mx;
}

// scratch sizing shared by both back-ends -------------------------------------------------------
struct IpxTbSizing { int arrcap, dircap, cigcap; };
static inline IpxTbSizing ipx_tb1_sizing(const IpxDims &d)
{
    IpxTbSizing s;
    const int len = d.max_read_len > d.max_ref_len ? d.max_read_len : d.max_ref_len;
    s.arrcap = 2 * (len > 0 ? len : 1) + 8;                      // band_width <= len (ssw.c:669)
    s.dircap = (2 * (len > 0 ? len : 1) + 1) * (d.max_read_len > 0 ? d.max_read_len : 1);
    s.cigcap = d.max_read_len + d.max_ref_len + 8;
    return s;
}
// the one-wave-per-job kernel needs direction bytes and CIGAR runs for ONE job per block
static inline size_t ipx_tbc_bytes_per_block(const IpxTbSizing &s)
{
    return (((size_t)s.dircap + 15) & ~(size_t)15) + 4ull * (size_t)s.cigcap + 16;
}
