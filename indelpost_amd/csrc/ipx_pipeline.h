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
#include "ipx_kernels.h"

struct IpxWorkspace {
    IpxPlan plan;
    uint32_t *tb_list, *tb_list_n;      // jobs that get a CIGAR
    uint32_t *tb_next, *tb_next_n;      // jobs whose band outgrew the tier-0 scratch
    IpxTbScratch tb0, tb1;
    int tb0_waves, tb1_waves;
};

struct IpxDims {
    int max_read_len;    // longest read of the batch
    int max_ref_len;     // longest window of the batch
    unsigned fwd8_mask;  // segLen buckets present among the reads, 8-bit pass (bit k = bucket k)
    unsigned fwd16_mask; // ... 16-bit pass
};

static inline void ipx_dims_add_read(IpxDims &d, int len)
{
    if (len > d.max_read_len) d.max_read_len = len;
    int c8 = (len + 15) / 16, c16 = (len + 7) / 8;
    if (c8 > IPX_MAX_SEG) c8 = IPX_MAX_SEG;
    if (c16 > IPX_MAX_SEG) c16 = IPX_MAX_SEG;
    for (int k = 0; k < 6; ++k) {
        static const int smax[6] = {8, 16, 24, 32, 48, 64};
        if (c8 <= smax[k]) { d.fwd8_mask |= 1u << k; break; }
    }
    for (int k = 0; k < 6; ++k) {
        static const int smax[6] = {8, 16, 24, 32, 48, 64};
        if (c16 <= smax[k]) { d.fwd16_mask |= 1u << k; break; }
    }
}

// kernel classes for per-kernel timing (ipx_runtime.hip records HIP events around each launch)
enum {
    IPX_K_INIT = 0, IPX_K_PLAN, IPX_K_BYTE_FWD, IPX_K_WORD_FWD, IPX_K_BYTE_REV, IPX_K_WORD_REV,
    IPX_K_TB_LIST, IPX_K_TRACEBACK, IPX_K_PACK, IPX_K_NUM
};

static inline int ipx_dp_lds_bytes(int W, int SMAX, bool rev, int maxcols)
{
    return 768 * SMAX + (rev ? 0 : 4 * (64 / W) * maxcols) + 64;
}

// timing key of a launch: kernel class * 8 + segLen bucket (0 for non-DP kernels)
#define IPX_NUM_BUCKETS 6
static const int ipx_bucket_smax[IPX_NUM_BUCKETS] = {8, 16, 24, 32, 48, 64};
static inline int ipx_bucket_of_class(int cls)
{
    for (int k = 0; k < IPX_NUM_BUCKETS; ++k) if (cls <= ipx_bucket_smax[k]) return k;
    return IPX_NUM_BUCKETS - 1;
}

// one register-resident instantiation per segLen bucket; `mask` bit k = launch bucket k
template <class BE, int W, bool REV>
static void ipx_launch_dp(BE &be, const IpxBatch &b, const IpxPlan &p, unsigned mask, int maxcols, int kclass)
{
    if (mask & 1u)
        be.launch(kclass * 8 + 0, k_dp_pass<W, 8, REV>, be.dp_grid(), 64, ipx_dp_lds_bytes(W, 8, REV, maxcols), b, p, 0, 8, maxcols);
    if (mask & 2u)
        be.launch(kclass * 8 + 1, k_dp_pass<W, 16, REV>, be.dp_grid(), 64, ipx_dp_lds_bytes(W, 16, REV, maxcols), b, p, 9, 16, maxcols);
    if (mask & 4u)
        be.launch(kclass * 8 + 2, k_dp_pass<W, 24, REV>, be.dp_grid(), 64, ipx_dp_lds_bytes(W, 24, REV, maxcols), b, p, 17, 24, maxcols);
    if (mask & 8u)
        be.launch(kclass * 8 + 3, k_dp_pass<W, 32, REV>, be.dp_grid(), 64, ipx_dp_lds_bytes(W, 32, REV, maxcols), b, p, 25, 32, maxcols);
    if (mask & 16u)
        be.launch(kclass * 8 + 4, k_dp_pass<W, 48, REV>, be.dp_grid(), 64, ipx_dp_lds_bytes(W, 48, REV, maxcols), b, p, 33, 48, maxcols);
    if (mask & 32u)
        be.launch(kclass * 8 + 5, k_dp_pass<W, 64, REV>, be.dp_grid(), 64, ipx_dp_lds_bytes(W, 64, REV, maxcols), b, p, 49, 64, maxcols);
}

template <class BE>
static void ipx_plan_pass(BE &be, const IpxBatch &b, const IpxPlan &p, int pass, int na)
{
    be.launch(IPX_K_PLAN * 8, k_plan_zero, 1, 128, 0, p);
    be.launch(IPX_K_PLAN * 8, k_plan_count, be.flat_grid(b.n_jobs), 256, 0, b, p, pass);
    be.launch(IPX_K_PLAN * 8, k_plan_scan, 1, 64, 0, p, na);
    be.launch(IPX_K_PLAN * 8, k_plan_scatter, be.flat_grid(b.n_jobs), 256, 0, b, p, pass);
}

template <class BE>
static void ipx_run_pipeline(BE &be, const IpxBatch &b, const IpxWorkspace &ws, const IpxDims &d)
{
    const int maxcols = d.max_ref_len + 4;
    // forward passes: only the segLen buckets that occur among the reads (known on the host);
    // reverse passes: the read prefix can be any length up to the longest read
    unsigned rev8 = 0, rev16 = 0;
    for (int k = 0; k < IPX_NUM_BUCKETS; ++k) {
        if ((d.fwd8_mask >> k) != 0) rev8 |= 1u << k;
        if ((d.fwd16_mask >> k) != 0) rev16 |= 1u << k;
    }

    be.launch(IPX_K_INIT * 8, k_init, be.flat_grid(b.n_jobs), 256, 0, b);
    be.zero_u32(b.cigar_cursor, 1);
    be.zero_u32(ws.tb_list_n, 1);
    be.zero_u32(ws.tb_next_n, 1);

    if (b.score_size != 1) {                                     // 8-bit forward pass (ssw.c:842-843)
        ipx_plan_pass(be, b, ws.plan, IPX_PASS_BYTE_FWD, 8);
        ipx_launch_dp<BE, 16, false>(be, b, ws.plan, d.fwd8_mask, maxcols, IPX_K_BYTE_FWD);
    }
    if (b.score_size != 0) {                                     // 16-bit forward pass (ssw.c:844-847, 853-855)
        ipx_plan_pass(be, b, ws.plan, IPX_PASS_WORD_FWD, 16);
        ipx_launch_dp<BE, 8, false>(be, b, ws.plan, d.fwd16_mask, maxcols, IPX_K_WORD_FWD);
    }
    if (b.flag != 0) {                                           // begin position (ssw.c:872-886)
        if (b.score_size != 1) {
            ipx_plan_pass(be, b, ws.plan, IPX_PASS_BYTE_REV, 8);
            ipx_launch_dp<BE, 16, true>(be, b, ws.plan, rev8, maxcols, IPX_K_BYTE_REV);
        }
        if (b.score_size != 0) {
            ipx_plan_pass(be, b, ws.plan, IPX_PASS_WORD_REV, 16);
            ipx_launch_dp<BE, 8, true>(be, b, ws.plan, rev16, maxcols, IPX_K_WORD_REV);
        }
        if ((7 & b.flag) != 0) {                                 // CIGAR (ssw.c:894-916)
            be.launch(IPX_K_TB_LIST * 8, k_tb_list, be.flat_grid(b.n_jobs), 256, 0, b, ws.tb_list, ws.tb_list_n);
            be.launch(IPX_K_TRACEBACK * 8, k_traceback, ws.tb0_waves, 64, 64, b, (const uint32_t *)ws.tb_list,
                      (const uint32_t *)ws.tb_list_n, ws.tb0, ws.tb_next, ws.tb_next_n);
            be.launch(IPX_K_TRACEBACK * 8 + 1, k_traceback, ws.tb1_waves, 64, 64, b, (const uint32_t *)ws.tb_next,
                      (const uint32_t *)ws.tb_next_n, ws.tb1, (uint32_t *)nullptr, (uint32_t *)nullptr);
        }
    }
}

// scratch sizing shared by both back-ends -------------------------------------------------------
struct IpxTbSizing { int arrcap, dircap, cigcap; };
static inline IpxTbSizing ipx_tb0_sizing(const IpxDims &d)
{
    IpxTbSizing s;
    s.arrcap = 2 * 8 + 4;                                        // band_width <= 8
    s.dircap = (2 * 8 + 1) * (d.max_read_len > 0 ? d.max_read_len : 1);
    s.cigcap = 64;
    return s;
}
static inline IpxTbSizing ipx_tb1_sizing(const IpxDims &d)
{
    IpxTbSizing s;
    const int len = d.max_read_len > d.max_ref_len ? d.max_read_len : d.max_ref_len;
    s.arrcap = 2 * (len > 0 ? len : 1) + 8;                      // band_width <= len (ssw.c:669)
    s.dircap = (2 * (len > 0 ? len : 1) + 1) * (d.max_read_len > 0 ? d.max_read_len : 1);
    s.cigcap = d.max_read_len + d.max_ref_len + 8;
    return s;
}
static inline size_t ipx_tb_bytes_per_wave(const IpxTbSizing &s)
{
    return 64ull * (3ull * 4ull * (size_t)s.arrcap + (size_t)s.dircap + 4ull * (size_t)s.cigcap);
}
