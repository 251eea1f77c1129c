// emu.cpp -- lock-step wavefront emulator for the CPU-only test suite (TEST INFRASTRUCTURE).
//
// Compiles the PRODUCT kernel source (indelpost_amd/csrc/ipx_kernels.h, ipx_pipeline.h) with
// -DIPX_CPU_EMU and runs every thread of a block as a fiber; cross-lane primitives (DPP moves,
// ballots) rendezvous through a strict round-robin scheduler, so wave-uniform control flow behaves
// exactly as on the GPU.  This lets `pytest -m "not gpu"` check the kernel logic and the launch
// sequence against the oracle without a device.  It is not a fallback: the shipped library
// (libindelpost_hip.so) contains none of this and refuses to run without a GPU.
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <functional>
#include <vector>

#include "../../indelpost_amd/csrc/ipx_pipeline.h"

// ---- minimal x86-64 context switch ----------------------------------------------------------------
extern "C" void ipx_ctx_switch(void **save_sp, void *load_sp);
asm(".text\n"
    ".globl ipx_ctx_switch\n"
    ".type ipx_ctx_switch,@function\n"
    "ipx_ctx_switch:\n"
    "  pushq %rbp\n  pushq %rbx\n  pushq %r12\n  pushq %r13\n  pushq %r14\n  pushq %r15\n"
    "  movq %rsp, (%rdi)\n"
    "  movq %rsi, %rsp\n"
    "  popq %r15\n  popq %r14\n  popq %r13\n  popq %r12\n  popq %rbx\n  popq %rbp\n"
    "  ret\n");

namespace ipx_emu {

struct Fiber {
    void *sp;
    char *stack;
    bool done;
    LaneCtx lc;
};

static std::vector<Fiber> g_fibers;
static int g_cur = -1;
static void *g_sched_sp;
static std::function<void()> g_body;
static std::vector<uint32_t> g_slot[2];
static std::vector<int> g_parity;          // per fiber: which exchange buffer comes next
static std::vector<long> g_epoch;          // per fiber: number of cross-lane operations executed so far
static const size_t STACK_BYTES = 512 * 1024;

LaneCtx &cur() { return g_fibers[g_cur].lc; }

static void yield_to_sched() { ipx_ctx_switch(&g_fibers[g_cur].sp, g_sched_sp); }

static void fiber_entry()
{
    g_body();
    g_fibers[g_cur].done = true;
    for (;;) yield_to_sched();
}

// A cross-lane operation must be reached by every lane of the wave (on the GPU an inactive lane feeds
// zeros/stale data into DPP and ballots): abort loudly when the lanes of a wave disagree.
static void check_convergent(int me, const char *what)
{
    const int base = me & ~63;
    const int n = (int)g_fibers.size();
    // lanes scheduled before `me` have already resumed and may have reached the NEXT operation
    for (int k = 0; k < 64 && base + k < n; ++k)
        if ((g_epoch[base + k] < g_epoch[me] || g_epoch[base + k] > g_epoch[me] + (base + k < me ? 1 : 0))) {
            fprintf(stderr, "ipx_emu: divergent %s: lane %d is at cross-lane op %ld, lane %d at %ld (block %d)\n",
                    what, me & 63, g_epoch[me], k, g_epoch[base + k], g_fibers[me].lc.bid);
            abort();
        }
}

// Block-wide barrier: waves may reach it after different numbers of cross-lane operations (e.g. a loop over the
// distinct classes of a wave), so a fiber waits here until every fiber of the block that is still running has
// arrived -- one scheduler round is not enough.
// Everybody then leaves in the NEXT scheduler round, in lane order, so the lock-step phase the lane exchanges
// rely on (a lane is at most one operation behind the lanes before it) is restored.
static long g_bar_gen = 0, g_round = 0, g_bar_open_round = -1;
static int g_bar_count = 0;

void block_barrier()
{
    const long gen = g_bar_gen;
    int alive = 0;
    for (const Fiber &f : g_fibers) alive += f.done ? 0 : 1;
    if (++g_bar_count >= alive) { g_bar_count = 0; ++g_bar_gen; g_bar_open_round = g_round; }   // last arrival opens the barrier
    while (g_bar_gen == gen || g_round <= g_bar_open_round) yield_to_sched();
}

uint32_t exchange(uint32_t v, int src_tid)
{
    const int me = g_cur;
    const int par = g_parity[me];
    g_parity[me] ^= 1;
    g_slot[par][me] = v;
    ++g_epoch[me];
    yield_to_sched();
    check_convergent(me, "lane exchange");
    return src_tid >= 0 ? g_slot[par][src_tid] : 0u;
}

uint64_t ballot(bool p)
{
    const int me = g_cur;
    const int par = g_parity[me];
    g_parity[me] ^= 1;
    g_slot[par][me] = p ? 1u : 0u;
    ++g_epoch[me];
    yield_to_sched();
    check_convergent(me, "ballot");
    const int base = me & ~63;
    uint64_t m = 0;
    const int n = (int)g_fibers.size();
    for (int k = 0; k < 64 && base + k < n; ++k)
        if (g_slot[par][base + k]) m |= (1ull << k);
    return m;
}

static void run_block(int bid, int gdim, int bdim, int lds_bytes, const std::function<void()> &body)
{
    const size_t LDS_GUARD = 16384;                                  // canary behind the block's LDS: a kernel that writes past its launch's allocation is caught below
    std::vector<unsigned char> lds((size_t)lds_bytes + LDS_GUARD, 0xCD);
    g_fibers.assign((size_t)bdim, Fiber());
    g_slot[0].assign((size_t)bdim, 0);
    g_slot[1].assign((size_t)bdim, 0);
    g_parity.assign((size_t)bdim, 0);
    g_epoch.assign((size_t)bdim, 0);
    g_bar_count = 0;
    g_bar_open_round = -1;
    g_body = body;
    for (int t = 0; t < bdim; ++t) {
        Fiber &f = g_fibers[t];
        f.stack = (char *)aligned_alloc(64, STACK_BYTES);
        f.done = false;
        f.lc.tid = t; f.lc.bid = bid; f.lc.gdim = gdim; f.lc.bdim = bdim; f.lc.lds = lds.data();
        uintptr_t top = ((uintptr_t)f.stack + STACK_BYTES) & ~(uintptr_t)63;
        void **sp = (void **)top;
        *--sp = nullptr;                   // fake return address of fiber_entry's caller
        *--sp = (void *)&fiber_entry;      // `ret` target of the first switch
        for (int k = 0; k < 6; ++k) *--sp = nullptr;   // rbp rbx r12 r13 r14 r15
        f.sp = (void *)sp;
    }
    for (;;) {
        bool any = false;
        ++g_round;
        for (int t = 0; t < bdim; ++t) {
            if (g_fibers[t].done) continue;
            any = true;
            g_cur = t;
            ipx_ctx_switch(&g_sched_sp, g_fibers[t].sp);
        }
        if (!any) break;
    }
    g_cur = -1;
    for (size_t k = 0; k < LDS_GUARD; ++k)
        if (lds[(size_t)lds_bytes + k] != 0xCD) {
            fprintf(stderr, "emu: block %d wrote LDS byte %zu, the launch reserved %d\n", bid, (size_t)lds_bytes + k, lds_bytes);
            abort();
        }
    for (int t = 0; t < bdim; ++t) free(g_fibers[t].stack);
    g_fibers.clear();
}

} // namespace ipx_emu

// ---- launcher with the interface ipx_pipeline.h expects -----------------------------------------
struct EmuBackend {
    int launches[IPX_NUM_KEYS];
    EmuBackend() { memset(launches, 0, sizeof launches); }
    int dp_grid() const { return 3; }
    int dp_grid(int, int) const { return 3; }
    int sweep_grid(int, uint64_t, uint64_t) const { return 2; }
    int flat_grid(int64_t n) const { return n > 512 ? 2 : 1; }
    int plan_grid(int64_t n) const { return n > 100 ? 3 : 1; }
    void zero_u32(uint32_t *p, int n) { memset(p, 0, sizeof(uint32_t) * (size_t)n); }
    void copy_u32(uint32_t *dst, const uint32_t *src, int n) { memcpy(dst, src, sizeof(uint32_t) * (size_t)n); }
    void note_dp(int, int, int, int) {}
    void note_dp_set(int, int, int, uint32_t, int) {}
    int dp_grid_set(int, int, uint32_t) const { return 3; }
    bool pass_predicted_empty(int) const { return false; }
    void note_f16(int kind, int S) { ++launches[IPX_KEY(IPX_K_PACK, 10 + 40 * kind + S)]; }   // (test visibility, under unused keys: half-precision launches per segLen; kind 0 16-bit column by column, 1 16-bit wavefront, 2 8-bit upper bound as a wavefront, 3 8-bit lower bound)
    template <class K, class... A>
    void launch(int kclass, K kern, int grid, int block, int lds, A... args)
    {
        ++launches[kclass];
        if (getenv("IPX_EMU_TRACE")) fprintf(stderr, "emu launch key %d grid %d block %d lds %d\n", kclass, grid, block, lds);
        std::function<void()> body = [=]() { kern(args...); };
        for (int bidx = 0; bidx < grid; ++bidx) ipx_emu::run_block(bidx, grid, block, lds, body);
    }
};

template <class T> static T *zalloc(size_t n) { return (T *)calloc(n ? n : 1, sizeof(T)); }

// C entry used by tests/emu_backend.py: same arguments as ipx_align_batch (include/indelpost_hip.h) plus the
// scoring / routing parameters the HIP context holds; launches_out (optional): launches per timing key;
// pass_jobs_out (optional): jobs that went through each of the IPX_NUM_PASSES passes
extern "C" int emu_align_batch(const int8_t *reads, const int64_t *read_off, const int8_t *refs,
                               const int64_t *ref_off, const int32_t *ref_id, const uint8_t *gap_open,
                               const uint8_t *gap_ext, const int32_t *mask_len, const int8_t *mat,
                               int64_t n_jobs, int32_t n_refs, int flag, int filters, int filterd,
                               int score_size, int routing, IpxResult *out, uint32_t *cigar_pool, uint32_t cigar_cap,
                               uint32_t *status_out, int32_t *launches_out, uint32_t *pass_jobs_out)
{
    EmuBackend be;
    IpxBatch b;
    memset(&b, 0, sizeof b);
    IpxDims *dp = zalloc<IpxDims>(1);
    IpxDims &d = *dp;
    std::vector<int64_t> refp_off((size_t)n_refs + 1);
    std::vector<int32_t> ref_len((size_t)n_refs + 1);
    int64_t tot = 0;
    for (int r = 0; r < n_refs; ++r) {
        const int len = (int)(ref_off[r + 1] - ref_off[r]);
        refp_off[r] = tot;
        ref_len[r] = len;
        tot += ((len + 3) & ~3) + IPX_REF_PAD;
        if (len > d.max_ref_len) d.max_ref_len = len;
    }
    for (int64_t i = 0; i < n_jobs; ++i) ipx_dims_add_read(d, (int)(read_off[i + 1] - read_off[i]), gap_open[i] <= gap_ext[i]);
    int8_t *packed = zalloc<int8_t>((size_t)tot + 64);
    be.launch(IPX_KEY(IPX_K_PACK, 0), k_pack_refs, 2, 256, 0, refs, ref_off, (const int64_t *)refp_off.data(), packed, n_refs);

    uint32_t status = 0, cursor = 0;
    b.n_jobs = n_jobs; b.n_refs = n_refs; b.reads = reads; b.read_off = read_off;
    b.refs_packed = packed; b.refp_off = refp_off.data(); b.ref_len = ref_len.data(); b.ref_id = ref_id;
    b.gap_open = gap_open; b.gap_ext = gap_ext; b.mask_len = mask_len;
    memcpy(b.mat, mat, 25);
    int bias = 0;
    for (int k = 0; k < 25; ++k) if (mat[k] < bias) bias = mat[k];
    b.bias = -bias;
    b.word_first_len = (routing & IPX_ROUTE_NO_WORD_FIRST) ? 0 : ipx_word_first_len(mat, -bias);
    b.use_bracket = ipx_perm_profile_ok(mat, routing) && !(routing & IPX_ROUTE_NO_BRACKET);
    b.bracket_min_len = ipx_bracket_min_len(mat);
    b.f16_max_len = ipx_f16_max_len(mat);
    b.byte_safe_len = ipx_byte_safe_len(mat, -bias);
    { int mx = 0; for (int k = 0; k < 25; ++k) if (mat[k] > mx) mx = mat[k]; b.max_match = mx; }
    b.exact_direct = ipx_perm_profile_ok(mat, routing) && !(routing & IPX_ROUTE_NO_EXACT_DIRECT);   // (the stepped selector-profile kernels: cheap where no cut can happen)
    b.flag = (uint8_t)flag; b.score_size = (uint8_t)score_size; b.filters = (uint16_t)filters; b.filterd = filterd;
    b.res = out; b.cigar_pool = cigar_pool; b.cigar_cap = cigar_cap; b.cigar_cursor = &cursor; b.status = &status;
    ipx_plan_classes(d, b, routing);
    b.plain_first = d.plain_first;
    b.plain_max_len = d.plain_max_len;
    b.cls_map = &d.cls_map[0][0];
    ipx_dims_finish(d, b.word_first_len, score_size, b.plain_first ? 0 : ipx_exact_start_len(b.byte_safe_len, b.bracket_min_len, b.use_bracket));

    b.maxcol_scratch = zalloc<uint32_t>((size_t)be.dp_grid() * 16 * (size_t)(d.max_ref_len + 8));
    IpxWorkspace ws;
    memset(&ws, 0, sizeof ws);
    ws.plan_tables = zalloc<uint32_t>(IPX_PLAN_TABLE_WORDS);
    ws.exact_starters = zalloc<uint32_t>(IPX_NUM_CLASSES);
    b.plan_counts = nullptr;
    uint32_t *offs = zalloc<uint32_t>((size_t)IPX_NUM_PASSES * 2 * (IPX_NUM_CLASSES + 1));
    uint32_t *perms = zalloc<uint32_t>(3 * (size_t)n_jobs);      // two static passes + one shared by the dynamic ones
    for (int ps = 0; ps < IPX_NUM_PASSES; ++ps) {
        IpxPlan &p = ws.plan[ps];
        p.count = ipx_plan_count_of(ws.plan_tables, ps);
        p.cursor = p.count + IPX_NUM_CLASSES;
        p.cls_off = offs + (size_t)ps * 2 * (IPX_NUM_CLASSES + 1);
        p.tile_off = p.cls_off + IPX_NUM_CLASSES + 1;
        p.perm = perms + (size_t)(ps < IPX_FIRST_DYNAMIC_PASS ? ps : IPX_FIRST_DYNAMIC_PASS) * (size_t)n_jobs;
        p.stats = nullptr;
    }
    // plan_counts[pass][class]: the count rows of plan_tables have a stride of 2 * IPX_NUM_CLASSES
    b.plan_counts = ws.plan_tables;
    ws.tb_list = zalloc<uint32_t>(IPX_TB_NLISTS * (size_t)n_jobs);
    ws.tb_esc = zalloc<uint32_t>((size_t)n_jobs);
    ws.tb_esc_n = nullptr;
    ws.tb_list_n = zalloc<uint32_t>(IPX_TB_NCOUNTERS);
    ws.tb_esc_n = ws.tb_list_n + IPX_TB_CLS_COOP;
    b.tb_bw = zalloc<uint16_t>((size_t)n_jobs);
    b.tb_diag = !(routing & IPX_ROUTE_TB_NO_DIAG);
    ws.tbd_waves = 2;
    const IpxTbSizing s1 = ipx_tb1_sizing(d);
    ws.tbf_waves = 7;
    ws.tbf_scratch = (unsigned char *)malloc(ipx_tbf_scratch_bytes_per_block(IPX_TBF_ROWCAP) * 7 + 64);
    memset(ws.tbf_scratch, 0x5A, ipx_tbf_scratch_bytes_per_block(IPX_TBF_ROWCAP) * 7);
    ws.tb1_waves = 2;
    memset(&ws.tb1, 0, sizeof ws.tb1);
    ws.tb1.arrcap = s1.arrcap; ws.tb1.dircap = s1.dircap; ws.tb1.cigcap = s1.cigcap; ws.tb1.arrcap_lds = s1.arrcap_lds;
    ws.tb1.band = (int32_t *)malloc(16ull * (size_t)s1.arrcap * ws.tb1_waves + 64);
    ws.tb1.dir = (uint8_t *)malloc((size_t)s1.dircap * ws.tb1_waves + 64);
    ws.tb1.cig = (uint32_t *)malloc(sizeof(uint32_t) * (size_t)s1.cigcap * ws.tb1_waves + 64);
    memset(ws.tb1.dir, 0x5A, (size_t)s1.dircap * ws.tb1_waves);

    {   // banded reverse pass: two job lists, counters per class, the second list's class / tile offsets
        uint32_t *rv = zalloc<uint32_t>(ipx_rev_words(n_jobs));
        ws.rev_listA = rv; ws.rev_listB = rv + n_jobs; ws.rev_cnt = rv + 2 * (size_t)n_jobs;
        ws.rev_cls_off = ws.rev_cnt + 2 * IPX_NUM_CLASSES; ws.rev_tile_off = ws.rev_cls_off + IPX_NUM_CLASSES + 2;
    }
    ws.long_state = nullptr; ws.long_stride = 0; ws.long_blocks = 0;
    if (d.max_read_len > 8 * (IPX_MAX_SEG - 1)) {          // class 64 ("64 segments or more") can occur: from 505 bp in the 16-bit passes
        ws.long_stride = (int64_t)ipx_long_state_bytes(d.max_read_len);
        ws.long_blocks = be.dp_grid();
        ws.long_state = (unsigned char *)malloc((size_t)ws.long_stride * ws.long_blocks);
    }
    if (n_jobs > 0) {
        ipx_build_static_plans(be, b, ws, d, routing);
        ipx_run_pipeline(be, b, ws, d, routing, false, (routing & IPX_ROUTE_TEST_SKIP_ALL) != 0);
        if (status & IPX_STATUS_RERUN) {                          // a pass that was left out held a job (speculation, latency tier): again, with every pass
            if (!(routing & IPX_ROUTE_TEST_SKIP_ALL)) status |= IPX_STATUS_INTERNAL;
            status &= ~(uint32_t)IPX_STATUS_RERUN;
            ++be.launches[IPX_KEY(IPX_K_PACK, 5)];              // (test visibility: the run was repeated)
            ipx_run_pipeline(be, b, ws, d, routing, false, false);
        }
    }

    *status_out = status;
    if (launches_out) for (int k = 0; k < IPX_NUM_KEYS; ++k) launches_out[k] = be.launches[k];
    if (pass_jobs_out)
        for (int ps = 0; ps < IPX_NUM_PASSES; ++ps) {
            pass_jobs_out[ps] = 0;
            for (int c = 0; c < IPX_NUM_CLASSES; ++c) pass_jobs_out[ps] += ipx_plan_count_of(ws.plan_tables, ps)[c];
        }
    if (pass_jobs_out)                                           // jobs pushed to k_tb_coop's list and to the three anti-diagonal tiers' lists
        for (int q = 0; q < 4; ++q) pass_jobs_out[11 + q] = ws.tb_list_n[IPX_TB_CLS_COOP + q];
    free(packed);
    free(b.maxcol_scratch);
    free(ws.plan_tables); free(ws.exact_starters); free(offs); free(perms);
    free(ws.tb_list); free(ws.tb_esc); free(ws.tb_list_n); free(b.tb_bw);
    free(ws.tb1.dir); free(ws.tb1.cig); free(ws.tb1.band); free(ws.tbf_scratch); free(ws.long_state);
    free(dp);
    return 0;
}

extern "C" int emu_num_keys(void) { return IPX_NUM_KEYS; }
extern "C" int emu_key(int kclass, int sub) { return IPX_KEY(kclass, sub); }
