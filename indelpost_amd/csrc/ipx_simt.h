// ipx_simt.h -- the handful of wavefront primitives the kernels are written against.
//
// Product build (hipcc, gfx950): every primitive is a CDNA4 instruction -- DPP lane moves inside a
// 16-lane row, packed 16-bit VALU (v_pk_*_i16/u16 with clamp), ballots -- no portability layer.
// Test build (g++ -DIPX_CPU_EMU, tests/emu/): the same kernel source runs under a lock-step
// 64-fiber wave emulator so the kernel logic can be checked (and sanitised) without a GPU.  The
// emulator is test infrastructure; the shipped library contains none of it.
#pragma once
#include <stdint.h>

typedef uint32_t pk16; // two 16-bit DP cells: lo half = even alignment slot, hi half = odd slot

#if defined(IPX_CPU_EMU)
// ------------------------------------------------------------------------------------------------
// lock-step emulation (tests only)
// ------------------------------------------------------------------------------------------------
#include <string.h>
#include <stdio.h>
#include <stdlib.h>
struct uint2 { uint32_t x, y; };
#define IPX_KERNEL
#define IPX_KERNEL_WAVE
#define IPX_KERNEL_WAVE_OCC(w)
#define IPX_DEV static inline
#define IPX_HD static inline
#define IPX_UNROLL
#define IPX_VMEM_FENCE() ((void)0)
#define IPX_COMPILER_FENCE() ((void)0)
#define IPX_NOUNROLL
#define IPX_RESTRICT
#define IPX_KEEP_VGPR(x) ((void)0)
#define IPX_RAISE_PRIO(b) ((void)0)
namespace ipx_emu {
struct LaneCtx { int tid; int bid; int gdim; int bdim; unsigned char *lds; };
LaneCtx &cur();
void block_barrier();                              // all fibers of the block rendezvous
uint32_t exchange(uint32_t v, int src_lane);       // read `v` of lane src_lane (same wave); -1 -> 0
uint64_t ballot(bool p);
} // namespace ipx_emu
#define IPX_TID (ipx_emu::cur().tid)
#define IPX_BID (ipx_emu::cur().bid)
#define IPX_GDIM (ipx_emu::cur().gdim)
#define IPX_BDIM (ipx_emu::cur().bdim)
#define IPX_LDS_BASE (ipx_emu::cur().lds)
#define IPX_SYNC() ipx_emu::block_barrier()

IPX_DEV uint32_t xl_shfl(uint32_t v, int src) { return ipx_emu::exchange(v, (IPX_TID & ~63) + src); }
IPX_DEV uint64_t xl_ballot(bool p) { return ipx_emu::ballot(p); }
IPX_DEV bool xl_any(bool p) { return ipx_emu::ballot(p) != 0; }
IPX_DEV uint32_t xl_first(uint32_t v) { return ipx_emu::exchange(v, (IPX_TID & ~63)); }
template <int LANE> IPX_DEV uint32_t xl_readlane(uint32_t v) { return ipx_emu::exchange(v, (IPX_TID & ~63) + LANE); }
IPX_DEV int lane_id() { return IPX_TID & 63; }
// lane i <- lane i-1 inside its 16-lane row, first lane of the row <- 0   (DPP row_shr:1)
IPX_DEV uint32_t xl_row_shr1(uint32_t v) { int l = lane_id(); return ipx_emu::exchange(v, (l & 15) ? (IPX_TID - 1) : -1); }
template <int N> IPX_DEV uint32_t xl_row_shr(uint32_t v) { int l = lane_id(); return ipx_emu::exchange(v, (l & 15) >= N ? (IPX_TID - N) : -1); }
// lane i <- lane i+1 inside its 16-lane row, last lane of the row <- 0   (DPP row_shl:1)
IPX_DEV uint32_t xl_row_shl1(uint32_t v) { int l = lane_id(); return ipx_emu::exchange(v, (l & 15) != 15 ? (IPX_TID + 1) : -1); }
// the same across the whole wavefront (DPP wave_shr:1 / wave_shl:1, GFX9): lane i <- lane i-1 (lane 0 <- 0) / lane i <- lane i+1 (lane 63 <- 0)
IPX_DEV uint32_t xl_wave_shr1(uint32_t v) { int l = lane_id(); return ipx_emu::exchange(v, l > 0 ? (IPX_TID - 1) : -1); }
IPX_DEV uint32_t xl_wave_shl1(uint32_t v) { int l = lane_id(); return ipx_emu::exchange(v, l < 63 ? (IPX_TID + 1) : -1); }
IPX_DEV uint32_t xl_xor1(uint32_t v) { return ipx_emu::exchange(v, IPX_TID ^ 1); }   // quad_perm [1,0,3,2]
IPX_DEV uint32_t xl_xor2(uint32_t v) { return ipx_emu::exchange(v, IPX_TID ^ 2); }   // quad_perm [2,3,0,1]
IPX_DEV uint32_t xl_half_mirror(uint32_t v) { return ipx_emu::exchange(v, IPX_TID ^ 7); }  // row_half_mirror
IPX_DEV uint32_t xl_mirror(uint32_t v) { return ipx_emu::exchange(v, IPX_TID ^ 15); }      // row_mirror

IPX_DEV int16_t sat16(int v) { return (int16_t)(v > 32767 ? 32767 : v < -32768 ? -32768 : v); }
IPX_DEV pk16 pk_make(int lo, int hi) { return (uint32_t)(uint16_t)lo | ((uint32_t)(uint16_t)hi << 16); }
IPX_DEV int pk_lo(pk16 a) { return (int16_t)(a & 0xFFFF); }
IPX_DEV int pk_hi(pk16 a) { return (int16_t)(a >> 16); }
IPX_DEV unsigned pk_ulo(pk16 a) { return a & 0xFFFF; }
IPX_DEV unsigned pk_uhi(pk16 a) { return a >> 16; }
IPX_DEV pk16 pk_add_sat(pk16 a, pk16 b) { return pk_make(sat16(pk_lo(a) + pk_lo(b)), sat16(pk_hi(a) + pk_hi(b))); }
IPX_DEV pk16 pk_add(pk16 a, pk16 b) { return pk_make(pk_lo(a) + pk_lo(b), pk_hi(a) + pk_hi(b)); }
IPX_DEV pk16 pk_sub(pk16 a, pk16 b) { return pk_make(pk_lo(a) - pk_lo(b), pk_hi(a) - pk_hi(b)); }
IPX_DEV pk16 pk_subus(pk16 a, pk16 b) {
    unsigned l = pk_ulo(a) > pk_ulo(b) ? pk_ulo(a) - pk_ulo(b) : 0, h = pk_uhi(a) > pk_uhi(b) ? pk_uhi(a) - pk_uhi(b) : 0;
    return l | (h << 16);
}
IPX_DEV pk16 pk_max(pk16 a, pk16 b) { return pk_make(pk_lo(a) > pk_lo(b) ? pk_lo(a) : pk_lo(b), pk_hi(a) > pk_hi(b) ? pk_hi(a) : pk_hi(b)); }
IPX_DEV pk16 pk_minu(pk16 a, pk16 b) {
    unsigned l = pk_ulo(a) < pk_ulo(b) ? pk_ulo(a) : pk_ulo(b), h = pk_uhi(a) < pk_uhi(b) ? pk_uhi(a) : pk_uhi(b);
    return l | (h << 16);
}
IPX_DEV pk16 pk_mul(pk16 a, pk16 b) { return ((pk_ulo(a) * pk_ulo(b)) & 0xFFFFu) | (((pk_uhi(a) * pk_uhi(b)) & 0xFFFFu) << 16); }
IPX_DEV pk16 pk_shr1(pk16 a) { return ((a >> 1) & 0x7FFF7FFFu); }
IPX_DEV uint32_t ubfe(uint32_t v, uint32_t off, uint32_t width) { return (v >> off) & ((1u << width) - 1u); }
// (low | high) 16 bits of a and of b -> packed pair (lo half from a, hi half from b)
IPX_DEV pk16 pk_lo16_pair(uint32_t a, uint32_t b) { return (a & 0xFFFFu) | (b << 16); }
IPX_DEV pk16 pk_hi16_pair(uint32_t a, uint32_t b) { return (a >> 16) | (b & 0xFFFF0000u); }
// v_perm_b32: result byte i = byte sel[i] of the 8-byte table {hi:lo} (0..3 = lo, 4..7 = hi), 0x0c = constant 0, 8..11 = 0x00 / 0xff by bit 15 / 31 / 47 / 63
IPX_DEV uint32_t pk_perm(uint32_t hi, uint32_t lo, uint32_t sel)
{
    const uint64_t t = ((uint64_t)hi << 32) | lo;
    uint32_t r = 0;
    for (int i = 0; i < 4; ++i) {
        const uint32_t k = (sel >> (8 * i)) & 0xFFu;
        uint32_t v;
        if (k < 8u) v = (uint32_t)(t >> (8 * k)) & 0xFFu;
        else if (k == 0x0cu) v = 0u;
        else if (k >= 8u && k <= 11u) v = ((t >> (16 * (k - 8u) + 15)) & 1u) ? 0xFFu : 0u;   // sign of byte 1 / 3 / 5 / 7, replicated
        else { fprintf(stderr, "emu: pk_perm selector %#x not modelled\n", k); abort(); }
        r |= v << (8 * i);
    }
    return r;
}
// per half: the signed high byte, sign-extended to 16 bit (v_pk_ashrrev_i16 by 8)
IPX_DEV pk16 pk_sext_hi8(pk16 x) { return pk_make(pk_lo(x) >> 8, pk_hi(x) >> 8); }
// v_alignbyte_b32: four consecutive bytes of the eight-byte value {hi:lo}, starting at byte sh & 3
IPX_DEV uint32_t xl_alignbyte(uint32_t hi, uint32_t lo, uint32_t sh) { return (uint32_t)(((((uint64_t)hi << 32) | lo) >> (8 * (sh & 3u))) & 0xFFFFFFFFu); }
IPX_DEV int8_t load_stream_i8(const int8_t *p) { return *p; }
IPX_DEV uint32_t load_global_u32(const uint32_t *p) { return *p; }
IPX_DEV void store_global_u32(uint32_t *p, uint32_t v) { *p = v; }
IPX_DEV uint32_t atomic_add_u32(uint32_t *p, uint32_t v) { uint32_t o = *p; *p = o + v; return o; }
IPX_DEV uint32_t atomic_or_u32(uint32_t *p, uint32_t v) { uint32_t o = *p; *p = o | v; return o; }
// packed IEEE half precision (v_pk_add_f16 / v_pk_max_f16 / v_pk_maximum3_f16), round to nearest even; no NaN handling
// (the kernels only ever hold finite values)
IPX_DEV float emu_h2f(uint32_t h)
{
    const uint32_t s = (h >> 15) & 1u, e = (h >> 10) & 31u, m = h & 0x3FFu;
    float v;
    if (e == 0) v = (float)m * (1.0f / 16777216.0f);                 // subnormal: m * 2^-24
    else if (e == 31) v = 65536.0f * 65536.0f;                      // inf (never produced here)
    else { v = (float)(0x400u | m); int k = (int)e - 25; while (k > 0) { v *= 2.0f; --k; } while (k < 0) { v *= 0.5f; ++k; } }
    return s ? -v : v;
}
IPX_DEV uint32_t emu_f2h(float f)
{
    uint32_t x; memcpy(&x, &f, 4);
    const uint32_t s = (x >> 16) & 0x8000u;
    const int e = (int)((x >> 23) & 0xFF) - 127 + 15;
    uint32_t m = x & 0x7FFFFFu;
    if (((x >> 23) & 0xFF) == 0) return s;                           // zero (float subnormals are far below half range)
    if (e >= 31) return s | 0x7C00u;
    if (e <= 0) {                                                    // half subnormal
        if (e < -10) return s;
        m |= 0x800000u;
        const int sh = 14 - e;                                       // bits dropped
        uint32_t r = m >> sh;
        const uint32_t rem = m & ((1u << sh) - 1u), half = 1u << (sh - 1);
        if (rem > half || (rem == half && (r & 1u))) ++r;
        return s | r;
    }
    uint32_t r = ((uint32_t)e << 10) | (m >> 13);
    const uint32_t rem = m & 0x1FFFu;
    if (rem > 0x1000u || (rem == 0x1000u && (r & 1u))) ++r;          // (a carry into the exponent is the right answer)
    return s | r;
}
IPX_DEV float emu_hmax(float a, float b) { return a > b ? a : b; }
IPX_DEV pk16 pkh_add(pk16 a, pk16 b) { return emu_f2h(emu_h2f(a & 0xFFFFu) + emu_h2f(b & 0xFFFFu)) | (emu_f2h(emu_h2f(a >> 16) + emu_h2f(b >> 16)) << 16); }
IPX_DEV pk16 pkh_max(pk16 a, pk16 b) { return emu_f2h(emu_hmax(emu_h2f(a & 0xFFFFu), emu_h2f(b & 0xFFFFu))) | (emu_f2h(emu_hmax(emu_h2f(a >> 16), emu_h2f(b >> 16))) << 16); }
IPX_DEV pk16 pkh_max3(pk16 a, pk16 b, pk16 c) { return pkh_max(pkh_max(a, b), c); }

#else
// ------------------------------------------------------------------------------------------------
// gfx950
// ------------------------------------------------------------------------------------------------
#include <hip/hip_runtime.h>
#define IPX_KERNEL __global__
#define IPX_KERNEL_WAVE __global__ __launch_bounds__(64)   // block = one wavefront: the whole VGPR file is available
#define IPX_KERNEL_WAVE_OCC(w) __global__ __launch_bounds__(64, w)   // ... but ask for w waves per SIMD (register budget 512/w)
#define IPX_DEV __device__ __forceinline__
#define IPX_HD __host__ __device__ inline
#define IPX_UNROLL _Pragma("unroll")
// wait for every outstanding global-memory operation here, and keep later ones below this point
#define IPX_VMEM_FENCE() asm volatile("s_waitcnt vmcnt(0)" ::: "memory")
#define IPX_NOUNROLL _Pragma("unroll 1")
// the compiler may not carry memory values (or move memory operations) across this point
#define IPX_COMPILER_FENCE() asm volatile("" ::: "memory")
#define IPX_RESTRICT __restrict__
// keep a value in a vector register of its own (the compiler may not re-derive it from a lane mask)
#define IPX_KEEP_VGPR(x) asm volatile("" : "+v"(x))
// r04: the latency-bound kernels (proofs, planner, traceback, the stepped passes' tier launch) raise their waves' issue priority: beside
// the other streams' wavefront kernels -- old waves whose hand-scheduled stripes never leave the vector ALU a free slot -- their dependent
// chains otherwise advance a few instructions per turn (config 4, four streams: 2.4 to 6 times their duration alone)
#define IPX_RAISE_PRIO(b) do { if ((b).lat_prio) __builtin_amdgcn_s_setprio(3); } while (0)
#define IPX_TID ((int)threadIdx.x)
#define IPX_BID ((int)blockIdx.x)
#define IPX_GDIM ((int)gridDim.x)
#define IPX_BDIM ((int)blockDim.x)
extern __shared__ __attribute__((aligned(16))) unsigned char ipx_dyn_lds[];
#define IPX_LDS_BASE (ipx_dyn_lds)
#define IPX_SYNC() __syncthreads()

typedef short ipx_s2 __attribute__((ext_vector_type(2)));
typedef unsigned short ipx_u2 __attribute__((ext_vector_type(2)));

IPX_DEV int lane_id() { return (int)(threadIdx.x & 63); }
IPX_DEV uint32_t xl_shfl(uint32_t v, int src) { return (uint32_t)__shfl((int)v, src, 64); }
IPX_DEV uint64_t xl_ballot(bool p) { return __builtin_amdgcn_ballot_w64(p); }
IPX_DEV bool xl_any(bool p) { return __builtin_amdgcn_ballot_w64(p) != 0; }
IPX_DEV uint32_t xl_first(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }
template <int LANE> IPX_DEV uint32_t xl_readlane(uint32_t v) { return (uint32_t)__builtin_amdgcn_readlane((int)v, LANE); }   // v_readlane_b32
// DPP controls: row_shr:1 = 0x111, quad_perm[1,0,3,2] = 0xB1, quad_perm[2,3,0,1] = 0x4E,
// row_mirror = 0x140, row_half_mirror = 0x141.  bound_ctrl=1 -> out-of-row source reads 0.
IPX_DEV uint32_t xl_row_shr1(uint32_t v) { return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, true); }
template <int N> IPX_DEV uint32_t xl_row_shr(uint32_t v) { return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x110 + N, 0xf, 0xf, true); }   // row_shr:N
IPX_DEV uint32_t xl_row_shl1(uint32_t v) { return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x101, 0xf, 0xf, true); }    // row_shl:1
// whole-wavefront shifts (wave_shr:1 = 0x138, wave_shl:1 = 0x130; GFX9 DPP controls, present on gfx950: tools/ubench_dpp.hip -- same
// cost as a row shift)
IPX_DEV uint32_t xl_wave_shr1(uint32_t v) { return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x138, 0xf, 0xf, true); }
IPX_DEV uint32_t xl_wave_shl1(uint32_t v) { return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x130, 0xf, 0xf, true); }
IPX_DEV uint32_t xl_xor1(uint32_t v) { return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xf, 0xf, true); }
IPX_DEV uint32_t xl_xor2(uint32_t v) { return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x4E, 0xf, 0xf, true); }
IPX_DEV uint32_t xl_half_mirror(uint32_t v) { return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x141, 0xf, 0xf, true); }
IPX_DEV uint32_t xl_mirror(uint32_t v) { return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x140, 0xf, 0xf, true); }

IPX_DEV pk16 pk_make(int lo, int hi) { return (uint32_t)(uint16_t)lo | ((uint32_t)(uint16_t)hi << 16); }
IPX_DEV int pk_lo(pk16 a) { return (int16_t)(a & 0xFFFF); }
IPX_DEV int pk_hi(pk16 a) { return (int16_t)(a >> 16); }
IPX_DEV unsigned pk_ulo(pk16 a) { return a & 0xFFFF; }
IPX_DEV unsigned pk_uhi(pk16 a) { return a >> 16; }
#define IPX_S2(x) __builtin_bit_cast(ipx_s2, (x))
#define IPX_U2(x) __builtin_bit_cast(ipx_u2, (x))
#define IPX_PK(x) __builtin_bit_cast(uint32_t, (x))
IPX_DEV pk16 pk_add_sat(pk16 a, pk16 b) { return IPX_PK(__builtin_elementwise_add_sat(IPX_S2(a), IPX_S2(b))); }  // v_pk_add_i16 clamp
IPX_DEV pk16 pk_add(pk16 a, pk16 b) { return IPX_PK(IPX_S2(a) + IPX_S2(b)); }                                      // v_pk_add_u16
IPX_DEV pk16 pk_sub(pk16 a, pk16 b) { return IPX_PK(IPX_S2(a) - IPX_S2(b)); }                                      // v_pk_sub_i16
IPX_DEV pk16 pk_subus(pk16 a, pk16 b) { return IPX_PK(__builtin_elementwise_sub_sat(IPX_U2(a), IPX_U2(b))); }      // v_pk_sub_u16 clamp
IPX_DEV pk16 pk_max(pk16 a, pk16 b) { return IPX_PK(__builtin_elementwise_max(IPX_S2(a), IPX_S2(b))); }            // v_pk_max_i16
IPX_DEV pk16 pk_minu(pk16 a, pk16 b) { return IPX_PK(__builtin_elementwise_min(IPX_U2(a), IPX_U2(b))); }           // v_pk_min_u16
IPX_DEV pk16 pk_sext_hi8(pk16 x) { return IPX_PK(IPX_S2(x) >> 8); }   // per half: signed high byte -> 16 bit: v_pk_ashrrev_i16
IPX_DEV uint32_t xl_alignbyte(uint32_t hi, uint32_t lo, uint32_t sh) { return __builtin_amdgcn_alignbyte(hi, lo, sh); }   // v_alignbyte_b32
IPX_DEV pk16 pk_mul(pk16 a, pk16 b) { return IPX_PK(IPX_U2(a) * IPX_U2(b)); }                                     // v_pk_mul_lo_u16
IPX_DEV pk16 pk_shr1(pk16 a) { return IPX_PK(IPX_U2(a) >> 1); }                                                    // v_pk_lshrrev_b16
IPX_DEV uint32_t ubfe(uint32_t v, uint32_t off, uint32_t width) { return __builtin_amdgcn_ubfe(v, off, width); }
// (low | high) 16 bits of a and of b -> packed pair (lo half from a, hi half from b): one v_perm_b32
IPX_DEV pk16 pk_lo16_pair(uint32_t a, uint32_t b) { return __builtin_amdgcn_perm(b, a, 0x05040100u); }
IPX_DEV pk16 pk_hi16_pair(uint32_t a, uint32_t b) { return __builtin_amdgcn_perm(b, a, 0x07060302u); }
// byte table lookup: result byte i = byte sel[i] of {hi:lo} (0..3 = lo, 4..7 = hi), selector 0x0c = constant 0
IPX_DEV uint32_t pk_perm(uint32_t hi, uint32_t lo, uint32_t sel) { return __builtin_amdgcn_perm(hi, lo, sel); }
// read-once data (the read letters): keep it out of the way of the L2-resident per-block scratch
IPX_DEV int8_t load_stream_i8(const int8_t *p) { return __builtin_nontemporal_load(p); }
// explicitly global (not flat) accesses: they count in vmcnt only, so LDS waits do not wait for them
IPX_DEV uint32_t load_global_u32(const uint32_t *p) { return *(const __attribute__((address_space(1))) uint32_t *)p; }
IPX_DEV void store_global_u32(uint32_t *p, uint32_t v) { *(__attribute__((address_space(1))) uint32_t *)p = v; }
IPX_DEV uint32_t atomic_add_u32(uint32_t *p, uint32_t v) { return atomicAdd(p, v); }
IPX_DEV uint32_t atomic_or_u32(uint32_t *p, uint32_t v) { return atomicOr(p, v); }
// packed half precision: v_pk_add_f16, v_pk_max_f16, v_pk_maximum3_f16 (new in gfx950)
typedef _Float16 ipx_h2 __attribute__((ext_vector_type(2)));
#define IPX_H2(x) __builtin_bit_cast(ipx_h2, (x))
IPX_DEV pk16 pkh_add(pk16 a, pk16 b) { return IPX_PK(IPX_H2(a) + IPX_H2(b)); }
IPX_DEV pk16 pkh_max(pk16 a, pk16 b) { return IPX_PK(__builtin_elementwise_max(IPX_H2(a), IPX_H2(b))); }
IPX_DEV pk16 pkh_max3(pk16 a, pk16 b, pk16 c) { return IPX_PK(__builtin_elementwise_maximum(__builtin_elementwise_maximum(IPX_H2(a), IPX_H2(b)), IPX_H2(c))); }
#endif

// ---- helpers shared by both builds -------------------------------------------------------------
// per-half "non-zero -> 0xFFFF" mask
IPX_DEV pk16 pk_nzmask(pk16 x) { return pk_sub(0u, pk_minu(x, 0x00010001u)); }
#if defined(IPX_CPU_EMU)
IPX_DEV pk16 pk_select(pk16 mask, pk16 a, pk16 b) { return (a & mask) | (b & ~mask); }
IPX_DEV pk16 pk_nzmask_pos(pk16 x) { return pk_nzmask(x); }
#else
// the same for halves below 0x8000: 0 - x is negative exactly when x is not 0, its sign spread over the half is the mask -- two
// packed instructions with inline constants (the compiler turns pk_nzmask into two compares, two selects and a v_perm); op_sel_hi 0
// on the shift count: an integer inline constant fills the LOW half only, both halves must read that one
IPX_DEV pk16 pk_nzmask_pos(pk16 x) { pk16 r; asm("v_pk_sub_i16 %0, 0, %1\n\tv_pk_ashrrev_i16 %0, 15, %0 op_sel_hi:[0,1]" : "=&v"(r) : "v"(x)); return r; }
// one v_bfi_b32 (left to itself the compiler sometimes splits it into v_and + v_and_or)
IPX_DEV pk16 pk_select(pk16 mask, pk16 a, pk16 b) { pk16 r; asm("v_bfi_b32 %0, %1, %2, %3" : "=v"(r) : "v"(mask), "v"(a), "v"(b)); return r; }
#endif
IPX_DEV pk16 pk_splat(int v) { return pk_make(v, v); }
// half-precision bit pattern of an integer 0..2047 (exact), and back (non-negative, integer valued)
IPX_HD uint32_t ipx_f16_from_uint(uint32_t v)
{
    if (v == 0) return 0;
    if (v > 2047u) v = 2047u;
    int e = 10;
    while (!((v >> e) & 1u)) --e;
    return ((uint32_t)(e + 15) << 10) | ((v << (10 - e)) & 0x3FFu);
}
IPX_HD uint32_t ipx_f16_to_uint(uint32_t p)
{
    const int e = (int)((p >> 10) & 31u);
    if (e < 15) return 0;
    const uint32_t m = 0x400u | (p & 0x3FFu);
    return e >= 25 ? m << (e - 25) : m >> (25 - e);
}
// signed small integer -> half pattern (sign bit + magnitude)
IPX_HD uint32_t ipx_f16_from_int(int v) { return v < 0 ? (0x8000u | ipx_f16_from_uint((uint32_t)(-v))) : ipx_f16_from_uint((uint32_t)v); }
