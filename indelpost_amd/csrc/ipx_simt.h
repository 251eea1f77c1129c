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
// v_perm_b32: result byte i = byte sel[i] of the 8-byte table {hi:lo} (0..3 = lo, 4..7 = hi), 0x0c = constant 0
IPX_DEV uint32_t pk_perm(uint32_t hi, uint32_t lo, uint32_t sel)
{
    const uint64_t t = ((uint64_t)hi << 32) | lo;
    uint32_t r = 0;
    for (int i = 0; i < 4; ++i) {
        const uint32_t k = (sel >> (8 * i)) & 0xFFu;
        uint32_t v;
        if (k < 8u) v = (uint32_t)(t >> (8 * k)) & 0xFFu;
        else if (k == 0x0cu) v = 0u;
        else { fprintf(stderr, "emu: pk_perm selector %#x not modelled\n", k); abort(); }
        r |= v << (8 * i);
    }
    return r;
}
// per half: the signed high byte, sign-extended to 16 bit (v_pk_ashrrev_i16 by 8)
IPX_DEV pk16 pk_sext_hi8(pk16 x) { return pk_make(pk_lo(x) >> 8, pk_hi(x) >> 8); }
IPX_DEV int8_t load_stream_i8(const int8_t *p) { return *p; }
IPX_DEV uint32_t load_global_u32(const uint32_t *p) { return *p; }
IPX_DEV void store_global_u32(uint32_t *p, uint32_t v) { *p = v; }
IPX_DEV uint32_t atomic_add_u32(uint32_t *p, uint32_t v) { uint32_t o = *p; *p = o + v; return o; }
IPX_DEV uint32_t atomic_or_u32(uint32_t *p, uint32_t v) { uint32_t o = *p; *p = o | v; return o; }

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
#endif

// ---- helpers shared by both builds -------------------------------------------------------------
// per-half "non-zero -> 0xFFFF" mask
IPX_DEV pk16 pk_nzmask(pk16 x) { return pk_sub(0u, pk_minu(x, 0x00010001u)); }
#if defined(IPX_CPU_EMU)
IPX_DEV pk16 pk_select(pk16 mask, pk16 a, pk16 b) { return (a & mask) | (b & ~mask); }
#else
// one v_bfi_b32 (left to itself the compiler sometimes splits it into v_and + v_and_or)
IPX_DEV pk16 pk_select(pk16 mask, pk16 a, pk16 b) { pk16 r; asm("v_bfi_b32 %0, %1, %2, %3" : "=v"(r) : "v"(mask), "v"(a), "v"(b)); return r; }
#endif
IPX_DEV pk16 pk_splat(int v) { return pk_make(v, v); }
