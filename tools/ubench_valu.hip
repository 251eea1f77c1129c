// ubench_valu.hip -- issue rate of the packed 16-bit VALU instructions the DP kernels are made of, on gfx950.
// For each instruction: ITER x 64 instructions on 8 independent accumulators (or one dependent chain),
// W waves per SIMD; reports shader cycles (s_memtime) per wave-instruction per SIMD and the clock
// (s_memtime / s_memrealtime).  Build: hipcc --offload-arch=gfx950 -O2 -o tools/ubench_valu tools/ubench_valu.hip
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <vector>
#include <algorithm>

#define ITER 4000
#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)

#define DEFK(NAME, INDEP, CHAIN)                                                                   \
    __global__ __launch_bounds__(64) void NAME(uint32_t *out, uint64_t *cyc, int chain)           \
    {                                                                                              \
        uint32_t a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, \
                 a7 = a0 + 7, b = 0x00030001u, c = 0x0c010c00u;                                    \
        const uint64_t t0 = __builtin_amdgcn_s_memtime();                                          \
        const uint64_t r0 = __builtin_amdgcn_s_memrealtime();                                      \
        if (!chain) {                                                                              \
            for (int i = 0; i < ITER; ++i) {                                                       \
                asm volatile(INDEP INDEP INDEP INDEP INDEP INDEP INDEP INDEP                       \
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) \
                             : "v"(b), "v"(c));                                                    \
            }                                                                                      \
        } else {                                                                                   \
            for (int i = 0; i < ITER; ++i) {                                                       \
                asm volatile(CHAIN CHAIN CHAIN CHAIN CHAIN CHAIN CHAIN CHAIN                       \
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) \
                             : "v"(b), "v"(c));                                                    \
            }                                                                                      \
        }                                                                                          \
        const uint64_t t1 = __builtin_amdgcn_s_memtime();                                          \
        const uint64_t r1 = __builtin_amdgcn_s_memrealtime();                                      \
        out[blockIdx.x * 64 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;                \
        if (threadIdx.x == 0) { cyc[2 * blockIdx.x] = t1 - t0; cyc[2 * blockIdx.x + 1] = r1 - r0; } \
    }

// 8 independent instructions / 8 dependent instructions (s_nop 0 after each dependent packed op, as the compiler emits)
#define I8(OP) OP " %0, %0, %8\n" OP " %1, %1, %8\n" OP " %2, %2, %8\n" OP " %3, %3, %8\n" OP " %4, %4, %8\n" OP " %5, %5, %8\n" OP " %6, %6, %8\n" OP " %7, %7, %8\n"
#define I8C(OP) OP " %0, %0, %8 clamp\n" OP " %1, %1, %8 clamp\n" OP " %2, %2, %8 clamp\n" OP " %3, %3, %8 clamp\n" OP " %4, %4, %8 clamp\n" OP " %5, %5, %8 clamp\n" OP " %6, %6, %8 clamp\n" OP " %7, %7, %8 clamp\n"
#define C8(OP) OP " %0, %0, %8\ns_nop 0\n" OP " %0, %0, %8\ns_nop 0\n" OP " %0, %0, %8\ns_nop 0\n" OP " %0, %0, %8\ns_nop 0\n" OP " %0, %0, %8\ns_nop 0\n" OP " %0, %0, %8\ns_nop 0\n" OP " %0, %0, %8\ns_nop 0\n" OP " %0, %0, %8\ns_nop 0\n"
#define C8N(OP) OP " %0, %0, %8\n" OP " %0, %0, %8\n" OP " %0, %0, %8\n" OP " %0, %0, %8\n" OP " %0, %0, %8\n" OP " %0, %0, %8\n" OP " %0, %0, %8\n" OP " %0, %0, %8\n"
#define P8 "v_perm_b32 %0, %0, %8, %9\nv_perm_b32 %1, %1, %8, %9\nv_perm_b32 %2, %2, %8, %9\nv_perm_b32 %3, %3, %8, %9\nv_perm_b32 %4, %4, %8, %9\nv_perm_b32 %5, %5, %8, %9\nv_perm_b32 %6, %6, %8, %9\nv_perm_b32 %7, %7, %8, %9\n"
#define M3 "v_max3_i32 %0, %0, %8, %9\nv_max3_i32 %1, %1, %8, %9\nv_max3_i32 %2, %2, %8, %9\nv_max3_i32 %3, %3, %8, %9\nv_max3_i32 %4, %4, %8, %9\nv_max3_i32 %5, %5, %8, %9\nv_max3_i32 %6, %6, %8, %9\nv_max3_i32 %7, %7, %8, %9\n"
#define DPP8 "v_mov_b32_dpp %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\nv_mov_b32_dpp %1, %2 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\nv_mov_b32_dpp %2, %3 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\nv_mov_b32_dpp %3, %4 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\nv_mov_b32_dpp %4, %5 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\nv_mov_b32_dpp %5, %6 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\nv_mov_b32_dpp %6, %7 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\nv_mov_b32_dpp %7, %0 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"

DEFK(k_pk_max_i16, I8("v_pk_max_i16"), C8("v_pk_max_i16"))
DEFK(k_pk_add_i16c, I8C("v_pk_add_i16"), C8N("v_pk_add_i16"))
DEFK(k_pk_sub_u16c, I8C("v_pk_sub_u16"), C8N("v_pk_sub_u16"))
DEFK(k_max_i32, I8("v_max_i32"), C8N("v_max_i32"))
DEFK(k_add_u32, I8("v_add_u32"), C8N("v_add_u32"))
DEFK(k_perm, P8, P8)
DEFK(k_max3_i32, M3, M3)
DEFK(k_pk_ashr, I8("v_pk_ashrrev_i16"), C8N("v_pk_ashrrev_i16"))
DEFK(k_dpp_mov, DPP8, DPP8)
DEFK(k_bfi, "v_bfi_b32 %0, %8, %0, %9\nv_bfi_b32 %1, %8, %1, %9\nv_bfi_b32 %2, %8, %2, %9\nv_bfi_b32 %3, %8, %3, %9\nv_bfi_b32 %4, %8, %4, %9\nv_bfi_b32 %5, %8, %5, %9\nv_bfi_b32 %6, %8, %6, %9\nv_bfi_b32 %7, %8, %7, %9\n", C8N("v_and_b32"))

typedef void (*kern_t)(uint32_t *, uint64_t *, int);

int main()
{
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, 0) != hipSuccess) { printf("no device\n"); return 1; }
    const int ncu = prop.multiProcessorCount;
    printf("device %s, %d CUs\n", prop.name, ncu);
    struct { const char *name; kern_t k; } ks[] = {
        {"v_pk_max_i16", k_pk_max_i16}, {"v_pk_add_i16 clamp", k_pk_add_i16c}, {"v_pk_sub_u16 clamp", k_pk_sub_u16c},
        {"v_pk_ashrrev_i16", k_pk_ashr}, {"v_perm_b32", k_perm}, {"v_bfi_b32", k_bfi}, {"v_mov_b32_dpp row_shr", k_dpp_mov},
        {"v_max_i32", k_max_i32}, {"v_add_u32", k_add_u32}, {"v_max3_i32", k_max3_i32}};
    const int maxblocks = ncu * 4 * 8;
    uint32_t *out; uint64_t *cyc;
    hipMalloc(&out, (size_t)maxblocks * 64 * 4);
    hipMalloc(&cyc, (size_t)maxblocks * 16);
    std::vector<uint64_t> h(2 * (size_t)maxblocks);
    printf("%-24s %5s %6s | cycles per wave-instruction per SIMD, clock GHz\n", "instruction", "chain", "waves");
    for (auto &k : ks)
        for (int chain = 0; chain < 2; ++chain)
            for (int w : {1, 2, 4}) {
                const int blocks = ncu * 4 * w;
                for (int rep = 0; rep < 2; ++rep) {
                    hipLaunchKernelGGL(k.k, dim3(blocks), dim3(64), 0, 0, out, cyc, chain);
                    hipDeviceSynchronize();
                }
                hipMemcpy(h.data(), cyc, (size_t)blocks * 16, hipMemcpyDeviceToHost);
                std::vector<double> c, ghz;
                for (int i = 0; i < blocks; ++i) { c.push_back((double)h[2 * i]); ghz.push_back((double)h[2 * i] / ((double)h[2 * i + 1] * 10.0)); }
                std::sort(c.begin(), c.end()); std::sort(ghz.begin(), ghz.end());
                const double med = c[c.size() / 2];
                const double n_inst = (double)ITER * 64.0;            // per wave
                printf("%-24s %5d %6d | %.2f  (per wave %.2f)  %.2f GHz\n", k.name, chain, w, med / (n_inst * w), med / n_inst, ghz[ghz.size() / 2]);
            }
    return 0;
}
