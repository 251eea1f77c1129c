// ubench_valu2.hip -- which gfx950 VALU instructions issue at the fast (2 cycles per wave-instruction per SIMD) rate?
// Same harness as ubench_valu.hip: 8 independent accumulators, W waves per SIMD, s_memtime.
// Build: hipcc --offload-arch=gfx950 -O2 -o tools/ubench_valu2 tools/ubench_valu2.hip
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <vector>
#include <algorithm>

#define ITER 3000

#define DEFK(NAME, BODY)                                                                          \
    __global__ __launch_bounds__(64) void NAME(uint32_t *out, uint64_t *cyc)                      \
    {                                                                                              \
        uint32_t a0 = threadIdx.x & 7, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, \
                 a7 = a0 + 7, b = 0x3c003c00u, c = 0x00000000u;                                    \
        const uint64_t t0 = __builtin_amdgcn_s_memtime();                                          \
        const uint64_t r0 = __builtin_amdgcn_s_memrealtime();                                      \
        for (int i = 0; i < ITER; ++i) {                                                           \
            asm volatile(BODY BODY BODY BODY BODY BODY BODY BODY                                   \
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) \
                         : "v"(b), "v"(c));                                                        \
        }                                                                                          \
        const uint64_t t1 = __builtin_amdgcn_s_memtime();                                          \
        const uint64_t r1 = __builtin_amdgcn_s_memrealtime();                                      \
        out[blockIdx.x * 64 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;                \
        if (threadIdx.x == 0) { cyc[2 * blockIdx.x] = t1 - t0; cyc[2 * blockIdx.x + 1] = r1 - r0; } \
    }

#define I2(OP, SUF) OP " %0, %0, %8" SUF "\n" OP " %1, %1, %8" SUF "\n" OP " %2, %2, %8" SUF "\n" OP " %3, %3, %8" SUF "\n" OP " %4, %4, %8" SUF "\n" OP " %5, %5, %8" SUF "\n" OP " %6, %6, %8" SUF "\n" OP " %7, %7, %8" SUF "\n"
#define I3(OP) OP " %0, %0, %8, %9\n" OP " %1, %1, %8, %9\n" OP " %2, %2, %8, %9\n" OP " %3, %3, %8, %9\n" OP " %4, %4, %8, %9\n" OP " %5, %5, %8, %9\n" OP " %6, %6, %8, %9\n" OP " %7, %7, %8, %9\n"
// interleave two different ops
#define MIX(A, B) A " %0, %0, %8\n" B " %1, %1, %8\n" A " %2, %2, %8\n" B " %3, %3, %8\n" A " %4, %4, %8\n" B " %5, %5, %8\n" A " %6, %6, %8\n" B " %7, %7, %8\n"

DEFK(k00, I2("v_pk_add_f16", ""))
DEFK(k01, I2("v_pk_max_f16", ""))
DEFK(k02, I3("v_pk_maximum3_f16"))
DEFK(k03, I3("v_pk_fma_f16"))
DEFK(k04, I2("v_pk_add_u16", ""))
DEFK(k05, I2("v_pk_max_u16", ""))
DEFK(k06, I2("v_pk_min_i16", ""))
DEFK(k07, I2("v_add_u16", ""))
DEFK(k08, I2("v_max_i16", ""))
DEFK(k09, I2("v_max_u16", ""))
DEFK(k10, I2("v_sub_u16", ""))
DEFK(k11, I2("v_add_f32", ""))
DEFK(k12, I2("v_max_f32", ""))
DEFK(k13, I2("v_add_f16", ""))
DEFK(k14, I2("v_max_f16", ""))
DEFK(k15, I2("v_max_u32", ""))
DEFK(k16, I2("v_min_u32", ""))
DEFK(k17, I2("v_sub_u32", ""))
DEFK(k18, I2("v_and_b32", ""))
DEFK(k19, I2("v_xor_b32", ""))
DEFK(k20, I2("v_lshlrev_b32", ""))
DEFK(k21, I3("v_add3_u32"))
DEFK(k22, I3("v_and_or_b32"))
DEFK(k23, I3("v_lshl_add_u32"))
DEFK(k24, I3("v_fma_f32"))
DEFK(k25, I3("v_maximum3_f32"))
DEFK(k26, I3("v_max3_f32"))
DEFK(k27, I2("v_mul_f32", ""))
DEFK(k28, MIX("v_pk_max_i16", "v_add_u32"))
DEFK(k29, MIX("v_pk_max_i16", "v_pk_add_f16"))
DEFK(k30, I2("v_add_u32_e64", ""))
DEFK(k31, I2("v_max_i32_e64", ""))
DEFK(k32, I2("v_pk_mul_f16", ""))
DEFK(k33, I2("v_pk_min_f16", ""))
DEFK(k34, I2("v_add_u16_sdwa", " dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:WORD_1 src1_sel:WORD_1"))
DEFK(k35, I2("v_max_i16_sdwa", " dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:WORD_1 src1_sel:WORD_1"))
DEFK(k36, I2("v_max_f32_e64", ""))
DEFK(k37, I2("v_pk_add_i16", " clamp"))

typedef void (*kern_t)(uint32_t *, uint64_t *);

int main()
{
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, 0) != hipSuccess) { printf("no device\n"); return 1; }
    const int ncu = prop.multiProcessorCount;
    printf("device %s, %d CUs\n", prop.name, ncu);
    struct { const char *name; kern_t k; } ks[] = {
        {"v_pk_add_f16", k00}, {"v_pk_max_f16", k01}, {"v_pk_maximum3_f16", k02}, {"v_pk_fma_f16", k03}, {"v_pk_add_u16", k04},
        {"v_pk_max_u16", k05}, {"v_pk_min_i16", k06}, {"v_add_u16", k07}, {"v_max_i16", k08}, {"v_max_u16", k09}, {"v_sub_u16", k10},
        {"v_add_f32", k11}, {"v_max_f32", k12}, {"v_add_f16", k13}, {"v_max_f16", k14}, {"v_max_u32", k15}, {"v_min_u32", k16},
        {"v_sub_u32", k17}, {"v_and_b32", k18}, {"v_xor_b32", k19}, {"v_lshlrev_b32", k20}, {"v_add3_u32", k21}, {"v_and_or_b32", k22},
        {"v_lshl_add_u32", k23}, {"v_fma_f32", k24}, {"v_maximum3_f32", k25}, {"v_max3_f32", k26}, {"v_mul_f32", k27},
        {"mix pk_max_i16/add_u32", k28}, {"mix pk_max_i16/pk_add_f16", k29}, {"v_add_u32_e64", k30}, {"v_max_i32_e64", k31},
        {"v_pk_mul_f16", k32}, {"v_pk_min_f16", k33}, {"v_add_u16_sdwa", k34}, {"v_max_i16_sdwa", k35}, {"v_max_f32_e64", k36},
        {"v_pk_add_i16 clamp", k37}};
    const int maxblocks = ncu * 4 * 8;
    uint32_t *out; uint64_t *cyc;
    hipMalloc(&out, (size_t)maxblocks * 64 * 4);
    hipMalloc(&cyc, (size_t)maxblocks * 16);
    std::vector<uint64_t> h(2 * (size_t)maxblocks);
    printf("%-28s %6s | cycles per wave-instruction per SIMD, clock GHz\n", "instruction", "waves");
    for (auto &k : ks)
        for (int w : {1, 2, 3, 4}) {
            const int blocks = ncu * 4 * w;
            for (int rep = 0; rep < 2; ++rep) {
                hipLaunchKernelGGL(k.k, dim3(blocks), dim3(64), 0, 0, out, cyc);
                hipDeviceSynchronize();
            }
            hipMemcpy(h.data(), cyc, (size_t)blocks * 16, hipMemcpyDeviceToHost);
            std::vector<double> c, ghz;
            for (int i = 0; i < blocks; ++i) { c.push_back((double)h[2 * i]); ghz.push_back((double)h[2 * i] / ((double)h[2 * i + 1] * 10.0)); }
            std::sort(c.begin(), c.end()); std::sort(ghz.begin(), ghz.end());
            const double med = c[c.size() / 2];
            const double n_inst = (double)ITER * 64.0;
            printf("%-28s %6d | %.2f  (per wave %.2f)  %.2f GHz\n", k.name, w, med / (n_inst * w), med / n_inst, ghz[ghz.size() / 2]);
        }
    return 0;
}
