// ubench_dispatch.hip -- where do the one-wave blocks of a SMALL grid land?  For grids of 64..2048 one-wave blocks of a fixed run of packed
// instructions: launch time (HIP events) and, per block, HW_ID (SIMD / CU / SE) and XCC_ID -- how many blocks share a SIMD and a CU.
// Build: hipcc --offload-arch=gfx950 -O2 -o tools/ubench_dispatch tools/ubench_dispatch.hip
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <map>
#include <vector>
#define ITER 1500
#define I8(OP) OP " %0, %0, %8\n" OP " %1, %1, %8\n" OP " %2, %2, %8\n" OP " %3, %3, %8\n" OP " %4, %4, %8\n" OP " %5, %5, %8\n" OP " %6, %6, %8\n" OP " %7, %7, %8\n"
__global__ __launch_bounds__(64) void k(uint32_t *out, uint64_t *rec)
{
    uint32_t a0 = threadIdx.x & 7, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7, b = 0x00030001u;
    uint32_t hwid, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    const uint64_t t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < ITER; ++i)
        asm volatile(I8("v_pk_max_i16") I8("v_pk_max_i16") I8("v_pk_max_i16") I8("v_pk_max_i16")
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));
    const uint64_t t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * 64 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
    if (threadIdx.x == 0) { rec[4 * blockIdx.x] = t0; rec[4 * blockIdx.x + 1] = t1; rec[4 * blockIdx.x + 2] = hwid; rec[4 * blockIdx.x + 3] = xcc; }
}
int main()
{
    const int maxblocks = 4096;
    uint32_t *out; uint64_t *rec;
    (void)hipMalloc(&out, (size_t)maxblocks * 64 * 4);
    (void)hipMalloc(&rec, (size_t)maxblocks * 32);
    std::vector<uint64_t> h(4 * (size_t)maxblocks);
    const int grids[] = {64, 128, 250, 256, 500, 512, 750, 1000, 1024, 2048};
    for (int blocks : grids) {
        hipEvent_t e0, e1;
        (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
        float ms = 0;
        for (int rep = 0; rep < 3; ++rep) {
            (void)hipEventRecord(e0, 0);
            hipLaunchKernelGGL(k, dim3(blocks), dim3(64), 0, 0, out, rec);
            (void)hipEventRecord(e1, 0);
            (void)hipDeviceSynchronize();
            (void)hipEventElapsedTime(&ms, e0, e1);
        }
        (void)hipMemcpy(h.data(), rec, (size_t)blocks * 32, hipMemcpyDeviceToHost);
        std::map<uint64_t, int> per_simd, per_cu;
        for (int i = 0; i < blocks; ++i) {
            const uint64_t hw = h[4 * i + 2], xcc = h[4 * i + 3] & 15;
            const uint64_t simd = (hw >> 4) & 3, cu = (hw >> 8) & 15, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;
            const uint64_t cukey = (xcc << 16) | (se << 8) | (sh << 7) | cu;
            ++per_cu[cukey]; ++per_simd[(cukey << 2) | simd];
        }
        int hist_cu[16] = {0}, hist_simd[16] = {0};
        for (auto &kv : per_cu) ++hist_cu[kv.second < 15 ? kv.second : 15];
        for (auto &kv : per_simd) ++hist_simd[kv.second < 15 ? kv.second : 15];
        printf("blocks %4d: %.4f ms | CUs used %zu, blocks per CU histogram:", blocks, ms, per_cu.size());
        for (int q = 1; q < 16; ++q) if (hist_cu[q]) printf(" %dx%d", hist_cu[q], q);
        printf(" | SIMDs used %zu, blocks per SIMD:", per_simd.size());
        for (int q = 1; q < 16; ++q) if (hist_simd[q]) printf(" %dx%d", hist_simd[q], q);
        printf("\n");
    }
    return 0;
}
