// ubench_issue.hip -- how does a gfx950 SIMD share its VALU issue between resident waves?  Per wave: HW_ID (SIMD, wave slot),
// start and end time of a fixed run of independent v_pk_max_i16 (or v_add_u32); printed raw for offline analysis.
// Build: hipcc --offload-arch=gfx950 -O2 -o tools/ubench_issue tools/ubench_issue.hip
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <vector>

#define ITER 3000
#define I8(OP) OP " %0, %0, %8\n" OP " %1, %1, %8\n" OP " %2, %2, %8\n" OP " %3, %3, %8\n" OP " %4, %4, %8\n" OP " %5, %5, %8\n" OP " %6, %6, %8\n" OP " %7, %7, %8\n"

template <int KIND>
__global__ __launch_bounds__(64) void k(uint32_t *out, uint64_t *rec)
{
    uint32_t a0 = threadIdx.x & 7, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7, b = 0x00030001u;
    uint32_t hwid, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    const uint64_t t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < ITER; ++i) {
        if (KIND == 0)
            asm volatile(I8("v_pk_max_i16") I8("v_pk_max_i16") I8("v_pk_max_i16") I8("v_pk_max_i16")
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));
        else
            asm volatile(I8("v_add_u32") I8("v_add_u32") I8("v_add_u32") I8("v_add_u32")
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));
    }
    const uint64_t t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * 64 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
    if (threadIdx.x == 0) { rec[4 * blockIdx.x] = t0; rec[4 * blockIdx.x + 1] = t1; rec[4 * blockIdx.x + 2] = hwid; rec[4 * blockIdx.x + 3] = xcc; }
}

int main()
{
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, 0) != hipSuccess) { printf("no device\n"); return 1; }
    const int ncu = prop.multiProcessorCount;
    const int maxblocks = ncu * 4 * 10;
    uint32_t *out; uint64_t *rec;
    (void)hipMalloc(&out, (size_t)maxblocks * 64 * 4);
    (void)hipMalloc(&rec, (size_t)maxblocks * 32);
    std::vector<uint64_t> h(4 * (size_t)maxblocks);
    for (int kind = 0; kind < 2; ++kind)
        for (int w = 1; w <= 8; ++w) {
            const int blocks = ncu * 4 * w;
            hipEvent_t e0, e1;
            (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
            float ms = 0;
            for (int rep = 0; rep < 2; ++rep) {
                (void)hipEventRecord(e0, 0);
                if (kind == 0) hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(64), 0, 0, out, rec);
                else hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(64), 0, 0, out, rec);
                (void)hipEventRecord(e1, 0);
                (void)hipDeviceSynchronize();
                (void)hipEventElapsedTime(&ms, e0, e1);
            }
            (void)hipMemcpy(h.data(), rec, (size_t)blocks * 32, hipMemcpyDeviceToHost);
            printf("# kind %d waves %d blocks %d ms %.4f inst_per_wave %d\n", kind, w, blocks, ms, ITER * 32);
            for (int i = 0; i < blocks; ++i)
                printf("%d %d %d %llu %llu %llx %llx\n", kind, w, i, (unsigned long long)h[4 * i], (unsigned long long)h[4 * i + 1],
                       (unsigned long long)h[4 * i + 2], (unsigned long long)h[4 * i + 3]);
        }
    return 0;
}
