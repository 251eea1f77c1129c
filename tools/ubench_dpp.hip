#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#define ITER 2000
template <int KIND>
__global__ __launch_bounds__(64) void k(uint32_t *out, uint64_t *rec)
{
    uint32_t a = threadIdx.x, b = threadIdx.x * 3 + 1;
    const uint64_t t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < ITER; ++i) {
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            uint32_t y;
            if (KIND == 0) y = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)a, 0x111, 0xf, 0xf, true);      // row_shr:1
            else if (KIND == 1) y = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)a, 0x138, 0xf, 0xf, true); // wave_shr:1
            else if (KIND == 2) y = (uint32_t)__builtin_amdgcn_ds_swizzle((int)a, 0x8000 | 0); // placeholder
            else y = (uint32_t)__shfl_up((int)a, 1, 64);
            a = y + b;   // dependent
        }
    }
    const uint64_t t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * 64 + threadIdx.x] = a;
    if (threadIdx.x == 0) { rec[2 * blockIdx.x] = t0; rec[2 * blockIdx.x + 1] = t1; }
}
__global__ void sem(uint32_t *out)
{
    uint32_t a = threadIdx.x + 100;
    out[threadIdx.x] = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)a, 0x138, 0xf, 0xf, true);
    out[64 + threadIdx.x] = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)a, 0x111, 0xf, 0xf, true);
    out[128 + threadIdx.x] = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)a, 0x130, 0xf, 0xf, true);
    out[192 + threadIdx.x] = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)a, 0x101, 0xf, 0xf, true);
}
int main()
{
    uint32_t *out; uint64_t *rec;
    (void)hipMalloc(&out, 1024 * 64 * 4);
    (void)hipMalloc(&rec, 1024 * 16);
    uint32_t h[256];
    hipLaunchKernelGGL(sem, dim3(1), dim3(64), 0, 0, out);
    (void)hipMemcpy(h, out, 1024, hipMemcpyDeviceToHost);
    printf("wave_shr:1 :"); for (int i = 0; i < 64; ++i) printf(" %u", h[i]); printf("\n");
    printf("row_shr:1  :"); for (int i = 0; i < 64; ++i) printf(" %u", h[64 + i]); printf("\n");
    printf("wave_shl:1 :"); for (int i = 0; i < 64; ++i) printf(" %u", h[128 + i]); printf("\n");
    printf("row_shl:1  :"); for (int i = 0; i < 64; ++i) printf(" %u", h[192 + i]); printf("\n");
    for (int kind = 0; kind < 4; ++kind) {
        if (kind == 2) continue;
        uint64_t r[2];
        for (int rep = 0; rep < 2; ++rep) {
            if (kind == 0) hipLaunchKernelGGL(k<0>, dim3(1), dim3(64), 0, 0, out, rec);
            if (kind == 1) hipLaunchKernelGGL(k<1>, dim3(1), dim3(64), 0, 0, out, rec);
            if (kind == 3) hipLaunchKernelGGL(k<3>, dim3(1), dim3(64), 0, 0, out, rec);
            (void)hipDeviceSynchronize();
        }
        (void)hipMemcpy(r, rec, 16, hipMemcpyDeviceToHost);
        printf("kind %d: %.2f memtime ticks per (dpp + add) pair\n", kind, (double)(r[1] - r[0]) / (ITER * 16));
    }
    return 0;
}
