// v_perm_b32 selectors 8..11 on gfx950 (k_dp_wide's score lookup relies on them): which source bit does each replicate?
//   hipcc --offload-arch=gfx950 -O2 tools/ubench_perm.hip -o tools/ubench_perm && tools/ubench_perm
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void k(const uint32_t *in, uint32_t *out)
{
    const uint32_t hi = in[0], lo = in[1];
    out[0] = __builtin_amdgcn_perm(hi, lo, 0x08080801u);
    out[1] = __builtin_amdgcn_perm(hi, lo, 0x09090903u);
    out[2] = __builtin_amdgcn_perm(hi, lo, 0x0a0a0a05u);
    out[3] = __builtin_amdgcn_perm(hi, lo, 0x0b0b0b07u);
    out[4] = __builtin_amdgcn_perm(hi, lo, 0x0c0c0c0cu);
    out[5] = __builtin_amdgcn_perm(hi, lo, 0x0d0d0d0du);
}
int main()
{
    uint32_t *d, *o, h[2], r[6];
    hipMalloc(&d, 8); hipMalloc(&o, 24);
    const uint32_t cases[4][2] = {{0x80007f00u, 0x7f008000u}, {0x7f008000u, 0x80007f00u}, {0xfe000300u, 0x0300fe00u}, {0x00000000u, 0xffffffffu}};
    for (auto &c : cases) {
        h[0] = c[0]; h[1] = c[1];
        hipMemcpy(d, h, 8, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(k, dim3(1), dim3(1), 0, 0, d, o);
        hipMemcpy(r, o, 24, hipMemcpyDeviceToHost);
        printf("hi %08x lo %08x: sel 08080801 -> %08x | 09090903 -> %08x | 0a0a0a05 -> %08x | 0b0b0b07 -> %08x | 0c.. -> %08x | 0d.. -> %08x\n", h[0], h[1], r[0], r[1], r[2], r[3], r[4], r[5]);
    }
    return 0;
}
