// which way do the whole-wave DPP shifts move data on gfx950?  hipcc --offload-arch=gfx950 -O3 scripts/exp/dpp_wave_shift.hip -o /tmp/dppt && /tmp/dppt
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(int *out)
{
    const int v = 100 + (int)threadIdx.x;
    out[threadIdx.x] = __builtin_amdgcn_update_dpp(-1, v, 0x130, 0xf, 0xf, false);        // wave_shl:1
    out[64 + threadIdx.x] = __builtin_amdgcn_update_dpp(-1, v, 0x138, 0xf, 0xf, false);   // wave_shr:1
}
int main()
{
    int *d, h[128];
    hipMalloc(&d, sizeof h);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
    printf("wave_shl:1 lane0 %d lane1 %d lane31 %d lane32 %d lane62 %d lane63 %d\n", h[0], h[1], h[31], h[32], h[62], h[63]);
    printf("wave_shr:1 lane0 %d lane1 %d lane31 %d lane32 %d lane62 %d lane63 %d\n", h[64], h[65], h[64 + 31], h[64 + 32], h[64 + 62], h[64 + 63]);
    return 0;
}
