// Semantics of the whole-wavefront DPP shifts on gfx950 (v_mov_b32_dpp wave_shl:1 / wave_shr:1): which lane reads which, and what the
// lane without a source keeps.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(int* o)
{
    const int v = o[threadIdx.x];
    const int r = __builtin_amdgcn_update_dpp(-1, v, 0x130, 0xf, 0xf, false);   // wave_shl:1
    const int q = __builtin_amdgcn_update_dpp(-1, v, 0x138, 0xf, 0xf, false);   // wave_shr:1
    o[threadIdx.x] = r; o[threadIdx.x + 64] = q;
}
int main()
{
    int* d; int h[128];
    if (hipMalloc(&d, 512) != hipSuccess) return 1;
    for (int i = 0; i < 64; i++) h[i] = i;
    (void)hipMemcpy(d, h, 256, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    (void)hipMemcpy(h, d, 512, hipMemcpyDeviceToHost);
    printf("wave_shl:1 "); for (int i = 0; i < 64; i++) printf("%d ", h[i]);
    printf("\nwave_shr:1 "); for (int i = 64; i < 128; i++) printf("%d ", h[i]);
    printf("\n");
    return 0;
}
