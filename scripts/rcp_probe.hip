// Properties of v_rcp_f32 on gfx950 that the EASU / RCAS kernels and the oracle's device-reciprocal model rely on.
// The reference's OpenCL kernels compile `native_recip(x)` and `1.0f / x` to
//     v_frexp_mant_f32, v_rcp_f32, v_frexp_exp_i32_f32, v_sub/ldexp        (oracle/_ref/fsr_*.hsaco)
// This program checks over ALL 2^32 inputs: (1) where the plain instruction equals that sequence, (2) sign symmetry,
// (3) exponent invariance of the mantissa result, (4) the distance to the correctly rounded reciprocal.
// hipcc --offload-arch=gfx950 -O3 -ffp-contract=off scripts/rcp_probe.hip -o scripts/rcp_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cmath>

__device__ __forceinline__ float rcp_hw(float x) { return __builtin_amdgcn_rcpf(x); }
__device__ __forceinline__ float rcp_seq(float x)
{
    const float m = __builtin_amdgcn_frexp_mantf(x);
    const int e = __builtin_amdgcn_frexp_expf(x);
    return __builtin_amdgcn_ldexpf(__builtin_amdgcn_rcpf(m), -e);
}

// counters: 0 plain != seq (normal in, normal out), 1 plain != seq (other), 2 sign asym, 3 exponent variance, 4 delta -1, 5 delta 0, 6 delta +1, 7 |delta| > 1
__global__ void probe(unsigned long long* cnt, uint32_t* first_bad)
{
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    unsigned long long c[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < (1ull << 32); i += stride)
    {
        const uint32_t bits = (uint32_t)i;
        const float x = __uint_as_float(bits);
        const uint32_t ex = (bits >> 23) & 0xffu;
        if (ex == 0xffu) continue;                       // inf / nan
        const float a = rcp_hw(x), b = rcp_seq(x);
        const uint32_t ab = __float_as_uint(a), bb = __float_as_uint(b);
        const uint32_t rex = (bb >> 23) & 0xffu;
        const bool normal = ex != 0 && rex != 0 && rex != 0xffu;
        if (ab != bb) { if (normal) { c[0]++; atomicMin(first_bad, bits); } else c[1]++; }
        if (__float_as_uint(rcp_hw(-x)) != (ab ^ 0x80000000u)) c[2]++;
        if (normal && !(bits >> 31))
        {
            // mantissa of rcp(1.m * 2^k) must not depend on k: compare with k = 0
            const float x0 = __uint_as_float((bits & 0x7fffffu) | (127u << 23));
            const uint32_t r0 = __float_as_uint(rcp_hw(x0));
            if ((r0 & 0x7fffffu) != (ab & 0x7fffffu)) c[3]++;
            if (ex == 127u)
            {
                // correctly rounded reciprocal via double (exact to well below half an ulp of binary32 except for ties, which cannot occur for 1/x)
                const float cr = (float)(1.0 / (double)x);
                const int d = (int)(ab - __float_as_uint(cr));
                if (d == -1) c[4]++; else if (d == 0) c[5]++; else if (d == 1) c[6]++; else c[7]++;
            }
        }
    }
    for (int k = 0; k < 8; k++) if (c[k]) atomicAdd(&cnt[k], c[k]);
}

int main()
{
    unsigned long long* d_cnt; uint32_t* d_bad;
    hipMalloc(&d_cnt, 8 * sizeof(unsigned long long)); hipMemset(d_cnt, 0, 8 * sizeof(unsigned long long));
    hipMalloc(&d_bad, 4); hipMemset(d_bad, 0xff, 4);
    hipLaunchKernelGGL(probe, dim3(256 * 8), dim3(256), 0, 0, d_cnt, d_bad);
    unsigned long long c[8]; uint32_t bad;
    hipMemcpy(c, d_cnt, sizeof(c), hipMemcpyDeviceToHost); hipMemcpy(&bad, d_bad, 4, hipMemcpyDeviceToHost);
    printf("v_rcp_f32 vs frexp/rcp/ldexp sequence: %llu differences with normal input and output (first bits 0x%08x), %llu with a denormal / overflowing side\n", c[0], bad, c[1]);
    printf("sign asymmetries: %llu   exponent-dependent mantissas: %llu\n", c[2], c[3]);
    printf("mantissas in [1, 2): delta to the correctly rounded reciprocal  -1: %llu   0: %llu   +1: %llu   |d| > 1: %llu\n", c[4], c[5], c[6], c[7]);
    return 0;
}
