// Stress test: DPP-based wavefront reductions vs LDS-crossbar (__shfl_xor) reductions on gfx950.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <math.h>
template <typename Op>
__device__ __forceinline__ int wave_reduce_i32(int v, int ident, Op op) {
    v = op(v, __builtin_amdgcn_update_dpp(ident, v, 0x111, 0xf, 0xf, false));
    v = op(v, __builtin_amdgcn_update_dpp(ident, v, 0x112, 0xf, 0xf, false));
    v = op(v, __builtin_amdgcn_update_dpp(ident, v, 0x114, 0xf, 0xf, false));
    v = op(v, __builtin_amdgcn_update_dpp(ident, v, 0x118, 0xf, 0xf, false));
    v = op(v, __builtin_amdgcn_update_dpp(ident, v, 0x142, 0xa, 0xf, false));
    v = op(v, __builtin_amdgcn_update_dpp(ident, v, 0x143, 0xc, 0xf, false));
    return __builtin_amdgcn_readlane(v, 63);
}
__device__ __forceinline__ float wave_max_dpp(float v) {
    const int r = wave_reduce_i32(__float_as_int(v), __float_as_int(-INFINITY),
                                  [](int a, int b) { return __float_as_int(fmaxf(__int_as_float(a), __int_as_float(b))); });
    return __int_as_float(r);
}
__device__ __forceinline__ int wave_sum_dpp(int v) { return wave_reduce_i32(v, 0, [](int a, int b) { return a + b; }); }

__global__ __launch_bounds__(256) void k(unsigned *bad, unsigned seed, int iters)
{
    unsigned x = (blockIdx.x * 256 + threadIdx.x) * 2654435761u + seed;
    unsigned nbad = 0;
    float acc = 0.f;
    for (int it = 0; it < iters; ++it) {
        x = x * 1664525u + 1013904223u;
        float f = (float)(x >> 8) * (1.0f / 16777216.0f) - 0.3f;
        // some VALU / trans work around, like the kernel's epilogue
        f = f * __builtin_amdgcn_rsqf(1.0f + (float)(it & 7)) + acc * 1e-9f;
        int iv = (int)(x >> 20) - 2048;
        const float m1 = wave_max_dpp(f);
        const int s1 = wave_sum_dpp(iv);
        float m2 = f; int s2 = iv;
        for (int o = 32; o > 0; o >>= 1) { m2 = fmaxf(m2, __shfl_xor(m2, o)); s2 += __shfl_xor(s2, o); }
        if (m1 != m2) nbad += 1;
        if (s1 != s2) nbad += 1000;
        acc += m1;
        if ((x >> 13) & 1) {                       // divergent region between reductions
            x ^= (unsigned)__float_as_int(acc);
            if ((x >> 17) & 1) atomicAdd(bad + 1, 1u);
        }
    }
    if (nbad) atomicAdd(bad, nbad);
}
int main() {
    unsigned *d; hipMalloc(&d, 8); hipMemset(d, 0, 8);
    for (int r = 0; r < 20; ++r) hipLaunchKernelGGL(k, dim3(256 * 12), dim3(256), 0, 0, d, 17u + r, 2000);
    hipDeviceSynchronize();
    unsigned h[2]; hipMemcpy(h, d, 8, hipMemcpyDeviceToHost);
    printf("DPP vs shuffle mismatches (max: +1 each, sum: +1000 each): %u\n", h[0]);
    return 0;
}
