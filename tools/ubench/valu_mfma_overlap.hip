// Micro-benchmark: do VALU instructions of one wavefront overlap the MFMAs of ANOTHER wavefront on the same SIMD?
// A workgroup of 8 wavefronts puts two on every SIMD.  Mode 0: all 8 run an MFMA loop.  Mode 1: all 8 run a VALU loop.
// Mode 2: wavefronts 0-3 MFMA, 4-7 VALU (one of each per SIMD).  Mode 3: every wavefront alternates 1 MFMA : R VALU.
// If the two pipes overlap across wavefronts, mode 2 takes max(t_mfma_half, t_valu_half); if the SIMD runs one vector
// instruction at a time, it takes their sum.  Also times dependent-free v_cvt_f64 / v_fma_f64 / v_rsq_f32 / v_pk_fma_f32
// per instruction.   Build: hipcc --offload-arch=gfx950 -O3 valu_mfma_overlap.hip -o valu_mfma_overlap
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef int v4i __attribute__((ext_vector_type(4)));
typedef float v2f __attribute__((ext_vector_type(2)));
#define ITER 8192

__device__ __forceinline__ void mfma_loop(int *out, int seed, int iters)
{
    v4i a = {seed + (int)threadIdx.x, 2, 3, 4}, b = {5, 6, 7, seed};
    v4i c[8];
    for (int i = 0; i < 8; ++i) c[i] = v4i{0, 0, 0, 0};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) c[i] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, c[i], 0, 0, 0);
    }
    int r = 0;
    for (int i = 0; i < 8; ++i) r += c[i][0] + c[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}
template <int KIND>
__device__ __forceinline__ void valu_loop(int *out, int seed, int iters)
{
    // 32 independent chains, one instruction each per iteration
    float f[32]; double d[16]; v2f p[16];
    for (int i = 0; i < 32; ++i) f[i] = (float)(i + seed);
    for (int i = 0; i < 16; ++i) { d[i] = (double)(i + seed); p[i] = v2f{(float)i, (float)seed}; }
    const float x = __int_as_float(0x3f800001 + threadIdx.x);
    const double xd = 1.0000001 + threadIdx.x * 1e-9;
    for (int it = 0; it < iters; ++it) {
        if (KIND == 0) {
#pragma unroll
            for (int i = 0; i < 32; ++i) f[i] = __builtin_fmaf(f[i], x, 1.0f);
        } else if (KIND == 1) {
#pragma unroll
            for (int i = 0; i < 16; ++i) d[i] = __builtin_fma(d[i], xd, 1.0);
#pragma unroll
            for (int i = 0; i < 16; ++i) d[i] = __builtin_fma(d[i], xd, 2.0);
        } else if (KIND == 2) {
#pragma unroll
            for (int i = 0; i < 32; ++i) f[i] = __builtin_amdgcn_rsqf(f[i] + 1.0f);
        } else if (KIND == 3) {
#pragma unroll
            for (int i = 0; i < 16; ++i) p[i] = __builtin_elementwise_fma(p[i], v2f{x, x}, v2f{1.0f, 2.0f});
#pragma unroll
            for (int i = 0; i < 16; ++i) p[i] = __builtin_elementwise_fma(p[i], v2f{x, x}, v2f{3.0f, 4.0f});
        } else if (KIND == 4) {
#pragma unroll
            for (int i = 0; i < 16; ++i) d[i] = (double)(int)__float_as_int(f[i]) + d[i];   // v_cvt_f64_i32 + v_add_f64
#pragma unroll
            for (int i = 0; i < 16; ++i) f[i] = (float)d[i] + f[i + 16];                       // v_cvt_f32_f64 + v_add_f32
        } else if (KIND == 5) {
#pragma unroll
            for (int i = 0; i < 32; ++i) f[i] = __int_as_float(__builtin_amdgcn_alignbyte(__float_as_int(f[i]), __float_as_int(f[(i + 1) & 31]), 1));
        } else if (KIND == 6) {                                         // v_add_u32 with a run-time operand
#pragma unroll
            for (int i = 0; i < 32; ++i) f[i] = __int_as_float(__float_as_int(f[i]) + __float_as_int(x) + i);
        } else if (KIND == 7) {                                         // v_mad_i32_i24
#pragma unroll
            for (int i = 0; i < 32; ++i) f[i] = __int_as_float(__mul24(__float_as_int(f[i]), __float_as_int(x)) + i);
        } else if (KIND == 8) {                                         // v_cvt_f32_i32
#pragma unroll
            for (int i = 0; i < 32; ++i) f[i] = (float)(__float_as_int(f[i]) ^ i);
        } else if (KIND == 9) {                                         // v_max_f32 / v_cndmask
#pragma unroll
            for (int i = 0; i < 32; ++i) f[i] = fmaxf(f[i], x) > 2.0f ? x : f[i];
        } else if (KIND == 10) {                                        // v_mul_f32
#pragma unroll
            for (int i = 0; i < 32; ++i) f[i] = f[i] * x;
        }
    }
    float r = 0;
    for (int i = 0; i < 32; ++i) r += f[i];
    for (int i = 0; i < 16; ++i) r += (float)d[i] + p[i][0] + p[i][1];
    out[blockIdx.x * blockDim.x + threadIdx.x] = __float_as_int(r);
}

template <int MODE, int KIND>
__global__ __launch_bounds__(512) void k(int *out, int seed, int it_m, int it_v)
{
    const int wv = threadIdx.x >> 6;
    if (MODE == 0) mfma_loop(out, seed, it_m);
    else if (MODE == 1) valu_loop<KIND>(out, seed, it_v);
    else if (MODE == 2) { if (wv < 4) mfma_loop(out, seed, it_m); else valu_loop<KIND>(out, seed, it_v); }
}

template <typename F>
static double timeit(F launch)
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    launch(); hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int r = 0; r < 5; ++r) launch();
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    return ms / 5.0 * 1e-3;
}

template <int KIND>
static void run(const char *name, int *out, int blocks)
{
    // per SIMD: mode 0 = 2 waves x ITER x 8 MFMAs; mode 1 = 2 waves x ITER x 32 VALU; mode 2 = 1 wave of each
    const double tm = timeit([&] { hipLaunchKernelGGL((k<0, KIND>), dim3(blocks), dim3(512), 0, 0, out, 1, ITER, ITER); });
    const double tv = timeit([&] { hipLaunchKernelGGL((k<1, KIND>), dim3(blocks), dim3(512), 0, 0, out, 1, ITER, ITER); });
    const double tb = timeit([&] { hipLaunchKernelGGL((k<2, KIND>), dim3(blocks), dim3(512), 0, 0, out, 1, ITER, ITER); });
    const double clk = 2.4e9;
    const double cyc_m = tm * clk / (2.0 * ITER * 8), cyc_v = tv * clk / (2.0 * ITER * 32);
    printf("%-28s all-MFMA %.3f ms (%.1f clk/MFMA)  all-VALU %.3f ms (%.2f clk/instr)  half/half %.3f ms  [sum of halves %.3f, max of halves %.3f]\n",
           name, tm * 1e3, cyc_m, tv * 1e3, cyc_v, tb * 1e3, (tm + tv) * 0.5e3, (tm > tv ? tm : tv) * 0.5e3);
}

int main()
{
    int *out; hipMalloc(&out, 256 * 8 * 512 * sizeof(int));
    const int blocks = 256;              // one workgroup of 8 wavefronts per CU (2 per SIMD)
    run<0>("v_fma_f32", out, blocks);
    run<1>("v_fma_f64", out, blocks);
    run<2>("v_add_f32 + v_rsq_f32", out, blocks);
    run<3>("v_pk_fma_f32", out, blocks);
    run<4>("cvt f64<->i32/f32 + add", out, blocks);
    run<5>("v_alignbyte_b32", out, blocks);
    run<6>("v_add_u32 (x2)", out, blocks);
    run<7>("v_mad_i32_i24", out, blocks);
    run<8>("v_xor + v_cvt_f32_i32", out, blocks);
    run<9>("v_max + v_cmp + v_cndmask", out, blocks);
    run<10>("v_mul_f32", out, blocks);
    run<0>("v_fma_f32 (again)", out, blocks);
    return 0;
}
