// Micro-benchmark: issue rate of candidate inner-product instructions on gfx950.
// Build: hipcc --offload-arch=gfx950 -O3 valu_rates.hip -o valu_rates ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

typedef _Float16 half2_t __attribute__((ext_vector_type(2)));
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

#define ITER 4096
#define NACC 16

template <int OP>
__global__ __launch_bounds__(256) void k(uint32_t *out, uint32_t seed)
{
    uint32_t a[NACC];
    float f[NACC];
    uint32_t x = threadIdx.x * 2654435761u + seed, y = x ^ 0x9e3779b9u;
    for (int i = 0; i < NACC; ++i) { a[i] = i; f[i] = (float)i; }
    for (int it = 0; it < ITER; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) {
            if (OP == 0) a[i] = __builtin_amdgcn_udot4(x, y + i, a[i], false);
            if (OP == 1) a[i] = (uint32_t)__builtin_amdgcn_sdot4((int)x, (int)(y + i), (int)a[i], false);
            if (OP == 2) f[i] = __builtin_fmaf(__uint_as_float(x), __uint_as_float(y + i), f[i]);
            if (OP == 3) a[i] = __builtin_amdgcn_alignbyte(x, a[i], 1) + i;
            if (OP == 4) {
                half2_t hx = __builtin_bit_cast(half2_t, x), hy = __builtin_bit_cast(half2_t, y + i);
                f[i] = __builtin_amdgcn_fdot2(hx, hy, f[i], false);
            }
            if (OP == 5) a[i] = __umul24(x, y + i) + a[i];          // v_mad_u32_u24
            if (OP == 6) a[i] = x * (y + i) + a[i];                  // v_mul_lo_u32 / mad_u64_u32
            if (OP == 7) a[i] = __builtin_amdgcn_sad_u8(x, y + i, a[i]);
            if (OP == 8) a[i] = __builtin_amdgcn_perm(x, a[i], y + i);
        }
        x += 0x01010101u;
    }
    uint32_t r = 0;
    for (int i = 0; i < NACC; ++i) r += a[i] + __float_as_uint(f[i]);
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

template <int OP>
__global__ __launch_bounds__(256) void kmfma(int *out, int seed)
{
    // OP 0: 16x16x64 i8 (4 accumulators), OP 1: 32x32x32 i8 (2 accumulators)
    v4i a = {seed + (int)threadIdx.x, 2, 3, 4}, b = {5, 6, 7, seed};
    if (OP == 0) {
        v4i c[4] = {{0,0,0,0},{0,0,0,0},{0,0,0,0},{0,0,0,0}};
        for (int it = 0; it < ITER; ++it) {
#pragma unroll
            for (int i = 0; i < 4; ++i) c[i] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, c[i], 0, 0, 0);
        }
        out[blockIdx.x * blockDim.x + threadIdx.x] = c[0][0] + c[1][1] + c[2][2] + c[3][3];
    } else {
        v16i c[2];
        for (int i = 0; i < 16; ++i) { c[0][i] = 0; c[1][i] = 0; }
        for (int it = 0; it < ITER; ++it) {
#pragma unroll
            for (int i = 0; i < 2; ++i) c[i] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, c[i], 0, 0, 0);
        }
        out[blockIdx.x * blockDim.x + threadIdx.x] = c[0][0] + c[1][5];
    }
}

template <typename F>
static double timeit(F launch)
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    launch();
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int r = 0; r < 5; ++r) launch();
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    return ms / 5.0 * 1e-3;
}

int main()
{
    uint32_t *d;
    const int blocks = 256 * 8, threads = 256;
    hipMalloc(&d, sizeof(uint32_t) * blocks * threads);
    const char *names[] = {"v_dot4_u32_u8", "v_dot4_i32_i8", "v_fma_f32", "v_alignbyte+add", "v_dot2_f32_f16",
                           "v_mad_u32_u24", "v_mul_lo_u32+add", "v_sad_u8", "v_perm_b32"};
    const double lane_ops = (double)blocks * threads * ITER * NACC;
#define RUN(OP) { double t = timeit([&] { hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(threads), 0, 0, d, 1u); }); \
                  printf("%-18s %8.3f ms  %8.2f T lane-instr/s  (%.2f cycles/wave-instr/SIMD @2.4GHz)\n", names[OP], t * 1e3, \
                         lane_ops / t / 1e12, 2.4e9 * 1024 * t / (lane_ops / 64)); }
    RUN(0) RUN(1) RUN(2) RUN(3) RUN(4) RUN(5) RUN(6) RUN(7) RUN(8)
    {
        double t = timeit([&] { hipLaunchKernelGGL(kmfma<0>, dim3(blocks), dim3(threads), 0, 0, (int *)d, 1); });
        double ops = (double)blocks * (threads / 64) * ITER * 4 * 2.0 * 16 * 16 * 64;
        printf("mfma_i32_16x16x64_i8 %8.3f ms  %8.1f TOPS  (%.1f cycles/mfma/SIMD)\n", t * 1e3, ops / t / 1e12,
               2.4e9 * 1024 * t / ((double)blocks * 4 * ITER * 4));
        t = timeit([&] { hipLaunchKernelGGL(kmfma<1>, dim3(blocks), dim3(threads), 0, 0, (int *)d, 1); });
        ops = (double)blocks * (threads / 64) * ITER * 2 * 2.0 * 32 * 32 * 32;
        printf("mfma_i32_32x32x32_i8 %8.3f ms  %8.1f TOPS  (%.1f cycles/mfma/SIMD)\n", t * 1e3, ops / t / 1e12,
               2.4e9 * 1024 * t / ((double)blocks * 4 * ITER * 2));
    }
    return 0;
}
