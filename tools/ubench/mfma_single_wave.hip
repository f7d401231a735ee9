// Micro-benchmark: MFMA issue rate with ONE wavefront per SIMD (latency-bound regime of pm_kernel_mfma).
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef int v4i __attribute__((ext_vector_type(4)));
#define ITER 2048
template <int NACC, bool SAMEB>
__global__ __launch_bounds__(256) void k(int *out, int seed)
{
    v4i a[8], b[8], c[NACC];
    for (int i = 0; i < 8; ++i) { a[i] = v4i{seed + i, 2, 3, (int)threadIdx.x}; b[i] = v4i{5, i, 7, seed}; }
    for (int i = 0; i < NACC; ++i) c[i] = v4i{0, 0, 0, 0};
    for (int it = 0; it < ITER; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) c[i] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a[i & 7], b[SAMEB ? 0 : (i & 7)], c[i], 0, 0, 0);
    }
    int r = 0;
    for (int i = 0; i < NACC; ++i) r += c[i][0] + c[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}
template <typename F> static double timeit(F launch) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    launch(); hipDeviceSynchronize(); hipEventRecord(e0);
    for (int r = 0; r < 5; ++r) launch();
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms = 0; hipEventElapsedTime(&ms, e0, e1); return ms / 5.0 * 1e-3;
}
int main() {
    int *d; hipMalloc(&d, sizeof(int) * 256 * 256 * 8);
#define RUN(NACC, SAMEB, BLOCKS) { double t = timeit([&] { hipLaunchKernelGGL((k<NACC, SAMEB>), dim3(BLOCKS), dim3(256), 0, 0, d, 1); }); \
    printf("blocks/CU %d  acc %d  sameB %d : %.1f shader-cycles per MFMA per wave (@2.4 GHz)\n", BLOCKS / 256, NACC, (int)SAMEB, \
           t * 2.4e9 / ((double)ITER * NACC) / ((BLOCKS) / 256)); }
    RUN(8, true, 256) RUN(8, false, 256) RUN(4, true, 256) RUN(2, true, 256) RUN(1, true, 256)
    RUN(8, true, 512) RUN(8, true, 1024)
    return 0;
}
