// LDS read cost of the sweep's B-fragment access pattern on gfx950:
//   lane (n = l & 15, g = l >> 4) needs the 16 bytes at byte offset x0 + n + 16 g of a window row.
// mode 0: 5 aligned dwords + 4 v_alignbyte_b32;  mode 1: one unaligned ds_read_b128;
// mode 2: aligned ds_read_b128 at 16 * lane (conflict-free reference, like the A fragment).
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef int v4i __attribute__((ext_vector_type(4)));
struct __attribute__((packed, aligned(1))) U16 { v4i v; };
template <int MODE>
__global__ __launch_bounds__(256) void k(int *sink, int mis, int iters)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    for (int i = threadIdx.x; i < 16384; i += 256) reinterpret_cast<int *>(smem)[i] = i * 2654435761u;
    __syncthreads();
    const int lane = threadIdx.x & 63, n = lane & 15, g = lane >> 4;
    const unsigned char *p = MODE == 2 ? smem + 16 * lane : smem + (n + 16 * g) + mis;
    v4i acc = {0, 0, 0, 0};
#pragma unroll 8
    for (int it = 0; it < iters; ++it) {
        const unsigned char *q = p + (it & 63) * (MODE == 2 ? 1024 : 116);
        if (MODE == 0) {
            const unsigned off = (unsigned)(q - smem);                         // stay in the LDS address space
            const unsigned *a = reinterpret_cast<const unsigned *>(smem + (off & ~3u));
            const unsigned sh = off & 3u;
            unsigned r0 = a[0], r1 = a[1], r2 = a[2], r3 = a[3], r4 = a[4];
            acc.x += __builtin_amdgcn_alignbyte(r1, r0, sh); acc.y += __builtin_amdgcn_alignbyte(r2, r1, sh);
            acc.z += __builtin_amdgcn_alignbyte(r3, r2, sh); acc.w += __builtin_amdgcn_alignbyte(r4, r3, sh);
        } else if (MODE == 1) {
            acc += reinterpret_cast<const U16 *>(q)->v;
        } else {
            acc += *reinterpret_cast<const v4i *>(q);
        }
    }
    sink[blockIdx.x * 256 + threadIdx.x] = acc.x ^ acc.y ^ acc.z ^ acc.w;
}
int main() {
    int *s; hipMalloc(&s, 4 * 256 * 2048);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 20000, blocks = 2048;              // 2 blocks of 64 KB per CU -> 8 wavefronts per CU
    const char *names[3] = {"5 dwords + alignbyte", "unaligned ds_read_b128", "aligned ds_read_b128 (16 B x lane)"};
    for (int mode = 0; mode < 3; ++mode)
        for (int mis = 0; mis < (mode == 2 ? 1 : 4); mis += 3) {
            float ms = 0;
            for (int rep = 0; rep < 2; ++rep) {
                hipEventRecord(e0);
                if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(256), 65536, 0, s, mis, iters);
                else if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(256), 65536, 0, s, mis, iters);
                else hipLaunchKernelGGL(k<2>, dim3(blocks), dim3(256), 65536, 0, s, mis, iters);
                hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
            }
            const double wave_reads = (double)blocks * 4 * iters;           // 16 B x 64 lanes each
            const double per_cu_per_s = wave_reads / 256.0 / (ms * 1e-3);
            printf("%-36s mis %d: %7.2f ms, %6.1f ns of CU time per wavefront fragment (= %5.1f clk @2.4GHz), %6.1f GB/s per CU\n",
                   names[mode], mis, ms, 1e9 / per_cu_per_s, 2.4e9 / per_cu_per_s, per_cu_per_s * 1024 / 1e9);
        }
    return 0;
}
