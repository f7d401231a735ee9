// LDS cost of window-fragment builds for the sweep (gfx950).
//   mode 0: one fragment  = 5 ds_read_b32 + 4 v_alignbyte_b32 (lane (n,g): 16 bytes at x0 + n + 16 g)
//   mode 1: two fragments = 3 aligned ds_read_b64 + 8 v_alignbyte_b32: the 32 placements x0 .. x0+31 are split
//           into tile A (x offsets {0-3, 8-11, 16-19, 24-27}) and tile B (the same + 4); lane (n,g) of both
//           tiles reads the 24 bytes at the 8-aligned address x0 + 8 (n >> 2) + 16 g and shifts by (n & 3)
//   mode 2: column-pair layout WP[u][rho] (2 bytes per rho): lane n reads its own row u = x0 + n, 3 ds_read_b64
// Reports CU clocks per fragment with 8 wavefronts per CU.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef unsigned u32;
template <int MODE>
__global__ __launch_bounds__(256) void k(int *sink, int iters, int pitch)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    for (int i = threadIdx.x; i < 16384; i += 256) reinterpret_cast<int *>(smem)[i] = i * 2654435761u;
    __syncthreads();
    const int lane = threadIdx.x & 63, n = lane & 15, g = lane >> 4;
    u32 a0 = 0, a1 = 0, a2 = 0, a3 = 0, b0 = 0, b1 = 0, b2 = 0, b3 = 0;
    for (int it0 = 0; it0 < iters; it0 += 4) {
#pragma unroll
    for (int it = it0; it < it0 + 4; ++it) {
        const u32 row = (u32)(it & 63) * (u32)pitch;
        if (MODE == 0) {
            const u32 off = row + n + 16 * g + 1;
            const u32 *p = reinterpret_cast<const u32 *>(smem + (off & ~3u));
            const u32 sh = off & 3u;
            const u32 r0 = p[0], r1 = p[1], r2 = p[2], r3 = p[3], r4 = p[4];
            a0 += __builtin_amdgcn_alignbyte(r1, r0, sh); a1 += __builtin_amdgcn_alignbyte(r2, r1, sh);
            a2 += __builtin_amdgcn_alignbyte(r3, r2, sh); a3 += __builtin_amdgcn_alignbyte(r4, r3, sh);
        } else if (MODE == 1) {
            // three separately laundered addresses: the compiler would otherwise merge neighbours into ds_read2_b64
            // (two accesses per instruction, 8 LDS cycles: half the rate of plain ds_read_b64)
            u32 o0 = row + 8 * (n >> 2) + 16 * g, o1 = o0 + 8, o2 = o0 + 16;
            asm("" : "+v"(o1)); asm("" : "+v"(o2));
            const u32 sh = n & 3u;
            const uint2 q0 = *reinterpret_cast<const uint2 *>(smem + o0), q1 = *reinterpret_cast<const uint2 *>(smem + o1),
                        q2 = *reinterpret_cast<const uint2 *>(smem + o2);
            a0 += __builtin_amdgcn_alignbyte(q0.y, q0.x, sh); a1 += __builtin_amdgcn_alignbyte(q1.x, q0.y, sh);
            a2 += __builtin_amdgcn_alignbyte(q1.y, q1.x, sh); a3 += __builtin_amdgcn_alignbyte(q2.x, q1.y, sh);
            b0 += __builtin_amdgcn_alignbyte(q1.x, q0.y, sh); b1 += __builtin_amdgcn_alignbyte(q1.y, q1.x, sh);
            b2 += __builtin_amdgcn_alignbyte(q2.x, q1.y, sh); b3 += __builtin_amdgcn_alignbyte(q2.y, q2.x, sh);
        } else {
            u32 o0 = (u32)n * (u32)pitch + 16 * g + 8 * (u32)(it & 7), o1 = o0 + 8, o2 = o0 + 16;
            asm("" : "+v"(o1)); asm("" : "+v"(o2));
            const uint2 q0 = *reinterpret_cast<const uint2 *>(smem + o0), q1 = *reinterpret_cast<const uint2 *>(smem + o1),
                        q2 = *reinterpret_cast<const uint2 *>(smem + o2);
            a0 += q0.x; a1 += q0.y; a2 += q1.x; a3 += q1.y;
            b0 += __builtin_amdgcn_alignbyte(q0.y, q0.x, 2); b1 += __builtin_amdgcn_alignbyte(q1.x, q0.y, 2);
            b2 += __builtin_amdgcn_alignbyte(q1.y, q1.x, 2); b3 += __builtin_amdgcn_alignbyte(q2.x, q1.y, 2);
        }
    }
    }
    sink[blockIdx.x * 256 + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ b0 ^ b1 ^ b2 ^ b3;
}
int main() {
    int *s; hipMalloc(&s, 4 * 256 * 2048);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 20000, blocks = 2048;
    const char *names[3] = {"5 x b32 -> 1 fragment", "3 x b64 -> 2 fragments (interleaved pair)", "3 x b64, lane-private rows (column-pair layout)"};
    const int pitches[3][3] = {{116, 128, 144}, {120, 128, 144}, {168, 152, 280}};
    for (int mode = 0; mode < 3; ++mode)
        for (int pi = 0; pi < 3; ++pi) {
            float ms = 0;
            const int pitch = pitches[mode][pi];
            for (int rep = 0; rep < 2; ++rep) {
                hipEventRecord(e0);
                if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(256), 65536, 0, s, iters, pitch);
                else if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(256), 65536, 0, s, iters, pitch);
                else hipLaunchKernelGGL(k<2>, dim3(blocks), dim3(256), 65536, 0, s, iters, pitch);
                hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
            }
            const double frags = (double)blocks * 4 * iters * (mode == 0 ? 1 : 2);
            const double per_cu_per_s = frags / 256.0 / (ms * 1e-3);
            printf("%-50s pitch %3d: %7.2f ms, %5.1f clk of CU time per fragment @2.4GHz\n", names[mode], pitch, ms, 2.4e9 / per_cu_per_s);
        }
    return 0;
}
