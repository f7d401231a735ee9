// Probe: operand/result lane layout of v_mfma_i32_16x16x64_i8 on gfx950.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <stdlib.h>
typedef int v4i __attribute__((ext_vector_type(4)));

__global__ void k(const int8_t *A /*16x64 row-major*/, const int8_t *B /*64x16 (k-major): B[k*16+n]*/, int *D /*16x16*/)
{
    const int l = threadIdx.x, m = l & 15, g = l >> 4;
    v4i a, b, c = {0, 0, 0, 0};
    for (int q = 0; q < 4; ++q) {
        uint32_t wa = 0, wb = 0;
        for (int j = 0; j < 4; ++j) {
            const int kk = 16 * g + 4 * q + j;
            wa |= (uint32_t)(uint8_t)A[m * 64 + kk] << (8 * j);
            wb |= (uint32_t)(uint8_t)B[kk * 16 + m] << (8 * j);
        }
        a[q] = (int)wa; b[q] = (int)wb;
    }
    c = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, c, 0, 0, 0);
    for (int r = 0; r < 4; ++r) D[(4 * g + r) * 16 + m] = c[r];      // row = 4*(l>>4)+reg, col = l&15
}

int main()
{
    int8_t hA[16 * 64], hB[64 * 16];
    int hD[256], ref[256];
    srand(1);
    for (int i = 0; i < 1024; ++i) { hA[i] = (int8_t)(rand() % 256 - 128); hB[i] = (int8_t)(rand() % 256 - 128); }
    for (int m = 0; m < 16; ++m) for (int n = 0; n < 16; ++n) {
        int s = 0; for (int kk = 0; kk < 64; ++kk) s += (int)hA[m * 64 + kk] * (int)hB[kk * 16 + n];
        ref[m * 16 + n] = s;
    }
    int8_t *dA, *dB; int *dD;
    hipMalloc(&dA, 1024); hipMalloc(&dB, 1024); hipMalloc(&dD, 1024);
    hipMemcpy(dA, hA, 1024, hipMemcpyHostToDevice); hipMemcpy(dB, hB, 1024, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dA, dB, dD);
    hipMemcpy(hD, dD, 1024, hipMemcpyDeviceToHost);
    int bad = 0; for (int i = 0; i < 256; ++i) bad += hD[i] != ref[i];
    printf("mfma_i32_16x16x64_i8 layout probe: %d mismatches of 256 (D[0]=%d ref=%d)\n", bad, hD[0], ref[0]);
    return bad != 0;
}
