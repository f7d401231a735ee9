// Probe: do the CBSZ / ABID / BLGP modifiers act on v_mfma_i32_16x16x64_i8 (gfx950), and how?
// For every BLGP value the result is compared with the plain product of A and a lane-group-permuted B
// (lane group = 16 lanes = one k-group of 16 bytes): rotate / broadcast patterns of the CDNA3 ISA table.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <stdlib.h>
typedef int v4i __attribute__((ext_vector_type(4)));

template <int BLGP>
__global__ void k(const int8_t *A, const int8_t *B, int *D)
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
    c = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, c, 0, 0, BLGP);
    for (int r = 0; r < 4; ++r) D[(4 * g + r) * 16 + m] = c[r];
}

static int8_t hA[16 * 64], hB[64 * 16];
static void ref_with_groups(const int src[4], int *ref)
{
    // B'(k-group g) = B(k-group src[g])
    for (int m = 0; m < 16; ++m) for (int n = 0; n < 16; ++n) {
        int s = 0;
        for (int g = 0; g < 4; ++g) for (int j = 0; j < 16; ++j)
            s += (int)hA[m * 64 + 16 * g + j] * (int)hB[(16 * src[g] + j) * 16 + n];
        ref[m * 16 + n] = s;
    }
}

int main()
{
    srand(7);
    for (int i = 0; i < 1024; ++i) { hA[i] = (int8_t)(rand() % 256 - 128); hB[i] = (int8_t)(rand() % 256 - 128); }
    int8_t *dA, *dB; int *dD;
    hipMalloc(&dA, 1024); hipMalloc(&dB, 1024); hipMalloc(&dD, 1024);
    hipMemcpy(dA, hA, 1024, hipMemcpyHostToDevice); hipMemcpy(dB, hB, 1024, hipMemcpyHostToDevice);
    int hD[256], ref[256];
    for (int blgp = 0; blgp < 8; ++blgp) {
        switch (blgp) {
        case 0: hipLaunchKernelGGL(k<0>, dim3(1), dim3(64), 0, 0, dA, dB, dD); break;
        case 1: hipLaunchKernelGGL(k<1>, dim3(1), dim3(64), 0, 0, dA, dB, dD); break;
        case 2: hipLaunchKernelGGL(k<2>, dim3(1), dim3(64), 0, 0, dA, dB, dD); break;
        case 3: hipLaunchKernelGGL(k<3>, dim3(1), dim3(64), 0, 0, dA, dB, dD); break;
        case 4: hipLaunchKernelGGL(k<4>, dim3(1), dim3(64), 0, 0, dA, dB, dD); break;
        case 5: hipLaunchKernelGGL(k<5>, dim3(1), dim3(64), 0, 0, dA, dB, dD); break;
        case 6: hipLaunchKernelGGL(k<6>, dim3(1), dim3(64), 0, 0, dA, dB, dD); break;
        default: hipLaunchKernelGGL(k<7>, dim3(1), dim3(64), 0, 0, dA, dB, dD); break;
        }
        hipError_t e = hipDeviceSynchronize();
        hipMemcpy(hD, dD, 1024, hipMemcpyDeviceToHost);
        printf("blgp %d (%s):", blgp, hipGetErrorString(e));
        // candidate lane-group sources for k-groups 0..3
        const int cand[][4] = {{0, 1, 2, 3}, {0, 1, 0, 1}, {2, 3, 2, 3}, {1, 2, 3, 0}, {3, 0, 1, 2}, {0, 0, 0, 0}, {1, 1, 1, 1},
                               {2, 2, 2, 2}, {3, 3, 3, 3}, {2, 3, 0, 1}};
        const char *names[] = {"identity", "bcast lo32", "bcast hi32", "rotate down 16 (g<-g+1)", "rotate up 16 (g<-g-1)", "bcast g0",
                               "bcast g1", "bcast g2", "bcast g3", "swap halves"};
        bool any = false;
        for (unsigned c = 0; c < sizeof cand / sizeof cand[0]; ++c) {
            ref_with_groups(cand[c], ref);
            int bad = 0; for (int i = 0; i < 256; ++i) bad += hD[i] != ref[i];
            if (!bad) { printf(" == %s", names[c]); any = true; }
        }
        if (!any) printf(" matches none of the candidates");
        printf("\n");
    }
    return 0;
}
