// Does private (scratch) memory keep its contents while three 52 KB-LDS workgroups share a CU?
// Every lane parks 24 known dwords in scratch, the workgroup then churns LDS, MFMAs and global loads for a while
// (with calls to a non-inlined function that saves registers on the same stack), and the parked values are read
// back and compared.  Prints the number of corrupted dwords per configuration (LDS bytes per workgroup -> 3, 2 or
// 1 workgroups per CU).  Background: DESIGN.md section 6b (the determinism failures of spilling kernel builds).
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef int v4i __attribute__((ext_vector_type(4)));

__device__ __noinline__ int churn(int *lds, int n, int seed)
{
    v4i acc = {0, 0, 0, 0};
    v4i a = {seed, seed + 1, seed + 2, seed + 3}, b = {seed ^ 5, seed ^ 9, seed ^ 17, seed ^ 33};
    for (int i = 0; i < 64; ++i) {
        acc = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, acc, 0, 0, 0);
        const int k = (threadIdx.x * 7 + i * 131 + seed) % n;
        atomicAdd(&lds[k], acc[0] & 3);
        a[0] += lds[(k + 64) % n];
    }
    return acc[0] + acc[1] + acc[2] + acc[3];
}

__global__ __launch_bounds__(256) void k(const int *src, int *bad, int rounds)
{
    extern __shared__ int lds[];
    const int n = 12 * 1024;
    for (int i = threadIdx.x; i < n; i += 256) lds[i] = i;
    __syncthreads();
    volatile int park[24];                                             // volatile: lives in scratch, not in registers
    const int me = blockIdx.x * 256 + threadIdx.x;
    for (int j = 0; j < 24; ++j) park[j] = me * 31 + j * 1009;
    int sink = 0;
    for (int r = 0; r < rounds; ++r) {
        sink += churn(lds, n, r + src[(me + r * 4099) & 0xfffff]);
        __syncthreads();
    }
    int nb = 0;
    for (int j = 0; j < 24; ++j) nb += park[j] != me * 31 + j * 1009;
    if (nb) atomicAdd(bad, nb);
    if (sink == 0x7fffffff) atomicAdd(bad, 1 << 20);
}

int main()
{
    int *src, *bad;
    hipMalloc(&src, 4 << 20); hipMemset(src, 1, 4 << 20); hipMalloc(&bad, 4);
    const int lds_cfg[3] = {52 * 1024, 70 * 1024, 120 * 1024};
    for (int c = 0; c < 3; ++c) {
        hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize, lds_cfg[c]);
        int total = 0;
        for (int rep = 0; rep < 20; ++rep) {
            hipMemset(bad, 0, 4);
            hipLaunchKernelGGL(k, dim3(256 * 24), dim3(256), lds_cfg[c], 0, src, bad, 40);
            int h = 0; hipMemcpy(&h, bad, 4, hipMemcpyDeviceToHost);
            total += h;
        }
        printf("LDS %6d B per workgroup (%d per CU): %d corrupted scratch dwords in 20 launches of 6144 workgroups\n",
               lds_cfg[c], 160 * 1024 / lds_cfg[c], total);
    }
    return 0;
}
