// Probe: LDS allocation granularity and blocks per CU on gfx950 (reads HW_REG_LDS_ALLOC / HW_REG_HW_ID).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <set>
__global__ void k(int *o) {
    extern __shared__ int s[];
    s[threadIdx.x] = threadIdx.x;
    __syncthreads();
    // keep the block resident for a while so that co-resident blocks pile up
    long long t0 = clock64(); while (clock64() - t0 < 200000) {}
    if (threadIdx.x == 0) {
        o[2 * blockIdx.x] = (int)__builtin_amdgcn_s_getreg((31 << 11) | 6);     // LDS_ALLOC
        o[2 * blockIdx.x + 1] = (int)__builtin_amdgcn_s_getreg((31 << 11) | 4);  // HW_ID
    }
}
int main() {
    int *d; hipMalloc(&d, 8 * 4096);
    const int sizes[] = {1024, 1280, 1281, 2560, 2561, 52032, 53760, 53761, 53904, 54613, 81920, 100000, 163840};
    for (int sz : sizes) {
        hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, sz);
        hipLaunchKernelGGL(k, dim3(4096), dim3(256), sz, 0, d);
        if (hipDeviceSynchronize() != hipSuccess) { printf("%d: launch failed\n", sz); continue; }
        static int h[8192]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
        std::set<int> bases; int size_field = 0;
        for (int b = 0; b < 4096; ++b) { bases.insert(h[2 * b] & 0xfff); size_field = (h[2 * b] >> 12) & 0xfff; }
        printf("dyn lds %6d: size field %4d (x256 = %6d B), distinct base fields %zu\n", sz, size_field, size_field * 256, bases.size());
    }
    return 0;
}
