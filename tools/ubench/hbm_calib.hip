// Calibration of rocprofv3 FETCH_SIZE / WRITE_SIZE for dword-per-lane access (the PM kernel's pattern:
// ph_window reads dwords, the callee-saved register spills are scratch dword stores/loads).
// Reads 2 GiB and writes 2 GiB once each; compare the counters with these byte counts.
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void read_dwords(const unsigned *p, size_t n, unsigned *sink) {
    unsigned acc = 0;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) acc += p[i];
    if (acc == 0x12345678u) *sink = acc;
}
__global__ void write_dwords(unsigned *p, size_t n) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = (unsigned)i;
}
int main() {
    const size_t n = (size_t)512 << 20;                 // 512 Mi dwords = 2 GiB
    unsigned *p, *s; hipMalloc(&p, n * 4); hipMalloc(&s, 4);
    hipMemset(p, 1, n * 4); hipDeviceSynchronize();
    hipLaunchKernelGGL(write_dwords, dim3(8192), dim3(256), 0, 0, p, n);
    hipLaunchKernelGGL(read_dwords, dim3(8192), dim3(256), 0, 0, p, n, s);
    hipDeviceSynchronize();
    printf("bytes written by write_dwords: %zu, bytes read by read_dwords: %zu\n", n * 4, n * 4);
    return 0;
}
