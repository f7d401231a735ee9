#include <hip/hip_runtime.h>
// Probe behind PMArgs::gs_pool (round 4): (XCC_ID, HW_ID.se / cu, LDS_ALLOC.base) is a unique key among CO-RESIDENT workgroups -
// every workgroup marks its key busy with an atomic on entry (a key found busy is a violation) and free on exit.
// ./resident_slot <dynamic LDS bytes> <threads>; MI355X: 0 violations in 4 x 20 000 workgroups at 53 764 / 81 536 / 91 040 / 40 964 bytes.
#include <cstdio>
#include <cstdlib>
#include <set>
// probe: is (XCC_ID, HW_ID.se/sh/cu, LDS_ALLOC.base) a unique key among CO-RESIDENT workgroups?
__global__ void k(unsigned *occ, unsigned *viol, unsigned *keys, int spin)
{
    extern __shared__ unsigned char smem[];
    __shared__ unsigned key_s;
    if (threadIdx.x == 0) {
        const unsigned xcc = __builtin_amdgcn_s_getreg((3 << 11) | 20) & 7u;
        const unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | 4);        // HW_REG_HW_ID
        const unsigned la = __builtin_amdgcn_s_getreg((31 << 11) | 6);        // HW_REG_LDS_ALLOC
        keys[3 * blockIdx.x] = xcc; keys[3 * blockIdx.x + 1] = hw; keys[3 * blockIdx.x + 2] = la;
        const unsigned cu = (hw >> 8) & 15u, sh = (hw >> 12) & 1u, se = (hw >> 13) & 7u;   // gfx9 HW_ID layout: cu_id [11:8], sh_id [12], se_id [15:13]
        const unsigned base = la & 0x1ffu;                                    // lds_base (units unknown: probe)
        const unsigned key = (((xcc * 8 + se) * 2 + sh) * 16 + cu) * 512 + base;
        key_s = key;
        if (atomicAdd(&occ[key], 1u) != 0u) atomicAdd(viol, 1u);
    }
    __syncthreads();
    volatile unsigned char *p = smem;
    unsigned acc = 0;
    for (int i = 0; i < spin; ++i) { p[(threadIdx.x * 7 + i) & 1023] = (unsigned char)i; acc += p[(threadIdx.x + i) & 1023]; }
    __syncthreads();
    if (threadIdx.x == 0) { atomicSub(&occ[key_s], 1u); if (acc == 0xffffffffu) viol[1] = 1; }
}
int main(int argc, char **argv)
{
    const int lds = argc > 1 ? atoi(argv[1]) : 53760, blocks = 20000, thr = argc > 2 ? atoi(argv[2]) : 256;
    unsigned *occ, *viol, *keys;
    const size_t nk = (size_t)8 * 8 * 2 * 16 * 512;
    hipMalloc(&occ, nk * 4); hipMemset(occ, 0, nk * 4); hipMalloc(&viol, 8); hipMemset(viol, 0, 8); hipMalloc(&keys, 12 * blocks);
    hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipLaunchKernelGGL(k, dim3(blocks), dim3(thr), lds, 0, occ, viol, keys, 3000);
    hipDeviceSynchronize();
    unsigned v[2]; hipMemcpy(v, viol, 8, hipMemcpyDeviceToHost);
    unsigned *h = (unsigned *)malloc(12 * blocks); hipMemcpy(h, keys, 12 * blocks, hipMemcpyDeviceToHost);
    std::set<unsigned> bases, cus; std::set<unsigned long long> all;
    for (int i = 0; i < blocks; ++i) { bases.insert(h[3 * i + 2] & 0xffffu); cus.insert((h[3 * i + 1] >> 8) & 0xffu); all.insert(((unsigned long long)h[3 * i] << 40) | ((unsigned long long)(h[3 * i + 1] & 0xff00u) << 8) | (h[3 * i + 2] & 0x1ffu)); }
    printf("lds %d thr %d: violations %u, distinct keys %zu, distinct LDS_ALLOC[15:0] values:", lds, thr, v[0], all.size());
    for (unsigned b : bases) printf(" %04x", b);
    printf("\ndistinct HW_ID[15:8]: %zu;", cus.size());
    for (int i = 0; i < 6; ++i) printf(" [xcc %u hw %08x lds_alloc %08x]", h[3 * i], h[3 * i + 1], h[3 * i + 2]);
    printf("\n");
    return 0;
}
