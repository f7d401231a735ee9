// Packed-FP32 arithmetic next to another wavefront's MFMAs (gfx950) - the micro-benchmark behind DESIGN.md section 6b.
//
// hipcc -O3 with the SLP vectoriser on turned the float32 fast path of the on-the-fly template sampler into packed
// instructions; the stretch below is copied from that build (registers renamed):
//
//   v_cvt_f32_f64 v9, v[8:9]      v_cvt_f32_f64 v8, v[6:7]            ; rotation terms  (float)a, (float)b
//   v_add_f64 ... (4x)            v_cvt_f32_f64 v7, v[6:7]   v_cvt_f32_f64 v6, v[82:83]   ; offsets
//   v_pk_add_f32 v[82:83], v[8:9], 0 neg_lo:[1,1] neg_hi:[1,1]
//   v_pk_fma_f32 v[6:7], v[18:19], v[8:9], v[6:7] op_sel:[0,1,0] op_sel_hi:[1,0,1]     ; low result takes the HIGH half of v[8:9]
//   v_mov_b32    v82, v8
//   v_pk_fma_f32 v[6:7], v[20:21], v[82:83], v[6:7]
//
// In the PM kernel - and only with three workgroups per CU - lanes 48..63 of a wavefront occasionally came out of this
// with a wrong coordinate.  Here every wavefront runs the stretch `iters` times on changing data and compares the
// results with the same arithmetic written in C, per quarter of the wavefront, in several variants and with different
// company on the SIMD: none, workgroups that only issue VALU FMAs, workgroups that issue v_mfma_i32_16x16x64_i8.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

typedef int v4i __attribute__((ext_vector_type(4)));

#define CONSTS "v_mov_b32 v18, 1.0\n v_mov_b32 v19, 2.0\n v_mov_b32 v20, 0.5\n v_mov_b32 v21, 4.0\n"
#define LOADS_F64 \
    "ds_read2_b64 v[6:9], %4 offset0:0 offset1:1\n ds_read2_b64 v[40:43], %4 offset0:2 offset1:3\n" \
    "s_waitcnt lgkmcnt(1)\n v_cvt_f32_f64 v9, v[8:9]\n v_cvt_f32_f64 v8, v[6:7]\n s_waitcnt lgkmcnt(0)\n" \
    "v_add_f64 v[6:7], %5, -v[42:43]\n v_add_f64 v[82:83], %6, -v[40:41]\n" \
    "v_add_f64 v[82:83], v[82:83], -%8\n v_add_f64 v[6:7], v[6:7], -%7\n" \
    "v_cvt_f32_f64 v7, v[6:7]\n v_cvt_f32_f64 v6, v[82:83]\n"
#define PK_NEG  "v_pk_add_f32 v[82:83], v[8:9], 0 neg_lo:[1,1] neg_hi:[1,1]\n"
#define PK_FMA1 "v_pk_fma_f32 v[6:7], v[18:19], v[8:9], v[6:7] op_sel:[0,1,0] op_sel_hi:[1,0,1]\n"
#define MOVE    "v_mov_b32 v82, v8\n"
#define PK_FMA2 "v_pk_fma_f32 v[6:7], v[20:21], v[82:83], v[6:7]\n"
#define RESULTS "s_nop 7\n v_mov_b32 %0, v82\n v_mov_b32 %1, v6\n v_mov_b32 %2, v7\n v_mov_b32 %3, v8\n"
#define OPERANDS : "=v"(r82), "=v"(r6), "=v"(r7), "=v"(r8) : "v"(addr), "v"(c2), "v"(c4), "v"(k26), "v"(k28) \
                 : "v6", "v7", "v8", "v9", "v18", "v19", "v20", "v21", "v40", "v41", "v42", "v43", "v82", "v83", "memory"

// VARIANT 0: as emitted   1: s_nop 7 twice between the last convert and the packed instructions
//         2: the first packed FMA as two plain v_fma_f32 (no packed instruction reads across halves)
//         3: all four packed instructions replaced by plain VALU instructions (what -fno-slp-vectorize gives)
//         4: first packed FMA with op_sel_hi:[0,1,1] (the HIGH result takes the LOW half of src0: the form the PM kernel's
//            scoring uses for a broadcast operand)   5: first packed FMA without any op_sel
// COMPANY 0: every workgroup runs the test   1: every third workgroup spins on v_fma_f32   2: ... on MFMAs
template <int VARIANT>
__global__ __launch_bounds__(256) void k(unsigned *bad_per_lane, int iters, int company, float *dump)
{
    extern __shared__ unsigned lds[];
    asm volatile("v_mov_b32 v167, 0" ::: "v167");      // 168 VGPRs per wavefront as in the PM kernel: three per SIMD, bases 0 / 168 / 336
    if (company && blockIdx.x % 3u == 0u) {
        if (company == 2) {
            v4i acc = {0, 0, 0, 0}, x = {(int)threadIdx.x, 2, 3, 4}, y = {5, 6, 7, (int)blockIdx.x};
            for (int it = 0; it < iters * 8; ++it) {
                acc = __builtin_amdgcn_mfma_i32_16x16x64_i8(x, y, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_i32_16x16x64_i8(y, x, acc, 0, 0, 0);
            }
            if (acc[0] == 0x12345678) bad_per_lane[255] = 1;
        } else {
            float u = (float)threadIdx.x, w = 1.0001f;
            for (int it = 0; it < iters * 64; ++it) { u = fmaf(u, w, 0.5f); w = fmaf(w, 0.9999f, 1e-6f); }
            if (u == 0.12345f) bad_per_lane[255] = 1;
        }
        return;
    }
    double *ld = reinterpret_cast<double *>(lds) + 4 * threadIdx.x;
    const int lane = threadIdx.x & 63;
    unsigned bad_copy = 0, bad_lo = 0, bad_hi = 0;
    double a = (double)(threadIdx.x + 1) * 0.37 + 100.0, b = (double)(blockIdx.x + 3) * 1.3;
    for (int it = 0; it < iters; ++it) {
        a = a * 1.0001 + 0.5; b = b * 0.9999 - 0.25;
        ld[0] = a; ld[1] = b; ld[2] = a + 3.0; ld[3] = b - 7.0;
        __syncthreads();
        const unsigned addr = (unsigned)(threadIdx.x * 32);
        float r82, r6, r7, r8;
        const double c2 = 1000.0 + a, c4 = 2000.0 + b, k26 = 5.5, k28 = 6.5;
        if (VARIANT == 0) asm volatile(CONSTS LOADS_F64 PK_NEG PK_FMA1 MOVE PK_FMA2 RESULTS OPERANDS);
        else if (VARIANT == 1) asm volatile(CONSTS LOADS_F64 "s_nop 7\n s_nop 7\n" PK_NEG PK_FMA1 MOVE PK_FMA2 RESULTS OPERANDS);
        else if (VARIANT == 2) asm volatile(CONSTS LOADS_F64 PK_NEG "v_fma_f32 v6, v18, v9, v6\n v_fma_f32 v7, v19, v8, v7\n" MOVE PK_FMA2 RESULTS OPERANDS);
        else if (VARIANT == 4) asm volatile(CONSTS LOADS_F64 PK_NEG "v_pk_fma_f32 v[6:7], v[18:19], v[8:9], v[6:7] op_sel_hi:[0,1,1]\n" MOVE PK_FMA2 RESULTS OPERANDS);
        else if (VARIANT == 5) asm volatile(CONSTS LOADS_F64 PK_NEG "v_pk_fma_f32 v[6:7], v[18:19], v[8:9], v[6:7]\n" MOVE PK_FMA2 RESULTS OPERANDS);
        else asm volatile(CONSTS LOADS_F64 "v_xor_b32 v83, 0x80000000, v9\n v_fma_f32 v6, v18, v9, v6\n v_fma_f32 v7, v19, v8, v7\n" MOVE
                          "v_fma_f32 v6, v20, v82, v6\n v_fma_f32 v7, v21, v83, v7\n" RESULTS OPERANDS);
        // the same arithmetic in C
        const float f8 = (float)a, f9 = (float)b;
        const float f7 = (float)((c2 - (b - 7.0)) - k26), f6 = (float)((c4 - (a + 3.0)) - k28);
        const float p6 = VARIANT == 4 || VARIANT == 5 ? fmaf(1.0f, f8, f6) : fmaf(1.0f, f9, f6);
        const float p7 = VARIANT == 4 ? fmaf(1.0f, f9, f7) : VARIANT == 5 ? fmaf(2.0f, f9, f7) : fmaf(2.0f, f8, f7);
        const float e6 = fmaf(0.5f, f8, p6), e7 = fmaf(4.0f, -f9, p7);
        bad_copy += (__float_as_uint(r82) != __float_as_uint(f8) || __float_as_uint(r8) != __float_as_uint(f8)) ? 1u : 0u;
        const bool wlo = __float_as_uint(r6) != __float_as_uint(e6), whi = __float_as_uint(r7) != __float_as_uint(e7);
        bad_lo += wlo ? 1u : 0u; bad_hi += whi ? 1u : 0u;
        if ((wlo || whi) && dump) {
            const unsigned slot = atomicAdd(reinterpret_cast<unsigned *>(dump), 1u);
            if (slot < 4) { float *o = dump + 16 + 16 * slot; o[0] = (float)lane; o[1] = r6; o[2] = e6; o[3] = r7; o[4] = e7; o[5] = f8; o[6] = f9; o[7] = f6; o[8] = f7; o[9] = (float)it; }
        }
        __syncthreads();
    }
    if (bad_copy) atomicAdd(&bad_per_lane[lane], bad_copy);
    if (bad_lo) atomicAdd(&bad_per_lane[64 + lane], bad_lo);
    if (bad_hi) atomicAdd(&bad_per_lane[128 + lane], bad_hi);
}

int main(int argc, char **argv)
{
    const int iters = argc > 1 ? atoi(argv[1]) : 20000;
    unsigned *d; float *dd;
    (void)hipMalloc(&d, 256 * sizeof(unsigned)); (void)hipMalloc(&dd, 1024);
    const char *vn[6] = {"as emitted", "16 idle cycles before the packed ops", "first packed FMA as two v_fma_f32", "no packed instruction at all",
                         "first packed FMA with op_sel_hi:[0,1,1]", "first packed FMA without op_sel"};
    const char *cn[3] = {"all workgroups run the test", "every 3rd workgroup: v_fma_f32 loop", "every 3rd workgroup: MFMA loop"};
    void (*kerns[6])(unsigned *, int, int, float *) = {k<0>, k<1>, k<2>, k<3>, k<4>, k<5>};
    for (int company = 0; company < 3; ++company)
        for (int variant = 0; variant < 6; ++variant) {
            if (company < 2 && variant > 0) continue;
            (void)hipMemset(d, 0, 256 * sizeof(unsigned)); (void)hipMemset(dd, 0, 1024);
            (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kerns[variant]), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            hipLaunchKernelGGL(kerns[variant], dim3(256 * 3 * 4), dim3(256), 50 * 1024, 0, d, iters, company, dd);   // 50 KB: three workgroups per CU
            const hipError_t e = hipDeviceSynchronize();
            unsigned h[256]; float hd[256];
            (void)hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost); (void)hipMemcpy(hd, dd, 1024, hipMemcpyDeviceToHost);
            unsigned long long q[3][4] = {};
            for (int l = 0; l < 64; ++l) for (int c = 0; c < 3; ++c) q[c][l >> 4] += h[64 * c + l];
            const double evals = (double)iters * 64.0 * 4.0 * 3072.0 * (company ? 2.0 / 3.0 : 1.0) / 4.0;   // per quarter of the wavefronts
            printf("[%s | %s] wrong results per quarter of the wavefront (lanes 0-15, 16-31, 32-47, 48-63) of %.3g evaluations each:\n"
                   "    copy v82 = v8: %llu %llu %llu %llu   low half of the packed FMAs: %llu %llu %llu %llu   high half: %llu %llu %llu %llu   (%s)\n",
                   cn[company], vn[variant], evals, q[0][0], q[0][1], q[0][2], q[0][3], q[1][0], q[1][1], q[1][2], q[1][3],
                   q[2][0], q[2][1], q[2][2], q[2][3], hipGetErrorString(e));
            for (int k2 = 0; k2 < 2 && k2 < (int)*reinterpret_cast<unsigned *>(hd); ++k2) {
                const float *o = hd + 16 + 16 * k2;
                printf("      e.g. lane %2.0f, repetition %.0f: v6 = %.9g (expected %.9g), v7 = %.9g (expected %.9g); (float)a %.9g (float)b %.9g offsets %.9g %.9g\n",
                       o[0], o[9], o[1], o[2], o[3], o[4], o[5], o[6], o[7], o[8]);
            }
        }
    return 0;
}
