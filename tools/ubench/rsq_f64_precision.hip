// How accurate is v_rsq_f64 on gfx950?  (round 5)  The shortened normalisation of the winner's NCC matrix and the Hessian's hypot
// refine v_rsq_f64 with coupled Newton steps; one step suffices if the instruction is good to ~2^-26 (the result is then within
// a few ulp of the correctly rounded double, far inside the 64-ulp guard band that sends lanes to the specification's route).
// Prints the largest relative error of v_rsq_f64, and of one and of two coupled Newton steps, against a reference refined four
// times, over 2^31 arguments spread over [2^-60, 2^60] (mantissas uniform).
//   hipcc --offload-arch=gfx950 -O2 -ffp-contract=off -o rsq_f64_precision rsq_f64_precision.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
__global__ void k(unsigned long long seed, int per_thread, double *out)
{
    unsigned long long st = seed ^ (0x9e3779b97f4a7c15ull * (unsigned long long)(blockIdx.x * blockDim.x + threadIdx.x + 1));
    auto next = [&]() { st ^= st << 13; st ^= st >> 7; st ^= st << 17; return st; };
    double e0 = 0, e1 = 0, e2 = 0;
    for (int it = 0; it < per_thread; ++it) {
        const unsigned long long a = next();
        const int ex = 1023 - 60 + (int)((a >> 52) % 121);
        const double x = __longlong_as_double(((unsigned long long)ex << 52) | (a & 0xfffffffffffffull));
        const double y0 = __builtin_amdgcn_rsq(x);
        // coupled Newton steps as in exact_from_sums_fast / hypot_fast: g ~ sqrt(x), h ~ 1 / (2 sqrt(x))
        double g = x * y0, h = 0.5 * y0;
        double r = __builtin_fma(-h, g, 0.5);
        g = __builtin_fma(g, r, g); h = __builtin_fma(h, r, h);
        const double h1 = h, g1 = g;
        r = __builtin_fma(-h, g, 0.5);
        g = __builtin_fma(g, r, g); h = __builtin_fma(h, r, h);
        const double h2 = h;
        // reference: two more steps (quadratic convergence: far below one ulp), then the residual of 2 h against it
        double gr = g, hr = h;
        for (int q = 0; q < 2; ++q) { r = __builtin_fma(-hr, gr, 0.5); gr = __builtin_fma(gr, r, gr); hr = __builtin_fma(hr, r, hr); }
        const double ref = 2.0 * hr;
        e0 = fmax(e0, fabs(y0 - ref) / ref);
        e1 = fmax(e1, fabs(2.0 * h1 - ref) / ref);
        e2 = fmax(e2, fabs(2.0 * h2 - ref) / ref);
        (void)g1;
    }
    // block maximum by atomics on the bit patterns (non-negative doubles order like their bits)
    atomicMax(reinterpret_cast<unsigned long long *>(out), (unsigned long long)__double_as_longlong(e0));
    atomicMax(reinterpret_cast<unsigned long long *>(out) + 1, (unsigned long long)__double_as_longlong(e1));
    atomicMax(reinterpret_cast<unsigned long long *>(out) + 2, (unsigned long long)__double_as_longlong(e2));
}
int main()
{
    double *d; hipMalloc(&d, 24); hipMemset(d, 0, 24);
    const int blocks = 8192, threads = 256, per = 1024;           // 2^31 arguments
    hipLaunchKernelGGL(k, dim3(blocks), dim3(threads), 0, 0, 0x1234567ull, per, d);
    hipDeviceSynchronize();
    double h[3]; hipMemcpy(h, d, 24, hipMemcpyDeviceToHost);
    printf("v_rsq_f64 over %lld arguments: max relative error %.3e (2^%.1f); after one coupled Newton step %.3e (2^%.1f = %.1f ulp of a double); after two %.3e (2^%.1f)\n",
           (long long)blocks * threads * per, h[0], log2(h[0]), h[1], log2(h[1]), h[1] / 1.11e-16, h[2], log2(h[2] > 0 ? h[2] : 1e-300));
    return 0;
}
