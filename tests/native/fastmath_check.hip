// fastmath_check.hip -- the claim behind the sweep's fast arithmetic (sweep_kernels.hip: rcp_refined / div_by /
// fast_sqrt), checked on the device against the compiler's own IEEE sequences: for operands inside
// [2^-150, 2^150] (products / quotients of such operands included, as they occur in the closed forms of
// src/cfmms.jl:125-126, :321-337) the quotient and the square root are THE SAME BITS, over N random operand pairs
// per launch.  Compiled and run by tests/test_gpu_fastmath.py (hipcc is part of the image on the GPU box).
// fast_exp (the log-space GeometricMean form; not bit-exact by construction, the family's bar is 1e-12) is compared
// with the device library's exp over |x| <= 320 (ulp distance) and, on a sample, with the host's long-double expl.
// Prints "FASTMATH_CHECK pairs=<n> div_mismatch=<k> sqrt_mismatch=<k> zero_mismatch=<k> exp_max_ulp_vs_lib=<k>
// exp_over_1ulp_vs_lib=<k> exp_max_err_ulp_vs_expl=<x>".
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

__device__ __forceinline__ double rcp_refined(double b)
{
    double y = __builtin_amdgcn_rcp(b);
    double e = __builtin_fma(-b, y, 1.0);
    y = __builtin_fma(y, e, y);
    e = __builtin_fma(-b, y, 1.0);
    return __builtin_fma(y, e, y);
}
__device__ __forceinline__ double div_by(double a, double b, double yb)
{
    const double q = a * yb;
    const double r = __builtin_fma(-b, q, a);
    return __builtin_fma(r, yb, q);
}
__device__ __forceinline__ double fast_sqrt(double x)
{
    const double y = __builtin_amdgcn_rsq(x);
    double s = x * y;
    double h = y * 0.5;
    const double r = __builtin_fma(-h, s, 0.5);
    s = __builtin_fma(s, r, s);
    double d = __builtin_fma(-s, s, x);
    h = __builtin_fma(h, r, h);
    s = __builtin_fma(d, h, s);
    d = __builtin_fma(-s, s, x);
    return __builtin_fma(d, h, s);
}

__device__ __forceinline__ double fma_sc(double x, double acc, double c)
{
    double r;
    asm("v_fma_f64 %0, %1, %2, %3" : "=v"(r) : "v"(x), "v"(acc), "s"(c));
    return r;
}
__device__ __forceinline__ double fast_exp(double x)
{
    const double k = __builtin_rint(x * 0x1.71547652b82fep+0);
    double r = __builtin_fma(-k, 0x1.62e42fefa39efp-1, x);
    r = __builtin_fma(-k, 0x1.abc9e3b39803fp-56, r);
    double p = 0x1.af39091a8441ap-26;
    p = fma_sc(r, p, 0x1.2891d2ecb3ed9p-22);
    p = fma_sc(r, p, 0x1.71de0d863c737p-19);
    p = fma_sc(r, p, 0x1.a019b8cbe6585p-16);
    p = fma_sc(r, p, 0x1.a01a01a7ce75dp-13);
    p = fma_sc(r, p, 0x1.6c16c1789caa1p-10);
    p = fma_sc(r, p, 0x1.11111111109a6p-7);
    p = fma_sc(r, p, 0x1.5555555553d38p-5);
    p = fma_sc(r, p, 0x1.5555555555556p-3);
    p = fma_sc(r, p, 0x1.0000000000001p-1);
    p = __builtin_fma(r, p, 1.0);
    p = __builtin_fma(r, p, 1.0);
    return __builtin_ldexp(p, (int)k);
}

__device__ __forceinline__ uint64_t splitmix(uint64_t& s)
{
    uint64_t z = (s += 0x9e3779b97f4a7c15ull);
    z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
    z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
    return z ^ (z >> 31);
}
// random double with a uniformly random mantissa and an exponent uniform in [-range, range]
__device__ __forceinline__ double rnd(uint64_t& s, int range, bool neg_ok)
{
    const uint64_t m = splitmix(s);
    const int e = (int)(splitmix(s) % (uint64_t)(2 * range + 1)) - range;
    const uint64_t bits = ((uint64_t)(1023 + e) << 52) | (m & 0xfffffffffffffull) | ((neg_ok && (m >> 63)) ? (1ull << 63) : 0);
    return __longlong_as_double((long long)bits);
}

constexpr int kSample = 1 << 16;   // (x, fast_exp(x)) pairs handed to the host for the expl comparison

__global__ void check(uint64_t seed, long long per_thread, unsigned long long* out, double2* sample)
{
    const uint64_t gid = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
    uint64_t s = seed + 0x1000003ull * gid;
    unsigned long long bad_div = 0, bad_sqrt = 0, bad_zero = 0, exp_over = 0, exp_max = 0;
    for (long long k = 0; k < per_thread; ++k) {
        // exponent arguments: logarithms of reserves, uniform in [-320, 320] (the window allows ~312), plus tiny ones
        const uint64_t u = splitmix(s);
        double xe = ((double)(u >> 11) * 0x1p-53 - 0.5) * 640.0;
        if ((u & 15) == 0) xe *= 0x1p-30;
        const double e1 = exp(xe), e2 = fast_exp(xe);
        const long long du = __double_as_longlong(e1) - __double_as_longlong(e2);
        const unsigned long long ad = (unsigned long long)(du < 0 ? -du : du);
        exp_over += ad > 1;
        exp_max = ad > exp_max ? ad : exp_max;
        if (k == 0 && gid < (uint64_t)kSample) sample[gid] = make_double2(xe, e2);
        // numerators as wide as the closed forms produce them (products of up to four window operands), any sign
        const double b = rnd(s, 320, false);          // divisors: prices, fees, gm = γ·m, internal prices
        const double a = rnd(s, 620, true);
        const double q1 = a / b, q2 = div_by(a, b, rcp_refined(b));
        bad_div += __double_as_longlong(q1) != __double_as_longlong(q2);
        const double x = rnd(s, 640, false);          // radicands: gm·k, k/gm, k/price, price·k
        bad_sqrt += __double_as_longlong(sqrt(x)) != __double_as_longlong(fast_sqrt(x));
        // +0 numerators (max0(...) = 0: pools at the no-arbitrage boundary) give +0
        const double z = div_by(0.0, b, rcp_refined(b));
        bad_zero += __double_as_longlong(z) != 0;
    }
    atomicAdd(out + 0, bad_div);
    atomicAdd(out + 1, bad_sqrt);
    atomicAdd(out + 2, bad_zero);
    atomicAdd(out + 3, exp_over);
    atomicMax(out + 4, exp_max);
}

int main(int argc, char** argv)
{
    const long long per_thread = argc > 1 ? atoll(argv[1]) : 1024;
    unsigned long long* d = nullptr;
    double2* d_sample = nullptr;
    if (hipMalloc(&d, 5 * sizeof(unsigned long long)) != hipSuccess) { fprintf(stderr, "no device\n"); return 2; }
    if (hipMalloc(&d_sample, kSample * sizeof(double2)) != hipSuccess) { fprintf(stderr, "no memory\n"); return 2; }
    hipMemset(d, 0, 5 * sizeof(unsigned long long));
    const int blocks = 4096, threads = 256;
    hipLaunchKernelGGL(check, dim3(blocks), dim3(threads), 0, 0, 0x2545f4914f6cdd1dull, per_thread, d, d_sample);
    unsigned long long h[5] = {0, 0, 0, 0, 0};
    if (hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost) != hipSuccess) { fprintf(stderr, "kernel failed\n"); return 3; }
    std::vector<double2> sample(kSample);
    if (hipMemcpy(sample.data(), d_sample, kSample * sizeof(double2), hipMemcpyDeviceToHost) != hipSuccess) return 3;
    double worst = 0.0;   // |fast_exp(x) - expl(x)| in ulps of the result (x87 extended precision: 64-bit mantissa)
    for (const double2& p : sample) {
        const long double t = expl((long double)p.x);
        int e;
        (void)frexpl(t, &e);
        const long double ulp = ldexpl(1.0L, e - 53);
        const double err = (double)(fabsl((long double)p.y - t) / ulp);
        worst = err > worst ? err : worst;
    }
    printf("FASTMATH_CHECK pairs=%lld div_mismatch=%llu sqrt_mismatch=%llu zero_mismatch=%llu exp_max_ulp_vs_lib=%llu "
           "exp_over_1ulp_vs_lib=%llu exp_max_err_ulp_vs_expl=%.4f\n",
           (long long)blocks * threads * per_thread, h[0], h[1], h[2], h[4], h[3], worst);
    return (h[0] | h[1] | h[2]) ? 1 : 0;
}
