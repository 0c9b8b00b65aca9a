// fastmath_check.hip -- the claim behind the sweep's fast arithmetic (sweep_kernels.hip: rcp_refined / div_by /
// fast_sqrt), checked on the device against the compiler's own IEEE sequences: for operands inside
// [2^-150, 2^150] (products / quotients of such operands included, as they occur in the closed forms of
// src/cfmms.jl:125-126, :321-337) the quotient and the square root are THE SAME BITS, over N random operand pairs
// per launch.  Compiled and run by tests/test_gpu_fastmath.py (hipcc is part of the image on the GPU box).
// Prints "FASTMATH_CHECK pairs=<n> div_mismatch=<k> sqrt_mismatch=<k> zero_mismatch=<k>".
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>

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

__global__ void check(uint64_t seed, long long per_thread, unsigned long long* out)
{
    uint64_t s = seed + 0x1000003ull * (blockIdx.x * (uint64_t)blockDim.x + threadIdx.x);
    unsigned long long bad_div = 0, bad_sqrt = 0, bad_zero = 0;
    for (long long k = 0; k < per_thread; ++k) {
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
}

int main(int argc, char** argv)
{
    const long long per_thread = argc > 1 ? atoll(argv[1]) : 1024;
    unsigned long long* d = nullptr;
    if (hipMalloc(&d, 3 * sizeof(unsigned long long)) != hipSuccess) { fprintf(stderr, "no device\n"); return 2; }
    hipMemset(d, 0, 3 * sizeof(unsigned long long));
    const int blocks = 4096, threads = 256;
    hipLaunchKernelGGL(check, dim3(blocks), dim3(threads), 0, 0, 0x2545f4914f6cdd1dull, per_thread, d);
    unsigned long long h[3] = {0, 0, 0};
    if (hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost) != hipSuccess) { fprintf(stderr, "kernel failed\n"); return 3; }
    printf("FASTMATH_CHECK pairs=%lld div_mismatch=%llu sqrt_mismatch=%llu zero_mismatch=%llu\n",
           (long long)blocks * threads * per_thread, h[0], h[1], h[2]);
    return (h[0] | h[1] | h[2]) ? 1 : 0;
}
