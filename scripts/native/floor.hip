// floor.hip -- launch-latency floors on the MI355X box, for DESIGN.md: what the cheapest possible kernel
// costs when measured the way bench.py measures the sweep and the fold (hipExtLaunchKernel start/stop
// events written by the command processor), and what a dependent kernel pair costs per step.
//   (a) empty kernel, 1 block            (b) empty kernel, 512 blocks x 512 threads
//   (c) a fold-shaped kernel: 33 blocks x 512 threads summing 512 rows x 257 doubles (1 MB)
//   (d) a 16-byte-per-lane copy of B bytes for B = 6.4 MB, 32 MB, 64 MB, 72 MB (the practical roofline of
//       a kernel of that size: read B/2, write B/2)
// build: hipcc --offload-arch=gfx950 -O3 -o scripts/native/floor.bin scripts/native/floor.hip
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
#include <vector>

__global__ void empty_k() {}
__global__ __launch_bounds__(512) void fold_k(const double* __restrict__ p, int rows, int n1, double* __restrict__ out)
{
    __shared__ double red[8 * 8];
    const int c = threadIdx.x % 8, r = threadIdx.x / 8, col = blockIdx.x * 8 + c;
    double s = 0.0;
    if (col < n1) {
        double x[8];
#pragma unroll
        for (int b = 0; b < 8; ++b) x[b] = (r + b * 64 < rows) ? p[(size_t)(r + b * 64) * n1 + col] : 0.0;
#pragma unroll
        for (int b = 0; b < 8; ++b) s += x[b];
    }
    for (int off = 32; off >= 8; off >>= 1) s += __shfl_down(s, off, 64);
    if ((threadIdx.x & 63) < 8) red[(threadIdx.x >> 6) * 8 + (threadIdx.x & 63)] = s;
    __syncthreads();
    if (threadIdx.x < 8 && col < n1) {
        double t = 0;
        for (int k = 0; k < 8; ++k) t += red[k * 8 + c];
        out[col] = t;
    }
}
__global__ __launch_bounds__(256) void copy_k(const double2* __restrict__ a, double2* __restrict__ b, long n)
{
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) b[i] = a[i];
}

// read-only stream: every lane sums 16-byte loads; one partial per block (what a FUSED sweep's memory side is)
__global__ __launch_bounds__(256) void read_k(const double2* __restrict__ a, double* __restrict__ out, long n)
{
    double s = 0.0;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const double2 x = a[i];
        s += x.x + x.y;
    }
    for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
    if ((threadIdx.x & 63) == 0) atomicAdd(&out[blockIdx.x], s);
}

// the same read-only stream with a per-launch direction: block-strided "phases" k = 0..K-1 walked forwards or
// backwards, as a sweep's tile loop would -- does the XCD's 4 MB L2 keep the tail of the previous launch?
__global__ __launch_bounds__(256) void read_dir_k(const double2* __restrict__ a, double* __restrict__ out, long n, int reverse)
{
    const long stride = (long)gridDim.x * 256;
    const long first = (long)blockIdx.x * 256 + threadIdx.x;
    const long count = first < n ? (n - first + stride - 1) / stride : 0;
    double s = 0.0;
    for (long k = 0; k < count; ++k) {
        const long i = first + (reverse ? count - 1 - k : k) * stride;
        const double2 x = a[i];
        s += x.x + x.y;
    }
    for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
    if ((threadIdx.x & 63) == 0) atomicAdd(&out[blockIdx.x], s);
}

template <class F>
static double span_us(F launch, hipStream_t s, int reps = 200)
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    double tot = 0;
    for (int i = 0; i < reps + 20; ++i) {
        launch(e0, e1);
        hipStreamSynchronize(s);
        float ms = 0;
        hipEventElapsedTime(&ms, e0, e1);
        if (i >= 20) tot += ms;
    }
    return 1e3 * tot / reps;
}

int main()
{
    hipStream_t s;
    hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    double *rows, *out;
    hipMalloc(&rows, 512 * 257 * 8);
    hipMalloc(&out, 257 * 8);
    hipMemset(rows, 0, 512 * 257 * 8);
    printf("empty 1x64        : %6.2f us (kernel span, CP start/stop events)\n",
           span_us([&](hipEvent_t a, hipEvent_t b) { hipExtLaunchKernelGGL(empty_k, dim3(1), dim3(64), 0, s, a, b, 0); }, s));
    printf("empty 512x512     : %6.2f us\n",
           span_us([&](hipEvent_t a, hipEvent_t b) { hipExtLaunchKernelGGL(empty_k, dim3(512), dim3(512), 0, s, a, b, 0); }, s));
    printf("fold 33x512, 1 MB : %6.2f us\n",
           span_us([&](hipEvent_t a, hipEvent_t b) { hipExtLaunchKernelGGL(fold_k, dim3(33), dim3(512), 0, s, a, b, 0, rows, 512, 257, out); }, s));
    for (double mb : {6.4, 32.0, 64.0, 72.0, 96.0}) {
        const long n = (long)(mb * 1e6 / 2 / 16);
        double2 *a, *b;
        hipMalloc(&a, n * 16);
        hipMalloc(&b, n * 16);
        hipMemset(a, 1, n * 16);
        for (int grid : {512, 1024, 2048}) {
            const double us = span_us([&](hipEvent_t x, hipEvent_t y) { hipExtLaunchKernelGGL(copy_k, dim3(grid), dim3(256), 0, s, x, y, 0, a, b, n); }, s, 100);
            printf("copy %5.1f MB grid %4d: %6.2f us = %5.2f TB/s\n", mb, grid, us, mb * 1e6 / us / 1e6);
        }
        hipFree(a);
        hipFree(b);
    }
    for (double mb : {32.0, 44.0, 88.0}) {   // read-only streams, cache-warm
        const long n = (long)(mb * 1e6 / 16);
        double2* a;
        double* o;
        hipMalloc(&a, n * 16);
        hipMalloc(&o, 4096 * 8);
        hipMemset(a, 0, n * 16);
        hipMemset(o, 0, 4096 * 8);
        for (int grid : {512, 1024, 2048}) {
            const double us = span_us([&](hipEvent_t x, hipEvent_t y) { hipExtLaunchKernelGGL(read_k, dim3(grid), dim3(256), 0, s, x, y, 0, a, o, n); }, s, 100);
            printf("read %5.1f MB grid %4d: %6.2f us = %5.2f TB/s\n", mb, grid, us, mb * 1e6 / us / 1e6);
        }
        hipFree(a);
        hipFree(o);
    }
    for (double mb : {44.0, 64.0}) {   // alternating direction between launches vs always forwards
        const long n = (long)(mb * 1e6 / 16);
        double2* a;
        double* o;
        hipMalloc(&a, n * 16);
        hipMalloc(&o, 4096 * 8);
        hipMemset(a, 0, n * 16);
        hipMemset(o, 0, 4096 * 8);
        for (int alt = 0; alt < 2; ++alt) {
            int launch = 0;
            const double us = span_us([&](hipEvent_t x, hipEvent_t y) {
                hipExtLaunchKernelGGL(read_dir_k, dim3(512), dim3(256), 0, s, x, y, 0, a, o, n, alt ? (launch++ & 1) : 0); }, s, 100);
            printf("read %5.1f MB, tile order %s: %6.2f us = %5.2f TB/s\n", mb, alt ? "alternating" : "always forwards", us, mb * 1e6 / us / 1e6);
        }
        hipFree(a);
        hipFree(o);
    }
    // HBM-resident copy: rotate over 6 buffer pairs of 72 MB (432 MB > the 256 MB Infinity Cache)
    {
        const long n = (long)(72e6 / 2 / 16);
        double2 *a[6], *b[6];
        for (int k = 0; k < 6; ++k) { hipMalloc(&a[k], n * 16); hipMalloc(&b[k], n * 16); hipMemset(a[k], 1, n * 16); }
        hipEvent_t e0, e1;
        hipEventCreate(&e0);
        hipEventCreate(&e1);
        double tot = 0;
        for (int i = 0; i < 66; ++i) {
            hipExtLaunchKernelGGL(copy_k, dim3(1024), dim3(256), 0, s, e0, e1, 0, a[i % 6], b[i % 6], n);
            hipStreamSynchronize(s);
            float ms = 0;
            hipEventElapsedTime(&ms, e0, e1);
            if (i >= 6) tot += ms;
        }
        printf("copy  72.0 MB HBM-resident (6 rotating buffer pairs): %6.2f us = %5.2f TB/s\n", 1e3 * tot / 60, 72.0 / (1e3 * tot / 60));
        for (int k = 0; k < 6; ++k) { hipFree(a[k]); hipFree(b[k]); }
    }
    // dependent pair per step, wall clock over 200 steps (what a sweep + fold step pays in boundaries)
    {
        hipDeviceSynchronize();
        hipEvent_t e0, e1;
        hipEventCreate(&e0);
        hipEventCreate(&e1);
        for (int pass = 0; pass < 2; ++pass) {
            hipEventRecord(e0, s);
            for (int i = 0; i < 200; ++i) {
                hipLaunchKernelGGL(empty_k, dim3(512), dim3(512), 0, s);
                hipLaunchKernelGGL(fold_k, dim3(33), dim3(512), 0, s, rows, 512, 257, out);
            }
            hipEventRecord(e1, s);
            hipStreamSynchronize(s);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            if (pass) printf("empty 512x512 + fold, back to back: %6.2f us per pair\n", 1e3 * ms / 200);
        }
    }
    return 0;
}
