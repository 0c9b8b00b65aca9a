// streams.hip -- what does the MEMORY side of a ProductTwoCoin-shaped sweep cost when the pool state comes from HBM?
// Per pool: read 16 B (reserve pair) + 8 B (packed record), write one 16-byte trade record; 256 blocks x 1024 threads,
// block-strided tiles -- the geometry of the real launch (csrc/sweep_kernels.hip) -- with a dependent FMA chain standing in for
// the closed form.  The steps rotate over enough copies that the bytes touched are >= 2 x the 256 MiB Infinity Cache.
// Variants: tiles in flight per lane (1 = the plain loop, 2 = next-tile prefetch, 4), SoA arrays vs one contiguous block per
// tile (AoSoA), trade stores write-through (sc1) / non-temporal / none.
// build: hipcc --offload-arch=gfx950 -O3 -o scripts/native/streams.bin scripts/native/streams.hip
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
#include <vector>

typedef double d2v __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void store_sc1(double2* dst, double x, double y)
{
    d2v v = {x, y};
    asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" : : "v"(dst), "v"(v) : "memory");
}
__device__ __forceinline__ void store_nt(double2* dst, double x, double y)
{
    d2v v = {x, y};
    asm volatile("global_store_dwordx4 %0, %1, off nt\n\ts_nop 1" : : "v"(dst), "v"(v) : "memory");
}

struct Raw { double2 R; uint2 pk; };

// LAYOUT 0: R[i], pk[i] in two arrays.  LAYOUT 1: tile t (1024 pools) = 16 KB of R then 8 KB of pk, contiguous.
template <int LAYOUT>
__device__ __forceinline__ Raw load(const char* base, long m, long i)
{
    Raw r;
    if (LAYOUT == 0) {
        r.R = reinterpret_cast<const double2*>(base)[i];
        r.pk = reinterpret_cast<const uint2*>(base + m * 16)[i];
    } else {
        const long t = i >> 10, l = i & 1023;
        const char* tile = base + t * (1024 * 24);
        r.R = reinterpret_cast<const double2*>(tile)[l];
        r.pk = reinterpret_cast<const uint2*>(tile + 1024 * 16)[l];
    }
    return r;
}

template <int DEPTH, int LAYOUT, int STORE, int WORK>
__global__ __launch_bounds__(1024) void sweep_like(const char* __restrict__ pools, double2* __restrict__ trades, long m, double* __restrict__ out)
{
    const long stride = (long)gridDim.x * 1024;
    long i = (long)blockIdx.x * 1024 + threadIdx.x;
    long left = i < m ? (m - i + stride - 1) / stride : 0;
    Raw q[DEPTH];
#pragma unroll
    for (int d = 0; d < DEPTH; ++d)
        if (d < left) q[d] = load<LAYOUT>(pools, m, i + d * stride);
    double acc = 0.0;
    while (left > 0) {
        const Raw cur = q[0];
#pragma unroll
        for (int d = 0; d + 1 < DEPTH; ++d) q[d] = q[d + 1];
        if (left > DEPTH - 1 && DEPTH > 0) {
            if (left > DEPTH) q[DEPTH - 1] = load<LAYOUT>(pools, m, i + DEPTH * stride);
        }
        double x = cur.R.x, y = cur.R.y + (double)cur.pk.x * 1e-9;
#pragma unroll 8
        for (int k = 0; k < WORK; ++k) x = __builtin_fma(x, 0.999999, y * 1e-6);   // dependent chain: stands in for the closed form
        if (STORE == 1) store_sc1(trades + i, x, y);
        else if (STORE == 2) store_nt(trades + i, x, y);
        acc += x;
        i += stride;
        --left;
    }
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
    if ((threadIdx.x & 63) == 0) atomicAdd(out + blockIdx.x, acc);
}

template <int DEPTH, int LAYOUT, int STORE, int WORK>
static void run(const char* label, std::vector<char*>& pools, std::vector<double2*>& trades, long m, double* out, hipStream_t s)
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const int copies = (int)pools.size();
    double tot = 0, tmin = 1e9;
    const int reps = 3 * copies;
    for (int k = 0; k < reps + copies; ++k) {
        hipExtLaunchKernelGGL((sweep_like<DEPTH, LAYOUT, STORE, WORK>), dim3(256), dim3(1024), 0, s, e0, e1, 0, pools[k % copies],
                              trades[k % copies], m, out);
        hipStreamSynchronize(s);
        float ms = 0;
        hipEventElapsedTime(&ms, e0, e1);
        if (k >= copies) { tot += ms; tmin = ms < tmin ? ms : tmin; }
    }
    const double us = 1e3 * tot / reps, bytes = m * (24.0 + (STORE ? 16.0 : 0.0));
    printf("%-64s %6.2f us (min %6.2f)  %5.2f TB/s of %4.1f MB\n", label, us, 1e3 * tmin, bytes / us / 1e6, bytes / 1e6);
    fflush(stdout);
}

int main(int argc, char** argv)
{
    const long m = argc > 1 ? atol(argv[1]) : 1000000;
    const int copies = (int)(2.0 * 268435456.0 / (m * 40.0)) + 2;
    hipStream_t s;
    hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    std::vector<char*> pools(copies);
    std::vector<double2*> trades(copies);
    const long mpad = (m + 1023) / 1024 * 1024;
    for (int k = 0; k < copies; ++k) {
        hipMalloc(&pools[k], mpad * 24);
        hipMalloc(&trades[k], mpad * 16);
        hipMemset(pools[k], 0, mpad * 24);
    }
    double* out;
    hipMalloc(&out, 4096 * 8);
    hipMemset(out, 0, 4096 * 8);
    printf("# %ld pools, %d copies = %.0f MB touched per rotation (>= 2 x 256 MiB); 256 blocks x 1024 threads; kernel span by CP events\n", m, copies,
           copies * m * 40.0 / 1e6);
    run<1, 0, 1, 64>("SoA, 1 tile in flight, sc1 stores, 64-fma chain", pools, trades, m, out, s);
    run<2, 0, 1, 64>("SoA, 2 tiles in flight (prefetch), sc1 stores, 64-fma chain", pools, trades, m, out, s);
    run<4, 0, 1, 64>("SoA, 4 tiles in flight, sc1 stores, 64-fma chain", pools, trades, m, out, s);
    run<1, 1, 1, 64>("AoSoA (24 KB per tile), 1 tile in flight, sc1 stores", pools, trades, m, out, s);
    run<2, 1, 1, 64>("AoSoA, 2 tiles in flight, sc1 stores", pools, trades, m, out, s);
    run<2, 0, 2, 64>("SoA, 2 tiles in flight, NON-TEMPORAL stores", pools, trades, m, out, s);
    run<4, 0, 2, 64>("SoA, 4 tiles in flight, NON-TEMPORAL stores", pools, trades, m, out, s);
    run<2, 0, 0, 64>("SoA, 2 tiles in flight, no stores (fused evaluation)", pools, trades, m, out, s);
    run<4, 0, 0, 64>("SoA, 4 tiles in flight, no stores", pools, trades, m, out, s);
    run<2, 0, 1, 0>("SoA, 2 tiles in flight, sc1 stores, NO arithmetic", pools, trades, m, out, s);
    run<2, 0, 1, 256>("SoA, 2 tiles in flight, sc1 stores, 256-fma chain", pools, trades, m, out, s);
    // cache-warm reference: the same copy every launch
    {
        std::vector<char*> p1(1, pools[0]);
        std::vector<double2*> t1(1, trades[0]);
        run<2, 0, 1, 64>("cache-warm (one copy): SoA, 2 tiles in flight, sc1 stores", p1, t1, m, out, s);
        run<2, 0, 2, 64>("cache-warm (one copy): SoA, 2 tiles in flight, NT stores", p1, t1, m, out, s);
    }
    return 0;
}
