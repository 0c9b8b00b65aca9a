// gran.hip -- at what GRANULARITY does a scattered read reach the fabric / HBM on gfx950, and how does rocprofv3's
// FETCH_SIZE tally it?  (MI355X_MICROARCH.md calibrates FETCH_SIZE only for wide coalesced reads: x2.)  The multi-tick UniV3
// walk reads ONE 64-byte landing record per walking pool at an address no neighbouring lane shares; whether that costs
// 64 or 128 bytes of HBM traffic decides what its traffic floor is (UniV3 section).
// Every lane reads BYTES bytes (16 .. 128) at the start of a pseudo-randomly chosen 128-byte-aligned slot of a 4 GiB buffer
// (HALF = 1: at the slot's second 64-byte half for odd lanes), with the cache-policy bits FLAGS on the load; no slot is
// visited twice, so nothing is served by a cache.  Also a coalesced read of the same byte count as the calibration point.
// build: hipcc --offload-arch=gfx950 -O3 -o scripts/native/gran.bin scripts/native/gran.hip
// run:   scripts/native/gran.bin            (times)      rocprofv3 --pmc FETCH_SIZE --kernel-trace ... -- scripts/native/gran.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef double d2v __attribute__((ext_vector_type(2)));
template <int FLAGS>
__device__ __forceinline__ d2v load16(const char* p)
{
    d2v v;
    if (FLAGS == 0) asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(v) : "v"(p) : "memory");
    if (FLAGS == 1) asm volatile("global_load_dwordx4 %0, %1, off nt" : "=v"(v) : "v"(p) : "memory");
    if (FLAGS == 2) asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(v) : "v"(p) : "memory");
    if (FLAGS == 3) asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1" : "=v"(v) : "v"(p) : "memory");
    if (FLAGS == 4) asm volatile("global_load_dwordx4 %0, %1, off sc0" : "=v"(v) : "v"(p) : "memory");
    if (FLAGS == 5) asm volatile("global_load_dwordx4 %0, %1, off sc1 nt" : "=v"(v) : "v"(p) : "memory");
    return v;
}

// slot of lane t: a bijection on [0, 2^bits) (odd multiplier, xor-shift), so no slot is read twice
__device__ __forceinline__ unsigned long slot_of(unsigned long t, int bits)
{
    unsigned long x = (t * 0x9E3779B97F4A7C15ul) >> (64 - bits);
    return x;
}

template <int BYTES, int FLAGS, int HALF>
__global__ __launch_bounds__(256) void scattered(const char* __restrict__ buf, int bits, double* __restrict__ out, long lanes, long first)
{
    const long t = (long)blockIdx.x * 256 + threadIdx.x;
    if (t >= lanes) return;
    const char* p = buf + slot_of((unsigned long)(t + first), bits) * 128 + ((HALF && (t & 1)) ? 64 : 0);
    d2v s = {0.0, 0.0};
    d2v v[BYTES / 16];
#pragma unroll
    for (int k = 0; k < BYTES / 16; ++k) v[k] = load16<FLAGS>(p + 16 * k);
#pragma unroll
    for (int k = 0; k < BYTES / 16; ++k) asm volatile("s_waitcnt vmcnt(0)" : "+v"(v[k]) : : "memory");   // the loads are asynchronous: nothing reads v[k] before this
#pragma unroll
    for (int k = 0; k < BYTES / 16; ++k) s += v[k];
    out[t] = s.x + s.y;
}

// The same 64-byte records fetched COOPERATIVELY: instruction k (of 4) fetches the records of lanes 16k .. 16k+15, four lanes
// reading the four 16-byte quarters of one record -- 16 distinct lines per load instruction instead of 64 -- and the quarters
// travel to the owning lane through ds_bpermute (REDIST = 1) or are summed where they are (REDIST = 0: the fetch alone).
template <int REDIST>
__global__ __launch_bounds__(256) void cooperative(const char* __restrict__ buf, int bits, double* __restrict__ out, long lanes, long first)
{
    const long t = (long)blockIdx.x * 256 + threadIdx.x;
    if (t >= lanes) return;
    const int lane = threadIdx.x & 63;
    const unsigned long my_slot = slot_of((unsigned long)(t + first), bits);
    d2v v[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int owner = 16 * k + (lane >> 2);
        const unsigned long s = __shfl(my_slot, owner, 64);
        v[k] = load16<0>(buf + s * 128 + 16 * (lane & 3));
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) asm volatile("s_waitcnt vmcnt(0)" : "+v"(v[k]) : : "memory");
    d2v s = {0.0, 0.0};
    if (REDIST) {
        // owner lane q = 16k + j receives quarter c of its record from lane 4j + c of instruction k
        const int k_mine = lane >> 4, j = lane & 15;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            d2v got = {0.0, 0.0};
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const double x = __shfl(v[k].x, 4 * j + c, 64), y = __shfl(v[k].y, 4 * j + c, 64);
                if (k == k_mine) got = d2v{x, y};
            }
            s += got;
        }
    } else {
#pragma unroll
        for (int k = 0; k < 4; ++k) s += v[k];
    }
    out[t] = s.x + s.y;
}

template <int BYTES>
__global__ __launch_bounds__(256) void coalesced(const char* __restrict__ buf, double* __restrict__ out, long lanes, long first)
{
    const long t = (long)blockIdx.x * 256 + threadIdx.x;
    if (t >= lanes) return;
    d2v s = {0.0, 0.0};
#pragma unroll
    for (int k = 0; k < BYTES / 16; ++k) s += *reinterpret_cast<const d2v*>(buf + ((first + t) + (long)k * lanes) * 16);
    out[t] = s.x + s.y;
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

int main()
{
    const int bits = 25;                        // 2^25 slots x 128 B = 4 GiB
    const long lanes = 1l << 22;                // 4M lanes per launch
    char* buf;
    double* out;
    CK(hipMalloc(&buf, (size_t)128 << bits));
    CK(hipMalloc(&out, lanes * 8));
    CK(hipMemset(buf, 1, (size_t)128 << bits));
    hipEvent_t a, b;
    CK(hipEventCreate(&a));
    CK(hipEventCreate(&b));
    long first = 0;
    auto timeit = [&](const char* name, auto launch, double bytes_useful) {
        float best = 1e30f, ms;
        for (int rep = 0; rep < 3; ++rep) {     // every launch visits slots no earlier launch has touched (2^25 / 2^22 = 8 windows; the
            (void)hipEventRecord(a);                 // 4 GiB written by the memset left the caches long ago)
            launch(first);
            (void)hipEventRecord(b);
            (void)hipEventSynchronize(b);
            (void)hipEventElapsedTime(&ms, a, b);
            if (ms < best) best = ms;
            first = (first + lanes) & ((1l << bits) - 1);
        }
        printf("%-34s %8.1f us   useful %6.1f MB -> %6.2f TB/s useful   (as 128 B lines: %6.2f TB/s)\n", name, best * 1e3, bytes_useful / 1e6,
               bytes_useful / (best * 1e-3) / 1e12, 128.0 * lanes / (best * 1e-3) / 1e12);
    };
    dim3 g((unsigned)(lanes / 256)), bl(256);
#define SC(B, F, H) timeit("scattered<" #B "," #F "," #H ">", [&](long f) { hipLaunchKernelGGL((scattered<B, F, H>), g, bl, 0, 0, buf, bits, out, lanes, f); }, (double)B * lanes)
    SC(16, 0, 0); SC(32, 0, 0); SC(64, 0, 0); SC(128, 0, 0); SC(64, 0, 1);
    SC(64, 1, 0); SC(64, 2, 0); SC(64, 3, 0); SC(64, 4, 0); SC(64, 5, 0);
    SC(128, 3, 0); SC(48, 0, 0);
    timeit("cooperative<fetch only>", [&](long f) { hipLaunchKernelGGL((cooperative<0>), g, bl, 0, 0, buf, bits, out, lanes, f); }, 64.0 * lanes);
    timeit("cooperative<fetch + bpermute>", [&](long f) { hipLaunchKernelGGL((cooperative<1>), g, bl, 0, 0, buf, bits, out, lanes, f); }, 64.0 * lanes);
    timeit("coalesced<64>", [&](long f) { hipLaunchKernelGGL((coalesced<64>), g, bl, 0, 0, buf, out, lanes, (f * 4) & ((1l << 27) - 1)); }, 64.0 * lanes);
    timeit("coalesced<128>", [&](long f) { hipLaunchKernelGGL((coalesced<128>), g, bl, 0, 0, buf, out, lanes, (f * 8) & ((1l << 27) - 1)); }, 128.0 * lanes);
    CK(hipDeviceSynchronize());
    return 0;
}
