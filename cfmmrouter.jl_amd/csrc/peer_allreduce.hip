// peer_allreduce.hip -- one-shot all-reduce(sum) of the n_tokens+1 doubles {Ψ, acc} across the
// GPUs of one node, over xGMI peer mappings (symmetric memory), for the sharded sweep.
//
// Why not RCCL for this: the message is 8·(n_tokens+1) bytes (2 KB at 256 tokens) -- pure latency.
// A ring/tree collective pays several hops of launch + protocol latency (tens of µs); here every
// rank publishes its vector once and reads the other N−1 vectors directly:
//     fold kernel writes {Ψ, acc} into data[parity] of this rank's symmetric buffer
//     lane 0: system-scope release, flag[parity] = seq                      (publish)
//     for p = 0..N−1 (FIXED order): wait flag_p[parity] >= seq, add data_p[parity][j]   (gather)
// The sum order is the rank order on every GPU, so all ranks obtain bit-identical Ψ (the lockstep
// L-BFGS-B of dist.py depends on that) and the result is reproducible.
// Double buffering by seq parity is enough: a rank rewrites data[parity] for seq+2 only after its
// seq+1 gather, which waited for every peer's seq+1 flag, i.e. for every peer to have finished
// reading seq.  Peer data is read with system-scope atomic loads (L2-bypassing): the same
// addresses were read two steps earlier and a cached copy would be stale.
// Every spin is bounded by wall-clock time (CFMM_AMD_PEER_TIMEOUT_S, default 30 s); on timeout the
// output is poisoned with NaN, which cfmm_abi.hip (host_sweep_end) and dist.py check.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include "../../include/cfmm_amd.h"

namespace {

constexpr int kMaxPeers = 16;

struct PeerArgs {
    const double* data[kMaxPeers];          // peer p's symmetric buffer: [2][count] doubles ...
    unsigned long long* flags[kMaxPeers];   // ... followed by 2 flags (one per parity)
    int world, rank;
    long long count;
    unsigned long long seq;
    double* out;
    long long timeout_ticks;   // wall_clock64() ticks (100 MHz) a rank waits for a peer before it gives up
};

__global__ __launch_bounds__(256) void peer_allreduce_kernel(PeerArgs a)
{
    __shared__ int ok;
    const int parity = (int)(a.seq & 1ull);
    if (threadIdx.x == 0) {
        __threadfence_system();   // release the fold kernel's stores to the other GPUs
        __hip_atomic_store(a.flags[a.rank] + parity, a.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        ok = 1;
    }
    __syncthreads();
    double acc[4] = {0.0, 0.0, 0.0, 0.0};   // up to 4 elements per lane per pass (count <= 1024 in one pass)
    for (long long base = 0; base < a.count; base += 4ll * blockDim.x) {
        for (int k = 0; k < 4; ++k) acc[k] = 0.0;
        for (int p = 0; p < a.world; ++p) {
            if (base == 0) {   // wait once per peer
                if (threadIdx.x == 0) {
                    const long long t0 = (long long)wall_clock64();
                    while (__hip_atomic_load(a.flags[p] + parity, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) < a.seq) {
                        __builtin_amdgcn_s_sleep(2);
                        if ((long long)wall_clock64() - t0 > a.timeout_ticks) { ok = 0; break; }
                    }
                }
                __syncthreads();
            }
            const double* src = a.data[p] + (long long)parity * a.count;
            for (int k = 0; k < 4; ++k) {
                const long long j = base + (long long)k * blockDim.x + threadIdx.x;
                if (j < a.count)
                    acc[k] += __hip_atomic_load(src + j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            }
        }
        for (int k = 0; k < 4; ++k) {
            const long long j = base + (long long)k * blockDim.x + threadIdx.x;
            if (j < a.count) a.out[j] = ok ? acc[k] : __builtin_nan("");
        }
    }
}

} // namespace

extern "C" int cfmm_peer_allreduce(void* hip_stream, const uint64_t* peer_buffers, int32_t world, int32_t rank,
                                   int64_t count, uint64_t seq, double* d_out)
{
    if (!peer_buffers || !d_out || world < 1 || world > kMaxPeers || rank < 0 || rank >= world || count < 1)
        return CFMM_ERR_INVALID_ARG;
    PeerArgs a;
    for (int p = 0; p < world; ++p) {
        a.data[p] = reinterpret_cast<const double*>(peer_buffers[p]);
        a.flags[p] = reinterpret_cast<unsigned long long*>(peer_buffers[p] + (uint64_t)(2 * count) * sizeof(double));
    }
    a.world = world;
    a.rank = rank;
    a.count = count;
    a.seq = seq;
    a.out = d_out;
    // Plain rank skew (a GC pause, a pool reload, a trade download on one rank) must not become a
    // failure: wait up to CFMM_AMD_PEER_TIMEOUT_S seconds (default 30) of wall clock, then give up
    // (NaN output, which every caller checks).
    static const double timeout_s = [] {
        const char* e = getenv("CFMM_AMD_PEER_TIMEOUT_S");
        const double v = e ? atof(e) : 0.0;
        return v > 0.0 ? v : 30.0;
    }();
    a.timeout_ticks = (long long)(timeout_s * 1e8);
    hipLaunchKernelGGL(peer_allreduce_kernel, dim3(1), dim3(256), 0, static_cast<hipStream_t>(hip_stream), a);
    return hipGetLastError() == hipSuccess ? CFMM_OK : CFMM_ERR_HIP;
}
