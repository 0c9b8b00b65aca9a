// abi_peers.cpp -- one process per GPU: rank-to-rank buffers and cfmm_set_peers (include/cfmm_amd.h).
// The exchange itself is reduce_gather (sweep_kernels.hip): the launch that folds a rank's partial rows publishes
// its {Ψ, acc} as self-validating granules in the rank's buffer and reads every peer's over xGMI.
#include "ctx.h"

#include <cstdlib>
#include <cstring>

using namespace cfmm;

extern "C" {

int cfmm_set_peers(cfmm_ctx* c, const uint64_t* peer_buffers, int32_t world, int32_t rank, uint64_t seq)
{
    if (!c) return CFMM_ERR_INVALID_ARG;
    CFMM_SINGLE_ONLY(c, "cfmm_set_peers");
    armed_cancel(c);
    if (world == 0) { // back to single-GPU operation
        c->peers.clear();
        return CFMM_OK;
    }
    if (c->rccl_comm) return fail(c, CFMM_ERR_STATE, "an RCCL communicator is active on this context: one exchange at a time (cfmm_set_rccl_comm(ctx, NULL) first)");
    if (!peer_buffers || world < 1 || world > kMaxPeers || rank < 0 || rank >= world)
        return fail(c, CFMM_ERR_INVALID_ARG, "bad peer configuration");
    if (global_bins(c)) return fail(c, CFMM_ERR_UNSUPPORTED, "sharded operation is limited to n_tokens <= %d", kMaxLdsTokens);
    c->peers.assign(peer_buffers, peer_buffers + world);
    c->peer_rank = rank;
    c->peer_seq = seq;
    if (const char* e = getenv("CFMM_AMD_PEER_TIMEOUT_S")) {
        const double t = atof(e);
        if (t > 0.0) c->peer_timeout_ticks = (long long)(t * 1e8);
    }
    return CFMM_OK;
}

int64_t cfmm_peer_buffer_bytes(int32_t n_tokens) { return (int64_t)4 * ((int64_t)n_tokens + 1) * (int64_t)sizeof(uint64_t); }

int cfmm_peer_buffer_alloc(cfmm_ctx* c, uint64_t* d_buf, unsigned char handle[CFMM_IPC_HANDLE_BYTES])
{
    if (!c || !d_buf || !handle) return CFMM_ERR_INVALID_ARG;
    CFMM_SINGLE_ONLY(c, "cfmm_peer_buffer_alloc");
    static_assert(sizeof(hipIpcMemHandle_t) == CFMM_IPC_HANDLE_BYTES, "IPC handle size");
    HIP_TRY(c, hipSetDevice(c->device));
    const size_t bytes = (size_t)cfmm_peer_buffer_bytes(c->n);
    // Fine-grained device memory: stores from this GPU become visible to a peer's system-scope loads over xGMI without
    // waiting for a kernel boundary (coarse-grained allocations only guarantee that at the end of the writing kernel).
    // Falls back to a plain allocation if the fine-grained one cannot be exported (the bench's start-up self-check then
    // decides between this path and RCCL).
    void* p = nullptr;
    hipError_t e = hipExtMallocWithFlags(&p, bytes, hipDeviceMallocFinegrained);
    hipIpcMemHandle_t h;
    if (e == hipSuccess) e = hipMemset(p, 0, bytes);
    if (e == hipSuccess) e = hipDeviceSynchronize();
    if (e == hipSuccess) e = hipIpcGetMemHandle(&h, p);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        if (p) (void)hipFree(p);
        p = nullptr;
        HIP_TRY(c, hipMalloc(&p, bytes));
        e = hipMemset(p, 0, bytes);
        if (e == hipSuccess) e = hipDeviceSynchronize();
        if (e == hipSuccess) e = hipIpcGetMemHandle(&h, p);
        if (e != hipSuccess) {
            (void)hipFree(p);
            return fail(c, CFMM_ERR_HIP, "peer buffer export failed: %s (HSA_ENABLE_IPC_MODE_LEGACY=0 set?)", hipGetErrorString(e));
        }
    }
    std::memcpy(handle, &h, sizeof h);
    *d_buf = reinterpret_cast<uint64_t>(p);
    return CFMM_OK;
}

int cfmm_peer_buffer_open(cfmm_ctx* c, const unsigned char handle[CFMM_IPC_HANDLE_BYTES], uint64_t* d_peer)
{
    if (!c || !d_peer || !handle) return CFMM_ERR_INVALID_ARG;
    CFMM_SINGLE_ONLY(c, "cfmm_peer_buffer_open");
    HIP_TRY(c, hipSetDevice(c->device));
    hipIpcMemHandle_t h;
    std::memcpy(&h, handle, sizeof h);
    void* p = nullptr;
    HIP_TRY(c, hipIpcOpenMemHandle(&p, h, hipIpcMemLazyEnablePeerAccess));
    *d_peer = reinterpret_cast<uint64_t>(p);
    return CFMM_OK;
}

int cfmm_peer_buffer_close(cfmm_ctx* c, uint64_t d_peer)
{
    if (!c) return CFMM_ERR_INVALID_ARG;
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    HIP_TRY(c, hipIpcCloseMemHandle(reinterpret_cast<void*>(d_peer)));
    return CFMM_OK;
}

int cfmm_peer_buffer_free(cfmm_ctx* c, uint64_t d_buf)
{
    if (!c) return CFMM_ERR_INVALID_ARG;
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    HIP_TRY(c, hipFree(reinterpret_cast<void*>(d_buf)));
    return CFMM_OK;
}

} // extern "C"
