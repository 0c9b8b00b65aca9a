// abi_rccl.cpp -- the north_star's collective at the C ABI: "RCCL all-reduce of Ψ and ∇g over xGMI per outer iteration".
// One process per GPU, pools sharded along the axis the reference threads over (src/router.jl:39); after every sweep's row
// fold the library enqueues, on the context's stream,
//     ncclAllReduce(d_out, d_out, n_tokens + 1, ncclDouble, ncclSum, comm, stream)
// so cfmm_find_arb / cfmm_eval / cfmm_route / cfmm_sweep_dev return the GLOBAL {Ψ, acc} on every rank -- the same contract
// as cfmm_set_peers (abi_peers.cpp: the library's own one-launch exchange, the fast path), with no torch, no Python, no IPC
// hand-rolling: a Julia / C host needs three calls (cfmm_rccl_unique_id on rank 0, its own broadcast of 128 bytes,
// cfmm_rccl_init_rank everywhere).
//
// RCCL is resolved at FIRST USE, not at load time (librccl.so is a 570 MB object; single-GPU users never touch it), and
// ALL entry points come from ONE library image -- a communicator must only ever meet the ncclAllReduce of the RCCL that
// created it:
//   1. CFMM_AMD_RCCL_LIB (environment: a path or soname) names the image -- the host that hands over its own communicator
//      (cfmm_set_rccl_comm) says which RCCL owns it; nothing else is tried, a failure is CFMM_ERR_UNSUPPORTED;
//   2. else the process's GLOBAL scope, if ncclAllReduce is visible there (a host linked against RCCL): all five symbols
//      must then resolve there;
//   3. else dlopen("librccl.so.1") (ROCm's soname: an image already loaded under that soname -- e.g. a framework's bundled
//      copy -- is the one returned), then /opt/rocm/lib/librccl.so.1.
// cfmm_set_rccl_comm accepts a foreign communicator only under 1. or 2. (under 3. the library cannot know that the caller's
// communicator belongs to the image it found by itself); cfmm_rccl_init_rank creates its communicator with the resolved
// image and works under all three.
#include "ctx.h"

#include <cstdlib>
#include <dlfcn.h>
#include <rccl/rccl.h>

using namespace cfmm;

namespace {

struct Rccl {
    ncclResult_t (*get_unique_id)(ncclUniqueId*) = nullptr;
    ncclResult_t (*comm_init_rank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*comm_destroy)(ncclComm_t) = nullptr;
    ncclResult_t (*all_reduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    const char* (*error_string)(ncclResult_t) = nullptr;
    std::string why;      // non-empty: resolution failed
    std::string source;   // "CFMM_AMD_RCCL_LIB=...", "global scope" or the soname / path the library opened by itself
    bool caller_named = false;   // resolved under rule 1 or 2: the caller's own RCCL
    bool ok = false;
};

Rccl& rccl()
{
    static Rccl r;
    static std::once_flag once;
    std::call_once(once, [] {
        void* h = nullptr;             // RTLD_DEFAULT is a null handle on glibc: `global` tells the two apart
        bool global = false;
        std::string tried;
        auto open = [&](const char* name) {
            h = dlopen(name, RTLD_NOW | RTLD_LOCAL);
            if (h) {
                r.source = name;
                return true;
            }
            const char* de = dlerror();        // ONE call: dlerror() clears the message it returns
            tried += std::string(tried.empty() ? "" : "; ") + name + ": " + (de ? de : "not loadable");
            return false;
        };
        const char* named = std::getenv("CFMM_AMD_RCCL_LIB");
        if (named && *named) {
            r.caller_named = true;
            if (open(named)) r.source = std::string("CFMM_AMD_RCCL_LIB=") + named;
        } else if (dlsym(RTLD_DEFAULT, "ncclAllReduce")) {
            global = true;
            r.caller_named = true;
            r.source = "global scope";
        } else if (!open("librccl.so.1")) {
            open("/opt/rocm/lib/librccl.so.1");
        }
        if (!h && !global) {
            r.why = "RCCL is not available (" + tried + ")";
            return;
        }
        auto sym = [&](const char* name) -> void* {
            void* p = dlsym(global ? RTLD_DEFAULT : h, name);
            if (!p && r.why.empty()) r.why = std::string("RCCL symbol ") + name + " not found in " + r.source;
            return p;
        };
        r.get_unique_id = reinterpret_cast<decltype(r.get_unique_id)>(sym("ncclGetUniqueId"));
        r.comm_init_rank = reinterpret_cast<decltype(r.comm_init_rank)>(sym("ncclCommInitRank"));
        r.comm_destroy = reinterpret_cast<decltype(r.comm_destroy)>(sym("ncclCommDestroy"));
        r.all_reduce = reinterpret_cast<decltype(r.all_reduce)>(sym("ncclAllReduce"));
        r.error_string = reinterpret_cast<decltype(r.error_string)>(sym("ncclGetErrorString"));
        r.ok = r.get_unique_id && r.comm_init_rank && r.comm_destroy && r.all_reduce && r.error_string;
    });
    return r;
}

} // namespace

namespace cfmm {

// enqueue_sweep: the all-reduce of {Ψ, acc} behind the fold, in-stream
int rccl_all_reduce_out(cfmm_ctx* c, double* d_out)
{
    Rccl& r = rccl();
    if (!r.ok) return fail(c, CFMM_ERR_UNSUPPORTED, "%s", r.why.c_str());
    const ncclResult_t e = r.all_reduce(d_out, d_out, (size_t)c->n + 1, ncclDouble, ncclSum, static_cast<ncclComm_t>(c->rccl_comm), c->stream);
    if (e != ncclSuccess) return fail(c, CFMM_ERR_HIP, "ncclAllReduce failed: %s", r.error_string(e));
    return CFMM_OK;
}

void rccl_release(cfmm_ctx* c)
{
    if (c->rccl_comm && c->rccl_owned) {
        (void)hipStreamSynchronize(c->stream);
        Rccl& r = rccl();
        if (r.ok) (void)r.comm_destroy(static_cast<ncclComm_t>(c->rccl_comm));
    }
    c->rccl_comm = nullptr;
    c->rccl_owned = false;
}

} // namespace cfmm

extern "C" {

int cfmm_rccl_unique_id(unsigned char id[CFMM_RCCL_ID_BYTES])
{
    static_assert(sizeof(ncclUniqueId) == CFMM_RCCL_ID_BYTES, "ncclUniqueId size");
    if (!id) return CFMM_ERR_INVALID_ARG;
    Rccl& r = rccl();
    if (!r.ok) return fail(nullptr, CFMM_ERR_UNSUPPORTED, "%s", r.why.c_str());
    ncclUniqueId u;
    const ncclResult_t e = r.get_unique_id(&u);
    if (e != ncclSuccess) return fail(nullptr, CFMM_ERR_HIP, "ncclGetUniqueId failed: %s", r.error_string(e));
    std::memcpy(id, &u, sizeof u);
    return CFMM_OK;
}

int cfmm_set_rccl_comm(cfmm_ctx* c, void* comm)
{
    if (!c) return CFMM_ERR_INVALID_ARG;
    CFMM_SINGLE_ONLY(c, "cfmm_set_rccl_comm");
    armed_cancel(c);
    if (comm && global_bins(c)) return fail(c, CFMM_ERR_UNSUPPORTED, "sharded operation is limited to n_tokens <= %d", kMaxLdsTokens);
    if (comm && !c->peers.empty())
        return fail(c, CFMM_ERR_STATE, "cfmm_set_peers is active on this context: one exchange at a time (cfmm_set_peers(ctx, NULL, 0, 0, 0) first)");
    if (comm && !rccl().ok) return fail(c, CFMM_ERR_UNSUPPORTED, "%s", rccl().why.c_str());
    if (comm && !rccl().caller_named)
        return fail(c, CFMM_ERR_UNSUPPORTED,
                    "cannot tell which RCCL owns this communicator: ncclAllReduce is not in the process's global scope and the library "
                    "resolved %s by itself -- name the owning image in CFMM_AMD_RCCL_LIB (or use cfmm_rccl_init_rank)", rccl().source.c_str());
    rccl_release(c);
    c->rccl_comm = comm;      // caller-owned (NULL: back to single-GPU operation)
    c->rccl_owned = false;
    c->have_out = false;
    return CFMM_OK;
}

int cfmm_rccl_init_rank(cfmm_ctx* c, const unsigned char id[CFMM_RCCL_ID_BYTES], int32_t world, int32_t rank)
{
    if (!c || !id) return CFMM_ERR_INVALID_ARG;
    CFMM_SINGLE_ONLY(c, "cfmm_rccl_init_rank");
    if (world < 1 || rank < 0 || rank >= world) return fail(c, CFMM_ERR_INVALID_ARG, "bad world / rank");
    if (global_bins(c)) return fail(c, CFMM_ERR_UNSUPPORTED, "sharded operation is limited to n_tokens <= %d", kMaxLdsTokens);
    if (!c->peers.empty()) return fail(c, CFMM_ERR_STATE, "cfmm_set_peers is active on this context: one exchange at a time");
    Rccl& r = rccl();
    if (!r.ok) return fail(c, CFMM_ERR_UNSUPPORTED, "%s", r.why.c_str());
    armed_cancel(c);
    HIP_TRY(c, hipSetDevice(c->device));
    ncclUniqueId u;
    std::memcpy(&u, id, sizeof u);
    ncclComm_t comm = nullptr;
    const ncclResult_t e = r.comm_init_rank(&comm, world, u, rank);
    if (e != ncclSuccess) return fail(c, CFMM_ERR_HIP, "ncclCommInitRank failed: %s", r.error_string(e));
    rccl_release(c);
    c->rccl_comm = comm;      // owned by the context: destroyed with it (or by cfmm_set_rccl_comm(ctx, NULL))
    c->rccl_owned = true;
    c->have_out = false;
    return CFMM_OK;
}

} // extern "C"
