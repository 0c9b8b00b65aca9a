// abi_multi.cpp -- single-process multi-device context (cfmm_ctx_create_multi).
// The parent context owns one ordinary single-device context per shard.  Pools of every batch are
// split into contiguous blocks over the shards (no pool is replicated); a host-pointer sweep stages
// the same v on every device, runs the shard sweeps concurrently (one host worker thread per shard,
// or -- option "multi_threads" = 0 -- all launches from the calling thread, then all waits) and sums
// the shards' {Ψ, acc} ON THE HOST in shard order: v comes from the host and Ψ returns to it on
// every evaluation anyway, so the "all-reduce" of SURVEY 8e degenerates to N·(n+1) additions --
// no peer access, no IPC, no torch.  One L-BFGS-B (cfmm_route) drives all shards; its evaluations are pre-armed on
// every shard when the shards sit on distinct devices (abi_sweep.cpp, armed_eval).
#include "ctx.h"

#include <algorithm>

using namespace cfmm;

namespace {

void worker_main(cfmm_ctx* parent, int d)
{
    Workers& w = *parent->workers;
    cfmm_ctx* child = parent->shards[(size_t)d];
    (void)hipSetDevice(child->device);
    uint64_t seen = 0;
    for (;;) {
        int spins = 0;
        while (w.go.load(std::memory_order_acquire) == seen && !w.quit.load(std::memory_order_relaxed)) {
            if (++spins < 8000) { __builtin_ia32_pause(); continue; }   // ~0.1 ms of spinning (an evaluation follows the previous one within tens of us), then sleep
            std::unique_lock<std::mutex> lk(w.mu);
            w.sleepers.fetch_add(1);
            w.cv.wait(lk, [&] { return w.go.load(std::memory_order_acquire) != seen || w.quit.load(); });
            w.sleepers.fetch_sub(1);
        }
        if (w.quit.load()) return;
        seen = w.go.load(std::memory_order_acquire);
        w.rc[(size_t)d] = single_host_sweep(child, w.v, w.materialize);
        w.pending.fetch_sub(1, std::memory_order_release);
    }
}

void pop_last_segment(cfmm_ctx* child)
{
    (void)hipSetDevice(child->device);
    (void)hipStreamSynchronize(child->stream);
    free_segment(child->segs.back());
    child->segs.pop_back();
    child->geometry_dirty = true;
    child->have_out = child->have_trades = false;
}

} // namespace

namespace cfmm {

void shard_range(int64_t m, int d, int nd, int64_t& lo, int64_t& hi)
{
    const int64_t base = m / nd, rem = m % nd;
    lo = d * base + std::min<int64_t>(d, rem);
    hi = lo + base + (d < rem ? 1 : 0);
}

int multi_host_sweep(cfmm_ctx* c, const double* v, bool materialize)
{
    const int nd = (int)c->shards.size();
    std::vector<int>& rcs = c->workers->rc;
    std::fill(rcs.begin(), rcs.end(), CFMM_OK);
    if (c->opt_multi_threads != 0 && nd > 1) {
        Workers& w = *c->workers;
        if (w.threads.empty())
            for (int d = 1; d < nd; ++d) w.threads.emplace_back(worker_main, c, d);
        w.v = v;
        w.materialize = materialize;
        w.pending.store(nd - 1, std::memory_order_relaxed);
        w.go.fetch_add(1, std::memory_order_release);
        if (w.sleepers.load() > 0) {
            std::lock_guard<std::mutex> lk(w.mu);
            w.cv.notify_all();
        }
        rcs[0] = single_host_sweep(c->shards[0], v, materialize);
        while (w.pending.load(std::memory_order_acquire) > 0) __builtin_ia32_pause();
    } else {
        for (int d = 0; d < nd; ++d) rcs[(size_t)d] = host_sweep_begin(c->shards[(size_t)d], v, materialize);
        for (int d = 0; d < nd; ++d)
            if (rcs[(size_t)d] == CFMM_OK) rcs[(size_t)d] = host_sweep_end(c->shards[(size_t)d]);
    }
    for (int d = 0; d < nd; ++d)
        if (rcs[(size_t)d] != CFMM_OK) {
            c->have_out = false;
            return fail(c, rcs[(size_t)d], "shard %d (device %d): %s", d, c->shards[(size_t)d]->device,
                        c->shards[(size_t)d]->err.c_str());
        }
    // the all-reduce: shard order, on the host
    c->last_out.assign((size_t)c->n + 1, 0.0);
    for (int d = 0; d < nd; ++d) {
        const std::vector<double>& o = c->shards[(size_t)d]->last_out;
        for (int j = 0; j <= c->n; ++j) c->last_out[(size_t)j] += o[(size_t)j];
    }
    c->have_out = true;
    c->have_trades = materialize;
    return CFMM_OK;
}

// Run add(child, lo, hi) on every shard with a non-empty block [lo, hi) of the m pools; all or nothing.
int multi_add(cfmm_ctx* c, int kind, int64_t m, const std::function<int(cfmm_ctx*, int64_t, int64_t)>& add)
{
    if (m < 0) return fail(c, CFMM_ERR_INVALID_ARG, "negative pool count");
    if (m == 0) return CFMM_OK;
    const int nd = (int)c->shards.size();
    std::vector<int> added;
    for (int d = 0; d < nd; ++d) {
        int64_t lo, hi;
        shard_range(m, d, nd, lo, hi);
        if (hi == lo) continue;
        cfmm_ctx* child = c->shards[(size_t)d];
        const int rc = add(child, lo, hi);
        if (rc != CFMM_OK) {
            fail(c, rc, "pools [%lld, %lld) -> shard %d (device %d): %s", (long long)lo, (long long)hi, d, child->device,
                 child->err.c_str());
            for (int a : added) pop_last_segment(c->shards[(size_t)a]);
            return rc;
        }
        added.push_back(d);
    }
    c->psegs.push_back({kind, m, c->m_total});
    c->m_total += m;
    c->have_out = c->have_trades = false;
    return CFMM_OK;
}

// child segment index that holds shard d's block of parent segment `pseg` (-1: that block is empty)
int child_segment(const cfmm_ctx* c, int pseg, int d)
{
    const int nd = (int)c->shards.size();
    int idx = 0;
    for (int k = 0; k <= pseg; ++k) {
        int64_t lo, hi;
        shard_range(c->psegs[(size_t)k].m, d, nd, lo, hi);
        if (k == pseg) return hi > lo ? idx : -1;
        if (hi > lo) ++idx;
    }
    return -1;
}

int multi_get_trades_range(cfmm_ctx* c, int32_t seg, int64_t first, int64_t count, double* Delta, double* Lambda)
{
    if (!c->have_trades) return fail(c, CFMM_ERR_STATE, "no materialised trades: call cfmm_find_arb first");
    if (seg < 0 || seg >= (int32_t)c->psegs.size()) return fail(c, CFMM_ERR_INVALID_ARG, "segment out of range");
    const int64_t m = c->psegs[(size_t)seg].m;
    if (first < 0 || count < 0 || first + count > m) return fail(c, CFMM_ERR_INVALID_ARG, "row range out of bounds");
    const int nd = (int)c->shards.size();
    for (int d = 0; d < nd; ++d) {
        int64_t lo, hi;
        shard_range(m, d, nd, lo, hi);
        const int64_t a = std::max(lo, first), b = std::min(hi, first + count);
        if (b <= a) continue;
        cfmm_ctx* child = c->shards[(size_t)d];
        const int rc = cfmm_get_trades_range(child, child_segment(c, seg, d), a - lo, b - a,
                                             Delta ? Delta + 2 * (a - first) : nullptr,
                                             Lambda ? Lambda + 2 * (a - first) : nullptr);
        if (rc != CFMM_OK) return fail(c, rc, "shard %d: %s", d, child->err.c_str());
    }
    return CFMM_OK;
}

} // namespace cfmm
