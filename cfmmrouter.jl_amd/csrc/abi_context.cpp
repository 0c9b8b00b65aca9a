// abi_context.cpp -- context life cycle, options, streams, introspection (include/cfmm_amd.h).
#include "ctx.h"

#include <cstdlib>
#include <cstring>

using namespace cfmm;

namespace cfmm {

thread_local std::string g_create_error = "";

// Kernel arguments in device memory: measured 22.2 vs 24.9 us per config-3 step and 7.0 vs 9.1 us per config-2
// step against host-memory kernargs (r01).  The HIP runtime reads the variable when it initialises, so it is
// set when this library is loaded -- unless the caller has decided otherwise (an existing value is kept).
__attribute__((constructor)) static void cfmm_default_environment() { setenv("HIP_FORCE_DEV_KERNARG", "1", 0); }

int fail(const cfmm_ctx* c, int code, const char* fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if (c) c->err = buf;
    else g_create_error = buf;
    return code;
}

} // namespace cfmm

extern "C" {

const char* cfmm_version(void) { return "cfmm_amd 0.6.0 (gfx950)"; }

const char* cfmm_last_error(const cfmm_ctx* ctx) { return ctx ? ctx->err.c_str() : g_create_error.c_str(); }

int cfmm_ctx_create(int device_id, int32_t n_tokens, cfmm_ctx** out)
{
    if (!out) return fail(nullptr, CFMM_ERR_INVALID_ARG, "out is null");
    *out = nullptr;
    if (n_tokens < 1) return fail(nullptr, CFMM_ERR_INVALID_ARG, "n_tokens must be >= 1");
    if (n_tokens > (1 << 26))
        return fail(nullptr, CFMM_ERR_UNSUPPORTED, "n_tokens %d exceeds the supported maximum %d", n_tokens, 1 << 26);
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess || count <= 0)
        return fail(nullptr, CFMM_ERR_HIP, "no HIP device available (%s); this library has no CPU fallback",
                    e == hipSuccess ? "device count is 0" : hipGetErrorString(e));
    if (device_id < 0 || device_id >= count)
        return fail(nullptr, CFMM_ERR_INVALID_ARG, "device_id %d out of range [0, %d)", device_id, count);
    hipDeviceProp_t prop;
    HIP_TRY(nullptr, hipGetDeviceProperties(&prop, device_id));
    if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return fail(nullptr, CFMM_ERR_UNSUPPORTED, "device %d is %s; this library is built for gfx950 only", device_id,
                    prop.gcnArchName);
    HIP_TRY(nullptr, hipSetDevice(device_id));

    cfmm_ctx* c = new cfmm_ctx();
    c->device = device_id;
    c->n = n_tokens;
    c->n_pad = (n_tokens + 1) & ~1;
    auto bail = [&](int code) {
        g_create_error = c->err;
        cfmm_ctx_destroy(c);
        return code;
    };
#define HIP_TRY_C(expr)                                                                               \
    do {                                                                                              \
        hipError_t _e = (expr);                                                                       \
        if (_e != hipSuccess) {                                                                       \
            fail(c, CFMM_ERR_HIP, "%s failed: %s", #expr, hipGetErrorString(_e));                     \
            return bail(CFMM_ERR_HIP);                                                                \
        }                                                                                             \
    } while (0)
    HIP_TRY_C(hipStreamCreateWithFlags(&c->own_stream, hipStreamNonBlocking));
    c->stream = c->own_stream;
    HIP_TRY_C(hipMalloc(reinterpret_cast<void**>(&c->d_v), (size_t)c->n * sizeof(double)));
    HIP_TRY_C(hipMalloc(reinterpret_cast<void**>(&c->d_out), (size_t)(c->n + 1) * sizeof(double)));
    // [n] v, [n+1] {psi, acc}, padding to a 128-byte boundary, then the output granules: 16 per fold
    // block = 2 per column, columns padded to a multiple of 8 (see fold_finish)
    c->gran_off = (size_t)((2 * c->n + 2 + 15) & ~15);
    c->flag_off = c->gran_off + 2 * (size_t)((c->n + 1 + 7) & ~7);
    const size_t stage_words = c->flag_off + 16;
    HIP_TRY_C(hipHostMalloc(reinterpret_cast<void**>(&c->h_stage), stage_words * sizeof(double), hipHostMallocMapped));
    std::memset(c->h_stage, 0, stage_words * sizeof(double));
    if (hipHostGetDevicePointer(reinterpret_cast<void**>(&c->d_stage), c->h_stage, 0) != hipSuccess) {
        (void)hipGetLastError();
        c->d_stage = nullptr; // fall back to explicit copies
    }
    {   // armed evaluations: fine-grained device memory the host can write through the PCIe BAR (optional)
        int large_bar = 0;
        const char* env = std::getenv("CFMM_AMD_ARMED");
        if (!(env && env[0] == '0') &&
            hipDeviceGetAttribute(&large_bar, hipDeviceAttributeIsLargeBar, c->device) == hipSuccess && large_bar != 0) {
            const size_t words = (size_t)c->n_pad + 8;
            if (hipExtMallocWithFlags(reinterpret_cast<void**>(&c->d_arm), words * sizeof(double), hipDeviceMallocFinegrained) == hipSuccess) {
                // self-check: what the host stores must be what the device holds
                std::vector<double> probe(words), back(words, 0.0);
                for (size_t j = 0; j < words; ++j) probe[j] = 1.0 + (double)j;
                std::memcpy(c->d_arm, probe.data(), words * sizeof(double));
                __builtin_ia32_sfence();
                if (hipMemcpy(back.data(), c->d_arm, words * sizeof(double), hipMemcpyDeviceToHost) != hipSuccess || back != probe) {
                    (void)hipGetLastError();
                    (void)hipFree(c->d_arm);
                    c->d_arm = nullptr;
                } else {
                    std::memset(c->d_arm, 0, words * sizeof(double));
                    __builtin_ia32_sfence();
                }
            } else {
                (void)hipGetLastError();
                c->d_arm = nullptr;
            }
        }
    }
    // the dynamic-LDS ceiling is a per-function, process-wide attribute: always raise it to the
    // full 160 KiB so that contexts with different n_tokens cannot shrink each other's limit
    HIP_TRY_C(prepare_kernels(160 * 1024));
#undef HIP_TRY_C
    *out = c;
    return CFMM_OK;
}

int cfmm_ctx_create_multi(int32_t n_devices, const int32_t* device_ids, int32_t n_tokens, cfmm_ctx** out)
{
    if (!out) return fail(nullptr, CFMM_ERR_INVALID_ARG, "out is null");
    *out = nullptr;
    if (n_devices < 1 || n_devices > 64 || !device_ids)
        return fail(nullptr, CFMM_ERR_INVALID_ARG, "n_devices must be in [1, 64] and device_ids non-null");
    if (n_tokens > kMaxLdsTokens)
        return fail(nullptr, CFMM_ERR_UNSUPPORTED, "multi-device contexts are limited to n_tokens <= %d", kMaxLdsTokens);
    cfmm_ctx* c = new cfmm_ctx();
    c->device = -1;
    c->n = n_tokens;
    c->n_pad = (n_tokens + 1) & ~1;
    c->workers.reset(new Workers());
    c->workers->rc.assign((size_t)n_devices, CFMM_OK);
    for (int d = 0; d < n_devices; ++d) {
        cfmm_ctx* child = nullptr;
        int rc = cfmm_ctx_create(device_ids[d], n_tokens, &child);   // validates the ordinal, the arch, n_tokens
        if (rc != CFMM_OK) {
            cfmm_ctx_destroy(c);
            return rc;   // g_create_error already holds the message
        }
        for (int e = 0; e < d; ++e)
            if (device_ids[e] == device_ids[d]) c->shards_distinct = false;
        c->shards.push_back(child);
    }
    *out = c;
    return CFMM_OK;
}

int32_t cfmm_device_count(const cfmm_ctx* c) { return c ? (c->shards.empty() ? 1 : (int32_t)c->shards.size()) : 0; }

void cfmm_ctx_destroy(cfmm_ctx* c)
{
    if (!c) return;
    if (c->workers) {
        Workers& w = *c->workers;
        {
            std::lock_guard<std::mutex> lk(w.mu);
            w.quit.store(true);
            w.cv.notify_all();
        }
        for (auto& t : w.threads) t.join();
    }
    if (is_parent(c)) {
        for (cfmm_ctx* child : c->shards) cfmm_ctx_destroy(child);
        delete c;
        return;
    }
    (void)hipSetDevice(c->device);
    armed_cancel(c);
    if (c->stream && c->stream != c->own_stream) (void)hipStreamSynchronize(c->stream);
    if (c->own_stream) (void)hipStreamSynchronize(c->own_stream);
    rccl_release(c);          // a communicator created by cfmm_rccl_init_rank goes with the context
    for (auto& s : c->segs) free_segment(s);
    for (auto e : c->ev_pool) (void)hipEventDestroy(e);
    free_trade_staging(c);
    (void)hipFree(c->d_v); (void)hipFree(c->d_out); (void)hipFree(c->d_partials); (void)hipFree(c->d_gtab);
    (void)hipFree(c->d_delta); (void)hipFree(c->d_lambda); (void)hipFree(c->d_over);
    (void)hipFree(c->d_xdelta); (void)hipFree(c->d_xlambda);
    (void)hipFree(c->d_flow); (void)hipFree(c->d_entries); (void)hipFree(c->d_chunks);
    (void)hipFree(c->d_tok_chunk_off); (void)hipFree(c->d_chunk_sums);
    if (c->h_stage) (void)hipHostFree(c->h_stage);
    if (c->d_arm) (void)hipFree(c->d_arm);
    if (c->own_stream) (void)hipStreamDestroy(c->own_stream);
    delete c;
}

int cfmm_set_stream(cfmm_ctx* c, void* hip_stream)
{
    if (!c) return CFMM_ERR_INVALID_ARG;
    CFMM_SINGLE_ONLY(c, "cfmm_set_stream");
    c->stream = static_cast<hipStream_t>(hip_stream); // NULL is HIP's default (null) stream
    return CFMM_OK;
}

int cfmm_reset_stream(cfmm_ctx* c)
{
    if (!c) return CFMM_ERR_INVALID_ARG;
    CFMM_SINGLE_ONLY(c, "cfmm_reset_stream");
    c->stream = c->own_stream;
    return CFMM_OK;
}

static int64_t* option_slot(cfmm_ctx* c, const char* key)
{
    if (!key) return nullptr;
    struct { const char* name; int64_t* slot; } table[] = {
        {"max_grid", &c->opt_max_grid}, {"block", &c->opt_block}, {"bin_copies", &c->opt_bin_copies},
        {"time_kernels", &c->opt_time_kernels}, {"geomean_exact", &c->opt_geomean_exact},
        {"fuse_segments", &c->opt_fuse_segments}, {"zero_copy", &c->opt_zero_copy},
        {"alternate", &c->opt_alternate}, {"pack", &c->opt_pack}, {"compact_trades", &c->opt_compact_trades},
        {"fast_math", &c->opt_fast_math}, {"armed", &c->opt_armed}, {"arm_timeout_ms", &c->opt_arm_timeout_ms},
        {"cost_geomean", &c->opt_cost_geomean}, {"cost_univ3", &c->opt_cost_univ3}, {"host_flag", &c->opt_host_flag},
        {"stop_in_noise", &c->opt_stop_in_noise}, {"multi_threads", &c->opt_multi_threads},
        {"dev_prices_in_window", &c->opt_dev_prices_in_window}, {"univ3_heads", &c->opt_univ3_heads}, {"direct_small", &c->opt_direct_small}, {"stream_stores", &c->opt_stream_stores},
#ifdef CFMM_TEST_HOOKS
        {"debug_stall_ms", &c->opt_debug_stall_ms},
#endif
    };
    for (auto& t : table)
        if (!std::strcmp(key, t.name)) return t.slot;
    return nullptr;
}

int cfmm_set_option(cfmm_ctx* c, const char* key, int64_t value)
{
    if (!c) return CFMM_ERR_INVALID_ARG;
    int64_t* slot = option_slot(c, key);
    if (!slot) return fail(c, CFMM_ERR_INVALID_ARG, "unknown option '%s'", key ? key : "(null)");
    if (slot == &c->opt_max_grid && value < 0) return fail(c, CFMM_ERR_INVALID_ARG, "max_grid must be >= 0 (0 = auto)");
    if (slot == &c->opt_block && !(value == 0 || value == kMidBlock || value == kBigBlock))
        return fail(c, CFMM_ERR_INVALID_ARG, "block must be 0 (auto), %d or %d", kMidBlock, kBigBlock);
    if (slot == &c->opt_bin_copies && !(value == 0 || value == 1 || value == 2))
        return fail(c, CFMM_ERR_INVALID_ARG, "bin_copies must be 0 (auto), 1 (shared) or 2 (per wavefront)");
    if (slot == &c->opt_stream_stores && !(value == 0 || value == 1 || value == 2))
        return fail(c, CFMM_ERR_INVALID_ARG, "stream_stores must be 0 (auto), 1 (write-through) or 2 (non-temporal)");
    *slot = value;
    if (slot != &c->opt_multi_threads)
        for (cfmm_ctx* child : c->shards) {
            int rc = cfmm_set_option(child, key, value);
            if (rc != CFMM_OK) return fail(c, rc, "%s", child->err.c_str());
        }
    if (slot == &c->opt_max_grid || slot == &c->opt_block || slot == &c->opt_fuse_segments ||
        slot == &c->opt_geomean_exact || slot == &c->opt_pack ||
        slot == &c->opt_cost_geomean || slot == &c->opt_cost_univ3 || slot == &c->opt_direct_small)
        c->geometry_dirty = true;
    return CFMM_OK;
}

int cfmm_get_option(const cfmm_ctx* c, const char* key, int64_t* value)
{
    if (!c || !value) return CFMM_ERR_INVALID_ARG;
    if (key && !std::strcmp(key, "peer_seq")) {   // read-only: sharded sweeps performed on the current peer buffers
        *value = (int64_t)c->peer_seq;            // (cfmm_set_peers' `seq` to continue from; ranks re-align on the maximum)
        return CFMM_OK;
    }
    int64_t* slot = option_slot(const_cast<cfmm_ctx*>(c), key);
    if (!slot) return fail(c, CFMM_ERR_INVALID_ARG, "unknown option '%s'", key ? key : "(null)");
    *value = *slot;
    return CFMM_OK;
}

int64_t cfmm_pools_count(const cfmm_ctx* c)
{
    if (!c) return 0;
    int64_t m = 0;
    for (auto& s : c->segs) m += s.m;
    for (auto& ps : c->psegs) m += ps.m;
    return m;
}

int32_t cfmm_n_tokens(const cfmm_ctx* c) { return c ? c->n : 0; }

int cfmm_kernel_times(cfmm_ctx* c, int64_t* sweep_launches, double* sweep_ms, int64_t* reduce_launches,
                      double* reduce_ms)
{
    if (!c) return CFMM_ERR_INVALID_ARG;
    if (!c->shards.empty()) {   // totals over the shards
        int64_t sn = 0, rn = 0;
        double sm = 0, rm = 0;
        for (cfmm_ctx* child : c->shards) {
            int64_t a = 0, b = 0;
            double x = 0, y = 0;
            int rc = cfmm_kernel_times(child, &a, &x, &b, &y);
            if (rc != CFMM_OK) return fail(c, rc, "%s", child->err.c_str());
            sn += a; rn += b; sm += x; rm += y;
        }
        if (sweep_launches) *sweep_launches = sn;
        if (sweep_ms) *sweep_ms = sm;
        if (reduce_launches) *reduce_launches = rn;
        if (reduce_ms) *reduce_ms = rm;
        return CFMM_OK;
    }
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    for (auto& p : c->pending) {
        float ms = 0.f;
        HIP_TRY(c, hipEventElapsedTime(&ms, p.a, p.b));
        if (p.what == 0) { c->t_sweep_n++; c->t_sweep_ms += ms; }
        else { c->t_reduce_n++; c->t_reduce_ms += ms; }
    }
    c->pending.clear();
    c->ev_used = 0;
    if (sweep_launches) *sweep_launches = c->t_sweep_n;
    if (sweep_ms) *sweep_ms = c->t_sweep_ms;
    if (reduce_launches) *reduce_launches = c->t_reduce_n;
    if (reduce_ms) *reduce_ms = c->t_reduce_ms;
    c->t_sweep_n = c->t_reduce_n = 0;
    c->t_sweep_ms = c->t_reduce_ms = 0;
    return CFMM_OK;
}

int32_t cfmm_segment_count(const cfmm_ctx* c)
{
    return c ? (int32_t)(c->shards.empty() ? c->segs.size() : c->psegs.size()) : 0;
}

int cfmm_segment_info(const cfmm_ctx* c, int32_t seg, int32_t* kind, int64_t* m, int32_t* block, int32_t* grid)
{
    if (!c) return CFMM_ERR_INVALID_ARG;
    if (!c->shards.empty()) {   // kind and size of the whole segment; launch geometry of shard 0's block
        if (seg < 0 || seg >= (int32_t)c->psegs.size()) return fail(c, CFMM_ERR_INVALID_ARG, "segment out of range");
        const int cs = child_segment(c, seg, 0);
        if (cs >= 0) {
            int rc = cfmm_segment_info(c->shards[0], cs, kind, nullptr, block, grid);
            if (rc != CFMM_OK) return rc;
        }
        if (kind) *kind = c->psegs[(size_t)seg].kind;
        if (m) *m = c->psegs[(size_t)seg].m;
        return CFMM_OK;
    }
    if (seg < 0 || seg >= (int32_t)c->segs.size()) return fail(c, CFMM_ERR_INVALID_ARG, "segment out of range");
    int rc = ensure_geometry(const_cast<cfmm_ctx*>(c));
    if (rc != CFMM_OK) return rc;
    const Segment& s = c->segs[(size_t)seg];
    if (kind) *kind = s.kind;
    if (m) *m = s.m;
    if (block) *block = s.block;
    if (grid) *grid = s.grid;
    return CFMM_OK;
}

} // extern "C"
