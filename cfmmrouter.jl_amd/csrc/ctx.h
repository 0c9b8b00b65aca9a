// ctx.h -- internal: the context behind the C ABI (include/cfmm_amd.h) and what the abi_*.cpp translation units
// share.  Host-side only.  There is no CPU fallback anywhere behind this header: without a gfx950 device every
// entry point that needs one fails with CFMM_ERR_HIP.
//
//   abi_context.cpp   create / destroy / options / streams / introspection
//   abi_upload.cpp    pool validation + upload (src/cfmms.jl:76-111, :152-165, :226-245), prepared constants
//   abi_sweep.cpp     launch geometry, one evaluation = sweep launches + row fold, host-pointer sweeps,
//                     pre-armed evaluations
//   abi_trades.cpp    trade download / device views, update_reserves!, reserves / prices read-back
//   abi_route.cpp     route! in one call (L-BFGS-B + objectives), the bare solver
//   abi_multi.cpp     single-process multi-device parents
//   abi_peers.cpp     one process per GPU: peer buffers, cfmm_set_peers
//   abi_rccl.cpp      one process per GPU: the all-reduce through RCCL (cfmm_set_rccl_comm, cfmm_rccl_init_rank)
#pragma once

#include "../../include/cfmm_amd.h"
#include "sweep.h"

#include <atomic>
#include <condition_variable>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <functional>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

namespace cfmm {

struct Segment {
    int kind = 0;
    int64_t m = 0;
    int64_t trade_off = 0; // first row of this segment in the trade buffers
    int64_t n_ticks_total = 0;
    int fast_ok = 0; // every constant the sweep divides by / takes roots of lies in [2^-kFastExp, 2^kFastExp] (sweep.h)
    // device arrays (owned)
    double2* R = nullptr;
    double2* w = nullptr;
    double* eta = nullptr;   // geomean: η = w1/w2
    double2* lR = nullptr;   // geomean: {Q1, Q2}, the v-independent constants of the log-space exponents (GeoMeanLogOps)
    double* gamma = nullptr;
    int2* Ai = nullptr;
    double2* pg = nullptr;
    double* cp = nullptr;    // univ3: current_price alone (packed records)
    int has_walk = 1;        // univ3: some pool has a tick beyond its current one
    double2* cur_a = nullptr;
    double2* cur_b = nullptr;
    double* cur_c = nullptr;
    double2* curR = nullptr;
    int4* walk = nullptr;
    TickRec* ticks = nullptr;   // univ3: walk lists (sweep.h TickRec)
    double* thr = nullptr;      // univ3: drain thresholds, one per record
    uint4* head = nullptr;      // univ3: per pool the first four thresholds of both walk lists as rounded-down floats (sweep.h UniV3Pools)
    // launch geometry (decided by ensure_geometry)
    int block = kMidBlock;
    int grid = 0;
    int64_t row_off = 0; // first partial row
    PackedFeeTok* pk = nullptr;     // {i1 | i2 << 16, fee-table index} per pool, or null (too many distinct fees / tokens)
    std::vector<double> gvals;      // the segment's distinct fees, in order of first appearance (index = PackedFeeTok::gidx)
    int gbase = 0;                  // first entry of this segment in its launch's fee table (ensure_geometry)
    std::vector<int32_t> h_ai; // host copy of Ai: large-market mode (incidence build) and UniV3 segments
    // UniV3 only: the pool definitions as uploaded (update_reserves! moves current_price and re-derives the constants)
    std::vector<double> h_cp, h_gamma, h_lt, h_liq;
    std::vector<int64_t> h_tick_off;
};

// A launch: either one segment (sweep_kernel) or up to kMaxMulti segments fused (sweep_multi).
struct Group {
    int first = 0, nseg = 1;
    bool multi = false;
    int block = kMidBlock;
    int grid = 0;       // total blocks of the launch
    int64_t row_off = 0;
    int gtab_n = 0;     // entries of this launch's fee table (0: its segments use the plain gamma / Ai arrays)
    // XCD-aware weighted block -> segment map of a fused launch (see sweep_multi); xcd_map == false: block b -> segment b % nseg
    bool xcd_map = false;
    unsigned char pattern[32] = {0}, rank[32] = {0};
    int seg_w[kMaxMulti] = {0};
};

struct Workers {
    // Multi-device parents: one persistent thread per shard >= 1 (shard 0 runs on the calling thread).  A call publishes
    // {v, materialize} and bumps `go`; workers spin briefly on it (an L-BFGS-B evaluation follows
    // the previous one within microseconds), then sleep on the condition variable.
    std::vector<std::thread> threads;
    std::mutex mu;
    std::condition_variable cv;
    std::atomic<uint64_t> go{0};
    std::atomic<int> pending{0};
    std::atomic<int> sleepers{0};
    std::atomic<bool> quit{false};
    const double* v = nullptr;
    bool materialize = false;
    std::vector<int> rc;
};

// Pinned staging of the trade download (abi_trades.cpp): a few worker threads, each with its own stream and two slots.
struct TradeStaging {
    static constexpr int kThreads = 4, kSlots = 2;
    static constexpr int64_t kChunkRows = 1 << 16;   // 1 MiB per slot
    double2* slot[kThreads][kSlots] = {{nullptr}};
    hipStream_t stream[kThreads] = {nullptr};
    hipEvent_t done[kThreads][kSlots] = {{nullptr}};
    bool ready = false;
};

} // namespace cfmm

struct cfmm_ctx {
    int device = 0;
    int n = 0;
    int n_pad = 0;
    hipStream_t own_stream = nullptr;
    hipStream_t stream = nullptr;
    std::vector<cfmm::Segment> segs;
    std::vector<cfmm::Group> groups;
    int64_t m_total = 0;
    int64_t touched_bytes = 0;    // what one materialising sweep moves by construction (packed layout; ensure_geometry): decides "stream_stores" = auto
    int64_t rows_total = 0;

    double* d_v = nullptr;        // [n]
    double* d_out = nullptr;      // [n+1]
    double* d_partials = nullptr; // [rows_cap][row_width]: n+1 columns, rows padded to 128 bytes
    int64_t rows_cap = 0;
    // trade buffers [trade_cap] each.  Compact layout (option "compact_trades", default): d_delta holds ONE 16-byte
    // record per pool, d_lambda / d_over the four values of the rare pools that trade in both directions (sweep.h
    // SweepArgs); plain layout: d_delta = {Δ₁, Δ₂}, d_lambda = {Λ₁, Λ₂}.  d_xdelta / d_xlambda: expanded copies, for
    // cfmm_trades_dev and the trade download.
    double2* d_delta = nullptr;
    double2* d_lambda = nullptr;
    double2* d_over = nullptr;
    double2* d_xdelta = nullptr;
    double2* d_xlambda = nullptr;
    int64_t x_cap = 0;
    bool x_valid = false;         // d_xdelta / d_xlambda hold the expansion of the trades currently on the device
    int trades_compact = 0;       // layout of the trades currently on the device
    int64_t trade_cap = 0;
    cfmm::TradeStaging tstage;
    // large-market mode (n > kMaxLdsTokens): token -> (pool, side) incidence and flow scratch
    double2* d_flow = nullptr;    // [m_total] {Λ₁−Δ₁, Λ₂−Δ₂}
    int* d_entries = nullptr;     // [2·m_total] flat flow indices grouped by token
    int2* d_chunks = nullptr;     // [n_chunks] {begin, end} into d_entries
    int* d_tok_chunk_off = nullptr; // [n+1]
    double* d_chunk_sums = nullptr; // [n_chunks]
    int n_chunks = 0;
    // sharded operation (cfmm_set_peers): the fold launch of every sweep also gathers the peers' {Ψ, acc} over xGMI
    // (reduce_gather), so eval / find_arb / route return GLOBAL {Ψ, acc}
    std::vector<uint64_t> peers;  // device addresses of all ranks' symmetric buffers
    int peer_rank = 0;
    uint64_t peer_seq = 0;
    long long peer_timeout_ticks = 3000000000ll;   // 30 s of wall_clock64() at 100 MHz (CFMM_AMD_PEER_TIMEOUT_S)
    // sharded operation through RCCL (cfmm_set_rccl_comm / cfmm_rccl_init_rank, abi_rccl.cpp): every sweep's fold is followed,
    // in-stream, by ncclAllReduce(d_out, n + 1 doubles) -- same contract as `peers`, one exchange at a time
    void* rccl_comm = nullptr;    // ncclComm_t
    bool rccl_owned = false;      // created by cfmm_rccl_init_rank: destroyed with the context
    // pinned + device-mapped staging: [n] v in, [n+1] {Ψ, acc} out, padding to a 128-byte boundary, then the output
    // granules (16 per fold block = 2 per column, columns padded to a multiple of 8; see fold_finish)
    double* h_stage = nullptr;
    double* d_stage = nullptr;    // device address of h_stage
    size_t gran_off = 0;          // first output granule in h_stage / d_stage (doubles)
    size_t flag_off = 0;          // the sweeps' sticky report word (sweep.h kFlagWindow / kFlagGaveUp), its own 128-byte line
    double* d_gtab = nullptr;     // [groups][kMaxFeeTable] fee tables of the launches (packed pool records)
    size_t gtab_cap = 0;
    // pre-armed evaluations of cfmm_route (sweep.h SweepArgs::arm_word): [n_pad] v, then the word, in FINE-GRAINED
    // device memory that the host writes through the PCIe BAR (null: no large BAR, or the self-check failed)
    double* d_arm = nullptr;
    uint64_t arm_seq = 0;         // sequence number of the latest armed launch
    bool arm_pending = false;     // an armed launch is enqueued and has not been signalled or cancelled yet
    uint64_t arm_tag = 0;         // output tag that launch will deliver
    uint64_t out_seq = 0;         // host-visible outputs: sequence number of the latest granule-delivering sweep
    bool last_host_out = false;   // the latest enqueue_sweep delivers {psi, acc} as granules (the caller polls them)
    std::vector<double> last_out; // psi..., acc of the latest host-pointer sweep
    std::vector<double> trade_v;  // v of the latest MATERIALISING host-pointer sweep (empty: none / device-pointer sweep)
    bool have_out = false;
    bool have_trades = false;
    bool geometry_dirty = true;

    // options (cfmm_set_option)
    int64_t opt_max_grid = 0;    // 0 = auto
    int64_t opt_block = 0;       // 0 = auto, else kMidBlock or kBigBlock
    int64_t opt_bin_copies = 0;  // 0 = auto, 1 = one shared copy, 2 = one copy per wavefront
    int64_t opt_time_kernels = 0;
    int64_t opt_geomean_exact = 0; // 1: pow-based reference-order forms instead of log-space
    int64_t opt_fuse_segments = 1; // 1: sweep all pool families in one launch (sweep_multi)
    int64_t opt_zero_copy = 1;     // 1: host-pointer calls read v / receive Ψ through mapped pinned memory
    int64_t opt_cost_geomean = 10; // cost of a GeometricMean / UniV3 evaluation in tenths of a ProductTwoCoin one (10 = blocks in
    int64_t opt_cost_univ3 = 10;   // proportion to pool counts)
    int64_t opt_compact_trades = 1; // 1: a materialising sweep writes one 16-byte trade record per pool (+ overflow rows)
    int64_t opt_pack = 1;          // 1: sweeps read the packed fee + token record when the launch's distinct fees fit the LDS table
    int64_t opt_alternate = 1;     // 1: consecutive sweeps walk the tiles in alternating directions (L2 reuse across sweeps)
    int64_t opt_fast_math = 1;     // 1: division / square root without range scaffolding where operands are inside the window (same bits)
    int64_t opt_armed = 1;         // 1: cfmm_route enqueues evaluation k+1 while evaluation k runs (see abi_sweep.cpp)
    int64_t opt_arm_timeout_ms = 2000; // bound of that wait
    int64_t opt_host_flag = 1;     // 1: zero-copy host-pointer sweeps deliver {Ψ, acc} as self-validating granules that the caller
                                   //    polls, instead of waiting for the stream (saves the end-of-kernel + signal path)
    int64_t opt_stop_in_noise = 0; // cfmm_route: 1 = end the run when a line-search trial point sits on the rounding-noise floor
                                   //    (LbfgsbOptions::stop_in_noise; fewer evaluations, departs from L-BFGS-B 3.0); 0 = reference behaviour
    int64_t opt_multi_threads = 1;
    int64_t opt_stream_stores = 0; // trade-record stores: 0 = auto (non-temporal when one sweep touches more than the 256 MiB Infinity Cache,
                                   //    i.e. the pool state cannot stay cache-resident between sweeps; write-through otherwise), 1 = always
                                   //    write-through, 2 = always non-temporal (a caller that rotates over many markets says so)
    int64_t opt_direct_small = 1;  // 1: single-family markets of up to kDirectPools pools are swept by ONE block that publishes {Ψ, acc}
                                   //    itself (no fold launch); 0: the general two-launch geometry
    int64_t opt_univ3_heads = 1;   // 1: multi-tick UniV3 walks decide their first four list ticks from the per-pool float threshold heads
    int64_t opt_dev_prices_in_window = 0; // 1 = the CALLER vouches that the prices of device-pointer sweeps lie in [2^-kFastExp, 2^kFastExp]
                                     //    (what the host checks itself for host-pointer calls): cfmm_sweep_dev launches the fast kernels
                                     //    instead of the ones that carry both arithmetics; every block still checks what it stages and
                                     //    poisons its row (NaN in {psi, acc}: an error, never a wrong number) when the promise is broken
    int64_t opt_debug_stall_ms = 0; // test hook, reachable only in libcfmm_amd_hooks.so (-DCFMM_TEST_HOOKS: the option key and the stall
                                    //    exist there alone; the FIELD is unconditional so that every translation unit sees one layout)
    uint64_t sweep_count = 0;

    // kernel timing
    std::vector<hipEvent_t> ev_pool;
    size_t ev_used = 0;
    struct Pending { hipEvent_t a, b; int what; };
    std::vector<Pending> pending;
    int64_t t_sweep_n = 0, t_reduce_n = 0;
    double t_sweep_ms = 0, t_reduce_ms = 0;

    // single-process multi-device parent (cfmm_ctx_create_multi): shards non-empty, no device state of its own
    std::vector<cfmm_ctx*> shards;
    struct ParentSeg { int kind; int64_t m; int64_t trade_off; };
    std::vector<ParentSeg> psegs;          // one per cfmm_pools_add_* call with m > 0
    std::unique_ptr<cfmm::Workers> workers;
    bool shards_distinct = true;           // no two shards share a device (pre-armed evaluations need that)

    mutable std::string err = "";
};

namespace cfmm {

extern thread_local std::string g_create_error;

int fail(const cfmm_ctx* c, int code, const char* fmt, ...);

#define HIP_TRY(ctx, expr)                                                                            \
    do {                                                                                              \
        hipError_t _e = (expr);                                                                       \
        if (_e != hipSuccess)                                                                         \
            return ::cfmm::fail(ctx, CFMM_ERR_HIP, "%s failed: %s", #expr, hipGetErrorString(_e));    \
    } while (0)

#define CFMM_SINGLE_ONLY(c, what)                                                                     \
    if (!(c)->shards.empty() || (c)->device < 0)                                                      \
        return ::cfmm::fail(c, CFMM_ERR_UNSUPPORTED, what " is not available on a multi-device context (host-pointer calls only)")

template <class T>
int upload(cfmm_ctx* c, T** dst, const void* src, size_t count)
{
    *dst = nullptr;
    if (count == 0) return CFMM_OK;
    HIP_TRY(c, hipMalloc(reinterpret_cast<void**>(dst), count * sizeof(T)));
    HIP_TRY(c, hipMemcpy(*dst, src, count * sizeof(T), hipMemcpyHostToDevice));
    return CFMM_OK;
}

inline bool global_bins(const cfmm_ctx* c) { return c->n > kMaxLdsTokens; }
inline int row_width(const cfmm_ctx* c) { return global_bins(c) ? 1 : row_pitch_of(c->n + 1); }   // doubles between partial rows
inline bool is_parent(const cfmm_ctx* c) { return !c->shards.empty() || c->device < 0; }

// abi_upload.cpp
void free_segment(Segment& s);
int univ3_build(cfmm_ctx* c, Segment& s, int64_t m, const double* current_price, const double* gamma, const int32_t* Ai,
                const int64_t* tick_off, const double* lower_ticks, const double* liquidity);

// abi_sweep.cpp
int ensure_geometry(cfmm_ctx* c);
constexpr int kPricesUnknown = 0, kPricesInWindow = 1, kPricesOutside = 2;
int enqueue_sweep(cfmm_ctx* c, const double* d_v, double* d_out, bool materialize, bool want_host_out = false,
                  uint64_t arm_seq = 0, int price_window = kPricesUnknown);
bool prices_in_fast_window(const double* v, int n);
unsigned long long take_flags(cfmm_ctx* c, unsigned long long mask);
int check_prices(cfmm_ctx* c, const double* v);
int host_sweep_begin(cfmm_ctx* c, const double* v, bool materialize);
int host_sweep_end(cfmm_ctx* c);
int single_host_sweep(cfmm_ctx* c, const double* v, bool materialize);
int host_sweep(cfmm_ctx* c, const double* v, bool materialize);   // single device or parent
bool can_arm(cfmm_ctx* c);
void armed_cancel(cfmm_ctx* c);
int armed_eval(cfmm_ctx* c, const double* v, bool* lost_out);      // single device or parent

// abi_multi.cpp
void shard_range(int64_t m, int d, int nd, int64_t& lo, int64_t& hi);
int multi_host_sweep(cfmm_ctx* c, const double* v, bool materialize);
int multi_add(cfmm_ctx* c, int kind, int64_t m, const std::function<int(cfmm_ctx*, int64_t, int64_t)>& add);
int child_segment(const cfmm_ctx* c, int pseg, int d);
int multi_get_trades_range(cfmm_ctx* c, int32_t seg, int64_t first, int64_t count, double* Delta, double* Lambda);

// abi_trades.cpp
void free_trade_staging(cfmm_ctx* c);

// abi_rccl.cpp
int rccl_all_reduce_out(cfmm_ctx* c, double* d_out);
void rccl_release(cfmm_ctx* c);

} // namespace cfmm
