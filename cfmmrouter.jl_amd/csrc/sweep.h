// sweep.h -- internal C++ interface between the C ABI (abi_*.cpp) and the gfx950 kernels
// (sweep_kernels.hip).  Not installed; the public surface is include/cfmm_amd.h.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace cfmm {

constexpr int kMidBlock = 512;       // 8 wavefronts: small markets (one tile per block) and fused multi-family launches
constexpr int kBigBlock = 1024;      // 16 wavefronts: single-family launches, few partial rows
constexpr int kFoldBlock = 512;      // reduce_partials / reduce_gather
constexpr int kResidentThreads = 2048 * 256; // one machine of resident threads (256 CUs x 2048)
constexpr int kReduceCols = 8;       // tokens per fold block (64 B = half a 128-byte line of each partial row; the two blocks that share
                                     // a line run on the SAME XCD, i.e. behind the same L2: fold_colblock in sweep_kernels.hip)
constexpr int kRowAlign = 16;        // partial rows are padded to a multiple of this many doubles (128 B)
constexpr int kMaxLdsTokens = 8192;  // up to here the prices + one bin copy fit the 160 KiB LDS of a CU;
                                     // larger markets pull Ψ per token (sweep_body<..., GBINS=true>)
constexpr int kGatherChunk = 512;    // incidence entries per wavefront in gather_chunks

// SoA-of-pairs pool stores, one struct per pool family.  All pointers are device pointers.
// Packed fee + token-index record (8 B instead of 16): tok = i1 | i2 << 16 (n_tokens <= 65536), gidx = index of
// the pool's fee in the launch's fee table (markets use a handful of fee tiers; the table, <= kMaxFeeTable distinct
// values per launch, is staged in LDS).  Segments whose fees do not fit keep the plain gamma / Ai arrays (pk == null).
struct PackedFeeTok {
    uint32_t tok;
    uint32_t gidx;
};
constexpr int kMaxFeeTable = 256;

// Window of the "fast" arithmetic (sweep_kernels.hip, div_by / fast_sqrt): when every reserve, fee, liquidity and
// price of a launch lies in [2^-kFastExp, 2^kFastExp], the IEEE division / square-root sequences run without their
// range scaffolding (v_div_scale, v_div_fmas, v_div_fixup, ldexp pairs) and still return the correctly rounded bits.
constexpr int kFastExp = 150;

struct ProductPools {            // src/cfmms.jl:101-111
    const double2* R;            // [m] {R1, R2}
    const double* gamma;         // [m]
    const int2* Ai;              // [m] {i1, i2}, 0-based
    const PackedFeeTok* pk;      // [m] {tokens, fee-table index}: replaces Ai (and gamma) in the sweep (24 B per pool instead of
                                 //     32); null only in large-market mode (n_tokens > kMaxLdsTokens: plain arrays)
    int gbase;                   // this segment's first entry in the launch's fee table, or -1: no table (too many fee tiers,
                                 //     or option "pack" = 0): the fee comes from gamma[]
};
struct GeoMeanPools {            // src/cfmms.jl:152-165
    const double2* R;
    const double2* w;            // [m] {w1, w2}
    const double* gamma;
    const int2* Ai;
    const double* eta;           // [m] η = w1/w2                  prepared at upload
    const double2* Q;            // [m] {Q1, Q2}: the v-independent part of the two log-space exponents, prepared at
                                 //     upload: Q1 = log γ + log η + log R2 + η·log R1,  Q2 = η·(log γ + log R1 − log η) + log R2
    int reference_order;         // 1: evaluate with pow in the reference's operation order
    const PackedFeeTok* pk;      // see ProductPools (48 B per pool instead of 56)
    int gbase;
};
// One tick of a walk list (UniV3Ops::list_tick): everything find_arb_pos (src/cfmms.jl:321-337) needs for one prepared
// tick in ONE 64-byte line -- lanes walk different pools, so every tick visit is a scattered access; round 2 kept
// {k, s_in}, {δmax, s_out} and R_out in three arrays = three lines per visit.
struct alignas(64) TickRec {
    double2 ks;                  // {k, R_in + alpha_in}
    double2 dt;                  // {delta_max, R_out + beta_out}
    double rout;                 // R_out
    double thr;                  // this tick's drain threshold (the same value as UniV3Pools::thr[...]; 0: never / closing record): the
                                 // walk's 2^-40 band test needs it exactly where the record is already in registers
    double2 psum;                // {Σδ, Σλ} of the current tick and of every list tick BEFORE this one, all drained, summed in
                                 // walk order (UniV3Ops::solve_dir); every list is closed by a record that carries only this
};
struct UniV3Pools {              // src/cfmms.jl:226-245 as find_arb_pos constants (see UniV3Ops)
    const double2* pg;           // [m] {current_price, gamma}
    const int2* Ai;              // [m]
    const double2* cur_a;        // [m] current tick {k, R1+alpha}
    const double2* cur_b;        // [m] current tick {R2+beta, k/beta - (R1+alpha)}
    const double* cur_c;         // [m] current tick  k/alpha - (R2+beta)
    const double2* curR;         // [m] current tick {R1, R2}
    const int4* walk;            // [m] ticks beyond the current one: {up_begin, up_count, lo_begin, lo_count}
    const TickRec* ticks;        // [W] the walk lists: per pool the non-empty ticks above, then below, its current one, each
                                 //     list closed by one extra record (psum of the whole list)
    const double* thr;           // [W] per record: the largest price at which the walk drains that tick (0: never / closing record)
    const uint4* head;           // [2m] or null.  Round 5: the first FOUR drain thresholds of pool i's two walk lists ({up, down} =
                                 //     head[2i], head[2i+1]) as binary32 values rounded DOWN (bits; 0 = never drains, a NaN pattern =
                                 //     "not representable: ask the exact array"), read with the pool's other coalesced streams.  With
                                 //     lo = the float and hi = the next float up, lo <= T <= hi: price <= lo proves that the tick
                                 //     drains, price > hi proves that it does not, anything in between falls back to the exact scan
                                 //     of thr[] -- so the walk decisions are the exact ones, while four of five walking pools no longer
                                 //     touch thr[] (whose 128-byte lines were being fetched almost in full for 32 useful bytes each)
    int has_walk;                // 0: no pool of the segment has a tick beyond its current one (every BoundedProduct
                                 //    pool): the walk spans are not even loaded
    const double* cp;            // [m] current_price alone, read with pk instead of pg + Ai (packed records)
    const PackedFeeTok* pk;      // see ProductPools
    int gbase;
};

// How a fold launch hands {Ψ, acc} to the host (mapped pinned memory), if at all: gran != null -> the block's 8 columns
// leave as 16 SELF-VALIDATING 8-byte granules {tag, 32 bits of the double} (two per column) written by one store
// instruction = two full 64-byte lines; the host re-reads them until all carry the tag -- no drain of the output stores,
// no ticket, no flag word.  gran == null: plain stores to `out` (device consumers).
struct HostOut {
    unsigned long long* gran;    // [2 * ceil8(n1)] words in mapped host memory, or null
    unsigned long long tag;      // 1 .. 2^32 - 1 (0 = an empty buffer)
};
struct SweepArgs {
    const double* v;             // [n] device
    int n;                       // n_tokens
    int n_pad;                   // n rounded up to even (LDS row pitch)
    int v_shift;                 // 4: prices staged in LDS as {v, rcp_refined(v)} pairs (16 B per token: the fast arithmetic's
                                 //    divisions by a price); 3: the prices alone (markets too wide for the pairs)
    int need_logv;               // 1: also stage log v per token in LDS (launches with a log-space GeometricMean segment)
    const double* gtab;          // the launch's fee table (device), staged in LDS when gtab_n > 0
    int gtab_n;
    int copies;                  // private bin copies per block (1 or one per wavefront)
    unsigned long long* flags;   // sticky report word in mapped host memory (or null): kFlagWindow / kFlagGaveUp, set by a block
                                 //    that poisons its row (see sweep_tiles)
    int64_t m;                   // pools in this segment
    // Trade buffers of the segment (null when !materialize).  compact == 0: Delta[i] = {Δ₁, Δ₂}, Lambda[i] = {Λ₁, Λ₂}
    // (32 B written per pool).  compact == 1: at most one direction of a pool trades, so ONE 16-byte record goes
    // to Delta[i]:  {+Δ₁, Λ₂}  (direction 1 or no trade)  or  {−Δ₂, Λ₁}  (direction 2; the sign bit of the first
    // entry carries the direction, −0.0 included); any other pool -- both directions non-zero (γ > 1), NaN, or the tiny
    // negative / −0.0 values the reference's tick arithmetic yields on degenerate UniV3 boundaries -- writes the record
    // {0, −1} and its four values to Lambda[i] = {Δ₁, Δ₂}, Over[i] = {Λ₁, Λ₂}.  Lossless bit for bit; decoded by
    // cfmm_get_trades / expand_trades / update_two_coin.
    double2* Delta;
    double2* Lambda;
    double2* Over;
    int compact;
    double* partials;            // [grid][row_pitch] rows of this launch ([grid][1] with global bins)
    int row_pitch;               // doubles between two partial rows: n+1 rounded up to a multiple of 16 (128-byte rows: a fold block's
                                 //    64-byte column group never straddles a line; round 4's pitch of n+1 doubles shifted every row by
                                 //    8 bytes: 1491 KiB fetched for 526 KB of rows), 1 with global bins
    double2* gflow;              // null: LDS bins; else [m] {Λ₁−Δ₁, Λ₂−Δ₂} of this segment (large markets)
    int reverse;                 // 1: every lane walks its tiles last-to-first (alternates between consecutive sweeps, so a
                                 //    sweep starts on the pool data the previous one left in the XCD's L2)
    // Pre-armed launch (arm_word != null; cfmm_route): the kernel is enqueued BEFORE its price vector exists and every
    // block, after issuing its first pool loads and clearing its bins, waits until the host has written v into `v`
    // (fine-grained device memory, through the PCIe BAR) and then arm_seq into *arm_word -- the launch latency is
    // spent while the previous evaluation and the host solver run.  *arm_word == arm_seq | kArmCancel: the launch
    // is not needed (the blocks skip their pools, the fold returns without output).  Waits are bounded by
    // arm_timeout ticks of the 100 MHz wall clock; a block that gives up poisons the dual column with NaN.
    const unsigned long long* arm_word;
    unsigned long long arm_seq;
    long long arm_timeout;
    // Single-block launches (grid == 1: markets of up to kDirectPools pools, one family): the block's row IS the result, so
    // it goes straight to `direct_out` (device consumers) or -- direct_host.gran set -- to the host as granules, and NO fold
    // launch follows: one kernel per evaluation instead of two (the reference's own benchmark grid, benchmark/scaling.jl:8-38,
    // has six of its ten market sizes in this range, where an evaluation is launch latency and nothing else).
    int nt_stores;               // 1: trade records leave through non-temporal stores (HBM-resident markets); 0: write-through
    int direct;
    double* direct_out;          // [n + 1] {Ψ, acc}
    HostOut direct_host;
};
constexpr int kDirectPools = 2048;   // two tiles of a 1024-thread block
constexpr unsigned long long kArmCancel = 1ull << 63;
constexpr unsigned long long kFlagWindow = 1;   // a fast kernel staged a price outside [2^-kFastExp, 2^kFastExp]: rows poisoned
constexpr unsigned long long kFlagGaveUp = 2;   // a pre-armed launch gave up waiting for its price vector: rows poisoned

// One launch over up to kMaxMulti segments (sweep_multi).
constexpr int kMaxMulti = 4;
union AnyPools {
    ProductPools p;
    GeoMeanPools g;
    UniV3Pools u;
};
struct MultiSeg {
    int kind;
    int64_t m;
    AnyPools pools;
    double2* Delta;
    double2* Lambda;
    double2* Over;
    double2* gflow;
};
struct MultiArgs {
    int nseg;
    int xcd_map;                 // 1: XCD-aware, cost-weighted block -> segment map (needs grid % 256 == 0): deal j = b / 8
                                 //    (the 8 blocks that land on the 8 XCDs together) belongs to segment
                                 //    pattern[(j + j / 32) % 32] and is that segment's deal number (j / 32)·w + rank[...];
                                 // 0: block b -> segment b % nseg (grids that are no multiple of 256 blocks: small markets)
    unsigned char pattern[32];   // segment of each of 32 consecutive deals; segment s appears seg_w[s] times
    unsigned char rank[32];      // rank[p] = #{p' < p : pattern[p'] == pattern[p]}
    int seg_w[kMaxMulti];        // deals out of 32 given to each segment (in proportion to pools x cost per pool)
    MultiSeg seg[kMaxMulti];
    SweepArgs common;            // v, n, n_pad, copies, partials (row 0 of this launch)
};

struct LaunchCfg {
    int block;                   // kMidBlock or kBigBlock
    int grid;
    size_t lds_bytes;
    int arith = 0;               // the kernel's arithmetic (sweep_kernels.hip, FASTK): 0 = the compiler's full-range sequences,
                                 // 1 = fast (every pool constant of the launch and -- the host KNOWS -- every price inside the
                                 // kFastExp window), 2 = auto (pool constants inside the window, prices unknown to the host:
                                 // device-pointer sweeps; every block picks the loop from the prices it stages)
    hipEvent_t ev_start = nullptr; // both set: the launch is timed by the command processor
    hipEvent_t ev_stop = nullptr;  // (hipExtLaunchKernel), i.e. the kernel's own execution span
};

hipError_t launch_sweep(const ProductPools& p, const SweepArgs& a, const LaunchCfg& c, bool materialize,
                        hipStream_t s);
hipError_t launch_sweep(const GeoMeanPools& p, const SweepArgs& a, const LaunchCfg& c, bool materialize,
                        hipStream_t s);
hipError_t launch_sweep(const UniV3Pools& p, const SweepArgs& a, const LaunchCfg& c, bool materialize,
                        hipStream_t s);

// block b writes partial row b (see sweep_multi for the block -> segment map); without xcd_map the grid must be a
// multiple of ma.nseg.
hipError_t launch_multi(const MultiArgs& ma, const LaunchCfg& c, bool materialize, hipStream_t s);

// Large markets: chunk_sums[c] = sum of flow[entries[chunks[c].x .. chunks[c].y)], then
// out[t] = sum of chunk_sums[tok_chunk_off[t] .. tok_chunk_off[t+1]) for t < n and
// out[n] = sum of acc_rows[0 .. rows) (the dual-scalar column of the partial rows).
hipError_t launch_gather(const int2* chunks, const int* entries, const double* flow, double* chunk_sums, int n_chunks,
                         const int* tok_chunk_off, double* out, int n, const double* acc_rows, int rows, hipStream_t s);

struct ArmWord {                 // see SweepArgs::arm_word; {nullptr, 0} = not armed
    const unsigned long long* word;
    unsigned long long seq;
};
// rows of n1 doubles, `pitch` doubles apart (SweepArgs::row_pitch)
inline int row_pitch_of(int n1) { return (n1 + kRowAlign - 1) / kRowAlign * kRowAlign; }
// out[j] = sum over rows of partials[row][j], j in [0, n1); fixed summation order.
hipError_t launch_reduce(const double* partials, int rows, int n1, int pitch, double* out, hipStream_t s,
                         hipEvent_t ev_start = nullptr, hipEvent_t ev_stop = nullptr, HostOut host = HostOut{nullptr, 0},
                         ArmWord arm = ArmWord{nullptr, 0});

// Sharded runs (cfmm_set_peers): the row fold fused with the one-shot all-reduce over xGMI peer
// mappings (reduce_gather in sweep_kernels.hip): block b folds its kReduceCols columns, publishes
// them as self-validating granules in this rank's symmetric buffer and adds the same columns of
// every peer in rank order -- one launch, one hop on the critical path, no hand-off between the blocks of a rank.
constexpr int kMaxPeers = 16;
struct PeerSet {
    unsigned long long* gran[kMaxPeers];    // peer p's granules: [2][count][2] uint64 (see include/cfmm_amd.h)
    int world, rank;
    long long count;                        // n_tokens + 1
    unsigned long long seq;                 // 1, 2, ... identical on every rank
    long long timeout_ticks;                // wall_clock64() ticks (100 MHz) before a wait gives up (NaN output)
    HostOut host;                           // optional: the global {Ψ, acc} also travel to the host as granules
    ArmWord arm;                            // pre-armed evaluation: a cancelled launch publishes nothing
};
hipError_t launch_reduce_gather(const double* partials, int rows, int n1, int pitch, double* out, hipStream_t s,
                                const PeerSet& ps, hipEvent_t ev_start = nullptr, hipEvent_t ev_stop = nullptr);

// R <- (R + gamma*Delta) - Lambda in place; GeometricMean (Q, eta non-null): Q <- the exponents' constants for the new R;
// *left_window <- 1 if a new reserve lies outside [2^-kFastExp, 2^kFastExp]
hipError_t launch_update_two_coin(double2* R, const double* gamma, const double2* Delta, const double2* Lambda,
                                  const double2* Over, int compact, double2* Q, const double* eta, int64_t m, int* left_window,
                                  hipStream_t s);
// compact trade records -> full {Δ₁, Δ₂} / {Λ₁, Λ₂} arrays (cfmm_trades_dev, cfmm_get_trades*)
hipError_t launch_expand_trades(const double2* rec, const double2* ovA, const double2* ovB, double2* Delta, double2* Lambda,
                                int64_t m, hipStream_t s);

size_t sweep_lds_bytes(int n_pad, int copies, int block, int need_logv, int gtab_n, int stage_y);
hipError_t prepare_kernels(size_t max_lds_bytes);

} // namespace cfmm
