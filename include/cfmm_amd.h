/*
 * cfmm_amd.h -- C ABI of libcfmm_amd.so: the MI355X (gfx950) implementation of
 * CFMMRouter.jl's per-CFMM arbitrage sweep and the reductions route! consumes.
 *
 * The reference has no FFI seam (it is one Julia module); this header IS the seam a
 * maintainer binds with `ccall` (julia/CFMMRouterAMD.jl).  Each entry
 * point names the reference code it replaces (paths relative to the reference root).
 *
 * Conventions
 *   - plain C, no exceptions: every call returns CFMM_OK (0) or a negative error code;
 *     cfmm_last_error() returns the message.  (Reference: ArgumentError from the
 *     constructors src/cfmms.jl:77-78, src/objectives.jl:54,97 -- nothing else.)
 *   - all values are IEEE binary64, computed in binary64 on the device.
 *   - token indices are 0-based int32 on this side (reference: 1-based Int64,
 *     src/cfmms.jl:10,16); the binding subtracts 1 when packing.
 *   - "pair" arrays are [m][2] row-major: R = {R1,R2}, Ai = {i1,i2}, w = {w1,w2},
 *     Delta = {D1,D2}, Lambda = {L1,L2} -- the reference's per-pool 2-vectors, contiguous.
 *   - host pointers are only read/written during the call; the library copies.
 *   - a cfmm_ctx is bound to one device and is not re-entrant; distinct contexts are
 *     independent.  Host-pointer calls are synchronous; *_dev calls are asynchronous on
 *     the context's stream.
 *   - pools are stored in SEGMENTS (one per cfmm_pools_add_* call with m > 0, each homogeneous
 *     in pool family; an empty batch is accepted and ignored).  Trade arrays are laid out segment after segment in call order, so a
 *     router whose cfmms vector is grouped by family keeps the reference's pool order.
 */
#ifndef CFMM_AMD_H
#define CFMM_AMD_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CFMM_OK 0
#define CFMM_ERR_INVALID_ARG (-1) /* bad pointer / size / pool data (the reference's ArgumentError) */
#define CFMM_ERR_HIP (-2)         /* a HIP runtime call failed; message carries hipGetErrorString */
#define CFMM_ERR_STATE (-3)       /* call sequence error (e.g. netflows before any sweep) */
#define CFMM_ERR_UNSUPPORTED (-4) /* valid request outside the implemented envelope */

#define CFMM_KIND_PRODUCT 0 /* ProductTwoCoin        src/cfmms.jl:101-140 */
#define CFMM_KIND_GEOMEAN 1 /* GeometricMeanTwoCoin  src/cfmms.jl:152-196 */
#define CFMM_KIND_UNIV3 2   /* UniV3/BoundedProduct  src/cfmms.jl:226-395 */

typedef struct cfmm_ctx cfmm_ctx;

/* ---- lifetime ------------------------------------------------------------------------ */

/* Router(objective, cfmms, n_tokens) -- src/router.jl:18-36: the device-side half of the
 * Router (pool store, trade buffers, v).  device_id is a HIP ordinal. */
int cfmm_ctx_create(int device_id, int32_t n_tokens, cfmm_ctx** out);
void cfmm_ctx_destroy(cfmm_ctx* ctx);

/* The same Router sharded over the GPUs of one node from ONE host thread / process:
 * the parallel axis of src/router.jl:39 (`Threads.@threads for i in 1:length(r.Δs)`) split into
 * n_devices contiguous blocks.  Every cfmm_pools_add_* batch is divided into contiguous blocks over
 * the devices (no pool is replicated); every host-pointer sweep (cfmm_find_arb, cfmm_eval, and
 * therefore cfmm_route: ONE L-BFGS-B drives all shards) stages v on every device, runs the shard
 * sweeps concurrently and adds the shards' {psi, acc} in device-list order on the host -- v comes
 * from the host and psi returns to it on every evaluation, so the all-reduce of a sharded route! is
 * n_devices * (n_tokens + 1) host additions; no peer mapping, no IPC, no RCCL, no torch.  Results
 * are independent of the option "multi_threads" (1, default: one host worker thread per device
 * launches and waits; 0: the calling thread launches on every device, then waits for each).
 * A device id may be listed more than once (several shards on one GPU: used by the 1-GPU tests).
 * Trade arrays keep the single-device layout (segment after segment, pools in upload order).
 * Not available on such a context: cfmm_set_stream, cfmm_sweep_dev, cfmm_trades_dev, cfmm_set_peers
 * (CFMM_ERR_UNSUPPORTED), and n_tokens > 8192.  For one process PER GPU use cfmm_ctx_create +
 * cfmm_set_peers / RCCL instead (cfmmrouter.jl_amd/dist.py). */
int cfmm_ctx_create_multi(int32_t n_devices, const int32_t* device_ids, int32_t n_tokens, cfmm_ctx** out);
int32_t cfmm_device_count(const cfmm_ctx* ctx); /* shards of the context (1 for cfmm_ctx_create) */

/* Message of the last failing call on ctx (ctx == NULL: last failing cfmm_ctx_create on
 * this thread).  Never NULL; valid until the next call on the same ctx/thread. */
const char* cfmm_last_error(const cfmm_ctx* ctx);
const char* cfmm_version(void);

/* Launch on a caller-owned hipStream_t (e.g. torch's current stream) instead of the
 * context's own non-blocking stream.  NULL means HIP's default (null) stream -- exactly the
 * handle given is used.  cfmm_reset_stream returns to the context's own stream. */
int cfmm_set_stream(cfmm_ctx* ctx, void* hip_stream);
int cfmm_reset_stream(cfmm_ctx* ctx);

/* Options (int64 values; unknown keys are CFMM_ERR_INVALID_ARG).  Results never depend on the tuning knobs beyond
 * summation-order rounding of psi / acc; trades are bit-identical under all of them.
 *   launch geometry   "block" (0 = auto | 512 | 1024 threads), "max_grid" (0 = auto), "bin_copies" (0 = auto, 1 = one
 *                     LDS netflow copy per block, 2 = one per wavefront), "fuse_segments" (default 1: all pool families
 *                     swept by one launch; 0: one launch per segment), "cost_geomean" / "cost_univ3" (cost of one
 *                     evaluation in tenths of a ProductTwoCoin one: how a fused launch divides its blocks; 10 / 10)
 *   data layout       "pack" (default 1: sweeps read an 8-byte {token pair, fee-table index} record instead of gamma +
 *                     Ai when a launch's distinct fees fit a 256-entry table), "compact_trades" (default 1: a
 *                     materialising sweep stores ONE 16-byte record per pool -- {+Delta1, Lambda2} or {-Delta2,
 *                     Lambda1}, a two-coin trade has one direction -- plus overflow rows for pools whose four values
 *                     do not fit that form; cfmm_get_trades* / cfmm_trades_dev return the reference's rows bit for bit),
 *                     "stream_stores" (trade records leave through 0 = auto: non-temporal stores when one sweep touches more
 *                     than the 256 MiB Infinity Cache -- the pool state then comes from HBM on every sweep and the trade lines
 *                     should not displace it: -4 % at 8M-16M pools -- and write-through stores otherwise; 1 = always
 *                     write-through (best while the market stays cache-resident between sweeps: route!); 2 = always
 *                     non-temporal: a caller that rotates over many markets, each of which fits the cache, says so: -2..-5 %
 *                     when the pool state really comes from HBM, +0..3 % when it does not),
 *                     "direct_small" (default 1: a context with one pool family and at most 2048 pools is swept by ONE block
 *                     that delivers {psi, acc} itself -- one kernel per evaluation instead of sweep + fold; 0: the general
 *                     two-launch geometry),
 *                     "univ3_heads" (default 1: multi-tick UniV3 walks decide their first four list ticks from a per-pool
 *                     head of rounded-down binary32 thresholds read with the pool's coalesced streams, and consult the
 *                     exact threshold array only for deeper walks or a price within 2^-23 of a threshold: same decisions,
 *                     same bits, 11 % fewer bytes; 0: always the exact array)
 *   arithmetic        "fast_math" (default 1: where every operand lies in [2^-150, 2^150] -- pool constants checked at
 *                     upload, prices by the host (host-pointer calls, cfmm_route) and again by every block as it stages
 *                     them; the library then launches kernels in which divisions and square roots run the compiler's
 *                     own correctly-rounded instruction sequences WITHOUT their range scaffolding, and divisions by a
 *                     price or a fee reuse a reciprocal refined once per token / fee tier: same bits, ~half the
 *                     instructions; the log-space GeometricMean form -- within 1e-12 of the reference either way --
 *                     also evaluates its exponential with an own < 1 ulp polynomial instead of the device library's;
 *                     anything outside the window runs the full-range arithmetic -- device-pointer sweeps, whose prices
 *                     the library cannot see, decide per block inside the launch (cfmm_sweep_dev);
 *                     0: the compiler's sequences and the library's exp everywhere), "dev_prices_in_window" (default 0;
 *                     1 = the caller vouches that the prices it passes to cfmm_sweep_dev lie in [2^-150, 2^150] -- the
 *                     check the library makes itself on host-pointer calls -- and device-pointer sweeps launch the fast
 *                     kernels alone (fewer registers than the kernels that carry both arithmetics); every block still
 *                     checks the prices it stages, and a broken promise yields NaN in every entry of {psi, acc}: an
 *                     error, never a wrong number), "geomean_exact" (1 = GeometricMeanTwoCoin
 *                     with pow in the reference's operation order instead of the default log-space form; both are within
 *                     1e-12 of the reference), "alternate" (default 1: consecutive evaluations walk every lane's tiles in
 *                     alternating directions, so that a sweep starts on the pool data the previous one left in the XCD's
 *                     L2 -- two fused evaluations at the same v then agree to summation-order rounding, every second one
 *                     bit for bit.  cfmm_find_arb and the final sweep of cfmm_route always walk forwards: find_arb!(r, v)
 *                     is a function of v alone, and cfmm_route restarts the alternation, so a route is a function of its
 *                     arguments alone; 0: always forwards, every sweep bit-identical)
 *   host boundary     "zero_copy" (default 1: host-pointer calls read v through mapped pinned memory), "host_flag"
 *                     (default 1: such a call ends when {psi, acc} have arrived in pinned host memory as self-validating
 *                     8-byte granules, which the library polls, instead of on the stream's completion signal),
 *                     "time_kernels" (see cfmm_kernel_times), "multi_threads" (multi-device contexts)
 *   route!            "armed" (default 1: cfmm_route enqueues evaluation k+1 while evaluation k runs; its blocks wait on
 *                     the device -- bounded by "arm_timeout_ms", default 2000 -- until the host has written the next price
 *                     vector straight into device memory through the PCIe BAR, which takes the launch latency off the
 *                     critical path of every evaluation; needs a large-BAR system, otherwise -- or with CFMM_AMD_ARMED=0
 *                     in the environment -- the evaluations are launched when their prices are ready; identical results
 *                     either way; also on cfmm_set_peers contexts and on multi-device contexts whose shards sit on
 *                     distinct devices), "stop_in_noise" (default 0 = the stopping rules of L-BFGS-B 3.0, the
 *                     reference's solver; 1 = additionally end the run when a line-search trial point differs from the
 *                     current dual value by no more than the factr tolerance: fewer evaluations, v* up to a decade
 *                     further from the reference's).
 * Environment: HIP_FORCE_DEV_KERNARG is set to 1 when the library is loaded unless already set (kernel arguments in
 * device memory: -10 % per step); CFMM_AMD_PEER_TIMEOUT_S (see cfmm_set_peers). */
int cfmm_set_option(cfmm_ctx* ctx, const char* key, int64_t value);
int cfmm_get_option(const cfmm_ctx* ctx, const char* key, int64_t* value);

/* ---- pool upload (struct definitions + constructors, packed once) --------------------- */

/* m x ProductTwoCoin(R, gamma, idx) -- src/cfmms.jl:101-111 (+ checks of :76-90).
 * R[m][2] > 0, gamma[m] > 0, Ai[m][2] distinct and in [0, n_tokens). */
int cfmm_pools_add_product(cfmm_ctx* ctx, int64_t m, const double* R, const double* gamma,
                           const int32_t* Ai);

/* m x GeometricMeanTwoCoin(R, w, gamma, idx) -- src/cfmms.jl:152-165.  w[m][2] > 0. */
int cfmm_pools_add_geomean(cfmm_ctx* ctx, int64_t m, const double* R, const double* w,
                           const double* gamma, const int32_t* Ai);

/* m x UniV3(current_price, lower_ticks, liquidity, gamma, Ai) -- src/cfmms.jl:226-245,
 * ragged ticks in CSR form: pool i owns lower_ticks/liquidity[tick_off[i] .. tick_off[i+1]).
 * lower_ticks strictly descending and > 0 per pool, liquidity >= 0, and
 * current_price <= lower_ticks[first] (the reference would index tick 0 otherwise, :235,:316).
 * current_tick is derived here exactly as :235 does.  A stand-alone BoundedProduct pool
 * (src/cfmms.jl:272-289) is a UniV3 with two ticks whose second liquidity is 0. */
int cfmm_pools_add_univ3(cfmm_ctx* ctx, int64_t m, const double* current_price, const double* gamma,
                         const int32_t* Ai, const int64_t* tick_off, const double* lower_ticks,
                         const double* liquidity);

int cfmm_pools_clear(cfmm_ctx* ctx);
int64_t cfmm_pools_count(const cfmm_ctx* ctx); /* length(r.cfmms) */
int32_t cfmm_n_tokens(const cfmm_ctx* ctx);    /* length(r.v) */

/* ---- the hot path --------------------------------------------------------------------- */

/* find_arb!(r::Router, v) -- src/router.jl:38-42, dispatching to src/cfmms.jl:130-140,
 * :185-196, :339-395 per segment.  Materialising sweep: writes every pool's Delta/Lambda
 * to device memory AND reduces psi = sum_i A_i(Lambda_i - Delta_i) and the dual scalar
 * acc = sum_i (Lambda_i - Delta_i)'v[A_i] in the same pass.  v: n_tokens host doubles. */
int cfmm_find_arb(cfmm_ctx* ctx, const double* v);

/* The evaluation route!'s closures fn / g! need (src/router.jl:73-86, :89-102) without the
 * O(m) trade write-back: psi_out[n_tokens] = pool part of G (== netflows at v),
 * *acc_out = pool part of fn.  Either output pointer may be NULL. */
int cfmm_eval(cfmm_ctx* ctx, const double* v, double* psi_out, double* acc_out);

/* r.Δs / r.Λs after find_arb! -- src/router.jl:7-8,40.  [m_total][2] each, segment order.
 * Requires a preceding cfmm_find_arb (cfmm_eval does not produce trades). */
int cfmm_get_trades(cfmm_ctx* ctx, double* Delta, double* Lambda);
/* Same, one segment: rows [first, first+count) of segment `seg`. */
int cfmm_get_trades_range(cfmm_ctx* ctx, int32_t seg, int64_t first, int64_t count, double* Delta,
                          double* Lambda);

/* update_reserves!(r) -- src/router.jl:127-132.  The reference's router method calls a per-pool
 * update_reserves!(c, Δ, Λ, v) that is defined nowhere (its own test is disabled, test/arb.jl:30-39);
 * implemented here is the update the routing problem prescribes (find_arb! docstring,
 * src/cfmms.jl:26-31: the pool ends at R + γΔ − Λ), applied IN PLACE ON THE DEVICE from the trades of
 * the latest materialising sweep (cfmm_find_arb, cfmm_route; consumed by this call):
 *   ProductTwoCoin / GeometricMeanTwoCoin:  R <- (R + γΔ) − Λ            (one kernel, no host traffic)
 *   UniV3 / BoundedProduct: the state is the price.  A pool that traded moves to the internal price
 *     P = p/γ (price falling, src/cfmms.jl:361) or γ·p (price rising, :381), p = v₁/v₂, clamped to the
 *     first tick's upper price; its tick constants are re-derived as at upload (:294-313).  That is
 *     exactly the pool R + γΔ − Λ tick by tick (tested), and needs the prices of the trades: the
 *     materialising sweep must have been a host-pointer call.
 * Afterwards a sweep at the same prices finds no arbitrage in any pool. */
int cfmm_update_reserves(cfmm_ctx* ctx);
/* Current reserves R[m][2] of a two-coin segment / current prices [m] of a UniV3 segment. */
int cfmm_get_reserves(cfmm_ctx* ctx, int32_t seg, double* R);
int cfmm_get_prices(cfmm_ctx* ctx, int32_t seg, double* current_price);

/* netflows!(psi, r) -- src/router.jl:111-119, for the most recent sweep. */
int cfmm_netflows(cfmm_ctx* ctx, double* psi);
/* the `acc` of fn (src/router.jl:79-83) for the most recent sweep. */
int cfmm_dual_value(cfmm_ctx* ctx, double* acc);

/* ---- device-resident variants (stream / RCCL interop; no host round trip) -------------- */

/* d_v: n_tokens device doubles (must be finite and > 0, src/cfmms.jl:129: the library cannot validate device
 * memory; non-positive prices give undefined trades).  The library cannot see these prices, so with "fast_math" (default)
 * the launch carries BOTH arithmetics and every block picks from the prices it stages: inside [2^-150, 2^150] the fast
 * one, anything else -- NaN included -- the compiler's full-range sequences, under which a NaN price propagates into the
 * trades and psi of every pool that touches the token exactly as the reference's arithmetic does.  The call never refuses
 * and never changes the context's state.  (Round 4 launched the fast kernels on trust and reported a refusal on a LATER
 * call; host-pointer calls and cfmm_route choose the kernel per call, from the prices they are given.)
 * d_out: n_tokens+1 device doubles = {psi..., acc}: the
 * buffer a sharded run all-reduces (one collective per evaluation).  Asynchronous on the
 * context's stream.  materialize != 0: also write Delta/Lambda (find_arb! semantics). */
int cfmm_sweep_dev(cfmm_ctx* ctx, const double* d_v, double* d_out, int materialize);
/* Device addresses of the trade rows of the latest materialising sweep ([m_total][2] doubles each, the
 * reference's Delta / Lambda layout), valid until pools change.  With "compact_trades" (default) these
 * are expanded copies written on the context's stream by this call: call it again after a later sweep. */
int cfmm_trades_dev(cfmm_ctx* ctx, const double** d_delta, const double** d_lambda);

/* With option "time_kernels"=1 every sweep launch is bracketed by hipEvents on the launch
 * stream.  Returns and resets: number of timed sweep-kernel launches, their summed
 * duration, and the same for the partial-reduction kernel. */
int cfmm_kernel_times(cfmm_ctx* ctx, int64_t* sweep_launches, double* sweep_ms,
                      int64_t* reduce_launches, double* reduce_ms);

/* ---- sharded runs: the all-reduce of {psi, acc} over xGMI peer mappings, inside the fold launch -- */

/* Sharded operation of a context (one process per GPU): after this call EVERY sweep of the context --
 * cfmm_find_arb, cfmm_eval, cfmm_route, and cfmm_sweep_dev -- returns the psi / acc of the WHOLE
 * market while the context stores only this rank's shard: the launch that folds the partial rows
 * also performs the all-reduce over the given symmetric buffers (block b publishes its columns as
 * self-validating 8-byte granules {sequence tag, 32 payload bits} in this rank's buffer -- no flag, no
 * fence -- and adds the same columns of every peer in rank order: every rank obtains bit-identical
 * results).  Buffer layout, per rank, with count = n_tokens + 1, zero-initialised before first use:
 *     [2][count][2] uint64 granules  =  cfmm_peer_buffer_bytes(n_tokens) bytes
 * (peer_buffers[p] = address of rank p's buffer mapped into THIS process; cfmm_peer_buffer_alloc / _open below, or any
 * other symmetric allocation).  It replaces nothing in the reference (which has no distributed path): it is the
 * collective of SURVEY 8e.  Every rank
 * must issue the same sequence of sweeps (route! does: all ranks take bit-identical L-BFGS-B steps).
 * seq = number of sharded sweeps already performed on these buffers (0 for fresh ones).  A rank
 * waits up to CFMM_AMD_PEER_TIMEOUT_S seconds (environment, default 30) for a peer, then the output
 * is NaN and host-pointer calls fail with CFMM_ERR_STATE.  world = 0 switches sharding off. */
int cfmm_set_peers(cfmm_ctx* ctx, const uint64_t* peer_buffers, int32_t world, int32_t rank, uint64_t seq);

/* The symmetric buffers of cfmm_set_peers without any framework: every rank allocates its buffer
 * (fine-grained device memory, sized and zeroed for the context's n_tokens), publishes the 64-byte IPC handle through whatever channel
 * its launcher has (MPI, torch.distributed, a file), and maps the other ranks' buffers from their handles
 * (hipIpcGetMemHandle / hipIpcOpenMemHandle; needs HSA_ENABLE_IPC_MODE_LEGACY=0 on this driver stack).
 * The pointers go to cfmm_set_peers (own rank: the pointer from _alloc).  _close unmaps a peer's buffer,
 * _free releases the own one (after every peer has closed it). */
#define CFMM_IPC_HANDLE_BYTES 64
int64_t cfmm_peer_buffer_bytes(int32_t n_tokens);
int cfmm_peer_buffer_alloc(cfmm_ctx* ctx, uint64_t* d_buf, unsigned char handle[CFMM_IPC_HANDLE_BYTES]);
int cfmm_peer_buffer_open(cfmm_ctx* ctx, const unsigned char handle[CFMM_IPC_HANDLE_BYTES], uint64_t* d_peer);
int cfmm_peer_buffer_close(cfmm_ctx* ctx, uint64_t d_peer);
int cfmm_peer_buffer_free(cfmm_ctx* ctx, uint64_t d_buf);

/* ---- sharded runs through RCCL: north_star's "RCCL all-reduce of psi and grad g over xGMI per outer iteration" ---- */

/* The same contract as cfmm_set_peers -- after the call EVERY sweep of the context (cfmm_find_arb, cfmm_eval, cfmm_route,
 * cfmm_sweep_dev) returns the psi / acc of the WHOLE market while the context stores this rank's shard of the axis the
 * reference threads over (src/router.jl:39) -- with the collective done by RCCL: behind every sweep's row fold the library
 * enqueues ncclAllReduce(d_out, d_out, n_tokens + 1, ncclDouble, ncclSum, comm, stream) on the context's stream (one
 * 4 KB collective per evaluation: latency, never link bandwidth).  No torch, no Python, no IPC handles: a Julia / C host
 * shards a market in three calls --
 *     rank 0:      cfmm_rccl_unique_id(id);          then broadcasts the 128 bytes with whatever its launcher has
 *     every rank:  cfmm_rccl_init_rank(ctx, id, world, rank);   (collective: returns when all ranks have joined)
 * -- or hands over a communicator it created itself (cfmm_set_rccl_comm: the caller keeps ownership; NULL switches the
 * exchange off).  RCCL sums in its own (ring / tree) order: every rank receives the SAME bits (the lockstep L-BFGS-B of
 * cfmm_route relies on it), but not the rank-ordered sum of cfmm_set_peers.  cfmm_set_peers (fold + exchange in ONE
 * launch: no second launch on the critical path of an evaluation) stays the fast path; this is the portable one
 * (cfmmrouter.jl_amd/dist.py tries the peer exchange first, then this, then torch.distributed).  One exchange at a time per context (CFMM_ERR_STATE otherwise);
 * evaluations are launched when their prices are ready (no pre-arming); n_tokens <= 8192.  RCCL is resolved at first use,
 * ALL entry points from ONE image: the one named by CFMM_AMD_RCCL_LIB (environment; path or soname), else the process's
 * global scope if ncclAllReduce is visible there, else librccl.so.1 -- CFMM_ERR_UNSUPPORTED if there is none.
 * cfmm_set_rccl_comm takes a foreign communicator only in the first two cases (the caller's own RCCL: a communicator must
 * meet the ncclAllReduce of the image that created it); cfmm_rccl_init_rank works in all three.
 * Failure semantics: RCCL's, not cfmm_set_peers'.  The all-reduce has NO time limit of its own -- a rank that fails before
 * it enqueues its ncclAllReduce (a launch error, an exception between two evaluations of a host-driven loop) leaves the
 * other ranks waiting in their stream synchronisation, exactly like any NCCL program; bound it with RCCL's own watchdog /
 * ncclCommAbort from the launcher, or use cfmm_set_peers, whose waits are bounded (CFMM_AMD_PEER_TIMEOUT_S).  A sweep whose
 * all-reduce could not be enqueued returns the error AFTER its local sweep and fold were launched: trades of a materialising
 * sweep exist, {psi, acc} do not (cfmm_netflows: CFMM_ERR_STATE). */
#define CFMM_RCCL_ID_BYTES 128
int cfmm_rccl_unique_id(unsigned char id[CFMM_RCCL_ID_BYTES]);
int cfmm_rccl_init_rank(cfmm_ctx* ctx, const unsigned char id[CFMM_RCCL_ID_BYTES], int32_t world, int32_t rank);
int cfmm_set_rccl_comm(cfmm_ctx* ctx, void* nccl_comm /* ncclComm_t, or NULL */);

/* ---- route! without an interpreter in the loop ----------------------- */

#define CFMM_OBJ_LINEAR_NONNEGATIVE 0 /* LinearNonnegative(c)      src/objectives.jl:51-79 */
#define CFMM_OBJ_BASKET_LIQUIDATION 1 /* BasketLiquidation(i, Din) src/objectives.jl:92-129 */

typedef struct cfmm_route_info {
    double f;            /* dual value g(v*) */
    double proj_grad;    /* max-norm of the projected gradient at v* */
    int32_t iterations;  /* L-BFGS-B iterations */
    int32_t evaluations; /* fn/g! evaluations == device sweeps inside the solver */
    int32_t sweeps;      /* all device sweeps incl. prologue and epilogue (src/router.jl:104,107) */
    int32_t status;      /* 0 pgtol, 1 factr, 2 maxiter, 3 maxfun, 4 line search, 5 non-finite f */
    double sweep_seconds;  /* wall time spent inside device sweeps (launch to results on the host), all sweeps */
    double total_seconds;  /* wall time of the whole call; the difference is the host-side L-BFGS-B */
} cfmm_route_info;

/* route!(r; v, m, factr, pgtol, maxfun, maxiter) -- src/router.jl:58-108, with the external
 * LBFGSB.jl solver (src/router.jl:60,105) replaced by this library's own L-BFGS-B
 * (csrc/lbfgsb.cpp, written from the published algorithm).  objective_vec is `c`
 * (LinearNonnegative) or `Din` (BasketLiquidation, objective_index = 0-based output token);
 * v0 may be NULL (the reference's default ones(n)/n, :62).  On return v_out[n] = r.v,
 * psi_out[n] = netflows(r), trades are materialised at v* (cfmm_get_trades).  Pass
 * maxfun = maxiter = 15000, m = 5, factr = 1e1, pgtol = 1e-5 for the reference's defaults. */
int cfmm_route(cfmm_ctx* ctx, int32_t objective_kind, const double* objective_vec, int32_t objective_index,
               const double* v0, int32_t m, double factr, double pgtol, int32_t maxfun, int32_t maxiter,
               double* v_out, double* psi_out, cfmm_route_info* info);

/* The solver alone on a caller-supplied objective (used by the CPU tests to compare it with
 * SciPy's L-BFGS-B).  nbd[i]: 0 free, 1 lower, 2 both, 3 upper.  fg returns f and fills g.
 * boxed_from_nbd != 0: like the Fortran code, treat the problem as "boxed" (unit first step) when
 * every nbd[i] == 2 even if a bound is infinite -- what the reference's call does
 * (src/router.jl:67-70) and what cfmm_route uses; 0: infinite bounds are no bounds (SciPy). */
typedef double (*cfmm_fg_callback)(void* user, const double* x, double* g);
int cfmm_lbfgsb_minimize(int32_t n, double* x, const double* lower, const double* upper, const int32_t* nbd,
                         cfmm_fg_callback fg, void* user, int32_t m, double factr, double pgtol, int32_t maxfun,
                         int32_t maxiter, int32_t boxed_from_nbd, cfmm_route_info* info);

/* Number of segments and their description (kind, pool count, launch geometry). */
int32_t cfmm_segment_count(const cfmm_ctx* ctx);
int cfmm_segment_info(const cfmm_ctx* ctx, int32_t seg, int32_t* kind, int64_t* m, int32_t* block,
                      int32_t* grid);

/* ==== EXPERIMENTAL -- beyond the reference's verbs ====================================================================
 * Everything ABOVE this line is the drop-in surface: the reference's own verbs (Router, find_arb!, route!, netflows,
 * update_reserves!, the objectives) and what sharding them needs.  What follows is NOT in CFMMRouter.jl, may change, and
 * is not needed to replace the reference's path:
 *   - cfmm_polish (below),
 *   - the option "stop_in_noise" (cfmm_set_option; default 0 = the stopping rules of L-BFGS-B 3.0).
 * (Test hooks are not in this library: `make -C cfmmrouter.jl_amd/csrc hooks` builds libcfmm_amd_hooks.so with
 * -DCFMM_TEST_HOOKS for tests/test_gpu_armed.py.)
 * ====================================================================================================================== */

/* Tighten a route!'s result beyond what L-BFGS-B's stopping rules can (NOT part of the reference: its route! ends where
 * LBFGSB.jl ends, src/router.jl:105-107).  L-BFGS-B's line search compares dual VALUES, whose rounding noise -- a sum
 * over all pools -- hides decreases below ~1e-15 relative; on interior optima that leaves a stationarity residual of
 * ~1e-6 max|psi|, on the reference's side as well.  cfmm_polish continues from v (in: a route!'s v*, out: the polished
 * point) with a projected chord-Newton iteration on the optimality conditions of the dual problem of src/router.jl:58-108,
 *     G_j(v) = 0 for l_j < v_j,   G_j(v) >= 0 for v_j = l_j,   G = grad f(objective, v) + psi(v),  l = lower_limit(objective),
 * using GRADIENTS only: J = forward-difference Jacobian of G at the start (n_tokens + 1 fused sweeps, step rel_step * v_j;
 * pass rel_step <= 0 for 1e-7), free set F = {j : not (v_j = l_j and G_j > 0)}, v_F <- max(l_F, v_F - J_FF^-1 G_F); steps that
 * do not reduce the residual are halved, the best iterate is kept, at most max_iters iterations (8 is plenty).  Ends with
 * find_arb!(r, v) like route! does: psi_out[n] = netflows(r), trades materialised at the polished point.  Objective
 * arguments as for cfmm_route.  The Python mirror's polish_ (cfmmrouter.jl_amd/router.py) is the same iteration.  Works on
 * multi-device contexts; on a cfmm_set_peers context every rank must make the same call (its sweeps are collective, and
 * all ranks see bit-identical psi, hence take identical steps). */
typedef struct cfmm_polish_info {
    double residual0;    /* max |G_F| before (and -G_j where a variable on its bound has G_j < 0) */
    double residual;     /* ... after */
    int32_t iterations;
    int32_t sweeps;      /* device sweeps of the call, Jacobian and final find_arb! included */
    double total_seconds;
} cfmm_polish_info;
int cfmm_polish(cfmm_ctx* ctx, int32_t objective_kind, const double* objective_vec, int32_t objective_index,
                double* v, int32_t max_iters, double rel_step, double* psi_out, cfmm_polish_info* info);

#ifdef __cplusplus
}
#endif
#endif
