// kernels.hip.h -- HIP kernels of the particle-filter hot path (gfx950, wave64).
//
// Hot path of the reference, per observation (R/particle_filter_core.R:123-246):
//   transition_fn -> weight_fn -> max/exp/sum normalise -> loglik, ESS ->
//   resample_fn (src/resampling.cpp) -> gather -> state estimate.
// Kernel map of the MULTI-LAUNCH path (all on the context's stream; 4 launches per observation with resampling up to 2^20 particles,
// 6 above; the fused path -- fused.hip.h, ONE launch per observation -- runs the same *_block bodies):
//   k_step / k_step_sir   transition_fn + weight_fn + per-block (max, sum exp, sum exp^2) partials, grid max by atomics   :127,:177-183
//   k_weights             = k_local<W, from_lw>: w = exp(lw - max)/sum (:205-207); loglik / ESS / resample decision from the
//                         partials (:208-218); block records of the exact sequential sum(w)          src/resampling.cpp:20
//   k_local<P>(+resolve)  every workgroup resolves the W records itself -> exact total; block records of cumsum(w / total)
//                                                                                                     src/resampling.cpp:24-25
//   k_apply<kind>(+resolve) every workgroup resolves the P records up to itself -> its exact incoming state; exact cum_sum ->
//                         output counts -> ancestors -> particles[indices, ]                          src/resampling.cpp:28-37, R/resampling.R:40
//   k_resolve_all<W|P>    grids above 512 blocks: one 1024-thread workgroup resolves a pass and emits every block's state
//                         (k_resolve: the round-1 resolver, behind option inkernel_resolve = 0)
//   k_multinomial         inverse-CDF search on the exact cum_sum                                src/resampling.cpp:11 (distributional)
//   k_multinomial_r       Rcpp::sample's own algorithm on R's unif_rand() stream (parity mode)   src/resampling.cpp:11
//   k_carry               no resample at this observation: carry particles, sum(x * w)           :238
//   k_move                resample_move_filter's random-walk Metropolis move                     :226-234
//   k_bsum, k_plan        stand-alone resampler front end (validation, approximate block prefixes)
//   k_pf_batch            many small filters per launch: one workgroup = one whole filter (all T observations on chip),
//                         built from the same *_block bodies as the kernels above
// mv.hip.h: the multivariate linear-Gaussian family's model kernels; multi.hip.h: K filters per launch (blockIdx.y = filter).
// Particles are SoA: x[dim][N] doubles; weights/log-weights [N]; ancestors int32.
//
// All arithmetic that the reference defines is done in fp64 with contraction
// off (the build passes -ffp-contract=off): a*b+c is two roundings, as in R.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <math.h>
#include "seqsum.h"
#include "rng.h"

namespace bssm {

#ifndef BSSM_NT                  // build-time shape of a scan workgroup (make NT=.. EL=..): NT threads x EL terms = EB
#define BSSM_NT 256
#endif
#ifndef BSSM_EL
#define BSSM_EL 8
#endif
constexpr int NT = BSSM_NT;      // threads per workgroup (NWV waves)
constexpr int EL = BSSM_EL;      // terms per thread in the scan kernels
constexpr int NWV = NT / 64;
static_assert(NT * EL == 2048 && (NT == 256 || NT == 512) , "a scan workgroup covers 2048 terms with 4 or 8 waves");
constexpr int EB = NT * EL;      // 2048 terms per workgroup
constexpr int MAXB = 2048;       // max scan workgroups  => N <= 2^22 per filter

// The bulk stores of a kernel (resampled particles, normalised weights) are streaming stores: the data is next read by another
// kernel, from other XCDs, and a kernel ends when its stores have drained -- non-temporal ones drain ~0.5 us sooner per kernel.
typedef double v2d_nt __attribute__((ext_vector_type(2)));
// Build-time store flavour of the three bulk outputs (experiment switches; make ST_W=.. ST_X=.. ST_S=..):
//   0 = non-temporal (streaming), 1 = sc1 write-through (leaves nothing dirty in L2 for the end-of-kernel write-back), 2 = plain
#ifndef BSSM_ST_W
#define BSSM_ST_W 0
#endif
#ifndef BSSM_ST_X
#define BSSM_ST_X 1
#endif
#ifndef BSSM_ST_S
#define BSSM_ST_S 2
#endif
typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
template <int F>
__device__ __forceinline__ void bulk_store16(double* p /* 16-byte aligned */, double a, double b)
{
    if constexpr (F == 0) { v2d_nt q; q.x = a; q.y = b; __builtin_nontemporal_store(q, reinterpret_cast<v2d_nt*>(p)); }
    else if constexpr (F == 1) {
        const unsigned long long ua = (unsigned long long)__double_as_longlong(a), ub = (unsigned long long)__double_as_longlong(b);
        u32x4_t v; v.x = (unsigned)ua; v.y = (unsigned)(ua >> 32); v.z = (unsigned)ub; v.w = (unsigned)(ub >> 32);
        asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" :: "v"(p), "v"(v) : "memory");
    } else { double2 q; q.x = a; q.y = b; *reinterpret_cast<double2*>(p) = q; }
}
template <int F>
__device__ __forceinline__ void bulk_store8(double* p, double a)
{
    if constexpr (F == 0) __builtin_nontemporal_store(a, p);
    else if constexpr (F == 1) __hip_atomic_store(p, a, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else *p = a;
}
constexpr int MODE_W = 0;        // terms are the weights themselves        (total = sum(w))
constexpr int MODE_P = 1;        // terms are prob = w / total              (cum = cumsum(prob))

constexpr int PLAN_RESAMPLE_ONLY = 0;   // stand-alone resampler
constexpr int PLAN_PF = 1;              // filter: weights of an observation
constexpr int PLAN_AUX = 2;             // APF first stage (R/particle_filter_core.R:152-155)

constexpr uint32_t FLAG_NEGATIVE = 1u, FLAG_ZERO_SUM = 2u, FLAG_NONFINITE = 4u;

#define BSSM_LN_SQRT_2PI 0.918938533204672741780329736406

struct BlockRec { Rec prefix; int32_t tail_from; int32_t nside; };   // 64 bytes
// Blocks with several boundaries publish them, in lane order, next to the block record: the resolver
// steps through  prefix, e[0].leaf, e[0].post, e[1].leaf, ...  with the exact state.
constexpr int MAXBND = 64;       // boundaries handled per block before the literal fallback
struct SideEntry { Rec leaf; int64_t d0, d1; uint64_t post_base; int32_t lo, hi; int32_t lane; int32_t pad; double terms[EL]; };  // 160 bytes
struct SideList { SideEntry e[MAXBND]; };
// Block records live in HBM as FOUR PLANES of 16-byte pieces (piece q of block b at plane q, index b; plane stride
// BREC_STRIDE blocks): the grid-level resolve has thread t fetch the records of blocks 2t, 2t+1, and with whole 64-byte
// records side by side every lane of such a load touched its own cache line (512 tag look-ups per wave for 8 KiB); by
// planes a wave's load covers 2 KiB contiguously.  `stride` 1 = one record on its own (the batched kernel's local record).
constexpr int BREC_STRIDE = MAXB;
static_assert(sizeof(BlockRec) == 64, "four 16-byte pieces");
__device__ __forceinline__ void store_brec(BlockRec* __restrict__ base, int b, const BlockRec& r, int stride = BREC_STRIDE)
{
    uint4* u = reinterpret_cast<uint4*>(base);
    const uint4* q = reinterpret_cast<const uint4*>(&r);
#pragma unroll
    for (int k = 0; k < 4; k++) u[(size_t)k * stride + b] = q[k];
}
__device__ __forceinline__ BlockRec load_brec(const BlockRec* __restrict__ base, int b, int stride = BREC_STRIDE)
{
    BlockRec r;
    const uint4* u = reinterpret_cast<const uint4*>(base);
    uint4* q = reinterpret_cast<uint4*>(&r);
#pragma unroll
    for (int k = 0; k < 4; k++) q[k] = u[(size_t)k * stride + b];
    return r;
}

// Per-run scalars living in HBM; written by single-workgroup kernels, read by all.
struct DevState {
    double loglike;
    double lse_max, lse_sum;
    double ess;
    uint64_t total_bits;          // exact sequential sum(weights), bit pattern
    int32_t do_resample;          // this observation
    int32_t dead;                 // 0 alive; i = all log-weights < -1e8 at observation i
    uint32_t flags;               // FLAG_*
    int32_t res_calls;            // resample calls made so far
    int32_t cur_call;             // index of the resample call in flight
    int32_t debug_stop;           // dev tool: kernels return after stage N (0 = run everything)
    int32_t out_lo, out_hi;       // outputs [out_lo, out_hi) produced by this launch's blocks in the last k_apply (particle-block sharding)
    long long stat_hard_blocks, stat_serial_walks, stat_literal_terms;
    long long stamps[4][16];      // dev tool: clock64() at stage boundaries (debug_stop == 99)
};

// dev tool: debug_stop 99 stamps a typical block (block 100 of a large grid), 98 the head block (block 0)
#ifdef BSSM_DEV_STAMPS
#define BSSM_STAMP(st, dbg, row, col, cond) do { if (((dbg) == 99 || (dbg) == 98) && (cond)) (st)->stamps[row][col] = clock64(); } while (0)
#else
#define BSSM_STAMP(st, dbg, row, col, cond) do { } while (0)      // (the stamps sit in the kernels' hot instruction stream: a dev build only, make DEV=1)
#endif

// ---------------------------------------------------------------------------
// small block-level helpers (wave64 shuffles, then LDS across the 4 waves)
// ---------------------------------------------------------------------------
// Cross-lane moves by DPP (data-parallel primitives: a VALU operand modifier, a few cycles) instead of
// ds_bpermute shuffles (an LDS-crossbar round trip each).  gfx9-family controls: row_shr:n shifts inside
// a row of 16 lanes; row_bcast:15 / row_bcast:31 hand the last lane of a row / of the lower half-wave to
// the following rows; wave_shr:1 shifts the whole wave64 by one lane.  Lanes without a source keep `old`.
constexpr int DPP_SHR1 = 0x111, DPP_SHR2 = 0x112, DPP_SHR4 = 0x114, DPP_SHR8 = 0x118;
constexpr int DPP_BCAST15 = 0x142, DPP_BCAST31 = 0x143, DPP_WAVE_SHR1 = 0x138;

template <int CTRL, int RM>
__device__ __forceinline__ int dpp_i32(int old, int src) { return __builtin_amdgcn_update_dpp(old, src, CTRL, RM, 0xf, false); }
template <int CTRL, int RM>
__device__ __forceinline__ int64_t dpp_i64(int64_t old, int64_t src)
{
    const int lo = dpp_i32<CTRL, RM>((int)(uint32_t)(uint64_t)old, (int)(uint32_t)(uint64_t)src);
    const int hi = dpp_i32<CTRL, RM>((int)(uint32_t)((uint64_t)old >> 32), (int)(uint32_t)((uint64_t)src >> 32));
    return (int64_t)(((uint64_t)(uint32_t)hi << 32) | (uint32_t)lo);
}
template <int CTRL, int RM>
__device__ __forceinline__ double dpp_f64(double old, double src)
{
    return __longlong_as_double(dpp_i64<CTRL, RM>(__double_as_longlong(old), __double_as_longlong(src)));
}

// inclusive wave64 scans / reductions; every lane must be active
#define BSSM_WAVE_SCAN_STEPS(STEP) STEP(DPP_SHR1, 0xf) STEP(DPP_SHR2, 0xf) STEP(DPP_SHR4, 0xf) STEP(DPP_SHR8, 0xf) STEP(DPP_BCAST15, 0xa) STEP(DPP_BCAST31, 0xc)

__device__ __forceinline__ double wave_incl_sum(double x)
{
#define STEP(C, R) x = dpp_f64<C, R>(0.0, x) + x;
    BSSM_WAVE_SCAN_STEPS(STEP)
#undef STEP
    return x;
}
__device__ __forceinline__ double wave_incl_max(double x)
{
#define STEP(C, R) x = fmax(dpp_f64<C, R>(-INFINITY, x), x);
    BSSM_WAVE_SCAN_STEPS(STEP)
#undef STEP
    return x;
}
__device__ __forceinline__ int64_t wave_incl_min_i64(int64_t x)
{
#define STEP(C, R) { const int64_t y = dpp_i64<C, R>(x, x); x = y < x ? y : x; }
    BSSM_WAVE_SCAN_STEPS(STEP)
#undef STEP
    return x;
}
__device__ __forceinline__ int64_t wave_incl_max_i64(int64_t x)
{
#define STEP(C, R) { const int64_t y = dpp_i64<C, R>(x, x); x = y > x ? y : x; }
    BSSM_WAVE_SCAN_STEPS(STEP)
#undef STEP
    return x;
}
__device__ __forceinline__ double lane63_f64(double x)
{
    const long long b = __double_as_longlong(x);
    const int lo = __builtin_amdgcn_readlane((int)(uint32_t)(uint64_t)b, 63), hi = __builtin_amdgcn_readlane((int)(uint32_t)((uint64_t)b >> 32), 63);
    return __longlong_as_double((long long)(((uint64_t)(uint32_t)hi << 32) | (uint32_t)lo));
}
__device__ __forceinline__ int64_t lane63_i64(int64_t b)
{
    const int lo = __builtin_amdgcn_readlane((int)(uint32_t)(uint64_t)b, 63), hi = __builtin_amdgcn_readlane((int)(uint32_t)((uint64_t)b >> 32), 63);
    return (int64_t)(((uint64_t)(uint32_t)hi << 32) | (uint32_t)lo);
}
// wave totals, valid in every lane
__device__ __forceinline__ double wave_sum(double v) { return lane63_f64(wave_incl_sum(v)); }
__device__ __forceinline__ double wave_max(double v) { return lane63_f64(wave_incl_max(v)); }
// pairwise sum of NW per-wave partials: (p0 + p1) + (p2 + p3) for four waves
template <int NW>
__device__ __forceinline__ double tree_sum(const double* p)
{
    if constexpr (NW == 1) return p[0];
    else return tree_sum<NW / 2>(p) + tree_sum<NW / 2>(p + NW / 2);
}
// result valid in every thread
__device__ __forceinline__ double block_sum(double v, double* sh4)
{
    v = wave_sum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sh4[threadIdx.x >> 6] = v;
    __syncthreads();
    return tree_sum<NWV>(sh4);
}
__device__ __forceinline__ double block_max(double v, double* sh4)
{
    v = wave_max(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sh4[threadIdx.x >> 6] = v;
    __syncthreads();
    double m = sh4[0];
#pragma unroll
    for (int i = 1; i < NWV; i++) m = fmax(m, sh4[i]);
    return m;
}
template <int NW>
__device__ __forceinline__ double block_sum_n(double v, double* sh)
{
    v = wave_sum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
    __syncthreads();
    double s = 0.0;
#pragma unroll
    for (int i = 0; i < NW; i++) s += sh[i];
    return s;
}
// two sums at once (one barrier pair instead of two); each is added in the same order as block_sum_n adds it.  sh: 2 NW doubles
template <int NW>
__device__ __forceinline__ void block_sum2_n(double& a, double& b, double* sh)
{
    a = wave_sum(a); b = wave_sum(b);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) { sh[threadIdx.x >> 6] = a; sh[NW + (threadIdx.x >> 6)] = b; }
    __syncthreads();
    double s = 0.0, q = 0.0;
#pragma unroll
    for (int i = 0; i < NW; i++) { s += sh[i]; q += sh[NW + i]; }
    a = s; b = q;
}
template <int NW>
__device__ __forceinline__ double block_max_n(double v, double* sh)
{
    v = wave_max(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
    __syncthreads();
    double m = sh[0];
#pragma unroll
    for (int i = 1; i < NW; i++) m = fmax(m, sh[i]);
    return m;
}
// exclusive prefix of x over the block (re-associated: only an approximation
// of the sequential prefix is needed here)
__device__ __forceinline__ double block_excl_scan(double x, double* sh4)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const double inc = wave_incl_sum(x);
    const double exc = dpp_f64<DPP_WAVE_SHR1, 0xf>(0.0, inc);        // lane 0 keeps 0
    __syncthreads();
    if (lane == 63) sh4[wave] = inc;
    __syncthreads();
    double pre = 0.0;
    for (int i = 0; i < wave; i++) pre += sh4[i];
    return pre + exc;
}

// ---------------------------------------------------------------------------
// Built-in models: the reference's user closures restated for the device.
// theta = (phi, sigma_x, sigma_y).
// ---------------------------------------------------------------------------
struct ModelPar { double phi, sx, sy, log_sy; double n_total, s0, i0, lgy; };   // SIR: phi = lambda, sx = gamma, lgy = lgamma(y+1)

// rnorm(n, mu, sd) = mu + sd * z   (R nmath/rnorm.c)
__device__ __forceinline__ double r_rnorm(double mu, double sd, double z) { return mu + sd * z; }
// dnorm(x, mu, sd, log=TRUE)       (R nmath/dnorm.c)
__device__ __forceinline__ double r_dnorm_log(double x, double mu, double sd, double log_sd)
{
    double z = (x - mu) / sd;
    if (!isfinite(z)) return -INFINITY;
    z = fabs(z);
    return -(BSSM_LN_SQRT_2PI + 0.5 * z * z + log_sd);
}

template <int MODEL> struct Model;
template <> struct Model<0> {   // BSSM_MODEL_LG   tests/testthat/test-pmmh_tuning.R:163-173
    static constexpr int D = 1;
    __device__ static double transition(double x, double z, const ModelPar& p) { return p.phi * x + r_rnorm(0.0, p.sx, z); }
    __device__ static double forecast(double x, const ModelPar& p) { return p.phi * x; }
};
template <> struct Model<1> {   // BSSM_MODEL_AR1SIN   README.md:137-146
    static constexpr int D = 1;
    __device__ static double transition(double x, double z, const ModelPar& p) { return p.phi * x + sin(x) + r_rnorm(0.0, p.sx, z); }
    __device__ static double forecast(double x, const ModelPar& p) { return p.phi * x + sin(x); }
};

// Stochastic SIR (vignettes/articles/stochastic-sir-model.Rmd:152-176, 285-310): state (s, i), one Gillespie day
// per transition, y ~ Poisson(i).  theta = (lambda, gamma, n_total, s0, i0).
struct Sir {
    // epidemic_step (:152-176) for ONE particle; draws keyed (particle, event) inside call `call`
    __device__ static void transition(double& s, double& i, const ModelPar& p, PhiloxKey key, uint32_t call, uint32_t particle)
    {
        if (i == 0.0) return;                                   // transition_fn: if (i == 0) return(c(s, i))  (:296-298)
        double t = 0.0;
        uint32_t ev = 0;
        while (t < 1.0 && i > 0.0 && ev < (1u << 20)) {
            const double rate_infection = (p.phi / p.n_total) * s * i;
            const double rate_removal = p.sx * i;
            const double rate_total = rate_infection + rate_removal;
            if (rate_total <= 0.0) break;
            u32x4 c; c.x = particle; c.y = call; c.z = DRAW_TRANS | (ev << 8); c.w = key.stream;
            const u32x4 r = philox4x32_10(c, key.k0, key.k1);
            const double dt = -log(u01_from_bits(r.x, r.y)) / rate_total;      // rexp(1, rate_total)
            if (t + dt > 1.0) break;
            t = t + dt;
            if (u01_from_bits(r.z, r.w) < rate_infection / rate_total) { s = s - 1.0; i = i + 1.0; }
            else { i = i - 1.0; }
            ev++;
        }
    }
    // dpois(y, lambda, log = TRUE) = y log(lambda) - lambda - lgamma(y + 1)   (R's dpois_raw evaluates the same
    // quantity through stirlerr/bd0; the two agree to ~1e-14 relative)
    __device__ static double dpois_log(double y, double lambda, double lgy)
    {
        if (lambda <= 0.0) return (y == 0.0 && lambda == 0.0) ? 0.0 : -INFINITY;
        if (y == 0.0) return -lambda;
        return y * log(lambda) - lambda - lgy;
    }
    __device__ static double loglik(double y, double i, const ModelPar& p) { return dpois_log(y, i, p.lgy); }
    // APF look-ahead (this build's choice; the reference defines none for SIR): Poisson at the one-day mean of i
    __device__ static double aux_loglik(double y, double s, double i, const ModelPar& p)
    {
        const double m = i + (p.phi / p.n_total) * s * i - p.sx * i;
        return dpois_log(y, m > 0.0 ? m : 0.0, p.lgy);
    }
};

// Order-preserving map double -> uint64 (for an atomic max over doubles): x < y  <=>  key(x) < key(y); 0 is below every key.
__device__ __forceinline__ unsigned long long f64_key(double x)
{
    const unsigned long long b = (unsigned long long)__double_as_longlong(x);
    return (b >> 63) ? ~b : (b | 0x8000000000000000ull);
}
__device__ __forceinline__ double key_f64(unsigned long long k)
{
    return __longlong_as_double((long long)((k >> 63) ? (k & 0x7fffffffffffffffull) : ~k));
}

// The grid maximum is kept in GM_SLOTS partial maxima on cache lines of their own (block b adds to slot b % GM_SLOTS): 512
// atomics on ONE address serialise at the memory side (~5 ns each: the step kernel ran 2.5 us longer), 32 per line do not.
constexpr int GM_SLOTS = 16, GM_STRIDE = 16;       // slots, and their spacing in uint64 (128 B)

struct NoiseSrc {
    const double* arr;     // parity mode: N draws for this call; nullptr => generator
    PhiloxKey key;
    uint32_t purpose, call;
};

// ---------------------------------------------------------------------------
// k_init: init_fn = rnorm(N, 0, 1)   (R/particle_filter_core.R:76), t = 0 state
// estimate partial sums (:109)
// ---------------------------------------------------------------------------
__device__ __forceinline__ void init_block(double* sh4, const int bidx, double* __restrict__ x, long long N, const NoiseSrc& ns,
                                           double* __restrict__ se_part /* [nblocks][dim] */, int model, const ModelPar& par)
{
    const long long base = (long long)bidx * EB;
    const double invN = 1.0 / (double)N;
    double acc = 0.0, acc1 = 0.0;
#pragma unroll
    for (int r = 0; r < EL / 2; r++) {
        const long long j = base + 2 * (threadIdx.x + NT * r);
        if (j < N) {
            if (model == 2) {   // SIR: every particle starts at (s0, i0)  (stochastic-sir-model.Rmd:286-293)
                x[j] = par.s0; x[N + j] = par.i0; acc += par.s0 * invN; acc1 += par.i0 * invN;
                if (j + 1 < N) { x[j + 1] = par.s0; x[N + j + 1] = par.i0; acc += par.s0 * invN; acc1 += par.i0 * invN; }
            } else {
                double z0, z1;
                if (ns.arr) { z0 = ns.arr[j]; z1 = (j + 1 < N) ? ns.arr[j + 1] : 0.0; }
                else normal_pair(ns.key, ns.purpose, ns.call, 0, (uint32_t)(j >> 1), z0, z1);
                const double x0 = r_rnorm(0.0, 1.0, z0), x1 = r_rnorm(0.0, 1.0, z1);
                x[j] = x0; acc += x0 * invN;
                if (j + 1 < N) { x[j + 1] = x1; acc += x1 * invN; }
            }
        }
    }
    const int dim = (model == 2) ? 2 : 1;
    acc = block_sum(acc, sh4);
    if (dim > 1) acc1 = block_sum(acc1, sh4);
    if (threadIdx.x == 0) { se_part[(long long)bidx * dim] = acc; if (dim > 1) se_part[(long long)bidx * dim + 1] = acc1; }
}

// (boff: first block of this launch in the global block numbering -- 0 unless one filter's particle blocks are sharded)
__global__ __launch_bounds__(NT) void k_init(double* __restrict__ x, long long N, NoiseSrc ns,
                                             double* __restrict__ se_part /* [nblocks][dim] */, int model, ModelPar par, int boff)
{
    __shared__ double sh4[NWV];
    init_block(sh4, (int)blockIdx.x + boff, x, N, ns, se_part, model, par);
}

// ---------------------------------------------------------------------------
// k_step: transition_fn and/or weight_fn for one call, fused with the
// per-block (max, sum exp) partial of the log-sum-exp normalisation.
//   TRANS   : x <- transition(x, z)                     (:127, :159)
//   WEIGHT 1: lw = dnorm(y, x', sy, log)                (:177-182)
//   WEIGHT 2: lw = aux log-lik at the CURRENT particles  (:142-147), no transition
//   SUBAUX  : lw -= aux_lw[ancestor] (already gathered)  (:175)
// ---------------------------------------------------------------------------
constexpr int NTS = 1024;        // threads per workgroup of k_step: one particle PAIR per thread, 16 waves
                                 // (the transcendental chains need the occupancy to hide their latency)

template <int MODEL, bool TRANS, int WEIGHT, bool SUBAUX>
__device__ __forceinline__ void step_block(double* sh /* [2 NTS / 64] */, const int bx, const double* xin, double* xout /* may alias xin */,
                                           double* __restrict__ lw, const double* __restrict__ auxg,
                                           long long N, const ModelPar& par, double y, const NoiseSrc& ns,
                                           double* __restrict__ pm, double* __restrict__ ps, double* __restrict__ pq,
                                           const DevState* __restrict__ st, unsigned long long* __restrict__ gmax)
{
    // (no early return on st->dead here: a dependent read of the run state in front of the particle loads would
    //  cost every launch a memory round trip; propagating a dead run is harmless, its results are never read)
    (void)st;
    const long long j = (long long)bx * EB + 2 * (long long)threadIdx.x;
    double l0 = -INFINITY, l1 = -INFINITY;
    if (j < N) {
        const bool two = (j + 1 < N);
        double x0, x1;
        if (two) { const double2 q = *reinterpret_cast<const double2*>(xin + j); x0 = q.x; x1 = q.y; }
        else { x0 = xin[j]; x1 = 0.0; }
        if (TRANS) {
            double z0, z1;
            if (ns.arr) { z0 = ns.arr[j]; z1 = two ? ns.arr[j + 1] : 0.0; }
            else normal_pair(ns.key, ns.purpose, ns.call, 0, (uint32_t)(j >> 1), z0, z1);
            x0 = Model<MODEL>::transition(x0, z0, par);
            x1 = Model<MODEL>::transition(x1, z1, par);
            if (two) bulk_store16<BSSM_ST_S>(xout + j, x0, x1);
            else xout[j] = x0;
        }
        if (WEIGHT) {
            if (WEIGHT == 2) {
                l0 = r_dnorm_log(y, Model<MODEL>::forecast(x0, par), par.sy, par.log_sy);
                l1 = r_dnorm_log(y, Model<MODEL>::forecast(x1, par), par.sy, par.log_sy);
            } else {
                l0 = r_dnorm_log(y, x0, par.sy, par.log_sy);
                l1 = r_dnorm_log(y, x1, par.sy, par.log_sy);
            }
            if (SUBAUX) { l0 = l0 - auxg[j]; if (two) l1 = l1 - auxg[j + 1]; }
            // (lw == nullptr: the normalising kernel recomputes the log-weights from the particles it reads anyway -- FromLw::xw)
            if (!two) l1 = -INFINITY;
            if (lw) { if (two) { double2 q; q.x = l0; q.y = l1; *reinterpret_cast<double2*>(lw + j) = q; } else lw[j] = l0; }
        }
    }
    if (WEIGHT) {
        const double bm = block_max_n<NTS / 64>(fmax(l0, l1), sh);
        double s = 0.0, q = 0.0;                       // sum exp(l - bm) and sum exp(l - bm)^2 (the latter feeds the ESS)
        if (bm > -INFINITY) {
            if (l0 > -INFINITY) { const double e = exp(l0 - bm); s += e; q += e * e; }
            if (l1 > -INFINITY) { const double e = exp(l1 - bm); s += e; q += e * e; }
        }
        block_sum2_n<NTS / 64>(s, q, sh);
        // max(log_weights) over the whole grid (R/particle_filter_core.R:204) by one atomic per block: the next kernel reads
        // it with one load instead of reducing the B block maxima again in every workgroup
        if (threadIdx.x == 0) { pm[bx] = bm; ps[bx] = s; pq[bx] = q; if (gmax) atomicMax(gmax + (bx % GM_SLOTS) * GM_STRIDE, f64_key(bm)); }
    }
}

template <int MODEL, bool TRANS, int WEIGHT, bool SUBAUX>
__global__ __launch_bounds__(NTS) void k_step(const double* xin, double* xout /* may alias xin */,
                                              double* __restrict__ lw, const double* __restrict__ auxg,
                                              long long N, ModelPar par, double y, NoiseSrc ns,
                                              double* __restrict__ pm, double* __restrict__ ps, double* __restrict__ pq,
                                              const DevState* __restrict__ st, unsigned long long* __restrict__ gmax, int boff)
{
    __shared__ double sh[2 * (NTS / 64)];
    step_block<MODEL, TRANS, WEIGHT, SUBAUX>(sh, (int)blockIdx.x + boff, xin, xout, lw, auxg, N, par, y, ns, pm, ps, pq, st, gmax);
}

// k_lw_partials: the per-block (max, sum exp, sum exp^2) partials of k_step for log-weights that are ALREADY in HBM --
// closure mode: the model's log_likelihood_fn ran on the host (R/particle_filter_core.R:177-183), the normalisation,
// log-likelihood, ESS, resample decision and resampling (:204-224) run here.  Same block shape and reduction order as k_step.
__global__ __launch_bounds__(NTS) void k_lw_partials(const double* __restrict__ lw, long long N, double* __restrict__ pm,
                                                     double* __restrict__ ps, double* __restrict__ pq,
                                                     unsigned long long* __restrict__ gmax)
{
    __shared__ double sh[2 * (NTS / 64)];
    const long long j = (long long)blockIdx.x * EB + 2 * (long long)threadIdx.x;
    double l0 = -INFINITY, l1 = -INFINITY;
    if (j < N) { l0 = lw[j]; if (j + 1 < N) l1 = lw[j + 1]; }
    const double bm = block_max_n<NTS / 64>(fmax(l0, l1), sh);
    double s = 0.0, q = 0.0;
    if (bm > -INFINITY) {
        if (l0 > -INFINITY) { const double e = exp(l0 - bm); s += e; q += e * e; }
        if (l1 > -INFINITY) { const double e = exp(l1 - bm); s += e; q += e * e; }
    }
    block_sum2_n<NTS / 64>(s, q, sh);
    if (threadIdx.x == 0) { pm[blockIdx.x] = bm; ps[blockIdx.x] = s; pq[blockIdx.x] = q; if (gmax) atomicMax(gmax + (blockIdx.x % GM_SLOTS) * GM_STRIDE, f64_key(bm)); }
}

// SIR variant of k_step (state dimension 2, data-dependent Gillespie loop: VALU/divergence-bound, not HBM-bound)
template <bool TRANS, int WEIGHT, bool SUBAUX>
__global__ __launch_bounds__(NTS) void k_step_sir(const double* xin, double* xout /* may alias xin */,
                                                  double* __restrict__ lw, const double* __restrict__ auxg,
                                                  long long N, ModelPar par, double y, NoiseSrc ns,
                                                  double* __restrict__ pm, double* __restrict__ ps, double* __restrict__ pq,
                                                  unsigned long long* __restrict__ gmax)
{
    __shared__ double sh[2 * (NTS / 64)];
    const long long j = (long long)blockIdx.x * EB + 2 * (long long)threadIdx.x;
    double l0 = -INFINITY, l1 = -INFINITY;
    if (j < N) {
        const bool two = (j + 1 < N);
        double s0 = xin[j], i0 = xin[N + j];
        double s1 = two ? xin[j + 1] : 0.0, i1 = two ? xin[N + j + 1] : 0.0;
        if (TRANS) {
            Sir::transition(s0, i0, par, ns.key, ns.call, (uint32_t)j);
            if (two) Sir::transition(s1, i1, par, ns.key, ns.call, (uint32_t)(j + 1));
            xout[j] = s0; xout[N + j] = i0;
            if (two) { xout[j + 1] = s1; xout[N + j + 1] = i1; }
        }
        if (WEIGHT) {
            if (WEIGHT == 2) { l0 = Sir::aux_loglik(y, s0, i0, par); l1 = Sir::aux_loglik(y, s1, i1, par); }
            else { l0 = Sir::loglik(y, i0, par); l1 = Sir::loglik(y, i1, par); }
            if (SUBAUX) { l0 = l0 - auxg[j]; if (two) l1 = l1 - auxg[j + 1]; }
            lw[j] = l0;
            if (two) lw[j + 1] = l1; else l1 = -INFINITY;
        }
    }
    if (WEIGHT) {
        const double bm = block_max_n<NTS / 64>(fmax(l0, l1), sh);
        double s = 0.0, q = 0.0;
        if (bm > -INFINITY) {
            if (l0 > -INFINITY) { const double e = exp(l0 - bm); s += e; q += e * e; }
            if (l1 > -INFINITY) { const double e = exp(l1 - bm); s += e; q += e * e; }
        }
        block_sum2_n<NTS / 64>(s, q, sh);
        if (threadIdx.x == 0) { pm[blockIdx.x] = bm; ps[blockIdx.x] = s; pq[blockIdx.x] = q; atomicMax(gmax + (blockIdx.x % GM_SLOTS) * GM_STRIDE, f64_key(bm)); }
    }
}

// stand-alone resampler front end: validation (src/resampling.cpp:6,18,45) + block sums
__global__ __launch_bounds__(NT) void k_bsum(const double* __restrict__ w, long long nw, double* __restrict__ bsum, DevState* st)
{
    __shared__ double sh4[NWV];
    const long long base = (long long)blockIdx.x * EB;
    double s = 0.0;
    uint32_t f = 0;
#pragma unroll
    for (int r = 0; r < EL; r++) {
        const long long j = base + threadIdx.x + NT * r;
        if (j < nw) {
            const double x = w[j];
            if (x < 0) f |= FLAG_NEGATIVE;
            if (!isfinite(x)) f |= FLAG_NONFINITE;
            s += x;
        }
    }
    s = block_sum(s, sh4);
    if (threadIdx.x == 0) bsum[blockIdx.x] = s;
    if (f) atomicOr(&st->flags, f);
}

// ---------------------------------------------------------------------------
// k_plan (one workgroup, stand-alone resampler only): approximate exclusive block prefixes ain[b] from the
// block sums.  (In the filter these come from the log-sum-exp partials inside k_local<W, from_lw>.)
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(NT) void k_plan(const double* __restrict__ bsum, int B, double* __restrict__ ain, DevState* st)
{
    __shared__ double sh4[NWV];
    if (threadIdx.x == 0) { st->do_resample = 1; st->cur_call = st->res_calls; st->res_calls += 1; }
    double loc[MAXB / NT];
    double tsum = 0.0;
#pragma unroll
    for (int k = 0; k < MAXB / NT; k++) {
        const int i = threadIdx.x * (MAXB / NT) + k;
        loc[k] = (i < B) ? bsum[i] : 0.0;
        tsum += loc[k];
    }
    double pre = block_excl_scan(tsum, sh4);
#pragma unroll
    for (int k = 0; k < MAXB / NT; k++) {
        const int i = threadIdx.x * (MAXB / NT) + k;
        if (i < B) ain[i] = pre;
        pre += loc[k];
    }
}

// ---------------------------------------------------------------------------
// exact sequential scan (seqsum.h) -- device orchestration
//
// Almost every thread-chunk (8 terms) and every block of a real weight vector
// is PURE: the running sum stays inside one binade, so its action on the state
// is "add d0 ulps if the incoming low bit is 0, d1 if it is 1".  PURE maps
// compose with two integer adds, so they are scanned with wave shuffles.  The
// few lanes that are not PURE (a binade crossing, the exact-zero head, the
// neighbourhood of cum == 1.0) are BOUNDARIES: they cut the scan into segments
// and are stepped through one after the other, by one lane, with the exact
// state.  Afterwards every PURE lane checks that its own exact incoming state
// lies inside the window its record was built for (|in - hypothesis| <= lim);
// one failed check anywhere discards the result and the block is re-run
// literally, in order -- correctness never depends on the approximation.
// ---------------------------------------------------------------------------
struct Pure { int64_t d0, d1; };

// (bit-mask selects, not ?: on struct members: the compiler turns the latter into an indexed load and
//  keeps the whole record in scratch memory)
__device__ __forceinline__ int64_t sel_i64(int64_t a, int64_t b, uint64_t odd)
{
    const uint64_t m = 0ull - (odd & 1ull);
    return (int64_t)(((uint64_t)a & ~m) | ((uint64_t)b & m));
}
__device__ __forceinline__ Pure pure_compose(const Pure& f, const Pure& g)
{
    Pure c;
    c.d0 = f.d0 + sel_i64(g.d0, g.d1, (uint64_t)f.d0);
    c.d1 = f.d1 + sel_i64(g.d0, g.d1, (uint64_t)f.d1 + 1ull);
    return c;
}
__device__ __forceinline__ uint64_t pure_step(const Pure& f, uint64_t in) { return in + (uint64_t)sel_i64(f.d0, f.d1, in); }

// A boundary step "PURE run, then one record" folded into constants, so that the serial walk over the boundaries
// has a short dependent chain per step:   out = ((s - c) >> sh) + K[c],  c = s & 3.
// Derivation: e = s + ex.d[c&1];  diff = e - base = (s - c) + R_c with R_c = c + ex.d[c&1] - base and (s - c) a
// multiple of 4, so the record's hypothesis index is sidx = R_c & 3 and
//   PURE: out = o[sidx] + (diff - sidx)       = (s - c)      + (o[sidx] + R_c - sidx)
//   X1  : out = o[sidx] + (diff - sidx) / 2   = (s - c) / 2  + (o[sidx] + (R_c - sidx) / 2).
// Valid while diff stays in the record's window, i.e. (s - sref) in [wlo, whi].
struct StepFn {
    uint64_t K0, K1, K2, K3;
    uint64_t sref; int64_t wlo, whi;
    int sh;          // 0 PURE, 1 X1
    int mode;        // 0 = formula, 1 = constant (ABS), 2 = no fast form (HARD): caller re-runs the terms
};

__device__ __forceinline__ StepFn stepfn_build(const Pure& ex, const Rec& r)
{
    StepFn f; f.K0 = f.K1 = f.K2 = f.K3 = 0; f.sref = 0; f.wlo = 0; f.whi = -1; f.sh = 0; f.mode = 2;
    if (r.kind == REC_ABS) { f.mode = 1; f.K0 = f.K1 = f.K2 = f.K3 = r.o[0]; return f; }
    if (r.kind == REC_HARD) return f;
    f.mode = 0; f.sh = (r.kind == REC_X1) ? 1 : 0;
    uint64_t K[4];
    int64_t lo = -(1ll << 62), hi = (1ll << 62);
    f.sref = r.base - (uint64_t)ex.d0;                       // the incoming state the hypotheses were built around
#pragma unroll
    for (int c = 0; c < 4; c++) {
        const uint64_t eoff = (uint64_t)((c & 1) ? ex.d1 : ex.d0);
        const int64_t R = (int64_t)((uint64_t)c + eoff - r.base);
        const int64_t sidx = R & 3;
        const uint64_t os = (sidx == 0) ? r.o[0] : (sidx == 1) ? r.o[1] : (sidx == 2) ? r.o[2] : r.o[3];
        const int64_t rem = R - sidx;                        // multiple of 4
        K[c] = os + (uint64_t)(f.sh ? (rem >> 1) : rem);
        // window: r.lo <= (s - c) + R <= r.hi   <=>   s - sref in [r.lo - R + c - (sref offset) ...]
        const int64_t shift = (int64_t)((uint64_t)c - (uint64_t)R - f.sref);     // s >= r.lo + shift + sref ...
        const int64_t l = (int64_t)r.lo + shift, h = (int64_t)r.hi + shift;
        lo = l > lo ? l : lo; hi = h < hi ? h : hi;
    }
    f.K0 = K[0]; f.K1 = K[1]; f.K2 = K[2]; f.K3 = K[3];
    f.wlo = lo + 4; f.whi = hi - 4;
    return f;
}

__device__ __forceinline__ uint64_t stepfn_apply(const StepFn& f, uint64_t s, bool& ok)
{
    const uint64_t c = s & 3ull;
    const uint64_t Kc = (c == 0) ? f.K0 : (c == 1) ? f.K1 : (c == 2) ? f.K2 : f.K3;
    const uint64_t out = ((s - c) >> f.sh) + Kc;
    const int64_t off = (int64_t)(s - f.sref);
    ok = (f.mode == 1) || (f.mode == 0 && off >= f.wlo && off <= f.whi);
    return (f.mode == 1) ? f.K0 : out;
}

// A boundary step as plain data.  The serial part of a scan is a chain of these, evaluated branch-free (walk_chain_regs):
// ~10 dependent VALU ops a link.  Links that are not "simple" (HARD, side entries, window misses) stop the chain; the caller
// continues from there with the general loop.
struct alignas(16) WalkFn { uint64_t K0, K1, K2, K3; uint64_t nref, range; int sh; int pad0, pad1, pad2; };   // 64 B
// a link accepts state s  <=>  (s + nref) <u range      (nref = -(sref + wlo), range = whi - wlo + 1; range 0 = never)

__device__ __forceinline__ WalkFn walkfn_from(const StepFn& f, bool simple)
{
    WalkFn w; w.K0 = f.K0; w.K1 = f.K1; w.K2 = f.K2; w.K3 = f.K3; w.sh = f.sh; w.pad0 = w.pad1 = w.pad2 = 0;
    const bool okwin = simple && f.mode == 0 && f.whi >= f.wlo;
    w.nref = okwin ? (0ull - (f.sref + (uint64_t)f.wlo)) : 0ull;
    w.range = okwin ? (uint64_t)(f.whi - f.wlo) + 1ull : 0ull;
    if (simple && f.mode == 1) {   // ABS: out = K whatever comes in.  States are bit patterns of non-negative doubles (< 2^63), so >> 63 clears them.
        w.sh = 63; w.nref = 0; w.range = ~0ull;
    }
    return w;
}

__device__ __forceinline__ uint64_t mk64(uint32_t lo, uint32_t hi) { return ((uint64_t)hi << 32) | lo; }

__device__ __forceinline__ uint64_t readlane_u64(uint64_t v, int j)
{
    const int lo = __builtin_amdgcn_readlane((int)(uint32_t)v, j), hi = __builtin_amdgcn_readlane((int)(uint32_t)(v >> 32), j);
    return ((uint64_t)(uint32_t)hi << 32) | (uint32_t)lo;
}

// The same walk with the chain in REGISTERS: lane j of the wave holds link j.  The state travels in scalar registers
// (wave-uniform); per link every lane evaluates ITS link on the current state and lane j's result is read back with
// v_readlane (j is the loop counter, a scalar).  No LDS round trip per link: the LDS form above waited ~500 cycles a link for
// its four 16-byte reads (profiles/r02_a_stage_stamps_typical_block.txt: 10 links in 5000 cycles).
// Returns how many leading links accepted their state (jd); lane j < jd gets (my_s, my_out) = the state before / after its
// link; s = the state before link jd.
// (Round 3 tried the opposite split: lane j's constants read into scalar registers one link ahead (13 v_readlane per link) and the link
//  evaluated on the scalar unit.  The compiler does emit s_cselect / s_lshr_b64 / s_add for it, but the chain of 10 links took 3.9k cycles
//  against 2.6k here: the readlanes cost more than the VALU -> SGPR round trip they remove.  profiles/r03_c_rejected_variants.txt)
// (Round 3, third attempt at the ~260 cycles a link: the state kept in vector registers and handed from link to link by DPP row_newbcast:J
//  -- 16 unrolled steps per row, no v_readlane, no SGPR, no data-dependent branch; accept bits broadcast the same way.  Bit-identical
//  (225 GPU tests), but SLOWER: the fused launch 48.2 -> 48.8 us, k_local<P> 11.17 -> 11.45: ~30 vector instructions a link with the
//  DPP wait states and three unrolled copies of 16 steps in the instruction stream cost more than three readlanes.  Removed.)
__device__ __forceinline__ int walk_chain_regs(const WalkFn& w, int nent, uint64_t& s, uint64_t& my_s, uint64_t& my_out)
{
    const int lane = threadIdx.x & 63;
    uint64_t sv = readlane_u64(s, 0);
    int j = 0;
#pragma unroll 1
    for (; j < nent; j++) {
        const bool bad = (sv + w.nref) >= w.range;
        const uint32_t lo = (uint32_t)sv;
        const uint64_t k01 = (lo & 1u) ? w.K1 : w.K0, k23 = (lo & 1u) ? w.K3 : w.K2;
        const uint64_t out = ((sv & ~3ull) >> (w.sh & 63)) + ((lo & 2u) ? k23 : k01);
        if (__builtin_amdgcn_readlane((int)bad, j)) break;
        if (lane == j) { my_s = sv; my_out = out; }
        sv = readlane_u64(out, j);
    }
    s = sv;
    return j;
}

// (Round 3 also tried the links' constants in LDS, read at wave-uniform addresses so that every lane evaluates every link and the state
//  never leaves the vector registers -- no v_readlane, one uniform branch per four links.  With a provably uniform address the compiler
//  moved the chain to the scalar unit (16 v_readfirstlane a link: +3.5 us per resolve); kept on the vector unit by an opaque zero in the
//  index it ran 0.3 us per resolve SLOWER than walk_chain_regs (N = 2^20: k_local<P> 11.35 -> 11.8 us, k_apply 15.2 -> 15.45, the fused
//  launch 47.8 -> 48.3): the LDS write -> read turn-around and 16 ds_read per four links cost what the readlanes cost.  Removed.)
__device__ __forceinline__ int64_t shfl_up_i64(int64_t v, int off)
{
    int lo = (int)(uint32_t)(uint64_t)v, hi = (int)(uint32_t)((uint64_t)v >> 32);
    lo = __shfl_up(lo, off, 64); hi = __shfl_up(hi, off, 64);
    return (int64_t)(((uint64_t)(uint32_t)hi << 32) | (uint32_t)lo);
}

struct SegSmem {
    Pure wagg[16]; int wflag[16]; int wnb[16]; Pure wcarry[17]; int wsegbase[17];
    Pure bnd_excl[MAXBND]; Rec bnd_rec[MAXBND]; int bnd_lane[MAXBND]; uint64_t bnd_ent[MAXBND];
    double bnd_terms[MAXBND][EL];    // the terms of every boundary lane (so a literal re-run never chases global memory)
    uint64_t seg_start[MAXBND + 1];
    int red_min[NWV][2], red_max[NWV][2];
    uint64_t segbase[MAXBND + 1];    // hypothesis base of every segment = that of its first lane (published by block_scan: no barrier round later)
    int wlastb[16];                  // per wave: its last lane is a boundary
    int smin[MAXBND + 1]; int smax[MAXBND + 1];
    int nb; int fail; int big;
    double sh4[16];
    double bcast;
};

// Segmented exclusive scan over the NT lanes of a block.  `isb`: this lane is a
// boundary (contributes the identity, ends the segment).  Returns the
// composite of the PURE lanes between the previous boundary and this lane,
// the lane's segment index (= number of boundaries before it), and leaves the
// boundary count in sm.nb.
// Inside the scan a PURE map travels as (d0, dd = d1 - d0): dd is tiny (it only moves at rounding ties) and is zero for
// every lane of almost every wave, in which case the segmented scan degenerates to a segmented int64 add.
//   compose(f, g):  a = odd(f.d0) ? g.dd : 0;  b = odd(f.d0 + f.dd + 1) ? g.dd : 0;
//                   d0 = f.d0 + g.d0 + a;      dd = f.dd + b - a.
struct PureC { int64_t d0; int32_t dd; };
__device__ __forceinline__ PureC purec_compose(const PureC& f, const PureC& g)
{
    const int32_t a = (f.d0 & 1) ? g.dd : 0;
    const int32_t b = ((f.d0 + f.dd + 1) & 1) ? g.dd : 0;
    PureC c; c.d0 = f.d0 + g.d0 + a; c.dd = f.dd + b - a;
    return c;
}

template <int NW>
__device__ __forceinline__ Pure seg_excl_scan(SegSmem& sm, const Pure& leaf, bool isb, int& seg, int& nb_total, Pure& block_incl_last_seg,
                                              bool* prev_is_boundary = nullptr /* out: the lane in front of this one is a boundary lane */)
{
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const unsigned long long bal = __ballot(isb);
    const unsigned long long below = bal & ((1ull << lane) - 1ull);
    // head flag: the previous lane is a boundary (lane 0 of a wave: decided across waves)
    PureC v; v.d0 = isb ? 0 : leaf.d0; v.dd = isb ? 0 : (int32_t)(leaf.d1 - leaf.d0);
    int fi = ((lane > 0) && ((bal >> (lane - 1)) & 1ull)) ? 1 : 0;
    // segmented inclusive scan: combine(prev, cur) = cur.f ? cur : (prev.v . cur.v, prev.f)
    if (__all(v.dd == 0)) {
#define STEP(C, R) { const int64_t p0 = dpp_i64<C, R>(0, v.d0); const int pf = dpp_i32<C, R>(0, fi); if (!fi) { v.d0 += p0; fi = pf; } }
        BSSM_WAVE_SCAN_STEPS(STEP)
#undef STEP
    } else {
#define STEP(C, R) { PureC pv; pv.d0 = dpp_i64<C, R>(0, v.d0); pv.dd = dpp_i32<C, R>(0, v.dd); const int pf = dpp_i32<C, R>(0, fi); \
                     if (!fi) { v = purec_compose(pv, v); fi = pf; } }
        BSSM_WAVE_SCAN_STEPS(STEP)
#undef STEP
    }
    // v = composite from the segment head (or the wave start) up to and including this lane
    const bool last_is_b = (bal >> 63) & 1ull;
    if (lane == 63) {
        Pure a; a.d0 = v.d0; a.d1 = v.d0 + v.dd;
        if (last_is_b) { a.d0 = 0; a.d1 = 0; }                        // a boundary in the last lane: nothing is carried
        sm.wagg[wave] = a; sm.wflag[wave] = (bal != 0ull) ? 1 : 0;    // a boundary anywhere in the wave cuts the carry
        sm.wnb[wave] = __popcll(bal);
        sm.wlastb[wave] = last_is_b ? 1 : 0;
    }
    __syncthreads();
    if (prev_is_boundary) *prev_is_boundary = (lane > 0) ? (((bal >> (lane - 1)) & 1ull) != 0) : (wave > 0 && sm.wlastb[wave - 1] != 0);
    Pure carry; carry.d0 = 0; carry.d1 = 0;
    int segbase = 0;
    if (NW <= 4) {
        // few waves: every lane folds the aggregates of the waves before its own
        Pure all; all.d0 = 0; all.d1 = 0;
        int nball = 0;
#pragma unroll
        for (int w = 0; w < NW; w++) {
            if (w == wave) { carry = all; segbase = nball; }
            if (sm.wflag[w]) all = sm.wagg[w]; else all = pure_compose(all, sm.wagg[w]);
            nball += sm.wnb[w];
        }
        nb_total = nball;
        block_incl_last_seg = all;
    } else {
        // many waves (the resolve workgroup): wave 0 scans the NW aggregates and publishes exclusive carries
        if (wave == 0) {
            Pure a; a.d0 = 0; a.d1 = 0; int af = 0, an = 0;
            if (lane < NW) { a = sm.wagg[lane]; af = sm.wflag[lane]; an = sm.wnb[lane]; }
            int cn = an;
#define STEP(C, R) { Pure pv; pv.d0 = dpp_i64<C, R>(0, a.d0); pv.d1 = dpp_i64<C, R>(0, a.d1); const int pf = dpp_i32<C, R>(0, af); \
                     cn += dpp_i32<C, R>(0, cn); if (!af) { a = pure_compose(pv, a); af = pf; } }
            STEP(DPP_SHR1, 0xf) STEP(DPP_SHR2, 0xf) STEP(DPP_SHR4, 0xf) STEP(DPP_SHR8, 0xf)
#undef STEP
            Pure e; e.d0 = dpp_i64<DPP_WAVE_SHR1, 0xf>(0, a.d0); e.d1 = dpp_i64<DPP_WAVE_SHR1, 0xf>(0, a.d1);
            const int en = dpp_i32<DPP_WAVE_SHR1, 0xf>(0, cn);
            if (lane < NW) { sm.wcarry[lane] = e; sm.wsegbase[lane] = en; }
            if (lane == NW - 1) { sm.wcarry[NW] = a; sm.wsegbase[NW] = cn; }
        }
        __syncthreads();
        carry = sm.wcarry[wave];
        segbase = sm.wsegbase[wave];
        nb_total = sm.wsegbase[NW];
        block_incl_last_seg = sm.wcarry[NW];
    }
    // lanes at or before the first boundary of the wave (inclusive of that boundary lane) still belong to the carried segment
    const bool carried = (below == 0ull);
    Pure vi; vi.d0 = v.d0; vi.d1 = v.d0 + v.dd;
    const Pure inc = carried ? pure_compose(carry, vi) : vi;
    // exclusive = inclusive of the previous lane, unless the previous lane is a boundary
    Pure exc; exc.d0 = dpp_i64<DPP_WAVE_SHR1, 0xf>(carry.d0, inc.d0); exc.d1 = dpp_i64<DPP_WAVE_SHR1, 0xf>(carry.d1, inc.d1);
    if (lane > 0 && ((bal >> (lane - 1)) & 1ull)) { exc.d0 = 0; exc.d1 = 0; }
    seg = segbase + __popcll(below);
    return exc;
}

// The same scan for the grid-level resolve, where a lane may hold non-PURE blocks WITH a PURE run behind them: a boundary
// lane contributes `val` = the run behind its last non-PURE block (its "tail") and restarts the segment at itself; `nent` is
// the number of its non-PURE blocks.  Returns the PURE composite between the previous boundary lane (tail included) and this
// lane, the number of non-PURE blocks in front of the lane (ent_before) and in the whole workgroup (ent_total).
template <int NW>
__device__ __forceinline__ Pure seg_excl_scan_tail(SegSmem& sm, const Pure& val, bool isb, int nent, int& ent_before, int& ent_total)
{
    static_assert(NW <= 16, "workgroups of up to sixteen waves");
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const unsigned long long bal = __ballot(isb);
    Pure v = val;
    int fi = isb ? 1 : 0, cn = nent;
#define STEP(C, R) { Pure pv; pv.d0 = dpp_i64<C, R>(0, v.d0); pv.d1 = dpp_i64<C, R>(0, v.d1); const int pf = dpp_i32<C, R>(0, fi); \
                     cn += dpp_i32<C, R>(0, cn); if (!fi) { v = pure_compose(pv, v); fi = pf; } }
    BSSM_WAVE_SCAN_STEPS(STEP)
#undef STEP
    // v = composite from the last boundary lane at or below this lane (its tail included), or from the wave start
    if (lane == 63) { sm.wagg[wave] = v; sm.wflag[wave] = (bal != 0ull) ? 1 : 0; sm.wnb[wave] = cn; }
    __syncthreads();
    Pure carry; carry.d0 = 0; carry.d1 = 0;
    int before = 0, all_n = 0;
    {
        Pure all; all.d0 = 0; all.d1 = 0;
#pragma unroll
        for (int wv = 0; wv < NW; wv++) {
            if (wv == wave) { carry = all; before = all_n; }
            if (sm.wflag[wv]) all = sm.wagg[wv]; else all = pure_compose(all, sm.wagg[wv]);
            all_n += sm.wnb[wv];
        }
    }
    ent_total = all_n;
    // lanes with no boundary at or below them still belong to the segment carried in from the waves before
    const bool carried = (bal & ((lane == 63) ? ~0ull : ((2ull << lane) - 1ull))) == 0ull;
    const Pure inc = carried ? pure_compose(carry, v) : v;
    Pure exc; exc.d0 = dpp_i64<DPP_WAVE_SHR1, 0xf>(carry.d0, inc.d0); exc.d1 = dpp_i64<DPP_WAVE_SHR1, 0xf>(carry.d1, inc.d1);
    ent_before = before + (cn - nent);
    return exc;
}

template <int MODE>
__device__ __forceinline__ void load_terms(const double* __restrict__ w, long long nw, double total, long long j0, double (&v)[EL])
{
    if (j0 + EL <= nw) {
        const double2* p = reinterpret_cast<const double2*>(w + j0);
#pragma unroll
        for (int k = 0; k < EL / 2; k++) { const double2 q = p[k]; v[2 * k] = q.x; v[2 * k + 1] = q.y; }
    } else {
#pragma unroll
        for (int k = 0; k < EL; k++) v[k] = (j0 + k < nw) ? w[j0 + k] : 0.0;
    }
    if (MODE == MODE_P) {
#pragma unroll
        for (int k = 0; k < EL; k++) v[k] = v[k] / total;          // prob = weights / total (src/resampling.cpp:24,51)
    }
}

template <int MODE>
__device__ __forceinline__ double term_at(const double* __restrict__ w, long long nw, double total, long long j)
{
    double x = (j < nw) ? w[j] : 0.0;
    if (MODE == MODE_P) x = x / total;
    return x;
}

// literal left-to-right run over terms [j0, j1) from exact state `in`
template <int MODE>
__device__ __attribute__((noinline)) uint64_t literal_run(const double* __restrict__ w, long long nw, double total, long long j0, long long j1, uint64_t in)
{
    double c = b2d(in);
    for (long long j = j0; j < j1; j++) c = c + term_at<MODE>(w, nw, total, j);
    return d2b(c);
}

// Everything a block knows before the exact incoming state is available.
struct BlockScan {
    Rec leaf;          // this lane's record (general form)
    Pure pleaf;        // PURE form (valid when !isb)
    Pure exc;          // composite of the PURE lanes between the previous boundary and this lane
    Pure last_seg;     // composite of the block's last segment through the last lane
    bool isb;
    int seg, nb;
};

template <int MODE>
__device__ __forceinline__ void block_scan(SegSmem& sm, const double (&v)[EL], double a_in, int lim, BlockScan& bs)
{
    double ts = 0.0;
#pragma unroll
    for (int k = 0; k < EL; k++) ts += v[k];
    const double h = a_in + block_excl_scan(ts, sm.sh4);
    bs.leaf = chunk_record_fixed<EL>(v, h, lim);
    bs.isb = bs.leaf.kind != REC_PURE;
    bs.pleaf.d0 = (int64_t)(bs.leaf.o[0] - bs.leaf.base);
    bs.pleaf.d1 = (int64_t)(bs.leaf.o[1] - (bs.leaf.base + 1));
    bool prev_b = false;
    bs.exc = seg_excl_scan<NT / 64>(sm, bs.pleaf, bs.isb, bs.seg, bs.nb, bs.last_seg, &prev_b);
    if (!bs.isb && prev_b && bs.seg <= MAXBND) sm.segbase[bs.seg] = bs.leaf.base;      // the first lane of segment bs.seg (>= 1)
    // publish the boundaries in lane order
    if (bs.isb && bs.seg < MAXBND) {
        Rec& d = sm.bnd_rec[bs.seg];
        d.base = bs.leaf.base; d.o[0] = bs.leaf.o[0]; d.o[1] = bs.leaf.o[1]; d.o[2] = bs.leaf.o[2]; d.o[3] = bs.leaf.o[3];
        d.kind = bs.leaf.kind; d.lo = bs.leaf.lo; d.hi = bs.leaf.hi; d.pad = 0;
        sm.bnd_excl[bs.seg].d0 = bs.exc.d0; sm.bnd_excl[bs.seg].d1 = bs.exc.d1; sm.bnd_lane[bs.seg] = threadIdx.x;
#pragma unroll
        for (int q = 0; q < EL; q++) sm.bnd_terms[bs.seg][q] = v[q];
    }
}

// With the exact incoming state of the block: exact incoming state of every lane.
// Returns false (block-uniform) if the records did not cover the exact states;
// the caller then falls back to block_literal().
template <int MODE>
__device__ __forceinline__ bool block_resolve(SegSmem& sm, const BlockScan& bs, uint64_t cin, int lim,
                                              const double* __restrict__ w, long long nw, double total, long long b0,
                                              uint64_t& ent)
{
    const int t = threadIdx.x;
    if (t == 0) sm.fail = (bs.nb > MAXBND) ? 1 : 0;
    __syncthreads();
    // Boundary walk by wave 0: lane k holds boundary k's record in registers; the exact state is handed from lane
    // to lane with v_readlane, so a step costs one record evaluation and no memory round trip.
    if (t < 64 && bs.nb == 0) {
        // no boundary in this block (nearly every block): nothing to walk -- and no step functions to build for an empty chain (~100
        // 64-bit operations that every workgroup used to spend here, between the arrival of its exact state and its expansion)
        if (t == 0) sm.seg_start[0] = cin;
    } else if (t < 64 && bs.nb <= MAXBND) {
        const int lane = t, nb = bs.nb;
        Rec rec = rec_identity(0);
        Pure ex; ex.d0 = 0; ex.d1 = 0;
        if (lane < nb) { rec = sm.bnd_rec[lane]; ex = sm.bnd_excl[lane]; }
        const StepFn fn = stepfn_build(ex, rec);             // "PURE run + this boundary" folded into constants
        uint64_t s = cin, my_s = 0, my_out = 0;
        const WalkFn wfn = walkfn_from(fn, lane < nb);
        const int j0 = walk_chain_regs(wfn, nb, s, my_s, my_out);
        for (int j = j0; j < nb; j++) {                      // what the chain could not take (HARD leaves, window misses)
            bool ok;
            uint64_t o = stepfn_apply(fn, s, ok);
            if (!__builtin_amdgcn_readlane((int)ok, j)) {    // boundary j's 8 terms literally, from LDS
                double c = b2d(pure_step(ex, s));
#pragma unroll
                for (int q = 0; q < EL; q++) c = c + sm.bnd_terms[j][q];
                o = d2b(c);                                  // (every lane computes from the shared terms; only lane j's ex is right)
            }
            if (lane == j) { my_s = s; my_out = o; }
            const int olo = __builtin_amdgcn_readlane((int)(uint32_t)o, j), ohi = __builtin_amdgcn_readlane((int)(uint32_t)(o >> 32), j);
            s = ((uint64_t)(uint32_t)ohi << 32) | (uint32_t)olo;
        }
        const uint64_t my_ent = pure_step(ex, my_s);
        if (lane == 0) sm.seg_start[0] = cin;
        if (lane < nb) { sm.bnd_ent[lane] = my_ent; sm.seg_start[lane + 1] = my_out; }
    }
    __syncthreads();
    // (not a re-read of sm.fail: a wave that is late here could see the flag another wave sets below for a window miss, leave
    //  early and miss the barrier behind the checks.  The only setter in front of this point is the block-uniform test.)
    if (bs.nb > MAXBND) return false;
    bool ok = true;
    if (bs.isb) ent = sm.bnd_ent[bs.seg];
    else {
        ent = pure_step(bs.exc, sm.seg_start[bs.seg]);
        const int64_t diff = (int64_t)(ent - bs.leaf.base);
        ok = (diff >= -(int64_t)lim) && (diff <= (int64_t)lim);
    }
    if (!ok) sm.fail = 1;
    __syncthreads();
    return sm.fail == 0;
}

// Literal fallback: thread 0 runs the whole block in order and records each lane's incoming state.
template <int MODE>
__device__ __attribute__((noinline)) void block_literal(uint64_t* tin /* LDS [NT] */, uint64_t cin, const double* __restrict__ w,
                                              long long nw, double total, long long b0, DevState* st)
{
    if (threadIdx.x == 0) {
        double c = b2d(cin);
        for (int tt = 0; tt < NT; tt++) {
            tin[tt] = d2b(c);
            for (int k = 0; k < EL; k++) c = c + term_at<MODE>(w, nw, total, b0 + (long long)tt * EL + k);
        }
        atomicAdd((unsigned long long*)&st->stat_serial_walks, 1ull);
        atomicAdd((unsigned long long*)&st->stat_literal_terms, (unsigned long long)EB);
    }
    __syncthreads();
}

// One block, few terms (the batched small filters): thread 0 adds the terms in order from exact +0 -- the reference's own
// loop (src/resampling.cpp:20,25) -- and records every lane's incoming state.  For a few hundred terms this is faster
// than the parallel record machinery, whose fixed latency (scan + boundary chain) is ~13k cycles per pass.
// `terms`: the n terms in order (LDS), already published and synchronised by the caller.
__device__ __forceinline__ void block_literal_terms(uint64_t* tin /* LDS [NT] */, const double* terms, int n)
{
    const int t = threadIdx.x;
    const int nl = (n + EL - 1) / EL;                      // lanes that hold terms (terms[] is zero-padded to a multiple of EL by the caller)
    if (t == 0) {
        double c = 0.0;
        // 16 terms (LPT lanes) per trip, the next trip's terms loaded a whole trip ahead: the LDS latency (~130 cycles)
        // hides under 16 dependent adds.  Reading up to a trip past the last lane is harmless (inside the EB-sized buffer;
        // a lane's values are added only when the lane holds terms).
        constexpr int LPT = 16 / EL;
        const double2* p = reinterpret_cast<const double2*>(terms);
        double2 a[8], b[8];
#pragma unroll
        for (int k = 0; k < 8; k++) a[k] = p[k];
        for (int tt = 0; tt < nl; tt += LPT) {
            const int nx = (tt + LPT < nl) ? tt + LPT : tt;
            const double2* q = p + (EL / 2) * nx;
#pragma unroll
            for (int k = 0; k < 8; k++) b[k] = q[k];
#pragma unroll
            for (int l = 0; l < LPT; l++) {
                if (l == 0 || tt + l < nl) {
                    tin[tt + l] = d2b(c);
#pragma unroll
                    for (int k = 0; k < EL / 2; k++) { c = c + a[l * (EL / 2) + k].x; c = c + a[l * (EL / 2) + k].y; }
                }
            }
#pragma unroll
            for (int k = 0; k < 8; k++) a[k] = b[k];
        }
        tin[NT] = d2b(c);                                   // the block's outgoing state
    }
    __syncthreads();
    if (t >= nl) tin[t] = tin[NT];                          // lanes past the end: nothing left to add
    __syncthreads();
}

// block-wide min and max of two pairs (segments 0 and 1), lanes opt in per segment.  The values are offsets in ulps; every use clamps what
// it derives from them to +-2^30 (pure_to_rec, the side entries), so the reductions run on values saturated to 32 bits (half the DPP
// moves of the 64-bit scans; results identical after the clamp).  The caller has passed a barrier since the last read of sm.red_*.
__device__ __forceinline__ int wave_min_i32(int x)
{
#define STEP(C, R) { const int y = dpp_i32<C, R>(x, x); x = y < x ? y : x; }
    BSSM_WAVE_SCAN_STEPS(STEP)
#undef STEP
    return __builtin_amdgcn_readlane(x, 63);
}
__device__ __forceinline__ int wave_max_i32(int x)
{
#define STEP(C, R) { const int y = dpp_i32<C, R>(x, x); x = y > x ? y : x; }
    BSSM_WAVE_SCAN_STEPS(STEP)
#undef STEP
    return __builtin_amdgcn_readlane(x, 63);
}
__device__ __forceinline__ int sat_i32(long long x) { return x > 0x7fffffffll ? 0x7fffffff : (x < -0x7fffffffll ? -0x7fffffff : (int)x); }
// Window values travel as 32-bit integers saturated at +-WSAT = +-(2^30 + 2^24): every consumer clamps to +-2^30 in the end, and the margin
// keeps  -lim - x + 4  and friends (lim < 2^23) inside int32 AND on the same side of +-2^30 as the exact 64-bit value, so the clamped
// result is the 64-bit code's result for every input (round 3: the 64-bit form of this arithmetic was ~400 dependent instructions on the
// four lanes that compose a crossing block's record, and every grid-level resolve waits for the last record).
constexpr int WSAT = (1 << 30) + (1 << 24);
__device__ __forceinline__ int sat_w(long long x) { return x > (long long)WSAT ? WSAT : (x < -(long long)WSAT ? -WSAT : (int)x); }
__device__ __forceinline__ int clamp30i(int x) { return x < -(1 << 30) ? -(1 << 30) : (x > (1 << 30) ? (1 << 30) : x); }
__device__ __forceinline__ void block_minmax2(SegSmem& sm, int mn[2], int mx[2])
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int a0 = wave_min_i32(mn[0]), a1 = wave_min_i32(mn[1]);
    const int b0 = wave_max_i32(mx[0]), b1 = wave_max_i32(mx[1]);
    if (lane == 0) { sm.red_min[wave][0] = a0; sm.red_min[wave][1] = a1; sm.red_max[wave][0] = b0; sm.red_max[wave][1] = b1; }
    __syncthreads();
#pragma unroll
    for (int s = 0; s < 2; s++) {
        int a = sm.red_min[0][s], b = sm.red_max[0][s];
#pragma unroll
        for (int wv = 1; wv < NWV; wv++) { a = min(a, sm.red_min[wv][s]); b = max(b, sm.red_max[wv][s]); }
        mn[s] = a; mx[s] = b;
    }
}

__device__ __forceinline__ Rec pure_to_rec(const Pure& p, uint64_t base, long long lo, long long hi)
{
    Rec r; r.pad = 0; r.base = base & ~3ull; r.kind = REC_PURE;
    r.o[0] = r.base + (uint64_t)p.d0; r.o[1] = r.base + 1 + (uint64_t)p.d1;
    r.o[2] = r.base + 2 + (uint64_t)p.d0; r.o[3] = r.base + 3 + (uint64_t)p.d1;
    const long long L = -(1ll << 30), H = (1ll << 30);
    r.lo = (int32_t)(lo < L ? L : (lo > H ? H : lo));
    r.hi = (int32_t)(hi < L ? L : (hi > H ? H : hi));
    return r;
}

// Where a block's record of a pass goes: the planes in HBM that the next kernel's workgroups read (GlobalRecSink), or
// wherever a fused kernel publishes it.  `literal`: the in-order fallback of a block that starts from exact +0.
template <int MODE>
struct GlobalRecSink {
    BlockRec* brec; SideList* sides; int bidx, bstride;
    const double* w; long long nw; double total; long long b0; DevState* st;
    __device__ __forceinline__ void rec(const BlockRec& br) const { store_brec(brec, bidx, br, bstride); }
    __device__ __forceinline__ void side(int k, const SideEntry& e) const { this->side_list()[k] = e; }
    __device__ __forceinline__ SideEntry* side_list() const { return sides[bidx].e; }
    __device__ __forceinline__ void literal(uint64_t* tin, uint64_t cin, const double (&)[EL]) const { block_literal<MODE>(tin, cin, w, nw, total, b0, st); }
};

// The block's record for the grid-level resolve, from its scan (see k_local below for the four cases).
template <int MODE, int STAMPCOL, class Sink>
__device__ __forceinline__ void block_record_tail(SegSmem& sm, uint64_t* tin, const BlockScan& bs, const double (&v)[EL], const double a_in,
                                                  const int lim, DevState* st, const int dbg, const bool stamper, const Sink& sink)
{
    const int t = threadIdx.x;
    (void)dbg; (void)stamper;
    BSSM_STAMP(st, dbg, 2, 12, stamper);
    const uint64_t hb0 = d2b(a_in) & ~3ull;
    if (a_in == 0.0) {
        // exact incoming state (+0): resolve now and publish the exact outgoing state
        uint64_t ent;
        bool good = block_resolve<MODE>(sm, bs, 0ull, lim, nullptr, 0, 1.0, 0, ent);
        if (!good) { sink.literal(tin, 0ull, v); ent = tin[t]; }
        if (t == NT - 1) {
            double c = b2d(ent);
#pragma unroll
            for (int k = 0; k < EL; k++) c = c + v[k];
            BlockRec br; br.prefix = rec_abs(d2b(c)); br.tail_from = NT; br.nside = 0;
            sink.rec(br);
        }
        return;
    }
    // validity windows of segment 0 (lanes before the first boundary) and of the last segment when nb == 1
    // (round 3: the first boundary lane and the base of the segment behind it are read from what block_scan published -- the three
    //  barrier rounds that used to find them made the blocks with a binade crossing publish ~1 us after the others, and a grid-level
    //  resolve waits for the last record)
    int fb = NT;
    uint64_t seg1_base = 0;
    if (bs.nb > 0) { fb = sm.bnd_lane[0]; seg1_base = sm.segbase[1]; }       // (segbase[1]: only meaningful -- and only used -- when segment 1 has a lane)
    int mn[2] = {WSAT, WSAT}, mx[2] = {-WSAT, -WSAT};          // (no lane in the segment: no constraint, as +-2^40 was in the 64-bit form)
    if (!bs.isb && bs.seg <= 1) {
        const uint64_t sb = (bs.seg == 0) ? hb0 : (seg1_base & ~3ull);
        const long long m0 = (long long)(sb + (uint64_t)bs.exc.d0 - bs.leaf.base);
        const long long m1 = (long long)(sb + 1 + (uint64_t)bs.exc.d1 - bs.leaf.base);
        const int lo_ = sat_w(m0 < m1 ? m0 : m1), hi_ = sat_w(m0 > m1 ? m0 : m1);
        if (bs.seg == 0) { mn[0] = lo_; mx[0] = hi_; } else { mn[1] = lo_; mx[1] = hi_; }   // no runtime index: keeps mn/mx in registers
    }
    block_minmax2(sm, mn, mx);
    BSSM_STAMP(st, dbg, 2, STAMPCOL, stamper);
    // segment-0 prefix record: lanes [0, fb)
    if (bs.nb == 0) {
        if (t == NT - 1) {
            BlockRec br; br.prefix = pure_to_rec(bs.last_seg, hb0, (long long)clamp30i(-lim - mn[0] + 4), (long long)clamp30i(lim - mx[0] - 4));
            br.tail_from = NT; br.nside = 0; sink.rec(br);
        }
        return;
    }
    BSSM_STAMP(st, dbg, 2, 13, stamper);
    // One boundary, a binade crossing (REC_X1): the block's record is  pre . crossing . post  composed into ONE X1 record.  The four
    // hypotheses s = 0..3 are independent, so lanes 0..3 compose one each (rec_compose's arithmetic, lane-parallel: the serial form
    // ran ~600 instructions on one lane while the other 255 waited -- and with them every block of the grid, see above).
    if (t < 4) {
        const Pure ex0 = sm.bnd_excl[0];                     // composite of segment 0 (the first boundary lane's exclusive scan)
        const Rec leaf = sm.bnd_rec[0];
        const int C30 = 1 << 30;
        // (lanes 0..3 are one DPP quad: quad_perm [1,0,3,2] = 0xB1 and [2,3,0,1] = 0x4E exchange with lane ^ 1 and lane ^ 2)
        auto min4 = [&](int x) { x = min(x, dpp_i32<0xB1, 0xf>(x, x)); return min(x, dpp_i32<0x4E, 0xf>(x, x)); };
        auto max4 = [&](int x) { x = max(x, dpp_i32<0xB1, 0xf>(x, x)); return max(x, dpp_i32<0x4E, 0xf>(x, x)); };
        // pre = rec_identity(hb0) or pure_to_rec(ex0, hb0, ...): this lane's hypothesis
        const uint64_t pbase = hb0 & ~3ull;
        const int plo = (fb == 0) ? -C30 : clamp30i(-lim - mn[0] + 4), phi = (fb == 0) ? C30 : clamp30i(lim - mx[0] - 4);
        const uint64_t po = pbase + (uint64_t)t + ((fb == 0) ? 0ull : (uint64_t)((t & 1) ? ex0.d1 : ex0.d0));
        bool done1 = false;
        uint64_t fo = 0; int flo = 0, fhi = 0;
        BSSM_STAMP(st, dbg, 2, 5, stamper);
        if (bs.nb == 1 && leaf.kind == REC_X1) {
            // r = rec_compose(pre, leaf)
            bool ok = true;
            {
                const int64_t diff = (int64_t)(po - leaf.base);
                const int m = sat_w(diff);
                // (|leaf.lo|, |leaf.hi| <= 2^30 and mm within +-(2^30 - 16): the sums stay inside int32; a lane whose diff lies beyond that
                //  fails the range test below, so the record is refused whatever lo / hi say -- as in the 64-bit form)
                const int mm = m < -(C30 - 16) ? -(C30 - 16) : (m > C30 - 16 ? C30 - 16 : m);
                int lo = leaf.lo - min4(mm) + 8, hi = leaf.hi - max4(mm) - 8;
                lo = max(lo, plo); hi = min(hi, phi);
                ok = !(lo > 0 || hi < 3);
                ok = ok && !(m < leaf.lo || m > leaf.hi);
                const int64_t sx = diff & 3, q4 = diff - sx;
                const uint64_t os = (sx == 0) ? leaf.o[0] : (sx == 1) ? leaf.o[1] : (sx == 2) ? leaf.o[2] : leaf.o[3];
                fo = os + (uint64_t)(q4 >> 1); flo = lo; fhi = hi;
            }
            ok = __all(ok);                                       // (lanes 0..3 are the active ones here)
            BSSM_STAMP(st, dbg, 2, 6, stamper);
            if (ok && fb < NT - 1) {
                // r = rec_compose(r, post),  post = pure_to_rec(last_seg, seg1_base, ...)
                const uint64_t gbase = seg1_base & ~3ull;
                const int glo = clamp30i(-lim - mn[1] + 4), ghi = clamp30i(lim - mx[1] - 4);
                const int64_t diff = (int64_t)(fo - gbase);
                const int m = sat_w(diff);
                // (the factor 2: a crossing halves the distance.  Clamped to +-(2^30 - 1) before doubling: beyond that the exact double is
                //  below every flo / above every fhi, or makes the record invalid -- the same outcome)
                auto dbl = [&](int x) { const int c = x < -(C30 - 1) ? -(C30 - 1) : (x > C30 - 1 ? C30 - 1 : x); return 2 * c; };
                const int mm = m < -(C30 - 16) ? -(C30 - 16) : (m > C30 - 16 ? C30 - 16 : m);
                int lo = dbl(glo - min4(mm) + 8), hi = dbl(ghi - max4(mm) - 8);
                lo = max(lo, flo); hi = min(hi, fhi);
                bool ok2 = !(lo > 0 || hi < 3);
                ok2 = ok2 && !(m < glo || m > ghi);
                const int64_t sx = diff & 3, q4 = diff - sx;
                // post.o[sx] = gbase + sx + (sx odd ? d1 : d0)
                fo = gbase + (uint64_t)sx + (uint64_t)((sx & 1) ? bs.last_seg.d1 : bs.last_seg.d0) + (uint64_t)q4;
                flo = lo; fhi = hi;
                ok = __all(ok2);
            }
            done1 = ok;
        }
        BSSM_STAMP(st, dbg, 2, 7, stamper);
        // lane 0 collects the four outgoing states (of the composite, or of pre) and publishes the record
        const uint64_t src = done1 ? fo : po;
        const uint64_t o1 = (uint64_t)dpp_i64<0x55, 0xf>((long long)src, (long long)src), o2 = (uint64_t)dpp_i64<0xAA, 0xf>((long long)src, (long long)src),
                       o3 = (uint64_t)dpp_i64<0xFF, 0xf>((long long)src, (long long)src);      // quad_perm broadcasts of lanes 1, 2, 3
        if (t == 0) {
            BlockRec br;
            br.prefix.base = pbase; br.prefix.o[0] = src; br.prefix.o[1] = o1; br.prefix.o[2] = o2; br.prefix.o[3] = o3; br.prefix.pad = 0;
            if (done1) { br.prefix.kind = REC_X1; br.prefix.lo = (int32_t)flo; br.prefix.hi = (int32_t)fhi; br.tail_from = NT; br.nside = 0; }
            else {
                br.prefix.kind = REC_PURE; br.prefix.lo = (int32_t)plo; br.prefix.hi = (int32_t)phi;
                if (bs.nb <= MAXBND) { br.tail_from = NT; br.nside = bs.nb; }     // side list below
                else { br.tail_from = fb; br.nside = 0; }                          // too many: literal tail
                atomicAdd((unsigned long long*)&st->stat_hard_blocks, 1ull);
            }
            sink.rec(br);
            sm.fail = done1 ? 1 : 0;      // reuse as "record complete" flag for the block
        }
    }
    BSSM_STAMP(st, dbg, 2, 14, stamper);
    __syncthreads();
    BSSM_STAMP(st, dbg, 2, 15, stamper);
    if (sm.fail || bs.nb > MAXBND) return;
    // ---- side list: every boundary leaf + the PURE segment after it, with that segment's window ----
    for (int k = t; k <= MAXBND; k += NT) { sm.smin[k] = 0x7fffffff; sm.smax[k] = -0x7fffffff; }
    // (a segment's hypothesis base is that of its first lane: sm.segbase, published by block_scan)
    __syncthreads();
    if (!bs.isb && bs.seg >= 1) {
        const uint64_t sb = sm.segbase[bs.seg] & ~3ull;
        long long m0 = (long long)(sb + (uint64_t)bs.exc.d0 - bs.leaf.base);
        long long m1 = (long long)(sb + 1 + (uint64_t)bs.exc.d1 - bs.leaf.base);
        long long lo_ = m0 < m1 ? m0 : m1, hi_ = m0 > m1 ? m0 : m1;
        const long long C = (1ll << 30);
        lo_ = lo_ < -C ? -C : (lo_ > C ? C : lo_); hi_ = hi_ < -C ? -C : (hi_ > C ? C : hi_);
        atomicMin(&sm.smin[bs.seg], (int)lo_);
        atomicMax(&sm.smax[bs.seg], (int)hi_);
    }
    __syncthreads();
    if (bs.isb) {
        const int k = bs.seg;
        SideEntry e;
        e.leaf = bs.leaf;
        const Pure post = (k + 1 < bs.nb) ? sm.bnd_excl[k + 1] : bs.last_seg;
        e.d0 = post.d0; e.d1 = post.d1;
        const bool has_post = (sm.smin[k + 1] != 0x7fffffff);
        e.post_base = has_post ? (sm.segbase[k + 1] & ~3ull) : 0ull;
        const long long lo = has_post ? (-(long long)lim - (long long)sm.smin[k + 1] + 4) : -(1ll << 30);
        const long long hi = has_post ? ((long long)lim - (long long)sm.smax[k + 1] - 4) : (1ll << 30);
        e.lo = (int32_t)(lo < -(1ll << 30) ? -(1ll << 30) : (lo > (1ll << 30) ? (1ll << 30) : lo));
        e.hi = (int32_t)(hi < -(1ll << 30) ? -(1ll << 30) : (hi > (1ll << 30) ? (1ll << 30) : hi));
        e.lane = t; e.pad = has_post ? 1 : 0;
#pragma unroll
        for (int q = 0; q < EL; q++) e.terms[q] = v[q];     // so the resolver can re-run this leaf without chasing w[]
        sink.side(k, e);
    }
}

// ---------------------------------------------------------------------------
// k_local: the block's record for the grid-level resolve.
//   no boundary            -> PURE record (d0,d1) + validity window
//   one X1 boundary        -> X1 record  (pre . crossing . post)
//   incoming state is +0   -> the block resolves itself now: ABS record
//   otherwise              -> prefix record up to the first boundary + tail_from
// ---------------------------------------------------------------------------
// FROM_LW (filter path, MODE_W): the kernel also IS the normalisation step -- it turns log-weights
// into weights  w = exp(lw - max) / sum  (R/particle_filter_core.R:205-207), writes them, and derives
// the approximate block prefixes from the log-sum-exp partials instead of a separate pass.
// The grid-level resolve of the pass BEFORE a kernel, run inside that kernel right after its own loads have been issued
// (their latency then hides under the resolve).  NoResolve: the value comes from a k_resolve launch (or is not needed).
struct NoResolve { static constexpr bool active = false; __device__ __forceinline__ uint64_t operator()() const { return 0ull; } };

struct FromLw {
    const double* lw; double* w_out; const double* pm; const double* ps; const double* pq; int nb;
    const double* xw; double yw, syw, lsyw;   // xw != nullptr: log-weights = dnorm(yw, xw[j], syw, log = TRUE), evaluated here exactly as k_step evaluated them (it did not store them)
    const unsigned long long* gmax;      // key of max(pm[0..nb)) left by the step kernel's atomics, or nullptr (then reduced here)
    int fold;                            // 1: the resampler's own normalisation (prob = w / sum(w), src/resampling.cpp:24,51) is folded into
                                         // this kernel's  w = exp(lw - max) / sum: total is taken as 1 and this pass's records ARE the
                                         // records of cumsum(prob) -- no exact total, no second local pass
    int lead, pub;                       // the block that records the per-observation scalars (0) / publishes the prefixes (nblk / 2);
                                         // other values only when a filter's blocks are sharded over ranks (every rank keeps its own run state)
    double* ain_out;
    // per-observation bookkeeping that only needs the partials (R/particle_filter_core.R:189-218): every block
    // derives the same numbers; block 0 records them
    int plan;                 // PLAN_PF / PLAN_AUX
    long long N; int obs_i; int resample_algorithm; double threshold;
    double* ess_out; double* llh_out; int* resampled_out;
};

// The body of k_local for workgroup `bidx` of `nblk`; shared storage is handed in so that the batched small-filter kernel
// (k_pf_batch: one workgroup runs a whole filter, nblk == 1) executes exactly this code.  MAXBL: capacity of the partials.
template <int MODE, bool FROM_LW, int MAXBL, bool LIT = false, class Pro = NoResolve>
__device__ __forceinline__ void local_block(SegSmem& sm, uint64_t* tin /* [NT + 1] */, double* es /* [MAXBL], FROM_LW only */, const int bidx, const int nblk,
                                            const double* __restrict__ w, long long nw, const double* __restrict__ ain,
                                            int lim, BlockRec* __restrict__ brec, SideList* __restrict__ side, DevState* st,
                                            const FromLw& f, const Pro pro = Pro(), double* __restrict__ ain_p_out = nullptr,
                                            const int bstride = BREC_STRIDE)
{
    const int t = threadIdx.x;
    const long long b0 = (long long)bidx * EB;
    // run-state words and this lane's terms are fetched together (one memory round trip, not two)
    const int s_dead = st->dead, s_do = st->do_resample, dbg = st->debug_stop;
    const uint32_t s_flags = st->flags;
    double total = (MODE == MODE_P && !Pro::active) ? b2d(st->total_bits) : 1.0;
    double vraw[EL];
    if (!FROM_LW) load_terms<MODE_W>(w, nw, 1.0, b0 + (long long)t * EL, vraw);
    double a_in_pre = FROM_LW ? 0.0 : ain[bidx];
    if (s_dead || s_flags) return;
    if (!FROM_LW && !s_do) return;
    if constexpr (Pro::active) {
        // this pass's workgroups resolve the MODE_W pass before them themselves (exact total = sum(w)), with their own
        // weight loads already in flight; `ain` holds the prefixes of w, this block's prefix of w / total is derived here
        const uint64_t tb = pro();
        const double tot = b2d(tb);
        if (bidx == f.lead && t == 0) {
            st->total_bits = tb;
            if (tot == 0.0) atomicOr(&st->flags, FLAG_ZERO_SUM);       // src/resampling.cpp:8,22,49
            if (!isfinite(tot)) atomicOr(&st->flags, FLAG_NONFINITE);
        }
        if (tot == 0.0 || !isfinite(tot)) return;
        total = tot;
        a_in_pre = a_in_pre / tot;
        if (t == 0) ain_p_out[bidx] = a_in_pre;
    }
    const bool stamper = (t == 0 && bidx == ((nblk > 100 && dbg != 98) ? 100 : 0)); (void)stamper;
    BSSM_STAMP(st, dbg, 2, FROM_LW ? 4 : 0, stamper);
    double v[EL];
    double a_in;
    if (FROM_LW) {
        // global (max, sum exp) from the per-block partials, and -- from the same numbers -- the approximate
        // exclusive block prefixes of w:  sum_b w = ps[b] exp(pm[b] - M) / S
        // issue this thread's log-weight loads first: their latency hides under the reductions below
        const long long j0 = b0 + (long long)t * EL;
        double l8[EL];
        const double* lsrc = f.xw ? f.xw : f.lw;
        if (j0 + EL <= nw) {
            const double2* p2 = reinterpret_cast<const double2*>(lsrc + j0);
#pragma unroll
            for (int k = 0; k < EL / 2; k++) { const double2 q2 = p2[k]; l8[2 * k] = q2.x; l8[2 * k + 1] = q2.y; }
        } else {
#pragma unroll
            for (int k = 0; k < EL; k++) l8[k] = (j0 + k < nw) ? lsrc[j0 + k] : (f.xw ? 0.0 : -INFINITY);
        }
        // Thread t holds the partials of the L consecutive blocks t L .. t L + L - 1 (L = nb / NT rounded up): all three
        // arrays and the grid maximum are fetched up front (one memory round trip).
        constexpr int KMAX = MAXBL / NT;
        const int L = (f.nb + NT - 1) / NT;
        double pmv[KMAX], psv[KMAX], pqv[KMAX];
#pragma unroll
        for (int k = 0; k < KMAX; k++) {
            const int i = t * L + k;
            const bool in = (k < L) && (i < f.nb);
            pmv[k] = in ? f.pm[i] : -INFINITY; psv[k] = in ? f.ps[i] : 0.0; pqv[k] = in ? f.pq[i] : 0.0;
        }
        double M;
        if (f.gmax) {                             // every wave reduces the GM_SLOTS partial maxima itself: one load, no barrier
            const unsigned long long kq = f.gmax[(t % GM_SLOTS) * GM_STRIDE];
            M = wave_max(kq ? key_f64(kq) : -INFINITY);           // (key 0: no block added to that slot)
        } else {
            // one block (nb == 1, finite max): M = pm[0] -- exactly what the reduction returns (it takes max with -inf)
            const bool one = (MAXBL == NT) && (f.nb == 1) && (f.pm[0] > -INFINITY);
            double m = -INFINITY;
#pragma unroll
            for (int k = 0; k < KMAX; k++) m = fmax(m, pmv[k]);
            M = one ? f.pm[0] : block_max(m, sm.sh4);
        }
        const bool degenerate = (f.plan == PLAN_PF) && (M < -1e8);       // all(log_weights < -1e8)  (:189-202)
        if (degenerate) {
            if (bidx == f.lead && t == 0) {
                st->loglike = -INFINITY; f.llh_out[f.obs_i - 1] = -INFINITY; st->dead = f.obs_i; st->do_resample = 0;
            }
            return;
        }
        // sum_b w = ps[b] exp(pm[b] - M) / S: this thread's slice, then ONE block scan gives the slice prefixes and S
        double esv[KMAX];
        double ts0 = 0.0, tq = 0.0;
#pragma unroll
        for (int k = 0; k < KMAX; k++) {
            double x = 0.0;
            if (pmv[k] > -INFINITY) { const double ex = exp(pmv[k] - M); x = psv[k] * ex; tq += pqv[k] * ex * ex; }
            esv[k] = x; ts0 += x;
        }
        double pre, S, sq;
        {
            const int lane = t & 63, wave = t >> 6;
            const double inc = wave_incl_sum(ts0);
            const double exc = dpp_f64<DPP_WAVE_SHR1, 0xf>(0.0, inc);        // lane 0 keeps 0
            const double wq = wave_sum(tq);
            __syncthreads();
            if (lane == 63) { sm.sh4[wave] = inc; sm.sh4[8 + wave] = wq; }
            __syncthreads();
            pre = exc;
            for (int i = 0; i < wave; i++) pre += sm.sh4[i];
            S = tree_sum<NWV>(sm.sh4);
            sq = tree_sum<NWV>(sm.sh4 + 8);
        }
        int doit = 1;
        if (f.plan == PLAN_PF) {
            const double ess = 1.0 / (sq / (S * S));                                          // :211
            doit = (f.resample_algorithm == 0) ? 0 : (f.resample_algorithm == 1) ? 1 : (ess < f.threshold);   // :214-218
            if (bidx == f.lead && t == 0) {
                const double ll = st->loglike + (M + log(S) - log((double)f.N));              // :208
                st->loglike = ll; f.llh_out[f.obs_i - 1] = ll;                                // :209
                st->do_resample = doit;
                f.ess_out[f.obs_i] = doit ? (double)f.N : ess;                                // :212,:223
                if (f.resampled_out) f.resampled_out[f.obs_i - 1] = doit;
                st->ess = ess; st->lse_max = M; st->lse_sum = S;
                if (doit) { st->cur_call = st->res_calls; st->res_calls += 1; }
                if (f.fold) st->total_bits = d2b(1.0);
            }
        } else if (bidx == f.lead && t == 0) { st->do_resample = 1; st->cur_call = st->res_calls; st->res_calls += 1; if (f.fold) st->total_bits = d2b(1.0); }
        if (t == bidx / L) {                      // the one lane whose slice holds this block: one division
            double pp = pre;
#pragma unroll
            for (int k = 0; k < KMAX; k++) if (k < bidx % L) pp += esv[k];
            sm.bcast = pp / S;
        }
        if (bidx == f.pub) {                      // one (ordinary) block also publishes every block's prefix, for ain_p later;
                                                             // not block 0: that one already walks the exact-zero head
            double pp = pre;
#pragma unroll
            for (int k = 0; k < KMAX; k++) {
                const int i = t * L + k;
                if (k < L && i < f.nb) f.ain_out[i] = pp / S;
                pp += esv[k];
            }
        }
        __syncthreads();
        a_in = sm.bcast;
        BSSM_STAMP(st, dbg, 2, 8, stamper);
        if (f.xw) {
#pragma unroll
            for (int k = 0; k < EL; k++) l8[k] = r_dnorm_log(f.yw, l8[k], f.syw, f.lsyw);
        }
#pragma unroll
        for (int k = 0; k < EL; k++) v[k] = (j0 + k < nw) ? exp(l8[k] - M) / S : 0.0;        // :205-207
        if (j0 + EL <= nw || (LIT && j0 < nw)) {             // (LIT: the partly filled lane stores its zero padding too -- the in-order pass reads whole lanes)
            double2* p2 = reinterpret_cast<double2*>(f.w_out + j0);
#pragma unroll
            for (int k = 0; k < EL / 2; k++) bulk_store16<BSSM_ST_W>(reinterpret_cast<double*>(p2 + k), v[2 * k], v[2 * k + 1]);
        } else {
#pragma unroll
            for (int k = 0; k < EL; k++) if (j0 + k < nw) f.w_out[j0 + k] = v[k];
        }
        if (!doit) return;                                   // SIS / SISAR without a resample: weights are all that is needed
        // NaN/Inf log-weights: the scan stands down.  (lw <= M, so every exp(lw - M) / S is finite unless S or M is not: a NaN
        // log-weight makes its block's sum NaN, +Inf makes M +Inf, all -Inf leaves S == 0 -- no pass over the weights needed)
        if (!(S > 0.0) || !isfinite(S) || !isfinite(M)) {
            if (t == 0) atomicOr(&st->flags, FLAG_NONFINITE);
            return;
        }
        BSSM_STAMP(st, dbg, 2, 9, stamper);
    } else {
#pragma unroll
        for (int k = 0; k < EL; k++) v[k] = (MODE == MODE_P) ? vraw[k] / total : vraw[k];   // prob = weights / total (src/resampling.cpp:24,51)
        a_in = a_in_pre;
    }
    BSSM_STAMP(st, dbg, 2, 1, stamper && v[0] >= 0.0);
    if (LIT) {
        // one block of few terms, incoming state +0 (FROM_LW: the terms are already in f.w_out, in order, zero beyond nw)
        __syncthreads();
        block_literal_terms(tin, FROM_LW ? f.w_out : w, (int)nw);
        if (t == NT - 1) { BlockRec br; br.prefix = rec_abs(tin[NT]); br.tail_from = NT; br.nside = 0; store_brec(brec, bidx, br, bstride); }
        return;
    }
    BlockScan bs;
    block_scan<MODE>(sm, v, a_in, lim, bs);
    __syncthreads();
    BSSM_STAMP(st, dbg, 2, FROM_LW ? 10 : 2, stamper);
    GlobalRecSink<MODE> sink; sink.brec = brec; sink.sides = side; sink.bidx = bidx; sink.bstride = bstride;
    sink.w = w; sink.nw = nw; sink.total = total; sink.b0 = b0; sink.st = st;
    if constexpr (FROM_LW) block_record_tail<MODE, 11>(sm, tin, bs, v, a_in, lim, st, dbg, stamper, sink);
    else block_record_tail<MODE, 3>(sm, tin, bs, v, a_in, lim, st, dbg, stamper, sink);
}


// ---------------------------------------------------------------------------
// k_resolve (one workgroup): exact incoming state cin[b] of every block.
// Same scheme one level up: each thread owns CB consecutive block records; a
// run of PURE blocks is a PURE chunk, anything else is a boundary chunk.
// ---------------------------------------------------------------------------
// literal-term counters: terms re-run from a side entry's own copy count as they are; a run that had to read the weights of a
// block from HBM (literal_run) also adds LIT_FROM_W -- a fused launch, whose weights never reach HBM, must then stand down
constexpr long long LIT_FROM_W = 1ll << 40;
template <int MODE>
__device__ __forceinline__ uint64_t block_out_exact_body(const BlockRec& br, const SideList* __restrict__ side, const double* __restrict__ w,
                                                         long long nw, double total, long long b, uint64_t in, long long& lit)
{
    bool ok = true;
    uint64_t o = rec_step(br.prefix, in, ok);
    const long long e0 = b * EB, e1 = (e0 + EB < nw) ? e0 + EB : nw;
    if (!ok) {                                   // the prefix record does not cover this state: whole block literally
        lit += ((e1 > e0) ? e1 - e0 : 0) + LIT_FROM_W;
        return literal_run<MODE>(w, nw, total, e0, e1, in);
    }
    if (br.nside > 0) {
        const SideEntry* se = side[b].e;
        for (int k = 0; k < br.nside; k++) {
            const SideEntry e = se[k];
            bool ok2 = true;
            uint64_t o2 = rec_step(e.leaf, o, ok2);
            if (!ok2) {                          // HARD leaf (or window miss): its 8 terms literally
                double c = b2d(o);
#pragma unroll
                for (int q = 0; q < EL; q++) c = c + e.terms[q];
                o2 = d2b(c);
                lit += EL;
            }
            o = o2;
            if (e.pad) {                         // PURE segment after the boundary
                const int64_t diff = (int64_t)(o - e.post_base);
                if (diff < (int64_t)e.lo || diff > (int64_t)e.hi) {
                    const long long j0 = e0 + (long long)(e.lane + 1) * EL;   // not covered: rest of the block literally
                    if (j0 < e1) { lit += e1 - j0 + LIT_FROM_W; o = literal_run<MODE>(w, nw, total, j0, e1, o); }
                    return o;
                }
                Pure p; p.d0 = e.d0; p.d1 = e.d1;
                o = pure_step(p, o);
            }
        }
        return o;
    }
    if (br.tail_from >= NT) return o;
    const long long j0 = e0 + (long long)br.tail_from * EL;
    if (j0 >= e1) return o;
    lit += e1 - j0 + LIT_FROM_W;
    return literal_run<MODE>(w, nw, total, j0, e1, o);
}

// Out of line (rare, and the resolver's hot instruction stream should stay short); results by value so that the
// caller's counters stay in registers.
struct OutLit { uint64_t out; long long lit; };
template <int MODE>
__device__ __attribute__((noinline)) OutLit block_out_exact_nl(const BlockRec* br, const SideList* side, const double* w,
                                                               long long nw, double total, long long b, uint64_t in)
{
    OutLit r; r.lit = 0;
    r.out = block_out_exact_body<MODE>(*br, side, w, nw, total, b, in, r.lit);
    return r;
}
template <int MODE>
__device__ __forceinline__ uint64_t block_out_exact(const BlockRec& br, const SideList* __restrict__ side, const double* __restrict__ w,
                                                    long long nw, double total, long long b, uint64_t in, long long& lit)
{
    const OutLit r = block_out_exact_nl<MODE>(&br, side, w, nw, total, b, in);
    lit += r.lit;
    return r.out;
}

// ---------------------------------------------------------------------------
// resolve_in_block: the grid-level resolve, run by EVERY workgroup of the consuming kernel for itself (B <= 2 NT blocks).
// The records of a pass are tiny (64 B a block) and the resolve is a latency chain, not work: as a launch of its own
// (k_resolve, one workgroup) it cost 6.4-7.4 us per pass at N = 2^20 with the other 255 CUs idle.
// Here each consumer workgroup loads the records it needs straight into registers (two per thread: one round trip), folds
// them into PURE runs around its non-PURE blocks, scans the runs with its own four waves (seg_excl_scan_tail: a thread's run
// BEHIND its last non-PURE block travels on, so every non-PURE block is exactly one link) and one wave walks the chain of
// links (walk_chain_regs); every thread then checks its runs' windows against the exact states.
// Returns the exact state after blocks [0, upto)   (upto == B in a MODE_W pass: total = sum(w), src/resampling.cpp:20).
// Anything the records do not cover (a window miss, three non-PURE blocks in one lane's range, more than 64 of them) falls
// back to one lane walking the blocks in order: exact, merely slow.
// ---------------------------------------------------------------------------
struct ResolveSmem {
    BlockRec bnd[64];                // the non-PURE blocks' records, in walk order
    Pure ex[64];                     // the PURE run folded in front of each
    int ebidx[64];
    uint64_t sout[65];               // exact state after link j (sout[j + 1]); sout[0] = 0
    __attribute__((aligned(16))) SideEntry sideC[64];
    uint64_t result;
    int fail;
};

struct LaneRun { Pure p; long long wlo, whi; uint64_t base; int any; };      // a run of PURE blocks

__device__ __forceinline__ void run_reset(LaneRun& r) { r.p.d0 = 0; r.p.d1 = 0; r.wlo = -(1ll << 40); r.whi = (1ll << 40); r.base = 0; r.any = 0; }
__device__ __forceinline__ void run_fold(LaneRun& r, const Rec& pr)
{
    if (!r.any) { r.base = pr.base; r.any = 1; }
    const long long m0 = (long long)(r.base + (uint64_t)r.p.d0 - pr.base);
    const long long m1 = (long long)(r.base + 1 + (uint64_t)r.p.d1 - pr.base);
    const long long mmin = m0 < m1 ? m0 : m1, mmax = m0 > m1 ? m0 : m1;
    const long long lo = (long long)pr.lo - mmin + 2, hi = (long long)pr.hi - mmax - 2;
    r.wlo = lo > r.wlo ? lo : r.wlo; r.whi = hi < r.whi ? hi : r.whi;
    Pure q; q.d0 = (int64_t)(pr.o[0] - pr.base); q.d1 = (int64_t)(pr.o[1] - (pr.base + 1));
    r.p = pure_compose(r.p, q);
}
// append the run q to the run r (same window algebra as a block record: q's window is relative to q.base)
__device__ __forceinline__ void run_merge(LaneRun& r, const LaneRun& q)
{
    if (!q.any) return;
    if (!r.any) { r = q; return; }
    const long long m0 = (long long)(r.base + (uint64_t)r.p.d0 - q.base);
    const long long m1 = (long long)(r.base + 1 + (uint64_t)r.p.d1 - q.base);
    const long long mmin = m0 < m1 ? m0 : m1, mmax = m0 > m1 ? m0 : m1;
    const long long lo = q.wlo - mmin + 2, hi = q.whi - mmax - 2;
    r.wlo = lo > r.wlo ? lo : r.wlo; r.whi = hi < r.whi ? hi : r.whi;
    r.p = pure_compose(r.p, q.p);
}
__device__ __forceinline__ bool run_ok(const LaneRun& r, uint64_t ent)
{
    if (!r.any) return true;
    const long long diff = (long long)(ent - r.base);
    return diff >= r.wlo && diff <= r.whi;
}

// NTX: threads of the calling workgroup (NT inside the scan kernels; NTR in k_resolve_all, which also EMITs every block's
// exact incoming state to cin_out).
// LateSide: where the LAST block's side entry comes from when it is not in `side` yet.  The multi-launch kernels read it from HBM at the
// start (NoLateSide); the fused launch's resolver fetches it from the wire only when the walk reaches it -- that entry is published ~0.5 us
// after its block's record, and waiting for it in front of the scan put it on every workgroup's critical path.
struct NoLateSide {
    static constexpr bool active = false;
    __device__ __forceinline__ unsigned long long issue() const { return 0ull; }
    __device__ __forceinline__ bool finish(SideEntry*, unsigned long long) const { return true; }
};
template <int MODE, int NTX = NT, bool EMIT = false, class LateSide = NoLateSide>
__device__ __forceinline__ uint64_t resolve_in_block(SegSmem& sm, ResolveSmem& rs, const BlockRec* __restrict__ brec,
                                                     const SideList* __restrict__ side, const int B, const int upto,
                                                     const double* __restrict__ w, long long nw, double total, DevState* st,
                                                     const bool count_stats, uint64_t* __restrict__ cin_out = nullptr,
                                                     long long* lit_out = nullptr /* this thread's count of literally re-run terms */,
                                                     const BlockRec* pre0 = nullptr, const BlockRec* pre1 = nullptr /* the thread's records, already in registers */,
                                                     const LateSide late = LateSide())
{
    const int t = threadIdx.x;
    if (upto <= 0) return 0ull;                        // (block-uniform)
    const int dbg = st->debug_stop; (void)dbg;
    const bool stamper = (t == 0 && (EMIT || blockIdx.x == 100)); (void)stamper;
    BSSM_STAMP(st, dbg, MODE, 0, stamper);
    // ---- all threads: thread t holds CB (1 or 2) consecutive blocks, straight from global memory into registers ----
    const int CB = (B + NTX - 1) / NTX;                // (the caller guarantees B <= 2 NTX)
    const int c0 = t * CB, c1 = (c0 + CB < upto) ? c0 + CB : upto;
    BlockRec r0, r1;
    r0.prefix = rec_identity(0); r0.tail_from = NT; r0.nside = 0; r1 = r0;
    if (c0 < c1) r0 = pre0 ? *pre0 : load_brec(brec, c0);
    if (c0 + 1 < c1) r1 = pre1 ? *pre1 : load_brec(brec, c0 + 1);
    if (t == 0) { rs.sout[0] = 0; rs.fail = 0; }
    // the last block of a full pass nearly always carries one side entry (the lanes next to cum == 1.0 are never PURE):
    // fetch it now, with the records, so that the walk does not wait for a dependent global load later
    if (!LateSide::active && upto == B && t >= NTX - (int)(sizeof(SideEntry) / 16)) {
        const int q = t - (NTX - (int)(sizeof(SideEntry) / 16));
        reinterpret_cast<uint4*>(&rs.sideC[63])[q] = reinterpret_cast<const uint4*>(&side[upto - 1].e[0])[q];
    }
    // head = the PURE run in front of the thread's first non-PURE block, tail = the run behind its last one
    //   (P P): head = both | (X P): tail = block 1 | (P X): head = block 0 | (X X): neither
    LaneRun head, tail;
    run_reset(head); run_reset(tail);
    const bool x0 = (c0 < c1) && !(r0.prefix.kind == REC_PURE && r0.tail_from >= NT && r0.nside == 0);
    const bool x1 = (c0 + 1 < c1) && !(r1.prefix.kind == REC_PURE && r1.tail_from >= NT && r1.nside == 0);
    if (c0 < c1 && !x0) run_fold(head, r0.prefix);
    if (c0 + 1 < c1 && !x1) { if (x0) run_fold(tail, r1.prefix); else run_fold(head, r1.prefix); }
    const int nent = (x0 ? 1 : 0) + (x1 ? 1 : 0);
    int ent_before, E;
    const Pure carry = seg_excl_scan_tail<NTX / 64>(sm, nent ? tail.p : head.p, nent > 0, nent, ent_before, E);
    const bool fits = (E <= 64);
    // the links: every non-PURE block with the PURE run in front of it
    if (fits && nent) {
        int slot = ent_before;
        if (x0) { rs.bnd[slot] = r0; rs.ebidx[slot] = c0; rs.ex[slot] = carry; slot++; }
        if (x1) { rs.bnd[slot] = r1; rs.ebidx[slot] = c0 + 1; if (x0) { rs.ex[slot].d0 = 0; rs.ex[slot].d1 = 0; } else rs.ex[slot] = pure_compose(carry, head.p); }
    }
    __syncthreads();
    BSSM_STAMP(st, dbg, MODE, 2, stamper);
    long long lit = 0;
    if (fits && t < 64) {
        // ---- boundary walk, one wave: lane j owns link j ----
        const int lane = t;
        const bool have = lane < E;
        // (the last block's side entry: the loads go out now, ~4k cycles of link set-up and chain walk before the value is looked at)
        unsigned long long late_v = 0ull;
        if constexpr (LateSide::active) { if (upto == B) late_v = late.issue(); }
        int bidx = -1;
        Pure ex; ex.d0 = 0; ex.d1 = 0;
        Rec pr = rec_identity(0);
        int nside = 0;
        bool slow = false;
        if (have) {
            bidx = rs.ebidx[lane]; ex = rs.ex[lane];
            pr = rs.bnd[lane].prefix; nside = rs.bnd[lane].nside;
            slow = (nside > 1) || (rs.bnd[lane].tail_from < NT);
            if (nside == 1 && !slow && !(upto == B && bidx == B - 1 && lane < 63)) {      // (the last block's entry is already in slot 63)
                const uint4* sp = reinterpret_cast<const uint4*>(&side[bidx].e[0]);
                uint4* dp = reinterpret_cast<uint4*>(&rs.sideC[lane]);
#pragma unroll
                for (int q = 0; q < (int)(sizeof(SideEntry) / 16); q++) dp[q] = sp[q];
            }
        }
        const int side_slot = (upto == B && bidx == B - 1 && lane < 63) ? 63 : lane;
        uint64_t sw = 0, my_out = 0, my_s0 = 0;
        const bool special = have && (slow || nside == 1);
        const StepFn fn = stepfn_build(ex, pr);          // "PURE run + this block's record" folded into constants
        const WalkFn wfn = walkfn_from(fn, have && !special);
        BSSM_STAMP(st, dbg, MODE, 3, stamper && wfn.sh >= 0);
        const int j0 = walk_chain_regs(wfn, E, sw, my_s0, my_out);
        BSSM_STAMP(st, dbg, MODE, 4, stamper && sw != 1);
        if constexpr (LateSide::active) {
            // the last block's side entry (slot 63), if the walk is about to need it: the last link is block B - 1 with exactly one entry
            if (upto == B && E > 0 && E <= 64) {
                const int lb = __builtin_amdgcn_readlane(bidx, E - 1), ln = __builtin_amdgcn_readlane(nside, E - 1), ls = __builtin_amdgcn_readlane((int)slow, E - 1);
                if (lb == B - 1 && ln == 1 && !ls) { if (!late.finish(&rs.sideC[63], late_v)) lit += LIT_FROM_W; }      // (a time-out voids the fused result)
            }
        }
        for (int j = j0; j < E; j++) {
            bool ok;
            uint64_t o = stepfn_apply(fn, sw, ok);
            const uint64_t e = pure_step(ex, sw);         // (only lane j's copy is used)
            if (lane == j && (special || !ok)) {             // rare: side entry / tails / window miss
                bool done = false;
                if (!slow && ok && nside == 1) {
                    const SideEntry& se = rs.sideC[side_slot];
                    bool ok2 = true;
                    uint64_t o2 = rec_step(se.leaf, o, ok2);
                    if (!ok2) {                          // HARD leaf: its 8 terms literally (kept in the entry)
                        double c = b2d(o);
#pragma unroll
                        for (int q = 0; q < EL; q++) c = c + se.terms[q];
                        o2 = d2b(c);
                    }
                    done = true;
                    if (se.pad) {
                        const int64_t diff = (int64_t)(o2 - se.post_base);
                        if (diff < (int64_t)se.lo || diff > (int64_t)se.hi) done = false;
                        else { Pure pp; pp.d0 = se.d0; pp.d1 = se.d1; o2 = pure_step(pp, o2); }
                    }
                    if (done) o = o2;
                }
                if (!done) o = block_out_exact<MODE>(rs.bnd[lane], side, w, nw, total, bidx, e, lit);   // the general (slower) routine
            }
            if (lane == j) my_out = o;
            sw = readlane_u64(o, j);
        }
        BSSM_STAMP(st, dbg, MODE, 5, stamper && sw != 1);
        if (have) rs.sout[lane + 1] = my_out;
    }
    if (!fits && t == 0) rs.fail = 1;
    __syncthreads();
    BSSM_STAMP(st, dbg, MODE, 6, stamper);
    // ---- all threads: the exact states entering the PURE runs must lie inside the runs' windows ----
    if (fits && c0 < c1) {
        const uint64_t ent = pure_step(carry, rs.sout[ent_before]);      // exact state at the thread's first block
        bool ok = true;
        uint64_t end;
        if (nent == 0) { ok = run_ok(head, ent); end = pure_step(head.p, ent); }
        else {
            ok = run_ok(head, ent);                                      // (empty unless (P X))
            const uint64_t after = rs.sout[ent_before + nent];            // exact state behind the thread's last non-PURE block
            ok = ok && run_ok(tail, after);
            end = pure_step(tail.p, after);
        }
        if (!ok) rs.fail = 1;
        if (c1 == upto) rs.result = end;                                 // this thread holds the last block of the range
        if constexpr (EMIT) {
            // every block's exact incoming state: the thread's first block enters at `ent`; its second one behind the first
            // block's link (non-PURE) or behind the first block's own PURE record
            cin_out[c0] = ent;
            if (c0 + 1 < c1) {
                LaneRun h0; run_reset(h0); run_fold(h0, r0.prefix);
                cin_out[c0 + 1] = x0 ? rs.sout[ent_before + 1] : pure_step(h0.p, ent);
            }
        }
    }
    __syncthreads();
    if (rs.fail) {
        // the records did not cover the exact states: one lane walks every block (each step is still exact)
        if (pre0) {      // (records handed over in registers: the walking lane reads them from `brec`, which the caller owns)
            if (c0 < c1) store_brec(const_cast<BlockRec*>(brec), c0, r0);
            if (c0 + 1 < c1) store_brec(const_cast<BlockRec*>(brec), c0 + 1, r1);
            __syncthreads();
        }
        if (t == 0) {
            uint64_t sf = 0;
            for (int b = 0; b < upto; b++) { const BlockRec r = load_brec(brec, b); if (EMIT) cin_out[b] = sf; sf = block_out_exact<MODE>(r, side, w, nw, total, b, sf, lit); }
            rs.result = sf;
            if (count_stats) atomicAdd((unsigned long long*)&st->stat_serial_walks, 1ull);
        }
        __syncthreads();
    }
    if (lit && count_stats) atomicAdd((unsigned long long*)&st->stat_literal_terms, (unsigned long long)(lit & (LIT_FROM_W - 1)));
    if (lit_out) *lit_out = lit;
    BSSM_STAMP(st, dbg, MODE, 7, stamper);
#ifdef BSSM_DEV_STAMPS
    if ((dbg == 99 || dbg == 98) && stamper) { st->stamps[MODE][8] = E; st->stamps[MODE][9] = upto; }
#endif
    return rs.result;
}

template <int MODE>
struct InResolve {
    static constexpr bool active = true;
    SegSmem* sm; ResolveSmem* rs; const BlockRec* brec; const SideList* side; int B, upto;
    const double* w; long long nw; double total; DevState* st; bool stats;
    __device__ __forceinline__ uint64_t operator()() const { return resolve_in_block<MODE>(*sm, *rs, brec, side, B, upto, w, nw, total, st, stats); }
};

// INRES (MODE_P only): this pass's workgroups resolve the MODE_W pass before them themselves -- exact total = sum(w), their
// own approximate prefix of w / total -- instead of a k_resolve<W> launch in between.
// (INRES is a template parameter so that the plain variants keep their register budget: the resolve costs ~40 VGPRs)
template <int MODE, bool FROM_LW, bool INRES = false>
__global__ __launch_bounds__(NT) void k_local(const double* __restrict__ w, long long nw, const double* __restrict__ ain,
                                              int lim, BlockRec* __restrict__ brec, SideList* __restrict__ side, DevState* st,
                                              FromLw f, const BlockRec* __restrict__ prev_brec, const SideList* __restrict__ prev_side,
                                              double* __restrict__ ain_p_out, int boff, int nblk_g)
{
    __shared__ SegSmem sm;
    __shared__ uint64_t tin[NT];
    __shared__ double es[1];
    const int bidx = (int)blockIdx.x + boff, nblk = nblk_g ? nblk_g : (int)gridDim.x;
    if constexpr (INRES) {
        static_assert(!INRES || (MODE == MODE_P && !FROM_LW), "only the cumsum pass resolves the pass before it");
        __shared__ ResolveSmem rs;
        InResolve<MODE_W> pro; pro.sm = &sm; pro.rs = &rs; pro.brec = prev_brec; pro.side = prev_side; pro.B = nblk; pro.upto = nblk;
        pro.w = w; pro.nw = nw; pro.total = 1.0; pro.st = st; pro.stats = (bidx == 0);
        local_block<MODE, FROM_LW, MAXB, false, InResolve<MODE_W>>(sm, tin, es, bidx, nblk, w, nw, ain, lim, brec, side, st, f, pro, ain_p_out);
    } else {
        (void)prev_brec; (void)prev_side; (void)ain_p_out;
        local_block<MODE, FROM_LW, MAXB>(sm, tin, es, bidx, nblk, w, nw, ain, lim, brec, side, st, f);
    }
}

constexpr int NTR = 1024;        // threads of the single resolve workgroup (16 waves)
constexpr int SIDE_CACHE = 12;   // side entries staged in LDS for the boundary walk

template <int MODE>
__global__ __launch_bounds__(NTR) void k_resolve(const double* __restrict__ w, long long nw, int B,
                                                 const BlockRec* __restrict__ brec, const SideList* __restrict__ side,
                                                 uint64_t* __restrict__ cin,
                                                 const double* __restrict__ ain_w, double* __restrict__ ain_p, DevState* st)
{
    extern __shared__ __attribute__((aligned(16))) char smraw[];
    BlockRec* br = reinterpret_cast<BlockRec*>(smraw);
    __shared__ SegSmem sm;
    __shared__ uint64_t final_state;
    __shared__ __attribute__((aligned(16))) SideEntry sideC[64];
    const int t = threadIdx.x;
    // run-state words are fetched together with the block records (checked after the staging barrier)
    const int s_dead = st->dead, s_do = st->do_resample, dbg0 = st->debug_stop; (void)dbg0;
    const uint32_t s_flags = st->flags;
    const uint64_t s_total = st->total_bits;
    BSSM_STAMP(st, dbg0, MODE, 0, t == 0);
    BSSM_STAMP(st, dbg0, MODE, 1, t == 0);
    // The last block nearly always carries one side entry (the lanes next to cum == 1.0 are never PURE): fetch it
    // now, together with the block records, so the walk does not wait for a dependent global load later.
    if (t >= NTR - (int)(sizeof(SideEntry) / 16)) {
        const int q = t - (NTR - (int)(sizeof(SideEntry) / 16));
        reinterpret_cast<uint4*>(&sideC[63])[q] = reinterpret_cast<const uint4*>(&side[B - 1].e[0])[q];
    }
    {   // stage the block records in LDS (16-byte pieces)
        const uint4* src = reinterpret_cast<const uint4*>(brec);
        uint4* dst = reinterpret_cast<uint4*>(smraw);
        const int n16 = B * (int)(sizeof(BlockRec) / 16);
        for (int i = t; i < n16; i += NTR) dst[i] = src[(size_t)(i & 3) * BREC_STRIDE + (i >> 2)];      // planes -> whole records
    }
    if (t == 0) sm.fail = 0;
    __syncthreads();
    if (s_dead || s_flags || !s_do) return;
    BSSM_STAMP(st, dbg0, MODE, 2, t == 0);
    const double total = (MODE == MODE_P) ? b2d(s_total) : 1.0;
    const int CB = (B + NTR - 1) / NTR;                // blocks per thread (1 or 2)
    const int c0 = t * CB, c1 = (c0 + CB < B) ? c0 + CB : B;
    // chunk record: PURE composite + validity window relative to the first block's base
    bool isb = false;
    Pure comp; comp.d0 = 0; comp.d1 = 0;
    long long wlo = -(1ll << 40), whi = (1ll << 40);
    uint64_t cbase = 0;
    for (int b = c0; b < c1; b++) {
        const BlockRec& r = br[b];
        if (r.prefix.kind != REC_PURE || r.tail_from < NT || r.nside > 0) { isb = true; break; }
        if (b == c0) cbase = r.prefix.base;
        const long long m0 = (long long)(cbase + (uint64_t)comp.d0 - r.prefix.base);
        const long long m1 = (long long)(cbase + 1 + (uint64_t)comp.d1 - r.prefix.base);
        const long long mmin = m0 < m1 ? m0 : m1, mmax = m0 > m1 ? m0 : m1;
        const long long lo = (long long)r.prefix.lo - mmin + 2, hi = (long long)r.prefix.hi - mmax - 2;
        wlo = lo > wlo ? lo : wlo; whi = hi < whi ? hi : whi;
        Pure p; p.d0 = (int64_t)(r.prefix.o[0] - r.prefix.base); p.d1 = (int64_t)(r.prefix.o[1] - (r.prefix.base + 1));
        comp = pure_compose(comp, p);
    }
    if (c0 >= B) { isb = false; comp.d0 = comp.d1 = 0; }    // padding lanes: identity
    BSSM_STAMP(st, dbg0, MODE, 3, t == 0);
    int seg, nb; Pure lastseg;
    const Pure exc = seg_excl_scan<NTR / 64>(sm, comp, isb, seg, nb, lastseg);
    if (isb && seg < MAXBND) { sm.bnd_excl[seg] = exc; sm.bnd_lane[seg] = t; }
    __syncthreads();
    BSSM_STAMP(st, dbg0, MODE, 4, t == 0);
    long long lit = 0;
    // ---- boundary walk.  One wave; lane j owns the j-th boundary block with its record in registers; the
    // exact state is handed from lane to lane by readlane, so each step costs one record evaluation. ----
    const int nent = nb * CB;
#ifdef BSSM_DEV_STAMPS
    if (dbg0 == 99 && t == 0) { st->stamps[MODE][8] = nb; st->stamps[MODE][9] = B; }
#endif
    if (t < 64) {
        const int lane = t;
        const bool fits = (nb <= MAXBND) && (nent <= 64);
        const bool have = fits && lane < nent;
        const int k = have ? lane / CB : 0;
        int bidx = -1;
        Pure ex; ex.d0 = 0; ex.d1 = 0;
        if (have) {
            bidx = sm.bnd_lane[k] * CB + (lane % CB);
            if (bidx >= B) bidx = -1;
            if (lane % CB == 0) ex = sm.bnd_excl[k];
        }
        Rec pr = rec_identity(0);
        int nside = 0;
        bool slow = false;
        if (bidx >= 0) {
            pr = br[bidx].prefix; nside = br[bidx].nside;
            slow = (nside > 1) || (br[bidx].tail_from < NT);
            if (nside == 1 && !slow && fits && !(bidx == B - 1 && lane < 63)) {   // (the last block's entry is already in slot 63)
                const uint4* sp = reinterpret_cast<const uint4*>(&side[bidx].e[0]);
                uint4* dp = reinterpret_cast<uint4*>(&sideC[lane]);
#pragma unroll
                for (int q = 0; q < (int)(sizeof(SideEntry) / 16); q++) dp[q] = sp[q];
            }
        }
        if (!fits) { if (lane == 0) sm.fail = 1; }
        else {
            uint64_t s = 0, my_in = 0, my_out = 0;
            const bool special = (bidx >= 0) && (slow || nside == 1);
            const StepFn fn = stepfn_build(ex, pr);          // "PURE run + this block's record" folded into constants
            const WalkFn wfn = walkfn_from(fn, lane < nent && bidx >= 0 && !special);
            BSSM_STAMP(st, dbg0, MODE, 10, t == 0 && fn.mode >= 0);
            // (MODE_P: nobody needs the state after the last block -- skip its link, usually the slow one)
            const int blast = __builtin_amdgcn_readlane(bidx, nent > 0 ? nent - 1 : 0);
            const int nstep = (MODE == MODE_P && nent > 0 && blast == B - 1) ? nent - 1 : nent;
            uint64_t my_s0 = 0;
            const int j0 = walk_chain_regs(wfn, nstep, s, my_s0, my_out);
            if (lane < j0) my_in = pure_step(ex, my_s0);
#ifdef BSSM_DEV_STAMPS
            if (dbg0 == 99 && t == 0) { st->stamps[MODE][12] = j0; st->stamps[MODE][13] = clock64(); }
#endif
            for (int j = j0; j < nstep; j++) {
                bool ok;
                uint64_t o = stepfn_apply(fn, s, ok);
                const uint64_t e = pure_step(ex, s);         // (only lane j's copy is used)
                if (bidx < 0) { o = e; ok = true; }
                if (lane == j && (special || !ok)) {             // rare: side entry / tails / window miss
                    bool done = false;
                    if (!slow && ok && nside == 1) {
                        const SideEntry& se = sideC[(bidx == B - 1) ? 63 : lane];
                        bool ok2 = true;
                        uint64_t o2 = rec_step(se.leaf, o, ok2);
                        if (!ok2) {                          // HARD leaf: its 8 terms literally (kept in the entry)
                            double c = b2d(o);
#pragma unroll
                            for (int q = 0; q < EL; q++) c = c + se.terms[q];
                            o2 = d2b(c);
                        }
                        done = true;
                        if (se.pad) {
                            const int64_t diff = (int64_t)(o2 - se.post_base);
                            if (diff < (int64_t)se.lo || diff > (int64_t)se.hi) done = false;
                            else { Pure pp; pp.d0 = se.d0; pp.d1 = se.d1; o2 = pure_step(pp, o2); }
                        }
                        if (done) o = o2;
                    }
                    if (!done) o = block_out_exact<MODE>(br[bidx], side, w, nw, total, bidx, e, lit);   // the general (slower) routine
                }
                if (lane == j) { my_in = e; my_out = o; }
                // hand lane j's result to everyone (j is wave-uniform: v_readlane, no LDS round trip)
                const int olo = __builtin_amdgcn_readlane((int)(uint32_t)o, j), ohi = __builtin_amdgcn_readlane((int)(uint32_t)(o >> 32), j);
                s = ((uint64_t)(uint32_t)ohi << 32) | (uint32_t)olo;
            }
            if (nstep < nent && lane == nent - 1) { my_in = pure_step(ex, s); my_out = 0; }
            BSSM_STAMP(st, dbg0, MODE, 11, t == 0 && s != 1);
            if (lane == 0) sm.seg_start[0] = 0;
            if (have) {
                if (bidx >= 0) cin[bidx] = my_in;                // boundary blocks are finished here
                if (lane % CB == CB - 1) sm.seg_start[k + 1] = my_out;
                if (bidx == B - 1) final_state = my_out;
            }
        }
    }
    BSSM_STAMP(st, dbg0, MODE, 5, t == 0);
    __syncthreads();
    BSSM_STAMP(st, dbg0, MODE, 6, t == 0);
    uint64_t ent = 0;
    if (!sm.fail && c0 < B && !isb) {
        ent = pure_step(exc, sm.seg_start[seg]);
        const long long diff = (long long)(ent - cbase);
        if (diff < wlo || diff > whi) sm.fail = 1;
    }
    __syncthreads();
    if (sm.fail) {
        // records did not cover the exact states: one lane walks every block (each step is still exact)
        if (t == 0) {
            uint64_t s = 0;
            for (int b = 0; b < B; b++) { cin[b] = s; s = block_out_exact<MODE>(br[b], side, w, nw, total, b, s, lit); }
            final_state = s;
            atomicAdd((unsigned long long*)&st->stat_serial_walks, 1ull);
        }
    } else if (c0 < B && !isb) {
        uint64_t s = ent;
        for (int b = c0; b < c1; b++) {
            cin[b] = s;
            if (b + 1 < c1 || (c1 == B && MODE == MODE_W)) s = block_out_exact<MODE>(br[b], side, w, nw, total, b, s, lit);
        }
        if (c1 == B) final_state = s;
    }
    if (lit) atomicAdd((unsigned long long*)&st->stat_literal_terms, (unsigned long long)(lit & (LIT_FROM_W - 1)));
    __syncthreads();
    BSSM_STAMP(st, dbg0, MODE, 7, t == 0);
    if (MODE == MODE_W) {
        const double tot = b2d(final_state);
        if (t == 0) {
            st->total_bits = final_state;
            if (tot == 0.0) atomicOr(&st->flags, FLAG_ZERO_SUM);       // src/resampling.cpp:8,22,49
            if (!isfinite(tot)) atomicOr(&st->flags, FLAG_NONFINITE);
        }
        for (int b = t; b < B; b += NTR) ain_p[b] = ain_w[b] / tot;
    }
}

// k_resolve_all: grids of more than 2 NT blocks (N > 2^20): ONE workgroup of NTR threads runs the same block-wide resolve
// as the scan kernels do for themselves (resolve_in_block: records straight into registers, one segmented scan, the links in
// registers) and emits every block's exact incoming state.  Replaces k_resolve there (19 -> see DESIGN.md us at 2048 blocks).
template <int MODE>
__global__ __launch_bounds__(NTR) void k_resolve_all(const double* __restrict__ w, long long nw, int B,
                                                     const BlockRec* __restrict__ brec, const SideList* __restrict__ side,
                                                     uint64_t* __restrict__ cin,
                                                     const double* __restrict__ ain_w, double* __restrict__ ain_p, DevState* st)
{
    __shared__ SegSmem sm;
    __shared__ ResolveSmem rs;
    const int t = threadIdx.x;
    const int s_dead = st->dead, s_do = st->do_resample;
    const uint32_t s_flags = st->flags;
    const double total = (MODE == MODE_P) ? b2d(st->total_bits) : 1.0;
    if (s_dead || s_flags || !s_do) return;
    const uint64_t fin = resolve_in_block<MODE, NTR, true>(sm, rs, brec, side, B, B, w, nw, total, st, true, cin);
    if (MODE == MODE_W) {
        const double tot = b2d(fin);
        if (t == 0) {
            st->total_bits = fin;
            if (tot == 0.0) atomicOr(&st->flags, FLAG_ZERO_SUM);       // src/resampling.cpp:8,22,49
            if (!isfinite(tot)) atomicOr(&st->flags, FLAG_NONFINITE);
        }
        for (int b = t; b < B; b += NTR) ain_p[b] = ain_w[b] / tot;
    }
}

// ---------------------------------------------------------------------------
// k_apply: exact cum_sum -> output counts T -> ancestors (-> gather)
// ---------------------------------------------------------------------------
struct UniformSrc {
    const double* arr;       // parity mode: draws of this call (systematic: arr[0])
    PhiloxKey key;
    uint32_t call;
    __device__ __forceinline__ double operator()(int32_t i) const
    {
        return arr ? arr[i] : resample_uniform(key, call, (uint32_t)i);
    }
};

struct ApplyArgs {
    const double* w; long long nw; const double* ain_p; const uint64_t* cin; int lim;
    int n;                        // number of outputs
    const double* u_base;         // parity draws for ALL calls (or nullptr)
    long long u_stride;           // doubles per call in u_base
    PhiloxKey key;
    int* anc_out;                 // [n] 1-based, or nullptr
    long long anc_stride;         // per-call stride when recording every call's ancestors (0: single)
    double* cum_out;              // [nw] exact cum_sum, or nullptr
    const double* xsrc; double* xdst; int dim; long long xstride;   // gather particles[indices, ]
    const double* auxsrc; double* auxdst;                            // gather aux_log_weights[ancestors]
    double* se_part;              // [B][dim] partial sums of x * (1/N) after resampling, or nullptr
    int nstage;                   // k_apply: LDS staging arrays provided by the launch (0..3 x CAPX doubles)
    int lead, last;               // first / last block of this launch in the global numbering: they record the launch's output range
    // STEP kernels: the NEXT observation's transition_fn + weight_fn (R/particle_filter_core.R:127,177-183) applied to the
    // resampled particles before they leave the workgroup -- what k_step would do in a launch of its own
    int step_model; ModelPar step_par; double step_y; NoiseSrc step_ns; double* step_lw;
};

// Expansion when an element owns more than 64 outputs (degenerate weights: a few particles own almost everything): all
// lanes share the work through a binary search in the counts.  Rare: out of line.
// (arguments and results by value: taking the caller's accumulators or its argument block by reference would push them
// into scratch memory on the hot path as well)
__device__ __attribute__((noinline)) double2 expand_by_search(const int* Tl, int* anc, const double* xsrc, double* xdst, int dim, long long xstride,
                                                              const double* auxsrc, double* auxdst, long long b0, int Tb, int Te, double invN,
                                                              const double* xloc = nullptr /* the block's own particles in LDS, [dim][EB] */)
{
    double acc0 = 0.0, acc1 = 0.0;
    for (int i = Tb + (int)threadIdx.x; i < Te; i += NT) {
        // first local index whose count exceeds i
        int lo = 0, hi = EB - 1;
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if (Tl[mid] > i) hi = mid; else lo = mid + 1;
        }
        const long long src = b0 + lo;
        if (anc) anc[i] = (int)(src + 1);
        if (xdst) {
            const double x0 = xloc ? xloc[lo] : xsrc[src];
            xdst[i] = x0; acc0 += x0 * invN;
            if (dim > 1) { const double x1 = xloc ? xloc[EB + lo] : xsrc[xstride + src]; xdst[xstride + i] = x1; acc1 += x1 * invN; }
        }
        if (auxdst) auxdst[i] = auxsrc[src];
    }
    double2 r; r.x = acc0; r.y = acc1;
    return r;
}

constexpr int CAPX = 3072;       // outputs a block can stage in LDS for the coalesced store (it owns ~EB of them)

// The second half of the expansion kernel's body: from every lane's exact incoming state `ent` (and its EL terms v[] = prob)
// to the exact cum_sum, the output counts, the ancestors and particles[indices, ].  xs0 / xs1 / axs: the particles (and
// auxiliary log-weights) of the lane's EXPANSION elements  wave * 64 * EL + 64 k + lane  (lane-interleaved, see apply_block).
template <int KIND, bool STEP, int CAP, bool LEAN>
__device__ __forceinline__ void apply_tail(SegSmem& sm, int* Tl /* [EB] */, int& Tbegin, const int bidx, const int nblk, const ApplyArgs& a, DevState* st,
                                           double* lx, const int nstage, const double (&v)[EL], const uint64_t ent,
                                           const double (&xs0)[EL], const double (&xs1)[EL], const double (&axs)[EL],
                                           const int call, const UniformSrc& us, const double Usys, const int dbg, const bool stamper,
                                           const double* xloc = nullptr /* the block's own particles in LDS ([dim][EB]) when they are not in a.xsrc */)
{
    const int t = threadIdx.x;
    const bool d2 = !LEAN && (a.dim > 1), aux = !LEAN && (a.auxdst != nullptr);
    const long long b0 = (long long)bidx * EB;
    const int ebase = (t >> 6) * (64 * EL) + (t & 63);
    (void)nblk; (void)dbg; (void)stamper;
    BSSM_STAMP(st, dbg, 3, 3, stamper);
    // the reference chain itself, from the exact incoming state
    int Tk[EL];
    double c = b2d(ent);
#pragma unroll
    for (int k = 0; k < EL; k++) {
        c = c + v[k];
        const long long j = b0 + (long long)t * EL + k;
        if (a.cum_out && j < a.nw) a.cum_out[j] = c;
        int T;
        if (j >= a.nw - 1) T = a.n;                                   // j < size-1 clamp (src/resampling.cpp:33,59)
        else if (KIND == 1) T = count_le_systematic(c, a.n, Usys);
        else if (KIND == 0) T = count_le_stratified(c, a.n, us);
        else T = 0;
        Tl[t * EL + k] = T;
        Tk[k] = T;
    }
    if (KIND == 2) return;                                            // multinomial: k_multinomial searches cum_out
    BSSM_STAMP(st, dbg, 3, 4, stamper);
    // Outputs owned before this lane's first element: T of the lane's own exact incoming state (every lane evaluates it
    // itself -- no LDS hand-off from the neighbouring lane, and the check below is complete before the barrier).
    int tprev;
    {
        const long long jprev = b0 + (long long)t * EL - 1;
        if (jprev < 0) tprev = 0;
        else if (jprev >= a.nw - 1) tprev = a.n;                      // the clamp of src/resampling.cpp:33,59 already applied
        else tprev = (KIND == 1) ? count_le_systematic(b2d(ent), a.n, Usys) : count_le_stratified(b2d(ent), a.n, us);
    }
    int maxcnt = 0;
    {
        int p = tprev;
#pragma unroll
        for (int k = 0; k < EL; k++) { maxcnt = (Tk[k] - p) > maxcnt ? (Tk[k] - p) : maxcnt; p = Tk[k]; }
    }
    if (maxcnt > 64) sm.big = 1;                                      // (cleared at entry; barriers in between)
    if (t == 0) { Tbegin = tprev; if (bidx == a.lead) st->out_lo = tprev; }
    if (t == NT - 1 && bidx == a.last) st->out_hi = Tk[EL - 1];
    __syncthreads();
    BSSM_STAMP(st, dbg, 3, 7, stamper);
    const int Tb = Tbegin, Te = Tl[EB - 1];
    int* anc = a.anc_out ? a.anc_out + (long long)call * a.anc_stride : nullptr;
    const double invN = 1.0 / (double)a.n;
    double acc0 = 0.0, acc1 = 0.0;
    // Expansion.  Each element owns the outputs [T_prev, T) (src/resampling.cpp:30-37).  Common case (no element
    // owns more than 64 outputs): every lane stores its elements' particles straight to the outputs they own --
    // neighbouring lanes own neighbouring output ranges, and nothing has to be gathered.  Otherwise (degenerate
    // weights: a few particles own almost everything): all lanes share the work through a binary search in the counts.
    const int any_big = sm.big;
    bool step_in_place = false; (void)step_in_place;
    BSSM_STAMP(st, dbg, 3, 8, stamper);
    if (!any_big) {
        // Scattered 8-byte stores cost the CU one address per lane (64 cache lines per wave instruction): with the
        // particles going straight to HBM that address traffic, not the bytes, bounded this stage.  Plain case
        // (scalar state, nothing else to carry): scatter into LDS, then store the block's output range coalesced.
        // (the second state component and the auxiliary log-weights are staged the same way when the launch provides room)
        const int need = (d2 ? 2 : 1) + (aux ? 1 : 0);
        const bool stage = (lx != nullptr) && a.xdst && !anc && need <= nstage && (Te - Tb) <= CAP;
        double* lx1 = lx + CAP;                                    // second state component
        double* lxa = lx + (d2 ? 2 : 1) * CAP;                     // auxiliary log-weights
        int Tp[EL], Ts[EL];                                        // all the counts first: one LDS round trip, not EL
#pragma unroll
        for (int k = 0; k < EL; k++) { const int e = ebase + 64 * k; Ts[k] = Tl[e]; Tp[k] = (e == 0) ? Tb : Tl[e - 1]; }
        if (stage) {
            // one loop over "j-th output of every element" with the EL stores predicated, instead of EL short loops
            // (a divergent loop trip costs two scalar branches and a dozen instructions)
            int cmax = 0;
#pragma unroll
            for (int k = 0; k < EL; k++) { Ts[k] -= Tp[k]; Tp[k] -= Tb; cmax = Ts[k] > cmax ? Ts[k] : cmax; }     // counts, staging offsets
            for (int j = 0; __any(j < cmax); j++) {
#pragma unroll
                for (int k = 0; k < EL; k++) {
                    if (j < Ts[k]) {
                        lx[Tp[k] + j] = xs0[k];
                        if (d2) lx1[Tp[k] + j] = xs1[k];
                        if (aux) lxa[Tp[k] + j] = axs[k];
                    }
                }
            }
        } else {
#pragma unroll
            for (int k = 0; k < EL; k++) {
                const long long src = b0 + ebase + 64 * k;
                for (int i = Tp[k]; i < Ts[k]; i++) {
                    if (anc) anc[i] = (int)(src + 1);                    // 1-based (src/resampling.cpp:36,62)
                    if (a.xdst) {
                        a.xdst[i] = xs0[k];
                        if (d2) a.xdst[a.xstride + i] = xs1[k];
                    }
                    if (aux) a.auxdst[i] = axs[k];
                }
            }
        }
        // the state estimate sums the block's outputs in output order, strided over the lanes (every path does)
        BSSM_STAMP(st, dbg, 3, 9, stamper);
        if (a.xdst) {
            __syncthreads();
            BSSM_STAMP(st, dbg, 3, 10, stamper);
            if (stage) {
                // all the LDS reads first, then the stores (a read-store pair per trip waited ~450 cycles a trip)
                constexpr int R = CAP / NT;
                double xv[R];
#pragma unroll
                for (int r = 0; r < R; r++) { const int i = Tb + t + NT * r; xv[r] = (i < Te) ? lx[i - Tb] : 0.0; }
#pragma unroll
                for (int r = 0; r < R; r++) { const int i = Tb + t + NT * r; if (i < Te) { if (!STEP) bulk_store8<BSSM_ST_X>(a.xdst + i, xv[r]); acc0 += xv[r] * invN; } }
                if constexpr (STEP) {
                    // the next observation's transition + weight on the staged particles, a PAIR of outputs per lane (the
                    // generator gives two normals per block, keyed by the pair's index -- exactly k_step's arithmetic);
                    // a pair cut by the range's ends is finished by the neighbouring workgroup
                    constexpr int R2 = (CAP / 2 + 1 + NT - 1) / NT;
                    const int q0 = Tb >> 1;
#pragma unroll
                    for (int r = 0; r < R2; r++) {
                        const int q = q0 + t + NT * r, i0 = 2 * q, i1 = i0 + 1;
                        const bool v0 = (i0 >= Tb && i0 < Te), v1 = (i1 >= Tb && i1 < Te);
                        if (v0 || v1) {
                            double x0 = v0 ? lx[i0 - Tb] : 0.0, x1 = v1 ? lx[i1 - Tb] : 0.0;
                            double z0, z1;
                            if (a.step_ns.arr) { z0 = v0 ? a.step_ns.arr[i0] : 0.0; z1 = v1 ? a.step_ns.arr[i1] : 0.0; }
                            else normal_pair(a.step_ns.key, a.step_ns.purpose, a.step_ns.call, 0, (uint32_t)q, z0, z1);
                            if (a.step_model == 1) { x0 = Model<1>::transition(x0, z0, a.step_par); x1 = Model<1>::transition(x1, z1, a.step_par); }
                            else { x0 = Model<0>::transition(x0, z0, a.step_par); x1 = Model<0>::transition(x1, z1, a.step_par); }
                            const double l0 = r_dnorm_log(a.step_y, x0, a.step_par.sy, a.step_par.log_sy);
                            const double l1 = r_dnorm_log(a.step_y, x1, a.step_par.sy, a.step_par.log_sy);
                            if (v0 && v1) {
                                double2 qx; qx.x = x0; qx.y = x1; *reinterpret_cast<double2*>(a.xdst + i0) = qx;
                                double2 ql; ql.x = l0; ql.y = l1; *reinterpret_cast<double2*>(a.step_lw + i0) = ql;
                            } else if (v0) { a.xdst[i0] = x0; a.step_lw[i0] = l0; }
                            else { a.xdst[i1] = x1; a.step_lw[i1] = l1; }
                        }
                    }
                }
                if (d2) for (int i = Tb + t; i < Te; i += NT) { const double x1 = lx1[i - Tb]; a.xdst[a.xstride + i] = x1; acc1 += x1 * invN; }
                if (aux) for (int i = Tb + t; i < Te; i += NT) a.auxdst[i] = lxa[i - Tb];
            } else {
                for (int i = Tb + t; i < Te; i += NT) {
                    acc0 += a.xdst[i] * invN;
                    if (d2) acc1 += a.xdst[a.xstride + i] * invN;
                }
                if constexpr (STEP) step_in_place = true;
            }
        }
    } else {
        const double2 r = expand_by_search(Tl, anc, a.xsrc, a.xdst, a.dim, a.xstride, a.auxsrc, a.auxdst, b0, Tb, Te, invN, xloc);
        acc0 = r.x; acc1 = r.y;
        if constexpr (STEP) step_in_place = true;
    }
    if constexpr (STEP) {
        if (step_in_place) {
            // (the resampled particles of this workgroup's range went to global memory -- too many outputs to stage, or
            //  degenerate weights: the next observation's transition + weight in place, after everyone has read them)
            __syncthreads();
            for (int q = (Tb >> 1) + t; 2 * q < Te; q += NT) {
                const int i0 = 2 * q, i1 = i0 + 1;
                const bool v0 = (i0 >= Tb && i0 < Te), v1 = (i1 >= Tb && i1 < Te);
                if (v0 || v1) {
                    double x0 = v0 ? a.xdst[i0] : 0.0, x1 = v1 ? a.xdst[i1] : 0.0;
                    double z0, z1;
                    if (a.step_ns.arr) { z0 = v0 ? a.step_ns.arr[i0] : 0.0; z1 = v1 ? a.step_ns.arr[i1] : 0.0; }
                    else normal_pair(a.step_ns.key, a.step_ns.purpose, a.step_ns.call, 0, (uint32_t)q, z0, z1);
                    if (a.step_model == 1) { x0 = Model<1>::transition(x0, z0, a.step_par); x1 = Model<1>::transition(x1, z1, a.step_par); }
                    else { x0 = Model<0>::transition(x0, z0, a.step_par); x1 = Model<0>::transition(x1, z1, a.step_par); }
                    if (v0) { a.xdst[i0] = x0; a.step_lw[i0] = r_dnorm_log(a.step_y, x0, a.step_par.sy, a.step_par.log_sy); }
                    if (v1) { a.xdst[i1] = x1; a.step_lw[i1] = r_dnorm_log(a.step_y, x1, a.step_par.sy, a.step_par.log_sy); }
                }
            }
        }
    }
    BSSM_STAMP(st, dbg, 3, 5, stamper);
    if (a.se_part) {
        acc0 = block_sum(acc0, sm.sh4);
        if (d2) acc1 = block_sum(acc1, sm.sh4);
        if (t == 0) { a.se_part[(long long)bidx * a.dim] = acc0; if (d2) a.se_part[(long long)bidx * a.dim + 1] = acc1; }
    }
    BSSM_STAMP(st, dbg, 3, 6, stamper);
}


// (The body below and apply_tail above are the same expansion: the multi-launch kernels keep it in ONE function -- split into head + tail
//  the lean expansion kernel ran 39.6 instead of 35.1 us at N = 2^22 (same source otherwise, profiles/r03_h_*) -- and the fused kernel, which
//  arrives with the exact states already in registers, calls apply_tail.)
// LEAN (scalar state, no auxiliary log-weights to carry; grids that run several rounds of workgroups per CU): the registers of the
// second state component and of the auxiliary values are not allocated, and CAP < CAPX outputs are staged -- together four
// workgroups fit a CU instead of three (112 VGPRs, 40.7 KiB LDS).
template <int KIND, bool LIT = false, class Pro = NoResolve, bool STEP = false, int CAP = CAPX, bool LEAN = false>
__device__ __forceinline__ void apply_block(SegSmem& sm, uint64_t* tin /* [NT + 1] */, int* Tl /* [EB] */, int& Tbegin, const int bidx, const int nblk,
                                            const ApplyArgs& a, DevState* st, double* lx = nullptr /* LDS [nstage][CAPX] or nullptr */, const int nstage = 0,
                                            const Pro pro = Pro())
{
    const int t = threadIdx.x;
    const bool d2 = !LEAN && (a.dim > 1), aux = !LEAN && (a.auxdst != nullptr);
    const long long b0 = (long long)bidx * EB;
    // run-state words and this lane's terms are fetched together (one memory round trip, not two)
    const int s_dead = st->dead, s_do = st->do_resample, call = st->cur_call, dbg = st->debug_stop;
    const uint32_t s_flags = st->flags;
    const double total = b2d(st->total_bits);
    double v[EL];
    load_terms<MODE_W>(a.w, a.nw, 1.0, b0 + (long long)t * EL, v);
    // The particles travel with the elements that own them: the common expansion path stores them straight to the
    // outputs their elements own, with no dependent gather.  For the expansion a lane holds the elements
    // wave * 64 * EL + 64 k + lane (k = 0..EL-1) -- lane-interleaved, NOT the EL consecutive elements its terms cover:
    // neighbouring lanes then write neighbouring outputs.  (With consecutive elements per lane the lanes' output
    // positions were ~EL apart and the 8-byte LDS stores ran into 8-way bank conflicts: that stage took 9k of the kernel's
    // 19k cycles, profiles/r02_a_stage_stamps_typical_block.txt.)
    const int ebase = (t >> 6) * (64 * EL) + (t & 63);       // element (within the block) of k = 0
    double xs0[EL], xs1[EL], axs[EL];
#pragma unroll
    for (int k = 0; k < EL; k++) { xs0[k] = 0.0; xs1[k] = 0.0; axs[k] = 0.0; }
    if (KIND != 2) {
#pragma unroll
        for (int k = 0; k < EL; k++) {
            const long long j = b0 + ebase + 64 * k;
            if (j < a.nw) {
                if (a.xdst) { xs0[k] = a.xsrc[j]; if (d2) xs1[k] = a.xsrc[a.xstride + j]; }
                if (aux) axs[k] = a.auxsrc[j];
            }
        }
    }
    const double a_in_p = a.ain_p[bidx];
    const uint64_t cin_ld = Pro::active ? 0ull : a.cin[bidx];
    if (s_dead || !s_do || s_flags) return;
    // (INRES: every workgroup resolves the MODE_P pass for itself -- its own exact incoming state -- while its weight and
    //  particle loads are in flight; the resolve's scratch borrows the staging area, which is not in use yet)
    const uint64_t cinb = Pro::active ? pro() : cin_ld;
    if (Pro::active) __syncthreads();
    if (t == 0) sm.big = 0;
    // the uniform(s) of this resample call (systematic: one draw; computed here, under the load latency)
    UniformSrc us;
    us.arr = a.u_base ? a.u_base + (long long)call * a.u_stride : nullptr;
    us.key = a.key; us.call = (uint32_t)call;
    const double Usys = (KIND == 1) ? us(0) : 0.0;
    const bool stamper = (t == 0 && bidx == ((nblk > 100 && dbg != 98) ? 100 : 0)); (void)stamper;
    BSSM_STAMP(st, dbg, 3, 0, stamper);
#pragma unroll
    for (int k = 0; k < EL; k++) v[k] = v[k] / total;                  // prob = weights / total (src/resampling.cpp:24,51)
    BSSM_STAMP(st, dbg, 3, 1, stamper && v[0] >= 0.0);
    uint64_t ent;
    if (LIT) {
        // one block of few terms from exact +0: publish prob (zero beyond nw) in the not yet written destination buffer and add in order
        double2* p2 = reinterpret_cast<double2*>(a.xdst + (long long)t * EL);
#pragma unroll
        for (int k = 0; k < EL / 2; k++) { double2 q2; q2.x = v[2 * k]; q2.y = v[2 * k + 1]; p2[k] = q2; }
        __syncthreads();
        block_literal_terms(tin, a.xdst, (int)a.nw);
        ent = tin[t];
    } else {
        BlockScan bs;
        block_scan<MODE_P>(sm, v, a_in_p, a.lim, bs);
        __syncthreads();
        BSSM_STAMP(st, dbg, 3, 2, stamper);
        const bool good = block_resolve<MODE_P>(sm, bs, cinb, a.lim, a.w, a.nw, total, b0, ent);
        if (!good) { block_literal<MODE_P>(tin, cinb, a.w, a.nw, total, b0, st); ent = tin[t]; }
    }
    BSSM_STAMP(st, dbg, 3, 3, stamper);
    // the reference chain itself, from the exact incoming state
    int Tk[EL];
    double c = b2d(ent);
#pragma unroll
    for (int k = 0; k < EL; k++) {
        c = c + v[k];
        const long long j = b0 + (long long)t * EL + k;
        if (a.cum_out && j < a.nw) a.cum_out[j] = c;
        int T;
        if (j >= a.nw - 1) T = a.n;                                   // j < size-1 clamp (src/resampling.cpp:33,59)
        else if (KIND == 1) T = count_le_systematic(c, a.n, Usys);
        else if (KIND == 0) T = count_le_stratified(c, a.n, us);
        else T = 0;
        Tl[t * EL + k] = T;
        Tk[k] = T;
    }
    if (KIND == 2) return;                                            // multinomial: k_multinomial searches cum_out
    BSSM_STAMP(st, dbg, 3, 4, stamper);
    // Outputs owned before this lane's first element: T of the lane's own exact incoming state (every lane evaluates it
    // itself -- no LDS hand-off from the neighbouring lane, and the check below is complete before the barrier).
    int tprev;
    {
        const long long jprev = b0 + (long long)t * EL - 1;
        if (jprev < 0) tprev = 0;
        else if (jprev >= a.nw - 1) tprev = a.n;                      // the clamp of src/resampling.cpp:33,59 already applied
        else tprev = (KIND == 1) ? count_le_systematic(b2d(ent), a.n, Usys) : count_le_stratified(b2d(ent), a.n, us);
    }
    int maxcnt = 0;
    {
        int p = tprev;
#pragma unroll
        for (int k = 0; k < EL; k++) { maxcnt = (Tk[k] - p) > maxcnt ? (Tk[k] - p) : maxcnt; p = Tk[k]; }
    }
    if (maxcnt > 64) sm.big = 1;                                      // (cleared at entry; barriers in between)
    if (t == 0) { Tbegin = tprev; if (bidx == a.lead) st->out_lo = tprev; }
    if (t == NT - 1 && bidx == a.last) st->out_hi = Tk[EL - 1];
    __syncthreads();
    BSSM_STAMP(st, dbg, 3, 7, stamper);
    const int Tb = Tbegin, Te = Tl[EB - 1];
    int* anc = a.anc_out ? a.anc_out + (long long)call * a.anc_stride : nullptr;
    const double invN = 1.0 / (double)a.n;
    double acc0 = 0.0, acc1 = 0.0;
    // Expansion.  Each element owns the outputs [T_prev, T) (src/resampling.cpp:30-37).  Common case (no element
    // owns more than 64 outputs): every lane stores its elements' particles straight to the outputs they own --
    // neighbouring lanes own neighbouring output ranges, and nothing has to be gathered.  Otherwise (degenerate
    // weights: a few particles own almost everything): all lanes share the work through a binary search in the counts.
    const int any_big = sm.big;
    bool step_in_place = false; (void)step_in_place;
    BSSM_STAMP(st, dbg, 3, 8, stamper);
    if (!any_big) {
        // Scattered 8-byte stores cost the CU one address per lane (64 cache lines per wave instruction): with the
        // particles going straight to HBM that address traffic, not the bytes, bounded this stage.  Plain case
        // (scalar state, nothing else to carry): scatter into LDS, then store the block's output range coalesced.
        // (the second state component and the auxiliary log-weights are staged the same way when the launch provides room)
        const int need = (d2 ? 2 : 1) + (aux ? 1 : 0);
        const bool stage = (lx != nullptr) && a.xdst && !anc && need <= nstage && (Te - Tb) <= CAP;
        double* lx1 = lx + CAP;                                    // second state component
        double* lxa = lx + (d2 ? 2 : 1) * CAP;                     // auxiliary log-weights
        int Tp[EL], Ts[EL];                                        // all the counts first: one LDS round trip, not EL
#pragma unroll
        for (int k = 0; k < EL; k++) { const int e = ebase + 64 * k; Ts[k] = Tl[e]; Tp[k] = (e == 0) ? Tb : Tl[e - 1]; }
        if (stage) {
            // one loop over "j-th output of every element" with the EL stores predicated, instead of EL short loops
            // (a divergent loop trip costs two scalar branches and a dozen instructions)
            int cmax = 0;
#pragma unroll
            for (int k = 0; k < EL; k++) { Ts[k] -= Tp[k]; Tp[k] -= Tb; cmax = Ts[k] > cmax ? Ts[k] : cmax; }     // counts, staging offsets
            for (int j = 0; __any(j < cmax); j++) {
#pragma unroll
                for (int k = 0; k < EL; k++) {
                    if (j < Ts[k]) {
                        lx[Tp[k] + j] = xs0[k];
                        if (d2) lx1[Tp[k] + j] = xs1[k];
                        if (aux) lxa[Tp[k] + j] = axs[k];
                    }
                }
            }
        } else {
#pragma unroll
            for (int k = 0; k < EL; k++) {
                const long long src = b0 + ebase + 64 * k;
                for (int i = Tp[k]; i < Ts[k]; i++) {
                    if (anc) anc[i] = (int)(src + 1);                    // 1-based (src/resampling.cpp:36,62)
                    if (a.xdst) {
                        a.xdst[i] = xs0[k];
                        if (d2) a.xdst[a.xstride + i] = xs1[k];
                    }
                    if (aux) a.auxdst[i] = axs[k];
                }
            }
        }
        // the state estimate sums the block's outputs in output order, strided over the lanes (every path does)
        BSSM_STAMP(st, dbg, 3, 9, stamper);
        if (a.xdst) {
            __syncthreads();
            BSSM_STAMP(st, dbg, 3, 10, stamper);
            if (stage) {
                // all the LDS reads first, then the stores (a read-store pair per trip waited ~450 cycles a trip)
                constexpr int R = CAP / NT;
                double xv[R];
#pragma unroll
                for (int r = 0; r < R; r++) { const int i = Tb + t + NT * r; xv[r] = (i < Te) ? lx[i - Tb] : 0.0; }
#pragma unroll
                for (int r = 0; r < R; r++) { const int i = Tb + t + NT * r; if (i < Te) { if (!STEP) bulk_store8<BSSM_ST_X>(a.xdst + i, xv[r]); acc0 += xv[r] * invN; } }
                if constexpr (STEP) {
                    // the next observation's transition + weight on the staged particles, a PAIR of outputs per lane (the
                    // generator gives two normals per block, keyed by the pair's index -- exactly k_step's arithmetic);
                    // a pair cut by the range's ends is finished by the neighbouring workgroup
                    constexpr int R2 = (CAP / 2 + 1 + NT - 1) / NT;
                    const int q0 = Tb >> 1;
#pragma unroll
                    for (int r = 0; r < R2; r++) {
                        const int q = q0 + t + NT * r, i0 = 2 * q, i1 = i0 + 1;
                        const bool v0 = (i0 >= Tb && i0 < Te), v1 = (i1 >= Tb && i1 < Te);
                        if (v0 || v1) {
                            double x0 = v0 ? lx[i0 - Tb] : 0.0, x1 = v1 ? lx[i1 - Tb] : 0.0;
                            double z0, z1;
                            if (a.step_ns.arr) { z0 = v0 ? a.step_ns.arr[i0] : 0.0; z1 = v1 ? a.step_ns.arr[i1] : 0.0; }
                            else normal_pair(a.step_ns.key, a.step_ns.purpose, a.step_ns.call, 0, (uint32_t)q, z0, z1);
                            if (a.step_model == 1) { x0 = Model<1>::transition(x0, z0, a.step_par); x1 = Model<1>::transition(x1, z1, a.step_par); }
                            else { x0 = Model<0>::transition(x0, z0, a.step_par); x1 = Model<0>::transition(x1, z1, a.step_par); }
                            const double l0 = r_dnorm_log(a.step_y, x0, a.step_par.sy, a.step_par.log_sy);
                            const double l1 = r_dnorm_log(a.step_y, x1, a.step_par.sy, a.step_par.log_sy);
                            if (v0 && v1) {
                                double2 qx; qx.x = x0; qx.y = x1; *reinterpret_cast<double2*>(a.xdst + i0) = qx;
                                double2 ql; ql.x = l0; ql.y = l1; *reinterpret_cast<double2*>(a.step_lw + i0) = ql;
                            } else if (v0) { a.xdst[i0] = x0; a.step_lw[i0] = l0; }
                            else { a.xdst[i1] = x1; a.step_lw[i1] = l1; }
                        }
                    }
                }
                if (d2) for (int i = Tb + t; i < Te; i += NT) { const double x1 = lx1[i - Tb]; a.xdst[a.xstride + i] = x1; acc1 += x1 * invN; }
                if (aux) for (int i = Tb + t; i < Te; i += NT) a.auxdst[i] = lxa[i - Tb];
            } else {
                for (int i = Tb + t; i < Te; i += NT) {
                    acc0 += a.xdst[i] * invN;
                    if (d2) acc1 += a.xdst[a.xstride + i] * invN;
                }
                if constexpr (STEP) step_in_place = true;
            }
        }
    } else {
        const double2 r = expand_by_search(Tl, anc, a.xsrc, a.xdst, a.dim, a.xstride, a.auxsrc, a.auxdst, b0, Tb, Te, invN);
        acc0 = r.x; acc1 = r.y;
        if constexpr (STEP) step_in_place = true;
    }
    if constexpr (STEP) {
        if (step_in_place) {
            // (the resampled particles of this workgroup's range went to global memory -- too many outputs to stage, or
            //  degenerate weights: the next observation's transition + weight in place, after everyone has read them)
            __syncthreads();
            for (int q = (Tb >> 1) + t; 2 * q < Te; q += NT) {
                const int i0 = 2 * q, i1 = i0 + 1;
                const bool v0 = (i0 >= Tb && i0 < Te), v1 = (i1 >= Tb && i1 < Te);
                if (v0 || v1) {
                    double x0 = v0 ? a.xdst[i0] : 0.0, x1 = v1 ? a.xdst[i1] : 0.0;
                    double z0, z1;
                    if (a.step_ns.arr) { z0 = v0 ? a.step_ns.arr[i0] : 0.0; z1 = v1 ? a.step_ns.arr[i1] : 0.0; }
                    else normal_pair(a.step_ns.key, a.step_ns.purpose, a.step_ns.call, 0, (uint32_t)q, z0, z1);
                    if (a.step_model == 1) { x0 = Model<1>::transition(x0, z0, a.step_par); x1 = Model<1>::transition(x1, z1, a.step_par); }
                    else { x0 = Model<0>::transition(x0, z0, a.step_par); x1 = Model<0>::transition(x1, z1, a.step_par); }
                    if (v0) { a.xdst[i0] = x0; a.step_lw[i0] = r_dnorm_log(a.step_y, x0, a.step_par.sy, a.step_par.log_sy); }
                    if (v1) { a.xdst[i1] = x1; a.step_lw[i1] = r_dnorm_log(a.step_y, x1, a.step_par.sy, a.step_par.log_sy); }
                }
            }
        }
    }
    BSSM_STAMP(st, dbg, 3, 5, stamper);
    if (a.se_part) {
        acc0 = block_sum(acc0, sm.sh4);
        if (d2) acc1 = block_sum(acc1, sm.sh4);
        if (t == 0) { a.se_part[(long long)bidx * a.dim] = acc0; if (d2) a.se_part[(long long)bidx * a.dim + 1] = acc1; }
    }
    BSSM_STAMP(st, dbg, 3, 6, stamper);
}


// prev_brec != nullptr: every workgroup resolves the MODE_P pass for itself (its own exact incoming state) instead of a
// k_resolve<P> launch in between.
// STEP: the next observation's transition_fn + weight_fn run on the resampled particles before they leave the workgroup.
constexpr int CAP_LEAN = 2304;   // 18 KiB of staging: with the 22.3 KiB of static LDS four workgroups fit a CU
template <int KIND, bool INRES = false, bool STEP = false, bool LEAN = false>
__global__ __launch_bounds__(NT) void k_apply(ApplyArgs a, DevState* st, const BlockRec* __restrict__ prev_brec,
                                              const SideList* __restrict__ prev_side, int boff, int nblk_g)
{
    static_assert(!LEAN || (!INRES && !STEP), "the lean variant is the plain expansion");
    constexpr int CAP = LEAN ? CAP_LEAN : CAPX;
    const int bidx = (int)blockIdx.x + boff, nblk = nblk_g ? nblk_g : (int)gridDim.x;
    __shared__ SegSmem sm;
    __shared__ uint64_t tin[NT];
    __shared__ int Tl[EB];
    __shared__ int Tbegin;
    extern __shared__ __attribute__((aligned(16))) double lx[];      // [a.nstage][CAPX], sized by the launch
    if constexpr (INRES) {
        InResolve<MODE_P> pro; pro.sm = &sm; pro.rs = reinterpret_cast<ResolveSmem*>(lx); pro.brec = prev_brec; pro.side = prev_side;
        pro.B = nblk; pro.upto = bidx; pro.w = a.w; pro.nw = a.nw; pro.total = b2d(st->total_bits); pro.st = st; pro.stats = (bidx == nblk - 1);
        apply_block<KIND, false, InResolve<MODE_P>, STEP, CAP, LEAN>(sm, tin, Tl, Tbegin, bidx, nblk, a, st, a.nstage ? lx : nullptr, a.nstage, pro);
    } else {
        (void)prev_brec; (void)prev_side;
        apply_block<KIND, false, NoResolve, STEP, CAP, LEAN>(sm, tin, Tl, Tbegin, bidx, nblk, a, st, a.nstage ? lx : nullptr, a.nstage);
    }
}

// multinomial: inverse CDF on the exact cum_sum (distributional parity only)
__device__ __forceinline__ void multinomial_block(double* sh4, const int bidx, const double* __restrict__ cum, long long nw, int n,
                                                  const ApplyArgs& a, DevState* st)
{
    if (st->dead || !st->do_resample || st->flags) return;
    const int call = st->cur_call;
    UniformSrc us;
    us.arr = a.u_base ? a.u_base + (long long)call * a.u_stride : nullptr;
    us.key = a.key; us.call = (uint32_t)call;
    int* anc = a.anc_out ? a.anc_out + (long long)call * a.anc_stride : nullptr;
    const double invN = 1.0 / (double)n;
    double acc0 = 0.0, acc1 = 0.0;
#pragma unroll
    for (int r = 0; r < EL; r++) {
        const long long i = (long long)bidx * EB + threadIdx.x + NT * r;
        if (i < n) {
            const double u = us((int32_t)i);
            long long lo = 0, hi = nw - 1;
            while (lo < hi) {
                const long long mid = (lo + hi) >> 1;
                if (cum[mid] < u) lo = mid + 1; else hi = mid;
            }
            if (anc) anc[i] = (int)(lo + 1);
            if (a.xdst) {
                const double x0 = a.xsrc[lo];
                a.xdst[i] = x0; acc0 += x0 * invN;
                if (a.dim > 1) { const double x1 = a.xsrc[a.xstride + lo]; a.xdst[a.xstride + i] = x1; acc1 += x1 * invN; }
            }
            if (a.auxdst) a.auxdst[i] = a.auxsrc[lo];
        }
    }
    if (a.se_part) {
        acc0 = block_sum(acc0, sh4);
        if (a.dim > 1) acc1 = block_sum(acc1, sh4);
        if (threadIdx.x == 0) { a.se_part[(long long)bidx * a.dim] = acc0; if (a.dim > 1) a.se_part[(long long)bidx * a.dim + 1] = acc1; }
    }
}

// ---------------------------------------------------------------------------
// k_multinomial_r: resample_multinomial_cpp in R's own stream (parity mode).  src/resampling.cpp:5-13 draws through
// Rcpp::sample(n, n, true, prob) -- third-party (Rcpp sugar sample.h, mirroring R's do_sample): Normalize (p /= sum(p)),
// then Walker's alias method when more than 200 categories have n p > 0.1, sorted inversion (Rf_revsort + cumsum)
// otherwise.  Both set-ups are sequential algorithms whose exact order of operations decides the result, so ONE lane runs
// them as published; the n draws (one unif_rand() each, an input here) are then taken by all lanes.  One workgroup: this
// is the parity path (R-seeded shims and filters), not the throughput path (k_multinomial, inverse CDF).
// ---------------------------------------------------------------------------
__device__ __attribute__((noinline)) void revsort_dev(double* a, int* ib, int n)
{   // R's revsort (src/main/sort.c): heapsort into decreasing order, ib[] alongside
    int l, j, ir, i, ii;
    double ra;
    if (n <= 1) return;
    a--; ib--;
    l = (n >> 1) + 1;
    ir = n;
    for (;;) {
        if (l > 1) { l = l - 1; ra = a[l]; ii = ib[l]; }
        else {
            ra = a[ir]; ii = ib[ir];
            a[ir] = a[1]; ib[ir] = ib[1];
            if (--ir == 1) { a[1] = ra; ib[1] = ii; return; }
        }
        i = l; j = l << 1;
        while (j <= ir) {
            if (j < ir && a[j] > a[j + 1]) ++j;
            if (ra > a[j]) { a[i] = a[j]; ib[i] = ib[j]; j += (i = j); }
            else j = ir + 1;
        }
        a[i] = ra; ib[i] = ii;
    }
}

constexpr int NTM = 1024;
__global__ __launch_bounds__(NTM) void k_multinomial_r(const double* __restrict__ w, int n, const double* __restrict__ u_base,
                                                       long long u_stride, int* __restrict__ anc_base, long long anc_stride,
                                                       double* __restrict__ q /* [n] */, int* __restrict__ a /* [n] */,
                                                       int* __restrict__ HL /* [n] */, DevState* st)
{
    __shared__ int walker;
    if (st->dead || !st->do_resample || st->flags) return;
    const int call = st->cur_call;
    const double* U = u_base + (long long)call * u_stride;
    int* ans = anc_base + (long long)call * anc_stride;
    if (threadIdx.x == 0) {
        double total = 0.0;                                            // sum(weights)                    src/resampling.cpp:7
        for (int i = 0; i < n; i++) total += w[i];
        if (total == 0.0) { atomicOr(&st->flags, FLAG_ZERO_SUM); walker = -1; }
        else if (!isfinite(total)) { atomicOr(&st->flags, FLAG_NONFINITE); walker = -1; }
        else {
            double sum = 0.0;
            for (int i = 0; i < n; i++) { const double p = w[i] / total; q[i] = p; sum += p; }      // prob = weights / total (:10); Normalize(): sum
            int nc = 0;
            for (int i = 0; i < n; i++) { const double p = q[i] / sum; q[i] = p; nc += ((double)n * p > 0.1); }
            walker = nc > 200;
            if (nc > 200) {                                            // WalkerSample set-up
                int H = -1, L = n;
                for (int i = 0; i < n; i++) { a[i] = 0; q[i] = q[i] * n; if (q[i] < 1.0) HL[++H] = i; else HL[--L] = i; }
                if (H >= 0 && L < n) {
                    for (int k = 0; k < n - 1; k++) {
                        const int i = HL[k], j = HL[L];
                        a[i] = j;
                        q[j] += q[i] - 1;
                        L += (q[j] < 1.0);
                        if (L >= n) break;
                    }
                }
                for (int i = 0; i < n; i++) q[i] += i;
            } else {                                                   // SampleReplace set-up
                for (int i = 0; i < n; i++) a[i] = i + 1;
                revsort_dev(q, a, n);
                for (int i = 1; i < n; i++) q[i] += q[i - 1];
            }
        }
    }
    __syncthreads();
    if (walker < 0) return;
    // (the set-up lane's plain stores are visible to its own workgroup after the barrier: same CU, write-through L1)
    for (int i = threadIdx.x; i < n; i += NTM) {
        if (walker) {
            const double rU = U[i] * n;
            const int k = (int)rU;
            ans[i] = (rU < q[k]) ? k + 1 : a[k] + 1;
        } else {
            const double rU = U[i];
            int j = 0;
            for (; j < n - 1; j++) if (rU <= q[j]) break;
            ans[i] = a[j];
        }
    }
}

// particles[indices, ] for indices already in HBM (1-based), with the state-estimate partials (R/resampling.R:20,40,60)
__global__ __launch_bounds__(NT) void k_gather_anc(const int* __restrict__ anc_base, long long anc_stride, int n, const double* __restrict__ xsrc,
                                                   double* __restrict__ xdst, int dim, long long xstride, const double* __restrict__ auxsrc,
                                                   double* __restrict__ auxdst, double* __restrict__ se_part, DevState* st)
{
    __shared__ double sh4[NWV];
    if (st->dead || !st->do_resample || st->flags) return;
    const int* anc = anc_base + (long long)st->cur_call * anc_stride;
    const double invN = 1.0 / (double)n;
    double acc0 = 0.0, acc1 = 0.0;
#pragma unroll
    for (int r = 0; r < EL; r++) {
        const long long i = (long long)blockIdx.x * EB + threadIdx.x + NT * r;
        if (i < n) {
            const long long src = anc[i] - 1;
            if (xdst) {
                const double x0 = xsrc[src];
                xdst[i] = x0; acc0 += x0 * invN;
                if (dim > 1) { const double x1 = xsrc[xstride + src]; xdst[xstride + i] = x1; acc1 += x1 * invN; }
            }
            if (auxdst) auxdst[i] = auxsrc[src];
        }
    }
    if (se_part) {
        acc0 = block_sum(acc0, sh4);
        if (dim > 1) acc1 = block_sum(acc1, sh4);
        if (threadIdx.x == 0) { se_part[(long long)blockIdx.x * dim] = acc0; if (dim > 1) se_part[(long long)blockIdx.x * dim + 1] = acc1; }
    }
}

__global__ __launch_bounds__(NT) void k_multinomial(const double* __restrict__ cum, long long nw, int n, ApplyArgs a, DevState* st)
{
    __shared__ double sh4[NWV];
    multinomial_block(sh4, (int)blockIdx.x, cum, nw, n, a, st);
}

// resample_move_filter's move step (R/particle_filter_core.R:226-234) for the built-in random-walk Metropolis move
// of the reference's own example (R/resample_move_filter.R:166-176, tests/testthat/test-resample_move_filter.R:26-35):
//   proposal = particle + rnorm(1, 0, sd);  accept if log(runif(1)) < loglik(proposal) - loglik(particle).
// Also the state-estimate partials, which the core takes AFTER the move (:237-241).
template <int MODEL>
__device__ __forceinline__ void move_block(double* sh4, const int bidx, double* __restrict__ x, long long N, const ModelPar& par, double y,
                                           double move_sd, const double* __restrict__ zmv, const double* __restrict__ umv,
                                           const PhiloxKey& key, uint32_t call, double* __restrict__ se_part,
                                           const DevState* __restrict__ st)
{
    if (st->dead) return;
    const long long base = (long long)bidx * EB;
    const double invN = 1.0 / (double)N;
    double acc = 0.0;
#pragma unroll
    for (int r = 0; r < EL; r++) {
        const long long j = base + threadIdx.x + NT * r;
        if (j < N) {
            double z, u;
            if (zmv) { z = zmv[j]; u = umv[j]; } else move_draws(key, call, (uint32_t)j, z, u);
            const double cur = x[j];
            const double prop = cur + r_rnorm(0.0, move_sd, z);
            const double lp_cur = r_dnorm_log(y, cur, par.sy, par.log_sy);
            const double lp_prop = r_dnorm_log(y, prop, par.sy, par.log_sy);
            const double out = (log(u) < (lp_prop - lp_cur)) ? prop : cur;
            x[j] = out;
            acc += out * invN;
        }
    }
    acc = block_sum(acc, sh4);
    if (threadIdx.x == 0) se_part[bidx] = acc;
}

template <int MODEL>
__global__ __launch_bounds__(NT) void k_move(double* __restrict__ x, long long N, ModelPar par, double y, double move_sd,
                                             const double* __restrict__ zmv, const double* __restrict__ umv,
                                             PhiloxKey key, uint32_t call, double* __restrict__ se_part,
                                             const DevState* __restrict__ st)
{
    __shared__ double sh4[NWV];
    move_block<MODEL>(sh4, (int)blockIdx.x, x, N, par, y, move_sd, zmv, umv, key, call, se_part, st);
}

__global__ void k_dump_move(PhiloxKey key, uint32_t call, long long n, double* __restrict__ zout, double* __restrict__ uout)
{
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double z, u;
    move_draws(key, call, (uint32_t)i, z, u);
    zout[i] = z; uout[i] = u;
}

// no resampling at this observation: particles carry over; state estimate is
// sum(particles * weights) with the normalised weights (:238)
__device__ __forceinline__ void carry_block(double* sh4, const int bidx, const double* __restrict__ xsrc, double* __restrict__ xdst,
                                            const double* __restrict__ w, long long N, int dim,
                                            double* __restrict__ se_part, const DevState* __restrict__ st)
{
    if (st->dead || st->do_resample) return;
    const long long base = (long long)bidx * EB;
    double acc0 = 0.0, acc1 = 0.0;
#pragma unroll
    for (int r = 0; r < EL; r++) {
        const long long j = base + threadIdx.x + NT * r;
        if (j < N) {
            const double wj = w[j];
            const double x0 = xsrc[j];
            xdst[j] = x0; acc0 += x0 * wj;
            if (dim > 1) { const double x1 = xsrc[N + j]; xdst[N + j] = x1; acc1 += x1 * wj; }
        }
    }
    acc0 = block_sum(acc0, sh4);
    if (dim > 1) acc1 = block_sum(acc1, sh4);
    if (threadIdx.x == 0) { se_part[(long long)bidx * dim] = acc0; if (dim > 1) se_part[(long long)bidx * dim + 1] = acc1; }
}

__global__ __launch_bounds__(NT) void k_carry(const double* __restrict__ xsrc, double* __restrict__ xdst,
                                              const double* __restrict__ w, long long N, int dim,
                                              double* __restrict__ se_part, const DevState* __restrict__ st, int boff)
{
    __shared__ double sh4[NWV];
    carry_block(sh4, (int)blockIdx.x + boff, xsrc, xdst, w, N, dim, se_part, st);
}

// state_est[i] = sum over blocks of the partials written for observation i
// (a partial slot is only written by the kernel that ran for that observation)
__global__ __launch_bounds__(NT) void k_reduce_state_est(const double* __restrict__ se_part, int nblocks, int dim,
                                                         double* __restrict__ state_est)
{
    __shared__ double sh4[NWV];
    const long long row = blockIdx.x;
    for (int d = 0; d < dim; d++) {
        double s = 0.0;
        for (int b = threadIdx.x; b < nblocks; b += NT) s += se_part[(row * nblocks + b) * dim + d];
        s = block_sum(s, sh4);
        if (threadIdx.x == 0) state_est[row * dim + d] = s;
    }
}

// weights after the resample decision, for weights_history (:222,:244)
__global__ void k_record_history(const double* __restrict__ x, const double* __restrict__ w, long long N, int dim,
                                 double* __restrict__ ph_row, double* __restrict__ wh_row, const DevState* __restrict__ st)
{
    if (st->dead) return;
    const long long j = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= N) return;
    if (wh_row) wh_row[j] = st->do_resample ? 1.0 / (double)N : w[j];
    if (ph_row) for (int d = 0; d < dim; d++) ph_row[(long long)d * N + j] = x[(long long)d * N + j];
}

// generator dumps (parity tests feed these draws to the CPU oracle)
__global__ void k_dump_normals(PhiloxKey key, uint32_t purpose, uint32_t call, long long n, double* __restrict__ out)
{
    const long long pair = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long j = 2 * pair;
    if (j >= n) return;
    double z0, z1;
    normal_pair(key, purpose, call, 0, (uint32_t)pair, z0, z1);
    out[j] = z0;
    if (j + 1 < n) out[j + 1] = z1;
}
__global__ void k_dump_uniforms(PhiloxKey key, uint32_t call, long long n, double* __restrict__ out)
{
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = resample_uniform(key, call, (uint32_t)i);
}

// ---------------------------------------------------------------------------
// k_pf_batch: many small bootstrap filters per launch (the pilot's 100 filter runs at N = 100, PMMH
// chains at the reference's native N <= 1000).  ONE workgroup runs ONE whole filter: all T observations inside the
// kernel, particles / log-weights / weights in LDS, no launches and no HBM traffic inside the loop.
// The arithmetic is the multi-launch path's, call for call: the same *_block bodies with nblk == 1, and the 1024-thread
// k_step reductions re-enacted by 256 threads in the same association order -- so a batched filter returns bit-for-bit
// what bssm_pf_run returns for the same (theta, seed, stream).
// ---------------------------------------------------------------------------
struct BatchArgs {
    int N, T, resample_algorithm, resample_fn, lim;
    int fold;                                         // FromLw::fold
    int lit_max;                                      // N <= lit_max: exact sums by the in-order pass (block_literal_terms)
    double move_sd;                                   // resample-move: sd of the random-walk Metropolis proposal
    double threshold;
    const double* y; const int* obs_times;            // [T]; obs_times may be nullptr (1..T)
    const double* lgy;                                // [T] lgamma(y + 1) (SIR's Poisson observation density), or nullptr
    const double* theta; int theta_stride;            // [F][theta_stride]: phi, sigma_x, sigma_y
    const double* log_sy;                             // [F] log(sigma_y), taken on the host like bssm_pf_run does (device log may differ in the last bit)
    const PhiloxKey* keys;                            // [F]
    double* loglike; double* state_est; double* ess; double* llh;    // [F], [F][T+1][D], [F][T+1], [F][T]
    int* dead; uint32_t* flags; int* res_calls;       // [F]
    long long* phase_cycles;                          // dev tool: [8] cycles per phase summed over filter 0's observations + [2][16] stage stamps, or nullptr
};

// k_step<MODEL, TRANS, WEIGHT, false> for one block of up to EB particles, by NT threads: thread t plays the threads
// t, t + NT, t + 2 NT, t + 3 NT of the NTS-thread kernel (wave w of round r = wave w + 4 r there).
template <int MODEL, bool TRANS, int WEIGHT, bool SUBAUX = false>
__device__ __forceinline__ void step_emul(double* sh16, double* x, double* __restrict__ lw, const double* __restrict__ auxg, long long N,
                                          const ModelPar& par, double y, const NoiseSrc& ns, double* pm, double* ps, double* pq)
{
    constexpr int R = NTS / NT;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    double l0[R], l1[R];
    // rounds past the last particle contribute max = -inf and sums = 0 in the k_step they re-enact: skip their work
    const int rmax = (int)((N + 2 * NT - 1) / (2 * NT));
#pragma unroll
    for (int r = 0; r < R; r++) {
        const long long j = 2 * (long long)(t + NT * r);
        l0[r] = -INFINITY; l1[r] = -INFINITY;
        if (j < N) {
            const bool two = (j + 1 < N);
            if constexpr (MODEL == 2) {       // k_step_sir: state (s, i) as [2][N]
                double s0 = x[j], i0 = x[N + j];
                double s1 = two ? x[j + 1] : 0.0, i1 = two ? x[N + j + 1] : 0.0;
                if (TRANS) {
                    Sir::transition(s0, i0, par, ns.key, ns.call, (uint32_t)j);
                    if (two) Sir::transition(s1, i1, par, ns.key, ns.call, (uint32_t)(j + 1));
                    x[j] = s0; x[N + j] = i0;
                    if (two) { x[j + 1] = s1; x[N + j + 1] = i1; }
                }
                if (WEIGHT) {
                    if (WEIGHT == 2) { l0[r] = Sir::aux_loglik(y, s0, i0, par); l1[r] = Sir::aux_loglik(y, s1, i1, par); }
                    else { l0[r] = Sir::loglik(y, i0, par); l1[r] = Sir::loglik(y, i1, par); }
                    if (SUBAUX) { l0[r] = l0[r] - auxg[j]; if (two) l1[r] = l1[r] - auxg[j + 1]; }
                    lw[j] = l0[r];
                    if (two) lw[j + 1] = l1[r]; else l1[r] = -INFINITY;
                }
            } else {
                double x0 = x[j], x1 = two ? x[j + 1] : 0.0;
                if (TRANS) {
                    double z0, z1;
                    normal_pair(ns.key, ns.purpose, ns.call, 0, (uint32_t)(j >> 1), z0, z1);
                    x0 = Model<MODEL>::transition(x0, z0, par);
                    x1 = Model<MODEL>::transition(x1, z1, par);
                    x[j] = x0; if (two) x[j + 1] = x1;
                }
                if (WEIGHT) {
                    if (WEIGHT == 2) {
                        l0[r] = r_dnorm_log(y, Model<MODEL>::forecast(x0, par), par.sy, par.log_sy);
                        l1[r] = r_dnorm_log(y, Model<MODEL>::forecast(x1, par), par.sy, par.log_sy);
                    } else {
                        l0[r] = r_dnorm_log(y, x0, par.sy, par.log_sy);
                        l1[r] = r_dnorm_log(y, x1, par.sy, par.log_sy);
                    }
                    if (SUBAUX) { l0[r] = l0[r] - auxg[j]; if (two) l1[r] = l1[r] - auxg[j + 1]; }
                    lw[j] = l0[r];
                    if (two) lw[j + 1] = l1[r]; else l1[r] = -INFINITY;
                }
            }
        }
    }
    if (WEIGHT) {
        double v[R];
#pragma unroll
        for (int r = 0; r < R; r++) v[r] = (r < rmax) ? wave_max(fmax(l0[r], l1[r])) : -INFINITY;
        __syncthreads();
        if (lane == 0) {
#pragma unroll
            for (int r = 0; r < R; r++) sh16[wave + (NT / 64) * r] = v[r];
        }
        __syncthreads();
        double bm = sh16[0];
#pragma unroll
        for (int i = 1; i < NTS / 64; i++) bm = fmax(bm, sh16[i]);
        double sv[R], qv[R];
#pragma unroll
        for (int r = 0; r < R; r++) {
            double s_ = 0.0, q_ = 0.0;
            if (bm > -INFINITY) {
                if (l0[r] > -INFINITY) { const double e = exp(l0[r] - bm); s_ += e; q_ += e * e; }
                if (l1[r] > -INFINITY) { const double e = exp(l1[r] - bm); s_ += e; q_ += e * e; }
            }
            if (r < rmax) { sv[r] = wave_sum(s_); qv[r] = wave_sum(q_); } else { sv[r] = 0.0; qv[r] = 0.0; }
        }
        __syncthreads();
        if (lane == 0) {
#pragma unroll
            for (int r = 0; r < R; r++) sh16[wave + (NT / 64) * r] = sv[r];
        }
        __syncthreads();
        double S = 0.0;
#pragma unroll
        for (int i = 0; i < NTS / 64; i++) S += sh16[i];
        __syncthreads();
        if (lane == 0) {
#pragma unroll
            for (int r = 0; r < R; r++) sh16[wave + (NT / 64) * r] = qv[r];
        }
        __syncthreads();
        double Q = 0.0;
#pragma unroll
        for (int i = 0; i < NTS / 64; i++) Q += sh16[i];
        if (t == 0) { *pm = bm; *ps = S; *pq = Q; }
    }
}

// ALG: 0 bootstrap, 1 auxiliary, 2 resample-move
#ifndef BSSM_BATCH_WPE
#define BSSM_BATCH_WPE 2
#endif
template <int MODEL, int ALG>
__global__ __launch_bounds__(NT, (MODEL == 2 || ALG == 1) ? 1 : BSSM_BATCH_WPE) void k_pf_batch(BatchArgs g)
{
    constexpr bool APF = (ALG == 1), RMPF = (ALG == 2);
    __shared__ SegSmem sm;
    __shared__ uint64_t tin[NT + 1];
    __shared__ int Tl[EB];
    __shared__ int Tbegin;
    __shared__ double es[NT];
    __shared__ double sh16[NTS / 64];
    constexpr int D = (MODEL == 2) ? 2 : 1;                      // state dimension; particles are [D][N]
    __shared__ __attribute__((aligned(16))) double XA[D * EB];
    __shared__ __attribute__((aligned(16))) double XB[D * EB];
    __shared__ __attribute__((aligned(16))) double LW[EB];      // log-weights, then (in place) the normalised weights
    __shared__ __attribute__((aligned(16))) double AUXLW[APF ? EB : 2];     // auxiliary filter: first-stage log-weights ...
    __shared__ __attribute__((aligned(16))) double AUXG[APF ? EB : 2];      // ... and their gather through the first-stage ancestors
    __shared__ DevState st;
    __shared__ BlockRec br;
    __shared__ double pm1, ps1, pq1, ainw1, ainp1;
    __shared__ double sep[2];
    __shared__ uint64_t cin1;
    extern __shared__ __attribute__((aligned(16))) double CUM[];       // [EB], multinomial resampling only (sized by the launch)
    const int fi = blockIdx.x, t = threadIdx.x;
    const long long N = g.N;
    const int T = g.T;
    ModelPar par;
    {
        const double* th = g.theta + (long long)fi * g.theta_stride;
        par.phi = th[0]; par.sx = th[1]; par.sy = th[2]; par.log_sy = g.log_sy[fi];
        par.n_total = 0.0; par.s0 = 0.0; par.i0 = 0.0; par.lgy = 0.0;
        if (MODEL == 2) { par.n_total = th[2]; par.s0 = th[3]; par.i0 = th[4]; }      // SIR: (lambda, gamma, n_total, s0, i0)
    }
    const PhiloxKey key = g.keys[fi];
    const bool lit = g.N <= g.lit_max;      // few terms: the in-order pass beats the record machinery's fixed latency
    if (t == 0) {
        st.loglike = 0.0; st.lse_max = 0.0; st.lse_sum = 0.0; st.ess = 0.0; st.total_bits = 0;
        st.do_resample = 0; st.dead = 0; st.flags = 0; st.res_calls = 0; st.cur_call = 0;
        st.debug_stop = (g.phase_cycles != nullptr && fi == 0) ? 99 : 0;      // dev tool: stage stamps of the last observation
        st.stat_hard_blocks = 0; st.stat_serial_walks = 0; st.stat_literal_terms = 0;
        sep[0] = 0.0; sep[1] = 0.0;
    }
    __syncthreads();
    {   // t = 0  (R/particle_filter_core.R:76-116)
        NoiseSrc ns; ns.arr = nullptr; ns.key = key; ns.purpose = DRAW_INIT; ns.call = 0;
        init_block(sm.sh4, 0, XA, N, ns, sep, MODEL, par);
    }
    __syncthreads();
    if (t == 0 && g.state_est) { for (int d = 0; d < D; d++) { double s0 = 0.0; s0 += sep[d]; g.state_est[(long long)fi * (T + 1) * D + d] = s0; } }
    double* xa = XA; double* xb = XB;
    long long ph[6] = {0, 0, 0, 0, 0, 0};
#ifdef BSSM_DEV_STAMPS
    const bool prof = (g.phase_cycles != nullptr) && fi == 0 && t == 0;
#define PH(k) { if (prof) { const long long c_ = clock64(); ph[k] += c_ - pc; pc = c_; } }
#else
    const bool prof = false;
#define PH(k) { (void)pc; }
#endif
    int ktrans = 0, prev_t = 0;
    for (int i = 1; i <= T; i++) {                                                        // :123
        const int ot = g.obs_times ? g.obs_times[i - 1] : i;
        const int gap = ot - prev_t;                                                      // :124
        prev_t = ot;
        const double yi = g.y[i - 1];
        if (MODEL == 2) par.lgy = g.lgy[i - 1];                  // lgamma(y + 1), taken on the host like bssm_pf_run does
        long long pc = prof ? clock64() : 0;
        NoiseSrc ns; ns.arr = nullptr; ns.key = key; ns.purpose = DRAW_TRANS;
        for (int step = 1; step <= gap; step++) {                                         // :125-136
            ns.call = (uint32_t)ktrans;
            if (!APF && step == gap) step_emul<MODEL, true, 1>(sh16, xa, LW, nullptr, N, par, yi, ns, &pm1, &ps1, &pq1);
            else step_emul<MODEL, true, 0>(sh16, xa, LW, nullptr, N, par, yi, ns, &pm1, &ps1, &pq1);
            ktrans++;
            __syncthreads();
        }
        if (APF) {                                                                        // :140-175
            // first stage: look-ahead weights at the current particles, resample particles and carry the log-weights along
            ns.call = 0;
            step_emul<MODEL, false, 2>(sh16, xa, AUXLW, nullptr, N, par, yi, ns, &pm1, &ps1, &pq1);
            __syncthreads();
            FromLw fa;
            fa.lw = AUXLW; fa.xw = nullptr; fa.w_out = LW; fa.pm = &pm1; fa.ps = &ps1; fa.pq = &pq1; fa.nb = 1; fa.gmax = nullptr; fa.fold = g.fold; fa.lead = 0; fa.pub = 0; fa.ain_out = &ainw1;
            fa.plan = PLAN_AUX; fa.N = N; fa.obs_i = i; fa.resample_algorithm = g.resample_algorithm; fa.threshold = g.threshold;
            fa.ess_out = nullptr; fa.llh_out = nullptr; fa.resampled_out = nullptr;
            if (lit) local_block<MODE_W, true, NT, true>(sm, tin, es, 0, 1, LW, N, nullptr, g.lim, &br, nullptr, &st, fa, NoResolve(), nullptr, 1);
            else local_block<MODE_W, true, NT, false>(sm, tin, es, 0, 1, LW, N, nullptr, g.lim, &br, nullptr, &st, fa, NoResolve(), nullptr, 1);
            __syncthreads();
            if (t == 0 && !st.dead && !st.flags && st.do_resample) {
                const uint64_t fs = g.fold ? d2b(1.0) : br.prefix.o[0];
                const double tot = b2d(fs);
                st.total_bits = fs;
                if (tot == 0.0) st.flags |= FLAG_ZERO_SUM;
                if (!isfinite(tot)) st.flags |= FLAG_NONFINITE;
                ainp1 = ainw1 / tot; cin1 = 0;
            }
            __syncthreads();
            ApplyArgs aa;
            aa.w = LW; aa.nw = N; aa.ain_p = &ainp1; aa.cin = &cin1; aa.lim = g.lim; aa.n = (int)N;
            aa.u_base = nullptr; aa.u_stride = 0; aa.key = key; aa.anc_out = nullptr; aa.anc_stride = 0; aa.cum_out = nullptr;
            aa.xsrc = xa; aa.xdst = xb; aa.dim = D; aa.xstride = N; aa.auxsrc = AUXLW; aa.auxdst = AUXG; aa.se_part = nullptr; aa.nstage = 0; aa.lead = 0; aa.last = 0; aa.step_model = -1; aa.step_lw = nullptr;
            if (g.resample_fn == 1) {
                if (lit) apply_block<1, true>(sm, tin, Tl, Tbegin, 0, 1, aa, &st); else apply_block<1, false>(sm, tin, Tl, Tbegin, 0, 1, aa, &st);
            } else if (g.resample_fn == 0) {
                if (lit) apply_block<0, true>(sm, tin, Tl, Tbegin, 0, 1, aa, &st); else apply_block<0, false>(sm, tin, Tl, Tbegin, 0, 1, aa, &st);
            } else {                                                                      // multinomial: exact cum_sum, then inverse-CDF search
                aa.cum_out = CUM;
                if (lit) apply_block<2, true>(sm, tin, Tl, Tbegin, 0, 1, aa, &st); else apply_block<2, false>(sm, tin, Tl, Tbegin, 0, 1, aa, &st);
                __syncthreads();
                multinomial_block(sm.sh4, 0, CUM, N, (int)N, aa, &st);
            }
            __syncthreads();
            { double* tmp = xa; xa = xb; xb = tmp; }
            // second stage: propagate the resampled particles, weight by  loglik - aux_lw[ancestor]   (:159-175)
            ns.call = (uint32_t)ktrans;
            step_emul<MODEL, true, 1, true>(sh16, xa, LW, AUXG, N, par, yi, ns, &pm1, &ps1, &pq1);
            ktrans++;
            __syncthreads();
        } else if (gap <= 0) {                           // obs_times repeats a time: weights on the current particles
            ns.call = 0;
            step_emul<MODEL, false, 1>(sh16, xa, LW, nullptr, N, par, yi, ns, &pm1, &ps1, &pq1);
            __syncthreads();
        }
        if (t == 0) { sep[0] = 0.0; sep[1] = 0.0; }
        FromLw fl;
        fl.lw = LW; fl.xw = nullptr; fl.w_out = LW; fl.pm = &pm1; fl.ps = &ps1; fl.pq = &pq1; fl.nb = 1; fl.gmax = nullptr; fl.fold = g.fold; fl.lead = 0; fl.pub = 0; fl.ain_out = &ainw1;
        fl.plan = PLAN_PF; fl.N = N; fl.obs_i = i; fl.resample_algorithm = g.resample_algorithm; fl.threshold = g.threshold;
        fl.ess_out = g.ess + (long long)fi * (T + 1); fl.llh_out = g.llh + (long long)fi * T; fl.resampled_out = nullptr;
        // normalise + loglik/ESS/decision + the exact sum(weights) of the block (:204-218, src/resampling.cpp:20-24)
        if (lit) local_block<MODE_W, true, NT, true>(sm, tin, es, 0, 1, LW, N, nullptr, g.lim, &br, nullptr, &st, fl, NoResolve(), nullptr, 1);
        else local_block<MODE_W, true, NT, false>(sm, tin, es, 0, 1, LW, N, nullptr, g.lim, &br, nullptr, &st, fl, NoResolve(), nullptr, 1);
        __syncthreads();
        PH(1)
        if (t == 0 && !st.dead && !st.flags && st.do_resample) {      // what k_resolve<W> / k_resolve<P> come to for one block
            const uint64_t fs = g.fold ? d2b(1.0) : br.prefix.o[0];
            const double tot = b2d(fs);
            st.total_bits = fs;
            if (tot == 0.0) st.flags |= FLAG_ZERO_SUM;
            if (!isfinite(tot)) st.flags |= FLAG_NONFINITE;
            ainp1 = ainw1 / tot; cin1 = 0;
        }
        __syncthreads();
        ApplyArgs a;
        a.w = LW; a.nw = N; a.ain_p = &ainp1; a.cin = &cin1; a.lim = g.lim; a.n = (int)N;
        a.u_base = nullptr; a.u_stride = 0; a.key = key; a.anc_out = nullptr; a.anc_stride = 0; a.cum_out = nullptr;
        a.xsrc = xa; a.xdst = xb; a.dim = D; a.xstride = N; a.auxsrc = nullptr; a.auxdst = nullptr; a.se_part = sep; a.nstage = 0; a.lead = 0; a.last = 0; a.step_model = -1; a.step_lw = nullptr;
        if (g.resample_fn == 1) {                                                         // systematic
            if (lit) apply_block<1, true>(sm, tin, Tl, Tbegin, 0, 1, a, &st); else apply_block<1, false>(sm, tin, Tl, Tbegin, 0, 1, a, &st);
        } else if (g.resample_fn == 0) {                                                  // stratified
            if (lit) apply_block<0, true>(sm, tin, Tl, Tbegin, 0, 1, a, &st); else apply_block<0, false>(sm, tin, Tl, Tbegin, 0, 1, a, &st);
        } else {                                                                          // multinomial: exact cum_sum, then inverse-CDF search
            a.cum_out = CUM;
            if (lit) apply_block<2, true>(sm, tin, Tl, Tbegin, 0, 1, a, &st); else apply_block<2, false>(sm, tin, Tl, Tbegin, 0, 1, a, &st);
            __syncthreads();
            multinomial_block(sm.sh4, 0, CUM, N, (int)N, a, &st);
        }
        __syncthreads();
        PH(2)
        if (g.resample_algorithm != 1) {                 // SIS / SISAR: carry over when no resample ran (:238)
            carry_block(sm.sh4, 0, xa, xb, LW, N, D, sep, &st);
            __syncthreads();
        }
        { double* tmp = xa; xa = xb; xb = tmp; }
        if constexpr (RMPF && MODEL != 2) {              // move every particle, then take the state estimate (:226-241)
            move_block<MODEL>(sm.sh4, 0, xa, N, par, yi, g.move_sd, nullptr, nullptr, key, (uint32_t)i, sep, &st);
            __syncthreads();
        }
        if (t == 0 && g.state_est) { for (int d = 0; d < D; d++) { double s0 = 0.0; s0 += sep[d]; g.state_est[((long long)fi * (T + 1) + i) * D + d] = s0; } }   // :237-241
        if (st.dead) break;                              // degenerate weights: the reference returns at once (:189-202)
        __syncthreads();
        PH(3)
    }
    if (prof) { for (int k = 0; k < 6; k++) g.phase_cycles[k] = ph[k]; for (int r = 0; r < 2; r++) for (int k = 0; k < 16; k++) g.phase_cycles[8 + 16 * r + k] = st.stamps[2 + r][k]; }
#undef PH
    if (t == 0) { g.loglike[fi] = st.loglike; g.dead[fi] = st.dead; g.flags[fi] = st.flags; g.res_calls[fi] = st.res_calls; }
}

__global__ void k_reset_state(DevState* st)
{
    st->loglike = 0.0; st->lse_max = 0.0; st->lse_sum = 0.0; st->ess = 0.0; st->total_bits = 0;
    st->do_resample = 0; st->dead = 0; st->flags = 0; st->res_calls = 0; st->cur_call = 0; st->debug_stop = 0;
    st->out_lo = 0; st->out_hi = 0;
    st->stat_hard_blocks = 0; st->stat_serial_walks = 0; st->stat_literal_terms = 0;
}

}  // namespace bssm
