// kernels.hip.h -- HIP kernels of the particle-filter hot path (gfx950, wave64).
//
// Hot path of the reference, per observation (R/particle_filter_core.R:123-246):
//   transition_fn -> weight_fn -> max/exp/sum normalise -> loglik, ESS ->
//   resample_fn (src/resampling.cpp) -> gather -> state estimate.
// Kernel map (one launch each, all on the context's stream):
//   k_step        propagate (+ log-weight, + per-block log-sum-exp partial)   :127,:177-183,:204-206
//   k_normalize   w = exp(lw - max)/sum, per-block sum w and sum w^2          :205-207,:211
//   k_plan        loglik / ESS / resample decision; approximate block prefixes :208-218
//   k_local<W|P>  per-block "records" of the exact sequential sum (seqsum.h)  src/resampling.cpp:20,25
//   k_resolve<>   exact incoming state of every block; exact total            src/resampling.cpp:20
//   k_apply<>     exact cum_sum -> output counts -> ancestors -> gather        src/resampling.cpp:28-37, R/resampling.R:40
//   k_multinomial inverse-CDF search on the exact cum_sum                      src/resampling.cpp:11 (distributional)
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

constexpr int NT = 256;          // threads per workgroup (4 waves)
constexpr int EL = 8;            // terms per thread in the scan kernels
constexpr int EB = NT * EL;      // 2048 terms per workgroup
constexpr int MAXB = 2048;       // max scan workgroups  => N <= 2^22 per filter
constexpr int MAXG = 64;         // max groups in k_resolve

constexpr int MODE_W = 0;        // terms are the weights themselves        (total = sum(w))
constexpr int MODE_P = 1;        // terms are prob = w / total              (cum = cumsum(prob))

constexpr int PLAN_RESAMPLE_ONLY = 0;   // stand-alone resampler
constexpr int PLAN_PF = 1;              // filter: weights of an observation
constexpr int PLAN_AUX = 2;             // APF first stage (R/particle_filter_core.R:152-155)

constexpr uint32_t FLAG_NEGATIVE = 1u, FLAG_ZERO_SUM = 2u, FLAG_NONFINITE = 4u;

#define BSSM_LN_SQRT_2PI 0.918938533204672741780329736406

struct BlockRec { Rec prefix; int32_t tail_from; int32_t pad; };   // 64 bytes

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
    int32_t pad0;
    long long stat_hard_blocks, stat_serial_walks, stat_literal_terms;
};

// ---------------------------------------------------------------------------
// small block-level helpers (wave64 shuffles, then LDS across the 4 waves)
// ---------------------------------------------------------------------------
__device__ __forceinline__ double wave_sum(double v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    return v;
}
__device__ __forceinline__ double wave_max(double v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = fmax(v, __shfl_down(v, off, 64));
    return v;
}
// result valid in every thread
__device__ __forceinline__ double block_sum(double v, double* sh4)
{
    v = wave_sum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sh4[threadIdx.x >> 6] = v;
    __syncthreads();
    return (sh4[0] + sh4[1]) + (sh4[2] + sh4[3]);
}
__device__ __forceinline__ double block_max(double v, double* sh4)
{
    v = wave_max(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sh4[threadIdx.x >> 6] = v;
    __syncthreads();
    return fmax(fmax(sh4[0], sh4[1]), fmax(sh4[2], sh4[3]));
}
// exclusive prefix of x over the block (re-associated: only an approximation
// of the sequential prefix is needed here)
__device__ __forceinline__ double block_excl_scan(double x, double* sh4)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    double inc = x;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const double y = __shfl_up(inc, off, 64);
        if (lane >= off) inc += y;
    }
    double exc = __shfl_up(inc, 1, 64);
    if (lane == 0) exc = 0.0;
    __syncthreads();
    if (lane == 63) sh4[wave] = inc;
    __syncthreads();
    double pre = 0.0;
    for (int i = 0; i < wave; i++) pre += sh4[i];
    return pre + exc;
}

// (max, sum exp(. - max)) pairs: combine b into a
__device__ __forceinline__ void lse_combine(double& m, double& s, double mb, double sb)
{
    if (mb > m) { s = s * exp(m - mb) + sb; m = mb; }
    else if (mb > -INFINITY) { s = s + sb * exp(mb - m); }
}

// Reduce the per-block (max, sumexp) partials to the global pair; every
// thread of every calling block gets the same value (same order everywhere).
__device__ void reduce_lse_partials(const double* pm, const double* ps, int nb, double* sh4, double& M, double& S)
{
    double m = -INFINITY;
    for (int i = threadIdx.x; i < nb; i += NT) m = fmax(m, pm[i]);
    M = block_max(m, sh4);
    double s = 0.0;
    for (int i = threadIdx.x; i < nb; i += NT) {
        const double mb = pm[i];
        if (mb > -INFINITY) s += ps[i] * exp(mb - M);
    }
    S = block_sum(s, sh4);
}

// ---------------------------------------------------------------------------
// Built-in models: the reference's user closures restated for the device.
// theta = (phi, sigma_x, sigma_y).
// ---------------------------------------------------------------------------
struct ModelPar { double phi, sx, sy, log_sy; };

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

struct NoiseSrc {
    const double* arr;     // parity mode: N draws for this call; nullptr => generator
    PhiloxKey key;
    uint32_t purpose, call;
};

// ---------------------------------------------------------------------------
// k_init: init_fn = rnorm(N, 0, 1)   (R/particle_filter_core.R:76), t = 0 state
// estimate partial sums (:109)
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(NT) void k_init(double* __restrict__ x, long long N, NoiseSrc ns,
                                             double* __restrict__ se_part /* [nblocks] */)
{
    __shared__ double sh4[4];
    const long long base = (long long)blockIdx.x * EB;
    const double invN = 1.0 / (double)N;
    double acc = 0.0;
#pragma unroll
    for (int r = 0; r < EL / 2; r++) {
        const long long j = base + 2 * (threadIdx.x + NT * r);
        if (j < N) {
            double z0, z1;
            if (ns.arr) { z0 = ns.arr[j]; z1 = (j + 1 < N) ? ns.arr[j + 1] : 0.0; }
            else normal_pair(ns.key, ns.purpose, ns.call, 0, (uint32_t)(j >> 1), z0, z1);
            const double x0 = r_rnorm(0.0, 1.0, z0), x1 = r_rnorm(0.0, 1.0, z1);
            x[j] = x0; acc += x0 * invN;
            if (j + 1 < N) { x[j + 1] = x1; acc += x1 * invN; }
        }
    }
    acc = block_sum(acc, sh4);
    if (threadIdx.x == 0) se_part[blockIdx.x] = acc;
}

// ---------------------------------------------------------------------------
// k_step: transition_fn and/or weight_fn for one call, fused with the
// per-block (max, sum exp) partial of the log-sum-exp normalisation.
//   TRANS   : x <- transition(x, z)                     (:127, :159)
//   WEIGHT 1: lw = dnorm(y, x', sy, log)                (:177-182)
//   WEIGHT 2: lw = aux log-lik at the CURRENT particles  (:142-147), no transition
//   SUBAUX  : lw -= aux_lw[ancestor] (already gathered)  (:175)
// ---------------------------------------------------------------------------
template <int MODEL, bool TRANS, int WEIGHT, bool SUBAUX>
__global__ __launch_bounds__(NT) void k_step(const double* xin, double* xout /* may alias xin */,
                                             double* __restrict__ lw, const double* __restrict__ auxg,
                                             long long N, ModelPar par, double y, NoiseSrc ns,
                                             double* __restrict__ pm, double* __restrict__ ps,
                                             const DevState* __restrict__ st)
{
    if (st->dead) return;
    __shared__ double sh4[4];
    const long long base = (long long)blockIdx.x * EB;
    double lv[EL];
    double m = -INFINITY;
#pragma unroll
    for (int r = 0; r < EL / 2; r++) {
        const long long j = base + 2 * (threadIdx.x + NT * r);
        lv[2 * r] = lv[2 * r + 1] = -INFINITY;
        if (j < N) {
            const bool two = (j + 1 < N);
            double x0 = xin[j], x1 = two ? xin[j + 1] : 0.0;
            if (TRANS) {
                double z0, z1;
                if (ns.arr) { z0 = ns.arr[j]; z1 = two ? ns.arr[j + 1] : 0.0; }
                else normal_pair(ns.key, ns.purpose, ns.call, 0, (uint32_t)(j >> 1), z0, z1);
                x0 = Model<MODEL>::transition(x0, z0, par);
                x1 = Model<MODEL>::transition(x1, z1, par);
                xout[j] = x0;
                if (two) xout[j + 1] = x1;
            }
            if (WEIGHT) {
                double l0, l1;
                if (WEIGHT == 2) {
                    l0 = r_dnorm_log(y, Model<MODEL>::forecast(x0, par), par.sy, par.log_sy);
                    l1 = r_dnorm_log(y, Model<MODEL>::forecast(x1, par), par.sy, par.log_sy);
                } else {
                    l0 = r_dnorm_log(y, x0, par.sy, par.log_sy);
                    l1 = r_dnorm_log(y, x1, par.sy, par.log_sy);
                }
                if (SUBAUX) { l0 = l0 - auxg[j]; if (two) l1 = l1 - auxg[j + 1]; }
                lw[j] = l0; lv[2 * r] = l0; m = fmax(m, l0);
                if (two) { lw[j + 1] = l1; lv[2 * r + 1] = l1; m = fmax(m, l1); }
            }
        }
    }
    if (WEIGHT) {
        const double bm = block_max(m, sh4);
        double s = 0.0;
        if (bm > -INFINITY) {
#pragma unroll
            for (int k = 0; k < EL; k++) if (lv[k] > -INFINITY) s += exp(lv[k] - bm);
        }
        s = block_sum(s, sh4);
        if (threadIdx.x == 0) { pm[blockIdx.x] = bm; ps[blockIdx.x] = s; }
    }
}

// ---------------------------------------------------------------------------
// k_normalize: weights = exp(lw - max) / sum  (:205-207), per-block plain sums
// of w (approximate prefix for the exact scan) and of w^2 (ESS, :211)
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(NT) void k_normalize(const double* __restrict__ lw, double* __restrict__ w, long long N,
                                                  const double* __restrict__ pm, const double* __restrict__ ps, int nb,
                                                  double* __restrict__ bsum, double* __restrict__ bsq,
                                                  int check_degenerate, DevState* st)
{
    if (st->dead) return;
    __shared__ double sh4[4];
    double M, S;
    reduce_lse_partials(pm, ps, nb, sh4, M, S);
    if (check_degenerate && M < -1e8) return;   // degenerate: k_plan marks the run dead (:189)
    const long long base = (long long)blockIdx.x * EB;
    double s1 = 0.0, s2 = 0.0;
#pragma unroll
    for (int r = 0; r < EL / 2; r++) {
        const long long j = base + 2 * (threadIdx.x + NT * r);
        if (j < N) {
            const double w0 = exp(lw[j] - M) / S;
            w[j] = w0; s1 += w0; s2 += w0 * w0;
            if (j + 1 < N) { const double w1 = exp(lw[j + 1] - M) / S; w[j + 1] = w1; s1 += w1; s2 += w1 * w1; }
        }
    }
    s1 = block_sum(s1, sh4);
    s2 = block_sum(s2, sh4);
    if (threadIdx.x == 0) {
        bsum[blockIdx.x] = s1; bsq[blockIdx.x] = s2;
        if (!isfinite(s1)) atomicOr(&st->flags, FLAG_NONFINITE);   // NaN/Inf log-weights: the scan kernels stand down
    }
}

// stand-alone resampler front end: validation (src/resampling.cpp:6,18,45) + block sums
__global__ __launch_bounds__(NT) void k_bsum(const double* __restrict__ w, long long nw, double* __restrict__ bsum, DevState* st)
{
    __shared__ double sh4[4];
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
// k_plan (one workgroup): scalars of the observation + approximate exclusive
// block prefixes ain[b] for the exact scan.
// ---------------------------------------------------------------------------
template <int PLAN>
__global__ __launch_bounds__(NT) void k_plan(const double* __restrict__ pm, const double* __restrict__ ps, int nb_lse,
                                             const double* __restrict__ bsum, const double* __restrict__ bsq, int B,
                                             double* __restrict__ ain, DevState* st, long long N,
                                             int obs_i /* 1-based */, int resample_algorithm, double threshold,
                                             double* __restrict__ ess_out, double* __restrict__ llh_out, int* __restrict__ resampled_out)
{
    __shared__ double sh4[4];
    if (PLAN != PLAN_RESAMPLE_ONLY) {
        if (st->dead) return;
        if (PLAN == PLAN_PF) {
            double M, S;
            reduce_lse_partials(pm, ps, nb_lse, sh4, M, S);
            if (M < -1e8) {                     // all(log_weights < -1e8)  (:189-202)
                if (threadIdx.x == 0) {
                    st->loglike = -INFINITY; llh_out[obs_i - 1] = -INFINITY; st->dead = obs_i; st->do_resample = 0;
                }
                return;
            }
            double q = 0.0;
            for (int i = threadIdx.x; i < B; i += NT) q += bsq[i];
            q = block_sum(q, sh4);
            if (threadIdx.x == 0) {
                const double ll = st->loglike + (M + log(S) - log((double)N));   // :208
                st->loglike = ll; llh_out[obs_i - 1] = ll;                        // :209
                const double ess = 1.0 / q;                                       // :211
                const int doit = (resample_algorithm == 0) ? 0 : (resample_algorithm == 1) ? 1 : (ess < threshold);  // :214-218
                st->do_resample = doit;
                ess_out[obs_i] = doit ? (double)N : ess;                          // :212,:223
                if (resampled_out) resampled_out[obs_i - 1] = doit;
                st->ess = ess; st->lse_max = M; st->lse_sum = S;
                if (doit) { st->cur_call = st->res_calls; st->res_calls += 1; }
            }
        } else {
            if (threadIdx.x == 0) { st->do_resample = 1; st->cur_call = st->res_calls; st->res_calls += 1; }
        }
    } else {
        if (threadIdx.x == 0) { st->do_resample = 1; st->cur_call = st->res_calls; st->res_calls += 1; }
    }
    // exclusive scan of the block sums (B <= MAXB = NT * 8)
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
// exact sequential scan: shared pieces
// ---------------------------------------------------------------------------
struct ScanSmem {
    Rec a[NT];
    Rec b[NT];
    Rec orig[NT];
    uint64_t tin[NT];
    double sh4[4];
    int tail;
};

template <int MODE>
__device__ __forceinline__ void load_terms(const double* __restrict__ w, long long nw, double total, long long j0, double v[EL])
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
__device__ uint64_t literal_run(const double* __restrict__ w, long long nw, double total, long long j0, long long j1, uint64_t in)
{
    double c = b2d(in);
    for (long long j = j0; j < j1; j++) c = c + term_at<MODE>(w, nw, total, j);
    return d2b(c);
}

// Hillis-Steele inclusive scan of the 256 thread records in LDS.
__device__ __forceinline__ void scan_records(ScanSmem& sm, const Rec& mine, Rec& inc, Rec& exc)
{
    const int t = threadIdx.x;
    sm.orig[t] = mine;
    sm.a[t] = mine;
    __syncthreads();
    Rec* src = sm.a;
    Rec* dst = sm.b;
    for (int off = 1; off < NT; off <<= 1) {
        Rec r = src[t];
        if (t >= off) r = rec_compose(src[t - off], r);
        dst[t] = r;
        __syncthreads();
        Rec* tmp = src; src = dst; dst = tmp;
    }
    inc = src[t];
    if (t > 0) exc = src[t - 1]; else exc = rec_identity(0);
}

// ---------------------------------------------------------------------------
// k_local: block record = composite of threads [0, tail_from)
// ---------------------------------------------------------------------------
template <int MODE>
__global__ __launch_bounds__(NT) void k_local(const double* __restrict__ w, long long nw, const double* __restrict__ ain,
                                              int lim, BlockRec* __restrict__ brec, DevState* st)
{
    if (st->dead || !st->do_resample || st->flags) return;
    __shared__ ScanSmem sm;
    const int t = threadIdx.x;
    const long long b0 = (long long)blockIdx.x * EB;
    const double total = (MODE == MODE_P) ? b2d(st->total_bits) : 1.0;
    double v[EL];
    load_terms<MODE>(w, nw, total, b0 + (long long)t * EL, v);
    double ts = 0.0;
#pragma unroll
    for (int k = 0; k < EL; k++) ts += v[k];
    const double a_in = ain[blockIdx.x];
    const double h = a_in + block_excl_scan(ts, sm.sh4);
    const Rec mine = chunk_record(v, EL, 1, h, lim);
    Rec inc, exc;
    if (t == 0) sm.tail = NT;
    scan_records(sm, mine, inc, exc);
    if (inc.kind == REC_HARD) atomicMin(&sm.tail, t);
    __syncthreads();
    const int tail = sm.tail;
    if (tail < NT && a_in == 0.0) {
        // incoming state is exactly +0: walk the thread records with the exact state now
        if (t == 0) {
            uint64_t s = 0;
            long long lit = 0;
            for (int tt = 0; tt < NT; tt++) {
                bool ok = true;
                uint64_t o = rec_step(sm.orig[tt], s, ok);
                if (!ok) { o = literal_run<MODE>(w, nw, total, b0 + (long long)tt * EL, b0 + (long long)(tt + 1) * EL, s); lit += EL; }
                s = o;
            }
            BlockRec br; br.prefix = rec_abs(s); br.tail_from = NT; br.pad = 0;
            brec[blockIdx.x] = br;
            atomicAdd((unsigned long long*)&st->stat_serial_walks, 1ull);
            atomicAdd((unsigned long long*)&st->stat_literal_terms, (unsigned long long)lit);
        }
        return;
    }
    if (tail == NT) {
        if (t == NT - 1) { BlockRec br; br.prefix = inc; br.tail_from = NT; br.pad = 0; brec[blockIdx.x] = br; }
    } else if (tail == 0) {
        if (t == 0) {
            BlockRec br; br.prefix = rec_identity(d2b(a_in)); br.tail_from = 0; br.pad = 0; brec[blockIdx.x] = br;
            atomicAdd((unsigned long long*)&st->stat_hard_blocks, 1ull);
        }
    } else if (t == tail - 1) {
        BlockRec br; br.prefix = inc; br.tail_from = tail; br.pad = 0; brec[blockIdx.x] = br;
        atomicAdd((unsigned long long*)&st->stat_hard_blocks, 1ull);
    }
}

// ---------------------------------------------------------------------------
// k_resolve (one workgroup): exact incoming state cin[b] of every block.
// Blocks are taken in ~sqrt(B) groups: group records by composition, one
// serial walk over the groups with the exact state, then every group walks
// its own blocks.  Tails / window misses are re-run literally.
// ---------------------------------------------------------------------------
template <int MODE>
__device__ uint64_t block_out_exact(const BlockRec& br, const double* __restrict__ w, long long nw, double total,
                                    long long b, uint64_t in, long long& lit)
{
    bool ok = true;
    const uint64_t o = rec_step(br.prefix, in, ok);
    const long long e0 = b * EB, e1 = (e0 + EB < nw) ? e0 + EB : nw;
    if (ok) {
        if (br.tail_from >= NT) return o;
        const long long j0 = e0 + (long long)br.tail_from * EL;
        if (j0 >= e1) return o;
        lit += e1 - j0;
        return literal_run<MODE>(w, nw, total, j0, e1, o);
    }
    lit += (e1 > e0) ? e1 - e0 : 0;
    return literal_run<MODE>(w, nw, total, e0, e1, in);
}

template <int MODE>
__global__ __launch_bounds__(NT) void k_resolve(const double* __restrict__ w, long long nw, int B,
                                                const BlockRec* __restrict__ brec, uint64_t* __restrict__ cin,
                                                const double* __restrict__ ain_w, double* __restrict__ ain_p, DevState* st)
{
    if (st->dead || !st->do_resample || st->flags) return;
    extern __shared__ __attribute__((aligned(16))) char smraw[];
    BlockRec* br = reinterpret_cast<BlockRec*>(smraw);
    __shared__ Rec grec[MAXG];
    __shared__ uint64_t gin[MAXG];
    __shared__ uint64_t final_state;
    const int t = threadIdx.x;
    {   // stage the block records in LDS (16-byte pieces)
        const uint4* src = reinterpret_cast<const uint4*>(brec);
        uint4* dst = reinterpret_cast<uint4*>(smraw);
        const int n16 = B * (int)(sizeof(BlockRec) / 16);
        for (int i = t; i < n16; i += NT) dst[i] = src[i];
    }
    __syncthreads();
    const double total = (MODE == MODE_P) ? b2d(st->total_bits) : 1.0;
    int G = 1;
    while (G * G < B) G++;
    const int NG = (B + G - 1) / G;
    if (t < NG) {
        const int b0 = t * G, b1 = (b0 + G < B) ? b0 + G : B;
        Rec r = (br[b0].tail_from < NT) ? rec_hard(0) : br[b0].prefix;
        for (int b = b0 + 1; b < b1; b++) r = (br[b].tail_from < NT) ? rec_hard(0) : rec_compose(r, br[b].prefix);
        grec[t] = r;
    }
    __syncthreads();
    long long lit = 0;
    if (t == 0) {
        uint64_t s = 0;
        for (int g = 0; g < NG; g++) {
            gin[g] = s;
            if (MODE == MODE_P && g == NG - 1) break;      // the final state of cumsum(prob) is not needed
            bool ok = true;
            uint64_t o = rec_step(grec[g], s, ok);
            if (!ok) {
                o = s;
                const int b0 = g * G, b1 = (b0 + G < B) ? b0 + G : B;
                for (int b = b0; b < b1; b++) o = block_out_exact<MODE>(br[b], w, nw, total, b, o, lit);
            }
            s = o;
        }
        final_state = s;
    }
    __syncthreads();
    if (t < NG) {
        uint64_t s = gin[t];
        const int b0 = t * G, b1 = (b0 + G < B) ? b0 + G : B;
        for (int b = b0; b < b1; b++) {
            cin[b] = s;
            if (b + 1 < b1) s = block_out_exact<MODE>(br[b], w, nw, total, b, s, lit);
        }
    }
    if (lit) atomicAdd((unsigned long long*)&st->stat_literal_terms, (unsigned long long)lit);
    if (MODE == MODE_W) {
        const double tot = b2d(final_state);
        if (t == 0) {
            st->total_bits = final_state;
            if (tot == 0.0) atomicOr(&st->flags, FLAG_ZERO_SUM);       // src/resampling.cpp:8,22,49
            if (!isfinite(tot)) atomicOr(&st->flags, FLAG_NONFINITE);
        }
        for (int b = t; b < B; b += NT) ain_p[b] = ain_w[b] / tot;
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
};

template <int KIND>
__global__ __launch_bounds__(NT) void k_apply(ApplyArgs a, DevState* st)
{
    if (st->dead || !st->do_resample || st->flags) return;
    __shared__ ScanSmem sm;
    __shared__ int Tl[EB];
    __shared__ int first_bad;
    __shared__ int Tbegin;
    const int t = threadIdx.x;
    const long long b0 = (long long)blockIdx.x * EB;
    const double total = b2d(st->total_bits);
    const int call = st->cur_call;
    double v[EL];
    load_terms<MODE_P>(a.w, a.nw, total, b0 + (long long)t * EL, v);
    double ts = 0.0;
#pragma unroll
    for (int k = 0; k < EL; k++) ts += v[k];
    const double h = a.ain_p[blockIdx.x] + block_excl_scan(ts, sm.sh4);
    const Rec mine = chunk_record(v, EL, 1, h, a.lim);
    Rec inc, exc;
    if (t == 0) first_bad = NT;
    scan_records(sm, mine, inc, exc);
    const uint64_t cinb = a.cin[blockIdx.x];
    uint64_t tin = cinb;
    if (t > 0) {
        bool ok = true;
        tin = rec_step(exc, cinb, ok);
        if (!ok) atomicMin(&first_bad, t);
    }
    sm.tin[t] = tin;
    __syncthreads();
    if (first_bad < NT) {
        if (t == 0) {   // serial walk from the last thread whose exact state is known
            uint64_t s = sm.tin[first_bad - 1];
            long long lit = 0;
            for (int tt = first_bad - 1; tt < NT; tt++) {
                sm.tin[tt] = s;
                bool ok = true;
                uint64_t o = rec_step(sm.orig[tt], s, ok);
                if (!ok) { o = literal_run<MODE_P>(a.w, a.nw, total, b0 + (long long)tt * EL, b0 + (long long)(tt + 1) * EL, s); lit += EL; }
                s = o;
            }
            atomicAdd((unsigned long long*)&st->stat_serial_walks, 1ull);
            atomicAdd((unsigned long long*)&st->stat_literal_terms, (unsigned long long)lit);
        }
        __syncthreads();
        tin = sm.tin[t];
    }
    // the reference chain itself, from the exact incoming state
    UniformSrc us;
    us.arr = a.u_base ? a.u_base + (long long)call * a.u_stride : nullptr;
    us.key = a.key; us.call = (uint32_t)call;
    const double Usys = (KIND == 1) ? us(0) : 0.0;
    double c = b2d(tin);
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
    }
    if (KIND == 2) return;                                            // multinomial: k_multinomial searches cum_out
    if (t == 0) {
        int tb = 0;
        if (blockIdx.x > 0) {
            const double cprev = b2d(cinb);
            tb = (KIND == 1) ? count_le_systematic(cprev, a.n, Usys) : count_le_stratified(cprev, a.n, us);
        }
        Tbegin = tb;
    }
    __syncthreads();
    const int Tb = Tbegin, Te = Tl[EB - 1];
    int* anc = a.anc_out ? a.anc_out + (long long)call * a.anc_stride : nullptr;
    const double invN = 1.0 / (double)a.n;
    double acc0 = 0.0, acc1 = 0.0;
    for (int i = Tb + t; i < Te; i += NT) {
        // first local index whose count exceeds i
        int lo = 0, hi = EB - 1;
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if (Tl[mid] > i) hi = mid; else lo = mid + 1;
        }
        const long long src = b0 + lo;
        if (anc) anc[i] = (int)(src + 1);                            // 1-based (src/resampling.cpp:36,62)
        if (a.xdst) {
            const double x0 = a.xsrc[src];
            a.xdst[i] = x0; acc0 += x0 * invN;
            if (a.dim > 1) { const double x1 = a.xsrc[a.xstride + src]; a.xdst[a.xstride + i] = x1; acc1 += x1 * invN; }
        }
        if (a.auxdst) a.auxdst[i] = a.auxsrc[src];
    }
    if (a.se_part) {
        acc0 = block_sum(acc0, sm.sh4);
        if (a.dim > 1) acc1 = block_sum(acc1, sm.sh4);
        if (t == 0) { a.se_part[(long long)blockIdx.x * a.dim] = acc0; if (a.dim > 1) a.se_part[(long long)blockIdx.x * a.dim + 1] = acc1; }
    }
}

// multinomial: inverse CDF on the exact cum_sum (distributional parity only)
__global__ __launch_bounds__(NT) void k_multinomial(const double* __restrict__ cum, long long nw, int n, ApplyArgs a, DevState* st)
{
    if (st->dead || !st->do_resample || st->flags) return;
    __shared__ double sh4[4];
    const int call = st->cur_call;
    UniformSrc us;
    us.arr = a.u_base ? a.u_base + (long long)call * a.u_stride : nullptr;
    us.key = a.key; us.call = (uint32_t)call;
    int* anc = a.anc_out ? a.anc_out + (long long)call * a.anc_stride : nullptr;
    const double invN = 1.0 / (double)n;
    double acc0 = 0.0, acc1 = 0.0;
#pragma unroll
    for (int r = 0; r < EL; r++) {
        const long long i = (long long)blockIdx.x * EB + threadIdx.x + NT * r;
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
        if (threadIdx.x == 0) { a.se_part[(long long)blockIdx.x * a.dim] = acc0; if (a.dim > 1) a.se_part[(long long)blockIdx.x * a.dim + 1] = acc1; }
    }
}

// no resampling at this observation: particles carry over; state estimate is
// sum(particles * weights) with the normalised weights (:238)
__global__ __launch_bounds__(NT) void k_carry(const double* __restrict__ xsrc, double* __restrict__ xdst,
                                              const double* __restrict__ w, long long N, int dim,
                                              double* __restrict__ se_part, const DevState* __restrict__ st)
{
    if (st->dead || st->do_resample) return;
    __shared__ double sh4[4];
    const long long base = (long long)blockIdx.x * EB;
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
    if (threadIdx.x == 0) { se_part[(long long)blockIdx.x * dim] = acc0; if (dim > 1) se_part[(long long)blockIdx.x * dim + 1] = acc1; }
}

// state_est[i] = sum over blocks of the partials written for observation i
// (a partial slot is only written by the kernel that ran for that observation)
__global__ __launch_bounds__(NT) void k_reduce_state_est(const double* __restrict__ se_part, int nblocks, int dim,
                                                         double* __restrict__ state_est)
{
    __shared__ double sh4[4];
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

__global__ void k_reset_state(DevState* st)
{
    st->loglike = 0.0; st->lse_max = 0.0; st->lse_sum = 0.0; st->ess = 0.0; st->total_bits = 0;
    st->do_resample = 0; st->dead = 0; st->flags = 0; st->res_calls = 0; st->cur_call = 0; st->pad0 = 0;
    st->stat_hard_blocks = 0; st->stat_serial_walks = 0; st->stat_literal_terms = 0;
}

}  // namespace bssm
