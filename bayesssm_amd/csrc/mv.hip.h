// mv.hip.h -- multivariate linear-Gaussian family (state dimension d <= 8, observation dimension p <= 8) on the device.
//
// The reference's models are R closures over an N x d particle matrix (R/particle_filter_core.R:76-88; its own multi-dimensional
// cases: tests/testthat/test-bootstrap_filter.R:211-230, tests/testthat/test-pmmh.R:619-668 -- a 2-d random walk with a constant
// log-likelihood).  This family covers those and the general linear-Gaussian state-space model:
//   init_fn          x0 = m0 + L0 z                      z ~ N(0, I_d)          (matrix(rnorm(N d), ncol = d), shifted and scaled)
//   transition_fn    x' = A x + b + L z                  L lower triangular (a Cholesky factor of Q)
//   log_likelihood   p == 0: the constant c0             (the reference tests' rep(1, nrow(particles)))
//                    p >  0: sum_k dnorm(y_k, h0_k + (H x)_k, sd_k, log = TRUE)      (independent observation components)
// Only the model evaluation is new: normalisation, log-likelihood, ESS, the resample decision and the exact resampling run in the
// same kernels as every other filter (k_local / k_apply, which here emit ANCESTORS); particles[indices, ] is then a gather of d
// coalesced component rows (k_gather_mv).  Particles are SoA [d][N].
// Arithmetic (fp64, contraction off), in one fixed order of operations (the parity tests restate it on the CPU operation for operation):
//   x'_c = ((b_c + A_c0 x_0) + A_c1 x_1 + ... ) + L_c0 z_0 + ... + L_cc z_c        (left to right)
#pragma once
#include "kernels.hip.h"

namespace bssm {

constexpr int MVD = 8;            // largest state / observation dimension
// packed parameter block (doubles): d, p, m0[d], L0[d d], A[d d], b[d], L[d d], c0, H[p d], h0[p], sd[p], log(sd)[p]
struct MvPar {
    const double* P; int d, p;
    __host__ __device__ int o_m0() const { return 2; }
    __host__ __device__ int o_L0() const { return 2 + d; }
    __host__ __device__ int o_A() const { return o_L0() + d * d; }
    __host__ __device__ int o_b() const { return o_A() + d * d; }
    __host__ __device__ int o_L() const { return o_b() + d; }
    __host__ __device__ int o_c0() const { return o_L() + d * d; }
    __host__ __device__ int o_H() const { return o_c0() + 1; }
    __host__ __device__ int o_h0() const { return o_H() + p * d; }
    __host__ __device__ int o_sd() const { return o_h0() + p; }
    __host__ __device__ int o_lsd() const { return o_sd() + p; }
    __host__ __device__ int size() const { return o_lsd() + p; }
};

struct MvNoise { const double* arr; PhiloxKey key; uint32_t purpose, call; };      // arr: [d][N] injected draws of this call, or nullptr

// init_fn (R/particle_filter_core.R:76-88) + the t = 0 state estimate partials (:109-112)
__global__ __launch_bounds__(NT) void k_init_mv(double* __restrict__ x, long long N, MvPar mp, MvNoise ns, double* __restrict__ se_part /* [B][d] */)
{
    __shared__ double sh4[NWV];
    const int d = mp.d;
    const long long base = (long long)blockIdx.x * EB;
    const double invN = 1.0 / (double)N;
    double acc[MVD];
#pragma unroll
    for (int c = 0; c < MVD; c++) acc[c] = 0.0;
#pragma unroll 1
    for (int r = 0; r < EL; r++) {
        const long long j = base + threadIdx.x + NT * r;
        if (j < N) {
            double z[MVD];
#pragma unroll
            for (int c = 0; c < MVD; c++) {
                z[c] = 0.0;
                if (c < d) {
                    if (ns.arr) z[c] = ns.arr[(long long)c * N + j];
                    else { double z0, z1; normal_pair(ns.key, ns.purpose, ns.call, (uint32_t)c, (uint32_t)(j >> 1), z0, z1); z[c] = (j & 1) ? z1 : z0; }
                }
            }
#pragma unroll
            for (int c = 0; c < MVD; c++) {
                if (c < d) {
                    double v = mp.P[mp.o_m0() + c];
#pragma unroll
                    for (int k = 0; k < MVD; k++) if (k <= c) v = v + mp.P[mp.o_L0() + c * d + k] * z[k];
                    x[(long long)c * N + j] = v;
                    acc[c] += v * invN;
                }
            }
        }
    }
    for (int c = 0; c < d; c++) { const double s = block_sum(acc[c], sh4); if (threadIdx.x == 0) se_part[(long long)blockIdx.x * d + c] = s; }
}

// transition_fn and / or weight_fn (R/particle_filter_core.R:127,177-183) with the block partials of the log-sum-exp, as k_step
template <bool TRANS, bool WEIGHT>
__global__ __launch_bounds__(NTS) void k_step_mv(double* __restrict__ x, double* __restrict__ lw, long long N, MvPar mp, const double* __restrict__ yrow /* [p] */,
                                                 MvNoise ns, double* __restrict__ pm, double* __restrict__ ps, double* __restrict__ pq,
                                                 unsigned long long* __restrict__ gmax)
{
    __shared__ double sh[2 * (NTS / 64)];
    const int d = mp.d, p = mp.p;
    const long long j = (long long)blockIdx.x * EB + 2 * (long long)threadIdx.x;
    double l0 = -INFINITY, l1 = -INFINITY;
    if (j < N) {
        const bool two = (j + 1 < N);
        double x0[MVD], x1[MVD];
#pragma unroll
        for (int c = 0; c < MVD; c++) { x0[c] = 0.0; x1[c] = 0.0; if (c < d) { x0[c] = x[(long long)c * N + j]; if (two) x1[c] = x[(long long)c * N + j + 1]; } }
        if (TRANS) {
            double z0[MVD], z1[MVD];
#pragma unroll
            for (int c = 0; c < MVD; c++) {
                z0[c] = 0.0; z1[c] = 0.0;
                if (c < d) {
                    if (ns.arr) { z0[c] = ns.arr[(long long)c * N + j]; if (two) z1[c] = ns.arr[(long long)c * N + j + 1]; }
                    else normal_pair(ns.key, ns.purpose, ns.call, (uint32_t)c, (uint32_t)(j >> 1), z0[c], z1[c]);
                }
            }
            double n0[MVD], n1[MVD];
#pragma unroll
            for (int c = 0; c < MVD; c++) {
                n0[c] = 0.0; n1[c] = 0.0;
                if (c < d) {
                    double a0 = mp.P[mp.o_b() + c], a1 = a0;
#pragma unroll
                    for (int k = 0; k < MVD; k++) if (k < d) { const double A = mp.P[mp.o_A() + c * d + k]; a0 = a0 + A * x0[k]; a1 = a1 + A * x1[k]; }
#pragma unroll
                    for (int k = 0; k < MVD; k++) if (k <= c) { const double L = mp.P[mp.o_L() + c * d + k]; a0 = a0 + L * z0[k]; a1 = a1 + L * z1[k]; }
                    n0[c] = a0; n1[c] = a1;
                }
            }
#pragma unroll
            for (int c = 0; c < MVD; c++) if (c < d) { x0[c] = n0[c]; x1[c] = n1[c]; x[(long long)c * N + j] = n0[c]; if (two) x[(long long)c * N + j + 1] = n1[c]; }
        }
        if (WEIGHT) {
            if (p == 0) { l0 = mp.P[mp.o_c0()]; l1 = l0; }
            else {
                l0 = 0.0; l1 = 0.0;
#pragma unroll
                for (int k = 0; k < MVD; k++) {
                    if (k < p) {
                        double m0 = mp.P[mp.o_h0() + k], m1 = m0;
#pragma unroll
                        for (int c = 0; c < MVD; c++) if (c < d) { const double H = mp.P[mp.o_H() + k * d + c]; m0 = m0 + H * x0[c]; m1 = m1 + H * x1[c]; }
                        const double sd = mp.P[mp.o_sd() + k], lsd = mp.P[mp.o_lsd() + k];
                        l0 = l0 + r_dnorm_log(yrow[k], m0, sd, lsd);
                        l1 = l1 + r_dnorm_log(yrow[k], m1, sd, lsd);
                    }
                }
            }
            if (!two) l1 = -INFINITY;
            lw[j] = l0; if (two) lw[j + 1] = l1;
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
        if (threadIdx.x == 0) { pm[blockIdx.x] = bm; ps[blockIdx.x] = s; pq[blockIdx.x] = q; if (gmax) atomicMax(gmax + (blockIdx.x % GM_SLOTS) * GM_STRIDE, f64_key(bm)); }
    }
}

// particles[indices, ] (R/resampling.R:40,60) for the ancestors k_apply emitted, d component rows; state estimate partials (:237-241)
__global__ __launch_bounds__(NT) void k_gather_mv(const int* __restrict__ anc_base, long long anc_stride, long long N, int d,
                                                  const double* __restrict__ xsrc, double* __restrict__ xdst, double* __restrict__ se_part, DevState* st)
{
    __shared__ double sh4[NWV];
    if (st->dead || !st->do_resample || st->flags) return;
    const int* anc = anc_base + (long long)st->cur_call * anc_stride;
    const double invN = 1.0 / (double)N;
    double acc[MVD];
#pragma unroll
    for (int c = 0; c < MVD; c++) acc[c] = 0.0;
#pragma unroll 1
    for (int r = 0; r < EL; r++) {
        const long long i = (long long)blockIdx.x * EB + threadIdx.x + NT * r;
        if (i < N) {
            const long long src = anc[i] - 1;
#pragma unroll
            for (int c = 0; c < MVD; c++) if (c < d) { const double v = xsrc[(long long)c * N + src]; xdst[(long long)c * N + i] = v; acc[c] += v * invN; }
        }
    }
    for (int c = 0; c < d; c++) { const double s = block_sum(acc[c], sh4); if (threadIdx.x == 0) se_part[(long long)blockIdx.x * d + c] = s; }
}

// no resampling at this observation: carry over, state estimate = colSums(particles * weights) (:238)
__global__ __launch_bounds__(NT) void k_carry_mv(const double* __restrict__ xsrc, double* __restrict__ xdst, const double* __restrict__ w, long long N, int d,
                                                 double* __restrict__ se_part, const DevState* __restrict__ st)
{
    __shared__ double sh4[NWV];
    if (st->dead || st->do_resample) return;
    double acc[MVD];
#pragma unroll
    for (int c = 0; c < MVD; c++) acc[c] = 0.0;
#pragma unroll 1
    for (int r = 0; r < EL; r++) {
        const long long j = (long long)blockIdx.x * EB + threadIdx.x + NT * r;
        if (j < N) {
            const double wj = w[j];
#pragma unroll
            for (int c = 0; c < MVD; c++) if (c < d) { const double v = xsrc[(long long)c * N + j]; xdst[(long long)c * N + j] = v; acc[c] += v * wj; }
        }
    }
    for (int c = 0; c < d; c++) { const double s = block_sum(acc[c], sh4); if (threadIdx.x == 0) se_part[(long long)blockIdx.x * d + c] = s; }
}

__global__ void k_dump_normals_mv(PhiloxKey key, uint32_t purpose, uint32_t call, long long N, int d, double* __restrict__ out /* [d][N] */)
{
    const long long pair = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long j = 2 * pair;
    if (j >= N) return;
    for (int c = 0; c < d; c++) {
        double z0, z1;
        normal_pair(key, purpose, call, (uint32_t)c, (uint32_t)pair, z0, z1);
        out[(long long)c * N + j] = z0;
        if (j + 1 < N) out[(long long)c * N + j + 1] = z1;
    }
}

}  // namespace bssm
