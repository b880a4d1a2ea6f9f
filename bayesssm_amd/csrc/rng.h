// rng.h -- counter-based random draws for the device filter ("throughput mode").
//
// The reference draws from R's global Mersenne-Twister inside user closures
// (rnorm in transition_fn / init_fn; R::runif / Rcpp::runif in
// src/resampling.cpp:28,55).  A sequential generator cannot feed 2^20 lanes,
// so the device path keys every draw by WHAT it is for instead of WHEN it is
// drawn:   Philox4x32-10( key = seed,  counter = (index, call, purpose, 0) ).
// Results therefore do not depend on launch geometry, GPU count or chain
// placement (mirrors tests/testthat/test-pmmh.R:499-503: 1 core == 2 cores).
// In "parity mode" the same draws are supplied as arrays instead (and this
// header's stream can be dumped with bssm_dump_* to feed the CPU oracle).
//
// Philox4x32-10: Salmon, Moraes, Dror, Shaw, "Parallel random numbers: as easy
// as 1, 2, 3" (SC'11).  Device normals: Box-Muller on the two uniforms of one
// Philox block.  Host (PMMH proposal) normals: inversion (as R's default
// "Inversion" normal.kind does) with Wichura's AS 241 PPND16 rational
// approximation (Appl. Statist. 37 (1988) 477-484).
#pragma once
#include <stdint.h>
#include <math.h>

#if defined(__HIPCC__)
#define BSSM_HD __host__ __device__ __forceinline__
#else
#define BSSM_HD inline
#endif

namespace bssm {

enum : uint32_t { DRAW_INIT = 1, DRAW_TRANS = 2, DRAW_RESAMPLE = 3, DRAW_PROPOSAL = 5, DRAW_ACCEPT = 6, DRAW_MOVE = 7 };

struct u32x4 { uint32_t x, y, z, w; };

BSSM_HD uint32_t mulhi32(uint32_t a, uint32_t b) { return (uint32_t)(((uint64_t)a * (uint64_t)b) >> 32); }

BSSM_HD u32x4 philox4x32_10(u32x4 c, uint32_t k0, uint32_t k1)
{
    const uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u, W0 = 0x9E3779B9u, W1 = 0xBB67AE85u;
#pragma unroll
    for (int r = 0; r < 10; r++) {
        const uint32_t hi0 = mulhi32(M0, c.x), lo0 = M0 * c.x;
        const uint32_t hi1 = mulhi32(M1, c.z), lo1 = M1 * c.z;
        u32x4 n;
        n.x = hi1 ^ c.y ^ k0; n.y = lo1; n.z = hi0 ^ c.w ^ k1; n.w = lo0;
        c = n;
        k0 += W0; k1 += W1;
    }
    return c;
}

// 64 random bits -> uniform strictly inside (0,1): (k + 1/2) * 2^-53, k in [0, 2^53)
BSSM_HD double u01_from_bits(uint32_t lo, uint32_t hi)
{
    const uint64_t b = ((uint64_t)hi << 32) | lo;
    return ((double)(b >> 11) + 0.5) * 0x1.0p-53;
}

// AS 241 PPND16: standard normal quantile, |rel err| ~ 1e-16
BSSM_HD double qnorm_as241(double p)
{
    const double q = p - 0.5;
    if (fabs(q) <= 0.425) {
        const double r = 0.180625 - q * q;
        const double num = (((((((2.5090809287301226727e3 * r + 3.3430575583588128105e4) * r + 6.7265770927008700853e4) * r
                                + 4.5921953931549871457e4) * r + 1.3731693765509461125e4) * r + 1.9715909503065514427e3) * r
                             + 1.3314166789178437745e2) * r + 3.3871328727963666080e0);
        const double den = (((((((5.2264952788528545610e3 * r + 2.8729085735721942674e4) * r + 3.9307895800092710610e4) * r
                                + 2.1213794301586595867e4) * r + 5.3941960214247511077e3) * r + 6.8718700749205790830e2) * r
                             + 4.2313330701600911252e1) * r + 1.0);
        return q * num / den;
    }
    double r = (q < 0) ? p : (1.0 - p);
    r = sqrt(-log(r));
    double val;
    if (r <= 5.0) {
        r -= 1.6;
        const double num = (((((((7.74545014278341407640e-4 * r + 2.27238449892691845833e-2) * r + 2.41780725177450611770e-1) * r
                                + 1.27045825245236838258e0) * r + 3.64784832476320460504e0) * r + 5.76949722146069140550e0) * r
                             + 4.63033784615654529590e0) * r + 1.42343711074968357734e0);
        const double den = (((((((1.05075007164441684324e-9 * r + 5.47593808499534494600e-4) * r + 1.51986665636164571966e-2) * r
                                + 1.48103976427480074590e-1) * r + 6.89767334985100004550e-1) * r + 1.67638483018380384940e0) * r
                             + 2.05319162663775882187e0) * r + 1.0);
        val = num / den;
    } else {
        r -= 5.0;
        const double num = (((((((2.01033439929228813265e-7 * r + 2.71155556874348757815e-5) * r + 1.24266094738807843860e-3) * r
                                + 2.65321895265761230930e-2) * r + 2.96560571828504891230e-1) * r + 1.78482653991729133580e0) * r
                             + 5.46378491116411436990e0) * r + 6.65790464350110377720e0);
        const double den = (((((((2.04426310338993978564e-15 * r + 1.42151175831644588870e-7) * r + 1.84631831751005468180e-5) * r
                                + 7.86869131145613259100e-4) * r + 1.48753612908506148525e-2) * r + 1.36929880922735805310e-1) * r
                             + 5.99832206555887937690e-1) * r + 1.0);
        val = num / den;
    }
    return (q < 0) ? -val : val;
}

struct PhiloxKey { uint32_t k0, k1, stream; };   // key = seed; stream goes into counter word 3

// Two standard normals for the index PAIR (2*pair, 2*pair+1), dimension `dim`,
// of draw-call `call` with the given purpose.  One Philox block = 128 bits = two
// uniforms = one Box-Muller pair: no divergent tail branch, one log + sqrt +
// sincospi per two deviates (the inversion above costs about twice as much per
// deviate on a 64-wide wave because every wave runs both of its branches).
#if defined(__HIPCC__)
// log(x) for x strictly inside (0, 1) and normal (the uniforms above are >= 2^-54): fdlibm's __ieee754_log without its
// special cases -- about half the instructions of the library log, < 1 ulp (tools/micro: 2.2e-16 max relative
// difference to the library log over a sweep).  Box-Muller spends a log per pair; k_step is VALU-bound.
__device__ __forceinline__ double log_unit_interval(double x)
{
    const double ln2_hi = 6.93147180369123816490e-01, ln2_lo = 1.90821492927058770002e-10,
                 Lg1 = 6.666666666666735130e-01, Lg2 = 3.999999999940941908e-01, Lg3 = 2.857142874366239149e-01,
                 Lg4 = 2.222219843214978396e-01, Lg5 = 1.818357216161805012e-01, Lg6 = 1.531383769920937332e-01,
                 Lg7 = 1.479819860511658591e-01;
    const uint64_t b = (uint64_t)__double_as_longlong(x);
    int k = (int)(b >> 52) - 1023;
    const uint64_t m = b & 0x000FFFFFFFFFFFFFull;
    const bool up = m > 0x6A09E667F3BCDull;                 // mantissa above sqrt(2): use m/2 and k + 1, so that f = m - 1 is in [sqrt(2)/2 - 1, sqrt(2) - 1)
    k += up ? 1 : 0;
    const double mm = __longlong_as_double((long long)(m | (up ? 0x3FE0000000000000ull : 0x3FF0000000000000ull)));
    const double f = mm - 1.0;
    const double s = f / (2.0 + f);
    const double z = s * s, w = z * z;
    const double t1 = w * (Lg2 + w * (Lg4 + w * Lg6));
    const double t2 = z * (Lg1 + w * (Lg3 + w * (Lg5 + w * Lg7)));
    const double R = t2 + t1;
    const double hfsq = 0.5 * f * f;
    const double dk = (double)k;
    return dk * ln2_hi - ((hfsq - (s * (hfsq + R) + dk * ln2_lo)) - f);
}

__device__ __forceinline__ void normal_pair(PhiloxKey key, uint32_t purpose, uint32_t call, uint32_t dim, uint32_t pair,
                                            double& z0, double& z1)
{
    u32x4 c; c.x = pair; c.y = call; c.z = purpose | (dim << 8); c.w = key.stream;
    const u32x4 r = philox4x32_10(c, key.k0, key.k1);
    const double u1 = u01_from_bits(r.x, r.y), u2 = u01_from_bits(r.z, r.w);
    const double rad = sqrt(-2.0 * log_unit_interval(u1));
    double sn, cs;
    sincospi(2.0 * u2, &sn, &cs);
    z0 = rad * cs;
    z1 = rad * sn;
}
#endif

// Uniform (0,1) for output index i of resample call `call` (stratified /
// multinomial); systematic uses index 0.
BSSM_HD double resample_uniform(PhiloxKey key, uint32_t call, uint32_t i)
{
    u32x4 c; c.x = i >> 1; c.y = call; c.z = DRAW_RESAMPLE; c.w = key.stream;
    const u32x4 r = philox4x32_10(c, key.k0, key.k1);
    return (i & 1) ? u01_from_bits(r.z, r.w) : u01_from_bits(r.x, r.y);
}

// One (normal, uniform) pair for the resample-move step of particle i at observation `call`.
BSSM_HD void move_draws(PhiloxKey key, uint32_t call, uint32_t i, double& z, double& u)
{
    u32x4 c; c.x = i; c.y = call; c.z = DRAW_MOVE; c.w = key.stream;
    const u32x4 r = philox4x32_10(c, key.k0, key.k1);
    z = qnorm_as241(u01_from_bits(r.x, r.y));
    u = u01_from_bits(r.z, r.w);
}

}  // namespace bssm
