// seqsum.h -- parallel evaluation of a SEQUENTIAL fp64 running sum, bit for bit.
//
// Why: the reference resamplers (src/resampling.cpp:20,24-25 / :47,51-52)
// compute  total = sum(w),  prob = w/total,  cum = cumsum(prob)  with plain
// left-to-right double additions, then compare cum[j] < u_i.  Ancestor indices
// are only bit-exact if every cum[j] is the value that left-to-right chain
// produces.  A re-associated (tree) scan differs in the last bits and flips
// ancestors.  This header lets many lanes evaluate that chain exactly.
//
// Idea.  All terms are >= 0, so the running sum c is monotone.  While c stays
// inside one binade [2^e, 2^(e+1)) the update c <- fl(c + p) moves c by an
// integer number of ulps that depends on c only through the PARITY of its last
// mantissa bit (round-half-even ties); when the sum crosses into the next
// binade (ulp doubles) the rounded result depends on the last TWO bits.  So a
// chunk of terms acts on the incoming state as
//      out_bits = o[s] + (4q >> ncross),   in_bits = base + 4q + s, s in 0..3
// where o[s] is obtained by literally running the chunk in fp64 from the four
// HYPOTHETICAL starts base+s (base = an approximate prefix, low 2 bits
// cleared).  That map ("record") is valid when the true incoming state is in
// the same binade as base and the true trajectory changes binade at the same
// term as the hypothetical one; both are guaranteed when no state on the
// hypothetical trajectory lies within `lim` ulps of a power of two and the
// true incoming state is within the record's [lo,hi] ulp window of base.
// Records compose associatively (PURE.PURE=PURE, PURE.X1=X1.PURE=X1), so a
// parallel scan of records + one walk with the exact state reproduces the
// sequential chain.  Anything outside these conditions is marked HARD and is
// re-run literally (in order, in fp64) with the exact incoming state: always
// correct, merely slower -- correctness never rests on the error estimate.
//
// Compiled for gfx950 by hipcc (kernels) and for the host by g++ (the
// algorithm harness in tests/harness/, test infrastructure).  Build both with
// -ffp-contract=off; only IEEE add on doubles and integer ops are used.
#pragma once
#include <stdint.h>
#include <string.h>

#if defined(__HIPCC__)
#define BSSM_HD __host__ __device__ __forceinline__
#else
#define BSSM_HD inline
#endif

namespace bssm {

BSSM_HD uint64_t d2b(double x) { uint64_t b; memcpy(&b, &x, 8); return b; }
BSSM_HD double b2d(uint64_t b) { double x; memcpy(&x, &b, 8); return x; }

enum : int32_t { REC_PURE = 0, REC_X1 = 1, REC_ABS = 2, REC_HARD = 3 };

// 56-byte record: the action of a run of terms on the running-sum state.
struct Rec {
    uint64_t base;   // hypothetical incoming bit pattern, low two bits clear
    uint64_t o[4];   // outgoing bit pattern for incoming base+s
    int32_t kind;    // REC_*
    int32_t lo, hi;  // valid window of (true_in - base), in ulps of base's binade
    int32_t pad;
};

BSSM_HD Rec rec_hard(uint64_t base)
{
    Rec r; r.base = base; r.o[0] = r.o[1] = r.o[2] = r.o[3] = 0; r.kind = REC_HARD; r.lo = 0; r.hi = -1; r.pad = 0;
    return r;
}

BSSM_HD Rec rec_abs(uint64_t out)
{   // constant map: the incoming state is known exactly, the outgoing one is `out`
    Rec r; r.base = 0; r.o[0] = r.o[1] = r.o[2] = r.o[3] = out; r.kind = REC_ABS; r.lo = 0; r.hi = 0; r.pad = 0;
    return r;
}

BSSM_HD Rec rec_identity(uint64_t base)
{   // empty run of terms: out = in, for any incoming state
    Rec r; r.base = base & ~3ull; r.kind = REC_PURE; r.lo = -(1 << 30); r.hi = (1 << 30); r.pad = 0;
    for (int s = 0; s < 4; s++) r.o[s] = r.base + (uint64_t)s;
    return r;
}

// Is bit pattern b within `near` ulps of a binade boundary (or in a range
// where the record logic is not used: zero/denormal/first binade/inf/nan)?
BSSM_HD bool near_pow2(uint64_t b, uint64_t near)
{
    const uint64_t ef = b >> 52;
    const uint64_t mant = b & ((1ull << 52) - 1);
    return ef <= 1 || ef >= 0x7FE || mant <= near || mant >= (1ull << 52) - near;
}

// Ulp window half-width used for hypotheses over n terms; the window only
// affects speed (see header), not correctness.
BSSM_HD int32_t rec_window(long long n)
{
    // n + 1024 ulps.  Every add of the sequential sum is off by at most half an ulp of the final magnitude, so the
    // sequential prefix lies within n/2 ulps of the true one and the approximate (tree / log-sum-exp) prefix within a
    // few more: the window covers the WORST case.  Random weights only drift ~sqrt(n) ulps, but runs of EQUAL weights
    // (integer-valued states, e.g. the SIR model: a few dozen distinct weights among 2^18 particles) round the same
    // way add after add and drift linearly -- with a sqrt(n)-sized window those blocks all fell back to the in-order
    // pass.  The price of a wide window is only that lanes within `window` ulps of a power of two are HARD
    // (a 2^-30 sliver of the range at n = 2^22).
    const long long w = n + 1024;
    return (int32_t)(w > (1ll << 28) ? (1ll << 28) : w);
}

// Record of the L terms v[0..L) (stride vs) for hypothetical incoming value h.
// h == 0 exactly means "every earlier term is exactly zero": the incoming
// state is known (+0) and the chunk is run literally (REC_ABS).
BSSM_HD Rec chunk_record(const double* v, int L, int vs, double h, int32_t lim)
{
    Rec r; r.pad = 0;
    if (h == 0.0) {
        double c = 0.0;
        for (int k = 0; k < L; k++) c = c + v[k * vs];
        r.base = 0; r.kind = REC_ABS; r.lo = 0; r.hi = 0;
        r.o[0] = r.o[1] = r.o[2] = r.o[3] = d2b(c);
        return r;
    }
    const uint64_t hb = d2b(h) & ~3ull;
    {   // a chunk of zeros (the padding past the last weight, or truly zero weights) leaves ANY state as it is: c + 0.0 == c.
        // Without this, the padding lanes of a partly filled block sit next to cum == 1.0 and would all be HARD.
        bool allzero = true;
        for (int k = 0; k < L; k++) allzero = allzero && (v[k * vs] == 0.0);
        if (allzero) { r.base = hb; r.kind = REC_PURE; r.lo = -lim; r.hi = lim; r.o[0] = hb; r.o[1] = hb + 1; r.o[2] = hb + 2; r.o[3] = hb + 3; return r; }
    }
    const uint64_t near = (uint64_t)lim + 8;
    bool hard = near_pow2(hb, near);
    double c0 = b2d(hb), c1 = b2d(hb + 1), c2 = b2d(hb + 2), c3 = b2d(hb + 3);
    uint64_t eprev = hb >> 52, bprev = hb;
    int ncross = 0;
    for (int k = 0; k < L; k++) {
        const double p = v[k * vs];
        c0 = c0 + p; c1 = c1 + p; c2 = c2 + p; c3 = c3 + p;
        const uint64_t b0 = d2b(c0);
        const uint64_t e = b0 >> 52;
        if (e != eprev) {
            // binade change at this term: the state before must not be near the
            // upper boundary, the state after not near the lower one, and the
            // jump must be exactly one binade
            hard = hard || (e - eprev != 1) || near_pow2(bprev, near) || near_pow2(b0, near);
            ncross += 1;
            eprev = e;
        }
        bprev = b0;
    }
    hard = hard || near_pow2(bprev, near) || ncross > 1;
    if (hard) return rec_hard(hb);
    r.base = hb; r.o[0] = d2b(c0); r.o[1] = d2b(c1); r.o[2] = d2b(c2); r.o[3] = d2b(c3);
    r.kind = ncross ? REC_X1 : REC_PURE;
    r.lo = -lim; r.hi = lim;
    return r;
}

// Fixed-length variant for the kernels: the loop is fully unrolled so that the
// terms stay in registers (a runtime-length loop would put v[] in scratch).
template <int L>
BSSM_HD Rec chunk_record_fixed(const double (&v)[L], double h, int32_t lim)
{
    Rec r; r.pad = 0;
    if (h == 0.0) {
        double c = 0.0;
#pragma unroll
        for (int k = 0; k < L; k++) c = c + v[k];
        r.base = 0; r.kind = REC_ABS; r.lo = 0; r.hi = 0;
        r.o[0] = r.o[1] = r.o[2] = r.o[3] = d2b(c);
        return r;
    }
    const uint64_t hb = d2b(h) & ~3ull;
    {   // all-zero chunk: identity for any state (see chunk_record)
        bool allzero = true;
#pragma unroll
        for (int k = 0; k < L; k++) allzero = allzero && (v[k] == 0.0);
        if (allzero) { r.base = hb; r.kind = REC_PURE; r.lo = -lim; r.hi = lim; r.o[0] = hb; r.o[1] = hb + 1; r.o[2] = hb + 2; r.o[3] = hb + 3; return r; }
    }
    const uint64_t near = (uint64_t)lim + 8;
    bool hard = near_pow2(hb, near);
    double c0 = b2d(hb), c1 = b2d(hb + 1), c2 = b2d(hb + 2), c3 = b2d(hb + 3);
    uint64_t eprev = hb >> 52, bprev = hb;
    int ncross = 0;
#pragma unroll
    for (int k = 0; k < L; k++) {
        const double p = v[k];
        c0 = c0 + p; c1 = c1 + p; c2 = c2 + p; c3 = c3 + p;
        const uint64_t b0 = d2b(c0);
        const uint64_t e = b0 >> 52;
        if (e != eprev) {
            hard = hard || (e - eprev != 1) || near_pow2(bprev, near) || near_pow2(b0, near);
            ncross += 1;
            eprev = e;
        }
        bprev = b0;
    }
    hard = hard || near_pow2(bprev, near) || ncross > 1;
    if (hard) return rec_hard(hb);
    r.base = hb; r.o[0] = d2b(c0); r.o[1] = d2b(c1); r.o[2] = d2b(c2); r.o[3] = d2b(c3);
    r.kind = ncross ? REC_X1 : REC_PURE;
    r.lo = -lim; r.hi = lim;
    return r;
}

// Apply a record to the exact incoming bit pattern.  ok=false => the record
// does not cover this state; the caller must run the terms literally.
BSSM_HD uint64_t rec_step(const Rec& r, uint64_t in, bool& ok)
{
    if (r.kind == REC_ABS) return r.o[0];
    if (r.kind == REC_HARD) { ok = false; return 0; }
    const int64_t diff = (int64_t)(in - r.base);
    if (diff < (int64_t)r.lo || diff > (int64_t)r.hi) { ok = false; return 0; }
    const int64_t s = diff & 3;
    const int64_t q4 = diff - s;                    // multiple of 4
    // selects, not r.o[s]: a runtime-indexed member array would live in scratch on the GPU
    const uint64_t os = (s == 0) ? r.o[0] : (s == 1) ? r.o[1] : (s == 2) ? r.o[2] : r.o[3];
    return os + (uint64_t)(r.kind == REC_PURE ? q4 : (q4 >> 1));
}

// Composition: first f, then g.
BSSM_HD Rec rec_compose(const Rec& f, const Rec& g)
{
    if (f.kind == REC_HARD || g.kind == REC_HARD) return rec_hard(f.base);
    Rec r; r.pad = 0; r.base = f.base;
    if (g.kind == REC_ABS) {          // g starts from exact zero => so did f
        r = g; r.base = f.base; r.lo = f.lo; r.hi = f.hi;
        if (f.kind == REC_ABS) { r.lo = 0; r.hi = 0; }
        return r;
    }
    if (f.kind == REC_ABS) {
        bool ok = true;
        const uint64_t o = rec_step(g, f.o[0], ok);
        if (!ok) return rec_hard(f.base);
        r.kind = REC_ABS; r.lo = 0; r.hi = 0; r.o[0] = r.o[1] = r.o[2] = r.o[3] = o;
        return r;
    }
    if (f.kind == REC_X1 && g.kind == REC_X1) return rec_hard(f.base);
    // window of f-incoming offsets d for which g's entrance offset
    // m_s + shift_f(d) stays inside [g.lo, g.hi]; 8 ulps of slack for s and rounding
    int64_t mmin = 0, mmax = 0;
    for (int s = 0; s < 4; s++) {
        const int64_t m = (int64_t)(f.o[s] - g.base);
        if (s == 0 || m < mmin) mmin = m;
        if (s == 0 || m > mmax) mmax = m;
    }
    int64_t lo = (int64_t)g.lo - mmin + 8, hi = (int64_t)g.hi - mmax - 8;
    if (f.kind == REC_X1) { lo *= 2; hi *= 2; }
    if (lo < f.lo) lo = f.lo;
    if (hi > f.hi) hi = f.hi;
    if (lo > 0 || hi < 3) return rec_hard(f.base);   // must at least cover the 4 hypotheses
    bool ok = true;
    for (int s = 0; s < 4; s++) r.o[s] = rec_step(g, f.o[s], ok);
    if (!ok) return rec_hard(f.base);
    r.kind = (f.kind == REC_X1 || g.kind == REC_X1) ? REC_X1 : REC_PURE;
    r.lo = (int32_t)lo; r.hi = (int32_t)hi;
    return r;
}

// Literal sequential run (the reference arithmetic itself).
BSSM_HD uint64_t run_literal(const double* v, int L, int vs, uint64_t in)
{
    double c = b2d(in);
    for (int k = 0; k < L; k++) c = c + v[k * vs];
    return d2b(c);
}

// ---------------------------------------------------------------------------
// Output counting for systematic / stratified resampling.
//
// src/resampling.cpp:28-37 / :55-63:  u_i = (i + U_i) / n  (add, then divide,
// both rounded), non-decreasing in i; the walk assigns output i to the first j
// with !(cum[j] < u_i), clamped to the last weight.  Hence element j owns the
// outputs  T(cum[j-1]) <= i < T(cum[j])  with  T(c) = #{i in [0,n) : u_i <= c}
// (and the last element owns everything up to n).
// ---------------------------------------------------------------------------

// the reference's own predicate around the candidate (rare: c n - U within n 2^-49 of an integer); kept out of line so
// that the callers' hot instruction stream stays short
#if defined(__HIPCC__)
__host__ __device__ __attribute__((noinline))
#endif
inline int32_t count_le_systematic_exact(double c, int32_t n, double U, double g)
{
    const double dn = (double)n;
    int64_t t = (g < 0) ? 0 : (g >= dn ? (int64_t)n : (int64_t)g + 1);
    if (t > n) t = n;
    // pred(i): u_i <= c, true for small i, false for large i
    while (t < n && !((((double)t + U) / dn) > c)) t++;
    while (t > 0 && ((((double)(t - 1) + U) / dn) > c)) t--;
    return (int32_t)t;
}

// systematic: U scalar
BSSM_HD int32_t count_le_systematic(double c, int32_t n, double U)
{
    const double dn = (double)n;
    double g = c * dn - U;
    {   // Fast path.  u_i = (i+U)/n (1+e), |e| <= 2^-52, and g = c n - U carries an error <= n 2^-52, so
        // "u_i <= c" is decided by "i <= g" whenever g is further than n 2^-49 from an integer.
        // That covers every i in [0, n), so the count is floor(g) + 1 clamped to [0, n] -- also at the two ends.
        const double delta = dn * 0x1.0p-49;
        const double fl = floor(g), fr = g - fl;
        if (fr > delta && fr < 1.0 - delta) return fl < 0.0 ? 0 : (fl >= dn ? n : (int32_t)fl + 1);
    }
    return count_le_systematic_exact(c, n, U, g);
}

template <class UF>
#if defined(__HIPCC__)
__host__ __device__ __attribute__((noinline))
#endif
inline int32_t count_le_stratified_exact(double c, int32_t n, const UF& U, double g)
{
    const double dn = (double)n;
    int64_t t = (g < 0) ? 0 : (g >= dn ? (int64_t)n : (int64_t)g);
    if (t > n) t = n;
    while (t < n && !((((double)t + U((int32_t)t)) / dn) > c)) t++;
    while (t > 0 && ((((double)(t - 1) + U((int32_t)(t - 1))) / dn) > c)) t--;
    return (int32_t)t;
}

// stratified: U(i) per output (array in parity mode, generator otherwise)
template <class UF>
BSSM_HD int32_t count_le_stratified(double c, int32_t n, const UF& U)
{
    const double dn = (double)n;
    double g = c * dn;
    {   // Fast path: outputs below floor(g) are certainly <= c, those above certainly not (U in [0,1));
        // only output floor(g) needs the reference expression itself.
        const double delta = dn * 0x1.0p-49;
        const double fl = floor(g), fr = g - fl;
        if (fr > delta && fr < 1.0 - delta) {
            if (fl < 0.0) return 0;
            if (fl >= dn) return n;
            const int32_t i0 = (int32_t)fl;
            return i0 + ((((double)i0 + U(i0)) / dn) > c ? 0 : 1);
        }
    }
    return count_le_stratified_exact(c, n, U, g);
}

struct UniformArray {
    const double* p;
    BSSM_HD double operator()(int32_t i) const { return p[i]; }
};

}  // namespace bssm
