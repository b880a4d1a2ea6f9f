// Dev tool: cycle costs of the primitives the scan kernels are made of (clock64 deltas, one workgroup of 256).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <math.h>
#include "../../bayesssm_amd/csrc/kernels.hip.h"
using namespace bssm;

__global__ void k(long long* out, const double* g, double* sink, int reps)
{
    __shared__ double lds[4096];
    __shared__ double sh4[16];
    const int t = threadIdx.x;
    double x = g[t];
    long long c0, c1;
    int slot = 0;
#define MEASURE(body) { __syncthreads(); c0 = clock64(); for (int r = 0; r < reps; r++) { body; } c1 = clock64(); if (t == 0) out[slot] = (c1 - c0) / reps; slot++; }
    MEASURE(__syncthreads())                                                          // 0 barrier
    MEASURE(x = wave_incl_sum(x))                                                     // 1 DPP inclusive scan f64
    MEASURE(x = block_excl_scan(x, sh4))                                              // 2 block excl scan (2 barriers)
    MEASURE(lds[t] = x; __syncthreads(); x = lds[(t + 17) & 255] + 1.0; __syncthreads()) // 3 LDS write+barrier+read+barrier
    MEASURE(x = exp(x * 1e-3) )                                                       // 4 exp
    MEASURE(x = x / (1.0 + x * 1e-9))                                                 // 5 division
    MEASURE(x = log(fabs(x) + 1.5))                                                   // 6 log
    MEASURE(x = g[(t * 64 + (int)(x * 1e-30) + r * 4096) & 0xFFFFF] + x)              // 7 dependent global load (L2/MALL)
    MEASURE(x = block_sum(x, sh4) * 1e-3)                                             // 8 block_sum
    { Pure p; p.d0 = (int64_t)x; p.d1 = p.d0 + 1; int seg, nb; Pure last; __shared__ SegSmem sm;
      MEASURE(p = seg_excl_scan<4>(sm, p, (t & 63) == 17, seg, nb, last); p.d1 += seg) // 9 segmented record scan
      x += (double)p.d0; }
    { double v[8]; for (int k2 = 0; k2 < 8; k2++) v[k2] = x * 1e-6 + k2 * 1e-7; Rec rr;
      MEASURE(rr = chunk_record_fixed<8>(v, 0.4 + x * 1e-12, 1000); v[0] += (double)(rr.o[0] & 1) * 1e-9) // 10 chunk record
      x += v[0]; }
    { double u1, u2; PhiloxKey key{1, 2, 3};
      MEASURE(normal_pair(key, 2, r, 0, t, u1, u2); x += u1 * 1e-9 + u2 * 1e-9) }     // 11 philox + box-muller pair
    sink[t] = x;
}


// dependent-op latencies and the boundary chain, one wave
__global__ void k2(long long* out, const uint64_t* g, uint64_t* sink, int reps)
{
    __shared__ WalkFn wf[65];
    __shared__ uint64_t sseq[65];
    const int t = threadIdx.x;
    if (t >= 64) return;
    uint64_t x = g[t] + 12345;
    uint32_t y = (uint32_t)g[t] + 77;
    long long c0, c1; int slot = 0;
#define M2(body, div) { c0 = clock64(); for (int r = 0; r < reps; r++) { body; } c1 = clock64(); if (t == 0) out[slot] = (c1 - c0) * 10 / (reps * (div)); slot++; }
    M2(_Pragma("unroll") for (int q = 0; q < 32; q++) y = y * 3u + 1u;, 32)                       // 0 mad u32
    M2(_Pragma("unroll") for (int q = 0; q < 32; q++) y = (y ^ (y >> 3)) + 5u;, 32)               // 1 shift,xor,add u32 (3 ops)
    M2(_Pragma("unroll") for (int q = 0; q < 32; q++) x = x + (x >> 7);, 32)                       // 2 shift64 + add64
    M2(_Pragma("unroll") for (int q = 0; q < 32; q++) x = x + 0x123456789ull;, 32)                 // 3 add64
    M2(_Pragma("unroll") for (int q = 0; q < 32; q++) x = (x & 1) ? x + 3 : x + 0x55;, 32)         // 4 and,cmp,cndmask x2, add64
    M2(_Pragma("unroll") for (int q = 0; q < 32; q++) y = (y & 1) ? y + 3 : y + 0x55;, 32)         // 5 same in u32
    { StepFn f; f.K0 = 4; f.K1 = 9; f.K2 = 14; f.K3 = 19; f.sref = 0; f.wlo = -(1ll << 60); f.whi = 1ll << 60; f.sh = 0; f.mode = 0;
      for (int j = t; j < 65; j += 64) wf[j] = walkfn_from(f, true);
      __builtin_amdgcn_wave_barrier();
      uint64_t s = x & 0xffff;
      M2(walk_chain(wf, sseq, 16, s), 16)                                                          // 6 walk_chain per link
      x += s; }
    sink[t] = x + y;
}


// the boundary chain executed ONCE in a freshly launched kernel (cold instruction cache), as the resolver does
__global__ void k3(long long* out, const uint64_t* g, uint64_t* sink, int nlinks)
{
    __shared__ WalkFn wf[65];
    __shared__ uint64_t sseq[65];
    const int t = threadIdx.x;
    if (t >= 64) return;
    StepFn f; f.K0 = 4; f.K1 = 9; f.K2 = 14; f.K3 = 19; f.sref = 0; f.wlo = -(1ll << 60); f.whi = 1ll << 60; f.sh = 0; f.mode = 0;
    for (int j = t; j < 65; j += 64) wf[j] = walkfn_from(f, true);
    __builtin_amdgcn_wave_barrier();
    uint64_t s = g[t] & 0xffff;
    long long c0 = clock64();
    walk_chain(wf, sseq, nlinks, s);
    long long c1 = clock64();
    walk_chain(wf, sseq, nlinks, s);
    long long c2 = clock64();
    if (t == 0) { out[0] = c1 - c0; out[1] = c2 - c1; }
    sink[t] = s;
}


// the chain while the other waves of a 1024-thread workgroup wait at the barrier (as in k_resolve)
__global__ __launch_bounds__(1024) void k4(long long* out, const uint64_t* g, uint64_t* sink, int nlinks, int mode)
{
    __shared__ WalkFn wf[65];
    __shared__ uint64_t sseq[65];
    const int t = threadIdx.x;
    StepFn f; f.K0 = 4; f.K1 = 9; f.K2 = 14; f.K3 = 19; f.sref = 0; f.wlo = -(1ll << 60); f.whi = 1ll << 60; f.sh = 0; f.mode = 0;
    for (int j = t; j < 65; j += 1024) wf[j] = walkfn_from(f, true);
    __syncthreads();
    uint64_t s = g[t & 63] & 0xffff;
    long long acc = 0;
    for (int r = 0; r < 20; r++) {
        if (mode == 0) {                 // wave 0 walks, the rest wait at the barrier
            if (t < 64) { long long c0 = clock64(); walk_chain(wf, sseq, nlinks, s); acc += clock64() - c0; }
        } else {                         // every wave walks (redundantly): nobody waits
            long long c0 = clock64(); uint64_t s2 = s; walk_chain(wf, sseq, nlinks, s2); acc += clock64() - c0; s = s2;
        }
        __syncthreads();
    }
    if (t == 0) out[0] = acc / 20;
    sink[t & 63] = s;
}


// candidates for the Box-Muller transcendental part (u in (0,1)): fdlibm-style log without special cases, sin/cos(2 pi u) by
// quadrant reduction + the fdlibm kernels
__device__ __forceinline__ double fast_log01(double x)
{
    const double ln2_hi = 6.93147180369123816490e-01, ln2_lo = 1.90821492927058770002e-10,
                 Lg1 = 6.666666666666735130e-01, Lg2 = 3.999999999940941908e-01, Lg3 = 2.857142874366239149e-01,
                 Lg4 = 2.222219843214978396e-01, Lg5 = 1.818357216161805012e-01, Lg6 = 1.531383769920937332e-01,
                 Lg7 = 1.479819860511658591e-01;
    uint64_t b = (uint64_t)__double_as_longlong(x);
    int k = (int)(b >> 52) - 1023;
    uint64_t m = b & 0x000FFFFFFFFFFFFFull;
    // mantissa in [sqrt(2)/2, sqrt(2))
    const bool up = m > 0x6A09E667F3BCDull;
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
__device__ __forceinline__ void fast_sincos2pi(double u, double& sn, double& cs)
{
    // angle = 2 pi u, u in (0,1): q = round(4u) quarter turns, r = u - q/4 in [-1/8, 1/8], t = 2 pi r in [-pi/4, pi/4]
    const double S1 = -1.66666666666666324348e-01, S2 = 8.33333333332248946124e-03, S3 = -1.98412698298579493134e-04,
                 S4 = 2.75573137070700676789e-06, S5 = -2.50507602534068634195e-08, S6 = 1.58969099521155010221e-10,
                 C1 = 4.16666666666666019037e-02, C2 = -1.38888888888741095749e-03, C3 = 2.48015872894767294178e-05,
                 C4 = -2.75573143513906633035e-07, C5 = 2.08757232129817482790e-09, C6 = -1.13596475577881948265e-11;
    const double q = rint(4.0 * u);
    const double r = u - 0.25 * q;                       // exact
    const double t = 6.283185307179586476925 * r;
    const double z = t * t;
    const double sp = t + t * z * (S1 + z * (S2 + z * (S3 + z * (S4 + z * (S5 + z * S6)))));
    const double cp = 1.0 - 0.5 * z + z * z * (C1 + z * (C2 + z * (C3 + z * (C4 + z * (C5 + z * C6)))));
    const int qi = (int)q & 3;
    sn = (qi == 0) ? sp : (qi == 1) ? cp : (qi == 2) ? -sp : -cp;
    cs = (qi == 0) ? cp : (qi == 1) ? -sp : (qi == 2) ? -cp : sp;
}
__global__ void k5(long long* out, const double* g, double* sink, double* err)
{
    const int t = threadIdx.x;
    double x = 0.3 + 1e-3 * t + g[t];
    long long c0, c1; int slot = 0;
#define M5(body) { c0 = clock64(); for (int r = 0; r < 20; r++) { body; } c1 = clock64(); if (t == 0) out[slot] = (c1 - c0) / 20; slot++; }
    double a = 0, b2 = 0;
    M5(a += log(x); x = x * 0.999 + 1e-4)
    M5(a += fast_log01(x); x = x * 0.999 + 1e-4)
    double s_ = 0, c_ = 0;
    M5(sincospi(2.0 * x, &s_, &c_); a += s_ + c_; x = x * 0.999 + 1e-4)
    M5(fast_sincos2pi(x, s_, c_); a += s_ + c_; x = x * 0.999 + 1e-4)
    // accuracy over a sweep of u
    double e1 = 0, e2 = 0;
    for (int i = 0; i < 4000; i++) {
        const double u = (t * 4000 + i + 0.5) / (256.0 * 4000.0);
        e1 = fmax(e1, fabs(fast_log01(u) - log(u)) / fabs(log(u) - 1e-300));
        double s1, c1_, s2, c2; sincospi(2.0 * u, &s1, &c1_); fast_sincos2pi(u, s2, c2);
        e2 = fmax(e2, fmax(fabs(s1 - s2), fabs(c1_ - c2)));
    }
    err[t] = e1; err[256 + t] = e2;
    sink[t] = a + b2 + x;
}

int main()
{
    long long* out; double *g, *sink;
    hipMalloc(&out, 64 * 8); hipMalloc(&g, (1 << 20) * 8); hipMalloc(&sink, 256 * 8);
    hipMemset(g, 0, (1 << 20) * 8); hipMemset(out, 0, 64 * 8);
    for (int it = 0; it < 2; it++) hipLaunchKernelGGL(k, dim3(1), dim3(256), 0, 0, out, g, sink, 50);
    hipDeviceSynchronize();
    long long h[64]; hipMemcpy(h, out, 64 * 8, hipMemcpyDeviceToHost);
    const char* names[] = {"barrier", "wave DPP incl scan f64", "block_excl_scan", "LDS wr+bar+rd+bar", "exp", "div", "log",
                           "dependent global load", "block_sum", "seg_excl_scan<4>", "chunk_record<8>", "philox+box-muller pair"};
    for (int i = 0; i < 12; i++) printf("%-28s %6lld cycles\n", names[i], h[i]);
    hipMemset(out, 0, 64 * 8);
    for (int it = 0; it < 2; it++) hipLaunchKernelGGL(k2, dim3(1), dim3(256), 0, 0, out, (const uint64_t*)g, (uint64_t*)sink, 20);
    hipDeviceSynchronize();
    hipMemcpy(h, out, 64 * 8, hipMemcpyDeviceToHost);
    const char* n2[] = {"u32 mad (dep)", "u32 shift+xor+add (3 dep ops)", "u64 shift+add", "u64 add", "u64 and/cmp/cndmask/add", "u32 and/cmp/cndmask/add", "walk_chain per link"};
    for (int i = 0; i < 7; i++) printf("%-34s %6.1f cycles\n", n2[i], h[i] / 10.0);
    hipMemset(out, 0, 64 * 8);
    hipLaunchKernelGGL(k3, dim3(1), dim3(256), 0, 0, out, (const uint64_t*)g, (uint64_t*)sink, 10);
    hipDeviceSynchronize();
    hipMemcpy(h, out, 64 * 8, hipMemcpyDeviceToHost);
    printf("walk_chain, 10 links, first execution in a fresh kernel: %lld cycles; second execution: %lld cycles\n", h[0], h[1]);
    for (int mode = 0; mode < 2; mode++) {
        hipMemset(out, 0, 64 * 8);
        hipLaunchKernelGGL(k4, dim3(1), dim3(1024), 0, 0, out, (const uint64_t*)g, (uint64_t*)sink, 10, mode);
        hipDeviceSynchronize();
        hipMemcpy(h, out, 64 * 8, hipMemcpyDeviceToHost);
        printf("walk_chain 10 links inside a 1024-thread workgroup, %s: %lld cycles\n", mode ? "all 16 waves walking" : "15 waves waiting at the barrier", h[0]);
    }
    { double* err; hipMalloc(&err, 512 * 8); hipMemset(out, 0, 64 * 8);
      hipLaunchKernelGGL(k5, dim3(1), dim3(256), 0, 0, out, g, sink, err); hipDeviceSynchronize();
      hipMemcpy(h, out, 64 * 8, hipMemcpyDeviceToHost); double he[512]; hipMemcpy(he, err, 512 * 8, hipMemcpyDeviceToHost);
      double e1 = 0, e2 = 0; for (int i = 0; i < 256; i++) { e1 = fmax(e1, he[i]); e2 = fmax(e2, he[256 + i]); }
      printf("log %lld cycles, fast_log01 %lld; sincospi %lld, fast_sincos2pi %lld; max rel err log %.2e, max abs err sin/cos %.2e\n", h[0], h[1], h[2], h[3], e1, e2); }
    return 0;
}
