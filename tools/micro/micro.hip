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
    return 0;
}
