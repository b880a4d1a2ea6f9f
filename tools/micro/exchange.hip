// Dev tool: cost of an ALL-TO-ALL hand-off of small per-workgroup records between co-resident workgroups inside one
// launch -- the seams of a fused per-observation kernel (every workgroup publishes a few words, every workgroup needs
// all of them).  Forms, as in MI355X_MICROARCH.md (visibility, price list):
//   G   data-tagged granules: 8-byte {tag = epoch, 32-bit value}, ONE sc1 store each; consumers re-read with relaxed sc1
//       loads until every tag matches (no flag, no fence)
//   F   payload by 16-byte sc1 stores, drained (s_waitcnt vmcnt(0)) + workgroup barrier, then ONE agent atomic add on a
//       sharded arrival counter; consumers poll the shards with sc1 loads, then read the payload with 16-byte sc1 loads
// Geometry: B workgroups x 256 threads, 64 KiB of LDS each (two per CU), K 32-bit words per workgroup per exchange,
// three exchanges per "observation" separated by `work` dependent FMAs (uneven: + skew for some workgroups).
// Every word received is checked.  Spins are bounded (wall clock): the tool cannot hang the box.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <stdlib.h>
#include <vector>
#include <algorithm>

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned long long u64;
#define RLX_AGENT __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT

__device__ __forceinline__ unsigned hashv(unsigned b, unsigned k, unsigned e) { unsigned x = b * 2654435761u ^ (k * 40503u + e * 2246822519u); x ^= x >> 15; x *= 2654435761u; return x ^ (x >> 13); }

constexpr int MAXB = 512, NSH = 16, SH_STRIDE = 32;     // arrival counter: 16 shards, 128 B apart

struct Args {
    u64* gran;          // [3][K][MAXB] granules
    u32x4* pay;         // [3][K/4][MAXB] 16-byte pieces, plane-major
    unsigned* cnt;      // [3][NSH * SH_STRIDE]
    long long* out;     // [B][4]: cycles in exchanges, cycles total, passes, wall ticks
    unsigned* err;      // [4]: value errors, timeouts
    int reps, work, skew;
};

__device__ __forceinline__ bool timed_out(long long t0) { return (long long)wall_clock64() - t0 > 5000000ll; }   // 50 ms at 100 MHz

template <int K, int FORM>
__global__ __launch_bounds__(256) void k_ex(Args a)
{
    extern __shared__ __attribute__((aligned(16))) char lds[];
    (void)lds;
    const int t = threadIdx.x, lane = t & 63, b = blockIdx.x, B = gridDim.x;
    __shared__ int bail;
    if (t == 0) bail = 0;
    __syncthreads();
    double acc = 1.0 + t * 1e-9;
    long long cyc_ex = 0, passes = 0;
    const long long w0 = wall_clock64();
    const long long c_start = clock64();
    unsigned bad = 0;
    for (int r = 0; r < a.reps; r++) {
        for (int ph = 0; ph < 3; ph++) {
            const unsigned epoch = (unsigned)(r * 3 + ph + 1);
            // "compute phase": dependent chain; some workgroups take longer (uneven load)
            const int w = a.work + (((b * 7 + r) % 13) == 0 ? a.skew : 0);
            for (int i = 0; i < w; i++) acc = acc * 1.0000001 + 1e-12;
            const long long c0 = clock64();
            if (FORM == 0) {
                if (t < K) __hip_atomic_store(a.gran + ((size_t)ph * K + t) * MAXB + b, ((u64)epoch << 32) | hashv(b, t, epoch), RLX_AGENT);
                // gather: thread t takes blocks 2t, 2t+1 (one 16-byte load covers both granules of a plane)
                __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)(a.gran + (size_t)ph * K * MAXB), 0, K * MAXB * 8, 0x00020000);
                u32x4 v[K];
                const bool have = 2 * t < B;
                for (;;) {
                    bool ok = true;
#pragma unroll
                    for (int k = 0; k < K; k++) {
                        v[k] = __builtin_amdgcn_raw_buffer_load_b128(rs, (k * MAXB + 2 * t) * 8, 0, 16);
                        ok = ok && (!have || (v[k].y == epoch && (2 * t + 1 >= B || v[k].w == epoch)));
                    }
                    passes++;
                    if (__all(ok)) break;
                    if (lane == 0 && timed_out(w0)) bail = 1;
                    if (bail) break;
                    __builtin_amdgcn_s_sleep(2);
                }
                __syncthreads();
                if (bail) { if (t == 0) atomicAdd(a.err + 1, 1u); return; }
                if (have) {
#pragma unroll
                    for (int k = 0; k < K; k++) {
                        if (v[k].x != hashv(2 * t, k, epoch)) bad++;
                        if (2 * t + 1 < B && v[k].z != hashv(2 * t + 1, k, epoch)) bad++;
                    }
                }
            } else {
                // payload: K/4 16-byte pieces per workgroup, plane-major [K/4][MAXB]
                __amdgpu_buffer_rsrc_t rp = __builtin_amdgcn_make_buffer_rsrc((void*)(a.pay + (size_t)ph * (K / 4) * MAXB), 0, (K / 4) * MAXB * 16, 0x00020000);
                if (t < K / 4) {
                    u32x4 p; p.x = hashv(b, 4 * t, epoch); p.y = hashv(b, 4 * t + 1, epoch); p.z = hashv(b, 4 * t + 2, epoch); p.w = hashv(b, 4 * t + 3, epoch);
                    __builtin_amdgcn_raw_buffer_store_b128(p, rp, (t * MAXB + b) * 16, 0, 16);
                }
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __syncthreads();
                unsigned* cn = a.cnt + (size_t)ph * NSH * SH_STRIDE;
                if (t == 0) __hip_atomic_fetch_add(cn + (b % NSH) * SH_STRIDE, 1u, RLX_AGENT);
                // every wave polls the shards itself (lanes 0..15), then loads its part of the payload
                const unsigned want = (unsigned)(r + 1) * (unsigned)(B / NSH);      // (B is a multiple of NSH here)
                for (;;) {
                    bool ok = true;
                    if (lane < NSH) ok = __hip_atomic_load(cn + lane * SH_STRIDE, RLX_AGENT) >= want;
                    passes++;
                    if (__all(ok)) break;
                    if (lane == 0 && timed_out(w0)) bail = 1;
                    if (bail) break;
                    __builtin_amdgcn_s_sleep(2);
                }
                if (bail) { __syncthreads(); if (t == 0) atomicAdd(a.err + 1, 1u); return; }
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                u32x4 v[K / 4][2];
                const bool have = 2 * t < B;
#pragma unroll
                for (int k = 0; k < K / 4; k++) {
                    v[k][0] = __builtin_amdgcn_raw_buffer_load_b128(rp, (k * MAXB + 2 * t) * 16, 0, 16);
                    v[k][1] = __builtin_amdgcn_raw_buffer_load_b128(rp, (k * MAXB + 2 * t + 1) * 16, 0, 16);
                }
                if (have) {
#pragma unroll
                    for (int k = 0; k < K / 4; k++) {
                        for (int j = 0; j < 2; j++) {
                            const int bb = 2 * t + j;
                            if (bb >= B) continue;
                            if (v[k][j].x != hashv(bb, 4 * k, epoch) || v[k][j].y != hashv(bb, 4 * k + 1, epoch) ||
                                v[k][j].z != hashv(bb, 4 * k + 2, epoch) || v[k][j].w != hashv(bb, 4 * k + 3, epoch)) bad++;
                        }
                    }
                }
                __syncthreads();
            }
            cyc_ex += clock64() - c0;
        }
    }
    if (bad) atomicAdd(a.err, bad);
    if (t == 0) { a.out[b * 4] = cyc_ex; a.out[b * 4 + 1] = clock64() - c_start; a.out[b * 4 + 2] = passes; a.out[b * 4 + 3] = (long long)wall_clock64() - w0; }
    if (acc == 0.123) a.out[0] = 0;
}

template <int K, int FORM>
static void run(const char* name, int B, int reps, int work, int skew)
{
    Args a;
    hipMalloc(&a.gran, 3 * 16 * MAXB * 8); hipMalloc(&a.pay, 3 * 4 * MAXB * 16); hipMalloc(&a.cnt, 3 * NSH * SH_STRIDE * 4);
    hipMalloc(&a.out, MAXB * 4 * 8); hipMalloc(&a.err, 16);
    hipMemset(a.gran, 0, 3 * 16 * MAXB * 8); hipMemset(a.pay, 0, 3 * 4 * MAXB * 16); hipMemset(a.cnt, 0, 3 * NSH * SH_STRIDE * 4); hipMemset(a.err, 0, 16);
    a.reps = reps; a.work = work; a.skew = skew;
    hipFuncSetAttribute(reinterpret_cast<const void*>(&k_ex<K, FORM>), hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL((k_ex<K, FORM>), dim3(B), dim3(256), 60000, 0, a);
    hipEventRecord(e1, 0);
    hipError_t e = hipDeviceSynchronize();
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    std::vector<long long> h(B * 4); unsigned err[4];
    hipMemcpy(h.data(), a.out, B * 4 * 8, hipMemcpyDeviceToHost); hipMemcpy(err, a.err, 16, hipMemcpyDeviceToHost);
    long long mx = 0, mn = 1ll << 60, tot = 0, pas = 0;
    for (int i = 0; i < B; i++) { mx = std::max(mx, h[i * 4]); mn = std::min(mn, h[i * 4]); tot += h[i * 4 + 1]; pas += h[i * 4 + 2]; }
    const double n = (double)reps * 3;
    printf("%-34s B %3d work %4d skew %4d: %7.2f us per exchange+work (kernel %.3f ms); in-exchange cycles per exchange min %6.0f max %6.0f; "
           "%.1f poll passes per wave-exchange; clock64 %.0f MHz; value errors %u, timeouts %u%s\n",
           name, B, work, skew, ms * 1e3 / n, ms, mn / n, mx / n, (double)pas / B / n, (double)tot / B / (h[3] * 10.0) * 1e3, err[0], err[1],
           e == hipSuccess ? "" : "  HIP ERROR");
    hipFree(a.gran); hipFree(a.pay); hipFree(a.cnt); hipFree(a.out); hipFree(a.err);
}

int main(int argc, char** argv)
{
    const int reps = argc > 1 ? atoi(argv[1]) : 300;
    for (int B : {256, 512}) {
        for (int work : {0, 600}) {
            const int skew = work ? 600 : 0;
            run<6, 0>("granules K=6 (24 B payload)", B, reps, work, skew);
            run<8, 0>("granules K=8 (32 B payload)", B, reps, work, skew);
            run<16, 0>("granules K=16 (64 B payload)", B, reps, work, skew);
            run<8, 1>("flag form K=8 (32 B payload)", B, reps, work, skew);
            run<16, 1>("flag form K=16 (64 B payload)", B, reps, work, skew);
        }
    }
    return 0;
}
