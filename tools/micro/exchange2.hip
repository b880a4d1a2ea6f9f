// Dev tool (companion of exchange.hip): the REDUCER form of a grid-wide seam, and the raw one-to-one hand-off latency.
//   p2p      two workgroups ping-pong one 8-byte {tag, value} granule (sc1 store, relaxed sc1 load poll)
//   reducer  B workers publish K granules each; R dedicated reducer workgroups (blockIdx >= B) each gather all B x K
//            granules, "reduce" (sum of the values), and publish one 16-byte result PER WORKER (two granules in a slot
//            of that worker's own: no hot line); worker b polls the slot written by reducer b % R.
// Every word received is checked; spins are bounded by the wall clock.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <stdlib.h>
#include <vector>
#include <algorithm>

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned long long u64;
#define RLX_AGENT __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT
constexpr int MAXB = 512, MAXR = 8;

__device__ __forceinline__ unsigned hashv(unsigned b, unsigned k, unsigned e) { unsigned x = b * 2654435761u ^ (k * 40503u + e * 2246822519u); x ^= x >> 15; x *= 2654435761u; return x ^ (x >> 13); }
__device__ __forceinline__ bool timed_out(long long t0) { return (long long)wall_clock64() - t0 > 5000000ll; }

__global__ void k_p2p(u64* g, int reps, int partner, long long* out, unsigned* err)
{
    const int b = blockIdx.x;
    if (threadIdx.x != 0 || (b != 0 && b != partner)) return;
    const long long w0 = wall_clock64();
    const long long c0 = clock64();
    u64* mine = g + (b == 0 ? 0 : 32), *theirs = g + (b == 0 ? 32 : 0);      // 256 B apart
    for (int r = 1; r <= reps; r++) {
        if (b == 0) __hip_atomic_store(mine, ((u64)r << 32) | (unsigned)r, RLX_AGENT);
        for (;;) {
            const u64 x = __hip_atomic_load(theirs, RLX_AGENT);
            if ((x >> 32) == (u64)r) break;
            if (timed_out(w0)) { atomicAdd(err + 1, 1u); return; }
        }
        if (b != 0) __hip_atomic_store(mine, ((u64)r << 32) | (unsigned)r, RLX_AGENT);
    }
    out[b == 0 ? 0 : 1] = clock64() - c0;
}

struct Args { u64* gran; u64* res; long long* out; unsigned* err; int reps, B, R, work; };

template <int K>
__global__ __launch_bounds__(256) void k_red(Args a)
{
    extern __shared__ __attribute__((aligned(16))) char lds[];
    (void)lds;
    __shared__ int bail;
    __shared__ unsigned wsum[4];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, b = blockIdx.x, B = a.B, R = a.R;
    if (t == 0) bail = 0;
    __syncthreads();
    const long long w0 = wall_clock64();
    long long cyc = 0, passes = 0;
    unsigned bad = 0;
    double acc = 1.0 + t * 1e-9;
    const bool reducer = b >= B;
    for (int r = 0; r < a.reps; r++) {
        for (int ph = 0; ph < 3; ph++) {
            const unsigned epoch = (unsigned)(r * 3 + ph + 1);
            u64* gr = a.gran + (size_t)ph * K * MAXB;
            u64* rs = a.res + (size_t)ph * MAXR * MAXB * 2;
            if (!reducer) {
                for (int i = 0; i < a.work; i++) acc = acc * 1.0000001 + 1e-12;
                const long long c0 = clock64();
                if (t < K) __hip_atomic_store(gr + (size_t)t * MAXB + b, ((u64)epoch << 32) | hashv(b, t, epoch), RLX_AGENT);
                // wait for this worker's result slot: two granules written by reducer b % R
                if (wave == 0) {
                    u64* slot = rs + ((size_t)(b % R) * MAXB + b) * 2;
                    for (;;) {
                        bool ok = true; u64 x = 0;
                        if (lane < 2) { x = __hip_atomic_load(slot + lane, RLX_AGENT); ok = (x >> 32) == epoch; }
                        passes++;
                        if (__all(ok)) { if (lane < 2 && (unsigned)x != (lane == 0 ? hashv(12345u, 0, epoch) : (unsigned)b)) bad++; break; }
                        if (lane == 0 && timed_out(w0)) bail = 1;
                        if (bail) break;
                        __builtin_amdgcn_s_sleep(1);
                    }
                }
                __syncthreads();
                if (bail) { if (t == 0) atomicAdd(a.err + 1, 1u); return; }
                cyc += clock64() - c0;
            } else {
                __amdgpu_buffer_rsrc_t rr = __builtin_amdgcn_make_buffer_rsrc((void*)gr, 0, K * MAXB * 8, 0x00020000);
                u32x4 v[K];
                const bool have = 2 * t < B;
                const long long c0 = clock64();
                for (;;) {
                    bool ok = true;
#pragma unroll
                    for (int k = 0; k < K; k++) {
                        v[k] = __builtin_amdgcn_raw_buffer_load_b128(rr, (k * MAXB + 2 * t) * 8, 0, 16);
                        ok = ok && (!have || (v[k].y == epoch && (2 * t + 1 >= B || v[k].w == epoch)));
                    }
                    passes++;
                    if (__all(ok)) break;
                    if (lane == 0 && timed_out(w0)) bail = 1;
                    if (bail) break;
                    __builtin_amdgcn_s_sleep(1);
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
                // results: one slot per worker (this reducer serves workers b' with b' % R == b - B)
                const int me = b - B;
                for (int bb = t; bb < B; bb += 256) {
                    if (bb % R == me) {
                        u64* slot = rs + ((size_t)me * MAXB + bb) * 2;
                        __hip_atomic_store(slot, ((u64)epoch << 32) | hashv(12345u, 0, epoch), RLX_AGENT);
                        __hip_atomic_store(slot + 1, ((u64)epoch << 32) | (unsigned)bb, RLX_AGENT);
                    }
                }
                cyc += clock64() - c0;
            }
        }
    }
    if (bad) atomicAdd(a.err, bad);
    if (t == 0) { a.out[b * 4] = cyc; a.out[b * 4 + 2] = passes; a.out[b * 4 + 3] = (long long)wall_clock64() - w0; }
    if (acc == 0.123) a.out[0] = 0;
}

template <int K>
static void run_red(int B, int R, int reps, int work, int ldsbytes)
{
    Args a;
    hipMalloc(&a.gran, 3 * 16 * MAXB * 8); hipMalloc(&a.res, 3 * MAXR * MAXB * 2 * 8); hipMalloc(&a.out, (MAXB + MAXR) * 4 * 8); hipMalloc(&a.err, 16);
    hipMemset(a.gran, 0, 3 * 16 * MAXB * 8); hipMemset(a.res, 0, 3 * MAXR * MAXB * 2 * 8); hipMemset(a.err, 0, 16); hipMemset(a.out, 0, (MAXB + MAXR) * 4 * 8);
    a.reps = reps; a.B = B; a.R = R; a.work = work;
    hipFuncSetAttribute(reinterpret_cast<const void*>(&k_red<K>), hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL((k_red<K>), dim3(B + R), dim3(256), ldsbytes, 0, a);
    hipEventRecord(e1, 0);
    hipError_t e = hipDeviceSynchronize();
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    std::vector<long long> h((B + R) * 4); unsigned err[4];
    hipMemcpy(h.data(), a.out, (B + R) * 4 * 8, hipMemcpyDeviceToHost); hipMemcpy(err, a.err, 16, hipMemcpyDeviceToHost);
    long long wp = 0, rp = 0;
    for (int i = 0; i < B; i++) wp += h[i * 4 + 2];
    for (int i = B; i < B + R; i++) rp += h[i * 4 + 2];
    const double n = (double)reps * 3;
    printf("reducer form K=%2d B %3d R %d work %4d lds %5d: %6.2f us per exchange(+work) (kernel %.3f ms); poll passes per exchange: worker %.1f, reducer wave %.1f; value errors %u, timeouts %u%s\n",
           K, B, R, work, ldsbytes, ms * 1e3 / n, ms, (double)wp / B / n, (double)rp / R / 4 / n, err[0], err[1], e == hipSuccess ? "" : "  HIP ERROR");
    hipFree(a.gran); hipFree(a.res); hipFree(a.out); hipFree(a.err);
}

int main(int argc, char** argv)
{
    const int reps = argc > 1 ? atoi(argv[1]) : 300;
    {
        u64* g; long long* out; unsigned* err;
        hipMalloc(&g, 4096); hipMalloc(&out, 64); hipMalloc(&err, 16);
        for (int partner : {1, 8, 9, 255, 300}) {
            hipMemset(g, 0, 4096); hipMemset(err, 0, 16);
            hipLaunchKernelGGL(k_p2p, dim3(512), dim3(64), 0, 0, g, 2000, partner, out, err);
            hipDeviceSynchronize();
            long long h[2]; unsigned e[4]; hipMemcpy(h, out, 16, hipMemcpyDeviceToHost); hipMemcpy(e, err, 16, hipMemcpyDeviceToHost);
            printf("p2p ping-pong block 0 <-> block %3d: %lld cycles per one-way hand-off (timeouts %u)\n", partner, h[0] / 4000, e[1]);
        }
    }
    for (int B : {256, 512}) {
        for (int R : {1, 2, 8}) {
            run_red<6>(B, R, reps, 0, 40000);
            run_red<8>(B, R, reps, 0, 40000);
        }
        run_red<8>(B, 1, reps, 600, 40000);
        run_red<16>(B, 1, reps, 0, 40000);
    }
    return 0;
}
