// Dev tool: cost of a hand-rolled grid barrier (all workgroups co-resident) on MI355X.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

__device__ __forceinline__ void grid_barrier(unsigned* count, volatile unsigned* gen, unsigned nblocks)
{
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned g = __atomic_load_n((unsigned*)gen, __ATOMIC_RELAXED);
        __atomic_thread_fence(__ATOMIC_RELEASE);   // (agent scope in HIP by default for __atomic builtins w/o scope? use explicit builtin below)
        const unsigned prev = __hip_atomic_fetch_add(count, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
        if (prev == nblocks - 1) {
            __hip_atomic_store(count, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store((unsigned*)gen, g + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        } else {
            long spins = 0;
            while (__hip_atomic_load((unsigned*)gen, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) == g) {
                __builtin_amdgcn_s_sleep(1);
                if (++spins > 2000000) break;      // never hang the box
            }
        }
    }
    __syncthreads();
}

__global__ void k(unsigned* count, unsigned* gen, long long* out, double* data, int reps)
{
    const unsigned nb = gridDim.x;
    long long c0 = clock64();
    long long w0 = wall_clock64();
    for (int r = 0; r < reps; r++) {
        data[blockIdx.x * 256 + threadIdx.x] += 1.0;      // some global traffic that must be visible across the barrier
        grid_barrier(count, gen, nb);
    }
    long long c1 = clock64();
    long long w1 = wall_clock64();
    if (threadIdx.x == 0 && blockIdx.x == 0) { out[0] = (c1 - c0) / reps; out[1] = (w1 - w0) / reps; }
}

int main()
{
    unsigned *count, *gen; long long* out; double* data;
    hipMalloc(&count, 4); hipMalloc(&gen, 4); hipMalloc(&out, 64); hipMalloc(&data, 4096 * 256 * 8);
    hipMemset(count, 0, 4); hipMemset(gen, 0, 4); hipMemset(data, 0, 4096 * 256 * 8);
    int wc = 0; hipDeviceGetAttribute(&wc, hipDeviceAttributeWallClockRate, 0);
    for (int nb : {64, 256, 512, 1024}) {
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        const int reps = 200;
        hipLaunchKernelGGL(k, dim3(nb), dim3(256), 0, 0, count, gen, out, data, 10);
        hipDeviceSynchronize();
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL(k, dim3(nb), dim3(256), 0, 0, count, gen, out, data, reps);
        hipEventRecord(e1, 0);
        hipDeviceSynchronize();
        float ms; hipEventElapsedTime(&ms, e0, e1);
        long long h[2]; hipMemcpy(h, out, 16, hipMemcpyDeviceToHost);
        printf("blocks %4d: %.2f us per barrier (event), clock64 %lld, wall_clock64 %lld (rate %d kHz)\n", nb, 1e3 * ms / reps, h[0], h[1], wc);
    }
    return 0;
}
