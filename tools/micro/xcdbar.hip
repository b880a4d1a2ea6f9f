// Dev tool: (1) which XCD does workgroup i land on?  (2) cost of a counter barrier among the workgroups of ONE XCD.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

__device__ __forceinline__ unsigned xcc_id() { unsigned v; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v)); return v & 0xf; }

__device__ __forceinline__ void group_barrier(unsigned* count, unsigned* gen, unsigned nblocks)
{
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned g = __hip_atomic_load(gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned prev = __hip_atomic_fetch_add(count, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
        if (prev == nblocks - 1) {
            __hip_atomic_store(count, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(gen, g + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        } else {
            long spins = 0;
            while (__hip_atomic_load(gen, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) == g) {
                __builtin_amdgcn_s_sleep(1);
                if (++spins > 4000000) break;      // never hang the box
            }
        }
    }
    __syncthreads();
}

__global__ void k(unsigned* counters /* [8][32] */, long long* out, unsigned* xcc_of, double* data, int reps, int ngroups)
{
    const unsigned grp = blockIdx.x % ngroups, nb = gridDim.x / ngroups;
    if (threadIdx.x == 0) xcc_of[blockIdx.x] = xcc_id();
    unsigned* count = counters + grp * 64, *gen = counters + grp * 64 + 32;
    long long c0 = clock64();
    for (int r = 0; r < reps; r++) {
        data[blockIdx.x * 256 + threadIdx.x] += 1.0;
        group_barrier(count, gen, nb);
    }
    long long c1 = clock64();
    if (threadIdx.x == 0) out[blockIdx.x] = (c1 - c0) / reps;
}

int main()
{
    unsigned *counters, *xcc; long long* out; double* data;
    hipMalloc(&counters, 8 * 64 * 4); hipMalloc(&out, 4096 * 8); hipMalloc(&xcc, 4096 * 4); hipMalloc(&data, 4096 * 256 * 8);
    hipMemset(data, 0, 4096 * 256 * 8);
    for (int nb : {64, 256, 512}) {
        for (int ngroups : {1, 8}) {
            hipMemset(counters, 0, 8 * 64 * 4);
            hipLaunchKernelGGL(k, dim3(nb), dim3(256), 0, 0, counters, out, xcc, data, 100, ngroups);
            hipDeviceSynchronize();
            long long h[4096]; unsigned hx[4096];
            hipMemcpy(h, out, nb * 8, hipMemcpyDeviceToHost); hipMemcpy(hx, xcc, nb * 4, hipMemcpyDeviceToHost);
            int consistent = 1;
            for (int i = 0; i < nb; i++) if (hx[i] != hx[i % 8]) consistent = 0;
            printf("blocks %4d, %d group(s) of %d: %lld cycles per barrier (block 0); xcc of blocks 0..15:", nb, ngroups, nb / ngroups, h[0]);
            for (int i = 0; i < 16; i++) printf(" %u", hx[i]);
            printf("  [block i -> xcc(i %% 8) for all: %s]\n", consistent ? "yes" : "NO");
        }
    }
    return 0;
}
