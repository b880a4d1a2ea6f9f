// Dev tool (round 3, verdict item 2): the grid barrier in the form /opt/skills/guides/MI355X_MICROARCH.md prices as "barrier-xcd"
// (4.1 / 5.9 / 9.7 us at 256 / 512 / 1024 workgroups): XCD-hierarchical -- every workgroup adds to ITS XCC's counter (relaxed), the last
// arriver of an XCC is that XCC's leader: release fence, add to the top counter, poll the top counter (relaxed sc1 loads + s_sleep),
// acquire fence, bump its XCC's generation word; everybody else polls its own XCC's generation with relaxed sc1 loads and then
// fences (acquire, agent).  Counters never reset (monotonic targets), every spin bounded by the wall clock.
// Round 1's tools/micro/gridbar.hip (25 us at 512 workgroups) used ACQ_REL adds on one line and ACQUIRE polls: the slow forms.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

#define RLX __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT
__device__ __forceinline__ unsigned xcc_id() { unsigned v; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v)); return v & 0xf; }

struct Bar { unsigned xcount[8][32]; unsigned xgen[8][32]; unsigned top[32]; unsigned timeouts; };

// per_xcc: workgroups on this XCC (gridDim.x / nxcc for the round-robin dispatch); round: 1, 2, ...
__device__ __forceinline__ bool xcd_barrier(Bar* b, unsigned xcc, unsigned per_xcc, unsigned nxcc, unsigned round)
{
    __syncthreads();
    bool ok = true;
    if (threadIdx.x == 0) {
        const long long t0 = (long long)wall_clock64();
        const unsigned prev = __hip_atomic_fetch_add(&b->xcount[xcc][0], 1u, RLX);
        if (prev + 1 == per_xcc * round) {                       // this XCC's last arriver: its leader
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __hip_atomic_fetch_add(&b->top[0], 1u, RLX);
            while (__hip_atomic_load(&b->top[0], RLX) < nxcc * round) {
                __builtin_amdgcn_s_sleep(1);
                if ((long long)wall_clock64() - t0 > 2000000) { ok = false; break; }      // 20 ms
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            __hip_atomic_store(&b->xgen[xcc][0], round, RLX);
        } else {
            while (__hip_atomic_load(&b->xgen[xcc][0], RLX) < round) {
                __builtin_amdgcn_s_sleep(1);
                if ((long long)wall_clock64() - t0 > 2000000) { ok = false; break; }
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        }
        if (!ok) atomicAdd(&b->timeouts, 1u);
    }
    __syncthreads();
    return ok;
}

// record: every workgroup publishes a 128-byte record (plain stores, covered by the leader's release) and after the barrier re-reads its
// right-hand neighbour's (the guide's "re-reading a 128-B record per WG" column)
__global__ void k(Bar* b, double* rec /* [grid][16] */, long long* out, unsigned* census, int reps, int with_record, unsigned nxcc)
{
    const unsigned xcc = xcc_id();
    if (threadIdx.x == 0) atomicAdd(&census[xcc], 1u);
    const unsigned per_xcc = gridDim.x / nxcc;
    double acc = 0.0;
    const long long t0 = (long long)wall_clock64();
    for (int r = 1; r <= reps; r++) {
        if (with_record && threadIdx.x < 16) rec[blockIdx.x * 16 + threadIdx.x] = (double)(r * 1000 + threadIdx.x);
        if (!xcd_barrier(b, xcc, per_xcc, nxcc, (unsigned)(with_record ? 2 * r - 1 : r))) break;
        if (with_record && threadIdx.x < 16) {
            const double v = rec[((blockIdx.x + 1) % gridDim.x) * 16 + threadIdx.x];
            acc += (v == (double)(r * 1000 + threadIdx.x)) ? 0.0 : 1.0;         // stale reads counted
        }
        // (a second barrier per round keeps a fast workgroup from overwriting its record before its neighbour has read it)
        if (with_record && !xcd_barrier(b, xcc, per_xcc, nxcc, (unsigned)(2 * r))) break;
    }
    const long long t1 = (long long)wall_clock64();
    if (threadIdx.x < 16) atomicAdd((unsigned long long*)&out[1], (unsigned long long)acc);
    if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = t1 - t0;
}

int main()
{
    Bar* b; double* rec; long long* out; unsigned* census;
    hipMalloc(&b, sizeof(Bar)); hipMalloc(&rec, 4096 * 16 * 8); hipMalloc(&out, 64); hipMalloc(&census, 64);
    const int reps = 500;
    for (int nb : {256, 512}) {
        for (int with_record : {0, 1}) {
            hipMemset(b, 0, sizeof(Bar)); hipMemset(out, 0, 64); hipMemset(census, 0, 64); hipMemset(rec, 0, 4096 * 16 * 8);
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            hipEventRecord(e0, 0);
            hipLaunchKernelGGL(k, dim3(nb), dim3(256), 0, 0, b, rec, out, census, reps, with_record, 8u);
            hipEventRecord(e1, 0);
            hipDeviceSynchronize();
            float ms = 0; hipEventElapsedTime(&ms, e0, e1);
            long long h[2]; unsigned hc[16]; Bar hb;
            hipMemcpy(h, out, 16, hipMemcpyDeviceToHost); hipMemcpy(hc, census, 64, hipMemcpyDeviceToHost); hipMemcpy(&hb, b, sizeof(Bar), hipMemcpyDeviceToHost);
            const int nbar = with_record ? 2 * reps : reps;
            printf("barrier-xcd form, %4d workgroups (%d per CU), %s: %.2f us per barrier (kernel %.3f ms, %d barriers; wall-clock in-kernel %.2f us each); "
                   "stale record reads %lld, time-outs %u; workgroups per XCC:", nb, nb / 256, with_record ? "128-B record per WG re-read after it" : "nothing published",
                   ms * 1e3 / nbar, ms, nbar, (double)h[0] / 100.0 / nbar, h[1], hb.timeouts);
            for (int i = 0; i < 8; i++) printf(" %u", hc[i]);
            printf("\n");
        }
    }
    return 0;
}
