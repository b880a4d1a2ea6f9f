// multi.hip.h -- K independent LARGE filters per launch (blockIdx.y = filter).
//
// A PMMH iteration is one filter run (R/pmmh.R:445-457) and a chain is a strict dependency chain of 4 T launches; each of those
// kernels is latency-bound with most of the chip idle.  Independent chains (R/pmmh.R:511-531) can share the launches:
// the same kernel bodies as the one-filter path, one argument set per filter chosen by blockIdx.y, so K filters advance in
// lock-step on ONE stream and every filter's results are bit for bit those of bssm_pf_run.
#pragma once
#include "kernels.hip.h"

namespace bssm {

constexpr int MULTI_MAX = 4;      // filters per launch (the argument sets travel in the kernel arguments)

struct StepArgs {
    const double* xin; double* xout; double* lw; long long N; ModelPar par; double y; NoiseSrc ns;
    double* pm; double* ps; double* pq; const DevState* st; unsigned long long* gmax;
};
struct StepMulti { StepArgs a[MULTI_MAX]; };

template <int MODEL, bool TRANS, int WEIGHT>
__global__ __launch_bounds__(NTS) void k_step_multi(StepMulti m)
{
    __shared__ double sh[2 * (NTS / 64)];
    const StepArgs& a = m.a[blockIdx.y];
    step_block<MODEL, TRANS, WEIGHT, false>(sh, (int)blockIdx.x, a.xin, a.xout, a.lw, nullptr, a.N, a.par, a.y, a.ns, a.pm, a.ps, a.pq, a.st, a.gmax);
}

struct LocalArgs {
    const double* w; long long nw; const double* ain; int lim; BlockRec* brec; SideList* side; DevState* st; FromLw f;
    const BlockRec* prev_brec; const SideList* prev_side; double* ain_p_out;
};
struct LocalMulti { LocalArgs a[MULTI_MAX]; };

// k_local<MODE_W, true> (normalise + records of sum(w)) for K filters
__global__ __launch_bounds__(NT) void k_weights_multi(LocalMulti m)
{
    __shared__ SegSmem sm;
    __shared__ uint64_t tin[NT];
    __shared__ double es[1];
    const LocalArgs& a = m.a[blockIdx.y];
    local_block<MODE_W, true, MAXB>(sm, tin, es, (int)blockIdx.x, (int)gridDim.x, a.w, a.nw, a.ain, a.lim, a.brec, a.side, a.st, a.f);
}

// k_local<MODE_P, false, true> (resolve of sum(w) + records of cumsum(w / total)) for K filters
__global__ __launch_bounds__(NT) void k_localp_multi(LocalMulti m)
{
    __shared__ SegSmem sm;
    __shared__ uint64_t tin[NT];
    __shared__ double es[1];
    __shared__ ResolveSmem rs;
    const LocalArgs& a = m.a[blockIdx.y];
    const int bidx = (int)blockIdx.x, nblk = (int)gridDim.x;
    InResolve<MODE_W> pro; pro.sm = &sm; pro.rs = &rs; pro.brec = a.prev_brec; pro.side = a.prev_side; pro.B = nblk; pro.upto = nblk;
    pro.w = a.w; pro.nw = a.nw; pro.total = 1.0; pro.st = a.st; pro.stats = (bidx == 0);
    local_block<MODE_P, false, MAXB, false, InResolve<MODE_W>>(sm, tin, es, bidx, nblk, a.w, a.nw, a.ain, a.lim, a.brec, a.side, a.st, a.f, pro, a.ain_p_out);
}

struct ApplyOne { ApplyArgs a; DevState* st; const BlockRec* prev_brec; const SideList* prev_side; };
struct ApplyMulti { ApplyOne a[MULTI_MAX]; };

// k_apply<KIND, true> (resolve of the cumsum pass + expansion) for K filters
template <int KIND>
__global__ __launch_bounds__(NT) void k_apply_multi(ApplyMulti m)
{
    const ApplyOne& q = m.a[blockIdx.y];
    const int bidx = (int)blockIdx.x, nblk = (int)gridDim.x;
    __shared__ SegSmem sm;
    __shared__ uint64_t tin[NT];
    __shared__ int Tl[EB];
    __shared__ int Tbegin;
    extern __shared__ __attribute__((aligned(16))) double lx[];
    InResolve<MODE_P> pro; pro.sm = &sm; pro.rs = reinterpret_cast<ResolveSmem*>(lx); pro.brec = q.prev_brec; pro.side = q.prev_side;
    pro.B = nblk; pro.upto = bidx; pro.w = q.a.w; pro.nw = q.a.nw; pro.total = b2d(q.st->total_bits); pro.st = q.st; pro.stats = (bidx == nblk - 1);
    apply_block<KIND, false, InResolve<MODE_P>, false, CAPX, false>(sm, tin, Tl, Tbegin, bidx, nblk, q.a, q.st, q.a.nstage ? lx : nullptr, q.a.nstage, pro);
}

}  // namespace bssm
