// fused.hip.h -- ONE launch per observation of a bootstrap filter (scalar Gaussian-observation models, N <= 2^20).
//
// The multi-launch path (kernels.hip.h) spends one kernel per grid-wide dependency of the reference's arithmetic:
//   max / sum of the log-weights (R/particle_filter_core.R:204-206)  ->  total = sum(w) (src/resampling.cpp:20,47)
//   ->  cumsum(w / total) (:24-25,51-52)  ->  walk + gather (:31-37,57-63, R/resampling.R:40,60)
// and every one of those kernels re-loads what the previous one had in registers and re-derives the block scans.
// Here a workgroup keeps its 2048 particles, log-weights, weights and block scan in registers / LDS from the transition
// to the expansion; only small per-block records cross workgroups, HUB AND SPOKE:
//   every worker publishes its record as data-tagged 8-byte granules {tag = launch number, 32-bit value} (one sc1 store
//   each, no flag, no fence: the data is the flag);  ONE workgroup -- the "resolver", which is also the worker of a middle
//   block -- gathers all records, runs the grid-level step once (log-sum-exp combine / exact total / every block's exact
//   incoming state: the same resolve_in_block the multi-launch kernels run) and writes each worker's answer into a slot
//   of that worker's own; a worker polls nothing but its own slot.
// Measured background (tools/micro/exchange*.hip, profiles/r03_a_*): an all-to-all re-read by 512 workgroups costs 7-11 us
// per seam (every sc1 poll is a fabric read: B^2 x payload bytes per pass), the hub form 4.4-5 us, a one-to-one hand-off
// 0.35-0.55 us.
//
// All B <= 512 workgroups must be resident at once (two per CU): the launch is sized for that and EVERY spin is bounded
// by the wall clock; a time-out or anything the records cannot express in-launch (a literal re-run of another block's
// terms) raises FLAG_FUSED_* in the run state, the remaining launches of the run return at once and bssm_pf_run repeats
// the run on the multi-launch path -- a different HIP path, never a CPU one.
//
// Arithmetic: operation for operation that of k_step / k_local / k_apply (the reductions of the 1024-thread k_step are
// re-enacted in its association order), so a fused run returns bit for bit what the multi-launch run returns.
#pragma once
#include "kernels.hip.h"

namespace bssm {

typedef unsigned long long fz_u64;
#define BSSM_RLX_AGENT __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT

constexpr int FZ_MAXB = 2 * NT;        // workers (blocks of EB particles): N <= 2^20
constexpr int FZ_MINB = 256;           // grids up to this size (one workgroup per CU) stay on the multi-launch path by default: measured faster there (43.2 vs 46.9 us at
                                       // 256 blocks, but 47.8 vs 44.1 at 320: tools/diag_fused_threshold.py); option fused = 2 forces the fused launch at every size
constexpr int FZ_KREC = 8;             // granules of a block record on the wire
constexpr int FZ_KSIDE = (int)(sizeof(SideEntry) / 4);     // granules of a side entry
constexpr int FZ_NREP = 16;           // replicas of a result every worker reads (32 readers a line instead of 512)
constexpr int FZ_MAXSIDE = 16;         // side entries per block that travel in-launch (more: the run is repeated unfused)
constexpr uint32_t FLAG_FUSED_BAIL = 8u, FLAG_FUSED_TIMEOUT = 16u;
constexpr uint32_t FZ_ST_DOIT = 1u, FZ_ST_DEAD = 2u, FZ_ST_STOP = 4u, FZ_ST_BAIL = 8u;     // status word of a result slot
constexpr long long FZ_TIMEOUT_TICKS = 2000000ll;          // 20 ms of the 100 MHz wall clock

struct FusedWs {
    fz_u64 e1[6][FZ_MAXB];                      // worker -> resolver: (max, sum exp, sum exp^2) of the block's log-weights
    fz_u64 c1[FZ_NREP][16];                     // resolver -> all workers, replicated (worker b reads replica b % FZ_NREP): M, S, sum e^2, status
    fz_u64 a1[2][FZ_MAXB];                      // resolver -> worker b: approximate prefix of the block (plane-major: the resolver's stores are contiguous)
    fz_u64 e2[FZ_KREC][FZ_MAXB];                // records of the sum(w) pass
    fz_u64 c2[FZ_NREP][16];                     // total (bit pattern), status
    fz_u64 e3[FZ_KREC][FZ_MAXB];                // records of the cumsum(w / total) pass
    fz_u64 c3[FZ_NREP][16];                     // status
    fz_u64 a3[2][FZ_MAXB];                      // the block's exact incoming state
    fz_u64 side_w[FZ_MAXB][FZ_MAXSIDE][FZ_KSIDE];
    fz_u64 side_p[FZ_MAXB][FZ_MAXSIDE][FZ_KSIDE];
    // the resolver's private, plain copies: what the ordinary resolve code reads
    // (one set per pass: every address is written once and then read in a launch -- no line this CU's L1 may still hold from
    //  the other pass)
    BlockRec brec_priv[2][MAXB];                 // (four planes each, BREC_STRIDE apart)
    SideList side_priv[2][FZ_MAXB];
    long long stamps[2][24];
    long long startt[FZ_MAXB];                   // dev tool: wall clock at every block's first instruction
    long long endt[FZ_MAXB];                     // dev tool: wall clock at the end of every block's expansion
    long long waket[3][FZ_MAXB];                 // dev tool: wall clock at which every block had its three answers
    long long pubt[4][FZ_MAXB];                  // dev tool: wall clock (100 MHz) of every block's three publishes; [3][8..10] resolver's gather-done times                     // dev tool (make DEV=1): clock64() at the stages of a typical worker [0] / of the resolver [1], last launch
};

#ifdef BSSM_DEV_STAMPS
#define FZ_STAMP(k) do { if (fz_stamper) g.ws->stamps[fz_row][k] = clock64(); } while (0)
#else
#define FZ_STAMP(k) do { } while (0)
#endif

__device__ __forceinline__ fz_u64 fz_gran(uint32_t tag, uint32_t v) { return ((fz_u64)tag << 32) | v; }
__device__ __forceinline__ void fz_put(fz_u64* g, uint32_t tag, uint32_t v) { __hip_atomic_store(g, fz_gran(tag, v), BSSM_RLX_AGENT); }
__device__ __forceinline__ void fz_put64(fz_u64* g, uint32_t tag, uint64_t v) { fz_put(g, tag, (uint32_t)v); fz_put(g + 1, tag, (uint32_t)(v >> 32)); }
// (the two halves of a 64-bit value `stride` granules apart: plane-major results)
__device__ __forceinline__ void fz_put64x(fz_u64* g, int stride, uint32_t tag, uint64_t v) { fz_put(g, tag, (uint32_t)v); fz_put(g + stride, tag, (uint32_t)(v >> 32)); }
__device__ __forceinline__ fz_u64 fz_get(const fz_u64* g) { return __hip_atomic_load(g, BSSM_RLX_AGENT); }

// ---- the block record on the wire: 8 words ------------------------------------------------------------------------
//   w0,w1 base | w2,w3 o[0] | w4 = d1 | d2 << 16, w5 = d3 | meta << 16 (d_s = o[s] - o[0], int16) | w6 lo | w7 hi
//   meta: kind (2 bits) | nside (7) | literal tail (1) | not encodable (1)
__device__ __forceinline__ void fz_encode(const BlockRec& br, uint32_t (&w)[FZ_KREC])
{
    const Rec& r = br.prefix;
    const long long d1 = (long long)(r.o[1] - r.o[0]), d2 = (long long)(r.o[2] - r.o[0]), d3 = (long long)(r.o[3] - r.o[0]);
    const bool fits = d1 >= -32768 && d1 <= 32767 && d2 >= -32768 && d2 <= 32767 && d3 >= -32768 && d3 <= 32767;
    const uint32_t meta = (uint32_t)(r.kind & 3) | ((uint32_t)(br.nside & 127) << 2) | ((br.tail_from < NT) ? (1u << 9) : 0u) | (fits ? 0u : (1u << 10));
    w[0] = (uint32_t)r.base; w[1] = (uint32_t)(r.base >> 32); w[2] = (uint32_t)r.o[0]; w[3] = (uint32_t)(r.o[0] >> 32);
    w[4] = ((uint32_t)d1 & 0xffffu) | (((uint32_t)d2 & 0xffffu) << 16);
    w[5] = ((uint32_t)d3 & 0xffffu) | (meta << 16);
    w[6] = (uint32_t)r.lo; w[7] = (uint32_t)r.hi;
}
__device__ __forceinline__ BlockRec fz_decode(const uint32_t (&w)[FZ_KREC], bool& unsupported)
{
    BlockRec br;
    Rec& r = br.prefix;
    r.base = mk64(w[0], w[1]); r.o[0] = mk64(w[2], w[3]);
    r.o[1] = r.o[0] + (uint64_t)(long long)(int16_t)(w[4] & 0xffffu);
    r.o[2] = r.o[0] + (uint64_t)(long long)(int16_t)(w[4] >> 16);
    r.o[3] = r.o[0] + (uint64_t)(long long)(int16_t)(w[5] & 0xffffu);
    const uint32_t meta = w[5] >> 16;
    r.kind = (int32_t)(meta & 3); r.lo = (int32_t)w[6]; r.hi = (int32_t)w[7]; r.pad = 0;
    br.nside = (int32_t)((meta >> 2) & 127);
    const bool tail = (meta >> 9) & 1u, unenc = (meta >> 10) & 1u;
    br.tail_from = tail ? 0 : NT;
    unsupported = tail || unenc || br.nside > FZ_MAXSIDE;
    return br;
}

// ---- spins ---------------------------------------------------------------------------------------------------------
struct FzClock { long long t0; __device__ __forceinline__ bool expired() const { return (long long)wall_clock64() - t0 > FZ_TIMEOUT_TICKS; } };

// One wave waits for the NG granules of a slot (lane i < NG reads granule i); returns false on a time-out.  vals: lane i's word.
template <int NG>
__device__ __forceinline__ bool fz_wait_slot(const fz_u64* slot, uint32_t tag, const FzClock& clk, uint32_t& val)
{
    const int lane = threadIdx.x & 63;
    for (unsigned spins = 0;; spins++) {
        bool ok = true;
        if (lane < NG) { const fz_u64 x = fz_get(slot + lane); ok = (uint32_t)(x >> 32) == tag; val = (uint32_t)x; }
        if (__all(ok)) return true;
        if ((spins & 31u) == 31u && clk.expired()) return false;
        __builtin_amdgcn_s_sleep(1);
    }
}

// One wave waits for NG granules, lane i < NG polling the address it was given.
template <int NG>
__device__ __forceinline__ bool fz_wait_words(const fz_u64* mine, uint32_t tag, const FzClock& clk, uint32_t& val)
{
    const int lane = threadIdx.x & 63;
    for (unsigned spins = 0;; spins++) {
        bool ok = true;
        if (lane < NG) { const fz_u64 x = fz_get(mine); ok = (uint32_t)(x >> 32) == tag; val = (uint32_t)x; }
        if (__all(ok)) return true;
        if ((spins & 31u) == 31u && clk.expired()) return false;
        __builtin_amdgcn_s_sleep(1);
    }
}

// ---- sinks of block_record_tail ---------------------------------------------------------------------------------------
// The block's own terms for the in-order fallback live in LDS (nothing of a fused run's weights is in HBM).
// (the lane's terms travel BY VALUE: a pointer to the caller's register array handed to an out-of-line function forces that
//  array into scratch memory for the whole kernel -- 64 B a lane stored by every launch and written back to HBM at its end,
//  8 MiB per observation at N = 2^20, seen as WRITE_SIZE twice the algorithmic bytes)
struct LaneTerms { double a[EL]; };
__device__ __attribute__((noinline)) void fz_block_literal(uint64_t* tin /* LDS [NT] */, uint64_t cin, double* terms /* LDS [EB] */, const LaneTerms v /* this lane's EL terms */,
                                                           DevState* st)
{
    const int t = threadIdx.x;
#pragma unroll
    for (int k = 0; k < EL; k++) terms[t * EL + k] = v.a[k];
    __syncthreads();
    if (t == 0) {
        double c = b2d(cin);
        for (int tt = 0; tt < NT; tt++) {
            tin[tt] = d2b(c);
            for (int k = 0; k < EL; k++) c = c + terms[tt * EL + k];
        }
        atomicAdd((unsigned long long*)&st->stat_serial_walks, 1ull);
        atomicAdd((unsigned long long*)&st->stat_literal_terms, (unsigned long long)EB);
    }
    __syncthreads();
}
__device__ __forceinline__ LaneTerms fz_lane_terms(const double (&v)[EL])
{
    LaneTerms r;
#pragma unroll
    for (int k = 0; k < EL; k++) r.a[k] = v[k];
    return r;
}

struct FusedRecSink {
    fz_u64* rec_planes;            // [FZ_KREC][FZ_MAXB]
    fz_u64* side_base;             // [FZ_MAXB][FZ_MAXSIDE][FZ_KSIDE]
    int bidx; uint32_t tag;
    double* terms; DevState* st;
    __device__ __forceinline__ void rec(const BlockRec& br) const
    {
        uint32_t w[FZ_KREC];
        fz_encode(br, w);
#pragma unroll
        for (int k = 0; k < FZ_KREC; k++) fz_put(rec_planes + (size_t)k * FZ_MAXB + bidx, tag, w[k]);
    }
    __device__ __forceinline__ void side(int k, const SideEntry& e) const
    {
        if (k >= FZ_MAXSIDE) return;                         // (the record carries nside: the resolver stands down)
        uint32_t w[FZ_KSIDE];
        memcpy(w, &e, sizeof(SideEntry));
        fz_u64* dst = side_base + ((size_t)bidx * FZ_MAXSIDE + k) * FZ_KSIDE;
#pragma unroll
        for (int q = 0; q < FZ_KSIDE; q++) fz_put(dst + q, tag, w[q]);
    }
    __device__ __forceinline__ void literal(uint64_t* tin, uint64_t cin, const double (&v)[EL]) const { fz_block_literal(tin, cin, terms, fz_lane_terms(v), st); }
};

// ---- arguments -----------------------------------------------------------------------------------------------------
struct FusedArgs {
    const double* xin;            // particles before this observation's transition ([N])
    long long N; int nblk;
    ModelPar par; double y; NoiseSrc ns; int trans;      // trans 0: obs_times repeats a time, weights on the current particles
    int obs_i, resample_algorithm; double threshold;
    double* ess_out; double* llh_out; int* resampled_out;
    double* w_out;                // normalised weights to HBM (histories), or nullptr
    int lim;
    ApplyArgs a;                  // expansion: xdst, uniforms, ancestors, se_part, ...
    FusedWs* ws; uint32_t tag;    // launch number since the workspace was zeroed (never 0)
    // the NEXT fused launch's transition normals, drawn here while this workgroup waits for the resolver (the generator is
    // counter-based: a draw is a function of (key, call, particle slot)); nullptr: not wanted (last observation, injected draws)
    double* znext; uint32_t znext_call;
};

struct FusedSmem {
    double red[16]; double redq[16];
    uint32_t slot[16];
    int bail;
    double bm, bs_, bq;
};

// ---- the resolver's three duties ---------------------------------------------------------------------------------------
// Gather the NG-granule items of the blocks [c0, c1) this thread is responsible for (c1 - c0 <= 2) from plane-major granules.
template <int NG>
__device__ __forceinline__ bool fz_gather(const fz_u64* planes, int c0, int c1, uint32_t tag, const FzClock& clk, int* bail, uint32_t (&w0)[NG], uint32_t (&w1)[NG])
{
    // two blocks per thread (c0 even): ONE 16-byte sc1 load per plane fetches both granules; each 8-byte half is one store of its
    // producer, checked by its own tag
    // (c0 even and c0 < c1: the 16-byte form always -- also when only the first of the two is wanted (the cumsum duty's last thread), so that
    //  no wave runs both forms one after the other; FZ_MAXB is even, the second granule exists)
    const bool pair = ((c0 & 1) == 0) && (c0 < c1);
    const bool want1 = (c0 + 1 < c1);
    __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)const_cast<fz_u64*>(planes), 0, NG * FZ_MAXB * 8, 0x00020000);
    // a lane whose records have arrived stops polling: the passes that wait for the last few records (the head block, the crossing blocks) then
    // carry a handful of loads instead of 8 x 64 per wave and come back sooner
    bool have = false;
    for (unsigned spins = 0;; spins++) {
        if (!have) {
            bool ok = true;
            if (pair) {
#pragma unroll
                for (int k = 0; k < NG; k++) {
                    const u32x4_t x = __builtin_amdgcn_raw_buffer_load_b128(rs, (k * FZ_MAXB + c0) * 8, 0, 16);        // aux 16 = sc1
                    ok = ok && x.y == tag && (!want1 || x.w == tag); w0[k] = x.x; w1[k] = x.z;
                }
            } else {
#pragma unroll
                for (int k = 0; k < NG; k++) {
                    if (c0 < c1) { const fz_u64 x = fz_get(planes + (size_t)k * FZ_MAXB + c0); ok = ok && (uint32_t)(x >> 32) == tag; w0[k] = (uint32_t)x; }
                    if (c0 + 1 < c1) { const fz_u64 x = fz_get(planes + (size_t)k * FZ_MAXB + c0 + 1); ok = ok && (uint32_t)(x >> 32) == tag; w1[k] = (uint32_t)x; }
                }
            }
            have = ok;
        }
        if (__all(have)) return true;
        if ((spins & 15u) == 15u && clk.expired()) { *bail = 1; return false; }
        if (*(volatile int*)bail) return false;
        __builtin_amdgcn_s_sleep(1);
    }
}

// Copy the side entries of block b (nside of them) from the wire into the private plain list.
__device__ __forceinline__ bool fz_fetch_sides(const fz_u64* side_base, SideList* side_priv, int b, int nside, uint32_t tag, const FzClock& clk)
{
    for (int k = 0; k < nside; k++) {
        const fz_u64* src = side_base + ((size_t)b * FZ_MAXSIDE + k) * FZ_KSIDE;
        uint32_t w[FZ_KSIDE];
        for (unsigned spins = 0;; spins++) {
            bool ok = true;
#pragma unroll
            for (int q = 0; q < FZ_KSIDE; q++) { const fz_u64 x = fz_get(src + q); ok = ok && (uint32_t)(x >> 32) == tag; w[q] = (uint32_t)x; }
            if (ok) break;
            if ((spins & 15u) == 15u && clk.expired()) return false;
            __builtin_amdgcn_s_sleep(1);
        }
        SideEntry e;
        memcpy(&e, w, sizeof(SideEntry));
        side_priv[b].e[k] = e;
    }
    return true;
}

// The last block's single side entry, fetched from the wire by the resolver's wave 0 at the moment the chain walk needs it (lane q polls
// granule q; one round trip, usually none: the entry arrived while the scan ran) into the walk's LDS slot.
struct FzLateSide {
    static constexpr bool active = true;
    const fz_u64* src; uint32_t tag; FzClock clk;
    __device__ __forceinline__ unsigned long long issue() const
    {
        const int lane = threadIdx.x & 63;
        static_assert(FZ_KSIDE <= 64, "one granule per lane");
        return (lane < FZ_KSIDE) ? fz_get(src + lane) : 0ull;
    }
    __device__ __forceinline__ bool finish(SideEntry* dst, unsigned long long x) const
    {
        const int lane = threadIdx.x & 63;
        for (unsigned spins = 0;; spins++) {
            const bool ok = (lane >= FZ_KSIDE) || ((uint32_t)(x >> 32) == tag);
            if (__all(ok)) {
                if (lane < FZ_KSIDE) reinterpret_cast<uint32_t*>(dst)[lane] = (uint32_t)x;
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                return true;
            }
            if ((spins & 15u) == 15u && clk.expired()) return false;
            __builtin_amdgcn_s_sleep(1);
            if (lane < FZ_KSIDE) x = fz_get(src + lane);
        }
    }
};

template <int MODE, bool EMIT>
__device__ __forceinline__ uint64_t fz_resolve_duty(SegSmem& sm, ResolveSmem& rs, FusedSmem& fs, const FusedArgs& g, DevState* st, const FzClock& clk,
                                                    uint64_t* cin_lds /* [FZ_MAXB], EMIT */, bool& bail)
{
    const int t = threadIdx.x, B = g.nblk;
    const fz_u64* planes = (MODE == MODE_W) ? &g.ws->e2[0][0] : &g.ws->e3[0][0];
    const fz_u64* sides = (MODE == MODE_W) ? &g.ws->side_w[0][0][0] : &g.ws->side_p[0][0][0];
    const int CB = (B + NT - 1) / NT;
    // (cumsum pass: nobody needs the state BEHIND the last block, so its record -- the one that nearly always carries a side entry, published
    //  ~0.5 us after the record itself and fetched with a round trip of its own -- is not waited for at all)
    const int Bneed = (MODE == MODE_P) ? B - 1 : B;
    const int c0 = t * CB, c1 = (c0 + CB < Bneed) ? c0 + CB : Bneed;
    uint32_t w0[FZ_KREC], w1[FZ_KREC];
    bool ok = fz_gather<FZ_KREC>(planes, c0, c1, g.tag, clk, &fs.bail, w0, w1);
    bool unsup = false;
    BlockRec q0, q1;
    q0.prefix = rec_identity(0); q0.tail_from = NT; q0.nside = 0; q1 = q0;
    if (ok) {
        // (the records stay in registers; only side entries -- the last block's, rarely another -- go through the private list)
        // (the LAST block's single side entry is not waited for here: the walk fetches it when it gets there -- FzLateSide)
        if (c0 < c1) {
            bool u; q0 = fz_decode(w0, u); unsup = unsup || u;
            if (!u && q0.nside > 0 && !(c0 == B - 1 && q0.nside == 1 && q0.tail_from >= NT)) {
#ifdef BSSM_DEV_STAMPS
                g.ws->pubt[3][20 + MODE] = c0; atomicAdd((unsigned long long*)&g.ws->pubt[3][22 + MODE], 1ull);
#endif
                ok = ok && fz_fetch_sides(sides, g.ws->side_priv[MODE], c0, q0.nside, g.tag, clk);
            }
        }
        if (c0 + 1 < c1) {
            bool u; q1 = fz_decode(w1, u); unsup = unsup || u;
            if (!u && q1.nside > 0 && !(c0 + 1 == B - 1 && q1.nside == 1 && q1.tail_from >= NT)) {
#ifdef BSSM_DEV_STAMPS
                g.ws->pubt[3][20 + MODE] = c0 + 1; atomicAdd((unsigned long long*)&g.ws->pubt[3][22 + MODE], 1ull);
#endif
                ok = ok && fz_fetch_sides(sides, g.ws->side_priv[MODE], c0 + 1, q1.nside, g.tag, clk);
            }
        }
    }
    // (a wave that saw an unsupported record may have made the other waves leave their gather early: the larger code wins)
    if (unsup) atomicMax(&fs.bail, 2); else if (!ok) atomicMax(&fs.bail, 1);
    __syncthreads();                      // (also: this workgroup's plain stores above are visible to its own loads below)
#ifdef BSSM_DEV_STAMPS
    if (t == 0) { g.ws->stamps[1][MODE == MODE_W ? 8 : 13] = clock64(); g.ws->pubt[3][8 + (MODE == MODE_W ? 1 : 2)] = (long long)wall_clock64(); }
#endif
    if (fs.bail) { bail = true; return 0ull; }
    long long lit = 0;
    // the terms of OTHER blocks are not in HBM in a fused run: a literal re-run that reads them (lit >= LIT_FROM_W) voids the result;
    // terms re-run from a side entry's own copy are fine
    // (cumsum pass: nobody needs the state BEHIND the last block -- its record, the one with the lanes next to cum == 1, stays out of the
    //  walk; the state in front of it is the walk's result)
    const int upto = (MODE == MODE_P) ? B - 1 : B;
    FzLateSide late; late.src = sides + ((size_t)(B - 1) * FZ_MAXSIDE + 0) * FZ_KSIDE; late.tag = g.tag; late.clk = clk;
    const uint64_t fin = resolve_in_block<MODE, NT, EMIT, FzLateSide>(sm, rs, g.ws->brec_priv[MODE], g.ws->side_priv[MODE], B, upto, g.xin, g.N, 1.0, st, true, cin_lds, &lit, &q0, &q1, late);
    if (EMIT && t == 0) cin_lds[B - 1] = fin;
    if (lit >= LIT_FROM_W) fs.bail = 2;
    __syncthreads();
#ifdef BSSM_DEV_STAMPS
    if (t == 0) g.ws->stamps[1][MODE == MODE_W ? 9 : 14] = clock64();
#endif
    if (fs.bail) { bail = true; return 0ull; }
    return fin;
}

__host__ __device__ __forceinline__ int fz_resolver_block(int B) { return (5 * B) / 16; }

// The normals of transition call `call` for this lane's EL slots, to HBM (the next launch reads them as it reads injected draws).
__device__ __forceinline__ void fz_draw_next(const FusedArgs& g, long long j0, long long N, const int prio_after = 1)
{
    if (!g.znext) return;
    __builtin_amdgcn_s_setprio(0);
#pragma unroll
    for (int i = 0; i < EL / 2; i++) {
        const long long j = j0 + 2 * i;
        if (j < N) {
            double z0, z1;
            normal_pair(g.ns.key, g.ns.purpose, g.znext_call, 0, (uint32_t)(j >> 1), z0, z1);
            if (j + 1 < N) bulk_store16<1>(g.znext + j, z0, z1);      // write-through: nothing of it is left dirty in L2 for the end-of-kernel write-back
            else g.znext[j] = z0;
        }
    }
    if (prio_after == 3) __builtin_amdgcn_s_setprio(3); else __builtin_amdgcn_s_setprio(1);
}

// ---- the kernel ----------------------------------------------------------------------------------------------------
template <int MODEL, int KIND>
__global__ __launch_bounds__(NT) void k_obs(FusedArgs g, DevState* st)
{
    __shared__ SegSmem sm;
    __shared__ uint64_t tin[NT + 1];
    __shared__ FusedSmem fs;
    __shared__ int Tbegin;
    extern __shared__ __attribute__((aligned(16))) double dyn[];      // workers: Tl [EB] ints | lx [CAPX] doubles; resolver duty: ResolveSmem | cin
    int* Tl = reinterpret_cast<int*>(dyn);
    double* lx = dyn + EB / 2;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int bidx = (int)blockIdx.x, B = g.nblk;
    // the resolver: the worker of a block that (a) is dispatched in the first round of workgroups -- the older of the two
    // workgroups of its CU wins the VALU arbitration -- and (b) does not sit where the cumulative weight crosses a power of two
    // (blocks B/2, B/4, ...: their records take the longer path): block 5B/16, at cum ~ 0.31
    const bool resolver = (bidx == fz_resolver_block(B));
    // wave priorities: the resolver above everybody (its duties are on every workgroup's critical path), ordinary work 1, the
    // next observation's normals (fz_draw_next, drawn while a workgroup waits) 0: they only take VALU slots nobody else wants
    if (resolver) __builtin_amdgcn_s_setprio(3); else __builtin_amdgcn_s_setprio(1);
    const bool fz_stamper = (t == 0) && (resolver || bidx == ((B > 100) ? 100 : 0)); const int fz_row = resolver ? 1 : 0; (void)fz_stamper; (void)fz_row;
    FZ_STAMP(0);
#ifdef BSSM_DEV_STAMPS
    if (t == 0) g.ws->startt[bidx] = (long long)wall_clock64();
#define FZ_PUBT(k) do { if (t == 0) g.ws->pubt[k][bidx] = (long long)wall_clock64(); } while (0)
#define FZ_GDONE(k) do { if (t == 0) g.ws->pubt[3][8 + (k)] = (long long)wall_clock64(); } while (0)
#define FZ_WAKE(k) do { if (t == 0) g.ws->waket[k][bidx] = (long long)wall_clock64(); } while (0)
#else
#define FZ_PUBT(k) do { } while (0)
#define FZ_GDONE(k) do { } while (0)
#define FZ_WAKE(k) do { } while (0)
#endif
    const long long b0 = (long long)bidx * EB, j0 = b0 + (long long)t * EL;
    const long long N = g.N;
    FzClock clk; clk.t0 = (long long)wall_clock64();
    // run state, as the multi-launch kernels read it at their start (nothing in this launch changes it before every workgroup has read it:
    // the first writer is the resolver's bookkeeping, which needs every workgroup's first record)
    const int s_dead = st->dead, call = st->res_calls;
    const uint32_t s_flags = st->flags;
    if (t == 0) fs.bail = 0;
    // ---- phase A: transition_fn + weight_fn (R/particle_filter_core.R:127,177-183) on this lane's EL consecutive particles ----
    double x8[EL], l8[EL];
    if (j0 + EL <= N) {
        const double2* p = reinterpret_cast<const double2*>(g.xin + j0);
#pragma unroll
        for (int k = 0; k < EL / 2; k++) { const double2 q = p[k]; x8[2 * k] = q.x; x8[2 * k + 1] = q.y; }
    } else {
#pragma unroll
        for (int k = 0; k < EL; k++) x8[k] = (j0 + k < N) ? g.xin[j0 + k] : 0.0;
    }
    // the transition normals first: the generator needs nothing from memory, so its ~1 200 vector instructions run while the particle and
    // run-state loads above are in flight (the early exit below is the first thing that needs a loaded value)
    double z8[EL];
#pragma unroll
    for (int i = 0; i < EL / 2; i++) {
        const long long j = j0 + 2 * i;
        z8[2 * i] = 0.0; z8[2 * i + 1] = 0.0;
        if (g.trans && j < N) {
            if (g.ns.arr) { if (j + 1 < N) { const double2 q = *reinterpret_cast<const double2*>(g.ns.arr + j); z8[2 * i] = q.x; z8[2 * i + 1] = q.y; } else { z8[2 * i] = g.ns.arr[j]; } }
            else normal_pair(g.ns.key, g.ns.purpose, g.ns.call, 0, (uint32_t)(j >> 1), z8[2 * i], z8[2 * i + 1]);
        }
    }
    if (s_dead || s_flags) return;
#pragma unroll
    for (int i = 0; i < EL / 2; i++) {
        const long long j = j0 + 2 * i;
        if (g.trans && j < N) {
            x8[2 * i] = Model<MODEL>::transition(x8[2 * i], z8[2 * i], g.par);
            x8[2 * i + 1] = Model<MODEL>::transition(x8[2 * i + 1], z8[2 * i + 1], g.par);
        }
        l8[2 * i] = (j < N) ? r_dnorm_log(g.y, x8[2 * i], g.par.sy, g.par.log_sy) : -INFINITY;
        l8[2 * i + 1] = (j + 1 < N) ? r_dnorm_log(g.y, x8[2 * i + 1], g.par.sy, g.par.log_sy) : -INFINITY;
    }
    FZ_STAMP(1);
    // block (max, sum exp, sum exp^2) in k_step's association order: a k_step thread holds one PAIR, 64 of them make a wave
    // (= one row of 16 lanes here), the 16 wave totals are added in order
    double bm;
    {
        double m = l8[0];
#pragma unroll
        for (int k = 1; k < EL; k++) m = fmax(m, l8[k]);
        m = wave_max(m);
        __syncthreads();
        if (lane == 0) fs.red[wave] = m;
        __syncthreads();
        bm = fs.red[0];
#pragma unroll
        for (int i = 1; i < NWV; i++) bm = fmax(bm, fs.red[i]);
    }
    double bsum, bsq;
    {
        double sv[EL / 2], qv[EL / 2];
#pragma unroll
        for (int i = 0; i < EL / 2; i++) {
            double s_ = 0.0, q_ = 0.0;
            if (bm > -INFINITY) {
                if (l8[2 * i] > -INFINITY) { const double e = exp(l8[2 * i] - bm); s_ += e; q_ += e * e; }
                if (l8[2 * i + 1] > -INFINITY) { const double e = exp(l8[2 * i + 1] - bm); s_ += e; q_ += e * e; }
            }
            sv[i] = s_; qv[i] = q_;
        }
        static_assert(EL == 8, "four k_step pairs per lane");
        double s4 = (sv[0] + sv[1]) + (sv[2] + sv[3]), q4 = (qv[0] + qv[1]) + (qv[2] + qv[3]);
#define STEP(C) s4 = dpp_f64<C, 0xf>(0.0, s4) + s4; q4 = dpp_f64<C, 0xf>(0.0, q4) + q4;
        STEP(DPP_SHR1) STEP(DPP_SHR2) STEP(DPP_SHR4) STEP(DPP_SHR8)
#undef STEP
        __syncthreads();
        if ((lane & 15) == 15) { fs.red[4 * wave + (lane >> 4)] = s4; fs.redq[4 * wave + (lane >> 4)] = q4; }
        __syncthreads();
        double s_ = 0.0, q_ = 0.0;
#pragma unroll
        for (int i = 0; i < 16; i++) { s_ += fs.red[i]; q_ += fs.redq[i]; }
        bsum = s_; bsq = q_;
    }
    if (t < 6) {
        const uint64_t bits = d2b(t < 2 ? bm : (t < 4 ? bsum : bsq));
        fz_put(&g.ws->e1[t][bidx], g.tag, (t & 1) ? (uint32_t)(bits >> 32) : (uint32_t)bits);
    }
    FZ_STAMP(2); FZ_PUBT(0);
    // ---- resolver duty 1: log-sum-exp combine, log-likelihood, ESS, resample decision (:204-218), block prefixes ----
    if (resolver) {
        const int L = (B + NT - 1) / NT;
        const int c0 = t * L, c1 = (c0 + L < B) ? c0 + L : B;
        uint32_t w0[6], w1[6];
        const bool ok = fz_gather<6>(&g.ws->e1[0][0], c0, c1, g.tag, clk, &fs.bail, w0, w1);
        if (!ok) fs.bail = 1;
        __syncthreads();
        FZ_STAMP(3); FZ_GDONE(0);
        uint32_t status = 0;
        double M = 0.0, S = 0.0, sq = 0.0, ess1 = 0.0;
        bool book = false;
        double esv[2] = {0.0, 0.0}, pre = 0.0;
        if (!fs.bail) {
            double pmv[2], psv[2], pqv[2];
            pmv[0] = (c0 < c1) ? b2d(mk64(w0[0], w0[1])) : -INFINITY; psv[0] = (c0 < c1) ? b2d(mk64(w0[2], w0[3])) : 0.0; pqv[0] = (c0 < c1) ? b2d(mk64(w0[4], w0[5])) : 0.0;
            pmv[1] = (c0 + 1 < c1) ? b2d(mk64(w1[0], w1[1])) : -INFINITY; psv[1] = (c0 + 1 < c1) ? b2d(mk64(w1[2], w1[3])) : 0.0; pqv[1] = (c0 + 1 < c1) ? b2d(mk64(w1[4], w1[5])) : 0.0;
            M = block_max(fmax(pmv[0], pmv[1]), sm.sh4);
            const bool degenerate = (M < -1e8);                                              // all(log_weights < -1e8)  (:189-202)
            if (degenerate) {
                if (t == 0) { st->loglike = -INFINITY; g.llh_out[g.obs_i - 1] = -INFINITY; st->dead = g.obs_i; st->do_resample = 0; }
                status = FZ_ST_DEAD;
            } else {
                double ts0 = 0.0, tq = 0.0;
#pragma unroll
                for (int k = 0; k < 2; k++) {
                    double x = 0.0;
                    if (pmv[k] > -INFINITY) { const double ex = exp(pmv[k] - M); x = psv[k] * ex; tq += pqv[k] * ex * ex; }
                    esv[k] = x; ts0 += x;
                }
                const double inc = wave_incl_sum(ts0);
                const double exc = dpp_f64<DPP_WAVE_SHR1, 0xf>(0.0, inc);
                const double wq = wave_sum(tq);
                __syncthreads();
                if (lane == 63) { sm.sh4[wave] = inc; sm.sh4[8 + wave] = wq; }
                __syncthreads();
                pre = exc;
                for (int i = 0; i < wave; i++) pre += sm.sh4[i];
                S = tree_sum<NWV>(sm.sh4);
                sq = tree_sum<NWV>(sm.sh4 + 8);
                ess1 = 1.0 / (sq / (S * S));                                                      // :211
                const int doit = (g.resample_algorithm == 0) ? 0 : (g.resample_algorithm == 1) ? 1 : (ess1 < g.threshold);   // :214-218
                book = true;                   // (the run-state bookkeeping follows the answers: the workers are waiting for those)
                status = doit ? FZ_ST_DOIT : 0u;
                if (doit && (!(S > 0.0) || !isfinite(S) || !isfinite(M))) {                      // NaN/Inf log-weights: the scan stands down
                    if (t == 0) atomicOr(&st->flags, FLAG_NONFINITE);
                    status |= FZ_ST_STOP;
                }
            }
        } else {
            status = FZ_ST_BAIL;
            if (t == 0) atomicOr(&st->flags, FLAG_FUSED_TIMEOUT);
        }
        {
            double pp = pre;
#pragma unroll
            for (int k = 0; k < 2; k++) {
                const int i = c0 + k;
                if (k < L && i < c1) fz_put64x(&g.ws->a1[0][i], FZ_MAXB, g.tag, d2b(pp / S));
                pp += esv[k];
            }
            if (t < FZ_NREP) {
                fz_u64* rep = &g.ws->c1[t][0];
                fz_put64(rep + 0, g.tag, d2b(M)); fz_put64(rep + 2, g.tag, d2b(S)); fz_put64(rep + 4, g.tag, d2b(sq)); fz_put(rep + 6, g.tag, status);
            }
        }
        if (book && t == 0) {
            // log-likelihood, ESS, decision into the run state (:208-223) -- behind the answers: st->loglike is a load from HBM
            // and the two logs are ~100 instructions of one lane, neither of which the workers need
            const int doit = (status & FZ_ST_DOIT) ? 1 : 0;
            const double ll = st->loglike + (M + log(S) - log((double)N));                        // :208
            st->loglike = ll; g.llh_out[g.obs_i - 1] = ll;                                        // :209
            st->do_resample = doit;
            g.ess_out[g.obs_i] = doit ? (double)N : ess1;                                         // :212,:223
            if (g.resampled_out) g.resampled_out[g.obs_i - 1] = doit;
            st->ess = ess1; st->lse_max = M; st->lse_sum = S;
            if (doit) { st->cur_call = call; st->res_calls = call + 1; }
        }
    }
    if (resolver) FZ_STAMP(4);
    // ---- every worker: its slot of duty 1 ----
    double M, S, a_in;
    uint32_t status;
    {
        if (wave == 0) {
            uint32_t v = 0;
            const fz_u64* src = (lane < 7) ? &g.ws->c1[bidx % FZ_NREP][lane] : &g.ws->a1[(lane - 7) & 1][bidx];
            const bool ok = fz_wait_words<9>(src, g.tag, clk, v);
            if (lane < 9) fs.slot[lane] = v;
            if (!ok && lane == 0) fs.bail = 1;
        }
        __syncthreads();
        if (fs.bail) { if (t == 0 && fs.bail == 1) atomicOr(&st->flags, FLAG_FUSED_TIMEOUT); return; }
        M = b2d(mk64(fs.slot[0], fs.slot[1])); S = b2d(mk64(fs.slot[2], fs.slot[3])); a_in = b2d(mk64(fs.slot[7], fs.slot[8]));
        status = fs.slot[6];
    }
    FZ_STAMP(5); FZ_WAKE(0);
    if (status & (FZ_ST_DEAD | FZ_ST_BAIL)) return;
    const bool doit = status & FZ_ST_DOIT;
    // ---- phase B: w = exp(lw - max) / sum (:205-207) ----
    double v[EL];
#pragma unroll
    for (int k = 0; k < EL; k++) v[k] = (j0 + k < N) ? exp(l8[k] - M) / S : 0.0;
    if (g.w_out) {
        if (j0 + EL <= N) {
#pragma unroll
            for (int k = 0; k < EL / 2; k++) bulk_store16<BSSM_ST_W>(g.w_out + j0 + 2 * k, v[2 * k], v[2 * k + 1]);
        } else {
#pragma unroll
            for (int k = 0; k < EL; k++) if (j0 + k < N) g.w_out[j0 + k] = v[k];
        }
    }
    if (!doit) {
        // no resampling at this observation: the particles carry over, state estimate = sum(particles * weights) (:238), summed as
        // k_carry sums it (lane t: elements t, t + NT, ...)
        double* xa = lx; double* wa = lx + EB;
        __syncthreads();
#pragma unroll
        for (int k = 0; k < EL; k++) { xa[t * EL + k] = x8[k]; wa[t * EL + k] = v[k]; }
        __syncthreads();
        double acc0 = 0.0;
#pragma unroll
        for (int r = 0; r < EL; r++) {
            const long long j = b0 + t + NT * r;
            if (j < N) { const double x0 = xa[t + NT * r]; g.a.xdst[j] = x0; acc0 += x0 * wa[t + NT * r]; }
        }
        acc0 = block_sum(acc0, sm.sh4);
        if (t == 0) g.a.se_part[bidx] = acc0;
        fz_draw_next(g, j0, N);
        return;
    }
    if (status & FZ_ST_STOP) return;
    FZ_STAMP(6);
    // ---- W pass: the block's record of the exact sequential sum(w) (src/resampling.cpp:20,47) ----
    FusedRecSink sink; sink.bidx = bidx; sink.tag = g.tag; sink.terms = lx; sink.st = st;
    {
        BlockScan bsw;
        block_scan<MODE_W>(sm, v, a_in, g.lim, bsw);
        __syncthreads();
        sink.rec_planes = &g.ws->e2[0][0]; sink.side_base = &g.ws->side_w[0][0][0];
        block_record_tail<MODE_W, 11>(sm, tin, bsw, v, a_in, g.lim, st, 0, false, sink);
    }
    FZ_STAMP(7); FZ_PUBT(1);
    if (!resolver) fz_draw_next(g, j0, N);          // (the resolver draws behind its duty, below)
    ResolveSmem& rs = *reinterpret_cast<ResolveSmem*>(dyn);
    uint64_t* cin_lds = reinterpret_cast<uint64_t*>(reinterpret_cast<char*>(dyn) + ((sizeof(ResolveSmem) + 15) / 16) * 16);
    if (resolver) {
        // ---- duty 2: exact total = sum(w) ----
        __syncthreads();
        bool bail = false;
        const uint64_t fin = fz_resolve_duty<MODE_W, false>(sm, rs, fs, g, st, clk, nullptr, bail);
        uint32_t stw = 0;
        if (bail) { stw = FZ_ST_BAIL; if (t == 0) atomicOr(&st->flags, fs.bail == 1 ? FLAG_FUSED_TIMEOUT : FLAG_FUSED_BAIL); }
        else {
            const double tot = b2d(fin);
            if (t == 0) {
                st->total_bits = fin;
                if (tot == 0.0) atomicOr(&st->flags, FLAG_ZERO_SUM);       // src/resampling.cpp:8,22,49
                if (!isfinite(tot)) atomicOr(&st->flags, FLAG_NONFINITE);
            }
            if (tot == 0.0 || !isfinite(tot)) stw = FZ_ST_STOP;
        }
        if (t < FZ_NREP) { fz_put64(&g.ws->c2[t][0], g.tag, fin); fz_put(&g.ws->c2[t][2], g.tag, stw); }
        fz_draw_next(g, j0, N);
        FZ_STAMP(10);
    }
    double total;
    {
        __syncthreads();
        if (wave == 0) {
            uint32_t vv = 0;
            const bool ok = fz_wait_slot<3>(&g.ws->c2[bidx % FZ_NREP][0], g.tag, clk, vv);
            if (lane < 3) fs.slot[lane] = vv;
            if (!ok && lane == 0) fs.bail = 1;
        }
        __syncthreads();
        if (fs.bail) { if (t == 0 && fs.bail == 1) atomicOr(&st->flags, FLAG_FUSED_TIMEOUT); return; }
        if (fs.slot[2] & (FZ_ST_BAIL | FZ_ST_STOP)) return;
        total = b2d(mk64(fs.slot[0], fs.slot[1]));
    }
    FZ_STAMP(11); FZ_WAKE(1);
    // ---- P pass: prob = w / total, the block's record of cumsum(prob) (src/resampling.cpp:24-25,51-52) ----
    const double a_in_p = a_in / total;
#pragma unroll
    for (int k = 0; k < EL; k++) v[k] = v[k] / total;
    BlockScan bs;
#ifdef BSSM_DEV_STAMPS
    const bool xstamper = (t == 0 && bidx == B / 4);       // a block with a binade crossing (cum ~ 1/4)
    const int xdbg = 99;
    if (xstamper) st->stamps[2][10] = clock64();
#else
    const bool xstamper = false; const int xdbg = 0;
#endif
    block_scan<MODE_P>(sm, v, a_in_p, g.lim, bs);
    __syncthreads();
#ifdef BSSM_DEV_STAMPS
    if (xstamper) st->stamps[2][11] = clock64();
#endif
    sink.rec_planes = &g.ws->e3[0][0]; sink.side_base = &g.ws->side_p[0][0][0];
    block_record_tail<MODE_P, 3>(sm, tin, bs, v, a_in_p, g.lim, st, xdbg, xstamper, sink);
#ifdef BSSM_DEV_STAMPS
    if (xstamper) { st->stamps[2][9] = clock64(); st->stamps[2][8] = bs.nb; }
#endif
    FZ_STAMP(12); FZ_PUBT(2);
    uint64_t cinb = 0;
    if (resolver) {
        // ---- duty 3: every block's exact incoming state ----
        __syncthreads();
        bool bail = false;
        (void)fz_resolve_duty<MODE_P, true>(sm, rs, fs, g, st, clk, cin_lds, bail);
        uint32_t stp = 0;
        if (bail) { stp = FZ_ST_BAIL; if (t == 0) atomicOr(&st->flags, fs.bail == 1 ? FLAG_FUSED_TIMEOUT : FLAG_FUSED_BAIL); }
        for (int b = t; b < B; b += NT) fz_put64x(&g.ws->a3[0][b], FZ_MAXB, g.tag, bail ? 0ull : cin_lds[b]);
        if (t < FZ_NREP) fz_put(&g.ws->c3[t][0], g.tag, stp);
        FZ_STAMP(15);
        if (bail) return;
        cinb = cin_lds[bidx];
        __syncthreads();                  // (the staging area is about to be reused by this workgroup's own expansion)
    } else {
        __syncthreads();
        if (wave == 0) {
            uint32_t vv = 0;
            const fz_u64* src = (lane < 2) ? &g.ws->a3[lane & 1][bidx] : &g.ws->c3[bidx % FZ_NREP][0];
            const bool ok = fz_wait_words<3>(src, g.tag, clk, vv);
            if (lane < 3) fs.slot[lane] = vv;
            if (!ok && lane == 0) fs.bail = 1;
        }
        __syncthreads();
        if (fs.bail) { if (t == 0 && fs.bail == 1) atomicOr(&st->flags, FLAG_FUSED_TIMEOUT); return; }
        if (fs.slot[2] & FZ_ST_BAIL) return;
        cinb = mk64(fs.slot[0], fs.slot[1]);
    }
    FZ_STAMP(16); FZ_WAKE(2);
    // ---- phase D: exact cum_sum, output counts, ancestors, particles[indices, ] (src/resampling.cpp:28-37,55-63) ----
    uint64_t ent;
    {
        const bool good = block_resolve<MODE_P>(sm, bs, cinb, g.lim, nullptr, 0, total, b0, ent);
        if (!good) { fz_block_literal(tin, cinb, lx, fz_lane_terms(v), st); ent = tin[t]; }
    }
    FZ_STAMP(17);
    // the particles in the expansion's lane-interleaved layout (element wave * 64 * EL + 64 k + lane), through LDS
    double xs0[EL], xs1[EL], axs[EL];
    __syncthreads();
#pragma unroll
    for (int k = 0; k < EL; k++) lx[t * EL + k] = x8[k];
    __syncthreads();
    {
        const int ebase = wave * (64 * EL) + lane;
#pragma unroll
        for (int k = 0; k < EL; k++) { xs0[k] = lx[ebase + 64 * k]; xs1[k] = 0.0; axs[k] = 0.0; }
    }
    if (t == 0) sm.big = 0;
    UniformSrc us;
    us.arr = g.a.u_base ? g.a.u_base + (long long)call * g.a.u_stride : nullptr;
    us.key = g.a.key; us.call = (uint32_t)call;
    const double Usys = (KIND == 1) ? us(0) : 0.0;
    // (staging: the coalesced store goes through lx + EB, so that the block's own particles in lx[0, EB) stay readable for the
    //  degenerate-weights path, which does not stage)
    __syncthreads();
    apply_tail<KIND, false, CAPX, false>(sm, Tl, Tbegin, bidx, B, g.a, st, g.a.nstage ? lx + EB : nullptr, g.a.nstage, v, ent, xs0, xs1, axs, call, us, Usys, 0, false, lx);
    FZ_STAMP(18);
#ifdef BSSM_DEV_STAMPS
    if (t == 0) g.ws->pubt[3][32 + (bidx & 255)] = 0;      // (keeps the array referenced)
    __syncthreads();
    if (t == 0) g.ws->endt[bidx] = (long long)wall_clock64();
#endif
}

constexpr size_t FZ_DYN_LDS = (size_t)EB * sizeof(int) + (size_t)(EB + CAPX) * sizeof(double);
static_assert(((sizeof(ResolveSmem) + 15) / 16) * 16 + FZ_MAXB * sizeof(uint64_t) <= FZ_DYN_LDS, "the resolver's scratch borrows the workers' staging area");

}  // namespace bssm
