// bssm_api.hip -- C-ABI implementation (include/bayesssm_amd.h) over the kernels.
//
// Host-side sequencing of the reference's loops:
//   bssm_pf_run      = .particle_filter_core's T-loop       R/particle_filter_core.R:123-246
//   bssm_pmmh_chain  = chain_result's MH loop               R/pmmh.R:403-415,422-500
// One context = one GPU + one stream; device memory is owned by the context.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <math.h>
#include <stdio.h>
#include <string.h>
#include <string>
#include <vector>
#include <map>
#include <algorithm>
#include "kernels.hip.h"
#include "fused.hip.h"
#include "mv.hip.h"
#include "multi.hip.h"
#include <atomic>
#include "../../include/bayesssm_amd.h"

using namespace bssm;

static thread_local std::string g_err;

#define HIPCHK(expr)                                                                      \
    do {                                                                                  \
        hipError_t e__ = (expr);                                                          \
        if (e__ != hipSuccess) {                                                          \
            g_err = std::string(#expr) + ": " + hipGetErrorString(e__);                   \
            return BSSM_ERR_HIP;                                                          \
        }                                                                                 \
    } while (0)

#define ARGFAIL(msg) do { g_err = (msg); return BSSM_ERR_ARG; } while (0)

struct ProfEntry { double ms = 0; long long launches = 0; };

struct bssm_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    long long cap = 0;          // particles
    int max_dim = 1;
    int maxB = 0;
    // particle state
    double *x0 = nullptr, *x1 = nullptr, *lw = nullptr, *w = nullptr, *auxlw = nullptr, *auxg = nullptr, *cum = nullptr;
    // scan workspace
    double *pm = nullptr, *ps = nullptr, *pq = nullptr, *bsum = nullptr, *bsq = nullptr, *ain_w = nullptr, *ain_p = nullptr;
    BlockRec *brec = nullptr, *brec_p = nullptr;      // block records of the sum(w) pass / of the cumsum(w / total) pass
    SideList *side = nullptr, *side_p = nullptr;
    uint64_t* cin = nullptr;
    DevState* st = nullptr;
    unsigned long long* gmax_cur = nullptr;   // slot of the grid-wide max(log-weights) of the weight evaluation in flight
    int sh_boff = 0, sh_nloc = 0;  // particle-block sharding (bssm_pf_run_sharded): this rank's first block / block count; 0 = whole grid
    // per-context options (bssm_ctx_set_option): test aids and A/B switches -- no process-global state
    int opt_window = 0;            // > 0: override the validity window of the scan records (ulps); a tiny window drives the literal fallbacks
    int opt_batch_lit_max = 384;   // largest N that takes the in-order exact sums in k_pf_batch
    int opt_stage = 1;             // LDS staging of k_apply's particle stores
    int opt_inkernel_resolve = 1;  // grids of <= 2 NT blocks: resolve inside the consuming kernels instead of k_resolve launches
    int opt_renormalize = 1;       // filters: 1 = the resampler divides the normalised weights by their exact sum again (src/resampling.cpp:24,51),
                                   // 0 = that division (by 1 +- a few 1e-14) is folded away: one exact pass instead of two
    int opt_recompute_lw = 1;      // bootstrap filters on the Gaussian-observation models: log-weights re-evaluated in k_weights instead of stored by k_step
    int opt_fuse_step = 0;         // SISR bootstrap filters: the next observation's transition + weight inside the expansion kernel
                                   // (off: measured slower -- 7 generator pairs per lane at 2 waves per SIMD cost the expansion kernel 7.7 us,
                                   //  the k_step launch they replace costs 11.2 us but the per-block partials still need a 5.4 us launch)
    int opt_debug_stop = 0;        // DEV builds: stage stamps (99 typical block, 98 head block, 97 batched kernel)
    int opt_fused_prefetch = 0;    // fused path: the next observation's transition normals are drawn while the workgroups wait for the resolver
    int opt_fused = 1;             // bootstrap filters on the scalar Gaussian-observation models, N <= 2^20: one launch per observation (fused.hip.h)
    // fused path: workspace of the tagged records, launch counter (the tags), what happened
    FusedWs* fz = nullptr; uint32_t fz_tag = 0; bool fz_ok = false;
    long long fz_launches = 0, fz_runs = 0, fz_bails = 0, fz_timeouts = 0;
    // growable buffers
    std::map<std::string, std::pair<void*, size_t>> pool;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    void* h_stage = nullptr; size_t h_stage_bytes = 0;      // pinned host staging (batched filters: one upload, one download)
    // profiling
    bool profile = false;
    std::map<std::string, ProfEntry> prof;
    std::vector<std::pair<std::string, std::pair<hipEvent_t, hipEvent_t>>> prof_pending;
    std::vector<hipEvent_t> ev_pool;
    int last_device_status = 0;
};

// One fused run at a time per device (in this process): all workgroups of a fused launch must be resident together, and two
// such launches from two contexts could each hold half of the chip.  A run that does not get the token takes the multi-launch path.
static std::atomic<int> g_fused_busy[64];
// ... and a fused launch fills every residency slot of the chip, which starves the kernels of other contexts running at the same
// time (measured: two / four filter runs in flight 20.0 / 26.3 G particle-steps/s with one of them fused against 29.8 / 33 G all
// multi-launch).  So: runs in flight per device are counted, and after any overlap the next FZ_QUIET runs of the device stay multi-launch.
static std::atomic<int> g_runs_active[64];
static std::atomic<long long> g_run_serial[64], g_last_overlap[64];
static std::atomic<long long> g_fused_hold[64], g_fused_span[64];       // back-off after a time-out: no fused run before serial `hold`; `span` doubles while they keep coming
constexpr long long FZ_QUIET = 16;

static int pool_get(bssm_ctx* c, const char* name, size_t bytes, void** out)
{
    auto& e = c->pool[name];
    if (e.second < bytes) {
        if (e.first) HIPCHK(hipFree(e.first));
        e.first = nullptr; e.second = 0;
        const size_t want = bytes + bytes / 4 + 256;
        HIPCHK(hipMalloc(&e.first, want));
        e.second = want;
    }
    *out = e.first;
    return BSSM_OK;
}

static int host_stage(bssm_ctx* c, size_t bytes)
{
    if (c->h_stage_bytes >= bytes) return BSSM_OK;
    if (c->h_stage) HIPCHK(hipHostFree(c->h_stage));
    c->h_stage = nullptr; c->h_stage_bytes = 0;
    const size_t want = bytes + bytes / 4 + 4096;
    HIPCHK(hipHostMalloc(&c->h_stage, want, hipHostMallocDefault));
    c->h_stage_bytes = want;
    return BSSM_OK;
}

extern "C" const char* bssm_last_error(void) { return g_err.c_str(); }

extern "C" const char* bssm_status_string(int status)
{
    switch (status) {
        case BSSM_OK: return "ok";
        case BSSM_ERR_NEGATIVE_WEIGHT: return "Weights must be non-negative";
        case BSSM_ERR_ZERO_SUM: return "Sum of weights must be greater than 0";
        case BSSM_ERR_LENGTH: return "Number of particles must match the length of weights";
        case BSSM_ERR_ARG: return "invalid argument";
        case BSSM_ERR_HIP: return "HIP runtime error";
        case BSSM_ERR_CAPACITY: return "problem exceeds context capacity";
        default: return "unknown status";
    }
}

extern "C" int bssm_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

extern "C" int bssm_ctx_create(int device, long long max_particles, int max_dim, bssm_ctx** out)
{
    if (!out) ARGFAIL("bssm_ctx_create: out is NULL");
    *out = nullptr;
    if (max_particles <= 0) ARGFAIL("bssm_ctx_create: max_particles must be positive");
    if (max_dim < 1 || max_dim > MVD) ARGFAIL("bssm_ctx_create: max_dim must be 1 .. 8");
    const long long B = (max_particles + EB - 1) / EB;
    if (B > MAXB) { g_err = "bssm_ctx_create: max_particles exceeds 2^22 (scan workspace limit of this build)"; return BSSM_ERR_CAPACITY; }
    int ndev = 0;
    HIPCHK(hipGetDeviceCount(&ndev));
    if (ndev <= 0) { g_err = "no HIP device visible: this library has no CPU path"; return BSSM_ERR_HIP; }
    if (device < 0 || device >= ndev) ARGFAIL("bssm_ctx_create: device index out of range");
    HIPCHK(hipSetDevice(device));
    bssm_ctx* c = new bssm_ctx();
    c->device = device; c->cap = max_particles; c->max_dim = max_dim; c->maxB = (int)B;
    const size_t npad = (size_t)B * EB;
    hipError_t e = hipSuccess;
    // zero-fill ON THE CONTEXT'S STREAM: the stream is non-blocking, so a hipMemset on the null stream is not ordered with the
    // context's first kernels and could land after them (seen once as wrong weights in the first call of a fresh 2^20 context)
    auto A = [&](void** p, size_t bytes) { if (e == hipSuccess) { e = hipMalloc(p, bytes); if (e == hipSuccess) e = hipMemsetAsync(*p, 0, bytes, c->stream); } };
    e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
    A((void**)&c->x0, npad * 8 * max_dim); A((void**)&c->x1, npad * 8 * max_dim);
    A((void**)&c->lw, npad * 8); A((void**)&c->w, npad * 8); A((void**)&c->auxlw, npad * 8); A((void**)&c->auxg, npad * 8);
    A((void**)&c->cum, npad * 8);
    A((void**)&c->pm, MAXB * 8); A((void**)&c->ps, MAXB * 8); A((void**)&c->pq, MAXB * 8); A((void**)&c->bsum, MAXB * 8); A((void**)&c->bsq, MAXB * 8);
    A((void**)&c->ain_w, MAXB * 8); A((void**)&c->ain_p, MAXB * 8);
    A((void**)&c->brec, MAXB * sizeof(BlockRec)); A((void**)&c->brec_p, MAXB * sizeof(BlockRec)); A((void**)&c->cin, MAXB * 8);
    A((void**)&c->side, (size_t)B * sizeof(SideList)); A((void**)&c->side_p, (size_t)B * sizeof(SideList));
    A((void**)&c->st, sizeof(DevState));
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    if (e == hipSuccess) e = hipEventCreate(&c->ev0);
    if (e == hipSuccess) e = hipEventCreate(&c->ev1);
    if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_resolve<MODE_W>), hipFuncAttributeMaxDynamicSharedMemorySize, MAXB * (int)sizeof(BlockRec));
    if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_resolve<MODE_P>), hipFuncAttributeMaxDynamicSharedMemorySize, MAXB * (int)sizeof(BlockRec));
    if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_apply<0, false>), hipFuncAttributeMaxDynamicSharedMemorySize, 3 * CAPX * (int)sizeof(double));
    if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_apply<0, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 3 * CAPX * (int)sizeof(double));
    if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_apply<1, false>), hipFuncAttributeMaxDynamicSharedMemorySize, 3 * CAPX * (int)sizeof(double));
    if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_apply<1, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 3 * CAPX * (int)sizeof(double));
    if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_apply<2, false>), hipFuncAttributeMaxDynamicSharedMemorySize, 3 * CAPX * (int)sizeof(double));
    if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_apply<2, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 3 * CAPX * (int)sizeof(double));
    for (const void* fnp : {reinterpret_cast<const void*>(&k_apply<0, true, true>), reinterpret_cast<const void*>(&k_apply<0, false, true>),
                            reinterpret_cast<const void*>(&k_apply<1, true, true>), reinterpret_cast<const void*>(&k_apply<1, false, true>)})
        if (e == hipSuccess) e = hipFuncSetAttribute(fnp, hipFuncAttributeMaxDynamicSharedMemorySize, 3 * CAPX * (int)sizeof(double));
    static_assert(sizeof(ResolveSmem) <= 3 * CAPX * sizeof(double), "the in-kernel resolve borrows k_apply's staging area");
    for (const void* fnp : {reinterpret_cast<const void*>(&k_apply_multi<0>), reinterpret_cast<const void*>(&k_apply_multi<1>)})
        if (e == hipSuccess) e = hipFuncSetAttribute(fnp, hipFuncAttributeMaxDynamicSharedMemorySize, 3 * CAPX * (int)sizeof(double));
    if (e == hipSuccess) {
        // fused path: workspace (zeroed once: the tags only grow), dynamic LDS, and the residency it rests on -- two workgroups
        // per CU, every block of the largest grid resident at once
        e = hipMalloc((void**)&c->fz, sizeof(FusedWs));
        if (e == hipSuccess) e = hipMemsetAsync(c->fz, 0, sizeof(FusedWs), c->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
        int per_cu = 8, ncu = 0;
        for (const void* fnp : {reinterpret_cast<const void*>(&k_obs<0, 0>), reinterpret_cast<const void*>(&k_obs<0, 1>),
                                reinterpret_cast<const void*>(&k_obs<1, 0>), reinterpret_cast<const void*>(&k_obs<1, 1>)}) {
            if (e == hipSuccess) e = hipFuncSetAttribute(fnp, hipFuncAttributeMaxDynamicSharedMemorySize, (int)FZ_DYN_LDS);
            int nb = 0;
            if (e == hipSuccess) e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, fnp, NT, FZ_DYN_LDS);
            per_cu = std::min(per_cu, nb);
        }
        if (e == hipSuccess) e = hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, device);
        c->fz_ok = (e == hipSuccess) && per_cu >= 1 && (long long)per_cu * ncu >= FZ_MAXB;
    }
    if (e != hipSuccess) {
        g_err = std::string("bssm_ctx_create: ") + hipGetErrorString(e);
        bssm_ctx_destroy(c);
        return BSSM_ERR_HIP;
    }
    *out = c;
    return BSSM_OK;
}

extern "C" void bssm_ctx_destroy(bssm_ctx* c)
{
    if (!c) return;
    (void)hipSetDevice(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    void* ptrs[] = {c->x0, c->x1, c->lw, c->w, c->auxlw, c->auxg, c->cum, c->pm, c->ps, c->pq, c->bsum, c->bsq,
                    c->ain_w, c->ain_p, c->brec, c->brec_p, c->side, c->side_p, c->cin, c->st, c->fz};
    for (void* p : ptrs) if (p) (void)hipFree(p);
    for (auto& kv : c->pool) if (kv.second.first) (void)hipFree(kv.second.first);
    if (c->h_stage) (void)hipHostFree(c->h_stage);
    for (auto ev : c->ev_pool) (void)hipEventDestroy(ev);
    if (c->ev0) (void)hipEventDestroy(c->ev0);
    if (c->ev1) (void)hipEventDestroy(c->ev1);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

extern "C" int bssm_ctx_synchronize(bssm_ctx* c)
{
    if (!c) ARGFAIL("ctx is NULL");
    HIPCHK(hipSetDevice(c->device));
    HIPCHK(hipStreamSynchronize(c->stream));
    return BSSM_OK;
}

extern "C" void* bssm_ctx_stream(bssm_ctx* c) { return c ? (void*)c->stream : nullptr; }

extern "C" int bssm_ctx_set_option(bssm_ctx* c, int option, int value)
{
    if (!c) ARGFAIL("ctx is NULL");
    switch (option) {
        case BSSM_OPT_RECORD_WINDOW: c->opt_window = value; break;
        case BSSM_OPT_BATCH_LITERAL_MAX: c->opt_batch_lit_max = value; break;
        case BSSM_OPT_STAGE_EXPANSION: c->opt_stage = value; break;
        case BSSM_OPT_INKERNEL_RESOLVE: c->opt_inkernel_resolve = value; break;
        case BSSM_OPT_DEBUG_STOP: c->opt_debug_stop = value; break;
        case BSSM_OPT_FUSE_STEP: c->opt_fuse_step = value; break;
        case BSSM_OPT_RECOMPUTE_LW: c->opt_recompute_lw = value ? 1 : 0; break;
        case BSSM_OPT_RENORMALIZE: c->opt_renormalize = value ? 1 : 0; break;
        case BSSM_OPT_FUSED: c->opt_fused = value < 0 ? 0 : (value > 2 ? 2 : value); break;
        case BSSM_OPT_FUSED_PREFETCH: c->opt_fused_prefetch = value ? 1 : 0; break;
        default: ARGFAIL("bssm_ctx_set_option: unknown option");
    }
    return BSSM_OK;
}
__global__ void k_set_debug(DevState* st, int v) { st->debug_stop = v; }
__global__ void k_set_calls(DevState* st, int v) { st->res_calls = v; }       // the next resample call draws the generator's stream `v`
extern "C" int bssm_ctx_get_stamps(bssm_ctx* c, long long* out /* [4][16] */)
{   // DEV builds (make DEV=1): the clock64() stage stamps of the last run; zeros otherwise
    if (!c || !out) ARGFAIL("bssm_ctx_get_stamps: NULL argument");
    DevState h;
    HIPCHK(hipMemcpy(&h, c->st, sizeof(h), hipMemcpyDeviceToHost));
    memcpy(out, h.stamps, sizeof(h.stamps));
    return BSSM_OK;
}

extern "C" int bssm_ctx_set_profile(bssm_ctx* c, int enable)
{
    if (!c) ARGFAIL("ctx is NULL");
    c->profile = enable != 0;
    c->prof.clear();
    return BSSM_OK;
}

extern "C" int bssm_ctx_get_profile(bssm_ctx* c, int max_entries, const char** names, double* total_ms, long long* launches)
{
    if (!c) return 0;
    int k = 0;
    for (auto& kv : c->prof) {
        if (k >= max_entries) break;
        names[k] = kv.first.c_str(); total_ms[k] = kv.second.ms; launches[k] = kv.second.launches; k++;
    }
    return k;
}

// ---- launch helper: optional per-kernel-class timing ------------------------------------------
// Profile mode hands the launch a start and a stop event (hipExtLaunchKernelGGL): they receive the begin / end
// timestamps of the kernel's own dispatch, i.e. the duration rocprofv3 --kernel-trace reports.  (A pair of
// hipEventRecord calls around the launch would also time the dispatch of the launch, ~2.5 us here.)
static hipEvent_t prof_event(bssm_ctx* c)
{
    hipEvent_t e = nullptr;
    if (!c->ev_pool.empty()) { e = c->ev_pool.back(); c->ev_pool.pop_back(); } else (void)hipEventCreate(&e);
    return e;
}
#define LAUNCH(c, name, kern, grid, block, shmem, ...)                                                              \
    do {                                                                                                              \
        if ((c)->profile) {                                                                                           \
            hipEvent_t ea__ = prof_event(c), eb__ = prof_event(c);                                                    \
            hipExtLaunchKernelGGL(kern, dim3(grid), dim3(block), shmem, (c)->stream, ea__, eb__, 0, __VA_ARGS__);     \
            (c)->prof_pending.push_back({name, {ea__, eb__}});                                                        \
        } else hipLaunchKernelGGL(kern, dim3(grid), dim3(block), shmem, (c)->stream, __VA_ARGS__);                    \
    } while (0)

static void prof_collect(bssm_ctx* c)
{
    for (auto& p : c->prof_pending) {
        float ms = 0;
        (void)hipEventSynchronize(p.second.second);
        (void)hipEventElapsedTime(&ms, p.second.first, p.second.second);
        auto& e = c->prof[p.first]; e.ms += ms; e.launches++;
        c->ev_pool.push_back(p.second.first); c->ev_pool.push_back(p.second.second);
    }
    c->prof_pending.clear();
}
static PhiloxKey make_key(unsigned long long seed, unsigned long long stream)
{
    PhiloxKey k; k.k0 = (uint32_t)seed; k.k1 = (uint32_t)(seed >> 32);
    k.stream = (uint32_t)stream ^ (uint32_t)((stream >> 32) * 0x9E3779B9u);
    return k;
}

// ---- the exact resampling pipeline on weights already in HBM ---------------------
// resample_*_cpp: total = sum(w); prob = w/total; cum = cumsum(prob); walk.   src/resampling.cpp:16-66
struct ResampleLaunch {
    // filter path: weights are produced from log-weights inside the first scan kernel
    const double* d_lw = nullptr; int plan = PLAN_RESAMPLE_ONLY; int check_degenerate = 0;
    const double* xw = nullptr; double yw = 0, syw = 1, lsyw = 0;      // log-weights not stored: dnorm(yw, xw[j], syw) re-evaluated by the normalising kernel
    int obs_i = 0; int resample_algorithm = 1; double threshold = 0; double* d_ess = nullptr; double* d_llh = nullptr; int* d_resampled = nullptr;
    const double* d_w; long long nw; int n; int kind;
    const double* d_u; long long u_stride; PhiloxKey key;
    int* d_anc; long long anc_stride; double* d_cum;
    const double* xsrc; double* xdst; int dim; long long xstride;
    const double* auxsrc; double* auxdst; double* se_part;
    // the next observation's transition + weight fused into the expansion (k_apply<.., STEP>): model id or -1
    int step_model = -1; ModelPar step_par; double step_y = 0; NoiseSrc step_ns;
};

static void launch_scan_and_apply(bssm_ctx* c, const ResampleLaunch& r)
{
    const int B = (int)((r.nw + EB - 1) / EB);
    const int lim = c->opt_window > 0 ? c->opt_window : rec_window(r.nw);
    const size_t shm = (size_t)B * sizeof(BlockRec);
    const int boff = c->sh_boff, G = c->sh_nloc ? c->sh_nloc : B;        // launch grid: all blocks, or this rank's (sharded)
    const int Bg = c->sh_nloc ? B : 0;                                   // global block count handed to the kernels when sharded
    FromLw f; f.lw = r.d_lw; f.xw = r.xw; f.yw = r.yw; f.syw = r.syw; f.lsyw = r.lsyw; f.w_out = const_cast<double*>(r.d_w); f.pm = c->pm; f.ps = c->ps; f.pq = c->pq; f.nb = B;
    f.lead = boff; f.pub = c->sh_nloc ? boff + G / 2 : B / 2;
    const bool fold = r.d_lw && !c->opt_renormalize && !c->sh_nloc && r.kind != BSSM_MULTINOMIAL_R;
    f.fold = fold ? 1 : 0;
    f.gmax = (r.d_lw && !c->sh_nloc) ? c->gmax_cur : nullptr;       // (sharded: the partial maxima of other ranks are not in this rank's slots)
    f.ain_out = c->ain_w; f.plan = r.plan; f.N = r.nw; f.obs_i = r.obs_i; f.resample_algorithm = r.resample_algorithm;
    f.threshold = r.threshold; f.ess_out = r.d_ess; f.llh_out = r.d_llh; f.resampled_out = r.d_resampled;
    if (r.kind == BSSM_MULTINOMIAL_R) {
        // parity mode: Rcpp::sample's own algorithm on the injected unif_rand() stream; no exact scan involved
        if (r.d_lw) LAUNCH(c, "k_weights(normalize+local<W>)", (k_local<MODE_W, true>), G, NT, 0, r.d_w, r.nw, c->ain_w, lim, c->brec, c->side, c->st, f, nullptr, nullptr, nullptr, boff, Bg);
        void *dq, *da, *dh, *danc = r.d_anc;
        if (pool_get(c, "mr_q", (size_t)r.nw * 8, &dq) || pool_get(c, "mr_a", (size_t)r.nw * 4, &da) || pool_get(c, "mr_hl", (size_t)r.nw * 4, &dh)) return;
        long long astride = r.anc_stride;
        if (!danc) { if (pool_get(c, "mr_anc", (size_t)r.n * 4, &danc)) return; astride = 0; }
        LAUNCH(c, "k_multinomial_r", k_multinomial_r, 1, NTM, 0, r.d_w, r.n, r.d_u, r.u_stride, (int*)danc, astride, (double*)dq, (int*)da, (int*)dh, c->st);
        if (r.xdst || r.auxdst) {
            const int Bo = (int)(((long long)r.n + EB - 1) / EB);
            LAUNCH(c, "k_gather_anc", k_gather_anc, Bo, NT, 0, (const int*)danc, astride, r.n, r.xsrc, r.xdst, r.dim, r.xstride, r.auxsrc, r.auxdst, r.se_part, c->st);
        }
        return;
    }
    // B <= 2 NT: the consuming kernels resolve the pass before them in every workgroup (resolve_in_block); larger grids
    // keep the single-workgroup k_resolve launches (a thread would have to hold more than two block records)
    const bool inres = c->opt_inkernel_resolve && B <= 2 * NT;
    if (r.d_lw) LAUNCH(c, "k_weights(normalize+local<W>)", (k_local<MODE_W, true>), G, NT, 0, r.d_w, r.nw, c->ain_w, lim, c->brec, c->side, c->st, f, nullptr, nullptr, nullptr, boff, Bg);
    else LAUNCH(c, "k_local<W>", (k_local<MODE_W, false>), G, NT, 0, r.d_w, r.nw, c->ain_w, lim, c->brec, c->side, c->st, f, nullptr, nullptr, nullptr, boff, Bg);
    // grids of more than 2 NT blocks: one workgroup of NTR threads runs the block-wide resolve and emits every block's state
    // (k_resolve_all); k_resolve (the round-1 resolver) remains for inkernel_resolve = 0 on small grids
    const bool rall = c->opt_inkernel_resolve && !inres && B <= 2 * NTR;
    if (fold) {
        // the W records are the records of cumsum(prob) (total == 1): the expansion resolves them directly
        if (rall) LAUNCH(c, "k_resolve_all<P>", k_resolve_all<MODE_P>, 1, NTR, 0, r.d_w, r.nw, B, c->brec, c->side, c->cin, c->ain_w, c->ain_p, c->st);
        else if (!inres) LAUNCH(c, "k_resolve<P>", k_resolve<MODE_P>, 1, NTR, shm, r.d_w, r.nw, B, c->brec, c->side, c->cin, c->ain_w, c->ain_p, c->st);
    } else if (rall) {
        LAUNCH(c, "k_resolve_all<W>", k_resolve_all<MODE_W>, 1, NTR, 0, r.d_w, r.nw, B, c->brec, c->side, c->cin, c->ain_w, c->ain_p, c->st);
        LAUNCH(c, "k_local<P>", (k_local<MODE_P, false>), G, NT, 0, r.d_w, r.nw, c->ain_p, lim, c->brec_p, c->side_p, c->st, f, nullptr, nullptr, nullptr, boff, Bg);
        LAUNCH(c, "k_resolve_all<P>", k_resolve_all<MODE_P>, 1, NTR, 0, r.d_w, r.nw, B, c->brec_p, c->side_p, c->cin, c->ain_w, c->ain_p, c->st);
    } else if (inres) {
        LAUNCH(c, "k_local<P>(+resolve<W>)", (k_local<MODE_P, false, true>), G, NT, 0, r.d_w, r.nw, c->ain_w, lim, c->brec_p, c->side_p, c->st, f, c->brec, c->side, c->ain_p, boff, Bg);
    } else {
        LAUNCH(c, "k_resolve<W>", k_resolve<MODE_W>, 1, NTR, shm, r.d_w, r.nw, B, c->brec, c->side, c->cin, c->ain_w, c->ain_p, c->st);
        LAUNCH(c, "k_local<P>", (k_local<MODE_P, false>), G, NT, 0, r.d_w, r.nw, c->ain_p, lim, c->brec_p, c->side_p, c->st, f, nullptr, nullptr, nullptr, boff, Bg);
        LAUNCH(c, "k_resolve<P>", k_resolve<MODE_P>, 1, NTR, shm, r.d_w, r.nw, B, c->brec_p, c->side_p, c->cin, c->ain_w, c->ain_p, c->st);
    }
    const BlockRec* pb = inres ? (fold ? c->brec : c->brec_p) : nullptr;
    const SideList* psd = inres ? (fold ? c->side : c->side_p) : nullptr;
    ApplyArgs a;
    a.w = r.d_w; a.nw = r.nw; a.ain_p = fold ? c->ain_w : c->ain_p; a.cin = c->cin; a.lim = lim; a.n = r.n;
    a.u_base = r.d_u; a.u_stride = r.u_stride; a.key = r.key;
    a.anc_out = r.d_anc; a.anc_stride = r.anc_stride;
    a.cum_out = (r.kind == BSSM_MULTINOMIAL) ? (r.d_cum ? r.d_cum : c->cum) : r.d_cum;
    a.xsrc = r.xsrc; a.xdst = r.xdst; a.dim = r.dim; a.xstride = r.xstride;
    a.auxsrc = r.auxsrc; a.auxdst = r.auxdst; a.se_part = r.se_part;
    // LDS staging for the coalesced particle store: one array per thing carried to the outputs
    a.lead = boff; a.last = boff + G - 1;
    a.step_model = r.step_model; a.step_par = r.step_par; a.step_y = r.step_y; a.step_ns = r.step_ns; a.step_lw = c->lw;
    a.nstage = (c->opt_stage && r.xdst && !r.d_anc && r.kind != BSSM_MULTINOMIAL) ? (r.dim > 1 ? 2 : 1) + (r.auxdst ? 1 : 0) : 0;
    const bool step = r.step_model >= 0;
    // several rounds of workgroups per CU (more than 2 NT blocks), scalar state, nothing else to carry: the lean expansion
    // (fewer registers, a smaller staging area: four workgroups per CU instead of three)
    const bool lean = !inres && !step && B > 2 * NT && r.dim == 1 && !r.auxdst && !r.d_anc && a.nstage == 1;
    const size_t xshm = lean ? (size_t)CAP_LEAN * sizeof(double)
                             : std::max((size_t)a.nstage * CAPX * sizeof(double), inres ? sizeof(ResolveSmem) : (size_t)0);
#define APPLY(K, NAME) do { \
        if (lean) LAUNCH(c, "k_apply<" NAME ",lean>", (k_apply<K, false, false, true>), G, NT, xshm, a, c->st, pb, psd, boff, Bg); \
        else if (inres && step) LAUNCH(c, "k_apply+step<" NAME ">(+resolve<P>)", (k_apply<K, true, true>), G, NT, xshm, a, c->st, pb, psd, boff, Bg); \
        else if (inres) LAUNCH(c, "k_apply<" NAME ">(+resolve<P>)", (k_apply<K, true, false>), G, NT, xshm, a, c->st, pb, psd, boff, Bg); \
        else if (step) LAUNCH(c, "k_apply+step<" NAME ">", (k_apply<K, false, true>), G, NT, xshm, a, c->st, pb, psd, boff, Bg); \
        else LAUNCH(c, "k_apply<" NAME ">", (k_apply<K, false, false>), G, NT, xshm, a, c->st, pb, psd, boff, Bg); } while (0)
    if (r.kind == BSSM_SYSTEMATIC) APPLY(1, "systematic");
    else if (r.kind == BSSM_STRATIFIED) APPLY(0, "stratified");
#undef APPLY
    else {
        if (inres) LAUNCH(c, "k_apply<cum>(+resolve<P>)", (k_apply<2, true>), G, NT, xshm, a, c->st, pb, psd, boff, Bg); else LAUNCH(c, "k_apply<cum>", (k_apply<2, false>), G, NT, xshm, a, c->st, pb, psd, boff, Bg);
        const int Bo = (int)(((long long)r.n + EB - 1) / EB);
        LAUNCH(c, "k_multinomial", k_multinomial, Bo, NT, 0, a.cum_out, r.nw, r.n, a, c->st);
    }
}

static int flags_to_status(uint32_t f)
{
    f &= ~(FLAG_FUSED_BAIL | FLAG_FUSED_TIMEOUT);      // (not errors: the run is repeated on the multi-launch path)
    if (f & FLAG_NEGATIVE) return BSSM_ERR_NEGATIVE_WEIGHT;     // checked first, as the reference does
    if (f & FLAG_NONFINITE) { g_err = "weights contain NaN/Inf"; return BSSM_ERR_ARG; }
    if (f & FLAG_ZERO_SUM) return BSSM_ERR_ZERO_SUM;
    return BSSM_OK;
}

static int resample_common_device(bssm_ctx* c, int kind, int n, const double* d_w, int nw, const double* d_u,
                                  int* d_idx, double* d_cum)
{
    const int B = (nw + EB - 1) / EB;
    LAUNCH(c, "k_reset_state", k_reset_state, 1, 1, 0, c->st);
    if (c->opt_debug_stop) hipLaunchKernelGGL(k_set_debug, dim3(1), dim3(1), 0, c->stream, c->st, c->opt_debug_stop);
    LAUNCH(c, "k_bsum", k_bsum, B, NT, 0, d_w, (long long)nw, c->bsum, c->st);
    LAUNCH(c, "k_plan", k_plan, 1, NT, 0, c->bsum, B, c->ain_w, c->st);
    ResampleLaunch r;
    r.d_w = d_w; r.nw = nw; r.n = n; r.kind = kind; r.d_u = d_u; r.u_stride = (kind == BSSM_SYSTEMATIC) ? 1 : n;
    r.key = make_key(0, 0); r.d_anc = d_idx; r.anc_stride = 0; r.d_cum = d_cum;
    r.xsrc = nullptr; r.xdst = nullptr; r.dim = 1; r.xstride = 0; r.auxsrc = nullptr; r.auxdst = nullptr; r.se_part = nullptr;
    launch_scan_and_apply(c, r);
    return BSSM_OK;
}

extern "C" int bssm_resample_device(bssm_ctx* c, int kind, int n, const double* d_w, int nw, double U_scalar,
                                    const double* d_U, int* d_idx, double* d_cum)
{
    if (!c) ARGFAIL("ctx is NULL");
    if (n <= 0 || nw <= 0) ARGFAIL("bssm_resample_device: n and nw must be positive");
    if (kind < 0 || kind > 3) ARGFAIL("bssm_resample_device: unknown resampler kind");
    if ((long long)nw > c->cap || (long long)n > c->cap) { g_err = "bssm_resample_device: size exceeds context capacity"; return BSSM_ERR_CAPACITY; }
    HIPCHK(hipSetDevice(c->device));
    const double* du = d_U;
    if (kind == BSSM_SYSTEMATIC && !d_U) {
        void* p; int rc = pool_get(c, "u_scalar", 8, &p); if (rc) return rc;
        HIPCHK(hipMemcpyAsync(p, &U_scalar, 8, hipMemcpyHostToDevice, c->stream));
        HIPCHK(hipStreamSynchronize(c->stream));   // U_scalar lives on the caller's stack
        du = (const double*)p;
    }
    if (!du) ARGFAIL("bssm_resample_device: d_U is required for stratified/multinomial");
    return resample_common_device(c, kind, n, d_w, nw, du, d_idx, d_cum);
}

extern "C" int bssm_resample_device_status(bssm_ctx* c)
{
    if (!c) ARGFAIL("ctx is NULL");
    HIPCHK(hipSetDevice(c->device));
    DevState h;
    HIPCHK(hipMemcpyAsync(&h, c->st, sizeof(h), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    prof_collect(c);
    return flags_to_status(h.flags);
}

extern "C" int bssm_resample_ex(bssm_ctx* c, int kind, int n, const double* w, int nw, const double* U,
                                int* idx_out, double* cum_out, long long* stats)
{
    if (!c) ARGFAIL("ctx is NULL");
    if (!w || !idx_out || !U) ARGFAIL("bssm_resample: NULL pointer argument");
    if (kind < 0 || kind > 3) ARGFAIL("bssm_resample: unknown resampler kind");
    if (kind == BSSM_MULTINOMIAL_R && n != nw) ARGFAIL("probs.size() != n!");      // Rcpp::sample's own check
    if (n <= 0 || nw <= 0) ARGFAIL("bssm_resample: n and nw must be positive");
    if ((long long)nw > c->cap || (long long)n > c->cap) { g_err = "bssm_resample: size exceeds context capacity"; return BSSM_ERR_CAPACITY; }
    HIPCHK(hipSetDevice(c->device));
    void *dw, *du, *di, *dc = nullptr;
    int rc;
    const size_t nu = (kind == BSSM_SYSTEMATIC) ? 1 : (size_t)n;
    if ((rc = pool_get(c, "rs_w", (size_t)nw * 8, &dw))) return rc;
    if ((rc = pool_get(c, "rs_u", nu * 8, &du))) return rc;
    if ((rc = pool_get(c, "rs_idx", (size_t)n * 4, &di))) return rc;
    if (cum_out || kind == BSSM_MULTINOMIAL) { if ((rc = pool_get(c, "rs_cum", (size_t)nw * 8, &dc))) return rc; }
    HIPCHK(hipMemcpyAsync(dw, w, (size_t)nw * 8, hipMemcpyHostToDevice, c->stream));
    HIPCHK(hipMemcpyAsync(du, U, nu * 8, hipMemcpyHostToDevice, c->stream));
    rc = resample_common_device(c, kind, n, (const double*)dw, nw, (const double*)du, (int*)di, (double*)dc);
    if (rc) return rc;
    HIPCHK(hipGetLastError());
    DevState h;
    HIPCHK(hipMemcpyAsync(&h, c->st, sizeof(h), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    prof_collect(c);
    if (stats) { stats[0] = h.stat_hard_blocks; stats[1] = h.stat_serial_walks; stats[2] = h.stat_literal_terms; stats[3] = (nw + EB - 1) / EB; }
    const int st = flags_to_status(h.flags);
    if (st) { if (st != BSSM_ERR_ARG) g_err = bssm_status_string(st); return st; }
    HIPCHK(hipMemcpy(idx_out, di, (size_t)n * 4, hipMemcpyDeviceToHost));
    if (cum_out) HIPCHK(hipMemcpy(cum_out, dc, (size_t)nw * 8, hipMemcpyDeviceToHost));
    return BSSM_OK;
}

extern "C" int bssm_resample_systematic(bssm_ctx* c, int n, const double* w, int nw, double U, int* out)
{
    return bssm_resample_ex(c, BSSM_SYSTEMATIC, n, w, nw, &U, out, nullptr, nullptr);
}
extern "C" int bssm_resample_stratified(bssm_ctx* c, int n, const double* w, int nw, const double* U, int* out)
{
    return bssm_resample_ex(c, BSSM_STRATIFIED, n, w, nw, U, out, nullptr, nullptr);
}
extern "C" int bssm_resample_multinomial(bssm_ctx* c, int n, const double* w, int nw, const double* U, int* out)
{
    return bssm_resample_ex(c, BSSM_MULTINOMIAL, n, w, nw, U, out, nullptr, nullptr);
}

extern "C" int bssm_resample_multinomial_r(bssm_ctx* c, int n, const double* w, int nw, const double* U, int* out)
{
    return bssm_resample_ex(c, BSSM_MULTINOMIAL_R, n, w, nw, U, out, nullptr, nullptr);
}

// ---- generator dumps ---------------------------------------------------------------
extern "C" int bssm_dump_normals(bssm_ctx* c, unsigned long long seed, unsigned long long stream, int purpose, int call,
                                 long long n, double* out)
{
    if (!c || !out || n <= 0) ARGFAIL("bssm_dump_normals: bad argument");
    HIPCHK(hipSetDevice(c->device));
    void* d; int rc = pool_get(c, "dump", (size_t)n * 8, &d); if (rc) return rc;
    const long long pairs = (n + 1) / 2;
    hipLaunchKernelGGL(k_dump_normals, dim3((unsigned)((pairs + 255) / 256)), dim3(256), 0, c->stream,
                       make_key(seed, stream), (uint32_t)purpose, (uint32_t)call, n, (double*)d);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(out, d, (size_t)n * 8, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    return BSSM_OK;
}
extern "C" int bssm_dump_move_draws(bssm_ctx* c, unsigned long long seed, unsigned long long stream, int call, long long n,
                                    double* z_out, double* u_out)
{
    if (!c || !z_out || !u_out || n <= 0) ARGFAIL("bssm_dump_move_draws: bad argument");
    HIPCHK(hipSetDevice(c->device));
    void *dz, *du; int rc;
    if ((rc = pool_get(c, "dump", (size_t)n * 8, &dz))) return rc;
    if ((rc = pool_get(c, "dump2", (size_t)n * 8, &du))) return rc;
    hipLaunchKernelGGL(k_dump_move, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c->stream, make_key(seed, stream), (uint32_t)call, n, (double*)dz, (double*)du);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(z_out, dz, (size_t)n * 8, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipMemcpyAsync(u_out, du, (size_t)n * 8, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    return BSSM_OK;
}
extern "C" int bssm_dump_uniforms(bssm_ctx* c, unsigned long long seed, unsigned long long stream, int call, long long n, double* out)
{
    if (!c || !out || n <= 0) ARGFAIL("bssm_dump_uniforms: bad argument");
    HIPCHK(hipSetDevice(c->device));
    void* d; int rc = pool_get(c, "dump", (size_t)n * 8, &d); if (rc) return rc;
    hipLaunchKernelGGL(k_dump_uniforms, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c->stream,
                       make_key(seed, stream), (uint32_t)call, n, (double*)d);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(out, d, (size_t)n * 8, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    return BSSM_OK;
}

// ---- particle filter ---------------------------------------------------------------
extern "C" int bssm_pf_noise_shape(int algorithm, int T, const int* obs_times, int* max_trans, int* max_res)
{
    if (T < 0 || !max_trans || !max_res) ARGFAIL("bssm_pf_noise_shape: bad argument");
    const int last = (T > 0) ? (obs_times ? obs_times[T - 1] : T) : 0;
    *max_trans = last + ((algorithm == BSSM_APF) ? T : 0);
    *max_res = T * ((algorithm == BSSM_APF) ? 2 : 1);
    return BSSM_OK;
}

template <int MODEL>
static void launch_step(bssm_ctx* c, bool trans, int weight, bool subaux, double* x, long long N, int B,
                        const ModelPar& par, double y, const NoiseSrc& ns, bool store_lw = true)
{
    if (c->sh_nloc) B = c->sh_nloc;                  // sharded: this rank's blocks only (k_step adds the block offset)
    double* lw_out = store_lw ? c->lw : nullptr;     // nullptr: the normalising kernel re-evaluates the log-weights (FromLw::xw)
#define STEP_ARGS x, x, lw_out, c->auxg, N, par, y, ns, c->pm, c->ps, c->pq, c->st, c->gmax_cur, c->sh_boff
    if (trans && weight == 1 && !subaux) LAUNCH(c, "k_step<trans+weight>", (k_step<MODEL, true, 1, false>), B, NTS, 0, STEP_ARGS);
    else if (trans && weight == 1 && subaux) LAUNCH(c, "k_step<trans+weight-aux>", (k_step<MODEL, true, 1, true>), B, NTS, 0, STEP_ARGS);
    else if (trans && weight == 0) LAUNCH(c, "k_step<trans>", (k_step<MODEL, true, 0, false>), B, NTS, 0, STEP_ARGS);
    else if (!trans && weight == 2) {
        LAUNCH(c, "k_step<aux-weight>", (k_step<MODEL, false, 2, false>), B, NTS, 0, x, x, c->auxlw, c->auxg, N, par, y, ns, c->pm, c->ps, c->pq, c->st, c->gmax_cur, c->sh_boff);
    }
#undef STEP_ARGS
}

static void launch_step_sir(bssm_ctx* c, bool trans, int weight, bool subaux, double* x, long long N, int B,
                            const ModelPar& par, double y, const NoiseSrc& ns)
{
#define SIR_ARGS(LW) x, x, LW, c->auxg, N, par, y, ns, c->pm, c->ps, c->pq, c->gmax_cur
    if (trans && weight == 1 && !subaux) LAUNCH(c, "k_step_sir<trans+weight>", (k_step_sir<true, 1, false>), B, NTS, 0, SIR_ARGS(c->lw));
    else if (trans && weight == 1 && subaux) LAUNCH(c, "k_step_sir<trans+weight-aux>", (k_step_sir<true, 1, true>), B, NTS, 0, SIR_ARGS(c->lw));
    else if (trans && weight == 0) LAUNCH(c, "k_step_sir<trans>", (k_step_sir<true, 0, false>), B, NTS, 0, SIR_ARGS(c->lw));
    else if (!trans && weight == 2) LAUNCH(c, "k_step_sir<aux-weight>", (k_step_sir<false, 2, false>), B, NTS, 0, SIR_ARGS(c->auxlw));
    else if (!trans && weight == 1) LAUNCH(c, "k_step_sir<weight>", (k_step_sir<false, 1, false>), B, NTS, 0, SIR_ARGS(c->lw));
#undef SIR_ARGS
}

static void launch_step_model(bssm_ctx* c, int model, bool trans, int weight, bool subaux, double* x, long long N, int B,
                              const ModelPar& par, double y, const NoiseSrc& ns, bool store_lw = true)
{
    if (model == BSSM_MODEL_LG) launch_step<0>(c, trans, weight, subaux, x, N, B, par, y, ns, store_lw);
    else if (model == BSSM_MODEL_AR1SIN) launch_step<1>(c, trans, weight, subaux, x, N, B, par, y, ns, store_lw);
    else launch_step_sir(c, trans, weight, subaux, x, N, B, par, y, ns);
}

static constexpr int BSSM_RETRY_UNFUSED = -1000;     // internal: a fused run stood down, repeat it on the multi-launch path

static int pf_run_impl(bssm_ctx* c, const bssm_pf_config* cfg, bssm_pf_result* res, bool allow_fused)
{
    if (!c || !cfg || !res) ARGFAIL("bssm_pf_run: NULL argument");
    const long long N = cfg->num_particles;
    const int T = cfg->T;
    if (N <= 0) ARGFAIL("num_particles must be a positive count");                    // assert_count(..., positive = TRUE) :33
    if (T < 0) ARGFAIL("bssm_pf_run: T must be >= 0");
    if (N > c->cap) { g_err = "bssm_pf_run: num_particles exceeds context capacity"; return BSSM_ERR_CAPACITY; }
    if (cfg->model != BSSM_MODEL_LG && cfg->model != BSSM_MODEL_AR1SIN && cfg->model != BSSM_MODEL_SIR) ARGFAIL("bssm_pf_run: unknown model");
    const bool sir = cfg->model == BSSM_MODEL_SIR;
    if (sir && (cfg->z_init || cfg->z_trans)) ARGFAIL("bssm_pf_run: the SIR model draws a data-dependent number of variates; injected z_* are not supported");
    if (sir && cfg->n_theta < 5) ARGFAIL("bssm_pf_run: SIR theta must hold (lambda, gamma, n_total, s0, i0)");
    if (sir && c->max_dim < 2) { g_err = "bssm_pf_run: the SIR model needs a context created with max_dim = 2"; return BSSM_ERR_CAPACITY; }
    if (cfg->algorithm != BSSM_BPF && cfg->algorithm != BSSM_APF && cfg->algorithm != BSSM_RMPF) ARGFAIL("bssm_pf_run: unknown algorithm");
    const bool rmpf = cfg->algorithm == BSSM_RMPF;
    if (rmpf && cfg->model == BSSM_MODEL_SIR) ARGFAIL("bssm_pf_run: the built-in move step is defined for the scalar Gaussian-observation models only");
    if (rmpf && !(cfg->move_sd > 0)) ARGFAIL("bssm_pf_run: RMPF needs move_sd > 0");
    if (rmpf && ((cfg->z_move == nullptr) != (cfg->u_move == nullptr))) ARGFAIL("bssm_pf_run: z_move and u_move must be given together");
    if (cfg->resample_algorithm < 0 || cfg->resample_algorithm > 2) ARGFAIL("bssm_pf_run: unknown resample_algorithm");
    if (cfg->resample_fn < 0 || cfg->resample_fn > 3) ARGFAIL("bssm_pf_run: unknown resample_fn");
    if (cfg->resample_fn == BSSM_MULTINOMIAL_R && !cfg->u_res) ARGFAIL("bssm_pf_run: BSSM_MULTINOMIAL_R replays R's unif_rand() stream: u_res is required");
    if (!cfg->theta || cfg->n_theta < 3) ARGFAIL("bssm_pf_run: theta must hold (phi, sigma_x, sigma_y) or (lambda, gamma, n_total, s0, i0)");
    if (T > 0 && !cfg->y) ARGFAIL("bssm_pf_run: y is NULL");
    if (!res->state_est || !res->ess || !res->loglike || (T > 0 && !res->loglike_history)) ARGFAIL("bssm_pf_run: result buffers missing");
    for (int i = 0; i < T; i++) if (!isfinite(cfg->y[i])) ARGFAIL("Assertion on 'y' failed: Contains missing values");  // assert_numeric(y, any.missing = FALSE) :69
    if (cfg->obs_times) {                                                                   // assert_integerish(lower = 1, sorted = TRUE) :73
        int prev = 1;
        for (int i = 0; i < T; i++) { if (cfg->obs_times[i] < prev) ARGFAIL("Assertion on 'obs_times' failed: Must be sorted and >= 1"); prev = cfg->obs_times[i]; }
    }
    HIPCHK(hipSetDevice(c->device));
    const int dim = sir ? 2 : 1;
    const int B = (int)((N + EB - 1) / EB);
    const bool apf = cfg->algorithm == BSSM_APF;
    (void)rmpf;
    const double dN = (double)N;
    const int resample_algorithm = rmpf ? BSSM_SISR : cfg->resample_algorithm;     // RMPF forces SISR (R/resample_move_filter.R:229)
    double threshold = rmpf ? (double)NAN : cfg->threshold;            // NaN: NULL => auto; an explicit value, negative included, is kept
    if (isnan(threshold)) threshold = (resample_algorithm == BSSM_SIS) ? INFINITY : (resample_algorithm == BSSM_SISR) ? dN : dN / 2;   // :44-50
    int max_trans = 0, max_res = 0;
    bssm_pf_noise_shape(cfg->algorithm, T, cfg->obs_times, &max_trans, &max_res);
    const long long u_stride = (cfg->resample_fn == BSSM_SYSTEMATIC) ? 1 : N;

    // per-run device outputs
    void *d_ess, *d_llh, *d_se, *d_separt, *d_resampled, *d_anc = nullptr, *d_ph = nullptr, *d_wh = nullptr;
    void *d_zi = nullptr, *d_zt = nullptr, *d_ur = nullptr;
    int rc;
    if ((rc = pool_get(c, "ess", (size_t)(T + 1) * 8, &d_ess))) return rc;
    if ((rc = pool_get(c, "llh", (size_t)(T + 1) * 8, &d_llh))) return rc;
    if ((rc = pool_get(c, "se", (size_t)(T + 1) * dim * 8, &d_se))) return rc;
    if ((rc = pool_get(c, "separt", (size_t)(T + 1) * B * dim * 8, &d_separt))) return rc;
    if ((rc = pool_get(c, "resampled", (size_t)(T + 1) * 4, &d_resampled))) return rc;
    if (cfg->return_ancestors) { if (!res->ancestors) ARGFAIL("bssm_pf_run: ancestors buffer missing"); if ((rc = pool_get(c, "anc", (size_t)std::max(max_res, 1) * N * 4, &d_anc))) return rc; }
    if (cfg->return_particles) {
        if (!res->particles_history || !res->weights_history) ARGFAIL("bssm_pf_run: history buffers missing");
        if ((rc = pool_get(c, "ph", (size_t)(T + 1) * N * dim * 8, &d_ph))) return rc;
        if ((rc = pool_get(c, "wh", (size_t)(T + 1) * N * 8, &d_wh))) return rc;
    }
    if (cfg->z_init) { if ((rc = pool_get(c, "zi", (size_t)N * 8, &d_zi))) return rc; HIPCHK(hipMemcpyAsync(d_zi, cfg->z_init, (size_t)N * 8, hipMemcpyHostToDevice, c->stream)); }
    if (cfg->z_trans && max_trans > 0) { if ((rc = pool_get(c, "zt", (size_t)max_trans * N * 8, &d_zt))) return rc; HIPCHK(hipMemcpyAsync(d_zt, cfg->z_trans, (size_t)max_trans * N * 8, hipMemcpyHostToDevice, c->stream)); }
    void *d_zmv = nullptr, *d_umv = nullptr;
    if (rmpf && cfg->z_move && T > 0) {
        if ((rc = pool_get(c, "zmv", (size_t)T * N * 8, &d_zmv))) return rc;
        if ((rc = pool_get(c, "umv", (size_t)T * N * 8, &d_umv))) return rc;
        HIPCHK(hipMemcpyAsync(d_zmv, cfg->z_move, (size_t)T * N * 8, hipMemcpyHostToDevice, c->stream));
        HIPCHK(hipMemcpyAsync(d_umv, cfg->u_move, (size_t)T * N * 8, hipMemcpyHostToDevice, c->stream));
    }
    if (cfg->u_res && max_res > 0) { if ((rc = pool_get(c, "ur", (size_t)max_res * u_stride * 8, &d_ur))) return rc; HIPCHK(hipMemcpyAsync(d_ur, cfg->u_res, (size_t)max_res * u_stride * 8, hipMemcpyHostToDevice, c->stream)); }
    void* d_gmax;                                        // one slot per weight evaluation (two per observation in the auxiliary filter)
    const size_t n_gmax = (size_t)2 * T + 2, gm_words = (size_t)GM_SLOTS * GM_STRIDE;
    if ((rc = pool_get(c, "gmax", n_gmax * gm_words * 8, &d_gmax))) return rc;
    HIPCHK(hipMemsetAsync(d_gmax, 0, n_gmax * gm_words * 8, c->stream));      // key 0 lies below every double's key
    size_t wcall = 0;
    auto next_gmax = [&]() { c->gmax_cur = (unsigned long long*)d_gmax + (wcall < n_gmax ? wcall : n_gmax - 1) * gm_words; wcall++; };
    HIPCHK(hipMemsetAsync(d_separt, 0, (size_t)(T + 1) * B * dim * 8, c->stream));
    HIPCHK(hipMemsetAsync(d_ess, 0, (size_t)(T + 1) * 8, c->stream));
    HIPCHK(hipMemsetAsync(d_llh, 0, (size_t)(T + 1) * 8, c->stream));
    HIPCHK(hipMemsetAsync(d_resampled, 0, (size_t)(T + 1) * 4, c->stream));
    if (d_anc) HIPCHK(hipMemsetAsync(d_anc, 0, (size_t)std::max(max_res, 1) * N * 4, c->stream));

    ModelPar par; memset(&par, 0, sizeof(par));
    par.phi = cfg->theta[0]; par.sx = cfg->theta[1]; par.sy = cfg->theta[2]; par.log_sy = log(cfg->theta[2]);
    if (sir) { par.n_total = cfg->theta[2]; par.s0 = cfg->theta[3]; par.i0 = cfg->theta[4]; }
    const PhiloxKey key = make_key(cfg->seed, cfg->stream);
    double* X0 = c->x0; double* X1 = c->x1;
    double* separt = (double*)d_separt;

    HIPCHK(hipEventRecord(c->ev0, c->stream));
    LAUNCH(c, "k_reset_state", k_reset_state, 1, 1, 0, c->st);
    if (c->opt_debug_stop) hipLaunchKernelGGL(k_set_debug, dim3(1), dim3(1), 0, c->stream, c->st, c->opt_debug_stop);
    {   // t = 0  (:76-116)
        NoiseSrc ns; ns.arr = (const double*)d_zi; ns.key = key; ns.purpose = DRAW_INIT; ns.call = 0;
        LAUNCH(c, "k_init", k_init, B, NT, 0, X0, N, ns, separt, cfg->model, par, 0);
        if (cfg->return_particles) {
            // weights = rep(1/N, N): do_resample is 0 after reset, so seed row 0 from a constant fill
            std::vector<double> w0((size_t)N, 1.0 / dN);
            HIPCHK(hipMemcpyAsync(d_wh, w0.data(), (size_t)N * 8, hipMemcpyHostToDevice, c->stream));
            HIPCHK(hipStreamSynchronize(c->stream));
            HIPCHK(hipMemcpyAsync(d_ph, X0, (size_t)N * dim * 8, hipMemcpyDeviceToDevice, c->stream));
        }
    }
    int ktrans = 0, prev_t = 0;
    bool stepped_ahead = false;
    const double* z_ready = nullptr;      // fused path: the normals of the coming fused transition, drawn by the previous launch
    // bootstrap / resample-move filter on the Gaussian-observation models: k_step does not store the log-weights, k_weights
    // re-evaluates dnorm(y, x) on the particles (8 MB less written per observation at N = 2^20, the same 8 MB read)
    const bool relw = c->opt_recompute_lw && !apf && !sir && !c->opt_fuse_step;
    // one launch per observation (fused.hip.h): bootstrap / resample-move filter, scalar Gaussian-observation model, stratified or
    // systematic resampling, at most 512 blocks; the A/B and test switches that select other kernels keep the multi-launch path
    // (grids of at most FZ_MINB blocks -- one workgroup per CU -- stay on the multi-launch path unless the option is 2: the fused launch's three
    //  in-launch hand-offs cost the same whatever N is, the multi-launch kernels shrink with N: 46.9 vs 43.2 us per observation at 256 blocks,
    //  44.1 vs 47.8 at 320: tools/diag_fused_threshold.py)
    const bool fused = allow_fused && c->fz_ok && c->opt_fused && (B > FZ_MINB || c->opt_fused >= 2) && !apf && !sir && B <= FZ_MAXB && !c->sh_nloc && c->opt_renormalize &&
                       !c->opt_fuse_step && c->opt_inkernel_resolve &&
                       (cfg->resample_fn == BSSM_STRATIFIED || cfg->resample_fn == BSSM_SYSTEMATIC);
    if (fused) c->fz_runs++;
    for (int i = 1; i <= T; i++) {                                                        // :123
        const int ot = cfg->obs_times ? cfg->obs_times[i - 1] : i;
        const int gap = ot - prev_t;                                                      // :124
        prev_t = ot;
        const double yi = cfg->y[i - 1];
        if (sir) par.lgy = lgamma(yi + 1.0);
        auto noise = [&](int k) { NoiseSrc ns; ns.arr = d_zt ? (const double*)d_zt + (size_t)k * N : nullptr; ns.key = key; ns.purpose = DRAW_TRANS; ns.call = (uint32_t)k; return ns; };
        ResampleLaunch r;
        r.d_w = c->w; r.nw = N; r.n = (int)N; r.kind = cfg->resample_fn; r.d_u = (const double*)d_ur; r.u_stride = u_stride; r.key = key;
        r.d_anc = (int*)d_anc; r.anc_stride = N; r.d_cum = nullptr; r.dim = dim; r.xstride = N;
        if (fused) {
            for (int step = 1; step < gap; step++) { launch_step_model(c, cfg->model, true, 0, false, X0, N, B, par, yi, noise(ktrans)); ktrans++; }
            FusedArgs g;
            g.xin = X0; g.N = N; g.nblk = B; g.par = par; g.y = yi; g.trans = (gap >= 1) ? 1 : 0; g.ns = noise(gap >= 1 ? ktrans : 0);
            if (gap >= 1) ktrans++;
            g.obs_i = i; g.resample_algorithm = resample_algorithm; g.threshold = threshold;
            g.ess_out = (double*)d_ess; g.llh_out = (double*)d_llh; g.resampled_out = (int*)d_resampled;
            g.w_out = cfg->return_particles ? c->w : nullptr;
            g.lim = c->opt_window > 0 ? c->opt_window : rec_window(N);
            ApplyArgs& a = g.a;
            a.w = c->w; a.nw = N; a.ain_p = c->ain_p; a.cin = c->cin; a.lim = g.lim; a.n = (int)N;
            a.u_base = (const double*)d_ur; a.u_stride = u_stride; a.key = key;
            a.anc_out = (int*)d_anc; a.anc_stride = N; a.cum_out = nullptr;
            a.xsrc = X0; a.xdst = X1; a.dim = 1; a.xstride = N; a.auxsrc = nullptr; a.auxdst = nullptr;
            a.se_part = separt + (size_t)i * B * dim;
            a.nstage = (c->opt_stage && !d_anc) ? 1 : 0; a.lead = 0; a.last = B - 1;
            a.step_model = -1; a.step_par = par; a.step_y = 0; a.step_ns = g.ns; a.step_lw = nullptr;
            g.ws = c->fz; g.tag = ++c->fz_tag;
            // the normals of the NEXT fused transition are drawn inside this launch, in the time its workgroups wait for the resolver
            // (device generator only; injected draws are arrays already); two buffers alternate (this launch reads one, fills the other)
            g.znext = nullptr; g.znext_call = 0;
            if (z_ready) { g.ns.arr = z_ready; z_ready = nullptr; }
            if (!d_zt && i < T && c->opt_fused_prefetch) {
                const int next_gap = (cfg->obs_times ? cfg->obs_times[i] : i + 1) - ot;
                if (next_gap >= 1) {
                    g.znext = (i & 1) ? c->auxlw : c->auxg; g.znext_call = (uint32_t)(ktrans + next_gap - 1);
                    z_ready = g.znext;
                }
            }
            c->fz_launches++;
            const bool sysk = cfg->resample_fn == BSSM_SYSTEMATIC;
            if (cfg->model == BSSM_MODEL_LG) { if (sysk) LAUNCH(c, "k_obs<systematic>", (k_obs<0, 1>), B, NT, FZ_DYN_LDS, g, c->st); else LAUNCH(c, "k_obs<stratified>", (k_obs<0, 0>), B, NT, FZ_DYN_LDS, g, c->st); }
            else { if (sysk) LAUNCH(c, "k_obs<systematic>", (k_obs<1, 1>), B, NT, FZ_DYN_LDS, g, c->st); else LAUNCH(c, "k_obs<stratified>", (k_obs<1, 0>), B, NT, FZ_DYN_LDS, g, c->st); }
            std::swap(X0, X1);
            if (rmpf) {   // move every particle, then take the state estimate (:226-241)
                double* se_row = separt + (size_t)i * B * dim;
                const double* zm = d_zmv ? (const double*)d_zmv + (size_t)(i - 1) * N : nullptr;
                const double* um = d_umv ? (const double*)d_umv + (size_t)(i - 1) * N : nullptr;
                if (cfg->model == BSSM_MODEL_LG) LAUNCH(c, "k_move", k_move<0>, B, NT, 0, X0, N, par, yi, cfg->move_sd, zm, um, key, (uint32_t)i, se_row, c->st);
                else LAUNCH(c, "k_move", k_move<1>, B, NT, 0, X0, N, par, yi, cfg->move_sd, zm, um, key, (uint32_t)i, se_row, c->st);
            }
            if (cfg->return_particles) {
                LAUNCH(c, "k_record_history", k_record_history, (unsigned)((N + 255) / 256), 256, 0, X0, c->w, N, dim,
                       (double*)d_ph + (size_t)i * N * dim, (double*)d_wh + (size_t)i * N, c->st);
            }
            continue;
        }
        // gap transitions; the last one is fused with the weight evaluation unless APF  (:125-136)
        if (stepped_ahead) {
            // this observation's (single) transition and its log-weights were computed inside the previous observation's
            // expansion kernel; what is left of k_step are the per-block log-sum-exp partials
            next_gmax();
            LAUNCH(c, "k_lw_partials", k_lw_partials, B, NTS, 0, c->lw, N, c->pm, c->ps, c->pq, c->gmax_cur);
            ktrans++;
        } else for (int step = 1; step <= gap; step++) {
            const bool fuse_w = (!apf && step == gap);
            if (fuse_w) next_gmax();
            launch_step_model(c, cfg->model, true, fuse_w ? 1 : 0, false, X0, N, B, par, yi, noise(ktrans), !relw);
            ktrans++;
        }
        // can the NEXT observation's transition + weight ride along with this observation's expansion?  (bootstrap filter
        // that resamples at every observation, a single transition to the next observation, nothing that needs the
        // resampled particles themselves in HBM)
        stepped_ahead = false;
        if (c->opt_fuse_step && !apf && !rmpf && !sir && resample_algorithm == BSSM_SISR && i < T && !cfg->return_particles &&
            !cfg->return_ancestors && (cfg->resample_fn == BSSM_STRATIFIED || cfg->resample_fn == BSSM_SYSTEMATIC)) {
            const int next_gap = (cfg->obs_times ? cfg->obs_times[i] : i + 1) - ot;
            stepped_ahead = (next_gap == 1);
        }
        if (apf) {                                                                        // :140-175
            next_gmax();
            launch_step_model(c, cfg->model, false, 2, false, X0, N, B, par, yi, noise(0));
            r.d_lw = c->auxlw; r.plan = PLAN_AUX; r.check_degenerate = 0; r.obs_i = i;
            r.xsrc = X0; r.xdst = X1; r.auxsrc = c->auxlw; r.auxdst = c->auxg; r.se_part = nullptr;
            launch_scan_and_apply(c, r);
            std::swap(X0, X1);
            next_gmax();
            launch_step_model(c, cfg->model, true, 1, true, X0, N, B, par, yi, noise(ktrans));   // :159-175
            ktrans++;
        } else if (gap <= 0) {
            // obs_times repeats a time: no transition, weights on the current particles
            NoiseSrc ns = noise(0);
            next_gmax();
            double* lwo = relw ? nullptr : c->lw;
            if (cfg->model == BSSM_MODEL_LG) LAUNCH(c, "k_step<weight>", (k_step<0, false, 1, false>), B, NTS, 0, X0, X0, lwo, c->auxg, N, par, yi, ns, c->pm, c->ps, c->pq, c->st, c->gmax_cur, c->sh_boff);
            else if (cfg->model == BSSM_MODEL_AR1SIN) LAUNCH(c, "k_step<weight>", (k_step<1, false, 1, false>), B, NTS, 0, X0, X0, lwo, c->auxg, N, par, yi, ns, c->pm, c->ps, c->pq, c->st, c->gmax_cur, c->sh_boff);
            else launch_step_sir(c, false, 1, false, X0, N, B, par, yi, ns);
        }
        double* se_row = separt + (size_t)i * B * dim;
        // normalise (:204-207) + loglik/ESS/decision (:208-218) + resample (:220-224), fused into the scan kernels
        r.d_lw = c->lw; r.plan = PLAN_PF; r.check_degenerate = 1; r.obs_i = i; r.resample_algorithm = resample_algorithm;
        if (relw) { r.xw = X0; r.yw = yi; r.syw = par.sy; r.lsyw = par.log_sy; }
        r.threshold = threshold; r.d_ess = (double*)d_ess; r.d_llh = (double*)d_llh; r.d_resampled = (int*)d_resampled;
        r.xsrc = X0; r.xdst = X1; r.auxsrc = nullptr; r.auxdst = nullptr; r.se_part = se_row;
        if (stepped_ahead) { r.step_model = cfg->model; r.step_par = par; r.step_y = cfg->y[i]; r.step_ns = noise(ktrans); }
        launch_scan_and_apply(c, r);
        if (resample_algorithm != BSSM_SISR)
            LAUNCH(c, "k_carry", k_carry, B, NT, 0, X0, X1, c->w, N, dim, se_row, c->st, 0);
        std::swap(X0, X1);
        if (rmpf) {   // move every particle, then take the state estimate (:226-241)
            const double* zm = d_zmv ? (const double*)d_zmv + (size_t)(i - 1) * N : nullptr;
            const double* um = d_umv ? (const double*)d_umv + (size_t)(i - 1) * N : nullptr;
            if (cfg->model == BSSM_MODEL_LG) LAUNCH(c, "k_move", k_move<0>, B, NT, 0, X0, N, par, yi, cfg->move_sd, zm, um, key, (uint32_t)i, se_row, c->st);
            else LAUNCH(c, "k_move", k_move<1>, B, NT, 0, X0, N, par, yi, cfg->move_sd, zm, um, key, (uint32_t)i, se_row, c->st);
        }
        if (cfg->return_particles) {
            LAUNCH(c, "k_record_history", k_record_history, (unsigned)((N + 255) / 256), 256, 0, X0, c->w, N, dim,
                   (double*)d_ph + (size_t)i * N * dim, (double*)d_wh + (size_t)i * N, c->st);
        }
    }
    LAUNCH(c, "k_reduce_state_est", k_reduce_state_est, T + 1, NT, 0, separt, B, dim, (double*)d_se);
    HIPCHK(hipEventRecord(c->ev1, c->stream));
    HIPCHK(hipGetLastError());
    DevState h;
    HIPCHK(hipMemcpyAsync(&h, c->st, sizeof(h), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipMemcpyAsync(res->state_est, d_se, (size_t)(T + 1) * dim * 8, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipMemcpyAsync(res->ess, d_ess, (size_t)(T + 1) * 8, hipMemcpyDeviceToHost, c->stream));
    if (T > 0) HIPCHK(hipMemcpyAsync(res->loglike_history, d_llh, (size_t)T * 8, hipMemcpyDeviceToHost, c->stream));
    if (res->resampled && T > 0) HIPCHK(hipMemcpyAsync(res->resampled, d_resampled, (size_t)T * 4, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    prof_collect(c);
    if (fused && (h.flags & (FLAG_FUSED_BAIL | FLAG_FUSED_TIMEOUT))) {
        if (h.flags & FLAG_FUSED_BAIL) c->fz_bails++; else c->fz_timeouts++;
        return BSSM_RETRY_UNFUSED;
    }
    if (res->device_ms) { float ms = 0; HIPCHK(hipEventElapsedTime(&ms, c->ev0, c->ev1)); *res->device_ms = ms; }
    // ess[1] = 1 / sum(rep(1/N, N)^2)   (:106-107)
    res->ess[0] = 1.0 / (dN * ((1.0 / dN) * (1.0 / dN)));
    *res->loglike = h.loglike;
    if (res->early_return_step) *res->early_return_step = h.dead;
    if (res->n_res_calls) *res->n_res_calls = h.res_calls;
    if (res->scan_stats) { res->scan_stats[0] = h.stat_hard_blocks; res->scan_stats[1] = h.stat_serial_walks; res->scan_stats[2] = h.stat_literal_terms; }
    if (h.dead) {   // the reference returns at once: later rows keep their initial values -- numeric(out_steps) = 0 for a scalar
                    // state, matrix(NA, out_steps, d) for d > 1 (:90-97); NaN stands for NA_real_ at the C ABI
        const double se_init = (dim > 1) ? (double)NAN : 0.0;
        for (int i = h.dead; i <= T; i++) { res->ess[i] = 0.0; for (int d = 0; d < dim; d++) res->state_est[(size_t)i * dim + d] = se_init; }
        for (int i = h.dead; i < T; i++) res->loglike_history[i] = 0.0;
    }
    if (h.flags) {
        const int st = flags_to_status(h.flags);
        if (st != BSSM_ERR_ARG) g_err = bssm_status_string(st);
        return st;
    }
    if (cfg->return_ancestors && h.res_calls > 0) HIPCHK(hipMemcpy(res->ancestors, d_anc, (size_t)h.res_calls * N * 4, hipMemcpyDeviceToHost));
    if (cfg->return_particles) {
        const int rows = h.dead ? h.dead : T + 1;
        HIPCHK(hipMemcpy(res->particles_history, d_ph, (size_t)rows * N * dim * 8, hipMemcpyDeviceToHost));
        HIPCHK(hipMemcpy(res->weights_history, d_wh, (size_t)rows * N * 8, hipMemcpyDeviceToHost));
    }
    return BSSM_OK;
}


// ---- multivariate linear-Gaussian family (mv.hip.h): bootstrap filter, d <= 8 ---------------------------------------------
// cfg->theta: the packed block  d, p, m0[d], L0[d d], A[d d], b[d], L[d d], c0, H[p d], h0[p], sd[p];  cfg->y: [T][p] row-major
// (unused when p == 0);  injected draws: z_init [d][N], z_trans [calls][d][N] (component-major), u_res as for the scalar models.
static int pf_run_mv(bssm_ctx* c, const bssm_pf_config* cfg, bssm_pf_result* res)
{
    const long long N = cfg->num_particles;
    const int T = cfg->T;
    if (N <= 0) ARGFAIL("num_particles must be a positive count");
    if (T < 0) ARGFAIL("bssm_pf_run: T must be >= 0");
    if (N > c->cap) { g_err = "bssm_pf_run: num_particles exceeds context capacity"; return BSSM_ERR_CAPACITY; }
    if (!cfg->theta || cfg->n_theta < 2) ARGFAIL("bssm_pf_run: the multivariate model needs its packed parameter block");
    MvPar mp; mp.P = nullptr; mp.d = (int)cfg->theta[0]; mp.p = (int)cfg->theta[1];
    const int d = mp.d, p = mp.p;
    if (d < 1 || d > MVD || p < 0 || p > MVD) ARGFAIL("bssm_pf_run: multivariate model: 1 <= d <= 8, 0 <= p <= 8");
    if (cfg->n_theta != mp.o_lsd()) ARGFAIL("bssm_pf_run: multivariate model: parameter block has the wrong length");
    if (d > c->max_dim) { g_err = "bssm_pf_run: the context was created with a smaller max_dim than this model's state dimension"; return BSSM_ERR_CAPACITY; }
    if (cfg->algorithm != BSSM_BPF) ARGFAIL("bssm_pf_run: the multivariate family runs the bootstrap filter");
    if (cfg->resample_algorithm < 0 || cfg->resample_algorithm > 2) ARGFAIL("bssm_pf_run: unknown resample_algorithm");
    if (cfg->resample_fn != BSSM_STRATIFIED && cfg->resample_fn != BSSM_SYSTEMATIC) ARGFAIL("bssm_pf_run: the multivariate family resamples stratified / systematic");
    if (T > 0 && p > 0 && !cfg->y) ARGFAIL("bssm_pf_run: y is NULL");
    if (!res->state_est || !res->ess || !res->loglike || (T > 0 && !res->loglike_history)) ARGFAIL("bssm_pf_run: result buffers missing");
    for (int i = 0; i < T * p; i++) if (!isfinite(cfg->y[i])) ARGFAIL("Assertion on 'y' failed: Contains missing values");
    for (int k = 0; k < p; k++) if (!(cfg->theta[mp.o_sd() + k] > 0)) ARGFAIL("bssm_pf_run: multivariate model: observation sd must be positive");
    if (cfg->obs_times) { int prev = 1; for (int i = 0; i < T; i++) { if (cfg->obs_times[i] < prev) ARGFAIL("Assertion on 'obs_times' failed: Must be sorted and >= 1"); prev = cfg->obs_times[i]; } }
    HIPCHK(hipSetDevice(c->device));
    const int B = (int)((N + EB - 1) / EB);
    const double dN = (double)N;
    const int resample_algorithm = cfg->resample_algorithm;
    double threshold = cfg->threshold;
    if (isnan(threshold)) threshold = (resample_algorithm == BSSM_SIS) ? INFINITY : (resample_algorithm == BSSM_SISR) ? dN : dN / 2;
    int max_trans = 0, max_res = 0;
    bssm_pf_noise_shape(BSSM_BPF, T, cfg->obs_times, &max_trans, &max_res);
    const long long u_stride = (cfg->resample_fn == BSSM_SYSTEMATIC) ? 1 : N;
    void *d_ess, *d_llh, *d_se, *d_separt, *d_resampled, *d_anc, *d_P, *d_y = nullptr, *d_ph = nullptr, *d_wh = nullptr, *d_zi = nullptr, *d_zt = nullptr, *d_ur = nullptr;
    int rc;
    if ((rc = pool_get(c, "ess", (size_t)(T + 1) * 8, &d_ess))) return rc;
    if ((rc = pool_get(c, "llh", (size_t)(T + 1) * 8, &d_llh))) return rc;
    if ((rc = pool_get(c, "se", (size_t)(T + 1) * d * 8, &d_se))) return rc;
    if ((rc = pool_get(c, "separt", (size_t)(T + 1) * B * d * 8, &d_separt))) return rc;
    if ((rc = pool_get(c, "resampled", (size_t)(T + 1) * 4, &d_resampled))) return rc;
    const long long anc_stride = cfg->return_ancestors ? N : 0;
    if (cfg->return_ancestors && !res->ancestors) ARGFAIL("bssm_pf_run: ancestors buffer missing");
    if ((rc = pool_get(c, "anc", (size_t)(cfg->return_ancestors ? std::max(max_res, 1) : 1) * N * 4, &d_anc))) return rc;
    if (cfg->return_particles) {
        if (!res->particles_history || !res->weights_history) ARGFAIL("bssm_pf_run: history buffers missing");
        if ((rc = pool_get(c, "ph", (size_t)(T + 1) * N * d * 8, &d_ph))) return rc;
        if ((rc = pool_get(c, "wh", (size_t)(T + 1) * N * 8, &d_wh))) return rc;
    }
    std::vector<double> hp((size_t)mp.size());
    memcpy(hp.data(), cfg->theta, (size_t)mp.o_lsd() * 8);
    for (int k = 0; k < p; k++) hp[(size_t)mp.o_lsd() + k] = log(cfg->theta[mp.o_sd() + k]);      // (taken on the host, like log(sigma_y) of the scalar models)
    if ((rc = pool_get(c, "mv_par", hp.size() * 8, &d_P))) return rc;
    HIPCHK(hipMemcpyAsync(d_P, hp.data(), hp.size() * 8, hipMemcpyHostToDevice, c->stream));
    mp.P = (const double*)d_P;
    if (p > 0 && T > 0) { if ((rc = pool_get(c, "mv_y", (size_t)T * p * 8, &d_y))) return rc; HIPCHK(hipMemcpyAsync(d_y, cfg->y, (size_t)T * p * 8, hipMemcpyHostToDevice, c->stream)); }
    if (cfg->z_init) { if ((rc = pool_get(c, "zi", (size_t)N * d * 8, &d_zi))) return rc; HIPCHK(hipMemcpyAsync(d_zi, cfg->z_init, (size_t)N * d * 8, hipMemcpyHostToDevice, c->stream)); }
    if (cfg->z_trans && max_trans > 0) { if ((rc = pool_get(c, "zt", (size_t)max_trans * N * d * 8, &d_zt))) return rc; HIPCHK(hipMemcpyAsync(d_zt, cfg->z_trans, (size_t)max_trans * N * d * 8, hipMemcpyHostToDevice, c->stream)); }
    if (cfg->u_res && max_res > 0) { if ((rc = pool_get(c, "ur", (size_t)max_res * u_stride * 8, &d_ur))) return rc; HIPCHK(hipMemcpyAsync(d_ur, cfg->u_res, (size_t)max_res * u_stride * 8, hipMemcpyHostToDevice, c->stream)); }
    HIPCHK(hipStreamSynchronize(c->stream));                 // (hp lives on this stack frame)
    HIPCHK(hipMemsetAsync(d_separt, 0, (size_t)(T + 1) * B * d * 8, c->stream));
    HIPCHK(hipMemsetAsync(d_ess, 0, (size_t)(T + 1) * 8, c->stream));
    HIPCHK(hipMemsetAsync(d_llh, 0, (size_t)(T + 1) * 8, c->stream));
    HIPCHK(hipMemsetAsync(d_resampled, 0, (size_t)(T + 1) * 4, c->stream));
    const PhiloxKey key = make_key(cfg->seed, cfg->stream);
    double* X0 = c->x0; double* X1 = c->x1;
    double* separt = (double*)d_separt;
    c->gmax_cur = nullptr;
    HIPCHK(hipEventRecord(c->ev0, c->stream));
    LAUNCH(c, "k_reset_state", k_reset_state, 1, 1, 0, c->st);
    {
        MvNoise ns; ns.arr = (const double*)d_zi; ns.key = key; ns.purpose = DRAW_INIT; ns.call = 0;
        LAUNCH(c, "k_init_mv", k_init_mv, B, NT, 0, X0, N, mp, ns, separt);
        if (cfg->return_particles) {
            std::vector<double> w0((size_t)N, 1.0 / dN);
            HIPCHK(hipMemcpyAsync(d_wh, w0.data(), (size_t)N * 8, hipMemcpyHostToDevice, c->stream));
            HIPCHK(hipStreamSynchronize(c->stream));
            HIPCHK(hipMemcpyAsync(d_ph, X0, (size_t)N * d * 8, hipMemcpyDeviceToDevice, c->stream));
        }
    }
    int ktrans = 0, prev_t = 0;
    for (int i = 1; i <= T; i++) {                                                        // R/particle_filter_core.R:123
        const int ot = cfg->obs_times ? cfg->obs_times[i - 1] : i;
        const int gap = ot - prev_t;                                                      // :124
        prev_t = ot;
        const double* yrow = p > 0 ? (const double*)d_y + (size_t)(i - 1) * p : nullptr;
        auto noise = [&](int k) { MvNoise ns; ns.arr = d_zt ? (const double*)d_zt + (size_t)k * N * d : nullptr; ns.key = key; ns.purpose = DRAW_TRANS; ns.call = (uint32_t)k; return ns; };
        for (int step = 1; step <= gap; step++) {                                         // :125-136, the last one fused with weight_fn (:177-183)
            if (step == gap) LAUNCH(c, "k_step_mv<trans+weight>", (k_step_mv<true, true>), B, NTS, 0, X0, c->lw, N, mp, yrow, noise(ktrans), c->pm, c->ps, c->pq, (unsigned long long*)nullptr);
            else LAUNCH(c, "k_step_mv<trans>", (k_step_mv<true, false>), B, NTS, 0, X0, c->lw, N, mp, yrow, noise(ktrans), c->pm, c->ps, c->pq, (unsigned long long*)nullptr);
            ktrans++;
        }
        if (gap <= 0) LAUNCH(c, "k_step_mv<weight>", (k_step_mv<false, true>), B, NTS, 0, X0, c->lw, N, mp, yrow, noise(0), c->pm, c->ps, c->pq, (unsigned long long*)nullptr);
        double* se_row = separt + (size_t)i * B * d;
        ResampleLaunch r;
        r.d_lw = c->lw; r.plan = PLAN_PF; r.check_degenerate = 1; r.obs_i = i; r.resample_algorithm = resample_algorithm; r.threshold = threshold;
        r.d_ess = (double*)d_ess; r.d_llh = (double*)d_llh; r.d_resampled = (int*)d_resampled;
        r.d_w = c->w; r.nw = N; r.n = (int)N; r.kind = cfg->resample_fn; r.d_u = (const double*)d_ur; r.u_stride = u_stride; r.key = key;
        r.d_anc = (int*)d_anc; r.anc_stride = anc_stride; r.d_cum = nullptr;
        r.xsrc = nullptr; r.xdst = nullptr; r.dim = 1; r.xstride = 0; r.auxsrc = nullptr; r.auxdst = nullptr; r.se_part = nullptr;
        launch_scan_and_apply(c, r);                                                      // :204-224: ancestors only
        LAUNCH(c, "k_gather_mv", k_gather_mv, B, NT, 0, (const int*)d_anc, anc_stride, N, d, X0, X1, se_row, c->st);        // particles[indices, ]
        if (resample_algorithm != BSSM_SISR) LAUNCH(c, "k_carry_mv", k_carry_mv, B, NT, 0, X0, X1, c->w, N, d, se_row, c->st);
        std::swap(X0, X1);
        if (cfg->return_particles)
            LAUNCH(c, "k_record_history", k_record_history, (unsigned)((N + 255) / 256), 256, 0, X0, c->w, N, d,
                   (double*)d_ph + (size_t)i * N * d, (double*)d_wh + (size_t)i * N, c->st);
    }
    LAUNCH(c, "k_reduce_state_est", k_reduce_state_est, T + 1, NT, 0, separt, B, d, (double*)d_se);
    HIPCHK(hipEventRecord(c->ev1, c->stream));
    HIPCHK(hipGetLastError());
    DevState h;
    HIPCHK(hipMemcpyAsync(&h, c->st, sizeof(h), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipMemcpyAsync(res->state_est, d_se, (size_t)(T + 1) * d * 8, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipMemcpyAsync(res->ess, d_ess, (size_t)(T + 1) * 8, hipMemcpyDeviceToHost, c->stream));
    if (T > 0) HIPCHK(hipMemcpyAsync(res->loglike_history, d_llh, (size_t)T * 8, hipMemcpyDeviceToHost, c->stream));
    if (res->resampled && T > 0) HIPCHK(hipMemcpyAsync(res->resampled, d_resampled, (size_t)T * 4, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    prof_collect(c);
    if (res->device_ms) { float ms = 0; HIPCHK(hipEventElapsedTime(&ms, c->ev0, c->ev1)); *res->device_ms = ms; }
    res->ess[0] = 1.0 / (dN * ((1.0 / dN) * (1.0 / dN)));
    *res->loglike = h.loglike;
    if (res->early_return_step) *res->early_return_step = h.dead;
    if (res->n_res_calls) *res->n_res_calls = h.res_calls;
    if (res->scan_stats) { res->scan_stats[0] = h.stat_hard_blocks; res->scan_stats[1] = h.stat_serial_walks; res->scan_stats[2] = h.stat_literal_terms; }
    if (h.dead) {   // the reference returns at once (:189-202): numeric() zeros for a scalar state, matrix(NA) rows otherwise (:90-97)
        const double se_init = (d > 1) ? (double)NAN : 0.0;
        for (int i = h.dead; i <= T; i++) { res->ess[i] = 0.0; for (int k = 0; k < d; k++) res->state_est[(size_t)i * d + k] = se_init; }
        for (int i = h.dead; i < T; i++) res->loglike_history[i] = 0.0;
    }
    if (h.flags) { const int st = flags_to_status(h.flags); if (st != BSSM_ERR_ARG) g_err = bssm_status_string(st); return st; }
    if (cfg->return_ancestors && h.res_calls > 0) HIPCHK(hipMemcpy(res->ancestors, d_anc, (size_t)h.res_calls * N * 4, hipMemcpyDeviceToHost));
    if (cfg->return_particles) {
        const int rows = h.dead ? h.dead : T + 1;
        HIPCHK(hipMemcpy(res->particles_history, d_ph, (size_t)rows * N * d * 8, hipMemcpyDeviceToHost));
        HIPCHK(hipMemcpy(res->weights_history, d_wh, (size_t)rows * N * 8, hipMemcpyDeviceToHost));
    }
    return BSSM_OK;
}

extern "C" int bssm_dump_normals_mv(bssm_ctx* c, unsigned long long seed, unsigned long long stream, int purpose, int call,
                                    long long N, int d, double* out /* [d][N] */)
{
    if (!c || !out || N <= 0 || d < 1 || d > MVD) ARGFAIL("bssm_dump_normals_mv: bad argument");
    HIPCHK(hipSetDevice(c->device));
    void* dd; int rc = pool_get(c, "dump", (size_t)N * d * 8, &dd); if (rc) return rc;
    const long long pairs = (N + 1) / 2;
    hipLaunchKernelGGL(k_dump_normals_mv, dim3((unsigned)((pairs + 255) / 256)), dim3(256), 0, c->stream, make_key(seed, stream), (uint32_t)purpose, (uint32_t)call, N, d, (double*)dd);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(out, dd, (size_t)N * d * 8, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    return BSSM_OK;
}

extern "C" int bssm_pf_run(bssm_ctx* c, const bssm_pf_config* cfg, bssm_pf_result* res)
{
    if (!c) ARGFAIL("bssm_pf_run: NULL argument");
    if (cfg && res && cfg->model == BSSM_MODEL_LGMV) return pf_run_mv(c, cfg, res);
    // the fused path needs the device's token (one fused run at a time); without it the run takes the multi-launch path
    const int dev = c->device & 63;
    const long long serial = ++g_run_serial[dev];
    const bool alone = (++g_runs_active[dev] == 1);
    if (!alone) g_last_overlap[dev].store(serial);
    const bool quiet = alone && (g_last_overlap[dev].load() == 0 || serial - g_last_overlap[dev].load() > FZ_QUIET) && serial > g_fused_hold[dev].load();
    int expected = 0;
    const bool token = quiet && c->opt_fused && c->fz_ok && g_fused_busy[dev].compare_exchange_strong(expected, 1);
    const long long timeouts0 = c->fz_timeouts;
    int rc = pf_run_impl(c, cfg, res, token);
    if (token) g_fused_busy[dev].store(0);
    // a time-out means the launch's workgroups were not all resident: the GPU is shared with work this process cannot see (another
    // process, another library).  Each costs 20 ms before the run is repeated, so the device backs off: 256 runs without a fused attempt
    // after the first time-out, doubling up to 65 536 while they keep coming
    if (c->fz_timeouts != timeouts0) {
        const long long last = g_fused_span[dev].load();
        const long long span = last > 0 ? (last * 2 > 65536 ? 65536 : last * 2) : 256;
        g_fused_span[dev].store(span);
        g_fused_hold[dev].store(serial + span);
    } else if (token && rc == BSSM_OK) g_fused_span[dev].store(0);
    if (rc == BSSM_RETRY_UNFUSED) rc = pf_run_impl(c, cfg, res, false);      // deterministic: the same draws, the other kernels
    --g_runs_active[dev];
    return rc;
}

extern "C" int bssm_ctx_fused_stamps(bssm_ctx* c, long long* out /* [2][24] */)
{   // DEV builds (make DEV=1): clock64() stage stamps of the last fused launch (typical worker, resolver); zeros otherwise
    if (!c || !out || !c->fz) ARGFAIL("bssm_ctx_fused_stamps: NULL argument");
    HIPCHK(hipMemcpy(out, c->fz->stamps, sizeof(c->fz->stamps), hipMemcpyDeviceToHost));
    return BSSM_OK;
}

extern "C" int bssm_ctx_fused_endt(bssm_ctx* c, long long* out /* [2][512]: end, start */)
{
    if (!c || !out || !c->fz) ARGFAIL("bssm_ctx_fused_endt: NULL argument");
    HIPCHK(hipMemcpy(out, c->fz->endt, sizeof(c->fz->endt), hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(out + FZ_MAXB, c->fz->startt, sizeof(c->fz->startt), hipMemcpyDeviceToHost));
    return BSSM_OK;
}

extern "C" int bssm_ctx_fused_pubt(bssm_ctx* c, long long* out /* [4][512] */)
{
    if (!c || !out || !c->fz) ARGFAIL("bssm_ctx_fused_pubt: NULL argument");
    HIPCHK(hipMemcpy(out, c->fz->pubt, sizeof(c->fz->pubt), hipMemcpyDeviceToHost));
    return BSSM_OK;
}

extern "C" int bssm_ctx_fused_waket(bssm_ctx* c, long long* out /* [3][512] */)
{
    if (!c || !out || !c->fz) ARGFAIL("bssm_ctx_fused_waket: NULL argument");
    HIPCHK(hipMemcpy(out, c->fz->waket, sizeof(c->fz->waket), hipMemcpyDeviceToHost));
    return BSSM_OK;
}

extern "C" int bssm_ctx_fused_stats(bssm_ctx* c, long long* out /* [4]: runs, launches, stand-downs (unsupported record), time-outs */)
{
    if (!c || !out) ARGFAIL("bssm_ctx_fused_stats: NULL argument");
    out[0] = c->fz_runs; out[1] = c->fz_launches; out[2] = c->fz_bails; out[3] = c->fz_timeouts;
    return BSSM_OK;
}


// ---- K independent large filters in lock-step on ONE stream (multi.hip.h) -------------------------------------------------
// ctxs[k] holds filter k's buffers; every launch carries all K argument sets (blockIdx.y = filter).  Shared: data, N, T, model,
// resampling settings; per filter: theta, seed, stream.  Each filter's outputs are bit-identical to bssm_pf_run's.  Configurations
// outside the lock-step kernels' reach (APF / RMPF, SIR, multinomial, more than 2^20 particles, histories, injected draws) run
// one after the other through bssm_pf_run.
extern "C" int bssm_pf_run_multi(bssm_ctx* const* ctxs, int n_filters, const bssm_pf_config* cfg, const double* thetas,
                                 const unsigned long long* seeds, const unsigned long long* streams, bssm_pf_batch_result* res)
{
    if (!ctxs || !cfg || !res || !thetas || !seeds || !streams) ARGFAIL("bssm_pf_run_multi: NULL argument");
    const int F = n_filters;
    if (F < 1 || F > MULTI_MAX) ARGFAIL("bssm_pf_run_multi: 1 .. 4 filters per call");
    for (int k = 0; k < F; k++) { if (!ctxs[k]) ARGFAIL("bssm_pf_run_multi: NULL context"); for (int j = 0; j < k; j++) if (ctxs[j] == ctxs[k]) ARGFAIL("bssm_pf_run_multi: every filter needs a context of its own"); }
    const long long N = cfg->num_particles;
    const int T = cfg->T, nth = cfg->n_theta;
    if (N <= 0) ARGFAIL("num_particles must be a positive count");
    if (T < 0 || nth < 3) ARGFAIL("bssm_pf_run_multi: bad filter configuration");
    if (!res->loglike) ARGFAIL("bssm_pf_run_multi: result buffers missing");
    const int B = (int)((N + EB - 1) / EB);
    bssm_ctx* c0 = ctxs[0];
    bool lock = (cfg->model == BSSM_MODEL_LG || cfg->model == BSSM_MODEL_AR1SIN) && cfg->algorithm == BSSM_BPF && B <= 2 * NT &&
                (cfg->resample_fn == BSSM_STRATIFIED || cfg->resample_fn == BSSM_SYSTEMATIC) && !cfg->return_particles && !cfg->return_ancestors &&
                !cfg->z_init && !cfg->z_trans && !cfg->u_res && F > 1;
    for (int k = 0; k < F && lock; k++) {
        const bssm_ctx* c = ctxs[k];
        lock = c->device == c0->device && N <= c->cap && c->opt_inkernel_resolve && c->opt_renormalize && c->opt_recompute_lw && !c->opt_fuse_step &&
               !c->opt_debug_stop && c->opt_window == c0->opt_window && c->opt_stage == c0->opt_stage && !c->profile;
    }
    const int dim = 1;
    if (!lock) {
        // one after the other
        double ms_total = 0;
        int first_bad = BSSM_OK;
        for (int k = 0; k < F; k++) {
            bssm_pf_config q = *cfg;
            q.theta = thetas + (size_t)k * nth; q.seed = seeds[k]; q.stream = streams[k];
            const int d = (q.model == BSSM_MODEL_SIR) ? 2 : 1;
            std::vector<double> se((size_t)(T + 1) * d), ess((size_t)T + 1), llh((size_t)std::max(T, 1));
            double ll = 0, ms = 0; int early = 0, nres = 0;
            bssm_pf_result r; memset(&r, 0, sizeof(r));
            r.state_est = se.data(); r.ess = ess.data(); r.loglike_history = llh.data(); r.loglike = &ll; r.early_return_step = &early; r.n_res_calls = &nres; r.device_ms = &ms;
            const int rc = bssm_pf_run(ctxs[k], &q, &r);
            if (res->status) res->status[k] = rc;
            if (rc && !first_bad) first_bad = rc;
            res->loglike[k] = ll; ms_total += ms;
            if (res->state_est) memcpy(res->state_est + (size_t)k * (T + 1) * d, se.data(), se.size() * 8);
            if (res->ess) memcpy(res->ess + (size_t)k * (T + 1), ess.data(), ess.size() * 8);
            if (res->loglike_history && T > 0) memcpy(res->loglike_history + (size_t)k * T, llh.data(), (size_t)T * 8);
            if (res->early_return_step) res->early_return_step[k] = early;
            if (res->n_res_calls) res->n_res_calls[k] = nres;
        }
        if (res->device_ms) *res->device_ms = ms_total;
        return res->status ? BSSM_OK : first_bad;
    }
    for (int i = 0; i < T; i++) if (!isfinite(cfg->y[i])) ARGFAIL("Assertion on 'y' failed: Contains missing values");
    if (cfg->obs_times) { int prev = 1; for (int i = 0; i < T; i++) { if (cfg->obs_times[i] < prev) ARGFAIL("Assertion on 'obs_times' failed: Must be sorted and >= 1"); prev = cfg->obs_times[i]; } }
    HIPCHK(hipSetDevice(c0->device));
    hipStream_t stream = c0->stream;
    const double dN = (double)N;
    const int resample_algorithm = cfg->resample_algorithm;
    if (resample_algorithm < 0 || resample_algorithm > 2) ARGFAIL("bssm_pf_run_multi: unknown resample_algorithm");
    double threshold = cfg->threshold;
    if (isnan(threshold)) threshold = (resample_algorithm == BSSM_SIS) ? INFINITY : (resample_algorithm == BSSM_SISR) ? dN : dN / 2;
    const int lim = c0->opt_window > 0 ? c0->opt_window : rec_window(N);
    const size_t n_gmax = (size_t)T + 2, gm_words = (size_t)GM_SLOTS * GM_STRIDE;
    struct Per { void *d_ess, *d_llh, *d_se, *d_separt, *d_resampled, *d_gmax; ModelPar par; PhiloxKey key; double *X0, *X1; };
    std::vector<Per> P((size_t)F);
    for (int k = 0; k < F; k++) {
        bssm_ctx* c = ctxs[k];
        HIPCHK(hipStreamSynchronize(c->stream));                     // (its buffers may still be in use by its own stream)
        int rc;
        if ((rc = pool_get(c, "ess", (size_t)(T + 1) * 8, &P[k].d_ess))) return rc;
        if ((rc = pool_get(c, "llh", (size_t)(T + 1) * 8, &P[k].d_llh))) return rc;
        if ((rc = pool_get(c, "se", (size_t)(T + 1) * dim * 8, &P[k].d_se))) return rc;
        if ((rc = pool_get(c, "separt", (size_t)(T + 1) * B * dim * 8, &P[k].d_separt))) return rc;
        if ((rc = pool_get(c, "resampled", (size_t)(T + 1) * 4, &P[k].d_resampled))) return rc;
        if ((rc = pool_get(c, "gmax", n_gmax * gm_words * 8, &P[k].d_gmax))) return rc;
        HIPCHK(hipMemsetAsync(P[k].d_gmax, 0, n_gmax * gm_words * 8, stream));
        HIPCHK(hipMemsetAsync(P[k].d_separt, 0, (size_t)(T + 1) * B * dim * 8, stream));
        HIPCHK(hipMemsetAsync(P[k].d_ess, 0, (size_t)(T + 1) * 8, stream));
        HIPCHK(hipMemsetAsync(P[k].d_llh, 0, (size_t)(T + 1) * 8, stream));
        HIPCHK(hipMemsetAsync(P[k].d_resampled, 0, (size_t)(T + 1) * 4, stream));
        const double* th = thetas + (size_t)k * nth;
        memset(&P[k].par, 0, sizeof(ModelPar));
        P[k].par.phi = th[0]; P[k].par.sx = th[1]; P[k].par.sy = th[2]; P[k].par.log_sy = log(th[2]);
        P[k].key = make_key(seeds[k], streams[k]);
        P[k].X0 = c->x0; P[k].X1 = c->x1;
    }
    HIPCHK(hipEventRecord(c0->ev0, stream));
    for (int k = 0; k < F; k++) {
        hipLaunchKernelGGL(k_reset_state, dim3(1), dim3(1), 0, stream, ctxs[k]->st);
        NoiseSrc ns; ns.arr = nullptr; ns.key = P[k].key; ns.purpose = DRAW_INIT; ns.call = 0;
        hipLaunchKernelGGL(k_init, dim3(B), dim3(NT), 0, stream, P[k].X0, N, ns, (double*)P[k].d_separt, cfg->model, P[k].par, 0);
    }
    const bool sysk = cfg->resample_fn == BSSM_SYSTEMATIC;
    const long long u_stride = sysk ? 1 : N;
    int ktrans = 0, prev_t = 0;
    size_t wcall = 0;
    for (int i = 1; i <= T; i++) {                                                        // R/particle_filter_core.R:123
        const int ot = cfg->obs_times ? cfg->obs_times[i - 1] : i;
        const int gap = ot - prev_t;
        prev_t = ot;
        const double yi = cfg->y[i - 1];
        const size_t gslot = (wcall < n_gmax ? wcall : n_gmax - 1) * gm_words; wcall++;
        StepMulti sm; LocalMulti lw_, lp_; ApplyMulti am;
        for (int step = 1; step <= std::max(gap, 1); step++) {
            const bool last = (step >= gap), trans = gap >= 1;
            for (int k = 0; k < F; k++) {
                StepArgs& a = sm.a[k];
                a.xin = P[k].X0; a.xout = P[k].X0; a.lw = nullptr; a.N = N; a.par = P[k].par; a.y = yi;
                a.ns.arr = nullptr; a.ns.key = P[k].key; a.ns.purpose = DRAW_TRANS; a.ns.call = (uint32_t)ktrans;
                a.pm = ctxs[k]->pm; a.ps = ctxs[k]->ps; a.pq = ctxs[k]->pq; a.st = ctxs[k]->st; a.gmax = (unsigned long long*)P[k].d_gmax + gslot;
            }
            const dim3 grid(B, F);
            if (cfg->model == BSSM_MODEL_LG) {
                if (trans && last) hipLaunchKernelGGL((k_step_multi<0, true, 1>), grid, dim3(NTS), 0, stream, sm);
                else if (trans) hipLaunchKernelGGL((k_step_multi<0, true, 0>), grid, dim3(NTS), 0, stream, sm);
                else hipLaunchKernelGGL((k_step_multi<0, false, 1>), grid, dim3(NTS), 0, stream, sm);
            } else {
                if (trans && last) hipLaunchKernelGGL((k_step_multi<1, true, 1>), grid, dim3(NTS), 0, stream, sm);
                else if (trans) hipLaunchKernelGGL((k_step_multi<1, true, 0>), grid, dim3(NTS), 0, stream, sm);
                else hipLaunchKernelGGL((k_step_multi<1, false, 1>), grid, dim3(NTS), 0, stream, sm);
            }
            if (trans) ktrans++;
        }
        for (int k = 0; k < F; k++) {
            bssm_ctx* c = ctxs[k];
            FromLw f; f.lw = c->lw; f.xw = P[k].X0; f.yw = yi; f.syw = P[k].par.sy; f.lsyw = P[k].par.log_sy; f.w_out = c->w; f.pm = c->pm; f.ps = c->ps; f.pq = c->pq; f.nb = B;
            f.lead = 0; f.pub = B / 2; f.fold = 0; f.gmax = (unsigned long long*)P[k].d_gmax + gslot; f.ain_out = c->ain_w; f.plan = PLAN_PF; f.N = N; f.obs_i = i;
            f.resample_algorithm = resample_algorithm; f.threshold = threshold; f.ess_out = (double*)P[k].d_ess; f.llh_out = (double*)P[k].d_llh; f.resampled_out = (int*)P[k].d_resampled;
            LocalArgs& w_ = lw_.a[k];
            w_.w = c->w; w_.nw = N; w_.ain = c->ain_w; w_.lim = lim; w_.brec = c->brec; w_.side = c->side; w_.st = c->st; w_.f = f; w_.prev_brec = nullptr; w_.prev_side = nullptr; w_.ain_p_out = nullptr;
            LocalArgs& p_ = lp_.a[k];
            p_ = w_; p_.brec = c->brec_p; p_.side = c->side_p; p_.prev_brec = c->brec; p_.prev_side = c->side; p_.ain_p_out = c->ain_p;
            ApplyOne& q = am.a[k];
            ApplyArgs& a = q.a;
            a.w = c->w; a.nw = N; a.ain_p = c->ain_p; a.cin = c->cin; a.lim = lim; a.n = (int)N; a.u_base = nullptr; a.u_stride = u_stride; a.key = P[k].key;
            a.anc_out = nullptr; a.anc_stride = 0; a.cum_out = nullptr; a.xsrc = P[k].X0; a.xdst = P[k].X1; a.dim = 1; a.xstride = N; a.auxsrc = nullptr; a.auxdst = nullptr;
            a.se_part = (double*)P[k].d_separt + (size_t)i * B * dim; a.nstage = c0->opt_stage ? 1 : 0; a.lead = 0; a.last = B - 1;
            a.step_model = -1; a.step_par = P[k].par; a.step_y = 0; a.step_ns.arr = nullptr; a.step_ns.key = P[k].key; a.step_ns.purpose = 0; a.step_ns.call = 0; a.step_lw = nullptr;
            q.st = c->st; q.prev_brec = c->brec_p; q.prev_side = c->side_p;
        }
        const dim3 g2(B, F);
        hipLaunchKernelGGL(k_weights_multi, g2, dim3(NT), 0, stream, lw_);
        hipLaunchKernelGGL(k_localp_multi, g2, dim3(NT), 0, stream, lp_);
        const size_t xshm = std::max((size_t)(c0->opt_stage ? 1 : 0) * CAPX * sizeof(double), sizeof(ResolveSmem));
        if (sysk) hipLaunchKernelGGL((k_apply_multi<1>), g2, dim3(NT), xshm, stream, am);
        else hipLaunchKernelGGL((k_apply_multi<0>), g2, dim3(NT), xshm, stream, am);
        for (int k = 0; k < F; k++) {
            if (resample_algorithm != BSSM_SISR)
                hipLaunchKernelGGL(k_carry, dim3(B), dim3(NT), 0, stream, P[k].X0, P[k].X1, ctxs[k]->w, N, dim, (double*)P[k].d_separt + (size_t)i * B * dim, ctxs[k]->st, 0);
            std::swap(P[k].X0, P[k].X1);
        }
    }
    for (int k = 0; k < F; k++)
        hipLaunchKernelGGL(k_reduce_state_est, dim3(T + 1), dim3(NT), 0, stream, (double*)P[k].d_separt, B, dim, (double*)P[k].d_se);
    HIPCHK(hipEventRecord(c0->ev1, stream));
    HIPCHK(hipGetLastError());
    std::vector<DevState> hs((size_t)F);
    std::vector<double> se((size_t)(T + 1)), ess((size_t)T + 1), llh((size_t)std::max(T, 1));
    HIPCHK(hipStreamSynchronize(stream));
    int first_bad = BSSM_OK;
    for (int k = 0; k < F; k++) {
        HIPCHK(hipMemcpy(&hs[k], ctxs[k]->st, sizeof(DevState), hipMemcpyDeviceToHost));
        HIPCHK(hipMemcpy(se.data(), P[k].d_se, (size_t)(T + 1) * 8, hipMemcpyDeviceToHost));
        HIPCHK(hipMemcpy(ess.data(), P[k].d_ess, (size_t)(T + 1) * 8, hipMemcpyDeviceToHost));
        if (T > 0) HIPCHK(hipMemcpy(llh.data(), P[k].d_llh, (size_t)T * 8, hipMemcpyDeviceToHost));
        const DevState& h = hs[k];
        ess[0] = 1.0 / (dN * ((1.0 / dN) * (1.0 / dN)));
        if (h.dead) { for (int i = h.dead; i <= T; i++) { ess[i] = 0.0; se[i] = 0.0; } for (int i = h.dead; i < T; i++) llh[i] = 0.0; }
        res->loglike[k] = h.loglike;
        if (res->state_est) memcpy(res->state_est + (size_t)k * (T + 1), se.data(), (size_t)(T + 1) * 8);
        if (res->ess) memcpy(res->ess + (size_t)k * (T + 1), ess.data(), (size_t)(T + 1) * 8);
        if (res->loglike_history && T > 0) memcpy(res->loglike_history + (size_t)k * T, llh.data(), (size_t)T * 8);
        if (res->early_return_step) res->early_return_step[k] = h.dead;
        if (res->n_res_calls) res->n_res_calls[k] = h.res_calls;
        const int stf = h.flags ? flags_to_status(h.flags) : BSSM_OK;
        if (res->status) res->status[k] = stf;
        if (stf && !first_bad) { first_bad = stf; if (stf != BSSM_ERR_ARG) g_err = bssm_status_string(stf); }
    }
    if (res->device_ms) { float ms = 0; HIPCHK(hipEventElapsedTime(&ms, c0->ev0, c0->ev1)); *res->device_ms = ms; }
    return res->status ? BSSM_OK : first_bad;
}

// ---- closure mode: normalise, decide, resample for host-evaluated log-weights -------------------------------------
extern "C" int bssm_pf_weigh_resample(bssm_ctx* c, long long n, const double* lw, int always, int resample_algorithm,
                                      double threshold, int resample_fn, const double* U, unsigned long long seed,
                                      unsigned long long stream, int call, double* weights_out, int* ancestors_out,
                                      double* scalars_out, int* flags_out)
{
    if (!c || !lw || !scalars_out || !flags_out) ARGFAIL("bssm_pf_weigh_resample: NULL argument");
    if (n <= 0) ARGFAIL("num_particles must be a positive count");
    if (n > c->cap) { g_err = "bssm_pf_weigh_resample: num_particles exceeds context capacity"; return BSSM_ERR_CAPACITY; }
    if (resample_algorithm < 0 || resample_algorithm > 2) ARGFAIL("bssm_pf_weigh_resample: unknown resample_algorithm");
    if (resample_fn < 0 || resample_fn > 3) ARGFAIL("bssm_pf_weigh_resample: unknown resample_fn");
    if (resample_fn == BSSM_MULTINOMIAL_R && !U) ARGFAIL("bssm_pf_weigh_resample: BSSM_MULTINOMIAL_R replays R's unif_rand() stream: U is required");
    if (call < 0) ARGFAIL("bssm_pf_weigh_resample: call must be >= 0");
    HIPCHK(hipSetDevice(c->device));
    const int B = (int)((n + EB - 1) / EB);
    const double dN = (double)n;
    if (isnan(threshold)) threshold = (resample_algorithm == BSSM_SIS) ? INFINITY : (resample_algorithm == BSSM_SISR) ? dN : dN / 2;
    const long long nu = (resample_fn == BSSM_SYSTEMATIC) ? 1 : n;
    void *d_small, *d_anc, *d_u = nullptr;
    int rc;
    if ((rc = pool_get(c, "wr_small", 64, &d_small))) return rc;          // ess_out[2], llh_out[1], resampled[1]
    if ((rc = pool_get(c, "wr_anc", (size_t)n * 4, &d_anc))) return rc;
    HIPCHK(hipMemcpyAsync(c->lw, lw, (size_t)n * 8, hipMemcpyHostToDevice, c->stream));
    if (U) { if ((rc = pool_get(c, "wr_u", (size_t)nu * 8, &d_u))) return rc; HIPCHK(hipMemcpyAsync(d_u, U, (size_t)nu * 8, hipMemcpyHostToDevice, c->stream)); }
    HIPCHK(hipMemsetAsync(d_small, 0, 64, c->stream));
    LAUNCH(c, "k_reset_state", k_reset_state, 1, 1, 0, c->st);
    LAUNCH(c, "k_lw_partials", k_lw_partials, B, NTS, 0, c->lw, n, c->pm, c->ps, c->pq, (unsigned long long*)nullptr);
    c->gmax_cur = nullptr;
    ResampleLaunch r;
    r.d_lw = c->lw; r.plan = always ? PLAN_AUX : PLAN_PF; r.check_degenerate = always ? 0 : 1; r.obs_i = 1;
    r.resample_algorithm = resample_algorithm; r.threshold = threshold;
    r.d_ess = (double*)d_small; r.d_llh = (double*)d_small + 2; r.d_resampled = (int*)((double*)d_small + 4);
    r.d_w = c->w; r.nw = n; r.n = (int)n; r.kind = resample_fn;
    // the generator's resample draws are indexed by call: place the injected draws so that call 0 reads them
    r.d_u = (const double*)d_u; r.u_stride = 0; r.key = make_key(seed, stream);
    r.d_anc = (int*)d_anc; r.anc_stride = 0; r.d_cum = nullptr;
    r.xsrc = nullptr; r.xdst = nullptr; r.dim = 1; r.xstride = 0; r.auxsrc = nullptr; r.auxdst = nullptr; r.se_part = nullptr;
    if (!U && call > 0) hipLaunchKernelGGL(k_set_calls, dim3(1), dim3(1), 0, c->stream, c->st, call);
    launch_scan_and_apply(c, r);
    HIPCHK(hipGetLastError());
    DevState h;
    HIPCHK(hipMemcpyAsync(&h, c->st, sizeof(h), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    prof_collect(c);
    flags_out[0] = 0; flags_out[1] = h.dead ? 1 : 0;
    scalars_out[0] = h.loglike; scalars_out[1] = h.ess; scalars_out[2] = h.lse_max; scalars_out[3] = h.lse_sum;
    if (h.dead) return BSSM_OK;
    if (h.flags) { const int st = flags_to_status(h.flags); if (st != BSSM_ERR_ARG) g_err = bssm_status_string(st); return st; }
    if (weights_out) HIPCHK(hipMemcpy(weights_out, c->w, (size_t)n * 8, hipMemcpyDeviceToHost));
    if (h.do_resample) {
        if (!ancestors_out) ARGFAIL("bssm_pf_weigh_resample: ancestors_out is required when the observation resamples");
        HIPCHK(hipMemcpy(ancestors_out, d_anc, (size_t)n * 4, hipMemcpyDeviceToHost));
        flags_out[0] = 1;
    }
    return BSSM_OK;
}

// ---- one filter, particle blocks sharded over ranks (prototype) -------------------------------------------------
// The same kernels on this rank's blocks of the global numbering (block offset), host-staged collectives at the four
// points where a step needs the other ranks' blocks; every rank resolves the exact sums itself (resolve_in_block), so no
// rank waits for a "resolver" and the result does not depend on the number of ranks.
struct ShardScope {      // the context runs sharded only inside bssm_pf_run_sharded
    bssm_ctx* c;
    ShardScope(bssm_ctx* c_, int boff, int nloc) : c(c_) { c->sh_boff = boff; c->sh_nloc = nloc; }
    ~ShardScope() { c->sh_boff = 0; c->sh_nloc = 0; }
};

extern "C" int bssm_pf_run_sharded(bssm_ctx* c, const bssm_pf_config* cfg, const bssm_shard* sh, bssm_pf_result* res)
{
    if (!c || !cfg || !sh || !res) ARGFAIL("bssm_pf_run_sharded: NULL argument");
    if (!sh->all_gather || !sh->exchange || sh->world < 1 || sh->rank < 0 || sh->rank >= sh->world) ARGFAIL("bssm_pf_run_sharded: bad shard description");
    const long long N = cfg->num_particles;
    const int T = cfg->T, W = sh->world;
    if (cfg->model != BSSM_MODEL_LG && cfg->model != BSSM_MODEL_AR1SIN) ARGFAIL("bssm_pf_run_sharded: scalar-state Gaussian models only");
    if (cfg->algorithm != BSSM_BPF) ARGFAIL("bssm_pf_run_sharded: bootstrap filter only");
    if (cfg->resample_fn != BSSM_STRATIFIED && cfg->resample_fn != BSSM_SYSTEMATIC) ARGFAIL("bssm_pf_run_sharded: stratified / systematic resampling only");
    if (cfg->resample_algorithm < 0 || cfg->resample_algorithm > 2) ARGFAIL("bssm_pf_run_sharded: unknown resample_algorithm");
    if (cfg->return_particles || cfg->return_ancestors) ARGFAIL("bssm_pf_run_sharded: histories are not available");
    if (N <= 0 || N % ((long long)W * EB) != 0) ARGFAIL("bssm_pf_run_sharded: num_particles must be a multiple of world x 2048");
    const int B = (int)(N / EB), nloc = B / W, boff = sh->rank * nloc;
    if (B > MAXB) { g_err = "bssm_pf_run_sharded: at most 2^22 particles in all (the block-record workspace of this build; every rank resolves the records of all blocks itself)"; return BSSM_ERR_CAPACITY; }
    // up to 2 NT blocks every workgroup resolves the records itself (as on one GPU); above, one 1024-thread workgroup per rank does (k_resolve_all)
    const bool inres = B <= 2 * NT;
    if (N > c->cap) { g_err = "bssm_pf_run_sharded: num_particles exceeds context capacity"; return BSSM_ERR_CAPACITY; }
    if (T < 0 || (T > 0 && !cfg->y) || !cfg->theta || cfg->n_theta < 3) ARGFAIL("bssm_pf_run_sharded: bad filter configuration");
    if (!res->state_est || !res->ess || !res->loglike || (T > 0 && !res->loglike_history)) ARGFAIL("bssm_pf_run_sharded: result buffers missing");
    HIPCHK(hipSetDevice(c->device));
    ShardScope scope(c, boff, nloc);
    const double dN = (double)N;
    const int resample_algorithm = cfg->resample_algorithm;
    double threshold = cfg->threshold;
    if (isnan(threshold)) threshold = (resample_algorithm == BSSM_SIS) ? INFINITY : (resample_algorithm == BSSM_SISR) ? dN : dN / 2;
    int max_trans = 0, max_res = 0;
    bssm_pf_noise_shape(cfg->algorithm, T, cfg->obs_times, &max_trans, &max_res);
    const long long u_stride = (cfg->resample_fn == BSSM_SYSTEMATIC) ? 1 : N;
    const long long lo = (long long)boff * EB, cnt = (long long)nloc * EB;        // this rank's particles
    void *d_ess, *d_llh, *d_se, *d_separt, *d_resampled, *d_zi = nullptr, *d_zt = nullptr, *d_ur = nullptr;
    int rc;
    if ((rc = pool_get(c, "ess", (size_t)(T + 1) * 8, &d_ess))) return rc;
    if ((rc = pool_get(c, "llh", (size_t)(T + 1) * 8, &d_llh))) return rc;
    if ((rc = pool_get(c, "se", (size_t)(T + 1) * 8, &d_se))) return rc;
    if ((rc = pool_get(c, "separt", (size_t)(T + 1) * B * 8, &d_separt))) return rc;
    if ((rc = pool_get(c, "resampled", (size_t)(T + 1) * 4, &d_resampled))) return rc;
    if (cfg->z_init) { if ((rc = pool_get(c, "zi", (size_t)N * 8, &d_zi))) return rc; HIPCHK(hipMemcpyAsync(d_zi, cfg->z_init, (size_t)N * 8, hipMemcpyHostToDevice, c->stream)); }
    if (cfg->z_trans && max_trans > 0) { if ((rc = pool_get(c, "zt", (size_t)max_trans * N * 8, &d_zt))) return rc; HIPCHK(hipMemcpyAsync(d_zt, cfg->z_trans, (size_t)max_trans * N * 8, hipMemcpyHostToDevice, c->stream)); }
    if (cfg->u_res && max_res > 0) { if ((rc = pool_get(c, "ur", (size_t)max_res * u_stride * 8, &d_ur))) return rc; HIPCHK(hipMemcpyAsync(d_ur, cfg->u_res, (size_t)max_res * u_stride * 8, hipMemcpyHostToDevice, c->stream)); }
    HIPCHK(hipMemsetAsync(d_separt, 0, (size_t)(T + 1) * B * 8, c->stream));
    HIPCHK(hipMemsetAsync(d_ess, 0, (size_t)(T + 1) * 8, c->stream));
    HIPCHK(hipMemsetAsync(d_llh, 0, (size_t)(T + 1) * 8, c->stream));
    HIPCHK(hipMemsetAsync(d_resampled, 0, (size_t)(T + 1) * 4, c->stream));
    // host staging for the collectives
    const size_t rec_bytes = (size_t)nloc * sizeof(BlockRec), side_bytes = (size_t)nloc * sizeof(SideList);
    std::vector<char> hsend(std::max<size_t>({rec_bytes + side_bytes, (size_t)3 * nloc * 8, (size_t)(T + 1) * nloc * 8, (size_t)64})), hrecv(hsend.size() * W);
    std::vector<double> xsend((size_t)std::min<long long>(N, 2 * cnt + (long long)CAPX)), xrecv((size_t)cnt);
    auto sync = [&]() -> int { HIPCHK(hipStreamSynchronize(c->stream)); return BSSM_OK; };
#define SHCHK(expr) do { if ((expr) != 0) { g_err = "bssm_pf_run_sharded: a collective callback failed"; return BSSM_ERR_ARG; } } while (0)
    // all_gather of one per-block device array slice [boff, boff + nloc) x item bytes, written back to the full device array
    const bool devbuf = sh->device_buffers != 0;
    // device-buffer collectives of small HOST data (the ranks' output ranges + status): staged through two device words
    void* d_small = nullptr;
    if (devbuf) { if ((rc = pool_get(c, "sh_small", (size_t)64 * (W + 1), &d_small))) return rc; }
    auto gather_host = [&](const void* send, void* recv, size_t bytes) -> int {
        if (!devbuf) { SHCHK(sh->all_gather(sh->user, send, recv, (long long)bytes)); return BSSM_OK; }
        char* ds = (char*)d_small; char* dr = ds + 64;
        HIPCHK(hipMemcpyAsync(ds, send, bytes, hipMemcpyHostToDevice, c->stream));
        if (int r2 = sync()) return r2;
        SHCHK(sh->all_gather(sh->user, ds, dr, (long long)bytes));
        if (int r2 = sync()) return r2;
        HIPCHK(hipMemcpy(recv, dr, bytes * W, hipMemcpyDeviceToHost));
        return BSSM_OK;
    };
    auto gather_blocks = [&](void* d_arr, size_t item) -> int {
        if (devbuf) {      // in place on the device array: rank r's slice sits at r x nloc x item already
            if (int r2 = sync()) return r2;
            SHCHK(sh->all_gather(sh->user, (char*)d_arr + (size_t)boff * item, d_arr, (long long)((size_t)nloc * item)));
            return sync();
        }
        HIPCHK(hipMemcpyAsync(hsend.data(), (char*)d_arr + (size_t)boff * item, (size_t)nloc * item, hipMemcpyDeviceToHost, c->stream));
        if (int r2 = sync()) return r2;
        SHCHK(sh->all_gather(sh->user, hsend.data(), hrecv.data(), (long long)((size_t)nloc * item)));
        HIPCHK(hipMemcpyAsync(d_arr, hrecv.data(), (size_t)B * item, hipMemcpyHostToDevice, c->stream));      // rank order == block order
        return sync();            // (hrecv is reused by the next gather)
    };
    ModelPar par; memset(&par, 0, sizeof(par));
    par.phi = cfg->theta[0]; par.sx = cfg->theta[1]; par.sy = cfg->theta[2]; par.log_sy = log(cfg->theta[2]);
    const PhiloxKey key = make_key(cfg->seed, cfg->stream);
    double* X0 = c->x0; double* X1 = c->x1;
    double* separt = (double*)d_separt;
    const int lim = c->opt_window > 0 ? c->opt_window : rec_window(N);
    LAUNCH(c, "k_reset_state", k_reset_state, 1, 1, 0, c->st);
    {
        NoiseSrc ns; ns.arr = (const double*)d_zi; ns.key = key; ns.purpose = DRAW_INIT; ns.call = 0;
        LAUNCH(c, "k_init", k_init, nloc, NT, 0, X0, N, ns, separt, cfg->model, par, boff);
    }
    DevState h; memset(&h, 0, sizeof(h));
    int ktrans = 0, prev_t = 0;
    for (int i = 1; i <= T; i++) {                                                        // R/particle_filter_core.R:123
        const int ot = cfg->obs_times ? cfg->obs_times[i - 1] : i;
        const int gap = ot - prev_t;
        prev_t = ot;
        const double yi = cfg->y[i - 1];
        auto noise = [&](int k) { NoiseSrc ns; ns.arr = d_zt ? (const double*)d_zt + (size_t)k * N : nullptr; ns.key = key; ns.purpose = DRAW_TRANS; ns.call = (uint32_t)k; return ns; };
        c->gmax_cur = nullptr;
        for (int step = 1; step <= gap; step++) {
            launch_step_model(c, cfg->model, true, (step == gap) ? 1 : 0, false, X0, N, B, par, yi, noise(ktrans));
            ktrans++;
        }
        if (gap <= 0) {
            NoiseSrc ns = noise(0);
            if (cfg->model == BSSM_MODEL_LG) LAUNCH(c, "k_step<weight>", (k_step<0, false, 1, false>), nloc, NTS, 0, X0, X0, c->lw, c->auxg, N, par, yi, ns, c->pm, c->ps, c->pq, c->st, nullptr, boff);
            else LAUNCH(c, "k_step<weight>", (k_step<1, false, 1, false>), nloc, NTS, 0, X0, X0, c->lw, c->auxg, N, par, yi, ns, c->pm, c->ps, c->pq, c->st, nullptr, boff);
        }
        // (1) the log-sum-exp partials of every block
        if ((rc = gather_blocks(c->pm, 8)) || (rc = gather_blocks(c->ps, 8)) || (rc = gather_blocks(c->pq, 8))) return rc;
        double* se_row = separt + (size_t)i * B;
        FromLw f; f.lw = c->lw; f.xw = nullptr; f.w_out = c->w; f.pm = c->pm; f.ps = c->ps; f.pq = c->pq; f.nb = B; f.gmax = nullptr; f.fold = 0;
        f.lead = boff; f.pub = boff + nloc / 2; f.ain_out = c->ain_w; f.plan = PLAN_PF; f.N = N; f.obs_i = i;
        f.resample_algorithm = resample_algorithm; f.threshold = threshold;
        f.ess_out = (double*)d_ess; f.llh_out = (double*)d_llh; f.resampled_out = (int*)d_resampled;
        LAUNCH(c, "k_weights(normalize+local<W>)", (k_local<MODE_W, true>), nloc, NT, 0, c->w, N, c->ain_w, lim, c->brec, c->side, c->st, f, nullptr, nullptr, nullptr, boff, B);
        HIPCHK(hipMemcpyAsync(&h, c->st, sizeof(h), hipMemcpyDeviceToHost, c->stream));
        if ((rc = sync())) return rc;
        if (h.dead || h.flags) break;                                                     // identical on every rank: all leave together
        if (h.do_resample) {
            // (2) the records of sum(w), (3) the records of cumsum(w / total): every rank resolves them itself
            for (int q = 0; q < 4; q++) if ((rc = gather_blocks((char*)c->brec + (size_t)q * BREC_STRIDE * 16, 16))) return rc;      // (the records live in four planes)
            if ((rc = gather_blocks(c->side, sizeof(SideList)))) return rc;
            if (inres) LAUNCH(c, "k_local<P>(+resolve<W>)", (k_local<MODE_P, false, true>), nloc, NT, 0, c->w, N, c->ain_w, lim, c->brec_p, c->side_p, c->st, f, c->brec, c->side, c->ain_p, boff, B);
            else {
                // (ain_w holds the approximate prefix of EVERY block on every rank: the publishing block of k_weights derives them all from the gathered partials)
                LAUNCH(c, "k_resolve_all<W>", k_resolve_all<MODE_W>, 1, NTR, 0, c->w, N, B, c->brec, c->side, c->cin, c->ain_w, c->ain_p, c->st);
                LAUNCH(c, "k_local<P>", (k_local<MODE_P, false>), nloc, NT, 0, c->w, N, c->ain_p, lim, c->brec_p, c->side_p, c->st, f, nullptr, nullptr, nullptr, boff, B);
            }
            for (int q = 0; q < 4; q++) if ((rc = gather_blocks((char*)c->brec_p + (size_t)q * BREC_STRIDE * 16, 16))) return rc;
            if ((rc = gather_blocks(c->side_p, sizeof(SideList)))) return rc;
            ApplyArgs a;
            a.w = c->w; a.nw = N; a.ain_p = c->ain_p; a.cin = c->cin; a.lim = lim; a.n = (int)N;
            a.u_base = (const double*)d_ur; a.u_stride = u_stride; a.key = key; a.anc_out = nullptr; a.anc_stride = 0; a.cum_out = nullptr;
            a.xsrc = X0; a.xdst = X1; a.dim = 1; a.xstride = N; a.auxsrc = nullptr; a.auxdst = nullptr; a.se_part = se_row;
            a.lead = boff; a.last = boff + nloc - 1; a.step_model = -1; a.step_lw = nullptr;
            a.nstage = c->opt_stage ? 1 : 0;
            const size_t xshm = std::max((size_t)a.nstage * CAPX * sizeof(double), inres ? sizeof(ResolveSmem) : (size_t)0);
            if (inres) {
                if (cfg->resample_fn == BSSM_SYSTEMATIC) LAUNCH(c, "k_apply<systematic>(+resolve<P>)", (k_apply<1, true>), nloc, NT, xshm, a, c->st, c->brec_p, c->side_p, boff, B);
                else LAUNCH(c, "k_apply<stratified>(+resolve<P>)", (k_apply<0, true>), nloc, NT, xshm, a, c->st, c->brec_p, c->side_p, boff, B);
            } else {
                LAUNCH(c, "k_resolve_all<P>", k_resolve_all<MODE_P>, 1, NTR, 0, c->w, N, B, c->brec_p, c->side_p, c->cin, c->ain_w, c->ain_p, c->st);
                if (cfg->resample_fn == BSSM_SYSTEMATIC) LAUNCH(c, "k_apply<systematic>", (k_apply<1, false>), nloc, NT, xshm, a, c->st, nullptr, nullptr, boff, B);
                else LAUNCH(c, "k_apply<stratified>", (k_apply<0, false>), nloc, NT, xshm, a, c->st, nullptr, nullptr, boff, B);
            }
            HIPCHK(hipMemcpyAsync(&h, c->st, sizeof(h), hipMemcpyDeviceToHost, c->stream));
            if ((rc = sync())) return rc;
            // (4) the resampled particles: this rank produced the outputs [out_lo, out_hi); owners are cut at multiples of N / world
            // ... with this rank's status riding along: a rank that failed (its run state carries an error flag, its kernels returned
            // early) is seen by every rank HERE, and all of them leave the loop together instead of one returning while the others
            // wait in the next collective.  The tiling check is made on the gathered ranges, so every rank reaches the same verdict.
            long long mine[3] = {h.out_lo, h.out_hi, (long long)(h.flags | (h.dead ? 0x100u : 0u))};
            std::vector<long long> all((size_t)3 * W), scnt((size_t)W), rcnt((size_t)W);
            if ((rc = gather_host(mine, all.data(), 24))) return rc;
            auto overlap = [](long long a0, long long a1, long long b0, long long b1) { const long long l = std::max(a0, b0), r = std::min(a1, b1); return r > l ? r - l : 0LL; };
            bool any_fail = false, tiles = true;
            long long edge = 0;
            for (int r2 = 0; r2 < W; r2++) {
                any_fail = any_fail || all[3 * r2 + 2] != 0;
                tiles = tiles && all[3 * r2] == edge && all[3 * r2 + 1] >= edge;      // rank r2's outputs start where rank r2 - 1's ended
                edge = all[3 * r2 + 1];
                scnt[r2] = overlap(mine[0], mine[1], (long long)r2 * cnt, (long long)(r2 + 1) * cnt);
                rcnt[r2] = overlap(all[3 * r2], all[3 * r2 + 1], lo, lo + cnt);
            }
            if (any_fail) { for (int r2 = 0; r2 < W; r2++) h.flags |= (uint32_t)(all[3 * r2 + 2] & 0xff); break; }      // every rank: same decision
            if (!tiles || edge != N) { g_err = "bssm_pf_run_sharded: the ranks' output ranges do not tile the particles"; return BSSM_ERR_ARG; }
            const long long nsend = mine[1] - mine[0];
            if (devbuf) {
                // straight from the resampled particles on this GPU into a device staging area (the send and receive ranges of X1 overlap),
                // then into this rank's slice
                double* stage = c->auxlw;
                if ((rc = sync())) return rc;
                SHCHK(sh->exchange(sh->user, X1 + mine[0], scnt.data(), stage, rcnt.data()));
                if ((rc = sync())) return rc;
                HIPCHK(hipMemcpyAsync(X1 + lo, stage, (size_t)cnt * 8, hipMemcpyDeviceToDevice, c->stream));
            } else {
                if ((long long)xsend.size() < nsend) xsend.resize((size_t)nsend);
                if (nsend > 0) HIPCHK(hipMemcpyAsync(xsend.data(), X1 + mine[0], (size_t)nsend * 8, hipMemcpyDeviceToHost, c->stream));
                if ((rc = sync())) return rc;
                SHCHK(sh->exchange(sh->user, xsend.data(), scnt.data(), xrecv.data(), rcnt.data()));
                HIPCHK(hipMemcpyAsync(X1 + lo, xrecv.data(), (size_t)cnt * 8, hipMemcpyHostToDevice, c->stream));
            }
            if ((rc = sync())) return rc;
        } else {
            LAUNCH(c, "k_carry", k_carry, nloc, NT, 0, X0, X1, c->w, N, 1, se_row, c->st, boff);
        }
        std::swap(X0, X1);
    }
#undef SHCHK
    // state estimates: every rank's per-block partials, then the same reduction kernel as the single-GPU run
    {
        HIPCHK(hipMemcpy2DAsync(hsend.data(), (size_t)nloc * 8, separt + boff, (size_t)B * 8, (size_t)nloc * 8, (size_t)(T + 1), hipMemcpyDeviceToHost, c->stream));
        if ((rc = sync())) return rc;
        const size_t seb = (size_t)(T + 1) * nloc * 8;
        if (devbuf) {
            void* dse; if ((rc = pool_get(c, "sh_se", seb * (W + 1), &dse))) return rc;
            HIPCHK(hipMemcpyAsync(dse, hsend.data(), seb, hipMemcpyHostToDevice, c->stream));
            if ((rc = sync())) return rc;
            if (sh->all_gather(sh->user, dse, (char*)dse + seb, (long long)seb) != 0) { g_err = "bssm_pf_run_sharded: a collective callback failed"; return BSSM_ERR_ARG; }
            if ((rc = sync())) return rc;
            HIPCHK(hipMemcpy(hrecv.data(), (char*)dse + seb, seb * W, hipMemcpyDeviceToHost));
        } else if (sh->all_gather(sh->user, hsend.data(), hrecv.data(), (long long)seb) != 0) { g_err = "bssm_pf_run_sharded: a collective callback failed"; return BSSM_ERR_ARG; }
        std::vector<double> full((size_t)(T + 1) * B);
        const double* rv = (const double*)hrecv.data();
        for (int r2 = 0; r2 < W; r2++) for (int i = 0; i <= T; i++)
            memcpy(&full[(size_t)i * B + (size_t)r2 * nloc], rv + ((size_t)r2 * (T + 1) + i) * nloc, (size_t)nloc * 8);
        HIPCHK(hipMemcpyAsync(separt, full.data(), full.size() * 8, hipMemcpyHostToDevice, c->stream));
        LAUNCH(c, "k_reduce_state_est", k_reduce_state_est, T + 1, NT, 0, separt, B, 1, (double*)d_se);
        if ((rc = sync())) return rc;
    }
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(&h, c->st, sizeof(h), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipMemcpyAsync(res->state_est, d_se, (size_t)(T + 1) * 8, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipMemcpyAsync(res->ess, d_ess, (size_t)(T + 1) * 8, hipMemcpyDeviceToHost, c->stream));
    if (T > 0) HIPCHK(hipMemcpyAsync(res->loglike_history, d_llh, (size_t)T * 8, hipMemcpyDeviceToHost, c->stream));
    if (res->resampled && T > 0) HIPCHK(hipMemcpyAsync(res->resampled, d_resampled, (size_t)T * 4, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    prof_collect(c);
    res->ess[0] = 1.0 / (dN * ((1.0 / dN) * (1.0 / dN)));
    *res->loglike = h.loglike;
    if (res->early_return_step) *res->early_return_step = h.dead;
    if (res->n_res_calls) *res->n_res_calls = h.res_calls;
    if (res->device_ms) *res->device_ms = 0.0;
    if (h.dead) {
        for (int i = h.dead; i <= T; i++) { res->ess[i] = 0.0; res->state_est[i] = 0.0; }
        for (int i = h.dead; i < T; i++) res->loglike_history[i] = 0.0;
    }
    if (h.flags) { const int st = flags_to_status(h.flags); if (st != BSSM_ERR_ARG) g_err = bssm_status_string(st); return st; }
    return BSSM_OK;
}

// ---- many small filters per launch (one workgroup = one whole filter) -------------------------
extern "C" int bssm_pf_batch_max_particles(void) { return EB; }

extern "C" int bssm_pf_run_batch(bssm_ctx* c, const bssm_pf_config* cfg, int n_filters, const double* thetas,
                                 const unsigned long long* seeds, const unsigned long long* streams, bssm_pf_batch_result* res)
{
    if (!c || !cfg || !res || !thetas || !seeds || !streams) ARGFAIL("bssm_pf_run_batch: NULL argument");
    const long long N = cfg->num_particles;
    const int T = cfg->T, F = n_filters;
    if (F <= 0) ARGFAIL("bssm_pf_run_batch: n_filters must be positive");
    if (N <= 0) ARGFAIL("num_particles must be a positive count");
    if (N > EB) { g_err = "bssm_pf_run_batch: a batched filter holds at most 2048 particles (one workgroup); use bssm_pf_run"; return BSSM_ERR_CAPACITY; }
    if (T < 0) ARGFAIL("bssm_pf_run_batch: T must be >= 0");
    if (cfg->model != BSSM_MODEL_LG && cfg->model != BSSM_MODEL_AR1SIN && cfg->model != BSSM_MODEL_SIR) ARGFAIL("bssm_pf_run_batch: unknown model");
    const bool sir = cfg->model == BSSM_MODEL_SIR;
    const int dim = sir ? 2 : 1;
    if (cfg->algorithm != BSSM_BPF && cfg->algorithm != BSSM_APF && cfg->algorithm != BSSM_RMPF) ARGFAIL("bssm_pf_run_batch: unknown algorithm");
    const bool apf = cfg->algorithm == BSSM_APF, rmpf = cfg->algorithm == BSSM_RMPF;
    if (rmpf && cfg->model == BSSM_MODEL_SIR) ARGFAIL("bssm_pf_run_batch: the built-in move step is defined for the scalar Gaussian-observation models only");
    if (rmpf && !(cfg->move_sd > 0)) ARGFAIL("bssm_pf_run_batch: RMPF needs move_sd > 0");
    if (rmpf && (cfg->z_move || cfg->u_move)) ARGFAIL("bssm_pf_run_batch: injected draws are not available in the batched path");
    if (cfg->resample_algorithm < 0 || cfg->resample_algorithm > 2) ARGFAIL("bssm_pf_run_batch: unknown resample_algorithm");
    if (cfg->resample_fn < 0 || cfg->resample_fn > 2) ARGFAIL("bssm_pf_run_batch: unknown resample_fn");
    if (cfg->z_init || cfg->z_trans || cfg->u_res || cfg->return_particles || cfg->return_ancestors)
        ARGFAIL("bssm_pf_run_batch: injected draws and histories are not available in the batched path");
    if (cfg->n_theta < (sir ? 5 : 3)) ARGFAIL("bssm_pf_run_batch: theta rows must hold (phi, sigma_x, sigma_y) or (lambda, gamma, n_total, s0, i0)");
    if (T > 0 && !cfg->y) ARGFAIL("bssm_pf_run_batch: y is NULL");
    if (!res->loglike) ARGFAIL("bssm_pf_run_batch: loglike buffer missing");
    for (int i = 0; i < T; i++) if (!isfinite(cfg->y[i])) ARGFAIL("Assertion on 'y' failed: Contains missing values");
    if (cfg->obs_times) {
        int prev = 1;
        for (int i = 0; i < T; i++) { if (cfg->obs_times[i] < prev) ARGFAIL("Assertion on 'obs_times' failed: Must be sorted and >= 1"); prev = cfg->obs_times[i]; }
    }
    HIPCHK(hipSetDevice(c->device));
    const double dN = (double)N;
    const int resample_algorithm = rmpf ? BSSM_SISR : cfg->resample_algorithm;      // RMPF forces SISR (R/resample_move_filter.R:229)
    double threshold = rmpf ? (double)NAN : cfg->threshold;            // NaN: NULL => auto; an explicit value, negative included, is kept
    if (isnan(threshold)) threshold = (resample_algorithm == BSSM_SIS) ? INFINITY : (resample_algorithm == BSSM_SISR) ? dN : dN / 2;   // :44-50
    const int nth = cfg->n_theta;
    int rc;
    // One packed upload and one packed download per call (pinned staging): at T = 20 the filter itself takes ~0.3 ms,
    // a dozen separate small copies and memsets would double that.
    const size_t Tn = (size_t)std::max(T, 1);
    auto up8 = [](size_t x) { return (x + 15) & ~(size_t)15; };
    const size_t o_th = 0, o_lsy = o_th + up8((size_t)F * nth * 8), o_keys = o_lsy + up8((size_t)F * 8),
                 o_y = o_keys + up8((size_t)F * sizeof(PhiloxKey)), o_lgy = o_y + up8(Tn * 8), o_ot = o_lgy + up8(Tn * 8),
                 in_bytes = o_ot + up8(Tn * 4);
    const size_t rowsT1 = (size_t)F * (T + 1) * 8, rowsSe = rowsT1 * dim;
    const size_t q_ll = 0, q_se = q_ll + up8((size_t)F * 8), q_ess = q_se + up8(rowsSe), q_llh = q_ess + up8(rowsT1),
                 q_dead = q_llh + up8((size_t)F * Tn * 8), q_flags = q_dead + up8((size_t)F * 4), q_res = q_flags + up8((size_t)F * 4),
                 out_bytes = q_res + up8((size_t)F * 4);
    if ((rc = host_stage(c, std::max(in_bytes, out_bytes)))) return rc;
    void *d_in, *d_out;
    if ((rc = pool_get(c, "b_in", in_bytes, &d_in))) return rc;
    if ((rc = pool_get(c, "b_out", out_bytes, &d_out))) return rc;
    char* hs = (char*)c->h_stage;
    memcpy(hs + o_th, thetas, (size_t)F * nth * 8);
    for (int f = 0; f < F; f++) {
        ((double*)(hs + o_lsy))[f] = log(thetas[(size_t)f * nth + 2]);                      // as bssm_pf_run: log(sigma_y) on the host
        ((PhiloxKey*)(hs + o_keys))[f] = make_key(seeds[f], streams[f]);
    }
    if (T > 0) memcpy(hs + o_y, cfg->y, (size_t)T * 8);
    for (int i = 0; i < T; i++) ((double*)(hs + o_lgy))[i] = sir ? lgamma(cfg->y[i] + 1.0) : 0.0;   // as bssm_pf_run: lgamma(y + 1) on the host
    if (cfg->obs_times && T > 0) memcpy(hs + o_ot, cfg->obs_times, (size_t)T * 4);
    HIPCHK(hipMemcpyAsync(d_in, hs, in_bytes, hipMemcpyHostToDevice, c->stream));
    HIPCHK(hipMemsetAsync(d_out, 0, out_bytes, c->stream));
    char* di = (char*)d_in; char* dq = (char*)d_out;
    void *d_y = di + o_y, *d_ot = cfg->obs_times ? (void*)(di + o_ot) : nullptr, *d_lgy = sir ? (void*)(di + o_lgy) : nullptr;
    void *d_th = di + o_th, *d_lsy = di + o_lsy, *d_keys = di + o_keys;
    void *d_ll = dq + q_ll, *d_se = dq + q_se, *d_ess = dq + q_ess, *d_llh = dq + q_llh, *d_dead = dq + q_dead, *d_flags = dq + q_flags, *d_res = dq + q_res;
    BatchArgs g;
    g.N = (int)N; g.T = T; g.resample_algorithm = resample_algorithm; g.resample_fn = cfg->resample_fn;
    g.lim = c->opt_window > 0 ? c->opt_window : rec_window(N);
    g.lit_max = c->opt_batch_lit_max; g.move_sd = cfg->move_sd; g.fold = c->opt_renormalize ? 0 : 1;
    g.threshold = threshold; g.y = (const double*)d_y; g.obs_times = (const int*)d_ot; g.lgy = (const double*)d_lgy;
    g.theta = (const double*)d_th; g.theta_stride = nth; g.log_sy = (const double*)d_lsy; g.keys = (const PhiloxKey*)d_keys;
    g.loglike = (double*)d_ll; g.state_est = (double*)d_se; g.ess = (double*)d_ess; g.llh = (double*)d_llh;
    g.dead = (int*)d_dead; g.flags = (uint32_t*)d_flags; g.res_calls = (int*)d_res;
    g.phase_cycles = nullptr;
    void* d_ph = nullptr;
    if (c->opt_debug_stop == 97) { if ((rc = pool_get(c, "b_ph", 40 * 8, &d_ph))) return rc; HIPCHK(hipMemsetAsync(d_ph, 0, 40 * 8, c->stream)); g.phase_cycles = (long long*)d_ph; }
    HIPCHK(hipEventRecord(c->ev0, c->stream));
    const int alg = apf ? 1 : rmpf ? 2 : 0;
    const size_t bshm = (cfg->resample_fn == BSSM_MULTINOMIAL) ? (size_t)EB * sizeof(double) : 0;     // the exact cum_sum for the inverse-CDF search
#define BATCH(M, A) do { if (bshm) HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_pf_batch<M, A>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bshm)); \
                         LAUNCH(c, "k_pf_batch", (k_pf_batch<M, A>), F, NT, bshm, g); } while (0)
    if (cfg->model == BSSM_MODEL_LG) { if (alg == 0) BATCH(0, 0); else if (alg == 1) BATCH(0, 1); else BATCH(0, 2); }
    else if (cfg->model == BSSM_MODEL_AR1SIN) { if (alg == 0) BATCH(1, 0); else if (alg == 1) BATCH(1, 1); else BATCH(1, 2); }
    else { if (alg == 0) BATCH(2, 0); else BATCH(2, 1); }
#undef BATCH
    HIPCHK(hipEventRecord(c->ev1, c->stream));
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(hs, d_out, out_bytes, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    prof_collect(c);
    if (d_ph) { long long h[40]; HIPCHK(hipMemcpy(h, d_ph, 40 * 8, hipMemcpyDeviceToHost));
        const long long* w2 = h + 8; const long long* a3 = h + 24;
        fprintf(stderr, "   last observation, weights block: start->prologue %lld, ->weights %lld, ->terms ready %lld;  apply block: start->loaded %lld, ->in-order/resolve %lld, ->counts %lld, ->expanded %lld, ->state est %lld\n",
                w2[8] - w2[4], w2[9] - w2[8], w2[1] - w2[9], a3[1] - a3[0], a3[3] - a3[1], a3[4] - a3[3], a3[5] - a3[4], a3[6] - a3[5]); fprintf(stderr, "k_pf_batch filter 0 cycles/observation: step %lld, weights+scan<W> %lld, apply %lld, carry/store %lld\n", h[0] / std::max(T, 1), h[1] / std::max(T, 1), h[2] / std::max(T, 1), h[3] / std::max(T, 1)); }
    if (res->device_ms) { float ms = 0; HIPCHK(hipEventElapsedTime(&ms, c->ev0, c->ev1)); *res->device_ms = ms; }
    memcpy(res->loglike, hs + q_ll, (size_t)F * 8);
    if (res->state_est) memcpy(res->state_est, hs + q_se, rowsSe);
    if (res->ess) memcpy(res->ess, hs + q_ess, rowsT1);
    if (res->loglike_history && T > 0) memcpy(res->loglike_history, hs + q_llh, (size_t)F * T * 8);
    const int* dead = (const int*)(hs + q_dead); const int* nres = (const int*)(hs + q_res);
    const uint32_t* flags = (const uint32_t*)(hs + q_flags);
    int first_bad = BSSM_OK;
    for (int f = 0; f < F; f++) {
        if (res->ess) res->ess[(size_t)f * (T + 1)] = 1.0 / (dN * ((1.0 / dN) * (1.0 / dN)));       // :106-107
        if (res->early_return_step) res->early_return_step[f] = dead[f];
        if (res->n_res_calls) res->n_res_calls[f] = nres[f];
        if (dead[f]) {                                // the reference returns at once: later rows keep their initial values (:90-97)
            for (int i = dead[f]; i <= T; i++) {
                if (res->ess) res->ess[(size_t)f * (T + 1) + i] = 0.0;
                if (res->state_est) for (int d = 0; d < dim; d++) res->state_est[((size_t)f * (T + 1) + i) * dim + d] = (dim > 1) ? (double)NAN : 0.0;   // matrix(NA) for d > 1 (:90-95)
            }
            if (res->loglike_history) for (int i = dead[f]; i < T; i++) res->loglike_history[(size_t)f * T + i] = 0.0;
        }
        const int stf = flags[f] ? flags_to_status(flags[f]) : BSSM_OK;
        if (res->status) res->status[f] = stf;
        if (stf && !first_bad) first_bad = stf;
    }
    if (first_bad && !res->status) { g_err = bssm_status_string(first_bad); return first_bad; }
    return BSSM_OK;
}

// ---- PMMH: one chain --------------------------------------------------------------------
// Chain-level draws (proposal normals, acceptance uniform) come from the same
// counter-based generator, keyed by the chain seed: results do not depend on
// which GPU runs the chain.
static double host_normal(const PhiloxKey& key, uint32_t iter, uint32_t j)
{
    u32x4 cn; cn.x = j; cn.y = iter; cn.z = DRAW_PROPOSAL; cn.w = key.stream;
    const u32x4 r = philox4x32_10(cn, key.k0, key.k1);
    return qnorm_as241(u01_from_bits(r.x, r.y));
}
static double host_uniform(const PhiloxKey& key, uint32_t iter)
{
    u32x4 cn; cn.x = 0; cn.y = iter; cn.z = DRAW_ACCEPT; cn.w = key.stream;
    const u32x4 r = philox4x32_10(cn, key.k0, key.k1);
    return u01_from_bits(r.x, r.y);
}
static double log_prior(int kind, double a, double b, double x)
{
    switch (kind) {
        case BSSM_PRIOR_NORMAL: { const double z = (x - a) / b; return -(BSSM_LN_SQRT_2PI + 0.5 * z * z + log(b)); }   // dnorm(log=TRUE)
        case BSSM_PRIOR_EXP: return (x < 0) ? -INFINITY : (log(a) - a * x);                                         // dexp(rate, log=TRUE)
        case BSSM_PRIOR_UNIFORM: return (x >= a && x <= b) ? -log(b - a) : -INFINITY;                               // dunif(log=TRUE)
        case BSSM_PRIOR_HALFNORMAL: { if (x < 0) return -INFINITY; const double z = x / a;                          // extraDistr::dhnorm(x, sigma = a, log = TRUE)
                                      return log(2.0) - (BSSM_LN_SQRT_2PI + 0.5 * z * z + log(a)); }
        default: return 0.0;
    }
}
static double tr_fwd(int tr, double x) { return tr == BSSM_TR_LOG ? log(x) : tr == BSSM_TR_LOGIT ? log(x / (1 - x)) : x; }          // R/utils.R:102-112
static double tr_back(int tr, double z) { return tr == BSSM_TR_LOG ? exp(z) : tr == BSSM_TR_LOGIT ? 1 / (1 + exp(-z)) : z; }        // R/utils.R:122-132
static double tr_logjac(int tr, double x) { return tr == BSSM_TR_LOG ? log(x) : tr == BSSM_TR_LOGIT ? log(1 / (x * (1 - x))) : 0.0; }  // R/utils.R:142-152

// fac = V diag(sqrt(max(ev, 0))) of the symmetric matrix S (row-major p x p)
static int eigen_factor(int p, const double* S, double* fac)
{
    std::vector<double> A((size_t)p * p), V((size_t)p * p);
    for (int a = 0; a < p; a++) for (int b = 0; b < p; b++) { A[a * p + b] = 0.5 * (S[a * p + b] + S[b * p + a]); V[a * p + b] = (a == b); }
    for (int sweep = 0; sweep < 100; sweep++) {
        double off = 0.0;
        for (int a = 0; a < p; a++) for (int b = a + 1; b < p; b++) off += A[a * p + b] * A[a * p + b];
        if (off == 0.0) break;
        for (int a = 0; a < p - 1; a++) for (int b = a + 1; b < p; b++) {
            const double apq = A[a * p + b];
            if (apq == 0.0) continue;
            const double th = (A[b * p + b] - A[a * p + a]) / (2.0 * apq);
            const double t = (th >= 0 ? 1.0 : -1.0) / (fabs(th) + sqrt(th * th + 1.0));
            const double c = 1.0 / sqrt(t * t + 1.0), sn = t * c;
            for (int k = 0; k < p; k++) {
                const double aka = A[k * p + a], akb = A[k * p + b];
                A[k * p + a] = c * aka - sn * akb; A[k * p + b] = sn * aka + c * akb;
                const double vka = V[k * p + a], vkb = V[k * p + b];
                V[k * p + a] = c * vka - sn * vkb; V[k * p + b] = sn * vka + c * vkb;
            }
            for (int k = 0; k < p; k++) {
                const double aak = A[a * p + k], abk = A[b * p + k];
                A[a * p + k] = c * aak - sn * abk; A[b * p + k] = sn * aak + c * abk;
            }
        }
    }
    std::vector<int> order(p);
    for (int k = 0; k < p; k++) order[k] = k;
    for (int i = 1; i < p; i++) {
        const int o = order[i]; int j = i - 1;
        while (j >= 0 && A[order[j] * p + order[j]] < A[o * p + o]) { order[j + 1] = order[j]; j--; }
        order[j + 1] = o;
    }
    const double ev0 = A[order[0] * p + order[0]];
    for (int k = 0; k < p; k++) {
        const int o = order[k];
        const double ev = A[o * p + o];
        if (!(ev >= -1e-6 * fabs(ev0))) ARGFAIL("'Sigma' is not positive definite");
        int big = 0;
        for (int a = 1; a < p; a++) if (fabs(V[a * p + o]) > fabs(V[big * p + o])) big = a;
        const double sg = (V[big * p + o] < 0) ? -1.0 : 1.0;
        for (int a = 0; a < p; a++) fac[a * p + k] = (sg * V[a * p + o]) * sqrt(ev > 0 ? ev : 0.0);
    }
    return BSSM_OK;
}

// One chain's Metropolis-Hastings state: everything of R/pmmh.R:345-505 except the filter run itself, so that the
// same code drives one chain over bssm_pf_run and many chains in lock-step over bssm_pf_run_batch.
struct ChainState {
    const bssm_pmmh_config* cfg; bssm_pmmh_result* res;
    int p, m, T, dim, n_full;
    std::vector<double> L, cur, prop, lp_prop, th_full, se_cur, se_prop;
    PhiloxKey ckey;
    double cur_ll; int accepted;

    int init(const bssm_pmmh_config* cfg_, bssm_pmmh_result* res_)
    {
        cfg = cfg_; res = res_;
        if (!cfg || !res) ARGFAIL("bssm_pmmh_chain: NULL argument");
        p = cfg->n_params; m = cfg->m; T = cfg->pf.T;
        if (p < 1 || p > 16) ARGFAIL("bssm_pmmh_chain: n_params out of range");
        if (m < 1) ARGFAIL("Assertion on 'm' failed: Must be >= 1");                         // assert_int(m, lower = 1) R/pmmh.R:264
        if (!cfg->init_theta || !cfg->proposal_cov || !cfg->transform || !cfg->prior_kind || !cfg->prior_a || !cfg->prior_b)
            ARGFAIL("bssm_pmmh_chain: NULL configuration array");
        if (!res->theta_chain) ARGFAIL("bssm_pmmh_chain: theta_chain buffer missing");
        // proposal covariance on the transformed scale: J Sigma J, J = diag(dz/dtheta at init_theta)   R/pmmh.R:378-389
        std::vector<double> scale(p), cov((size_t)p * p);
        L.assign((size_t)p * p, 0.0);
        for (int j = 0; j < p; j++) {
            const double th = cfg->init_theta[j];
            scale[j] = cfg->transform[j] == BSSM_TR_LOG ? 1 / th : cfg->transform[j] == BSSM_TR_LOGIT ? 1 / (th * (1 - th)) : 1.0;
        }
        for (int a = 0; a < p; a++) for (int b = 0; b < p; b++) cov[a * p + b] = scale[a] * cfg->proposal_cov[a * p + b] * scale[b];
        // MASS::mvrnorm(1, mu, Sigma) (R/pmmh.R:425-428):  mu + V diag(sqrt(pmax(ev, 0))) z  with eigen(Sigma, symmetric = TRUE),
        // "'Sigma' is not positive definite" only when an eigenvalue is below -1e-6 |ev[1]| -- a positive SEMI-definite pilot
        // covariance (a parameter that never moved in the pilot's second half) is accepted, the chain just does not move in
        // the null directions.  Eigen-solver: cyclic Jacobi (p <= 16), eigenvalues decreasing, each eigenvector's largest
        // component positive (LAPACK's sign choice is build-dependent; same law either way).
        if (int rc = eigen_factor(p, cov.data(), L.data())) return rc;
        ckey = make_key(cfg->seed, 0x50000000ull + (unsigned long long)cfg->chain_index);
        cur.assign(cfg->init_theta, cfg->init_theta + p); prop.assign(p, 0.0); lp_prop.assign(p, 0.0);
        dim = (cfg->pf.model == BSSM_MODEL_SIR) ? 2 : 1;
        n_full = std::max(cfg->pf.n_theta, p);               // sampled parameters first, then fixed model constants
        th_full.assign((size_t)n_full, 0.0);
        for (int j = p; j < n_full; j++) th_full[j] = cfg->pf.theta ? cfg->pf.theta[j] : 0.0;
        se_cur.assign(((size_t)T + 1) * dim, 0.0); se_prop.assign(((size_t)T + 1) * dim, 0.0);
        cur_ll = 0; accepted = 0;
        return BSSM_OK;
    }
    // the filter's theta / stream for iteration `iter` with parameters `th`
    const double* full_theta(const std::vector<double>& th) { for (int j = 0; j < p; j++) th_full[j] = th[j]; return th_full.data(); }
    unsigned long long stream_of(unsigned iter) const { return ((unsigned long long)cfg->chain_index << 32) | iter; }
    void store(int i)
    {
        for (int j = 0; j < p; j++) res->theta_chain[(size_t)i * p + j] = cur[j];
        if (res->loglike_chain) res->loglike_chain[i] = cur_ll;
        if (res->state_est_chain) memcpy(res->state_est_chain + (size_t)i * (T + 1) * dim, se_cur.data(), sizeof(double) * (T + 1) * dim);
    }
    void start(double ll0) { cur_ll = ll0; store(0); }                                       // R/pmmh.R:403-417 (se_cur filled by the caller)
    // draw the proposal of iteration i (:424-432); false = a prior is -Inf, the chain stays put and no filter runs (:435-442)
    bool propose(int i)
    {
        std::vector<double> z(p), ztr(p);
        for (int j = 0; j < p; j++) {
            ztr[j] = tr_fwd(cfg->transform[j], cur[j]);
            z[j] = cfg->z_prop ? cfg->z_prop[(size_t)i * p + j] : host_normal(ckey, (uint32_t)i, (uint32_t)j);       // rnorm(p) of mvrnorm
        }
        for (int a = 0; a < p; a++) { double s = 0.0; for (int k = 0; k < p; k++) s += L[a * p + k] * z[k]; prop[a] = tr_back(cfg->transform[a], ztr[a] + s); }
        bool finite = true;
        for (int j = 0; j < p; j++) { lp_prop[j] = log_prior(cfg->prior_kind[j], cfg->prior_a[j], cfg->prior_b[j], prop[j]); if (!isfinite(lp_prop[j])) finite = false; }
        if (!finite) store(i);
        return finite;
    }
    // accept / reject with the proposal's log-likelihood (se_prop filled by the caller)   :461-496
    void finish(int i, double prop_ll)
    {
        long double lj_prop = 0, lj_cur = 0, slp_prop = 0, slp_cur = 0;                  // R's sum() accumulates in long double
        for (int j = 0; j < p; j++) {
            lj_prop += tr_logjac(cfg->transform[j], prop[j]); lj_cur += tr_logjac(cfg->transform[j], cur[j]);   // :461-469
            slp_prop += lp_prop[j]; slp_cur += log_prior(cfg->prior_kind[j], cfg->prior_a[j], cfg->prior_b[j], cur[j]);
        }
        const double num = prop_ll + (double)slp_prop + (double)lj_prop;                 // :475-478
        const double den = cur_ll + (double)slp_cur + (double)lj_cur;                    // :480-483
        double lar = num - den;                                                          // :485
        if (isnan(lar)) lar = -INFINITY;                                                 // :488-490
        const double u = cfg->u_accept ? cfg->u_accept[i] : host_uniform(ckey, (uint32_t)i);
        if (log(u) < lar) { cur = prop; cur_ll = prop_ll; se_cur = se_prop; accepted++; }   // :492-496
        store(i);
    }
};

extern "C" int bssm_pmmh_chain_draws(unsigned long long seed, int chain_index, int m, int n_params, double* z_prop_out, double* u_accept_out)
{
    if (m < 1 || n_params < 1 || !z_prop_out || !u_accept_out) ARGFAIL("bssm_pmmh_chain_draws: bad argument");
    const PhiloxKey ckey = make_key(seed, 0x50000000ull + (unsigned long long)chain_index);
    for (int i = 0; i < m; i++) {
        for (int j = 0; j < n_params; j++) z_prop_out[(size_t)i * n_params + j] = host_normal(ckey, (uint32_t)i, (uint32_t)j);
        u_accept_out[i] = host_uniform(ckey, (uint32_t)i);
    }
    return BSSM_OK;
}

extern "C" int bssm_pmmh_chain(bssm_ctx* c, const bssm_pmmh_config* cfg, bssm_pmmh_result* res)
{
    if (!c) ARGFAIL("bssm_pmmh_chain: NULL argument");
    ChainState ch;
    int rc = ch.init(cfg, res);
    if (rc) return rc;
    const int T = ch.T, m = ch.m;
    std::vector<double> ess((size_t)T + 1), llh((size_t)std::max(T, 1));
    double ll = 0, ms = 0, ms_total = 0;
    int ers = 0, nres = 0;
    bssm_pf_config pf = cfg->pf;
    pf.n_theta = ch.n_full; pf.return_particles = 0; pf.return_ancestors = 0; pf.z_init = pf.z_trans = pf.u_res = nullptr;
    pf.seed = cfg->seed;
    bssm_pf_result pr; memset(&pr, 0, sizeof(pr));
    pr.scan_stats = nullptr; pr.ess = ess.data(); pr.loglike_history = llh.data(); pr.loglike = &ll; pr.early_return_step = &ers; pr.n_res_calls = &nres; pr.device_ms = &ms;
    auto run_pf = [&](const std::vector<double>& th, std::vector<double>& se, unsigned iter) -> int {
        pf.theta = ch.full_theta(th); pr.state_est = se.data();
        pf.stream = ch.stream_of(iter);
        const int rc2 = bssm_pf_run(c, &pf, &pr);
        ms_total += ms;
        return rc2;
    };
    rc = run_pf(ch.cur, ch.se_cur, 0);                                                   // R/pmmh.R:403-417
    if (rc) return rc;
    ch.start(ll);
    for (int i = 1; i < m; i++) {                                                        // for (i in 2:m)  R/pmmh.R:422
        if (!ch.propose(i)) continue;
        rc = run_pf(ch.prop, ch.se_prop, (unsigned)i);                                   // :445-457
        if (rc) return rc;
        ch.finish(i, ll);
    }
    if (res->accepted) *res->accepted = ch.accepted;
    if (res->device_ms) *res->device_ms = ms_total;
    return BSSM_OK;
}

// Many chains in lock-step: iteration i of every chain proposes, ONE batched launch runs all the proposals' filters
// (one workgroup each), every chain accepts or rejects.  Chain k's result is exactly bssm_pmmh_chain's for cfgs[k]:
// chain draws are keyed by (seed, chain_index) and the batched filters are bit-identical to bssm_pf_run.
extern "C" int bssm_pmmh_chains_batch(bssm_ctx* c, int n_chains, const bssm_pmmh_config* cfgs, bssm_pmmh_result* ress)
{
    if (!c || !cfgs || !ress) ARGFAIL("bssm_pmmh_chains_batch: NULL argument");
    if (n_chains < 1) ARGFAIL("bssm_pmmh_chains_batch: n_chains must be positive");
    std::vector<ChainState> ch((size_t)n_chains);
    for (int k = 0; k < n_chains; k++) { const int rc = ch[k].init(&cfgs[k], &ress[k]); if (rc) return rc; }
    const bssm_pf_config& pf0 = cfgs[0].pf;
    const int T = ch[0].T, m = ch[0].m, nth = ch[0].n_full;
    for (int k = 1; k < n_chains; k++) {
        const bssm_pf_config& q = cfgs[k].pf;
        if (ch[k].m != m || ch[k].T != T || ch[k].n_full != nth || q.model != pf0.model || q.algorithm != pf0.algorithm ||
            q.num_particles != pf0.num_particles || q.resample_algorithm != pf0.resample_algorithm || q.resample_fn != pf0.resample_fn ||
            !(q.threshold == pf0.threshold || (isnan(q.threshold) && isnan(pf0.threshold))) || q.y != pf0.y || q.obs_times != pf0.obs_times)
            ARGFAIL("bssm_pmmh_chains_batch: the chains must share the data, the filter settings and m");
    }
    bssm_pf_config pf = pf0;
    pf.n_theta = nth; pf.theta = nullptr; pf.return_particles = 0; pf.return_ancestors = 0; pf.z_init = pf.z_trans = pf.u_res = nullptr;
    const int dim = ch[0].dim;
    std::vector<double> thetas((size_t)n_chains * nth), ll((size_t)n_chains), se((size_t)n_chains * (T + 1) * dim);
    std::vector<unsigned long long> seeds((size_t)n_chains), streams((size_t)n_chains);
    std::vector<int> who((size_t)n_chains);
    double ms = 0, ms_total = 0;
    bssm_pf_batch_result br; memset(&br, 0, sizeof(br));
    br.loglike = ll.data(); br.state_est = se.data(); br.device_ms = &ms;
    for (int i = 0; i < m; i++) {
        int F = 0;
        for (int k = 0; k < n_chains; k++) {
            const bool run = (i == 0) ? true : ch[k].propose(i);
            if (!run) continue;
            const double* th = ch[k].full_theta(i == 0 ? ch[k].cur : ch[k].prop);
            for (int j = 0; j < nth; j++) thetas[(size_t)F * nth + j] = th[j];
            seeds[F] = cfgs[k].seed; streams[F] = ch[k].stream_of((unsigned)i); who[F] = k; F++;
        }
        if (F == 0) continue;
        const int rc = bssm_pf_run_batch(c, &pf, F, thetas.data(), seeds.data(), streams.data(), &br);
        if (rc) return rc;
        ms_total += ms;
        for (int f = 0; f < F; f++) {
            ChainState& s = ch[who[f]];
            std::vector<double>& dst = (i == 0) ? s.se_cur : s.se_prop;
            memcpy(dst.data(), se.data() + (size_t)f * (T + 1) * dim, sizeof(double) * (T + 1) * dim);
            if (i == 0) s.start(ll[f]); else s.finish(i, ll[f]);
        }
    }
    for (int k = 0; k < n_chains; k++) {
        if (ress[k].accepted) *ress[k].accepted = ch[k].accepted;
        if (ress[k].device_ms) *ress[k].device_ms = ms_total / n_chains;
    }
    return BSSM_OK;
}

// The same lock-step loop for LARGE filters: iteration i of every chain proposes on the host, ONE run of bssm_pf_run_multi (the kernels of
// bssm_pf_run, one argument set per chain) filters all the proposals, every chain accepts or rejects (R/pmmh.R:422-500).  ctxs: one context per
// chain (at most 4).  Chain draws are keyed by (seed, chain index) and the filters are bit-identical to bssm_pf_run, so each chain equals
// bssm_pmmh_chain's.
extern "C" int bssm_pmmh_chains_multi(bssm_ctx* const* ctxs, int n_chains, const bssm_pmmh_config* cfgs, bssm_pmmh_result* ress)
{
    if (!ctxs || !cfgs || !ress) ARGFAIL("bssm_pmmh_chains_multi: NULL argument");
    if (n_chains < 1 || n_chains > MULTI_MAX) ARGFAIL("bssm_pmmh_chains_multi: 1 .. 4 chains per call");
    std::vector<ChainState> ch((size_t)n_chains);
    for (int k = 0; k < n_chains; k++) { const int rc = ch[k].init(&cfgs[k], &ress[k]); if (rc) return rc; }
    const bssm_pf_config& pf0 = cfgs[0].pf;
    const int T = ch[0].T, m = ch[0].m, nth = ch[0].n_full;
    for (int k = 1; k < n_chains; k++) {
        const bssm_pf_config& q = cfgs[k].pf;
        if (ch[k].m != m || ch[k].T != T || ch[k].n_full != nth || q.model != pf0.model || q.algorithm != pf0.algorithm ||
            q.num_particles != pf0.num_particles || q.resample_algorithm != pf0.resample_algorithm || q.resample_fn != pf0.resample_fn ||
            !(q.threshold == pf0.threshold || (isnan(q.threshold) && isnan(pf0.threshold))) || q.y != pf0.y || q.obs_times != pf0.obs_times)
            ARGFAIL("bssm_pmmh_chains_multi: the chains must share the data, the filter settings and m");
    }
    bssm_pf_config pf = pf0;
    pf.n_theta = nth; pf.theta = nullptr; pf.return_particles = 0; pf.return_ancestors = 0; pf.z_init = pf.z_trans = pf.u_res = nullptr;
    const int dim = ch[0].dim;
    std::vector<double> thetas((size_t)n_chains * nth), ll((size_t)n_chains), se((size_t)n_chains * (T + 1) * dim);
    std::vector<unsigned long long> seeds((size_t)n_chains), streams((size_t)n_chains);
    std::vector<int> who((size_t)n_chains);
    double ms = 0, ms_total = 0;
    bssm_pf_batch_result br; memset(&br, 0, sizeof(br));
    br.loglike = ll.data(); br.state_est = se.data(); br.device_ms = &ms;
    for (int i = 0; i < m; i++) {
        int F = 0;
        for (int k = 0; k < n_chains; k++) {
            const bool run = (i == 0) ? true : ch[k].propose(i);
            if (!run) continue;
            const double* th = ch[k].full_theta(i == 0 ? ch[k].cur : ch[k].prop);
            for (int j = 0; j < nth; j++) thetas[(size_t)F * nth + j] = th[j];
            seeds[F] = cfgs[k].seed; streams[F] = ch[k].stream_of((unsigned)i); who[F] = k; F++;
        }
        if (F == 0) continue;
        std::vector<bssm_ctx*> cx((size_t)F);
        for (int f = 0; f < F; f++) cx[f] = ctxs[who[f]];
        const int rc = bssm_pf_run_multi(cx.data(), F, &pf, thetas.data(), seeds.data(), streams.data(), &br);
        if (rc) return rc;
        ms_total += ms;
        for (int f = 0; f < F; f++) {
            ChainState& s = ch[who[f]];
            std::vector<double>& dst = (i == 0) ? s.se_cur : s.se_prop;
            memcpy(dst.data(), se.data() + (size_t)f * (T + 1) * dim, sizeof(double) * (T + 1) * dim);
            if (i == 0) s.start(ll[f]); else s.finish(i, ll[f]);
        }
    }
    for (int k = 0; k < n_chains; k++) {
        if (ress[k].accepted) *ress[k].accepted = ch[k].accepted;
        if (ress[k].device_ms) *ress[k].device_ms = ms_total / n_chains;
    }
    return BSSM_OK;
}
