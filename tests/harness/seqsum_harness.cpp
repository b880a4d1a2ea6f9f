// Host-side harness for bayesssm_amd/csrc/seqsum.h  (TEST INFRASTRUCTURE).
//
// Emulates, in plain loops, the hierarchy the HIP kernels use -- per-thread
// chunk records from hypothetical prefixes, block-level Hillis-Steele scan of
// records, group-wise resolve over block records, literal fallback for HARD
// records -- so that the ALGORITHM (not the kernel wiring) can be checked on
// the CPU against the plain sequential sum for millions of inputs, including
// adversarial ones (ties, binade crossings, zeros, denormals, huge jumps).
// Built by tests/test_seqsum_harness.py with g++ -O2 -ffp-contract=off.
#include <math.h>
#include <stdlib.h>
#include <vector>
#include "../../bayesssm_amd/csrc/seqsum.h"

using namespace bssm;

struct Stats { long long hard_threads, hard_blocks, slow_block_walks, hard_groups, literal_terms; };

// approximate (re-associated) block sum: per-thread sequential, then pairwise tree
static double approx_block_sum(const double* v, int cnt, int L, std::vector<double>& tsum)
{
    int nt = (cnt + L - 1) / L;
    tsum.assign(nt, 0.0);
    for (int t = 0; t < nt; t++) {
        double s = 0;
        for (int k = t * L; k < cnt && k < (t + 1) * L; k++) s += v[k];
        tsum[t] = s;
    }
    std::vector<double> a(tsum);
    for (int n = nt; n > 1; n = (n + 1) / 2)
        for (int i = 0; i < n / 2; i++) a[i] = a[2 * i] + a[2 * i + 1], (void)0;
    // note: odd tails handled crudely on purpose (any association is allowed)
    double s = 0; for (int t = 0; t < nt; t++) s += tsum[t];
    return s;
}

extern "C" int harness_exact_cumsum(long long n, const double* v, int L, int NT,
                                    int lim_override, double* out, Stats* st)
{
    const int EB = L * NT;                        // elements per block
    const long long B = (n + EB - 1) / EB;
    const int32_t lim = lim_override > 0 ? lim_override : rec_window(n);
    st->hard_threads = st->hard_blocks = st->slow_block_walks = st->hard_groups = st->literal_terms = 0;
    std::vector<double> bsum(B), ain(B), tsum;
    for (long long b = 0; b < B; b++) {
        int cnt = (int)((b + 1) * EB <= n ? EB : n - b * EB);
        bsum[b] = approx_block_sum(v + b * EB, cnt, L, tsum);
    }
    { double a = 0; for (long long b = 0; b < B; b++) { ain[b] = a; a += bsum[b]; } }

    // thread records + block records
    auto thread_recs = [&](long long b, std::vector<Rec>& recs) {
        int cnt = (int)((b + 1) * EB <= n ? EB : n - b * EB);
        const double* vb = v + b * EB;
        recs.resize(NT);
        // in-block approximate exclusive prefix (tree-ish: Hillis-Steele on thread sums)
        std::vector<double> ts(NT, 0.0), ex(NT, 0.0);
        for (int t = 0; t < NT; t++) { double s = 0; for (int k = t * L; k < cnt && k < (t + 1) * L; k++) s += vb[k]; ts[t] = s; }
        std::vector<double> inc(ts);
        for (int off = 1; off < NT; off <<= 1) { std::vector<double> nx(inc); for (int t = off; t < NT; t++) nx[t] = inc[t - off] + inc[t]; inc.swap(nx); }
        for (int t = 0; t < NT; t++) ex[t] = t ? inc[t - 1] : 0.0;
        for (int t = 0; t < NT; t++) {
            int lo = t * L, len = cnt - lo; if (len > L) len = L; if (len < 0) len = 0;
            double h = ain[b] + ex[t];
            recs[t] = chunk_record(vb + lo, len, 1, h, lim);
            if (recs[t].kind == REC_HARD) st->hard_threads++;
        }
    };
    auto scan_block = [&](const std::vector<Rec>& recs, std::vector<Rec>& excl, Rec& total) {
        // Hillis-Steele inclusive scan with rec_compose, as the kernel does in LDS
        std::vector<Rec> inc(recs);
        for (int off = 1; off < NT; off <<= 1) { std::vector<Rec> nx(inc); for (int t = off; t < NT; t++) nx[t] = rec_compose(inc[t - off], inc[t]); inc.swap(nx); }
        excl.resize(NT);
        for (int t = 0; t < NT; t++) excl[t] = t ? inc[t - 1] : Rec();
        total = inc[NT - 1];
    };
    // literal walk of one block with exact incoming state; fills per-thread exact in-states
    auto walk_block_exact = [&](long long b, const std::vector<Rec>& recs, uint64_t in, std::vector<uint64_t>* tin) -> uint64_t {
        int cnt = (int)((b + 1) * EB <= n ? EB : n - b * EB);
        const double* vb = v + b * EB;
        uint64_t s = in;
        for (int t = 0; t < NT; t++) {
            if (tin) (*tin)[t] = s;
            bool ok = true;
            uint64_t o = rec_step(recs[t], s, ok);
            if (!ok) { int lo = t * L, len = cnt - lo; if (len > L) len = L; if (len < 0) len = 0; o = run_literal(vb + lo, len, 1, s); st->literal_terms += len; }
            s = o;
        }
        return s;
    };

    // ---- "local" pass: block record = composite of threads [0, tail_from); the
    // threads from tail_from on (first HARD inclusive composite) are re-run
    // literally by the resolve pass.  Blocks whose incoming state is exactly 0
    // walk themselves and publish an ABS record.
    struct BRec { Rec prefix; int tail_from; };
    std::vector<BRec> brec(B);
    std::vector<Rec> recs, excl;
    auto block_cnt = [&](long long b) { return (int)((b + 1) * EB <= n ? EB : n - b * EB); };
    auto literal_tail = [&](long long b, int tail_from, uint64_t in) -> uint64_t {
        int cnt = block_cnt(b), lo = tail_from * L; if (lo > cnt) lo = cnt;
        st->literal_terms += cnt - lo;
        return run_literal(v + b * EB + lo, cnt - lo, 1, in);
    };
    for (long long b = 0; b < B; b++) {
        thread_recs(b, recs);
        Rec tot; scan_block(recs, excl, tot);
        int tail = NT;
        for (int t = 0; t < NT; t++) { Rec inc = (t + 1 < NT) ? excl[t + 1] : tot; if (inc.kind == REC_HARD) { tail = t; break; } }
        BRec br; br.tail_from = tail;
        br.prefix = tail == 0 ? rec_identity(d2b(ain[b])) : (tail < NT ? excl[tail] : tot);
        if (tail < NT && ain[b] == 0.0) {                // exact start known: walk now
            uint64_t o = walk_block_exact(b, recs, 0, nullptr);
            br.prefix = rec_abs(o); br.tail_from = NT; st->slow_block_walks++;
        }
        if (br.tail_from < NT) st->hard_blocks++;
        brec[b] = br;
    }
    // ---- "resolve" pass: groups of G blocks; group records by sequential
    // composition; exact walk over groups; literal tails where needed
    long long G = 1; while (G * G < B) G++;
    long long NG = (B + G - 1) / G;
    std::vector<Rec> grec(NG);
    for (long long g = 0; g < NG; g++) {
        Rec r = brec[g * G].tail_from < NT ? rec_hard(0) : brec[g * G].prefix;
        for (long long b = g * G + 1; b < B && b < (g + 1) * G; b++)
            r = brec[b].tail_from < NT ? rec_hard(0) : rec_compose(r, brec[b].prefix);
        grec[g] = r; if (r.kind == REC_HARD) st->hard_groups++;
    }
    std::vector<uint64_t> cin(B);
    auto block_out_exact = [&](long long b, uint64_t in) -> uint64_t {
        bool ok = true; uint64_t o = rec_step(brec[b].prefix, in, ok);
        if (ok) return brec[b].tail_from < NT ? literal_tail(b, brec[b].tail_from, o) : o;
        st->slow_block_walks++;                           // window miss: whole block literally
        return literal_tail(b, 0, in);
    };
    std::vector<uint64_t> gin(NG);
    { uint64_t s = 0;
      for (long long g = 0; g < NG; g++) {
          gin[g] = s; bool ok = true; uint64_t o = rec_step(grec[g], s, ok);
          if (!ok) { o = s; for (long long b = g * G; b < B && b < (g + 1) * G; b++) o = block_out_exact(b, o); }
          s = o;
      } }
    for (long long g = 0; g < NG; g++) { uint64_t s = gin[g]; for (long long b = g * G; b < B && b < (g + 1) * G; b++) { cin[b] = s; s = block_out_exact(b, s); } }

    // ---- "apply" pass: per-thread exact in-states from the exclusive scan; from
    // the first thread whose prefix does not cover the exact state, a serial
    // walk (record step or literal) continues to the end of the block
    std::vector<uint64_t> tin(NT);
    for (long long b = 0; b < B; b++) {
        int cnt = block_cnt(b);
        const double* vb = v + b * EB;
        thread_recs(b, recs);
        Rec tot; scan_block(recs, excl, tot);
        int first_bad = NT;
        for (int t = 0; t < NT; t++) {
            if (t == 0) { tin[0] = cin[b]; continue; }
            bool ok = true; tin[t] = rec_step(excl[t], cin[b], ok);
            if (!ok) { first_bad = t; break; }
        }
        if (first_bad < NT) {
            st->slow_block_walks++;
            uint64_t s = tin[first_bad - 1];
            for (int t = first_bad - 1; t < NT; t++) {
                tin[t] = s; bool ok = true; uint64_t o = rec_step(recs[t], s, ok);
                if (!ok) { int lo = t * L, len = cnt - lo; if (len > L) len = L; if (len < 0) len = 0; o = run_literal(vb + lo, len, 1, s); st->literal_terms += len; }
                s = o;
            }
        }
        for (int t = 0; t < NT; t++) {
            double c = b2d(tin[t]);
            for (int k = t * L; k < cnt && k < (t + 1) * L; k++) { c = c + vb[k]; out[b * EB + k] = c; }
        }
    }
    return 0;
}

extern "C" int harness_count_systematic(double c, int n, double U) { return count_le_systematic(c, n, U); }
extern "C" int harness_count_stratified(double c, int n, const double* U) { UniformArray ua{U}; return count_le_stratified(c, n, ua); }
