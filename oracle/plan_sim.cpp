// TEST INFRASTRUCTURE - CPU interpreter for the planner's step programs.
//
// Links the product's host-side planner (sorobn_amd/csrc/planner.cpp, pure C++) and executes the
// step program it emits with plain scalar loops, so that the *host logic* (relevance pruning, order
// selection, layouts, strides, axis merging, arena allocation, evidence slicing) can be checked
// against the golden vectors in the GPU-less build container (`pytest -m "not gpu"`).  It mirrors
// the step semantics documented in sorobn_amd/csrc/planner.h / ve_kernel.hip.h:
//     psi[out] = sum_x prod_j phi_j[base_j + idx_j(out) + x*xs_j]        (bayes_net.py:780-785)
// and the final normalisation (bayes_net.py:790).
//
// This library is NOT part of the product and is never loaded by sorobn_amd: the product path has
// no CPU fallback (mibn_query_batch fails without a gfx950 device).  Only tests/ may load it.
#include <cstdint>
#include <cstring>
#include <string>
#include <vector>

#include "../sorobn_amd/csrc/planner.h"

using namespace mibn;

static std::string g_err;

extern "C" const char *plan_sim_error() { return g_err.c_str(); }

extern "C" int plan_sim_query(int32_t n_vars, const int32_t *card, const int64_t *scope_off, const int32_t *scope_vars,
                              const int64_t *value_off, const double *values, int32_t n_hints, const int32_t *hints,
                              int32_t nq, const int32_t *qvars, int32_t ne, const int32_t *evars, const int32_t *ecodes,
                              double *out, double *stats /* bytes, flops, steps, max_cells, arena_cells */) {
    Network net;
    g_err = net.set(n_vars, card, scope_off, scope_vars, value_off, values);
    if (!g_err.empty()) return -1;
    for (int i = 0; i < n_hints; ++i)
        net.hints.emplace_back(hints + (size_t)i * n_vars, hints + (size_t)(i + 1) * n_vars);
    Request rq;
    rq.nq = nq; rq.qvars = qvars; rq.ne = ne; rq.evars = evars; rq.ecodes = ecodes; rq.out_off = 0;
    g_err = validate_request(net, rq);
    if (!g_err.empty()) return -1;
    int64_t out_cells = 1;
    for (int i = 0; i < nq; ++i) out_cells *= card[qvars[i]];
    for (int64_t i = 0; i < out_cells; ++i) out[i] = 0.0;
    for (int i = 0; i < ne; ++i)
        if (ecodes[i] < 0 || ecodes[i] >= card[evars[i]]) return 0;  // label outside the domain: empty posterior
    std::vector<uint32_t> prog;
    PlanStats st;
    g_err = plan_request(net, rq, prog, st);
    if (!g_err.empty()) return -6;
    if (stats) { stats[0] = st.alg_bytes; stats[1] = st.alg_flops; stats[2] = st.n_steps; stats[3] = st.max_step_cells; stats[4] = (double)st.arena_cells; }
    std::vector<double> arena((size_t)st.arena_cells + 2, -1e300);  // poison: reading unwritten scratch shows up
    const uint32_t *p = prog.data();
    int n_steps = (int)*p++;
    for (int s = 0; s < n_steps; ++s) {
        const uint32_t w0 = p[0];
        const int n_in = w0 & 0xff, na = (w0 >> 8) & 0xff;
        const bool fin = (w0 >> 24) & 1;
        const int cx = (int)p[1];
        const int64_t cells = (int64_t)p[2] * (int64_t)p[3];
        const uint64_t out_off = (uint64_t)p[4] | ((uint64_t)p[5] << 32);
        const int words = (int)p[6];
        double *outp = fin ? out + out_off : arena.data() + out_off;
        if (!fin && (int64_t)out_off + cells > st.arena_cells) { g_err = "step writes outside its arena"; return -7; }
        const double *inp[kMaxIn];
        int xs[kMaxIn];
        for (int j = 0; j < n_in; ++j) {
            const uint64_t o = (uint64_t)p[kHdrWords + 3 * j] | ((uint64_t)p[kHdrWords + 3 * j + 1] << 32);
            inp[j] = (o & kConstFlag) ? net.pool.data() + (o & ~kConstFlag) : arena.data() + o;
            xs[j] = (int)p[kHdrWords + 3 * j + 2];
        }
        const uint32_t *cd = p + kHdrWords + 3 * n_in;
        const int32_t *strd = (const int32_t *)(cd + na);
        std::vector<double> tmp((size_t)cells);
        for (int64_t o = 0; o < cells; ++o) {
            int64_t r = o;
            int64_t off[kMaxIn] = {0};
            for (int a = 0; a < na; ++a) {
                const int64_t d = r % cd[a];
                r /= cd[a];
                for (int j = 0; j < n_in; ++j) off[j] += d * strd[j * na + a];
            }
            double acc = 0.0;
            for (int x = 0; x < cx; ++x) {
                double v = 1.0;
                for (int j = 0; j < n_in; ++j) v *= inp[j][off[j] + (int64_t)x * xs[j]];
                acc += v;
            }
            tmp[(size_t)o] = acc;
        }
        std::memcpy(outp, tmp.data(), sizeof(double) * (size_t)cells);
        if (fin) {
            double total = 0;
            for (int64_t i = 0; i < cells; ++i) total += outp[i];
            if (total > 0)
                for (int64_t i = 0; i < cells; ++i) outp[i] /= total;
        }
        p += words;
    }
    return 0;
}
