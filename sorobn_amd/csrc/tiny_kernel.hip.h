// Small-network specialisation of the exact path (BASELINE config 2: Asia, 8 binary nodes, 36 CPT numbers).
//
// When a network is tiny - at most 32 variables, all CPTs together at most 4096 cells (32 KiB: they live in LDS) and a
// full joint of at most 65536 states - planning a step program per request costs far more than answering it.  Here one
// lane answers one request straight from the request arrays the caller handed over: no per-request planning, no step
// program, no arena.  The lane
//   * decodes its query / evidence variables and codes, builds the relevant set as a bit mask - query | evidence |
//     ancestors, exactly the pruning of BayesNet._variable_elimination (sorobn/bayes_net.py:763-765; MIBN_Q_NOPRUNE:
//     every variable, the semantics of full_joint_dist, bayes_net.py:460),
//   * for every joint state of the query variables enumerates the states of the hidden variables (766) in
//     TOPOLOGICAL order (Network::topo_asc: by depth, then id - the ids themselves may be any acyclic numbering) with running
//     prefix products - a state change at position k re-evaluates only the CPTs at positions >= k, which is every CPT
//     that mentions the variable because a CPT's parents come before it; about two table reads per state - and sums the product of the relevant CPTs: the
//     same sum-product the reference evaluates by pointwise_mul / sum_out (233-256, 100-103), in a different order of
//     additions (agreement ~1e-16),
//   * normalises (790) and writes the dense posterior; zero-probability or out-of-domain evidence gives all zeros
//     (the reference's empty Series).
// Lanes of a wave run different loop counts (divergence), which is irrelevant next to the host work this removes: the
// whole 100 k-request Asia batch is microseconds of kernel time.  Not HBM-bound: roofline n/a.
#pragma once
#include <hip/hip_runtime.h>

#include <string>
#include <vector>

#include "../../include/mibn.h"
#include "planner.h"

namespace mibn {

constexpr int kTinyMaxVars = 32;
constexpr int kTinyMaxPool = 4096;        // doubles of CPT tables held in LDS
constexpr int64_t kTinyMaxJoint = 65536;  // states of the full joint (bounds the enumeration of one request)
constexpr int kTinyMaxQCells = 4096;      // cells of one request's query table (the joint at most)

inline bool tiny_eligible(const Network &net) {
    if (net.n_vars < 1 || net.n_vars > kTinyMaxVars || (int64_t)net.pool.size() > kTinyMaxPool) return false;
    int64_t joint = 1;
    for (int v = 0; v < net.n_vars; ++v) {
        if (net.card[v] > 255) return false;
        joint *= net.card[v];
        if (joint > kTinyMaxJoint) return false;
    }
    return true;
}

constexpr uint32_t kTinyFlagHostBad = 0x80000000u;  // TinyArgs::flags, engine-internal (never a MIBN_Q_* bit of the C-ABI)

struct TinyArgs {
    const double *pool;
    const int32_t *meta;  // card[n], pool_off[n], scope_begin[n + 1], anc_mask[n], topo[n], then (scope_var, scope_stride) pairs
    const int64_t *q_off, *e_off, *out_off;
    const int32_t *q_vars, *e_vars, *e_codes;
    double *out;          // results, out_off[0]-relative
    int32_t *bad;         // bad[0] = lowest index of a malformed request (unknown variable, duplicate, query/evidence overlap,
                          // out_off that does not match the query table), INT32_MAX if none: the host then builds the
                          // reference's error message (bayes_net.py:840-845) for that request
    int64_t B;
    int32_t n_vars, pool_cells, meta_words;
    uint32_t flags;
};

__global__ __launch_bounds__(64) void tiny_kernel(const TinyArgs A) {
    extern __shared__ __attribute__((aligned(16))) unsigned char tiny_smem[];
    const int lane = threadIdx.x;
    const int n = A.n_vars;
    double *lpool = reinterpret_cast<double *>(tiny_smem);
    double *P = lpool + A.pool_cells;                                       // [n + 1][64] prefix products
    int32_t *meta = reinterpret_cast<int32_t *>(P + (kTinyMaxVars + 1) * 64);
    uint8_t *st = reinterpret_cast<uint8_t *>(meta + A.meta_words);        // [n][64] current code of every variable
    uint8_t *seq = st + kTinyMaxVars * 64;                                  // [n][64] relevant variables in topological order
    for (int i = lane; i < A.pool_cells; i += 64) lpool[i] = A.pool[i];
    for (int i = lane; i < A.meta_words; i += 64) meta[i] = A.meta[i];
    __syncthreads();
    const int32_t *card = meta, *pool_off = meta + n, *scope_begin = meta + 2 * n, *anc = meta + 3 * n + 1;
    const int32_t *topo = meta + 4 * n + 1;  // all variables, parents before children
    const int32_t *scope = meta + 5 * n + 1;
    const int64_t out0 = A.out_off[0];
    for (int64_t b = (int64_t)blockIdx.x * 64 + lane; b < A.B; b += (int64_t)gridDim.x * 64) {
        const int64_t q0 = A.q_off[b], q1 = A.q_off[b + 1], e0 = A.e_off[b], e1 = A.e_off[b + 1];
        double *out = A.out + (A.out_off[b] - out0);
        const int qcells = (int)(A.out_off[b + 1] - A.out_off[b]);
        uint32_t qmask = 0, emask = 0, rel = 0;
        bool valid = true, malformed = q1 <= q0;
        int64_t want_cells = 1;
        for (int64_t i = q0; i < q1 && !malformed; ++i) {
            const int v = A.q_vars[i];
            if (v < 0 || v >= n || ((qmask >> v) & 1u)) { malformed = true; break; }
            qmask |= 1u << v;
            rel |= anc[v];
            want_cells *= card[v];
        }
        for (int64_t i = e0; i < e1 && !malformed; ++i) {
            const int v = A.e_vars[i], c = A.e_codes[i];
            if (v < 0 || v >= n || (((qmask | emask) >> v) & 1u)) { malformed = true; break; }
            emask |= 1u << v;
            rel |= anc[v];
            if (c < 0 || c >= card[v]) valid = false;  // label outside the domain: empty posterior
            else st[v * 64 + lane] = (uint8_t)c;
        }
        if (malformed || want_cells != (int64_t)qcells) {
            if (A.flags & kTinyFlagHostBad) {
                // zero-copy call (engine.hip run_tiny): one wave, request b = lane, `bad` in pinned host memory - no atomic minimum
                // over PCIe: the lowest malformed lane of the wave stores its index
                const uint64_t m = __ballot(1);
                if (lane == __ffsll((unsigned long long)m) - 1) *A.bad = (int32_t)b;
            } else {
                atomicMin(A.bad, (int32_t)(b < 0x7fffffff ? b : 0x7ffffffe));
            }
            continue;
        }
        rel |= qmask | emask;
        if (A.flags & MIBN_Q_NOPRUNE) rel = n >= 32 ? 0xffffffffu : ((1u << n) - 1u);
        if (!valid) {
            for (int c = 0; c < qcells; ++c) out[c] = 0.0;
            continue;
        }
        const uint32_t hidden = rel & ~qmask & ~emask;
        int ns = 0;
        for (int t = 0; t < n; ++t) {  // (ids need not be topological: a parent may carry a larger id than its child)
            const int v = topo[t];
            if ((rel >> v) & 1u) seq[(ns++) * 64 + lane] = (uint8_t)v;
        }
        double total = 0.0;
        for (int qc = 0; qc < qcells; ++qc) {
            {   // joint state qc of the query variables (C-order over the caller's argument order, last fastest)
                int r = qc;
                for (int64_t i = q1 - 1; i >= q0; --i) {
                    const int v = A.q_vars[i], cd = card[v];
                    st[v * 64 + lane] = (uint8_t)(r % cd);
                    r /= cd;
                }
            }
            for (uint32_t m = hidden; m; m &= m - 1) st[__builtin_ctz(m) * 64 + lane] = 0;
            double acc = 0.0;
            P[lane] = 1.0;
            int i = 0;
            for (;;) {
                for (; i < ns; ++i) {  // forward: the prefix products of the relevant CPTs from position i on
                    const int v = seq[i * 64 + lane];
                    int off = pool_off[v];
                    for (int k = scope_begin[v]; k < scope_begin[v + 1]; ++k) off += (int)st[scope[2 * k] * 64 + lane] * scope[2 * k + 1];
                    P[(i + 1) * 64 + lane] = P[i * 64 + lane] * lpool[off];
                }
                acc += P[ns * 64 + lane];
                int j = ns - 1;  // next hidden state: increment the last hidden variable that can still grow
                for (; j >= 0; --j) {
                    const int v = seq[j * 64 + lane];
                    if (!((hidden >> v) & 1u)) continue;
                    const int c = st[v * 64 + lane] + 1;
                    if (c < card[v]) { st[v * 64 + lane] = (uint8_t)c; break; }
                    st[v * 64 + lane] = 0;
                }
                if (j < 0) break;
                i = j;
            }
            out[qc] = acc;
            total += acc;
        }
        if (total > 0.0)
            for (int c = 0; c < qcells; ++c) out[c] = out[c] / total;  // posterior / posterior.sum(), bayes_net.py:790
    }
}

// Device-side description of the network for tiny_kernel, built once per set_network.
inline std::vector<int32_t> tiny_meta(const Network &net) {
    const int n = net.n_vars;
    std::vector<int32_t> m;
    for (int v = 0; v < n; ++v) m.push_back(net.card[v]);
    for (int v = 0; v < n; ++v) m.push_back((int32_t)net.pool_off[v]);
    int32_t run = 0;
    for (int v = 0; v < n; ++v) { m.push_back(run); run += (int32_t)net.scope[v].size(); }
    m.push_back(run);
    for (int v = 0; v < n; ++v) m.push_back((int32_t)(uint32_t)net.anc[v].w[0]);
    for (int v = 0; v < n; ++v) m.push_back(net.topo_asc[v]);
    for (int v = 0; v < n; ++v)
        for (size_t k = 0; k < net.scope[v].size(); ++k) { m.push_back(net.scope[v][k]); m.push_back((int32_t)net.cstride[v][k]); }
    return m;
}

inline size_t tiny_lds_bytes(int pool_cells, int meta_words) {
    return (size_t)pool_cells * 8 + (size_t)(kTinyMaxVars + 1) * 64 * 8 + (size_t)meta_words * 4 + 2 * (size_t)kTinyMaxVars * 64;
}

}  // namespace mibn
