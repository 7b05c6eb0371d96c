// Forward (ancestral) sampling on gfx950 and the two approximate-inference loops built on it (SURVEY.md section 8f
// rank 2): BayesNet.sample / _forward_sample (sorobn/bayes_net.py:518-575), _rejection_sampling (577-619) and
// _llh_weighting (621-663).
//
// One sample per lane and trip: every lane walks the variables in topological order (= variable-id order, the order
// of `BayesNet.nodes`, bayes_net.py:319-322), looks up P(v | parents) in the dense CPT, draws v by inverse-CDF from a
// Philox4x32-10 uniform keyed by (seed, sample index, variable) - or takes the clamped `init` value - and multiplies
// the *likelihood* by P(value | parents) for EVERY node, clamped or not, exactly as the reference does (541-546).
//   mode SAMPLE     states[sample][v] = value code                                   (sample(n, init))
//   mode REJECTION  nothing is clamped; samples that agree with the event are counted per query cell   (606-619)
//   mode LIKELIHOOD the event is clamped; per query cell the likelihoods are summed and the samples counted - the
//                   reference returns groupby(query).mean() of the likelihood, normalised (658-663)
// The chain state of a lane lives in LDS (state[var][lane] bytes), histograms are accumulated per workgroup in LDS
// and flushed with one global atomic per cell.  Latency-bound like the Gibbs kernel; statistical parity only (the
// reference's stream depends on the absent third-party `vose` sampler, oracle/README.md).
#pragma once
#include <hip/hip_runtime.h>

#include <string>
#include <vector>

#include "../../include/mibn.h"
#include "gibbs_kernel.hip.h"
#include "planner.h"

namespace mibn {

struct SampleArgs {
    const double *pool;
    const GibbsVar *vars;         // is_evidence / ev_code = clamped (init) variables
    const int32_t *scope_var;
    const int32_t *scope_stride;
    const int32_t *qvars;
    const int32_t *qstride;
    const int32_t *evars;         // REJECTION: the event to agree with
    const int32_t *ecodes;
    uint8_t *states;              // SAMPLE: [n_samples][n_vars] (out); PROBE: the given joint states (in)
    double *cdf;                  // PROBE: [n_samples][n_vars][cdf_stride] running sums of the conditional row a draw of v compares u * total with
    double *lik;                  // PROBE: [n_samples] the likelihood of the given state
    int32_t cdf_stride;
    double *wsum;                 // LIKELIHOOD: sum of likelihoods per query cell
    unsigned long long *counts;   // samples per query cell (REJECTION: accepted ones)
    int32_t n_vars, n_q, n_e, hist_cells, mode;
    int64_t n_samples;
    uint64_t seed;
};

constexpr int kSampleMode = 0, kRejectionMode = 1, kLikelihoodMode = 2;
// PROBE (mibn_sample_probe, parity hook): the walk of the kernel over GIVEN joint states - same offsets, same row sums, same
// running sums a draw compares its uniform with, same likelihood product - written out instead of drawn from / histogrammed
constexpr int kProbeMode = 3;

__global__ __launch_bounds__(64) void sample_kernel(const SampleArgs A) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x;
    uint8_t *st = smem;                                                          // n_vars * 64 bytes
    double *hsum = (double *)(smem + ((A.n_vars * 64 + 15) & ~15));              // hist_cells
    unsigned int *hcnt = (unsigned int *)(hsum + A.hist_cells);                  // hist_cells
    for (int i = lane; i < A.hist_cells; i += 64) { hsum[i] = 0.0; hcnt[i] = 0u; }
    __syncthreads();
    const uint32_t k0 = (uint32_t)A.seed, k1 = (uint32_t)(A.seed >> 32) ^ 0x85EBCA6Bu;
    for (int64_t s0 = (int64_t)blockIdx.x * 64; s0 < A.n_samples; s0 += (int64_t)gridDim.x * 64) {
        const int64_t s = s0 + lane;
        const bool active = s < A.n_samples;
        double likelihood = 1.0;
        for (int v = 0; v < A.n_vars; ++v) {
            const GibbsVar V = A.vars[v];
            int off = V.table_off;
            for (int k = 0; k + 1 < V.scope_len; ++k)
                off += (int)st[A.scope_var[V.scope_begin + k] * 64 + lane] * A.scope_stride[V.scope_begin + k];
            int val = V.ev_code;
            const bool probe = A.mode == kProbeMode;
            if (!V.is_evidence || probe) {
                // P.cdt.sample(): a draw from the (possibly unnormalised / sparse) conditional row (bayes_net.py:28-42)
                double total = 0;
                for (int x = 0; x < V.card; ++x) total += A.pool[off + x];
                const double u = philox_uniform((uint64_t)s, 2u + (uint32_t)v, k0, k1) * total;
                double acc = 0;
                val = V.card - 1;
                for (int x = 0; x < V.card; ++x) {
                    acc += A.pool[off + x];
                    if (probe) { if (active) A.cdf[((size_t)s * A.n_vars + v) * A.cdf_stride + x] = acc; continue; }
                    if (u < acc) { val = x; break; }
                }
                if (probe) val = active ? (int)A.states[(size_t)s * A.n_vars + v] : 0;
            }
            st[v * 64 + lane] = (uint8_t)val;
            likelihood *= A.pool[off + val];  // P.get(node_value, 0): absent rows are 0 in the dense table
        }
        if (!active) continue;
        if (A.mode == kProbeMode) { A.lik[s] = likelihood; continue; }
        if (A.mode == kSampleMode) {
            for (int v = 0; v < A.n_vars; ++v) A.states[s * A.n_vars + v] = st[v * 64 + lane];
            continue;
        }
        bool keep = true;
        if (A.mode == kRejectionMode)
            for (int i = 0; i < A.n_e; ++i) keep = keep && (int)st[A.evars[i] * 64 + lane] == A.ecodes[i];
        if (keep) {
            int cell = 0;
            for (int q = 0; q < A.n_q; ++q) cell += (int)st[A.qvars[q] * 64 + lane] * A.qstride[q];
            atomicAdd(&hcnt[cell], 1u);
            if (A.mode == kLikelihoodMode) atomicAdd(&hsum[cell], likelihood);
        }
    }
    __syncthreads();
    if (A.mode != kSampleMode)
        for (int i = lane; i < A.hist_cells; i += 64) {
            if (hcnt[i]) atomicAdd(&A.counts[i], (unsigned long long)hcnt[i]);
            if (A.mode == kLikelihoodMode && hsum[i] != 0.0) atomicAdd(&A.wsum[i], hsum[i]);
        }
}

// host driver; returns MIBN_* code.  clamp_*: variables forced to a value (sample's `init`, likelihood weighting's
// event); ev_*: the event rejection sampling filters on.
inline int sample_run(const Network &net, const double *d_pool, hipStream_t stream, int mode, int32_t n_q, const int32_t *q_vars,
                      int32_t n_clamp, const int32_t *clamp_vars, const int32_t *clamp_codes, int32_t n_ev, const int32_t *ev_vars,
                      const int32_t *ev_codes, int64_t n_samples, uint64_t seed, uint8_t *states, double *wsum, int64_t *counts,
                      std::string &err, int32_t cdf_stride = 0, double *probe_lik = nullptr, double *probe_cdf = nullptr) {
    const int n = net.n_vars;
    std::vector<GibbsVar> vars(n);
    std::vector<int32_t> scope_var, scope_stride;
    for (int v = 0; v < n; ++v) {
        if (net.card[v] > 255) { err = "sampling: cardinality above 255"; return MIBN_E_LIMIT; }
        // topological order = id order: every parent must have a smaller id (true for BayesNet.nodes)
        for (size_t k = 0; k + 1 < net.scope[v].size(); ++k)
            if (net.scope[v][k] >= v) { err = "sampling: variable ids are not in topological order"; return MIBN_E_ARG; }
        GibbsVar g{};
        g.card = net.card[v];
        g.table_off = (int32_t)net.pool_off[v];
        g.scope_begin = (int32_t)scope_var.size();
        g.scope_len = (int32_t)net.scope[v].size();
        for (size_t k = 0; k < net.scope[v].size(); ++k) {
            scope_var.push_back(net.scope[v][k]);
            scope_stride.push_back((int32_t)net.cstride[v][k]);
        }
        vars[v] = g;
    }
    for (int i = 0; i < n_clamp; ++i) {
        const int v = clamp_vars[i];
        if (v < 0 || v >= n) { err = "sampling: unknown clamped variable"; return MIBN_E_ARG; }
        if (clamp_codes[i] < 0 || clamp_codes[i] >= net.card[v]) { err = "sampling: clamped label outside the domain"; return MIBN_E_ARG; }
        vars[v].is_evidence = 1;
        vars[v].ev_code = clamp_codes[i];
    }
    int64_t cells = 1;
    std::vector<int32_t> qstride(std::max(1, n_q));
    for (int i = n_q - 1; i >= 0; --i) { qstride[i] = (int32_t)cells; cells *= net.card[q_vars[i]]; }
    const size_t lds = ((size_t)n * 64 + 15) / 16 * 16 + (size_t)cells * 12;
    if (lds > 150 * 1024) { err = "sampling: network/query too large for the LDS-resident state"; return MIBN_E_LIMIT; }

    GibbsVar *d_vars = nullptr;
    int32_t *d_i32 = nullptr;
    unsigned char *d_out = nullptr;  // states | wsum + counts
    std::vector<int32_t> pack;
    auto put = [&](const std::vector<int32_t> &a) { size_t o = pack.size(); pack.insert(pack.end(), a.begin(), a.end()); return o; };
    const size_t o_sv = put(scope_var), o_ss = put(scope_stride);
    const size_t o_q = put(std::vector<int32_t>(q_vars, q_vars + n_q)), o_qs = put(qstride);
    const size_t o_ev = put(std::vector<int32_t>(ev_vars, ev_vars + n_ev)), o_ec = put(std::vector<int32_t>(ev_codes, ev_codes + n_ev));
    // PROBE: the given states | likelihoods | running sums
    const size_t probe_states = ((size_t)n_samples * n + 7) & ~size_t(7);
    const size_t out_bytes = mode == kSampleMode ? (size_t)n_samples * n
                             : mode == kProbeMode ? probe_states + (size_t)n_samples * 8 * (1 + (size_t)n * cdf_stride) : (size_t)cells * 16;
    auto fail = [&](hipError_t e) { err = std::string("sampling: ") + hipGetErrorString(e); hipFree(d_vars); hipFree(d_i32); hipFree(d_out); return MIBN_E_HIP; };
    hipError_t e;
    if ((e = hipMalloc(&d_vars, sizeof(GibbsVar) * std::max(1, n))) != hipSuccess) return fail(e);
    if ((e = hipMalloc(&d_i32, 4 * std::max<size_t>(1, pack.size()))) != hipSuccess) return fail(e);
    if ((e = hipMalloc(&d_out, std::max<size_t>(16, out_bytes))) != hipSuccess) return fail(e);
    if ((e = hipMemcpyAsync(d_vars, vars.data(), sizeof(GibbsVar) * n, hipMemcpyHostToDevice, stream)) != hipSuccess) return fail(e);
    if ((e = hipMemcpyAsync(d_i32, pack.data(), 4 * pack.size(), hipMemcpyHostToDevice, stream)) != hipSuccess) return fail(e);
    if ((e = hipMemsetAsync(d_out, 0, std::max<size_t>(16, out_bytes), stream)) != hipSuccess) return fail(e);
    if (mode == kProbeMode && (e = hipMemcpyAsync(d_out, states, (size_t)n_samples * n, hipMemcpyHostToDevice, stream)) != hipSuccess) return fail(e);
    SampleArgs A;
    A.pool = d_pool;
    A.vars = d_vars;
    A.scope_var = d_i32 + o_sv;
    A.scope_stride = d_i32 + o_ss;
    A.qvars = d_i32 + o_q;
    A.qstride = d_i32 + o_qs;
    A.evars = d_i32 + o_ev;
    A.ecodes = d_i32 + o_ec;
    A.states = d_out;
    A.wsum = (double *)d_out;
    A.counts = (unsigned long long *)(d_out + (size_t)cells * 8);
    A.lik = (double *)(d_out + probe_states);
    A.cdf = A.lik + n_samples;
    A.cdf_stride = cdf_stride;
    A.n_vars = n;
    A.n_q = n_q;
    A.n_e = n_ev;
    A.hist_cells = (mode == kSampleMode || mode == kProbeMode) ? 0 : (int32_t)cells;
    A.mode = mode;
    A.n_samples = n_samples;
    A.seed = seed;
    const unsigned blocks = (unsigned)std::min<int64_t>((n_samples + 63) / 64, 256 * 16);
    const size_t lds_launch = ((size_t)n * 64 + 15) / 16 * 16 + (size_t)A.hist_cells * 12;
    if (lds_launch > 64 * 1024)
        hipFuncSetAttribute((const void *)sample_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_launch);
    hipLaunchKernelGGL(sample_kernel, dim3(std::max(1u, blocks)), dim3(64), lds_launch, stream, A);
    if ((e = hipGetLastError()) != hipSuccess) return fail(e);
    std::vector<unsigned char> host(std::max<size_t>(16, out_bytes));
    if ((e = hipMemcpyAsync(host.data(), d_out, host.size(), hipMemcpyDeviceToHost, stream)) != hipSuccess) return fail(e);
    if ((e = hipStreamSynchronize(stream)) != hipSuccess) return fail(e);
    if (mode == kSampleMode) {
        std::memcpy(states, host.data(), out_bytes);
    } else if (mode == kProbeMode) {
        std::memcpy(probe_lik, host.data() + probe_states, (size_t)n_samples * 8);
        std::memcpy(probe_cdf, host.data() + probe_states + (size_t)n_samples * 8, (size_t)n_samples * 8 * n * cdf_stride);
    } else {
        const double *ws = (const double *)host.data();
        const unsigned long long *cn = (const unsigned long long *)(host.data() + (size_t)cells * 8);
        for (int64_t i = 0; i < cells; ++i) { if (wsum) wsum[i] = ws[i]; counts[i] = (int64_t)cn[i]; }
    }
    hipFree(d_vars);
    hipFree(d_i32);
    hipFree(d_out);
    return MIBN_OK;
}

}  // namespace mibn
