// CDNA4 (gfx950) variable-elimination interpreter kernel.
//
// One persistent 256-lane workgroup (4 wave64) executes whole requests: it pulls a request from a
// device-wide ticket counter (heaviest first), then runs the request's step program back to back.
// A step is the fused replacement of `pointwise_mul(...)` + `.cdt.sum_out(x)`
// (sorobn/bayes_net.py:780-785, 233-256, 100-103):
//
//        psi[out] = sum_{x < cx}  prod_{j < n_in}  phi_j[ base_j + idx_j(out) + x * xs_j ]
//
// over dense fp64 tables; the product table of the reference (up to 4^11 rows on the 10x10 grid) is
// never materialised.  Intermediates live in the workgroup's private arena slot in HBM, so steps of
// one request only need a workgroup barrier between them (same CU, same L1) - no grid-wide sync, no
// cross-workgroup visibility protocol.
//
// Index math: the planner lays every table out with the longest-living variable fastest, splits the
// output axes into a lane-varying block (lo, 256..1024 cells: consecutive lanes = consecutive cells
// of the output and, by construction of the layouts, near-consecutive cells of the big input -> coalesced
// 512-B wave stores, wide loads) and a wave-uniform block (hi).  Lane offsets are decoded once per
// step, hi offsets are decoded 256 at a time by all lanes in parallel into LDS and then broadcast
// from LDS in the streaming loop, so the inner loop is loads + fp64 multiplies/adds only.
#pragma once
#include <hip/hip_runtime.h>

#include "planner.h"

namespace mibn {

constexpr int kWG = 256;
constexpr int kMaxC = kLoMax / kWG;  // output cells per lane in the lane-varying block
constexpr int kStepWordsMax = kHdrWords + 3 * kMaxIn + kMaxAxes + kMaxIn * kMaxAxes;

struct KernelArgs {
    const uint32_t *prog;       // step programs of the batch
    const uint64_t *prog_off;   // word offset of request i's program
    const int32_t *order;       // execution order (heaviest first)
    const double *pool;         // CPT tables (constants pool)
    double *arena;              // scratch: n_workgroups slots
    uint64_t slot_cells;        // doubles per slot
    double *results;            // dense posteriors of the batch
    uint32_t *ticket;           // work counter
    int32_t n_requests;
};

template <int NIN, int CX>
__device__ __forceinline__ void step_body(const uint32_t *sw, int (*sh_hoff)[kWG], const double *__restrict__ pool,
                                          double *__restrict__ slot, double *__restrict__ results, const int tid) {
    const uint32_t w0 = sw[0];
    const int na = (w0 >> 8) & 0xff;
    const int nlo = (w0 >> 16) & 0xff;
    const bool fin = (w0 >> 24) & 1;
    const int cx = CX ? CX : (int)sw[1];
    const int lo_cells = (int)sw[2];
    const int hi_cells = (int)sw[3];
    const uint64_t out_off = (uint64_t)sw[4] | ((uint64_t)sw[5] << 32);
    double *__restrict__ outp = (fin ? results : slot) + out_off;

    const double *inp[NIN];
    int xs[NIN];
#pragma unroll
    for (int j = 0; j < NIN; ++j) {
        const uint64_t o = (uint64_t)sw[kHdrWords + 3 * j] | ((uint64_t)sw[kHdrWords + 3 * j + 1] << 32);
        inp[j] = (o & kConstFlag) ? pool + (o & ~kConstFlag) : slot + o;
        xs[j] = (int)sw[kHdrWords + 3 * j + 2];
    }
    const uint32_t *card = sw + kHdrWords + 3 * NIN;
    const int *strd = (const int *)(card + na);  // strd[j * na + a]

    // lane-varying offsets, decoded once per step
    int lo_off[NIN][kMaxC];
#pragma unroll
    for (int c = 0; c < kMaxC; ++c) {
#pragma unroll
        for (int j = 0; j < NIN; ++j) lo_off[j][c] = 0;
        const int l = tid + c * kWG;
        if (l < lo_cells) {
            int r = l;
            for (int a = 0; a < nlo; ++a) {
                const int cd = (int)card[a];
                const int q = r / cd;
                const int d = r - q * cd;
                r = q;
#pragma unroll
                for (int j = 0; j < NIN; ++j) lo_off[j][c] += d * strd[j * na + a];
            }
        }
    }

    for (int h0 = 0; h0 < hi_cells; h0 += kWG) {
        {  // decode 256 wave-uniform offsets in parallel
            const int h = h0 + tid;
            if (h < hi_cells) {
                int acc[NIN];
#pragma unroll
                for (int j = 0; j < NIN; ++j) acc[j] = 0;
                int r = h;
                for (int a = nlo; a < na; ++a) {
                    const int cd = (int)card[a];
                    const int q = r / cd;
                    const int d = r - q * cd;
                    r = q;
#pragma unroll
                    for (int j = 0; j < NIN; ++j) acc[j] += d * strd[j * na + a];
                }
#pragma unroll
                for (int j = 0; j < NIN; ++j) sh_hoff[j][tid] = acc[j];
            }
        }
        __syncthreads();
        const int nh = min(kWG, hi_cells - h0);
        for (int hh = 0; hh < nh; ++hh) {
            int ho[NIN];
#pragma unroll
            for (int j = 0; j < NIN; ++j) ho[j] = sh_hoff[j][hh];
            const size_t orow = (size_t)(h0 + hh) * (size_t)lo_cells;
#pragma unroll
            for (int c = 0; c < kMaxC; ++c) {
                const int l = tid + c * kWG;
                if (l < lo_cells) {
                    double acc = 0.0;
                    if (CX) {
#pragma unroll
                        for (int x = 0; x < (CX ? CX : 1); ++x) {
                            double p = inp[0][ho[0] + lo_off[0][c] + x * xs[0]];
#pragma unroll
                            for (int j = 1; j < NIN; ++j) p *= inp[j][ho[j] + lo_off[j][c] + x * xs[j]];
                            acc += p;
                        }
                    } else {
                        for (int x = 0; x < cx; ++x) {
                            double p = inp[0][ho[0] + lo_off[0][c] + x * xs[0]];
#pragma unroll
                            for (int j = 1; j < NIN; ++j) p *= inp[j][ho[j] + lo_off[j][c] + x * xs[j]];
                            acc += p;
                        }
                    }
                    outp[orow + l] = acc;
                }
            }
        }
        __syncthreads();
    }
}

template <int NIN>
__device__ __forceinline__ void step_cx(const uint32_t *sw, int (*sh_hoff)[kWG], const double *pool, double *slot,
                                        double *results, int tid) {
    const int cx = (int)sw[1];
    if (cx == 4) step_body<NIN, 4>(sw, sh_hoff, pool, slot, results, tid);
    else if (cx == 2) step_body<NIN, 2>(sw, sh_hoff, pool, slot, results, tid);
    else if (cx == 1) step_body<NIN, 1>(sw, sh_hoff, pool, slot, results, tid);
    else step_body<NIN, 0>(sw, sh_hoff, pool, slot, results, tid);
}

// posterior / posterior.sum()  (bayes_net.py:790); an all-zero table (zero-probability evidence) stays zero
__device__ __forceinline__ void normalise(double *__restrict__ p, int n, double *sh_red, int tid) {
    double s = 0.0;
    for (int i = tid; i < n; i += kWG) s += p[i];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_down(s, o, 64);
    if ((tid & 63) == 0) sh_red[tid >> 6] = s;
    __syncthreads();
    const double total = (sh_red[0] + sh_red[1]) + (sh_red[2] + sh_red[3]);
    if (total > 0.0)
        for (int i = tid; i < n; i += kWG) p[i] = p[i] / total;
    __syncthreads();
}

__global__ __launch_bounds__(kWG) void ve_kernel(const KernelArgs A) {
    __shared__ uint32_t sh_step[kStepWordsMax + 6];
    __shared__ int sh_hoff[kMaxIn][kWG];
    __shared__ double sh_red[kWG / 64];
    __shared__ int sh_ticket;
    const int tid = threadIdx.x;
    double *slot = A.arena + (size_t)blockIdx.x * A.slot_cells;

    for (;;) {
        if (tid == 0) sh_ticket = (int)atomicAdd(A.ticket, 1u);
        __syncthreads();
        const int t = sh_ticket;
        if (t >= A.n_requests) break;
        const int req = A.order[t];
        const uint32_t *p = A.prog + A.prog_off[req];
        const int n_steps = (int)p[0];
        ++p;
        for (int s = 0; s < n_steps; ++s) {
            const int words = (int)p[6];
            __syncthreads();  // previous step's stores are done and visible to the workgroup; sh_step reusable
            if (tid < words) sh_step[tid] = p[tid];
            __syncthreads();
            const int n_in = sh_step[0] & 0xff;
            switch (n_in) {
                case 1: step_cx<1>(sh_step, sh_hoff, A.pool, slot, A.results, tid); break;
                case 2: step_cx<2>(sh_step, sh_hoff, A.pool, slot, A.results, tid); break;
                case 3: step_cx<3>(sh_step, sh_hoff, A.pool, slot, A.results, tid); break;
                case 4: step_cx<4>(sh_step, sh_hoff, A.pool, slot, A.results, tid); break;
                case 5: step_cx<5>(sh_step, sh_hoff, A.pool, slot, A.results, tid); break;
                default: step_cx<6>(sh_step, sh_hoff, A.pool, slot, A.results, tid); break;
            }
            if ((sh_step[0] >> 24) & 1) {
                const uint64_t out_off = (uint64_t)sh_step[4] | ((uint64_t)sh_step[5] << 32);
                const int n = (int)(sh_step[2] * sh_step[3]);
                normalise(A.results + out_off, n, sh_red, tid);
            }
            p += words;
        }
        __syncthreads();  // sh_ticket is rewritten next
    }
}

}  // namespace mibn
