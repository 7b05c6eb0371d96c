// CDNA4 (gfx950) variable-elimination kernels.
//
// A step is the fused replacement of `pointwise_mul(...)` + `.cdt.sum_out(x)`
// (sorobn/bayes_net.py:780-785, 233-256, 100-103):
//
//        psi[out] = sum_{x < cx}  prod_{j < n_in}  phi_j[ base_j + idx_j(out) + x * xs_j ]
//
// over dense fp64 tables; the product table of the reference (up to 4^11 rows on the 10x10 grid) is
// never materialised.  Every request owns a private arena in HBM for its intermediates.
//
// Execution is level-synchronous (schedule: planner.h).  Three kernel families, all 256-lane
// workgroups (4 wave64), one work item per workgroup:
//  * fiber_tile_kernel<NBIG, CX, NCT> - the streaming form, > 95 % of the bytes on the 10x10 grid.  One
//    tile = 128 wave-uniform iterations x 256..512 lane cells of one big step.  The inputs are one or
//    two big tables (the elimination frontier, MBs, streamed from HBM) and a few CPT slices (<= 8 KiB)
//    whose product is tabulated once per tile in LDS (T, <= 16 KiB).  A lane owns one cell r of the big
//    tables' shared axes: it loads the cx values F[r, x] once - consecutive lanes read consecutive
//    addresses, 512 B per wave instruction - and produces the whole fiber over the new (CPT-only) axes
//    in registers,  out[r, n] = sum_x F[r, x] * T[n, x, ctrl(r)],  written as one contiguous NC*8-byte
//    vector store per lane.  One kernel per (NBIG, CX, NCT) shape: 44-76 VGPRs, 6 workgroups per CU.
//  * generic_tile_kernel<NIN> - big steps of any other shape, one output cell per lane-iteration.
//  * seg_kernel - a run of small steps of one request (start / end of a program, the final normalised
//    product) executed back to back by one workgroup; only `__syncthreads()` between steps (same CU).
//
// Index math: every table is laid out with the longest-living variable fastest, the iteration space
// is split into a lane-varying block (lo) and a wave-uniform block (hi).  Lane offsets are decoded
// once per item; hi offsets are decoded up to 256 at a time by all lanes in parallel into LDS and
// broadcast from LDS in the streaming loop, which therefore contains only loads, fp64 FMAs and stores.
#pragma once
#include <hip/hip_runtime.h>

#include "planner.h"

namespace mibn {

constexpr int kWG = 256;
constexpr int kFiberC = kFiberLoMax / kWG;  // R cells per lane in the lane-varying block

struct LevelArgs {
    const uint32_t *prog;       // step programs of the chunk
    const uint64_t *prog_off;   // word offset of request i's program
    const uint64_t *arena_off;  // offset (doubles) of request i's private arena
    const double *pool;         // CPT tables (constants pool)
    double *arena;              // scratch
    double *results;            // dense posteriors of the chunk
    const Item *items;          // work items of this launch (one per workgroup)
};

__device__ __forceinline__ const double *table_ptr(uint32_t lo, uint32_t hi, const double *pool, const double *slot) {
    const uint64_t o = (uint64_t)lo | ((uint64_t)hi << 32);
    return (o & kConstFlag) ? pool + (o & ~kConstFlag) : slot + o;
}

// ---------------------------------------------------------------------------------------- GENERIC
template <int NIN, int MAXC, int CX>
__device__ __forceinline__ void generic_body(const uint32_t *sw, int (*sh_hoff)[kWG], const double *__restrict__ pool,
                                             double *__restrict__ slot, double *__restrict__ results, const int tid,
                                             const int h_begin, const int h_end) {
    const uint32_t w0 = sw[0];
    const int na = (w0 >> 16) & 0xff;
    const int nlo = (w0 >> 24) & 0xff;
    const bool fin = (sw[1] >> 16) & 1;
    const int cx = CX ? CX : (int)(sw[1] & 0xffff);
    const int lo_cells = (int)sw[2];
    const uint64_t out_off = (uint64_t)sw[4] | ((uint64_t)sw[5] << 32);
    double *__restrict__ outp = (fin ? results : slot) + out_off;

    const double *inp[NIN];
    int xs[NIN];
#pragma unroll
    for (int j = 0; j < NIN; ++j) {
        inp[j] = table_ptr(sw[kHdrWords + 3 * j], sw[kHdrWords + 3 * j + 1], pool, slot);
        xs[j] = (int)sw[kHdrWords + 3 * j + 2];
    }
    const uint32_t *card = sw + kHdrWords + 3 * NIN;
    const int *strd = (const int *)(card + na);  // strd[j * na + a]

    int lo_off[NIN][MAXC];
#pragma unroll
    for (int c = 0; c < MAXC; ++c) {
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

    for (int h0 = h_begin; h0 < h_end; h0 += kWG) {
        {
            const int h = h0 + tid;
            if (h < h_end) {
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
        const int nh = min(kWG, h_end - h0);
        for (int hh = 0; hh < nh; ++hh) {
            int ho[NIN];
#pragma unroll
            for (int j = 0; j < NIN; ++j) ho[j] = sh_hoff[j][hh];
            const size_t orow = (size_t)(h0 + hh) * (size_t)lo_cells;
#pragma unroll
            for (int c = 0; c < MAXC; ++c) {
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

template <int NIN, int MAXC>
__device__ __forceinline__ void generic_cx(const uint32_t *sw, int (*sh_hoff)[kWG], const double *pool, double *slot,
                                           double *results, int tid, int h_begin, int h_end) {
    const int cx = (int)(sw[1] & 0xffff);
    if (cx == 4) generic_body<NIN, MAXC, 4>(sw, sh_hoff, pool, slot, results, tid, h_begin, h_end);
    else if (cx == 2) generic_body<NIN, MAXC, 2>(sw, sh_hoff, pool, slot, results, tid, h_begin, h_end);
    else generic_body<NIN, MAXC, 0>(sw, sh_hoff, pool, slot, results, tid, h_begin, h_end);
}

// ------------------------------------------------------------------------------------------ FIBER
// NBIG big inputs, CX compile-time x-cardinality (0 = runtime), NCT = register capacity of the N-fiber.
template <int NBIG, int CX, int NCT>
__device__ __forceinline__ void fiber_body(const uint32_t *sw, double *__restrict__ shT, int (*sh_hoff)[kWG],
                                           const double *__restrict__ pool, double *__restrict__ slot, const int tid,
                                           const int h_begin, const int h_end) {
    const uint32_t w0 = sw[0];
    const int na = (w0 >> 16) & 0xff;
    const int nlo = (w0 >> 24) & 0xff;
    const int cx = CX ? CX : (int)(sw[1] & 0xffff);
    const int lo_cells = (int)sw[2];
    double *__restrict__ outp = slot + ((uint64_t)sw[4] | ((uint64_t)sw[5] << 32));
    const int ns = (sw[7] >> 4) & 0xf, nN = (sw[7] >> 8) & 0xf, nctrl = (sw[7] >> 12) & 0xf;
    const int NC = (int)(sw[7] >> 16);
    const int T = (int)sw[8];
    const int nT = nN + nctrl;

    const uint32_t *q = sw + kHdrWords;
    const double *__restrict__ big[NBIG];
    int bxs[NBIG];
#pragma unroll
    for (int b = 0; b < NBIG; ++b) {
        big[b] = table_ptr(q[0], q[1], pool, slot);
        bxs[b] = (int)q[2];
        q += 3;
    }
    const uint32_t *smalls = q;  // ns records of (3 + nT) words
    q += ns * (3 + nT);
    const uint32_t *tcard = q;
    q += nT;
    const uint32_t *nout = q;
    q += NC;
    const uint32_t *rax = q;  // (card, ostride, tstride) per R axis
    q += 3 * na;
    const int *bst = (const int *)q;  // bst[b * na + a]

    // T[n + NC*(x + cx*ctrl)] = product of the small inputs (the CPT slices), once per step
    for (int t = tid; t < T; t += kWG) {
        int r = t;
        int qn = r / NC;
        int rn = r - qn * NC;
        r = qn;
        const int qx = r / cx;
        const int x = r - qx * cx;
        r = qx;
        double v = 1.0;
        for (int j = 0; j < ns; ++j) {
            const uint32_t *rec = smalls + j * (3 + nT);
            const int *sts = (const int *)(rec + 3);
            int off = x * (int)rec[2];
            int a = rn, c = r;
            for (int k = 0; k < nN; ++k) { const int cd = (int)tcard[k]; const int qq = a / cd; off += (a - qq * cd) * sts[k]; a = qq; }
            for (int k = nN; k < nT; ++k) { const int cd = (int)tcard[k]; const int qq = c / cd; off += (c - qq * cd) * sts[k]; c = qq; }
            v *= table_ptr(rec[0], rec[1], pool, slot)[off];
        }
        shT[t] = v;
    }
    // are the N-fiber's output cells contiguous (n fastest)?  then it is one vector store per lane
    bool contig = true;
    for (int n = 0; n < NC; ++n) contig = contig && (nout[n] == (uint32_t)n);

    int lo_o[kFiberC], lo_t[kFiberC], lo_b[NBIG][kFiberC];
#pragma unroll
    for (int c = 0; c < kFiberC; ++c) {
        lo_o[c] = 0;
        lo_t[c] = 0;
#pragma unroll
        for (int b = 0; b < NBIG; ++b) lo_b[b][c] = 0;
        const int l = tid + c * kWG;
        if (l < lo_cells) {
            int r = l;
            for (int a = 0; a < nlo; ++a) {
                const int cd = (int)rax[3 * a];
                const int qq = r / cd;
                const int d = r - qq * cd;
                r = qq;
                lo_o[c] += d * (int)rax[3 * a + 1];
                lo_t[c] += d * (int)rax[3 * a + 2];
#pragma unroll
                for (int b = 0; b < NBIG; ++b) lo_b[b][c] += d * bst[b * na + a];
            }
        }
    }

    for (int h0 = h_begin; h0 < h_end; h0 += kWG) {
        {
            const int h = h0 + tid;
            if (h < h_end) {
                int ao = 0, at = 0, ab[NBIG];
#pragma unroll
                for (int b = 0; b < NBIG; ++b) ab[b] = 0;
                int r = h;
                for (int a = nlo; a < na; ++a) {
                    const int cd = (int)rax[3 * a];
                    const int qq = r / cd;
                    const int d = r - qq * cd;
                    r = qq;
                    ao += d * (int)rax[3 * a + 1];
                    at += d * (int)rax[3 * a + 2];
#pragma unroll
                    for (int b = 0; b < NBIG; ++b) ab[b] += d * bst[b * na + a];
                }
                sh_hoff[0][tid] = ao;
                sh_hoff[1][tid] = at;
#pragma unroll
                for (int b = 0; b < NBIG; ++b) sh_hoff[2 + b][tid] = ab[b];
            }
        }
        __syncthreads();  // also orders the T build before its first use
        const int nh = min(kWG, h_end - h0);
        for (int hh = 0; hh < nh; ++hh) {
            const int ho = sh_hoff[0][hh], ht = sh_hoff[1][hh];
            int hb[NBIG];
#pragma unroll
            for (int b = 0; b < NBIG; ++b) hb[b] = sh_hoff[2 + b][hh];
            if (CX) {
                // all loads of this iteration (kFiberC cells x NBIG tables x CX values) are issued first
                double f[kFiberC][CX ? CX : 1];
#pragma unroll
                for (int c = 0; c < kFiberC; ++c) {
                    const bool ok = tid + c * kWG < lo_cells;
#pragma unroll
                    for (int x = 0; x < (CX ? CX : 1); ++x) {
                        double p = ok ? big[0][hb[0] + lo_b[0][c] + x * bxs[0]] : 0.0;
#pragma unroll
                        for (int b = 1; b < NBIG; ++b) p *= ok ? big[b][hb[b] + lo_b[b][c] + x * bxs[b]] : 0.0;
                        f[c][x] = p;
                    }
                }
#pragma unroll
                for (int c = 0; c < kFiberC; ++c) {
                    if (tid + c * kWG < lo_cells) {
                        const double *__restrict__ Tp = shT + (ht + lo_t[c]);
                        double *__restrict__ o = outp + (ho + lo_o[c]);
                        // the N-fiber is produced NCT outputs at a time (NCT = 4 covers NC = 16 in 4 rounds)
                        for (int n0 = 0; n0 < NC; n0 += NCT) {
                            double acc[NCT];
#pragma unroll
                            for (int n = 0; n < NCT; ++n) {
                                double s = 0.0;
                                if (n0 + n < NC) {
#pragma unroll
                                    for (int x = 0; x < (CX ? CX : 1); ++x) s += f[c][x] * Tp[x * NC + n0 + n];
                                }
                                acc[n] = s;
                            }
                            if (contig && NCT == 4 && n0 + 4 <= NC) {
                                *reinterpret_cast<double2 *>(o + n0) = make_double2(acc[0], acc[NCT >= 2 ? 1 : 0]);
                                *reinterpret_cast<double2 *>(o + n0 + 2) = make_double2(acc[NCT >= 4 ? 2 : 0], acc[NCT >= 4 ? 3 : 0]);
                            } else if (contig && NCT == 2 && n0 + 2 <= NC) {
                                *reinterpret_cast<double2 *>(o + n0) = make_double2(acc[0], acc[NCT >= 2 ? 1 : 0]);
                            } else {
#pragma unroll
                                for (int n = 0; n < NCT; ++n)
                                    if (n0 + n < NC) o[nout[n0 + n]] = acc[n];
                            }
                        }
                    }
                }
            } else {
#pragma unroll
                for (int c = 0; c < kFiberC; ++c) {
                    if (tid + c * kWG < lo_cells) {
                        const double *__restrict__ Tp = shT + (ht + lo_t[c]);
                        double *__restrict__ o = outp + (ho + lo_o[c]);
                        for (int n0 = 0; n0 < NC; n0 += NCT) {
                            double acc[NCT];
#pragma unroll
                            for (int n = 0; n < NCT; ++n) acc[n] = 0.0;
                            for (int x = 0; x < cx; ++x) {
                                double p = big[0][hb[0] + lo_b[0][c] + x * bxs[0]];
#pragma unroll
                                for (int b = 1; b < NBIG; ++b) p *= big[b][hb[b] + lo_b[b][c] + x * bxs[b]];
#pragma unroll
                                for (int n = 0; n < NCT; ++n)
                                    if (n0 + n < NC) acc[n] += p * Tp[x * NC + n0 + n];
                            }
#pragma unroll
                            for (int n = 0; n < NCT; ++n)
                                if (n0 + n < NC) o[nout[n0 + n]] = acc[n];
                        }
                    }
                }
            }
        }
        __syncthreads();
    }
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

// ---- out-of-line wrappers for the segment interpreter: each specialisation keeps its own register
// allocation instead of inflating one giant inlined kernel (which needed 153 VGPRs)
template <int NIN, int MAXC>
__device__ __noinline__ void generic_call(const uint32_t *sw, int (*sh_hoff)[kWG], const double *pool, double *slot,
                                          double *results, int tid) {
    generic_cx<NIN, MAXC>(sw, sh_hoff, pool, slot, results, tid, 0, (int)sw[3]);
}
template <int NBIG, int CX, int NCT>
__device__ __noinline__ void fiber_call(const uint32_t *sw, double *shT, int (*sh_hoff)[kWG], const double *pool, double *slot,
                                        int tid) {
    fiber_body<NBIG, CX, NCT>(sw, shT, sh_hoff, pool, slot, tid, 0, (int)sw[3]);
}
template <int NBIG>
__device__ __forceinline__ void fiber_dispatch(const uint32_t *sw, double *shT, int (*sh_hoff)[kWG], const double *pool,
                                               double *slot, int tid) {
    const int cx = (int)(sw[1] & 0xffff);
    const bool one = (sw[7] >> 16) <= 1;
    if (cx == 4) { if (one) fiber_call<NBIG, 4, 1>(sw, shT, sh_hoff, pool, slot, tid); else fiber_call<NBIG, 4, 4>(sw, shT, sh_hoff, pool, slot, tid); }
    else if (cx == 2) { if (one) fiber_call<NBIG, 2, 1>(sw, shT, sh_hoff, pool, slot, tid); else fiber_call<NBIG, 2, 4>(sw, shT, sh_hoff, pool, slot, tid); }
    else { if (one) fiber_call<NBIG, 0, 1>(sw, shT, sh_hoff, pool, slot, tid); else fiber_call<NBIG, 0, 4>(sw, shT, sh_hoff, pool, slot, tid); }
}

// SEGMENT: a run of (small) steps of one request, executed back to back by one workgroup
__global__ __launch_bounds__(kWG) void seg_kernel(const LevelArgs A) {
    __shared__ __attribute__((aligned(16))) double shT[kMaxT];
    __shared__ uint32_t sh_step[kMaxStepWords];
    __shared__ int sh_hoff[kMaxIn][kWG];
    __shared__ double sh_red[kWG / 64];
    const int tid = threadIdx.x;
    const Item it = A.items[blockIdx.x];
    double *slot = A.arena + A.arena_off[it.req];
    const uint32_t *p = A.prog + A.prog_off[it.req] + it.rel_off;
    const int n_steps = (int)it.a;
    for (int s = 0; s < n_steps; ++s) {
        const int words = (int)p[6];
        __syncthreads();  // previous step's stores are done and visible to the workgroup; sh_step reusable
        for (int i = tid; i < words; i += kWG) sh_step[i] = p[i];
        __syncthreads();
        const uint32_t kind = sh_step[0] & 0xff;
        const int n_in = (sh_step[0] >> 8) & 0xff;
        if (kind == kKindFiber) {
            if ((sh_step[7] & 0xf) == 1) fiber_dispatch<1>(sh_step, shT, sh_hoff, A.pool, slot, tid);
            else fiber_dispatch<2>(sh_step, shT, sh_hoff, A.pool, slot, tid);
        } else {
            switch (n_in) {
                case 1: generic_call<1, 2>(sh_step, sh_hoff, A.pool, slot, A.results, tid); break;
                case 2: generic_call<2, 2>(sh_step, sh_hoff, A.pool, slot, A.results, tid); break;
                case 3: generic_call<3, 2>(sh_step, sh_hoff, A.pool, slot, A.results, tid); break;
                case 4: generic_call<4, 1>(sh_step, sh_hoff, A.pool, slot, A.results, tid); break;
                case 5: generic_call<5, 1>(sh_step, sh_hoff, A.pool, slot, A.results, tid); break;
                default: generic_call<6, 1>(sh_step, sh_hoff, A.pool, slot, A.results, tid); break;
            }
            if ((sh_step[1] >> 16) & 1) {
                const uint64_t out_off = (uint64_t)sh_step[4] | ((uint64_t)sh_step[5] << 32);
                const int n = (int)(sh_step[2] * sh_step[3]);
                normalise(A.results + out_off, n, sh_red, tid);
            }
        }
        p += words;
    }
}

// TILE of a big FIBER step: hi iterations [a, b) of the step at rel_off
template <int NBIG, int CX, int NCT>
__global__ __launch_bounds__(kWG) void fiber_tile_kernel(const LevelArgs A) {
    __shared__ __attribute__((aligned(16))) double shT[kMaxT];
    __shared__ uint32_t sh_step[kMaxStepWords];
    __shared__ int sh_hoff[2 + NBIG][kWG];
    const int tid = threadIdx.x;
    const Item it = A.items[blockIdx.x];
    double *slot = A.arena + A.arena_off[it.req];
    const uint32_t *p = A.prog + A.prog_off[it.req] + it.rel_off;
    const int words = (int)p[6];
    for (int i = tid; i < words; i += kWG) sh_step[i] = p[i];
    __syncthreads();
    fiber_body<NBIG, CX, NCT>(sh_step, shT, sh_hoff, A.pool, slot, tid, (int)it.a, (int)it.b);
}

// TILE of a big GENERIC step
template <int NIN>
__global__ __launch_bounds__(kWG) void generic_tile_kernel(const LevelArgs A) {
    __shared__ uint32_t sh_step[kMaxStepWords];
    __shared__ int sh_hoff[NIN][kWG];
    const int tid = threadIdx.x;
    const Item it = A.items[blockIdx.x];
    double *slot = A.arena + A.arena_off[it.req];
    const uint32_t *p = A.prog + A.prog_off[it.req] + it.rel_off;
    const int words = (int)p[6];
    for (int i = tid; i < words; i += kWG) sh_step[i] = p[i];
    __syncthreads();
    generic_cx<NIN, (NIN <= 3 ? 2 : 1)>(sh_step, sh_hoff, A.pool, slot, A.results, tid, (int)it.a, (int)it.b);
}

}  // namespace mibn
