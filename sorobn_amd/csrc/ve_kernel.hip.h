// CDNA4 (gfx950) variable-elimination kernels.
//
// A step is the fused replacement of `pointwise_mul(...)` + `.cdt.sum_out(x)`
// (sorobn/bayes_net.py:780-785, 233-256, 100-103):
//
//        psi[out] = sum_{x < cx}  prod_{j < n_in}  phi_j[ base_j + idx_j(out) + off_j(x) ]
//
// over dense fp64 tables; the product table of the reference (up to 4^11 rows on the 10x10 grid) is
// never materialised.  Every request owns a private arena in HBM for its intermediates.
//
// Execution is level-synchronous (schedule: planner.h).  Three kernel families, all 256-lane
// workgroups (4 wave64):
//  * fiber_tile_kernel<NBIG, CXC, NCC> - the streaming form, > 95 % of the bytes on the 10x10 grid.  A tile =
//    a few wave-uniform iterations x <= 256 lane cells of one big step (~512 KiB of traffic).  The inputs are
//    one or two big tables (the elimination frontier, MBs, streamed from HBM) and a few CPT slices (<= 8 KiB)
//    whose product is tabulated once per tile in LDS (T, <= 16 KiB).  A lane owns one cell r of the big
//    tables' shared axes: it loads the cx values F[r, x] up front - the eliminated variables are the slowest
//    axes of F, so consecutive lanes read consecutive addresses, 512 B per wave instruction, cx (<= 16)
//    independent loads in flight per lane - and produces the whole fiber over the new (CPT-only) axes
//    in registers,  out[r, n] = sum_x F[r, x] * T[n, x, ctrl(r)].  With two variables eliminated per pass
//    (cx = 16, NC = 16: 256 FMAs per 256 bytes moved, still < 20 % of the fp64 vector rate) the frontier is
//    read and written once per *pair* of eliminations.  NC = 4 fibers are stored as two 16-byte vectors per
//    lane; NC = 16 fibers (128 B per lane) are transposed through a wave-private LDS buffer so that every
//    store instruction writes full 64-byte segments (direct 128-byte-strided stores reach only 3.8 TB/s
//    against 5.2 TB/s transposed - tools/ubench/stream_variants.hip).
//  * generic_tile_kernel<NIN> - big steps of any other shape, one output cell per lane-iteration.
//  * seg_kernel - a run of small steps of one request (start / end of a program, the final normalised
//    product) executed back to back by one workgroup; only `__syncthreads()` between steps (same CU).
//
// Index math: every table is laid out with the longest-living variable fastest, the iteration space
// is split into a lane-varying block (lo) and a wave-uniform block (hi).  Lane offsets are decoded
// once per tile; hi offsets are decoded by the lanes in parallel into LDS and broadcast from LDS in the
// streaming loop, which therefore contains only loads, fp64 FMAs and stores.
#pragma once
#include <hip/hip_runtime.h>

#include "planner.h"

namespace mibn {

constexpr int kWG = 256;
constexpr int kXRow = 20;  // dwords per lane row of the transpose buffer: 8 doubles + 16 B pad (conflict-free b128 writes)

struct LevelArgs {
    const uint32_t *prog;       // step programs of the chunk
    const uint64_t *prog_off;   // word offset of request i's program
    const uint64_t *arena_off;  // offset (doubles) of request i's private arena
    const double *pool;         // CPT tables (constants pool)
    double *arena;              // scratch
    double *results;            // dense posteriors of the chunk
    const Item *items;          // work items of this launch
    int n_items;
};

__device__ __forceinline__ const double *table_ptr(uint32_t lo, uint32_t hi, const double *pool, const double *slot) {
    const uint64_t o = (uint64_t)lo | ((uint64_t)hi << 32);
    return (o & kConstFlag) ? pool + (o & ~kConstFlag) : slot + o;
}

// ---------------------------------------------------------------------------------------- GENERIC
template <int NIN, int MAXC, int CX>
__device__ __forceinline__ void generic_body(const uint32_t *sw, int (*sh_hoff)[kTileMax], const double *__restrict__ pool,
                                             double *__restrict__ slot, double *__restrict__ results, const int tid,
                                             const int h_begin, const int h_end) {
    const uint32_t w0 = sw[0];
    const int na = (w0 >> 16) & 0xff;
    const int nlo = (w0 >> 24) & 0xff;
    const bool fin = (sw[1] >> 16) & kFlagFinal;
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

    for (int h0 = h_begin; h0 < h_end; h0 += kTileMax) {
        {
            const int h = h0 + tid;
            if (tid < kTileMax && h < h_end) {
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
        const int nh = min(kTileMax, h_end - h0);
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
__device__ __forceinline__ void generic_cx(const uint32_t *sw, int (*sh_hoff)[kTileMax], const double *pool, double *slot,
                                           double *results, int tid, int h_begin, int h_end) {
    const int cx = (int)(sw[1] & 0xffff);
    if (cx == 4) generic_body<NIN, MAXC, 4>(sw, sh_hoff, pool, slot, results, tid, h_begin, h_end);
    else if (cx == 2) generic_body<NIN, MAXC, 2>(sw, sh_hoff, pool, slot, results, tid, h_begin, h_end);
    else generic_body<NIN, MAXC, 0>(sw, sh_hoff, pool, slot, results, tid, h_begin, h_end);
}

// ------------------------------------------------------------------------------------------ FIBER
// NBIG big inputs.  CXC: 0 = cx 4 (one variable), 1 = cx 16 (two 4-state variables), 2 = runtime (cx <= 16).
// NCC: 0 = NC 1, 1 = NC 4 contiguous, 2 = NC 16 contiguous, 3 = runtime (NC <= 16, scattered stores).
// Tile = hi iterations [h_begin, h_end), at most kTileMax of them.
template <int NBIG, int CXC, int NCC>
__device__ __forceinline__ void fiber_body(const uint32_t *sw, double *__restrict__ shT, int (*sh_hoff)[kTileMax],
                                           uint32_t *__restrict__ shX, const double *__restrict__ pool,
                                           double *__restrict__ slot, const int tid, const int h_begin, const int h_end) {
    constexpr int CX = CXC == 0 ? 4 : (CXC == 1 ? 16 : 0);
    constexpr int NCT = NCC == 0 ? 1 : (NCC == 1 ? 4 : (NCC == 2 ? 16 : 0));
    const uint32_t w0 = sw[0];
    const int na = (w0 >> 16) & 0xff;
    const int nlo = (w0 >> 24) & 0xff;
    const int cx = CX ? CX : (int)(sw[1] & 0xffff);
    const int c1 = CX ? 4 : (int)(sw[8] >> 16);
    const int lo_cells = (int)sw[2];
    double *__restrict__ outp = slot + ((uint64_t)sw[4] | ((uint64_t)sw[5] << 32));
    const int ns = (sw[7] >> 4) & 0xf, nN = (sw[7] >> 8) & 0xf, nctrl = (sw[7] >> 12) & 0xf;
    const int NC = NCT ? NCT : (int)(sw[7] >> 16);
    const int T = (int)(sw[8] & 0xffff);
    const int nT = nN + nctrl;

    const uint32_t *q = sw + kHdrWords;
    const double *__restrict__ big[NBIG];
    int bxs1[NBIG], bxs2[NBIG];
#pragma unroll
    for (int b = 0; b < NBIG; ++b) {
        big[b] = table_ptr(q[0], q[1], pool, slot);
        bxs1[b] = (int)q[2];
        bxs2[b] = (int)q[3];
        q += 4;
    }
    const uint32_t *smalls = q;  // ns records of (4 + nT) words
    q += ns * (4 + nT);
    const uint32_t *tcard = q;
    q += nT;
    const uint32_t *nout = q;
    q += NC;
    const uint32_t *rax = q;  // (card, ostride, tstride) per R axis
    q += 3 * na;
    const int *bst = (const int *)q;  // bst[b * na + a]

    // T[n + NC*(x + cx*ctrl)] = product of the small inputs (the CPT slices), once per tile
    for (int t = tid; t < T; t += kWG) {
        int r = t;
        const int qn = r / NC;
        const int rn = r - qn * NC;
        r = qn;
        const int qx = r / cx;
        const int x = r - qx * cx;
        r = qx;
        const int x2 = x / c1, x1 = x - x2 * c1;
        double v = 1.0;
        for (int j = 0; j < ns; ++j) {
            const uint32_t *rec = smalls + j * (4 + nT);
            const int *sts = (const int *)(rec + 4);
            int off = x1 * (int)rec[2] + x2 * (int)rec[3];
            int a = rn, c = r;
            for (int k = 0; k < nN; ++k) { const int cd = (int)tcard[k]; const int qq = a / cd; off += (a - qq * cd) * sts[k]; a = qq; }
            for (int k = nN; k < nT; ++k) { const int cd = (int)tcard[k]; const int qq = c / cd; off += (c - qq * cd) * sts[k]; c = qq; }
            v *= table_ptr(rec[0], rec[1], pool, slot)[off];
        }
        shT[t] = v;
    }

    // lane offsets (one R cell per lane)
    const bool active = tid < lo_cells;
    int lo_o = 0, lo_t = 0, lo_b[NBIG];
#pragma unroll
    for (int b = 0; b < NBIG; ++b) lo_b[b] = 0;
    if (active) {
        int r = tid;
        for (int a = 0; a < nlo; ++a) {
            const int cd = (int)rax[3 * a];
            const int qq = r / cd;
            const int d = r - qq * cd;
            r = qq;
            lo_o += d * (int)rax[3 * a + 1];
            lo_t += d * (int)rax[3 * a + 2];
#pragma unroll
            for (int b = 0; b < NBIG; ++b) lo_b[b] += d * bst[b * na + a];
        }
    }
    // wave-uniform offsets of this tile's iterations
    const int nh = h_end - h_begin;
    if (tid < nh) {
        int ao = 0, at = 0, ab[NBIG];
#pragma unroll
        for (int b = 0; b < NBIG; ++b) ab[b] = 0;
        int r = h_begin + tid;
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
    __syncthreads();  // T and the offsets are ready

    const int wave = tid >> 6, lane = tid & 63;
    for (int hh = 0; hh < nh; ++hh) {
        const int ho = sh_hoff[0][hh], ht = sh_hoff[1][hh];
        int hb[NBIG];
#pragma unroll
        for (int b = 0; b < NBIG; ++b) hb[b] = sh_hoff[2 + b][hh];
        const double *__restrict__ Tp = shT + (ht + lo_t);
        double *__restrict__ o = outp + (ho + lo_o);
        if (CX) {
            // all loads of this iteration (NBIG tables x CX values) are issued before the first use
            double f[CX ? CX : 1];
#pragma unroll
            for (int x = 0; x < (CX ? CX : 1); ++x) {
                double p = active ? big[0][hb[0] + lo_b[0] + (x & 3) * bxs1[0] + (x >> 2) * bxs2[0]] : 0.0;
#pragma unroll
                for (int b = 1; b < NBIG; ++b) p *= active ? big[b][hb[b] + lo_b[b] + (x & 3) * bxs1[b] + (x >> 2) * bxs2[b]] : 0.0;
                f[x] = p;
            }
            if (NCT == 1) {
                double s = 0.0;
#pragma unroll
                for (int x = 0; x < (CX ? CX : 1); ++x) s += f[x] * Tp[x];
                if (active) o[0] = s;
            } else if (NCT == 4) {
                double acc[4];
#pragma unroll
                for (int n = 0; n < 4; ++n) {
                    double s = 0.0;
#pragma unroll
                    for (int x = 0; x < (CX ? CX : 1); ++x) s += f[x] * Tp[x * 4 + n];
                    acc[n] = s;
                }
                if (active) {
                    *reinterpret_cast<double2 *>(o) = make_double2(acc[0], acc[1]);
                    *reinterpret_cast<double2 *>(o + 2) = make_double2(acc[2], acc[3]);
                }
            } else if (NCT == 16) {
                // the wave's 64 fibers are one contiguous 64*16-cell region: two rounds of 8 cells per lane through
                // the wave-private transpose buffer, every store instruction then writes 64-byte segments
                uint32_t *__restrict__ X = shX + wave * (64 * kXRow);
                double *__restrict__ Ob = outp + ho + (wave * 64) * 16;
#pragma unroll
                for (int half = 0; half < 2; ++half) {
                    double acc[8];
#pragma unroll
                    for (int n = 0; n < 8; ++n) {
                        double s = 0.0;
#pragma unroll
                        for (int x = 0; x < (CX ? CX : 1); ++x) s += f[x] * Tp[x * 16 + half * 8 + n];
                        acc[n] = s;
                    }
#pragma unroll
                    for (int n = 0; n < 8; n += 2)
                        *reinterpret_cast<double2 *>(X + lane * kXRow + 2 * n) = make_double2(acc[n], acc[n + 1]);
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const int g = k * 64 + lane;  // 16-byte chunk of this round: owner lane g / 4, piece g % 4
                        const int owner = g >> 2, piece = g & 3;
                        const double2 v = *reinterpret_cast<const double2 *>(X + owner * kXRow + 4 * piece);
                        if (wave * 64 + owner < lo_cells)
                            *reinterpret_cast<double2 *>(Ob + owner * 16 + half * 8 + 2 * piece) = v;
                    }
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                }
            } else {
                if (active) {
                    for (int n0 = 0; n0 < NC; n0 += 4) {
                        double acc[4];
#pragma unroll
                        for (int n = 0; n < 4; ++n) {
                            double s = 0.0;
                            if (n0 + n < NC) {
#pragma unroll
                                for (int x = 0; x < (CX ? CX : 1); ++x) s += f[x] * Tp[x * NC + n0 + n];
                            }
                            acc[n] = s;
                        }
#pragma unroll
                        for (int n = 0; n < 4; ++n)
                            if (n0 + n < NC) o[nout[n0 + n]] = acc[n];
                    }
                }
            }
        } else {
            if (active) {
                for (int n0 = 0; n0 < NC; n0 += 4) {
                    double acc[4];
#pragma unroll
                    for (int n = 0; n < 4; ++n) acc[n] = 0.0;
                    for (int x = 0; x < cx; ++x) {
                        const int x2 = x / c1, x1 = x - x2 * c1;
                        double p = big[0][hb[0] + lo_b[0] + x1 * bxs1[0] + x2 * bxs2[0]];
#pragma unroll
                        for (int b = 1; b < NBIG; ++b) p *= big[b][hb[b] + lo_b[b] + x1 * bxs1[b] + x2 * bxs2[b]];
#pragma unroll
                        for (int n = 0; n < 4; ++n)
                            if (n0 + n < NC) acc[n] += p * Tp[x * NC + n0 + n];
                    }
#pragma unroll
                    for (int n = 0; n < 4; ++n)
                        if (n0 + n < NC) o[nout[n0 + n]] = acc[n];
                }
            }
        }
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

// ---- out-of-line wrappers for the segment interpreter: each specialisation keeps its own register allocation
template <int NIN, int MAXC>
__device__ __noinline__ void generic_call(const uint32_t *sw, int (*sh_hoff)[kTileMax], const double *pool, double *slot,
                                          double *results, int tid) {
    generic_cx<NIN, MAXC>(sw, sh_hoff, pool, slot, results, tid, 0, (int)sw[3]);
}

// SEGMENT: a run of small GENERIC steps of one request, executed back to back by one workgroup
__global__ __launch_bounds__(kWG) void seg_kernel(const LevelArgs A) {
    __shared__ uint32_t sh_step[kMaxStepWords];
    __shared__ int sh_hoff[kMaxIn][kTileMax];
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
        const int n_in = (sh_step[0] >> 8) & 0xff;
        switch (n_in) {
            case 1: generic_call<1, 2>(sh_step, sh_hoff, A.pool, slot, A.results, tid); break;
            case 2: generic_call<2, 2>(sh_step, sh_hoff, A.pool, slot, A.results, tid); break;
            case 3: generic_call<3, 2>(sh_step, sh_hoff, A.pool, slot, A.results, tid); break;
            case 4: generic_call<4, 1>(sh_step, sh_hoff, A.pool, slot, A.results, tid); break;
            case 5: generic_call<5, 1>(sh_step, sh_hoff, A.pool, slot, A.results, tid); break;
            default: generic_call<6, 1>(sh_step, sh_hoff, A.pool, slot, A.results, tid); break;
        }
        if ((sh_step[1] >> 16) & kFlagFinal) {
            const uint64_t out_off = (uint64_t)sh_step[4] | ((uint64_t)sh_step[5] << 32);
            const int n = (int)(sh_step[2] * sh_step[3]);
            normalise(A.results + out_off, n, sh_red, tid);
        }
        p += words;
    }
}

// Which tiled step does this workgroup work on?  items[k].b = first tile of step k within the launch (ascending).
__device__ __forceinline__ int find_item(const Item *__restrict__ items, int n, uint32_t wg) {
    int lo = 0, hi = n - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (items[mid].b <= wg) lo = mid;
        else hi = mid - 1;
    }
    return lo;
}

// TILE of a big FIBER step
template <int NBIG, int CXC, int NCC>
__global__ __launch_bounds__(kWG) void fiber_tile_kernel(const LevelArgs A) {
    __shared__ __attribute__((aligned(16))) double shT[kMaxT];
    __shared__ __attribute__((aligned(16))) uint32_t shX[NCC == 2 ? 4 * 64 * kXRow : 4];
    __shared__ uint32_t sh_step[kMaxStepWords];
    __shared__ int sh_hoff[2 + NBIG][kTileMax];
    const int tid = threadIdx.x;
    const Item it = A.items[find_item(A.items, A.n_items, blockIdx.x)];
    double *slot = A.arena + A.arena_off[it.req];
    const uint32_t *p = A.prog + A.prog_off[it.req] + it.rel_off;
    const int words = (int)p[6];
    for (int i = tid; i < words; i += kWG) sh_step[i] = p[i];
    __syncthreads();
    const int h0 = (int)((blockIdx.x - it.b) * it.a);
    const int h1 = min((int)sh_step[3], h0 + (int)it.a);
    fiber_body<NBIG, CXC, NCC>(sh_step, shT, sh_hoff, shX, A.pool, slot, tid, h0, h1);
}

// TILE of a big GENERIC step
template <int NIN>
__global__ __launch_bounds__(kWG) void generic_tile_kernel(const LevelArgs A) {
    __shared__ uint32_t sh_step[kMaxStepWords];
    __shared__ int sh_hoff[NIN][kTileMax];
    const int tid = threadIdx.x;
    const Item it = A.items[find_item(A.items, A.n_items, blockIdx.x)];
    double *slot = A.arena + A.arena_off[it.req];
    const uint32_t *p = A.prog + A.prog_off[it.req] + it.rel_off;
    const int words = (int)p[6];
    for (int i = tid; i < words; i += kWG) sh_step[i] = p[i];
    __syncthreads();
    const int h0 = (int)((blockIdx.x - it.b) * it.a);
    const int h1 = min((int)sh_step[3], h0 + (int)it.a);
    generic_cx<NIN, (NIN <= 3 ? 2 : 1)>(sh_step, sh_hoff, A.pool, slot, A.results, tid, h0, h1);
}

}  // namespace mibn
