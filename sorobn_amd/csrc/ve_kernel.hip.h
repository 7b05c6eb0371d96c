// CDNA4 (gfx950) variable-elimination kernel.
//
// A step is the fused replacement of `pointwise_mul(...)` + `.cdt.sum_out(x)`
// (sorobn/bayes_net.py:780-785, 233-256, 100-103):
//
//        psi[out] = sum_{x < cx}  prod_{j < n_in}  phi_j[ base_j + idx_j(out) + off_j(x) ]
//
// over dense fp64 tables; the product table of the reference (up to 4^11 rows on the 10x10 grid) is
// never materialised.  Every request owns a private arena in HBM for its intermediates.
//
// Execution is level-synchronous (schedule: planner.h): ONE launch of `ve_level_kernel` per level, 256-lane
// workgroups (4 wave64), one work item per workgroup - a tile of a big step or a segment of small steps; the only
// synchronisation between dependent steps is the launch boundary.  Three families of work:
//  * FIBER tiles <NBIG, CXC, NCC> - the streaming form, > 95 % of the bytes on the 10x10 grid.  A tile = a few
//    wave-uniform iterations x <= 256 lane cells of one big step (~512 KiB of traffic).  The inputs are one or two
//    big tables (the elimination frontier, MBs, streamed from HBM) and a few CPT slices (<= 8 KiB) whose product
//    is tabulated once per tile in LDS (T, <= 16 KiB).  A lane owns one cell r of the big tables' shared axes: it
//    loads the cx values F[r, x] - the eliminated variables are the slowest axes of F, so consecutive lanes read
//    consecutive addresses, 512 B per wave instruction - and produces the whole fiber over the new (CPT-only)
//    axes in registers,  out[r, n] = sum_x F[r, x] * T[n, x, ctrl(r)].  With two variables eliminated per pass
//    (cx = 16, NC = 16: 256 FMAs per 256 bytes moved, < 20 % of the fp64 vector rate) the frontier is read and
//    written once per *pair* of eliminations.  The loop is software-pipelined: the 16 loads of the next trip
//    (one cx = 16 iteration or four cx = 4 iterations) are in flight while the current one is reduced and
//    stored; their addresses are a per-trip scalar base plus one 32-bit lane offset.  NC = 4 fibers are stored
//    as two 16-byte vectors per lane; NC = 16 fibers (128 B per lane) are transposed through a wave-private LDS
//    buffer so that every store instruction writes full 64-byte segments (direct 128-byte-strided stores
//    reach only 3.8 TB/s against 5.2 TB/s transposed - tools/ubench/stream_variants.hip).
//    The NC = 16 shapes whose cells split into row blocks of 16 sharing one T slice run as fp64 MFMA products
//    instead (fiber_mfma_call: [16 cells x cx] x [cx x 16] per block; outer_mfma_call: the B operand from a second big
//    table; chain_mfma_call: a third variable summed out of the accumulators in registers) - the bulk of the bytes.
//  * GENERIC tiles <NIN> - big steps of any other shape, one output cell per lane-iteration.
//  * segments - a run of small steps of one request (start / end of a program, the final normalised
//    product) executed back to back by one workgroup; only `__syncthreads()` between steps (same CU).
//
// Index math: every table is laid out with the longest-living variable fastest, the iteration space
// is split into a lane-varying block (lo) and a wave-uniform block (hi).  Lane offsets are decoded
// once per tile; hi offsets are decoded by the lanes in parallel into LDS and read back as scalars in the
// streaming loop, which therefore contains only loads, fp64 FMAs and stores.
#pragma once
#include <hip/hip_runtime.h>

#include "planner.h"

namespace mibn {

constexpr int kWG = 256;
#ifndef MIBN_MIN_WAVES
#define MIBN_MIN_WAVES 3   // waves per SIMD the level kernel is compiled for (VGPR budget 512 / MIBN_MIN_WAVES)
#endif
#ifndef MIBN_PIPELINE
#define MIBN_PIPELINE 1    // software-pipelined FIBER loop (next trip's loads in flight during the reduction)
#endif
constexpr int kXRow = 20;  // dwords per lane row of the transpose buffer: 8 doubles + 16 B pad (conflict-free b128 writes)

struct LevelArgs {
    const uint32_t *prog;       // step programs of the chunk
    const uint64_t *prog_off;   // word offset of request i's program
    const uint64_t *arena_off;  // offset (doubles) of request i's private arena
    const double *pool;         // CPT tables (constants pool)
    double *arena;              // scratch
    double *results;            // dense posteriors of the chunk
    const Item *items;          // work items of this chunk
    const uint32_t *wg_item;    // item of every workgroup of this level
    uint32_t wg_base;           // level-relative index of this launch's first workgroup
};

__device__ __forceinline__ const double *table_ptr(uint32_t lo, uint32_t hi, const double *pool, const double *slot) {
    const uint64_t o = (uint64_t)lo | ((uint64_t)hi << 32);
    return (o & kConstFlag) ? pool + (o & ~kConstFlag) : slot + o;
}

__device__ __forceinline__ int uni(int v) { return __builtin_amdgcn_readfirstlane(v); }

// ---------------------------------------------------------------------------------------- GENERIC
// LANES = kWG: a tile of a big step, the whole workgroup.  LANES = 64: a step of a SEGMENT - one wave runs a request's chain
// of tiny steps on its own (four segments per workgroup, planner.h kSegPerWg): no workgroup barrier, the hand-over between
// dependent steps is the wave's own program order plus a fence.
template <int LANES>
__device__ __forceinline__ void lanes_sync() {
    if constexpr (LANES == kWG) {
        __syncthreads();
    } else {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");  // (this wave's stores to the arena / LDS before its next reads)
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    }
}

template <int NIN, int MAXC, int CX, int LANES>
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

    // the lane-varying block holds up to MAXC * kWG cells: a workgroup covers it in one pass of MAXC cells per lane, a single
    // wave (a segment's step) in up to four (static register indices either way: one pass's offsets live in registers)
    const int passes = LANES == kWG ? 1 : (lo_cells + MAXC * LANES - 1) / (MAXC * LANES);
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
        lanes_sync<LANES>();
        const int nh = min(kTileMax, h_end - h0);
        for (int pass = 0; pass < passes; ++pass) {
            const int l0 = tid + pass * (MAXC * LANES);
            int lo_off[NIN][MAXC];
#pragma unroll
            for (int c = 0; c < MAXC; ++c) {
#pragma unroll
                for (int j = 0; j < NIN; ++j) lo_off[j][c] = 0;
                const int l = l0 + c * LANES;
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
            for (int hh = 0; hh < nh; ++hh) {
                int ho[NIN];
#pragma unroll
                for (int j = 0; j < NIN; ++j) ho[j] = sh_hoff[j][hh];
                const size_t orow = (size_t)(h0 + hh) * (size_t)lo_cells;
#pragma unroll
                for (int c = 0; c < MAXC; ++c) {
                    const int l = l0 + c * LANES;
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
        }
        lanes_sync<LANES>();
    }
}

template <int NIN, int MAXC, int LANES>
__device__ __forceinline__ void generic_call(const uint32_t *sw, int (*sh_hoff)[kTileMax], const double *pool, double *slot,
                                          double *results, int tid, int h_begin, int h_end) {
    const int cx = (int)(sw[1] & 0xffff);
    if (cx == 4) generic_body<NIN, MAXC, 4, LANES>(sw, sh_hoff, pool, slot, results, tid, h_begin, h_end);
    else if (cx == 2) generic_body<NIN, MAXC, 2, LANES>(sw, sh_hoff, pool, slot, results, tid, h_begin, h_end);
    else generic_body<NIN, MAXC, 0, LANES>(sw, sh_hoff, pool, slot, results, tid, h_begin, h_end);
}

// (cells per lane: the lane-varying block holds up to kLoMax = 512 cells for <= 3 inputs, kLoTarget = 256 beyond)
template <int LANES>
__device__ __forceinline__ void generic_dispatch(int n_in, const uint32_t *sw, int (*sh_hoff)[kTileMax], const double *pool,
                                                 double *slot, double *results, int tid, int h_begin, int h_end) {
    constexpr int M3 = LANES == kWG ? kLoMax / kWG : 2, M6 = LANES == kWG ? kLoTarget / kWG : 1;  // cells per lane and pass
    switch (n_in) {
        case 1: generic_call<1, M3, LANES>(sw, sh_hoff, pool, slot, results, tid, h_begin, h_end); break;
        case 2: generic_call<2, M3, LANES>(sw, sh_hoff, pool, slot, results, tid, h_begin, h_end); break;
        case 3: generic_call<3, M3, LANES>(sw, sh_hoff, pool, slot, results, tid, h_begin, h_end); break;
        case 4: generic_call<4, M6, LANES>(sw, sh_hoff, pool, slot, results, tid, h_begin, h_end); break;
        case 5: generic_call<5, M6, LANES>(sw, sh_hoff, pool, slot, results, tid, h_begin, h_end); break;
        default: generic_call<6, M6, LANES>(sw, sh_hoff, pool, slot, results, tid, h_begin, h_end); break;
    }
}

// ------------------------------------------------------------------------------------------ FIBER
// Decoded layout of a FIBER step descriptor (LDS copy).
struct FiberDesc {
    int na, nlo, cx, c1, lo_cells, ns, nN, nctrl, NC, T, nT, nb;
    const uint32_t *bigs, *smalls, *tcard, *nout, *nB, *chain, *rax;
    const int *bst;
};

__device__ __forceinline__ FiberDesc fiber_desc(const uint32_t *sw) {
    FiberDesc d;
    d.na = (sw[0] >> 16) & 0xff;
    d.nlo = (sw[0] >> 24) & 0xff;
    d.cx = (int)(sw[1] & 0xffff);
    d.c1 = (int)(sw[8] >> 16);
    d.lo_cells = (int)sw[2];
    d.nb = sw[7] & 0xf;
    d.ns = (sw[7] >> 4) & 0xf;
    d.nN = (sw[7] >> 8) & 0xf;
    d.nctrl = (sw[7] >> 12) & 0xf;
    d.NC = (int)(sw[7] >> 16);
    d.T = (int)(sw[8] & 0xffff);
    d.nT = d.nN + d.nctrl;
    const uint32_t *q = sw + kHdrWords;
    d.bigs = q;
    q += 4 * d.nb;
    d.smalls = q;  // ns records of (4 + nT) words
    q += d.ns * (4 + d.nT);
    d.tcard = q;
    q += d.nT;
    d.nout = q;
    q += d.NC;
    d.nB = q;  // OUTER only: offset of N-combination n in the second big input
    if ((sw[1] >> 16) & kFlagOuter) q += d.NC;
    d.chain = q;  // CHAIN only: n_small3, n_dims3 | n12dep << 8, tcard3[], then (off lo, off hi, stride[n_dims3]) per input
    if ((sw[1] >> 16) & kFlagChain) { const int nd3 = (int)(q[1] & 0xff); q += 2 + nd3 + (int)q[0] * (2 + nd3); }
    d.rax = q;  // (card, ostride, tstride) per R axis
    q += 3 * d.na;
    d.bst = (const int *)q;  // bst[b * na + a]
    return d;
}

// Tile prologue shared by every FIBER specialisation:
//  * T[n + NC*(x + cx*ctrl)] = product of the small inputs (the CPT slices)
//  * wave-uniform offsets of the tile's iterations -> sh_hoff[0] (output), [1] (T), [2 + b] (big input b)
//  * this lane's offsets (one R cell per lane) -> lane_off[0] (output), [1] (T), [2 + b]
struct LaneOff { int o, t, b0, b1; };  // returned in registers (a pointer to the caller's array would live in scratch)
__device__ __noinline__ LaneOff fiber_prologue(const uint32_t *sw, double *__restrict__ shT, int (*sh_hoff)[kTileMax],
                                               const double *__restrict__ pool, const double *__restrict__ slot, const int tid,
                                               const int h_begin, const int h_end) {
    const FiberDesc d = fiber_desc(sw);
    for (int t0 = 0; t0 < d.T; t0 += 2 * kWG) {  // two entries per lane and trip: their loads overlap
        double v[2] = {1.0, 1.0};
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const int t = t0 + e * kWG + tid;
            if (t < d.T) {
                int r = t;
                const int qn = r / d.NC;
                const int rn = r - qn * d.NC;
                r = qn;
                const int qx = r / d.cx;
                const int x = r - qx * d.cx;
                r = qx;
                const int x2 = x / d.c1, x1 = x - x2 * d.c1;
                for (int j = 0; j < d.ns; ++j) {
                    const uint32_t *rec = d.smalls + j * (4 + d.nT);
                    const int *sts = (const int *)(rec + 4);
                    int off = x1 * (int)rec[2] + x2 * (int)rec[3];
                    int a = rn, c = r;
                    for (int k = 0; k < d.nN; ++k) { const int cd = (int)d.tcard[k]; const int qq = a / cd; off += (a - qq * cd) * sts[k]; a = qq; }
                    for (int k = d.nN; k < d.nT; ++k) { const int cd = (int)d.tcard[k]; const int qq = c / cd; off += (c - qq * cd) * sts[k]; c = qq; }
                    v[e] *= table_ptr(rec[0], rec[1], pool, slot)[off];
                }
            }
        }
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const int t = t0 + e * kWG + tid;
            if (t < d.T) shT[t] = v[e];
        }
    }
    int lo[4] = {0, 0, 0, 0};
    if (tid < d.lo_cells) {
        int r = tid;
        for (int a = 0; a < d.nlo; ++a) {
            const int cd = (int)d.rax[3 * a];
            const int qq = r / cd;
            const int dg = r - qq * cd;
            r = qq;
            lo[0] += dg * (int)d.rax[3 * a + 1];
            lo[1] += dg * (int)d.rax[3 * a + 2];
            for (int b = 0; b < d.nb; ++b) lo[2 + b] += dg * d.bst[b * d.na + a];
        }
    }
    const int nh = h_end - h_begin;
    if (tid < nh) {
        int hi[4] = {0, 0, 0, 0};
        int r = h_begin + tid;
        for (int a = d.nlo; a < d.na; ++a) {
            const int cd = (int)d.rax[3 * a];
            const int qq = r / cd;
            const int dg = r - qq * cd;
            r = qq;
            hi[0] += dg * (int)d.rax[3 * a + 1];
            hi[1] += dg * (int)d.rax[3 * a + 2];
            for (int b = 0; b < d.nb; ++b) hi[2 + b] += dg * d.bst[b * d.na + a];
        }
        sh_hoff[0][tid] = hi[0];
        sh_hoff[1][tid] = hi[1];
        sh_hoff[2][tid] = hi[2];
        sh_hoff[3][tid] = hi[3];
    }
    __syncthreads();  // T and the offsets are ready
    return LaneOff{lo[0], lo[1], lo[2], lo[3]};
}

// Reduce one loaded fiber f[0..CX) against T and store the NC outputs.  NCT = compile-time NC (1, 4, 16) or 0.
template <int CX, int NCT>
__device__ __forceinline__ void fiber_reduce_store(const double (&f)[CX], const double *__restrict__ Tp, double *__restrict__ o,
                                                   double *__restrict__ Ob, uint32_t *__restrict__ X, const uint32_t *nout,
                                                   const int NC, const bool active, const int lane, const int cells_in_wave) {
    if (NCT == 1) {
        double s = 0.0;
#pragma unroll
        for (int x = 0; x < CX; ++x) s += f[x] * Tp[x];
        if (active) o[0] = s;
    } else if (NCT == 4) {
        double acc[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int x = 0; x < CX; ++x) {
#pragma unroll
            for (int n = 0; n < 4; ++n) acc[n] += f[x] * Tp[x * 4 + n];
        }
        if (active) {
            *reinterpret_cast<double2 *>(o) = make_double2(acc[0], acc[1]);
            *reinterpret_cast<double2 *>(o + 2) = make_double2(acc[2], acc[3]);
        }
    } else if (NCT == 16) {
        // the wave's 64 fibers are one contiguous 64*16-cell region (Ob): two rounds of 8 cells per lane through the
        // wave-private transpose buffer X, every store instruction then writes 64-byte segments
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            double acc[8] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int x = 0; x < CX; ++x) {
#pragma unroll
                for (int n = 0; n < 8; ++n) acc[n] += f[x] * Tp[x * 16 + half * 8 + n];
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
                if (owner < cells_in_wave) *reinterpret_cast<double2 *>(Ob + owner * 16 + half * 8 + 2 * piece) = v;
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
                        for (int x = 0; x < CX; ++x) s += f[x] * Tp[x * NC + n0 + n];
                    }
                    acc[n] = s;
                }
#pragma unroll
                for (int n = 0; n < 4; ++n)
                    if (n0 + n < NC) o[nout[n0 + n]] = acc[n];
            }
        }
    }
}

// NBIG big inputs.  CXC: 0 = cx 4 (one variable), 1 = cx 16 (two 4-state variables), 2 = runtime (cx <= 16).
// NCC: 0 = NC 1, 1 = NC 4 contiguous, 2 = NC 16 contiguous, 3 = runtime (NC <= 16, scattered stores).
// Tile = hi iterations [h_begin, h_end), at most kTileMax of them.
template <int NBIG, int CXC, int NCC>
__device__ __forceinline__ void fiber_call(const uint32_t *sw, double *__restrict__ shT, int (*sh_hoff)[kTileMax],
                                        uint32_t *__restrict__ shX, const double *__restrict__ pool, double *__restrict__ slot,
                                        const int tid, const int h_begin, const int h_end) {
    constexpr int CX = CXC == 0 ? 4 : (CXC == 1 ? 16 : 0);
    constexpr int NCT = NCC == 0 ? 1 : (NCC == 1 ? 4 : (NCC == 2 ? 16 : 0));
    constexpr int U = CX ? 16 / CX : 1;  // iterations per trip: 16 loads in flight per lane and big input
    const LaneOff lane_off_ = fiber_prologue(sw, shT, sh_hoff, pool, slot, tid, h_begin, h_end);
    const int lane_off[4] = {lane_off_.o, lane_off_.t, lane_off_.b0, lane_off_.b1};
    const FiberDesc d = fiber_desc(sw);
    const int NC = NCT ? NCT : d.NC;
    double *__restrict__ outp = slot + ((uint64_t)sw[4] | ((uint64_t)sw[5] << 32));
    const double *__restrict__ big[NBIG];
    int bxs1[NBIG], bxs2[NBIG];
#pragma unroll
    for (int b = 0; b < NBIG; ++b) {
        big[b] = table_ptr(d.bigs[4 * b], d.bigs[4 * b + 1], pool, slot);
        bxs1[b] = (int)d.bigs[4 * b + 2];
        bxs2[b] = (int)d.bigs[4 * b + 3];
    }
    const bool active = tid < d.lo_cells;
    const int lo_o = lane_off[0], lo_t = lane_off[1];
    const int nh = h_end - h_begin;
    const int wave = tid >> 6, lane = tid & 63;
    uint32_t *__restrict__ X = shX + wave * (64 * kXRow);
    const int cells_in_wave = d.lo_cells - wave * 64;

    if (CX) {
        // issue the loads of iterations [hh, hh + U): the address is a scalar base (table + wave-uniform offset +
        // x offset) plus this lane's 32-bit offset
        auto issue = [&](const int hh, double (&dst)[U][CX ? CX : 1]) {
#pragma unroll
            for (int u = 0; u < U; ++u) {
                if (hh + u < nh) {
                    // A table of a two-table step often lacks one of the two eliminated variables (a sweep meeting
                    // another one): 4 distinct values then, not 16 - these steps are bound by L1 load issue, not by HBM.
                    // (inactive lanes carry offset 0: they load a valid cell and never store)
                    auto gather = [&](const int b, const int hrow, const uint32_t loff, const bool first) {
                        const int s1 = bxs1[b], s2 = bxs2[b];
                        const int hb = uni(sh_hoff[hrow][hh + u]);
                        if (NBIG > 1 && CX == 16 && s1 == 0) {
#pragma unroll
                            for (int x2 = 0; x2 < 4; ++x2) {
                                const double v = (big[b] + (hb + x2 * s2))[loff];
#pragma unroll
                                for (int x1 = 0; x1 < 4; ++x1) {
                                    double &t = dst[u][(x1 + 4 * x2) % (CX ? CX : 1)];
                                    t = first ? v : t * v;
                                }
                            }
                        } else if (NBIG > 1 && CX == 16 && s2 == 0) {
#pragma unroll
                            for (int x1 = 0; x1 < 4; ++x1) {
                                const double v = (big[b] + (hb + x1 * s1))[loff];
#pragma unroll
                                for (int x2 = 0; x2 < 4; ++x2) {
                                    double &t = dst[u][(x1 + 4 * x2) % (CX ? CX : 1)];
                                    t = first ? v : t * v;
                                }
                            }
                        } else {
#pragma unroll
                            for (int x = 0; x < (CX ? CX : 1); ++x) {
                                const double v = (big[b] + (hb + (x & 3) * s1 + (x >> 2) * s2))[loff];
                                dst[u][x] = first ? v : dst[u][x] * v;
                            }
                        }
                    };
                    gather(0, 2, (uint32_t)lane_off[2], true);
                    if (NBIG > 1) gather(NBIG - 1, 3, (uint32_t)lane_off[3], false);
                }
            }
        };
        auto finish = [&](const int hh, const double (&src)[U][CX ? CX : 1]) {
#pragma unroll
            for (int u = 0; u < U; ++u) {
                if (hh + u < nh) {
                    const int ho = uni(sh_hoff[0][hh + u]), ht = uni(sh_hoff[1][hh + u]);
                    fiber_reduce_store<(CX ? CX : 1), NCT>(src[u], shT + (ht + lo_t), outp + (ho + lo_o), outp + ho + (wave * 64) * 16, X,
                                                          d.nout, NC, active, lane, cells_in_wave);
                }
            }
        };
        if constexpr (NBIG == 1 && MIBN_PIPELINE) {
            // software pipeline: the next trip's loads are in flight while this trip is reduced and stored
            double fa[U][CX ? CX : 1], fb[U][CX ? CX : 1];
            issue(0, fa);
            for (int hh = 0; hh < nh; hh += 2 * U) {
                if (hh + U < nh) issue(hh + U, fb);
                finish(hh, fa);
                if (hh + U < nh) {
                    if (hh + 2 * U < nh) issue(hh + 2 * U, fa);
                    finish(hh + U, fb);
                }
            }
        } else if constexpr (NBIG == 2 && CX == 16) {
            // two tables: a table whose wave-uniform offset did not change since the previous iteration (the planner makes
            // the axes only the other table depends on run fastest) stays in registers - these steps are bound by L1 load
            // issue, a reused table saves 4-16 of their 8-32 loads per output cell
            // (one of the two stays - `keep` - the other is loaded into the product buffer every iteration: a third
            // 16-double array, both tables cached, made the register allocator spill)
            double keep[16];
            // the slow table: the one whose wave-uniform offset does not move between the first two iterations
            const bool slow_b = nh > 1 && uni(sh_hoff[2][0]) != uni(sh_hoff[2][1]) && uni(sh_hoff[3][0]) == uni(sh_hoff[3][1]);
            const int sb = slow_b ? NBIG - 1 : 0, fb = slow_b ? 0 : NBIG - 1;           // slow / fast table
            const int srow = slow_b ? 3 : 2, frow = slow_b ? 2 : 3;                       // their rows of sh_hoff
            const uint32_t sloff = (uint32_t)(slow_b ? lane_off[3] : lane_off[2]), floff = (uint32_t)(slow_b ? lane_off[2] : lane_off[3]);
            const double *__restrict__ sbig = big[sb], *__restrict__ fbig = big[fb];
            const int ss1 = bxs1[sb], ss2 = bxs2[sb], fs1 = bxs1[fb], fs2 = bxs2[fb];
            // a table of a two-table step often lacks one of the two eliminated variables (a sweep meeting another one):
            // 4 distinct values then, not 16
            auto fetch = [&](const double *__restrict__ tab, const int s1, const int s2, const int hb, const uint32_t loff, double (&dst)[16],
                             const bool multiply) {
                if (s1 == 0) {
#pragma unroll
                    for (int x2 = 0; x2 < 4; ++x2) {
                        const double v = (tab + (hb + x2 * s2))[loff];
#pragma unroll
                        for (int x1 = 0; x1 < 4; ++x1) dst[x1 + 4 * x2] = multiply ? v * keep[x1 + 4 * x2] : v;
                    }
                } else if (s2 == 0) {
#pragma unroll
                    for (int x1 = 0; x1 < 4; ++x1) {
                        const double v = (tab + (hb + x1 * s1))[loff];
#pragma unroll
                        for (int x2 = 0; x2 < 4; ++x2) dst[x1 + 4 * x2] = multiply ? v * keep[x1 + 4 * x2] : v;
                    }
                } else {
#pragma unroll
                    for (int x = 0; x < 16; ++x) {
                        const double v = (tab + (hb + (x & 3) * s1 + (x >> 2) * s2))[loff];
                        dst[x] = multiply ? v * keep[x] : v;
                    }
                }
            };
            int prev_s = 0;
            for (int hh = 0; hh < nh; ++hh) {
                const int hs = uni(sh_hoff[srow][hh]), hf = uni(sh_hoff[frow][hh]);
                if (hh == 0 || hs != prev_s) { fetch(sbig, ss1, ss2, hs, sloff, keep, false); prev_s = hs; }
                double f[1][CX ? CX : 1];
                fetch(fbig, fs1, fs2, hf, floff, f[0], true);
                finish(hh, f);
            }
        } else {
            double fa[U][CX ? CX : 1];
            for (int hh = 0; hh < nh; hh += U) {
                issue(hh, fa);
                finish(hh, fa);
            }
        }
    } else {
        const int cx = d.cx, c1 = d.c1;
        for (int hh = 0; hh < nh; ++hh) {
            if (active) {
                const double *__restrict__ Tp = shT + (sh_hoff[1][hh] + lo_t);
                double *__restrict__ o = outp + (sh_hoff[0][hh] + lo_o);
                for (int n0 = 0; n0 < NC; n0 += 4) {
                    double acc[4];
#pragma unroll
                    for (int n = 0; n < 4; ++n) acc[n] = 0.0;
                    for (int x = 0; x < cx; ++x) {
                        const int x2 = x / c1, x1 = x - x2 * c1;
                        double p = big[0][sh_hoff[2][hh] + lane_off[2] + x1 * bxs1[0] + x2 * bxs2[0]];
                        if (NBIG > 1) p *= big[NBIG - 1][sh_hoff[3][hh] + lane_off[3] + x1 * bxs1[NBIG - 1] + x2 * bxs2[NBIG - 1]];
#pragma unroll
                        for (int n = 0; n < 4; ++n)
                            if (n0 + n < NC) acc[n] += p * Tp[x * NC + n0 + n];
                    }
#pragma unroll
                    for (int n = 0; n < 4; ++n)
                        if (n0 + n < NC) o[d.nout[n0 + n]] = acc[n];
                }
            }
        }
    }
}

// fp64 MFMA classes: NC = 16 contiguous, cx = 4*KS (KS = 1: one 4-state variable, KS = 4: two), row stride s != 0.
// The 64 cells of a wave split into 4 row blocks of 16 cells sharing one T slice (cell of row i of block rb =
// (i % s) + s*rb + 4*s*(i / s); s = 16: consecutive cells); per block and iteration the step is the dense product
// out[16 cells x 16] = F[16 cells x cx] . T[cx x 16]:  v_mfma_f64_16x16x4_f64, 4 x values per instruction.  Lane l holds
//   A[row l&15][k l>>4] = F[x = 4*ks + (l>>4)][cell(rb, l&15)]        (s = 16: a load reads 4 x-slices x 128 B),
//   B[k l>>4][col l&15] = T[n = l&15][x = 4*ks + (l>>4)][ctrl(rb)]    (4*KS doubles per lane and iteration, from LDS),
//   D[row (l>>4) + 4*v][col l&15]  ->  out[cell(rb, 4*v + (l>>4))*16 + (l&15)]: s >= 4: 512 contiguous bytes per store.
// Against the VALU form this removes the 256 LDS operand reads per lane-iteration (the VALU form of this shape is
// LDS-bound: SQ_LDS_IDX_ACTIVE ~ 90 % of the kernel time) and the transposition of the 128-byte fibers.
typedef double v4d __attribute__((ext_vector_type(4)));

template <int NBIG, int KS, bool OCL = false>
__device__ __forceinline__ void fiber_mfma_call(const uint32_t *sw, double *__restrict__ shT, int (*sh_hoff)[kTileMax],
                                                uint32_t *__restrict__ shX, const double *__restrict__ pool,
                                                double *__restrict__ slot, const int tid, const int h_begin, const int h_end) {
    const LaneOff lane_off = fiber_prologue(sw, shT, sh_hoff, pool, slot, tid, h_begin, h_end);
    const FiberDesc d = fiber_desc(sw);
    double *__restrict__ outp = slot + ((uint64_t)sw[4] | ((uint64_t)sw[5] << 32));
    const double *__restrict__ big[NBIG];
    int bxs1[NBIG], bxs2[NBIG];
#pragma unroll
    for (int b = 0; b < NBIG; ++b) {
        big[b] = table_ptr(d.bigs[4 * b], d.bigs[4 * b + 1], pool, slot);
        bxs1[b] = (int)d.bigs[4 * b + 2];
        bxs2[b] = (int)d.bigs[4 * b + 3];
    }
    // every lane needs the offsets of the 4 cells (one per row block) it feeds: exchange them through LDS
    int *sh_cell = reinterpret_cast<int *>(shX);  // [4][kWG]: big input 0, big input 1, T, output
    sh_cell[tid] = lane_off.b0;
    sh_cell[kWG + tid] = lane_off.b1;
    sh_cell[2 * kWG + tid] = lane_off.t;
    sh_cell[3 * kWG + tid] = lane_off.o;
    __syncthreads();
    const int nh = h_end - h_begin;
    const int wave = tid >> 6, lane = tid & 63;
    const int lrow = lane & 15, lk = lane >> 4;
    const int rs = (int)((sw[1] >> kRowStrideShift) & 0xff);  // 1, 4 or 16
    uint32_t la[NBIG][4];
    int tb[4], oc[4][4];  // T offset of row block rb; output offset of accumulator element v of row block rb (-1: none)
    // OCL (ve_mfma_kernel: a register budget of 128): the sixteen output offsets of a lane are parked in LDS behind sh_cell, as
    // outer_mfma_call does - a lane reads back only what it wrote: no barrier
    int *sh_oc = sh_cell + 4 * kWG;  // [16][kWG]
#pragma unroll
    for (int rb = 0; rb < 4; ++rb) {
        const int c = wave * 64 + (lrow % rs) + rs * rb + 4 * rs * (lrow / rs);  // the cell this lane loads for block rb
#pragma unroll
        for (int b = 0; b < NBIG; ++b) la[b][rb] = (uint32_t)(sh_cell[b * kWG + c] + lk * bxs1[b]);
        tb[rb] = sh_cell[2 * kWG + wave * 64 + rs * rb];
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            const int i = lk + 4 * v;
            const int cell = wave * 64 + (i % rs) + rs * rb + 4 * rs * (i / rs);
            // (contiguous steps: cell*16 + lrow, i.e. 512 contiguous bytes per store instruction for rs >= 4)
            oc[rb][v] = cell < d.lo_cells ? sh_cell[3 * kWG + cell] + (int)d.nout[lrow] : -1;
            if (OCL) sh_oc[(rb * 4 + v) * kWG + tid] = oc[rb][v];
        }
    }

    auto issue = [&](const int hh, double (&dst)[4][KS]) {
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const double *__restrict__ b0 = big[0] + (uni(sh_hoff[2][hh]) + ks * bxs2[0]);
#pragma unroll
            for (int rb = 0; rb < 4; ++rb) {
                double p = b0[la[0][rb]];
                if (NBIG > 1) {
                    const double *__restrict__ b1 = big[NBIG - 1] + (uni(sh_hoff[3][hh]) + ks * bxs2[NBIG - 1]);
                    p *= b1[la[NBIG - 1][rb]];
                }
                dst[rb][ks] = p;
            }
        }
    };
    auto finish = [&](const int hh, const double (&a)[4][KS]) {
        const int ho = uni(sh_hoff[0][hh]), ht = uni(sh_hoff[1][hh]);
#pragma unroll
        for (int rb = 0; rb < 4; ++rb) {
            v4d acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int ks = 0; ks < KS; ++ks)
                acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a[rb][ks], shT[ht + tb[rb] + lrow + 16 * (4 * ks + lk)], acc, 0, 0, 0);
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                const int o = OCL ? sh_oc[(rb * 4 + v) * kWG + tid] : oc[rb][v];
                if (o >= 0) outp[ho + o] = acc[v];
            }
        }
    };
    if (NBIG == 1 && MIBN_PIPELINE) {
        double fa[4][KS], fb[4][KS];
        issue(0, fa);
        for (int hh = 0; hh < nh; hh += 2) {
            if (hh + 1 < nh) issue(hh + 1, fb);
            finish(hh, fa);
            if (hh + 1 < nh) {
                if (hh + 2 < nh) issue(hh + 2, fa);
                finish(hh + 1, fb);
            }
        }
    } else {
        double fa[4][KS];
        for (int hh = 0; hh < nh; ++hh) {
            issue(hh, fa);
            finish(hh, fa);
        }
    }
}

// OUTER class: out[r, n] = sum_x A[r, x] * B[r, n, x] with two big inputs (planner.h), as the same fp64 MFMA row-block
// product as fiber_mfma_call - the B operand comes from the second table instead of T: lane l of row block rb
// reads B[hi offset + row block's offset + nB[l & 15] + x(4*ks + (l >> 4))], mostly L2 hits (every B element feeds
// all the cells of A that share its batch axes).  2 x 16 loads per 1024 outputs instead of 32 per 64.
template <int KS>
__device__ __forceinline__ void outer_mfma_call(const uint32_t *sw, double *__restrict__ shT, int (*sh_hoff)[kTileMax],
                                                uint32_t *__restrict__ shX, const double *__restrict__ pool,
                                                double *__restrict__ slot, const int tid, const int h_begin, const int h_end) {
    const LaneOff lane_off = fiber_prologue(sw, shT, sh_hoff, pool, slot, tid, h_begin, h_end);
    const FiberDesc d = fiber_desc(sw);
    double *__restrict__ outp = slot + ((uint64_t)sw[4] | ((uint64_t)sw[5] << 32));
    const double *__restrict__ bigA = table_ptr(d.bigs[0], d.bigs[1], pool, slot);
    const double *__restrict__ bigB = table_ptr(d.bigs[4], d.bigs[5], pool, slot);
    const int axs1 = (int)d.bigs[2], axs2 = (int)d.bigs[3], bxs1 = (int)d.bigs[6], bxs2 = (int)d.bigs[7];
    int *sh_cell = reinterpret_cast<int *>(shX);  // [4][kWG]: A offset, B offset, output offset, T offset of every lane cell
    sh_cell[tid] = lane_off.b0;
    sh_cell[kWG + tid] = lane_off.b1;
    sh_cell[2 * kWG + tid] = lane_off.o;
    sh_cell[3 * kWG + tid] = lane_off.t;
    __syncthreads();
    const bool has_t = d.T > 0;  // CPT slices among the inputs: their product T[n, x, ctrl] scales the B operand
    const int nh = h_end - h_begin;
    const int wave = tid >> 6, lane = tid & 63;
    const int lrow = lane & 15, lk = lane >> 4;
    const int rs = (int)((sw[1] >> kRowStrideShift) & 0xff);
    uint32_t la[4], lb[4];
    const int t_lane = lrow + 16 * lk;  // (the T offset of row block rb is re-read from sh_cell in the loop: four registers fewer)
    // output offset of accumulator element v of row block rb (-1: beyond the lane block): 16 values per lane, parked in LDS
    // behind sh_cell (shX has exactly the room) - in registers they pushed the kernel over its budget (an 8-byte spill
    // reloaded in this loop was the level kernel's only scratch)
    int *sh_oc = sh_cell + 4 * kWG;  // [16][kWG]
    int oc_val[4][4];
#pragma unroll
    for (int rb = 0; rb < 4; ++rb) {
        const int c = wave * 64 + (lrow % rs) + rs * rb + 4 * rs * (lrow / rs);
        la[rb] = (uint32_t)(sh_cell[c] + lk * axs1);
        lb[rb] = (uint32_t)(sh_cell[kWG + wave * 64 + rs * rb] + (int)d.nB[lrow] + lk * bxs1);
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            const int i = lk + 4 * v;
            const int oc_cell = wave * 64 + (i % rs) + rs * rb + 4 * rs * (i / rs);
            oc_val[rb][v] = oc_cell < d.lo_cells ? sh_cell[2 * kWG + oc_cell] + (int)d.nout[lrow] : -1;
        }
    }
#pragma unroll
    for (int rb = 0; rb < 4; ++rb)
#pragma unroll
        for (int v = 0; v < 4; ++v) sh_oc[(rb * 4 + v) * kWG + tid] = oc_val[rb][v];  // (a lane reads back only what it wrote: no barrier)
    // An operand whose wave-uniform offset did not change since the previous iteration (the iteration moved along an
    // axis only the other table depends on) stays in registers: these steps multiply two tables over the union of
    // their axes, every element of A feeds all the cells of B that share its batch axes and vice versa.
    double a[4][KS], braw[4][KS];
    int prev_a = 0, prev_b = 0;
    for (int hh = 0; hh < nh; ++hh) {
        const int ho = uni(sh_hoff[0][hh]);
        const int ha = uni(sh_hoff[2][hh]), hb = uni(sh_hoff[3][hh]);
        if (hh == 0 || ha != prev_a) {
            const double *__restrict__ a0 = bigA + ha;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks)
#pragma unroll
                for (int rb = 0; rb < 4; ++rb) a[rb][ks] = a0[la[rb] + (uint32_t)(ks * axs2)];
            prev_a = ha;
        }
        if (hh == 0 || hb != prev_b) {
            const double *__restrict__ b0 = bigB + hb;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks)
#pragma unroll
                for (int rb = 0; rb < 4; ++rb) braw[rb][ks] = b0[lb[rb] + (uint32_t)(ks * bxs2)];
            prev_b = hb;
        }
        const int ht = has_t ? uni(sh_hoff[1][hh]) : 0;
#pragma unroll
        for (int rb = 0; rb < 4; ++rb) {
            v4d acc = {0.0, 0.0, 0.0, 0.0};
            const int tb = has_t ? sh_cell[3 * kWG + wave * 64 + rs * rb] + t_lane : 0;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const double bo = has_t ? braw[rb][ks] * shT[ht + tb + 64 * ks] : braw[rb][ks];
                acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a[rb][ks], bo, acc, 0, 0, 0);
            }
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                const int o = sh_oc[(rb * 4 + v) * kWG + tid];
                if (o >= 0) outp[ho + o] = acc[v];
            }
        }
    }
}

// CHAIN class (planner.h): three 4-state variables of one big table F per pass.  Per row block (16 cells sharing their
// table slices) and value of x3 the pair (x1, x2) is the [16 x 16] x [16 x 16] fp64-MFMA product of fiber_mfma_call
// against T12[., ., ctrl12, x3]; x3 is then summed out of the four accumulators in registers against T3 (16 values per
// lane: its output column's [n3][x3] slice).  16 loads, 16 MFMAs, 64 FMAs and 16 stores per lane and row block; the 64
// outputs of a cell are contiguous (n12 fastest), every store instruction writes four full 128-byte lines.
__device__ __forceinline__ void chain_mfma_call(const uint32_t *sw, double *__restrict__ shT, int (*sh_hoff)[kTileMax],
                                                uint32_t *__restrict__ shX, const double *__restrict__ pool,
                                                double *__restrict__ slot, const int tid, const int h_begin, const int h_end) {
    const LaneOff lane_off = fiber_prologue(sw, shT, sh_hoff, pool, slot, tid, h_begin, h_end);  // builds T12 (x3 = its last ctrl dimension)
    const FiberDesc d = fiber_desc(sw);
    const int T12 = (int)d.bigs[4], T3 = (int)d.bigs[5], fx3 = (int)d.bigs[6], t12x3 = (int)d.bigs[7];
    {   // T3[x3 + 4*n3 (+ 16*n12) + ctrl3...] after T12
        const int n3s = (int)d.chain[0], nd3 = (int)(d.chain[1] & 0xff);
        const uint32_t *tcard3 = d.chain + 2, *recs = d.chain + 2 + nd3;
        for (int t = tid; t < T3; t += kWG) {
            double v = 1.0;
            for (int j = 0; j < n3s; ++j) {
                const uint32_t *rec = recs + j * (2 + nd3);
                int off = 0, c = t;
                for (int k = 0; k < nd3; ++k) { const int cd = (int)tcard3[k]; const int qq = c / cd; off += (c - qq * cd) * (int)rec[2 + k]; c = qq; }
                v *= table_ptr(rec[0], rec[1], pool, slot)[off];
            }
            shT[T12 + t] = v;
        }
    }
    const bool n12dep = ((d.chain[1] >> 8) & 1) != 0;
    double *__restrict__ outp = slot + ((uint64_t)sw[4] | ((uint64_t)sw[5] << 32));
    const double *__restrict__ F = table_ptr(d.bigs[0], d.bigs[1], pool, slot);
    const int bxs1 = (int)d.bigs[2], bxs2 = (int)d.bigs[3];
    int *sh_cell = reinterpret_cast<int *>(shX);  // [4][kWG]: F offset, T3 offset, T12 offset, output offset of every lane cell
    sh_cell[tid] = lane_off.b0;
    sh_cell[kWG + tid] = lane_off.b1;
    sh_cell[2 * kWG + tid] = lane_off.t;
    sh_cell[3 * kWG + tid] = lane_off.o;
    __syncthreads();  // (also: T3 is complete)
    const int nh = h_end - h_begin;
    const int wave = tid >> 6, lane = tid & 63;
    const int lrow = lane & 15, lk = lane >> 4;
    const int rs = (int)((sw[1] >> kRowStrideShift) & 0xff);
    uint32_t la[4];
    int tb[4], t3b[4];
#pragma unroll
    for (int rb = 0; rb < 4; ++rb) {
        const int c = wave * 64 + (lrow % rs) + rs * rb + 4 * rs * (lrow / rs);
        la[rb] = (uint32_t)(sh_cell[c] + lk * bxs1);
        tb[rb] = sh_cell[2 * kWG + wave * 64 + rs * rb] + lrow + 16 * lk;
        t3b[rb] = T12 + sh_cell[kWG + wave * 64 + rs * rb] + (n12dep ? 16 * lrow : 0);
    }
    // output offset of accumulator element v of row block rb: the lane block of a CHAIN step is contiguous in the output
    // (cell l at 64 * l, emit_chain_as), so the 16 offsets are arithmetic in (rb, v) and need no registers
    const int lo_cells = d.lo_cells;
    auto out_cell = [&](const int rb, const int v) {
        const int i = lk + 4 * v;
        return wave * 64 + (i % rs) + rs * rb + 4 * rs * (i / rs);
    };
    // two load buffers: the 16 loads of the next row block are issued before the MFMAs of the current one
    double fa[4][4], fb[4][4];  // [x3][ks]
    auto issue = [&](const int hh, const int rb, double (&dst)[4][4]) {
#pragma unroll
        for (int x3 = 0; x3 < 4; ++x3)
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const double *__restrict__ bk = F + (uni(sh_hoff[2][hh]) + ks * bxs2 + x3 * fx3);
                dst[x3][ks] = bk[la[rb]];
            }
    };
    auto finish = [&](const int hh, const int rb, const double (&a)[4][4]) {
        const int ho = uni(sh_hoff[0][hh]), ht = uni(sh_hoff[1][hh]), ht3 = uni(sh_hoff[3][hh]);
        v4d acc[4];
#pragma unroll
        for (int x3 = 0; x3 < 4; ++x3) acc[x3] = v4d{0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
#pragma unroll
            for (int x3 = 0; x3 < 4; ++x3)
                acc[x3] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[x3][ks], shT[ht + tb[rb] + x3 * t12x3 + 64 * ks], acc[x3], 0, 0, 0);
        const double *__restrict__ t3p = shT + (ht3 + t3b[rb]);
#pragma unroll
        for (int half = 0; half < 2; ++half) {  // two n3 values at a time (register budget)
            v4d o[2];
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const int n3 = 2 * half + e;
                o[e] = acc[0] * t3p[4 * n3];
#pragma unroll
                for (int x3 = 1; x3 < 4; ++x3) o[e] += acc[x3] * t3p[4 * n3 + x3];
            }
#pragma unroll
            for (int e = 0; e < 2; ++e)
#pragma unroll
                for (int v = 0; v < 4; ++v) {
                    const int cell = out_cell(rb, v);
                    if (cell < lo_cells) outp[ho + 64 * cell + lrow + 16 * (2 * half + e)] = o[e][v];  // 16 lanes = one full 128-byte line
                }
        }
    };
    issue(0, 0, fa);
    for (int hh = 0; hh < nh; ++hh) {
        issue(hh, 1, fb);
        finish(hh, 0, fa);
        issue(hh, 2, fa);
        finish(hh, 1, fb);
        issue(hh, 3, fb);
        finish(hh, 2, fa);
        if (hh + 1 < nh) issue(hh + 1, 0, fa);
        finish(hh, 3, fb);
    }
}

// posterior / posterior.sum()  (bayes_net.py:790); an all-zero table (zero-probability evidence) stays zero.  One wave (a segment).
// The summation order - lane-strided partial sums, then the butterfly - is the one a workgroup of four waves used up to
// round 2 only for tables of at most 64 cells (every query of one or two variables): larger final tables round differently
// in the last bit.
__device__ __forceinline__ void normalise_wave(double *__restrict__ p, int n, int lane) {
    double s = 0.0;
    for (int i = lane; i < n; i += 64) s += p[i];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_down(s, o, 64);
    const double total = __shfl(s, 0, 64);
    if (total > 0.0)
        for (int i = lane; i < n; i += 64) p[i] = p[i] / total;
}

// A workgroup of SEGMENTS (planner.h kSegPerWg): wave w runs item first + w - the small GENERIC steps of one request, back to
// back, on its own.  A chain of dependent tiny steps is latency, not bandwidth, and a wave is enough for one (< 4 096 output
// cells per step): four chains per workgroup, twelve per CU, hide four times as much of it as one.  The wave's copy of the
// step descriptor and its offset table live in its quarter of the first 12 KB of shT (a segment has no T).
__device__ __forceinline__ void segment_wave(const LevelArgs &A, const uint32_t first, const int n_valid, double *shT, const int tid) {
    const int wave = tid >> 6, lane = tid & 63;
    if (wave >= n_valid) return;
    const uint32_t idx = first + (uint32_t)wave;
    const uint32_t req = (uint32_t)uni((int)A.items[idx].req), rel_off = (uint32_t)uni((int)A.items[idx].rel_off);
    const int n_steps = (int)((uint32_t)uni((int)A.items[idx].a) & ~kItemSegment);
    const uint64_t ao = A.arena_off[req], po = A.prog_off[req];
    double *slot = A.arena + (((uint64_t)(uint32_t)uni((int)(ao >> 32)) << 32) | (uint32_t)uni((int)(ao & 0xffffffffu)));
    const uint32_t *p = A.prog + (((uint64_t)(uint32_t)uni((int)(po >> 32)) << 32) | (uint32_t)uni((int)(po & 0xffffffffu))) + rel_off;
    unsigned char *wbase = reinterpret_cast<unsigned char *>(shT) + wave * (kMaxStepWords * 4 + kMaxIn * kTileMax * 4);
    uint32_t *w_step = reinterpret_cast<uint32_t *>(wbase);
    int (*w_hoff)[kTileMax] = reinterpret_cast<int (*)[kTileMax]>(wbase + kMaxStepWords * 4);
    // The descriptor of step s + 1 is fetched (kMaxStepWords words, whatever its length: the chunk's program buffer ends with that
    // much slack) while step s runs: a step of a chain is two dependent global round trips - its descriptor, then its inputs -
    // and this takes the first one off the chain.
    constexpr int kPre = kMaxStepWords / 64;
    uint32_t nxt[kPre];
#pragma unroll
    for (int k = 0; k < kPre; ++k) nxt[k] = p[lane + 64 * k];
    for (int s = 0; s < n_steps; ++s) {
        lanes_sync<64>();  // the previous step's stores are done and visible to this wave; its descriptor copy is reusable
#pragma unroll
        for (int k = 0; k < kPre; ++k) w_step[lane + 64 * k] = nxt[k];
        lanes_sync<64>();
        const int words = (int)w_step[6];
        if (s + 1 < n_steps) {
#pragma unroll
            for (int k = 0; k < kPre; ++k) nxt[k] = p[words + lane + 64 * k];
        }
        generic_dispatch<64>((w_step[0] >> 8) & 0xff, w_step, w_hoff, A.pool, slot, A.results, lane, 0, (int)w_step[3]);
        if ((w_step[1] >> 16) & kFlagFinal) {
            const uint64_t out_off = (uint64_t)w_step[4] | ((uint64_t)w_step[5] << 32);
            lanes_sync<64>();
            normalise_wave(A.results + out_off, (int)(w_step[2] * w_step[3]), lane);
        }
        p += words;
    }
}

#ifndef MIBN_PATHS
#define MIBN_PATHS 0xff  // build-time experiment hook: which families of tile code are compiled in (register / spill reports per family)
#endif
#define MIBN_FIBER_CASES(NB, C)                                                                        \
    case (NB - 1) * 18 + C * 6 + 0: fiber_call<NB, C, 0>(sh_step, shT, sh_hoff, shX, A.pool, slot, tid, h0, h1); break; \
    case (NB - 1) * 18 + C * 6 + 1: fiber_call<NB, C, 1>(sh_step, shT, sh_hoff, shX, A.pool, slot, tid, h0, h1); break; \
    case (NB - 1) * 18 + C * 6 + 2: fiber_call<NB, C, 2>(sh_step, shT, sh_hoff, shX, A.pool, slot, tid, h0, h1); break; \
    case (NB - 1) * 18 + C * 6 + 3: fiber_call<NB, C, 3>(sh_step, shT, sh_hoff, shX, A.pool, slot, tid, h0, h1); break;
#define MIBN_MFMA_CASES(NB)                                                                            \
    case (NB - 1) * 18 + 0 * 6 + 4: fiber_mfma_call<NB, 1>(sh_step, shT, sh_hoff, shX, A.pool, slot, tid, h0, h1); break; \
    case (NB - 1) * 18 + 1 * 6 + 4: fiber_mfma_call<NB, 4>(sh_step, shT, sh_hoff, shX, A.pool, slot, tid, h0, h1); break;

// One level of the schedule: workgroup b runs item wg_item[b] - a tile of a big step or a segment of small steps.
__global__ __launch_bounds__(kWG, MIBN_MIN_WAVES) void ve_level_kernel(const LevelArgs A) {
    __shared__ __attribute__((aligned(16))) double shT[kMaxT];
    __shared__ __attribute__((aligned(16))) uint32_t shX[4 * 64 * kXRow];
    __shared__ uint32_t sh_step[kMaxStepWords];
    __shared__ int sh_hoff[kMaxIn][kTileMax];
    const int tid = threadIdx.x;
    const uint32_t wg = blockIdx.x + A.wg_base;
    // (everything about the work item is wave-uniform: kept in scalar registers explicitly - as lane values the arena and
    //  program pointers were spilled around the switch over the step forms)
    const uint32_t item_idx = (uint32_t)uni((int)A.wg_item[wg]);
    Item it;
    it.req = (uint32_t)uni((int)A.items[item_idx].req);
    it.rel_off = (uint32_t)uni((int)A.items[item_idx].rel_off);
    it.a = (uint32_t)uni((int)A.items[item_idx].a);
    it.b = (uint32_t)uni((int)A.items[item_idx].b);
    if (it.a & kItemSegment) {
        segment_wave(A, item_idx, (int)it.b, shT, tid);
        return;
    }
    const uint64_t ao = A.arena_off[it.req], po = A.prog_off[it.req];
    double *slot = A.arena + (((uint64_t)(uint32_t)uni((int)(ao >> 32)) << 32) | (uint32_t)uni((int)(ao & 0xffffffffu)));
    const uint32_t *p = A.prog + (((uint64_t)(uint32_t)uni((int)(po >> 32)) << 32) | (uint32_t)uni((int)(po & 0xffffffffu))) + it.rel_off;
    // TILE of a big step: hi iterations [h0, h1)
    const int words = (int)p[6];
    for (int i = tid; i < words; i += kWG) sh_step[i] = p[i];
    __syncthreads();
    const int h0 = (int)((wg - it.b) * it.a);
    const int h1 = min((int)sh_step[3], h0 + (int)it.a);
    if ((sh_step[0] & 0xff) == kKindFiber && ((sh_step[1] >> 16) & kFlagChain)) {
        if constexpr (MIBN_PATHS & 1) chain_mfma_call(sh_step, shT, sh_hoff, shX, A.pool, slot, tid, h0, h1);
    } else if ((sh_step[0] & 0xff) == kKindFiber) {
        const int cx = (int)(sh_step[1] & 0xffff), c1 = (int)(sh_step[8] >> 16), NC = (int)(sh_step[7] >> 16);
        const bool contig = ((sh_step[1] >> 16) & kFlagContig) != 0;
        const int cxc = (cx == 4 && c1 == 4) ? 0 : ((cx == 16 && c1 == 4) ? 1 : 2);
        const bool mfma = ((sh_step[1] >> kRowStrideShift) & 0xff) != 0 && cxc < 2;
        const bool outer = ((sh_step[1] >> 16) & kFlagOuter) != 0;
        const int ncc = outer ? 5 : (NC == 1 ? 0 : ((NC == 4 && contig) ? 1 : ((NC == 16 && mfma) ? 4 : ((NC == 16 && contig) ? 2 : 3))));
        switch (((int)(sh_step[7] & 0xf) - 1) * 18 + cxc * 6 + ncc) {
#if MIBN_PATHS & 2
            MIBN_FIBER_CASES(1, 0)
            MIBN_FIBER_CASES(1, 1)
            MIBN_FIBER_CASES(1, 2)
#endif
#if MIBN_PATHS & 4
            MIBN_FIBER_CASES(2, 0)
            MIBN_FIBER_CASES(2, 1)
            MIBN_FIBER_CASES(2, 2)
#endif
#if MIBN_PATHS & 8
            MIBN_MFMA_CASES(1)
#endif
#if MIBN_PATHS & 16
            MIBN_MFMA_CASES(2)
#endif
#if MIBN_PATHS & 32
            case 18 + 0 * 6 + 5: outer_mfma_call<1>(sh_step, shT, sh_hoff, shX, A.pool, slot, tid, h0, h1); break;
            case 18 + 1 * 6 + 5: outer_mfma_call<4>(sh_step, shT, sh_hoff, shX, A.pool, slot, tid, h0, h1); break;
#endif
        }
    } else {
        if constexpr (MIBN_PATHS & 64) generic_dispatch<kWG>((sh_step[0] >> 8) & 0xff, sh_step, sh_hoff, A.pool, slot, A.results, tid, h0, h1);
    }
}
#undef MIBN_FIBER_CASES
#undef MIBN_MFMA_CASES

// Round 5: the one-table fp64-MFMA pair classes (fiber<1, cx4 | cx16, nc16-mfma>: a quarter of the C3 bytes) as a kernel of their
// own (option mfma_kernel, a fourth stream beside the level's other launches).  Inside ve_level_kernel they lived on that kernel's
// register budget - 168 VGPRs for the GENERIC and two-table forms = three waves per SIMD - and its SQ counters said occupancy
// (57 % of the wave cycles parked on s_waitcnt).  Alone, with the sixteen output offsets of a lane parked in LDS, they fit 128
// registers without scratch: four waves per SIMD, the four 40 KB workgroups a CU's LDS takes.
__global__ __launch_bounds__(kWG, 4) void ve_mfma_kernel(const LevelArgs A) {
    __shared__ __attribute__((aligned(16))) double shT[kMaxT];
    __shared__ __attribute__((aligned(16))) uint32_t shX[4 * 64 * kXRow];
    __shared__ uint32_t sh_step[kMaxStepWords];
    __shared__ int sh_hoff[kMaxIn][kTileMax];
    const int tid = threadIdx.x;
    const uint32_t wg = blockIdx.x + A.wg_base;
    const uint32_t item_idx = (uint32_t)uni((int)A.wg_item[wg]);
    Item it;
    it.req = (uint32_t)uni((int)A.items[item_idx].req);
    it.rel_off = (uint32_t)uni((int)A.items[item_idx].rel_off);
    it.a = (uint32_t)uni((int)A.items[item_idx].a);
    it.b = (uint32_t)uni((int)A.items[item_idx].b);
    const uint64_t ao = A.arena_off[it.req], po = A.prog_off[it.req];
    double *slot = A.arena + (((uint64_t)(uint32_t)uni((int)(ao >> 32)) << 32) | (uint32_t)uni((int)(ao & 0xffffffffu)));
    const uint32_t *p = A.prog + (((uint64_t)(uint32_t)uni((int)(po >> 32)) << 32) | (uint32_t)uni((int)(po & 0xffffffffu))) + it.rel_off;
    const int words = (int)p[6];
    for (int i = tid; i < words; i += kWG) sh_step[i] = p[i];
    __syncthreads();
    const int h0 = (int)((wg - it.b) * it.a);
    const int h1 = min((int)sh_step[3], h0 + (int)it.a);
    const int cx = (int)(sh_step[1] & 0xffff);
    if (cx == 4) fiber_mfma_call<1, 1, true>(sh_step, shT, sh_hoff, shX, A.pool, slot, tid, h0, h1);
    else fiber_mfma_call<1, 4, true>(sh_step, shT, sh_hoff, shX, A.pool, slot, tid, h0, h1);
}

// Round 4: the segments of a level as a kernel of their own (option seg_kernel, launched on a third stream beside the level's
// other launches).  A segment is a chain of dependent tiny steps - a few global round trips per step, nothing to stream - and
// inside ve_level_kernel it paid for that kernel's resources: 40 KB of LDS and 168 VGPRs per workgroup = three workgroups =
// twelve chains per CU.  Here a workgroup needs its four waves' descriptor copies and offset tables (12 KB) and the registers
// of the GENERIC form alone: the wave slots of a CU, not its LDS, bound the chains in flight.
#ifndef MIBN_SEG_WAVES
#define MIBN_SEG_WAVES 4  // waves per SIMD the segment kernel is compiled for (chains in flight per CU = 4 x this)
#endif
__global__ __launch_bounds__(kWG, MIBN_SEG_WAVES) void ve_segment_kernel(const LevelArgs A) {
    __shared__ __attribute__((aligned(16))) unsigned char sh_seg[kSegPerWg * (kMaxStepWords * 4 + kMaxIn * kTileMax * 4)];
    const uint32_t wg = blockIdx.x + A.wg_base;
    const uint32_t item_idx = (uint32_t)uni((int)A.wg_item[wg]);
    const int n_valid = uni((int)A.items[item_idx].b);
    segment_wave(A, item_idx, n_valid, reinterpret_cast<double *>(sh_seg), threadIdx.x);
}

}  // namespace mibn
