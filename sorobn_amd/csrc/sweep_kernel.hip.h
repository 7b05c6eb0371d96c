// CDNA4 (gfx950) kernel of the SWEEP form (planner.h): k = 3..5 four-state variables of one big table eliminated in ONE
// pass over it, with the tile resident in LDS.
//
// The streaming forms of ve_kernel.hip.h read and write the elimination frontier once per one, two or three eliminated
// variables (FIBER, pair, CHAIN); what bounds them is HBM, and the only thing left to gain is bytes.  A sweep along a row of
// the 10x10 grid eliminates ten variables of a 4^10-cell table; SWEEP does it in two passes instead of four:
//   * a workgroup (512 lanes) owns tiles of all 4^k combinations of the eliminated variables x Rt consecutive cells of
//     the remaining (R) axes, Rt = 8192 / 4^k: 64 KiB, read as 4^k runs of Rt * 8 bytes (the eliminated variables are
//     the slowest axes of the table) with 16-byte loads and staged as L[r + Rt * xc];
//   * stage j - in elimination order - contracts one digit of xc in place: every lane takes fibers of four cells along
//     that digit, multiplies by its 4 x cout slice of T_j (registers; the slice depends on up to three ctrl values: other
//     digits of the fiber or bits of r) and writes the cout results over the fiber: the new variable takes the digit of the
//     eliminated one (cout = 4) or the digit dies (cout = 1).  64 fp64 FMAs per 64 bytes of LDS traffic, five times;
//   * the tile's output - surviving digits fastest, then r - is ONE contiguous block of Rt * 4^kout cells;
//   * the next tile's loads are issued before the stores of the current one and land in registers (PMC calibration on a
//     known byte count, profiles/r02_u_pmc_calibration.log: WRITE_SIZE exact; FETCH_SIZE counts these 64-byte runs in full).
// LDS: 64 KiB tile + 8 KiB T + 1.5 KiB descriptor = 73.5 KiB -> two workgroups per CU, 4 waves per SIMD (<= 128 VGPRs).
// Measured form of the idea: tools/ubench/sweep_lds.hip (five variables, 4.0 - 4.3 TB/s of algorithmic traffic).
#pragma once
#include <hip/hip_runtime.h>

#include "planner.h"
#include "ve_kernel.hip.h"

namespace mibn {

constexpr int kSweepLdsBytes = kSweepTileCells * 8 + kSweepMaxT * 8 + kMaxStepWords * 4;

// LDS index of output cell c of a tile: c = surviving digits (2 bits each, ascending) | r << 2 kout
__device__ __forceinline__ int sweep_perm(const int c, const int kout, const int rb, const uint32_t surv) {
    int idx = c >> (2 * kout);
#pragma unroll
    for (int q = 0; q < 5; ++q)
        if (q < kout) idx += ((c >> (2 * q)) & 3) << (rb + 2 * (int)((surv >> (4 * q)) & 15));
    return idx;
}

// The four fibers of a lane in one stage.  SX / SL: LDS strides of the contracted digit and of the loop digit - compile-time
// for the canonical steps (every LDS access is base + immediate), 0 = runtime (sx, sl).
template <int COUT, int SX, int SL>
__device__ __forceinline__ void sweep_fibers(double *__restrict__ L, const double *__restrict__ T, const int base, const int sx_,
                                             const int sl_, const int toff, const int loop_ts) {
    const int sx = SX ? SX : sx_, sl = SL ? SL : sl_;
    double t[4 * COUT];
#pragma unroll
    for (int l = 0; l < 4; ++l) {
        if (l == 0 || loop_ts) {  // (uniform) the loop digit is a ctrl value of this stage: another T slice per trip
            const double *__restrict__ Tp = T + (toff + l * loop_ts);
#pragma unroll
            for (int q = 0; q < 4 * COUT; ++q) t[q] = Tp[q];
        }
        const int b = base + l * sl;
        double f[4];
#pragma unroll
        for (int x = 0; x < 4; ++x) f[x] = L[b + x * sx];
#pragma unroll
        for (int n = 0; n < COUT; ++n) {
            double s = f[0] * t[n];  // (explicit FMAs: with contraction left to the compiler, which product of a sum stays a
#pragma unroll                   //  plain multiply differs from one code shape to the next - the two sweep kernels then differ in the last bit)
            for (int x = 1; x < 4; ++x) s = __builtin_fma(f[x], t[n + COUT * x], s);
            L[b + n * sx] = s;
        }
    }
}

// T_j[n + cout * (x + 4 * ctrl)] = product of the CPT slices of stage j, all stages of the step (once per work item)
struct SweepTables {
    double *T;
    const uint32_t *stw, *smw;
    const double *pool;
    const double *slot;
    int k, t_total;
};
template <int WG>
__device__ __forceinline__ void sweep_build_tables_n(const SweepTables &b, const int tid) {
    for (int t = tid; t < b.t_total; t += WG) {
        int j = 0, rec0 = 0;
        while (j + 1 < b.k && t >= (int)((b.stw[j * kSweepStageWords + 1] & 0xffff) + (b.stw[j * kSweepStageWords + 1] >> 16))) {
            rec0 += (int)((b.stw[j * kSweepStageWords] >> 8) & 15);
            ++j;
        }
        const uint32_t s0 = b.stw[j * kSweepStageWords], s1 = b.stw[j * kSweepStageWords + 1];
        const int cout = (s0 >> 4) & 15, ns = (s0 >> 8) & 15;
        const int e = t - (int)(s1 & 0xffff);
        const int n = cout == 4 ? (e & 3) : 0;
        const int r = cout == 4 ? (e >> 2) : e;
        const int x = r & 3, c0 = (r >> 2) & 3, c1 = (r >> 4) & 3, c2 = (r >> 6) & 3;
        double v = 1.0;
        for (int i = 0; i < ns; ++i) {
            const uint32_t *rec = b.smw + (rec0 + i) * kSweepSmallWords;
            const int off = n * (int)rec[2] + x * (int)rec[3] + c0 * (int)rec[4] + c1 * (int)rec[5] + c2 * (int)rec[6];
            v *= table_ptr(rec[0], rec[1], b.pool, b.slot)[off];
        }
        b.T[t] = v;
    }
}
__device__ __forceinline__ void sweep_build_tables(const SweepTables &b, const int tid) { sweep_build_tables_n<kSweepWG>(b, tid); }

// What a lane keeps per stage across the tiles of a work item: the LDS index of its first fiber and the T offset its digits
// select.  Everything else of a stage is uniform and re-derived from the descriptor words (scalar registers) per tile.
__device__ __forceinline__ void sweep_stage_lane(const uint32_t s0, const uint32_t s1, const uint32_t (&cw)[3], const int rb, const int tid,
                                                 int &base, int &toff) {
    const int nctrl = (s0 >> 12) & 15, loop = (s0 >> 16) & 15;
    base = tid & ((1 << rb) - 1);
    int bits = tid >> rb;
#pragma unroll
    for (int f = 0; f < 3; ++f) {
        const int fd = (s0 >> (20 + 4 * f)) & 15;
        if (fd != 7) {
            base += (bits & 3) << (rb + 2 * fd);
            bits >>= 2;
        }
    }
    toff = (int)(s1 & 0xffff);
#pragma unroll
    for (int c = 0; c < 3; ++c)
        if (c < nctrl) {
            const int src = cw[c] & 0xff, ts = (int)(cw[c] >> 8);
            if (src < 8 && src != loop) toff += ((base >> (rb + 2 * src)) & 3) * ts;
        }
}

// The tiles [t_begin, t_end) of one work item.  K > 0: canonical step of K variables (stage j contracts digit K - 1 - j, the
// loop digit follows sweep_loop_digit): all LDS strides are immediates and the stages are unrolled.
template <int K>
__device__ __forceinline__ void sweep_tiles_impl(double *__restrict__ L, const double *__restrict__ T, const uint32_t *stw, const int k_rt,
                                                 const int rb_rt, const double *__restrict__ F, double *__restrict__ outp,
                                                 const long Rcells, const int t_begin, const int t_end, const int kout,
                                                 const uint32_t surv, const int tid, const SweepTables &tb) {
    constexpr int KS = K ? K : 5;  // stage slots
    const int k = K ? K : k_rt, rb = K ? 13 - 2 * K : rb_rt;
    const int Rt = 1 << rb;
    uint32_t sw0[KS], sw1[KS], swc[KS][3];  // the stage records: uniform (scalar registers)
    int base[KS], toff0[KS];                // per lane
#pragma unroll
    for (int j = 0; j < KS; ++j) {
        sw0[j] = sw1[j] = swc[j][0] = swc[j][1] = swc[j][2] = 0;
        base[j] = toff0[j] = 0;
        if (j < k) {
            sw0[j] = (uint32_t)uni((int)stw[j * kSweepStageWords]);
            sw1[j] = (uint32_t)uni((int)stw[j * kSweepStageWords + 1]);
#pragma unroll
            for (int c = 0; c < 3; ++c) swc[j][c] = (uint32_t)uni((int)stw[j * kSweepStageWords + 2 + c]);
            sweep_stage_lane(sw0[j], sw1[j], swc[j], rb, tid, base[j], toff0[j]);
        }
    }
    const int ocells = Rt << (2 * kout);
    const int p_tid2 = sweep_perm(2 * tid, kout, rb, surv);
    const int st0 = kout > 0 ? 1 << (rb + 2 * (int)(surv & 15)) : 1;  // LDS stride between output cells c and c + 1 (c even)
    // the tile's loads: pair c2 = i * 512 + tid -> cells (2 c2, 2 c2 + 1) of L = R cells (2 rp, 2 rp + 1) of combination xc
    const int rp = tid & ((Rt >> 1) - 1);
    const long g_tid = (long)(tid >> (rb - 1)) * Rcells + 2 * rp;
    const long g_step = (long)(kSweepWG >> (rb - 1)) * Rcells;  // per i
    double v[16];
    {
        const double *__restrict__ Ft = F + (long)t_begin * Rt + g_tid;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const double2 q = *reinterpret_cast<const double2 *>(Ft + i * g_step);
            v[2 * i] = q.x;
            v[2 * i + 1] = q.y;
        }
    }
    sweep_build_tables(tb, tid);  // (under the first tile's loads)
    for (int tile = t_begin; tile < t_end; ++tile) {
        __syncthreads();  // T is complete / the previous tile has been read out of L
#pragma unroll
        for (int i = 0; i < 8; ++i) *reinterpret_cast<double2 *>(L + 2 * (i * kSweepWG + tid)) = make_double2(v[2 * i], v[2 * i + 1]);
        __syncthreads();
        const int rg = tile * Rt + (tid & (Rt - 1));  // this lane's R cell in every stage (the lowest lane bits are r)
#define MIBN_SWEEP_STAGE(J)                                                                                                   \
        if (J < KS && J < k) {                                                                                                \
            constexpr int JJ = J < KS ? J : 0;                                                                                \
            const uint32_t s0 = sw0[JJ];                                                                                      \
            const int cout = (s0 >> 4) & 15, nctrl = (s0 >> 12) & 15, loop = (s0 >> 16) & 15;                                 \
            int toff = toff0[JJ], loop_ts = 0;                                                                                \
            _Pragma("unroll") for (int c = 0; c < 3; ++c)                                                                     \
                if (c < nctrl) {                                                                                              \
                    const int src = swc[JJ][c] & 0xff, ts = (int)(swc[JJ][c] >> 8);                                           \
                    if (src >= 8) toff += ((rg >> (src - 8)) & 3) * ts;                                                       \
                    else if (src == loop) loop_ts = ts;                                                                       \
                }                                                                                                             \
            constexpr int kRt = K ? (1 << (13 - 2 * (K ? K : 5))) : 0;                                                         \
            constexpr int kDig = K > J ? K - 1 - J : 0;                                                                       \
            constexpr int SX = K > J ? (kRt << (2 * kDig)) : 0;                                                               \
            constexpr int SL = K > J ? (kRt << (2 * sweep_loop_digit(K ? K : 5, kDig))) : 0;                                  \
            const int sx = 1 << (rb + 2 * (int)(s0 & 15)), sl = 1 << (rb + 2 * loop);                                         \
            if (cout == 4) sweep_fibers<4, SX, SL>(L, T, base[JJ], sx, sl, toff, loop_ts);                                    \
            else sweep_fibers<1, SX, SL>(L, T, base[JJ], sx, sl, toff, loop_ts);                                              \
            __syncthreads();                                                                                                  \
        }
        MIBN_SWEEP_STAGE(0)
        MIBN_SWEEP_STAGE(1)
        MIBN_SWEEP_STAGE(2)
        MIBN_SWEEP_STAGE(3)
        MIBN_SWEEP_STAGE(4)
#undef MIBN_SWEEP_STAGE
        {   // the next tile's loads fly during the stores.  No branch around them (the compiler then keeps v[] in scratch):
            // after the last tile every lane re-reads one and the same cell pair instead - one 64-byte request per wave
            const long more = tile + 1 < t_end ? 1 : 0;
            const double *__restrict__ Ft = F + (long)(tile + more) * Rt + more * g_tid;
            const long g_step_ = more * g_step;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const double2 q = *reinterpret_cast<const double2 *>(Ft + i * g_step_);
                v[2 * i] = q.x;
                v[2 * i + 1] = q.y;
            }
        }
        // the tile's output block, two cells (16 bytes) per lane and trip: cells c and c + 1 differ in the first surviving
        // digit (or, without one, in r)
        double *__restrict__ ot = outp + (long)tile * ocells;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int c0 = i * 2 * kSweepWG;  // (uniform)
            if (c0 < ocells) {
                const int c = c0 + 2 * tid;
                if (c < ocells) {
                    const int a = sweep_perm(c0, kout, rb, surv) + p_tid2;
                    *reinterpret_cast<double2 *>(ot + c) = make_double2(L[a], L[a + st0]);
                }
            }
        }
    }
}

template <int K>
__device__ __forceinline__ void sweep_tiles(double *__restrict__ L, const double *__restrict__ T, const uint32_t *stw, const double *__restrict__ F,
                                         double *__restrict__ outp, const long Rcells, const int t_begin, const int t_end, const int kout,
                                         const uint32_t surv, const int tid, const SweepTables &tb) {
    sweep_tiles_impl<K>(L, T, stw, K, 13 - 2 * K, F, outp, Rcells, t_begin, t_end, kout, surv, tid, tb);
}

// Any step (digits in any order): the stage records are re-read per tile and stage, all strides are runtime values.
__device__ __forceinline__ void sweep_tiles_any(double *__restrict__ L, const double *__restrict__ T, const uint32_t *stw, const int k, const int rb,
                                                const double *__restrict__ F, double *__restrict__ outp, const long Rcells, const int t_begin,
                                                const int t_end, const int kout, const uint32_t surv, const int tid, const SweepTables &tb) {
    sweep_build_tables(tb, tid);
    const int Rt = 1 << rb;
    const int ocells = Rt << (2 * kout);
    const int rp = tid & ((Rt >> 1) - 1);
    const long g_tid = (long)(tid >> (rb - 1)) * Rcells + 2 * rp;
    const long g_step = (long)(kSweepWG >> (rb - 1)) * Rcells;
    for (int tile = t_begin; tile < t_end; ++tile) {
        __syncthreads();
        const double *__restrict__ Ft = F + (long)tile * Rt + g_tid;
#pragma unroll
        for (int i = 0; i < 8; ++i) *reinterpret_cast<double2 *>(L + 2 * (i * kSweepWG + tid)) = *reinterpret_cast<const double2 *>(Ft + i * g_step);
        __syncthreads();
        const int rg = tile * Rt + (tid & (Rt - 1));
        for (int j = 0; j < k; ++j) {
            const uint32_t s0 = (uint32_t)uni((int)stw[j * kSweepStageWords]), s1 = (uint32_t)uni((int)stw[j * kSweepStageWords + 1]);
            uint32_t cw[3];
#pragma unroll
            for (int c = 0; c < 3; ++c) cw[c] = (uint32_t)uni((int)stw[j * kSweepStageWords + 2 + c]);
            int base, toff;
            sweep_stage_lane(s0, s1, cw, rb, tid, base, toff);
            const int cout = (s0 >> 4) & 15, nctrl = (s0 >> 12) & 15, loop = (s0 >> 16) & 15;
            int loop_ts = 0;
#pragma unroll
            for (int c = 0; c < 3; ++c)
                if (c < nctrl) {
                    const int src = cw[c] & 0xff, ts = (int)(cw[c] >> 8);
                    if (src >= 8) toff += ((rg >> (src - 8)) & 3) * ts;
                    else if (src == loop) loop_ts = ts;
                }
            const int sx = 1 << (rb + 2 * (int)(s0 & 15)), sl = 1 << (rb + 2 * loop);
            if (cout == 4) sweep_fibers<4, 0, 0>(L, T, base, sx, sl, toff, loop_ts);
            else sweep_fibers<1, 0, 0>(L, T, base, sx, sl, toff, loop_ts);
            __syncthreads();
        }
        double *__restrict__ ot = outp + (long)tile * ocells;
        for (int c = tid; c < ocells; c += kSweepWG) ot[c] = L[sweep_perm(c, kout, rb, surv)];
    }
}

// ------------------------------------------------------------------------------------------------------------------
// Round 3: ve_sweep_dma_kernel - the same pass, rebuilt around what the phase timers of tools/ubench/sweep_real.hip showed.
//
// The kernel above is NOT bound by HBM (a copy with its access pattern - 64-byte runs in, 64 KiB blocks out - streams at
// 5.05 TB/s through LDS-DMA, tools/ubench/pattern_copy.hip) but by LDS instruction issue: 1.0 - 1.3 us per stage with 8-byte
// accesses (hipcc pairs them into ds_read2_b64 / ds_write2st64_b64: 8 and 13 LDS cycles each) against 0.2 us of fp64 FMAs, a
// readout whose 32-lane groups hit ONE bank pair (1.6 us), a ds_write pass to fill the tile, and seven workgroup barriers
// per tile that keep all waves in the same phase (all read, all multiply, all write).  Two workgroups per CU serialised
// on the one LDS pipe: 8.1 us per tile and CU = 4.1 TB/s in isolation, 3.6 in the C3 mix.  What changed:
//   * the tile is filled by LDS-DMA (global_load_lds_dwordx4): no staging VGPRs, no ds_write pass.  LDS-DMA writes
//     base + 16 * lane: the staged layout L[r + Rt * xc] is kept, only the per-lane SOURCE address is strided.  hipcc does
//     not count the DMA (inline asm): the wave waits with s_waitcnt vmcnt(0) - VMEM operations of a wave retire in order,
//     the stores of the previous tile were issued before it - and __syncthreads() publishes every wave's share
//     (__syncthreads() lowers to s_waitcnt lgkmcnt(0) + s_barrier on gfx950: it does not drain the vector-memory queue);
//   * 16-byte LDS accesses: a lane owns the fibers of two adjacent R cells (one double2 per fiber position);
//   * wave-local stage pairs (0, 1) and (2, 3): both stages of a pair give a wave the same cells, so there is no workgroup
//     barrier between them and the waves drift out of phase - one wave's LDS writes overlap another one's FMAs;
//   * a wave-owned tail: the last stage writes the output block from its registers (no LDS write, no readout pass) and
//     refills the wave's own cells with the next tile's DMA before its FMAs: 3 workgroup barriers per tile instead of 7,
//     the DMA latency runs under the last stage and the stores.
// Two workgroups per CU as before (73.5 KiB of LDS, <= 128 VGPRs): while one waits for its tile the other one is in its
// stages.  Measured: 4.5 - 4.7 TB/s in isolation (sweep_real.hip), 4.2 - 4.3 TB/s in the C3 mix (profiles/r03_*).  Also measured
// and not kept: two tile buffers with one workgroup per CU (the next tile streams in under the stages: 4.0 - 4.1 TB/s, the
// lock-step of one workgroup's eight waves costs more than the exposed latency), 1 024-lane workgroups (3.1), a straight-
// line stage with all sixteen reads of a lane in flight (spills at 128 VGPRs), s_setprio around the memory phases (none).
// The step encoding and the arithmetic (order of the FMAs per output cell) are those of the kernel above: the two kernels
// agree bit for bit (tests/test_gpu_parity.py::test_sweep_kernels_agree_bit_for_bit).

#ifndef MIBN_SWEEP_TAIL_LOCAL
#define MIBN_SWEEP_TAIL_LOCAL 1  // wave-owned tail: no workgroup barrier between stage K - 2 and the last stage (same cells per wave)
#endif
#ifndef MIBN_SWEEP_DEAD_TAIL
#define MIBN_SWEEP_DEAD_TAIL 1   // the wave-owned tail also for five-variable passes in which one digit dies (kout = 4)
#endif
#ifndef MIBN_SWEEP_DEAD_SKIP
#define MIBN_SWEEP_DEAD_SKIP 1   // ... and the stages behind the dying one skip the fibers that carry nothing
#endif
#ifndef MIBN_SWEEP_VMCNT
#define MIBN_SWEEP_VMCNT 8       // wave-owned tail: the wait for the next tile's DMA leaves this many younger stores in flight (0: drain)
#endif
#ifndef MIBN_PROF_INIT  // (tools/ubench/sweep_real.hip defines these to time the phases of a tile with wall_clock64)
#define MIBN_PROF_INIT
#define MIBN_PROF_TICK(k)
#define MIBN_PROF_END
#endif

// byte address of an LDS location (what M0 takes)
__device__ __forceinline__ uint32_t lds_byte_addr(const void *p) {
    return (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const void *)p;
}

// one wave-instruction of LDS-DMA: lane l copies the 16 bytes at gsrc to LDS byte lds_base + 16 * l (lds_base wave-uniform)
__device__ __forceinline__ void dma16(const double *gsrc, const uint32_t lds_base) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(gsrc), "s"(lds_base)
                 : "memory");
}

// ---- the stages.  The canonical steps move 16 bytes per LDS instruction: a lane owns the fibers of TWO
// adjacent R cells (r = 2 rp, 2 rp + 1: one double2 per fiber position) for two values of the loop digit.  Lane bits, low
// to high: rp (rb - 1 bits), the free digits other than the contracted and the loop digit in ascending order (2 bits each),
// and one bit that selects the half of the loop digit's values.  For k = 5 (rp = 2 bits) the 16-lane groups of a
// ds_read_b128 then cover all 16 pairs (rp, digit 0) = all 64 banks as long as digit 0 is the first free digit.

// Lane geometry of stage J of a canonical K-variable step (this kernel's own choice - the planner's loop digit and thread
// fields in the stage record describe the register-staged kernel above).  Stage J contracts digit K - 1 - J.  The stages are
// PAIRED, (0, 1) and (2, 3): both stages of a pair give a wave the same set of cells - the contracted digit of either stage
// is a lane field or the fiber axis of the other one, the third field and the half of the loop digit are the same - so the
// second stage reads only what the wave itself wrote in the first: no workgroup barrier between them (the LDS serves a
// wave's instructions in order), the waves drift apart and one wave's writes overlap another one's reads and FMAs.
//   fields[0] must be digit 0 for k = 5 (bank spread of the 16-byte accesses, see above) unless digit 0 is contracted.
struct SweepGeom {
    int dig, loop, f[3];
    bool sync_after;  // a workgroup barrier follows the stage (else: the next stage is wave-local)
};
constexpr SweepGeom sweep_geom(const int K, const int J) {
    if (K == 5) {
        switch (J) {
            case 0: return {4, 1, {0, 3, 2}, false};
            case 1: return {3, 1, {0, 4, 2}, true};
            case 2: return {2, 4, {0, 1, 3}, false};
            case 3: return {1, 4, {0, 2, 3}, !MIBN_SWEEP_TAIL_LOCAL};  // (stages 2, 3 AND 4 give a wave the same cells: d3, a pair of d4)
            default: return {0, 4, {1, 2, 3}, true};
        }
    }
    if (K == 4) {
        switch (J) {
            case 0: return {3, 0, {2, 1, 7}, false};
            case 1: return {2, 0, {3, 1, 7}, true};
            case 2: return {1, 2, {0, 3, 7}, false};
            default: return {0, 2, {1, 3, 7}, true};
        }
    }
    if (K == 3) {
        switch (J) {  // one field, a wave = all R pairs of one (field, half): no two stages share their cells
            case 0: return {2, 0, {1, 7, 7}, true};
            case 1: return {1, 2, {0, 7, 7}, true};
            default: return {0, 2, {1, 7, 7}, true};
        }
    }
    return J == 0 ? SweepGeom{1, 0, {7, 7, 7}, true} : SweepGeom{0, 1, {7, 7, 7}, true};  // K == 2: no field at all
}

// cell index of a lane's first cell in stage (K, J): lane bits = rp, the fields in order, the half of the loop digit
template <int K, int J>
__device__ __forceinline__ int sweep_pair_base(const int tid) {
    constexpr int RB = 13 - 2 * K;
    constexpr SweepGeom G = sweep_geom(K, J);
    int base = 2 * (tid & ((1 << (RB - 1)) - 1));
    int bits = tid >> (RB - 1);
#pragma unroll
    for (int q = 0; q < K - 2; ++q) {
        base += (bits & 3) << (RB + 2 * G.f[q]);
        bits >>= 2;
    }
    return base + ((bits & 1) << (RB + 2 * G.loop + 1));  // loop values 2 h, 2 h + 1
}

// both R cells of the lane's two loop values: out[n] = sum_x f[x] * t[n + COUT x], the FMA order of sweep_fibers.
// PAR: bits 0-1 of r are a ctrl value of this stage - the odd cell takes the next T slice (par_ts cells further on); the two
// cells are then reduced one after the other through the same T registers (a second slice in registers spills at 128 VGPRs).
template <int COUT, int SX, int SL, bool PAR>
__device__ __forceinline__ void sweep_fiber_pairs_impl(double *__restrict__ L, const double *__restrict__ T, const int base, const int toff,
                                                       const int loop_ts, const int par_ts, const int nl) {
    constexpr int NT = 4 * COUT;
    double t0[NT];
#pragma unroll
    for (int l = 0; l < 2; ++l) {
        if (l >= nl) break;  // (nl = 1: the loop digit has died - only its value 0 carries anything.  A constant 2 everywhere else)
        if (PAR || l == 0 || loop_ts) {  // (uniform)
            const double2 *__restrict__ Tp = reinterpret_cast<const double2 *>(T + (toff + l * loop_ts));
#pragma unroll
            for (int q = 0; q < NT / 2; ++q) { const double2 v = Tp[q]; t0[2 * q] = v.x; t0[2 * q + 1] = v.y; }
        }
        double2 *__restrict__ Lp = reinterpret_cast<double2 *>(L + (base + l * SL));
        double2 f[4];
#pragma unroll
        for (int x = 0; x < 4; ++x) f[x] = Lp[x * (SX / 2)];
        if constexpr (PAR) {
            double s0[COUT], s1[COUT];
#pragma unroll
            for (int n = 0; n < COUT; ++n) {
                s0[n] = f[0].x * t0[n];
#pragma unroll
                for (int x = 1; x < 4; ++x) s0[n] = __builtin_fma(f[x].x, t0[n + COUT * x], s0[n]);
            }
            const double2 *__restrict__ Tq = reinterpret_cast<const double2 *>(T + (toff + l * loop_ts + par_ts));
#pragma unroll
            for (int q = 0; q < NT / 2; ++q) { const double2 v = Tq[q]; t0[2 * q] = v.x; t0[2 * q + 1] = v.y; }
#pragma unroll
            for (int n = 0; n < COUT; ++n) {
                s1[n] = f[0].y * t0[n];
#pragma unroll
                for (int x = 1; x < 4; ++x) s1[n] = __builtin_fma(f[x].y, t0[n + COUT * x], s1[n]);
            }
#pragma unroll
            for (int n = 0; n < COUT; ++n) Lp[n * (SX / 2)] = make_double2(s0[n], s1[n]);
        } else {
#pragma unroll
            for (int n = 0; n < COUT; ++n) {
                double s0 = f[0].x * t0[n], s1 = f[0].y * t0[n];
#pragma unroll
                for (int x = 1; x < 4; ++x) {
                    s0 = __builtin_fma(f[x].x, t0[n + COUT * x], s0);
                    s1 = __builtin_fma(f[x].y, t0[n + COUT * x], s1);
                }
                Lp[n * (SX / 2)] = make_double2(s0, s1);
            }
        }
    }
}
template <int COUT, int SX, int SL>
__device__ __forceinline__ void sweep_fiber_pairs(double *__restrict__ L, const double *__restrict__ T, const int base, const int toff,
                                                  const int loop_ts, const int par_ts, const int nl = 2) {
    // (uniform.  A straight-line variant for loop_ts = 0 - all sixteen reads of a lane in flight, then the FMAs, then the
    //  writes - was measured: no faster with one workgroup per CU, spills at the 128-VGPR budget of two)
    if (par_ts) sweep_fiber_pairs_impl<COUT, SX, SL, true>(L, T, base, toff, loop_ts, par_ts, nl);
    else sweep_fiber_pairs_impl<COUT, SX, SL, false>(L, T, base, toff, loop_ts, 0, nl);
}

// The last stage of a wave-owned tail (sweep_tiles_dma, OWN): the results go from the registers straight to the output
// block - no LDS write, no readout pass - and the moment the wave's reads of the stage have returned its cells are free:
// the DMA of the NEXT tile is issued before the FMAs and lands under them and the stores.  Lane bits, low to high: the first
// field d1 (four lanes = the 4 x 4 outputs of a 128-byte line: n = digit 0, then d1), rp, the other fields, the half of the
// loop digit.  The 4-way bank conflict of the reads (only rp spreads the lanes of a group) is the one this stage had anyway.
// Vector-memory instructions a lane issues BEHIND dma_next() in sweep_last_stage_out: two loop-digit halves x four unconditional
// 16-byte stores.  sweep_tiles_dma's wait for the next tile's DMA is `s_waitcnt vmcnt(kSweepOwnTailStores)` - correct only while
// (a) exactly this many VMEM operations follow the DMA in program order, none of them predicated away, and (b) the vector-memory
// operations of a wave retire in order (gfx9: loads and stores share one in-order vmcnt).  Both uses name this constant.
constexpr int kSweepOwnTailHalves = 2, kSweepOwnTailStoresPerHalf = 4;
constexpr int kSweepOwnTailStores = kSweepOwnTailHalves * kSweepOwnTailStoresPerHalf;

template <int K, bool PAR, class DmaNext>
__device__ __forceinline__ void sweep_last_stage_out(const double *__restrict__ L, const double *__restrict__ T, double *__restrict__ ot,
                                                     const int tid, const int rg_tile, const uint32_t s1, const uint32_t (&cw)[3],
                                                     const int nctrl, DmaNext dma_next) {
    constexpr int RB = 13 - 2 * K, RT = 1 << RB;
    constexpr SweepGeom G = sweep_geom(K, K - 1);  // dig 0, fields (1, ..), loop
    static_assert(G.dig == 0 && G.f[0] == 1, "last stage: digit 0 contracted, first field digit 1");
    const int d1 = tid & 3, rp = (tid >> 2) & ((RT >> 1) - 1);
    int bits = tid >> (RB + 1);
    int base = 2 * rp + (d1 << (RB + 2));
    int c_hi = 4 * d1;  // output cell without digit 0 and r
#pragma unroll
    for (int q = 1; q < K - 2; ++q) {
        base += (bits & 3) << (RB + 2 * G.f[q]);
        c_hi += (bits & 3) << (2 * G.f[q]);
        bits >>= 2;
    }
    const int h = bits & 1;
    base += (2 * h) << (RB + 2 * G.loop);
    c_hi += (2 * h) << (2 * G.loop);
    int toff = (int)(s1 & 0xffff), loop_ts = 0, par_ts = 0;
    const int rg = rg_tile + 2 * rp;
#pragma unroll
    for (int c = 0; c < 3; ++c)
        if (c < nctrl) {
            const int src = cw[c] & 0xff, ts = (int)(cw[c] >> 8);
            if (src >= 8) { toff += ((rg >> (src - 8)) & 3) * ts; if (src == 8) par_ts = ts; }
            else {
                toff += ((base >> (RB + 2 * src)) & 3) * ts;
                if (src == G.loop) loop_ts = ts;
            }
        }
    constexpr int SX = RT, SL = RT << (2 * G.loop);  // digit 0: stride Rt
    const double2 *__restrict__ Lp = reinterpret_cast<const double2 *>(L + base);
    double2 f[2][4];
#pragma unroll
    for (int l = 0; l < 2; ++l)
#pragma unroll
        for (int x = 0; x < 4; ++x) f[l][x] = Lp[(l * SL + x * SX) / 2];
    // the wave's cells have been read (the fence waits for the LDS): refill them
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // (the data is in the registers before the DMA may overwrite the cells)
    dma_next();
    double t0[16];
#pragma unroll
    for (int l = 0; l < kSweepOwnTailHalves; ++l) {
        double s0[4], s1v[4];
        if (PAR || l == 0 || loop_ts) {
            const double2 *__restrict__ Tp = reinterpret_cast<const double2 *>(T + (toff + l * loop_ts));
#pragma unroll
            for (int q = 0; q < 8; ++q) { const double2 v = Tp[q]; t0[2 * q] = v.x; t0[2 * q + 1] = v.y; }
        }
#pragma unroll
        for (int n = 0; n < 4; ++n) {
            s0[n] = f[l][0].x * t0[n];
#pragma unroll
            for (int x = 1; x < 4; ++x) s0[n] = __builtin_fma(f[l][x].x, t0[n + 4 * x], s0[n]);
        }
        if constexpr (PAR) {
            const double2 *__restrict__ Tq = reinterpret_cast<const double2 *>(T + (toff + l * loop_ts + par_ts));
#pragma unroll
            for (int q = 0; q < 8; ++q) { const double2 v = Tq[q]; t0[2 * q] = v.x; t0[2 * q + 1] = v.y; }
        }
#pragma unroll
        for (int n = 0; n < 4; ++n) {
            s1v[n] = f[l][0].y * t0[n];
#pragma unroll
            for (int x = 1; x < 4; ++x) s1v[n] = __builtin_fma(f[l][x].y, t0[n + 4 * x], s1v[n]);
        }
        // output cell: digits as in LDS (identity), r slowest: c = n + c_hi + (l << 2 loop) + 4^K r
        double *__restrict__ o0 = ot + (c_hi + (l << (2 * G.loop)) + ((2 * rp) << (2 * K)));
        double *__restrict__ o1 = o0 + (1 << (2 * K));
        static_assert(kSweepOwnTailStoresPerHalf == 4, "the four unconditional stores below are what vmcnt(kSweepOwnTailStores) counts");
        *reinterpret_cast<double2 *>(o0) = make_double2(s0[0], s0[1]);
        *reinterpret_cast<double2 *>(o0 + 2) = make_double2(s0[2], s0[3]);
        *reinterpret_cast<double2 *>(o1) = make_double2(s1v[0], s1v[1]);
        *reinterpret_cast<double2 *>(o1 + 2) = make_double2(s1v[2], s1v[3]);
    }
}

// The wave-owned tail of a five-variable pass in which ONE digit dies (kout = 4: a stage with cout = 1 - the variable of that digit has
// no successor in the frontier - 11 % of the C3 bytes).  Round 3 sent these steps through the readout path (a last stage that writes
// LDS, a workgroup barrier, a readout pass, another barrier, then the next tile's DMA with its whole latency exposed: 10.6 us per
// 80 KiB tile, 3.8 TB/s in isolation - tools/ubench/sweep_real.hip with a dying stage, profiles/r04_e_sweep_dead.log).  Here the last
// stage has the same geometry, the same refill of the wave's own cells and the same register-to-memory stores as
// sweep_last_stage_out; what differs is which lanes hold anything (a stage with cout = 1 leaves its result where the digit is 0:
// only lanes - for digit 3 waves, for digit 4 the first of the two loop values - whose dead digit is 0 store) and where a cell
// goes: the surviving digits are packed in ascending order, cell = sum of value(d) << pos(d) + (r << 8).  dead = 0: the last stage
// itself has cout = 1 and a lane stores one value per cell instead of four.  Same FMA order per output cell as every other path.
template <bool PAR, class DmaNext>
__device__ __forceinline__ void sweep_last_stage_out_dead(const double *__restrict__ L, const double *__restrict__ T, double *__restrict__ ot,
                                                          const int tid, const int rg_tile, const uint32_t s1, const uint32_t (&cw)[3],
                                                          const int nctrl, const int dead, DmaNext dma_next) {
    constexpr int K = 5, RB = 3, RT = 8;
    constexpr SweepGeom G = sweep_geom(K, K - 1);  // dig 0, fields (1, 2, 3), loop 4
    static_assert(G.dig == 0 && G.f[0] == 1 && G.f[1] == 2 && G.f[2] == 3 && G.loop == 4, "last stage of a five-variable pass");
    const int d1 = tid & 3, rp = (tid >> 2) & 3, d2 = (tid >> 4) & 3, d3 = (tid >> 6) & 3, h = (tid >> 8) & 1;
    const int base = 2 * rp + (d1 << (RB + 2)) + (d2 << (RB + 4)) + (d3 << (RB + 6)) + ((2 * h) << (RB + 8));
    // position of digit d in the packed output cell: digits below the dead one keep 2 d, digits above it move down to 2 (d - 1)
    // (the dead digit's value is 0 wherever a lane stores: its own shift never matters)
    const int q1 = 1 > dead ? 0 : 2, q2 = 2 > dead ? 2 : 4, q3 = 3 > dead ? 4 : 6, q4 = 4 > dead ? 6 : 8;
    int toff = (int)(s1 & 0xffff), loop_ts = 0, par_ts = 0;
    const int rg = rg_tile + 2 * rp;
#pragma unroll
    for (int c = 0; c < 3; ++c)
        if (c < nctrl) {
            const int src = cw[c] & 0xff, ts = (int)(cw[c] >> 8);
            if (src >= 8) { toff += ((rg >> (src - 8)) & 3) * ts; if (src == 8) par_ts = ts; }
            else {
                toff += ((base >> (RB + 2 * src)) & 3) * ts;
                if (src == G.loop) loop_ts = ts;
            }
        }
    constexpr int SX = RT, SL = RT << (2 * G.loop);
    const double2 *__restrict__ Lp = reinterpret_cast<const double2 *>(L + base);
    double2 f[2][4];
#pragma unroll
    for (int l = 0; l < 2; ++l)
#pragma unroll
        for (int x = 0; x < 4; ++x) f[l][x] = Lp[(l * SL + x * SX) / 2];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // (the data is in the registers before the DMA may overwrite the cells)
    dma_next();
    const bool lane_on = dead == 1 ? d1 == 0 : (dead == 2 ? d2 == 0 : (dead == 3 ? d3 == 0 : (dead == 4 ? h == 0 : true)));
    const int c_hi = (d1 << q1) + (d2 << q2) + (d3 << q3) + ((2 * h) << q4) + ((2 * rp) << 8);
    if (dead != 0) {
        if (!lane_on) return;  // (after the wave's reads and its DMA: nothing of this lane's cells is an answer)
        double t0[16];
#pragma unroll
        for (int l = 0; l < 2; ++l) {
            if (dead == 4 && l == 1) break;
            double s0[4], s1v[4];
            if (PAR || l == 0 || loop_ts) {
                const double2 *__restrict__ Tp = reinterpret_cast<const double2 *>(T + (toff + l * loop_ts));
#pragma unroll
                for (int q = 0; q < 8; ++q) { const double2 v = Tp[q]; t0[2 * q] = v.x; t0[2 * q + 1] = v.y; }
            }
#pragma unroll
            for (int n = 0; n < 4; ++n) {
                s0[n] = f[l][0].x * t0[n];
#pragma unroll
                for (int x = 1; x < 4; ++x) s0[n] = __builtin_fma(f[l][x].x, t0[n + 4 * x], s0[n]);
            }
            if constexpr (PAR) {
                const double2 *__restrict__ Tq = reinterpret_cast<const double2 *>(T + (toff + l * loop_ts + par_ts));
#pragma unroll
                for (int q = 0; q < 8; ++q) { const double2 v = Tq[q]; t0[2 * q] = v.x; t0[2 * q + 1] = v.y; }
            }
#pragma unroll
            for (int n = 0; n < 4; ++n) {
                s1v[n] = f[l][0].y * t0[n];
#pragma unroll
                for (int x = 1; x < 4; ++x) s1v[n] = __builtin_fma(f[l][x].y, t0[n + 4 * x], s1v[n]);
            }
            {
                double *__restrict__ o0 = ot + (c_hi + (l << q4));
                double *__restrict__ o1 = o0 + (1 << 8);
                *reinterpret_cast<double2 *>(o0) = make_double2(s0[0], s0[1]);
                *reinterpret_cast<double2 *>(o0 + 2) = make_double2(s0[2], s0[3]);
                *reinterpret_cast<double2 *>(o1) = make_double2(s1v[0], s1v[1]);
                *reinterpret_cast<double2 *>(o1 + 2) = make_double2(s1v[2], s1v[3]);
            }
        }
    } else {
        // the last stage's own digit dies: T_j[x + 4 ctrl], one value per cell
        double t0[4];
#pragma unroll
        for (int l = 0; l < 2; ++l) {
            if (PAR || l == 0 || loop_ts) {
                const double2 *__restrict__ Tp = reinterpret_cast<const double2 *>(T + (toff + l * loop_ts));
#pragma unroll
                for (int q = 0; q < 2; ++q) { const double2 v = Tp[q]; t0[2 * q] = v.x; t0[2 * q + 1] = v.y; }
            }
            double s0 = f[l][0].x * t0[0];
#pragma unroll
            for (int x = 1; x < 4; ++x) s0 = __builtin_fma(f[l][x].x, t0[x], s0);
            if constexpr (PAR) {
                const double2 *__restrict__ Tq = reinterpret_cast<const double2 *>(T + (toff + l * loop_ts + par_ts));
#pragma unroll
                for (int q = 0; q < 2; ++q) { const double2 v = Tq[q]; t0[2 * q] = v.x; t0[2 * q + 1] = v.y; }
            }
            double s1v = f[l][0].y * t0[0];
#pragma unroll
            for (int x = 1; x < 4; ++x) s1v = __builtin_fma(f[l][x].y, t0[x], s1v);
            double *__restrict__ o0 = ot + (c_hi + (l << q4));
            o0[0] = s0;
            o0[1 << 8] = s1v;
        }
    }
}

// Readout order.  Piece e (16 bytes = output cells c, c + 1) of a tile's output block: lanes e = trip * 512 + tid.  With the
// identity map (c = 2 e) a wave's lanes differ in digit values only, whose LDS strides are multiples of 32 cells: all
// 32 lanes of a ds_read_b64 group on one bank pair (1.6 us per tile).  Instead the five low bits of e are: two bits of the
// cell index (four lanes = 64 contiguous output bytes) and the three low bits of r - whose LDS stride is one cell - and the
// other bits follow in ascending order; for k = 5 a group then covers 16 of the 32 bank pairs.  kout < 2: identity.
__device__ __forceinline__ int sweep_readout_cell(const int e, const int kout) {
    if (kout < 2) return 2 * e;
    const int mr = 2 * kout - 1;  // position of r bit 0 in m = c / 2
    const int lo = e & 3, r3 = (e >> 2) & 7, hi = e >> 5;
    const int mid = hi & ((1 << (mr - 2)) - 1), top = hi >> (mr - 2);
    return 2 * (lo | (mid << 2) | (r3 << mr) | (top << (mr + 3)));
}

// The tiles [t_begin, t_end) of one work item.  K > 0: canonical step of K variables (compile-time stage geometry, 16-byte
// LDS accesses); K = 0: any step (runtime strides, 8-byte accesses).  OWN (K = 5 / 4, every digit survives in place): the
// wave-owned tail.  The last stage gives a wave the cells {one value fa of a digit, the pair 2 h, 2 h + 1 of another one} =
// two runs of 512 consecutive LDS cells (K = 5: run rho = d3 + 4 d4, fa = d3, pair = d4; K = 4: rho = d2 + 4 d3, fa = d3,
// pair = d2); the DMA fills the same cells per wave, so nothing between the barrier after stage K - 2 and the landing of
// the next tile needs the other waves.
// LOADER (round 6, tools/ubench/sweep_real.hip only - VERDICT r5 item 2): a NINTH wave issues every tile's LDS-DMA into a ring of two
// 64 KiB slots while the eight stage waves work on the other slot - the fill decoupled from the stage waves (one workgroup per CU:
// 137.5 KiB of LDS).  The loader joins the tile's two workgroup barriers: behind the first (the stage waves have left the previous
// tile's slot) it issues the next tile's 64 DMA instructions, behind the second it waits for them.
template <int K, bool OWN, bool DEAD = false, bool LOADER = false>
__device__ __forceinline__ void sweep_tiles_dma(double *__restrict__ L, double *__restrict__ T, const uint32_t *stw, const int k_rt,
                                                const int rb_rt, const double *__restrict__ F, double *__restrict__ outp,
                                                const long Rcells, const int t_begin, const int t_end, const int kout,
                                                const uint32_t surv, const int tid, const SweepTables &tb, const int dead = -1) {
    static_assert(!DEAD || (OWN && K == 5), "one dead digit: the wave-owned tail of a five-variable pass");
    constexpr int WG = kSweepWG;
    constexpr int KS = K ? K : 5;
    constexpr int PER = kSweepTileCells / 2 / WG;  // 16-byte pieces per lane and tile
    const int k = K ? K : k_rt, rb = K ? 13 - 2 * K : rb_rt;
    const int Rt = 1 << rb;
    // (the stage records are re-read from the descriptor per tile and stage and a lane's cell index re-derived from its id:
    //  kept across the tile loop they cost 25 scalar and 10 vector registers - spills at the 128-VGPR budget)
    const int ocells = Rt << (2 * kout);
    // readout (not OWN): piece e = q * WG + tid of the output block = cells c, c + 1 (sweep_readout_cell); the bit deposit is
    // linear in e, so a lane keeps its own part and adds a uniform part per trip
    const int c_lane = sweep_readout_cell(tid, kout);
    const int p_lane = sweep_perm(c_lane, kout, rb, surv);
    const int st0 = kout > 0 ? 1 << (rb + 2 * (int)(surv & 15)) : 1;
    const int c_wave = uni(sweep_readout_cell(tid & ~63, kout));  // the lowest cell of the wave (the deposit is monotonic in e)
    // DMA: a wave instruction fills 1 KiB = 128 consecutive LDS cells (lane l the cells lambda + 2 l, + 1); LDS cell lambda is
    // cell (lambda & (Rt - 1)) of combination lambda >> rb.  Default: instruction i of wave w fills the cells from 1024 i + 128 w.
    const int lane = tid & 63, wv = tid >> 6;
    const int own_rho0 = K == 5 ? (wv & 3) + 8 * (wv >> 2) : 2 * (wv >> 2) + 4 * (wv & 3);  // run of the pair's first value; the second one:
    constexpr int kOwnRhoStep = K == 5 ? 4 : 1;                                             // + this
    const int rp = tid & ((Rt >> 1) - 1);
    const long g_tid = OWN ? (long)((512 * own_rho0 + 2 * lane) >> rb) * Rcells + ((2 * lane) & (Rt - 1))
                           : (long)(tid >> (rb - 1)) * Rcells + 2 * rp;
    const long g_step = (long)(WG >> (rb - 1)) * Rcells;
    const uint32_t lds_l = lds_byte_addr(L);
    auto dma_tile = [&](const int tile) {
        long g_lane = g_tid;
        if constexpr (!OWN) {
            // (the readout paths are at their register budget: the lane's part of the address, kept across the tile loop, was the
            //  kernel's only spill - and its reload, with the s_waitcnt vmcnt(0) the compiler puts behind it, sat in front of every
            //  tile's DMA.  Re-derived per tile from an opaque copy of the lane id instead: a handful of integer operations)
            int t2 = tid;
            asm volatile("" : "+v"(t2));
            g_lane = (long)(t2 >> (rb - 1)) * Rcells + 2 * (t2 & ((Rt >> 1) - 1));
        }
        const double *__restrict__ Ft = F + (long)tile * Rt + g_lane;
        if constexpr (OWN) {
            const uint32_t lb = (uint32_t)uni((int)(lds_l + 8u * 512u * (uint32_t)own_rho0));
#pragma unroll
            for (int i = 0; i < PER; ++i) {
                const int lam = 512 * kOwnRhoStep * (i >> 2) + 128 * (i & 3);  // (compile-time)
                dma16(Ft + (long)(lam >> rb) * Rcells, lb + (uint32_t)(8 * lam));
            }
        } else {
            const uint32_t lb = (uint32_t)uni((int)(lds_l + 16u * (uint32_t)(tid & ~63)));
#pragma unroll
            for (int i = 0; i < PER; ++i) dma16(Ft + i * g_step, lb + (uint32_t)(i * WG * 16));
        }
    };
    const int n_tiles = t_end - t_begin;
    double *__restrict__ const Lbase = L;
    if constexpr (LOADER) {
        static_assert(!LOADER || (K == 5 && OWN && !DEAD), "the loader experiment covers the canonical five-variable pass");
        if (tid >= WG) {  // the loader wave: instruction j fills LDS cells [128 j, 128 j + 128) of the slot (lane l: cells 128 j + 2 l, + 1)
            auto fill = [&](const int tile, const int slot) {
                const uint32_t lb = (uint32_t)uni((int)(lds_l + 8u * (uint32_t)(slot * kSweepTileCells)));
                const double *__restrict__ Ft = F + (long)tile * Rt + ((2 * lane) & (Rt - 1));
#pragma unroll 8
                for (int j = 0; j < kSweepTileCells / 128; ++j) dma16(Ft + (long)((128 * j + 2 * lane) >> rb) * Rcells, lb + (uint32_t)(8 * 128 * j));
            };
            fill(t_begin, 0);
            for (int i = 0; i < n_tiles; ++i) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __syncthreads();
                if (i + 1 < n_tiles) fill(t_begin + i + 1, (i + 1) & 1);
                __syncthreads();
            }
            return;
        }
    }
    MIBN_PROF_INIT
    if constexpr (!LOADER) dma_tile(t_begin);
    sweep_build_tables(tb, tid);  // (its loads retire behind the first tile's: the tile has landed when T is built)
    for (int i = 0; i < n_tiles; ++i) {
        const int tile = t_begin + i;
        if constexpr (LOADER) L = Lbase + (i & 1) * kSweepTileCells;
        MIBN_PROF_TICK(0)
        // this wave's share of the tile has landed.  Wave-owned tail: the DMA was issued BEFORE the eight 16-byte stores of the
        // last stage (sweep_last_stage_out) and the vector-memory operations of a wave retire in order, so "at most eight
        // operations outstanding" means the DMA is done - the wave does not sit out the write acknowledgements of its stores
        // (DEAD: a wave may have stored less - or nothing - after its DMA: it drains)
        static_assert(MIBN_SWEEP_VMCNT == 0 || MIBN_SWEEP_VMCNT == kSweepOwnTailStores, "the wait must leave exactly the tail's stores in flight");
        if constexpr (LOADER) {}  // (the stage waves issue no loads: the loader has waited for the tile)
        else if (OWN && !DEAD && MIBN_SWEEP_VMCNT == kSweepOwnTailStores && i > 0) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(kSweepOwnTailStores) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        MIBN_PROF_TICK(1)
        __syncthreads();  // ... and everybody else's (first tile: T is complete)
        const int rg = tile * Rt + (K ? 2 * rp : (tid & (Rt - 1)));  // this lane's (even) R cell
#define MIBN_SWEEP_STAGE(J)                                                                                                   \
        if (J < KS && J < k && !(OWN && J == K - 1)) {                                                                        \
            const uint32_t s0 = (uint32_t)uni((int)stw[J * kSweepStageWords]), s1 = (uint32_t)uni((int)stw[J * kSweepStageWords + 1]); \
            uint32_t cw[3];                                                                                                   \
            _Pragma("unroll") for (int c = 0; c < 3; ++c) cw[c] = (uint32_t)uni((int)stw[J * kSweepStageWords + 2 + c]);      \
            constexpr SweepGeom G = sweep_geom(K > J ? K : 5, K > J ? J : 0);                                                 \
            const int cout = (s0 >> 4) & 15, nctrl = (s0 >> 12) & 15, loop = K > J ? G.loop : (int)((s0 >> 16) & 15);         \
            int bs, toff, loop_ts = 0, par_ts = 0;                                                                            \
            if constexpr (K > J) {  /* pair mapping: the digit values the T offset depends on come out of the cell index */   \
                bs = sweep_pair_base<(K > J ? K : 5), (K > J ? J : 0)>(tid);                                                  \
                toff = (int)(s1 & 0xffff);                                                                                    \
                _Pragma("unroll") for (int c = 0; c < 3; ++c)                                                                 \
                    if (c < nctrl) {                                                                                          \
                        const int src = cw[c] & 0xff, ts = (int)(cw[c] >> 8);                                                 \
                        if (src < 8) toff += ((bs >> (rb + 2 * src)) & 3) * ts;  /* (the loop digit: 2 h) */                  \
                    }                                                                                                         \
            } else {                                                                                                          \
                sweep_stage_lane(s0, s1, cw, rb, tid, bs, toff);                                                              \
            }                                                                                                                 \
            _Pragma("unroll") for (int c = 0; c < 3; ++c)                                                                     \
                if (c < nctrl) {                                                                                              \
                    const int src = cw[c] & 0xff, ts = (int)(cw[c] >> 8);                                                     \
                    if (src >= 8) { toff += ((rg >> (src - 8)) & 3) * ts; if (src == 8) par_ts = ts; }                        \
                    else if (src == loop) loop_ts = ts;                                                                       \
                }                                                                                                             \
            if constexpr (K > J) {                                                                                            \
                constexpr int kRt = 1 << (13 - 2 * (K ? K : 5));                                                               \
                constexpr int SX = kRt << (2 * G.dig), SL = kRt << (2 * G.loop);                                              \
                /* DEAD: a stage behind the one in which the digit died (its digit is below the dead one) works on garbage     \
                   wherever the dead digit is not 0 - three quarters of its fibers.  Those lanes (whole waves where the dead    \
                   digit is a wave-level field; the second loop value where it is the loop digit) sit the stage out */          \
                bool alive = true;                                                                                            \
                int nl = 2;                                                                                                   \
                if constexpr (DEAD) {                                                                                         \
                    if (MIBN_SWEEP_DEAD_SKIP && G.dig < dead) {                                                               \
                        if (dead == G.loop) { alive = ((bs >> (rb + 2 * G.loop + 1)) & 1) == 0; nl = 1; }                     \
                        else alive = ((bs >> (rb + 2 * dead)) & 3) == 0;                                                      \
                    }                                                                                                         \
                }                                                                                                             \
                if (alive) {                                                                                                  \
                    if (cout == 4) sweep_fiber_pairs<4, SX, SL>(L, T, bs, toff, loop_ts, par_ts, nl);                         \
                    else sweep_fiber_pairs<1, SX, SL>(L, T, bs, toff, loop_ts, par_ts, nl);                                   \
                }                                                                                                             \
                /* (wave-owned tail: stage K - 2 and the last stage give a wave the same cells - sweep_geom - so that hand-over \
                    is wave-local too: two workgroup barriers per tile, after the landing and after stage 1 / K - 3) */        \
                if constexpr (G.sync_after && !(MIBN_SWEEP_TAIL_LOCAL && OWN && J == K - 2)) __syncthreads();                 \
                else {  /* wave-local hand-over: order the wave's own LDS writes before its reads of the next stage */        \
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");                                                    \
                    __builtin_amdgcn_wave_barrier();                                                                          \
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");                                                    \
                }                                                                                                             \
            } else {                                                                                                          \
                const int sx = 1 << (rb + 2 * (int)(s0 & 15)), sl = 1 << (rb + 2 * loop);                                     \
                if (cout == 4) sweep_fibers<4, 0, 0>(L, T, bs, sx, sl, toff, loop_ts);                                        \
                else sweep_fibers<1, 0, 0>(L, T, bs, sx, sl, toff, loop_ts);                                                  \
                __syncthreads();                                                                                              \
            }                                                                                                                 \
            MIBN_PROF_TICK(2 + J)                                                                                             \
        }
        MIBN_SWEEP_STAGE(0)
        MIBN_SWEEP_STAGE(1)
        MIBN_SWEEP_STAGE(2)
        MIBN_SWEEP_STAGE(3)
        MIBN_SWEEP_STAGE(4)
#undef MIBN_SWEEP_STAGE
        double *__restrict__ ot = outp + (long)tile * ocells;
        if constexpr (OWN) {
            // the last stage writes the output block itself and refills the wave's cells (sweep_last_stage_out)
            constexpr int JL = K > 0 ? K - 1 : 0;
            const uint32_t s1 = (uint32_t)uni((int)stw[JL * kSweepStageWords + 1]);
            const int nctrl = (int)(((uint32_t)uni((int)stw[JL * kSweepStageWords]) >> 12) & 15);
            uint32_t cw[3];
#pragma unroll
            for (int c = 0; c < 3; ++c) cw[c] = (uint32_t)uni((int)stw[JL * kSweepStageWords + 2 + c]);
            bool par = false;
#pragma unroll
            for (int c = 0; c < 3; ++c) par = par || (c < nctrl && (cw[c] & 0xff) == 8);
            auto dma_next = [&]() { if constexpr (!LOADER) { if (i + 1 < n_tiles) dma_tile(tile + 1); } };
            if constexpr (DEAD) {
                if (par) sweep_last_stage_out_dead<true>(L, T, ot, tid, tile * Rt, s1, cw, nctrl, dead, dma_next);  // (uniform)
                else sweep_last_stage_out_dead<false>(L, T, ot, tid, tile * Rt, s1, cw, nctrl, dead, dma_next);
            } else {
                if (par) sweep_last_stage_out<(K > 0 ? K : 5), true>(L, T, ot, tid, tile * Rt, s1, cw, nctrl, dma_next);  // (uniform)
                else sweep_last_stage_out<(K > 0 ? K : 5), false>(L, T, ot, tid, tile * Rt, s1, cw, nctrl, dma_next);
            }
            MIBN_PROF_TICK(7)
        } else {
            // the tile's output block, two cells (16 bytes) per lane and trip; a trip in which no lane of the wave has a cell is
            // skipped by a scalar branch
#pragma unroll
            for (int q = 0; q < PER; ++q) {
                const int c_trip = sweep_readout_cell(q * WG, kout);  // (uniform)
                if (c_trip + c_wave < ocells) {
                    const int c = c_trip + c_lane;
                    const int a = sweep_perm(c_trip, kout, rb, surv) + p_lane;
                    if (c < ocells) *reinterpret_cast<double2 *>(ot + c) = make_double2(L[a], L[a + st0]);
                }
            }
            MIBN_PROF_TICK(7)
            if (i + 1 < n_tiles) {
                __syncthreads();  // every wave has read the tile out
                dma_tile(tile + 1);
            }
        }
        MIBN_PROF_TICK(8)
    }
    MIBN_PROF_END
}

// 73.5 KiB of dynamic LDS like the register-staged kernel: two workgroups per CU, <= 128 VGPRs
__global__ __launch_bounds__(kSweepWG, 4) void ve_sweep_dma_kernel(const LevelArgs A) {
    extern __shared__ __attribute__((aligned(16))) double sweep_lds[];
    double *__restrict__ L = sweep_lds;
    double *__restrict__ T = sweep_lds + kSweepTileCells;
    uint32_t *sh_step = reinterpret_cast<uint32_t *>(sweep_lds + kSweepTileCells + kSweepMaxT);
    const int tid = threadIdx.x;
    const uint32_t wg = blockIdx.x + A.wg_base;
    const Item it = A.items[A.wg_item[wg]];
    double *__restrict__ slot = A.arena + A.arena_off[it.req];
    const uint32_t *p = A.prog + A.prog_off[it.req] + it.rel_off;
    const int words = (int)p[6];
    for (int i = tid; i < words; i += kSweepWG) sh_step[i] = p[i];
    __syncthreads();
    const int k = uni((int)((sh_step[0] >> 16) & 0xff)), rb = uni((int)((sh_step[0] >> 24) & 0xff));
    const int tiles = uni((int)sh_step[3]);
    const int kout = uni((int)(sh_step[7] & 0xffff)), t_total = uni((int)(sh_step[7] >> 16));
    const uint32_t surv = (uint32_t)uni((int)sh_step[8]);
    const long Rcells = (long)tiles << rb;
    const uint32_t *stw = sh_step + kHdrWords + 2;
    const uint32_t *smw = stw + k * kSweepStageWords;
    const double *__restrict__ F = slot + ((uint64_t)sh_step[kHdrWords] | ((uint64_t)sh_step[kHdrWords + 1] << 32));
    double *__restrict__ outp = slot + ((uint64_t)sh_step[4] | ((uint64_t)sh_step[5] << 32));
    const SweepTables tb{T, stw, smw, A.pool, slot, k, t_total};
    const int t_begin = (int)((wg - it.b) * it.a);
    const int t_end = min(tiles, t_begin + (int)it.a);
    const bool canon = (sh_step[1] >> 16) & kFlagSweepCanon;
    // one digit dies (kout = 4, the others stay in place): which one = the digit missing from the ascending list of survivors
    int dead = -1;
    if (MIBN_SWEEP_DEAD_TAIL && canon && k == 5 && kout == 4) {
        const uint32_t packs[5] = {0x4321u, 0x4320u, 0x4310u, 0x4210u, 0x3210u};
#pragma unroll
        for (int d = 0; d < 5; ++d)
            if (surv == packs[d]) dead = d;
    }
    if (canon && k == 5 && kout == 5 && surv == 0x43210u) sweep_tiles_dma<5, true>(L, T, stw, 5, 3, F, outp, Rcells, t_begin, t_end, kout, surv, tid, tb);
    else if (dead >= 0) sweep_tiles_dma<5, true, true>(L, T, stw, 5, 3, F, outp, Rcells, t_begin, t_end, kout, surv, tid, tb, dead);
    else if (canon && k == 5) sweep_tiles_dma<5, false>(L, T, stw, 5, 3, F, outp, Rcells, t_begin, t_end, kout, surv, tid, tb);
    else if (canon && k == 4 && kout == 4 && surv == 0x3210u) sweep_tiles_dma<4, true>(L, T, stw, 4, 5, F, outp, Rcells, t_begin, t_end, kout, surv, tid, tb);
    else if (canon && k == 4) sweep_tiles_dma<4, false>(L, T, stw, 4, 5, F, outp, Rcells, t_begin, t_end, kout, surv, tid, tb);
    else if (canon && k == 3) sweep_tiles_dma<3, false>(L, T, stw, 3, 7, F, outp, Rcells, t_begin, t_end, kout, surv, tid, tb);
    else if (canon && k == 2) sweep_tiles_dma<2, false>(L, T, stw, 2, 9, F, outp, Rcells, t_begin, t_end, kout, surv, tid, tb);
    else sweep_tiles_dma<0, false>(L, T, stw, k, rb, F, outp, Rcells, t_begin, t_end, kout, surv, tid, tb);
}

// the loader experiment as a kernel (canonical five-variable passes only; tools/ubench/sweep_real.hip): 512 stage lanes + a loader wave,
// two tile slots + T + the step descriptor = 137.5 KiB of LDS, one workgroup per CU
constexpr int kSweepLoaderWG = kSweepWG + 64;
constexpr int kSweepLoaderLdsBytes = 2 * kSweepTileCells * 8 + kSweepMaxT * 8 + kMaxStepWords * 4;
__global__ __launch_bounds__(kSweepLoaderWG) void ve_sweep_loader_kernel(const LevelArgs A) {
    extern __shared__ __attribute__((aligned(16))) double sweep_lds[];
    double *__restrict__ L = sweep_lds;
    double *__restrict__ T = sweep_lds + 2 * kSweepTileCells;
    uint32_t *sh_step = reinterpret_cast<uint32_t *>(sweep_lds + 2 * kSweepTileCells + kSweepMaxT);
    const int tid = threadIdx.x;
    const uint32_t wg = blockIdx.x + A.wg_base;
    const Item it = A.items[A.wg_item[wg]];
    double *__restrict__ slot = A.arena + A.arena_off[it.req];
    const uint32_t *p = A.prog + A.prog_off[it.req] + it.rel_off;
    const int words = (int)p[6];
    for (int i = tid; i < words; i += kSweepLoaderWG) sh_step[i] = p[i];
    __syncthreads();
    const int k = uni((int)((sh_step[0] >> 16) & 0xff)), rb = uni((int)((sh_step[0] >> 24) & 0xff));
    const int tiles = uni((int)sh_step[3]);
    const int kout = uni((int)(sh_step[7] & 0xffff)), t_total = uni((int)(sh_step[7] >> 16));
    const uint32_t surv = (uint32_t)uni((int)sh_step[8]);
    const long Rcells = (long)tiles << rb;
    const uint32_t *stw = sh_step + kHdrWords + 2;
    const uint32_t *smw = stw + k * kSweepStageWords;
    const double *__restrict__ F = slot + ((uint64_t)sh_step[kHdrWords] | ((uint64_t)sh_step[kHdrWords + 1] << 32));
    double *__restrict__ outp = slot + ((uint64_t)sh_step[4] | ((uint64_t)sh_step[5] << 32));
    const SweepTables tb{T, stw, smw, A.pool, slot, k, t_total};
    const int t_begin = (int)((wg - it.b) * it.a);
    const int t_end = min(tiles, t_begin + (int)it.a);
    const bool canon = (sh_step[1] >> 16) & kFlagSweepCanon;
    if (canon && k == 5 && kout == 5 && surv == 0x43210u) sweep_tiles_dma<5, true, false, true>(L, T, stw, 5, 3, F, outp, Rcells, t_begin, t_end, kout, surv, tid, tb);
    (void)rb;
}

__global__ __launch_bounds__(kSweepWG, 4) void ve_sweep_kernel(const LevelArgs A) {
    extern __shared__ __attribute__((aligned(16))) double sweep_lds[];
    double *__restrict__ L = sweep_lds;
    double *__restrict__ T = sweep_lds + kSweepTileCells;
    uint32_t *sh_step = reinterpret_cast<uint32_t *>(sweep_lds + kSweepTileCells + kSweepMaxT);
    const int tid = threadIdx.x;
    const uint32_t wg = blockIdx.x + A.wg_base;
    const Item it = A.items[A.wg_item[wg]];
    double *__restrict__ slot = A.arena + A.arena_off[it.req];
    const uint32_t *p = A.prog + A.prog_off[it.req] + it.rel_off;
    const int words = (int)p[6];
    for (int i = tid; i < words; i += kSweepWG) sh_step[i] = p[i];
    __syncthreads();
    const int k = uni((int)((sh_step[0] >> 16) & 0xff)), rb = uni((int)((sh_step[0] >> 24) & 0xff));
    const int tiles = uni((int)sh_step[3]);
    const int kout = uni((int)(sh_step[7] & 0xffff)), t_total = uni((int)(sh_step[7] >> 16));
    const uint32_t surv = (uint32_t)uni((int)sh_step[8]);
    const long Rcells = (long)tiles << rb;
    const uint32_t *stw = sh_step + kHdrWords + 2;          // k stage records
    const uint32_t *smw = stw + k * kSweepStageWords;       // the small inputs, stage by stage
    const double *__restrict__ F = slot + ((uint64_t)sh_step[kHdrWords] | ((uint64_t)sh_step[kHdrWords + 1] << 32));
    double *__restrict__ outp = slot + ((uint64_t)sh_step[4] | ((uint64_t)sh_step[5] << 32));
    const SweepTables tb{T, stw, smw, A.pool, slot, k, t_total};
    const int t_begin = (int)((wg - it.b) * it.a);
    const int t_end = min(tiles, t_begin + (int)it.a);
    const bool canon = (sh_step[1] >> 16) & kFlagSweepCanon;
    if (canon && k == 5) sweep_tiles<5>(L, T, stw, F, outp, Rcells, t_begin, t_end, kout, surv, tid, tb);
    else if (canon && k == 4) sweep_tiles<4>(L, T, stw, F, outp, Rcells, t_begin, t_end, kout, surv, tid, tb);
    else if (canon && k == 3) sweep_tiles<3>(L, T, stw, F, outp, Rcells, t_begin, t_end, kout, surv, tid, tb);
    else sweep_tiles_any(L, T, stw, k, rb, F, outp, Rcells, t_begin, t_end, kout, surv, tid, tb);
}

}  // namespace mibn
