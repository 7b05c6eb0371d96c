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
            double s = f[0] * t[n];
#pragma unroll
            for (int x = 1; x < 4; ++x) s += f[x] * t[n + COUT * x];
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
    const int p_tid = sweep_perm(tid, kout, rb, surv);
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
// Round 3: the same pass with the tile DOUBLE-BUFFERED in LDS and filled by LDS-DMA (global_load_lds_dwordx4).
//
// What bounded the kernel above was not bytes (PMC traffic / algorithmic = 0.92) but memory-level parallelism: a workgroup
// had its 64 KiB of loads in flight only between the end of a tile's stages and the start of the next tile's - the next
// tile waited in 32 VGPRs per lane - and two workgroups per CU overlapped one another's phases only statistically
// (9.3 us per tile and CU at 3.6 TB/s against 5 x 0.8 us of stages).  Here ONE workgroup per CU owns two tile buffers:
// while tile i is contracted and written out of buffer i & 1, the 64 KiB of tile i + 1 stream from HBM straight into the
// other buffer - no VGPRs, no ds_write pass - and the DMA of tile i + 2 is issued the moment tile i has been read out.
// Every CU keeps 64 KiB of loads in flight all the time, the stores of tile i drain under the stages of tile i + 1, and the
// per-tile critical path is the stages alone.
//   * LDS: 2 x 64 KiB tiles + 8 KiB T + 1.5 KiB descriptor = 137.5 KiB (of 160 KiB): one workgroup per CU.
//   * LDS-DMA writes base + 16 * lane: the staged layout L[r + Rt * xc] is exactly the order in which 16-byte piece
//     c2 = i * WG + tid is laid down, only the per-lane SOURCE address is strided (runs of Rt * 8 bytes).
//   * hipcc does not count the DMA (inline asm): completion is tracked by hand with s_waitcnt vmcnt(N).  VMEM operations
//     of a wave retire in order, so "at most N outstanding" with N = the operations issued after the DMA of the tile
//     (the stores of the previous tile, the DMA of the next one) means the tile has landed; __syncthreads() then publishes
//     every wave's share.  __syncthreads() itself lowers to s_waitcnt lgkmcnt(0) + s_barrier on gfx950: it does not drain
//     the vector-memory queue, the DMA stays in flight across the stage barriers.
// The step encoding, the stage geometry and the arithmetic (order of the FMAs) are those of the kernel above: bit-identical
// results (tests/test_gpu_parity.py::test_sweep_dma_kernel_reproduces_the_register_staged_kernel).

#ifndef MIBN_PROF_INIT  // (tools/ubench/sweep_real.hip defines these to time the phases of a tile with wall_clock64)
#define MIBN_PROF_INIT
#define MIBN_PROF_TICK(k)
#define MIBN_PROF_END
#endif

constexpr int kSweepDmaLdsBytes = 2 * kSweepTileCells * 8 + kSweepMaxT * 8 + kMaxStepWords * 4;

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

// s_waitcnt vmcnt(n), n wave-uniform in [0, 16]
__device__ __forceinline__ void wait_vmcnt_le(const int n) {
#define MIBN_VMCNT(K) case K: asm volatile("s_waitcnt vmcnt(" #K ")" ::: "memory"); break;
    switch (n) {
        MIBN_VMCNT(1) MIBN_VMCNT(2) MIBN_VMCNT(3) MIBN_VMCNT(4) MIBN_VMCNT(5) MIBN_VMCNT(6) MIBN_VMCNT(7) MIBN_VMCNT(8)
        MIBN_VMCNT(9) MIBN_VMCNT(10) MIBN_VMCNT(11) MIBN_VMCNT(12) MIBN_VMCNT(13) MIBN_VMCNT(14) MIBN_VMCNT(15) MIBN_VMCNT(16)
        default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
    }
#undef MIBN_VMCNT
}

// ---- the stages.  What bounds a stage is LDS instruction issue (measured with the phase timers of tools/ubench/sweep_real.hip:
// 1.0 - 1.3 us per stage with 8-byte accesses - hipcc pairs them into ds_read2_b64 / ds_write2st64_b64, 8 and 13 LDS cycles each -
// against 0.2 us of fp64 FMAs), so the canonical steps move 16 bytes per LDS instruction: a lane owns the fibers of TWO
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
            case 3: return {1, 4, {0, 2, 3}, true};
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
    switch (J) {  // K == 3: one field, a wave = all R pairs of one (field, half): no two stages share their cells
        case 0: return {2, 0, {1, 7, 7}, true};
        case 1: return {1, 2, {0, 7, 7}, true};
        default: return {0, 2, {1, 7, 7}, true};
    }
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
                                                       const int loop_ts, const int par_ts) {
    constexpr int NT = 4 * COUT;
    double t0[NT];
#pragma unroll
    for (int l = 0; l < 2; ++l) {
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
                for (int x = 1; x < 4; ++x) s0[n] += f[x].x * t0[n + COUT * x];
            }
            const double2 *__restrict__ Tq = reinterpret_cast<const double2 *>(T + (toff + l * loop_ts + par_ts));
#pragma unroll
            for (int q = 0; q < NT / 2; ++q) { const double2 v = Tq[q]; t0[2 * q] = v.x; t0[2 * q + 1] = v.y; }
#pragma unroll
            for (int n = 0; n < COUT; ++n) {
                s1[n] = f[0].y * t0[n];
#pragma unroll
                for (int x = 1; x < 4; ++x) s1[n] += f[x].y * t0[n + COUT * x];
            }
#pragma unroll
            for (int n = 0; n < COUT; ++n) Lp[n * (SX / 2)] = make_double2(s0[n], s1[n]);
        } else {
#pragma unroll
            for (int n = 0; n < COUT; ++n) {
                double s0 = f[0].x * t0[n], s1 = f[0].y * t0[n];
#pragma unroll
                for (int x = 1; x < 4; ++x) {
                    s0 += f[x].x * t0[n + COUT * x];
                    s1 += f[x].y * t0[n + COUT * x];
                }
                Lp[n * (SX / 2)] = make_double2(s0, s1);
            }
        }
    }
}
template <int COUT, int SX, int SL>
__device__ __forceinline__ void sweep_fiber_pairs(double *__restrict__ L, const double *__restrict__ T, const int base, const int toff,
                                                  const int loop_ts, const int par_ts) {
    // (uniform.  A straight-line variant for loop_ts = 0 - all sixteen reads of a lane in flight, then the FMAs, then the
    //  writes - was measured: no faster with one workgroup per CU, spills at the 128-VGPR budget of two)
    if (par_ts) sweep_fiber_pairs_impl<COUT, SX, SL, true>(L, T, base, toff, loop_ts, par_ts);
    else sweep_fiber_pairs_impl<COUT, SX, SL, false>(L, T, base, toff, loop_ts, 0);
}

// any step: the lane mapping and the 8-byte accesses of the register-staged kernel (sweep_fibers)
template <int COUT>
__device__ __forceinline__ void sweep_fibers_rt(double *__restrict__ L, const double *__restrict__ T, const int base, const int sx,
                                                const int sl, const int toff, const int loop_ts) {
    sweep_fibers<COUT, 0, 0>(L, T, base, sx, sl, toff, loop_ts);
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

// The tiles [t_begin, t_end) of one work item, double-buffered.  K > 0: canonical step of K variables (compile-time stage
// geometry, 16-byte LDS accesses); K = 0: any step (runtime strides, 8-byte accesses).
template <int K, int NBUF>
__device__ __forceinline__ void sweep_tiles_dma(double *__restrict__ Lbuf, double *__restrict__ T, const uint32_t *stw, const int k_rt,
                                                const int rb_rt, const double *__restrict__ F, double *__restrict__ outp,
                                                const long Rcells, const int t_begin, const int t_end, const int kout,
                                                const uint32_t surv, const int tid, const SweepTables &tb) {
    constexpr int WG = kSweepWG;
    constexpr int KS = K ? K : 5;
    constexpr int PER = kSweepTileCells / 2 / WG;  // 16-byte pieces per lane and tile
    const int k = K ? K : k_rt, rb = K ? 13 - 2 * K : rb_rt;
    const int Rt = 1 << rb;
    // (the stage records are re-read from the descriptor per tile and stage and a lane's cell index re-derived from its id:
    //  kept across the tile loop they cost 25 scalar and 10 vector registers - spills at the 128-VGPR budget of two
    //  workgroups per CU)
    const int ocells = Rt << (2 * kout);
    // readout: piece e = q * WG + tid of the output block = cells c, c + 1 (sweep_readout_cell); the bit deposit is linear
    // in e, so a lane keeps its own part and adds a uniform part per trip
    const int c_lane = sweep_readout_cell(tid, kout);
    const int p_lane = sweep_perm(c_lane, kout, rb, surv);
    const int st0 = kout > 0 ? 1 << (rb + 2 * (int)(surv & 15)) : 1;
    // store instructions THIS WAVE issues per tile: a trip in which no lane of the wave has a cell is skipped by a scalar
    // branch, so that the count below is exactly what the wave's vmcnt sees
    const int c_wave = uni(sweep_readout_cell(tid & ~63, kout));  // (the lowest cell of the wave: the deposit is monotonic in e)
    int n_st = 0;
#pragma unroll
    for (int q = 0; q < PER; ++q) n_st += (sweep_readout_cell(q * WG, kout) + c_wave < ocells) ? 1 : 0;
    // DMA: piece c2 = i * WG + tid = cells (2 rp, 2 rp + 1) of combination xc
    const int rp = tid & ((Rt >> 1) - 1);
    const long g_tid = (long)(tid >> (rb - 1)) * Rcells + 2 * rp;
    const long g_step = (long)(WG >> (rb - 1)) * Rcells;
    const uint32_t lds0 = lds_byte_addr(Lbuf) + 16u * (uint32_t)(tid & ~63);
    auto dma_tile = [&](const int tile, const int buf) {
        const double *__restrict__ Ft = F + (long)tile * Rt + g_tid;
        const uint32_t lb = (uint32_t)uni((int)(lds0 + (uint32_t)buf * (kSweepTileCells * 8)));
#pragma unroll
        for (int i = 0; i < PER; ++i) dma16(Ft + i * g_step, lb + (uint32_t)(i * WG * 16));
    };
    const int n_tiles = t_end - t_begin;
    MIBN_PROF_INIT
    dma_tile(t_begin, 0);
    sweep_build_tables(tb, tid);  // (its loads retire behind the first tile's: the tile has landed when T is built)
    if (NBUF > 1 && n_tiles > 1) dma_tile(t_begin + 1, 1);
    for (int i = 0; i < n_tiles; ++i) {
        const int tile = t_begin + i;
        double *__restrict__ L = Lbuf + (NBUF > 1 ? (i & 1) * kSweepTileCells : 0);
        MIBN_PROF_TICK(0)
        // two buffers: behind this tile's DMA the wave has issued the stores of the previous tile and the DMA of the next one;
        // one buffer: nothing (the stores of the previous tile came before it)
        if (NBUF > 1) wait_vmcnt_le((i > 0 ? n_st : 0) + (i + 1 < n_tiles ? PER : 0));
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        MIBN_PROF_TICK(1)
        __syncthreads();  // every wave's share of the tile has landed (first tile: T is complete)
        if (NBUF == 1) __builtin_amdgcn_s_setprio(0);
        const int rg = tile * Rt + (K ? 2 * rp : (tid & (Rt - 1)));  // this lane's (even) R cell
#define MIBN_SWEEP_STAGE(J)                                                                                                   \
        if (J < KS && J < k) {                                                                                                \
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
                if (cout == 4) sweep_fiber_pairs<4, SX, SL>(L, T, bs, toff, loop_ts, par_ts);                                 \
                else sweep_fiber_pairs<1, SX, SL>(L, T, bs, toff, loop_ts, par_ts);                                           \
                if constexpr (G.sync_after) __syncthreads();                                                                  \
                else {  /* wave-local hand-over: order the wave's own LDS writes before its reads of the next stage */        \
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");                                                    \
                    __builtin_amdgcn_wave_barrier();                                                                          \
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");                                                    \
                }                                                                                                             \
            } else {                                                                                                          \
                const int sx = 1 << (rb + 2 * (int)(s0 & 15)), sl = 1 << (rb + 2 * loop);                                     \
                if (cout == 4) sweep_fibers_rt<4>(L, T, bs, sx, sl, toff, loop_ts);                                           \
                else sweep_fibers_rt<1>(L, T, bs, sx, sl, toff, loop_ts);                                                     \
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
        // the tile's output block, two cells (16 bytes) per lane and trip.  One buffer, two workgroups per CU: from here to the
        // landing of the next tile this workgroup has few instructions to issue - LDS reads, stores, the DMA - and all of them
        // sit on its critical path while the other workgroup is in its stages: they go first (instruction arbitration is by
        // priority, then age)
        if (NBUF == 1) __builtin_amdgcn_s_setprio(3);
        double *__restrict__ ot = outp + (long)tile * ocells;
#pragma unroll
        for (int q = 0; q < PER; ++q) {
            const int c_trip = sweep_readout_cell(q * WG, kout);  // (uniform)
            if (c_trip + c_wave < ocells) {                       // (wave-uniform: see n_st)
                const int c = c_trip + c_lane;
                const int a = sweep_perm(c_trip, kout, rb, surv) + p_lane;
                if (c < ocells) *reinterpret_cast<double2 *>(ot + c) = make_double2(L[a], L[a + st0]);
            }
        }
        MIBN_PROF_TICK(7)
        if (i + NBUF < n_tiles) {
            __syncthreads();  // every wave has read the tile out: its buffer takes the next tile (two buffers: the one after next)
            dma_tile(tile + NBUF, NBUF > 1 ? (i & 1) : 0);
        }
        MIBN_PROF_TICK(8)
    }
    MIBN_PROF_END
}

template <int NBUF>
__device__ __forceinline__ void sweep_dma_body(const LevelArgs &A) {
    extern __shared__ __attribute__((aligned(16))) double sweep_lds[];
    double *__restrict__ Lbuf = sweep_lds;
    double *__restrict__ T = sweep_lds + NBUF * kSweepTileCells;
    uint32_t *sh_step = reinterpret_cast<uint32_t *>(sweep_lds + NBUF * kSweepTileCells + kSweepMaxT);
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
    if (canon && k == 5) sweep_tiles_dma<5, NBUF>(Lbuf, T, stw, 5, 3, F, outp, Rcells, t_begin, t_end, kout, surv, tid, tb);
    else if (canon && k == 4) sweep_tiles_dma<4, NBUF>(Lbuf, T, stw, 4, 5, F, outp, Rcells, t_begin, t_end, kout, surv, tid, tb);
    else if (canon && k == 3) sweep_tiles_dma<3, NBUF>(Lbuf, T, stw, 3, 7, F, outp, Rcells, t_begin, t_end, kout, surv, tid, tb);
    else sweep_tiles_dma<0, NBUF>(Lbuf, T, stw, k, rb, F, outp, Rcells, t_begin, t_end, kout, surv, tid, tb);
}

// one tile buffer, two workgroups per CU (73.5 KiB each, <= 128 VGPRs): a workgroup's DMA latency and its store / DMA issue
// hide under the other workgroup's stages
__global__ __launch_bounds__(kSweepWG, 4) void ve_sweep_dma_kernel(const LevelArgs A) { sweep_dma_body<1>(A); }
// two tile buffers, one workgroup per CU (137.5 KiB): the next tile streams in under this tile's stages
__global__ __launch_bounds__(kSweepWG, 2) void ve_sweep_dma2_kernel(const LevelArgs A) { sweep_dma_body<2>(A); }

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
    const int Rt = 1 << rb;
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
