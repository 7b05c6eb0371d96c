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
__device__ __forceinline__ void sweep_build_tables(const SweepTables &b, const int tid) {
    for (int t = tid; t < b.t_total; t += kSweepWG) {
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
