// Gibbs sampling on gfx950: one chain per lane, all 64 chains of a wave update the SAME variable in
// the same iteration (the reference cycles deterministically through the non-evidence variables,
// sorobn/bayes_net.py:697,718-722), so there is no divergence; the Markov-blanket conditional
//     P(v | mb(v))  ~  P(v | pa(v)) * prod_{c in children(v)} P(c | pa(c))      (bayes_net.py:700-710)
// is evaluated on the fly from the dense CPTs (LDS resident when they fit, else L1/L2) instead of materialising the
// reference's per-node posterior tables (8^7 rows = 16 MiB per interior node in config 5).
// Chain state lives in LDS as state[var][lane] bytes.  Every cycle position has a precompiled *update program*
// (wave-uniform words -> scalar loads): the factors mentioning the variable (its CPT and its children's), for each
// the table offset, the variable's stride and the (other variable, stride) pairs.  An update computes each
// factor's base offset once from the chain state and then the card x n_factors table values as independent
// loads, the weights stay in registers (cards up to 16; larger cards take a two-pass loop).  The kernel is
// latency-bound by construction (config 5 = 128 chains = 2 waves per GPU): what counts is the number of dependent
// memory round trips per update.  Tables, update programs and chain state all sit in LDS when they fit, and grids take
// the FAST form - fixed-size update records fetched one iteration ahead, the state reads of all factors issued
// together, then all table reads (config 5: 2.38 -> 1.16 us per update, same counts bit for bit).  Random numbers:
// Philox4x32-10 keyed by (seed, chain), counter = update index; statistical parity only (see include/mibn.h).
#pragma once
#include <hip/hip_runtime.h>

#include <string>
#include <vector>

#include "../../include/mibn.h"
#include "planner.h"

namespace mibn {

struct GibbsVar {
    int32_t card, table_off, scope_begin, scope_len, child_begin, child_len, is_evidence, ev_code;
};

__device__ __forceinline__ void philox_round(uint32_t (&c)[4], uint32_t k0, uint32_t k1) {
    const uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u;
    const uint32_t hi0 = __umulhi(M0, c[0]), lo0 = M0 * c[0];
    const uint32_t hi1 = __umulhi(M1, c[2]), lo1 = M1 * c[2];
    const uint32_t n0 = hi1 ^ c[1] ^ k0, n1 = lo1, n2 = hi0 ^ c[3] ^ k1, n3 = lo0;
    c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
}

// uniform double in [0,1) from Philox4x32-10(counter = (i_lo, i_hi, stream, 0), key = (k0, k1))
__device__ __forceinline__ double philox_uniform(uint64_t i, uint32_t stream, uint32_t k0, uint32_t k1) {
    uint32_t c[4] = {(uint32_t)i, (uint32_t)(i >> 32), stream, 0u};
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        philox_round(c, k0, k1);
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    const uint64_t m = ((uint64_t)c[0] << 21) ^ (uint64_t)(c[1] >> 11);  // 53 bits
    return (double)(m & ((1ull << 53) - 1)) * (1.0 / 9007199254740992.0);
}

struct GibbsArgs {
    const double *pool;
    const GibbsVar *vars;
    const int32_t *scope_var;     // flattened (variable, stride) of every CPT scope
    const int32_t *scope_stride;
    const int32_t *children;      // flattened children lists
    const int32_t *cycle;         // update order (non-evidence variables)
    const int32_t *uprog;         // update programs: card, n_factors, {table_off, stride of v, n_other, {var, stride}*}*
    const int32_t *uprog_off;     // word offset of cycle position i's program
    const int32_t *qvars;
    const int32_t *qstride;       // stride of each query variable in the joint histogram
    unsigned long long *counts;   // global histogram
    int32_t n_vars, n_cycle, n_q, hist_cells;
    int32_t pool_cells;           // doubles in `pool` (the LDS-resident variant copies them all)
    int32_t prog_words;           // words of uprog + uprog_off + cycle when they are copied to LDS too (else 0)
    int32_t uprog_words;          // words of uprog alone
    int64_t n_chains, n_iterations;
    int64_t chain_first;          // global index of this launch's first chain (shards of one stream: mibn_gibbs_shard)
    uint64_t seed;
    // COND (mibn_gibbs_conditional, a parity hook): the rows' states come from the caller, ONE update of cycle position cond_pos is
    // evaluated and its normalised weights are written instead of a draw
    const uint8_t *cond_states;   // [n_chains][n_vars] codes
    double *cond_out;             // [n_chains][card of the variable]
    int32_t cond_pos;
};

constexpr int kGibbsWaves = 1;  // waves per workgroup (each wave = 64 independent chains)
constexpr int kFastWords = 28;  // words of one cycle position's update record in the FAST format (gibbs_kernel)

// weight of value x of variable v given the rest of the lane's state
template <typename PoolPtr>
__device__ __forceinline__ double gibbs_weight(const GibbsArgs &A, PoolPtr pool, const GibbsVar &V, int v, int x,
                                               const uint8_t *st /* state[var*64 + lane] */, int lane) {
    // own CPT: scope = [*parents, v], v last with stride 1
    int off = V.table_off + x;
    for (int k = 0; k + 1 < V.scope_len; ++k)
        off += (int)st[A.scope_var[V.scope_begin + k] * 64 + lane] * A.scope_stride[V.scope_begin + k];
    double w = pool[off];
    for (int ci = 0; ci < V.child_len; ++ci) {
        const int c = A.children[V.child_begin + ci];
        const GibbsVar C = A.vars[c];
        int o = C.table_off;
        for (int k = 0; k < C.scope_len; ++k) {
            const int u = A.scope_var[C.scope_begin + k];
            const int s = A.scope_stride[C.scope_begin + k];
            o += (u == v ? x : (int)st[u * 64 + lane]) * s;
        }
        w *= pool[o];
    }
    return w;
}

// POOL_LDS: every CPT is copied into LDS first (config 5: 154 KB of tables + 3 KB of chain state in the 160 KB of a
// CU), so the card x n_factors table reads of an update are ds_read_b64 instead of L2 round trips - the update chain is
// latency-bound and few chains (config 5: 2 waves per GPU) cannot hide it with occupancy.
// PROG_LDS: the update programs, their offsets and the cycle are copied into LDS as well (config 5: 4.4 KB beside the
// 154 KB of tables) - read from global memory they are a chain of dependent scalar loads per factor, most of the 2.4 us
// an update took; from LDS the same words cost a broadcast read each.
// FAST (with POOL_LDS and PROG_LDS; every variable of the cycle has at most 8 states, 4 factors and factors with at most
// two other variables - grids): fixed-size update records, so that the state reads of all factors, then all their table
// reads, are issued together instead of factor after factor - the update is a handful of dependent LDS round trips.
// COND: the parity hook of mibn_gibbs_conditional - the same set-up, the same weight computation of whichever of the three
// update forms this launch takes, the weights written out (normalised) where the chain would draw from them.
template <bool POOL_LDS, bool PROG_LDS, bool FAST = false, bool COND = false>
__global__ __launch_bounds__(64 * kGibbsWaves) void gibbs_kernel(const GibbsArgs A) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & 63;
    uint8_t *st = smem;                                        // n_vars * 64 bytes
    unsigned int *hist = (unsigned int *)(smem + ((A.n_vars * 64 + 15) & ~15));  // hist_cells
    double *lds_pool = (double *)(smem + ((((A.n_vars * 64 + 15) & ~15) + A.hist_cells * 4 + 15) & ~15));
    if (POOL_LDS)
        for (int i = threadIdx.x; i < A.pool_cells; i += blockDim.x) lds_pool[i] = A.pool[i];
    if (FAST && threadIdx.x == 0) lds_pool[A.pool_cells] = 1.0;  // the table of an unused factor slot
    // uprog | uprog_off | cycle (FAST: 16-byte aligned records of kFastWords words, one per cycle position, nothing else)
    int32_t *lds_prog = FAST ? (int32_t *)((unsigned char *)lds_pool + (((size_t)(A.pool_cells + 1) * 8 + 15) & ~size_t(15)))
                             : (int32_t *)(lds_pool + (POOL_LDS ? A.pool_cells : 0));
    if (PROG_LDS) {
        for (int i = threadIdx.x; i < A.uprog_words; i += blockDim.x) lds_prog[i] = A.uprog[i];
        for (int i = threadIdx.x; i < (FAST ? 0 : A.n_cycle); i += blockDim.x) {
            lds_prog[A.uprog_words + i] = A.uprog_off[i];
            lds_prog[A.uprog_words + A.n_cycle + i] = A.cycle[i];
        }
    }
    const int32_t *uprog = PROG_LDS ? lds_prog : A.uprog;
    const int32_t *uprog_off = PROG_LDS ? lds_prog + A.uprog_words : A.uprog_off;
    const int32_t *cycle = PROG_LDS ? lds_prog + A.uprog_words + A.n_cycle : A.cycle;
    const int64_t local = (int64_t)blockIdx.x * 64 + lane;
    const int64_t chain = A.chain_first + local;  // the Philox key follows the global index: shards reproduce the whole
    const bool active = local < A.n_chains;
    for (int i = threadIdx.x; i < A.hist_cells; i += blockDim.x) hist[i] = 0;
    const uint32_t k0 = (uint32_t)A.seed ^ (uint32_t)chain * 0x9E3779B1u;
    const uint32_t k1 = (uint32_t)(A.seed >> 32) ^ (uint32_t)(chain >> 32) ^ 0x85EBCA6Bu;

    // forward (ancestral) sample with the evidence clamped - BayesNet.sample(init=event), bayes_net.py:715
    for (int v = 0; v < A.n_vars; ++v) {
        const GibbsVar V = A.vars[v];
        int val = V.ev_code;
        if (COND) {
            if (!V.is_evidence) val = active ? (int)A.cond_states[local * A.n_vars + v] : 0;
        } else if (!V.is_evidence) {
            int off = V.table_off;
            for (int k = 0; k + 1 < V.scope_len; ++k)
                off += (int)st[A.scope_var[V.scope_begin + k] * 64 + lane] * A.scope_stride[V.scope_begin + k];
            double total = 0;
            for (int x = 0; x < V.card; ++x) total += A.pool[off + x];
            const double u = philox_uniform((uint64_t)v, 1u, k0, k1) * total;
            double acc = 0;
            val = V.card - 1;
            for (int x = 0; x < V.card; ++x) {
                acc += A.pool[off + x];
                if (u < acc) { val = x; break; }
            }
        }
        st[v * 64 + lane] = (uint8_t)val;
    }
    __syncthreads();

    int qv4[4] = {0, 0, 0, 0}, qs4[4] = {0, 0, 0, 0};  // (unused slots: variable 0 with stride 0)
    for (int q = 0; q < 4 && q < A.n_q; ++q) { qv4[q] = A.qvars[q]; qs4[q] = A.qstride[q]; }
    int cyc = COND ? A.cond_pos : 0;
    constexpr int kRegCard = 16;
    typedef int32_t i32x4 __attribute__((ext_vector_type(4)));
    i32x4 cur[kFastWords / 4];  // FAST: the update record of this iteration, loaded one iteration ahead
    if (FAST) {
#pragma unroll
        for (int q = 0; q < kFastWords / 4; ++q) cur[q] = reinterpret_cast<const i32x4 *>(lds_prog + cyc * kFastWords)[q];
    }
    for (int64_t it = 0; it < (COND ? 1 : A.n_iterations); ++it) {
        int v, card, nf = 0;
        const int32_t *up = nullptr;
        i32x4 rec[kFastWords / 4];
        if (FAST) {
            // [v, card, nf, -, then four records of (table base, stride of v, other variable 0, its stride, other variable 1,
            // its stride)]; an unused slot points at the 1.0 behind the tables with all strides 0.  The next position's
            // record is requested now - before this update's state write, which the compiler must assume aliases it.
#pragma unroll
            for (int q = 0; q < kFastWords / 4; ++q) rec[q] = cur[q];
            cyc = cyc + 1 == A.n_cycle ? 0 : cyc + 1;
#pragma unroll
            for (int q = 0; q < kFastWords / 4; ++q) cur[q] = reinterpret_cast<const i32x4 *>(lds_prog + cyc * kFastWords)[q];
            v = __builtin_amdgcn_readfirstlane(rec[0][0]);
            card = __builtin_amdgcn_readfirstlane(rec[0][1]);
        } else {
            // (LDS reads return the same word in every lane: readfirstlane keeps the control flow scalar)
            v = PROG_LDS ? __builtin_amdgcn_readfirstlane(cycle[cyc]) : cycle[cyc];
            up = uprog + (PROG_LDS ? __builtin_amdgcn_readfirstlane(uprog_off[cyc]) : uprog_off[cyc]);
            cyc = cyc + 1 == A.n_cycle ? 0 : cyc + 1;
            card = PROG_LDS ? __builtin_amdgcn_readfirstlane(up[0]) : up[0];
            nf = PROG_LDS ? __builtin_amdgcn_readfirstlane(up[1]) : up[1];
            up += 2;
        }
        if (FAST) {
            // the uniform of this update does not depend on the state: computed first, its ~500 integer operations run
            // under the LDS round trips below instead of after them
            const double u01 = philox_uniform((uint64_t)it, 0u, k0, k1);
            int base[4], sv[4];
#pragma unroll
            for (int f = 0; f < 4; ++f) {
                const int w0 = 4 + 6 * f;
                auto R = [&](int k) { return rec[(w0 + k) >> 2][(w0 + k) & 3]; };
                base[f] = R(0) + (int)st[R(2) * 64 + lane] * R(3) + (int)st[R(4) * 64 + lane] * R(5);
                sv[f] = R(1);
            }
            double w[8];
#pragma unroll
            for (int x = 0; x < 8; ++x) {
                const bool on = x < card;
                double p = lds_pool[on ? base[0] + x * sv[0] : A.pool_cells];
#pragma unroll
                for (int f = 1; f < 4; ++f) p *= lds_pool[on ? base[f] + x * sv[f] : A.pool_cells];
                w[x] = p;
            }
            double total = 0;
#pragma unroll
            for (int x = 0; x < 8; ++x)
                if (x < card) total += w[x];
            if constexpr (COND) {
                if (active) {
#pragma unroll
                    for (int x = 0; x < 8; ++x)
                        if (x < card) A.cond_out[local * card + x] = total > 0 ? w[x] / total : 0.0;
                }
            } else if (total > 0) {
                const double u = u01 * total;
                double acc = 0;
                int val = -1, last = 0;
#pragma unroll
                for (int x = 0; x < 8; ++x)
                    if (x < card) {
                        if (w[x] > 0) last = x;
                        acc += w[x];
                        if (val < 0 && u < acc) val = x;
                    }
                st[v * 64 + lane] = (uint8_t)(val < 0 ? last : val);
            }
        } else if (card <= kRegCard) {
            double w[kRegCard];
#pragma unroll
            for (int x = 0; x < kRegCard; ++x) w[x] = 1.0;
            for (int f = 0; f < nf; ++f) {
                const int sv = up[1];
                const int no = PROG_LDS ? __builtin_amdgcn_readfirstlane(up[2]) : up[2];
                int base = up[0];
                for (int k = 0; k < no; ++k) base += (int)st[up[3 + 2 * k] * 64 + lane] * up[4 + 2 * k];
                up += 3 + 2 * no;
                double tv[kRegCard];
#pragma unroll
                for (int x = 0; x < kRegCard; ++x)
                    if (x < card) tv[x] = POOL_LDS ? lds_pool[base + x * sv] : A.pool[base + x * sv];
#pragma unroll
                for (int x = 0; x < kRegCard; ++x)
                    if (x < card) w[x] *= tv[x];
            }
            double total = 0;
#pragma unroll
            for (int x = 0; x < kRegCard; ++x)
                if (x < card) total += w[x];
            if constexpr (COND) {
                if (active) {
#pragma unroll
                    for (int x = 0; x < kRegCard; ++x)
                        if (x < card) A.cond_out[local * card + x] = total > 0 ? w[x] / total : 0.0;
                }
            } else if (total > 0) {
                const double u = philox_uniform((uint64_t)it, 0u, k0, k1) * total;
                double acc = 0;
                int val = -1, last = 0;
#pragma unroll
                for (int x = 0; x < kRegCard; ++x)
                    if (x < card) {
                        if (w[x] > 0) last = x;
                        acc += w[x];
                        if (val < 0 && u < acc) val = x;
                    }
                st[v * 64 + lane] = (uint8_t)(val < 0 ? last : val);
            }
        } else {
            const GibbsVar V = A.vars[v];
            double total = 0;
            for (int x = 0; x < V.card; ++x) total += POOL_LDS ? gibbs_weight(A, lds_pool, V, v, x, st, lane) : gibbs_weight(A, A.pool, V, v, x, st, lane);
            if constexpr (COND) {
                if (active)
                    for (int x = 0; x < V.card; ++x) {
                        const double w = POOL_LDS ? gibbs_weight(A, lds_pool, V, v, x, st, lane) : gibbs_weight(A, A.pool, V, v, x, st, lane);
                        A.cond_out[local * V.card + x] = total > 0 ? w / total : 0.0;
                    }
            } else if (total > 0) {
                const double u = philox_uniform((uint64_t)it, 0u, k0, k1) * total;
                double acc = 0;
                int val = -1, last = 0;
                for (int x = 0; x < V.card; ++x) {
                    const double w = POOL_LDS ? gibbs_weight(A, lds_pool, V, v, x, st, lane) : gibbs_weight(A, A.pool, V, v, x, st, lane);
                    if (w > 0) last = x;
                    acc += w;
                    if (val < 0 && u < acc) val = x;
                }
                st[v * 64 + lane] = (uint8_t)(val < 0 ? last : val);
            }
        }
        // record the joint query state (bayes_net.py:732-733: every iteration, no burn-in)
        if (!COND && active) {
            int cell = 0;
            if (A.n_q <= 4) {  // the usual case: query variables and strides preloaded (no global loads in the loop)
#pragma unroll
                for (int q = 0; q < 4; ++q) cell += (int)st[qv4[q] * 64 + lane] * qs4[q];
            } else {
                for (int q = 0; q < A.n_q; ++q) cell += (int)st[A.qvars[q] * 64 + lane] * A.qstride[q];
            }
            atomicAdd(&hist[cell], 1u);
        }
        if ((it & 0xffffff) == 0xffffff) {  // flush before a 32-bit LDS counter can overflow
            __syncthreads();
            for (int i = threadIdx.x; i < A.hist_cells; i += blockDim.x) {
                if (hist[i]) atomicAdd(&A.counts[i], (unsigned long long)hist[i]);
                hist[i] = 0;
            }
            __syncthreads();
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < A.hist_cells; i += blockDim.x)
        if (hist[i]) atomicAdd(&A.counts[i], (unsigned long long)hist[i]);
}

// ------------------------------------------------------------------------------------------------------------------
// Round 4: gibbs_kernel8 - the FAST form with EIGHT LANES PER CHAIN (lane = chain g of the wave x candidate state x).
//
// One chain per lane made the update a serial program of ~450 vector instructions on a wave that has its SIMD to itself
// (config 5: 2 waves per GPU): 32 table reads and 24 fp64 multiplies for the 8 weights, an 8-long accumulate-and-select
// loop, and - a third of the 2 800 cycles - Philox4x32-10 (40 quarter-rate 32-bit multiplies) in every lane.  Here a chain
// is spread over 8 lanes: lane x reads the 4 table values of candidate state x and multiplies them (same order), the
// running sum is a 7-step DPP scan ALONG the lanes in the sequential order of the loop it replaces (acc_x = acc_(x-1) + w_x:
// bit for bit the same sums, so the same draws), the draw is one ballot + find-first, and lane x computes the Philox
// uniform of iteration it0 + x once per eight iterations (same counter, same key => the same stream).  Eight chains per
// wave, so config 5's 128 chains per GPU are 16 workgroups on 16 CUs instead of 2.  Histograms bit for bit those of
// gibbs_kernel (tests: the gibbs_lds=0 / gibbs_fast8=0 comparisons), COND as there.
__device__ __forceinline__ double dpp_f64(const double v, const int ctrl_is_shr1_mirror_q3) {
    // 0: row_shr:1 (lane i <- lane i - 1), 1: row_half_mirror (lane x <- lane 7 - x of its 8), 2: quad_perm [3, 3, 3, 3]
    int lo = __double2loint(v), hi = __double2hiint(v);
    if (ctrl_is_shr1_mirror_q3 == 0) {
        lo = __builtin_amdgcn_update_dpp(lo, lo, 0x111, 0xf, 0xf, false);
        hi = __builtin_amdgcn_update_dpp(hi, hi, 0x111, 0xf, 0xf, false);
    } else if (ctrl_is_shr1_mirror_q3 == 1) {
        lo = __builtin_amdgcn_update_dpp(lo, lo, 0x141, 0xf, 0xf, false);
        hi = __builtin_amdgcn_update_dpp(hi, hi, 0x141, 0xf, 0xf, false);
    } else {
        lo = __builtin_amdgcn_update_dpp(lo, lo, 0xff, 0xf, 0xf, false);
        hi = __builtin_amdgcn_update_dpp(hi, hi, 0xff, 0xf, 0xf, false);
    }
    return __hiloint2double(hi, lo);
}

template <bool COND>
__global__ __launch_bounds__(64) void gibbs_kernel8(const GibbsArgs A) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & 63, g = lane >> 3, x = lane & 7;
    uint8_t *st = smem;                                        // state[var * 8 + chain of the wave] (room for n_vars * 64 as in gibbs_kernel)
    unsigned int *hist = (unsigned int *)(smem + ((A.n_vars * 64 + 15) & ~15));
    double *lds_pool = (double *)(smem + ((((A.n_vars * 64 + 15) & ~15) + A.hist_cells * 4 + 15) & ~15));
    for (int i = threadIdx.x; i < A.pool_cells; i += blockDim.x) lds_pool[i] = A.pool[i];
    if (threadIdx.x == 0) lds_pool[A.pool_cells] = 1.0;  // the table of an unused factor slot
    int32_t *lds_prog = (int32_t *)((unsigned char *)lds_pool + (((size_t)(A.pool_cells + 1) * 8 + 15) & ~size_t(15)));
    for (int i = threadIdx.x; i < A.uprog_words; i += blockDim.x) lds_prog[i] = A.uprog[i];
    const int64_t local = (int64_t)blockIdx.x * 8 + g;
    const int64_t chain = A.chain_first + local;
    const bool active = local < A.n_chains;
    for (int i = threadIdx.x; i < A.hist_cells; i += blockDim.x) hist[i] = 0;
    const uint32_t k0 = (uint32_t)A.seed ^ (uint32_t)chain * 0x9E3779B1u;
    const uint32_t k1 = (uint32_t)(A.seed >> 32) ^ (uint32_t)(chain >> 32) ^ 0x85EBCA6Bu;
    // forward (ancestral) sample with the evidence clamped (bayes_net.py:715): the eight lanes of a chain compute the same values
    for (int v = 0; v < A.n_vars; ++v) {
        const GibbsVar V = A.vars[v];
        int val = V.ev_code;
        if (COND) {
            if (!V.is_evidence) val = active ? (int)A.cond_states[local * A.n_vars + v] : 0;
        } else if (!V.is_evidence) {
            int off = V.table_off;
            for (int k = 0; k + 1 < V.scope_len; ++k)
                off += (int)st[A.scope_var[V.scope_begin + k] * 8 + g] * A.scope_stride[V.scope_begin + k];
            double total = 0;
            for (int c = 0; c < V.card; ++c) total += A.pool[off + c];
            const double u = philox_uniform((uint64_t)v, 1u, k0, k1) * total;
            double acc = 0;
            val = V.card - 1;
            for (int c = 0; c < V.card; ++c) {
                acc += A.pool[off + c];
                if (u < acc) { val = c; break; }
            }
        }
        st[v * 8 + g] = (uint8_t)val;
    }
    __syncthreads();
    int qv4[4] = {0, 0, 0, 0}, qs4[4] = {0, 0, 0, 0};
    for (int q = 0; q < 4 && q < A.n_q; ++q) { qv4[q] = A.qvars[q]; qs4[q] = A.qstride[q]; }
    auto query_cell = [&]() {
        int cell = 0;
        if (A.n_q <= 4) {
#pragma unroll
            for (int q = 0; q < 4; ++q) cell += (int)st[qv4[q] * 8 + g] * qs4[q];
        } else {
            for (int q = 0; q < A.n_q; ++q) cell += (int)st[A.qvars[q] * 8 + g] * A.qstride[q];
        }
        return cell;
    };
    int cell = query_cell();  // changes only when a query variable is updated (record word 3)
    int cyc = COND ? A.cond_pos : 0;
    typedef int32_t i32x4 __attribute__((ext_vector_type(4)));
    i32x4 cur[kFastWords / 4];
#pragma unroll
    for (int q = 0; q < kFastWords / 4; ++q) cur[q] = reinterpret_cast<const i32x4 *>(lds_prog + cyc * kFastWords)[q];
    double my_u = 0.0;
    for (int64_t it = 0; it < (COND ? 1 : A.n_iterations); ++it) {
        i32x4 rec[kFastWords / 4];
#pragma unroll
        for (int q = 0; q < kFastWords / 4; ++q) rec[q] = cur[q];
        cyc = cyc + 1 == A.n_cycle ? 0 : cyc + 1;
#pragma unroll
        for (int q = 0; q < kFastWords / 4; ++q) cur[q] = reinterpret_cast<const i32x4 *>(lds_prog + cyc * kFastWords)[q];
        const int v = __builtin_amdgcn_readfirstlane(rec[0][0]);
        const int card = __builtin_amdgcn_readfirstlane(rec[0][1]);
        const int is_q = __builtin_amdgcn_readfirstlane(rec[0][3]);
        // the uniforms of iterations it .. it + 7: lane x draws the one of it + x (same counter and key as one chain per lane)
        if ((it & 7) == 0) my_u = philox_uniform((uint64_t)(it + x), 0u, k0, k1);
        const double u01 = __shfl(my_u, (lane & ~7) | (int)(it & 7), 64);
        int base[4], sv[4];
#pragma unroll
        for (int f = 0; f < 4; ++f) {
            const int w0 = 4 + 6 * f;
            auto R = [&](int k) { return rec[(w0 + k) >> 2][(w0 + k) & 3]; };
            base[f] = R(0) + (int)st[R(2) * 8 + g] * R(3) + (int)st[R(4) * 8 + g] * R(5);
            sv[f] = R(1);
        }
        const bool on = x < card;
        double p = lds_pool[on ? base[0] + x * sv[0] : A.pool_cells];
#pragma unroll
        for (int f = 1; f < 4; ++f) p *= lds_pool[on ? base[f] + x * sv[f] : A.pool_cells];
        const double w = on ? p : 0.0;
        // acc_x = acc_(x-1) + w_x, in the order of the serial loop (lanes beyond card add 0.0: their sum is the total)
        double acc = w;
#pragma unroll
        for (int k = 1; k < 8; ++k) {
            const double t = dpp_f64(acc, 0);
            if (x == k) acc = t + w;
        }
        // the total = lane 7's sum, to every lane of the chain: quad_perm [3,3,3,3] (lanes 4-7 have it), mirrored into lanes 0-3
        // (both DPP moves with every lane enabled - a `?:` around the second one would run it with lanes 4-7 masked off, and a DPP
        //  read of a disabled lane leaves the destination unchanged: round 4's first version drew lanes 0-3 against a partial sum)
        const double q3 = dpp_f64(acc, 2);
        const double q3m = dpp_f64(q3, 1);
        const double total = x < 4 ? q3m : q3;
        if constexpr (COND) {
            if (active && on) A.cond_out[local * card + x] = total > 0 ? w / total : 0.0;
        } else {
            const double u = u01 * total;
            const unsigned long long hit = __ballot(on && u < acc), pos = __ballot(on && w > 0);
            const unsigned hb = (unsigned)(hit >> (8 * g)) & 0xffu, pb = (unsigned)(pos >> (8 * g)) & 0xffu;
            if (total > 0) {
                const int val = hb ? __ffs((int)hb) - 1 : (pb ? 31 - __clz((int)pb) : 0);
                st[v * 8 + g] = (uint8_t)val;  // (the eight lanes write the same byte)
            }
            if (is_q) cell = query_cell();
            if (active && x == 0) atomicAdd(&hist[cell], 1u);  // bayes_net.py:732-733: every iteration, no burn-in
        }
        if ((it & 0xffffff) == 0xffffff) {  // flush before a 32-bit LDS counter can overflow
            __syncthreads();
            for (int i = threadIdx.x; i < A.hist_cells; i += blockDim.x) {
                if (hist[i]) atomicAdd(&A.counts[i], (unsigned long long)hist[i]);
                hist[i] = 0;
            }
            __syncthreads();
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < A.hist_cells; i += blockDim.x)
        if (hist[i]) atomicAdd(&A.counts[i], (unsigned long long)hist[i]);
}

// host driver; returns MIBN_* code
// cond_var >= 0 (mibn_gibbs_conditional): n_chains rows of cond_states, the conditional of cond_var into cond_out
inline int gibbs_run(const Network &net, const double *d_pool, hipStream_t stream, int lds_mode /* 0: tables in L2; 1: LDS; 2: LDS, no 8-lane form */, int32_t n_q, const int32_t *q_vars,
                     int32_t n_e, const int32_t *e_vars, const int32_t *e_codes, const int32_t *cycle_in, int64_t chain_first,
                     int64_t n_chains, int64_t n_iterations, uint64_t seed, int64_t *counts, std::string &err, double &kernel_ms,
                     int32_t cond_var = -1, const uint8_t *cond_states = nullptr, double *cond_out = nullptr) {
    const int n = net.n_vars;
    const bool allow_lds = lds_mode != 0;
    std::vector<GibbsVar> vars(n);
    std::vector<int32_t> scope_var, scope_stride, children, cycle;
    std::vector<std::vector<int32_t>> ch(n);
    for (int v = 0; v < n; ++v)
        for (size_t k = 0; k + 1 < net.scope[v].size(); ++k) ch[net.scope[v][k]].push_back(v);
    for (int v = 0; v < n; ++v) {
        if (net.card[v] > 255) { err = "gibbs: cardinality above 255"; return MIBN_E_LIMIT; }
        GibbsVar g{};
        g.card = net.card[v];
        g.table_off = (int32_t)net.pool_off[v];
        g.scope_begin = (int32_t)scope_var.size();
        g.scope_len = (int32_t)net.scope[v].size();
        for (size_t k = 0; k < net.scope[v].size(); ++k) {
            scope_var.push_back(net.scope[v][k]);
            scope_stride.push_back((int32_t)net.cstride[v][k]);
        }
        g.child_begin = (int32_t)children.size();
        g.child_len = (int32_t)ch[v].size();
        for (int c : ch[v]) children.push_back(c);
        vars[v] = g;
    }
    for (int i = 0; i < n_e; ++i) {
        if (e_codes[i] < 0 || e_codes[i] >= net.card[e_vars[i]]) { err = "gibbs: evidence label outside the domain"; return MIBN_E_ARG; }
        vars[e_vars[i]].is_evidence = 1;
        vars[e_vars[i]].ev_code = e_codes[i];
    }
    int n_free = 0;
    for (int v = 0; v < n; ++v) n_free += !vars[v].is_evidence;
    if (cycle_in) {  // caller's update order (the reference cycles through sorted(nodes - event), bayes_net.py:697,718)
        std::vector<char> seen(n, 0);
        for (int i = 0; i < n_free; ++i) {
            const int v = cycle_in[i];
            if (v < 0 || v >= n || vars[v].is_evidence || seen[v]) { err = "gibbs: cycle must list every non-evidence variable once"; return MIBN_E_ARG; }
            seen[v] = 1;
            cycle.push_back(v);
        }
    } else {
        for (int v = 0; v < n; ++v)
            if (!vars[v].is_evidence) cycle.push_back(v);
    }
    if (cycle.empty()) { err = "gibbs: every variable is evidence"; return MIBN_E_ARG; }
    int cond_pos = -1;
    if (cond_var >= 0) {
        for (size_t i = 0; i < cycle.size(); ++i)
            if (cycle[i] == cond_var) cond_pos = (int)i;
        if (cond_pos < 0) { err = "gibbs conditional: the variable is evidence or unknown"; return MIBN_E_ARG; }
    }
    // update programs, one per cycle position
    std::vector<int32_t> uprog, uprog_off;
    for (int v : cycle) {
        uprog_off.push_back((int32_t)uprog.size());
        uprog.push_back(net.card[v]);
        uprog.push_back(1 + (int32_t)ch[v].size());
        auto factor = [&](int c) {  // CPT of c as a function of v's value
            int32_t sv = 0;
            std::vector<int32_t> others;
            for (size_t k = 0; k < net.scope[c].size(); ++k) {
                const int u = net.scope[c][k];
                if (u == v) sv = (int32_t)net.cstride[c][k];
                else { others.push_back(u); others.push_back((int32_t)net.cstride[c][k]); }
            }
            uprog.push_back((int32_t)net.pool_off[c]);
            uprog.push_back(sv);
            uprog.push_back((int32_t)(others.size() / 2));
            uprog.insert(uprog.end(), others.begin(), others.end());
        };
        factor(v);
        for (int c : ch[v]) factor(c);
    }
    std::vector<int32_t> qstride(n_q);
    int64_t cells = 1;
    for (int i = n_q - 1; i >= 0; --i) { qstride[i] = (int32_t)cells; cells *= net.card[q_vars[i]]; }
    size_t lds = ((size_t)n * 64 + 15) / 16 * 16 + (size_t)cells * 4;
    if (lds > 150 * 1024) { err = "gibbs: network/query too large for the LDS-resident chain state"; return MIBN_E_LIMIT; }
    // tables in LDS too when they fit beside the state and the launch is small enough for one workgroup per CU
    const size_t pool_cells = net.pool.size();
    const size_t lds_with_pool = (lds + 15) / 16 * 16 + pool_cells * 8;
    // the 8-lane form (gibbs_kernel8) has eight chains per workgroup: it wins while its workgroups - one per CU, the tables fill the
    // LDS - need at most two rounds over the 256 CUs (~0.4 us per update against 1.16 us for gibbs_kernel's 64 chains per wave)
    const bool want8 = lds_mode == 1 && (n_chains + 7) / 8 <= 512;
    const bool pool_lds = allow_lds && lds_with_pool <= 160 * 1024 && (want8 || (n_chains + 63) / 64 <= 512);
    if (pool_lds) lds = lds_with_pool;
    // fixed-size update records (gibbs_kernel<.., FAST>) when every variable of the cycle qualifies and they still fit
    bool fast = pool_lds;
    std::vector<int32_t> fprog;
    for (size_t ci = 0; ci < cycle.size() && fast; ++ci) {
        const int v = cycle[ci];
        if (net.card[v] > 8 || 1 + ch[v].size() > 4) { fast = false; break; }
        int32_t is_query = 0;
        for (int i = 0; i < n_q; ++i) is_query |= q_vars[i] == v;
        const int32_t head[4] = {v, net.card[v], 1 + (int32_t)ch[v].size(), is_query};
        fprog.insert(fprog.end(), head, head + 4);
        int slots = 0;
        auto record = [&](int c) {
            int32_t rec[6] = {(int32_t)net.pool_off[c], 0, 0, 0, 0, 0};
            int n_other = 0;
            for (size_t k = 0; k < net.scope[c].size(); ++k) {
                const int u = net.scope[c][k];
                if (u == v) rec[1] = (int32_t)net.cstride[c][k];
                else if (n_other < 2) { rec[2 + 2 * n_other] = u; rec[3 + 2 * n_other] = (int32_t)net.cstride[c][k]; ++n_other; }
                else fast = false;
            }
            fprog.insert(fprog.end(), rec, rec + 6);
            ++slots;
        };
        record(v);
        for (int c : ch[v]) record(c);
        for (; slots < 4; ++slots) {  // unused slots: the 1.0 behind the tables, strides 0
            const int32_t rec[6] = {(int32_t)pool_cells, 0, 0, 0, 0, 0};
            fprog.insert(fprog.end(), rec, rec + 6);
        }
    }
    if (fast && (lds + 8 + 15) / 16 * 16 + fprog.size() * 4 > 160 * 1024) fast = false;
    if (fast) {
        uprog = fprog;
        lds += 8;
    }
    const size_t prog_words = fast ? uprog.size() : uprog.size() + 2 * cycle.size();
    const bool prog_lds = fast || (allow_lds && (lds + 15) / 16 * 16 + prog_words * 4 <= 160 * 1024 && (n_chains + 63) / 64 <= 512);
    if (prog_lds) lds = (lds + 15) / 16 * 16 + prog_words * 4;

    GibbsVar *d_vars = nullptr;
    int32_t *d_i32 = nullptr;
    unsigned long long *d_counts = nullptr;
    uint8_t *d_cond_states = nullptr;
    double *d_cond_out = nullptr;
    std::vector<int32_t> pack;
    auto put = [&](const std::vector<int32_t> &a) { size_t o = pack.size(); pack.insert(pack.end(), a.begin(), a.end()); return o; };
    const size_t o_sv = put(scope_var), o_ss = put(scope_stride), o_ch = put(children), o_cy = put(cycle);
    const size_t o_q = put(std::vector<int32_t>(q_vars, q_vars + n_q)), o_qs = put(qstride);
    const size_t o_up = put(uprog), o_uo = put(uprog_off);
    auto fail = [&](hipError_t e) { err = std::string("gibbs: ") + hipGetErrorString(e); hipFree(d_vars); hipFree(d_i32); hipFree(d_counts); hipFree(d_cond_states); hipFree(d_cond_out); return MIBN_E_HIP; };
    hipError_t e;
    if ((e = hipMalloc(&d_vars, sizeof(GibbsVar) * n)) != hipSuccess) return fail(e);
    if ((e = hipMalloc(&d_i32, 4 * std::max<size_t>(1, pack.size()))) != hipSuccess) return fail(e);
    if ((e = hipMalloc(&d_counts, 8 * (size_t)cells)) != hipSuccess) return fail(e);
    if ((e = hipMemcpyAsync(d_vars, vars.data(), sizeof(GibbsVar) * n, hipMemcpyHostToDevice, stream)) != hipSuccess) return fail(e);
    if ((e = hipMemcpyAsync(d_i32, pack.data(), 4 * pack.size(), hipMemcpyHostToDevice, stream)) != hipSuccess) return fail(e);
    if ((e = hipMemsetAsync(d_counts, 0, 8 * (size_t)cells, stream)) != hipSuccess) return fail(e);
    GibbsArgs A;
    A.pool = d_pool;
    A.vars = d_vars;
    A.scope_var = d_i32 + o_sv;
    A.scope_stride = d_i32 + o_ss;
    A.children = d_i32 + o_ch;
    A.cycle = d_i32 + o_cy;
    A.uprog = d_i32 + o_up;
    A.uprog_off = d_i32 + o_uo;
    A.qvars = d_i32 + o_q;
    A.qstride = d_i32 + o_qs;
    A.counts = d_counts;
    A.n_vars = n;
    A.pool_cells = (int32_t)pool_cells;
    A.uprog_words = (int32_t)uprog.size();
    A.prog_words = prog_lds ? (int32_t)prog_words : 0;
    A.n_cycle = (int32_t)cycle.size();
    A.n_q = n_q;
    A.hist_cells = (int32_t)cells;
    A.n_chains = n_chains;
    A.chain_first = chain_first;
    A.n_iterations = n_iterations;
    A.seed = seed;
    A.cond_states = nullptr;
    A.cond_out = nullptr;
    A.cond_pos = 0;
    const size_t cond_cells = cond_var >= 0 ? (size_t)n_chains * (size_t)net.card[cond_var] : 0;
    if (cond_var >= 0) {
        if ((e = hipMalloc(&d_cond_states, std::max<size_t>(1, (size_t)n_chains * n))) != hipSuccess) return fail(e);
        if ((e = hipMalloc(&d_cond_out, 8 * std::max<size_t>(1, cond_cells))) != hipSuccess) return fail(e);
        if ((e = hipMemcpyAsync(d_cond_states, cond_states, (size_t)n_chains * n, hipMemcpyHostToDevice, stream)) != hipSuccess) return fail(e);
        A.cond_states = d_cond_states;
        A.cond_out = d_cond_out;
        A.cond_pos = cond_pos;
    }
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const bool fast8 = fast && want8;
    const unsigned blocks = fast8 ? (unsigned)((n_chains + 7) / 8) : (unsigned)((n_chains + 63) / 64);
    auto kernel = fast8 ? gibbs_kernel8<false> : fast ? gibbs_kernel<true, true, true>
                       : pool_lds ? (prog_lds ? gibbs_kernel<true, true> : gibbs_kernel<true, false>)
                                  : (prog_lds ? gibbs_kernel<false, true> : gibbs_kernel<false, false>);
    if (cond_var >= 0)  // the same form, writing the weights instead of drawing from them
        kernel = fast8 ? gibbs_kernel8<true> : fast ? gibbs_kernel<true, true, true, true>
                      : pool_lds ? (prog_lds ? gibbs_kernel<true, true, false, true> : gibbs_kernel<true, false, false, true>)
                                 : (prog_lds ? gibbs_kernel<false, true, false, true> : gibbs_kernel<false, false, false, true>);
    if (lds > 64 * 1024) {
        if ((e = hipFuncSetAttribute((const void *)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)) != hipSuccess) return fail(e);
    }
    hipEventRecord(e0, stream);
    hipLaunchKernelGGL(kernel, dim3(blocks), dim3(64 * kGibbsWaves), lds, stream, A);
    if ((e = hipGetLastError()) != hipSuccess) return fail(e);
    hipEventRecord(e1, stream);
    std::vector<unsigned long long> hc((size_t)cells);
    if ((e = hipMemcpyAsync(hc.data(), d_counts, 8 * (size_t)cells, hipMemcpyDeviceToHost, stream)) != hipSuccess) return fail(e);
    if ((e = hipStreamSynchronize(stream)) != hipSuccess) return fail(e);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    kernel_ms = ms;
    hipEventDestroy(e0);
    hipEventDestroy(e1);
    if (counts)
        for (int64_t i = 0; i < cells; ++i) counts[i] = (int64_t)hc[(size_t)i];
    if (cond_var >= 0) {
        if ((e = hipMemcpy(cond_out, d_cond_out, 8 * cond_cells, hipMemcpyDeviceToHost)) != hipSuccess) return fail(e);
        hipFree(d_cond_states);
        hipFree(d_cond_out);
    }
    hipFree(d_vars);
    hipFree(d_i32);
    hipFree(d_counts);
    return MIBN_OK;
}

}  // namespace mibn
