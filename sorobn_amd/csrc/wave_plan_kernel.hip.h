// wave_plan_kernel: the wave-cooperative device planner (wave_plan.h) as a kernel - included by engine.hip (the product) and by
// tools/ubench/wave_plan_bench.hip (the kernel alone on a recorded request stream).
#pragma once
#include <hip/hip_runtime.h>

#include "../../include/mibn.h"
#include "wave_plan.h"

using namespace mibn;

struct EmitMeta {  // per request
    uint32_t words, n_tags, tag_first;
    int32_t err;  // kEmitErr* (any error: the host plans the chunk itself)
    uint32_t prog_first, pad_;  // words of the slot in front of the program (order_effort 1: the runner-up's program stays behind the first one)
    double alg_bytes, alg_flops, n_steps, max_step_cells;
    int64_t arena_cells;
};

// ---------------------------------------------------------------------------------------------- wave-cooperative device planner
// Round 6 (csrc/wave_plan.h): ONE request per wave - order search and emission in one launch, the request's planning state in LDS
// (order_kernel / emit_kernel above keep theirs in scratch and global memory, one request per lane, and wait for it 86 % of their
// cycles: profiles/r06_a_plansq_counters.txt).  The four waves of a workgroup share the network tables (WNet, 8 KB of LDS) and plan a
// request each - the next one from a counter when they are done; 39.8 KB per workgroup, four workgroups = sixteen waves per CU at 128
// registers.  The programs, the per-request results and the work items go where emit_kernel writes them, word for word what the host
// plans (gpu_emit = 2 compares every word); with order_effort 1 a request's program may start behind the first words of its slot
// (EmitMeta::prog_first: the runner-up's program, emitted behind the first one, won).
struct WavePlanArgs {
    const WNet *net;
    const B2 *anc;                           // [n_vars] ancestor sets
    const int64_t *q_off, *e_off, *out_off;  // [B + 1]; out_off relative to the chunk's first request
    const int32_t *q_vars, *e_vars, *e_codes;
    const char *skip;                        // [B] evidence outside the domain: zero steps
    int64_t B;
    uint32_t flags;
    uint32_t *prog;                          // [B][prog_stride]
    uint32_t prog_stride;
    EmitMeta *meta;                          // [B]
    Tag *tags;                               // [tag_cap]
    uint32_t *tag_cursor;                    // [0] work items handed out, [1] requests handed out (both zeroed by reset_cursor_kernel)
    uint32_t tag_cap;
    uint32_t *perm;                          // [2 B] the order in which the requests are handed out (plan_sort_kernel; behind it the sort's bins), or null: by index
};
constexpr int kWaveWG = 4;  // waves (requests) per workgroup
constexpr int kWaveMaxQ = 8, kWaveMaxE = 32;  // query / evidence variables of a request (more: the host plans the chunk)
struct WaveReq { int32_t q[kWaveMaxQ], e[kWaveMaxE], c[kWaveMaxE]; };

#if defined(MIBN_WAVE_PROF)
__device__ unsigned long long g_wave_prof[24];  // ticks per phase, summed over the waves (WV_TICK in wave_plan.h)
#endif

__global__ void reset_cursor_kernel(uint32_t *cursor) { cursor[0] = 0; cursor[1] = 0; }

// The long requests first: the waves draw requests from a counter, and a chunk that ends with its longest requests ends with a tail of a few busy
// waves (handed out in descending order of their relevant-variable count - query, evidence and ancestors: what the planning time follows - a chunk of
// 32 768 C3 requests takes 8.4 instead of 9.3 ms, profiles/NOTES_r06.md session BQ).  One workgroup: a counting sort over the 129 possible counts.
constexpr int kPlanSortThreads = 1024;
__global__ __launch_bounds__(kPlanSortThreads) void plan_sort_kernel(const WavePlanArgs A) {
    __shared__ uint32_t hist[kWVars + 2];
    for (int i = threadIdx.x; i < kWVars + 2; i += kPlanSortThreads) hist[i] = 0;
    __syncthreads();
    auto key_of = [&](int64_t b) -> int {
        if (A.skip[b]) return 0;
        const int64_t q0 = A.q_off[b], e0 = A.e_off[b];
        const int nq = (int)(A.q_off[b + 1] - q0), ne = (int)(A.e_off[b + 1] - e0);
        if (nq > kWaveMaxQ || ne > kWaveMaxE) return 0;
        B2 rel;
        for (int i = 0; i < nq; ++i) { const int v = A.q_vars[q0 + i]; rel.set(v); rel.a |= A.anc[v].a; rel.b |= A.anc[v].b; }
        for (int i = 0; i < ne; ++i) { const int v = A.e_vars[e0 + i]; rel.set(v); rel.a |= A.anc[v].a; rel.b |= A.anc[v].b; }
        return __builtin_popcountll(rel.a) + __builtin_popcountll(rel.b);
    };
    uint32_t *bin = A.perm + A.B;  // (the buffer holds 2 B words: the request arrays are in pinned host memory, read once)
    for (int64_t b = threadIdx.x; b < A.B; b += kPlanSortThreads) {
        const uint32_t k = (uint32_t)(kWVars - key_of(b));  // (bin 0: the longest)
        bin[b] = k;
        atomicAdd(&hist[k], 1u);
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t at = 0;
        for (int i = 0; i <= kWVars; ++i) { const uint32_t c = hist[i]; hist[i] = at; at += c; }
    }
    __syncthreads();
    for (int64_t b = threadIdx.x; b < A.B; b += kPlanSortThreads) A.perm[atomicAdd(&hist[bin[b]], 1u)] = (uint32_t)b;
}

#ifndef MIBN_WAVE_MIN_WGS
#define MIBN_WAVE_MIN_WGS 4  // workgroups per CU the register budget allows (4: 128 VGPRs - some 300 spilled, still the fastest: 11.0 ms per chunk against 13.8 at 3, profiles/r06_v_occupancy.log; LDS: 39.8 KB per workgroup)
#endif
__global__ __launch_bounds__(64 * kWaveWG, MIBN_WAVE_MIN_WGS) void wave_plan_kernel(const WavePlanArgs A) {
    __shared__ WNet N;
    __shared__ WState W[kWaveWG];
    __shared__ WaveReq RQ[kWaveWG];
    {
        const uint32_t *src = reinterpret_cast<const uint32_t *>(A.net);
        uint32_t *dst = reinterpret_cast<uint32_t *>(&N);
        for (unsigned i = threadIdx.x; i < sizeof(WNet) / 4; i += blockDim.x) dst[i] = src[i];
    }
    __syncthreads();
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int lane = (int)(threadIdx.x & 63);
    // Every wave takes the next request when it is done with its own (a counter): requests differ by a factor of ten in planning time - and
    // by two more where two orders are emitted - and a workgroup of four fixed requests would hold its LDS until the slowest is through.
    // The grid is what the chip holds at once; the workgroup keeps its copy of the network.
    // (Every lane adds - lane 0 one, the others nothing - and lane 0's result is the wave's: written as "if (lane == 0) x = atomicAdd(..)"
    //  with x = 0 for the others, the compiler threads the constant into the other lanes' path behind the readfirstlane, the loop becomes
    //  divergent and the lanes of a wave run different requests.  Measured: profiles/NOTES_r06.md, session AX.)
    for (;;) {
        const uint32_t next = atomicAdd(A.tag_cursor + 1, lane == 0 ? 1u : 0u);
        const int64_t at = (int64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)next);
        if (at >= A.B) break;
        const int64_t b = A.perm ? (int64_t)A.perm[at] : at;
        EmitMeta m;
        m.words = 1; m.n_tags = 0; m.tag_first = 0; m.err = 0; m.prog_first = 0; m.pad_ = 0;
        m.alg_bytes = m.alg_flops = m.n_steps = m.max_step_cells = 0;
        m.arena_cells = 0;
        uint32_t *slot = A.prog + (size_t)b * A.prog_stride;
        if (A.skip[b]) {
            if (lane == 0) { slot[0] = 0; A.meta[b] = m; }
            continue;
        }
        const int64_t q0 = A.q_off[b], e0 = A.e_off[b];
        const int nq = (int)(A.q_off[b + 1] - q0), ne = (int)(A.e_off[b + 1] - e0);
        if (nq > kWaveMaxQ || ne > kWaveMaxE) {
            m.err = kEmitErrDevice;
            if (lane == 0) A.meta[b] = m;
            continue;
        }
        // the request's variables: read once (the arrays are in pinned host memory), kept in LDS
        WaveReq &rq = RQ[wave];
        if (lane < nq) rq.q[lane] = A.q_vars[q0 + lane];
        if (lane < ne) { rq.e[lane] = A.e_vars[e0 + lane]; rq.c[lane] = A.e_codes[e0 + lane]; }
        wv::sync();
        WState &S = W[wave];
        WResult R;
    #if defined(MIBN_WAVE_PROF)
        WProf prof_;
        for (int k = 0; k < 24; ++k) prof_.a[k] = 0;
        prof_.t = __builtin_amdgcn_s_memtime();
        wave_plan_request(N, S, A.anc, nq, rq.q, ne, rq.e, rq.c, (A.flags & MIBN_Q_NOPRUNE) != 0, A.out_off[b], slot, A.prog_stride, R, prof_);
        if (lane == 0) for (int k = 0; k < 24; ++k) atomicAdd(&g_wave_prof[k], prof_.a[k]);
    #else
        wave_plan_request(N, S, A.anc, nq, rq.q, ne, rq.e, rq.c, (A.flags & MIBN_Q_NOPRUNE) != 0, A.out_off[b], slot, A.prog_stride, R);
    #endif
        int err = R.err;
        if (!err) {
            uint32_t first = atomicAdd(A.tag_cursor, lane == 0 ? R.n_tags : 0u);  // (every lane adds, see above)
            first = (uint32_t)__builtin_amdgcn_readfirstlane((int)first);
            if (first + R.n_tags <= A.tag_cap) {
                // (the lanes copy the items a dword each)
                static_assert(sizeof(Tag) % 4 == 0, "work items are copied dword by dword");
                constexpr uint32_t kTagWords = sizeof(Tag) / 4;
                const uint32_t *src = reinterpret_cast<const uint32_t *>(S.e.tags);
                uint32_t *dst = reinterpret_cast<uint32_t *>(A.tags + first);
                for (uint32_t i = (uint32_t)lane; i < R.n_tags * kTagWords; i += 64) dst[i] = src[i];
                m.n_tags = R.n_tags;
                m.tag_first = first;
            } else {
                err = kEmitErrWords;
            }
        }
        m.err = err;
        m.words = R.words;
        m.prog_first = R.base;
        m.alg_bytes = R.alg_bytes; m.alg_flops = R.alg_flops; m.n_steps = R.n_steps; m.max_step_cells = R.max_step_cells;
        m.arena_cells = R.arena_cells;
        if (lane == 0) A.meta[b] = m;
    }
}

