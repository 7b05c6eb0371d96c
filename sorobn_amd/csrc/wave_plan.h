// Wave-cooperative device planner: ONE request per 64-lane wave, the request's whole planning state in LDS.
//
// Replaces the bookkeeping half of BayesNet._variable_elimination (sorobn/bayes_net.py:763-789) on the device: relevance pruning
// (763-765), the hidden set (766), evidence slicing (768-776), the choice of the elimination order (the reference takes Python's
// set order, 779) and the elimination loop's factor selection (779-786) - the same decisions, the same step programs WORD FOR WORD
// as order_search.h / emit_core.h (which stay the host's planner and the reference implementation of every rule below), but shaped
// for a wave instead of a thread:
//   * round 3-5's order_kernel / emit_kernel ran that host code one request per LANE: 29 KB of scratch + 325 KB of global state per
//     lane, every dependent access a trip to HBM - SQ counters: waves parked on memory 86 % of their cycles, 8 % issuing
//     (profiles/r06_a_plansq_counters.txt).  Here a request's state is ~10 KB of LDS (created factors are kept only while alive,
//     CPT slices are views of the network tables the workgroup shares, sets are two words), twelve requests per CU;
//   * loops over axes, inputs, vertices (min-fill) and words run on the lanes (wv::for_n / mask / reductions, wave_prims.h); the
//     candidate sweeps of the order search are simulated side by side, one per lane; what is inherently serial - the first-fit
//     arena, the merge of adjacent axes, the choice between the step forms - every lane executes alike (lane 0 writes).
// Compiled for the host (one lane runs every iteration) by oracle/plan_sim.cpp: tests/test_wave_planner.py holds the programs
// against plan_request's on every golden network and on the C3 streams, forwards and with every for_n reversed.
//
// Covered: networks of <= 128 variables whose multi-state variables all have 2^l states (OrderNet::uniform_log2: the min-fill
// search then runs on integer keys), <= kWHints order hints, factors of <= kWAxes axes, <= kWEnt created factors alive at once.
// Anything else: wnet_build returns false (the engine keeps the lane-per-request kernels) or the request reports kEmitErrDevice
// (the host plans its chunk) - never a different program.
#pragma once
#include <cstdint>

#include "emit_core.h"
#include "wave_prims.h"

namespace mibn {

// (hipcc parses a kernel's body in its host pass too: there the functions below are __host__ __device__ over the host primitives)
#define WV_HD MIBN_HD inline
// (the big step forms out of line - __attribute__((noinline)) on emit_sweep / emit_outer / emit_chain_as - were measured worse: the
//  emitter's state then lives in private memory, 1 008 bytes of scratch per lane against 560)
#if defined(__HIP_DEVICE_COMPILE__)
#define WV_LANE0 if (wv::lane() == 0)
#else
#define WV_LANE0 if (true)
#endif

// Phase timers (tools/ubench/wave_plan_bench.hip -DMIBN_WAVE_PROF): a tick books the time since the previous one to phase k, per wave.
#if defined(MIBN_WAVE_PROF) && defined(__HIP_DEVICE_COMPILE__)
struct WProf { unsigned long long t, a[24]; };
#define WV_PROF_ARG , WProf &prof_
#define WV_PROF_PASS , prof_
#define WV_TICK(k) { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); prof_.a[k] += t_ - prof_.t; prof_.t = t_; }
#if MIBN_WAVE_PROF >= 2  // the parts of emit() as phases 13.. of their own (taken out of the phase that called it)
#define WV_ETICK(k) { if (pp_) { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); pp_->a[k] += t_ - pp_->t; pp_->t = t_; } }
#endif
#elif defined(MIBN_WAVE_PROF)
struct WProf { unsigned long long t, a[24]; };
#define WV_PROF_ARG , WProf &prof_
#define WV_PROF_PASS , prof_
#define WV_TICK(k)
#else
#define WV_PROF_ARG
#define WV_PROF_PASS
#define WV_TICK(k)
#endif
#if !defined(WV_ETICK)
#define WV_ETICK(k)
#endif

constexpr int kEmitErrDevice = 7;  // the request exceeds a device limit below: the host plans the chunk (like kEmitErrWords)
constexpr int kWVars = 128;
constexpr int kWHints = 4;
constexpr int kWCsr = 384;     // scope entries of all CPTs together
constexpr int kWAxes = 24;     // axes of a factor / of a step before merging
constexpr int kWEnt = 24;      // created factors alive at once (+ the inputs of the step in flight)
constexpr int kWSims = 4;      // candidate orders simulated side by side
constexpr int kWCands = kWHints + 6;  // meet, reverse, the hints, (effort 1: the meet sweeps a level up and down,) min-fill, (its opening + meet)
constexpr int kWSimEnt = 24;   // factors alive at once in a simulation
constexpr int kWTags = 56;     // work items of a request
constexpr int kWBlocks = 32;   // free blocks of the arena (the host's list holds Arena::kMaxBlocks = 64: a request that would need more goes to the host)
constexpr int kWIns = 160;     // factor handles of one product

// a B2 as it is stored in LDS / in unions (B2 itself has member initialisers: no trivial constructor)
struct B2S {
    uint64_t a, b;
    MIBN_HD operator B2() const { B2 r; r.a = a; r.b = b; return r; }
    MIBN_HD B2S &operator=(const B2 &o) { a = o.a; b = o.b; return *this; }
};

// The network and the options as the wave planner reads them: built once per network / option change on the host (wnet_build),
// copied to LDS by every workgroup.
struct WNet {
    B2S scope[kWVars], fam[kWVars];  // CPT scopes; fam[v] = the CPTs that mention v
    B2S multi;
    int32_t n_vars, n_hints, uniform_log2, csr_n;
    // options (EmitNet / OrderNet)
    int32_t small_cells, prune, outer, fuse, chain, sweep, sweep_min, sweep_canon, tile_h, sweep_iters;
    int64_t big_iters, tile_bytes;
    double log2_small, log2_big, minfill_above, chain_weight, big_cells;
    double second_above;     // effort 1: modelled bytes of the best order above which the runner-up is emitted too
    int32_t big_log2, effort;  // big_cells = 2^big_log2 (-1: not a power of two); OrderNet::effort
    uint32_t pool_off[kWVars];
    int32_t scope_stride[kWCsr];
    uint16_t card[kWVars], scope_off[kWVars + 1];
    uint8_t depth[kWVars], topo_asc[kWVars], topo_desc[kWVars], hint_sorted[kWHints][kWVars];
    uint8_t scope_var[kWCsr];
};

// ---- per-wave state -------------------------------------------------------------------------------------------------------
struct WEnt {  // a created factor: dense, C-order over vars - every axis has 2^l states, axis a the stride 2^(l a) (dstride): whatever
               // order a step form gives its output (emit: layout keys, CHAIN: its own, SWEEP: surviving digits first), it is dense in it
    int32_t cells, n;
    int64_t off;
    B2S scope;
    uint8_t vars[kWAxes];
};
struct WSim {  // one candidate order's byte-model state (a lane each): the created factors alive
    B2S scope[kWSimEnt];
    uint8_t cnt[kWSimEnt];
};
struct WGreedy {
    B2S adj[kWVars];
    uint64_t key[kWVars];
    int32_t miss[kWVars];
    uint8_t gdeg[kWVars];  // effort 1: the degree of every vertex when it was eliminated (OrderScratch::gdeg)
};
struct WStage { int32_t cout, ns, nctrl, loop, f[3], t_off, t_cells, src[3], newv, cvar[3]; uint8_t in[kSweepMaxSmall]; };

struct WState {
    // the request
    uint8_t order[kWVars];   // the elimination order the search chose
    uint16_t ecode[kWVars];
    uint8_t pvar[kWVars];    // layout position -> variable: the elimination order, the query variables behind it
    int8_t pos[kWVars];
    uint32_t cpt_cells[kWVars], cpt_off[kWVars];
    union {
        struct {  // order search
            uint8_t cand[kWCands][kWVars];
            int32_t n_cand[kWCands];
            double cost[kWCands];
            union {
                WSim sim[kWSims];
                WGreedy g;
            };
        } o;
        struct {  // emission
            WEnt ent[kWEnt];
            uint8_t live[kWEnt], live_tmp[kWEnt];  // alive created factors in creation order
            int64_t foff[kWBlocks], fsz[kWBlocks];
            uint8_t hl[kWIns], hs[kSweepMaxSmall + 4];  // handles: factors of the product in flight / of a SWEEP candidate
            int32_t s[kMaxIn][kWAxes], xs[kMaxIn][3];
            int32_t t[18][kWAxes];         // the step forms' axis tables
            uint32_t nout[16], nB[16];
            WStage stg[5];                 // SWEEP: the stages of the candidate pass
            uint32_t hdr[kHdrWords];       // the header of the step in flight (flushed to the program when the step is complete)
            struct {                       // the emitter's cold state (in LDS: as registers every lane would hold it alike - and spill it)
                double alg_bytes, alg_flops, n_steps, max_step_cells, seg_bytes;
                uint32_t seg_first, seg_steps;
            } c;
            Tag tags[kWTags];
        } e;
    };
};

// what the kernel hands back per request (EmitMeta of engine.hip, field for field)
struct WResult {
    uint32_t words, n_tags;
    uint32_t base;  // words of the slot in front of the program (effort 1: the runner-up's program, where it won, stays behind the first one)
    int32_t err;
    double alg_bytes, alg_flops, n_steps, max_step_cells;
    int64_t arena_cells;
};

// ---- small helpers ---------------------------------------------------------------------------------------------------------
WV_HD B2 b2_and(const B2 &x, const B2 &y) { B2 r; r.a = x.a & y.a; r.b = x.b & y.b; return r; }
WV_HD B2 b2_or(const B2 &x, const B2 &y) { B2 r; r.a = x.a | y.a; r.b = x.b | y.b; return r; }
WV_HD B2 b2_andn(const B2 &x, const B2 &y) { B2 r; r.a = x.a & ~y.a; r.b = x.b & ~y.b; return r; }
WV_HD B2 b2_above(int y) {  // the bits > y, 0 <= y < 128
    B2 r;
    const uint64_t lo = (~1ull) << (y & 63);  // bits > (y mod 64) of a word
    r.a = y < 64 ? lo : 0ull;
    r.b = y < 64 ? ~0ull : lo;
    return r;
}
WV_HD int b2_first(const B2 &s) { return s.a ? __builtin_ctzll(s.a) : 64 + __builtin_ctzll(s.b); }

// ===========================================================================================================================
// Order search (order_search.h: order_prepare, order_sweep, order_simulate, order_greedy_impl<true>, order_search)
// ===========================================================================================================================
struct WOrderCtx {
    const WNet &N;
    WState &W;
    B2 rel, hidden, keep;  // keep = the variables that can be axes (multi-state, not evidence)
    bool overflow = false;
    int l;
    double best_cost_ = 0;  // of the order in W.order
    int second_ = -1;       // effort 1: the runner-up's slot in W.o.cand (-1: none) - valid until the emission takes the memory
    WV_HD WOrderCtx(const WNet &n, WState &w) : N(n), W(w), l(n.uniform_log2) {}

    WV_HD double cells_of(const B2 &u) const { return order_pow2(l * b2_count(u)); }

    // SURVEY section 8(d) byte model of one candidate (order_simulate), run by ONE lane: the created factors live in a table of
    // kWSimEnt entries that are recycled as they are consumed; the entries that contain x are found by testing the live ones (a handful).
    // Integer arithmetic: every cell count is a power of two, the weights 8 and 8 x chain_weight are 8 / w8 (wave_view admits the
    // chain weights 1, 1/2, 1/4, 1/8 only), so the cost in bytes is an integer below 2^63 - the number the host computes in doubles
    // (sums of integers < 2^53: exact, the order of the terms does not matter).
    WV_HD double simulate(WSim &S, const uint8_t *order, int n_order, bool &over) const {
        B2 alive0 = rel;
        uint32_t calive = 0;
        uint64_t bytes = 0;
        const uint64_t w8 = (uint64_t)(8.0 * N.chain_weight);
        const int big_log2 = N.big_log2;  // a table is big iff its cells > big_cells: l x count > big_log2 (big_cells = 2^big_log2) or compared as doubles below
        for (int o = 0; o < n_order; ++o) {
            const int x = order[o];
            const B2 m0 = b2_and(N.fam[x], alive0);
            alive0 = b2_andn(alive0, m0);
            B2 u;
            uint64_t in = 0;
            int nbig = 0;
            b2_each(m0, [&](int i) {
                const B2 sc = b2_and(N.scope[i], keep);
                u = b2_or(u, sc);
                const int e2 = l * b2_count(sc);
                in += 1ull << e2;
                nbig += big_log2 >= 0 ? e2 > big_log2 : order_pow2(e2) > N.big_cells;
            });
            uint32_t mc = 0;
            for (uint32_t m = calive; m; m &= m - 1) {
                const int e = __builtin_ctz(m);
                const B2 sc = S.scope[e];
                if (!sc.test(x)) continue;
                mc |= 1u << e;
                u = b2_or(u, sc);
                const int e2 = l * S.cnt[e];
                in += 1ull << e2;
                nbig += big_log2 >= 0 ? e2 > big_log2 : order_pow2(e2) > N.big_cells;
            }
            calive &= ~mc;
            u.clr(x);
            const int ucnt = b2_count(u);
            bytes += (nbig == 1 ? w8 : 8ull) * (in + (1ull << (l * ucnt)));
            const uint32_t free_ = ~calive & ((1u << kWSimEnt) - 1);
            if (!free_ || l * ucnt >= 58) { over = true; return __builtin_inf(); }
            const int e = __builtin_ctz(free_);
            S.scope[e] = u;
            S.cnt[e] = (uint8_t)ucnt;
            calive |= 1u << e;
        }
        B2 u;
        uint64_t in = 0;
        b2_each(alive0, [&](int i) {
            const B2 sc = b2_and(N.scope[i], keep);
            u = b2_or(u, sc);
            in += 1ull << (l * b2_count(sc));
        });
        for (uint32_t m = calive; m; m &= m - 1) {
            const int e = __builtin_ctz(m);
            u = b2_or(u, S.scope[e]);
            in += 1ull << (l * S.cnt[e]);
        }
        if (l * b2_count(u) >= 58) { over = true; return __builtin_inf(); }
        return (double)(bytes + 8ull * (in + (1ull << (l * b2_count(u)))));
    }

    // candidates [c0, c1) side by side, one per lane -> W.o.cost
    WV_HD void simulate_batch(int c0, int c1) {
        bool over = false;
        wv::for_n(c1 - c0, [&](int k) {
            bool ov = false;
            const double c = simulate(W.o.sim[k], W.o.cand[c0 + k], W.o.n_cand[c0 + k], ov);
            W.o.cost[c0 + k] = ov ? -1.0 : c;
        });
        wv::sync();
        for (int k = c0; k < c1; ++k) over = over || W.o.cost[k] < 0;
        overflow = overflow || over;
    }

    // the hidden variables in the order of a per-network list, filtered by depth (order_sweep's `filtered`)
    WV_HD int filtered(uint8_t *out, int base, const uint8_t *sorted_all, int lo_depth, int hi_depth) const {
        const int n = wv::compact_n(N.n_vars, base, [&](int i) {
            const int v = sorted_all[i];
            return hidden.test(v) && (int)N.depth[v] >= lo_depth && (int)N.depth[v] < hi_depth; },
            [&](int i, int k) { out[k] = sorted_all[i]; });
        return n;
    }

    // greedy min-fill on the lanes (order_greedy_impl<true>: integer keys): a vertex per lane; -> W.o.cand[slot]; false: cannot win
    WV_HD bool greedy(int slot, double abort_above) {
        WGreedy &G = W.o.g;
        const int nv = N.n_vars;
        wv::for_n(nv, [&](int v) {
            if (!rel.test(v)) return;
            B2 a;
            if (keep.test(v)) {
                b2_each(b2_and(N.fam[v], rel), [&](int i) { a = b2_or(a, b2_and(N.scope[i], keep)); });
                a.clr(v);
            }
            G.adj[v] = a;
        });
        wv::sync();
        auto pack = [&](int x, int missing, int degree) {
            return ((uint64_t)(uint32_t)(missing * 64 + l * degree) << 16) | ((uint64_t)N.depth[x] << 8) | (uint64_t)x;
        };
        wv::for_n(nv, [&](int x) {
            if (!hidden.test(x)) return;
            const B2 ax = G.adj[x];
            int missing = 0;
            b2_each(ax, [&](int y) { missing += __builtin_popcountll(ax.a & ~G.adj[y].a) + __builtin_popcountll(ax.b & ~G.adj[y].b) - 1; });
            G.miss[x] = missing;
            G.key[x] = pack(x, missing, b2_count(ax));
        });
        wv::sync();
        B2 alive = hidden;
        const int total = b2_count(hidden);
        double created = 0;
        uint8_t *cand = W.o.cand[slot];
        int n = 0;
        for (int it = 0; it < total; ++it) {
            const uint64_t kbest = wv::min_u64_n(nv, [&](int x) { return alive.test(x) ? G.key[x] : ~0ull; });
            const int best = (int)(kbest & 0xff);
            const B2 nb = G.adj[best];
            WV_LANE0 { cand[n] = (uint8_t)best; G.gdeg[n] = (uint8_t)b2_count(nb); }
            ++n;
            alive.clr(best);
            created += order_pow2(l * b2_count(nb));  // (= ws[best]: l x degree, kept current for every live vertex)
            if (16.0 * N.chain_weight * created > abort_above) { WV_LANE0 W.o.n_cand[slot] = n; wv::sync(); return false; }
            // every common neighbour z of a pair (y, u) of nb that becomes adjacent loses that pair (two ordered pairs) from its count
            wv::for_n(nv, [&](int z) {
                if (!rel.test(z) || nb.test(z) || z == best) return;
                const B2 S = b2_and(nb, G.adj[z]);
                if (!S.any()) return;
                int cnt = 0;
                b2_each(S, [&](int y) {
                    const B2 up = b2_above(y);
                    cnt += __builtin_popcountll(S.a & ~G.adj[y].a & up.a) + __builtin_popcountll(S.b & ~G.adj[y].b & up.b);
                });
                if (cnt) { G.miss[z] -= 2 * cnt; G.key[z] -= (uint64_t)cnt * ((uint64_t)128 << 16); }
            });
            wv::sync();
            // nb becomes a clique; its live members recount: a missing pair of y has at least one end outside nb
            wv::for_n(nv, [&](int y) {
                if (!nb.test(y)) return;
                B2 ay = b2_or(G.adj[y], nb);
                ay.clr(best);
                ay.clr(y);
                G.adj[y] = ay;
                if (!alive.test(y)) return;
                const B2 ext = b2_andn(ay, nb);
                B2 nbm = nb;
                nbm.clr(y);
                int missing = 0;
                b2_each(ext, [&](int e) {
                    const uint64_t na = ~G.adj[e].a, nb_ = ~G.adj[e].b;
                    missing += __builtin_popcountll(nbm.a & na) + __builtin_popcountll(nbm.b & nb_) + __builtin_popcountll(ay.a & na) +
                               __builtin_popcountll(ay.b & nb_) - 1;
                });
                G.miss[y] = missing;
                G.key[y] = pack(y, missing, b2_count(ay));
            });
            wv::sync();
        }
        WV_LANE0 W.o.n_cand[slot] = n;
        wv::sync();
        return true;
    }

    // order_search: the whole search of one request -> W.order, returns the number of hidden variables in it
    WV_HD int search(int nq, const int32_t *qvars, int ne, const int32_t *evars, const B2 *anc, bool no_prune WV_PROF_ARG) {
        B2 qb, eb;
        rel = B2{};
        for (int i = 0; i < nq; ++i) { const int v = qvars[i]; qb.set(v); rel.set(v); rel = b2_or(rel, anc[v]); }
        for (int i = 0; i < ne; ++i) { const int v = evars[i]; eb.set(v); rel.set(v); rel = b2_or(rel, anc[v]); }
        if (!N.prune || no_prune)
            for (int v = 0; v < N.n_vars; ++v) rel.set(v);
        hidden = b2_andn(b2_andn(rel, qb), eb);
        keep = b2_andn(N.multi, eb);
        hidden = b2_and(hidden, N.multi);
        if (!hidden.any()) return 0;
        int qdepth = 0x7fffffff;
        for (int i = 0; i < nq; ++i) qdepth = (int)N.depth[qvars[i]] < qdepth ? (int)N.depth[qvars[i]] : qdepth;
        const int kNoDepth = 0x7fffffff;
        // the sweeps: meet, reverse, the hints (effort 1: the meet sweep a level above and a level below the query's)
        const bool two = N.effort >= 1;
        int nc = 0;
        {
            int n = filtered(W.o.cand[0], 0, N.topo_asc, 0, qdepth);
            n = filtered(W.o.cand[0], n, N.topo_desc, qdepth, kNoDepth);
            WV_LANE0 W.o.n_cand[0] = n;
            n = filtered(W.o.cand[1], 0, N.topo_desc, 0, kNoDepth);
            WV_LANE0 W.o.n_cand[1] = n;
            nc = 2;
            for (int h = 0; h < N.n_hints; ++h) {
                n = filtered(W.o.cand[nc], 0, N.hint_sorted[h], 0, kNoDepth);
                WV_LANE0 W.o.n_cand[nc] = n;
                ++nc;
            }
            if (two && nq > 0)
                for (int d = -1; d <= 1; d += 2) {
                    if (qdepth + d < 0) continue;
                    n = filtered(W.o.cand[nc], 0, N.topo_asc, 0, qdepth + d);
                    n = filtered(W.o.cand[nc], n, N.topo_desc, qdepth + d, kNoDepth);
                    WV_LANE0 W.o.n_cand[nc] = n;
                    ++nc;
                }
        }
        wv::sync();
        WV_TICK(0)  // relevant set, candidate sweeps
        for (int c0 = 0; c0 < nc; c0 += kWSims) simulate_batch(c0, c0 + kWSims < nc ? c0 + kWSims : nc);
        WV_TICK(1)  // byte model of the sweeps
        if (overflow) return -1;
        // the best and (effort 1) the runner-up - the cheapest candidate whose ORDER differs from the best's - in the host's
        // sequence of candidates: the first of equals wins (order_search's consider())
        auto same_order = [&](int a, int b) {
            if (W.o.n_cand[a] != W.o.n_cand[b]) return false;
            return !wv::any_n(W.o.n_cand[a], [&](int i) { return W.o.cand[a][i] != W.o.cand[b][i]; });
        };
        double best_cost = __builtin_inf(), second_cost = __builtin_inf();
        int best = -1, second = -1;
        auto rank = [&](int c0, int c1) {
            for (int c = c0; c < c1; ++c) {
                const double cost = W.o.cost[c];
                if (cost < best_cost) {
                    if (two && best >= 0) { second_cost = best_cost; second = best; }
                    best_cost = cost;
                    best = c;
                } else if (two && cost < second_cost && !same_order(c, best)) {
                    second_cost = cost;
                    second = c;
                }
            }
        };
        rank(0, nc);
        // greedy min-fill where the sweeps cost more than minfill_above
        if (best_cost > N.minfill_above * N.chain_weight) {
            const bool whole = greedy(nc, two ? second_cost : best_cost);
            WV_TICK(2)  // min-fill
            int n_new = whole ? 1 : 0, opening_slot = -1;
            if (two) {
                // min-fill's opening (its leading eliminations that create factors of at most kOrderOpening variables), then the
                // meet sweep of the rest - built before the simulations take the greedy state's memory
                const int ng = W.o.n_cand[nc];
                const uint64_t small = wv::mask64(ng < 64 ? ng : 64, [&](int i) { return W.o.g.gdeg[i] <= kOrderOpening; });
                // (an opening is a handful of eliminations; 64 or more of them in a row: the request goes to the host)
                if (small == ~0ull) return -1;
                const int np = (int)__builtin_ctzll(~small);
                if (np > 0 && !(np == ng && whole)) {
                    const int slot = nc + n_new;  // (an aborted min-fill order is no candidate: the opening is built in place)
                    const B2 all_hidden = hidden;
                    if (slot != nc) wv::for_n(np, [&](int i) { W.o.cand[slot][i] = W.o.cand[nc][i]; });
                    for (int i = 0; i < np; ++i) hidden.clr(W.o.cand[nc][i]);
                    int n = filtered(W.o.cand[slot], np, N.topo_asc, 0, qdepth);
                    n = filtered(W.o.cand[slot], n, N.topo_desc, qdepth, kNoDepth);
                    WV_LANE0 W.o.n_cand[slot] = n;
                    hidden = all_hidden;
                    opening_slot = slot;
                    ++n_new;
                    wv::sync();
                }
            }
            if (n_new) {
                simulate_batch(nc, nc + n_new);
                WV_TICK(3)  // their byte model
                if (overflow) return -1;
                if (opening_slot >= 0) {  // (order_search.h: the model overrates this candidate)
                    WV_LANE0 W.o.cost[opening_slot] *= kOrderOpeningPenalty;
                    wv::sync();
                }
                rank(nc, nc + n_new);
            }
        }
        best_cost_ = best_cost;
        second_ = second;
        const int nb = W.o.n_cand[best];
        wv::for_n(nb, [&](int i) { W.order[i] = W.o.cand[best][i]; });
        wv::sync();
        return nb;
    }
};


#if defined(MIBN_WAVE_COUNT) && !defined(__HIP_DEVICE_COMPILE__)  // (tools/wave_check.cpp: how often each step form is tried / emitted)
static long g_wave_count[32];
#define WV_COUNT(k) (++g_wave_count[k])
#else
#define WV_COUNT(k)
#endif
// ===========================================================================================================================
// Emission (emit_core.h: Arena, Emitter, emit_begin, emit_run, tag_program).  Factor handles: h < 128 = the evidence-sliced CPT
// of variable h (a view of the network tables: nothing is stored but its cells and its offset), 128 + e = created factor e.
// ===========================================================================================================================
// rows of WState::e.t
enum { T_RCARD, T_ROST, T_RTST, T_RB0, T_RB1, T_MC, T_MO, T_MT, T_MB0, T_MB1, T_NAX, T_RAX, T_CTRL, T_MAP, T_A, T_B, T_C, T_F };

// A value every lane of the wave holds alike and that lives across whole steps (the emitter's counters and masks): every update
// goes through wv::uni, so the compiler keeps it in a SCALAR register - scalar registers spill into the lanes of a vector register,
// vector registers into scratch memory, and at 128 vector registers the planner spills (profiles/NOTES_r06.md, session AP).
// The host build: a plain T.
#if defined(MIBN_WAVE_NO_UNI)
template <class T> using WUni = T;
#else
template <class T> struct WUni {
    T v;
    WV_HD WUni() : v() {}
    WV_HD WUni(T x) : v(wv::uni(x)) {}
    WV_HD operator T() const { return v; }
    WV_HD WUni &operator=(T x) { v = wv::uni(x); return *this; }
    WV_HD WUni &operator+=(T x) { v = wv::uni((T)(v + x)); return *this; }
    WV_HD WUni &operator-=(T x) { v = wv::uni((T)(v - x)); return *this; }
    WV_HD WUni &operator|=(T x) { v = wv::uni((T)(v | x)); return *this; }
    WV_HD WUni &operator&=(T x) { v = wv::uni((T)(v & x)); return *this; }
    WV_HD WUni &operator++() { v = wv::uni((T)(v + 1)); return *this; }
    WV_HD WUni &operator--() { v = wv::uni((T)(v - 1)); return *this; }
};
#endif

struct WEmit {
    const WNet &N;
    WState &W;
    // where the words go: the request's slot of the chunk's program buffer (EmitBuf, device branch)
    uint32_t *data;
    WUni<uint32_t> size = 0u, cap;
    WUni<bool> overflow = false;
    // (the statistics - EmitStats - and the open segment of the work items live in W.e.c)
    WUni<int> err = 0;
    // the request
    B2 keep;                 // the variables that can be axes (multi-state, not evidence)
    B2 alive0;               // CPT slices not yet consumed
    WUni<int> n_live = 0;          // created factors alive (W.e.live: creation order)
    WUni<uint32_t> ent_busy = 0u;  // entries in use: alive, or an input / the output of the step in flight
    WUni<uint32_t> consumed_ents = 0u;  // entries consumed by the step in flight (their data stays until the step is complete)
    const uint8_t *cur_hl = nullptr;  // the inputs of the step being emitted (handles)
    WUni<int> npos = 0;            // layout positions in use (W.pvar)
    WUni<int> n_pool = 0, pool_cap = 0;  // the host's pool accounting (its limits are part of the program's definition)
    WUni<int> nf = 0;              // arena: free blocks (W.e.foff / fsz), top
    WUni<int64_t> top = (int64_t)0;
    // work items (tag_program, step by step)
    WUni<int> n_tags = 0, level = 0;
#if defined(MIBN_WAVE_PROF)
    WProf *pp_ = nullptr;
#endif

    WV_HD WEmit(const WNet &n, WState &w, uint32_t *slot, uint32_t cap_) : N(n), W(w), data(slot), cap(cap_) {
        WV_LANE0 { W.e.c.alg_bytes = W.e.c.alg_flops = W.e.c.n_steps = W.e.c.max_step_cells = W.e.c.seg_bytes = 0; W.e.c.seg_first = W.e.c.seg_steps = 0; }
        wv::sync();
    }
    // the step's share of the statistics (EmitStats)
    WV_HD void book(double bytes, double flops, double step_cells) {
        WV_LANE0 {
            W.e.c.alg_bytes += bytes;
            W.e.c.alg_flops += flops;
            W.e.c.max_step_cells = emit_max(W.e.c.max_step_cells, step_cells);
            W.e.c.n_steps += 1;
        }
    }

    WV_HD int card(int v) const { return (int)N.card[v]; }
    WV_HD int32_t dstride(int a) const { return (int32_t)(1u << (N.uniform_log2 * a)); }  // stride of axis a of a created factor

    // ---- factors ------------------------------------------------------------------------------------------------------------
    // (measured and not kept, profiles/r06_l_uni.log: wv::uni on the handles and on what the accessors return - scalar registers, scalar
    //  branches - costs more in spilled scalar registers than it saves in vector instructions: 14.5 against 13.6 ms per chunk)
    WV_HD int64_t fcells(int h) const { return h < kWVars ? (int64_t)W.cpt_cells[h] : (int64_t)W.e.ent[h - kWVars].cells; }
    WV_HD B2 fscope(int h) const { return h < kWVars ? b2_and(N.scope[h], keep) : (B2)W.e.ent[h - kWVars].scope; }
    WV_HD int64_t falloc(int h) const { return h < kWVars ? 0 : (int64_t)W.e.ent[h - kWVars].cells; }
    WV_HD uint64_t foffset(int h) const { return h < kWVars ? ((uint64_t)W.cpt_off[h] | kConstFlag) : (uint64_t)W.e.ent[h - kWVars].off; }
    WV_HD uint64_t foffset_v(int h) const { return foffset(h); }
    // stride of variable v in factor h (0: not an axis)
    WV_HD int64_t fstride_of(int h, int v) const {
        int64_t st = 0;
        if (h < kWVars) {
            for (int k = N.scope_off[h]; k < N.scope_off[h + 1]; ++k)
                if (N.scope_var[k] == v && keep.test(v)) st = N.scope_stride[k];
        } else {
            const WEnt &E = W.e.ent[h - kWVars];
            for (int a = 0; a < E.n; ++a)
                if (E.vars[a] == v) st = dstride(a);
        }
        return st;
    }
    WV_HD int ent_alloc() {  // a free entry for the factor a step creates (-1: none - the request goes to the host)
        const uint32_t free_ = ~ent_busy & ((1u << kWEnt) - 1);
        if (!free_) { err = kEmitErrDevice; return -1; }
        const int e = __builtin_ctz(free_);
        ent_busy |= 1u << e;
        return e;
    }

    // ---- arena (Arena of emit_core.h: first fit, free list sorted by offset) -------------------------------------------------
    WV_HD int64_t arena_alloc(int64_t n) {
        n = (n + 15) & ~int64_t(15);
        int64_t *foff = W.e.foff, *fsz = W.e.fsz;
        const uint64_t fit = wv::mask64(nf, [&](int i) { return fsz[i] >= n; });
        if (fit) {
            const int i = __builtin_ctzll(fit);
            const int64_t o = foff[i], left = fsz[i] - n;
            if (left) {
                WV_LANE0 { foff[i] = o + n; fsz[i] = left; }
            } else {
                WV_LANE0 for (int k = i; k + 1 < nf; ++k) { foff[k] = foff[k + 1]; fsz[k] = fsz[k + 1]; }
                --nf;
            }
            wv::sync();
            return o;
        }
        if (nf && foff[nf - 1] + fsz[nf - 1] == top) {  // grow the free block that touches the top
            const int64_t o = foff[nf - 1];
            top = o + n;
            --nf;
            return o;
        }
        const int64_t o = top;
        top += n;
        return o;
    }
    WV_HD void arena_release(int64_t o, int64_t n) {
        n = (n + 15) & ~int64_t(15);
        int64_t *foff = W.e.foff, *fsz = W.e.fsz;
        const int i = __builtin_popcountll(wv::mask64(nf, [&](int k) { return foff[k] < o; }));  // (sorted: the blocks below o)
        const bool left = i > 0 && foff[i - 1] + fsz[i - 1] == o;
        const bool right = i < nf && o + n == foff[i];
        if (left && right) {
            WV_LANE0 {
                fsz[i - 1] += n + fsz[i];
                for (int k = i; k + 1 < nf; ++k) { foff[k] = foff[k + 1]; fsz[k] = fsz[k + 1]; }
            }
            --nf;
        } else if (left) {
            WV_LANE0 fsz[i - 1] += n;
        } else if (right) {
            WV_LANE0 { foff[i] = o; fsz[i] += n; }
        } else if (nf < kWBlocks) {
            WV_LANE0 {
                for (int k = nf; k > i; --k) { foff[k] = foff[k - 1]; fsz[k] = fsz[k - 1]; }
                foff[i] = o;
                fsz[i] = n;
            }
            ++nf;
        } else {
            err = kEmitErrDevice;  // (the host's list is twice as long and leaks beyond that: not reproduced here)
        }
        wv::sync();
    }

    // ---- words ----------------------------------------------------------------------------------------------------------------
    WV_HD uint32_t *extend(int words) {
        if ((size_t)size + (size_t)words + kMaxStepWords > cap) { overflow = true; return data + (cap - kMaxStepWords); }
        uint32_t *p = data + size;
        size += (uint32_t)words;
        return p;
    }
    WV_HD void put_off(uint32_t *&p, uint64_t off) {
        WV_LANE0 { p[0] = (uint32_t)(off & 0xffffffffu); p[1] = (uint32_t)(off >> 32); }
        p += 2;
    }
    WV_HD void put(uint32_t *&p, uint32_t v) {
        WV_LANE0 p[0] = v;
        ++p;
    }
    // the header of the step in flight, kept in LDS until the step is complete (flush_header)
    WV_HD void header(uint32_t kind, int n_in, int ma, int mlo, int cx, bool final_, int64_t lo, int64_t hi, uint64_t out_off, int words) {
        uint32_t *w = W.e.hdr;
        WV_LANE0 {
            w[0] = kind | ((uint32_t)n_in << 8) | ((uint32_t)ma << 16) | ((uint32_t)mlo << 24);
            w[1] = (uint32_t)cx | ((final_ ? kFlagFinal : 0u) << 16);
            w[2] = (uint32_t)lo;
            w[3] = (uint32_t)hi;
            w[4] = (uint32_t)(out_off & 0xffffffffu);
            w[5] = (uint32_t)(out_off >> 32);
            w[6] = (uint32_t)words;
            w[7] = w[8] = w[9] = 0;
        }
        wv::sync();
    }
    WV_HD void hdr_or(int i, uint32_t v) { WV_LANE0 W.e.hdr[i] |= v; }
    WV_HD void hdr_set(int i, uint32_t v) { WV_LANE0 W.e.hdr[i] = v; }
    WV_HD void flush_header(uint32_t *w) {
        wv::sync();
        wv::for_n(kHdrWords, [&](int i) { w[i] = W.e.hdr[i]; });
    }

    // ---- GENERIC encoding (emit_generic): iteration space = output cells -----------------------------------------------------
    WV_HD void emit_generic(int n_in, WEnt &out, int64_t cells, int cx, bool final_, uint32_t *&wout) {
        const int na = out.n, l = N.uniform_log2;
        const int64_t K = int64_t(1) << l;  // (every axis is a multi-state variable: 2^l states)
        int nlo = 0;
        int64_t lo = 1;
        const int64_t lomax = n_in <= 3 ? kLoMax : kLoTarget;
        while (nlo < na && lo < kLoTarget && lo * K <= lomax) { lo *= K; ++nlo; }
        // merge adjacent axes that are contiguous in every input: axis a continues the block of a - 1 iff every input's stride
        // along a is its stride along a - 1 times that axis' cardinality (the block's start stride times its cells so far)
        int32_t (*s)[kWAxes] = W.e.s;
        const uint64_t cont = wv::mask64(na, [&](int a) {
            if (a == 0 || a == nlo) return false;
            bool m = true;
            for (int j = 0; j < n_in; ++j) m = m && (int64_t)s[j][a] == (int64_t)s[j][a - 1] * K;
            return m;
        });
        int32_t *mcard = W.e.t[T_MC], *start = W.e.t[T_MAP];
        int ma = 0, mlo = 0;
        if (l * na < 30) {
            // (no block can reach 2^30 cells: the blocks are the runs between the axes that do not continue their predecessor)
            const uint32_t bnd = ~(uint32_t)cont & (na < 32 ? (1u << na) - 1 : ~0u);
            ma = __builtin_popcount(bnd);
            mlo = __builtin_popcount(bnd & ((1u << nlo) - 1));
            wv::for_n(ma, [&](int m) {
                uint32_t b = bnd;
                for (int q = 0; q < m; ++q) b &= b - 1;
                const int a0 = __builtin_ctz(b);
                b &= b - 1;
                const int a1 = b ? __builtin_ctz(b) : na;
                start[m] = a0;
                mcard[m] = (int32_t)(1u << (l * (a1 - a0)));
            });
        } else {
            uint64_t blk = 0;
            for (int a = 0; a < na; ++a) {
                if (((cont >> a) & 1) && ma > 0 && blk * (uint64_t)K < (1u << 30)) {
                    blk *= (uint64_t)K;
                    WV_LANE0 mcard[ma - 1] = (int32_t)blk;
                } else {
                    WV_LANE0 { mcard[ma] = (int32_t)K; start[ma] = a; }
                    blk = (uint64_t)K;
                    ++ma;
                    if (a < nlo) ++mlo;
                }
            }
        }
        wv::sync();
        if (ma > kMaxAxes) { err = kEmitErrStepAxes; return; }
        const int nbw = 3 * n_in, body = nbw + ma + n_in * ma;
        const int words = kHdrWords + body;
        uint32_t *w = extend(words);
        wout = w;
        header(kKindGeneric, n_in, ma, mlo, cx, final_, lo, cells / lo, (uint64_t)out.off, words);
        uint32_t *p = w + kHdrWords;
        const uint8_t *hl = cur_hl;
        wv::for_n(n_in * 32, [&](int i) {  // lane (j, a): the three words of input j (a < 3), the merged cardinalities (j = 0), its merged strides
            const int j = i >> 5, a = i & 31;
            if (a < 3) {
                const uint64_t off = foffset_v(hl[j]);
                p[3 * j + a] = a == 0 ? (uint32_t)(off & 0xffffffffu) : (a == 1 ? (uint32_t)(off >> 32) : (uint32_t)W.e.xs[j][0]);
            }
            if (a < ma) {
                if (j == 0) p[nbw + a] = (uint32_t)mcard[a];
                p[nbw + ma + j * ma + a] = (uint32_t)s[j][start[a]];
            }
        });
        (void)body;
    }

    WV_HD static int nth_bit(uint32_t m, int k) {
        for (; k > 0; --k) m &= m - 1;
        return __builtin_ctz(m);
    }

    // the row stride of the MFMA form over the lane-varying axes (emit_fiber's rule): dep(i) = the axis carries a T / B dependence
    template <class Dep> WV_HD int row_stride_rule(int nlo, const int32_t *rcard, Dep dep, bool &ok_out) {
        int row_stride = 16, inside = 0;
        bool ok = true;
        int64_t cs = 1;
        for (int i = 0; i < nlo; ++i) {
            if (dep(i)) {
                if (cs < 64) {
                    ++inside;
                    if (rcard[i] == 4 && (cs == 1 || cs == 4 || cs == 16)) row_stride = (int)cs;
                    else ok = false;
                } else if (cs % 64 != 0) {
                    ok = false;
                }
            }
            cs *= rcard[i];
        }
        ok_out = ok && inside <= 1;
        return row_stride;
    }

    // merge adjacent R axes that are contiguous in the output, in T and in the big inputs (tables T_RCARD.. read through T_MAP):
    // fills T_MC / T_MO / T_MT / T_MB0 / T_MB1, returns ma (mlo by reference)
    WV_HD int merge_r_axes(int nr, int nlo, int nb, bool use_map, int &mlo_out) {
        int32_t (*t)[kWAxes] = W.e.t;
        auto at = [&](int i) { return use_map ? t[T_MAP][i] : i; };
        const uint64_t cont = wv::mask64(nr, [&](int i) {
            if (i == 0 || i == nlo) return false;
            const int q = at(i), r = at(i - 1);
            const int64_t c = t[T_RCARD][r];
            bool m = (int64_t)t[T_ROST][r] * c == t[T_ROST][q] && (int64_t)t[T_RTST][r] * c == t[T_RTST][q];
            if (nb > 0) m = m && (int64_t)t[T_RB0][r] * c == t[T_RB0][q];
            if (nb > 1) m = m && (int64_t)t[T_RB1][r] * c == t[T_RB1][q];
            return m;
        });
        int ma = 0, mlo = 0;
        int64_t blk = 0;
        for (int i = 0; i < nr; ++i) {
            const int q = at(i);
            const int64_t c = t[T_RCARD][q];
            if (((cont >> i) & 1) && ma > 0 && blk * c < (1 << 30)) {
                blk *= c;
                WV_LANE0 t[T_MC][ma - 1] = (int32_t)blk;
            } else {
                WV_LANE0 { t[T_MC][ma] = (int32_t)c; t[T_MO][ma] = t[T_ROST][q]; t[T_MT][ma] = t[T_RTST][q]; t[T_MB0][ma] = t[T_RB0][q]; t[T_MB1][ma] = t[T_RB1][q]; }
                blk = c;
                ++ma;
                if (i < nlo) ++mlo;
            }
        }
        wv::sync();
        mlo_out = mlo;
        return ma;
    }

    // the wave-uniform (hi) axes only one of two big tables depends on run fastest: a stable partition of [nlo, nr) -> T_MAP
    WV_HD void order_hi_axes(int nr, int nlo) {
        int32_t (*t)[kWAxes] = W.e.t;
        wv::for_n(nlo, [&](int i) { t[T_MAP][i] = i; });
        int k = nlo;
        for (int pass = 0; pass < 2; ++pass)
            k = wv::compact_n(nr - nlo, k, [&](int i) { const int q = nlo + i; return ((t[T_RB0][q] == 0) != (t[T_RB1][q] == 0)) == (pass == 0); },
                              [&](int i, int pos) { t[T_MAP][pos] = nlo + i; });
        wv::sync();
    }

    // ---- FIBER encoding (emit_fiber); false: the step does not fit the form --------------------------------------------------
    WV_HD bool emit_fiber(int n_in, WEnt &out, int cx, int c1, uint32_t *&wout) {
        int32_t (*t)[kWAxes] = W.e.t;
        int32_t (*s)[kWAxes] = W.e.s;
        const int na = out.n;
        uint32_t bigm = 0;
        for (int j = 0; j < n_in; ++j)
            if (fcells(cur_hl[j]) > N.small_cells) bigm |= 1u << j;
        const int nb = __builtin_popcount(bigm), ns = n_in - nb;
        const uint32_t smallm = ~bigm & ((1u << n_in) - 1);
        if (nb < 1 || nb > 2 || ns > kMaxSmall || cx > kMaxCx) return false;
        const int big0 = nth_bit(bigm, 0), big1 = nb > 1 ? nth_bit(bigm, 1) : big0;
        // N axes: no big input depends on them; keep at most kMaxNC combinations (fastest axes first)
        const uint64_t freem = wv::mask64(na, [&](int a) { return s[big0][a] == 0 && (nb < 2 || s[big1][a] == 0); });
        int nN = 0, nr = 0;
        int64_t NC = 1;
        for (int a = 0; a < na; ++a) {
            const int c = card(out.vars[a]);
            if (((freem >> a) & 1) && NC * c <= kMaxNC && nN < 15) { WV_LANE0 t[T_NAX][nN] = a; ++nN; NC *= c; }
            else { WV_LANE0 t[T_RAX][nr] = a; ++nr; }
        }
        wv::sync();
        // R-axis tables (unmerged); ctrl axes = R axes a small input depends on
        wv::for_n(nr, [&](int i) {
            const int a = t[T_RAX][i];
            t[T_RCARD][i] = card(out.vars[a]);
            t[T_ROST][i] = dstride(a);
            t[T_RB0][i] = s[big0][a];
            t[T_RB1][i] = nb > 1 ? s[big1][a] : 0;
            t[T_RTST][i] = 0;
        });
        const uint64_t depm = wv::mask64(nr, [&](int i) {
            const int a = t[T_RAX][i];
            bool dep = false;
            for (uint32_t m = smallm; m; m &= m - 1) dep = dep || s[__builtin_ctz(m)][a] != 0;
            return dep;
        });
        wv::sync();
        int nctrl = 0;
        int64_t T = NC * cx;
        for (uint64_t m = depm; m; m &= m - 1) {
            const int i = __builtin_ctzll(m);
            if (nctrl >= 15) return false;
            WV_LANE0 { t[T_CTRL][nctrl] = t[T_RAX][i]; t[T_RTST][i] = (int32_t)T; }
            ++nctrl;
            T *= t[T_RCARD][i];
            if (T > kMaxT) return false;
        }
        wv::sync();
        const int nT = nN + nctrl;
        int nlo = 0;
        int64_t lo = 1;
        while (nlo < nr && lo < kLoTarget && lo * t[T_RCARD][nlo] <= kFiberLoMax) lo *= t[T_RCARD][nlo++];
        int64_t rcells = 1;
        for (int i = 0; i < nr; ++i) rcells *= t[T_RCARD][i];
        if (rcells * NC < N.big_iters) return false;
        bool contig = true;
        {
            int64_t expect = NC;
            for (int i = 0; i < nlo; ++i) { contig = contig && t[T_ROST][i] == expect; expect *= t[T_RCARD][i]; }
        }
        if (nb == 2) order_hi_axes(nr, nlo);
        int mlo = 0;
        const int ma = merge_r_axes(nr, nlo, nb, nb == 2, mlo);
        if (ma > kMaxAxes) return false;
        const int words = kHdrWords + 4 * nb + ns * (4 + nT) + nT + (int)NC + 3 * ma + nb * ma;
        if (words > kMaxStepWords) return false;
        wv::for_n((int)NC, [&](int n) {
            int64_t r = n, off = 0;
            for (int i = 0; i < nN; ++i) {
                const int a = t[T_NAX][i], c = card(out.vars[a]);
                off += (r % c) * dstride(a);
                r /= c;
            }
            W.e.nout[n] = (uint32_t)off;
        });
        wv::sync();
        contig = contig && !wv::any_n((int)NC, [&](int n) { return W.e.nout[n] != (uint32_t)n; });
        uint32_t *w = extend(words);
        wout = w;
        header(kKindFiber, nb + ns, ma, mlo, cx, false, lo, rcells / lo, (uint64_t)out.off, words);
        {
            bool ok = true;
            int row_stride = row_stride_rule(nlo, t[T_RCARD], [&](int i) { return t[T_RTST][i] != 0; }, ok);
            if (!ok) row_stride = 0;
            hdr_or(1, (contig ? kFlagContig << 16 : 0u) | ((uint32_t)row_stride << kRowStrideShift));
        }
        hdr_set(7, (uint32_t)nb | ((uint32_t)ns << 4) | ((uint32_t)nN << 8) | ((uint32_t)nctrl << 12) | ((uint32_t)NC << 16));
        hdr_set(8, (uint32_t)T | ((uint32_t)c1 << 16));
        uint32_t *p = w + kHdrWords;
        for (int b = 0; b < nb; ++b) {
            const int j = b ? big1 : big0;
            put_off(p, foffset(cur_hl[j]));
            put(p, (uint32_t)W.e.xs[j][0]);
            put(p, (uint32_t)W.e.xs[j][1]);
        }
        for (uint32_t m = smallm; m; m &= m - 1) {
            const int j = __builtin_ctz(m);
            put_off(p, foffset(cur_hl[j]));
            put(p, (uint32_t)W.e.xs[j][0]);
            put(p, (uint32_t)W.e.xs[j][1]);
            wv::for_n(nT, [&](int i) { p[i] = (uint32_t)s[j][i < nN ? t[T_NAX][i] : t[T_CTRL][i - nN]]; });
            p += nT;
        }
        wv::for_n(nT, [&](int i) { p[i] = (uint32_t)card(out.vars[i < nN ? t[T_NAX][i] : t[T_CTRL][i - nN]]); });
        p += nT;
        wv::for_n((int)NC, [&](int n) { p[n] = W.e.nout[n]; });
        p += NC;
        wv::for_n(ma, [&](int a) { p[3 * a] = (uint32_t)t[T_MC][a]; p[3 * a + 1] = (uint32_t)t[T_MO][a]; p[3 * a + 2] = (uint32_t)t[T_MT][a]; });
        p += 3 * ma;
        wv::for_n(nb * ma, [&](int i) { p[i] = (uint32_t)t[i < ma ? T_MB0 : T_MB1][i % ma]; });
        return true;
    }

    // ---- OUTER encoding (emit_outer) ------------------------------------------------------------------------------------------
    // one role assignment: B = big input `b`, A = `a_`; fills the R tables; -> accepted (key below best_key and feasible)
    struct OuterTry { bool ok; int64_t key; int nax0, nax1, nr, nlo, row_stride, nctrl; int64_t lo, rcells, T; };
    WV_HD OuterTry outer_try(int b, int a_, int ns, uint32_t smallm, WEnt &out, int cx, int64_t best_key) {
        int32_t (*t)[kWAxes] = W.e.t;
        int32_t (*s)[kWAxes] = W.e.s;
        OuterTry r;
        r.ok = false;
        const int na = out.n;
        const uint64_t fm = wv::mask64(na, [&](int ax) { return s[b][ax] != 0 && s[a_][ax] == 0 && card(out.vars[ax]) == 4; });
        if (__builtin_popcountll(fm) < 2) return r;
        r.nax0 = __builtin_ctzll(fm);
        r.nax1 = __builtin_ctzll(fm & (fm - 1));
        r.key = (int64_t)dstride(r.nax0) + dstride(r.nax1);
        if (r.key >= best_key) return r;
        // R axes: everything but the two N axes; ctrl axes = R axes a small input depends on
        const int nr = na - 2;
        wv::for_n(nr, [&](int i) {
            const int ax = i + (i >= r.nax0) + (i + (i >= r.nax0) >= r.nax1);
            t[T_RAX][i] = ax;
            t[T_RCARD][i] = card(out.vars[ax]);
            t[T_ROST][i] = dstride(ax);
            t[T_RB0][i] = s[a_][ax];
            t[T_RB1][i] = s[b][ax];
            t[T_RTST][i] = 0;
        });
        wv::sync();
        const uint64_t depm = wv::mask64(nr, [&](int i) {
            const int ax = t[T_RAX][i];
            bool dep = false;
            for (uint32_t m = smallm; m; m &= m - 1) dep = dep || s[__builtin_ctz(m)][ax] != 0;
            return dep;
        });
        int nctrl = 0;
        int64_t T = ns ? 16 * (int64_t)cx : 0;
        bool ok = true;
        for (uint64_t m = depm; m; m &= m - 1) {
            const int i = __builtin_ctzll(m);
            if (nctrl >= 13) ok = false;
            WV_LANE0 { if (nctrl < kWAxes) t[T_CTRL][nctrl] = t[T_RAX][i]; t[T_RTST][i] = (int32_t)T; }
            ++nctrl;
            T *= t[T_RCARD][i];
            if (T > kMaxT) { ok = false; break; }
        }
        wv::sync();
        if (!ok) return r;
        int nlo = 0;
        int64_t lo = 1;
        while (nlo < nr && lo < kLoTarget && lo * t[T_RCARD][nlo] <= kFiberLoMax) lo *= t[T_RCARD][nlo++];
        bool rs_ok = true;
        const int rs = row_stride_rule(nlo, t[T_RCARD], [&](int i) { return t[T_RB1][i] != 0 || t[T_RTST][i] != 0; }, rs_ok);
        if (!rs_ok) return r;
        int64_t rcells = 1;
        for (int i = 0; i < nr; ++i) rcells *= t[T_RCARD][i];
        r.ok = true;
        r.nr = nr; r.nlo = nlo; r.lo = lo; r.row_stride = rs; r.nctrl = nctrl; r.T = T; r.rcells = rcells;
        return r;
    }
    WV_HD bool emit_outer(int n_in, WEnt &out, int cx, int c1, uint32_t *&wout) {
        if (!((cx == 4 && c1 == 4) || (cx == 16 && c1 == 4))) return false;
        int32_t (*t)[kWAxes] = W.e.t;
        int32_t (*s)[kWAxes] = W.e.s;
        uint32_t bigm = 0;
        for (int j = 0; j < n_in; ++j)
            if (fcells(cur_hl[j]) > N.small_cells) bigm |= 1u << j;
        const int nbig = __builtin_popcount(bigm), ns = n_in - nbig;
        if (nbig != 2 || ns > kMaxSmall) return false;
        const uint32_t smallm = ~bigm & ((1u << n_in) - 1);
        const int bigs[2] = {nth_bit(bigm, 0), nth_bit(bigm, 1)};
        // B = the big input that owns two 4-state output axes the other one does not depend on; of the two role assignments the
        // feasible one with the faster N axes wins (the second is tried only below the first one's key)
        const int64_t kNoKey = 0x7fffffffffffffffll;
        OuterTry r0 = outer_try(bigs[0], bigs[1], ns, smallm, out, cx, kNoKey);
        OuterTry r1 = outer_try(bigs[1], bigs[0], ns, smallm, out, cx, r0.ok ? r0.key : kNoKey);
        OuterTry r = r1;
        int A = bigs[0], B = bigs[1];
        if (!r1.ok) {
            if (!r0.ok) return false;
            r = outer_try(bigs[0], bigs[1], ns, smallm, out, cx, kNoKey);  // (the tables hold the second attempt's axes: again)
            A = bigs[1]; B = bigs[0];
        }
        const int nr = r.nr, nlo = r.nlo, nctrl = r.nctrl;
        if (r.rcells * 16 < N.big_iters) return false;
        wv::for_n(16, [&](int n) {
            W.e.nout[n] = (uint32_t)((n & 3) * dstride(r.nax0) + (n >> 2) * dstride(r.nax1));
            W.e.nB[n] = (uint32_t)((n & 3) * s[B][r.nax0] + (n >> 2) * s[B][r.nax1]);
        });
        wv::sync();
        bool contig = !wv::any_n(16, [&](int n) { return W.e.nout[n] != (uint32_t)n; });
        {
            int64_t expect = 16;
            for (int i = 0; i < nlo; ++i) { contig = contig && t[T_ROST][i] == expect; expect *= t[T_RCARD][i]; }
        }
        order_hi_axes(nr, nlo);
        int mlo = 0;
        const int ma = merge_r_axes(nr, nlo, 2, true, mlo);
        if (ma > kMaxAxes) return false;
        const int nT = 2 + nctrl;
        const int words = kHdrWords + 4 * 2 + ns * (4 + nT) + nT + 16 + 16 + 3 * ma + 2 * ma;
        if (words > kMaxStepWords) return false;
        uint32_t *w = extend(words);
        wout = w;
        header(kKindFiber, 2 + ns, ma, mlo, cx, false, r.lo, r.rcells / r.lo, (uint64_t)out.off, words);
        hdr_or(1, ((kFlagOuter | (contig ? kFlagContig : 0u)) << 16) | ((uint32_t)r.row_stride << kRowStrideShift));
        hdr_set(7, 2u | ((uint32_t)ns << 4) | (2u << 8) | ((uint32_t)nctrl << 12) | (16u << 16));
        hdr_set(8, (uint32_t)r.T | ((uint32_t)c1 << 16));
        uint32_t *p = w + kHdrWords;
        for (int q = 0; q < 2; ++q) {
            const int j = q ? B : A;
            put_off(p, foffset(cur_hl[j]));
            put(p, (uint32_t)W.e.xs[j][0]);
            put(p, (uint32_t)W.e.xs[j][1]);
        }
        for (uint32_t m = smallm; m; m &= m - 1) {
            const int j = __builtin_ctz(m);
            put_off(p, foffset(cur_hl[j]));
            put(p, (uint32_t)W.e.xs[j][0]);
            put(p, (uint32_t)W.e.xs[j][1]);
            put(p, (uint32_t)s[j][r.nax0]);
            put(p, (uint32_t)s[j][r.nax1]);
            wv::for_n(nctrl, [&](int i) { p[i] = (uint32_t)s[j][t[T_CTRL][i]]; });
            p += nctrl;
        }
        put(p, 4);
        put(p, 4);
        wv::for_n(nctrl, [&](int i) { p[i] = (uint32_t)card(out.vars[t[T_CTRL][i]]); });
        p += nctrl;
        wv::for_n(16, [&](int n) { p[n] = W.e.nout[n]; p[16 + n] = W.e.nB[n]; });
        p += 32;
        wv::for_n(ma, [&](int a) { p[3 * a] = (uint32_t)t[T_MC][a]; p[3 * a + 1] = (uint32_t)t[T_MO][a]; p[3 * a + 2] = (uint32_t)t[T_MT][a]; });
        p += 3 * ma;
        wv::for_n(2 * ma, [&](int i) { p[i] = (uint32_t)t[i < ma ? T_MB0 : T_MB1][i % ma]; });
        return true;
    }


    // ---- CHAIN form (emit_chain / emit_chain_as): three 4-state variables of one big input ------------------------------------
    WV_HD bool emit_chain(int n_in, WEnt &out, uint32_t *&wout) {
        const int kPerm[3][3] = {{0, 1, 2}, {0, 2, 1}, {1, 2, 0}};
        for (int pi = 0; pi < 3; ++pi)
            if (emit_chain_as(n_in, out, kPerm[pi][0], kPerm[pi][1], kPerm[pi][2], wout)) return true;
        return false;
    }
    // (p0, p1, p2): which of the step's eliminated variables plays x1, x2, x3
    WV_HD bool emit_chain_as(int n_in, WEnt &out, int p0, int p1, int p2, uint32_t *&wout) {
        int32_t (*t)[kWAxes] = W.e.t;
        int32_t (*s_in)[kWAxes] = W.e.s;
        auto xs = [&](int j, int k) { return W.e.xs[j][k == 0 ? p0 : (k == 1 ? p1 : p2)]; };
        int big = -1, n12 = 0, n3s = 0;
        uint32_t g12m = 0, g3m = 0;
        bool x3dep = false;
        for (int j = 0; j < n_in; ++j) {
            if (fcells(cur_hl[j]) > N.small_cells) {
                if (big >= 0) return false;
                big = j;
            } else if (xs(j, 2) != 0 && xs(j, 0) == 0 && xs(j, 1) == 0) {
                g3m |= 1u << j; ++n3s;
            } else {
                g12m |= 1u << j; ++n12;
                x3dep = x3dep || xs(j, 2) != 0;
            }
        }
        if (big < 0 || xs(big, 0) == 0 || xs(big, 1) == 0 || xs(big, 2) == 0) return false;
        if (n12 > kMaxSmall || n3s < 1 || n3s > 2) return false;
        const int na = out.n;
        if (na < 3) return false;
        auto dep_of = [&](uint32_t gm, int a) { bool d = false; for (uint32_t m = gm; m; m &= m - 1) d = d || s_in[__builtin_ctz(m)][a] != 0; return d; };
        // the three new axes: two of the pair tables, one of the third variable's
        int nax12[2] = {-1, -1}, nn12 = 0, nax3 = -1;
        bool n12dep = false;
        for (int a = 0; a < na; ++a) {
            if (s_in[big][a] != 0) continue;
            const bool dep12 = dep_of(g12m, a), dep3 = dep_of(g3m, a);
            if (card(out.vars[a]) != 4) return false;
            if (dep12) {
                if (nn12 >= 2) return false;
                if (nn12 == 0) nax12[0] = a; else nax12[1] = a;
                ++nn12;
                n12dep = n12dep || dep3;
            } else if (dep3) {
                if (nax3 >= 0) return false;
                nax3 = a;
            } else {
                return false;
            }
        }
        if (nn12 != 2 || nax3 < 0) return false;
        // the output's axis order: the three new axes fastest (n12 at strides 1 and 4, n3 at 16), the others in the order F stores them
        int32_t *ord = t[T_A];
        WV_LANE0 {
            ord[0] = nax12[0]; ord[1] = nax12[1]; ord[2] = nax3;
            int k = 3;
            for (int a = 0; a < na; ++a)
                if (a != nax3 && a != nax12[0] && a != nax12[1]) ord[k++] = a;
            for (int i = 4; i < na; ++i) {
                const int a = ord[i];
                int j = i - 1;
                while (j >= 3 && s_in[big][ord[j]] > s_in[big][a]) { ord[j + 1] = ord[j]; --j; }
                ord[j + 1] = a;
            }
        }
        wv::sync();
        // vars2 / ostr: T_B / T_C; the inputs' strides along the new order: sc(j, a) = s_in[j][ord[a]]
        int32_t *vars2 = t[T_B], *ostr = t[T_C];
        {
            int64_t cells = 1;
            for (int a = 0; a < na; ++a) {
                const int v = out.vars[ord[a]];
                WV_LANE0 { vars2[a] = v; ostr[a] = (int32_t)cells; }
                cells *= card(v);
            }
        }
        wv::sync();
        auto sc = [&](int j, int a) { return s_in[j][ord[a]]; };
        auto dep_sc = [&](uint32_t gm, int a) { bool d = false; for (uint32_t m = gm; m; m &= m - 1) d = d || sc(__builtin_ctz(m), a) != 0; return d; };
        // R axes; ctrl axes of T12 / of T3 = R axes a pair-group / third-group small input depends on.  T_RTST = rt12, T_RB1 = rt3, T_RB0 = F
        int nc12 = 0, nc3 = 0, nr = 0;
        int64_t T12 = 256, T3 = n12dep ? 256 : 16;
        int32_t *c12 = t[T_CTRL], *c3 = t[T_NAX];
        for (int a = 3; a < na; ++a) {
            const int64_t rc = card(vars2[a]);
            const bool dep12 = dep_sc(g12m, a), dep3 = dep_sc(g3m, a);
            WV_LANE0 { t[T_RCARD][nr] = (int32_t)rc; t[T_ROST][nr] = ostr[a]; t[T_RB0][nr] = sc(big, a); t[T_RTST][nr] = 0; t[T_RB1][nr] = 0; }
            if (dep12) {
                if (nc12 >= 10) return false;
                WV_LANE0 { c12[nc12] = a; t[T_RTST][nr] = (int32_t)T12; }
                ++nc12;
                T12 *= rc;
            }
            if (dep3) {
                if (nc3 >= 10) return false;
                WV_LANE0 { c3[nc3] = a; t[T_RB1][nr] = (int32_t)T3; }
                ++nc3;
                T3 *= rc;
            }
            if (T12 * (x3dep ? 4 : 1) + T3 > kMaxT) return false;
            ++nr;
        }
        wv::sync();
        const int64_t t12x3 = x3dep ? T12 : 0;
        if (x3dep) T12 *= 4;
        int nlo = 0;
        int64_t lo = 1;
        while (nlo < nr && lo < kLoTarget && lo * t[T_RCARD][nlo] <= kFiberLoMax) lo *= t[T_RCARD][nlo++];
        int64_t rcells = 1;
        for (int i = 0; i < nr; ++i) rcells *= t[T_RCARD][i];
        if (rcells * 64 < N.big_iters) return false;
        {   // the lane-varying block is contiguous in the output: cell l at 64*l
            int64_t expect = 64;
            for (int i = 0; i < nlo; ++i) { if (t[T_ROST][i] != expect) return false; expect *= t[T_RCARD][i]; }
        }
        bool rs_ok = true;
        const int row_stride = row_stride_rule(nlo, t[T_RCARD], [&](int i) { return t[T_RTST][i] != 0 || t[T_RB1][i] != 0; }, rs_ok);
        if (!rs_ok) return false;
        // merge (output, T12, T3, F): merge_r_axes with the four tables in its rows (ROST, RTST = rt12, RB0 = F, RB1 = rt3)
        int mlo = 0;
        const int ma = merge_r_axes(nr, nlo, 2, false, mlo);
        if (ma > kMaxAxes || ma < 1) return false;
        const int nT = 2 + nc12 + (x3dep ? 1 : 0);
        const int nd3 = 2 + (n12dep ? 2 : 0) + nc3;
        const int words = kHdrWords + 8 + n12 * (4 + nT) + nT + 16 + 2 + nd3 + n3s * (2 + nd3) + 3 * ma + 2 * ma;
        if (words > kMaxStepWords) return false;
        // commit the axis order of the output
        wv::for_n(na, [&](int a) { out.vars[a] = (uint8_t)vars2[a]; });  // (ostr[a] = 2^(l a): dense in the new order too)
        wv::sync();
        uint32_t *w = extend(words);
        wout = w;
        header(kKindFiber, n_in, ma, mlo, 16, false, lo, rcells / lo, (uint64_t)out.off, words);
        hdr_or(1, ((kFlagChain | kFlagContig) << 16) | ((uint32_t)row_stride << kRowStrideShift));
        hdr_set(7, 2u | ((uint32_t)n12 << 4) | (2u << 8) | ((uint32_t)(nT - 2) << 12) | (16u << 16));
        hdr_set(8, (uint32_t)T12 | (4u << 16));
        uint32_t *p = w + kHdrWords;
        put_off(p, foffset(cur_hl[big]));
        put(p, (uint32_t)xs(big, 0));
        put(p, (uint32_t)xs(big, 1));
        put(p, (uint32_t)T12);
        put(p, (uint32_t)T3);
        put(p, (uint32_t)xs(big, 2));
        put(p, (uint32_t)t12x3);
        for (uint32_t m = g12m; m; m &= m - 1) {
            const int j = __builtin_ctz(m);
            put_off(p, foffset(cur_hl[j]));
            put(p, (uint32_t)xs(j, 0));
            put(p, (uint32_t)xs(j, 1));
            put(p, (uint32_t)sc(j, 0));
            put(p, (uint32_t)sc(j, 1));
            wv::for_n(nc12, [&](int i) { p[i] = (uint32_t)sc(j, c12[i]); });
            p += nc12;
            if (x3dep) put(p, (uint32_t)xs(j, 2));
        }
        put(p, 4);
        put(p, 4);
        wv::for_n(nc12, [&](int i) { p[i] = (uint32_t)card(vars2[c12[i]]); });
        p += nc12;
        if (x3dep) put(p, 4);
        wv::for_n(16, [&](int n) { p[n] = (uint32_t)n; });
        p += 16;
        put(p, (uint32_t)n3s);
        put(p, (uint32_t)nd3 | (n12dep ? 256u : 0u));
        put(p, 4);  // x3
        put(p, 4);  // n3
        if (n12dep) { put(p, 4); put(p, 4); }
        wv::for_n(nc3, [&](int i) { p[i] = (uint32_t)card(vars2[c3[i]]); });
        p += nc3;
        for (uint32_t m = g3m; m; m &= m - 1) {
            const int j = __builtin_ctz(m);
            put_off(p, foffset(cur_hl[j]));
            put(p, (uint32_t)xs(j, 2));
            put(p, (uint32_t)sc(j, 2));
            if (n12dep) { put(p, (uint32_t)sc(j, 0)); put(p, (uint32_t)sc(j, 1)); }
            wv::for_n(nc3, [&](int i) { p[i] = (uint32_t)sc(j, c3[i]); });
            p += nc3;
        }
        wv::for_n(ma, [&](int a) { p[3 * a] = (uint32_t)t[T_MC][a]; p[3 * a + 1] = (uint32_t)t[T_MO][a]; p[3 * a + 2] = (uint32_t)t[T_MT][a]; });
        p += 3 * ma;
        wv::for_n(ma, [&](int a) { p[a] = (uint32_t)t[T_MB0][a]; p[ma + a] = (uint32_t)t[T_MB1][a]; });
        return true;
    }


    // ---- SWEEP form (emit_sweep): k = 2..5 four-state variables X[0..k) of the one big input, the tile resident in LDS ---------
    // hs[0..n_in) = the candidate's factors; X = W.order + i.  Does its own bookkeeping; false: nothing emitted, nothing allocated.
    WV_HD bool emit_sweep(const uint8_t *hs, int n_in, const uint8_t *X, int k, int eo) {
        WV_COUNT(14 + k);
        if (k < 2 || k > 5 || n_in - 1 > kSweepMaxSmall) return false;
        int Fh = -1;
        for (int j = 0; j < n_in; ++j)
            if (fcells(hs[j]) > N.small_cells) {
                if (Fh >= 0) return false;
                Fh = hs[j];
            }
        if (Fh < kWVars) return false;  // (none, or a CPT slice)
        const WEnt &F = W.e.ent[Fh - kWVars];
        const B2 Fscope = F.scope;
        const int rb = 13 - 2 * k;
        const int64_t Rt = int64_t(1) << rb;
        if ((int64_t)F.cells & ((int64_t(1) << (2 * k)) - 1)) return false;
        const int64_t Rcells = (int64_t)F.cells >> (2 * k);
        if (Rcells < Rt || (Rcells & (Rt - 1))) return false;
        // digits: x_j must sit on one of F's k slowest axes.  var_on[d] + 1 in byte d of `von`, dig[j] in nibble j of `digs`
        uint64_t von = 0;
        uint32_t digs = 0;
        auto var_on = [&](int d) { return (int)((von >> (8 * d)) & 0xff) - 1; };
        auto set_var_on = [&](int d, int v) { von = (von & ~(0xffull << (8 * d))) | ((uint64_t)(v + 1) << (8 * d)); };
        auto dig = [&](int j) { return (int)((digs >> (4 * j)) & 0xf); };
        for (int j = 0; j < k; ++j) {
            const int xj = X[j];
            if (card(xj) != 4) return false;
            int d = -1;
            for (int a = 0; a < F.n; ++a)
                if (F.vars[a] == xj) {
                    for (int q = 0; q < k; ++q)
                        if ((int64_t)dstride(a) == Rcells << (2 * q)) d = q;
                }
            if (d < 0 || var_on(d) >= 0) return false;
            digs |= (uint32_t)d << (4 * j);
            set_var_on(d, xj);
        }
        // stages: a small input belongs to the first eliminated variable it mentions
        uint32_t used = 0;
        B2 introduced;
        int t_total = 0, ns_total = 0;
        double in_cells = (double)F.cells;
        for (int j = 0; j < k; ++j) {
            WStage &g = W.e.stg[j];
            const int xj = X[j];
            int gns = 0;
            B2 U;
            for (int i = 0; i < n_in; ++i) {
                if (hs[i] == Fh || ((used >> i) & 1)) continue;
                const B2 sc = fscope(hs[i]);
                if (!sc.test(xj)) continue;
                used |= 1u << i;
                WV_LANE0 g.in[gns] = hs[i];
                ++gns;
                U = b2_or(U, sc);
                in_cells += (double)fcells(hs[i]);
            }
            ns_total += gns;
            const B2 fresh = b2_andn(b2_andn(U, Fscope), introduced);
            const int nnew = b2_count(fresh);
            if (nnew > 1) return false;
            int newv = -1;
            if (nnew == 1) {
                newv = b2_first(fresh);
                if (card(newv) != 4) return false;
                introduced.set(newv);
            }
            const int cout = nnew ? 4 : 1;
            int nctrl = 0, nfromr = 0;
            bool ok = true;
            int src3[3] = {0, 0, 0}, cvar3[3] = {0, 0, 0};
            b2_each(U, [&](int v) {
                if (!ok || v == xj || v == newv) return;
                int src = -1;
                for (int d = 0; d < k; ++d)
                    if (var_on(d) == v && d != dig(j)) src = d;
                if (src < 0) {
                    // an R axis of F: four states, power-of-two stride
                    for (int a = 0; a < F.n; ++a)
                        if (F.vars[a] == v && (int64_t)dstride(a) < Rcells) {
                            const int64_t st_ = dstride(a);
                            if (card(v) == 4 && (st_ & (st_ - 1)) == 0) src = 8 + __builtin_ctzll((unsigned long long)st_);
                        }
                }
                if (src < 0 || nctrl >= 3) { ok = false; return; }
                if (src >= 8) {  // at most two ctrl values come from r
                    if (nfromr >= 2) { ok = false; return; }
                    ++nfromr;
                }
                if (nctrl == 0) { src3[0] = src; cvar3[0] = v; } else if (nctrl == 1) { src3[1] = src; cvar3[1] = v; } else { src3[2] = src; cvar3[2] = v; }
                ++nctrl;
            });
            if (!ok) return false;
            const int t_cells = cout * 4 << (2 * nctrl);
            const int t_off = t_total;
            t_total += t_cells;
            if (t_total > kSweepMaxT) return false;
            const int loop = sweep_loop_digit(k, dig(j));
            int f3[3] = {7, 7, 7};
            for (int d = 0, m = 0; d < k; ++d)
                if (d != dig(j) && d != loop) { if (m == 0) f3[0] = d; else if (m == 1) f3[1] = d; else f3[2] = d; ++m; }
            WV_LANE0 {
                g.cout = cout; g.ns = gns; g.nctrl = nctrl; g.loop = loop; g.t_off = t_off; g.t_cells = t_cells; g.newv = newv;
                for (int c = 0; c < 3; ++c) { g.f[c] = f3[c]; g.src[c] = src3[c]; g.cvar[c] = cvar3[c]; }
            }
            set_var_on(dig(j), newv);  // (-1: the digit is dead from here on)
        }
        wv::sync();
        for (int i = 0; i < n_in; ++i)
            if (hs[i] != Fh && !((used >> i) & 1)) return false;  // (a small input that mentions none of the eliminated variables)
        // output: surviving digits (ascending) fastest, then F's R axes in F's order
        int kout = 0;
        uint32_t survs = 0;
        for (int d = 0; d < k; ++d)
            if (var_on(d) >= 0) { survs |= (uint32_t)d << (4 * kout); ++kout; }
        const int64_t out_cells = Rcells << (2 * kout);
        if (out_cells >= (1ll << 31)) return false;
        const int nR = wv::sum_n(F.n, [&](int a) { return (int64_t)dstride(a) < Rcells ? 1 : 0; });
        if (kout + nR > kWAxes) { err = kEmitErrDevice; return false; }
        WEnt &out = W.e.ent[eo];
        B2 oscope;
        for (int q = 0; q < kout; ++q) {
            const int v = var_on((int)((survs >> (4 * q)) & 0xf));
            WV_LANE0 out.vars[q] = (uint8_t)v;  // (stride 4^q = dstride(q): SWEEP variables have four states, l = 2)
            oscope.set(v);
        }
        // (F's R axes are its first ones - the eliminated variables are its slowest - so R axis a lands on position kout + a: its stride
        //  F's 2^(l a) times 4^kout = dstride(kout + a))
        wv::compact_n(F.n, kout, [&](int a) { return (int64_t)dstride(a) < Rcells; }, [&](int a, int pos) { out.vars[pos] = F.vars[a]; });
        for (int a = 0; a < F.n; ++a)
            if ((int64_t)dstride(a) < Rcells) oscope.set(F.vars[a]);
        const int na = kout + nR;
        const int words = kHdrWords + 2 + k * kSweepStageWords + ns_total * kSweepSmallWords;
        if (words > kMaxStepWords) return false;
        const int64_t ooff = arena_alloc(out_cells);
        WV_LANE0 { out.n = na; out.cells = (int32_t)out_cells; out.off = ooff; out.scope = oscope; }
        wv::sync();
        uint32_t *w = extend(words);
        header(kKindSweep, n_in, k, rb, 1 << (2 * k), false, kSweepTileCells, Rcells / Rt, (uint64_t)ooff, words);
        hdr_set(7, (uint32_t)kout | ((uint32_t)t_total << 16));
        hdr_set(8, survs);
        {
            bool canon = N.sweep_canon != 0;
            for (int j = 0; j < k; ++j) canon = canon && dig(j) == k - 1 - j;
            if (canon) hdr_or(1, kFlagSweepCanon << 16);
        }
        uint32_t *p = w + kHdrWords;
        put_off(p, (uint64_t)F.off);
        wv::for_n(k, [&](int j) {
            const WStage &g = W.e.stg[j];
            uint32_t *q = p + kSweepStageWords * j;
            q[0] = (uint32_t)dig(j) | ((uint32_t)g.cout << 4) | ((uint32_t)g.ns << 8) | ((uint32_t)g.nctrl << 12) | ((uint32_t)g.loop << 16) |
                   ((uint32_t)g.f[0] << 20) | ((uint32_t)g.f[1] << 24) | ((uint32_t)g.f[2] << 28);
            q[1] = (uint32_t)g.t_off | ((uint32_t)g.t_cells << 16);
            for (int c = 0; c < 3; ++c) q[2 + c] = c < g.nctrl ? ((uint32_t)g.src[c] | ((uint32_t)(g.cout * 4 << (2 * c)) << 8)) : 0u;
        });
        p += kSweepStageWords * k;
        for (int j = 0; j < k; ++j) {
            const WStage &g = W.e.stg[j];
            for (int i = 0; i < g.ns; ++i) {
                const int h = g.in[i];
                put_off(p, foffset(h));
                put(p, (uint32_t)(int32_t)(g.newv >= 0 ? fstride_of(h, g.newv) : 0));
                put(p, (uint32_t)(int32_t)fstride_of(h, X[j]));
                for (int c = 0; c < 3; ++c) put(p, (uint32_t)(int32_t)(c < g.nctrl ? fstride_of(h, g.cvar[c]) : 0));
            }
        }
        hdr_set(9, (uint32_t)(((int64_t)in_cells + out_cells + 2) >> 2));
        flush_header(w);
        book(8.0 * (in_cells + (double)out_cells), (double)k * 4.0 * (double)F.cells, (double)F.cells);
        tag_step((uint32_t)(w - data));
        WV_COUNT(20 + k);
        for (int j = 0; j < n_in; ++j)
            if (falloc(hs[j])) arena_release((int64_t)foffset(hs[j]), falloc(hs[j]));
        return true;
    }

    // ---- one step (Emitter::emit): multiply hl[0..n_in), sum out X[0..nx); the new factor is entry eo --------------------------
    WV_HD bool emit(const uint8_t *hl, int n_in, const int *X, int nx, bool final_, int64_t final_off, int eo, bool fiber_only) {
        WV_COUNT(nx * 2 + (fiber_only ? 1 : 0));
        WV_ETICK(13)  // (the caller's part up to here)
        WEnt &out = W.e.ent[eo];
        cur_hl = hl;
        const int l = N.uniform_log2;
        B2 scope;
        double in_cells = 0;
        int maxn = 1;
        int64_t in_sum = 0;  // (cell counts: integers - summed exactly, converted once)
        for (int j = 0; j < n_in; ++j) {
            const int h = hl[j];
            scope = b2_or(scope, fscope(h));
            in_sum += fcells(h);
            const int rn = h < kWVars ? (int)N.scope_off[h + 1] - (int)N.scope_off[h] : (int)W.e.ent[h - kWVars].n;
            maxn = rn > maxn ? rn : maxn;
        }
        in_cells = (double)in_sum;
        for (int k = 0; k < nx; ++k) scope.clr(X[k]);
        const int na = b2_count(scope);
        if (na > kWAxes) { err = kEmitErrDevice; return false; }
        // every axis is a multi-state variable of 2^l states: the dense strides are powers of two
        if (l * na >= 31) { if (!fiber_only) err = kEmitErrCells; return false; }
        const int64_t cells = int64_t(1) << (l * na);
        WV_ETICK(14)  // scope
        // layout: longest-living variable fastest.  In layout-position space (W.key / W.pvar: elimination position, the query variables
        // behind) the scope is a bit set, the rank of a variable the number of scope bits above its own
        const B2 ps = wv::mask128(npos, [&](int p) { return scope.test(W.pvar[p]); });
        wv::for_n(npos, [&](int p) {
            if (!ps.test(p)) return;
            const int r = b2_count(b2_and(ps, b2_above(p)));
            const int v = W.pvar[p];
            out.vars[r] = (uint8_t)v;
            W.pos[v] = (int8_t)r;
        });
        WV_LANE0 { out.n = na; out.cells = (int32_t)cells; out.scope = scope; }
        WV_ETICK(15)  // layout
        // per-input strides along the output axes, and along the eliminated variables
        int32_t (*s)[kWAxes] = W.e.s;
        wv::for_n(n_in * 32, [&](int i) { const int j = i >> 5, a = i & 31; if (a < na) s[j][a] = 0; else if (a < na + 3) W.e.xs[j][a - na] = 0; });
        wv::sync();
        int sh = 0;
        while ((1 << sh) < maxn) ++sh;
        wv::for_n(n_in << sh, [&](int i) {
            const int j = i >> sh, k = i & ((1 << sh) - 1);
            const int h = hl[j];
            int v, st;
            if (h < kWVars) {
                const int q = (int)N.scope_off[h] + k;
                if (q >= (int)N.scope_off[h + 1]) return;
                v = N.scope_var[q];
                if (!keep.test(v)) return;
                st = N.scope_stride[q];
            } else {
                const WEnt &E = W.e.ent[h - kWVars];
                if (k >= E.n) return;
                v = E.vars[k];
                st = dstride(k);
            }
            if (nx > 0 && v == X[0]) W.e.xs[j][0] = st;
            else if (nx > 1 && v == X[1]) W.e.xs[j][1] = st;
            else if (nx > 2 && v == X[2]) W.e.xs[j][2] = st;
            else s[j][W.pos[v]] = st;
        });
        wv::sync();
        WV_ETICK(16)  // strides
        const int c1 = nx > 0 ? card(X[0]) : 1;
        const int cx = nx > 1 ? c1 * card(X[1]) : c1;
        int64_t ooff, oalloc;
        if (final_) { ooff = final_off; oalloc = 0; }
        else { ooff = arena_alloc(cells); oalloc = cells; }
        WV_LANE0 out.off = ooff;
        wv::sync();
        WV_ETICK(17)  // arena_alloc
        uint32_t *w = nullptr;
        const bool streaming = !final_ && cells >= N.big_iters;
        const bool fiber = !streaming ? false
                           : nx == 3  ? emit_chain(n_in, out, w)
                                      : ((N.outer && nx > 0 && emit_outer(n_in, out, cx, c1, w)) || emit_fiber(n_in, out, cx, c1, w));
        if (err) return false;
        WV_ETICK(18)  // streaming forms
        if (!fiber) {
            if (fiber_only) {
                if (oalloc) arena_release(ooff, oalloc);
                WV_COUNT(8 + nx);
                return false;
            }
            WV_COUNT(streaming ? 12 : 13);
            emit_generic(n_in, out, cells, cx, final_, w);
        }
        if (err) return false;
        WV_ETICK(19)  // GENERIC
        hdr_set(9, (uint32_t)(((int64_t)in_cells + cells + 2) >> 2));
        flush_header(w);
        double pc = (double)cells;  // cells of the product scope = the output's cells x the eliminated cardinalities
        for (int k = 0; k < nx; ++k) pc *= card(X[k]);
        book(8.0 * (in_cells + (double)cells), n_in * pc, pc);
        WV_ETICK(20)  // header, statistics
        tag_step((uint32_t)(w - data));
        WV_ETICK(21)  // work item
        for (int j = 0; j < n_in; ++j)
            if (falloc(hl[j])) arena_release((int64_t)foffset(hl[j]), falloc(hl[j]));
        WV_ETICK(22)  // arena_release
        return true;
    }

    // ---- work items (tag_program, one step at a time: the header of the step just emitted is in W.e.hdr) ----------------------
    WV_HD void put_tag(const Tag &tg) {
        if (n_tags >= kWTags) { err = kEmitErrDevice; return; }
        WV_LANE0 W.e.tags[n_tags] = tg;
        ++n_tags;
    }
    WV_HD void flush_segment() {
        const uint32_t seg_steps = W.e.c.seg_steps;
        if (seg_steps) {
            put_tag(Tag{W.e.c.seg_first, seg_steps | kItemSegment, 1u, (uint16_t)level, (uint16_t)kKidSeg, (float)W.e.c.seg_bytes});
            ++level;
            wv::sync();
            WV_LANE0 { W.e.c.seg_steps = 0; W.e.c.seg_bytes = 0; }
            wv::sync();
        }
    }
    // (step_is_tiled / step_tile_h of emit_core.h on the options in WNet)
    WV_HD bool hdr_is_tiled(const uint32_t *w) const {
        if ((w[0] & 0xff) == kKindFiber || (w[0] & 0xff) == kKindSweep) return true;
        const bool fin = (w[1] >> 16) & kFlagFinal;
        return !fin && (int64_t)w[2] * (int64_t)w[3] >= N.big_iters;
    }
    WV_HD int hdr_tile_h(const uint32_t *w) const {
        if ((w[0] & 0xff) == kKindSweep) return emit_max(1, emit_min((int)N.sweep_iters, kTileMax));
        if (N.tile_h > 0) return emit_min((int)N.tile_h, kTileMax);
        const int64_t per_iter = emit_max<int64_t>(1, step_cost_bytes(w) / emit_max<int64_t>(1, (int64_t)w[3]));
        return (int)emit_max<int64_t>(1, emit_min<int64_t>(kTileMax, N.tile_bytes / per_iter));
    }
    WV_HD void tag_step(uint32_t off) {
        if (overflow) return;
        wv::sync();
        const uint32_t *w = W.e.hdr;
        const double bytes = (double)step_cost_bytes(w);
        if (hdr_is_tiled(w)) {
            flush_segment();
            const uint32_t th = (uint32_t)hdr_tile_h(w);
            put_tag(Tag{off, (w[0] & 0xff) == kKindSweep ? w[3] : th, (w[3] + th - 1) / th, (uint16_t)level, (uint16_t)kernel_id_of_step(w), (float)bytes});
            ++level;
        } else {
            WV_LANE0 {
                if (!W.e.c.seg_steps) W.e.c.seg_first = off;
                ++W.e.c.seg_steps;
                W.e.c.seg_bytes += bytes;
            }
        }
    }

    // ---- the request (emit_begin) ---------------------------------------------------------------------------------------------
    WV_HD int begin(int ne, const int32_t *evars, const int32_t *ecodes, const B2 &rel, const B2 &keep_, const B2 &eb) {
        keep = wv::uni(keep_);
        wv::for_n(ne, [&](int i) { W.ecode[evars[i]] = (uint16_t)(ecodes ? ecodes[i] : 0); });
        wv::sync();
        const bool bad = wv::any_n(N.n_vars, [&](int v) {
            if (!rel.test(v)) return false;
            uint32_t off = N.pool_off[v];
            int64_t cells = 1;
            int cnt = 0;
            for (int k = N.scope_off[v]; k < N.scope_off[v + 1]; ++k) {
                const int u = N.scope_var[k];
                if (eb.test(u)) off += (uint32_t)(N.scope_stride[k] * (int32_t)W.ecode[u]);
                else if (card(u) > 1) { cells *= card(u); ++cnt; }
            }
            W.cpt_off[v] = off;
            W.cpt_cells[v] = (uint32_t)cells;
            return cnt > kWAxes || cells >= (1ll << 31);
        });
        wv::sync();
        if (bad) return kEmitErrDevice;
        alive0 = wv::uni(rel);
        n_pool = b2_count(rel);
        pool_cap = emit_pool_cap(N.n_vars);
        return 0;
    }

    // factors alive whose scope contains a (or b, if b >= 0), in slot order (CPT slices by variable, then creation order);
    // f(handle) returns false to stop early
    template <class F> WV_HD void each_with(int a, int b, F f) {
        B2 m0 = N.fam[a];
        if (b >= 0) m0 = b2_or(m0, N.fam[b]);
        m0 = b2_and(m0, alive0);
        // (a CPT mentions a variable that is evidence in it?  fam is over the raw scopes: a and b are hidden variables, never evidence)
        const uint64_t mc = wv::mask64(n_live, [&](int p) {
            const B2 sc = W.e.ent[W.e.live[p]].scope;
            return sc.test(a) || (b >= 0 && sc.test(b));
        });
        bool go = true;
        for (uint64_t m = m0.a; m && go; m &= m - 1) go = f(__builtin_ctzll(m));
        for (uint64_t m = m0.b; m && go; m &= m - 1) go = f(64 + __builtin_ctzll(m));
        for (uint64_t m = mc; m && go; m &= m - 1) go = f(kWVars + (int)W.e.live[__builtin_ctzll(m)]);
    }
    // the factors list[a..b) leave the alive sets: CPT slices by bit, created ones by ONE order-preserving compaction of the live list
    // (their entries stay busy - consumed_ents - until the step in flight is complete)
    WV_HD void consume_range(const uint8_t *list, int a, int b) {
        uint32_t cons = 0;
        for (int j = a; j < b; ++j) {
            const int h = list[j];
            if (h < kWVars) { B2 a = alive0; a.clr(h); alive0 = wv::uni(a); }
            else cons |= 1u << (h - kWVars);
        }
        if (cons) remove_live(wv::mask64(n_live, [&](int p) { return ((cons >> W.e.live[p]) & 1) != 0; }));
    }
    WV_HD void remove_live(uint64_t rm) {  // rm: positions of the live list
        if (!rm) return;
        uint32_t cons = 0;
        for (uint64_t m = rm; m; m &= m - 1) cons |= 1u << W.e.live[__builtin_ctzll(m)];
        consumed_ents |= cons;
        const int n = n_live;
        uint8_t *live = W.e.live, *tmp = W.e.live_tmp;
        wv::for_n(n, [&](int p) { tmp[p] = live[p]; });
        wv::sync();
        wv::for_n(n, [&](int p) { if (!((rm >> p) & 1)) live[p - __builtin_popcountll(rm & ((1ull << p) - 1))] = tmp[p]; });
        n_live = n - __builtin_popcountll(rm);
        wv::sync();
    }
    WV_HD void add_factor(int e) {
        WV_LANE0 W.e.live[n_live] = (uint8_t)e;
        ++n_live;
        wv::sync();
    }
    WV_HD void ent_free(int e) { ent_busy &= ~(1u << e); consumed_ents &= ~(1u << e); }

    // multiply / eliminate with at most kMaxIn inputs per step: larger products are pre-multiplied (smallest tables first)
    WV_HD int emit_limited(int n_in, int x, bool final_, int64_t final_off) {
        uint8_t *hl = W.e.hl;
        while (n_in > kMaxIn && !err) {
            if (n_pool + 2 > pool_cap) { err = kEmitErrPool; return -1; }
            WV_LANE0 for (int i = 1; i < n_in; ++i) {  // (stable insertion sort by cells)
                const uint8_t h = hl[i];
                const int64_t c = fcells(h);
                int k = i - 1;
                while (k >= 0 && fcells(hl[k]) > c) { hl[k + 1] = hl[k]; --k; }
                hl[k + 1] = h;
            }
            wv::sync();
            const int eo = ent_alloc();
            if (eo < 0) return -1;
            ++n_pool;
            emit(hl, kMaxIn, nullptr, 0, false, 0, eo, false);
            if (err) return -1;
            for (int k = 0; k < kMaxIn; ++k)
                if (hl[k] >= kWVars) ent_free(hl[k] - kWVars);  // (consumed and multiplied: dead)
            wv::sync();
            WV_LANE0 {
                for (int k = kMaxIn; k < n_in; ++k) hl[k - kMaxIn] = hl[k];
                hl[n_in - kMaxIn] = (uint8_t)(kWVars + eo);
            }
            n_in -= kMaxIn - 1;
            wv::sync();
        }
        if (err) return -1;
        if (n_pool + 1 > pool_cap) { err = kEmitErrPool; return -1; }
        const int eo = ent_alloc();
        if (eo < 0) return -1;
        ++n_pool;
        int xx = x;
        emit(hl, n_in, &xx, x >= 0 ? 1 : 0, final_, final_off, eo, false);
        for (int k = 0; k < n_in; ++k)
            if (hl[k] >= kWVars) ent_free(hl[k] - kWVars);
        return eo;
    }

    // ---- the elimination loop and the final product (emit_run); the order is W.order[0..n_best) ---------------------------------
    WV_HD int run(int nq, const int32_t *qvars, int64_t out_off, int n_best WV_PROF_ARG) {
#if defined(MIBN_WAVE_PROF)
        pp_ = &prof_;
#endif
        if (nq > 127) return kEmitErrDevice;
        wv::for_n(n_best, [&](int i) { W.pvar[i] = W.order[i]; });
        wv::for_n(nq, [&](int i) { W.pvar[n_best + i] = (uint8_t)qvars[i]; });
        npos = n_best + nq;
        if (npos > kWVars) return kEmitErrDevice;
        wv::sync();
        uint32_t *count_word = extend(1);
        WV_LANE0 count_word[0] = 0;
        uint8_t *hl = W.e.hl, *hs = W.e.hs;
        const uint8_t *best = W.order;
        int32_t *sweep_n = W.e.t[T_F];
        const double lcard = (double)N.uniform_log2;  // log2card of a multi-state variable
        for (int i = 0; i < n_best; ++i) {
            if (overflow) return kEmitErrWords;
            if (err) return err;
            ent_busy &= ~consumed_ents;  // the previous step's inputs are gone
            consumed_ents = 0;
            const int x = best[i];
            // pop every factor mentioning x (bayes_net.py:780-784): the CPT slices by variable, then the created ones in creation order
            int n_in = 0;
            {
                const B2 m0 = b2_and(N.fam[x], alive0);
                alive0 = wv::uni(b2_andn(alive0, m0));
                const uint64_t mc = wv::mask64(n_live, [&](int p) { return ((B2)W.e.ent[W.e.live[p]].scope).test(x); });
                const int n0 = b2_count(m0);
                n_in = n0 + __builtin_popcountll(mc);
                if (n_in > kWIns) return kEmitErrDevice;
                WV_LANE0 { int k = 0; b2_each(m0, [&](int v) { hl[k++] = (uint8_t)v; }); }
                wv::for_n(n_live, [&](int p) { if ((mc >> p) & 1) hl[n0 + __builtin_popcountll(mc & ((1ull << p) - 1))] = (uint8_t)(kWVars + W.e.live[p]); });
                wv::sync();
                remove_live(mc);
            }
            WV_TICK(5)  // factors of x
            // the one big input of the step, if there is exactly one
            int nbig = 0, bigh = -1;
            for (int j = 0; j < n_in; ++j)
                if (fcells(hl[j]) > N.small_cells) { ++nbig; bigh = hl[j]; }
            // SWEEP candidates: how many of the next variables could join - all on the one big input, every other factor that
            // mentions them small
            int sweep_max = 0;
            if (N.fuse && N.sweep >= 3 && i + emit_min(2, N.sweep_min - 1) < n_best && card(x) == 4 && n_in - 1 <= kSweepMaxSmall &&
                n_pool + 1 <= pool_cap) {
                if (nbig == 1 && bigh >= kWVars && fcells(bigh) >= 16 * (int64_t)N.big_iters && fcells(bigh) >= 2 * kSweepTileCells) {
                    const B2 bigscope = fscope(bigh);
                    B2 taken0;
                    uint32_t takenc = 0;
                    int n_all = n_in;
                    wv::for_n(n_in, [&](int j) { hs[j] = hl[j]; });
                    WV_LANE0 sweep_n[0] = n_in;
                    wv::sync();
                    sweep_max = 1;
                    for (int j = 1; j < emit_min(N.sweep, 5) && i + j < n_best; ++j) {
                        const int xj = best[i + j];
                        if (card(xj) != 4 || !bigscope.test(xj)) break;
                        bool ok = true;
                        each_with(xj, -1, [&](int h) {
                            if (h < kWVars ? taken0.test(h) : ((takenc >> (h - kWVars)) & 1) != 0) return true;
                            if (fcells(h) > N.small_cells || n_all - 1 >= kSweepMaxSmall) { ok = false; return false; }
                            if (h < kWVars) taken0.set(h); else takenc |= 1u << (h - kWVars);
                            WV_LANE0 hs[n_all] = (uint8_t)h;
                            ++n_all;
                            return true;
                        });
                        if (!ok) break;
                        WV_LANE0 sweep_n[j] = n_all;
                        sweep_max = j + 1;
                    }
                    wv::sync();
                }
            }
            auto try_sweep = [&](int k_hi, int k_lo) -> bool {
                for (int k = emit_min(k_hi, sweep_max); k >= k_lo; --k) {
                    const int eo = ent_alloc();
                    if (eo < 0) return false;
                    ++n_pool;
                    if (emit_sweep(hs, sweep_n[k - 1], best + i, k, eo)) {
                        consume_range(hs, n_in, sweep_n[k - 1]);
                        add_factor(eo);
                        i += k - 1;
                        return true;
                    }
                    --n_pool;
                    ent_free(eo);
                    if (err) return false;
                }
                return false;
            };
            WV_TICK(6)  // sweep candidates
            if (sweep_max >= 4 && try_sweep(5, 4)) { WV_TICK(7) continue; }
            WV_TICK(7)  // SWEEP 5 / 4
            if (err) return err;
            // CHAIN: three consecutive 4-state variables of one big table in a single pass
            if (N.fuse && N.chain && i + 2 < n_best && n_in < kMaxIn && n_pool + 1 <= pool_cap && card(x) == 4 && card(best[i + 1]) == 4 &&
                card(best[i + 2]) == 4) {
                const int x2 = best[i + 1], x3 = best[i + 2];
                if (nbig == 1) {
                    const B2 bigscope = fscope(bigh);
                    if (bigscope.test(x2) && bigscope.test(x3) && fcells(bigh) >= 16 * (int64_t)N.big_iters) {
                        int n3 = n_in;
                        bool fits = true;
                        each_with(x2, x3, [&](int h) {
                            if (n3 >= kMaxIn || fcells(h) > N.small_cells) { fits = false; return false; }
                            WV_LANE0 hl[n3] = (uint8_t)h;
                            ++n3;
                            return true;
                        });
                        wv::sync();
                        if (fits) {  // exactly three new variables (the frontier keeps its width)
                            B2 u;
                            for (int j = 0; j < n3; ++j) u = b2_or(u, fscope(hl[j]));
                            fits = b2_count(u) - b2_count(bigscope) == 3;
                        }
                        if (fits) {
                            const int X[3] = {x, x2, x3};
                            const int eo = ent_alloc();
                            if (eo < 0) return err;
                            ++n_pool;
                            if (emit(hl, n3, X, 3, false, 0, eo, true)) {
                                consume_range(hl, n_in, n3);
                                add_factor(eo);
                                i += 2;
                                WV_TICK(8)
                                continue;
                            }
                            --n_pool;
                            ent_free(eo);
                            if (err) return err;
                        }
                    }
                }
            }
            WV_TICK(8)  // CHAIN
            if (sweep_max >= 3 && try_sweep(3, 3)) { WV_TICK(9) continue; }
            if (err) return err;
            if (N.sweep_min <= 2 && sweep_max >= 2 && try_sweep(2, 2)) { WV_TICK(9) continue; }  // (a pair of one big table: before the FIBER pair form)
            WV_TICK(9)  // SWEEP 3 / 2
            if (err) return err;
            // joint elimination of two variables in one FIBER pass
            if (N.fuse && i + 1 < n_best && n_in < kMaxIn && n_pool + 1 <= pool_cap) {
                const int x2 = best[i + 1];
                bool link = false;
                B2 u;
                for (int j = 0; j < n_in; ++j) { const B2 sc = fscope(hl[j]); link = link || sc.test(x2); u = b2_or(u, sc); }
                if (link && card(x) * card(x2) <= kMaxCx && (double)(N.uniform_log2 * b2_count(u)) - lcard > N.log2_small) {
                    int n2 = n_in;
                    bool fits = true;
                    each_with(x2, -1, [&](int h) {
                        if (n2 >= kMaxIn) { fits = false; return false; }
                        WV_LANE0 hl[n2] = (uint8_t)h;
                        ++n2;
                        u = b2_or(u, fscope(h));
                        return true;
                    });
                    wv::sync();
                    if (fits) {
                        int nb2 = 0;
                        for (int j = 0; j < n2; ++j) nb2 += fcells(hl[j]) > N.small_cells;
                        const double out_log2 = (double)(N.uniform_log2 * b2_count(u)) - lcard - lcard;
                        if (nb2 < 1 || nb2 > 2 || out_log2 < N.log2_big) fits = false;
                    }
                    if (fits) {
                        const int X[2] = {x, x2};
                        const int eo = ent_alloc();
                        if (eo < 0) return err;
                        ++n_pool;
                        if (emit(hl, n2, X, 2, false, 0, eo, true)) {
                            consume_range(hl, n_in, n2);
                            add_factor(eo);
                            ++i;
                            WV_TICK(10)
                            continue;
                        }
                        --n_pool;
                        ent_free(eo);
                        if (err) return err;
                    }
                }
            }
            WV_TICK(10)  // pair
            const int out = emit_limited(n_in, x, false, 0);  // pointwise_mul + sum_out (785)
            if (err) return err;
            add_factor(out);
            WV_TICK(11)  // single elimination
        }
        if (err) return err;
        ent_busy &= ~consumed_ents;
        consumed_ents = 0;
        // posterior = pointwise_mul(factors) / sum (bayes_net.py:789-790), written in the caller's query order
        int n_in = 0;
        b2_each(alive0, [&](int v) { if (n_in < kWIns) { WV_LANE0 hl[n_in] = (uint8_t)v; } ++n_in; });
        for (int p = 0; p < n_live; ++p) { if (n_in < kWIns) { WV_LANE0 hl[n_in] = (uint8_t)(kWVars + W.e.live[p]); } ++n_in; }
        if (n_in > kWIns) return kEmitErrDevice;
        wv::sync();
        emit_limited(n_in, -1, true, out_off);
        if (err) return err;
        if (overflow) return kEmitErrWords;
        wv::sync();
        WV_LANE0 count_word[0] = (uint32_t)W.e.c.n_steps;
        flush_segment();
        WV_TICK(12)  // final product
        if (err) return err;
        return 0;
    }
};

// One request, start to finish: order search, then emission into `slot` (cap words).  `anc` = the network's ancestor sets (global
// memory: read once per query / evidence variable).  Fills R; the work items are in W.e.tags[0..R.n_tags).
// effort 1: words parked behind the usable part of a request's slot while two orders are emitted - the runner-up's order (a byte per
// variable) and the first program's work items
constexpr uint32_t kWStashOrderWords = kWVars / 4, kWStashWords = kWStashOrderWords + kWTags * (uint32_t)(sizeof(Tag) / 4);

WV_HD void wave_plan_request(const WNet &N, WState &W, const B2 *anc, int nq, const int32_t *qvars, int ne, const int32_t *evars,
                             const int32_t *ecodes, bool no_prune, int64_t out_off, uint32_t *slot, uint32_t cap, WResult &R WV_PROF_ARG) {
    WOrderCtx oc(N, W);
    const int n_best = oc.search(nq, qvars, ne, evars, anc, no_prune WV_PROF_PASS);
    R.words = 1; R.n_tags = 0; R.err = 0; R.base = 0;
    R.alg_bytes = R.alg_flops = R.n_steps = R.max_step_cells = 0;
    R.arena_cells = 0;
    if (n_best < 0) { R.err = kEmitErrDevice; return; }
    B2 eb;
    for (int i = 0; i < ne; ++i) eb.set(evars[i]);
    // effort 1 (plan_request_rec): where the best order is expensive the runner-up is emitted too, behind the first program in the same slot,
    // and the program that moves fewer bytes stays.  Its order waits behind the usable part of the slot (the search's memory is the emission's).
    const bool try_second = N.effort >= 1 && oc.second_ >= 0 && oc.best_cost_ >= N.second_above;
    int n_second = 0;
    uint32_t *stash = nullptr;
    if (try_second) {
        if (cap < kWStashWords + 4 * (uint32_t)kMaxStepWords) { R.err = kEmitErrWords; return; }
        cap -= kWStashWords;
        stash = slot + cap;
        n_second = W.o.n_cand[oc.second_];
        uint8_t *so = reinterpret_cast<uint8_t *>(stash);
        const uint8_t *src = W.o.cand[oc.second_];
        wv::for_n(n_second, [&](int i) { so[i] = src[i]; });
        wv::sync();
    }
    // (hidden as emit_begin has it: the multi-state variables only - single-state ones are never axes, never eliminated)
    constexpr uint32_t kTagWords = sizeof(Tag) / 4;
    uint32_t *tag_stash = stash ? stash + kWStashOrderWords : nullptr;
    uint32_t base = 0;  // words of the first program (the second one is emitted behind it)
#if defined(__HIP_DEVICE_COMPILE__)
#pragma nounroll  // (ONE copy of the emitter in the kernel)
#endif
    for (int pass = 0; pass < 2; ++pass) {
        if (pass == 1) {
            // the first program's work items leave the emission's memory, the runner-up's order comes back
            const uint32_t *src = reinterpret_cast<const uint32_t *>(W.e.tags);
            wv::for_n((int)(R.n_tags * kTagWords), [&](int i) { tag_stash[i] = src[i]; });
            wv::gsync();
            const uint8_t *so = reinterpret_cast<const uint8_t *>(stash);
            wv::for_n(n_second, [&](int i) { W.order[i] = so[i]; });
            wv::sync();
        }
        WEmit em(N, W, slot + base, cap - base);
        int err = em.begin(ne, evars, ecodes, oc.rel, oc.keep, eb);
        WV_TICK(4)  // CPT slices
        if (!err) err = em.run(nq, qvars, out_off, pass ? n_second : n_best WV_PROF_PASS);
        wv::sync();
        if (pass == 0) {
            R.err = err;
            R.words = em.size;
            R.n_tags = (uint32_t)em.n_tags;
            R.alg_bytes = W.e.c.alg_bytes; R.alg_flops = W.e.c.alg_flops; R.n_steps = W.e.c.n_steps; R.max_step_cells = W.e.c.max_step_cells;
            R.arena_cells = em.top;
            if (err || !try_second) return;
            base = R.words;
            // (the emitter parks an overflowing step kMaxStepWords below the end of its buffer: the second program needs at least that much)
            if (cap - base < 2 * (uint32_t)kMaxStepWords) { R.err = kEmitErrWords; return; }
            continue;
        }
        // what only the device's limits refuse is not "the second order cannot be emitted": the request (its chunk, for the slot size) goes to the host
        if (err == kEmitErrDevice || err == kEmitErrWords) { R.err = err; return; }
        if (!err && W.e.c.alg_bytes < R.alg_bytes) {
            R.base = base;  // (it stays where it is: the work items' offsets are relative to the program's first word)
            R.words = em.size;
            R.n_tags = (uint32_t)em.n_tags;
            R.alg_bytes = W.e.c.alg_bytes; R.alg_flops = W.e.c.alg_flops; R.n_steps = W.e.c.n_steps; R.max_step_cells = W.e.c.max_step_cells;
            R.arena_cells = em.top;
        } else {
            uint32_t *dst = reinterpret_cast<uint32_t *>(W.e.tags);
            wv::for_n((int)(R.n_tags * kTagWords), [&](int i) { dst[i] = tag_stash[i]; });
            wv::sync();
        }
    }
}

}  // namespace mibn
