// Elimination-order search for networks of up to 128 variables - ONE implementation, compiled for the host (planner.cpp)
// and for the device (order_kernel in engine.hip: one request per lane), so that both choose the same order, bit for bit.
//
// The reference eliminates in Python-set iteration order (sorobn/bayes_net.py:766, 779), which is arbitrary and
// catastrophic on grids.  Here the hidden set of a request (relevant = query | evidence | ancestors, 763-765, minus query
// and evidence, 766) is ordered by the cheapest of
//   * "meet": sweep down from the roots to the query's depth, then up from the leaves,
//   * the reverse topological sweep,
//   * every order hint of the caller (row-major on the grid),
//   * greedy min-fill on the interaction graph, searched only when the sweeps cost more than `minfill_above` bytes,
// under the SURVEY section 8(d) byte model (per elimination 8 x (input cells + output cells), evidence axes collapsed).
//
// Everything is plain arrays and two-word bit sets: no heap, no std containers, no recursion - the whole search state is
// `OrderScratch` (host: one per planning thread; device: the lane's private memory).
#pragma once
#include <cmath>
#include <cstdint>

#if defined(__HIPCC__)
#define MIBN_HD __host__ __device__
#else
#define MIBN_HD
#endif

namespace mibn {

struct B2 {
    uint64_t a = 0, b = 0;
    MIBN_HD bool test(int i) const { return ((i < 64 ? a : b) >> (i & 63)) & 1; }
    MIBN_HD void set(int i) { (i < 64 ? a : b) |= 1ull << (i & 63); }
    MIBN_HD void clr(int i) { (i < 64 ? a : b) &= ~(1ull << (i & 63)); }
    MIBN_HD bool any() const { return (a | b) != 0; }
};
template <class F> MIBN_HD inline void b2_each(const B2 &s, F f) {
    for (uint64_t m = s.a; m; m &= m - 1) f(__builtin_ctzll(m));
    for (uint64_t m = s.b; m; m &= m - 1) f(64 + __builtin_ctzll(m));
}

// What the search reads of a network (pointers into host vectors, or into one device buffer).
struct OrderNet {
    int32_t n_vars = 0, n_hints = 0;
    const int32_t *card = nullptr;       // [n]
    const double *log2card = nullptr;    // [n]
    const int32_t *depth = nullptr;      // [n] longest path from a root
    const B2 *anc = nullptr;             // [n] ancestors
    const B2 *cpt_scope = nullptr;       // [n] scope of variable v's CPT ([*parents, v])
    const B2 *fam = nullptr;             // [n] the CPTs that mention v: v itself and its children
    const int32_t *topo_asc = nullptr;   // [n] all variables by (depth ascending, id)
    const int32_t *topo_desc = nullptr;  // [n] all variables by (depth descending, id)
    const int32_t *hint_sorted = nullptr;  // [n_hints][n] every hint as a variable list in ascending (priority, id) order
    B2 multi;                            // the variables with more than one state (the others are never axes)
    int32_t uniform_log2 = -1;           // l >= 0: every multi-state variable has 2^l states - cell counts are 2^(l x axes) and
                                         // sums of log2card are l x axes, exactly what the loops over the axes compute
    int32_t prune = 1;
    double minfill_above = 2e7;
    // Weight of an elimination that consumes exactly ONE big table (> big_cells cells) in the byte model below.  Such steps
    // chain into multi-variable passes (pair / CHAIN / SWEEP: up to five eliminations for one read and one write of the
    // table), the joins of two big tables do not and are the slowest bytes of a plan: with the SWEEP form on, candidate
    // orders are compared with the single-table steps at a quarter of their section-8(d) bytes.  (Powers of two only: the
    // host and the device search must round alike.)
    double chain_weight = 1.0, big_cells = 1024.0;
    // effort 1 (Network::order_effort): more candidate orders - the meet sweep around the query's depth, and the small eliminations min-fill
    // starts with followed by the meet sweep of the rest - and the SECOND best order kept beside the best: where the best is expensive the
    // planner emits both and keeps the program that moves fewer bytes (the byte model ranks candidates well - among its best two the
    // emitter finds 8 % fewer bytes on the C3 stream than the model's first choice of round 5's four, tools/order_quality.cpp)
    int32_t effort = 0;
};

// factor slots of the byte model: slot v < 128 = the evidence-sliced CPT of variable v, slot 128 + o = the factor the o-th
// elimination creates
constexpr int kOrderSlotWords = 4, kOrderSlots = 64 * kOrderSlotWords;
constexpr int kOrderOpening = 3;  // effort 1: variables of the factors min-fill's opening may create
// The byte model overrates the opening + meet candidate: where it ranks first with a plain meet sweep within a fifth behind, the EMITTED program of the sweep
// moves fewer bytes four times in five (/tmp-style experiment in profiles/NOTES_r06.md, session BT).  Its modelled cost counts 9/8 (exact in doubles, the same
// number on the host and on the device): the first choice alone then emits 2.9 % fewer bytes on C3, 1.6 % with eight evidence nodes.
constexpr double kOrderOpeningPenalty = 1.125;

struct OrderScratch {
    // the request: relevant variables, and per relevant variable v the scope of its CPT without the evidence axes (f[v]) and
    // its cells (fc[v]); behind them the factors the byte model creates
    B2 rel;
    B2 f[kOrderSlots];
    double fc[kOrderSlots];
    uint64_t mem[128][kOrderSlotWords];
    // min-fill
    B2 adj[128];
    double ws[128], score[128];
    int32_t miss[128];
    // candidates
    uint8_t cand[128], best[128];
    int32_t n_cand, n_best;
    // effort 1: the runner-up, and min-fill's order with the degree of every vertex when it was eliminated
    uint8_t second[128], greedy[128], gdeg[128];
    int32_t n_second, n_greedy;
    double best_cost, second_cost;
};

MIBN_HD inline double order_pow2(int e) {  // 2^e, e >= 0
    if (e > 1023) return __builtin_inf();
    const uint64_t bits = (uint64_t)(1023 + e) << 52;
    double d;
    __builtin_memcpy(&d, &bits, 8);
    return d;
}

MIBN_HD inline double order_exp2(double x) {
    // (integers - every network of two- / four- / eight-state variables - without the library call: exp2 is exact there)
    if (x >= 0 && x < 1024) {
        const int e = (int)x;
        if ((double)e == x) return order_pow2(e);
    }
#if defined(__HIP_DEVICE_COMPILE__)
    return ::exp2(x);
#else
    return std::exp2(x);
#endif
}

MIBN_HD inline int b2_count(const B2 &s) { return __builtin_popcountll(s.a) + __builtin_popcountll(s.b); }

MIBN_HD inline double order_cells(const OrderNet &net, const B2 &u) {
    if (net.uniform_log2 >= 0) return order_pow2(net.uniform_log2 * b2_count(u));
    double c = 1;
    b2_each(u, [&](int v) { c *= net.card[v]; });
    return c;
}

// sum of log2card over a set, in ascending order of the variables
MIBN_HD inline double order_log2sum(const OrderNet &net, const B2 &u) {
    if (net.uniform_log2 >= 0) return (double)(net.uniform_log2 * b2_count(u));
    double w = 0;
    b2_each(u, [&](int y) { w += net.log2card[y]; });
    return w;
}

// SURVEY section 8(d) byte model of eliminating `order` from the request's factors (order_prepare): every variable keeps the
// set of factor slots whose scope contains it, so an elimination touches only the factors it consumes.  The slots of the
// request's own factors are the variable ids, so a variable's initial set is a property of the network (OrderNet::fam)
// restricted to the relevant set - nothing is built per candidate order.
MIBN_HD inline double order_simulate(const OrderNet &net, OrderScratch &S, const uint8_t *order, int n_order, double abort_above) {
    const int kw = 2 + (n_order + 63) / 64;  // slot words in use: the request's factors + one slot per elimination
    const B2 rel = S.rel;
    b2_each(rel, [&](int v) {
        S.mem[v][0] = net.fam[v].a & rel.a;
        S.mem[v][1] = net.fam[v].b & rel.b;
        for (int k = 2; k < kw; ++k) S.mem[v][k] = 0;
    });
    uint64_t alive[kOrderSlotWords];
    alive[0] = rel.a;
    alive[1] = rel.b;
    for (int k = 2; k < kOrderSlotWords; ++k) alive[k] = 0;
    int nf = 128;
    double bytes = 0;
    for (int o = 0; o < n_order; ++o) {
        const int x = order[o];
        B2 u;
        double in = 0;
        int nbig = 0;
        for (int k = 0; k < kw; ++k) {
            uint64_t m = S.mem[x][k] & alive[k];
            alive[k] &= ~m;
            for (; m; m &= m - 1) {
                const int i = k * 64 + __builtin_ctzll(m);
                u.a |= S.f[i].a;
                u.b |= S.f[i].b;
                in += S.fc[i];
                nbig += S.fc[i] > net.big_cells;
            }
        }
        u.clr(x);
        const double uc = order_cells(net, u);
        bytes += (nbig == 1 ? 8.0 * net.chain_weight : 8.0) * (in + uc);
        if (bytes > abort_above) return bytes;
        S.f[nf] = u;
        S.fc[nf] = uc;
        alive[nf >> 6] |= 1ull << (nf & 63);
        b2_each(u, [&](int v) { S.mem[v][nf >> 6] |= 1ull << (nf & 63); });
        ++nf;
    }
    B2 u;
    double in = 0;
    for (int k = 0; k < kw; ++k)
        for (uint64_t m = alive[k]; m; m &= m - 1) {
            const int i = k * 64 + __builtin_ctzll(m);
            u.a |= S.f[i].a;
            u.b |= S.f[i].b;
            in += S.fc[i];
        }
    return bytes + 8.0 * (in + order_cells(net, u));
}

// Greedy min-fill elimination order on the interaction graph: eliminate the vertex whose elimination adds the fewest
// edges, ties by the size of the factor it creates, then by depth and id.  The fill counts are maintained
// incrementally: eliminating a vertex makes a clique of its neighbours - their counts are recomputed, from the neighbours
// they have OUTSIDE that clique only (a pair inside it is connected by construction) - and every common neighbour of a
// newly connected pair loses that pair from its count.
// `abort_above`: every factor an elimination creates is written once and read once later, so 16 bytes x the cells
// created so far is a lower bound of the order's section-8(d) cost - once it passes the best sweep the search stops
// (returns false: the order cannot win).  The order goes to S.cand.
// (Networks whose multi-state variables all have 2^l states - OrderNet::uniform_log2 - run the same search on integers: the score
// miss * 64 + l * degree is exact in a few bits, so score, depth and id pack into one 64-bit key per vertex and the choice of
// the next vertex is a branch-free minimum over the keys instead of a floating-point comparison with a tolerance and two
// tie-breaks per vertex.  Same order as the general form, bit for bit: the scores are the same numbers and an exact tie is what
// the tolerance of 1e-12 detects there.)
template <bool kUniform>
MIBN_HD inline bool order_greedy_impl(const OrderNet &net, OrderScratch &S, const B2 &hidden, double abort_above) {
    B2 *adj = S.adj;
    const B2 rel = S.rel;
    b2_each(rel, [&](int v) { adj[v] = B2{}; S.miss[v] = 0; S.ws[v] = 0; S.score[v] = 0; });
    b2_each(rel, [&](int i) {
        const B2 sc = S.f[i];
        b2_each(sc, [&](int v) { adj[v].a |= sc.a; adj[v].b |= sc.b; });
    });
    b2_each(rel, [&](int v) { adj[v].clr(v); });
    double *ws = S.ws, *score = S.score;
    int32_t *miss = S.miss;
    uint64_t *key = reinterpret_cast<uint64_t *>(S.score);  // (kUniform: the packed keys live where the scores would)
    const int l = net.uniform_log2;
    // key = (miss * 64 + l * degree) << 16 | depth << 8 | id   (depth, id < 256: networks of <= 128 variables)
    auto pack = [&](int x, int missing, int degree) { return ((uint64_t)(uint32_t)(missing * 64 + l * degree) << 16) | ((uint64_t)(uint32_t)net.depth[x] << 8) | (uint64_t)x; };
    B2 alive = hidden;  // the vertices not yet eliminated - a register pair, scanned in ascending order
    int n_alive = b2_count(hidden);
    b2_each(alive, [&](int x) {
        const B2 ax = adj[x];
        int missing = 0;  // (ordered) pairs of neighbours that are not adjacent
        b2_each(ax, [&](int y) { missing += __builtin_popcountll(ax.a & ~adj[y].a) + __builtin_popcountll(ax.b & ~adj[y].b) - 1; });
        miss[x] = missing;
        if (kUniform) {
            const int deg = b2_count(ax);
            ws[x] = (double)(l * deg);
            key[x] = pack(x, missing, deg);
        } else {
            ws[x] = order_log2sum(net, ax);
            score[x] = missing * 64.0 + ws[x];
        }
    });
    S.n_cand = 0;
    const int total = n_alive;
    double created = 0;
    for (int it = 0; it < total; ++it) {
        int best = -1;
        if (kUniform) {
            uint64_t kbest = ~0ull;
            b2_each(alive, [&](int x) { const uint64_t k = key[x]; kbest = k < kbest ? k : kbest; });
            best = (int)(kbest & 0xff);
        } else {
            int dbest = 0;
            double wbest = 0;
            b2_each(alive, [&](int x) {
                const double wx = score[x];
                const double d = wx - wbest;
                if (best < 0 || wx < wbest - 1e-12) {
                    best = x;
                    wbest = wx;
                    dbest = net.depth[x];
                } else if ((d < 0 ? -d : d) <= 1e-12) {
                    const int dx = net.depth[x];
                    if (dx < dbest || (dx == dbest && x < best)) { best = x; wbest = wx; dbest = dx; }
                }
            });
        }
        S.gdeg[S.n_cand] = (uint8_t)b2_count(adj[best]);
        S.cand[S.n_cand++] = (uint8_t)best;
        alive.clr(best);
        created += kUniform ? order_pow2((int)ws[best]) : order_exp2(ws[best]);  // cells of the factor this elimination creates (its scope = the neighbours)
        if (16.0 * net.chain_weight * created > abort_above) return false;
        const B2 nb = adj[best];
        b2_each(nb, [&](int y) {
            B2 fresh;  // members of nb not yet adjacent to y
            fresh.a = nb.a & ~adj[y].a;
            fresh.b = nb.b & ~adj[y].b;
            fresh.clr(y);
            b2_each(fresh, [&](int u) {
                if (u < y) return;
                B2 common;
                common.a = adj[y].a & adj[u].a & ~nb.a;
                common.b = adj[y].b & adj[u].b & ~nb.b;
                common.clr(best);
                b2_each(common, [&](int z) {
                    miss[z] -= 2;
                    if (kUniform) key[z] -= (uint64_t)128 << 16;  // (two ordered pairs fewer: the score drops by 2 * 64)
                    else score[z] = miss[z] * 64.0 + ws[z];
                });
            });
        });
        b2_each(nb, [&](int y) {
            adj[y].a |= nb.a;
            adj[y].b |= nb.b;
            adj[y].clr(best);
            adj[y].clr(y);
        });
        // the neighbours' own counts: nb is a clique now, so a missing pair of y has at least one end outside it -
        //   miss[y] = sum over e in ext = adj[y] - nb of  |nb - y - adj[e]| (pairs (w, e), counted from e's side: once
        //             here and once as (e, w) below)  +  |adj[y] - adj[e]| - 1 (pairs (e, w'), e itself taken off)
        const B2 live_nb{nb.a & alive.a, nb.b & alive.b};
        b2_each(live_nb, [&](int y) {
            const B2 ay = adj[y];
            B2 ext, nbm = nb;
            ext.a = ay.a & ~nb.a;
            ext.b = ay.b & ~nb.b;
            nbm.clr(y);
            int missing = 0;
            b2_each(ext, [&](int e) {
                const uint64_t na = ~adj[e].a, nb_ = ~adj[e].b;
                missing += __builtin_popcountll(nbm.a & na) + __builtin_popcountll(nbm.b & nb_) + __builtin_popcountll(ay.a & na) +
                           __builtin_popcountll(ay.b & nb_) - 1;
            });
            miss[y] = missing;
            if (kUniform) {
                const int deg = b2_count(ay);
                ws[y] = (double)(l * deg);
                key[y] = pack(y, missing, deg);
            } else {
                ws[y] = order_log2sum(net, ay);
                score[y] = missing * 64.0 + ws[y];
            }
        });
    }
    return true;
}

MIBN_HD inline bool order_greedy(const OrderNet &net, OrderScratch &S, const B2 &hidden, double abort_above) {
    // (depth and id take 8 bits each of a packed key; a score of a 128-vertex graph stays far below 2^31)
    if (net.uniform_log2 >= 0 && net.n_vars <= 128) return order_greedy_impl<true>(net, S, hidden, abort_above);
    return order_greedy_impl<false>(net, S, hidden, abort_above);
}

// Relevant set, hidden set and the factor scopes (S.f / S.fc of the relevant variables) of one request.
MIBN_HD inline void order_prepare(const OrderNet &net, OrderScratch &S, int nq, const int32_t *qvars, int ne, const int32_t *evars,
                                  bool no_prune, B2 &rel, B2 &hidden) {
    B2 qb, eb;
    rel = B2{};
    for (int i = 0; i < nq; ++i) { const int v = qvars[i]; qb.set(v); rel.set(v); rel.a |= net.anc[v].a; rel.b |= net.anc[v].b; }
    for (int i = 0; i < ne; ++i) { const int v = evars[i]; eb.set(v); rel.set(v); rel.a |= net.anc[v].a; rel.b |= net.anc[v].b; }
    if (!net.prune || no_prune)
        for (int v = 0; v < net.n_vars; ++v) rel.set(v);
    hidden.a = rel.a & ~qb.a & ~eb.a;
    hidden.b = rel.b & ~qb.b & ~eb.b;
    S.rel = rel;
    // factors = the CPTs of the relevant variables with the evidence (and single-state) axes removed
    const B2 keep{net.multi.a & ~eb.a, net.multi.b & ~eb.b};
    b2_each(rel, [&](int v) {
        B2 sc;
        sc.a = net.cpt_scope[v].a & keep.a;
        sc.b = net.cpt_scope[v].b & keep.b;
        S.f[v] = sc;
        S.fc[v] = order_cells(net, sc);
    });
    // single-state variables carry no information: they are never axes, never eliminated
    hidden.a &= net.multi.a;
    hidden.b &= net.multi.b;
}

// The two sweep candidates into S.cand (behind its first n0 entries): which = 0 "meet" (down from the roots to depth `qdepth`, then up
// from the leaves), 1 = reverse topological.
MIBN_HD inline void order_sweep(const OrderNet &net, OrderScratch &S, const B2 &hidden, int qdepth, int which, int n0 = 0) {
    S.n_cand = n0;
    auto filtered = [&](const int32_t *sorted_all, int lo_depth, int hi_depth) {
        for (int i = 0; i < net.n_vars; ++i) {
            const int v = sorted_all[i];
            if (hidden.test(v) && net.depth[v] >= lo_depth && net.depth[v] < hi_depth) S.cand[S.n_cand++] = (uint8_t)v;
        }
    };
    const int kNoDepth = 0x7fffffff;
    if (which == 0) {
        filtered(net.topo_asc, 0, qdepth);
        filtered(net.topo_desc, qdepth, kNoDepth);
    } else {
        filtered(net.topo_desc, 0, kNoDepth);
    }
}

// The whole search for one request.  Fills S.best / S.n_best (hidden variables, first eliminated first) and returns the
// modelled cost of that order (infinity when nothing is hidden).  net.effort >= 1: more candidates, and the runner-up in S.second /
// S.n_second (0: none) with S.best_cost / S.second_cost.
MIBN_HD inline double order_search(const OrderNet &net, OrderScratch &S, int nq, const int32_t *qvars, int ne, const int32_t *evars,
                                   bool no_prune) {
    B2 rel, hidden;
    order_prepare(net, S, nq, qvars, ne, evars, no_prune, rel, hidden);
    S.n_best = 0;
    S.n_second = 0;
    S.n_greedy = 0;
    double best_cost = __builtin_inf(), second_cost = __builtin_inf();
    S.best_cost = S.second_cost = best_cost;
    if (!hidden.any()) return best_cost;
    const bool two = net.effort >= 1;
    auto same = [&](const uint8_t *a, int na, const uint8_t *b, int nb) {
        if (na != nb) return false;
        for (int i = 0; i < na; ++i)
            if (a[i] != b[i]) return false;
        return true;
    };
    auto consider = [&](double penalty = 1.0) {  // evaluates S.cand
        const double c = penalty * order_simulate(net, S, S.cand, S.n_cand, two ? second_cost : best_cost);
        if (c < best_cost) {
            if (two && S.n_best > 0) {
                second_cost = best_cost;
                S.n_second = S.n_best;
                for (int i = 0; i < S.n_best; ++i) S.second[i] = S.best[i];
            }
            best_cost = c;
            S.n_best = S.n_cand;
            for (int i = 0; i < S.n_cand; ++i) S.best[i] = S.cand[i];
        } else if (two && c < second_cost && !same(S.cand, S.n_cand, S.best, S.n_best)) {
            second_cost = c;
            S.n_second = S.n_cand;
            for (int i = 0; i < S.n_cand; ++i) S.second[i] = S.cand[i];
        }
    };
    int qdepth = 0x7fffffff;
    for (int i = 0; i < nq; ++i) qdepth = net.depth[qvars[i]] < qdepth ? net.depth[qvars[i]] : qdepth;
    order_sweep(net, S, hidden, qdepth, 0);
    consider();
    // (a plain topological sweep wins on < 1 % of the C3 requests: not worth its simulation)
    order_sweep(net, S, hidden, qdepth, 1);
    consider();
    for (int h = 0; h < net.n_hints; ++h) {
        const int32_t *sorted_all = net.hint_sorted + (int64_t)h * net.n_vars;
        S.n_cand = 0;
        for (int i = 0; i < net.n_vars; ++i)
            if (hidden.test(sorted_all[i])) S.cand[S.n_cand++] = (uint8_t)sorted_all[i];
        consider();
    }
    if (two && nq > 0)  // the two sweeps meet one level above and one below the query's
        for (int d = -1; d <= 1; d += 2) {
            if (qdepth + d < 0) continue;
            order_sweep(net, S, hidden, qdepth + d, 0);
            consider();
        }
    // greedy min-fill: the best order on 60 % of the C3 requests (52.7 MB mean against 67.6 MB for the sweeps alone),
    // skipped where the sweeps already found a plan too cheap to be worth the time
    if (best_cost > net.minfill_above * net.chain_weight) {
        const bool whole = order_greedy(net, S, hidden, two ? second_cost : best_cost);
        if (two) {
            S.n_greedy = S.n_cand;  // (aborted: what it had eliminated so far - the small eliminations come first)
            for (int i = 0; i < S.n_cand; ++i) S.greedy[i] = S.cand[i];
        }
        if (whole) consider();
        if (two) {
            // min-fill's opening - its leading eliminations that create factors of at most kOrderOpening variables - then the meet sweep
            // of the rest (wider openings add little: 2.0 / 1.7 / 1.6 / 1.3 % fewer bytes alone at 4 .. 7 against 5.4 % at 3)
            int np = 0;
            while (np < S.n_greedy && S.gdeg[np] <= kOrderOpening) ++np;
            if (np > 0 && !(np == S.n_greedy && whole)) {
                B2 rest = hidden;
                for (int i = 0; i < np; ++i) { S.cand[i] = S.greedy[i]; rest.clr(S.greedy[i]); }
                order_sweep(net, S, rest, qdepth, 0, np);
                consider(kOrderOpeningPenalty);
            }
        }
    }
    S.best_cost = best_cost;
    S.second_cost = second_cost;
    return best_cost;
}

}  // namespace mibn
