// Host-side planner: request -> step program (see planner.h for the role and the encoding).
#include "planner.h"

#include <algorithm>
#include <unordered_map>
#include <atomic>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <limits>
#include <memory>
#include <numeric>
#include <condition_variable>
#include <mutex>
#include <thread>

namespace mibn {

#ifdef MIBN_PLAN_PROFILE
#include <chrono>
double g_prof[8];
struct ProfT { int k; std::chrono::steady_clock::time_point t0; ProfT(int k_) : k(k_), t0(std::chrono::steady_clock::now()) {} ~ProfT() { g_prof[k] += std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count(); } };
#define PROF(k) ProfT prof_##k(k)
#else
#define PROF(k)
#endif

// ------------------------------------------------------------------------------------ Network

std::string Network::set(int32_t n, const int32_t *card_, const int64_t *scope_off, const int32_t *scope_vars,
                         const int64_t *value_off, const double *values) {
    if (n < 0 || n > kMaxVars) return "n_vars out of range (max " + std::to_string(kMaxVars) + ")";
    static std::atomic<uint64_t> versions{0};
    version = ++versions;
    n_vars = n;
    nw = std::max(1, (n + 63) / 64);
    card.assign(card_, card_ + n);
    log2card.resize(n);
    scope.assign(n, {});
    cstride.assign(n, {});
    pool_off.assign(n, 0);
    cells.assign(n, 0);
    for (int v = 0; v < n; ++v) {
        if (card[v] < 1) return "cardinality of variable " + std::to_string(v) + " must be >= 1";
        log2card[v] = std::log2((double)card[v]);
    }
    for (int v = 0; v < n; ++v) {
        int64_t b = scope_off[v], e = scope_off[v + 1];
        if (e <= b) return "factor " + std::to_string(v) + " has an empty scope";
        scope[v].assign(scope_vars + b, scope_vars + e);
        if (scope[v].back() != v) return "scope of factor " + std::to_string(v) + " must end with the variable itself";
        for (size_t i = 0; i < scope[v].size(); ++i) {
            int32_t u = scope[v][i];
            if (u < 0 || u >= n) return "scope of factor " + std::to_string(v) + " names an unknown variable";
            for (size_t k = 0; k < i; ++k)
                if (scope[v][k] == u) return "scope of factor " + std::to_string(v) + " repeats a variable";
        }
        cstride[v].assign(scope[v].size(), 1);
        int64_t s = 1;
        for (int i = (int)scope[v].size() - 1; i >= 0; --i) {
            cstride[v][i] = s;
            s *= card[scope[v][i]];
            if (s >= (1ll << 31)) return "CPT of variable " + std::to_string(v) + " has >= 2^31 cells";
        }
        cells[v] = s;
        if (value_off[v + 1] - value_off[v] != s) return "value table of factor " + std::to_string(v) + " has the wrong size";
        pool_off[v] = value_off[v];
    }
    pool.assign(values, values + value_off[n]);
    // ancestors + depth by DFS with cycle detection
    anc.assign(n, Bits{});
    for (auto &b : anc) b.nw = nw;
    depth.assign(n, -1);
    std::vector<int8_t> state(n, 0);
    std::string err;
    std::function<void(int)> visit = [&](int v) {
        if (state[v] == 2 || !err.empty()) return;
        if (state[v] == 1) { err = "the network has a cycle"; return; }
        state[v] = 1;
        int d = 0;
        for (size_t i = 0; i + 1 < scope[v].size(); ++i) {
            int p = scope[v][i];
            visit(p);
            if (!err.empty()) return;
            anc[v].set(p);
            anc[v].or_(anc[p]);
            d = std::max(d, depth[p] + 1);
        }
        depth[v] = d;
        state[v] = 2;
    };
    for (int v = 0; v < n; ++v) visit(v);
    hints.clear();
    hint_sorted.clear();
    // the sweep orders of the planner are filtered from these lists (no per-request sorting)
    topo_asc.resize(n);
    std::iota(topo_asc.begin(), topo_asc.end(), 0);
    topo_desc = topo_asc;
    if (err.empty()) {
        std::stable_sort(topo_asc.begin(), topo_asc.end(), [&](int a, int b) { return depth[a] < depth[b]; });
        std::stable_sort(topo_desc.begin(), topo_desc.end(), [&](int a, int b) { return depth[a] > depth[b]; });
    }
    anc2.clear();
    scope2.clear();
    fam2.clear();
    multi2 = B2{};
    uniform_log2 = -1;
    hint_flat.clear();
    if (n <= 128 && err.empty()) {
        anc2.resize(n);
        scope2.resize(n);
        fam2.resize(n);
        int l = -2;  // -2: no multi-state variable seen yet
        for (int v = 0; v < n; ++v) {
            anc2[v].a = anc[v].w[0];
            anc2[v].b = anc[v].w[1];
            for (int32_t u : scope[v]) { scope2[v].set(u); fam2[u].set(v); }
            if (card[v] > 1) {
                multi2.set(v);
                const int lv = (card[v] & (card[v] - 1)) == 0 ? __builtin_ctz((unsigned)card[v]) : -1;
                l = l == -2 ? lv : (l == lv ? l : -1);
            }
        }
        uniform_log2 = l < 0 ? -1 : l;
    }
    scope_off32.assign(1, 0);
    scope_flat.clear();
    cstride_flat.clear();
    anc_flat.assign((size_t)n * nw, 0);
    for (int v = 0; v < n; ++v) {
        scope_flat.insert(scope_flat.end(), scope[v].begin(), scope[v].end());
        cstride_flat.insert(cstride_flat.end(), cstride[v].begin(), cstride[v].end());
        scope_off32.push_back((int32_t)scope_flat.size());
        for (int k = 0; k < nw; ++k) anc_flat[(size_t)v * nw + k] = anc[v].w[k];
    }
    if (err.empty()) set_hints(0, nullptr);  // the built-in sweep lists
    return err;
}

EmitNet Network::emit_view() const {
    EmitNet e;
    e.n_vars = n_vars;
    e.nw = nw;
    e.card = card.data();
    e.log2card = log2card.data();
    e.pool_off = pool_off.data();
    e.scope_off = scope_off32.data();
    e.scope_vars = scope_flat.data();
    e.scope_stride = cstride_flat.data();
    e.anc = anc_flat.data();
    e.small_cells = small_cells;
    e.big_iters = big_iters;
    e.log2_small = std::log2((double)small_cells);
    e.log2_big = std::log2((double)big_iters);
    e.uniform_log2 = uniform_log2;
    e.prune = prune; e.outer = outer; e.fuse = fuse; e.chain = chain; e.sweep = sweep; e.sweep_min = sweep_min; e.sweep_canon = sweep_canon;
    e.tile_h = tile_h;
    e.sweep_iters = sweep_iters;
    e.tile_bytes = tile_bytes;
    return e;
}

// Two depth-first topological orders (Kahn's algorithm with a stack; the children of a finished node pushed in ascending /
// descending id order): on a grid the column-major and the row-major sweep, on any DAG two sweeps that finish one branch
// before they start the next.  Candidate elimination orders next to the host's hints.
static void lifo_topological(const Network &net, bool ascending, std::vector<int32_t> &out) {
    const int n = net.n_vars;
    std::vector<int> indeg(n, 0);
    std::vector<std::vector<int32_t>> children(n);
    for (int v = 0; v < n; ++v)
        for (size_t i = 0; i + 1 < net.scope[v].size(); ++i) {
            children[net.scope[v][i]].push_back(v);
            ++indeg[v];
        }
    std::vector<int32_t> stack;
    for (int v = 0; v < n; ++v) {
        const int r = ascending ? v : n - 1 - v;
        if (indeg[r] == 0) stack.push_back(r);
    }
    out.clear();
    while (!stack.empty()) {
        const int v = stack.back();
        stack.pop_back();
        out.push_back(v);
        std::vector<int32_t> &ch = children[v];
        std::sort(ch.begin(), ch.end());
        if (!ascending) std::reverse(ch.begin(), ch.end());
        for (int32_t c : ch)
            if (--indeg[c] == 0) stack.push_back(c);
    }
}

OrderNet Network::order_view() const {
    OrderNet o;
    o.n_vars = n_vars;
    o.n_hints = (int32_t)hint_sorted.size();
    o.card = card.data();
    o.log2card = log2card.data();
    o.depth = depth.data();
    o.anc = anc2.data();
    o.cpt_scope = scope2.data();
    o.fam = fam2.data();
    o.multi = multi2;
    o.uniform_log2 = uniform_log2;
    o.topo_asc = topo_asc.data();
    o.topo_desc = topo_desc.data();
    o.hint_sorted = hint_flat.data();
    o.prune = prune;
    o.minfill_above = minfill_above;
    o.chain_weight = (fuse && sweep >= 4 && order_weights) ? (order_weights == 1 ? 0.25 : 1.0 / (double)order_weights) : 1.0;  // (option value k > 1: weight 1 / k)
    o.big_cells = (double)small_cells;
    o.effort = order_effort;
    return o;
}

// The network and the options of the moment, packed for the wave planner (wave_plan.h: one request per wave, this struct in LDS).
bool Network::wave_view(WNet &w) const {
    if (n_vars < 1 || n_vars > kWVars || uniform_log2 < 0 || (int)hint_sorted.size() > kWHints || (int)scope_flat.size() > kWCsr) return false;
    if ((int64_t)pool.size() >= (1ll << 32)) return false;
    const OrderNet ov = order_view();
    const EmitNet ev = emit_view();
    std::memset(&w, 0, sizeof(w));
    w.n_vars = n_vars;
    w.n_hints = (int32_t)hint_sorted.size();
    w.uniform_log2 = uniform_log2;
    w.csr_n = (int32_t)scope_flat.size();
    w.multi = multi2;
    w.small_cells = ev.small_cells; w.prune = ev.prune; w.outer = ev.outer; w.fuse = ev.fuse; w.chain = ev.chain; w.sweep = ev.sweep;
    w.sweep_min = ev.sweep_min; w.sweep_canon = ev.sweep_canon; w.tile_h = ev.tile_h; w.sweep_iters = ev.sweep_iters;
    w.big_iters = ev.big_iters; w.tile_bytes = ev.tile_bytes;
    w.log2_small = ev.log2_small; w.log2_big = ev.log2_big;
    w.minfill_above = ov.minfill_above; w.chain_weight = ov.chain_weight; w.big_cells = ov.big_cells;
    w.effort = ov.effort; w.second_above = second_above;
    // (the byte model runs on integers: 8 x the chain weight must be one, and the cost of an order stays below 2^53)
    if (!(ov.chain_weight == 1.0 || ov.chain_weight == 0.5 || ov.chain_weight == 0.25 || ov.chain_weight == 0.125)) return false;
    w.big_log2 = -1;
    for (int e = 0; e < 40; ++e)
        if (ov.big_cells == (double)(1ull << e)) w.big_log2 = e;
    for (int v = 0; v < n_vars; ++v) {
        if (card[v] > 65535 || depth[v] > 255 || pool_off[v] < 0) return false;
        w.scope[v] = scope2[v];
        w.fam[v] = fam2[v];
        w.card[v] = (uint16_t)card[v];
        w.depth[v] = (uint8_t)depth[v];
        w.topo_asc[v] = (uint8_t)topo_asc[v];
        w.topo_desc[v] = (uint8_t)topo_desc[v];
        w.pool_off[v] = (uint32_t)pool_off[v];
        w.scope_off[v] = (uint16_t)scope_off32[v];
        for (size_t h = 0; h < hint_sorted.size(); ++h) w.hint_sorted[h][v] = (uint8_t)hint_sorted[h][v];
    }
    w.scope_off[n_vars] = (uint16_t)scope_off32[n_vars];
    for (size_t k = 0; k < scope_flat.size(); ++k) {
        if (cstride_flat[k] < 0 || cstride_flat[k] >= (1ll << 31)) return false;
        w.scope_var[k] = (uint8_t)scope_flat[k];
        w.scope_stride[k] = (int32_t)cstride_flat[k];
    }
    return true;
}

void Network::set_hints(int32_t n_hints, const int32_t *priorities) {
    hints.clear();
    hint_sorted.clear();
    hint_flat.clear();
    auto add_list = [&](std::vector<int32_t> &&o) {
        if ((int)o.size() != n_vars) return;
        for (const auto &have : hint_sorted)
            if (have == o) return;  // (the host's name order of a grid IS one of the built-in sweeps)
        hint_flat.insert(hint_flat.end(), o.begin(), o.end());
        hint_sorted.push_back(std::move(o));
    };
    for (int i = 0; i < n_hints; ++i) {
        hints.emplace_back(priorities + (size_t)i * n_vars, priorities + (size_t)(i + 1) * n_vars);
        std::vector<int32_t> o(n_vars);
        std::iota(o.begin(), o.end(), 0);
        const std::vector<int32_t> &h = hints.back();
        std::stable_sort(o.begin(), o.end(), [&](int a, int b) { return h[a] < h[b]; });
        add_list(std::move(o));
    }
    if (builtin_sweeps)
        for (int asc = 0; asc < 2; ++asc) {
            std::vector<int32_t> o;
            lifo_topological(*this, asc != 0, o);
            add_list(std::move(o));
        }
}

bool request_is_valid(const Network &net, const Request &rq) {
    if (rq.nq < 1) return false;
    uint64_t seen[kWords];  // (only the words the network uses: a batch validates 100 k requests on one thread)
    for (int k = 0; k < net.nw; ++k) seen[k] = 0;
    for (int i = 0; i < rq.nq + rq.ne; ++i) {
        const int v = i < rq.nq ? rq.qvars[i] : rq.evars[i - rq.nq];
        if (v < 0 || v >= net.n_vars) return false;
        const uint64_t bit = 1ull << (v & 63);
        if (seen[v >> 6] & bit) return false;
        seen[v >> 6] |= bit;
    }
    return true;
}

std::string validate_request(const Network &net, const Request &rq) {
    if (request_is_valid(net, rq)) return "";
    if (rq.nq < 1) return "At least one query variable has to be specified";  // bayes_net.py:840-841
    Bits seen;
    for (int i = 0; i < rq.nq; ++i) {
        int v = rq.qvars[i];
        if (v < 0 || v >= net.n_vars) return "unknown query variable id " + std::to_string(v);
        if (seen.test(v)) return "duplicate query variable id " + std::to_string(v);
        seen.set(v);
    }
    Bits ev;
    for (int i = 0; i < rq.ne; ++i) {
        int v = rq.evars[i];
        if (v < 0 || v >= net.n_vars) return "unknown evidence variable id " + std::to_string(v);
        if (seen.test(v)) return "A query variable cannot be part of the event";  // bayes_net.py:843-845
        if (ev.test(v)) return "duplicate evidence variable id " + std::to_string(v);
        ev.set(v);
    }
    return "";
}

// ------------------------------------------------------------------------------------ orders

namespace {

inline double scope_log2(const Network &net, const Bits &b) {
    double s = 0;
    b.for_each([&](int v) { s += net.log2card[v]; });
    return s;
}

// per-thread scratch reused across requests
struct Scratch {
    std::vector<Bits> sim;
    std::vector<double> simc;
    std::vector<int32_t> hid;
    std::vector<int> miss;
    std::vector<Bits> scopes;
    std::vector<double> scope_cells;
    std::vector<Bits> adj;
    std::vector<double> w;
    std::vector<char> alive;
    // the request's emission state (EmitScratch of emit_core.h points into these)
    std::vector<PF> pool;
    std::vector<int32_t> pos, ecode;
    std::vector<double> key;
    std::vector<uint64_t> mem, slot_alive;
    std::vector<const PF *> ins;
    EmitScratch es;
    std::vector<int32_t> cand, best;   // candidate elimination orders (no per-request heap traffic)
    std::vector<int32_t> second;       // the byte model's runner-up (order_effort >= 1)
    std::vector<uint32_t> keep_words;  // the first order's program while the runner-up is emitted
};
Scratch &scratch() {
    static thread_local Scratch s;
    return s;
}
OrderScratch &order_scratch() {  // state of the shared host / device order search (order_search.h), ~35 KB per planning thread
    static thread_local OrderScratch s;
    return s;
}

inline double scope_cells(const Network &net, const Bits &b) {
    double c = 1;
    b.for_each([&](int v) { c *= net.card[v]; });
    return c;
}

// SURVEY section 8(d) byte model of an elimination order over factor scopes (f0c = cells of every scope).
double simulate(const Network &net, const std::vector<Bits> &f0, const std::vector<double> &f0c, const std::vector<int32_t> &order,
                double abort_above) {
    Scratch &S = scratch();
    std::vector<Bits> &f = S.sim;
    std::vector<double> &fc = S.simc;
    f.assign(f0.begin(), f0.end());
    fc.assign(f0c.begin(), f0c.end());
    double bytes = 0;
    for (int32_t x : order) {
        Bits u;
        u.nw = net.nw;
        double in = 0;
        const int xw = x >> 6;
        const uint64_t xm = 1ull << (x & 63);
        for (size_t i = 0; i < f.size();) {
            if (f[i].w[xw] & xm) {  // consumed: swap-remove (the order of the factors does not matter)
                u.or_(f[i]);
                in += fc[i];
                if (i + 1 != f.size()) { f[i] = f.back(); fc[i] = fc.back(); }
                f.pop_back();
                fc.pop_back();
            } else {
                ++i;
            }
        }
        u.clr(x);
        const double uc = scope_cells(net, u);
        bytes += 8.0 * (in + uc);
        if (bytes > abort_above) return bytes;
        f.push_back(u);
        fc.push_back(uc);
    }
    Bits u;
    u.nw = net.nw;
    double in = 0;
    for (size_t i = 0; i < f.size(); ++i) {
        u.or_(f[i]);
        in += fc[i];
    }
    bytes += 8.0 * (in + scope_cells(net, u));
    return bytes;
}

// Networks of more than 128 variables (order_search.h handles the others, on the host and on the device).
// Greedy min-fill elimination order on the interaction graph: eliminate the vertex whose elimination adds the
// fewest edges, ties by the size of the factor it creates, then by depth and id.  The fill counts are maintained
// incrementally: eliminating a vertex changes the neighbourhood of its neighbours (recomputed) and connects pairs
// of them - every common neighbour of a newly connected pair loses that pair from its fill count.
bool greedy_order(const Network &net, const std::vector<Bits> &f, const Bits &hidden, std::vector<int32_t> &order, double abort_above) {
    const int n = net.n_vars, nw = net.nw;
    (void)abort_above;  // (networks above 128 variables: the generic bit sets, no lower-bound abort)
    Scratch &S = scratch();
    S.adj.assign(n, Bits{});
    for (auto &a : S.adj) a.nw = nw;
    for (auto &s : f) s.for_each([&](int v) { S.adj[v].or_(s); });
    for (int v = 0; v < n; ++v) S.adj[v].clr(v);
    std::vector<Bits> &adj = S.adj;
    std::vector<int32_t> &hid = S.hid;
    hid.clear();
    hidden.for_each([&](int v) { hid.push_back(v); });
    S.w.assign(n, 0.0);
    S.miss.assign(n, 0);
    S.alive.assign(n, 0);
    auto full = [&](int x) {
        S.w[x] = scope_log2(net, adj[x]);
        int missing = 0;  // (ordered) pairs of neighbours that are not yet adjacent
        const Bits &ax = adj[x];
        ax.for_each([&](int y) {
            const Bits &ay = adj[y];
            for (int k = 0; k < nw; ++k) missing += __builtin_popcountll(ax.w[k] & ~ay.w[k]);
            missing -= 1;  // y itself is in adj[x] but not in adj[y]
        });
        S.miss[x] = missing;
    };
    for (int x : hid) { full(x); S.alive[x] = 1; }
    order.clear();
    size_t n_alive = hid.size();
    for (size_t it = 0, total = hid.size(); it < total; ++it) {
        int best = -1;
        double wbest = 0;
        size_t k = 0;
        for (size_t i = 0; i < n_alive; ++i) {  // compacts the alive list while scanning it
            const int x = hid[i];
            if (!S.alive[x]) continue;
            hid[k++] = x;
            const double wx = S.miss[x] * 64.0 + S.w[x];
            if (best < 0 || wx < wbest - 1e-12 ||
                (std::fabs(wx - wbest) <= 1e-12 &&
                 (net.depth[x] < net.depth[best] || (net.depth[x] == net.depth[best] && x < best)))) {
                best = x;
                wbest = wx;
            }
        }
        n_alive = k;
        order.push_back(best);
        S.alive[best] = 0;
        const Bits nb = adj[best];
        // pairs of neighbours this elimination connects: their common neighbours outside nb lose one missing pair
        nb.for_each([&](int y) {
            Bits fresh = nb;  // members of nb not yet adjacent to y
            fresh.andnot(adj[y]);
            fresh.clr(y);
            fresh.for_each([&](int u) {
                if (u < y) return;
                Bits common;
                common.nw = nw;
                for (int q = 0; q < nw; ++q) common.w[q] = adj[y].w[q] & adj[u].w[q] & ~nb.w[q];
                common.clr(best);
                common.for_each([&](int z) { S.miss[z] -= 2; });
            });
        });
        nb.for_each([&](int y) {
            adj[y].or_(nb);
            adj[y].clr(best);
            adj[y].clr(y);
        });
        nb.for_each([&](int y) { if (S.alive[y]) full(y); });
    }
    return true;
}

// Where a request's program depends on its evidence *codes* and on its position in the batch - and nowhere else: the
// offsets of the evidence-sliced CPTs and the result offset of the final step (plan templates, plan_batch).
struct PlanRecord {
    std::vector<std::pair<uint32_t, int32_t>> consts;  // (word index of an (off lo, off hi) pair in the buffer, CPT variable)
    std::vector<uint32_t> finals;                       // word index of the (out_off lo, out_off hi) pair of a FINAL step
};

}  // namespace

#if defined(MIBN_EMIT_PROF)
EmitProf g_host_emit_prof = {};
#endif

// host side of emit_core.h: growth of the ProgBuf behind an EmitBuf, the plan templates' record
uint32_t *emit_buf_grow(EmitBuf &b, size_t words) {
    uint32_t *p = b.host->extend(words);
    b.data = b.host->data;
    b.size = b.host->size;
    b.cap = b.host->cap;
    return p;
}
void emit_rec_const(void *rec, uint32_t word, int32_t cpt_var) { static_cast<PlanRecord *>(rec)->consts.emplace_back(word, cpt_var); }
void emit_rec_final(void *rec, uint32_t word) { static_cast<PlanRecord *>(rec)->finals.push_back(word); }

std::string emit_error_message(int err) {
    switch (err) {
        case 0: return "";
        case kEmitErrStepAxes: return "a step has more than " + std::to_string(kMaxAxes) + " axes";
        case kEmitErrFactorAxes: return "a factor has more than " + std::to_string(kRawAxes) + " axes";
        case kEmitErrCells: return "an intermediate factor has >= 2^31 cells";
        case kEmitErrCptAxes: return "a CPT has more than " + std::to_string(kRawAxes) + " free axes";
        case kEmitErrPool: return "planner factor pool exhausted";
        case kEmitErrWords: return "a request's program does not fit its slot of the device program buffer";
    }
    return "planner error " + std::to_string(err);
}

// One request on the host: the shared emission (emit_core.h) around the choice of the elimination order.
static std::string plan_request_rec(const Network &net, const Request &rq, ProgBuf &prog, PlanStats &st, PlanRecord *rec) {
    PROF(0);
    Scratch &S = scratch();
    const EmitNet en = net.emit_view();
    // the request's planning state lives in per-thread vectors
    EmitScratch &ES = S.es;
    {
        const size_t cap = (size_t)emit_pool_cap(net.n_vars), sw = (cap + 63) / 64, n = (size_t)net.n_vars;
        if (S.pool.size() < cap) S.pool.resize(cap);
        if (S.key.size() < n) { S.key.resize(n); S.pos.resize(n); S.ecode.resize(n); }
        if (S.mem.size() < n * sw) S.mem.resize(n * sw);
        if (S.slot_alive.size() < sw) S.slot_alive.resize(sw);
        if (S.ins.size() < n + 8) S.ins.resize(n + 8);
        ES.pool = S.pool.data();
        ES.pool_cap = (int32_t)cap;
        ES.sw = (int32_t)sw;
        ES.key = S.key.data();
        ES.pos = S.pos.data();
        ES.ecode = S.ecode.data();
        ES.mem = S.mem.data();
        ES.alive = S.slot_alive.data();
        ES.ins = S.ins.data();
    }
    // relevant = query | event | ancestors(...)  (bayes_net.py:763-765); hidden = relevant - query - event (766); factors =
    // evidence-sliced CPTs of the relevant nodes (768-776)
    if (int e = emit_begin(en, ES, rq.nq, rq.qvars, rq.ne, rq.evars, rq.ecodes, rq.no_prune)) return emit_error_message(e);
    const Bits &hidden = ES.hidden;

    // candidate elimination orders, cheapest by the byte model wins
    std::vector<int32_t> &best = S.best, &cand = S.cand;
    best.clear();
    std::vector<int32_t> &second = S.second;
    second.clear();
    if (rq.n_order >= 0) {
        // the order was found by the device order search (same code, order_search.h)
        best.assign(rq.order, rq.order + rq.n_order);
    } else if (net.n_vars <= 128) {
        PROF(1);
        OrderScratch &OS = order_scratch();
        order_search(net.order_view(), OS, rq.nq, rq.qvars, rq.ne, rq.evars, rq.no_prune);
        best.assign(OS.best, OS.best + OS.n_best);
        if (net.order_effort >= 1 && OS.n_second > 0 && OS.best_cost >= net.second_above) second.assign(OS.second, OS.second + OS.n_second);
    } else if (hidden.any()) {
        std::vector<Bits> &scopes = S.scopes;
        std::vector<double> &scells = S.scope_cells;
        scopes.clear();
        scells.clear();
        for (int i = 0; i < ES.n0; ++i) { scopes.push_back(ES.pool[i].scope); scells.push_back((double)ES.pool[i].cells); }
        double best_cost = std::numeric_limits<double>::infinity();
        auto consider = [&]() {  // evaluates `cand`
            const double c = simulate(net, scopes, scells, cand, best_cost);
            if (c < best_cost) { best_cost = c; best.swap(cand); }
        };
        int qdepth = std::numeric_limits<int>::max();
        for (int i = 0; i < rq.nq; ++i) qdepth = std::min(qdepth, (int)net.depth[rq.qvars[i]]);
        // the candidate sweeps are the hidden variables in the order of a per-network sorted list (Network::set /
        // set_hints): filtered, not sorted, per request
        auto filtered = [&](const std::vector<int32_t> &sorted_all, int lo_depth, int hi_depth) {
            for (int32_t v : sorted_all)
                if (hidden.test(v) && net.depth[v] >= lo_depth && net.depth[v] < hi_depth) cand.push_back(v);
        };
        constexpr int kNoDepth = std::numeric_limits<int>::max();
        {
            PROF(1);
            // "meet": sweep down from the roots to the query's depth, then up from the leaves
            cand.clear();
            filtered(net.topo_asc, 0, qdepth);
            filtered(net.topo_desc, qdepth, kNoDepth);
            consider();
            // (a plain topological sweep wins on < 1 % of the C3 requests: not worth its simulation)
            cand.clear();
            filtered(net.topo_desc, 0, kNoDepth);  // reverse sweep
            consider();
            for (auto &h : net.hint_sorted) {
                cand.clear();
                filtered(h, 0, kNoDepth);
                consider();
            }
        }
        // greedy min-fill: the best order on 60 % of the C3 requests (52.7 MB mean against 67.6 MB for the sweeps alone),
        // skipped where the sweeps already found a plan too cheap to be worth the host time
        if (best_cost > net.minfill_above) {
            PROF(3);
            cand.clear();
            if (greedy_order(net, scopes, hidden, cand, best_cost)) consider();
        }
    }

    PROF(4);
    if (st.order) *st.order = best;
    EmitBuf eb;
    eb.host = &prog;
    eb.data = prog.data;
    eb.size = prog.size;
    eb.cap = prog.cap;
    EmitStats es;
    es.alg_bytes = st.alg_bytes; es.alg_flops = st.alg_flops; es.n_steps = st.n_steps; es.max_step_cells = st.max_step_cells;
    es.arena_cells = st.arena_cells; es.out_cells = st.out_cells;
#if defined(MIBN_EMIT_PROF)  // (tools/planner_prof.cpp, single-threaded: phase times of the emission in TSC ticks)
    EmitProf &prof_ = g_host_emit_prof;
    prof_.t = __builtin_ia32_rdtsc();
#endif
    const size_t start_words = prog.size;
    const EmitStats es0 = es;
    const size_t rec_consts0 = rec ? rec->consts.size() : 0, rec_finals0 = rec ? rec->finals.size() : 0;
    int e = emit_run(en, ES, eb, es, rec, rq.nq, rq.qvars, rq.out_off, best.data(), (int)best.size() MIBN_PROF_PASS);
    if (!e && !second.empty()) {
        // the runner-up of the byte model emitted too: the program that moves fewer bytes stays (the first on a tie, or if the
        // second cannot be emitted)
        std::vector<uint32_t> &keep = S.keep_words;
        keep.assign(prog.data + start_words, prog.data + eb.size);
        const EmitStats es_first = es;
        PlanRecord rec_first;
        if (rec) {
            rec_first.consts.assign(rec->consts.begin() + rec_consts0, rec->consts.end());
            rec_first.finals.assign(rec->finals.begin() + rec_finals0, rec->finals.end());
            rec->consts.resize(rec_consts0);
            rec->finals.resize(rec_finals0);
        }
        prog.size = start_words;
        eb.data = prog.data; eb.size = prog.size; eb.cap = prog.cap;
        es = es0;
        int e2 = emit_begin(en, ES, rq.nq, rq.qvars, rq.ne, rq.evars, rq.ecodes, rq.no_prune);
        if (!e2) e2 = emit_run(en, ES, eb, es, rec, rq.nq, rq.qvars, rq.out_off, second.data(), (int)second.size() MIBN_PROF_PASS);
        if (e2 || !(es.alg_bytes - es0.alg_bytes < es_first.alg_bytes - es0.alg_bytes)) {
            prog.size = start_words;
            std::memcpy(prog.extend(keep.size()), keep.data(), keep.size() * sizeof(uint32_t));
            es = es_first;
            if (rec) {
                rec->consts.resize(rec_consts0);
                rec->finals.resize(rec_finals0);
                rec->consts.insert(rec->consts.end(), rec_first.consts.begin(), rec_first.consts.end());
                rec->finals.insert(rec->finals.end(), rec_first.finals.begin(), rec_first.finals.end());
            }
        } else if (st.order) {
            *st.order = second;
        }
    }
    st.alg_bytes = es.alg_bytes; st.alg_flops = es.alg_flops; st.n_steps = es.n_steps; st.max_step_cells = es.max_step_cells;
    st.arena_cells = es.arena_cells; st.out_cells = es.out_cells;
    return emit_error_message(e);
}

// ------------------------------------------------------------------------------------ cost estimate
// Byte model of the cheaper sweep order of one request (no emission, no min-fill search): what mibn_estimate_costs
// reports for shard balancing.
static double sweep_cost(const Network &net, const Request &rq) {
    if (net.n_vars <= 128) {
        OrderScratch &OS = order_scratch();
        const OrderNet on = net.order_view();
        B2 rel, hidden;
        order_prepare(on, OS, rq.nq, rq.qvars, rq.ne, rq.evars, rq.no_prune, rel, hidden);
        int qdepth = std::numeric_limits<int>::max();
        for (int i = 0; i < rq.nq; ++i) qdepth = std::min(qdepth, (int)net.depth[rq.qvars[i]]);
        order_sweep(on, OS, hidden, qdepth, 0);
        double best = order_simulate(on, OS, OS.cand, OS.n_cand, std::numeric_limits<double>::infinity());
        order_sweep(on, OS, hidden, qdepth, 1);
        return std::min(best, order_simulate(on, OS, OS.cand, OS.n_cand, best));
    }
    Scratch &S = scratch();
    Bits rel, qb, eb;
    rel.nw = qb.nw = eb.nw = net.nw;
    for (int i = 0; i < rq.nq; ++i) { qb.set(rq.qvars[i]); rel.set(rq.qvars[i]); rel.or_(net.anc[rq.qvars[i]]); }
    for (int i = 0; i < rq.ne; ++i) { eb.set(rq.evars[i]); rel.set(rq.evars[i]); rel.or_(net.anc[rq.evars[i]]); }
    if (!net.prune || rq.no_prune)
        for (int v = 0; v < net.n_vars; ++v) rel.set(v);
    Bits hidden = rel;
    hidden.andnot(qb);
    hidden.andnot(eb);
    std::vector<Bits> &scopes = S.scopes;
    std::vector<double> &scells = S.scope_cells;
    scopes.clear();
    scells.clear();
    rel.for_each([&](int v) {
        Bits sc;
        sc.nw = net.nw;
        double cells = 1;
        for (int u : net.scope[v])
            if (!eb.test(u) && net.card[u] > 1) { sc.set(u); cells *= net.card[u]; }
        scopes.push_back(sc);
        scells.push_back(cells);
    });
    Bits live = hidden;
    hidden.for_each([&](int v) { if (net.card[v] <= 1) live.clr(v); });
    int qdepth = std::numeric_limits<int>::max();
    for (int i = 0; i < rq.nq; ++i) qdepth = std::min(qdepth, (int)net.depth[rq.qvars[i]]);
    std::vector<int32_t> &o = S.hid;
    o.clear();
    for (int32_t v : net.topo_asc)
        if (live.test(v) && net.depth[v] < qdepth) o.push_back(v);
    for (int32_t v : net.topo_desc)
        if (live.test(v) && net.depth[v] >= qdepth) o.push_back(v);
    double best = simulate(net, scopes, scells, o, std::numeric_limits<double>::infinity());
    o.clear();
    for (int32_t v : net.topo_desc)
        if (live.test(v)) o.push_back(v);
    best = std::min(best, simulate(net, scopes, scells, o, best));
    return best;
}

void estimate_costs(const Network &net, ThreadPool &pool, int64_t B, const int64_t *q_off, const int32_t *q_vars,
                    const int64_t *e_off, const int32_t *e_vars, double *cost) {
    std::atomic<int64_t> next{0};
    constexpr int64_t kBlock = 256;
    pool.run([&](int) {
        for (;;) {
            const int64_t lo = next.fetch_add(kBlock, std::memory_order_relaxed);
            if (lo >= B) break;
            for (int64_t b = lo, hi = std::min(B, lo + kBlock); b < hi; ++b) {
                Request rq;
                rq.nq = (int32_t)(q_off[b + 1] - q_off[b]);
                rq.qvars = q_vars + q_off[b];
                rq.ne = (int32_t)(e_off[b + 1] - e_off[b]);
                rq.evars = e_vars + e_off[b];
                cost[b] = request_is_valid(net, rq) ? sweep_cost(net, rq) : 0.0;
            }
        }
    });
}

// ------------------------------------------------------------------------------------ buffers / threads

uint32_t *ProgBuf::extend(size_t words) {
    if (size + words > cap) {
        size_t ncap = std::max<size_t>(cap * 2, size + words + 4096);
        if (grow) {
            data = grow(ctx, data, size, ncap);
        } else {
            data = (uint32_t *)std::realloc(data, ncap * sizeof(uint32_t));
        }
        cap = ncap;
    }
    uint32_t *p = data + size;
    size += words;
    return p;
}

void ProgBuf::release() {
    if (!grow) std::free(data);
    data = nullptr;
    size = cap = 0;
}

std::string plan_request(const Network &net, const Request &rq, ProgBuf &prog, PlanStats &st) {
    return plan_request_rec(net, rq, prog, st, nullptr);
}

std::string plan_request(const Network &net, const Request &rq, std::vector<uint32_t> &prog, PlanStats &st) {
    ProgBuf b;
    std::string e = plan_request(net, rq, b, st);
    prog.insert(prog.end(), b.data, b.data + b.size);
    b.release();
    return e;
}

struct ThreadPool::Impl {
    std::vector<std::thread> th;
    std::mutex m;
    std::condition_variable cv_go, cv_done;
    const std::function<void(int)> *job = nullptr;
    uint64_t generation = 0;
    int pending = 0;
    bool stop = false;
};

ThreadPool::ThreadPool(int n) : impl_(new Impl), n_(std::max(1, n)) {
    for (int t = 1; t < n_; ++t)  // worker 0 is the calling thread
        impl_->th.emplace_back([this, t] {
            uint64_t seen = 0;
            for (;;) {
                const std::function<void(int)> *job;
                {
                    std::unique_lock<std::mutex> lk(impl_->m);
                    impl_->cv_go.wait(lk, [&] { return impl_->stop || impl_->generation != seen; });
                    if (impl_->stop) return;
                    seen = impl_->generation;
                    job = impl_->job;
                }
                (*job)(t);
                {
                    std::lock_guard<std::mutex> lk(impl_->m);
                    if (--impl_->pending == 0) impl_->cv_done.notify_one();
                }
            }
        });
}

ThreadPool::~ThreadPool() {
    {
        std::lock_guard<std::mutex> lk(impl_->m);
        impl_->stop = true;
    }
    impl_->cv_go.notify_all();
    for (auto &t : impl_->th) t.join();
    delete impl_;
}

void ThreadPool::run(const std::function<void(int)> &job) {
    if (n_ > 1) {
        std::lock_guard<std::mutex> lk(impl_->m);
        impl_->job = &job;
        impl_->pending = n_ - 1;
        ++impl_->generation;
    }
    impl_->cv_go.notify_all();
    job(0);
    if (n_ > 1) {
        std::unique_lock<std::mutex> lk(impl_->m);
        impl_->cv_done.wait(lk, [&] { return impl_->pending == 0; });
    }
}

// Cut one request's program into work items (tag_program, emit_core.h); appends to `out`, returns the number of items.
static uint32_t tag_request(const EmitNet &en, const uint32_t *prog, std::vector<Tag> &out) {
    return tag_program(en, prog, [&](const Tag &t) { out.push_back(t); });
}

// ------------------------------------------------------------------------------------ plan templates
// The program of a request depends on its *shape* - the query variables (in order) and the evidence variables (in
// order) - and, through the offsets of the evidence-sliced CPTs and the result offset of its final step only, on the
// evidence codes and the request's position.  Workloads repeat shapes (BayesNet.predict_proba: one shape for every row;
// the 100 k Asia requests of BASELINE config 2: 504 shapes), so every planning worker keeps the programs it planned as
// templates and instantiates a repeated shape by copying and patching.  Random shapes (the C3 stream: 3.8e8 of them)
// never repeat: the cache probes the first requests of every window and stays off for the rest when they miss.
namespace {

struct PlanTemplate {
    std::vector<uint32_t> words;
    struct Patch { uint32_t pos; uint32_t first, count; uint64_t base; };  // off = base + sum stride[k] * code[slot[k]]
    std::vector<Patch> patches;
    std::vector<uint32_t> slot;
    std::vector<int64_t> stride;
    std::vector<uint32_t> finals;
    std::vector<Tag> tags;
    PlanStats st;
};

// Shared by the planning workers of one network: 64 shards, each a mutex + a map; templates are immutable once
// published and are only freed at the start of a plan_batch call (no worker is running then), so a worker holds a
// shard's lock for the lookup only and copies from the template without it.
struct TemplateStore {
    struct Shard {
        std::mutex m;
        std::unordered_map<std::string, std::unique_ptr<PlanTemplate>> map;
    };
    static constexpr int kShards = 64;
    static constexpr size_t kMaxWords = 64u << 20;  // 256 MB of templates per network: cleared at the next batch
    Shard shard[kShards];
    std::atomic<size_t> words{0};
    uint64_t version = 0, opt_sig = 0;
    void clear() {
        for (auto &sh : shard) sh.map.clear();
        words = 0;
    }
};

// per planning worker: the probe state (is the stream repeating shapes?) and scratch
struct PlanCache {
    const TemplateStore *store = nullptr;
    uint64_t version = 0;
    uint64_t seen = 0, probe_hits = 0;
    bool on = true;
    // The first kProbe requests of every window are looked up (and recorded); a quarter of them hitting keeps the cache on for the
    // rest of the window.  Round 5: the window of a worker whose probes fail doubles from kMinWindow to kMaxWindow (a stream that never
    // repeats a shape - C3 - ends up probing 0.4 % of its requests, 1.6 % before) and falls back to kMinWindow with the first window
    // in which at least a quarter of the probes hit (a sparse-hit stream keeps doubling: its templates would not pay) - a stream of few shapes whose store was empty at the first probe (the n_evidence = 1 variant: 9 900 shapes, thirty-two
    // workers of 1 600 requests per call) used to wait 32 768 requests PER WORKER - twenty calls - for its second chance.
    static constexpr uint64_t kMinWindow = 1024, kMaxWindow = 65536, kProbe = 256;
    uint64_t window = kMinWindow;
    PlanRecord rec;
    std::string key;
};

uint64_t option_signature(const Network &net) {
    uint64_t h = 1469598103934665603ull;
    auto mix = [&](uint64_t v) { h = (h ^ v) * 1099511628211ull; };
    mix((uint64_t)net.small_cells); mix((uint64_t)net.big_iters); mix((uint64_t)net.tile_h); mix((uint64_t)net.fuse);
    mix((uint64_t)net.chain); mix((uint64_t)net.sweep); mix((uint64_t)net.sweep_iters); mix((uint64_t)net.sweep_min); mix((uint64_t)net.sweep_adapt); mix((uint64_t)net.order_weights); mix((uint64_t)net.order_effort); mix((uint64_t)net.second_above); mix((uint64_t)net.hint_sorted.size()); mix((uint64_t)net.sweep_canon); mix((uint64_t)net.outer); mix((uint64_t)net.prune); mix((uint64_t)net.minfill_above);
    mix((uint64_t)net.hints.size()); mix((uint64_t)net.tile_bytes);
    return h;
}

// called by plan_batch before its workers start
TemplateStore *template_store(const Network &net) {
    if (!net.templates) net.templates = std::make_shared<TemplateStore>();
    TemplateStore *ts = static_cast<TemplateStore *>(net.templates.get());
    const uint64_t sig = option_signature(net);
    if (ts->version != net.version || ts->opt_sig != sig || ts->words.load() > TemplateStore::kMaxWords) {
        ts->clear();
        ts->version = net.version;
        ts->opt_sig = sig;
    }
    return ts;
}

PlanCache &plan_cache(const TemplateStore *ts) {
    static thread_local PlanCache c;
    if (c.store != ts || c.version != ts->version) {
        c.store = ts;
        c.version = ts->version;
        c.seen = c.probe_hits = 0;
        c.on = true;
        c.window = PlanCache::kMinWindow;  // (another network / option set starts from the short window again)
    }
    return c;
}

}  // namespace

void plan_batch(const Network &net, ThreadPool &pool, std::vector<ProgBuf> &bufs, int64_t b0, int64_t b1,
                const int64_t *q_off, const int32_t *q_vars, const int64_t *e_off, const int32_t *e_vars,
                const int32_t *e_codes, const int64_t *out_off, const char *skip, BatchPlan &ck, bool no_prune,
                const uint8_t *orders, const int32_t *order_len, int64_t out_first) {
    const int64_t n = b1 - b0;
    const int T = pool.size();
    if ((int)bufs.size() < T) bufs.resize(T);
    ck.st = PlanStats{};
    ck.arena_cells = 0;
    ck.err.clear();
    ck.prog_off.assign(n, 0);
    ck.cost.assign(n, 0.0);
    ck.arena_need.assign(n, 0);
    ck.thread_of.assign(n, 0);
    ck.local_off.assign(n, 0);
    ck.thread_words.assign(T, 0);
    ck.tag_first.assign(n, 0);
    ck.tag_count.assign(n, 0);
    if ((int)ck.tags.size() < T) ck.tags.resize(T);
    std::vector<PlanStats> tst(T);
    std::vector<std::string> terr(T);
    // dynamic distribution in blocks of 32 requests: request costs vary 100x and a worker may lose its core to
    // another rank's planner, a static split would wait for the slowest worker
    TemplateStore *store = net.plan_cache ? template_store(net) : nullptr;
    std::atomic<int64_t> next{0};
    constexpr int64_t kBlock = 32;
    const EmitNet en = net.emit_view();
    pool.run([&](int t) {
        ProgBuf &prog = bufs[t];
        prog.size = 0;
        std::vector<Tag> &tags = ck.tags[t];
        tags.clear();
        for (;;) {
          const int64_t lo = next.fetch_add(kBlock, std::memory_order_relaxed);
          if (lo >= n) break;
          const int64_t hi = std::min(n, lo + kBlock);
          for (int64_t i = lo; i < hi; ++i) {
            const int64_t b = b0 + i;
            ck.prog_off[i] = prog.size;
            ck.local_off[i] = prog.size;
            ck.thread_of[i] = t;
            ck.tag_first[i] = (uint32_t)tags.size();
            if (skip && skip[b]) { prog.push(0); continue; }  // zero steps: result stays all-zero
            Request rq;
            rq.nq = (int32_t)(q_off[b + 1] - q_off[b]);
            rq.qvars = q_vars + q_off[b];
            rq.ne = (int32_t)(e_off[b + 1] - e_off[b]);
            rq.evars = e_vars + e_off[b];
            rq.ecodes = e_codes + e_off[b];
            rq.out_off = out_off[b] - out_off[out_first >= 0 ? out_first : b0];
            rq.no_prune = no_prune;
            if (orders) { rq.order = orders + (size_t)i * 128; rq.n_order = order_len[i]; }
            PlanStats st;
            // plan templates (see above): probe at the start of every window, stay on while shapes repeat
            PlanCache *pc = store ? &plan_cache(store) : nullptr;
            bool use_cache = false;
            if (pc) {
                const uint64_t w = pc->seen++;
                if (w == 0) { pc->probe_hits = 0; pc->on = true; }
                if (w == PlanCache::kProbe) {
                    pc->on = pc->probe_hits * 4 >= PlanCache::kProbe;
                    pc->window = pc->on ? PlanCache::kMinWindow : std::min(pc->window * 2, PlanCache::kMaxWindow);
                }
                if (pc->seen >= pc->window) pc->seen = 0;
                use_cache = pc->on;
            }
            if (use_cache) {
                std::string &key = pc->key;
                key.assign(reinterpret_cast<const char *>(&rq.nq), sizeof(rq.nq));
                key.push_back(no_prune ? 'N' : 'P');
                key.append(reinterpret_cast<const char *>(rq.qvars), sizeof(int32_t) * (size_t)rq.nq);
                key.append(reinterpret_cast<const char *>(rq.evars), sizeof(int32_t) * (size_t)rq.ne);
                TemplateStore::Shard &sh = store->shard[std::hash<std::string>{}(key) % TemplateStore::kShards];
                const PlanTemplate *hit = nullptr;
                {
                    std::lock_guard<std::mutex> lk(sh.m);
                    auto it = sh.map.find(key);
                    if (it != sh.map.end()) hit = it->second.get();
                }
                if (hit) {
                    const PlanTemplate &tp = *hit;
                    ++pc->probe_hits;
                    uint32_t *w = prog.extend(tp.words.size());
                    std::copy(tp.words.begin(), tp.words.end(), w);
                    for (const auto &pt : tp.patches) {
                        uint64_t off = pt.base;
                        for (uint32_t k = pt.first; k < pt.first + pt.count; ++k) off += (uint64_t)(tp.stride[k] * rq.ecodes[tp.slot[k]]);
                        off |= kConstFlag;
                        w[pt.pos] = (uint32_t)(off & 0xffffffffu);
                        w[pt.pos + 1] = (uint32_t)(off >> 32);
                    }
                    for (uint32_t fp : tp.finals) {
                        w[fp] = (uint32_t)((uint64_t)rq.out_off & 0xffffffffu);
                        w[fp + 1] = (uint32_t)((uint64_t)rq.out_off >> 32);
                    }
                    st = tp.st;
                    tags.insert(tags.end(), tp.tags.begin(), tp.tags.end());
                    ck.tag_count[i] = (uint32_t)tp.tags.size();
                } else {
                    pc->rec.consts.clear();
                    pc->rec.finals.clear();
                    std::string e = plan_request_rec(net, rq, prog, st, &pc->rec);
                    if (!e.empty()) { terr[t] = e; return; }
                    ck.tag_count[i] = tag_request(en, prog.data + ck.local_off[i], tags);
                    if (store->words.load(std::memory_order_relaxed) <= TemplateStore::kMaxWords) {
                        std::unique_ptr<PlanTemplate> up(new PlanTemplate);
                        PlanTemplate &tp = *up;
                        const size_t start = (size_t)ck.local_off[i];
                        tp.words.assign(prog.data + start, prog.data + prog.size);
                        for (const auto &cs : pc->rec.consts) {
                            const int v = cs.second;
                            PlanTemplate::Patch pt{(uint32_t)(cs.first - start), (uint32_t)tp.slot.size(), 0u, (uint64_t)net.pool_off[v]};
                            for (size_t k = 0; k < net.scope[v].size(); ++k)
                                for (int sl = 0; sl < rq.ne; ++sl)
                                    if (rq.evars[sl] == net.scope[v][k]) { tp.slot.push_back((uint32_t)sl); tp.stride.push_back(net.cstride[v][k]); ++pt.count; }
                            if (pt.count) tp.patches.push_back(pt);
                        }
                        for (uint32_t fp : pc->rec.finals) tp.finals.push_back((uint32_t)(fp - start));
                        tp.tags.assign(tags.begin() + ck.tag_first[i], tags.end());
                        tp.st = st;
                        store->words.fetch_add(tp.words.size(), std::memory_order_relaxed);
                        std::lock_guard<std::mutex> lk(sh.m);
                        sh.map.emplace(key, std::move(up));  // (another worker may have published the shape meanwhile: kept)
                    }
                }
            } else {
                std::string e = plan_request(net, rq, prog, st);
                if (!e.empty()) { terr[t] = e; return; }
                ck.tag_count[i] = tag_request(en, prog.data + ck.local_off[i], tags);
            }
            ck.cost[i] = st.alg_bytes;
            ck.arena_need[i] = st.arena_cells;
            tst[t].alg_bytes += st.alg_bytes;
            tst[t].alg_flops += st.alg_flops;
            tst[t].n_steps += st.n_steps;
            tst[t].max_step_cells = std::max(tst[t].max_step_cells, st.max_step_cells);
            tst[t].arena_cells = std::max(tst[t].arena_cells, st.arena_cells);
          }
        }
        ck.thread_words[t] = prog.size;
    });
    std::vector<size_t> tbase(T, 0);
    size_t base = 0;
    for (int t = 0; t < T; ++t) {
        tbase[t] = base;
        base += ck.thread_words[t];
    }
    for (int64_t i = 0; i < n; ++i) ck.prog_off[i] = tbase[ck.thread_of[i]] + ck.local_off[i];
    for (int t = 0; t < T; ++t) {
        if (!terr[t].empty()) ck.err = terr[t];
        ck.st.alg_bytes += tst[t].alg_bytes;
        ck.st.alg_flops += tst[t].alg_flops;
        ck.st.n_steps += tst[t].n_steps;
        ck.st.max_step_cells = std::max(ck.st.max_step_cells, tst[t].max_step_cells);
        ck.arena_cells = std::max(ck.arena_cells, tst[t].arena_cells);
    }
    ck.total_words = base;
}

// ------------------------------------------------------------------------------------ schedule

const char *kernel_name(int kid) {
    static std::string names[kNumKernels];
    static bool init = false;
    if (!init) {
        names[kKidSeg] = "segments";
        static const char *cxn[3] = {"cx4", "cx16", "cxN"}, *ncn[6] = {"nc1", "nc4", "nc16", "ncN", "nc16-mfma", "outer-mfma"};
        for (int nb = 1; nb <= 2; ++nb)
            for (int c = 0; c < 3; ++c)
                for (int n = 0; n < 6; ++n)
                    names[kKidFiber0 + (nb - 1) * 18 + c * 6 + n] =
                        "fiber<" + std::to_string(nb) + "," + cxn[c] + "," + ncn[n] + ">";
        names[kKidChain] = "fiber<1,cx64,chain-mfma>";  // (the slot of the impossible class <2,cxN,outer-mfma>)
        for (int j = 0; j < kMaxIn; ++j) names[kKidGeneric0 + j] = "generic<" + std::to_string(j + 1) + ">";
        names[kKidSweep] = "ve_sweep_kernel";
        init = true;
    }
    return kid >= 0 && kid < kNumKernels ? names[kid].c_str() : "?";
}

// Order of the classes of work inside a level's launch (workgroups are dispatched in this order): the classes whose
// workgroups run long and move little - segments, GENERIC tiles, the two-table VALU forms: 80-170 us per workgroup -
// go first, so that they finish under the streaming classes instead of forming the tail of every level.
struct ClassOrder {
    int rank_of[kNumKernels], kid_at[kNumKernels];
    ClassOrder() {
        std::vector<int> order;
        auto add = [&](int kid) { if (std::find(order.begin(), order.end(), kid) == order.end()) order.push_back(kid); };
        add(kKidSeg);
        for (int j = 0; j < kMaxIn; ++j) add(kKidGeneric0 + j);
        for (int n : {3, 1, 2, 0})  // two tables, VALU: ncN, nc4, nc16, nc1
            for (int c = 2; c >= 0; --c) add(kKidFiber0 + 18 + c * 6 + n);
        for (int c = 2; c >= 0; --c) add(kKidFiber0 + c * 6 + 3);                          // one table, ncN
        for (int n = 0; n < 6; ++n) add(kKidFiber0 + 2 * 6 + n);                            // one table, runtime cx
        for (int c = 0; c < 2; ++c) add(kKidFiber0 + 18 + c * 6 + 5);                       // OUTER
        for (int kid = 0; kid < kNumKernels; ++kid)                                        // the streaming classes ...
            if (kid != kKidFiber0 + 4 && kid != kKidFiber0 + 6 + 4 && kid != kKidSweep) add(kid);
        add(kKidFiber0 + 4);      // ... the one-table fp64-MFMA pair classes last among them (contiguous: engine option mfma_kernel
        add(kKidFiber0 + 6 + 4);  //     launches the two as ve_mfma_kernel), then the sweep class (a kernel of its own)
        add(kKidSweep);
        for (int r = 0; r < kNumKernels; ++r) { kid_at[r] = order[(size_t)r]; rank_of[order[(size_t)r]] = r; }
    }
};
static const ClassOrder kClassOrder;

void build_schedule(const Network &net, const BatchPlan &bp, const std::vector<ProgBuf> &bufs, int64_t r0, int64_t r1,
                    Schedule &out) {
    (void)bufs;
    const int64_t n = r1 - r0;
    out.items.clear();
    out.wg_item.clear();
    out.launches.clear();
    out.arena_off.assign(n, 0);
    int64_t top = 0;
    for (int64_t i = 0; i < n; ++i) {
        out.arena_off[i] = (uint64_t)top;
        top += (bp.arena_need[r0 + i] + 15) & ~int64_t(15);  // 128-byte aligned private arenas
    }
    out.arena_cells = top;
    // pass 1: bucket sizes per (level, class of work) from the tags the planning workers left behind
    int n_levels = 0;
    size_t n_tags = 0, n_wg = 0;
    for (int64_t i = 0; i < n; ++i) {
        const int64_t r = r0 + i;
        const Tag *tg = bp.tags[bp.thread_of[r]].data() + bp.tag_first[r];
        const uint32_t cnt = bp.tag_count[r];
        if (cnt) n_levels = std::max(n_levels, (int)tg[cnt - 1].level + 1);
        n_tags += cnt;
    }
    // Staggered levels (Network::stagger = G > 1): the requests of a chunk are dealt into G groups and group g starts
    // g * (levels / G) launches late, so that every launch mixes the phases of a request's program - the latency-bound
    // chains of tiny steps at its start, the frontier sweeps that stream at the HBM rate in the middle, the write-
    // dominated joins where two sweeps meet - instead of running each phase for all requests at once.  Dependencies are
    // per request (item k of a request still runs one launch after item k - 1), arenas are private: nothing else changes.
    const int G = std::max(1, std::min(net.stagger, 8));
    const int delta = G > 1 ? (n_levels + G - 1) / G : 0;
    auto shift = [&](int64_t i) { return (int)(i % G) * delta; };
    n_levels += (G - 1) * delta;
    out.n_levels = n_levels;
    const size_t nb = (size_t)n_levels * kNumKernels;
    std::vector<size_t> count(nb + 1, 0);
    std::vector<double> bytes(nb, 0.0);
    std::vector<uint64_t> sweep_tiles((size_t)n_levels, 0);
    for (int64_t i = 0; i < n; ++i) {
        const int64_t r = r0 + i;
        const Tag *tg = bp.tags[bp.thread_of[r]].data() + bp.tag_first[r];
        const int sh = shift(i);
        for (uint32_t k = 0; k < bp.tag_count[r]; ++k) {
            const size_t bkt = (size_t)(tg[k].level + sh) * kNumKernels + kClassOrder.rank_of[tg[k].kid];
            ++count[bkt + 1];
            bytes[bkt] += tg[k].bytes;
            if (tg[k].kid == kKidSweep) sweep_tiles[(size_t)(tg[k].level + sh)] += tg[k].a;
            else n_wg += tg[k].wgs;
        }
    }
    // Tiles per workgroup of a level's sweep launch: net.sweep_iters where the launch is big enough to keep every CU busy
    // for several workgroup lifetimes (one lives ~ 20 us per tile, 512 are resident), fewer - down to 2: the next tile's
    // loads fly under the current one's stores - where it is not, so that the tail of the launch stays short.
    std::vector<uint32_t> sweep_iters((size_t)n_levels, (uint32_t)std::max(1, net.sweep_iters));
    if (net.sweep_adapt)
        for (int l = 0; l < n_levels; ++l)
            while (sweep_iters[(size_t)l] > 2 && sweep_tiles[(size_t)l] / sweep_iters[(size_t)l] < (uint64_t)net.sweep_adapt) sweep_iters[(size_t)l] /= 2;
    for (int64_t i = 0; i < n; ++i) {
        const int64_t r = r0 + i;
        const Tag *tg = bp.tags[bp.thread_of[r]].data() + bp.tag_first[r];
        const int sh = shift(i);
        for (uint32_t k = 0; k < bp.tag_count[r]; ++k)
            if (tg[k].kid == kKidSweep) { const uint32_t it = sweep_iters[(size_t)(tg[k].level + sh)]; n_wg += (tg[k].a + it - 1) / it; }
    }
    for (size_t k = 0; k < nb; ++k) count[k + 1] += count[k];
    // pass 2: scatter (Item::b = workgroups of the item for now)
    out.items.resize(n_tags);
    std::vector<size_t> cur(count.begin(), count.end() - 1);
    for (int64_t i = 0; i < n; ++i) {
        const int64_t r = r0 + i;
        const Tag *tg = bp.tags[bp.thread_of[r]].data() + bp.tag_first[r];
        const int sh = shift(i);
        for (uint32_t k = 0; k < bp.tag_count[r]; ++k) {
            uint32_t a = tg[k].a, wgs = tg[k].wgs;
            if (tg[k].kid == kKidSweep) { const uint32_t it = sweep_iters[(size_t)(tg[k].level + sh)]; wgs = (a + it - 1) / it; a = it; }
            out.items[cur[(size_t)(tg[k].level + sh) * kNumKernels + kClassOrder.rank_of[tg[k].kid]]++] = Item{(uint32_t)i, tg[k].rel_off, a, wgs};
        }
    }
    // pass 3: workgroup -> item table, level by level
    out.wg_item.resize(n_wg);
    size_t wg = 0, wg_level = 0;
    int cur_level = -1;
    for (size_t k = 0; k < nb; ++k)
        if (count[k + 1] > count[k]) {
            const int level = (int)(k / kNumKernels), kid = kClassOrder.kid_at[k % kNumKernels];
            if (level != cur_level) { cur_level = level; wg_level = wg; }
            if (kid == kKidSeg && count[k + 1] - count[k] > 1) {
                // longest segments first (they are the tail of their level): stable counting sort on the step count -
                // a comparison sort of the 100 k one-segment requests of a tiny network was the bulk of this function
                Item *first = out.items.data() + count[k];
                const size_t m = count[k + 1] - count[k];
                uint32_t maxa = 0;
                for (size_t q = 0; q < m; ++q) maxa = std::max(maxa, first[q].a & ~kItemSegment);
                if (maxa < (1u << 16)) {
                    std::vector<size_t> pos((size_t)maxa + 2, 0);
                    for (size_t q = 0; q < m; ++q) ++pos[(size_t)(maxa - (first[q].a & ~kItemSegment)) + 1];
                    for (size_t v = 0; v <= maxa; ++v) pos[v + 1] += pos[v];
                    std::vector<Item> tmp(first, first + m);
                    for (size_t q = 0; q < m; ++q) first[pos[(size_t)(maxa - (tmp[q].a & ~kItemSegment))]++] = tmp[q];
                } else {
                    std::stable_sort(first, first + m, [](const Item &a, const Item &b) { return a.a > b.a; });
                }
            }
            const size_t wg_first = wg;
            if (kid == kKidSeg) {
                // Segments are chains of dependent tiny steps - latency, not bandwidth - and a wave is enough for one (< 4 096
                // output cells per step): a workgroup runs FOUR, one per wave, with nothing shared but the launch.  Twelve
                // chains per CU in flight instead of three (the level kernel's 40 KB of LDS admit three workgroups per CU).
                for (size_t q = count[k]; q < count[k + 1]; q += kSegPerWg) {
                    const size_t m = std::min<size_t>(kSegPerWg, count[k + 1] - q);
                    for (size_t t = 0; t < m; ++t) out.items[q + t].b = t == 0 ? (uint32_t)m : 0u;
                    out.wg_item[wg++] = (uint32_t)q;
                }
            } else {
                for (size_t q = count[k]; q < count[k + 1]; ++q) {
                    const uint32_t wgs = out.items[q].b;
                    out.items[q].b = (uint32_t)(wg - wg_level);
                    std::fill(out.wg_item.begin() + wg, out.wg_item.begin() + wg + wgs, (uint32_t)q);
                    wg += wgs;
                }
            }
            out.launches.push_back({level, kid, count[k], count[k + 1] - count[k], wg_first, wg - wg_first, wg_level, bytes[k]});
        }
    out.wg_item.resize(wg);  // (sized for one workgroup per segment above)
}

}  // namespace mibn
