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
    hint_flat.clear();
    if (n <= 128 && err.empty()) {
        anc2.resize(n);
        scope2.resize(n);
        for (int v = 0; v < n; ++v) {
            anc2[v].a = anc[v].w[0];
            anc2[v].b = anc[v].w[1];
            for (int32_t u : scope[v]) scope2[v].set(u);
        }
    }
    if (err.empty()) set_hints(0, nullptr);  // the built-in sweep lists
    return err;
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
    o.topo_asc = topo_asc.data();
    o.topo_desc = topo_desc.data();
    o.hint_sorted = hint_flat.data();
    o.prune = prune;
    o.minfill_above = minfill_above;
    o.chain_weight = (fuse && sweep >= 4 && order_weights) ? (order_weights == 1 ? 0.25 : 1.0 / (double)order_weights) : 1.0;  // (option value k > 1: weight 1 / k)
    o.big_cells = (double)small_cells;
    return o;
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

constexpr int kRawAxes = 40;  // axes of one factor before merging (cells < 2^31 => <= 31 non-trivial axes)

struct PF {  // planning-time factor (plain data: no heap allocation on the planning path)
    Bits scope;                  // free (non-evidence) variables
    int n = 0;                   // axes
    int32_t vars[kRawAxes];
    int64_t strides[kRawAxes];   // stride (doubles) per axis
    uint64_t off = 0;            // arena offset, or pool offset | kConstFlag
    int64_t cells = 0;           // product of the free cardinalities
    int64_t alloc = 0;           // arena cells owned (0 for constants)
    int32_t src = -1;            // initial factor: the variable whose CPT it slices (its offset depends on the evidence codes)
    PF() {}                      // (user-provided: emplace_back() does not zero the 480 bytes of vars / strides)
};

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
    std::vector<PF> pool;
    std::vector<int> live;
    std::vector<int32_t> pos;  // variable -> axis position in the current output (or -1)
    std::vector<double> key;
    std::vector<uint64_t> mem;    // per variable: the factor slots (pool indices) whose scope contains it
    std::vector<uint64_t> slot_alive;  // factor slots not yet consumed
    std::vector<int32_t> cand, best;   // candidate elimination orders (no per-request heap traffic)
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

struct Arena {
    static constexpr int kMaxBlocks = 64;
    int64_t foff[kMaxBlocks], fsz[kMaxBlocks];  // free list sorted by offset
    int nf = 0;
    int64_t top = 0;
    int64_t alloc(int64_t n) {
        n = (n + 15) & ~int64_t(15);  // 128-byte aligned tables: a wave's 512-byte load or store touches exactly 4 cache lines
        for (int i = 0; i < nf; ++i)
            if (fsz[i] >= n) {
                const int64_t o = foff[i];
                foff[i] += n;
                fsz[i] -= n;
                if (!fsz[i]) { for (int k = i; k + 1 < nf; ++k) { foff[k] = foff[k + 1]; fsz[k] = fsz[k + 1]; } --nf; }
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
    void release(int64_t o, int64_t n) {
        n = (n + 15) & ~int64_t(15);
        int i = 0;
        while (i < nf && foff[i] < o) ++i;
        const bool left = i > 0 && foff[i - 1] + fsz[i - 1] == o;
        const bool right = i < nf && o + n == foff[i];
        if (left && right) {
            fsz[i - 1] += n + fsz[i];
            for (int k = i; k + 1 < nf; ++k) { foff[k] = foff[k + 1]; fsz[k] = fsz[k + 1]; }
            --nf;
        } else if (left) {
            fsz[i - 1] += n;
        } else if (right) {
            foff[i] = o;
            fsz[i] += n;
        } else if (nf < kMaxBlocks) {
            for (int k = nf; k > i; --k) { foff[k] = foff[k - 1]; fsz[k] = fsz[k - 1]; }
            foff[i] = o;
            fsz[i] = n;
            ++nf;
        }  // else: leak the block (only costs scratch space)
    }
};

// Where a request's program depends on its evidence *codes* and on its position in the batch - and nowhere else: the
// offsets of the evidence-sliced CPTs and the result offset of the final step (plan templates, plan_batch).
struct PlanRecord {
    std::vector<std::pair<uint32_t, int32_t>> consts;  // (word index of an (off lo, off hi) pair in the buffer, CPT variable)
    std::vector<uint32_t> finals;                       // word index of the (out_off lo, out_off hi) pair of a FINAL step
};

struct Emitter {
    const Network &net;
    ProgBuf &prog;
    PlanStats &st;
    Arena arena;
    std::vector<double> &key;   // layout key per variable: larger = lives longer = faster axis
    std::vector<int32_t> &pos;  // variable -> output axis (scratch, -1 outside emit)
    std::string err;
    PlanRecord *rec = nullptr;  // optional: where the program depends on evidence codes / batch position

    // the (off lo, off hi) pair of input table f
    void put_off(uint32_t *&p, const PF *f) {
        if (rec && f->src >= 0) rec->consts.emplace_back((uint32_t)(p - prog.data), f->src);
        *p++ = (uint32_t)(f->off & 0xffffffffu);
        *p++ = (uint32_t)(f->off >> 32);
    }

    void header(uint32_t *w, uint32_t kind, int n_in, int ma, int mlo, int cx, bool final_, int64_t lo, int64_t hi,
                uint64_t out_off, int words) {
        w[0] = kind | ((uint32_t)n_in << 8) | ((uint32_t)ma << 16) | ((uint32_t)mlo << 24);
        w[1] = (uint32_t)cx | ((final_ ? kFlagFinal : 0u) << 16);
        w[2] = (uint32_t)lo;
        w[3] = (uint32_t)hi;
        w[4] = (uint32_t)(out_off & 0xffffffffu);
        w[5] = (uint32_t)(out_off >> 32);
        if (rec && final_) rec->finals.push_back((uint32_t)(w - prog.data) + 4);
        w[6] = (uint32_t)words;
        w[7] = w[8] = w[9] = 0;
    }

    using Strides = int64_t[kMaxIn][kRawAxes];
    using XStrides = int64_t[kMaxIn][3];

    // GENERIC encoding: iteration space = output cells
    void emit_generic(const PF *const *ins, int n_in, const Strides &s, const int64_t *xs, const PF &out, int64_t cells,
                      int cx, bool final_) {
        const int na = out.n;
        int nlo = 0;
        int64_t lo = 1;
        const int64_t lomax = n_in <= 3 ? kLoMax : kLoTarget;  // kernel: 2 cells per lane up to 3 inputs, else 1
        while (nlo < na && lo < kLoTarget && lo * net.card[out.vars[nlo]] <= lomax) lo *= net.card[out.vars[nlo++]];
        // merge adjacent axes that are contiguous in every input (the output is dense by construction)
        uint32_t mcard[kRawAxes];
        int64_t ms[kMaxIn][kRawAxes];
        int ma = 0, mlo = 0;
        for (int a = 0; a < na; ++a) {
            bool merge = ma > 0 && a != nlo;
            for (int j = 0; j < n_in && merge; ++j) merge = s[j][a] == ms[j][ma - 1] * (int64_t)mcard[ma - 1];
            const uint32_t c = (uint32_t)net.card[out.vars[a]];
            if (merge && (uint64_t)mcard[ma - 1] * c < (1u << 30)) {
                mcard[ma - 1] *= c;
            } else {
                mcard[ma] = c;
                for (int j = 0; j < n_in; ++j) ms[j][ma] = s[j][a];
                ++ma;
                if (a < nlo) ++mlo;
            }
        }
        if (ma > kMaxAxes) { err = "a step has more than " + std::to_string(kMaxAxes) + " axes"; return; }
        const int words = kHdrWords + 3 * n_in + ma + n_in * ma;
        uint32_t *w = prog.extend(words);
        header(w, kKindGeneric, n_in, ma, mlo, cx, final_, lo, cells / lo, out.off, words);
        uint32_t *p = w + kHdrWords;
        for (int j = 0; j < n_in; ++j) {
            put_off(p, ins[j]);
            *p++ = (uint32_t)(int32_t)xs[j];
        }
        for (int a = 0; a < ma; ++a) *p++ = mcard[a];
        for (int j = 0; j < n_in; ++j)
            for (int a = 0; a < ma; ++a) *p++ = (uint32_t)(int32_t)ms[j][a];
    }

    // FIBER encoding (see planner.h); returns false when the step does not fit the form
    bool emit_fiber(const PF *const *ins, int n_in, const Strides &s, const XStrides &xs, const PF &out, int cx, int c1) {
        const int na = out.n;
        int big[kMaxIn], small[kMaxIn], nb = 0, ns = 0;
        for (int j = 0; j < n_in; ++j) {
            if (ins[j]->cells > net.small_cells) big[nb++] = j;
            else small[ns++] = j;
        }
        if (nb < 1 || nb > 2 || ns > kMaxSmall || cx > kMaxCx) return false;
        // N axes: no big input depends on them; keep at most kMaxNC combinations (fastest axes first)
        int naxes[kRawAxes], raxes[kRawAxes], nN = 0, nr = 0;
        int64_t NC = 1;
        for (int a = 0; a < na; ++a) {
            bool free_ = true;
            for (int b = 0; b < nb; ++b) free_ = free_ && s[big[b]][a] == 0;
            const int c = net.card[out.vars[a]];
            if (free_ && NC * c <= kMaxNC && nN < 15) { naxes[nN++] = a; NC *= c; }
            else raxes[nr++] = a;
        }
        // R-axis tables (unmerged); ctrl axes = R axes a small input depends on
        int64_t rcard[kRawAxes], rost[kRawAxes], rtst[kRawAxes], rb[2][kRawAxes];
        int ctrl[kRawAxes], nctrl = 0;
        int64_t T = NC * cx;
        for (int i = 0; i < nr; ++i) {
            const int a = raxes[i];
            rcard[i] = net.card[out.vars[a]];
            rost[i] = out.strides[a];
            for (int b = 0; b < nb; ++b) rb[b][i] = s[big[b]][a];
            bool dep = false;
            for (int k = 0; k < ns; ++k) dep = dep || s[small[k]][a] != 0;
            rtst[i] = 0;
            if (dep) {
                if (nctrl >= 15) return false;
                ctrl[nctrl++] = a;
                rtst[i] = T;
                T *= rcard[i];
                if (T > kMaxT) return false;
            }
        }
        const int nT = nN + nctrl;
        int nlo = 0;
        int64_t lo = 1;
        while (nlo < nr && lo < kLoTarget && lo * rcard[nlo] <= kFiberLoMax) lo *= rcard[nlo++];
        int64_t rcells = 1;
        for (int i = 0; i < nr; ++i) rcells *= rcard[i];
        if (rcells * NC < net.big_iters) return false;  // small steps (< big_iters output cells) run in the segment interpreter (GENERIC form)
        // contiguous fibers: N-combination n at offset n, lane cell l at l*NC
        bool contig = true;
        {
            int64_t expect = NC;
            for (int i = 0; i < nlo; ++i) { contig = contig && rost[i] == expect; expect *= rcard[i]; }
        }
        if (nb == 2) {
            // two tables: the wave-uniform (hi) axes only one of them depends on run fastest, the other table's values
            // stay in the kernel's registers over those iterations (fiber_call<2, ...>)
            int64_t tc[kRawAxes], to[kRawAxes], tt[kRawAxes], tb0[kRawAxes], tb1[kRawAxes];
            int k = 0;
            for (int pass = 0; pass < 2; ++pass)
                for (int i = nlo; i < nr; ++i) {
                    const bool single = (rb[0][i] == 0) != (rb[1][i] == 0);
                    if (single == (pass == 0)) { tc[k] = rcard[i]; to[k] = rost[i]; tt[k] = rtst[i]; tb0[k] = rb[0][i]; tb1[k] = rb[1][i]; ++k; }
                }
            for (int i = 0; i < k; ++i) { rcard[nlo + i] = tc[i]; rost[nlo + i] = to[i]; rtst[nlo + i] = tt[i]; rb[0][nlo + i] = tb0[i]; rb[1][nlo + i] = tb1[i]; }
        }
        // merge adjacent R axes contiguous in the output, in T and in every big input
        int64_t mc[kRawAxes], mo[kRawAxes], mt[kRawAxes], mb[2][kRawAxes];
        int ma = 0, mlo = 0;
        for (int i = 0; i < nr; ++i) {
            bool merge = ma > 0 && i != nlo && mo[ma - 1] * mc[ma - 1] == rost[i] && mt[ma - 1] * mc[ma - 1] == rtst[i] &&
                         mc[ma - 1] * rcard[i] < (1 << 30);
            for (int b = 0; b < nb && merge; ++b) merge = mb[b][ma - 1] * mc[ma - 1] == rb[b][i];
            if (merge) {
                mc[ma - 1] *= rcard[i];
            } else {
                mc[ma] = rcard[i];
                mo[ma] = rost[i];
                mt[ma] = rtst[i];
                for (int b = 0; b < nb; ++b) mb[b][ma] = rb[b][i];
                ++ma;
                if (i < nlo) ++mlo;
            }
        }
        if (ma > kMaxAxes) return false;
        const int words = kHdrWords + 4 * nb + ns * (4 + nT) + nT + (int)NC + 3 * ma + nb * ma;
        if (words > kMaxStepWords) return false;
        uint32_t nout[kMaxNC];
        for (int64_t n = 0; n < NC; ++n) {
            int64_t r = n, off = 0;
            for (int i = 0; i < nN; ++i) {
                const int c = net.card[out.vars[naxes[i]]];
                off += (r % c) * out.strides[naxes[i]];
                r /= c;
            }
            nout[n] = (uint32_t)off;
            contig = contig && off == n;
        }
        uint32_t *w = prog.extend(words);
        header(w, kKindFiber, nb + ns, ma, mlo, cx, false, lo, rcells / lo, out.off, words);
        if (contig) w[1] |= kFlagContig << 16;
        {
            // row stride of the MFMA form: at most one ctrl axis inside a wave's 64 cells (4 states, cell stride 1/4/16);
            // ctrl axes further out must not change inside a wave (cell stride a multiple of 64)
            int row_stride = 16, inside = 0;
            bool ok = true;
            int64_t cs = 1;
            for (int i = 0; i < nlo; ++i) {
                if (rtst[i] != 0) {
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
            if (inside > 1 || !ok) row_stride = 0;
            w[1] |= (uint32_t)row_stride << kRowStrideShift;
        }
        w[7] = (uint32_t)nb | ((uint32_t)ns << 4) | ((uint32_t)nN << 8) | ((uint32_t)nctrl << 12) | ((uint32_t)NC << 16);
        w[8] = (uint32_t)T | ((uint32_t)c1 << 16);
        uint32_t *p = w + kHdrWords;
        for (int b = 0; b < nb; ++b) {
            put_off(p, ins[big[b]]);
            *p++ = (uint32_t)(int32_t)xs[big[b]][0];
            *p++ = (uint32_t)(int32_t)xs[big[b]][1];
        }
        for (int k = 0; k < ns; ++k) {
            const int j = small[k];
            put_off(p, ins[j]);
            *p++ = (uint32_t)(int32_t)xs[j][0];
            *p++ = (uint32_t)(int32_t)xs[j][1];
            for (int i = 0; i < nN; ++i) *p++ = (uint32_t)(int32_t)s[j][naxes[i]];
            for (int i = 0; i < nctrl; ++i) *p++ = (uint32_t)(int32_t)s[j][ctrl[i]];
        }
        for (int i = 0; i < nN; ++i) *p++ = (uint32_t)net.card[out.vars[naxes[i]]];
        for (int i = 0; i < nctrl; ++i) *p++ = (uint32_t)net.card[out.vars[ctrl[i]]];
        for (int64_t n = 0; n < NC; ++n) *p++ = nout[n];
        for (int a = 0; a < ma; ++a) { *p++ = (uint32_t)mc[a]; *p++ = (uint32_t)mo[a]; *p++ = (uint32_t)mt[a]; }
        for (int b = 0; b < nb; ++b)
            for (int a = 0; a < ma; ++a) *p++ = (uint32_t)(int32_t)mb[b][a];
        return true;
    }

    // OUTER encoding (see planner.h): two big inputs (+ CPT slices), the product is a batched dense
    // [cells of A] x [cx] x [16 cells only B spans] contraction for the fp64 MFMA.  Returns false when it does not fit.
    bool emit_outer(const PF *const *ins, int n_in, const Strides &s, const XStrides &xs, const PF &out, int cx, int c1) {
        if (!((cx == 4 && c1 == 4) || (cx == 16 && c1 == 4))) return false;
        int bigs[kMaxIn], small[kMaxIn], nbig = 0, ns = 0;
        for (int j = 0; j < n_in; ++j) {
            if (ins[j]->cells > net.small_cells) bigs[nbig++] = j;
            else small[ns++] = j;
        }
        if (nbig != 2 || ns > kMaxSmall) return false;
        const int na = out.n;
        // B = the big input that owns two 4-state output axes the other one does not depend on (its N axes); of the
        // two possible role assignments the feasible one with the faster N axes wins
        int A = -1, B = -1, nax[2] = {-1, -1};
        int64_t rcard[kRawAxes], rost[kRawAxes], rtst[kRawAxes], rb[2][kRawAxes];
        int raxis[kRawAxes], ctrl[kRawAxes];
        int nr = 0, nlo = 0, row_stride = 0, nctrl = 0;
        int64_t lo = 1, rcells = 1, T = 0;
        int64_t best_key = std::numeric_limits<int64_t>::max();
        for (int cand = 0; cand < 2; ++cand) {
            const int b = bigs[cand], a_ = bigs[1 - cand];
            int found[2], nf = 0;
            for (int ax = 0; ax < na && nf < 2; ++ax)
                if (s[b][ax] != 0 && s[a_][ax] == 0 && net.card[out.vars[ax]] == 4) found[nf++] = ax;
            if (nf < 2) continue;
            const int64_t key = out.strides[found[0]] + out.strides[found[1]];
            if (key >= best_key) continue;
            // R axes: everything but the two N axes; ctrl axes = R axes a small input depends on
            int64_t c_card[kRawAxes], c_ost[kRawAxes], c_tst[kRawAxes], c_rb[2][kRawAxes];
            int c_axis[kRawAxes], c_ctrl[kRawAxes];
            int c_nr = 0, c_nctrl = 0;
            int64_t c_T = ns ? 16 * (int64_t)cx : 0;
            bool ok = true;
            for (int ax = 0; ax < na && ok; ++ax) {
                if (ax == found[0] || ax == found[1]) continue;
                c_axis[c_nr] = ax;
                c_card[c_nr] = net.card[out.vars[ax]];
                c_ost[c_nr] = out.strides[ax];
                c_rb[0][c_nr] = s[a_][ax];
                c_rb[1][c_nr] = s[b][ax];
                bool dep = false;
                for (int k = 0; k < ns; ++k) dep = dep || s[small[k]][ax] != 0;
                c_tst[c_nr] = 0;
                if (dep) {
                    if (c_nctrl >= 13) ok = false;
                    c_ctrl[c_nctrl++] = ax;
                    c_tst[c_nr] = c_T;
                    c_T *= c_card[c_nr];
                    if (c_T > kMaxT) ok = false;
                }
                ++c_nr;
            }
            if (!ok) continue;
            int c_nlo = 0;
            int64_t c_lo = 1;
            while (c_nlo < c_nr && c_lo < kLoTarget && c_lo * c_card[c_nlo] <= kFiberLoMax) c_lo *= c_card[c_nlo++];
            // row stride: what the B operand (B and T) depends on inside a wave's 64 cells (same rule as emit_fiber)
            int c_rs = 16, inside = 0;
            int64_t cs = 1;
            for (int i = 0; i < c_nlo; ++i) {
                if (c_rb[1][i] != 0 || c_tst[i] != 0) {
                    if (cs < 64) {
                        ++inside;
                        if (c_card[i] == 4 && (cs == 1 || cs == 4 || cs == 16)) c_rs = (int)cs;
                        else ok = false;
                    } else if (cs % 64 != 0) {
                        ok = false;
                    }
                }
                cs *= c_card[i];
            }
            if (inside > 1 || !ok) continue;
            best_key = key;
            A = a_; B = b; nax[0] = found[0]; nax[1] = found[1];
            nr = c_nr; nlo = c_nlo; lo = c_lo; row_stride = c_rs; nctrl = c_nctrl; T = c_T;
            rcells = 1;
            for (int i = 0; i < c_nr; ++i) {
                raxis[i] = c_axis[i]; rcard[i] = c_card[i]; rost[i] = c_ost[i]; rtst[i] = c_tst[i];
                rb[0][i] = c_rb[0][i]; rb[1][i] = c_rb[1][i];
                rcells *= c_card[i];
            }
            for (int i = 0; i < c_nctrl; ++i) ctrl[i] = c_ctrl[i];
        }
        if (B < 0) return false;
        (void)raxis;
        const int big[2] = {A, B};
        if (rcells * 16 < net.big_iters) return false;  // (an R cell is 16 output cells here)
        uint32_t nout[16], nB[16];
        bool contig = true;
        for (int n = 0; n < 16; ++n) {
            nout[n] = (uint32_t)((n & 3) * out.strides[nax[0]] + (n >> 2) * out.strides[nax[1]]);
            nB[n] = (uint32_t)((n & 3) * s[B][nax[0]] + (n >> 2) * s[B][nax[1]]);
            contig = contig && nout[n] == (uint32_t)n;
        }
        {
            int64_t expect = 16;
            for (int i = 0; i < nlo; ++i) { contig = contig && rost[i] == expect; expect *= rcard[i]; }
        }
        // iteration order of the wave-uniform (hi) axes: those only one of the two tables depends on run fastest, so that
        // the other table's operand stays in the kernel's registers over consecutive iterations (outer_mfma_call)
        {
            int64_t tc[kRawAxes], to[kRawAxes], tt[kRawAxes], tb0[kRawAxes], tb1[kRawAxes];
            int k = 0;
            for (int pass = 0; pass < 2; ++pass)
                for (int i = nlo; i < nr; ++i) {
                    const bool single = (rb[0][i] == 0) != (rb[1][i] == 0);
                    if (single == (pass == 0)) { tc[k] = rcard[i]; to[k] = rost[i]; tt[k] = rtst[i]; tb0[k] = rb[0][i]; tb1[k] = rb[1][i]; ++k; }
                }
            for (int i = 0; i < k; ++i) { rcard[nlo + i] = tc[i]; rost[nlo + i] = to[i]; rtst[nlo + i] = tt[i]; rb[0][nlo + i] = tb0[i]; rb[1][nlo + i] = tb1[i]; }
        }
        // merge adjacent R axes contiguous in the output, in T and in both inputs
        int64_t mc[kRawAxes], mo[kRawAxes], mt[kRawAxes], mb[2][kRawAxes];
        int ma = 0, mlo = 0;
        for (int i = 0; i < nr; ++i) {
            bool merge = ma > 0 && i != nlo && mo[ma - 1] * mc[ma - 1] == rost[i] && mt[ma - 1] * mc[ma - 1] == rtst[i] &&
                         mc[ma - 1] * rcard[i] < (1 << 30);
            for (int b = 0; b < 2 && merge; ++b) merge = mb[b][ma - 1] * mc[ma - 1] == rb[b][i];
            if (merge) {
                mc[ma - 1] *= rcard[i];
            } else {
                mc[ma] = rcard[i];
                mo[ma] = rost[i];
                mt[ma] = rtst[i];
                for (int b = 0; b < 2; ++b) mb[b][ma] = rb[b][i];
                ++ma;
                if (i < nlo) ++mlo;
            }
        }
        if (ma > kMaxAxes) return false;
        const int nT = 2 + nctrl;
        const int words = kHdrWords + 4 * 2 + ns * (4 + nT) + nT + 16 + 16 + 3 * ma + 2 * ma;
        if (words > kMaxStepWords) return false;
        uint32_t *w = prog.extend(words);
        header(w, kKindFiber, 2 + ns, ma, mlo, cx, false, lo, rcells / lo, out.off, words);
        w[1] |= (kFlagOuter | (contig ? kFlagContig : 0u)) << 16;
        w[1] |= (uint32_t)row_stride << kRowStrideShift;
        w[7] = 2u | ((uint32_t)ns << 4) | (2u << 8) | ((uint32_t)nctrl << 12) | (16u << 16);
        w[8] = (uint32_t)T | ((uint32_t)c1 << 16);
        uint32_t *p = w + kHdrWords;
        for (int b = 0; b < 2; ++b) {
            put_off(p, ins[big[b]]);
            *p++ = (uint32_t)(int32_t)xs[big[b]][0];
            *p++ = (uint32_t)(int32_t)xs[big[b]][1];
        }
        for (int k = 0; k < ns; ++k) {
            const int j = small[k];
            put_off(p, ins[j]);
            *p++ = (uint32_t)(int32_t)xs[j][0];
            *p++ = (uint32_t)(int32_t)xs[j][1];
            *p++ = (uint32_t)(int32_t)s[j][nax[0]];
            *p++ = (uint32_t)(int32_t)s[j][nax[1]];
            for (int i = 0; i < nctrl; ++i) *p++ = (uint32_t)(int32_t)s[j][ctrl[i]];
        }
        *p++ = 4;  // tcard: the two N axes, then the ctrl axes
        *p++ = 4;
        for (int i = 0; i < nctrl; ++i) *p++ = (uint32_t)net.card[out.vars[ctrl[i]]];
        for (int n = 0; n < 16; ++n) *p++ = nout[n];
        for (int n = 0; n < 16; ++n) *p++ = nB[n];
        for (int a = 0; a < ma; ++a) { *p++ = (uint32_t)mc[a]; *p++ = (uint32_t)mo[a]; *p++ = (uint32_t)mt[a]; }
        for (int b = 0; b < 2; ++b)
            for (int a = 0; a < ma; ++a) *p++ = (uint32_t)(int32_t)mb[b][a];
        return true;
    }

    // CHAIN form (planner.h): three 4-state variables, one big input.  Any of the three may be the one summed out in
    // registers (its own small inputs must not depend on the other two).
    bool emit_chain(const PF *const *ins, int n_in, const Strides &s, const XStrides &xs_in, PF &out) {
        static const int kPerm[3][3] = {{0, 1, 2}, {0, 2, 1}, {1, 2, 0}};
        for (int pi = 0; pi < 3; ++pi) {
            XStrides xs;
            for (int j = 0; j < n_in; ++j)
                for (int k = 0; k < 3; ++k) xs[j][k] = xs_in[j][kPerm[pi][k]];
            if (emit_chain_as(ins, n_in, s, xs, out)) return true;
        }
        return false;
    }

    bool emit_chain_as(const PF *const *ins, int n_in, const Strides &s_in, const XStrides &xs, PF &out) {
        int big = -1, g12[kMaxIn], g3[kMaxIn], n12 = 0, n3s = 0;
        bool x3dep = false;
        for (int j = 0; j < n_in; ++j) {
            if (ins[j]->cells > net.small_cells) {
                if (big >= 0) return false;
                big = j;
            } else if (xs[j][2] != 0 && xs[j][0] == 0 && xs[j][1] == 0) {
                g3[n3s++] = j;
            } else {
                g12[n12++] = j;
                x3dep = x3dep || xs[j][2] != 0;
            }
        }
        if (big < 0 || xs[big][0] == 0 || xs[big][1] == 0 || xs[big][2] == 0) return false;
        if (n12 > kMaxSmall || n3s < 1 || n3s > 2) return false;
        const int na = out.n;
        if (na < 3) return false;
        // the three new axes: two of the pair tables, one of the third variable's
        int nax12[2], nn12 = 0, nax3 = -1;
        bool n12dep = false;
        for (int a = 0; a < na; ++a) {
            if (s_in[big][a] != 0) continue;
            bool dep12 = false, dep3 = false;
            for (int k = 0; k < n12; ++k) dep12 = dep12 || s_in[g12[k]][a] != 0;
            for (int k = 0; k < n3s; ++k) dep3 = dep3 || s_in[g3[k]][a] != 0;
            if (net.card[out.vars[a]] != 4) return false;
            if (dep12) {
                if (nn12 >= 2) return false;
                nax12[nn12++] = a;
                n12dep = n12dep || dep3;
            } else if (dep3) {
                if (nax3 >= 0) return false;
                nax3 = a;
            } else {
                return false;
            }
        }
        if (nn12 != 2 || nax3 < 0) return false;
        // the layout of the output is ours to choose: the three new axes become the fastest ones - n12 at strides 1 and
        // 4 (16 lanes of the kernel write one full 128-byte line), n3 at 16 - and the other axes follow ...  (n3 fastest + 16-byte stores was tried: the half-written lines made L2 fetch the
        // output before overwriting it, +19 % HBM reads.)
        int ord[kRawAxes];
        ord[0] = nax12[0]; ord[1] = nax12[1]; ord[2] = nax3;
        {   // ... in the order F stores them: a wave's 64 cells are then 512 contiguous bytes of every F slice too (with
            // the output's own order the lanes of a row block gathered 32-byte pieces of four lines, and the four row
            // blocks - loaded at different times - fetched every line of F 1.5 times from HBM)
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
        int64_t s[kMaxIn][kRawAxes], ostr[kRawAxes];
        int32_t vars2[kRawAxes];
        {
            int64_t cells = 1;
            for (int a = 0; a < na; ++a) {
                vars2[a] = out.vars[ord[a]];
                ostr[a] = cells;
                cells *= net.card[vars2[a]];
                for (int j = 0; j < n_in; ++j) s[j][a] = s_in[j][ord[a]];
            }
        }
        // R axes; ctrl axes of T12 / of T3 = R axes a pair-group / third-group small input depends on
        int64_t rcard[kRawAxes], rost[kRawAxes], rt12[kRawAxes], rt3[kRawAxes], rbig[kRawAxes];
        int c12[kRawAxes], c3[kRawAxes], nc12 = 0, nc3 = 0, nr = 0;
        int64_t T12 = 256, T3 = n12dep ? 256 : 16;
        for (int a = 3; a < na; ++a) {
            rcard[nr] = net.card[vars2[a]];
            rost[nr] = ostr[a];
            rbig[nr] = s[big][a];
            bool dep12 = false, dep3 = false;
            for (int k = 0; k < n12; ++k) dep12 = dep12 || s[g12[k]][a] != 0;
            for (int k = 0; k < n3s; ++k) dep3 = dep3 || s[g3[k]][a] != 0;
            rt12[nr] = rt3[nr] = 0;
            if (dep12) {
                if (nc12 >= 10) return false;
                c12[nc12++] = a;
                rt12[nr] = T12;
                T12 *= rcard[nr];
            }
            if (dep3) {
                if (nc3 >= 10) return false;
                c3[nc3++] = a;
                rt3[nr] = T3;
                T3 *= rcard[nr];
            }
            if (T12 * (x3dep ? 4 : 1) + T3 > kMaxT) return false;
            ++nr;
        }
        const int64_t t12x3 = x3dep ? T12 : 0;
        if (x3dep) T12 *= 4;
        int nlo = 0;
        int64_t lo = 1;
        while (nlo < nr && lo < kLoTarget && lo * rcard[nlo] <= kFiberLoMax) lo *= rcard[nlo++];
        int64_t rcells = 1;
        for (int i = 0; i < nr; ++i) rcells *= rcard[i];
        if (rcells * 64 < net.big_iters) return false;
        {   // the lane-varying block is contiguous in the output: cell l at 64*l
            int64_t expect = 64;
            for (int i = 0; i < nlo; ++i) { if (rost[i] != expect) return false; expect *= rcard[i]; }
        }
        // row stride (rule of emit_fiber, over the ctrl axes of both tables)
        int row_stride = 16, inside = 0;
        {
            bool ok = true;
            int64_t cs = 1;
            for (int i = 0; i < nlo; ++i) {
                if (rt12[i] != 0 || rt3[i] != 0) {
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
            if (inside > 1 || !ok) return false;
        }
        int64_t mc[kRawAxes], mo[kRawAxes], m12[kRawAxes], m3[kRawAxes], mb[kRawAxes];
        int ma = 0, mlo = 0;
        for (int i = 0; i < nr; ++i) {
            const bool merge = ma > 0 && i != nlo && mo[ma - 1] * mc[ma - 1] == rost[i] && m12[ma - 1] * mc[ma - 1] == rt12[i] &&
                               m3[ma - 1] * mc[ma - 1] == rt3[i] && mb[ma - 1] * mc[ma - 1] == rbig[i] && mc[ma - 1] * rcard[i] < (1 << 30);
            if (merge) {
                mc[ma - 1] *= rcard[i];
            } else {
                mc[ma] = rcard[i];
                mo[ma] = rost[i];
                m12[ma] = rt12[i];
                m3[ma] = rt3[i];
                mb[ma] = rbig[i];
                ++ma;
                if (i < nlo) ++mlo;
            }
        }
        if (ma > kMaxAxes || ma < 1) return false;
        const int nT = 2 + nc12 + (x3dep ? 1 : 0);
        const int nd3 = 2 + (n12dep ? 2 : 0) + nc3;
        const int words = kHdrWords + 8 + n12 * (4 + nT) + nT + 16 + 2 + nd3 + n3s * (2 + nd3) + 3 * ma + 2 * ma;
        if (words > kMaxStepWords) return false;
        // commit the axis order of the output
        for (int a = 0; a < na; ++a) { out.vars[a] = vars2[a]; out.strides[a] = ostr[a]; }
        uint32_t *w = prog.extend(words);
        header(w, kKindFiber, n_in, ma, mlo, 16, false, lo, rcells / lo, out.off, words);
        w[1] |= (kFlagChain | kFlagContig) << 16;
        w[1] |= (uint32_t)row_stride << kRowStrideShift;
        w[7] = 2u | ((uint32_t)n12 << 4) | (2u << 8) | ((uint32_t)(nT - 2) << 12) | (16u << 16);
        w[8] = (uint32_t)T12 | (4u << 16);
        uint32_t *p = w + kHdrWords;
        put_off(p, ins[big]);
        *p++ = (uint32_t)(int32_t)xs[big][0];
        *p++ = (uint32_t)(int32_t)xs[big][1];
        *p++ = (uint32_t)T12;
        *p++ = (uint32_t)T3;
        *p++ = (uint32_t)(int32_t)xs[big][2];
        *p++ = (uint32_t)t12x3;
        for (int k = 0; k < n12; ++k) {
            const int j = g12[k];
            put_off(p, ins[j]);
            *p++ = (uint32_t)(int32_t)xs[j][0];
            *p++ = (uint32_t)(int32_t)xs[j][1];
            *p++ = (uint32_t)(int32_t)s[j][0];
            *p++ = (uint32_t)(int32_t)s[j][1];
            for (int i = 0; i < nc12; ++i) *p++ = (uint32_t)(int32_t)s[j][c12[i]];
            if (x3dep) *p++ = (uint32_t)(int32_t)xs[j][2];
        }
        *p++ = 4;
        *p++ = 4;
        for (int i = 0; i < nc12; ++i) *p++ = (uint32_t)net.card[vars2[c12[i]]];
        if (x3dep) *p++ = 4;
        for (int n = 0; n < 16; ++n) *p++ = (uint32_t)n;
        *p++ = (uint32_t)n3s;
        *p++ = (uint32_t)nd3 | (n12dep ? 256u : 0u);
        *p++ = 4;  // x3
        *p++ = 4;  // n3
        if (n12dep) { *p++ = 4; *p++ = 4; }
        for (int i = 0; i < nc3; ++i) *p++ = (uint32_t)net.card[vars2[c3[i]]];
        for (int k = 0; k < n3s; ++k) {
            const int j = g3[k];
            put_off(p, ins[j]);
            *p++ = (uint32_t)(int32_t)xs[j][2];
            *p++ = (uint32_t)(int32_t)s[j][2];
            if (n12dep) { *p++ = (uint32_t)(int32_t)s[j][0]; *p++ = (uint32_t)(int32_t)s[j][1]; }
            for (int i = 0; i < nc3; ++i) *p++ = (uint32_t)(int32_t)s[j][c3[i]];
        }
        for (int a = 0; a < ma; ++a) { *p++ = (uint32_t)mc[a]; *p++ = (uint32_t)mo[a]; *p++ = (uint32_t)m12[a]; }
        for (int a = 0; a < ma; ++a) *p++ = (uint32_t)(int32_t)mb[a];
        for (int a = 0; a < ma; ++a) *p++ = (uint32_t)(int32_t)m3[a];
        return true;
    }

    // SWEEP form (planner.h): k = 3..5 four-state variables X[0..k) - in elimination order - of the one big input, the
    // tile resident in LDS.  Does its own bookkeeping (layout of the output, arena, statistics); returns false - nothing
    // emitted, nothing allocated - when the step does not fit.
    bool emit_sweep(const PF *const *ins, int n_in, const int *X, int k, PF &out) {
        if (k < 2 || k > 5 || n_in - 1 > kSweepMaxSmall) return false;
        const PF *F = nullptr;
        for (int j = 0; j < n_in; ++j)
            if (ins[j]->cells > net.small_cells) {
                if (F) return false;
                F = ins[j];
            }
        if (!F || (F->off & kConstFlag)) return false;
        const int rb = 13 - 2 * k;
        const int64_t Rt = int64_t(1) << rb;
        if (F->cells & ((int64_t(1) << (2 * k)) - 1)) return false;
        const int64_t Rcells = F->cells >> (2 * k);
        if (Rcells < Rt || (Rcells & (Rt - 1))) return false;
        // digits: x_j must sit on one of F's k slowest axes
        int dig[5], var_on[5] = {-1, -1, -1, -1, -1};
        for (int j = 0; j < k; ++j) {
            if (net.card[X[j]] != 4) return false;
            int d = -1;
            for (int a = 0; a < F->n; ++a)
                if (F->vars[a] == X[j]) {
                    for (int q = 0; q < k; ++q)
                        if (F->strides[a] == Rcells << (2 * q)) d = q;
                }
            if (d < 0 || var_on[d] >= 0) return false;
            dig[j] = d;
            var_on[d] = X[j];
        }
        // stages: a small input belongs to the first eliminated variable it mentions
        struct Stage { int cout, ns, nctrl, loop, f[3], t_off, t_cells, src[3], newv; const PF *in[kSweepMaxSmall]; int cvar[3]; } S[5];
        bool used[kSweepMaxSmall + 1] = {};
        Bits introduced;
        introduced.nw = net.nw;
        int t_total = 0, ns_total = 0;
        double in_cells = (double)F->cells;
        for (int j = 0; j < k; ++j) {
            Stage &g = S[j];
            g.ns = 0;
            Bits U;
            U.nw = net.nw;
            for (int i = 0; i < n_in; ++i) {
                if (ins[i] == F || used[i] || !ins[i]->scope.test(X[j])) continue;
                used[i] = true;
                g.in[g.ns++] = ins[i];
                U.or_(ins[i]->scope);
                in_cells += (double)ins[i]->cells;
            }
            ns_total += g.ns;
            Bits fresh = U;
            fresh.andnot(F->scope);
            fresh.andnot(introduced);
            const int nnew = fresh.count();
            if (nnew > 1) return false;
            g.newv = -1;
            if (nnew == 1) {
                fresh.for_each([&](int v) { g.newv = v; });
                if (net.card[g.newv] != 4) return false;
                introduced.set(g.newv);
            }
            g.cout = nnew ? 4 : 1;
            g.nctrl = 0;
            bool ok = true;
            U.for_each([&](int v) {
                if (!ok || v == X[j] || v == g.newv) return;
                int src = -1;
                for (int d = 0; d < k; ++d)
                    if (var_on[d] == v && d != dig[j]) src = d;
                if (src < 0) {
                    // an R axis of F: four states, power-of-two stride
                    for (int a = 0; a < F->n; ++a)
                        if (F->vars[a] == v && F->strides[a] < Rcells) {
                            const int64_t st_ = F->strides[a];
                            if (net.card[v] == 4 && (st_ & (st_ - 1)) == 0) src = 8 + __builtin_ctzll((unsigned long long)st_);
                        }
                }
                if (src < 0 || g.nctrl >= 3) { ok = false; return; }
                if (src >= 8) {  // at most two ctrl values come from r (the kernel keeps their shifts in scalar registers)
                    int nr = 0;
                    for (int c = 0; c < g.nctrl; ++c) nr += g.src[c] >= 8;
                    if (nr >= 2) { ok = false; return; }
                }
                g.src[g.nctrl] = src;
                g.cvar[g.nctrl] = v;
                ++g.nctrl;
            });
            if (!ok) return false;
            g.t_cells = g.cout * 4 << (2 * g.nctrl);
            g.t_off = t_total;
            t_total += g.t_cells;
            if (t_total > kSweepMaxT) return false;
            // thread fields / loop digit (planner.h): a fixed rule, so that the kernel's stage geometry is known at compile time
            g.loop = sweep_loop_digit(k, dig[j]);
            g.f[0] = g.f[1] = g.f[2] = 7;
            for (int d = 0, m = 0; d < k; ++d)
                if (d != dig[j] && d != g.loop) g.f[m++] = d;
            var_on[dig[j]] = g.newv;  // (-1: the digit is dead from here on)
        }
        for (int i = 0; i < n_in; ++i)
            if (ins[i] != F && !used[i]) return false;  // (a small input that mentions none of the eliminated variables)
        // output: surviving digits (ascending) fastest, then F's R axes in F's order
        int kout = 0, surv[5];
        for (int d = 0; d < k; ++d)
            if (var_on[d] >= 0) surv[kout++] = d;
        const int64_t out_cells = Rcells << (2 * kout);
        if (out_cells >= (1ll << 31)) return false;
        int na = 0;
        out.scope.nw = net.nw;
        for (int q = 0; q < net.nw; ++q) out.scope.w[q] = 0;
        for (int q = 0; q < kout; ++q) {
            out.vars[na] = var_on[surv[q]];
            out.strides[na] = int64_t(1) << (2 * q);
            out.scope.set(out.vars[na]);
            ++na;
        }
        for (int a = 0; a < F->n; ++a) {
            if (F->strides[a] >= Rcells) continue;  // (the eliminated variables)
            if (na >= kRawAxes) return false;
            out.vars[na] = F->vars[a];
            out.strides[na] = F->strides[a] << (2 * kout);
            out.scope.set(out.vars[na]);
            ++na;
        }
        out.n = na;
        out.cells = out_cells;
        const int words = kHdrWords + 2 + k * kSweepStageWords + ns_total * kSweepSmallWords;
        if (words > kMaxStepWords) return false;
        out.off = (uint64_t)arena.alloc(out_cells);
        out.alloc = out_cells;
        out.src = -1;
        uint32_t *w = prog.extend(words);
        header(w, kKindSweep, n_in, k, rb, 1 << (2 * k), false, kSweepTileCells, Rcells / Rt, out.off, words);
        w[7] = (uint32_t)kout | ((uint32_t)t_total << 16);
        w[8] = 0;
        for (int q = 0; q < kout; ++q) w[8] |= (uint32_t)surv[q] << (4 * q);
        {
            bool canon = net.sweep_canon != 0;
            for (int j = 0; j < k; ++j) canon = canon && dig[j] == k - 1 - j;
            if (canon) w[1] |= kFlagSweepCanon << 16;
        }
        uint32_t *p = w + kHdrWords;
        put_off(p, F);
        for (int j = 0; j < k; ++j) {
            const Stage &g = S[j];
            *p++ = (uint32_t)dig[j] | ((uint32_t)g.cout << 4) | ((uint32_t)g.ns << 8) | ((uint32_t)g.nctrl << 12) | ((uint32_t)g.loop << 16) |
                   ((uint32_t)g.f[0] << 20) | ((uint32_t)g.f[1] << 24) | ((uint32_t)g.f[2] << 28);
            *p++ = (uint32_t)g.t_off | ((uint32_t)g.t_cells << 16);
            for (int c = 0; c < 3; ++c) *p++ = c < g.nctrl ? ((uint32_t)g.src[c] | ((uint32_t)(g.cout * 4 << (2 * c)) << 8)) : 0u;
        }
        auto stride_of = [](const PF *f, int v) -> int64_t {
            for (int a = 0; a < f->n; ++a)
                if (f->vars[a] == v) return f->strides[a];
            return 0;
        };
        for (int j = 0; j < k; ++j) {
            const Stage &g = S[j];
            for (int i = 0; i < g.ns; ++i) {
                put_off(p, g.in[i]);
                *p++ = (uint32_t)(int32_t)(g.newv >= 0 ? stride_of(g.in[i], g.newv) : 0);
                *p++ = (uint32_t)(int32_t)stride_of(g.in[i], X[j]);
                for (int c = 0; c < 3; ++c) *p++ = (uint32_t)(int32_t)(c < g.nctrl ? stride_of(g.in[i], g.cvar[c]) : 0);
            }
        }
        w[9] = (uint32_t)(((int64_t)in_cells + out_cells + 2) >> 2);
        st.alg_bytes += 8.0 * (in_cells + (double)out_cells);
        st.alg_flops += (double)k * 4.0 * (double)F->cells;
        st.max_step_cells = std::max(st.max_step_cells, (double)F->cells);
        st.n_steps += 1;
        for (int j = 0; j < n_in; ++j)
            if (ins[j]->alloc) arena.release((int64_t)ins[j]->off, ins[j]->alloc);
        return true;
    }

    // Emit one step: multiply `ins`, sum out the nx (0..2) variables X (nx = 0: product only); the new factor is
    // written to `out`.  fiber_only: emit nothing and return false unless the step fits the FIBER form (used to try
    // the joint elimination of two variables).
    bool emit(const PF *const *ins, int n_in, const int *X, int nx, bool final_, int64_t final_off, PF &out, bool fiber_only) {
        out.scope.nw = net.nw;  // (only the words the network uses: the rest of a pool entry's scope is never read)
        for (int k = 0; k < net.nw; ++k) out.scope.w[k] = 0;
        for (int j = 0; j < n_in; ++j) out.scope.or_(ins[j]->scope);
        for (int k = 0; k < nx; ++k) out.scope.clr(X[k]);
        int na = 0;
        bool overflow = false;
        out.scope.for_each([&](int v) { if (na < kRawAxes) out.vars[na++] = v; else overflow = true; });
        if (overflow) { if (!fiber_only) err = "a factor has more than " + std::to_string(kRawAxes) + " axes"; return false; }
        // layout: longest-living variable fastest (insertion sort on the key, descending)
        for (int i = 1; i < na; ++i) {
            const int v = out.vars[i];
            int k = i - 1;
            while (k >= 0 && (key[out.vars[k]] < key[v] || (key[out.vars[k]] == key[v] && out.vars[k] > v))) { out.vars[k + 1] = out.vars[k]; --k; }
            out.vars[k + 1] = v;
        }
        out.n = na;
        int64_t cells = 1;
        for (int a = 0; a < na; ++a) {
            out.strides[a] = cells;
            cells *= net.card[out.vars[a]];
            if (cells >= (1ll << 31)) { if (!fiber_only) err = "an intermediate factor has >= 2^31 cells"; return false; }
        }
        for (int a = 0; a < na; ++a) pos[out.vars[a]] = a;
        out.cells = cells;
        // per-input strides along the output axes, and along the eliminated variables
        Strides s;
        XStrides xs;
        double in_cells = 0;
        for (int j = 0; j < n_in; ++j) {
            for (int a = 0; a < na; ++a) s[j][a] = 0;
            xs[j][0] = xs[j][1] = xs[j][2] = 0;
            for (int k = 0; k < ins[j]->n; ++k) {
                const int v = ins[j]->vars[k];
                if (nx > 0 && v == X[0]) xs[j][0] = ins[j]->strides[k];
                else if (nx > 1 && v == X[1]) xs[j][1] = ins[j]->strides[k];
                else if (nx > 2 && v == X[2]) xs[j][2] = ins[j]->strides[k];
                else s[j][pos[v]] = ins[j]->strides[k];
            }
            in_cells += (double)ins[j]->cells;
        }
        for (int a = 0; a < na; ++a) pos[out.vars[a]] = -1;
        const int c1 = nx > 0 ? net.card[X[0]] : 1;
        const int cx = nx > 1 ? c1 * net.card[X[1]] : c1;
        if (final_) {
            out.off = (uint64_t)final_off;
            out.alloc = 0;
        } else {
            out.off = (uint64_t)arena.alloc(cells);
            out.alloc = cells;
        }
        const size_t step_base = prog.size;
        const bool fiber = nx == 3 ? (!final_ && emit_chain(ins, n_in, s, xs, out))
                                   : !final_ && ((net.outer && nx > 0 && emit_outer(ins, n_in, s, xs, out, cx, c1)) || emit_fiber(ins, n_in, s, xs, out, cx, c1));
        if (!fiber) {
            if (fiber_only) {
                if (out.alloc) arena.release((int64_t)out.off, out.alloc);
                return false;
            }
            int64_t xs1[kMaxIn];
            for (int j = 0; j < n_in; ++j) xs1[j] = xs[j][0];
            emit_generic(ins, n_in, s, xs1, out, cells, cx, final_);
        }
        if (!err.empty()) return false;
        prog.data[step_base + 9] = (uint32_t)(((int64_t)in_cells + cells + 2) >> 2);  // section-8(d) cells of this step, units of 4
        st.alg_bytes += 8.0 * (in_cells + (double)cells);
        double pc = (double)cells;  // cells of the product scope = the output's cells x the eliminated cardinalities
        for (int k = 0; k < nx; ++k) pc *= net.card[X[k]];
        st.alg_flops += n_in * pc;
        st.max_step_cells = std::max(st.max_step_cells, pc);
        st.n_steps += 1;
        for (int j = 0; j < n_in; ++j)
            if (ins[j]->alloc) arena.release((int64_t)ins[j]->off, ins[j]->alloc);
        return true;
    }
};

}  // namespace

static std::string plan_request_rec(const Network &net, const Request &rq, ProgBuf &prog, PlanStats &st, PlanRecord *rec) {
    PROF(0);
    Scratch &S = scratch();
    // relevant = query | event | ancestors(...)  (bayes_net.py:763-765); hidden = relevant - query - event (766)
    Bits rel, qb, eb;
    rel.nw = qb.nw = eb.nw = net.nw;
    for (int i = 0; i < rq.nq; ++i) { qb.set(rq.qvars[i]); rel.set(rq.qvars[i]); rel.or_(net.anc[rq.qvars[i]]); }
    for (int i = 0; i < rq.ne; ++i) { eb.set(rq.evars[i]); rel.set(rq.evars[i]); rel.or_(net.anc[rq.evars[i]]); }
    if (!net.prune || rq.no_prune)  // full_joint_dist / predict_proba multiply *all* CPTs (bayes_net.py:460): with sparse or
        for (int v = 0; v < net.n_vars; ++v) rel.set(v);  // unnormalised CPTs a barren node does not sum to 1
    Bits hidden = rel;
    hidden.andnot(qb);
    hidden.andnot(eb);
    int32_t ecode_buf[kMaxVars];
    if (rq.ecodes)
        for (int i = 0; i < rq.ne; ++i) ecode_buf[rq.evars[i]] = rq.ecodes[i];
    else
        for (int i = 0; i < rq.ne; ++i) ecode_buf[rq.evars[i]] = 0;

    // factors = evidence-sliced CPTs of the relevant nodes (bayes_net.py:768-776): the evidence axis is
    // not copied away but folded into the base offset (stride 0 afterwards)
    std::vector<PF> &pool = S.pool;
    std::vector<int> &live = S.live;
    pool.clear();
    live.clear();
    pool.reserve(3 * (size_t)net.n_vars + 64);  // never reallocates below: `ins` holds pointers into it
    std::vector<Bits> &scopes = S.scopes;
    std::vector<double> &scells = S.scope_cells;
    scopes.clear();
    scells.clear();
    std::string err;
    rel.for_each([&](int v) {
        pool.emplace_back();
        PF &f = pool.back();
        f.scope.nw = net.nw;
        uint64_t off = (uint64_t)net.pool_off[v];
        int64_t cells = 1;
        for (size_t k = 0; k < net.scope[v].size(); ++k) {
            const int u = net.scope[v][k];
            if (eb.test(u)) {
                off += (uint64_t)(net.cstride[v][k] * ecode_buf[u]);
            } else if (net.card[u] > 1) {
                if (f.n >= kRawAxes) { err = "a CPT has more than " + std::to_string(kRawAxes) + " free axes"; return; }
                f.scope.set(u);
                f.vars[f.n] = u;
                f.strides[f.n] = net.cstride[v][k];
                ++f.n;
                cells *= net.card[u];
            }
        }
        f.off = off | kConstFlag;
        f.src = v;
        f.cells = cells;
        live.push_back((int)pool.size() - 1);
        if (net.n_vars > 128) {
            scopes.push_back(f.scope);
            scells.push_back((double)cells);
        }
    });
    if (!err.empty()) return err;
    // single-state variables carry no information: they are never axes, never eliminated
    Bits trivial;
    trivial.nw = net.nw;
    hidden.for_each([&](int v) { if (net.card[v] <= 1) trivial.set(v); });
    hidden.andnot(trivial);

    // candidate elimination orders, cheapest by the byte model wins
    std::vector<int32_t> &best = S.best, &cand = S.cand;
    best.clear();
    double best_cost = std::numeric_limits<double>::infinity();
    auto consider = [&]() {  // evaluates `cand`
        const double c = simulate(net, scopes, scells, cand, best_cost);
        if (c < best_cost) { best_cost = c; best.swap(cand); }
    };
    if (rq.n_order >= 0) {
        // the order was found by the device order search (same code, order_search.h)
        best.assign(rq.order, rq.order + rq.n_order);
    } else if (net.n_vars <= 128) {
        PROF(1);
        OrderScratch &OS = order_scratch();
        order_search(net.order_view(), OS, rq.nq, rq.qvars, rq.ne, rq.evars, rq.no_prune);
        best.assign(OS.best, OS.best + OS.n_best);
    } else if (hidden.any()) {
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
    S.key.assign(net.n_vars, 0.0);
    S.pos.assign(net.n_vars, -1);
    Emitter em{net, prog, st, Arena{}, S.key, S.pos, "", rec};
    for (size_t i = 0; i < best.size(); ++i) S.key[best[i]] = (double)i;
    for (int i = 0; i < rq.nq; ++i) S.key[rq.qvars[i]] = 1e9 + i;

    const size_t count_pos = prog.size;
    prog.push(0);
    const double steps0 = st.n_steps;
    const PF *ins[kMaxVars + 8];
    // multiply/eliminate with at most kMaxIn inputs per step: larger products are pre-multiplied
    auto emit_limited = [&](int n_in, int x, bool final_, int64_t final_off) -> int {
        while (n_in > kMaxIn && em.err.empty()) {
            if (pool.size() + 2 > pool.capacity()) { em.err = "planner factor pool exhausted"; return -1; }
            std::sort(ins, ins + n_in, [](const PF *a, const PF *b) { return a->cells < b->cells; });
            pool.emplace_back();
            em.emit(ins, kMaxIn, nullptr, 0, false, 0, pool.back(), false);
            for (int k = kMaxIn; k < n_in; ++k) ins[k - kMaxIn] = ins[k];
            n_in -= kMaxIn;
            ins[n_in++] = &pool.back();
        }
        if (!em.err.empty()) return -1;
        if (pool.size() + 1 > pool.capacity()) { em.err = "planner factor pool exhausted"; return -1; }
        pool.emplace_back();
        em.emit(ins, n_in, &x, x >= 0 ? 1 : 0, final_, final_off, pool.back(), false);
        return (int)pool.size() - 1;
    };
    // Which factors mention a variable: every variable keeps the set of factor slots (pool indices) whose scope contains
    // it, `alive` the slots not yet consumed.  Slots are handed out in creation order, so walking a set in ascending order
    // visits the factors in the order of the reference's factor list (bayes_net.py:780-784 pops from it, 786 appends).
    const int sw = (int)((pool.capacity() + 63) / 64);
    std::vector<uint64_t> &mem = S.mem, &alive = S.slot_alive;
    if (mem.size() < (size_t)net.n_vars * sw) mem.resize((size_t)net.n_vars * sw);
    alive.assign(sw, 0);
    rel.for_each([&](int v) { std::fill(mem.begin() + (size_t)v * sw, mem.begin() + (size_t)(v + 1) * sw, 0ull); });
    auto add_factor = [&](int idx) {
        alive[idx >> 6] |= 1ull << (idx & 63);
        pool[idx].scope.for_each([&](int v) { mem[(size_t)v * sw + (idx >> 6)] |= 1ull << (idx & 63); });
    };
    for (int idx : live) add_factor(idx);
    // factors alive whose scope contains a (or b, if b >= 0), in slot order; f(idx) returns false to stop early
    auto each_with = [&](int a, int b, auto f) {
        for (int k = 0; k < sw; ++k) {
            uint64_t m = (mem[(size_t)a * sw + k] | (b >= 0 ? mem[(size_t)b * sw + k] : 0ull)) & alive[k];
            for (; m; m &= m - 1)
                if (!f(k * 64 + __builtin_ctzll(m))) return;
        }
    };
    const double log2_small = std::log2((double)net.small_cells), log2_big = std::log2((double)net.big_iters);
    auto consume = [&](const PF *f) {
        const int idx = (int)(f - pool.data());
        alive[idx >> 6] &= ~(1ull << (idx & 63));
    };
    for (size_t i = 0; i < best.size(); ++i) {
        const int32_t x = best[i];
        // pop every factor mentioning x (bayes_net.py:780-784)
        int n_in = 0;
        each_with(x, -1, [&](int idx) { ins[n_in++] = &pool[idx]; return true; });
        for (int j = 0; j < n_in; ++j) consume(ins[j]);
        // SWEEP: up to five consecutive 4-state variables of one big table in a single pass, the tile resident in LDS
        // (planner.h).  sweep_max = how many of the next variables could join: all on the one big input, every other factor
        // that mentions them small.  Four and five variables are tried first, three only after the CHAIN form below.
        int sweep_max = 0, sweep_n[5] = {0, 0, 0, 0, 0};
        const PF *sweep_ins[kSweepMaxSmall + 2];
        if (net.fuse && net.sweep >= 3 && i + (size_t)std::min(2, net.sweep_min - 1) < best.size() && net.card[x] == 4 && sw <= 16 && n_in - 1 <= kSweepMaxSmall &&
            pool.size() + 1 <= pool.capacity()) {
            int nbig = 0;
            const PF *bigf = nullptr;
            for (int j = 0; j < n_in; ++j)
                if (ins[j]->cells > net.small_cells) { ++nbig; bigf = ins[j]; }
            if (nbig == 1 && !(bigf->off & kConstFlag) && bigf->cells >= 16 * (int64_t)net.big_iters && bigf->cells >= 2 * kSweepTileCells) {
                uint64_t taken[16];
                for (int q = 0; q < sw; ++q) taken[q] = 0;
                int n_all = n_in;
                for (int j = 0; j < n_in; ++j) sweep_ins[j] = ins[j];
                sweep_max = 1;
                sweep_n[0] = n_in;
                for (int j = 1; j < std::min(net.sweep, 5) && i + j < best.size(); ++j) {
                    const int32_t xj = best[i + j];
                    if (net.card[xj] != 4 || !bigf->scope.test(xj)) break;
                    bool ok = true;
                    each_with(xj, -1, [&](int idx) {
                        if (taken[idx >> 6] >> (idx & 63) & 1) return true;
                        if (pool[idx].cells > net.small_cells || n_all - 1 >= kSweepMaxSmall) { ok = false; return false; }
                        taken[idx >> 6] |= 1ull << (idx & 63);
                        sweep_ins[n_all++] = &pool[idx];
                        return true;
                    });
                    if (!ok) break;
                    sweep_n[j] = n_all;
                    sweep_max = j + 1;
                }
            }
        }
        auto try_sweep = [&](int k_hi, int k_lo) -> bool {
            for (int k = std::min(k_hi, sweep_max); k >= k_lo; --k) {
                int X[5];
                for (int j = 0; j < k; ++j) X[j] = best[i + j];
                pool.emplace_back();
                if (em.emit_sweep(sweep_ins, sweep_n[k - 1], X, k, pool.back())) {
                    for (int j = n_in; j < sweep_n[k - 1]; ++j) consume(sweep_ins[j]);
                    add_factor((int)pool.size() - 1);
                    i += (size_t)k - 1;
                    return true;
                }
                pool.pop_back();
            }
            return false;
        };
        if (sweep_max >= 4 && try_sweep(5, 4)) continue;
        // Joint elimination: if the factor this step creates is a big table that the very next step consumes, both
        // variables are summed out in one pass over the inputs and the intermediate never touches HBM.
        // CHAIN: three consecutive 4-state variables of one big table in a single pass (planner.h)
        if (net.fuse && net.chain && i + 2 < best.size() && n_in < kMaxIn && pool.size() + 1 <= pool.capacity() &&
            net.card[x] == 4 && net.card[best[i + 1]] == 4 && net.card[best[i + 2]] == 4) {
            const int32_t x2 = best[i + 1], x3 = best[i + 2];
            int nbig = 0;
            const PF *bigf = nullptr;
            for (int j = 0; j < n_in; ++j)
                if (ins[j]->cells > net.small_cells) { ++nbig; bigf = ins[j]; }
            if (nbig == 1 && bigf->scope.test(x2) && bigf->scope.test(x3) && bigf->cells >= 16 * (int64_t)net.big_iters) {
                int n3 = n_in;
                bool fits = true;
                each_with(x2, x3, [&](int idx) {
                    const PF &f = pool[idx];
                    if (n3 >= kMaxIn || f.cells > net.small_cells) { fits = false; return false; }
                    ins[n3++] = &f;
                    return true;
                });
                if (fits) {  // exactly three new variables (the frontier keeps its width), cheap to check before any layout work
                    Bits u;
                    u.nw = net.nw;
                    for (int j = 0; j < n3; ++j) u.or_(ins[j]->scope);
                    fits = u.count() - bigf->scope.count() == 3;
                }
                if (fits) {
                    const int X[3] = {x, x2, x3};
                    pool.emplace_back();
                    if (em.emit(ins, n3, X, 3, false, 0, pool.back(), true)) {
                        for (int j = n_in; j < n3; ++j) consume(ins[j]);
                        add_factor((int)pool.size() - 1);
                        i += 2;
                        continue;
                    }
                    pool.pop_back();
                    if (!em.err.empty()) return em.err;
                }
            }
        }
        if (sweep_max >= 3 && try_sweep(3, 3)) continue;
        if (net.sweep_min <= 2 && sweep_max >= 2 && try_sweep(2, 2)) continue;  // (a pair of one big table: before the FIBER pair form)
        if (net.fuse && i + 1 < best.size() && n_in < kMaxIn && pool.size() + 1 <= pool.capacity()) {
            const int32_t x2 = best[i + 1];
            bool link = false;
            Bits u;
            u.nw = net.nw;
            for (int j = 0; j < n_in; ++j) { link = link || ins[j]->scope.test(x2); u.or_(ins[j]->scope); }
            if (link && net.card[x] * net.card[x2] <= kMaxCx &&
                scope_log2(net, u) - net.log2card[x] > log2_small) {
                int n2 = n_in;
                bool fits = true;
                each_with(x2, -1, [&](int idx) {
                    if (n2 >= kMaxIn) { fits = false; return false; }
                    ins[n2++] = &pool[idx];
                    u.or_(pool[idx].scope);
                    return true;
                });
                // cheap necessary conditions of the FIBER form, before any emission work (most candidates fail here):
                // one or two big inputs, and enough R cells (output cells / predicted NC) for a tiled step
                if (fits) {
                    int nbig = 0;
                    Bits bigscope;
                    bigscope.nw = net.nw;
                    for (int j = 0; j < n2; ++j)
                        if (ins[j]->cells > net.small_cells) { ++nbig; bigscope.or_(ins[j]->scope); }
                    Bits nvars = u;  // output variables no big input depends on: the N axes
                    nvars.andnot(bigscope);
                    double nc = 1;
                    nvars.for_each([&](int v) { if (v != x && v != x2 && nc * net.card[v] <= kMaxNC) nc *= net.card[v]; });
                    const double out_log2 = scope_log2(net, u) - net.log2card[x] - net.log2card[x2];
                    if (nbig < 1 || nbig > 2 || out_log2 < log2_big) fits = false;
                }
                if (fits) {
                    const int X[2] = {x, x2};
                    pool.emplace_back();
                    if (em.emit(ins, n2, X, 2, false, 0, pool.back(), true)) {
                        for (int j = n_in; j < n2; ++j) consume(ins[j]);
                        add_factor((int)pool.size() - 1);
                        ++i;
                        continue;
                    }
                    pool.pop_back();
                    if (!em.err.empty()) return em.err;
                }
            }
        }
        const int out = emit_limited(n_in, x, false, 0);  // pointwise_mul + sum_out (785)
        if (!em.err.empty()) return em.err;
        add_factor(out);
    }
    // posterior = pointwise_mul(factors) / sum (bayes_net.py:789-790), written in the caller's
    // query order (C-order, last query variable fastest)
    st.out_cells = 1;
    for (int i = 0; i < rq.nq; ++i) st.out_cells *= net.card[rq.qvars[i]];
    int n_in = 0;
    for (int k = 0; k < sw; ++k)
        for (uint64_t m = alive[k]; m; m &= m - 1) ins[n_in++] = &pool[k * 64 + __builtin_ctzll(m)];
    emit_limited(n_in, -1, true, rq.out_off);
    if (!em.err.empty()) return em.err;
    prog.data[count_pos] = (uint32_t)(st.n_steps - steps0);
    st.arena_cells = std::max(st.arena_cells, em.arena.top);
    return "";
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

// Cut one request's program into work items (see Schedule): a maximal run of small steps is one SEGMENT, every big
// step is a level of its own, tiled.  Appends to `out`, returns the number of items.
static uint32_t tag_request(const Network &net, const uint32_t *prog, std::vector<Tag> &out) {
    const int n_steps = (int)prog[0];
    const size_t first = out.size();
    uint32_t off = 1;
    uint16_t level = 0;
    uint32_t seg_first = 0, seg_steps = 0;
    double seg_bytes = 0;
    auto flush = [&]() {
        if (seg_steps) {
            out.push_back({seg_first, seg_steps | kItemSegment, 1u, level, (uint16_t)kKidSeg, (float)seg_bytes});
            ++level;
            seg_steps = 0;
            seg_bytes = 0;
        }
    };
    for (int s = 0; s < n_steps; ++s) {
        const uint32_t *w = prog + off;
        const double bytes = (double)step_cost_bytes(w);
        if (step_is_tiled(net, w)) {
            flush();
            const uint32_t th = (uint32_t)step_tile_h(net, w);
            // (SWEEP items carry their tile count: build_schedule sizes the workgroups of a level's sweep launch as a whole)
            out.push_back({off, (w[0] & 0xff) == kKindSweep ? w[3] : th, (w[3] + th - 1) / th, level, (uint16_t)kernel_id_of_step(w), (float)bytes});
            ++level;
        } else {
            if (!seg_steps) seg_first = off;
            ++seg_steps;
            seg_bytes += bytes;
        }
        off += w[6];
    }
    flush();
    return (uint32_t)(out.size() - first);
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
    static constexpr uint64_t kWindow = 32768, kProbe = 512;
    PlanRecord rec;
    std::string key;
};

uint64_t option_signature(const Network &net) {
    uint64_t h = 1469598103934665603ull;
    auto mix = [&](uint64_t v) { h = (h ^ v) * 1099511628211ull; };
    mix((uint64_t)net.small_cells); mix((uint64_t)net.big_iters); mix((uint64_t)net.tile_h); mix((uint64_t)net.fuse);
    mix((uint64_t)net.chain); mix((uint64_t)net.sweep); mix((uint64_t)net.sweep_iters); mix((uint64_t)net.sweep_min); mix((uint64_t)net.sweep_adapt); mix((uint64_t)net.order_weights); mix((uint64_t)net.hint_sorted.size()); mix((uint64_t)net.sweep_canon); mix((uint64_t)net.outer); mix((uint64_t)net.prune); mix((uint64_t)net.minfill_above);
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
    }
    return c;
}

}  // namespace

void plan_batch(const Network &net, ThreadPool &pool, std::vector<ProgBuf> &bufs, int64_t b0, int64_t b1,
                const int64_t *q_off, const int32_t *q_vars, const int64_t *e_off, const int32_t *e_vars,
                const int32_t *e_codes, const int64_t *out_off, const char *skip, BatchPlan &ck, bool no_prune,
                const uint8_t *orders, const int32_t *order_len) {
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
            rq.out_off = out_off[b] - out_off[b0];
            rq.no_prune = no_prune;
            if (orders) { rq.order = orders + (size_t)i * 128; rq.n_order = order_len[i]; }
            PlanStats st;
            // plan templates (see above): probe at the start of every window, stay on while shapes repeat
            PlanCache *pc = store ? &plan_cache(store) : nullptr;
            bool use_cache = false;
            if (pc) {
                const uint64_t w = pc->seen++ % PlanCache::kWindow;
                if (w == 0) { pc->probe_hits = 0; pc->on = true; }
                if (w == PlanCache::kProbe) pc->on = pc->probe_hits * 4 >= PlanCache::kProbe;
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
                    ck.tag_count[i] = tag_request(net, prog.data + ck.local_off[i], tags);
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
                ck.tag_count[i] = tag_request(net, prog.data + ck.local_off[i], tags);
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

int fiber_cx_class(const uint32_t *w) {
    const int cx = (int)(w[1] & 0xffff), c1 = (int)(w[8] >> 16);
    if (cx == 4 && c1 == 4) return 0;
    if (cx == 16 && c1 == 4) return 1;
    return 2;
}

int fiber_nc_class(const uint32_t *w) {
    const int NC = (int)(w[7] >> 16);
    const bool contig = ((w[1] >> 16) & kFlagContig) != 0;
    if ((w[1] >> 16) & kFlagOuter) return 5;
    if (NC == 1) return 0;
    if (NC == 4 && contig) return 1;
    if (NC == 16 && ((w[1] >> kRowStrideShift) & 0xff) && fiber_cx_class(w) < 2) return 4;
    if (NC == 16 && contig) return 2;
    return 3;
}

int kernel_id_of_step(const uint32_t *w) {
    const uint32_t kind = w[0] & 0xff;
    if (kind == kKindSweep) return kKidSweep;
    if (kind == kKindFiber) {
        const int nb = w[7] & 0xf;
        if ((w[1] >> 16) & kFlagChain) return kKidChain;
        return kKidFiber0 + (nb - 1) * 18 + fiber_cx_class(w) * 6 + fiber_nc_class(w);
    }
    const int n_in = (w[0] >> 8) & 0xff;
    return kKidGeneric0 + std::min(std::max(n_in, 1), kMaxIn) - 1;
}

// section-8(d) algorithmic bytes of one step (the planner stores (input + output cells) / 4 in w9)
int64_t step_cost_bytes(const uint32_t *w) { return 32 * (int64_t)w[9]; }

bool step_is_tiled(const Network &net, const uint32_t *w) {
    if ((w[0] & 0xff) == kKindFiber || (w[0] & 0xff) == kKindSweep) return true;  // FIBER / SWEEP steps are only emitted above big_iters
    const bool fin = (w[1] >> 16) & kFlagFinal;
    return !fin && (int64_t)w[2] * (int64_t)w[3] >= net.big_iters;
}

int step_tile_h(const Network &net, const uint32_t *w) {
    if ((w[0] & 0xff) == kKindSweep) return std::max(1, std::min(net.sweep_iters, kTileMax));  // (tiles of 64 KiB in, <= 64 KiB out)
    if (net.tile_h > 0) return std::min(net.tile_h, kTileMax);
    // bytes one hi iteration moves = the step's section-8(d) traffic / hi (broadcast re-reads of a small "big" input
    // are cache hits, they do not count)
    const int64_t per_iter = std::max<int64_t>(1, step_cost_bytes(w) / std::max<int64_t>(1, (int64_t)w[3]));
    // (CHAIN steps - 256 KiB per iteration - end up with one iteration per tile; 4 per tile measured 11 % slower)
    return (int)std::max<int64_t>(1, std::min<int64_t>(kTileMax, net.tile_bytes / per_iter));
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
        for (int kid = 0; kid < kNumKernels; ++kid) add(kid);                               // the streaming classes
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
