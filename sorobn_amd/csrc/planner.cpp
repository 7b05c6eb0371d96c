// Host-side planner: request -> step program (see planner.h for the role and the encoding).
#include "planner.h"

#include <algorithm>
#include <cmath>
#include <cstring>
#include <functional>
#include <limits>

namespace mibn {

// ------------------------------------------------------------------------------------ Network

std::string Network::set(int32_t n, const int32_t *card_, const int64_t *scope_off, const int32_t *scope_vars,
                         const int64_t *value_off, const double *values) {
    if (n < 0 || n > kMaxVars) return "n_vars out of range (max " + std::to_string(kMaxVars) + ")";
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
    return err;
}

std::string validate_request(const Network &net, const Request &rq) {
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

struct PF {  // planning-time factor
    Bits scope;                    // free (non-evidence) variables
    std::vector<int32_t> vars;     // axes, any order
    std::vector<int64_t> strides;  // stride (doubles) per axis
    uint64_t off = 0;              // arena offset, or pool offset | kConstFlag
    int64_t cells = 0;             // product of the free cardinalities
    int64_t alloc = 0;             // arena cells owned (0 for constants)
};

inline double scope_log2(const Network &net, const Bits &b) {
    double s = 0;
    b.for_each([&](int v) { s += net.log2card[v]; });
    return s;
}

// SURVEY section 8(d) byte model of an elimination order over factor scopes.
double simulate(const Network &net, std::vector<Bits> f, const std::vector<int32_t> &order, double abort_above) {
    double bytes = 0;
    for (int32_t x : order) {
        Bits u;
        u.nw = net.nw;
        double in = 0;
        size_t k = 0;
        for (size_t i = 0; i < f.size(); ++i) {
            if (f[i].test(x)) {
                u.or_(f[i]);
                in += std::exp2(scope_log2(net, f[i]));
            } else {
                if (k != i) f[k] = f[i];
                ++k;
            }
        }
        f.resize(k);
        u.clr(x);
        bytes += 8.0 * (in + std::exp2(scope_log2(net, u)));
        if (bytes > abort_above) return bytes;
        f.push_back(u);
    }
    Bits u;
    u.nw = net.nw;
    double in = 0;
    for (auto &s : f) {
        u.or_(s);
        in += std::exp2(scope_log2(net, s));
    }
    bytes += 8.0 * (in + std::exp2(scope_log2(net, u)));
    return bytes;
}

// greedy min-weight on the interaction graph (weight = size of the factor the elimination creates)
std::vector<int32_t> greedy_min_weight(const Network &net, const std::vector<Bits> &f, const Bits &hidden,
                                       bool fill) {
    int n = net.n_vars;
    std::vector<Bits> adj(n);
    for (auto &a : adj) a.nw = net.nw;
    for (auto &s : f) s.for_each([&](int v) { adj[v].or_(s); });
    for (int v = 0; v < n; ++v) adj[v].clr(v);
    std::vector<int32_t> hid;
    hidden.for_each([&](int v) { hid.push_back(v); });
    std::vector<double> w(n, 0);
    auto weight = [&](int x) {
        double s = scope_log2(net, adj[x]);
        if (!fill) return s;
        // weighted min-fill flavour: created factor size minus what the neighbours already share
        double removed = 0;
        adj[x].for_each([&](int y) {
            Bits t = adj[x];
            t.andnot(adj[y]);
            t.clr(y);
            removed += t.count();
        });
        return removed * 64.0 + s;  // primary: number of fill edges, secondary: size
    };
    for (int x : hid) w[x] = weight(x);
    std::vector<int32_t> order;
    std::vector<char> alive(n, 0);
    for (int x : hid) alive[x] = 1;
    for (size_t it = 0; it < hid.size(); ++it) {
        int best = -1;
        for (int x : hid)
            if (alive[x]) {
                if (best < 0 || w[x] < w[best] - 1e-12 ||
                    (std::fabs(w[x] - w[best]) <= 1e-12 &&
                     (net.depth[x] < net.depth[best] || (net.depth[x] == net.depth[best] && x < best))))
                    best = x;
            }
        order.push_back(best);
        alive[best] = 0;
        Bits nb = adj[best];
        nb.for_each([&](int y) {
            adj[y].or_(nb);
            adj[y].clr(best);
            adj[y].clr(y);
        });
        nb.for_each([&](int y) { if (alive[y]) w[y] = weight(y); });
        if (fill)  // second-ring weights change too
            nb.for_each([&](int y) { adj[y].for_each([&](int z) { if (alive[z]) w[z] = weight(z); }); });
    }
    return order;
}

struct Arena {
    std::vector<std::pair<int64_t, int64_t>> free_;  // (offset, size), sorted by offset
    int64_t top = 0;
    int64_t alloc(int64_t n) {
        n = (n + 1) & ~int64_t(1);  // keep 16-byte alignment
        for (size_t i = 0; i < free_.size(); ++i)
            if (free_[i].second >= n) {
                int64_t o = free_[i].first;
                free_[i].first += n;
                free_[i].second -= n;
                if (!free_[i].second) free_.erase(free_.begin() + i);
                return o;
            }
        // extend: if the last free block touches the top, grow it
        if (!free_.empty() && free_.back().first + free_.back().second == top) {
            int64_t o = free_.back().first;
            top = o + n;
            free_.pop_back();
            return o;
        }
        int64_t o = top;
        top += n;
        return o;
    }
    void release(int64_t o, int64_t n) {
        n = (n + 1) & ~int64_t(1);
        auto it = std::lower_bound(free_.begin(), free_.end(), std::make_pair(o, int64_t(0)));
        it = free_.insert(it, {o, n});
        size_t i = it - free_.begin();
        if (i + 1 < free_.size() && free_[i].first + free_[i].second == free_[i + 1].first) {
            free_[i].second += free_[i + 1].second;
            free_.erase(free_.begin() + i + 1);
        }
        if (i > 0 && free_[i - 1].first + free_[i - 1].second == free_[i].first) {
            free_[i - 1].second += free_[i].second;
            free_.erase(free_.begin() + i);
        }
    }
};

struct Emitter {
    const Network &net;
    std::vector<uint32_t> &prog;
    PlanStats &st;
    Arena arena;
    std::vector<double> key;  // layout key per variable: larger = lives longer = faster axis
    std::string err;

    void header(uint32_t *w, uint32_t kind, int n_in, int ma, int mlo, int cx, bool final_, int64_t lo, int64_t hi,
                uint64_t out_off, int words) {
        w[0] = kind | ((uint32_t)n_in << 8) | ((uint32_t)ma << 16) | ((uint32_t)mlo << 24);
        w[1] = (uint32_t)cx | ((final_ ? kFlagFinal : 0u) << 16);
        w[2] = (uint32_t)lo;
        w[3] = (uint32_t)hi;
        w[4] = (uint32_t)(out_off & 0xffffffffu);
        w[5] = (uint32_t)(out_off >> 32);
        w[6] = (uint32_t)words;
        w[7] = w[8] = w[9] = 0;
    }

    // GENERIC encoding: iteration space = output cells
    void emit_generic(const std::vector<PF> &ins, const std::vector<std::vector<int64_t>> &s, const std::vector<int64_t> &xs,
                      const PF &out, int na, int64_t cells, int cx, bool final_) {
        const int n_in = (int)ins.size();
        int nlo = 0;
        int64_t lo = 1;
        const int64_t lomax = n_in <= 3 ? kLoMax : kLoTarget;  // kernel: 2 cells per lane up to 3 inputs, else 1
        while (nlo < na && lo < kLoTarget && lo * net.card[out.vars[nlo]] <= lomax) lo *= net.card[out.vars[nlo++]];
        // merge adjacent axes that are contiguous in every input (the output is dense by construction)
        std::vector<uint32_t> mcard;
        std::vector<std::vector<int64_t>> ms(n_in);
        int mlo = 0;
        for (int a = 0; a < na; ++a) {
            bool merge = !mcard.empty() && a != nlo;
            if (merge)
                for (int j = 0; j < n_in && merge; ++j) merge = s[j][a] == ms[j].back() * (int64_t)mcard.back();
            uint32_t c = (uint32_t)net.card[out.vars[a]];
            if (merge && (uint64_t)mcard.back() * c < (1u << 30)) {
                mcard.back() *= c;
            } else {
                mcard.push_back(c);
                for (int j = 0; j < n_in; ++j) ms[j].push_back(s[j][a]);
                if (a < nlo) ++mlo;
            }
        }
        const int ma = (int)mcard.size();
        if (ma > kMaxAxes) { err = "a step has more than " + std::to_string(kMaxAxes) + " axes"; return; }
        const size_t base = prog.size();
        const int words = kHdrWords + 3 * n_in + ma + n_in * ma;
        prog.resize(base + words);
        uint32_t *w = prog.data() + base;
        header(w, kKindGeneric, n_in, ma, mlo, cx, final_, lo, cells / lo, out.off, words);
        uint32_t *p = w + kHdrWords;
        for (int j = 0; j < n_in; ++j) {
            *p++ = (uint32_t)(ins[j].off & 0xffffffffu);
            *p++ = (uint32_t)(ins[j].off >> 32);
            *p++ = (uint32_t)(int32_t)xs[j];
        }
        for (int a = 0; a < ma; ++a) *p++ = mcard[a];
        for (int j = 0; j < n_in; ++j)
            for (int a = 0; a < ma; ++a) *p++ = (uint32_t)(int32_t)ms[j][a];
    }

    // FIBER encoding (see planner.h); returns false when the step does not fit the form
    bool emit_fiber(const std::vector<PF> &ins, const std::vector<std::vector<int64_t>> &s, const std::vector<int64_t> &xs,
                    const PF &out, int na, int64_t cells, int cx) {
        std::vector<int> big, small;
        for (int j = 0; j < (int)ins.size(); ++j) (ins[j].cells > net.small_cells ? big : small).push_back(j);
        if (big.empty() || big.size() > 2 || (int)small.size() > kMaxSmall || cx > 16) return false;
        // N axes: no big input depends on them; keep at most kMaxNC combinations (fastest axes first)
        std::vector<int> naxes, raxes;
        int64_t NC = 1;
        for (int a = 0; a < na; ++a) {
            bool free_ = true;
            for (int b : big) free_ = free_ && s[b][a] == 0;
            const int c = net.card[out.vars[a]];
            if (free_ && NC * c <= kMaxNC) { naxes.push_back(a); NC *= c; }
            else raxes.push_back(a);
        }
        // ctrl axes: R axes a small input depends on
        std::vector<int> ctrl;
        int64_t T = NC * cx;
        for (int a : raxes) {
            bool dep = false;
            for (int j : small) dep = dep || s[j][a] != 0;
            if (dep) { ctrl.push_back(a); T *= net.card[out.vars[a]]; if (T > kMaxT) return false; }
        }
        const int nT = (int)naxes.size() + (int)ctrl.size();
        if (naxes.size() > 15 || ctrl.size() > 15) return false;
        // R-axis tables (unmerged)
        const int nr = (int)raxes.size();
        std::vector<int64_t> rcard(nr), rost(nr), rtst(nr, 0);
        std::vector<std::vector<int64_t>> rb(big.size(), std::vector<int64_t>(nr));
        {
            int64_t tmul = NC * cx;
            for (int i = 0; i < nr; ++i) {
                const int a = raxes[i];
                rcard[i] = net.card[out.vars[a]];
                rost[i] = out.strides[a];
                for (size_t b = 0; b < big.size(); ++b) rb[b][i] = s[big[b]][a];
                if (std::find(ctrl.begin(), ctrl.end(), a) != ctrl.end()) { rtst[i] = tmul; tmul *= rcard[i]; }
            }
        }
        int nlo = 0;
        int64_t lo = 1;
        while (nlo < nr && lo < kLoTarget && lo * rcard[nlo] <= kFiberLoMax) lo *= rcard[nlo++];
        int64_t rcells = 1;
        for (int i = 0; i < nr; ++i) rcells *= rcard[i];
        // merge adjacent R axes contiguous in the output, in T and in every big input
        std::vector<int64_t> mc, mo, mt;
        std::vector<std::vector<int64_t>> mb(big.size());
        int mlo = 0;
        for (int i = 0; i < nr; ++i) {
            bool merge = !mc.empty() && i != nlo && mo.back() * mc.back() == rost[i] && mt.back() * mc.back() == rtst[i] &&
                         mc.back() * rcard[i] < (1 << 30);
            for (size_t b = 0; b < big.size() && merge; ++b) merge = mb[b].back() * mc.back() == rb[b][i];
            if (merge) {
                mc.back() *= rcard[i];
            } else {
                mc.push_back(rcard[i]);
                mo.push_back(rost[i]);
                mt.push_back(rtst[i]);
                for (size_t b = 0; b < big.size(); ++b) mb[b].push_back(rb[b][i]);
                if (i < nlo) ++mlo;
            }
        }
        const int ma = (int)mc.size();
        if (ma > kMaxAxes) return false;
        const int nb = (int)big.size(), ns = (int)small.size();
        const int words = kHdrWords + 3 * nb + ns * (3 + nT) + nT + (int)NC + 3 * ma + nb * ma;
        if (words > kMaxStepWords) return false;
        const size_t base = prog.size();
        prog.resize(base + words);
        uint32_t *w = prog.data() + base;
        header(w, kKindFiber, nb + ns, ma, mlo, cx, false, lo, rcells / lo, out.off, words);
        w[7] = (uint32_t)nb | ((uint32_t)ns << 4) | ((uint32_t)naxes.size() << 8) | ((uint32_t)ctrl.size() << 12) | ((uint32_t)NC << 16);
        w[8] = (uint32_t)T;
        uint32_t *p = w + kHdrWords;
        for (int b : big) {
            *p++ = (uint32_t)(ins[b].off & 0xffffffffu);
            *p++ = (uint32_t)(ins[b].off >> 32);
            *p++ = (uint32_t)(int32_t)xs[b];
        }
        for (int j : small) {
            *p++ = (uint32_t)(ins[j].off & 0xffffffffu);
            *p++ = (uint32_t)(ins[j].off >> 32);
            *p++ = (uint32_t)(int32_t)xs[j];
            for (int a : naxes) *p++ = (uint32_t)(int32_t)s[j][a];
            for (int a : ctrl) *p++ = (uint32_t)(int32_t)s[j][a];
        }
        for (int a : naxes) *p++ = (uint32_t)net.card[out.vars[a]];
        for (int a : ctrl) *p++ = (uint32_t)net.card[out.vars[a]];
        for (int64_t n = 0; n < NC; ++n) {
            int64_t r = n, off = 0;
            for (int a : naxes) { off += (r % net.card[out.vars[a]]) * out.strides[a]; r /= net.card[out.vars[a]]; }
            *p++ = (uint32_t)off;
        }
        for (int a = 0; a < ma; ++a) { *p++ = (uint32_t)mc[a]; *p++ = (uint32_t)mo[a]; *p++ = (uint32_t)mt[a]; }
        for (int b = 0; b < nb; ++b)
            for (int a = 0; a < ma; ++a) *p++ = (uint32_t)(int32_t)mb[b][a];
        (void)cells;
        return true;
    }

    // Emit one step: multiply `ins`, sum out x (x < 0: product only).  Returns the new factor.
    PF emit(const std::vector<PF> &ins, int x, bool final_, int64_t final_off) {
        PF out;
        out.scope.nw = net.nw;
        for (auto &f : ins) out.scope.or_(f.scope);
        double prod_log2 = scope_log2(net, out.scope);
        if (x >= 0) out.scope.clr(x);
        out.scope.for_each([&](int v) { out.vars.push_back(v); });
        std::sort(out.vars.begin(), out.vars.end(), [&](int a, int b) { return key[a] > key[b] || (key[a] == key[b] && a < b); });
        int na = (int)out.vars.size();
        int64_t cells = 1;
        out.strides.resize(na);
        for (int a = 0; a < na; ++a) {
            out.strides[a] = cells;
            cells *= net.card[out.vars[a]];
            if (cells >= (1ll << 31)) { err = "an intermediate factor has >= 2^31 cells"; return out; }
        }
        out.cells = cells;
        int n_in = (int)ins.size();
        // per-input strides along the output axes, and along x
        std::vector<std::vector<int64_t>> s(n_in, std::vector<int64_t>(na, 0));
        std::vector<int64_t> xs(n_in, 0);
        for (int j = 0; j < n_in; ++j)
            for (size_t k = 0; k < ins[j].vars.size(); ++k) {
                int v = ins[j].vars[k];
                if (v == x) { xs[j] = ins[j].strides[k]; continue; }
                int a = (int)(std::find(out.vars.begin(), out.vars.end(), v) - out.vars.begin());
                s[j][a] = ins[j].strides[k];
            }
        int cx = x >= 0 ? net.card[x] : 1;
        if (final_) {
            out.off = (uint64_t)final_off;
            out.alloc = 0;
        } else {
            out.off = (uint64_t)arena.alloc(cells);
            out.alloc = cells;
        }
        double in_cells = 0;
        for (int j = 0; j < n_in; ++j) in_cells += (double)ins[j].cells;
        if (!(!final_ && emit_fiber(ins, s, xs, out, na, cells, cx)))
            emit_generic(ins, s, xs, out, na, cells, cx, final_);
        if (!err.empty()) return out;
        st.alg_bytes += 8.0 * (in_cells + (double)cells);
        double pc = std::exp2(prod_log2);
        st.alg_flops += n_in * pc;
        st.max_step_cells = std::max(st.max_step_cells, pc);
        st.n_steps += 1;
        for (auto &f : ins)
            if (f.alloc) arena.release((int64_t)f.off, f.alloc);
        return out;
    }

    // multiply/eliminate with at most kMaxIn inputs per step
    PF emit_limited(std::vector<PF> ins, int x, bool final_, int64_t final_off) {
        while ((int)ins.size() > kMaxIn && err.empty()) {
            std::sort(ins.begin(), ins.end(), [](const PF &a, const PF &b) { return a.cells < b.cells; });
            std::vector<PF> head(ins.begin(), ins.begin() + kMaxIn);
            PF prod = emit(head, -1, false, 0);
            ins.erase(ins.begin(), ins.begin() + kMaxIn);
            ins.push_back(prod);
        }
        if (!err.empty()) return PF{};
        return emit(ins, x, final_, final_off);
    }
};

}  // namespace

std::string plan_request(const Network &net, const Request &rq, std::vector<uint32_t> &prog, PlanStats &st) {
    // relevant = query | event | ancestors(...)  (bayes_net.py:763-765); hidden = relevant - query - event (766)
    Bits rel, qb, eb;
    rel.nw = qb.nw = eb.nw = net.nw;
    for (int i = 0; i < rq.nq; ++i) { qb.set(rq.qvars[i]); rel.set(rq.qvars[i]); rel.or_(net.anc[rq.qvars[i]]); }
    for (int i = 0; i < rq.ne; ++i) { eb.set(rq.evars[i]); rel.set(rq.evars[i]); rel.or_(net.anc[rq.evars[i]]); }
    Bits hidden = rel;
    hidden.andnot(qb);
    hidden.andnot(eb);
    std::vector<int32_t> ecode(net.n_vars, 0);
    if (rq.ecodes)
        for (int i = 0; i < rq.ne; ++i) ecode[rq.evars[i]] = rq.ecodes[i];

    // factors = evidence-sliced CPTs of the relevant nodes (bayes_net.py:768-776): the evidence axis is
    // not copied away but folded into the base offset (stride 0 afterwards)
    std::vector<PF> fs;
    std::vector<Bits> scopes;
    rel.for_each([&](int v) {
        PF f;
        f.scope.nw = net.nw;
        uint64_t off = (uint64_t)net.pool_off[v];
        int64_t cells = 1;
        for (size_t k = 0; k < net.scope[v].size(); ++k) {
            int u = net.scope[v][k];
            if (eb.test(u)) {
                off += (uint64_t)(net.cstride[v][k] * ecode[u]);
            } else {
                f.scope.set(u);
                f.vars.push_back(u);
                f.strides.push_back(net.cstride[v][k]);
                cells *= net.card[u];
            }
        }
        f.off = off | kConstFlag;
        f.cells = cells;
        fs.push_back(f);
        scopes.push_back(f.scope);
    });

    // candidate elimination orders, cheapest by the byte model wins
    std::vector<int32_t> hid;
    hidden.for_each([&](int v) { hid.push_back(v); });
    std::vector<int32_t> best;
    double best_cost = std::numeric_limits<double>::infinity();
    auto consider = [&](std::vector<int32_t> &&o) {
        double c = simulate(net, scopes, o, best_cost);
        if (c < best_cost) { best_cost = c; best = std::move(o); }
    };
    if (!hid.empty()) {
        int qdepth = std::numeric_limits<int>::max();
        for (int i = 0; i < rq.nq; ++i) qdepth = std::min(qdepth, (int)net.depth[rq.qvars[i]]);
        auto sorted_by = [&](auto keyfn) {
            std::vector<int32_t> o = hid;
            std::stable_sort(o.begin(), o.end(), [&](int a, int b) { return keyfn(a) < keyfn(b); });
            return o;
        };
        // "meet": sweep down from the roots to the query's depth, then up from the leaves
        consider(sorted_by([&](int v) { return net.depth[v] < qdepth ? (double)net.depth[v] : 1e6 - net.depth[v]; }));
        consider(sorted_by([&](int v) { return (double)net.depth[v]; }));   // topological sweep
        consider(sorted_by([&](int v) { return -(double)net.depth[v]; }));  // reverse sweep
        for (auto &h : net.hints) consider(sorted_by([&](int v) { return (double)h[v]; }));
        consider(greedy_min_weight(net, scopes, hidden, false));
        consider(greedy_min_weight(net, scopes, hidden, true));
    }

    Emitter em{net, prog, st, Arena{}, std::vector<double>(net.n_vars, 0.0), ""};
    for (size_t i = 0; i < best.size(); ++i) em.key[best[i]] = (double)i;
    for (int i = 0; i < rq.nq; ++i) em.key[rq.qvars[i]] = 1e9 + i;

    size_t count_pos = prog.size();
    prog.push_back(0);
    double steps0 = st.n_steps;
    for (int32_t x : best) {
        std::vector<PF> ins;
        // pop every factor mentioning x (bayes_net.py:780-784)
        size_t k = 0;
        for (size_t i = 0; i < fs.size(); ++i) {
            if (fs[i].scope.test(x)) ins.push_back(fs[i]);
            else { if (k != i) fs[k] = fs[i]; ++k; }
        }
        fs.resize(k);
        PF out = em.emit_limited(ins, x, false, 0);  // pointwise_mul + sum_out (785)
        if (!em.err.empty()) return em.err;
        fs.push_back(out);
    }
    // posterior = pointwise_mul(factors) / sum (bayes_net.py:789-790), written in the caller's
    // query order (C-order, last query variable fastest)
    st.out_cells = 1;
    for (int i = 0; i < rq.nq; ++i) st.out_cells *= net.card[rq.qvars[i]];
    em.emit_limited(fs, -1, true, rq.out_off);
    if (!em.err.empty()) return em.err;
    prog[count_pos] = (uint32_t)(st.n_steps - steps0);
    st.arena_cells = std::max(st.arena_cells, em.arena.top);
    return "";
}

}  // namespace mibn
