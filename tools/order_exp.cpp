// Order-search experiment (CPU only, round 5): per request of the C3 stream the byte-model cost of EVERY candidate order (meet, reverse
// topological, the hint lists) and of greedy min-fill - who wins how often, what each piece costs in time, and what cheaper searches
// (fewer candidates, a higher min-fill threshold, min-degree instead of min-fill) would cost in bytes.  Results: profiles/NOTES_r05.md.
//   g++ -O3 -mpopcnt -std=c++17 tools/order_exp.cpp sorobn_amd/csrc/planner.cpp -lpthread -o /tmp/order_exp && /tmp/order_exp; NEV=16 /tmp/order_exp
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>
#include "../sorobn_amd/csrc/planner.h"
using namespace mibn;
static double now() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

static bool greedy_minweight(const OrderNet &net, OrderScratch &S, const B2 &hidden, int mode) {
    B2 *adj = S.adj; const B2 rel = S.rel;
    b2_each(rel, [&](int v) { adj[v] = B2{}; });
    b2_each(rel, [&](int i) { const B2 sc = S.f[i]; b2_each(sc, [&](int v) { adj[v].a |= sc.a; adj[v].b |= sc.b; }); });
    b2_each(rel, [&](int v) { adj[v].clr(v); });
    B2 alive = hidden; int total = b2_count(hidden); S.n_cand = 0;
    for (int it = 0; it < total; ++it) {
        int best = -1; long bk = 0;
        b2_each(alive, [&](int x) {
            long k;
            const int deg = b2_count(adj[x]);
            if (mode == 0) k = ((long)deg << 16) | (net.depth[x] << 8) | x;            // min degree (= min weight for uniform cards)
            else { // min degree, ties by fewer alive neighbours... (mode 1: prefer deeper)
                k = ((long)deg << 16) | ((255 - net.depth[x]) << 8) | x; }
            if (best < 0 || k < bk) { best = x; bk = k; }
        });
        S.cand[S.n_cand++] = (uint8_t)best; alive.clr(best);
        const B2 nb = adj[best];
        b2_each(nb, [&](int y) { adj[y].a |= nb.a; adj[y].b |= nb.b; adj[y].clr(best); adj[y].clr(y); });
    }
    return true;
}

int main(int argc, char **argv) {
    const int R = 10, C = 10, K = 4, n = 100;
    const int64_t B = 8192;
    const int NE = getenv("NEV") ? atoi(getenv("NEV")) : 4;
    std::vector<int32_t> card(n, K), scope_vars; std::vector<int64_t> scope_off{0}, value_off{0}; std::vector<double> values;
    std::mt19937_64 rng(1); std::uniform_real_distribution<double> U(0.1, 1.0);
    for (int v = 0; v < n; ++v) { int r = v / C, c = v % C; if (r) scope_vars.push_back(v - C); if (c) scope_vars.push_back(v - 1); scope_vars.push_back(v);
        scope_off.push_back(scope_vars.size()); int64_t cells = K; if (r) cells *= K; if (c) cells *= K; for (int64_t i = 0; i < cells; ++i) values.push_back(U(rng)); value_off.push_back(values.size()); }
    Network net; net.set(n, card.data(), scope_off.data(), scope_vars.data(), value_off.data(), values.data());
    std::vector<int32_t> hint(n); for (int v = 0; v < n; ++v) hint[v] = v; net.set_hints(1, hint.data());
    OrderNet on = net.order_view();
    printf("n_hints %d chain_weight %g minfill_above %g\n", on.n_hints, on.chain_weight, on.minfill_above);
    static OrderScratch S;
    const int NC = 2 + on.n_hints + 3;
    std::vector<double> cost(B * NC, 0.0); std::vector<double> t(NC + 2, 0.0);
    for (int64_t b = 0; b < B; ++b) {
        int pick[40]; for (int k = 0; k < NE + 1;) { int v = rng() % n; bool dup = false; for (int j = 0; j < k; ++j) dup |= pick[j] == v; if (!dup) pick[k++] = v; }
        int32_t q = pick[0]; int32_t ev[40]; for (int k = 0; k < NE; ++k) { ev[k] = pick[1 + k]; rng(); }
        B2 rel, hidden; double t0 = now();
        order_prepare(on, S, 1, &q, NE, ev, false, rel, hidden);
        t[NC] += now() - t0;
        int qd = on.depth[q];
        const double inf = __builtin_inf();
        for (int c = 0; c < NC; ++c) {
            t0 = now();
            bool ok = true;
            if (c < 2) order_sweep(on, S, hidden, qd, c);
            else if (c < 2 + on.n_hints) { const int32_t *s = on.hint_sorted + (int64_t)(c - 2) * on.n_vars; S.n_cand = 0; for (int i = 0; i < on.n_vars; ++i) if (hidden.test(s[i])) S.cand[S.n_cand++] = s[i]; }
            else if (c == NC - 3) { ok = order_greedy(on, S, hidden, inf); t[NC + 1] += now() - t0; t0 = now(); }
            else { greedy_minweight(on, S, hidden, c - (NC - 2)); }
            cost[b * NC + c] = ok ? order_simulate(on, S, S.cand, S.n_cand, inf) : inf;
            t[c] += now() - t0;
        }
    }
    for (int c = 0; c < NC; ++c) printf("cand %d: sim+gen %.2f us/request\n", c, t[c] / B);
    printf("prepare %.2f us, greedy itself %.2f us\n", t[NC] / B, t[NC + 1] / B);
    // strategies
    auto eval = [&](const char *name, std::vector<int> sweeps, bool greedy, double thr, int gi = -1) { if (gi < 0) gi = NC - 3;
        double tot = 0, ng = 0; std::vector<int> wins(NC, 0);
        for (int64_t b = 0; b < B; ++b) { double best = 1e300; int w = -1; for (int c : sweeps) if (cost[b * NC + c] < best) { best = cost[b * NC + c]; w = c; }
            if (greedy && best > thr) { ng++; if (cost[b * NC + gi] < best) { best = cost[b * NC + gi]; w = gi; } } tot += best; wins[w]++; }
        printf("%-40s mean weighted cost %.3f MB  greedy runs %.1f%%  wins:", name, tot / B / 1e6, 100 * ng / B); for (int c = 0; c < NC; ++c) printf(" %d", wins[c]); printf("\n");
    };
    const double thr = on.minfill_above * on.chain_weight;
    eval("current (4 sweeps + greedy>thr)", {0, 1, 2, 3}, true, thr);
    eval("all + mindeg(shallow first) > thr", {0, 1, 2, 3}, true, thr, NC - 2);
    eval("all + mindeg(deep first) > thr", {0, 1, 2, 3}, true, thr, NC - 1);
    eval("all + mindeg always", {0, 1, 2, 3}, true, 0, NC - 2);
    eval("no rev", {0, 2, 3}, true, thr);
    eval("no meet", {1, 2, 3}, true, thr);
    eval("meet + hint2", {0, 2}, true, thr);
    eval("meet + hint3", {0, 3}, true, thr);
    eval("meet only", {0}, true, thr);
    eval("greedy always + meet", {0}, true, 0);
    eval("greedy always, no sweeps at all", {}, true, -1);
    eval("greedy always, all", {0, 1, 2, 3}, true, 0);
    eval("no greedy", {0, 1, 2, 3}, false, 0);
    eval("all, greedy > 2 thr", {0, 1, 2, 3}, true, 2 * thr);
    eval("all, greedy > 4 thr", {0, 1, 2, 3}, true, 4 * thr);
    eval("all, greedy > thr/2", {0, 1, 2, 3}, true, thr / 2);
    return 0;
}
