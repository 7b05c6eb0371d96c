// How far is the planner's elimination order from the best one?  (CPU only; profiles/NOTES_r06.md, "order quality".)
// For every request of a C3 stream: the planner's own order (order_search: sweeps, hints, min-fill ranked by the section-8(d) byte
// model), emitted (emit_core.h) = the algorithmic bytes the VE kernels would move.  Then a hill climb on that order - move one
// variable to another position, keep the move if the EMITTED bytes drop - as a probe of what a better search could find, and the
// modelled cost (order_simulate) beside it to see whether the model ranks the way the emitter does.
//   g++ -O2 -mpopcnt -std=c++17 tools/order_quality.cpp sorobn_amd/csrc/planner.cpp -lpthread -o /tmp/order_quality && /tmp/order_quality [requests] [evidence nodes] [tries]
#include <time.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <string>
#include <vector>

#include "../sorobn_amd/csrc/planner.h"
using namespace mibn;

int main(int argc, char **argv) {
    const int64_t B = argc > 1 ? atoll(argv[1]) : 300;
    const int NE = argc > 2 ? atoi(argv[2]) : 4;
    const int TRIES = argc > 3 ? atoi(argv[3]) : 300;
    // (ROWS / COLS / CARD: other grids than C3's 10 x 10 four-state one - does what was tuned there hold elsewhere?)
    const int R = std::getenv("ROWS") ? atoi(std::getenv("ROWS")) : 10, C = std::getenv("COLS") ? atoi(std::getenv("COLS")) : 10, K = std::getenv("CARD") ? atoi(std::getenv("CARD")) : 4, n = R * C;
    std::vector<int32_t> card(n, K), scope_vars;
    std::vector<int64_t> scope_off{0}, value_off{0};
    std::vector<double> values;
    std::mt19937_64 rng(1);
    std::uniform_real_distribution<double> U(0.1, 1.0);
    for (int v = 0; v < n; ++v) {
        const int r = v / C, c = v % C;
        if (r) scope_vars.push_back(v - C);
        if (c) scope_vars.push_back(v - 1);
        scope_vars.push_back(v);
        scope_off.push_back((int64_t)scope_vars.size());
        int64_t cells = K;
        if (r) cells *= K;
        if (c) cells *= K;
        for (int64_t i = 0; i < cells; ++i) values.push_back(U(rng));
        value_off.push_back((int64_t)values.size());
    }
    Network net;
    std::string e = net.set(n, card.data(), scope_off.data(), scope_vars.data(), value_off.data(), values.data());
    if (!e.empty()) { std::fprintf(stderr, "%s\n", e.c_str()); return 1; }
    const EmitNet en = net.emit_view();
    const OrderNet on = net.order_view();
    std::vector<char> slice(emit_scratch_bytes(n) + 64);
    std::vector<uint32_t> slot(1 << 16);
    OrderScratch *os = new OrderScratch;
    char *base = slice.data() + ((64 - (reinterpret_cast<uintptr_t>(slice.data()) & 63)) & 63);
    int32_t qv[1], ev[40], ec[40];
    auto emitted = [&](const uint8_t *order, int n_order, double *steps = nullptr) -> double {
        EmitScratch S;
        emit_scratch_carve(S, base, n);
        if (emit_begin(en, S, 1, qv, NE, ev, ec, false)) return -1;
        EmitBuf buf;
        buf.data = slot.data();
        buf.cap = slot.size();
        EmitStats st;
        if (emit_run(en, S, buf, st, nullptr, 1, qv, 0, order, n_order)) return -1;
        if (steps) *steps = st.n_steps;
        return st.alg_bytes;
    };
    std::vector<int32_t> hint(n);
    for (int v = 0; v < n; ++v) hint[v] = v;
    if (!std::getenv("NO_HINT")) net.set_hints(1, hint.data());
    const OrderNet on2 = net.order_view();
    if (std::getenv("PLANNER")) {
        // the planner itself (plan_request) at order_effort 0 and 1: bytes per request and planning time on this core
        std::vector<std::vector<int32_t>> reqs;
        for (int64_t b = 0; b < B; ++b) {
            int pick[40];
            for (int k = 0; k < NE + 1;) {
                const int v = (int)(rng() % n);
                bool dup = false;
                for (int j = 0; j < k; ++j) dup = dup || pick[j] == v;
                if (!dup) pick[k++] = v;
            }
            reqs.emplace_back(pick, pick + NE + 1);
        }
        double bytes[2] = {0, 0}, us[2] = {0, 0};
        for (int eff = 0; eff < 2; ++eff) {
            net.order_effort = eff;
            if (std::getenv("SECOND_ABOVE")) net.second_above = atof(std::getenv("SECOND_ABOVE"));
            if (std::getenv("ORDER_WEIGHTS")) net.order_weights = atoi(std::getenv("ORDER_WEIGHTS"));  // (k > 1: a single-table elimination counts 1 / k of its bytes in the model)
            std::vector<uint32_t> prog;
            timespec t0, t1;
            clock_gettime(CLOCK_MONOTONIC, &t0);
            for (auto &r : reqs) {
                for (int k = 0; k < NE; ++k) { ev[k] = r[1 + k]; ec[k] = 0; }
                qv[0] = r[0];
                Request rq;
                rq.nq = 1; rq.qvars = qv; rq.ne = NE; rq.evars = ev; rq.ecodes = ec;
                prog.clear();
                PlanStats st;
                const std::string err = plan_request(net, rq, prog, st);
                if (!err.empty()) { std::fprintf(stderr, "%s\n", err.c_str()); return 1; }
                bytes[eff] += st.alg_bytes;
            }
            clock_gettime(CLOCK_MONOTONIC, &t1);
            us[eff] = ((t1.tv_sec - t0.tv_sec) * 1e9 + (t1.tv_nsec - t0.tv_nsec)) / 1e3 / (double)B;
        }
        std::printf("%lld requests, %d evidence nodes: order_effort 0: %.3f MB per request, %.1f us per request planned; order_effort 1: %.3f MB (%.1f %% less), %.1f us\n", (long long)B, NE,
                    bytes[0] / B / 1e6, us[0], bytes[1] / B / 1e6, 100.0 * (1 - bytes[1] / bytes[0]), us[1]);
        return 0;
    }
    if (std::getenv("CANDIDATES")) {
        // every candidate of the search emitted: what the model chose against what the emitter would have chosen - and what further
        // generators (EXTRA=1) would add: the meet sweep at other depths, a min-fill prefix (the eliminations that create small factors)
        // followed by the meet sweep of the rest
        const bool extra = std::getenv("EXTRA") != nullptr;
        std::vector<std::string> names = {"meet", "reverse", "row-major hint", "min-fill"};
        if (extra) for (const char *x : {"meet -2", "meet -1", "meet +1", "meet +2", "min-fill <= 3 then meet", "min-fill <= 4 then meet", "min-fill <= 5 then meet", "min-fill <= 6 then meet", "min-fill <= 7 then meet"}) names.push_back(x);
        // LAZY=1: the meet sweep, but whenever a vertex of the current interaction graph has at most d neighbours it goes first (the opening rule applied all along)
        const bool lazy = extra && std::getenv("LAZY") != nullptr;
        if (lazy) for (const char *x : {"meet, degree <= 2 first", "meet, degree <= 3 first", "meet, degree <= 4 first", "meet, degree <= 5 first"}) names.push_back(x);
        const int NC = (int)names.size();
        double chosen = 0, best_e = 0, chosen4 = 0, topk[4] = {0, 0, 0, 0}, extra_emits[4] = {0, 0, 0, 0}, divers = 0;
        std::vector<double> per(NC, 0.0);
        std::vector<int64_t> wins_model(NC, 0), wins_emit(NC, 0);
        std::vector<double> costs;
        for (int64_t b = 0; b < B; ++b) {
            int pick[40];
            for (int k = 0; k < NE + 1;) {
                const int v = (int)(rng() % n);
                bool dup = false;
                for (int j = 0; j < k; ++j) dup = dup || pick[j] == v;
                if (!dup) pick[k++] = v;
            }
            qv[0] = pick[0];
            for (int k = 0; k < NE; ++k) { ev[k] = pick[1 + k]; ec[k] = 0; }
            B2 rel, hidden;
            order_prepare(on2, *os, 1, qv, NE, ev, false, rel, hidden);
            if (!hidden.any()) continue;
            std::vector<double> em(NC), mo(NC);
            order_greedy(on2, *os, hidden, __builtin_inf());
            const std::vector<uint8_t> G(os->cand, os->cand + os->n_cand);
            // widths of min-fill's eliminations (replay on scopes)
            std::vector<int> gw;
            {
                std::vector<B2> fs;
                B2 eb;
                for (int k = 0; k < NE; ++k) eb.set(ev[k]);
                b2_each(rel, [&](int v) { B2 sc = net.scope2[v]; sc.a &= ~eb.a; sc.b &= ~eb.b; fs.push_back(sc); });
                std::vector<char> alive(fs.size(), 1);
                for (uint8_t x : G) {
                    B2 u;
                    for (size_t i = 0; i < fs.size(); ++i)
                        if (alive[i] && fs[i].test(x)) { alive[i] = 0; u.a |= fs[i].a; u.b |= fs[i].b; }
                    u.clr(x);
                    fs.push_back(u);
                    alive.push_back(1);
                    gw.push_back(b2_count(u));
                }
            }
            for (int c = 0; c < NC; ++c) {
                std::vector<uint8_t> o;
                if (c < 2) { order_sweep(on2, *os, hidden, on2.depth[qv[0]], c); o.assign(os->cand, os->cand + os->n_cand); }
                else if (c == 2) { for (int i = 0; i < n; ++i) if (hidden.test(on2.hint_sorted[i])) o.push_back((uint8_t)on2.hint_sorted[i]); }
                else if (c == 3) o = G;
                else if (c < 8) { const int d = std::max(0, on2.depth[qv[0]] + (c < 6 ? c - 6 : c - 5)); order_sweep(on2, *os, hidden, d, 0); o.assign(os->cand, os->cand + os->n_cand); }
                else if (c >= 13) {
                    const int d = c - 13 + 2;
                    // interaction graph of the request (scopes without evidence), eliminated as we go
                    std::vector<B2> adj(n);
                    B2 eb;
                    for (int k = 0; k < NE; ++k) eb.set(ev[k]);
                    b2_each(rel, [&](int v) {
                        B2 sc = net.scope2[v]; sc.a &= ~eb.a; sc.b &= ~eb.b;
                        b2_each(sc, [&](int u) { adj[u].a |= sc.a; adj[u].b |= sc.b; });
                    });
                    for (int v = 0; v < n; ++v) adj[v].clr(v);
                    order_sweep(on2, *os, hidden, on2.depth[qv[0]], 0);
                    const std::vector<uint8_t> sweep(os->cand, os->cand + os->n_cand);
                    B2 alive = hidden;
                    size_t next = 0;
                    while (alive.any()) {
                        int pick_v = -1, pick_deg = 1 << 30;
                        b2_each(alive, [&](int v) { const int dg = b2_count(adj[v]); if (dg <= d && dg < pick_deg) { pick_deg = dg; pick_v = v; } });
                        if (pick_v < 0) {
                            while (!alive.test(sweep[next])) ++next;
                            pick_v = sweep[next];
                        }
                        o.push_back((uint8_t)pick_v);
                        alive.clr(pick_v);
                        const B2 nb = adj[pick_v];
                        b2_each(nb, [&](int y) { adj[y].a |= nb.a; adj[y].b |= nb.b; adj[y].clr(y); adj[y].clr(pick_v); });
                    }
                }
                else {
                    const int w = c - 8 + 3;
                    B2 rest = hidden;
                    for (size_t i = 0; i < G.size() && gw[i] <= w; ++i) { o.push_back(G[i]); rest.clr(G[i]); }
                    order_sweep(on2, *os, rest, on2.depth[qv[0]], 0);
                    o.insert(o.end(), os->cand, os->cand + os->n_cand);
                }
                mo[c] = order_simulate(on2, *os, o.data(), (int)o.size(), __builtin_inf());
                em[c] = emitted(o.data(), (int)o.size());
                if (em[c] < 0) em[c] = 1e30;
                per[c] += em[c] < 1e29 ? em[c] : 0;
            }
            int cm = 0, ce = 0, cm4 = 0;
            for (int c = 1; c < NC; ++c) { if (mo[c] < mo[cm]) cm = c; if (em[c] < em[ce]) ce = c; if (c < 4 && mo[c] < mo[cm4]) cm4 = c; }
            {   // two stages: the model's best k, the emitter among them (only where the model's best is heavy)
                std::vector<int> idx(NC);
                for (int c = 0; c < NC; ++c) idx[c] = c;
                std::sort(idx.begin(), idx.end(), [&](int a_, int b_) { return mo[a_] < mo[b_]; });
                const double heavy = std::getenv("HEAVY_ABOVE") ? atof(std::getenv("HEAVY_ABOVE")) * 1e6 : 0;
                {   // diversity: the runner-up = the model's best of the OTHER family (min-fill and its openings / the sweeps)
                    auto fam = [&](int c) { return c == 3 || c >= 8; };
                    double d = em[idx[0]];
                    if (mo[idx[0]] >= heavy)
                        for (int q = 1; q < NC; ++q)
                            if (fam(idx[q]) != fam(idx[0]) && mo[idx[q]] < 1e300) { d = std::min(d, em[idx[q]]); break; }
                    divers += d;
                }
                for (int k = 1; k <= 4; ++k) {
                    double bestk = em[idx[0]];
                    if (mo[idx[0]] >= heavy) { for (int q = 1; q < k; ++q) bestk = std::min(bestk, em[idx[q]]); extra_emits[k - 1] += k - 1; }
                    topk[k - 1] += bestk;
                }
            }
            chosen += em[cm]; best_e += em[ce]; chosen4 += em[cm4];
            ++wins_model[cm]; ++wins_emit[ce];
            costs.push_back(em[cm]);
        }
        std::printf("%zu requests, %d evidence nodes: today's four candidates by the model %.3f MB per request emitted; all %d by the model %.3f MB (%.1f %% less); all by the emitter %.3f MB (%.1f %% less)\n", costs.size(), NE,
                    chosen4 / costs.size() / 1e6, NC, chosen / costs.size() / 1e6, 100.0 * (1 - chosen / chosen4), best_e / costs.size() / 1e6, 100.0 * (1 - best_e / chosen4));
        for (int k = 1; k <= 4; ++k) std::printf("  the model's best %d, the emitter among them: %.3f MB (%.1f %% less than today), %.2f extra emissions per request\n", k, topk[k - 1] / costs.size() / 1e6, 100.0 * (1 - topk[k - 1] / chosen4), extra_emits[k - 1] / costs.size());
        std::printf("  the model's best and its best of the other family (min-fill and openings / sweeps), the emitter between them: %.3f MB (%.1f %% less than today)\n", divers / costs.size() / 1e6, 100.0 * (1 - divers / chosen4));
        for (int c = 0; c < NC; ++c) std::printf("  %-26s alone %.3f MB; chosen by the model %lld times, by the emitter %lld times\n", names[c].c_str(), per[c] / costs.size() / 1e6, (long long)wins_model[c], (long long)wins_emit[c]);
        std::sort(costs.begin(), costs.end());
        double tot = 0, acc = 0;
        for (double c : costs) tot += c;
        std::printf("  bytes by decile of requests (cheapest first):");
        for (size_t i = 0; i < costs.size(); ++i) { acc += costs[i]; if ((i + 1) % (costs.size() / 10) == 0) std::printf(" %.1f%%", 100.0 * acc / tot); }
        std::printf("   median %.2f MB, 90th percentile %.2f MB, max %.2f MB\n", costs[costs.size() / 2] / 1e6, costs[costs.size() * 9 / 10] / 1e6, costs.back() / 1e6);
        return 0;
    }
    double base_bytes = 0, climbed_bytes = 0, base_model = 0, climbed_model = 0, model_of_best_model = 0, bytes_of_best_model = 0;
    int64_t improved = 0, moves = 0, failed = 0, searched = 0;
    for (int64_t b = 0; b < B; ++b) {
        int pick[40];
        for (int k = 0; k < NE + 1;) {
            const int v = (int)(rng() % n);
            bool dup = false;
            for (int j = 0; j < k; ++j) dup = dup || pick[j] == v;
            if (!dup) pick[k++] = v;
        }
        qv[0] = pick[0];
        for (int k = 0; k < NE; ++k) { ev[k] = pick[1 + k]; ec[k] = 0; }
        order_search(on2, *os, 1, qv, NE, ev, false);
        const int m = os->n_best;
        std::vector<uint8_t> cur(os->best, os->best + m), best_model_order = cur;
        double cur_bytes = emitted(cur.data(), m);
        if (cur_bytes < 0 || m < 3) { ++failed; continue; }
        const double b0 = cur_bytes;
        if (std::getenv("HEAVY_ABOVE") && b0 < atof(std::getenv("HEAVY_ABOVE")) * 1e6) {  // only the heavy requests are searched
            base_bytes += b0; climbed_bytes += b0; bytes_of_best_model += b0;
            continue;
        }
        ++searched;
        double cur_model = order_simulate(on, *os, cur.data(), m, __builtin_inf());
        const double m0 = cur_model;
        double bm = cur_model;  // a second climb, on the model alone (what a better SEARCH with the same model would find)
        for (int t = 0; t < TRIES; ++t) {
            const int i = (int)(rng() % m);
            int j = (int)(rng() % m);
            if (i == j) continue;
            std::vector<uint8_t> cand = cur;
            const uint8_t v = cand[i];
            cand.erase(cand.begin() + i);
            cand.insert(cand.begin() + j, v);
            const double cb = emitted(cand.data(), m);
            if (cb >= 0 && cb < cur_bytes) {
                if (std::getenv("MOVES")) std::printf("move: request %lld m %d: %d%d from %d to %d (%+d), %.2f -> %.2f MB (%.1f %%)\n", (long long)b, m, v / 10, v % 10, i, j, j - i, cur_bytes / 1e6, cb / 1e6, 100.0 * (1 - cb / cur_bytes));
                cur = cand; cur_bytes = cb; ++moves;
            }
            std::vector<uint8_t> cm = best_model_order;
            const uint8_t u = cm[i];
            cm.erase(cm.begin() + i);
            cm.insert(cm.begin() + j, u);
            const double mm = order_simulate(on, *os, cm.data(), m, bm);
            if (mm < bm) { bm = mm; best_model_order = cm; }
        }
        base_bytes += b0;
        climbed_bytes += cur_bytes;
        base_model += m0;
        climbed_model += order_simulate(on, *os, cur.data(), m, __builtin_inf());
        model_of_best_model += bm;
        const double bb = emitted(best_model_order.data(), m);
        bytes_of_best_model += bb < 0 ? b0 : bb;
        improved += cur_bytes < b0;
        if (std::getenv("ORDER_DUMP") && b < atoi(std::getenv("ORDER_DUMP"))) {
            auto show = [&](const char *name, const std::vector<uint8_t> &o, double bytes) {
                std::printf("  %s %.2f MB:", name, bytes / 1e6);
                // replay on scopes: the width (variables of the created factor) of every elimination
                std::vector<B2> fs;
                B2 rel = os->rel, eb;
                for (int k = 0; k < NE; ++k) eb.set(ev[k]);
                b2_each(rel, [&](int v) { B2 sc = net.scope2[v]; sc.a &= ~eb.a; sc.b &= ~eb.b; fs.push_back(sc); });
                std::vector<char> alive(fs.size(), 1);
                for (uint8_t x : o) {
                    B2 u;
                    for (size_t i = 0; i < fs.size(); ++i)
                        if (alive[i] && fs[i].test(x)) { alive[i] = 0; u.a |= fs[i].a; u.b |= fs[i].b; }
                    u.clr(x);
                    fs.push_back(u);
                    alive.push_back(1);
                    std::printf(" %d%d:%d", x / 10, x % 10, b2_count(u));
                }
                std::printf("\n");
            };
            std::printf("request %lld: query %d%d evidence", (long long)b, qv[0] / 10, qv[0] % 10);
            for (int k = 0; k < NE; ++k) std::printf(" %d%d", ev[k] / 10, ev[k] % 10);
            std::printf("\n");
            show("planner", std::vector<uint8_t>(os->best, os->best + m), b0);
            show("climbed", cur, cur_bytes);
        }
    }
    const double nb = (double)(B - failed);
    std::printf("%lld requests (%lld skipped, %lld searched), %d evidence nodes, %d moves tried per searched request\n", (long long)B, (long long)failed, (long long)searched, NE, TRIES);
    std::printf("  the planner's order:                      emitted %.3f MB per request, modelled %.3f MB\n", base_bytes / nb / 1e6, base_model / nb / 1e6);
    std::printf("  hill climb on the EMITTED bytes:          emitted %.3f MB (%.1f %% less; %lld requests improved, %.1f moves kept each), modelled %.3f MB\n", climbed_bytes / nb / 1e6,
                100.0 * (1 - climbed_bytes / base_bytes), (long long)improved, (double)moves / nb, climbed_model / nb / 1e6);
    std::printf("  hill climb on the MODEL (order_simulate): emitted %.3f MB (%.1f %% less), modelled %.3f MB (%.1f %% less)\n", bytes_of_best_model / nb / 1e6,
                100.0 * (1 - bytes_of_best_model / base_bytes), model_of_best_model / nb / 1e6, 100.0 * (1 - model_of_best_model / base_model));
    return 0;
}
