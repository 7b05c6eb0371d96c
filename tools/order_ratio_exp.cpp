// second_cost / best_cost of the byte model against "the runner-up's EMITTED program wins", by pair of candidates (CPU; profiles/NOTES_r06.md, session BT):
// where is the second emission worth its time, and which candidate does the model overrate?   g++ -O2 -mpopcnt -std=c++17 tools/order_ratio_exp.cpp sorobn_amd/csrc/planner.cpp -lpthread
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <string>
#include <vector>
#include "../sorobn_amd/csrc/planner.h"
using namespace mibn;
int main(int argc, char **argv) {
    const int64_t B = argc > 1 ? atoll(argv[1]) : 4000;
    const int NE = argc > 2 ? atoi(argv[2]) : 4;
    const int R = 10, C = 10, K = 4, n = R * C;
    std::vector<int32_t> card(n, K), scope_vars; std::vector<int64_t> scope_off{0}, value_off{0}; std::vector<double> values;
    std::mt19937_64 rng(1); std::uniform_real_distribution<double> U(0.1, 1.0);
    for (int v = 0; v < n; ++v) { const int r = v / C, c = v % C; if (r) scope_vars.push_back(v - C); if (c) scope_vars.push_back(v - 1); scope_vars.push_back(v); scope_off.push_back((int64_t)scope_vars.size());
        int64_t cells = K; if (r) cells *= K; if (c) cells *= K; for (int64_t i = 0; i < cells; ++i) values.push_back(U(rng)); value_off.push_back((int64_t)values.size()); }
    Network net; net.set(n, card.data(), scope_off.data(), scope_vars.data(), value_off.data(), values.data());
    std::vector<int32_t> hint(n); for (int v = 0; v < n; ++v) hint[v] = v; net.set_hints(1, hint.data());
    net.order_effort = 1;
    const EmitNet en = net.emit_view(); const OrderNet on = net.order_view();
    std::vector<char> slice(emit_scratch_bytes(n) + 64); std::vector<uint32_t> slot(1 << 16); OrderScratch *os = new OrderScratch;
    char *base = slice.data() + ((64 - (reinterpret_cast<uintptr_t>(slice.data()) & 63)) & 63);
    int32_t qv[1], ev[40], ec[40];
    auto emitted = [&](const uint8_t *order, int n_order) -> double { EmitScratch S; emit_scratch_carve(S, base, n); if (emit_begin(en, S, 1, qv, NE, ev, ec, false)) return -1; EmitBuf buf; buf.data = slot.data(); buf.cap = slot.size(); EmitStats st; if (emit_run(en, S, buf, st, nullptr, 1, qv, 0, order, n_order)) return -1; return st.alg_bytes; };
    const double edges[] = {1.0, 1.05, 1.1, 1.2, 1.35, 1.5, 2.0, 3.0, 1e30};
    static double first_only[6];
    static double pair_cnt[7][7], pair_win[7][7], pair_gain[7][7];
    double tot = 0, cnt[8] = {0}, wins[8] = {0}, gain[8] = {0}, bytes[8] = {0};
    for (int64_t b = 0; b < B; ++b) {
        int pick[40]; for (int k = 0; k < NE + 1;) { const int v = (int)(rng() % n); bool dup = false; for (int j = 0; j < k; ++j) dup = dup || pick[j] == v; if (!dup) pick[k++] = v; }
        qv[0] = pick[0]; for (int k = 0; k < NE; ++k) { ev[k] = pick[1 + k]; ec[k] = 0; }
        order_search(on, *os, 1, qv, NE, ev, false);
        std::vector<uint8_t> A(os->best, os->best + os->n_best), S2(os->second, os->second + os->n_second);
        const double ca = os->best_cost, cb = os->second_cost;
        const double ea = emitted(A.data(), (int)A.size());
        tot += ea;
        if (S2.empty() || ca < 2e7) continue;
        const double eb = emitted(S2.data(), (int)S2.size());
        const double ratio = cb / ca;
        // which candidates are they?  (regenerate: 0 meet, 1 reverse, 2 hint, 3 meet-1, 4 meet+1, 5 min-fill, 6 opening + meet)
        auto ident = [&](const std::vector<uint8_t> &o) {
            B2 rel, hidden; order_prepare(on, *os, 1, qv, NE, ev, false, rel, hidden);
            auto eq = [&](int nc) { return nc == (int)o.size() && std::equal(o.begin(), o.end(), os->cand); };
            const int qd = on.depth[qv[0]];
            order_sweep(on, *os, hidden, qd, 0); if (eq(os->n_cand)) return 0;
            order_sweep(on, *os, hidden, qd, 1); if (eq(os->n_cand)) return 1;
            os->n_cand = 0; for (int i = 0; i < n; ++i) if (hidden.test(on.hint_sorted[i])) os->cand[os->n_cand++] = (uint8_t)on.hint_sorted[i]; if (eq(os->n_cand)) return 2;
            if (qd > 0) { order_sweep(on, *os, hidden, qd - 1, 0); if (eq(os->n_cand)) return 3; }
            order_sweep(on, *os, hidden, qd + 1, 0); if (eq(os->n_cand)) return 4;
            order_greedy(on, *os, hidden, __builtin_inf()); if (eq(os->n_cand)) return 5;
            return 6;
        };
        {   // a penalty on the opening candidate's modelled cost: what the FIRST choice alone would emit
            const int ia0 = ident(A), ib0 = ident(S2);
            const double pens[6] = {1.0, 1.0625, 1.125, 1.1875, 1.25, 1.5};
            for (int p = 0; p < 6; ++p) {
                bool swap_ = false;
                if (ia0 == 6 && ib0 != 6 && cb < pens[p] * ca) swap_ = true;
                if (ib0 == 6 && ia0 != 6 && cb * pens[p] < ca) swap_ = false;  // (the second is penalised: stays second)
                first_only[p] += (swap_ && eb >= 0 ? eb : ea) - ea;
            }
        }
        if (ratio < 1.2) { const int ia = ident(A), ib = ident(S2); pair_cnt[ia][ib] += 1; if (eb >= 0 && eb < ea) { pair_win[ia][ib] += 1; pair_gain[ia][ib] += ea - eb; } }
        int k = 0; while (ratio >= edges[k + 1]) ++k;
        cnt[k] += 1; bytes[k] += ea;
        if (eb >= 0 && eb < ea) { wins[k] += 1; gain[k] += ea - eb; }
    }
    std::printf("%lld requests, %d evidence nodes, best >= 2e7: total %.3f MB per request emitted (first choice)\n", (long long)B, NE, tot / B / 1e6);
    for (int k = 0; k < 8; ++k) std::printf("  second / best cost in [%.2f, %.2g): %5.0f requests (%.1f %%), runner-up wins %4.0f (%.0f %%), bytes saved %.2f %% of all bytes\n", edges[k], edges[k + 1], cnt[k], 100.0 * cnt[k] / B, wins[k], cnt[k] ? 100.0 * wins[k] / cnt[k] : 0.0, 100.0 * gain[k] / tot);
    { const double pens[6] = {1.0, 1.0625, 1.125, 1.1875, 1.25, 1.5}; for (int p = 0; p < 6; ++p) std::printf("  opening penalised x %.4f: the first choice alone emits %.2f %% fewer bytes (over the requests >= 2e7; all bytes as base)\n", pens[p], -100.0 * first_only[p] / tot); }
    const char *nm[7] = {"meet", "reverse", "hint", "meet-1", "meet+1", "min-fill", "opening+meet"};
    for (int a = 0; a < 7; ++a) for (int b = 0; b < 7; ++b) if (pair_cnt[a][b] >= 10) std::printf("  best %-13s second %-13s: %4.0f requests, runner-up wins %3.0f %%, saves %.2f %% of all bytes\n", nm[a], nm[b], pair_cnt[a][b], 100.0 * pair_win[a][b] / pair_cnt[a][b], 100.0 * pair_gain[a][b] / tot);
    return 0;
}
