// Host check of the wave planner (csrc/wave_plan.h compiled for the host: one lane runs every iteration) against the host planner
// (order_search.h / emit_core.h) on a random stream over an R x C grid of K-state variables: orders, programs, statistics and
// work items must agree word for word.   g++ -O2 -mpopcnt -std=c++17 tools/wave_check.cpp sorobn_amd/csrc/planner.cpp -lpthread
// (-DMIBN_WAVE_REVERSE: every wv::for_n runs backwards - iterations that depend on each other show up as differences)
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <string>
#include <vector>

#include "../sorobn_amd/csrc/planner.h"
using namespace mibn;

int main(int argc, char **argv) {
    const int64_t B = argc > 1 ? atoll(argv[1]) : 2000;
    const int NE = argc > 2 ? atoi(argv[2]) : 4;
    const int R = argc > 3 ? atoi(argv[3]) : 10, C = argc > 4 ? atoi(argv[4]) : 10, K = argc > 5 ? atoi(argv[5]) : 4;
    const int n = R * C;
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
    if (const char *s = std::getenv("MINFILL_ABOVE")) net.minfill_above = atof(s);
    if (const char *s = std::getenv("ORDER_EFFORT")) net.order_effort = atoi(s);
    if (const char *s = std::getenv("SECOND_ABOVE")) net.second_above = atof(s);
    std::vector<int32_t> hint(n);
    for (int v = 0; v < n; ++v) hint[v] = v;
    net.set_hints(1, hint.data());
    WNet *wn = new WNet;
    if (!net.wave_view(*wn)) { std::fprintf(stderr, "network outside the wave planner's coverage\n"); return 1; }
    WState *ws = new WState;
    std::printf("sizeof WNet %zu, WState %zu (order part %zu, emission part %zu)\n", sizeof(WNet), sizeof(WState), sizeof(ws->o), sizeof(ws->e));
    OrderScratch *os = new OrderScratch;
    const OrderNet on = net.order_view();
    int64_t bad = 0, n_hidden = 0;
    size_t max_tags = 0;
    for (int64_t b = 0; b < B; ++b) {
        int pick[40];
        for (int k = 0; k < NE + 1;) {
            const int v = (int)(rng() % n);
            bool dup = false;
            for (int j = 0; j < k; ++j) dup = dup || pick[j] == v;
            if (!dup) pick[k++] = v;
        }
        int32_t qv[1] = {pick[0]}, ev[40];
        for (int k = 0; k < NE; ++k) ev[k] = pick[1 + k];
        order_search(on, *os, 1, qv, NE, ev, false);
        WOrderCtx oc(*wn, *ws);
        const int nw = oc.search(1, qv, NE, ev, net.anc2.data(), false);
        n_hidden += os->n_best;
        if (nw != os->n_best || std::memcmp(ws->order, os->best, (size_t)os->n_best) != 0) {
            if (++bad <= 5) std::printf("request %lld: order differs (host %d, wave %d)\n", (long long)b, os->n_best, nw);
            continue;
        }
        // the program, the statistics and the work items
        int32_t ec[40];
        for (int k = 0; k < NE; ++k) ec[k] = (int)(rng() % K);
        Request rq;
        rq.nq = 1; rq.qvars = qv; rq.ne = NE; rq.evars = ev; rq.ecodes = ec; rq.out_off = 4 * b;
        std::vector<uint32_t> hp;
        PlanStats st;
        const std::string pe = plan_request(net, rq, hp, st);
        if (!pe.empty()) { std::printf("request %lld: host planner: %s\n", (long long)b, pe.c_str()); ++bad; continue; }
        std::vector<Tag> htags;
        tag_program(net.emit_view(), hp.data(), [&](const Tag &t) { htags.push_back(t); });
        std::vector<uint32_t> slot(2 * 8192 + kWStashWords + 4 * kMaxStepWords, 0xdeadbeefu);
        WResult R;
        wave_plan_request(*wn, *ws, net.anc2.data(), 1, qv, NE, ev, ec, false, 4 * b, slot.data(), (uint32_t)slot.size(), R);
        bool same = R.err == 0 && R.words == hp.size() && std::memcmp(slot.data() + R.base, hp.data(), hp.size() * 4) == 0;
        if (same) same = R.alg_bytes == st.alg_bytes && R.alg_flops == st.alg_flops && R.n_steps == st.n_steps && R.max_step_cells == st.max_step_cells && R.arena_cells == st.arena_cells;
        if (same) same = R.n_tags == htags.size() && std::memcmp(ws->e.tags, htags.data(), htags.size() * sizeof(Tag)) == 0;
        max_tags = std::max<size_t>(max_tags, htags.size());
        if (!same) {
            if (++bad <= 8) {
                size_t d = 0;
                while (d < hp.size() && d < R.words && slot[R.base + d] == hp[d]) ++d;
                // which step holds the first differing word
                size_t off = 1, step = 0;
                while (step < hp[0] && off + hp[off + 6] <= d) { off += hp[off + 6]; ++step; }
                std::printf("request %lld (q %d): err %d words %u / %zu, first difference at word %zu = step %zu (+%zu, kind %u flags %x) host %08x wave %08x; steps %.0f / %.0f bytes %.0f / %.0f arena %lld / %lld tags %u / %zu\n",
                            (long long)b, qv[0], R.err, R.words, hp.size(), d, step, d - off, hp[off] & 0xff, hp[off + 1] >> 16, d < hp.size() ? hp[d] : 0u, slot[d], R.n_steps, st.n_steps,
                            R.alg_bytes, st.alg_bytes, (long long)R.arena_cells, (long long)st.arena_cells, R.n_tags, htags.size());
            }
        }
    }
    std::printf("most work items of a request %zu\n", max_tags);
#if defined(MIBN_WAVE_COUNT)
    {
        const long *c = g_wave_count;
        const double n = (double)B;
        std::printf("per request: emit() calls nx=0 %.2f  nx=1 %.2f  nx=2 (pair attempts) %.2f  nx=3 (CHAIN attempts) %.2f; failed pair %.2f failed CHAIN %.2f; GENERIC of a streaming-size step %.2f, of a small step %.2f\n",
                    c[0] / n, c[2] / n, c[5] / n, c[7] / n, c[10] / n, c[11] / n, c[12] / n, c[13] / n);
        std::printf("             SWEEP attempts k=2..5: %.2f %.2f %.2f %.2f   emitted: %.2f %.2f %.2f %.2f\n", c[16] / n, c[17] / n, c[18] / n, c[19] / n, c[22] / n, c[23] / n, c[24] / n, c[25] / n);
    }
#endif
    std::printf("%lld requests, %d evidence nodes: %lld requests differ (order / program / statistics / work items; mean %.1f hidden variables)\n", (long long)B, NE, (long long)bad, (double)n_hidden / B);
    return bad != 0;
}
