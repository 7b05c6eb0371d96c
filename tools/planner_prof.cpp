// Host-planner profiling harness (CPU only): plans a random 1-query + 4-evidence stream on an R x C grid network with K
// states and reports us per request.  Build with -pg for gprof:
//   g++ -O2 -g -pg -mpopcnt -std=c++17 tools/planner_prof.cpp sorobn_amd/csrc/planner.cpp -lpthread -o /tmp/planner_prof
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <string>
#include <vector>

#include "../sorobn_amd/csrc/planner.h"
using namespace mibn;
#if defined(MIBN_PLAN_PROFILE)
namespace mibn { extern double g_prof[8]; }
#endif

int main(int argc, char **argv) {
    const int R = 10, C = 10, K = 4;
    const int64_t B = argc > 1 ? atoll(argv[1]) : 8192;
    const int threads = argc > 2 ? atoi(argv[2]) : 1;
    const int reps = argc > 3 ? atoi(argv[3]) : 3;
    const int n = R * C;
    // variable ids in anti-diagonal topological order like BayesNet.nodes (any topological order would do here)
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
    if (const char *e = std::getenv("ORDER_WEIGHTS")) net.order_weights = atoi(e);
    if (const char *e = std::getenv("MINFILL_ABOVE")) net.minfill_above = atof(e);
    if (const char *e = std::getenv("SWEEP_MIN")) net.sweep_min = atoi(e);
    net.plan_cache = std::getenv("PLAN_CACHE") ? atoi(std::getenv("PLAN_CACHE")) : 0;  // (the repetitions below would turn into template copies)
    std::vector<int32_t> hint(n);
    for (int v = 0; v < n; ++v) hint[v] = v;
    net.set_hints(1, hint.data());
    std::vector<int64_t> q_off(B + 1), e_off(B + 1), out_off(B + 1);
    const int NE = std::getenv("NEV") ? atoi(std::getenv("NEV")) : 4;  // evidence nodes per request (1 / 8 / 16 variants)
    std::vector<int32_t> qv(B), ev((size_t)NE * B), ec((size_t)NE * B);
    for (int64_t b = 0; b < B; ++b) {
        int pick[33];
        for (int k = 0; k < NE + 1;) {
            const int v = (int)(rng() % n);
            bool dup = false;
            for (int j = 0; j < k; ++j) dup = dup || pick[j] == v;
            if (!dup) pick[k++] = v;
        }
        qv[b] = pick[0];
        for (int k = 0; k < NE; ++k) { ev[NE * b + k] = pick[1 + k]; ec[NE * b + k] = (int)(rng() % K); }
    }
    for (int64_t b = 0; b <= B; ++b) { q_off[b] = b; e_off[b] = NE * b; out_off[b] = 4 * b; }
    ThreadPool pool(threads);
    std::vector<ProgBuf> bufs;
    BatchPlan bp;
    double best = 1e30;
    for (int r = 0; r < reps; ++r) {
        auto t0 = std::chrono::steady_clock::now();
        plan_batch(net, pool, bufs, 0, B, q_off.data(), qv.data(), e_off.data(), ev.data(), ec.data(), out_off.data(), nullptr, bp);
        const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        if (ms < best) best = ms;
    }
    if (!bp.err.empty()) { std::fprintf(stderr, "%s\n", bp.err.c_str()); return 1; }
    // fingerprint of the emitted programs in request order (independent of which worker planned a request): planner
    // optimisations must not change a single word
    uint64_t h = 1469598103934665603ull;
    for (int64_t b = 0; b < B; ++b) {
        const uint32_t *w = bufs[bp.thread_of[b]].data + bp.local_off[b];
        const uint32_t *end = w + 1;
        for (uint32_t s = 0, off = 1; s < w[0]; ++s) { off += w[off + 6]; end = w + off; }
        for (const uint32_t *p = w; p < end; ++p) h = (h ^ *p) * 1099511628211ull;
    }
    std::printf("program fingerprint %016llx\n", (unsigned long long)h);
    {   // the longest program and the most work items of a request (what the device emission has to reserve per request)
        size_t max_words = 0, max_tags = 0;
        for (int64_t b = 0; b < B; ++b) {
            const uint32_t *w = bufs[bp.thread_of[b]].data + bp.local_off[b];
            size_t off = 1;
            for (uint32_t s = 0; s < w[0]; ++s) off += w[off + 6];
            max_words = std::max(max_words, off);
            max_tags = std::max<size_t>(max_tags, bp.tag_count[b]);
        }
        std::printf("longest program %zu words, most work items of a request %zu\n", max_words, max_tags);
    }
    {   // bytes per class of work and a time estimate from the per-class rates of profiles/r03_g_probe_classes_32768.log
        std::vector<double> by(kNumKernels, 0.0);
        for (int t = 0; t < (int)bp.tags.size(); ++t)
            for (const Tag &tg : bp.tags[t]) by[tg.kid] += tg.bytes;
        double est_ms = 0, sweep = 0, mfma1 = 0, joins = 0, rest = 0;
        for (int k = 0; k < kNumKernels; ++k) {
            if (by[k] == 0) continue;
            const std::string nm = kernel_name(k);
            double rate = 1000;  // GB/s
            if (k == kKidSweep) { rate = 4400; sweep += by[k]; }
            else if (nm.find("fiber<1") == 0 && nm.find("mfma") != std::string::npos) { rate = nm.find("chain") != std::string::npos ? 3000 : 4300; mfma1 += by[k]; }
            else if (nm.find("fiber<2") == 0) { rate = nm.find("outer") != std::string::npos ? 2250 : 1300; joins += by[k]; }
            else if (nm == "fiber<1,cx4,nc4>") { rate = 3200; rest += by[k]; }
            else rest += by[k];
            est_ms += by[k] / rate / 1e6;
        }
        for (int k = 0; k < kNumKernels; ++k)
            if (by[k] / B > 2e4) std::printf("  %-28s %7.3f MB per request\n", kernel_name(k), by[k] / B / 1e6);
        std::printf("per request: sweep %.2f MB, one-table MFMA %.2f MB, two-table joins %.2f MB, rest %.2f MB; estimated kernel time %.2f us per request\n",
                    sweep / B / 1e6, mfma1 / B / 1e6, joins / B / 1e6, rest / B / 1e6, est_ms * 1e3 / B);
    }
#if defined(MIBN_EMIT_PROF)  // g++ ... -DMIBN_EMIT_PROF, one thread: where the emission's time goes (phases of emit_run, emit_core.h)
    {
        static const char *names[11] = {"begin", "key / pos / slot sets", "factors of x", "sweep candidates", "SWEEP 5 / 4", "CHAIN", "SWEEP 3 / 2", "pair",
                                        "single elimination", "final product", "work items"};
        double tot = 0;
        for (int k = 0; k < 11; ++k) tot += (double)g_host_emit_prof.a[k];
        for (int k = 0; k < 11; ++k) std::printf("  emit phase %-24s %5.1f %%\n", names[k], 100.0 * (double)g_host_emit_prof.a[k] / tot);
    }
#endif
#if defined(MIBN_PLAN_PROFILE)  // g++ ... -DMIBN_PLAN_PROFILE, one thread: plan_request's phases (planner.cpp PROF)
    {
        const double n = (double)B * reps;
        std::printf("  per request: plan_request %.2f us = order search %.2f + emission %.2f + rest %.2f\n", g_prof[0] / n, g_prof[1] / n, g_prof[4] / n,
                    (g_prof[0] - g_prof[1] - g_prof[4]) / n);
    }
#endif
    std::printf("threads %d: %.1f ms for %lld requests = %.2f us/request/thread (x%d threads), %.0f req/s; %.1f steps, %.0f words, %.2f MB per request\n",
                threads, best, (long long)B, best * 1e3 / B * threads, threads, B / best * 1e3, bp.st.n_steps / B, (double)bp.total_words / B, bp.st.alg_bytes / B / 1e6);
    return 0;
}
