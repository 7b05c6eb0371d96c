// Regression guard for planner optimisations (CPU only): plans seeded random requests on a family of synthetic networks -
// uniform and mixed cardinalities, grids and random DAGs, every option set the tests force - and prints one fingerprint of the
// emitted programs + work items + statistics per (network, option set).  A change that is meant to make the planner faster must
// not move a single line of this output.
//   g++ -O2 -mpopcnt -std=c++17 tools/plan_fingerprint.cpp sorobn_amd/csrc/planner.cpp -lpthread -o /tmp/plan_fingerprint
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <string>
#include <vector>

#include "../sorobn_amd/csrc/planner.h"
using namespace mibn;

struct Spec {
    std::string name;
    std::vector<int32_t> card, scope_vars;
    std::vector<int64_t> scope_off{0}, value_off{0};
    std::vector<double> values;
    void add(const std::vector<int32_t> &parents, int v) {
        int64_t cells = card[v];
        for (int p : parents) { scope_vars.push_back(p); cells *= card[p]; }
        scope_vars.push_back(v);
        scope_off.push_back((int64_t)scope_vars.size());
        for (int64_t i = 0; i < cells; ++i) values.push_back(0.25 + 0.5 * ((i * 2654435761u) % 97) / 97.0);
        value_off.push_back((int64_t)values.size());
    }
};

static Spec grid(int R, int C, const std::vector<int> &cards, uint64_t seed, const char *name) {
    Spec s;
    s.name = name;
    std::mt19937_64 rng(seed);
    for (int v = 0; v < R * C; ++v) s.card.push_back(cards[rng() % cards.size()]);
    for (int v = 0; v < R * C; ++v) {
        std::vector<int32_t> pa;
        if (v / C) pa.push_back(v - C);
        if (v % C) pa.push_back(v - 1);
        s.add(pa, v);
    }
    return s;
}

static Spec random_dag(int n, int max_parents, const std::vector<int> &cards, uint64_t seed, const char *name) {
    Spec s;
    s.name = name;
    std::mt19937_64 rng(seed);
    for (int v = 0; v < n; ++v) s.card.push_back(cards[rng() % cards.size()]);
    for (int v = 0; v < n; ++v) {
        std::vector<int32_t> pa;
        const int np = v ? (int)(rng() % (std::min(v, max_parents) + 1)) : 0;
        while ((int)pa.size() < np) {
            const int p = (int)(rng() % v);
            // mostly recent nodes: long chains of interaction, like a layered network
            const int q = v - 1 - (int)(rng() % std::min(v, 12));
            const int pick = (rng() & 3) ? q : p;
            if (std::find(pa.begin(), pa.end(), pick) == pa.end()) pa.push_back(pick);
        }
        std::sort(pa.begin(), pa.end());
        s.add(pa, v);
    }
    return s;
}

struct Opt {
    const char *name;
    int small_cells;
    int64_t big_iters;
    double minfill_above;
    int fuse, chain, sweep, sweep_min, outer, prune, order_weights;
};

static uint64_t fnv(uint64_t h, const void *p, size_t n) {
    const unsigned char *c = (const unsigned char *)p;
    for (size_t i = 0; i < n; ++i) h = (h ^ c[i]) * 1099511628211ull;
    return h;
}

int main(int argc, char **argv) {
    const int64_t B = argc > 1 ? atoll(argv[1]) : 1500;
    std::vector<Spec> specs;
    specs.push_back(grid(10, 10, {4}, 1, "grid10x10_k4"));
    specs.push_back(grid(5, 10, {8}, 2, "grid5x10_k8"));
    specs.push_back(grid(8, 8, {3}, 3, "grid8x8_k3"));
    specs.push_back(grid(9, 9, {2, 3, 4, 5}, 4, "grid9x9_mixed"));
    specs.push_back(grid(12, 10, {2, 4}, 5, "grid12x10_k24"));
    specs.push_back(grid(7, 18, {4}, 6, "grid7x18_k4"));
    specs.push_back(grid(6, 6, {1, 2, 4, 16}, 7, "grid6x6_k1_16"));
    specs.push_back(random_dag(40, 3, {2, 3, 4, 5}, 8, "dag40_mixed"));
    specs.push_back(random_dag(90, 3, {2, 3, 4}, 9, "dag90_mixed"));
    specs.push_back(random_dag(128, 2, {4}, 10, "dag128_k4"));
    specs.push_back(random_dag(160, 2, {2, 3}, 11, "dag160_k23"));  // > 128 variables: the generic bit sets
    specs.push_back(random_dag(12, 3, {2, 3, 4, 5, 17, 33}, 12, "dag12_wide"));
    const Opt opts[] = {
        {"default", 1024, 4096, 2e7, 1, 1, 5, 2, 1, 1, 1},
        {"minfill_always", 1024, 4096, 0.0, 1, 1, 5, 2, 1, 1, 1},
        {"forced_small", 4, 16, 0.0, 1, 1, 5, 2, 1, 1, 1},
        {"forced_16_64", 16, 64, 2e7, 1, 1, 5, 3, 1, 1, 1},
        {"no_fuse", 1024, 4096, 2e7, 0, 0, 0, 2, 0, 1, 0},
        {"no_sweep_noprune", 64, 256, 1e5, 1, 1, 0, 2, 1, 0, 1},
    };
    for (const Spec &s : specs) {
        const int n = (int)s.card.size();
        for (const Opt &o : opts) {
            Network net;
            std::string e = net.set(n, s.card.data(), s.scope_off.data(), s.scope_vars.data(), s.value_off.data(), s.values.data());
            if (!e.empty()) { std::printf("%s: %s\n", s.name.c_str(), e.c_str()); return 1; }
            net.small_cells = o.small_cells; net.big_iters = o.big_iters; net.minfill_above = o.minfill_above; net.fuse = o.fuse;
            net.chain = o.chain; net.sweep = o.sweep; net.sweep_min = o.sweep_min; net.outer = o.outer; net.prune = o.prune;
            net.order_weights = o.order_weights;
            net.plan_cache = 0;
            std::vector<int32_t> hint(n);
            for (int v = 0; v < n; ++v) hint[v] = v;
            net.set_hints(1, hint.data());
            std::mt19937_64 rng(1234);
            std::vector<int64_t> q_off{0}, e_off{0}, out_off{0};
            std::vector<int32_t> qv, ev, ec;
            const bool heavy = !o.prune || n > 128;
            const int64_t nb = heavy ? std::max<int64_t>(B / 8, 16) : B;
            for (int64_t b = 0; b < nb; ++b) {
                const int nq = 1 + (int)(rng() % 3 == 0) + (int)(rng() % 7 == 0);
                const int ne = (int)(rng() % 5) + (rng() % 4 == 0 ? 8 : 0);
                std::vector<int> pick;
                while ((int)pick.size() < std::min(n, nq + ne)) {
                    const int v = (int)(rng() % n);
                    if (std::find(pick.begin(), pick.end(), v) == pick.end()) pick.push_back(v);
                }
                int64_t cells = 1;
                for (int i = 0; i < (int)pick.size(); ++i) {
                    if (i < nq) { qv.push_back(pick[i]); cells *= s.card[pick[i]]; }
                    else { ev.push_back(pick[i]); ec.push_back((int)(rng() % s.card[pick[i]])); }
                }
                q_off.push_back((int64_t)qv.size());
                e_off.push_back((int64_t)ev.size());
                out_off.push_back(out_off.back() + cells);
            }
            ThreadPool pool(1);
            std::vector<ProgBuf> bufs;
            BatchPlan bp;
            if (ev.empty()) { ev.push_back(0); ec.push_back(0); }
            plan_batch(net, pool, bufs, 0, nb, q_off.data(), qv.data(), e_off.data(), ev.data(), ec.data(), out_off.data(), nullptr, bp,
                       !o.prune);
            uint64_t h = 1469598103934665603ull;
            if (!bp.err.empty()) {
                h = fnv(h, bp.err.data(), bp.err.size());
            } else {
                h = fnv(h, bufs[0].data, bufs[0].size * 4);
                for (const Tag &t : bp.tags[0]) {  // (field by field: the struct has padding)
                    h = fnv(h, &t.rel_off, 4); h = fnv(h, &t.a, 4); h = fnv(h, &t.wgs, 4); h = fnv(h, &t.level, 2); h = fnv(h, &t.kid, 2);
                    h = fnv(h, &t.bytes, 4);
                }
                h = fnv(h, bp.cost.data(), bp.cost.size() * 8);
                h = fnv(h, bp.arena_need.data(), bp.arena_need.size() * 8);
                h = fnv(h, &bp.st.alg_flops, 8);
                h = fnv(h, &bp.st.n_steps, 8);
                Schedule sc;
                build_schedule(net, bp, bufs, 0, nb, sc);
                h = fnv(h, sc.items.data(), sc.items.size() * sizeof(Item));
                h = fnv(h, sc.wg_item.data(), sc.wg_item.size() * 4);
                h = fnv(h, sc.arena_off.data(), sc.arena_off.size() * 8);
            }
            std::printf("%-16s %-18s %016llx  words %zu steps %.0f bytes %.6g%s\n", s.name.c_str(), o.name, (unsigned long long)h,
                        bp.total_words, bp.st.n_steps, bp.st.alg_bytes, bp.err.empty() ? "" : (" ERR " + bp.err).c_str());
            for (auto &b : bufs) b.release();
        }
    }
    return 0;
}
