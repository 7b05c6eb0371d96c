// What could cross-request sharing of evidence-free eliminations save on the C3 stream?  (CPU only.)
// For every request the planner's own elimination order is replayed on factor SCOPES with the section-8(d) byte model, and every
// factor carries a signature of where it comes from: a CPT slice is "pure" when no evidence variable is in its scope (its numbers are
// the network's, whatever the request), an elimination's result is pure when all its inputs are - the same signature in two requests
// = the same table.  Reported: the bytes of the pure eliminations (what a perfect cache of results could skip, an upper bound), the
// bytes of their distinct signatures (what filling that cache costs once), and how many distinct pure results there are.
//   g++ -O2 -mpopcnt -std=c++17 tools/share_exp.cpp sorobn_amd/csrc/planner.cpp -lpthread -o /tmp/share_exp && /tmp/share_exp [requests] [evidence nodes]
#include <cstdio>
#include <cstdlib>
#include <random>
#include <string>
#include <unordered_map>
#include <vector>

#include "../sorobn_amd/csrc/planner.h"
using namespace mibn;

int main(int argc, char **argv) {
    const int64_t B = argc > 1 ? atoll(argv[1]) : 20000;
    const int NE = argc > 2 ? atoi(argv[2]) : 4;
    const int R = 10, C = 10, K = 4, n = R * C;
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
    std::vector<int32_t> hint(n);
    for (int v = 0; v < n; ++v) hint[v] = v;
    net.set_hints(1, hint.data());
    struct F { B2 scope; uint64_t sig; bool pure; };
    auto mix = [](uint64_t h, uint64_t v) { return (h ^ (v + 0x9e3779b97f4a7c15ull + (h << 6) + (h >> 2))) * 1099511628211ull; };
    double total = 0, pure_bytes = 0, planner_bytes = 0;
    std::unordered_map<uint64_t, double> distinct;  // signature of a pure elimination -> its bytes
    for (int64_t b = 0; b < B; ++b) {
        int pick[40];
        for (int k = 0; k < NE + 1;) {
            const int v = (int)(rng() % n);
            bool dup = false;
            for (int j = 0; j < k; ++j) dup = dup || pick[j] == v;
            if (!dup) pick[k++] = v;
        }
        int32_t qv[1] = {pick[0]}, ev[40], ec[40];
        for (int k = 0; k < NE; ++k) { ev[k] = pick[1 + k]; ec[k] = 0; }
        Request rq;
        rq.nq = 1; rq.qvars = qv; rq.ne = NE; rq.evars = ev; rq.ecodes = ec;
        std::vector<uint32_t> prog;
        std::vector<int32_t> order;
        PlanStats st;
        st.order = &order;
        if (!plan_request(net, rq, prog, st).empty()) return 1;
        planner_bytes += st.alg_bytes;
        B2 rel, eb;
        rel.set(qv[0]); rel.a |= net.anc2[qv[0]].a; rel.b |= net.anc2[qv[0]].b;
        for (int k = 0; k < NE; ++k) { eb.set(ev[k]); rel.set(ev[k]); rel.a |= net.anc2[ev[k]].a; rel.b |= net.anc2[ev[k]].b; }
        std::vector<F> fs;
        b2_each(rel, [&](int v) {
            F f;
            f.scope.a = net.scope2[v].a & ~eb.a; f.scope.b = net.scope2[v].b & ~eb.b;
            f.pure = !((net.scope2[v].a & eb.a) | (net.scope2[v].b & eb.b));
            f.sig = mix(0x1234, (uint64_t)v);
            fs.push_back(f);
        });
        std::vector<char> alive(fs.size(), 1);
        for (int x : order) {
            B2 u;
            double in = 0;
            bool pure = true;
            uint64_t sig = mix(0xabcd, (uint64_t)x);
            for (size_t i = 0; i < fs.size(); ++i)
                if (alive[i] && fs[i].scope.test(x)) {
                    alive[i] = 0;
                    u.a |= fs[i].scope.a; u.b |= fs[i].scope.b;
                    in += std::exp2(2.0 * b2_count(fs[i].scope));
                    pure = pure && fs[i].pure;
                    sig = mix(sig, fs[i].sig);  // (the consumed factors in slot order: the same order in every request that has them)
                }
            u.clr(x);
            const double bytes = 8.0 * (in + std::exp2(2.0 * b2_count(u)));
            total += bytes;
            if (pure) {
                pure_bytes += bytes;
                distinct.emplace(sig, bytes);
            }
            fs.push_back(F{u, sig, pure});
            alive.push_back(1);
        }
    }
    double distinct_bytes = 0;
    for (auto &kv : distinct) distinct_bytes += kv.second;
    std::printf("%lld requests, %d evidence nodes: planner %.2f MB per request (its passes fuse eliminations); one elimination at a time %.2f MB per request, of which in evidence-free\n"
                "eliminations %.2f MB = %.1f %% (the upper bound of what sharing results could skip); %zu distinct evidence-free results, %.1f MB to compute each once (%.3f MB per request over this stream)\n",
                (long long)B, NE, planner_bytes / B / 1e6, total / B / 1e6, pure_bytes / B / 1e6, 100.0 * pure_bytes / total, distinct.size(), distinct_bytes / 1e6, distinct_bytes / B / 1e6);
    return 0;
}
