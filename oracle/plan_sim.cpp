// TEST INFRASTRUCTURE - CPU interpreter for the planner's step programs.
//
// Links the product's host-side planner (sorobn_amd/csrc/planner.cpp, pure C++) and executes the
// step program it emits with plain scalar loops, so that the *host logic* (relevance pruning, order
// selection, layouts, strides, axis merging, arena allocation, evidence slicing) can be checked
// against the golden vectors in the GPU-less build container (`pytest -m "not gpu"`).  It mirrors
// the step semantics documented in sorobn_amd/csrc/planner.h / ve_kernel.hip.h:
//     psi[out] = sum_x prod_j phi_j[base_j + idx_j(out) + x*xs_j]        (bayes_net.py:780-785)
// and the final normalisation (bayes_net.py:790).
//
// This library is NOT part of the product and is never loaded by sorobn_amd: the product path has
// no CPU fallback (mibn_query_batch fails without a gfx950 device).  Only tests/ may load it.
#include <algorithm>
#include <chrono>
#include <cstdint>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../sorobn_amd/csrc/planner.h"

using namespace mibn;

static std::string g_err;
static int g_order_effort = 0;
static double g_second_above = 1e7;
static int g_small_cells = 1024, g_big_iters = 4096, g_tile_h = 0, g_fuse = 1, g_prune = 1, g_chain = 1, g_sweep = 5, g_sweep_min = 2;
extern "C" void plan_sim_set_small_cells(int v) { g_small_cells = v; }
extern "C" void plan_sim_set_tiling(int big_iters, int tile_h) { g_big_iters = big_iters; g_tile_h = tile_h; }
extern "C" void plan_sim_set_fuse(int fuse) { g_fuse = fuse; }
extern "C" void plan_sim_set_chain(int chain) { g_chain = chain; }
extern "C" void plan_sim_set_sweep(int sweep) { g_sweep = sweep; }
extern "C" void plan_sim_set_sweep_min(int k) { g_sweep_min = k; }
extern "C" void plan_sim_set_prune(int prune) { g_prune = prune; }
extern "C" void plan_sim_set_order_effort(int effort, double second_above) { g_order_effort = effort; g_second_above = second_above; }

extern "C" const char *plan_sim_error() { return g_err.c_str(); }

// Execute iterations [h_begin, h_end) x lo_cells of one step (the whole step when the range covers hi_cells).
// Mirrors generic_body / fiber_body of ve_kernel.hip.h with scalar loops.
static int exec_step(const Network &net, const uint32_t *p, int64_t h_begin, int64_t h_end, std::vector<double> &arena,
                     int64_t arena_base, int64_t arena_cells, double *out) {
    const uint32_t w0 = p[0];
    const uint32_t kind = w0 & 0xff;
    const int n_in = (w0 >> 8) & 0xff, na = (w0 >> 16) & 0xff;
    const int cx = (int)(p[1] & 0xffff);
    const bool fin = (p[1] >> 16) & kFlagFinal;
    const int64_t lo = p[2];
    const uint64_t out_off = (uint64_t)p[4] | ((uint64_t)p[5] << 32);
    double *slot = arena.data() + arena_base;
    double *outp = fin ? out + out_off : slot + out_off;
    auto table = [&](uint64_t o) { return (o & kConstFlag) ? net.pool.data() + (o & ~kConstFlag) : slot + o; };
    const int64_t it0 = h_begin * lo, it1 = h_end * lo;
    if (kind == kKindGeneric) {
        if (!fin && (int64_t)out_off + it1 > arena_cells) { g_err = "step writes outside its arena"; return -7; }
        const double *inp[kMaxIn];
        int xs[kMaxIn];
        for (int j = 0; j < n_in; ++j) {
            inp[j] = table((uint64_t)p[kHdrWords + 3 * j] | ((uint64_t)p[kHdrWords + 3 * j + 1] << 32));
            xs[j] = (int)p[kHdrWords + 3 * j + 2];
        }
        const uint32_t *cd = p + kHdrWords + 3 * n_in;
        const int32_t *strd = (const int32_t *)(cd + na);
        std::vector<double> tmp((size_t)(it1 - it0));
        for (int64_t o = it0; o < it1; ++o) {
            int64_t r = o;
            int64_t off[kMaxIn] = {0};
            for (int a = 0; a < na; ++a) {
                const int64_t d = r % cd[a];
                r /= cd[a];
                for (int j = 0; j < n_in; ++j) off[j] += d * strd[j * na + a];
            }
            double acc = 0.0;
            for (int x = 0; x < cx; ++x) {
                double v = 1.0;
                for (int j = 0; j < n_in; ++j) v *= inp[j][off[j] + (int64_t)x * xs[j]];
                acc += v;
            }
            tmp[(size_t)(o - it0)] = acc;
        }
        std::memcpy(outp + it0, tmp.data(), sizeof(double) * tmp.size());
        return 0;
    }
    if (kind == kKindSweep) {
        // SWEEP (planner.h): tiles [h_begin, h_end) of Rt consecutive R cells x all 4^k combinations of the eliminated
        // variables, the k stages applied in place on a tile-sized scratch - what ve_sweep_kernel does in LDS
        const int k = na, rb = (w0 >> 24) & 0xff;
        const int64_t Rt = int64_t(1) << rb, tiles = p[3], Rcells = tiles * Rt;
        if (k < 2 || k > 5 || rb != 13 - 2 * k || lo != kSweepTileCells || cx != (1 << (2 * k))) { g_err = "malformed SWEEP header"; return -9; }
        const int kout = (int)(p[7] & 0xffff), t_total = (int)(p[7] >> 16);
        if (kout > k || t_total > kSweepMaxT) { g_err = "SWEEP: bad output rank or T size"; return -9; }
        const uint32_t *q = p + kHdrWords;
        const uint64_t f_off = (uint64_t)q[0] | ((uint64_t)q[1] << 32);
        if (f_off & kConstFlag) { g_err = "SWEEP: the big input is a constant"; return -9; }
        const double *F = slot + f_off;
        if ((int64_t)f_off + (Rcells << (2 * k)) > arena_cells) { g_err = "SWEEP reads outside its arena"; return -7; }
        if ((int64_t)out_off + (Rcells << (2 * kout)) > arena_cells) { g_err = "SWEEP writes outside its arena"; return -7; }
        q += 2;
        const uint32_t *stage = q;
        const uint32_t *small = q + k * kSweepStageWords;
        // T tables
        std::vector<double> T((size_t)t_total, 0.0);
        {
            const uint32_t *sm = small;
            bool live[5] = {true, true, true, true, true}, seen[5] = {};
            int t_expect = 0;
            for (int j = 0; j < k; ++j) {
                const uint32_t s0 = stage[j * kSweepStageWords], s1 = stage[j * kSweepStageWords + 1];
                const int dig = s0 & 15, cout = (s0 >> 4) & 15, ns = (s0 >> 8) & 15, nctrl = (s0 >> 12) & 15;
                const int t_off = (int)(s1 & 0xffff), t_cells = (int)(s1 >> 16);
                if (dig >= k || seen[dig] || (cout != 1 && cout != 4) || nctrl > 3 || t_off != t_expect || t_cells != (cout * 4 << (2 * nctrl)) ||
                    t_off + t_cells > t_total) { g_err = "SWEEP: malformed stage"; return -9; }
                seen[dig] = true;
                t_expect += t_cells;
                {   // the lane / loop split of the fibers follows the rule the kernel compiles in
                    const int loop = (s0 >> 16) & 15;
                    int m = 0;
                    bool okf = loop == sweep_loop_digit(k, dig);
                    for (int d = 0; d < k; ++d)
                        if (d != dig && d != loop) { okf = okf && (int)((s0 >> (20 + 4 * m)) & 15) == d; ++m; }
                    for (; m < 3; ++m) okf = okf && ((s0 >> (20 + 4 * m)) & 15) == 7;
                    if (!okf) { g_err = "SWEEP: thread fields / loop digit off the rule"; return -9; }
                    if (((p[1] >> 16) & kFlagSweepCanon) && dig != k - 1 - j) { g_err = "SWEEP: canonical flag on a permuted step"; return -9; }
                }
                int n_rctrl = 0;
                for (int c = 0; c < nctrl; ++c) {
                    const uint32_t cw = stage[j * kSweepStageWords + 2 + c];
                    const int src = cw & 0xff, ts = (int)(cw >> 8);
                    if (ts != (cout * 4 << (2 * c))) { g_err = "SWEEP: ctrl stride"; return -9; }
                    if (src >= 8 && ++n_rctrl > 2) { g_err = "SWEEP: more than two ctrl values from r"; return -9; }
                    if (src < 8 && (src >= k || src == dig || !live[src])) { g_err = "SWEEP: ctrl on a dead or contracted digit"; return -9; }
                    if (src >= 8 && (int64_t(4) << (src - 8)) > Rcells) { g_err = "SWEEP: ctrl bits beyond the R cells"; return -9; }
                }
                for (int t = 0; t < t_cells; ++t) {
                    const int n = t % cout, x = (t / cout) & 3;
                    int cc = t / (cout * 4);
                    int cv[3];
                    for (int c = 0; c < 3; ++c) { cv[c] = cc & 3; cc >>= 2; }
                    double v = 1.0;
                    for (int i = 0; i < ns; ++i) {
                        const uint32_t *rec = sm + i * kSweepSmallWords;
                        const double *tab = table((uint64_t)rec[0] | ((uint64_t)rec[1] << 32));
                        int64_t off = (int64_t)n * (int32_t)rec[2] + (int64_t)x * (int32_t)rec[3];
                        for (int c = 0; c < nctrl; ++c) off += (int64_t)cv[c] * (int32_t)rec[4 + c];
                        v *= tab[off];
                    }
                    T[(size_t)(t_off + t)] = v;
                }
                sm += ns * kSweepSmallWords;
                if (cout == 1) live[dig] = false;
            }
            int surv = 0;
            for (int d = 0; d < k; ++d)
                if (live[d]) { if (((p[8] >> (4 * surv)) & 15) != (uint32_t)d) { g_err = "SWEEP: surviving digits"; return -9; } ++surv; }
            if (surv != kout) { g_err = "SWEEP: output rank"; return -9; }
        }
        std::vector<double> L((size_t)kSweepTileCells);
        for (int64_t tile = h_begin; tile < h_end; ++tile) {
            for (int64_t c = 0; c < kSweepTileCells; ++c) {
                const int64_t r = c & (Rt - 1), xc = c >> rb;
                L[(size_t)c] = F[xc * Rcells + tile * Rt + r];
            }
            for (int j = 0; j < k; ++j) {
                const uint32_t *sj = stage + j * kSweepStageWords;
                const int dig = sj[0] & 15, cout = (sj[0] >> 4) & 15, nctrl = (sj[0] >> 12) & 15;
                const int t_off = (int)(sj[1] & 0xffff);
                const int64_t sx = Rt << (2 * dig);
                // every fiber along `dig`: the thread-field / loop-digit split of the kernel must enumerate each exactly once
                const int loop = (sj[0] >> 16) & 15, fld[3] = {(int)((sj[0] >> 20) & 15), (int)((sj[0] >> 24) & 15), (int)((sj[0] >> 28) & 15)};
                std::vector<char> done((size_t)kSweepTileCells / 4, 0);
                for (int tid = 0; tid < kSweepWG; ++tid)
                    for (int l = 0; l < 4; ++l) {
                        int64_t base = tid & (Rt - 1);
                        int bits = tid >> rb;
                        for (int f = 0; f < 3; ++f)
                            if (fld[f] != 7) { base += (int64_t)(bits & 3) * (Rt << (2 * fld[f])); bits >>= 2; }
                        if (bits) { g_err = "SWEEP: thread fields do not use up the lane bits"; return -9; }
                        base += (int64_t)l * (Rt << (2 * loop));
                        if (loop == dig || loop >= k) { g_err = "SWEEP: loop digit"; return -9; }
                        const int64_t rg = tile * Rt + (base & (Rt - 1));
                        int toff = t_off;
                        for (int c = 0; c < nctrl; ++c) {
                            const uint32_t cw = sj[2 + c];
                            const int src = cw & 0xff, ts = (int)(cw >> 8);
                            const int val = src >= 8 ? (int)((rg >> (src - 8)) & 3) : (int)((base >> (rb + 2 * src)) & 3);
                            toff += val * ts;
                        }
                        // fiber id = base without the contracted digit
                        if ((base >> (rb + 2 * dig)) & 3) { g_err = "SWEEP: fiber base on the contracted digit"; return -9; }
                        const int64_t lowmask = (Rt << (2 * dig)) - 1;
                        const int64_t fid = (base & lowmask) | ((base >> 2) & ~lowmask);
                        if (done[(size_t)fid]) { g_err = "SWEEP: a fiber is visited twice"; return -9; }
                        done[(size_t)fid] = 1;
                        double f[4], o[4];
                        for (int x = 0; x < 4; ++x) f[x] = L[(size_t)(base + x * sx)];
                        for (int n = 0; n < cout; ++n) {
                            double sacc = 0.0;
                            for (int x = 0; x < 4; ++x) sacc += f[x] * T[(size_t)(toff + n + cout * x)];
                            o[n] = sacc;
                        }
                        for (int n = 0; n < cout; ++n) L[(size_t)(base + n * sx)] = o[n];
                    }
            }
            const int64_t ocells = Rt << (2 * kout);
            for (int64_t c = 0; c < ocells; ++c) {
                const int64_t lo_ = c & ((int64_t(1) << (2 * kout)) - 1), r = c >> (2 * kout);
                int64_t idx = r;
                for (int qd = 0; qd < kout; ++qd) idx += ((lo_ >> (2 * qd)) & 3) * (Rt << (2 * ((p[8] >> (4 * qd)) & 15)));
                outp[lo_ + ((tile * Rt + r) << (2 * kout))] = L[(size_t)idx];
            }
        }
        return 0;
    }
    // FIBER
    const int nb = p[7] & 0xf, ns = (p[7] >> 4) & 0xf, nN = (p[7] >> 8) & 0xf, nctrl = (p[7] >> 12) & 0xf;
    const int NC = (int)(p[7] >> 16);
    const int T = (int)(p[8] & 0xffff);
    const int c1 = (int)(p[8] >> 16);
    if (c1 < 1 || cx % c1) { g_err = "fiber step: c1 does not divide cx"; return -9; }
    const int nT = nN + nctrl;
    if ((p[1] >> 16) & kFlagChain) {
        // CHAIN (planner.h): out[r, n12, n3] = sum_x3 T3[..] * sum_x12 F[r, x12, x3] * T12[n12, x12, ctrl12(r), x3]
        const uint32_t *q = p + kHdrWords;
        if (nb != 2 || NC != 16 || cx != 16 || c1 != 4 || nN != 2 || !((p[1] >> 16) & kFlagContig)) { g_err = "malformed CHAIN step"; return -9; }
        const double *F = table((uint64_t)q[0] | ((uint64_t)q[1] << 32));
        const int fx1 = (int)q[2], fx2 = (int)q[3];
        const int T12 = (int)q[4], T3 = (int)q[5], fx3 = (int)q[6], t12x3 = (int)q[7];
        q += 8;
        if (T12 != T || T12 + T3 > kMaxT || (t12x3 != 0 && t12x3 * 4 != T12)) { g_err = "CHAIN step: bad table sizes"; return -9; }
        const double *sm[kMaxSmall];
        int sxs1[kMaxSmall], sxs2[kMaxSmall];
        const int32_t *sts[kMaxSmall];
        for (int j = 0; j < ns; ++j) {
            sm[j] = table((uint64_t)q[0] | ((uint64_t)q[1] << 32));
            sxs1[j] = (int)q[2];
            sxs2[j] = (int)q[3];
            sts[j] = (const int32_t *)(q + 4);
            q += 4 + nT;
        }
        const uint32_t *tcard = q; q += nT;
        const uint32_t *nout = q; q += 16;
        for (int n = 0; n < 16; ++n)
            if (nout[n] != (uint32_t)n) { g_err = "CHAIN step with scattered N offsets"; return -9; }
        const int n3s = (int)q[0], nd3 = (int)(q[1] & 0xff);
        const bool n12dep = (q[1] >> 8) & 1;
        q += 2;
        const uint32_t *tcard3 = q; q += nd3;
        if (n3s < 1 || n3s > 2 || nd3 < 2 || tcard3[0] != 4 || tcard3[1] != 4) { g_err = "CHAIN step: bad third-variable table"; return -9; }
        const double *sm3[2];
        const int32_t *sts3[2];
        for (int j = 0; j < n3s; ++j) { sm3[j] = table((uint64_t)q[0] | ((uint64_t)q[1] << 32)); sts3[j] = (const int32_t *)(q + 2); q += 2 + nd3; }
        const uint32_t *rax = q; q += 3 * na;
        const int32_t *bst = (const int32_t *)q;  // [0][a] = F, [1][a] = T3 strides
        {
            int64_t c = 1;
            for (int k = 0; k < nd3; ++k) c *= tcard3[k];
            if (c != T3) { g_err = "CHAIN step: T3 size does not match its dimensions"; return -9; }
            c = 256;
            for (int k = 2; k < nT; ++k) c *= tcard[k];
            if (c != T12) { g_err = "CHAIN step: T12 size does not match its dimensions"; return -9; }
            const int nlo_m = (p[0] >> 24) & 0xff;
            int64_t expect = 64;
            for (int a = 0; a < nlo_m; ++a) {
                if ((int64_t)rax[3 * a + 1] != expect) { g_err = "CHAIN step whose lane block is not contiguous"; return -9; }
                expect *= rax[3 * a];
            }
            // the kernel reads n12-dependent T3 entries at 16 * lane-column: dimensions 2 and 3 must then be (n1, n2)
            const int rs = (p[1] >> kRowStrideShift) & 0xff;
            if (rs != 1 && rs != 4 && rs != 16) { g_err = "bad MFMA row stride"; return -9; }
            auto lo_t = [&](int64_t cell, int which) {
                int64_t r = cell, to = 0;
                for (int a = 0; a < nlo_m; ++a) { to += (r % rax[3 * a]) * (which ? (int64_t)bst[na + a] : (int64_t)rax[3 * a + 2]); r /= rax[3 * a]; }
                return to;
            };
            for (int64_t w0 = 0; w0 < lo; w0 += 64)
                for (int rb = 0; rb < 4; ++rb)
                    for (int i = 0; i < 16; ++i) {
                        const int64_t c0 = w0 + rs * rb, cc = w0 + (i % rs) + rs * rb + 4 * rs * (i / rs);
                        if (cc < lo && c0 < lo && (lo_t(cc, 0) != lo_t(c0, 0) || lo_t(cc, 1) != lo_t(c0, 1))) { g_err = "CHAIN row block mixes table slices"; return -9; }
                    }
        }
        std::vector<double> Tt((size_t)T12), T3t((size_t)T3);
        for (int t = 0; t < T12; ++t) {
            int r = t;
            const int n = r % 16; r /= 16;
            const int x = r % 16; r /= 16;
            double v = 1.0;
            for (int j = 0; j < ns; ++j) {
                int64_t off = (int64_t)(x % 4) * sxs1[j] + (int64_t)(x / 4) * sxs2[j] + (int64_t)(n % 4) * sts[j][0] + (int64_t)(n / 4) * sts[j][1];
                int c = r;
                for (int k = 2; k < nT; ++k) { off += (int64_t)(c % tcard[k]) * sts[j][k]; c /= tcard[k]; }
                v *= sm[j][off];
            }
            Tt[(size_t)t] = v;
        }
        for (int t = 0; t < T3; ++t) {
            double v = 1.0;
            for (int j = 0; j < n3s; ++j) {
                int64_t off = 0;
                int c = t;
                for (int k = 0; k < nd3; ++k) { off += (int64_t)(c % tcard3[k]) * sts3[j][k]; c /= tcard3[k]; }
                v *= sm3[j][off];
            }
            T3t[(size_t)t] = v;
        }
        if (n12dep && (nd3 < 4 || tcard3[2] != 4 || tcard3[3] != 4)) { g_err = "CHAIN step: n12-dependent T3 without (n1, n2) dimensions"; return -9; }
        const int64_t total_cells = (int64_t)p[2] * (int64_t)p[3] * 64;
        if ((int64_t)out_off + total_cells > arena_cells) { g_err = "chain step writes outside its arena"; return -7; }
        std::vector<std::pair<int64_t, double>> writes;
        writes.reserve((size_t)((it1 - it0) * 64));
        for (int64_t rr = it0; rr < it1; ++rr) {
            int64_t r = rr, oo = 0, to = 0, fo = 0, t3o = 0;
            for (int a = 0; a < na; ++a) {
                const int64_t d = r % rax[3 * a];
                r /= rax[3 * a];
                oo += d * rax[3 * a + 1];
                to += d * rax[3 * a + 2];
                fo += d * bst[a];
                t3o += d * bst[na + a];
            }
            for (int n = 0; n < 16; ++n) {
                double g[4];
                for (int x3 = 0; x3 < 4; ++x3) {
                    double acc = 0.0;
                    for (int x = 0; x < 16; ++x)
                        acc += F[fo + (int64_t)(x % 4) * fx1 + (int64_t)(x / 4) * fx2 + (int64_t)x3 * fx3] * Tt[(size_t)(to + (int64_t)x3 * t12x3 + x * 16 + n)];
                    g[x3] = acc;
                }
                for (int n3 = 0; n3 < 4; ++n3) {
                    double acc = 0.0;
                    for (int x3 = 0; x3 < 4; ++x3) acc += g[x3] * T3t[(size_t)(t3o + (n12dep ? 16 * n : 0) + 4 * n3 + x3)];
                    const int64_t o = oo + n + 16 * n3;
                    if (o < 0 || o >= total_cells) { g_err = "chain output offset out of range"; return -8; }
                    writes.emplace_back(o, acc);
                }
            }
        }
        for (auto &w : writes) outp[w.first] = w.second;
        return 0;
    }
    const uint32_t *q = p + kHdrWords;
    const double *big[2];
    int bxs1[2], bxs2[2];
    for (int b = 0; b < nb; ++b) { big[b] = table((uint64_t)q[0] | ((uint64_t)q[1] << 32)); bxs1[b] = (int)q[2]; bxs2[b] = (int)q[3]; q += 4; }
    const double *sm[kMaxSmall];
    int sxs1[kMaxSmall], sxs2[kMaxSmall];
    const int32_t *sts[kMaxSmall];
    for (int j = 0; j < ns; ++j) {
        sm[j] = table((uint64_t)q[0] | ((uint64_t)q[1] << 32));
        sxs1[j] = (int)q[2];
        sxs2[j] = (int)q[3];
        sts[j] = (const int32_t *)(q + 4);
        q += 4 + nT;
    }
    const uint32_t *tcard = q; q += nT;
    const uint32_t *nout = q; q += NC;
    const bool outer = ((p[1] >> 16) & kFlagOuter) != 0;
    const uint32_t *nB = q;  // OUTER: offset of N-combination n in the second big input
    if (outer) { if (nb != 2 || NC != 16 || (ns == 0) != (T == 0)) { g_err = "malformed OUTER step"; return -9; } q += NC; }
    const uint32_t *rax = q; q += 3 * na;   // card, ostride, tstride per R axis
    const int32_t *bst = (const int32_t *)q;  // [b][a]
    if ((p[1] >> 16) & kFlagContig) {  // the kernel's vector / transposed stores rely on this
        for (int n = 0; n < NC; ++n)
            if (nout[n] != (uint32_t)n) { g_err = "CONTIG step with scattered N offsets"; return -9; }
        const int nlo_m = (p[0] >> 24) & 0xff;
        int64_t expect = NC;
        for (int a = 0; a < nlo_m; ++a) {
            if ((int64_t)rax[3 * a + 1] != expect) { g_err = "CONTIG step whose lane block is not contiguous"; return -9; }
            expect *= rax[3 * a];
        }
    }
    if (const int rs = (p[1] >> kRowStrideShift) & 0xff) {
        // MFMA form: within every wave (64 consecutive lane cells) the 16 cells of a row block share their T offset
        if (rs != 1 && rs != 4 && rs != 16) { g_err = "bad MFMA row stride"; return -9; }
        const int nlo_m = (p[0] >> 24) & 0xff;
        auto lo_t = [&](int64_t cell) {
            int64_t r = cell, to = 0;
            for (int a = 0; a < nlo_m; ++a) { to += (r % rax[3 * a]) * (outer ? (int64_t)bst[na + a] * 4096 + (int64_t)rax[3 * a + 2] : (int64_t)rax[3 * a + 2]); r /= rax[3 * a]; }
            return to;
        };
        for (int64_t w0 = 0; w0 < lo; w0 += 64)
            for (int rb = 0; rb < 4; ++rb)
                for (int i = 0; i < 16; ++i) {
                    const int64_t c0 = w0 + rs * rb, c = w0 + (i % rs) + rs * rb + 4 * rs * (i / rs);
                    if (c < lo && c0 < lo && lo_t(c) != lo_t(c0)) { g_err = "MFMA row block mixes T slices"; return -9; }
                }
    }
    std::vector<double> Tt((size_t)T);
    for (int t = 0; t < T; ++t) {
        int r = t;
        const int n = r % NC; r /= NC;
        const int x = r % cx; r /= cx;
        int64_t off[kMaxSmall] = {0};
        int rn = n;
        for (int k = 0; k < nN; ++k) { int d = rn % tcard[k]; rn /= tcard[k]; for (int j = 0; j < ns; ++j) off[j] += (int64_t)d * sts[j][k]; }
        for (int k = nN; k < nT; ++k) { int d = r % tcard[k]; r /= tcard[k]; for (int j = 0; j < ns; ++j) off[j] += (int64_t)d * sts[j][k]; }
        double v = 1.0;
        for (int j = 0; j < ns; ++j) v *= sm[j][off[j] + (int64_t)(x % c1) * sxs1[j] + (int64_t)(x / c1) * sxs2[j]];
        Tt[(size_t)t] = v;
    }
    const int64_t total_cells = (int64_t)p[2] * (int64_t)p[3] * NC;
    if ((int64_t)out_off + total_cells > arena_cells) { g_err = "fiber step writes outside its arena"; return -7; }
    std::vector<std::pair<int64_t, double>> writes;
    writes.reserve((size_t)((it1 - it0) * NC));
    for (int64_t rr = it0; rr < it1; ++rr) {
        int64_t r = rr, oo = 0, to = 0, bo[2] = {0, 0};
        for (int a = 0; a < na; ++a) {
            const int64_t d = r % rax[3 * a];
            r /= rax[3 * a];
            oo += d * rax[3 * a + 1];
            to += d * rax[3 * a + 2];
            for (int b = 0; b < nb; ++b) bo[b] += d * bst[b * na + a];
        }
        for (int n = 0; n < NC; ++n) {
            double acc = 0.0;
            for (int x = 0; x < cx; ++x) {
                double f = 1.0;
                for (int b = 0; b < nb; ++b)
                    f *= big[b][bo[b] + (outer && b == 1 ? (int64_t)nB[n] : 0) + (int64_t)(x % c1) * bxs1[b] + (int64_t)(x / c1) * bxs2[b]];
                acc += T ? f * Tt[(size_t)(to + (int64_t)x * NC + n)] : f;
            }
            const int64_t o = oo + nout[n];
            if (o < 0 || o >= total_cells) { g_err = "fiber output offset out of range"; return -8; }
            writes.emplace_back(o, acc);
        }
    }
    for (auto &w : writes) outp[w.first] = w.second;  // (inputs never alias the output)
    return 0;
}

// A CSR batch of requests through the product's batch planner (`threads` workers) and level-synchronous scheduler
// (`stagger` groups), executed exactly as the level kernel sees it: level by level, workgroup by workgroup, every
// workgroup looking its item up in wg_item and deriving its tile from (workgroup index - Item::b); every request in its
// own arena.  out[out_off[b] ..) receives the dense posterior of request b.
extern "C" int plan_sim_query_batch(int32_t n_vars, const int32_t *card, const int64_t *scope_off, const int32_t *scope_vars,
                                    const int64_t *value_off, const double *values, int32_t n_hints, const int32_t *hints,
                                    int64_t B, const int64_t *q_off, const int32_t *q_vars, const int64_t *e_off, const int32_t *e_vars,
                                    const int32_t *e_codes, const int64_t *out_off, double *out, int32_t stagger, int32_t threads,
                                    double *stats /* bytes, flops, steps, max_cells, arena_cells */) {
    Network net;
    g_err = net.set(n_vars, card, scope_off, scope_vars, value_off, values);
    if (!g_err.empty()) return -1;
    net.small_cells = g_small_cells;
    net.big_iters = g_big_iters;
    net.tile_h = g_tile_h;
    net.fuse = g_fuse;
    net.chain = g_chain;
    net.sweep = g_sweep;
    net.sweep_min = g_sweep_min;
    net.order_effort = g_order_effort; net.second_above = g_second_above;
    net.prune = g_prune;
    net.stagger = stagger;
    net.set_hints(n_hints, hints);
    std::vector<char> skip((size_t)B, 0);
    for (int64_t b = 0; b < B; ++b) {
        Request rq;
        rq.nq = (int32_t)(q_off[b + 1] - q_off[b]); rq.qvars = q_vars + q_off[b];
        rq.ne = (int32_t)(e_off[b + 1] - e_off[b]); rq.evars = e_vars + e_off[b];
        g_err = validate_request(net, rq);
        if (!g_err.empty()) return -1;
        for (int i = 0; i < rq.ne; ++i) {
            const int32_t c = e_codes[e_off[b] + i];
            if (c < 0 || c >= card[rq.evars[i]]) skip[(size_t)b] = 1;  // label outside the domain: empty posterior
        }
    }
    for (int64_t i = out_off[0]; i < out_off[B]; ++i) out[i] = 0.0;
    ThreadPool pool(std::max(1, threads));
    std::vector<ProgBuf> bufs;
    BatchPlan bp;
    plan_batch(net, pool, bufs, 0, B, q_off, q_vars, e_off, e_vars, e_codes, out_off, skip.data(), bp);
    g_err = bp.err;
    if (!g_err.empty()) { for (auto &b : bufs) b.release(); return -6; }
    if (stats) { stats[0] = bp.st.alg_bytes; stats[1] = bp.st.alg_flops; stats[2] = bp.st.n_steps; stats[3] = bp.st.max_step_cells; stats[4] = (double)bp.arena_cells; }
    Schedule sc;
    build_schedule(net, bp, bufs, 0, B, sc);
    std::vector<double> arena((size_t)sc.arena_cells + 16, -1e300);  // poison: reading unwritten scratch shows up
    double *res = out + out_off[0];  // the programs address the result buffer relative to the batch's first cell
    int rc = 0;
    double bytes_check = 0;
    size_t wg_seen = 0;
    int last_level = -1;
    std::vector<int> req_level((size_t)B, -1);  // dependencies: the items of a request run in strictly increasing levels
    for (const Launch &L : sc.launches) {
        bytes_check += L.alg_bytes;
        if (L.wg_first != wg_seen) { g_err = "launches do not tile wg_item"; rc = -11; break; }
        if (L.level < last_level) { g_err = "launches out of level order"; rc = -11; break; }
        last_level = L.level;
        wg_seen += L.grid;
        for (size_t wgi = L.wg_first; wgi < L.wg_first + L.grid && rc == 0; ++wgi) {
            const uint32_t k = sc.wg_item[wgi];
            if (k < L.first || k >= L.first + L.count) { g_err = "workgroup mapped to an item of another launch"; rc = -11; break; }
            const Item &it = sc.items[k];
            if (it.req >= (uint32_t)B) { g_err = "item of an unknown request"; rc = -11; break; }
            const uint32_t wg = (uint32_t)(wgi - L.wg_level);  // level-relative index = blockIdx.x + wg_base
            const uint32_t *prog = bufs[(size_t)bp.thread_of[it.req]].data + bp.local_off[it.req];
            const uint32_t *p = prog + it.rel_off;
            const int64_t need = bp.arena_need[it.req];
            const bool first_wg = (it.a & kItemSegment) || wg == it.b;
            if (first_wg) {
                if (req_level[it.req] >= L.level) { g_err = "two items of a request in one level"; rc = -11; break; }
                req_level[it.req] = L.level;
            }
            if (it.a & kItemSegment) {
                // a workgroup of segments: item k and the it.b - 1 items behind it, one per wave (planner.h kSegPerWg)
                if (L.kid != kKidSeg || it.b < 1 || it.b > (uint32_t)kSegPerWg || k + it.b > L.first + L.count) { g_err = "segment group inconsistent"; rc = -11; break; }
                if ((k - L.first) % kSegPerWg != 0) { g_err = "segment group not aligned"; rc = -11; break; }
                for (uint32_t wv = 0; wv < it.b && rc == 0; ++wv) {
                const Item &sg = sc.items[k + wv];
                if (!(sg.a & kItemSegment) || (wv > 0 && sg.b != 0) || sg.req >= (uint32_t)B) { g_err = "segment group member inconsistent"; rc = -11; break; }
                if (wv > 0) {
                    if (req_level[sg.req] >= L.level) { g_err = "two items of a request in one level"; rc = -11; break; }
                    req_level[sg.req] = L.level;
                }
                const uint32_t *p = bufs[(size_t)bp.thread_of[sg.req]].data + bp.local_off[sg.req] + sg.rel_off;
                const int64_t need = bp.arena_need[sg.req];
                const Item &it = sg;  // (the steps below belong to this wave's item)
                const uint32_t n_steps = it.a & ~kItemSegment;
                for (uint32_t s = 0; s < n_steps && rc == 0; ++s) {
                    if ((p[0] & 0xff) != kKindGeneric) { g_err = "FIBER / SWEEP step inside a segment"; rc = -10; break; }
                    rc = exec_step(net, p, 0, p[3], arena, (int64_t)sc.arena_off[it.req], need, res);
                    if (rc == 0 && ((p[1] >> 16) & kFlagFinal)) {
                        const uint64_t oo = (uint64_t)p[4] | ((uint64_t)p[5] << 32);
                        const int64_t cells = (int64_t)p[2] * (int64_t)p[3];
                        double total = 0;
                        for (int64_t i = 0; i < cells; ++i) total += res[oo + i];
                        if (total > 0)
                            for (int64_t i = 0; i < cells; ++i) res[oo + i] /= total;
                    }
                    p += p[6];
                }
                }
            } else {
                if (kernel_id_of_step(p) != L.kid) { g_err = "tile scheduled under the wrong class"; rc = -11; break; }
                if (it.a < 1 || it.a > (uint32_t)kTileMax || wg < it.b) { g_err = "tile geometry out of range"; rc = -11; break; }
                const int64_t h0 = (int64_t)(wg - it.b) * it.a;
                const int64_t h1 = std::min<int64_t>(p[3], h0 + it.a);
                if (h0 >= (int64_t)p[3]) { g_err = "workgroup beyond the step's last tile"; rc = -11; break; }
                // every tile is executed exactly once: the first tile of an item checks the item's tile count
                if (wg == it.b) {
                    const uint32_t tiles = (p[3] + it.a - 1) / it.a;
                    for (uint32_t t = 0; t < tiles; ++t)
                        if (wgi + t >= sc.wg_item.size() || sc.wg_item[wgi + t] != k) { g_err = "item does not own all of its tiles"; rc = -11; break; }
                    if (rc == 0 && wgi + tiles < L.wg_first + L.grid && sc.wg_item[wgi + tiles] == k) { g_err = "item owns too many workgroups"; rc = -11; }
                }
                if (rc == 0) rc = exec_step(net, p, h0, h1, arena, (int64_t)sc.arena_off[it.req], need, res);
            }
        }
    }
    if (rc == 0 && wg_seen != sc.wg_item.size()) { g_err = "launches do not cover wg_item"; rc = -11; }
    if (rc == 0 && std::fabs(bytes_check - bp.st.alg_bytes) > 64.0 * bp.st.n_steps + 1e-9 * bp.st.alg_bytes) {
        g_err = "schedule bytes do not add up to the plan's algorithmic bytes";
        rc = -12;
    }
    for (auto &b : bufs) b.release();
    return rc;
}

extern "C" int plan_sim_query(int32_t n_vars, const int32_t *card, const int64_t *scope_off, const int32_t *scope_vars,
                              const int64_t *value_off, const double *values, int32_t n_hints, const int32_t *hints,
                              int32_t nq, const int32_t *qvars, int32_t ne, const int32_t *evars, const int32_t *ecodes,
                              double *out, double *stats /* bytes, flops, steps, max_cells, arena_cells */) {
    // a one-request batch
    int64_t q_off[2] = {0, nq}, e_off[2] = {0, ne}, out_off[2] = {0, 1};
    for (int i = 0; i < nq; ++i) out_off[1] *= card[qvars[i]];
    const int32_t zero = 0;
    return plan_sim_query_batch(n_vars, card, scope_off, scope_vars, value_off, values, n_hints, hints, 1, q_off, qvars, e_off,
                                ne ? evars : &zero, ne ? ecodes : &zero, out_off, out, 1, 1, stats);
}

// debugging aid: return the raw step program of one request (words copied into `out`, count returned)
// Plan templates (planner.cpp, plan_batch): the same batch planned by one worker with and without the template cache
// must give the same programs, word for word, the same items and the same statistics.  Returns the number of requests
// answered from a template (>= 0), or < 0 with plan_sim_error() set.
static double g_cache_ms[2];
extern "C" void plan_sim_cache_times(double *out) { out[0] = g_cache_ms[0]; out[1] = g_cache_ms[1]; }
extern "C" int64_t plan_sim_cache_check(int32_t n_vars, const int32_t *card, const int64_t *scope_off, const int32_t *scope_vars,
                                        const int64_t *value_off, const double *values, int64_t B, const int64_t *q_off,
                                        const int32_t *q_vars, const int64_t *e_off, const int32_t *e_vars, const int32_t *e_codes) {
    Network net;
    g_err = net.set(n_vars, card, scope_off, scope_vars, value_off, values);
    if (!g_err.empty()) return -1;
    net.small_cells = g_small_cells;
    net.big_iters = g_big_iters;
    net.tile_h = g_tile_h;
    net.fuse = g_fuse;
    net.chain = g_chain;
    net.sweep = g_sweep;
    net.sweep_min = g_sweep_min;
    net.order_effort = g_order_effort; net.second_above = g_second_above;
    net.prune = g_prune;
    std::vector<int64_t> out_off(B + 1, 0);
    for (int64_t b = 0; b < B; ++b) {
        int64_t cells = 1;
        for (int64_t k = q_off[b]; k < q_off[b + 1]; ++k) cells *= card[q_vars[k]];
        out_off[b + 1] = out_off[b] + cells;
    }
    const int n_threads = getenv("PLAN_SIM_THREADS") ? atoi(getenv("PLAN_SIM_THREADS")) : 1;  // (> 1: timing only, the comparison needs one worker)
    ThreadPool pool(n_threads);
    std::vector<ProgBuf> bufs[2];
    BatchPlan bp[2];
    for (int pass = 0; pass < 2; ++pass) {
        net.plan_cache = pass;  // pass 0: every request planned; pass 1: templates
        auto t0 = std::chrono::steady_clock::now();
        plan_batch(net, pool, bufs[pass], 0, B, q_off, q_vars, e_off, e_vars, e_codes, out_off.data(), nullptr, bp[pass]);
        g_cache_ms[pass] = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        if (!bp[pass].err.empty()) { g_err = bp[pass].err; return -2; }
    }
    int64_t rc = 0;
    if (n_threads > 1) { for (auto &v : bufs) for (auto &b : v) b.release(); return 0; }
    if (bp[0].total_words != bp[1].total_words || std::memcmp(bufs[0][0].data, bufs[1][0].data, bp[0].total_words * 4) != 0) { g_err = "programs differ"; rc = -3; }
    else if (bp[0].tags[0].size() != bp[1].tags[0].size() || std::memcmp(bp[0].tags[0].data(), bp[1].tags[0].data(), bp[0].tags[0].size() * sizeof(Tag)) != 0) { g_err = "items differ"; rc = -4; }
    else if (bp[0].arena_need != bp[1].arena_need || bp[0].cost != bp[1].cost || bp[0].st.alg_bytes != bp[1].st.alg_bytes || bp[0].st.n_steps != bp[1].st.n_steps) { g_err = "statistics differ"; rc = -5; }
    else {
        // how many requests repeat an earlier shape (what the cache can answer from a template)
        std::vector<std::string> seen;
        for (int64_t b = 0; b < B; ++b) {
            std::string key((const char *)(q_vars + q_off[b]), 4 * (size_t)(q_off[b + 1] - q_off[b]));
            key.push_back('|');
            key.append((const char *)(e_vars + e_off[b]), 4 * (size_t)(e_off[b + 1] - e_off[b]));
            if (std::find(seen.begin(), seen.end(), key) != seen.end()) ++rc; else seen.push_back(key);
        }
    }
    for (auto &v : bufs) for (auto &b : v) b.release();
    return rc;
}

extern "C" int64_t plan_sim_program(int32_t n_vars, const int32_t *card, const int64_t *scope_off, const int32_t *scope_vars,
                                    const int64_t *value_off, const double *values, int32_t n_hints, const int32_t *hints,
                                    int32_t nq, const int32_t *qvars, int32_t ne, const int32_t *evars, const int32_t *ecodes,
                                    uint32_t *out, int64_t cap) {
    Network net;
    g_err = net.set(n_vars, card, scope_off, scope_vars, value_off, values);
    if (!g_err.empty()) return -1;
    net.small_cells = g_small_cells;
    net.big_iters = g_big_iters;
    net.tile_h = g_tile_h;
    net.fuse = g_fuse;
    net.chain = g_chain;
    net.sweep = g_sweep;
    net.sweep_min = g_sweep_min;
    net.order_effort = g_order_effort; net.second_above = g_second_above;
    net.prune = g_prune;
    net.set_hints(n_hints, hints);
    Request rq;
    rq.nq = nq; rq.qvars = qvars; rq.ne = ne; rq.evars = evars; rq.ecodes = ecodes; rq.out_off = 0;
    std::vector<uint32_t> prog;
    PlanStats st;
    g_err = plan_request(net, rq, prog, st);
    if (!g_err.empty()) return -6;
    if ((int64_t)prog.size() > cap) return -2;
    std::memcpy(out, prog.data(), prog.size() * 4);
    return (int64_t)prog.size();
}

// The wave-cooperative device planner (csrc/wave_plan.h) compiled for the host - one lane runs every iteration of its lane loops
// (wave_prims.h) - on one request: the program in `out` (words returned; -3: the network / options are outside what it covers; -4: the
// request exceeds a device limit, kEmitErrDevice; -5: another emission error), its work items in `tags_out` (four words each),
// {bytes, flops, steps, max cells, arena cells, work items} in `stats`.  tests/test_wave_planner.py holds it against plan_sim_program.
extern "C" int64_t wave_plan_program(int32_t n_vars, const int32_t *card, const int64_t *scope_off, const int32_t *scope_vars,
                                     const int64_t *value_off, const double *values, int32_t n_hints, const int32_t *hints,
                                     int32_t nq, const int32_t *qvars, int32_t ne, const int32_t *evars, const int32_t *ecodes,
                                     int32_t no_prune, uint32_t *out, int64_t cap, uint32_t *tags_out, int64_t tags_cap, double *stats) {
    Network net;
    g_err = net.set(n_vars, card, scope_off, scope_vars, value_off, values);
    if (!g_err.empty()) return -1;
    net.small_cells = g_small_cells;
    net.big_iters = g_big_iters;
    net.tile_h = g_tile_h;
    net.fuse = g_fuse;
    net.chain = g_chain;
    net.sweep = g_sweep;
    net.sweep_min = g_sweep_min;
    net.order_effort = g_order_effort; net.second_above = g_second_above;
    net.prune = g_prune;
    net.set_hints(n_hints, hints);
    std::unique_ptr<WNet> wn(new WNet);
    if (!net.wave_view(*wn)) return -3;
    std::unique_ptr<WState> ws(new WState);
    std::vector<uint32_t> slot(((size_t)cap + kMaxStepWords) * (g_order_effort ? 2 : 1) + (g_order_effort ? kWStashWords + 4 * kMaxStepWords : 0));  // (effort 1: two programs and the stash share the slot)
    WResult R;
    wave_plan_request(*wn, *ws, net.anc2.data(), nq, qvars, ne, evars, ecodes, no_prune != 0, 0, slot.data(), (uint32_t)slot.size(), R);
    if (R.err == kEmitErrDevice) return -4;
    if (R.err) { g_err = emit_error_message(R.err); return -5; }
    if ((int64_t)R.words > cap || (int64_t)R.n_tags > tags_cap) return -2;
    std::memcpy(out, slot.data() + R.base, (size_t)R.words * 4);
    std::memcpy(tags_out, ws->e.tags, (size_t)R.n_tags * sizeof(Tag));
    stats[0] = R.alg_bytes; stats[1] = R.alg_flops; stats[2] = R.n_steps; stats[3] = R.max_step_cells; stats[4] = (double)R.arena_cells; stats[5] = (double)R.n_tags;
    return (int64_t)R.words;
}

// the host planner's program of the same request with its work items and statistics, in the same shape (the reference of the above)
extern "C" int64_t plan_sim_program_tags(int32_t n_vars, const int32_t *card, const int64_t *scope_off, const int32_t *scope_vars,
                                         const int64_t *value_off, const double *values, int32_t n_hints, const int32_t *hints,
                                         int32_t nq, const int32_t *qvars, int32_t ne, const int32_t *evars, const int32_t *ecodes,
                                         int32_t no_prune, uint32_t *out, int64_t cap, uint32_t *tags_out, int64_t tags_cap, double *stats) {
    Network net;
    g_err = net.set(n_vars, card, scope_off, scope_vars, value_off, values);
    if (!g_err.empty()) return -1;
    net.small_cells = g_small_cells;
    net.big_iters = g_big_iters;
    net.tile_h = g_tile_h;
    net.fuse = g_fuse;
    net.chain = g_chain;
    net.sweep = g_sweep;
    net.sweep_min = g_sweep_min;
    net.order_effort = g_order_effort; net.second_above = g_second_above;
    net.prune = g_prune;
    net.set_hints(n_hints, hints);
    Request rq;
    rq.nq = nq; rq.qvars = qvars; rq.ne = ne; rq.evars = evars; rq.ecodes = ecodes; rq.out_off = 0; rq.no_prune = no_prune != 0;
    std::vector<uint32_t> prog;
    PlanStats st;
    g_err = plan_request(net, rq, prog, st);
    if (!g_err.empty()) return -6;
    std::vector<Tag> tags;
    tag_program(net.emit_view(), prog.data(), [&](const Tag &t) { tags.push_back(t); });
    if ((int64_t)prog.size() > cap || (int64_t)tags.size() > tags_cap) return -2;
    std::memcpy(out, prog.data(), prog.size() * 4);
    std::memcpy(tags_out, tags.data(), tags.size() * sizeof(Tag));
    stats[0] = st.alg_bytes; stats[1] = st.alg_flops; stats[2] = st.n_steps; stats[3] = st.max_step_cells; stats[4] = (double)st.arena_cells; stats[5] = (double)tags.size();
    return (int64_t)prog.size();
}

// host-side planning throughput probe (no execution): plans B fixed-shape requests on `threads` threads
extern "C" double plan_sim_bench(int32_t n_vars, const int32_t *card, const int64_t *scope_off, const int32_t *scope_vars,
                                 const int64_t *value_off, const double *values, int32_t n_hints, const int32_t *hints,
                                 int64_t B, int32_t nq, const int32_t *qvars, int32_t ne, const int32_t *evars,
                                 const int32_t *ecodes, int threads, double *stats /* bytes, steps, words, schedule ms, items, workgroups */) {
    Network net;
    g_err = net.set(n_vars, card, scope_off, scope_vars, value_off, values);
    if (!g_err.empty()) return -1;
    net.chain = g_chain;
    net.sweep = g_sweep;
    net.sweep_min = g_sweep_min;
    net.order_effort = g_order_effort; net.second_above = g_second_above;
    net.set_hints(n_hints, hints);
    net.plan_cache = 0;  // PLANNING is what is timed: the second pass over the same requests must not be answered from templates of the first
    std::vector<int64_t> q_off(B + 1), e_off(B + 1), out_off(B + 1);
    for (int64_t b = 0; b <= B; ++b) { q_off[b] = b * nq; e_off[b] = b * ne; out_off[b] = b * 4; }
    BatchPlan bp;
    ThreadPool pool(threads);
    std::vector<ProgBuf> bufs;
    plan_batch(net, pool, bufs, 0, B, q_off.data(), qvars, e_off.data(), evars, ecodes, out_off.data(), nullptr, bp);  // warm-up
    auto t0 = std::chrono::steady_clock::now();
    plan_batch(net, pool, bufs, 0, B, q_off.data(), qvars, e_off.data(), evars, ecodes, out_off.data(), nullptr, bp);
    double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    Schedule sc;
    build_schedule(net, bp, bufs, 0, B, sc);  // warm-up (allocations)
    auto t1 = std::chrono::steady_clock::now();
    build_schedule(net, bp, bufs, 0, B, sc);
    const double sched_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t1).count();
    for (auto &b : bufs) b.release();
    if (stats) { stats[0] = bp.st.alg_bytes; stats[1] = bp.st.n_steps; stats[2] = (double)bp.total_words; stats[3] = sched_ms; stats[4] = (double)sc.items.size(); stats[5] = (double)sc.wg_item.size(); }
    g_err = bp.err;
    return ms;
}

// The device planner's configuration of emit_core.h, run on the CPU: the planning state of a request carved out of ONE raw
// (deliberately dirty) slice like a lane's, the program written into a fixed slot of `slot_words` words, the elimination order
// taken from order_search as bytes - compared word for word, work item for work item, with plan_request.  Returns the number
// of requests whose programs fit their slot (the others must report kEmitErrWords and leave no damage), or -1 and
// plan_sim_error().
extern "C" int64_t plan_sim_device_style(int32_t n_vars, const int32_t *card, const int64_t *scope_off, const int32_t *scope_vars,
                                         const int64_t *value_off, const double *values, int32_t n_hints, const int32_t *hints,
                                         int64_t B, const int64_t *q_off, const int32_t *q_vars, const int64_t *e_off,
                                         const int32_t *e_vars, const int32_t *e_codes, int32_t slot_words, int32_t no_prune) {
    Network net;
    g_err = net.set(n_vars, card, scope_off, scope_vars, value_off, values);
    if (!g_err.empty()) return -1;
    if (n_vars > 128) { g_err = "the device planner covers networks of up to 128 variables"; return -1; }
    net.small_cells = g_small_cells; net.big_iters = g_big_iters; net.tile_h = g_tile_h; net.fuse = g_fuse; net.chain = g_chain;
    net.sweep = g_sweep; net.sweep_min = g_sweep_min; net.prune = g_prune;
    net.order_effort = g_order_effort; net.second_above = g_second_above;
    net.set_hints(n_hints, hints);
    const EmitNet en = net.emit_view();
    const OrderNet on = net.order_view();
    const size_t bytes = emit_scratch_bytes(n_vars);
    std::vector<char> slice(bytes + 64);
    std::vector<uint32_t> slot((size_t)slot_words);
    std::vector<OrderScratch> os(1);
    int64_t fitted = 0;
    for (int64_t b = 0; b < B; ++b) {
        const int nq = (int)(q_off[b + 1] - q_off[b]), ne = (int)(e_off[b + 1] - e_off[b]);
        const int32_t *qv = q_vars + q_off[b], *ev = e_vars + e_off[b], *ec = e_codes + e_off[b];
        std::memset(slice.data(), 0xA5 + (int)(b & 7), slice.size());  // a lane's slice is never cleared between requests
        std::fill(slot.begin(), slot.end(), 0xDEADBEEFu);
        char *base = slice.data() + ((64 - (reinterpret_cast<uintptr_t>(slice.data()) & 63)) & 63);
        EmitScratch S;
        emit_scratch_carve(S, base, n_vars);
        order_search(on, os[0], nq, qv, ne, ev, no_prune != 0);
        int err = emit_begin(en, S, nq, qv, ne, ev, ec, no_prune != 0);
        EmitBuf buf;
        buf.data = slot.data();
        buf.cap = (size_t)slot_words;
        EmitStats st;
        if (!err) err = emit_run(en, S, buf, st, nullptr, nq, qv, 1000 * b, os[0].best, (int)os[0].n_best);
        // the host's planner on the same request
        Request rq;
        rq.nq = nq; rq.qvars = qv; rq.ne = ne; rq.evars = ev; rq.ecodes = ec; rq.out_off = 1000 * b; rq.no_prune = no_prune != 0;
        std::vector<uint32_t> ref;
        PlanStats rs;
        g_err = plan_request(net, rq, ref, rs);
        if (!g_err.empty()) return -1;
        if (err == kEmitErrWords) {
            if (ref.size() + kMaxStepWords <= (size_t)slot_words) { g_err = "request " + std::to_string(b) + ": refused although the program fits its slot"; return -1; }
            continue;
        }
        if (err) { g_err = "request " + std::to_string(b) + ": " + emit_error_message(err); return -1; }
        if (buf.size != ref.size() || std::memcmp(slot.data(), ref.data(), ref.size() * 4) != 0) {
            g_err = "request " + std::to_string(b) + ": the slot's program differs from plan_request's (" + std::to_string(buf.size) + " vs " + std::to_string(ref.size()) + " words)";
            return -1;
        }
        if (st.alg_bytes != rs.alg_bytes || st.n_steps != rs.n_steps || st.arena_cells != rs.arena_cells) { g_err = "request " + std::to_string(b) + ": statistics differ"; return -1; }
        uint32_t n_items = 0;
        tag_program(en, slot.data(), [&](const Tag &t) { n_items += t.rel_off > 0 && t.wgs > 0; });
        if (!n_items) { g_err = "request " + std::to_string(b) + ": no work items"; return -1; }
        ++fitted;
    }
    return fitted;
}
