// Program emission - request + elimination order -> step program - ONE implementation, compiled for the host (planner.cpp:
// the planning workers) and for the device (emit_kernel in engine.hip: one request per lane), so that both write the same
// program, word for word.  Like order_search.h: plain arrays and fixed-size bit sets, no heap, no std containers, no
// recursion; the whole planning state of a request is an `EmitScratch` (host: backed by per-thread vectors; device: a slice
// of one buffer per lane).
//
// Replaces the bookkeeping half of BayesNet._variable_elimination (sorobn/bayes_net.py:763-789): evidence slicing of the
// CPTs (768-776), the elimination loop's factor selection (779-786) and the final product (789-790) - see planner.h for the
// role of the planner and for the encoding of the step programs.
#pragma once
#include <cstdint>

#include "order_search.h"

namespace mibn {

constexpr int kMaxVars = 1024;            // bitset capacity
constexpr int kWords = kMaxVars / 64;
constexpr int kMaxIn = 6;                 // input factors per step (larger products are pre-multiplied)
constexpr int kMaxAxes = 32;              // output axes per step after merging
constexpr int kLoTarget = 256;            // lane-varying block: first axes whose product reaches this
constexpr int kLoMax = 512;               // ... but never more than this many cells (2 per lane)
constexpr uint64_t kConstFlag = 1ull << 63;  // in_off bit: table lives in the constants pool
constexpr int kSweepItersDefault = 8;     // SWEEP: tiles per workgroup (Network::sweep_iters)

// The device planner covers networks of up to 128 variables: there every bit set is two words, the loops below have a
// compile-time bound, unroll, and a local set lives in registers instead of the lane's private memory.
#if defined(__HIP_DEVICE_COMPILE__)
#define MIBN_BITS_WORDS(nw) 2
#else
#define MIBN_BITS_WORDS(nw) (nw)
#endif

struct Bits {
    int nw = kWords;  // words in use (first: in the same cache line as w[0], w[1] - all a network of <= 128 variables touches)
    uint64_t w[kWords] = {};
    MIBN_HD void set(int i) { w[i >> 6] |= 1ull << (i & 63); }
    MIBN_HD void clr(int i) { w[i >> 6] &= ~(1ull << (i & 63)); }
    MIBN_HD bool test(int i) const { return (w[i >> 6] >> (i & 63)) & 1; }
    MIBN_HD bool any() const { for (int k = 0; k < MIBN_BITS_WORDS(nw); ++k) if (w[k]) return true; return false; }
    MIBN_HD void or_(const Bits &o) { for (int k = 0; k < MIBN_BITS_WORDS(nw); ++k) w[k] |= o.w[k]; }
    MIBN_HD void andnot(const Bits &o) { for (int k = 0; k < MIBN_BITS_WORDS(nw); ++k) w[k] &= ~o.w[k]; }
    MIBN_HD bool intersects(const Bits &o) const { for (int k = 0; k < MIBN_BITS_WORDS(nw); ++k) if (w[k] & o.w[k]) return true; return false; }
    MIBN_HD int count() const { int c = 0; for (int k = 0; k < MIBN_BITS_WORDS(nw); ++k) c += __builtin_popcountll(w[k]); return c; }
    template <class F> MIBN_HD void for_each(F f) const {
        for (int k = 0; k < MIBN_BITS_WORDS(nw); ++k) { uint64_t m = w[k]; while (m) { int b = __builtin_ctzll(m); f(k * 64 + b); m &= m - 1; } }
    }
};

// (the encoding of the step programs is documented in planner.h)
constexpr uint32_t kFlagFinal = 1, kFlagContig = 2, kFlagOuter = 4, kFlagChain = 8;
constexpr uint32_t kKindSweep = 2;
constexpr int kSweepTileCells = 8192;  // 64 KiB of LDS
constexpr int kSweepMaxT = 1024;       // T cells of all stages together (8 KiB)
constexpr int kSweepMaxSmall = 16;     // small inputs of all stages together
constexpr int kSweepWG = 512;
constexpr int kSweepStageWords = 5, kSweepSmallWords = 7;
constexpr uint32_t kFlagSweepCanon = 16;  // stage j contracts digit k-1-j (first eliminated = slowest axis, the layout rule): the
                                          // kernel's compile-time stage geometry applies
// the loop digit of a stage (the digit a lane's four fibers differ in): the highest one that is neither contracted nor -
// the usual ctrl of a grid sweep - its lower neighbour
MIBN_HD constexpr int sweep_loop_digit(int k, int dig) {
    for (int d = k - 1; d >= 0; --d)
        if (d != dig && d != dig - 1) return d;
    for (int d = k - 1; d >= 0; --d)  // k = 2: the only other digit, lower neighbour or not
        if (d != dig) return d;
    return -1;
}
constexpr int kRowStrideShift = 20;  // w1 bits 20..27
constexpr int kHdrWords = 10;
constexpr uint32_t kKindGeneric = 0, kKindFiber = 1;  // (kKindSweep = 2 above)
constexpr int kMaxNC = 16;         // N-fiber length held in registers
constexpr int kMaxT = 2048;        // T cells (16 KiB of LDS)
constexpr int kMaxSmall = 4;
constexpr int kFiberLoMax = 256;    // R cells in the lane-varying block of a FIBER step (1 per lane)
constexpr int kMaxCx = 16;          // eliminated combinations of a FIBER step (register fiber of loads)
constexpr int kMaxStepWords = 384;  // LDS copy of one step descriptor
constexpr int kTileMax = 64;        // hi iterations per tile (their offsets are decoded in one go)

// classes of work (see Launch in planner.h)
constexpr int kKidSeg = 0;         // segments of small GENERIC steps
constexpr int kKidFiber0 = 1;      // 36 FIBER tile classes: 1 + (n_big-1)*18 + cx_class*6 + nc_class
constexpr int kKidChain = 36;      // CHAIN steps (takes the id of the impossible FIBER class <2, cxN, outer-mfma>)
constexpr int kKidGeneric0 = 37;   // 6 GENERIC tile classes: 37 + (n_in - 1)
constexpr int kKidSweep = 43;      // SWEEP steps: a kernel of their own (sweep_kernel.hip.h), launched beside the level kernel
constexpr int kNumKernels = 44;
constexpr uint32_t kItemSegment = 1u << 31;

// Phase timers of emit_kernel (tools/gpu_r03_emit_phases.sh builds the library with -DMIBN_EMIT_PROF): a tick books the time since
// the previous one to phase k - per lane, so under divergence a phase also holds the time a lane waits for the others' branches.
#if defined(MIBN_EMIT_PROF)  // (only the translation unit of the kernels is built with it)
struct EmitProf { unsigned long long t, a[12]; };
extern EmitProf g_host_emit_prof;  // (planner.cpp, host build with the flag: tools/planner_prof.cpp prints it)
#define MIBN_PROF_ARG , EmitProf &prof_
#define MIBN_PROF_PASS , prof_
#if defined(__HIP_DEVICE_COMPILE__)
#define MIBN_TICK(k) { const unsigned long long t_ = __builtin_amdgcn_s_memrealtime(); prof_.a[k] += t_ - prof_.t; prof_.t = t_; }
#else
#define MIBN_TICK(k) { const unsigned long long t_ = __builtin_ia32_rdtsc(); prof_.a[k] += t_ - prof_.t; prof_.t = t_; }
#endif
#else
#define MIBN_PROF_ARG
#define MIBN_PROF_PASS
#define MIBN_TICK(k)
#endif

template <class T> MIBN_HD constexpr T emit_max(T a, T b) { return a < b ? b : a; }
template <class T> MIBN_HD constexpr T emit_min(T a, T b) { return b < a ? b : a; }

// What the emission reads of a network and of its options (pointers into host vectors, or into one device buffer).
struct EmitNet {
    int32_t n_vars = 0, nw = 1;
    const int32_t *card = nullptr;        // [n]
    const double *log2card = nullptr;     // [n]
    const int64_t *pool_off = nullptr;    // [n] offset (doubles) of CPT v in the constants pool
    const int32_t *scope_off = nullptr;   // [n + 1] CSR of the CPT scopes ([*parents, v])
    const int32_t *scope_vars = nullptr;
    const int64_t *scope_stride = nullptr;  // dense C-order stride of every scope entry
    const uint64_t *anc = nullptr;        // [n][nw] ancestors (bayes_net.py:373-378)
    int32_t small_cells = 1024;
    int64_t big_iters = 4096;
    double log2_small = 10, log2_big = 12;  // log2 of the two, computed once on the host (host and device must agree to the bit)
    int32_t uniform_log2 = -1;            // l >= 0: every multi-state variable has 2^l states and the others one (OrderNet::uniform_log2): a
                                          // scope's log2 cells is l x its popcount, exactly the sum the loop over its variables computes
    int32_t prune = 1, outer = 1, fuse = 1, chain = 1, sweep = 5, sweep_min = 2, sweep_canon = 1;
    int32_t tile_h = 0, sweep_iters = kSweepItersDefault;
    int64_t tile_bytes = 512 << 10;
};

// errors of the emission (planner.cpp turns them into the messages of mibn_last_error)
constexpr int kEmitErrStepAxes = 1, kEmitErrFactorAxes = 2, kEmitErrCells = 3, kEmitErrCptAxes = 4, kEmitErrPool = 5,
              kEmitErrWords = 6;  // 6: the program does not fit its slot (device only: the chunk is then planned on the host)

// Where the words go.  Host: a view of a growable ProgBuf (planner.h); device: the request's slot of the chunk's program buffer,
// whose last kMaxStepWords words take the writes of a step that no longer fits (kEmitErrWords).
struct ProgBuf;
struct EmitBuf {
    uint32_t *data = nullptr;
    size_t size = 0, cap = 0;
    ProgBuf *host = nullptr;
    bool overflow = false;
    MIBN_HD uint32_t *extend(size_t words);
    MIBN_HD void push(uint32_t w) { *extend(1) = w; }
};

struct EmitStats {
    double alg_bytes = 0, alg_flops = 0, n_steps = 0, max_step_cells = 0;
    int64_t arena_cells = 0;  // scratch cells this request needs in its arena slot
    int64_t out_cells = 0;
};

// host only (planner.cpp): growth of a ProgBuf-backed buffer, and the plan templates' record of where a program depends on the
// evidence codes / on the request's position in the batch
#if !defined(__HIP_DEVICE_COMPILE__)
uint32_t *emit_buf_grow(EmitBuf &b, size_t words);
void emit_rec_const(void *rec, uint32_t word, int32_t cpt_var);
void emit_rec_final(void *rec, uint32_t word);
#endif

MIBN_HD inline uint32_t *EmitBuf::extend(size_t words) {
#if !defined(__HIP_DEVICE_COMPILE__)
    if (host) return emit_buf_grow(*this, words);
#endif
    if (size + words + kMaxStepWords > cap) { overflow = true; return data + (cap - kMaxStepWords); }
    uint32_t *p = data + size;
    size += words;
    return p;
}

constexpr int kRawAxes = 40;  // axes of one factor before merging (cells < 2^31 => <= 31 non-trivial axes)

struct alignas(64) PF {  // planning-time factor (plain data).  Field order = cache lines of a typical factor (<= 8 axes): the header and
             // vars[0..8) share one, strides[0..8) is the next one touched, the scope's live words the third
    int n = 0;                   // axes
    int32_t src = -1;            // initial factor: the variable whose CPT it slices (its offset depends on the evidence codes)
    uint64_t off = 0;            // arena offset, or pool offset | kConstFlag
    int64_t cells = 0;           // product of the free cardinalities
    int64_t alloc = 0;           // arena cells owned (0 for constants)
    int32_t vars[kRawAxes];
    int64_t strides[kRawAxes];   // stride (doubles) per axis
    Bits scope;                  // free (non-evidence) variables
    MIBN_HD PF() {}              // (user-provided: a new pool entry does not zero the 480 bytes of vars / strides)
};

MIBN_HD inline void pf_reset(PF &f) {  // a fresh pool entry (vars / strides / scope are written by whoever fills it)
    f.n = 0;
    f.off = 0;
    f.cells = 0;
    f.alloc = 0;
    f.src = -1;
}

struct Arena {
    static constexpr int kMaxBlocks = 64;
    int64_t foff[kMaxBlocks], fsz[kMaxBlocks];  // free list sorted by offset
    int nf = 0;
    int64_t top = 0;
    MIBN_HD int64_t alloc(int64_t n) {
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
    MIBN_HD void release(int64_t o, int64_t n) {
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

struct Emitter {
    const EmitNet &net;
    EmitBuf &prog;
    EmitStats &st;
    Arena arena;
    double *key;    // layout key per variable: larger = lives longer = faster axis
    int32_t *pos;   // variable -> output axis (scratch, -1 outside emit)
    int err = 0;    // kEmitErr*
    void *rec = nullptr;  // optional (host): where the program depends on evidence codes / batch position

    // the (off lo, off hi) pair of input table f
    MIBN_HD void put_off(uint32_t *&p, const PF *f) {
#if !defined(__HIP_DEVICE_COMPILE__)
        if (rec && f->src >= 0) emit_rec_const(rec, (uint32_t)(p - prog.data), f->src);
#endif
        *p++ = (uint32_t)(f->off & 0xffffffffu);
        *p++ = (uint32_t)(f->off >> 32);
    }

    MIBN_HD void header(uint32_t *w, uint32_t kind, int n_in, int ma, int mlo, int cx, bool final_, int64_t lo, int64_t hi,
                uint64_t out_off, int words) {
        w[0] = kind | ((uint32_t)n_in << 8) | ((uint32_t)ma << 16) | ((uint32_t)mlo << 24);
        w[1] = (uint32_t)cx | ((final_ ? kFlagFinal : 0u) << 16);
        w[2] = (uint32_t)lo;
        w[3] = (uint32_t)hi;
        w[4] = (uint32_t)(out_off & 0xffffffffu);
        w[5] = (uint32_t)(out_off >> 32);
#if !defined(__HIP_DEVICE_COMPILE__)
        if (rec && final_) emit_rec_final(rec, (uint32_t)(w - prog.data) + 4);
#endif
        w[6] = (uint32_t)words;
        w[7] = w[8] = w[9] = 0;
    }

    using Strides = int64_t[kMaxIn][kRawAxes];
    using XStrides = int64_t[kMaxIn][3];

    // GENERIC encoding: iteration space = output cells
    MIBN_HD void emit_generic(const PF *const *ins, int n_in, const Strides &s, const int64_t *xs, const PF &out, int64_t cells,
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
        if (ma > kMaxAxes) { err = kEmitErrStepAxes; return; }
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
    MIBN_HD bool emit_fiber(const PF *const *ins, int n_in, const Strides &s, const XStrides &xs, const PF &out, int cx, int c1) {
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
    MIBN_HD bool emit_outer(const PF *const *ins, int n_in, const Strides &s, const XStrides &xs, const PF &out, int cx, int c1) {
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
        int64_t best_key = 0x7fffffffffffffffll;
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
    MIBN_HD bool emit_chain(const PF *const *ins, int n_in, const Strides &s, const XStrides &xs_in, PF &out) {
        const int kPerm[3][3] = {{0, 1, 2}, {0, 2, 1}, {1, 2, 0}};
        for (int pi = 0; pi < 3; ++pi) {
            XStrides xs;
            for (int j = 0; j < n_in; ++j)
                for (int k = 0; k < 3; ++k) xs[j][k] = xs_in[j][kPerm[pi][k]];
            if (emit_chain_as(ins, n_in, s, xs, out)) return true;
        }
        return false;
    }

    MIBN_HD bool emit_chain_as(const PF *const *ins, int n_in, const Strides &s_in, const XStrides &xs, PF &out) {
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
    MIBN_HD bool emit_sweep(const PF *const *ins, int n_in, const int *X, int k, PF &out) {
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
        for (int q = 0; q < MIBN_BITS_WORDS(net.nw); ++q) out.scope.w[q] = 0;
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
        st.max_step_cells = emit_max(st.max_step_cells, (double)F->cells);
        st.n_steps += 1;
        for (int j = 0; j < n_in; ++j)
            if (ins[j]->alloc) arena.release((int64_t)ins[j]->off, ins[j]->alloc);
        return true;
    }

    // Emit one step: multiply `ins`, sum out the nx (0..2) variables X (nx = 0: product only); the new factor is
    // written to `out`.  fiber_only: emit nothing and return false unless the step fits the FIBER form (used to try
    // the joint elimination of two variables).
    MIBN_HD bool emit(const PF *const *ins, int n_in, const int *X, int nx, bool final_, int64_t final_off, PF &out, bool fiber_only) {
        out.scope.nw = net.nw;  // (only the words the network uses: the rest of a pool entry's scope is never read)
        for (int k = 0; k < MIBN_BITS_WORDS(net.nw); ++k) out.scope.w[k] = 0;
        for (int j = 0; j < n_in; ++j) out.scope.or_(ins[j]->scope);
        for (int k = 0; k < nx; ++k) out.scope.clr(X[k]);
        int na = 0;
        bool overflow = false;
        out.scope.for_each([&](int v) { if (na < kRawAxes) out.vars[na++] = v; else overflow = true; });
        if (overflow) { if (!fiber_only) err = kEmitErrFactorAxes; return false; }
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
            if (cells >= (1ll << 31)) { if (!fiber_only) err = kEmitErrCells; return false; }
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
        // (a step of fewer than big_iters output cells runs in the segment interpreter, GENERIC form: every streaming form ends with
        //  that test - emit_fiber `rcells * NC`, emit_outer `rcells * 16`, emit_chain_as `rcells * 64` are all the output's cells -
        //  so two thirds of the steps skip their layout work here)
        const bool streaming = !final_ && cells >= net.big_iters;
        const bool fiber = !streaming ? false
                           : nx == 3  ? emit_chain(ins, n_in, s, xs, out)
                                      : ((net.outer && nx > 0 && emit_outer(ins, n_in, s, xs, out, cx, c1)) || emit_fiber(ins, n_in, s, xs, out, cx, c1));
        if (!fiber) {
            if (fiber_only) {
                if (out.alloc) arena.release((int64_t)out.off, out.alloc);
                return false;
            }
            int64_t xs1[kMaxIn];
            for (int j = 0; j < n_in; ++j) xs1[j] = xs[j][0];
            emit_generic(ins, n_in, s, xs1, out, cells, cx, final_);
        }
        if (err) return false;
        prog.data[step_base + 9] = (uint32_t)(((int64_t)in_cells + cells + 2) >> 2);  // section-8(d) cells of this step, units of 4
        st.alg_bytes += 8.0 * (in_cells + (double)cells);
        double pc = (double)cells;  // cells of the product scope = the output's cells x the eliminated cardinalities
        for (int k = 0; k < nx; ++k) pc *= net.card[X[k]];
        st.alg_flops += n_in * pc;
        st.max_step_cells = emit_max(st.max_step_cells, pc);
        st.n_steps += 1;
        for (int j = 0; j < n_in; ++j)
            if (ins[j]->alloc) arena.release((int64_t)ins[j]->off, ins[j]->alloc);
        return true;
    }
};

// ------------------------------------------------------------------------------------ one request
// The planning state of one request.  pool_cap >= 3 * n_vars + 64 factor slots (every elimination creates one; products of more
// than kMaxIn factors a few more), sw = words of a slot set.
struct EmitScratch {
    PF *pool = nullptr;
    int32_t pool_cap = 0, sw = 0;
    double *key = nullptr;        // [n_vars]
    int32_t *pos = nullptr;       // [n_vars]
    int32_t *ecode = nullptr;     // [n_vars] evidence code per variable
    uint64_t *mem = nullptr;      // [n_vars][sw] per variable: the factor slots whose scope contains it
    uint64_t *alive = nullptr;    // [sw] factor slots not yet consumed
    const PF **ins = nullptr;     // [n_vars + 8]
    // filled by emit_begin
    Bits rel, hidden;             // relevant = query | event | ancestors (bayes_net.py:763-765); hidden = relevant - query - event (766)
    int32_t n_pool = 0, n0 = 0;   // factors in the pool; n0 = the evidence-sliced CPTs (the first n0 of them)
};
MIBN_HD inline int emit_pool_cap(int n_vars) { return 3 * n_vars + 64; }
// bytes of one request's scratch behind a 16-byte aligned base (device: one slice per lane), and the carving of it
MIBN_HD inline size_t emit_scratch_bytes(int n_vars) {
    const size_t cap = (size_t)emit_pool_cap(n_vars), sw = (cap + 63) / 64, n = (size_t)n_vars;
    size_t b = cap * sizeof(PF);
    b += n * 8;                      // key
    b += (n * sw + sw) * 8;          // mem, alive
    b += (n + 8) * sizeof(void *);   // ins
    b += 2 * ((n * 4 + 15) & ~(size_t)15);  // pos, ecode
    return (b + 63) & ~(size_t)63;
}
MIBN_HD inline void emit_scratch_carve(EmitScratch &S, char *base, int n_vars) {
    const size_t cap = (size_t)emit_pool_cap(n_vars), sw = (cap + 63) / 64, n = (size_t)n_vars;
    S.pool = reinterpret_cast<PF *>(base); base += cap * sizeof(PF);
    S.pool_cap = (int32_t)cap;
    S.sw = (int32_t)sw;
    S.key = reinterpret_cast<double *>(base); base += n * 8;
    S.mem = reinterpret_cast<uint64_t *>(base); base += n * sw * 8;
    S.alive = reinterpret_cast<uint64_t *>(base); base += sw * 8;
    S.ins = reinterpret_cast<const PF **>(base); base += (n + 8) * sizeof(void *);
    S.pos = reinterpret_cast<int32_t *>(base); base += (n * 4 + 15) & ~(size_t)15;
    S.ecode = reinterpret_cast<int32_t *>(base);
}

MIBN_HD inline double emit_scope_log2(const EmitNet &net, const Bits &b) {
    if (net.uniform_log2 >= 0) return (double)(net.uniform_log2 * b.count());  // (scopes hold multi-state variables only: emit_begin)
    double s = 0;
    b.for_each([&](int v) { s += net.log2card[v]; });
    return s;
}

MIBN_HD inline void emit_or_ancestors(const EmitNet &net, Bits &b, int v) {
#if defined(__HIP_DEVICE_COMPILE__)
    b.w[0] |= net.anc[(size_t)v * net.nw];
    if (net.nw > 1) b.w[1] |= net.anc[(size_t)v * net.nw + 1];
#else
    for (int k = 0; k < net.nw; ++k) b.w[k] |= net.anc[(size_t)v * net.nw + k];
#endif
}

// Relevant / hidden sets and the factors of a request = the evidence-sliced CPTs of the relevant nodes (bayes_net.py:768-776):
// the evidence axis is not copied away but folded into the base offset.  Returns 0 or a kEmitErr*.
MIBN_HD inline int emit_begin(const EmitNet &net, EmitScratch &S, int nq, const int32_t *qvars, int ne, const int32_t *evars,
                              const int32_t *ecodes, bool no_prune) {
    Bits &rel = S.rel, &hidden = S.hidden;
    Bits qb, eb;
    rel = Bits{};
    rel.nw = qb.nw = eb.nw = net.nw;
    for (int i = 0; i < nq; ++i) {
        qb.set(qvars[i]); rel.set(qvars[i]);
        emit_or_ancestors(net, rel, qvars[i]);
    }
    for (int i = 0; i < ne; ++i) {
        eb.set(evars[i]); rel.set(evars[i]);
        emit_or_ancestors(net, rel, evars[i]);
    }
    if (!net.prune || no_prune)  // full_joint_dist / predict_proba multiply *all* CPTs (bayes_net.py:460): with sparse or
        for (int v = 0; v < net.n_vars; ++v) rel.set(v);  // unnormalised CPTs a barren node does not sum to 1
    hidden = rel;
    hidden.andnot(qb);
    hidden.andnot(eb);
    for (int i = 0; i < ne; ++i) S.ecode[evars[i]] = ecodes ? ecodes[i] : 0;
    S.n_pool = 0;
    int err = 0;
    rel.for_each([&](int v) {
        if (err) return;
        PF &f = S.pool[S.n_pool++];
        f.n = 0;
        f.alloc = 0;
        f.scope.nw = net.nw;
        for (int k = 0; k < MIBN_BITS_WORDS(net.nw); ++k) f.scope.w[k] = 0;
        uint64_t off = (uint64_t)net.pool_off[v];
        int64_t cells = 1;
        for (int k = net.scope_off[v]; k < net.scope_off[v + 1]; ++k) {
            const int u = net.scope_vars[k];
            if (eb.test(u)) {
                off += (uint64_t)(net.scope_stride[k] * S.ecode[u]);
            } else if (net.card[u] > 1) {
                if (f.n >= kRawAxes) { err = kEmitErrCptAxes; return; }
                f.scope.set(u);
                f.vars[f.n] = u;
                f.strides[f.n] = net.scope_stride[k];
                ++f.n;
                cells *= net.card[u];
            }
        }
        f.off = off | kConstFlag;
        f.src = v;
        f.cells = cells;
    });
    S.n0 = S.n_pool;
    // single-state variables carry no information: they are never axes, never eliminated
    const Bits h0 = hidden;
    h0.for_each([&](int v) { if (net.card[v] <= 1) hidden.clr(v); });
    return err;
}

// The elimination loop over `order` (the hidden variables, first eliminated first) and the final product: appends the
// request's program to `prog`.  Returns 0 or a kEmitErr*.
template <class OrderT>
MIBN_HD inline int emit_run(const EmitNet &net, EmitScratch &S, EmitBuf &prog, EmitStats &st, void *rec, int nq, const int32_t *qvars,
                            int64_t out_off, const OrderT *best, int n_best MIBN_PROF_ARG) {
    PF *pool = S.pool;
    S.rel.for_each([&](int v) { S.key[v] = 0.0; S.pos[v] = -1; });  // (only relevant variables are axes of anything)
    Emitter em{net, prog, st, Arena{}, S.key, S.pos, 0, rec};
    for (int i = 0; i < n_best; ++i) S.key[best[i]] = (double)i;
    for (int i = 0; i < nq; ++i) S.key[qvars[i]] = 1e9 + i;

    const size_t count_pos = prog.size;
    prog.push(0);
    const double steps0 = st.n_steps;
    const PF **ins = S.ins;
    // multiply/eliminate with at most kMaxIn inputs per step: larger products are pre-multiplied (smallest tables first)
    auto emit_limited = [&](int n_in, int x, bool final_, int64_t final_off) -> int {
        while (n_in > kMaxIn && !em.err) {
            if (S.n_pool + 2 > S.pool_cap) { em.err = kEmitErrPool; return -1; }
            for (int i = 1; i < n_in; ++i) {  // (stable insertion sort by cells: the same order on the host and on the device)
                const PF *f = ins[i];
                int k = i - 1;
                while (k >= 0 && ins[k]->cells > f->cells) { ins[k + 1] = ins[k]; --k; }
                ins[k + 1] = f;
            }
            PF &o = pool[S.n_pool++];
            pf_reset(o);
            em.emit(ins, kMaxIn, nullptr, 0, false, 0, o, false);
            for (int k = kMaxIn; k < n_in; ++k) ins[k - kMaxIn] = ins[k];
            n_in -= kMaxIn;
            ins[n_in++] = &o;
        }
        if (em.err) return -1;
        if (S.n_pool + 1 > S.pool_cap) { em.err = kEmitErrPool; return -1; }
        PF &o = pool[S.n_pool++];
        pf_reset(o);
        em.emit(ins, n_in, &x, x >= 0 ? 1 : 0, final_, final_off, o, false);
        return S.n_pool - 1;
    };
    // Which factors mention a variable: every variable keeps the set of factor slots (pool indices) whose scope contains
    // it, `alive` the slots not yet consumed.  Slots are handed out in creation order, so walking a set in ascending order
    // visits the factors in the order of the reference's factor list (bayes_net.py:780-784 pops from it, 786 appends).
    const int sw = S.sw;
    uint64_t *mem = S.mem, *alive = S.alive;
    for (int k = 0; k < sw; ++k) alive[k] = 0;
    S.rel.for_each([&](int v) { for (int k = 0; k < sw; ++k) mem[(size_t)v * sw + k] = 0; });
    auto add_factor = [&](int idx) {
        alive[idx >> 6] |= 1ull << (idx & 63);
        pool[idx].scope.for_each([&](int v) { mem[(size_t)v * sw + (idx >> 6)] |= 1ull << (idx & 63); });
    };
    for (int idx = 0; idx < S.n0; ++idx) add_factor(idx);
    MIBN_TICK(1)  // key / pos / slot sets
    // factors alive whose scope contains a (or b, if b >= 0), in slot order; f(idx) returns false to stop early
    auto each_with = [&](int a, int b, auto f) {
        const int kw = (S.n_pool + 63) >> 6;  // (slots handed out so far: the words beyond are all zero)
        for (int k = 0; k < kw; ++k) {
            uint64_t m = (mem[(size_t)a * sw + k] | (b >= 0 ? mem[(size_t)b * sw + k] : 0ull)) & alive[k];
            for (; m; m &= m - 1)
                if (!f(k * 64 + __builtin_ctzll(m))) return;
        }
    };
    const double log2_small = net.log2_small, log2_big = net.log2_big;
    auto consume = [&](const PF *f) {
        const int idx = (int)(f - pool);
        alive[idx >> 6] &= ~(1ull << (idx & 63));
    };
    for (int i = 0; i < n_best; ++i) {
        if (prog.overflow) return kEmitErrWords;
        const int32_t x = best[i];
        // pop every factor mentioning x (bayes_net.py:780-784)
        int n_in = 0;
        each_with(x, -1, [&](int idx) { ins[n_in++] = &pool[idx]; return true; });
        for (int j = 0; j < n_in; ++j) consume(ins[j]);
        MIBN_TICK(2)  // factors of x
        // SWEEP: up to five consecutive 4-state variables of one big table in a single pass, the tile resident in LDS
        // (planner.h).  sweep_max = how many of the next variables could join: all on the one big input, every other factor
        // that mentions them small.  Four and five variables are tried first, three only after the CHAIN form below.
        int sweep_max = 0, sweep_n[5] = {0, 0, 0, 0, 0};
        const PF *sweep_ins[kSweepMaxSmall + 2];
        if (net.fuse && net.sweep >= 3 && i + emit_min(2, net.sweep_min - 1) < n_best && net.card[x] == 4 && sw <= 16 && n_in - 1 <= kSweepMaxSmall &&
            S.n_pool + 1 <= S.pool_cap) {
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
                for (int j = 1; j < emit_min(net.sweep, 5) && i + j < n_best; ++j) {
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
            for (int k = emit_min(k_hi, sweep_max); k >= k_lo; --k) {
                int X[5];
                for (int j = 0; j < k; ++j) X[j] = best[i + j];
                PF &o = pool[S.n_pool++];
                pf_reset(o);
                if (em.emit_sweep(sweep_ins, sweep_n[k - 1], X, k, o)) {
                    for (int j = n_in; j < sweep_n[k - 1]; ++j) consume(sweep_ins[j]);
                    add_factor(S.n_pool - 1);
                    i += k - 1;
                    return true;
                }
                --S.n_pool;
            }
            return false;
        };
        MIBN_TICK(3)  // sweep candidates
        if (sweep_max >= 4 && try_sweep(5, 4)) { MIBN_TICK(4) continue; }
        MIBN_TICK(4)  // SWEEP 5 / 4
        // Joint elimination: if the factor this step creates is a big table that the very next step consumes, both
        // variables are summed out in one pass over the inputs and the intermediate never touches HBM.
        // CHAIN: three consecutive 4-state variables of one big table in a single pass (planner.h)
        if (net.fuse && net.chain && i + 2 < n_best && n_in < kMaxIn && S.n_pool + 1 <= S.pool_cap &&
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
                    PF &o = pool[S.n_pool++];
                    pf_reset(o);
                    if (em.emit(ins, n3, X, 3, false, 0, o, true)) {
                        for (int j = n_in; j < n3; ++j) consume(ins[j]);
                        add_factor(S.n_pool - 1);
                        i += 2;
                        MIBN_TICK(5)
                        continue;
                    }
                    --S.n_pool;
                    if (em.err) return em.err;
                }
            }
        }
        MIBN_TICK(5)  // CHAIN
        if (sweep_max >= 3 && try_sweep(3, 3)) { MIBN_TICK(6) continue; }
        if (net.sweep_min <= 2 && sweep_max >= 2 && try_sweep(2, 2)) { MIBN_TICK(6) continue; }  // (a pair of one big table: before the FIBER pair form)
        MIBN_TICK(6)  // SWEEP 3 / 2
        if (net.fuse && i + 1 < n_best && n_in < kMaxIn && S.n_pool + 1 <= S.pool_cap) {
            const int32_t x2 = best[i + 1];
            bool link = false;
            Bits u;
            u.nw = net.nw;
            for (int j = 0; j < n_in; ++j) { link = link || ins[j]->scope.test(x2); u.or_(ins[j]->scope); }
            if (link && net.card[x] * net.card[x2] <= kMaxCx &&
                emit_scope_log2(net, u) - net.log2card[x] > log2_small) {
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
                    for (int j = 0; j < n2; ++j) nbig += ins[j]->cells > net.small_cells;
                    const double out_log2 = emit_scope_log2(net, u) - net.log2card[x] - net.log2card[x2];
                    if (nbig < 1 || nbig > 2 || out_log2 < log2_big) fits = false;
                }
                if (fits) {
                    const int X[2] = {x, x2};
                    PF &o = pool[S.n_pool++];
                    pf_reset(o);
                    if (em.emit(ins, n2, X, 2, false, 0, o, true)) {
                        for (int j = n_in; j < n2; ++j) consume(ins[j]);
                        add_factor(S.n_pool - 1);
                        ++i;
                        MIBN_TICK(7)
                        continue;
                    }
                    --S.n_pool;
                    if (em.err) return em.err;
                }
            }
        }
        MIBN_TICK(7)  // pair
        const int out = emit_limited(n_in, x, false, 0);  // pointwise_mul + sum_out (785)
        if (em.err) return em.err;
        add_factor(out);
        MIBN_TICK(8)  // single elimination
    }
    // posterior = pointwise_mul(factors) / sum (bayes_net.py:789-790), written in the caller's
    // query order (C-order, last query variable fastest)
    st.out_cells = 1;
    for (int i = 0; i < nq; ++i) st.out_cells *= net.card[qvars[i]];
    int n_in = 0;
    for (int k = 0; k < ((S.n_pool + 63) >> 6); ++k)
        for (uint64_t m = alive[k]; m; m &= m - 1) ins[n_in++] = &pool[k * 64 + __builtin_ctzll(m)];
    emit_limited(n_in, -1, true, out_off);
    if (em.err) return em.err;
    if (prog.overflow) return kEmitErrWords;
    prog.data[count_pos] = (uint32_t)(st.n_steps - steps0);
    st.arena_cells = emit_max(st.arena_cells, em.arena.top);
    MIBN_TICK(9)  // final product
    return 0;
}

// ------------------------------------------------------------------------------------ work items of a program
// One work item of a request (see Schedule in planner.h): a maximal run of small steps is one SEGMENT, every big step is a
// level of its own, tiled.
struct Tag {
    uint32_t rel_off;  // word offset of the (first) step inside the request's program
    uint32_t a;        // SEGMENT: number of steps | kItemSegment.  TILED step: hi iterations per tile.  SWEEP step: its tiles
    uint32_t wgs;      // workgroups: 1, or the number of tiles
    uint16_t level, kid;
    float bytes;       // algorithmic bytes
};

MIBN_HD inline int fiber_cx_class(const uint32_t *w) {  // 0: cx = 4   1: cx = 16 = 4 x 4   2: anything else (runtime loop)
    const int cx = (int)(w[1] & 0xffff), c1 = (int)(w[8] >> 16);
    if (cx == 4 && c1 == 4) return 0;
    if (cx == 16 && c1 == 4) return 1;
    return 2;
}

// 0: NC = 1   1: NC = 4 contiguous   2: NC = 16 contiguous   3: anything else   4: NC = 16, cx 4 or 16, row stride != 0 -> fp64 MFMA
// 16x16x4   5: OUTER form (always MFMA)
MIBN_HD inline int fiber_nc_class(const uint32_t *w) {
    const int NC = (int)(w[7] >> 16);
    const bool contig = ((w[1] >> 16) & kFlagContig) != 0;
    if ((w[1] >> 16) & kFlagOuter) return 5;
    if (NC == 1) return 0;
    if (NC == 4 && contig) return 1;
    if (NC == 16 && ((w[1] >> kRowStrideShift) & 0xff) && fiber_cx_class(w) < 2) return 4;
    if (NC == 16 && contig) return 2;
    return 3;
}

MIBN_HD inline int kernel_id_of_step(const uint32_t *w) {  // which class of work executes this step
    const uint32_t kind = w[0] & 0xff;
    if (kind == kKindSweep) return kKidSweep;
    if (kind == kKindFiber) {
        const int nb = w[7] & 0xf;
        if ((w[1] >> 16) & kFlagChain) return kKidChain;
        return kKidFiber0 + (nb - 1) * 18 + fiber_cx_class(w) * 6 + fiber_nc_class(w);
    }
    const int n_in = (w[0] >> 8) & 0xff;
    return kKidGeneric0 + emit_min(emit_max(n_in, 1), kMaxIn) - 1;
}

// section-8(d) algorithmic bytes of one step (the planner stores (input + output cells) / 4 in w9)
MIBN_HD inline int64_t step_cost_bytes(const uint32_t *w) { return 32 * (int64_t)w[9]; }

MIBN_HD inline bool step_is_tiled(const EmitNet &net, const uint32_t *w) {
    if ((w[0] & 0xff) == kKindFiber || (w[0] & 0xff) == kKindSweep) return true;  // FIBER / SWEEP steps are only emitted above big_iters
    const bool fin = (w[1] >> 16) & kFlagFinal;
    return !fin && (int64_t)w[2] * (int64_t)w[3] >= net.big_iters;
}

MIBN_HD inline int step_tile_h(const EmitNet &net, const uint32_t *w) {
    if ((w[0] & 0xff) == kKindSweep) return emit_max(1, emit_min(net.sweep_iters, kTileMax));  // (tiles of 64 KiB in, <= 64 KiB out)
    if (net.tile_h > 0) return emit_min(net.tile_h, kTileMax);
    // bytes one hi iteration moves = the step's section-8(d) traffic / hi (broadcast re-reads of a small "big" input
    // are cache hits, they do not count)
    const int64_t per_iter = emit_max<int64_t>(1, step_cost_bytes(w) / emit_max<int64_t>(1, (int64_t)w[3]));
    // (CHAIN steps - 256 KiB per iteration - end up with one iteration per tile; 4 per tile measured 11 % slower)
    return (int)emit_max<int64_t>(1, emit_min<int64_t>(kTileMax, net.tile_bytes / per_iter));
}

// Cut one request's program into work items: put(tag) receives them in level order.  Returns their number.
template <class Put>
MIBN_HD inline uint32_t tag_program(const EmitNet &net, const uint32_t *prog, Put put) {
    const int n_steps = (int)prog[0];
    uint32_t off = 1, n = 0;
    uint16_t level = 0;
    uint32_t seg_first = 0, seg_steps = 0;
    double seg_bytes = 0;
    auto flush = [&]() {
        if (seg_steps) {
            put(Tag{seg_first, seg_steps | kItemSegment, 1u, level, (uint16_t)kKidSeg, (float)seg_bytes});
            ++n;
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
            put(Tag{off, (w[0] & 0xff) == kKindSweep ? w[3] : th, (w[3] + th - 1) / th, level, (uint16_t)kernel_id_of_step(w), (float)bytes});
            ++n;
            ++level;
        } else {
            if (!seg_steps) seg_first = off;
            ++seg_steps;
            seg_bytes += bytes;
        }
        off += w[6];
    }
    flush();
    return n;
}

}  // namespace mibn
