// Host-side planner for the variable-elimination hot path (pure C++17, no HIP dependency).
//
// Replaces the bookkeeping half of BayesNet._variable_elimination (sorobn/bayes_net.py:763-789):
// relevance pruning to ancestors (763-765, `ancestors` 373-378), the hidden set (766), evidence
// slicing of the CPTs (768-776) and the elimination loop's factor selection (779-786) - but instead
// of executing pandas joins it emits a *step program* for the level kernel (ve_kernel.hip.h):
// every step is one fused  psi[out] = sum_x prod_j phi_j[idx_j(out, x)]  over dense strided tables.
//
// Where the reference's elimination order is the iteration order of a Python set (766, 779), the
// planner evaluates several candidate orders with the SURVEY section 8(d) byte model and executes
// the cheapest.
#pragma once
#include <array>
#include <cstdint>
#include <functional>
#include <memory>
#include <string>
#include <vector>

#include "emit_core.h"
#include "wave_plan.h"

namespace mibn {

struct Network {
    int n_vars = 0;
    std::vector<int32_t> card;
    std::vector<double> log2card;
    std::vector<std::vector<int32_t>> scope;     // [*parents, v]
    std::vector<std::vector<int64_t>> cstride;   // dense C-order stride of each scope entry
    std::vector<int64_t> pool_off;               // offset (doubles) of table v in the constants pool
    std::vector<int64_t> cells;
    std::vector<double> pool;
    std::vector<Bits> anc;                       // ancestors (bayes_net.py:373-378), memoised
    std::vector<int32_t> depth;                  // longest path from a root
    std::vector<std::vector<int32_t>> hints;     // optional priority arrays (lower = earlier); install with set_hints()
    std::vector<std::vector<int32_t>> hint_sorted;  // every hint as a variable list in ascending (priority, id) order, then the built-in sweeps
    std::vector<int32_t> topo_asc, topo_desc;    // all variables by (depth ascending, id) / (depth descending, id)
    // networks of up to 128 variables: what the shared host / device order search reads (order_search.h)
    std::vector<B2> anc2, scope2, fam2;          // fam2[v]: the CPTs that mention v (v and its children)
    B2 multi2;                                   // the variables with more than one state
    int32_t uniform_log2 = -1;                   // l: every multi-state variable has 2^l states (OrderNet::uniform_log2)
    std::vector<int32_t> hint_flat;
    OrderNet order_view() const;
    // what the shared host / device emission reads (emit_core.h): the CPT scopes as one CSR, the ancestor sets as n x nw words
    std::vector<int32_t> scope_off32, scope_flat;
    std::vector<int64_t> cstride_flat;
    std::vector<uint64_t> anc_flat;
    EmitNet emit_view() const;
    bool wave_view(WNet &w) const;               // the wave planner's packed copy (wave_plan.h); false: the network or an option is outside what it covers
    int nw = 1;
    int small_cells = 1024;  // FIBER: inputs up to this size are folded into the LDS table
    int64_t big_iters = 4096;  // a step with at least this many lane-iterations is a level of its own (tiled, FIBER form if it fits)
    int tile_h = 0;          // hi iterations per tile; 0 = sized for tile_bytes of traffic per tile
    int64_t tile_bytes = 512 << 10;  // traffic one tile should move (tile_h = 0)
    double minfill_above = 2e7;  // run the greedy min-fill order search only if the sweep orders cost more bytes than this
    int prune = 1;           // restrict a request to the ancestors of its query / evidence variables (bayes_net.py:763-765)
    int outer = 1;           // OUTER form (fp64 MFMA) for products of two big tables
    int fuse = 1;            // eliminate two consecutive variables in one FIBER step when the first result would be a big table
    int plan_cache = 1;      // plan templates: requests that repeat a (query, evidence set) shape re-use its program (plan_batch)
    uint64_t version = 0;    // changes with every set(): invalidates cached plan templates
    mutable std::shared_ptr<void> templates;  // the template store of this network (planner.cpp)
    int chain = 1;           // CHAIN form: a third 4-state variable eliminated in the registers of the same pass
    int stagger = 1;         // build_schedule: groups of requests whose levels are staggered inside a chunk (1 = all in phase)
    int sweep = 5;           // SWEEP form: up to this many 4-state variables of one big table per pass, tile resident in LDS (0 = off)
    int sweep_min = 2;       // fewest variables of a SWEEP pass: 2 also takes the PAIR steps of one big table (two stages: a pass bound by
                             // HBM rather than by the LDS) away from the level kernel's MFMA pair class
    int sweep_iters = kSweepItersDefault;  // tiles per workgroup of the sweep kernel
    int sweep_adapt = 4096;  // build_schedule: fewer tiles per workgroup (down to 2) in sweep launches of fewer workgroups than this (0: off)
    int builtin_sweeps = 1;  // two depth-first topological orders (grid: row- and column-major) as candidate orders next to the host's
                             // hints: -2 % bytes on C3.  Off in round 2 (no measurable time then); with round 3's kernels the bytes
                             // show: 269 -> 278 k queries/s, 4 ms more planning per 32 768 requests (profiles/r03_j_orders.log)
    int order_effort = 0;    // 1: more candidate orders, and the two the byte model ranks first both emitted where the first costs more than second_above (order_search.h)
    double second_above = 1e7;  // modelled bytes of the best order above which the runner-up is emitted too (order_effort >= 1)
    int order_weights = 1;   // compare candidate orders with single-table eliminations at a quarter of their bytes (order_search.h)
    int sweep_canon = 1;     // 0 (test hook): never flag a SWEEP step canonical - the kernel's general path runs everything

    // returns "" or an error message
    std::string set(int32_t n, const int32_t *card_, const int64_t *scope_off, const int32_t *scope_vars,
                    const int64_t *value_off, const double *values);
    void set_hints(int32_t n_hints, const int32_t *priorities);  // n_hints arrays of n_vars priorities
};

struct Request {
    int32_t nq = 0, ne = 0;
    const int32_t *qvars = nullptr;
    const int32_t *evars = nullptr;
    const int32_t *ecodes = nullptr;  // may be null for plan-only statistics
    int64_t out_off = 0;              // offset (doubles) into the batch result buffer
    bool no_prune = false;            // MIBN_Q_NOPRUNE: every CPT takes part (full_joint_dist / predict_proba, bayes_net.py:460)
    const uint8_t *order = nullptr;   // elimination order found elsewhere (the device order search), n_order entries
    int32_t n_order = -1;             // -1: search on the host
};

struct PlanStats {
    double alg_bytes = 0, alg_flops = 0, n_steps = 0, max_step_cells = 0;
    int64_t arena_cells = 0;  // scratch cells this request needs in its arena slot
    int64_t out_cells = 0;
    std::vector<int32_t> *order = nullptr;  // optional: receives the elimination order the plan executes (mibn_plan_order)
};

// Step program encoding (uint32 words), consumed by ve_kernel.hip.h and oracle/plan_sim.cpp:
//   program := n_steps, step*
//   every step starts with a 10-word header
//      w0 = kind | n_in<<8 | n_axes<<16 | nlo<<24      kind: 0 = GENERIC, 1 = FIBER
//      w1 = cx | flags<<16      cx = number of eliminated combinations (1 = product only)
//      w2 = lo_cells   w3 = hi_cells     lane-varying / wave-uniform blocks of the iteration space
//      w4,w5 = out_off (u64, doubles; arena-relative, or result-buffer-relative if FINAL)
//      w6 = step_words (total words of this step)
//      w7 = FIBER: n_big | n_small<<4 | n_N<<8 | n_ctrl<<12 | NC<<16
//      w8 = FIBER: T_cells | c1<<16    (c1 = cardinality of the first eliminated variable; cx = c1 * c2)
//      w9 = (input cells + output cells) / 4  - the step's section-8(d) traffic, for per-kernel rooflines
//
//   GENERIC  psi[o] = sum_x prod_j phi_j[off_j(o) + x*xs_j], iteration space = output cells, one eliminated variable:
//      per input j<n_in:  in_off lo, in_off hi (bit 63 = constants pool), xs_j
//      card[a]            a < n_axes          (output axes, fastest first, merged)
//      stride[j][a]       j < n_in, a < n_axes (int32, doubles)
//
//   FIBER    the inputs are split into big tables (<= 2, streamed from HBM) and small ones (CPT
//      slices, <= Network::small_cells cells each).  One or two variables are eliminated in the same pass
//      (x enumerates their joint values, x = x1 + c1*x2).  Output axes no big input depends on are the N axes
//      (NC = prod of their cards <= kMaxNC): one lane owns one cell r of the remaining R axes, loads
//      the cx values of each big input once and produces the whole N-fiber in registers:
//           out[r, n] = sum_x (prod_b F_b[r, x]) * T[n, x, ctrl(r)]
//      where T = product of the small inputs, tabulated once per tile in LDS over
//      (n fastest, x, ctrl axes = the R axes a small input depends on).  Iteration space = R cells.
//      per big input b<n_big:     off lo, off hi, xs1_b, xs2_b        (offset of x = x1*xs1 + x2*xs2)
//      per small input s<n_small: off lo, off hi, xs1_s, xs2_s, tstride[s][k] k < n_N + n_ctrl
//      tcard[k]                   k < n_N + n_ctrl     (N axes, then ctrl axes)
//      nout[n]                    n < NC               (output offset of N-combination n)
//      card[a], ostride[a], tstride[a]    a < n_axes   (R axes, merged; tstride in T cells)
//      bstride[b][a]              b < n_big, a < n_axes
//      flag CONTIG: nout[n] == n and the lane-varying block is contiguous in the output (cell l of the block at
//      l*NC), i.e. the lanes of a wave own one contiguous 64*NC-cell region (vector / transposed stores).
//      flag OUTER (2 big inputs, no small ones): the second big input B takes the place of T.  The N axes are two
//      4-state output axes only B depends on (NC = 16), the R axes are all the others, and
//           out[r, n] = sum_x A[r, x] * B[r, n, x]        (B's offset: bstride[1][.] along R, nB[n] along N)
//      - per 16-cell row block a dense [16 x cx] x [cx x 16] product.  After nout[] the step carries nB[n], n < 16.
//      w1 bits 20..27 = row stride s of the fp64-MFMA form (0 = not applicable): the 64 cells of a wave split into
//      4 row blocks of 16 cells that share one T[., ., ctrl] slice - block rb = cells (i % s) + s*rb + 4*s*(i / s),
//      i < 16 - so that per block the step is a dense [16 x cx] x [cx x 16] product.  s = 16 when no ctrl axis lies
//      inside a wave's 64 cells, s = 1 / 4 / 16 = the cell stride of the single 4-state ctrl axis that does.
//      flag CHAIN (one big input F, three 4-state variables x1, x2, x3 eliminated in one pass):
//           out[r, n12, n3] = sum_x3 T3[x3, n3, n12?, ctrl3(r)] * sum_x12 F[r, x12, x3] * T12[n12, x12, ctrl12(r), x3?]
//      - the pair (x1, x2) goes through the fp64-MFMA row-block product of the plain form, once per value of x3,
//      and x3 is summed out of the four accumulators in registers.  T12 = product of the small inputs that depend on
//      x1 or x2 (x3, if they depend on it, is its last ctrl dimension), T3 = product of those that depend on x3 only,
//      tabulated after T12 in LDS over (x3, n3, [n1, n2,] ctrl3 axes).  The three new variables are the three fastest
//      output axes (n12 at strides 1 and 4, n3 at 16).  Encoded with n_big = 2: big record 1 is (T12 cells = offset
//      of T3, T3 cells, stride of x3 in F, stride of x3 in T12) and bstride[1][.] are the T3 strides of the R axes.
//      After nout[]: n_small3, n_dims3 | (T3 depends on n12) << 8, tcard3[n_dims3], then per small-3 input: off lo,
//      off hi, stride[n_dims3].
//
//   SWEEP (kind 2)  k = 3..5 four-state variables x_1..x_k of ONE big table F (+ CPT slices) eliminated in one pass with
//      the tile resident in LDS (ve_sweep_kernel, sweep_kernel.hip.h).  The x_j are F's slowest axes, x_j on digit dig_j of
//      xc:  F index = r + Rcells * xc,  xc = sum_d digit_d * 4^d.  A tile = all 4^k xc x Rt consecutive r (Rt = 8192 / 4^k
//      cells: runs of Rt * 8 bytes in F), staged as L[r_local + Rt * xc].  Stage j, in elimination order, works in place
//      along digit dig_j:
//           L[.., n, ..] = sum_x L[.., x, ..] * T_j[n + cout_j * (x + 4 * (c0 + 4 * (c1 + 4 * c2)))]        n < cout_j
//      cout_j = 4: the one new variable of the CPT slices that mention x_j takes over the digit; 1: the digit dies.  The
//      ctrl values c0..c2 are four-state variables that are on a live digit right now (x_j' not yet eliminated, n_j'
//      already introduced) or R axes of F with a power-of-two stride (bits of r; at most two of the three).  T_j = product of the step's small
//      inputs that mention x_j and none of x_1..x_{j-1}.  Output: the surviving digits (ascending) are the fastest axes,
//      then F's R axes in F's order:  out index = sum_q val_q * 4^q + 4^kout * r.
//      header: w0 = kind | n_in << 8 | k << 16 | log2(Rt) << 24;  w1 = 4^k | flags << 16;  w2 = 8192;  w3 = tiles = Rcells / Rt
//              w7 = kout | T cells (all stages) << 16;  w8 = LDS digit of surviving rank q at bits 4q (q < kout)
//      body:   F off lo, off hi
//              per stage j < k (5 words): s0 = dig | cout << 4 | n_small << 8 | nctrl << 12 | loop digit << 16 |
//                                              thread fields f0 << 20 | f1 << 24 | f2 << 28   (digits, unused = 7)
//                                         s1 = T_j offset (cells) | T_j cells << 16
//                                         ctrl c < 3:  src | tstride << 8     src 0..4 = digit, 8 + s = bits (r >> s) & 3
//              per stage, per small input (7 words): off lo, off hi, stride of the new variable, of x_j, of ctrl 0..2
//      flag SWEEP_CANON (w1): see kFlagSweepCanon.  The loop digit of a stage is sweep_loop_digit(k, dig), the thread fields are
//      the other free digits in ascending order.
//      A work item = Network::sweep_iters consecutive tiles; one workgroup of kSweepWG lanes per item.
// Growable word buffer the planner appends programs to.  The engine backs it with pinned host memory
// (so the upload is a true async DMA) and keeps it across calls; the default backing is malloc.
struct ProgBuf {
    uint32_t *data = nullptr;
    size_t size = 0, cap = 0;
    // returns a buffer of >= new_cap words whose first `used` words are copied from `old` (and releases old)
    uint32_t *(*grow)(void *ctx, uint32_t *old, size_t used, size_t new_cap) = nullptr;
    void *ctx = nullptr;
    uint32_t *extend(size_t words);  // appends `words` uninitialised words, returns their address
    void push(uint32_t w) { *extend(1) = w; }
    void release();                  // frees a malloc-backed buffer (no-op for custom backings)
};

// Plan one request; appends the program to `prog`.  Returns "" or an error message.
std::string plan_request(const Network &net, const Request &rq, ProgBuf &prog, PlanStats &st);
std::string emit_error_message(int err);  // message of a kEmitErr* code (emit_core.h)
std::string plan_request(const Network &net, const Request &rq, std::vector<uint32_t> &prog, PlanStats &st);

// Persistent host worker threads (planning a 16 k-request batch spawns no threads and faults no pages).
class ThreadPool {
public:
    explicit ThreadPool(int n);
    ~ThreadPool();
    int size() const { return n_; }
    void run(const std::function<void(int)> &job);  // job(t) on every worker t in [0, n); returns when all are done
private:
    struct Impl;
    Impl *impl_;
    int n_;
};

// Plan requests [b0, b1) of a CSR batch: worker t plans a contiguous share into bufs[t].
// (Tag - one work item of a request, tagged while the request's program is still in the planning worker's cache - lives in emit_core.h)
struct BatchPlan {
    std::vector<std::vector<Tag>> tags;        // per worker: the items of its requests, request by request
    std::vector<uint32_t> tag_first, tag_count;  // per request: its range in tags[thread_of[i]]
    std::vector<size_t> thread_words;          // words written by each worker (bufs[t].size)
    std::vector<uint64_t> prog_off;            // per request: word offset into the concatenated buffers
    std::vector<double> cost;                  // per request: algorithmic bytes
    std::vector<int64_t> arena_need;           // per request: scratch cells
    std::vector<int32_t> thread_of;            // per request: worker that planned it
    std::vector<uint64_t> local_off;           // per request: word offset inside that worker's buffer
    int64_t arena_cells = 0;                   // largest per-request scratch need
    size_t total_words = 0;
    PlanStats st;                              // totals
    std::string err;
};
void plan_batch(const Network &net, ThreadPool &pool, std::vector<ProgBuf> &bufs, int64_t b0, int64_t b1,
                const int64_t *q_off, const int32_t *q_vars, const int64_t *e_off, const int32_t *e_vars,
                const int32_t *e_codes, const int64_t *out_off, const char *skip, BatchPlan &bp, bool no_prune = false,
                const uint8_t *orders = nullptr, const int32_t *order_len = nullptr,  // orders[(b - b0) * 128 ..]: device order search
                int64_t out_first = -1);  // result offsets relative to out_off[out_first] (default: b0) - the host's share of a chunk whose
                                          // first requests the device plans

// Shard-balancing estimate (mibn_estimate_costs): section-8(d) bytes of the cheaper of the two sweep orders of every
// request of a CSR batch - the byte model only, nothing is emitted.
void estimate_costs(const Network &net, ThreadPool &pool, int64_t B, const int64_t *q_off, const int32_t *q_vars,
                    const int64_t *e_off, const int32_t *e_vars, double *cost);

// ---------------------------------------------------------------------------------------------------
// Level-synchronous schedule.  A request's program is cut into *items*: a maximal run of small steps
// is one SEGMENT item (one workgroup runs the steps back to back), every big step is its own level
// and is split into TILE items over its wave-uniform iteration range, executed by a kernel specialised
// for the step's shape.  Item k of a request runs at level k; level L of all requests of a wave is one
// launch per kernel id, launches are stream-ordered, so the only synchronisation is the launch
// boundary.  Big steps thereby use the whole chip (no single-workgroup tails) and every
// specialisation gets its own register budget.
struct Item {
    uint32_t req;      // request index within the wave
    uint32_t rel_off;  // word offset of the (first) step inside the request's program
    uint32_t a, b;     // SEGMENT: a = number of steps | kItemSegment; b = segments of its workgroup (first item of a group of
                       // kSegPerWg) or 0.  TILED step: a = hi iterations per tile; b = index of the item's first workgroup
                       // within its level
};
constexpr int kSegPerWg = 4;  // segments per workgroup of the level kernel: one per wave (the items of a level's segments are
                              // contiguous; workgroup g runs items first + 4 g .. + 3, Item::b of the group's first item = how
                              // many of the four exist)
struct Launch {
    int level, kid;       // kid: the class of work (kernel_name), for per-class accounting
    size_t first, count;  // range in Schedule::items
    size_t wg_first;      // range in Schedule::wg_item: workgroups [wg_first, wg_first + grid)
    size_t grid;
    size_t wg_level;      // wg_item index of the level's first workgroup (Item::b is relative to it)
    double alg_bytes;     // algorithmic bytes of the steps in this launch
};
const char *kernel_name(int kid);
// (kernel_id_of_step, fiber_cx_class / fiber_nc_class, step_cost_bytes, step_is_tiled, step_tile_h: emit_core.h)

struct Schedule {
    std::vector<Item> items;
    std::vector<uint32_t> wg_item;    // item index of every workgroup, level by level
    std::vector<Launch> launches;     // ordered by (level, kid); the launches of a level are contiguous in wg_item
    std::vector<uint64_t> arena_off;  // per request of the wave: offset (doubles) of its private arena
    int64_t arena_cells = 0;          // total
    int n_levels = 0;
};
// requests [r0, r1) of the planned chunk (indices into BatchPlan::prog_off); `arena_need[i]` = scratch cells of request i
void build_schedule(const Network &net, const BatchPlan &bp, const std::vector<ProgBuf> &bufs, int64_t r0, int64_t r1,
                    Schedule &out);

// Validate a request (unknown ids, duplicates, overlap) - bayes_net.py:840-845 and the KeyError of 770.
std::string validate_request(const Network &net, const Request &rq);  // "" or the reference's error message
bool request_is_valid(const Network &net, const Request &rq);          // the same checks without building a message

}  // namespace mibn
