// mibn engine: C-ABI (include/mibn.h) over the planner and the gfx950 kernels.
// No CPU fallback: every query entry point needs a live HIP device.
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <rccl/rccl.h>  // types and prototypes only: librccl.so is dlopen'ed by mibn_comm_init, never linked

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <mutex>
#include <numeric>
#include <string>
#include <thread>
#include <vector>

#include "../../include/mibn.h"
#include "gibbs_kernel.hip.h"
#include "sample_kernel.hip.h"
#include "count_kernel.hip.h"
#include "planner.h"
#include "tiny_kernel.hip.h"
#include "sweep_kernel.hip.h"
#include "ve_kernel.hip.h"
#include "wave_plan_kernel.hip.h"

using namespace mibn;

namespace {
double now_ms() {
    return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
}
}  // namespace

// ---------------------------------------------------------------------------------------------- device order search
// The elimination-order search of the planner (order_search.h - the same code the host runs) as a kernel: one request
// per lane, the search state in a global buffer.  Pure bit-set / integer work, independent per request and latency-bound:
// it occupies a few hundred wave slots for a few milliseconds beside the level kernels and takes ~40 % of the planning
// time off the host cores (option "gpu_search").
struct OrderArgs {
    OrderNet net;
    const int64_t *q_off, *e_off;
    const int32_t *q_vars, *e_vars;
    int64_t B;
    uint32_t flags;
    uint8_t *orders;       // [B][128]
    int32_t *order_len;    // [B]
    OrderScratch *scratch; // [B]
    uint32_t *zero;        // a counter the launch resets (emit_kernel's item cursor: no memset on the planning stream), or null
    const int32_t *perm;   // [B] or null: lane position -> request of the slice (requests of similar size share a wave: plan_on_device_launch)
    int32_t lanes;         // requests per wave (the other lanes of the 64 leave at once: the lanes of a wave run different requests -
                           // every data-dependent branch diverges - so a wave's time grows with the lanes in use; fewer lanes in more
                           // waves finish sooner, as long as the chip has SIMDs to spare: option plan_lanes)
};

// (Workgroups of up to 16 waves - no cooperation between them, only placement: the waves of a workgroup share a CU, so the 512
// waves of a chunk take 32 CUs whole instead of one workgroup's worth of registers on every CU - beside the level and sweep
// kernels, whose workgroups fill a CU's register file exactly, that is an eighth of the chip instead of a third to a half.)
#ifndef MIBN_ORDER_WAVES_PER_EU
#define MIBN_ORDER_WAVES_PER_EU 4  // build-time experiment hooks: the register budget of the planner's kernels (waves per SIMD)
#endif
#ifndef MIBN_EMIT_WAVES_PER_EU
#define MIBN_EMIT_WAVES_PER_EU 4
#endif
#ifndef MIBN_PLAN_WG
#define MIBN_PLAN_WG 1024  // largest workgroup of the planner's kernels (option plan_waves x 64 must not exceed it)
#endif
__global__ __launch_bounds__(MIBN_PLAN_WG, MIBN_ORDER_WAVES_PER_EU) void order_kernel(const OrderArgs A) {
    if (A.zero && blockIdx.x == 0 && threadIdx.x == 0) *A.zero = 0;
    const int lane = (int)(threadIdx.x & 63), wave = (int)(threadIdx.x >> 6), waves = (int)(blockDim.x >> 6);
    if (lane >= A.lanes) return;
    const int64_t pos = ((int64_t)blockIdx.x * waves + wave) * A.lanes + lane;
    if (pos >= A.B) return;
    const int64_t b = A.perm ? A.perm[pos] : pos;
    OrderScratch &S = A.scratch[pos];
    const int64_t q0 = A.q_off[b], e0 = A.e_off[b];
    order_search(A.net, S, (int)(A.q_off[b + 1] - q0), A.q_vars + q0, (int)(A.e_off[b + 1] - e0), A.e_vars + e0,
                 (A.flags & MIBN_Q_NOPRUNE) != 0);
    uint8_t *out = A.orders + b * 128;
    for (int i = 0; i < S.n_best; ++i) out[i] = S.best[i];
    A.order_len[b] = S.n_best;
}

// Device program emission (emit_core.h: the code the host's planning workers run): one request per lane turns its elimination
// order (order_kernel's output, never copied to the host) into its step program, written straight into the request's slot
// of the chunk's device program buffer, and cuts it into work items.  The host only lays out the schedule of the chunk
// (build_schedule, from the work items): a rank needs no planning threads.
struct EmitArgs {
    EmitNet net;
    const int64_t *q_off, *e_off, *out_off;  // [B + 1]; out_off relative to the chunk's first request
    const int32_t *q_vars, *e_vars, *e_codes;
    const char *skip;                        // [B] evidence outside the domain: zero steps
    const uint8_t *orders;                   // [B][128]
    const int32_t *order_len;                // [B]
    int64_t B;
    uint32_t flags;
    uint32_t *prog;                          // [B][prog_stride]
    uint32_t prog_stride;
    EmitMeta *meta;                          // [B]
    Tag *tags;                               // [tag_cap]: the work items of all requests, a request's contiguous
    uint32_t *tag_cursor;
    uint32_t tag_cap;
    char *scratch;                           // [B][scratch_stride] planning state (EmitScratch)
    size_t scratch_stride;
    const int32_t *perm;                     // [B] or null (see OrderArgs)
    int32_t lanes;                           // requests per wave (see OrderArgs)
};

#if defined(MIBN_EMIT_PROF)
__device__ unsigned long long g_emit_prof[12];  // 100 MHz ticks per phase, summed over the lanes (see MIBN_TICK in emit_core.h)
#endif

__global__ __launch_bounds__(MIBN_PLAN_WG, MIBN_EMIT_WAVES_PER_EU) void emit_kernel(const EmitArgs A) {
    const int lane = (int)(threadIdx.x & 63), wave = (int)(threadIdx.x >> 6), waves = (int)(blockDim.x >> 6);
    if (lane >= A.lanes) return;
    const int64_t pos = ((int64_t)blockIdx.x * waves + wave) * A.lanes + lane;
    if (pos >= A.B) return;
    const int64_t b = A.perm ? A.perm[pos] : pos;
    EmitMeta m;
    m.words = 1; m.n_tags = 0; m.tag_first = 0; m.err = 0; m.prog_first = 0; m.pad_ = 0;
    m.alg_bytes = m.alg_flops = m.n_steps = m.max_step_cells = 0;
    m.arena_cells = 0;
    uint32_t *slot = A.prog + (size_t)b * A.prog_stride;
    if (A.skip[b]) { slot[0] = 0; A.meta[b] = m; return; }
#if defined(MIBN_EMIT_PROF)
    EmitProf prof_;
    for (int k = 0; k < 12; ++k) prof_.a[k] = 0;
#if defined(__HIP_DEVICE_COMPILE__)
    prof_.t = __builtin_amdgcn_s_memrealtime();
#endif
#endif
    EmitScratch S;
    emit_scratch_carve(S, A.scratch + (size_t)pos * A.scratch_stride, A.net.n_vars);
    const int64_t q0 = A.q_off[b], e0 = A.e_off[b];
    const int nq = (int)(A.q_off[b + 1] - q0), ne = (int)(A.e_off[b + 1] - e0);
    int err = emit_begin(A.net, S, nq, A.q_vars + q0, ne, A.e_vars + e0, A.e_codes + e0, (A.flags & MIBN_Q_NOPRUNE) != 0);
    EmitBuf buf;
    buf.data = slot;
    buf.cap = A.prog_stride;
    EmitStats st;
    MIBN_TICK(0)  // relevant set, evidence-sliced CPTs
    if (!err) err = emit_run(A.net, S, buf, st, nullptr, nq, A.q_vars + q0, A.out_off[b], A.orders + b * 128, (int)A.order_len[b] MIBN_PROF_PASS);
    if (!err) {
        const uint32_t nt = tag_program(A.net, slot, [](const Tag &) {});
        const uint32_t first = atomicAdd(A.tag_cursor, nt);
        if (first + nt <= A.tag_cap) {
            uint32_t k = first;
            tag_program(A.net, slot, [&](const Tag &t) { A.tags[k++] = t; });
            m.n_tags = nt;
            m.tag_first = first;
        } else {
            err = kEmitErrWords;
        }
    }
    MIBN_TICK(10)  // work items
#if defined(MIBN_EMIT_PROF)
    for (int k = 0; k < 12; ++k) atomicAdd(&g_emit_prof[k], prof_.a[k]);
#endif
    m.err = err;
    m.words = (uint32_t)buf.size;
    m.alg_bytes = st.alg_bytes; m.alg_flops = st.alg_flops; m.n_steps = st.n_steps; m.max_step_cells = st.max_step_cells;
    m.arena_cells = st.arena_cells;
    A.meta[b] = m;
}

// Chunk sets (pinned program buffers, device copies, schedule, launch events): with n sets in use the host plans chunk
// k + n - 1 at the earliest when chunk k has completed.  Two by default.  With two sets and a call of three chunks the
// second chunk of a call can only be planned once the previous call has finished - the GPU runs the short first chunk and
// then idles 5-12 ms (the "[mibn gap]" lines of option trace); a third set (option chunk_sets=3) closes those gaps but
// proved erratic end to end (planning wall time doubles, runs between 155 k and 204 k queries/s against a steady
// 194-204 k with two: profiles/r02_q_chunk_sets.log), so it stays an experiment.
constexpr int kChunkSets = 4;

struct mibn_ctx {
    Network net;
    bool has_net = false;
    bool planner_only = false;
    int device = -1;
    int n_cu = 0;
    hipStream_t stream = nullptr;       // kernels, gibbs, result download
    hipStream_t copy_stream = nullptr;  // program / schedule uploads: overlap the previous chunk's kernels
    double *d_pool = nullptr;
    // Lanes: consecutive chunks of a call alternate between two streams with an arena each and run CONCURRENTLY - requests are
    // independent, and the launch boundaries of one chunk's levels (the only synchronisation of the level-synchronous schedule)
    // then overlap with the other chunk's launches instead of leaving the tail of every launch to half-empty CUs.
    double *d_arena[2] = {nullptr, nullptr};  // private arenas of the requests of the wave in flight, per lane
    size_t arena_bytes[2] = {0, 0};
    double *d_results[2] = {nullptr, nullptr};  // dense posteriors of the call in flight (two calls may overlap)
    size_t results_cap[2] = {0, 0};             // doubles
    struct Pending {                            // an asynchronous call whose results have not been collected yet
        bool active = false;
        double *out = nullptr;
        size_t cells = 0;
        hipEvent_t done = nullptr;
    } pend[2];
    int next_slot = 0;
    // chunk pipeline: workers plan the next chunks into pinned buffers while the GPU runs chunk i
    ThreadPool *pool = nullptr;
    struct Staging {  // pinned host staging: pageable sources would make hipMemcpyAsync block on the stream
        char *p = nullptr;
        size_t cap = 0;
    };
    struct Set {
        Staging stage[4];           // prog_off, arena_off, items, wg_item
        std::vector<ProgBuf> bufs;  // pinned host program buffers, one per worker
        std::vector<ProgBuf> xbufs; // ... and the ones the host re-plans single requests of a device-planned chunk into (beyond a device limit)
        BatchPlan xplan;
        uint32_t *d_prog = nullptr;
        size_t prog_cap = 0;
        uint64_t *d_prog_off = nullptr;
        size_t prog_off_cap = 0;
        uint64_t *d_arena_off = nullptr;
        size_t arena_off_cap = 0;
        uint32_t *d_wg_item = nullptr;
        size_t wg_item_cap = 0;
        Item *d_items = nullptr;
        size_t items_cap = 0;
        hipEvent_t uploaded = nullptr;       // the copy stream has delivered this set's programs and schedule
        std::vector<hipEvent_t> ev;          // launch boundaries of the waves in flight
        struct Timed {  // kid >= 0: one launch; -1: the wall time of a wave (kernel_ms); -2: the launches of one level on the two streams of
                        // option overlap, as ONE concurrent launch pair - events (e0, e1) of the level kernel's launch, (e2, e3) of the sweep
                        // kernel's (kNone: the level has no such launch); its duration is the span from the earlier start to the later end
            int kid; size_t e0, e1; double bytes, items; uint64_t call; size_t e2 = kNone, e3 = kNone, e4 = kNone, e5 = kNone, e6 = kNone, e7 = kNone;  // (e4, e5: the segment kernel's; e6, e7: the MFMA kernel's)
            static constexpr size_t kNone = ~size_t(0);
        };
        std::vector<Timed> timed;
        size_t ev_used = 0;
        bool busy = false;
        int gap_from = -1;  // trace: index (in mibn_ctx::gap_ev) of the event that closed the previous wave
        BatchPlan plan;
        Schedule sched;
    } set[kChunkSets];
    Staging res_stage[2];  // pinned landing buffers of asynchronous calls
    // small-network specialisation (tiny_kernel.hip.h): one lane per request, no planning
    bool tiny_ok = false;
    int tiny = 1;              // option: 1 = use it where the network is eligible, 0 = always plan step programs
    int tiny_zero_copy = 1;    // option: calls of at most kTinyZeroCopyRequests requests read / write pinned host memory directly (run_tiny)
    int32_t *d_tiny_meta = nullptr;
    int32_t tiny_meta_words = 0;
    char *d_tiny_req = nullptr;  // request arrays of the call in flight
    size_t tiny_req_cap = 0;
    Staging tiny_stage;
    // adaptive planning (option "adaptive"): when planning, not the GPU, bounds a stream of calls (few host cores per
    // GPU), the elimination-order search - 40 % of the planning time - moves to the device (order_kernel); for networks the
    // device search does not cover (> 128 variables) the greedy min-fill search is reserved for ever more expensive
    // requests instead; both are given back when the host has slack again
    int adaptive = 0;
    bool auto_search = false;  // gpu_search was switched on by the adaptive policy
    bool auto_emit = false;    // gpu_emit was
    int host_bound_streak = 0;
    bool adaptive_seeded = false;
    double base_minfill = 5e6, seen_plan_ms = 0, seen_kernel_ms = 0;  // (minfill_above: 2e7 up to round 6's last day; with order_effort 1 min-fill also supplies the opening candidate - 5e6: 1.7 % less GPU time per step for a tenth more planning, profiles/r06_cd_ab.log)
    double base_second_above = 2e7;  // option second_above (the calls the device plans run without the second emission: run_batch)
    int second_on_device = 0;        // option: 1 = device-planned calls emit the runner-up too (tests: the wave planner's second emission)
    double retired_requests = 0, seen_requests = 0;  // requests whose kernel time has been booked (the unit of kernel_ms in the policy's windows)
    double host_rate = 0;                            // requests per ms the host's workers planned beside the device planner (smoothed; 0: not measured)
    double kernel_ms_per_req = 0;                    // retired kernel time per request over the policy's last windows (smoothed; 0: not measured)
    double fixed_ms_per_req = 0;                     // the host's side of a call besides planning - validation, schedule - per request (smoothed)
    // device order search (order_kernel)
    int plan_lanes = 0;              // requests per wave of order_kernel / emit_kernel (1..64); 0 = by the rank's planning threads: plan_lanes_now()
    int plan_waves = 16;             // waves per workgroup of the two (1..16): see order_kernel
    int plan_sort = 0;               // experiment (off): the requests of a slice dealt to the lanes by descending relevant-set size - 1: consecutive
                                     // ranks share a wave; 2: round-robin over the workgroups (every CU the same mix, homogeneous waves); 3: round-
                                     // robin over the waves (every wave the same sum).  Measured at 24 lanes per wave, two planning threads
                                     // (profiles/r05_h/i/j_*.log): 250 k queries/s unsorted, 231 k (1), 231 k (2), 249 k (3) - homogeneous waves
                                     // are the slow ones: the lanes of a wave diverge almost completely, a wave's time is the SUM of its lanes',
                                     // and the stream's own order already mixes the sizes; balancing the sums exactly (3) shortens the
                                     // planner's kernels by 2 % and moves nothing.  32 lanes, four threads: 259 / 265 / 265 / 258 k.
    int gpu_search = 0;              // option: 1 = search elimination orders on the device (networks of <= 128 variables)
    hipStream_t search_stream = nullptr;
    char *d_order_net = nullptr;     // the OrderNet arrays
    OrderNet order_net_dev;          // pointers into d_order_net
    bool order_net_ok = false;
    uint8_t *d_orders = nullptr;
    size_t orders_cap = 0;
    int32_t *d_order_len = nullptr;
    size_t order_len_cap = 0;
    OrderScratch *d_order_scratch = nullptr;
    size_t order_scratch_cap = 0;
    Staging search_in, search_out;   // pinned: request arrays in, orders + lengths out
    double search_ms = 0;            // host wall time spent waiting for the device search (last call)
    // device program emission (emit_kernel)
    int gpu_emit = 0;                // option: 1 = the device plans whole chunks (order search + emission), 2 = the same, checked
                                     // word for word against the host's planner (tests)
    char *d_emit_net = nullptr;      // the EmitNet arrays
    EmitNet emit_net_dev;            // pointers into d_emit_net
    bool emit_net_ok = false;
    char *d_emit_scratch = nullptr;  // planning state, one slice per lane
    size_t emit_scratch_cap = 0;
    uint32_t *d_emit_cursor = nullptr;
    uint32_t *d_plan_perm = nullptr;  // the order in which wave_plan_kernel's waves draw the requests of a chunk (plan_sort_kernel: the long ones first)
    size_t plan_perm_cap = 0;
    int wave_sort = 1;                // option: 0 = by index
    Staging emit_in, emit_out;       // pinned, read / written by the kernels themselves: request arrays in, per-request results and work items out
    hipEvent_t emit_ev[2] = {nullptr, nullptr};  // around the planner's kernels
    double emit_share = 0.75;        // the device's share of a chunk, the host's workers plan the rest meanwhile: follows the two measured
                                     // rates so that both finish together (option emit_share: 0 < x <= 1 pins it)
    double emit_share_opt = -1;
    BatchPlan emit_dev, emit_host;   // the two parts of a chunk before they are joined
    uint32_t emit_words = 6144;      // words of a request's program slot (doubles after a chunk that did not fit)
    int64_t emit_single = 0;         // requests of device-planned chunks the host planned because they exceeded a device limit
    // wave-cooperative device planner (wave_plan_kernel)
    int wave_plan = 1;               // option: 1 = chunks the device plans go through wave_plan_kernel where the network is covered (wave_plan.h)
    int plan_priority = 2;           // option: the device planner's stream priority (2 highest - the default since round 4 -, 1 normal, 0 lowest); before the first device-planned chunk
    int wave_wgs = 0;                // option: workgroups of a wave_plan_kernel launch (0: one per four requests - the whole chip at once)
    WNet *wnet_host = nullptr;       // the packed network + options as uploaded last
    WNet *d_wnet = nullptr;
    bool wnet_ok = false;
    double emit_ms = 0;              // host wall time spent waiting for the device planner (last call)
    uint64_t emit_chunks = 0, emit_fallbacks = 0;
    hipEvent_t gap_ev[4] = {nullptr, nullptr, nullptr, nullptr};  // trace: ends of the last waves (GPU idle time between waves)
    uint64_t n_waves = 0;
    int n_sets = 2;           // chunk sets in use (option "chunk_sets": 2..4; two per lane: one on the GPU, one being planned)
    int set_cursor = 0;    // the chunk set the next chunk plans into: alternates across calls, so that a call of one chunk
                           // plans into the idle set while the previous call's kernels still run from the other one
    uint64_t call_id = 0;  // kernel time retired later is booked to the call that launched it
    // RCCL (multi-GPU gather / reduce), loaded on demand
    struct Comm {
        void *dl = nullptr;
        ncclComm_t comm = nullptr;
        int rank = 0, world = 1;
        decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
        decltype(&ncclCommInitRank) CommInitRank = nullptr;
        decltype(&ncclCommDestroy) CommDestroy = nullptr;
        decltype(&ncclCommCount) CommCount = nullptr;
        decltype(&ncclCommUserRank) CommUserRank = nullptr;
        decltype(&ncclAllGather) AllGather = nullptr;
        decltype(&ncclReduce) Reduce = nullptr;
        decltype(&ncclAllReduce) AllReduce = nullptr;
        decltype(&ncclGetErrorString) GetErrorString = nullptr;
        void *d_send = nullptr, *d_recv = nullptr;
        size_t send_cap = 0, recv_cap = 0;
        hipStream_t stream = nullptr;  // collectives run on their own stream: the gather of batch s must not queue behind the
                                       // kernels of batch s + 1, which the two-deep pipeline has already submitted
    } comm;
    std::string err;
    mibn_stats stats{}, total{};               // last call / since creation
    mibn_kernel_stat kstats[kNumKernels + 7];  // per class (split_kinds) + the level kernel as a whole + the tiny kernel + the LDS-DMA sweep kernel
    mibn_kernel_stat ktotal[kNumKernels + 7];  // + the device planner's pair of kernels + the concurrent launch pair of a level (option overlap)
    // options
    double arena_gb = 200.0;  // scratch budget of all lanes together (of the 288 GB)
    hipStream_t stream2 = nullptr;  // lane 1 (lane 0 = stream)
    hipStream_t aux[2][3] = {{nullptr, nullptr, nullptr}, {nullptr, nullptr, nullptr}};  // option overlap, per lane: the sweep kernel's, the segment kernel's and the MFMA kernel's launches
    int mfma_kernel = 1;            // round 5: the one-table fp64-MFMA pair classes of a level as a launch of ve_mfma_kernel (128 VGPRs, four
                                    // waves per SIMD) on a fourth stream instead of workgroups of ve_level_kernel (168 VGPRs, three): 296.2 -> 300.4 k queries/s,
                                    // all kernels 4.56 -> 4.63 TB/s, three interleaved repetitions each (profiles/r05_b_ab.log)
    int seg_kernel = 1;             // the segments of a level as a launch of ve_segment_kernel (12 KB of LDS per workgroup) instead of workgroups
                                    // of ve_level_kernel (40 KB, 168 VGPRs)
    hipEvent_t epoch = nullptr;     // reference of the busy-time bookkeeping (re-recorded when the GPU is idle)
    hipEvent_t lane_ev = nullptr, zero_ev = nullptr;  // end of lane 1's work of a call / results buffer zeroed
    double busy_until = 0;          // ms since epoch up to which GPU time has been booked as kernel_ms
    int n_streams = 1;  // lanes.  1 (default): every chunk on the main stream, one after the other.  2: consecutive chunks run concurrently
                        // (+3 % with 8 192-request chunks, profiles/r02_y_lanes*.log; the per-launch event times then include the other lane's
                        // contention and stop being a roofline measurement, which is why it is not the default)
    int first_chunk = 1;  // short first chunk of a call: 0 never, 1 when the GPU is idle, 2 always
    int threads = 0;
    int trace = 0;        // debug: one stderr line per launch
    int split_kinds = 0;  // profiling: one launch per (level, class of work) instead of one per level
    int overlap = 1;      // the sweep launch of a level goes to a second stream, beside the level kernel's launch of the same level (their
                          // items belong to different requests): the tail of one runs under the body of the other, +1-2 % (profiles/r03_k_overlap.log)
    int sweep_dma = 1;    // the sweep kernel: 1 = ve_sweep_dma_kernel (round 3: LDS-DMA fill, 16-byte LDS accesses, wave-local stage pairs,
                          // wave-owned tail); 0 = round 2's register-staged ve_sweep_kernel (reference for A/B runs and the bit-for-bit test)
    int gibbs_lds = 1;    // Gibbs: 1 = CPTs in LDS when they fit + the eight-lanes-per-chain form for grids (gibbs_kernel8); 2 = LDS, one chain per
                          // lane (round 3's kernel); 0 = always read the tables through L2
    int64_t chunk = 32768;  // requests per launch; planning of chunk i+1 overlaps the kernel of chunk i.  Round 3: 32 768 (16 384 before):
                            // half as many, twice as large launches - the tails of the ~150 launches of a chunk cost the same time
                            // whatever their size: 4.13-4.24 -> 4.30-4.38 TB/s over all kernels (profiles/r03_d_chunk.log); the
                            // arena of a C3 chunk doubles to ~145 GB of the 288
};

#define HIP_TRY(h, expr)                                                                              \
    do {                                                                                              \
        hipError_t e_ = (expr);                                                                       \
        if (e_ != hipSuccess) {                                                                       \
            (h)->err = std::string(#expr) + ": " + hipGetErrorString(e_);                             \
            return MIBN_E_HIP;                                                                        \
        }                                                                                             \
    } while (0)

// The arrays order_search.h reads, in one device buffer (rebuilt by set_network / set_order_hints).
static int upload_order_net(mibn_ctx *h) {
    h->order_net_ok = false;
    if (h->planner_only || h->net.n_vars > 128 || h->net.n_vars < 1) return MIBN_OK;
    const Network &net = h->net;
    const size_t n = (size_t)net.n_vars, nh = net.hint_sorted.size();
    std::vector<char> buf;
    auto put = [&](const void *p, size_t bytes) {
        const size_t off = (buf.size() + 15) & ~size_t(15);
        buf.resize(off + bytes);
        if (bytes) std::memcpy(buf.data() + off, p, bytes);
        return off;
    };
    const size_t o_card = put(net.card.data(), n * 4), o_depth = put(net.depth.data(), n * 4);
    const size_t o_asc = put(net.topo_asc.data(), n * 4), o_desc = put(net.topo_desc.data(), n * 4);
    const size_t o_hint = put(net.hint_flat.data(), nh * n * 4), o_log = put(net.log2card.data(), n * 8);
    const size_t o_anc = put(net.anc2.data(), n * sizeof(B2)), o_sc = put(net.scope2.data(), n * sizeof(B2));
    const size_t o_fam = put(net.fam2.data(), n * sizeof(B2));
    if (h->d_order_net) { HIP_TRY(h, hipFree(h->d_order_net)); h->d_order_net = nullptr; }
    HIP_TRY(h, hipMalloc(&h->d_order_net, buf.size()));
    HIP_TRY(h, hipMemcpy(h->d_order_net, buf.data(), buf.size(), hipMemcpyHostToDevice));
    OrderNet &o = h->order_net_dev;
    o = net.order_view();
    char *d = h->d_order_net;
    o.card = reinterpret_cast<const int32_t *>(d + o_card);
    o.depth = reinterpret_cast<const int32_t *>(d + o_depth);
    o.topo_asc = reinterpret_cast<const int32_t *>(d + o_asc);
    o.topo_desc = reinterpret_cast<const int32_t *>(d + o_desc);
    o.hint_sorted = reinterpret_cast<const int32_t *>(d + o_hint);
    o.log2card = reinterpret_cast<const double *>(d + o_log);
    o.anc = reinterpret_cast<const B2 *>(d + o_anc);
    o.cpt_scope = reinterpret_cast<const B2 *>(d + o_sc);
    o.fam = reinterpret_cast<const B2 *>(d + o_fam);
    h->order_net_ok = true;
    return MIBN_OK;
}

// The arrays emit_core.h reads, in one device buffer (rebuilt by set_network).  Networks the device order search covers only.
static int upload_emit_net(mibn_ctx *h) {
    h->emit_net_ok = false;
    if (h->planner_only || h->net.n_vars > 128 || h->net.n_vars < 1) return MIBN_OK;
    const Network &net = h->net;
    const size_t n = (size_t)net.n_vars;
    std::vector<char> buf;
    auto put = [&](const void *p, size_t bytes) {
        const size_t off = (buf.size() + 15) & ~size_t(15);
        buf.resize(off + bytes);
        if (bytes) std::memcpy(buf.data() + off, p, bytes);
        return off;
    };
    const size_t o_card = put(net.card.data(), n * 4), o_log = put(net.log2card.data(), n * 8), o_pool = put(net.pool_off.data(), n * 8);
    const size_t o_soff = put(net.scope_off32.data(), (n + 1) * 4), o_sv = put(net.scope_flat.data(), net.scope_flat.size() * 4);
    const size_t o_ss = put(net.cstride_flat.data(), net.cstride_flat.size() * 8), o_anc = put(net.anc_flat.data(), net.anc_flat.size() * 8);
    if (h->d_emit_net) { HIP_TRY(h, hipFree(h->d_emit_net)); h->d_emit_net = nullptr; }
    HIP_TRY(h, hipMalloc(&h->d_emit_net, buf.size()));
    HIP_TRY(h, hipMemcpy(h->d_emit_net, buf.data(), buf.size(), hipMemcpyHostToDevice));
    EmitNet &e = h->emit_net_dev;
    e = net.emit_view();
    char *d = h->d_emit_net;
    e.card = reinterpret_cast<const int32_t *>(d + o_card);
    e.log2card = reinterpret_cast<const double *>(d + o_log);
    e.pool_off = reinterpret_cast<const int64_t *>(d + o_pool);
    e.scope_off = reinterpret_cast<const int32_t *>(d + o_soff);
    e.scope_vars = reinterpret_cast<const int32_t *>(d + o_sv);
    e.scope_stride = reinterpret_cast<const int64_t *>(d + o_ss);
    e.anc = reinterpret_cast<const uint64_t *>(d + o_anc);
    h->emit_net_ok = true;
    return MIBN_OK;
}

extern "C" {

const char *mibn_version(void) { return "mibn 0.1 (gfx950)"; }

int mibn_device_count(int *count) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) n = 0;
    if (count) *count = n;
    return MIBN_OK;
}

// Factors of at most this many cells are "small" (segments, the small inputs of the passes, the byte model's big / small line).  1 024 from round 2 to the
// last day of round 6; with order_effort 1's plans 512 runs 1.3 % less GPU time per step (808.6 - 810.1 against 819.0 - 820.4 ms, interleaved; 256: 810.5; 2 048: 824.7 -
// profiles/r06_ca_ab.log, r06_cb_ab.log).  A bare Network (tools, oracle, the pinned fingerprints) keeps 1 024.
constexpr int kEngineSmallCells = 512;

int mibn_create_planner(mibn_t **out) {
    if (!out) return MIBN_E_ARG;
    auto *h = new mibn_ctx();
    h->net.order_effort = 1; h->net.second_above = h->base_second_above;  // (the engine's default; a bare Network - the tools, the oracle - keeps round 5's search)
    h->net.small_cells = kEngineSmallCells;
    h->net.minfill_above = h->base_minfill;
    h->planner_only = true;
    *out = h;
    return MIBN_OK;
}

int mibn_create(int device, mibn_t **out) {
    if (!out) return MIBN_E_ARG;
    *out = nullptr;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return MIBN_E_NODEVICE;
    if (device < 0 || device >= n) return MIBN_E_ARG;
    auto *h = new mibn_ctx();
    h->net.order_effort = 1; h->net.second_above = h->base_second_above;
    h->net.small_cells = kEngineSmallCells;
    h->net.minfill_above = h->base_minfill;
    h->device = device;
    if (hipSetDevice(device) != hipSuccess) { delete h; return MIBN_E_NODEVICE; }
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) != hipSuccess) { delete h; return MIBN_E_NODEVICE; }
    if (std::string(prop.gcnArchName).rfind("gfx950", 0) != 0) {  // kernels are built for gfx950 only
        delete h;
        return MIBN_E_NODEVICE;
    }
    h->n_cu = prop.multiProcessorCount;
    bool ok = hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking) == hipSuccess &&
              hipStreamCreateWithFlags(&h->stream2, hipStreamNonBlocking) == hipSuccess &&
              hipStreamCreateWithFlags(&h->aux[0][0], hipStreamNonBlocking) == hipSuccess &&
              hipStreamCreateWithFlags(&h->aux[0][1], hipStreamNonBlocking) == hipSuccess &&
              hipStreamCreateWithFlags(&h->aux[1][0], hipStreamNonBlocking) == hipSuccess &&
              hipStreamCreateWithFlags(&h->aux[1][1], hipStreamNonBlocking) == hipSuccess &&
              hipStreamCreateWithFlags(&h->aux[0][2], hipStreamNonBlocking) == hipSuccess &&
              hipStreamCreateWithFlags(&h->aux[1][2], hipStreamNonBlocking) == hipSuccess &&
              hipStreamCreateWithFlags(&h->copy_stream, hipStreamNonBlocking) == hipSuccess;
    // the sweep kernel keeps its 64 KiB tile, the T tables and the step descriptor in dynamic LDS (two workgroups per CU)
    ok = ok && hipFuncSetAttribute((const void *)ve_sweep_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, kSweepLdsBytes) == hipSuccess;
    ok = ok && hipFuncSetAttribute((const void *)ve_sweep_dma_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, kSweepLdsBytes) == hipSuccess;
    if (!ok) { delete h; return MIBN_E_HIP; }
    *out = h;
    return MIBN_OK;
}

void mibn_destroy(mibn_t *h) {
    if (!h) return;
    delete h->pool;
    if (!h->planner_only) {
        (void)hipSetDevice(h->device);
        if (h->stream) (void)hipStreamSynchronize(h->stream);
        (void)mibn_comm_destroy(h);
        (void)hipFree(h->d_pool);
        if (h->stream2) (void)hipStreamSynchronize(h->stream2);
        (void)hipFree(h->d_arena[0]);
        (void)hipFree(h->d_arena[1]);
        if (h->epoch) (void)hipEventDestroy(h->epoch);
        if (h->lane_ev) (void)hipEventDestroy(h->lane_ev);
        if (h->zero_ev) (void)hipEventDestroy(h->zero_ev);
        (void)hipFree(h->d_tiny_meta);
        (void)hipFree(h->d_tiny_req);
        (void)hipFree(h->d_order_net);
        (void)hipFree(h->d_orders);
        (void)hipFree(h->d_order_len);
        (void)hipFree(h->d_order_scratch);
        (void)hipFree(h->d_emit_net);
        (void)hipFree(h->d_wnet);
        delete h->wnet_host;
        (void)hipFree(h->d_emit_scratch);
        (void)hipFree(h->d_emit_cursor);
        if (h->d_plan_perm) (void)hipFree(h->d_plan_perm);
        if (h->emit_in.p) (void)hipHostFree(h->emit_in.p);
        if (h->emit_out.p) (void)hipHostFree(h->emit_out.p);
        if (h->search_in.p) (void)hipHostFree(h->search_in.p);
        if (h->search_out.p) (void)hipHostFree(h->search_out.p);
        if (h->search_stream) (void)hipStreamDestroy(h->search_stream);
        if (h->tiny_stage.p) (void)hipHostFree(h->tiny_stage.p);
        for (int k = 0; k < 2; ++k) {
            (void)hipFree(h->d_results[k]);
            if (h->pend[k].done) (void)hipEventDestroy(h->pend[k].done);
        }
        for (auto &st : h->set) {
            for (auto &b : st.bufs)
                if (b.data) (void)hipHostFree(b.data);
            for (auto &b : st.xbufs)
                if (b.data) (void)hipHostFree(b.data);
            (void)hipFree(st.d_prog);
            (void)hipFree(st.d_prog_off);
            (void)hipFree(st.d_arena_off);
            (void)hipFree(st.d_items);
            (void)hipFree(st.d_wg_item);
            for (auto &sg : st.stage)
                if (sg.p) (void)hipHostFree(sg.p);
            for (auto e : st.ev) (void)hipEventDestroy(e);
            if (st.uploaded) (void)hipEventDestroy(st.uploaded);
        }
        for (auto &sg : h->res_stage)
            if (sg.p) (void)hipHostFree(sg.p);
        if (h->copy_stream) (void)hipStreamDestroy(h->copy_stream);
        if (h->stream2) (void)hipStreamDestroy(h->stream2);
        for (auto &la : h->aux)
            for (hipStream_t a : la)
                if (a) (void)hipStreamDestroy(a);
        if (h->stream) (void)hipStreamDestroy(h->stream);
    }
    delete h;
}

const char *mibn_last_error(const mibn_t *h) { return h ? h->err.c_str() : "null handle"; }

int mibn_set_option(mibn_t *h, const char *name, double value) {
    if (!h || !name) return MIBN_E_ARG;
    std::string n(name);
    if (n == "arena_gb") h->arena_gb = value;
    else if (n == "threads") h->threads = (int)value;
    else if (n == "chunk") h->chunk = std::max<int64_t>(1, (int64_t)value);
    else if (n == "big_iters") h->net.big_iters = std::max<int64_t>(1, (int64_t)value);  // test hooks: force tiling
    else if (n == "tile_h") h->net.tile_h = std::max(0, std::min(kTileMax, (int)value));  // 0 = sized by traffic
    else if (n == "tile_kb") h->net.tile_bytes = std::max<int64_t>(16, (int64_t)value) << 10;  // traffic per tile when tile_h = 0
    else if (n == "chunk_sets") { if (mibn_drain(h) != MIBN_OK) return MIBN_E_HIP; h->n_sets = std::max(2, std::min(kChunkSets, (int)value)); h->set_cursor = 0; }
    else if (n == "trace") h->trace = (int)value;
    else if (n == "split_kinds") h->split_kinds = value != 0;
    else if (n == "overlap") { if (mibn_drain(h) != MIBN_OK) return MIBN_E_HIP; h->overlap = value != 0; }
    else if (n == "sweep_dma") h->sweep_dma = value != 0;
    else if (n == "seg_kernel") h->seg_kernel = value != 0;
    else if (n == "gibbs_lds") h->gibbs_lds = std::max(0, std::min(2, (int)value));
    else if (n == "tiny") h->tiny = value != 0;
    else if (n == "tiny_zero_copy") h->tiny_zero_copy = value != 0;
    else if (n == "mfma_kernel") h->mfma_kernel = value != 0;  // (round 6 measured CHAIN and OUTER in ve_mfma_kernel too: at its 128 registers they spill 74 and the
                                                                //  whole kernel pays - 291 against 298 k queries/s in three interleaved repetitions, profiles/r06_y_ab.log)
    else if (n == "gpu_emit") h->gpu_emit = std::max(0, std::min(2, (int)value));  // whole chunks planned on the device (order search + program emission)
    else if (n == "wave_wgs") h->wave_wgs = std::max(0, (int)value);
    else if (n == "plan_priority") h->plan_priority = std::max(0, std::min(2, (int)value));
    else if (n == "wave_plan") h->wave_plan = value != 0;  // 0: the device plans with order_kernel + emit_kernel (one request per lane)
    else if (n == "plan_waves") h->plan_waves = std::max(1, std::min(16, (int)value));  // waves per workgroup of the device planner's kernels
    else if (n == "plan_sort") h->plan_sort = (int)value;
    else if (n == "wave_sort") h->wave_sort = value != 0;  // wave_plan_kernel hands out the requests with the most relevant variables first (default 1)
    else if (n == "plan_lanes") h->plan_lanes = std::max(0, std::min(64, (int)value));  // requests per wave of the device planner's kernels
    else if (n == "emit_share") { h->emit_share_opt = value > 0 ? std::min(1.0, value) : -1; if (value > 0) h->emit_share = h->emit_share_opt; }  // the device's share of a chunk (<= 0: follows the measured rates)
    else if (n == "emit_words") h->emit_words = (uint32_t)std::max(1024, std::min(1 << 20, (int)value));  // words of a request's device program slot
    else if (n == "gpu_search") h->gpu_search = std::max(0, std::min(2, (int)value));  // elimination-order search on the device
        // (order_kernel): 1 = the first chunk of a call on the host, the rest of the call by one launch; 2 = every chunk, synchronously (tests)
    else if (n == "adaptive") {
        h->adaptive = value != 0;
        if (!h->adaptive) {  // what the policy had switched goes back
            h->net.minfill_above = h->base_minfill;
            if (h->auto_emit) { h->gpu_emit = 0; h->auto_emit = false; }
            if (h->auto_search) { h->gpu_search = 0; h->auto_search = false; }
            h->adaptive_seeded = false;
        }
    }
    else if (n == "minfill_above") { h->base_minfill = value; h->net.minfill_above = value; }  // bytes of the best sweep above which min-fill runs  // small-network kernel (one lane per request, no planning) where eligible
    else if (n == "fuse") h->net.fuse = value != 0;
    else if (n == "plan_cache") h->net.plan_cache = value != 0;  // plan templates for repeated request shapes
    else if (n == "chain") h->net.chain = value != 0;  // CHAIN form: three variables per pass
    else if (n == "streams") { if (mibn_drain(h) != MIBN_OK) return MIBN_E_HIP; h->n_streams = value >= 2 ? 2 : 1; h->n_sets = 2 * h->n_streams; h->set_cursor = 0; }  // lanes: 2 (default) = consecutive chunks run concurrently on two streams; 1 = one after the other
    else if (n == "first_chunk") h->first_chunk = std::max(0, std::min(2, (int)value));
    else if (n == "builtin_sweeps") {  // two depth-first topological orders as candidate elimination orders (rebuilds the hint lists)
        h->net.builtin_sweeps = value != 0;
        if (h->net.n_vars > 0) {
            std::vector<int32_t> flat;
            for (const auto &hh : h->net.hints) flat.insert(flat.end(), hh.begin(), hh.end());
            h->net.set_hints((int32_t)h->net.hints.size(), flat.data());
            if (!h->planner_only) return upload_order_net(h);
        }
    }
    else if (n == "order_effort") h->net.order_effort = std::max(0, std::min(1, (int)value));  // 1: more candidate orders, the byte model's best two both emitted where the best is expensive (order_search.h)
    else if (n == "second_above") { h->base_second_above = value; h->net.second_above = value; }  // modelled bytes above which the runner-up order is emitted too
    else if (n == "second_on_device") h->second_on_device = value != 0;  // 1: also in the calls the device plans (default: those run without the second emission)
    else if (n == "order_weights") h->net.order_weights = std::max(0, std::min(64, (int)value));  // class-weighted byte model of the order search (0: plain section-8(d) bytes)
    else if (n == "sweep_canon") h->net.sweep_canon = value != 0;  // test hook: 0 = the sweep kernel's general path for every step
    else if (n == "sweep_adapt") h->net.sweep_adapt = std::max(0, (int)value);  // fewer tiles per workgroup in sweep launches below this many workgroups
    else if (n == "sweep_min") h->net.sweep_min = std::max(2, std::min(5, (int)value));  // fewest variables of a SWEEP pass
    else if (n == "sweep_iters") h->net.sweep_iters = std::max(1, std::min(kTileMax, (int)value));  // tiles per workgroup of the sweep kernel
    else if (n == "sweep") h->net.sweep = std::max(0, std::min(5, (int)value));  // SWEEP form: up to this many variables per pass, tile in LDS (0 / < 3: off)
    else if (n == "outer") h->net.outer = value != 0;
    else if (n == "stagger") h->net.stagger = std::max(1, std::min(8, (int)value));  // groups of requests with staggered levels per chunk
    else if (n == "prune") h->net.prune = value != 0;  // 0: multiply every CPT (full_joint_dist / predict_proba semantics)  // OUTER (MFMA) form for products of two big tables  // joint elimination of two variables per pass
    else if (n == "small_cells") h->net.small_cells = std::max(1, std::min(kMaxT, (int)value));  // test hook: forces FIBER steps on small networks
    else { h->err = "unknown option " + n; return MIBN_E_ARG; }
    return MIBN_OK;
}

int mibn_set_network(mibn_t *h, int32_t n_vars, const int32_t *card, const int64_t *scope_off,
                     const int32_t *scope_vars, const int64_t *value_off, const double *values) {
    if (!h || !card || !scope_off || !scope_vars || !value_off || !values) return MIBN_E_ARG;
    std::string e = h->net.set(n_vars, card, scope_off, scope_vars, value_off, values);
    if (!e.empty()) { h->err = e; h->has_net = false; return MIBN_E_ARG; }
    h->has_net = true;
    h->adaptive_seeded = false;  // (the adaptive policy starts over with a new network)
    if (h->auto_emit) { h->gpu_emit = 0; h->auto_emit = false; }
    if (!h->planner_only) {
        HIP_TRY(h, hipSetDevice(h->device));
        if (h->d_pool) { HIP_TRY(h, hipFree(h->d_pool)); h->d_pool = nullptr; }
        size_t bytes = std::max<size_t>(8, h->net.pool.size() * sizeof(double));
        HIP_TRY(h, hipMalloc(&h->d_pool, bytes));
        HIP_TRY(h, hipMemcpy(h->d_pool, h->net.pool.data(), h->net.pool.size() * sizeof(double), hipMemcpyHostToDevice));
        h->tiny_ok = tiny_eligible(h->net);
        if (h->d_tiny_meta) { HIP_TRY(h, hipFree(h->d_tiny_meta)); h->d_tiny_meta = nullptr; }
        if (h->tiny_ok) {
            const std::vector<int32_t> meta = tiny_meta(h->net);
            h->tiny_meta_words = (int32_t)meta.size();
            HIP_TRY(h, hipMalloc(&h->d_tiny_meta, meta.size() * 4));
            HIP_TRY(h, hipMemcpy(h->d_tiny_meta, meta.data(), meta.size() * 4, hipMemcpyHostToDevice));
            h->tiny_ok = tiny_lds_bytes((int)h->net.pool.size(), h->tiny_meta_words) <= 64 * 1024;
        }
        if (int rc = upload_emit_net(h)) return rc;
        return upload_order_net(h);
    }
    return MIBN_OK;
}

int mibn_set_order_hints(mibn_t *h, int32_t n_hints, const int32_t *priorities) {
    if (!h || n_hints < 0 || (n_hints && !priorities)) return MIBN_E_ARG;
    if (!h->has_net) { h->err = "set_network first"; return MIBN_E_STATE; }
    h->net.set_hints(n_hints, priorities);
    if (!h->planner_only) return upload_order_net(h);
    return MIBN_OK;
}

int mibn_last_stats(const mibn_t *h, mibn_stats *out) {
    if (!h || !out) return MIBN_E_ARG;
    *out = h->stats;
    return MIBN_OK;
}

int mibn_plan_stats(mibn_t *h, int32_t n_q, const int32_t *q_vars, int32_t n_e, const int32_t *e_vars,
                    mibn_stats *out) {
    if (!h || !out) return MIBN_E_ARG;
    if (!h->has_net) { h->err = "set_network first"; return MIBN_E_STATE; }
    Request rq;
    rq.nq = n_q; rq.qvars = q_vars; rq.ne = n_e; rq.evars = e_vars;
    std::string e = validate_request(h->net, rq);
    if (!e.empty()) { h->err = e; return MIBN_E_ARG; }
    std::vector<uint32_t> prog;
    PlanStats st;
    e = plan_request(h->net, rq, prog, st);
    if (!e.empty()) { h->err = e; return MIBN_E_LIMIT; }
    std::memset(out, 0, sizeof(*out));
    out->alg_bytes = st.alg_bytes;
    out->alg_flops = st.alg_flops;
    out->n_steps = st.n_steps;
    out->max_step_cells = st.max_step_cells;
    out->arena_bytes = 8.0 * (double)st.arena_cells;
    return MIBN_OK;
}

}  // extern "C"

namespace {

template <class T>
int ensure(mibn_ctx *h, T *&ptr, size_t &cap, size_t need) {
    if (need <= cap) return MIBN_OK;
    if (ptr) { HIP_TRY(h, hipFree(ptr)); ptr = nullptr; cap = 0; }
    size_t n = need + need / 2 + 1024;
    HIP_TRY(h, hipMalloc(&ptr, n * sizeof(T)));
    cap = n;
    return MIBN_OK;
}

// pinned backing of the program buffers: the upload is then a real async DMA that overlaps planning
uint32_t *pinned_grow(void *, uint32_t *old, size_t used, size_t new_cap) {
    uint32_t *p = nullptr;
    if (hipHostMalloc((void **)&p, new_cap * sizeof(uint32_t), hipHostMallocDefault) != hipSuccess) return nullptr;
    if (old) {
        std::memcpy(p, old, used * sizeof(uint32_t));
        (void)hipHostFree(old);
    }
    return p;
}

// async upload through a pinned staging buffer (one per destination and set: reused only after retire())
int upload(mibn_ctx *h, mibn_ctx::Staging &sg, void *dst, const void *src, size_t bytes) {
    if (!bytes) return MIBN_OK;
    if (bytes > sg.cap) {
        if (sg.p) { HIP_TRY(h, hipHostFree(sg.p)); sg.p = nullptr; sg.cap = 0; }
        const size_t cap = bytes + bytes / 4 + 4096;
        HIP_TRY(h, hipHostMalloc((void **)&sg.p, cap, hipHostMallocDefault));
        sg.cap = cap;
    }
    std::memcpy(sg.p, src, bytes);
    HIP_TRY(h, hipMemcpyAsync(dst, sg.p, bytes, hipMemcpyHostToDevice, h->copy_stream));
    return MIBN_OK;
}

// CPUs this process may actually use: the cgroup CPU quota when there is one (a container that sees 256 hardware
// threads may be limited to 16 CPUs' worth of time; more runnable threads than ~2x the quota only get throttled)
double cpu_quota() {
    double q = 0;
    if (FILE *f = std::fopen("/sys/fs/cgroup/cpu.max", "r")) {  // cgroup v2: "<quota|max> <period>"
        char a[64] = {0};
        double period = 0;
        if (std::fscanf(f, "%63s %lf", a, &period) == 2 && std::strcmp(a, "max") != 0 && period > 0) q = std::atof(a) / period;
        std::fclose(f);
    } else if (FILE *g = std::fopen("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "r")) {  // cgroup v1
        double quota = 0, period = 0;
        if (std::fscanf(g, "%lf", &quota) == 1 && quota > 0) {
            if (FILE *p = std::fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r")) {
                if (std::fscanf(p, "%lf", &period) == 1 && period > 0) q = quota / period;
                std::fclose(p);
            }
        }
        std::fclose(g);
    }
    return q;
}

int default_threads() {
    int hw = (int)std::thread::hardware_concurrency();
    int local_world = 1;
    if (const char *e = std::getenv("LOCAL_WORLD_SIZE")) local_world = std::max(1, std::atoi(e));
    double cpus = hw;
    const double quota = cpu_quota();
    if (quota > 0) cpus = std::min(cpus, 2.0 * quota);  // planning is bursty: 2 threads per quota CPU finish a chunk sooner inside the same
                                                        // CPU-time budget; 4 per CPU (round 1's choice) run into the cgroup's throttling now that
                                                        // a step is 135 ms instead of 170 (profiles/r02_y_threads.log: 16-32 threads 237-239 k
                                                        // queries/s, 64 threads 235 k, 128 threads 221 k with 44 thread-seconds throttled)
    return std::max(1, std::min(64, (int)(cpus / local_world)));
}

void ensure_pool(mibn_ctx *h) {
    if (h->pool) return;
    h->pool = new ThreadPool(h->threads > 0 ? h->threads : default_threads());
    for (auto &st : h->set) {
        st.bufs.resize(h->pool->size());
        st.xbufs.resize(h->pool->size());
        if (!h->planner_only) {
            for (auto &b : st.bufs) b.grow = pinned_grow;
            for (auto &b : st.xbufs) b.grow = pinned_grow;
        }
    }
}

// name of statistics slot k: the classes of work (split_kinds), then the kernels as launched
const char *stat_name(int k) {
    return k < kNumKernels ? kernel_name(k) : (k == kNumKernels ? "ve_level_kernel" : (k == kNumKernels + 1 ? "tiny_kernel" : (k == kNumKernels + 2 ? "ve_sweep_dma_kernel" : (k == kNumKernels + 3 ? "order_kernel+emit_kernel" : (k == kNumKernels + 4 ? "level:ve_level_kernel||ve_mfma_kernel||ve_sweep_dma_kernel||ve_segment_kernel" : (k == kNumKernels + 5 ? "ve_segment_kernel" : "ve_mfma_kernel"))))));
}

// wait for a set's launches and book their HIP-event durations per kernel
int retire(mibn_ctx *h, mibn_ctx::Set &st) {
    if (!st.busy) return MIBN_OK;
    HIP_TRY(h, hipEventSynchronize(st.ev[st.ev_used - 1]));
    for (auto &t : st.timed) {
        float ms = 0;
        if (t.kid != -2) HIP_TRY(h, hipEventElapsedTime(&ms, st.ev[t.e0], st.ev[t.e1]));  // (-2: a level's launch group - any of its event pairs may be absent)
        const bool mine = t.call == h->call_id;  // (launches of an earlier asynchronous call only count in the totals)
        if (t.kid == -1) {
            h->retired_requests += t.items;  // (the wave's requests)
            // GPU time of a wave, first launch to last.  With two lanes the waves of consecutive chunks overlap: what is
            // booked as kernel_ms is the time the GPU was busy (the union of the intervals, kept as a high-water mark
            // since the epoch event; sets retire in launch order)
            float t0 = 0, t1 = 0;
            HIP_TRY(h, hipEventElapsedTime(&t0, h->epoch, st.ev[t.e0]));
            HIP_TRY(h, hipEventElapsedTime(&t1, h->epoch, st.ev[t.e1]));
            const double add = std::max(0.0, (double)t1 - std::max((double)t0, h->busy_until));
            h->busy_until = std::max(h->busy_until, (double)t1);
            if (mine) h->stats.kernel_ms += add;
            h->total.kernel_ms += add;
            continue;
        }
        if (t.kid == -2) {
            // one level = one concurrent launch pair: from the earlier start to the later end (timestamps against the epoch event)
            double lo = 1e300, hi = -1e300;
            for (size_t e : {t.e0, t.e2, t.e4, t.e6})
                if (e != mibn_ctx::Set::Timed::kNone) { float x = 0; HIP_TRY(h, hipEventElapsedTime(&x, h->epoch, st.ev[e])); lo = std::min(lo, (double)x); }
            for (size_t e : {t.e1, t.e3, t.e5, t.e7})
                if (e != mibn_ctx::Set::Timed::kNone) { float x = 0; HIP_TRY(h, hipEventElapsedTime(&x, h->epoch, st.ev[e])); hi = std::max(hi, (double)x); }
            for (mibn_kernel_stat *ks : {&h->kstats[kNumKernels + 4], &h->ktotal[kNumKernels + 4]}) {
                if (ks == &h->kstats[kNumKernels + 4] && !mine) continue;
                if (!ks->name[0]) std::snprintf(ks->name, sizeof(ks->name), "%s", stat_name(kNumKernels + 4));
                ks->launches += 1;
                ks->ms += std::max(0.0, hi - lo);
                ks->alg_bytes += t.bytes;
                ks->items += t.items;
            }
            continue;
        }
        if (mine) h->stats.n_launches += 1;
        h->total.n_launches += 1;
        for (mibn_kernel_stat *ks : {&h->kstats[t.kid], &h->ktotal[t.kid]}) {
            if (ks == &h->kstats[t.kid] && !mine) continue;
            if (!ks->name[0]) std::snprintf(ks->name, sizeof(ks->name), "%s", stat_name(t.kid));
            ks->launches += 1;
            ks->ms += ms;
            ks->alg_bytes += t.bytes;
            ks->items += t.items;
        }
        if (h->trace) std::fprintf(stderr, "[mibn launch] %-32s wgs %8.0f MB %10.2f ms %8.4f -> %7.1f GB/s\n", h->kstats[t.kid].name, t.items, t.bytes / 1e6, ms, t.bytes / ms / 1e6);
    }
    if (h->trace && st.gap_from >= 0 && st.ev_used > 0) {
        float gap = 0;
        if (hipEventElapsedTime(&gap, h->gap_ev[st.gap_from], st.ev[0]) == hipSuccess)
            std::fprintf(stderr, "[mibn gap] %.3f ms between the end of the previous wave and the first launch of this one\n", gap);
        st.gap_from = -1;
    }
    st.timed.clear();
    st.ev_used = 0;
    st.busy = false;
    return MIBN_OK;
}

// every set, oldest launches first (set_cursor points at the set that is re-used next = the oldest): the busy-time
// bookkeeping of retire() wants the waves in launch order
int retire_all(mibn_ctx *h) {
    for (int i = 0; i < h->n_sets; ++i) {
        int rc = retire(h, h->set[(h->set_cursor + i) % h->n_sets]);
        if (rc) return rc;
    }
    for (auto &st : h->set) {  // (sets beyond n_sets: idle unless the option has just changed)
        int rc = retire(h, st);
        if (rc) return rc;
    }
    return MIBN_OK;
}

int next_event(mibn_ctx *h, mibn_ctx::Set &st, size_t &idx, hipStream_t stream = nullptr) {
    if (st.ev_used == st.ev.size()) {
        hipEvent_t e;
        HIP_TRY(h, hipEventCreate(&e));
        st.ev.push_back(e);
    }
    idx = st.ev_used++;
    HIP_TRY(h, hipEventRecord(st.ev[idx], stream ? stream : h->stream));
    return MIBN_OK;
}

}  // namespace

namespace {
int pinned(mibn_ctx *h, mibn_ctx::Staging &sg, size_t bytes) {
    if (bytes <= sg.cap) return MIBN_OK;
    if (sg.p) { HIP_TRY(h, hipHostFree(sg.p)); sg.p = nullptr; sg.cap = 0; }
    HIP_TRY(h, hipHostMalloc((void **)&sg.p, bytes + bytes / 4 + 4096, hipHostMallocDefault));
    sg.cap = bytes + bytes / 4 + 4096;
    return MIBN_OK;
}

// Requests per wave of the device planner's kernels.  The lanes of a wave run different requests, every data-dependent branch
// diverges, so a wave's time grows with its lanes in use: fewer lanes in more waves plan faster and take more of the chip from the
// VE kernels.  A rank of one or two planning threads is bound by its planning capacity (host + device), not by its kernels: 24
// lanes (230 -> 246 k queries/s at one and two threads); from four threads on the kernels' share of the chip counts for more: 32
// (258 / 254 k at 24 / 32 lanes and four threads, 263 / 269 k at six; profiles/r05_f_planlanes.log, r05_g_planlanes.log).
// (Calls of 32 768 requests - such a rank's shard of a 2^18-request step on eight GPUs - are bound by the kernels again, whatever the lanes:
//  250.3 / 251.5 k at 24 / 32 lanes and two threads, 240 / 250 k at one: 32 there.  profiles/r05_k_calls32k.log)
// chunks the device plans go through wave_plan_kernel (its time grows with the requests it is given - order_kernel / emit_kernel are
// latency-bound: theirs hardly does): the host's share of a chunk may shrink to a hundredth
static bool wave_mode(const mibn_ctx *h) { return h->wave_plan && h->wnet_ok; }

static int plan_lanes_now(const mibn_ctx *h) {
    if (h->plan_lanes > 0) return h->plan_lanes;
    return h->pool && h->pool->size() <= 2 && h->chunk > 32768 ? 24 : 32;
}

// Device order search for requests [b0, b1): uploads their query / evidence variables and launches order_kernel on a
// high-priority stream of its own (it must not queue behind the level kernels of the previous chunk); the orders (128
// bytes per request) and their lengths land in h->search_out.  Asynchronous: search_wait() before they are read.  The
// kernel is latency-bound with one request per lane - its duration hardly depends on the number of requests - so the
// whole remainder of a call is searched by one launch (slices of kSearchSlice requests share the scratch buffer).
constexpr int64_t kSearchSlice = 65536;  // 1.3 GB of search state
int search_orders_async(mibn_ctx *h, uint32_t flags, int64_t b0, int64_t b1, const int64_t *q_off, const int32_t *q_vars,
                        const int64_t *e_off, const int32_t *e_vars) {
    const int64_t n = b1 - b0;
    int rc;
    if (!h->search_stream) {
        int lo = 0, hi = 0;
        HIP_TRY(h, hipDeviceGetStreamPriorityRange(&lo, &hi));
        HIP_TRY(h, hipStreamCreateWithPriority(&h->search_stream, hipStreamNonBlocking, h->plan_priority >= 2 ? hi : (h->plan_priority == 1 ? (lo + hi) / 2 : lo)));
    }
    const size_t nq = (size_t)(q_off[b1] - q_off[b0]), ne = (size_t)(e_off[b1] - e_off[b0]);
    const size_t off_bytes = (size_t)(n + 1) * 8;
    const size_t in_bytes = 2 * off_bytes + (nq + ne) * 4 + 64;
    if ((rc = pinned(h, h->search_in, in_bytes))) return rc;
    if ((rc = pinned(h, h->search_out, (size_t)n * 132))) return rc;
    if ((rc = ensure(h, h->d_order_scratch, h->order_scratch_cap, (size_t)std::min(n, kSearchSlice)))) return rc;
    int64_t *qo = reinterpret_cast<int64_t *>(h->search_in.p), *eo = qo + (n + 1);
    for (int64_t i = 0; i <= n; ++i) { qo[i] = q_off[b0 + i] - q_off[b0]; eo[i] = e_off[b0 + i] - e_off[b0]; }
    char *pv = h->search_in.p + 2 * off_bytes;
    std::memcpy(pv, q_vars + q_off[b0], nq * 4);
    if (ne) std::memcpy(pv + nq * 4, e_vars + e_off[b0], ne * 4);
    // (no copies on this stream - the kernel reads the request arrays from, and writes the orders to, pinned host memory: a DMA
    // copy here queues behind the previous call's result download on the copy engine, i.e. behind that call's kernels)
    const char *d_req = h->search_in.p;
    uint8_t *orders_out = reinterpret_cast<uint8_t *>(h->search_out.p);
    int32_t *len_out = reinterpret_cast<int32_t *>(h->search_out.p + (size_t)n * 128);
    for (int64_t s0 = 0; s0 < n; s0 += kSearchSlice) {
        const int64_t m = std::min(kSearchSlice, n - s0);
        OrderArgs A;
        A.net = h->order_net_dev;
        A.net.prune = h->net.prune;
        A.net.minfill_above = h->net.minfill_above;
        {   // (options set after the network: the model weights follow them like on the host)
            const OrderNet hv = h->net.order_view();
            A.net.chain_weight = hv.chain_weight;
            A.net.big_cells = hv.big_cells;
            A.net.effort = hv.effort;
        }
        A.q_off = reinterpret_cast<const int64_t *>(d_req) + s0;
        A.e_off = reinterpret_cast<const int64_t *>(d_req + off_bytes) + s0;
        A.q_vars = reinterpret_cast<const int32_t *>(d_req + 2 * off_bytes);
        A.e_vars = A.q_vars + nq;
        A.B = m;
        A.flags = flags;
        A.orders = orders_out + s0 * 128;
        A.order_len = len_out + s0;
        A.scratch = h->d_order_scratch;
        A.zero = nullptr;
        A.perm = nullptr;
        A.lanes = plan_lanes_now(h);
        {
            const int64_t per_wg = (int64_t)A.lanes * h->plan_waves;
            hipLaunchKernelGGL(order_kernel, dim3((unsigned)((m + per_wg - 1) / per_wg)), dim3(64 * h->plan_waves), 0, h->search_stream, A);
        }
        HIP_TRY(h, hipGetLastError());
    }
    return MIBN_OK;
}

int search_wait(mibn_ctx *h) {
    HIP_TRY(h, hipStreamSynchronize(h->search_stream));
    return MIBN_OK;
}

// Device planning of requests [b0, b1) - one chunk - into set `st`: order_kernel, then emit_kernel, on the planning stream (high
// priority: its few waves must not queue behind the level kernels of the chunk in flight); the programs stay on the device
// (st.d_prog, one slot of h->emit_words words per request), the per-request results and the work items come back and fill
// the BatchPlan like plan_batch would.  Returns MIBN_OK, an error, or 1: a request did not fit its slot / the item buffer
// (the caller plans the chunk on the host; the slots double for the next chunk).
constexpr int64_t kPlanSlice = 32768;  // requests per launch pair: 1.1 GB of search state + ~8 GB of emission state (100 variables)
int plan_on_device_launch(mibn_ctx *h, uint32_t flags, int64_t b0, int64_t b1, const int64_t *q_off, const int32_t *q_vars, const int64_t *e_off,
                          const int32_t *e_vars, const int32_t *e_codes, const int64_t *out_off, const char *skip, mibn_ctx::Set &st, size_t prog_words) {
    const int64_t n = b1 - b0;
    int rc;
    if (!h->search_stream) {
        int lo = 0, hi = 0;
        HIP_TRY(h, hipDeviceGetStreamPriorityRange(&lo, &hi));
        HIP_TRY(h, hipStreamCreateWithPriority(&h->search_stream, hipStreamNonBlocking, h->plan_priority >= 2 ? hi : (h->plan_priority == 1 ? (lo + hi) / 2 : lo)));
    }
    hipStream_t P = h->search_stream;
    const size_t nq = (size_t)(q_off[b1] - q_off[b0]), ne = (size_t)(e_off[b1] - e_off[b0]);
    const size_t off_bytes = (size_t)(n + 1) * 8;
    const size_t perm_off = (3 * off_bytes + (nq + 2 * ne) * 4 + (size_t)n + 63) & ~(size_t)63;
    const size_t in_bytes = perm_off + (size_t)n * 4 + 64;
    const size_t stride = h->emit_words;
    const size_t tag_cap = (size_t)n * 64;
    const size_t scratch_stride = emit_scratch_bytes(h->net.n_vars);
    const int64_t slice = std::min(n, kPlanSlice);
    // the wave-cooperative planner (wave_plan_kernel) where the network and the options of the moment are covered: its packed copy of
    // them is rebuilt per chunk (10 KB) and uploaded when it has changed
    bool wave = false;
    if (h->wave_plan && h->order_net_ok) {
        if (!h->wnet_host) { h->wnet_host = new WNet; std::memset(h->wnet_host, 0, sizeof(WNet)); h->wnet_ok = false; }
        WNet *fresh = new WNet;
        if (h->net.wave_view(*fresh)) {
            if (!h->d_wnet) HIP_TRY(h, hipMalloc(&h->d_wnet, sizeof(WNet)));
            if (!h->wnet_ok || std::memcmp(fresh, h->wnet_host, sizeof(WNet)) != 0) {
                HIP_TRY(h, hipMemcpy(h->d_wnet, fresh, sizeof(WNet), hipMemcpyHostToDevice));
                std::memcpy(h->wnet_host, fresh, sizeof(WNet));
                h->wnet_ok = true;
            }
            wave = true;
        }
        delete fresh;
    }
    if ((rc = pinned(h, h->emit_in, in_bytes))) return rc;
    if ((rc = pinned(h, h->emit_out, (size_t)n * sizeof(EmitMeta) + tag_cap * sizeof(Tag) + 64))) return rc;
    if (!h->d_emit_cursor) HIP_TRY(h, hipMalloc(&h->d_emit_cursor, 64));
    if (!wave) {
    if ((rc = ensure(h, h->d_orders, h->orders_cap, (size_t)n * 128))) return rc;
    if ((rc = ensure(h, h->d_order_len, h->order_len_cap, (size_t)n))) return rc;
    if ((rc = ensure(h, h->d_order_scratch, h->order_scratch_cap, (size_t)slice))) return rc;
    }
    if (!wave && (size_t)slice * scratch_stride > h->emit_scratch_cap) {
        if (h->d_emit_scratch) { HIP_TRY(h, hipFree(h->d_emit_scratch)); h->d_emit_scratch = nullptr; h->emit_scratch_cap = 0; }
        const size_t want = (size_t)std::max<int64_t>(slice, std::min<int64_t>(h->chunk, kPlanSlice)) * scratch_stride;
        HIP_TRY(h, hipMalloc(&h->d_emit_scratch, want));
        h->emit_scratch_cap = want;
    }
    if ((rc = ensure(h, st.d_prog, st.prog_cap, std::max(prog_words, (size_t)n * stride) + kMaxStepWords))) return rc;  // (slack: segment_wave prefetches whole descriptor slots)  // (+ room for the host's share of the chunk)
    if (!h->emit_ev[0]) { HIP_TRY(h, hipEventCreate(&h->emit_ev[0])); HIP_TRY(h, hipEventCreate(&h->emit_ev[1])); }
    // the request arrays of the chunk in one pinned buffer: [q_off | e_off | out_off | q_vars | e_vars | e_codes | skip]
    char *pin = h->emit_in.p;
    int64_t *qo = reinterpret_cast<int64_t *>(pin), *eo = qo + (n + 1), *oo = eo + (n + 1);
    for (int64_t i = 0; i <= n; ++i) { qo[i] = q_off[b0 + i] - q_off[b0]; eo[i] = e_off[b0 + i] - e_off[b0]; oo[i] = out_off[b0 + i] - out_off[b0]; }
    char *pv = pin + 3 * off_bytes;
    std::memcpy(pv, q_vars + q_off[b0], nq * 4);
    if (ne) { std::memcpy(pv + nq * 4, e_vars + e_off[b0], ne * 4); std::memcpy(pv + (nq + ne) * 4, e_codes + e_off[b0], ne * 4); }
    std::memcpy(pv + (nq + 2 * ne) * 4, skip + b0, (size_t)n);
    // lane position -> request, slice by slice: descending size of the relevant set (query, evidence and their ancestors: what the order
    // search and the emission loop over), a counting sort on 129 keys
    int32_t *perm = reinterpret_cast<int32_t *>(pin + perm_off);
    const bool sorted = h->plan_sort && h->net.n_vars <= 128 && !h->net.anc2.empty();
    if (sorted) {
        std::vector<uint8_t> key((size_t)std::min(n, kPlanSlice));
        for (int64_t s0 = 0; s0 < n; s0 += kPlanSlice) {
            const int64_t m = std::min(kPlanSlice, n - s0);
            int32_t count[130] = {0};
            for (int64_t i = 0; i < m; ++i) {
                const int64_t b = b0 + s0 + i;
                B2 rel;
                for (int64_t k = q_off[b]; k < q_off[b + 1]; ++k) { const int v = q_vars[k]; rel.set(v); rel.a |= h->net.anc2[v].a; rel.b |= h->net.anc2[v].b; }
                for (int64_t k = e_off[b]; k < e_off[b + 1]; ++k) { const int v = e_vars[k]; rel.set(v); rel.a |= h->net.anc2[v].a; rel.b |= h->net.anc2[v].b; }
                const int c = 128 - b2_count(rel);  // (descending)
                key[(size_t)i] = (uint8_t)c;
                ++count[c + 1];
            }
            for (int c = 0; c < 129; ++c) count[c + 1] += count[c];
            if (h->plan_sort == 1) {
                for (int64_t i = 0; i < m; ++i) perm[s0 + count[key[(size_t)i]]++] = (int32_t)i;
            } else {
                // plan_sort = 2: the sorted requests dealt round-robin over the workgroups - every workgroup (= CU) gets the same mix of
                // sizes, and inside it the waves are homogeneous (rank r -> workgroup r mod G, slot r div G)
                // (plan_sort = 3: the same over the WAVES - rank r -> wave r mod W: every wave gets the same sum of sizes, what counts if the
                //  lanes of a wave serialise)
                const int64_t per = h->plan_sort == 3 ? (int64_t)plan_lanes_now(h) : (int64_t)plan_lanes_now(h) * h->plan_waves, G = (m + per - 1) / per;
                std::vector<int32_t> rank_of((size_t)m);
                for (int64_t i = 0; i < m; ++i) rank_of[(size_t)count[key[(size_t)i]]++] = (int32_t)i;  // rank -> request
                std::vector<int64_t> fill((size_t)G, 0);
                // (the last workgroup may be short: a slot beyond the slice is skipped by giving the rank to the next workgroup with room)
                int64_t g = 0;
                for (int64_t r = 0; r < m; ++r) {
                    for (int tries = 0; tries < G; ++tries, g = (g + 1) % G) {
                        const int64_t pos = g * per + fill[(size_t)g];
                        if (fill[(size_t)g] < per && pos < m) { perm[s0 + pos] = rank_of[(size_t)r]; ++fill[(size_t)g]; g = (g + 1) % G; break; }
                    }
                }
            }
        }
    }
    // No copies on the planning stream: its kernels read the request arrays from, and write their results to, pinned host
    // memory.  (A DMA copy on this stream queued behind the previous call's result download - which waits for that call's
    // kernels - on the copy engine: the planner started when the chunk in flight had finished, never beside it.)
    HIP_TRY(h, hipEventRecord(h->emit_ev[0], P));
    const char *d = pin;
    EmitMeta *meta_out = reinterpret_cast<EmitMeta *>(h->emit_out.p);
    Tag *tags_out = reinterpret_cast<Tag *>(h->emit_out.p + (size_t)n * sizeof(EmitMeta) + 64);
    if (wave) {
        WavePlanArgs A;
        A.net = h->d_wnet;
        A.anc = h->order_net_dev.anc;
        A.q_off = reinterpret_cast<const int64_t *>(d);
        A.e_off = reinterpret_cast<const int64_t *>(d + off_bytes);
        A.out_off = reinterpret_cast<const int64_t *>(d + 2 * off_bytes);
        A.q_vars = reinterpret_cast<const int32_t *>(d + 3 * off_bytes);
        A.e_vars = A.q_vars + nq;
        A.e_codes = A.e_vars + ne;
        A.skip = reinterpret_cast<const char *>(A.e_codes + ne);
        A.B = n;
        A.flags = flags;
        A.prog = st.d_prog;
        A.prog_stride = (uint32_t)stride;
        A.meta = meta_out;
        A.tags = tags_out;
        A.tag_cursor = h->d_emit_cursor;
        A.tag_cap = (uint32_t)tag_cap;
        A.perm = nullptr;
        if (h->wave_sort && n > 4 * kWaveWG * (int64_t)h->n_cu) {  // (fewer requests than waves in flight: nothing to order)
            if ((rc = ensure(h, h->d_plan_perm, h->plan_perm_cap, 2 * (size_t)n))) return rc;
            A.perm = h->d_plan_perm;
        }
        hipLaunchKernelGGL(reset_cursor_kernel, dim3(1), dim3(1), 0, P, h->d_emit_cursor);
        if (A.perm) hipLaunchKernelGGL(plan_sort_kernel, dim3(1), dim3(kPlanSortThreads), 0, P, A);
        {
            // (the waves draw requests from a counter: the grid is what the chip holds at once - MIBN_WAVE_MIN_WGS workgroups per CU)
            int64_t grid = std::min<int64_t>((n + kWaveWG - 1) / kWaveWG, (int64_t)h->n_cu * MIBN_WAVE_MIN_WGS);
            if (h->wave_wgs > 0) grid = std::min<int64_t>(grid, h->wave_wgs);
            hipLaunchKernelGGL(wave_plan_kernel, dim3((unsigned)grid), dim3(64 * kWaveWG), 0, P, A);
        }
        HIP_TRY(h, hipGetLastError());
    }
    for (int64_t s0 = 0; !wave && s0 < n; s0 += kPlanSlice) {
        const int64_t m = std::min(kPlanSlice, n - s0);
        OrderArgs O;
        O.net = h->order_net_dev;
        O.net.prune = h->net.prune;
        O.net.minfill_above = h->net.minfill_above;
        {   // (options set after the network: the model weights follow them like on the host)
            const OrderNet hv = h->net.order_view();
            O.net.chain_weight = hv.chain_weight;
            O.net.big_cells = hv.big_cells;
            O.net.effort = hv.effort;
        }
        O.q_off = reinterpret_cast<const int64_t *>(d) + s0;
        O.e_off = reinterpret_cast<const int64_t *>(d + off_bytes) + s0;
        O.q_vars = reinterpret_cast<const int32_t *>(d + 3 * off_bytes);
        O.e_vars = O.q_vars + nq;
        O.B = m;
        O.flags = flags;
        O.orders = h->d_orders + s0 * 128;
        O.order_len = h->d_order_len + s0;
        O.scratch = h->d_order_scratch;
        O.zero = s0 == 0 ? h->d_emit_cursor : nullptr;
        O.perm = sorted ? reinterpret_cast<const int32_t *>(d + perm_off) + s0 : nullptr;
        O.lanes = plan_lanes_now(h);
        const int64_t per_wg = (int64_t)O.lanes * h->plan_waves;
        hipLaunchKernelGGL(order_kernel, dim3((unsigned)((m + per_wg - 1) / per_wg)), dim3(64 * h->plan_waves), 0, P, O);
        EmitArgs A;
        {   // the pointers of the device copy, the options of the moment
            A.net = h->net.emit_view();
            const EmitNet &dv = h->emit_net_dev;
            A.net.card = dv.card; A.net.log2card = dv.log2card; A.net.pool_off = dv.pool_off; A.net.scope_off = dv.scope_off;
            A.net.scope_vars = dv.scope_vars; A.net.scope_stride = dv.scope_stride; A.net.anc = dv.anc;
        }
        A.q_off = O.q_off;
        A.e_off = O.e_off;
        A.out_off = reinterpret_cast<const int64_t *>(d + 2 * off_bytes) + s0;
        A.q_vars = O.q_vars;
        A.e_vars = O.e_vars;
        A.e_codes = A.e_vars + ne;
        A.skip = reinterpret_cast<const char *>(A.e_codes + ne) + s0;
        A.orders = O.orders;
        A.order_len = O.order_len;
        A.B = m;
        A.flags = flags;
        A.prog = st.d_prog + (size_t)s0 * stride;
        A.prog_stride = (uint32_t)stride;
        A.meta = meta_out + s0;
        A.tags = tags_out;
        A.tag_cursor = h->d_emit_cursor;
        A.tag_cap = (uint32_t)tag_cap;
        A.scratch = h->d_emit_scratch;
        A.scratch_stride = scratch_stride;
        A.perm = O.perm;
        A.lanes = plan_lanes_now(h);
        hipLaunchKernelGGL(emit_kernel, dim3((unsigned)((m + per_wg - 1) / per_wg)), dim3(64 * h->plan_waves), 0, P, A);
        HIP_TRY(h, hipGetLastError());
    }
    HIP_TRY(h, hipEventRecord(h->emit_ev[1], P));
    return MIBN_OK;
}

// Waits for the device planner and fills `ck` for the n requests it planned, like plan_batch would (one "worker": the device).
// *dev_ms = duration of the kernels.  Returns MIBN_OK, an error, or 1 (see above).
#ifndef MIBN_HOST_KEEPS_UP
#define MIBN_HOST_KEEPS_UP 1.25  // adaptive policy: (requests the host plans per ms) x (kernel ms per request) at or above which the device planner is dropped
#endif
#ifndef MIBN_POLICY_WINDOW_MS
#define MIBN_POLICY_WINDOW_MS 150.0  // adaptive policy: planning and retired kernel time a window must hold before it is judged (ranks of more than four planning threads)
#endif
#ifndef MIBN_HOST_BOUND_RATIO
#define MIBN_HOST_BOUND_RATIO 1.15  // adaptive policy: planner wall time over GPU kernel time above which a stream of calls counts as host-bound
#endif
int plan_on_device_collect(mibn_ctx *h, int64_t b0, int64_t n, BatchPlan &ck, double *dev_ms, std::vector<int64_t> *beyond_list) {
    hipStream_t P = h->search_stream;
    const size_t stride = h->emit_words;
    const size_t tag_cap = (size_t)n * 64;
    const EmitMeta *meta = reinterpret_cast<const EmitMeta *>(h->emit_out.p);
    const Tag *tags = reinterpret_cast<const Tag *>(h->emit_out.p + (size_t)n * sizeof(EmitMeta) + 64);
    HIP_TRY(h, hipStreamSynchronize(P));
    {
        float ms = 0;
        HIP_TRY(h, hipEventElapsedTime(&ms, h->emit_ev[0], h->emit_ev[1]));
        if (dev_ms) *dev_ms = ms;
    }
    size_t n_tags = 0;  // (a request's items are contiguous, the requests' ranges in the order the lanes finished)
    for (int64_t i = 0; i < n; ++i)
        if (!meta[i].err) n_tags = std::max<size_t>(n_tags, (size_t)meta[i].tag_first + meta[i].n_tags);
    n_tags = std::min(n_tags, tag_cap);
    // the BatchPlan plan_batch would have filled (one "worker": the device)
    ck.st = PlanStats{};
    ck.arena_cells = 0;
    ck.err.clear();
    ck.prog_off.assign(n, 0);
    ck.cost.assign(n, 0.0);
    ck.arena_need.assign(n, 0);
    ck.thread_of.assign(n, 0);
    ck.local_off.assign(n, 0);
    ck.thread_words.clear();  // (nothing to upload: the programs are on the device)
    ck.tag_first.assign(n, 0);
    ck.tag_count.assign(n, 0);
    ck.tags.resize(1);
    ck.tags[0].assign(tags, tags + n_tags);
    bool refused = false, beyond = false;
    for (int64_t i = 0; i < n; ++i) {
        const EmitMeta &m = meta[i];
        if (m.err == kEmitErrWords) { refused = true; continue; }
        if (m.err == kEmitErrDevice) {  // beyond what wave_plan_kernel covers: the host plans this request (the chunk, if they are many)
            if (beyond_list && beyond_list->size() < 256) { beyond_list->push_back(i); continue; }
            refused = true; beyond = true; continue;
        }
        if (m.err) { h->err = "request " + std::to_string(b0 + i) + ": " + emit_error_message(m.err); return MIBN_E_LIMIT; }
        ck.prog_off[i] = (uint64_t)i * stride + m.prog_first;
        ck.local_off[i] = ck.prog_off[i];
        ck.tag_first[i] = m.tag_first;
        ck.tag_count[i] = m.n_tags;
        ck.cost[i] = m.alg_bytes;
        ck.arena_need[i] = m.arena_cells;
        ck.st.alg_bytes += m.alg_bytes;
        ck.st.alg_flops += m.alg_flops;
        ck.st.n_steps += m.n_steps;
        ck.st.max_step_cells = std::max(ck.st.max_step_cells, m.max_step_cells);
        ck.arena_cells = std::max(ck.arena_cells, m.arena_cells);
    }
    ck.total_words = (size_t)n * stride;
    if (refused) {
        if (h->trace || h->emit_fallbacks < 4) {  // (a rare event worth a line: the first few are always reported)
            int64_t nw = 0, nb = 0, first = -1;
            for (int64_t i = 0; i < n; ++i) { nw += meta[i].err == kEmitErrWords; nb += meta[i].err == kEmitErrDevice; if (meta[i].err && first < 0) first = i; }
            std::fprintf(stderr, "[mibn plan] device planner: %lld requests beyond a device limit, %lld did not fit their slot / the item buffer (first: request %lld, %u words, %u items; %zu items of %zu in the buffer): the host plans the chunk\n",
                         (long long)nb, (long long)nw, (long long)(b0 + first), first >= 0 ? meta[first].words : 0u, first >= 0 ? meta[first].n_tags : 0u, n_tags, tag_cap);
        }
        if (!beyond) h->emit_words = std::min<uint32_t>(h->emit_words * 2, 1u << 20);
        ++h->emit_fallbacks;
        return 1;
    }
    ++h->emit_chunks;
    return MIBN_OK;
}

// Small-network path (tiny_kernel.hip.h): the request arrays go to the device as they are, one lane answers one request,
// malformed requests are detected by the kernel (the host then re-validates to build the reference's message).
// Returns MIBN_OK / an error, or 1 when the batch does not fit the kernel and has to be planned.
constexpr int64_t kTinyZeroCopyRequests = 64;  // one wave

int run_tiny(mibn_t *h, uint32_t flags, int64_t B, const int64_t *q_off, const int32_t *q_vars, const int64_t *e_off,
             const int32_t *e_vars, const int32_t *e_codes, const int64_t *out_off, double *out, double t_start) {
    const size_t nq = (size_t)(q_off[B] - q_off[0]), ne = (size_t)(e_off[B] - e_off[0]);
    const size_t res_cells = (size_t)(out_off[B] - out_off[0]);
    if (q_off[0] != 0 || e_off[0] != 0) return 1;                    // (offsets relative to a larger array: plan)
    if (res_cells > (size_t)B * kTinyMaxQCells) return 1;
    int rc;
    // one pinned staging buffer: [q_off | e_off | out_off | q_vars | e_vars | e_codes | bad | results (zero-copy calls only)]
    const size_t off_bytes = (size_t)(B + 1) * 8;
    const size_t req_bytes = (3 * off_bytes + (nq + 2 * ne) * 4 + 15) & ~(size_t)15;
    // A call of at most one wave of requests (a single query(): BASELINE config 1) is latency, not bandwidth: the kernel reads the
    // request arrays from - and writes the posteriors and the bad-request flag to - the pinned, device-mapped staging buffer
    // itself.  No H2D / D2H copy, no timing events: one launch and one stream synchronisation (round 4 spent two DMAs each way
    // and two event records around a kernel of a few microseconds).  stats.kernel_ms is 0 for such a call.
    const bool zero_copy = B <= kTinyZeroCopyRequests && h->tiny_zero_copy;
    const size_t bad_off = req_bytes, res_off = req_bytes + 16;
    const size_t bytes = res_off + (zero_copy ? res_cells * 8 : 0) + 64;
    mibn_ctx::Staging &sg = h->tiny_stage;
    if (bytes > sg.cap) {
        if (sg.p) { HIP_TRY(h, hipHostFree(sg.p)); sg.p = nullptr; sg.cap = 0; }
        HIP_TRY(h, hipHostMalloc((void **)&sg.p, bytes + bytes / 4, hipHostMallocDefault));
        sg.cap = bytes + bytes / 4;
    }
    if (!zero_copy) {
        if ((rc = ensure(h, h->d_tiny_req, h->tiny_req_cap, bytes))) return rc;
        if ((rc = ensure(h, h->d_results[0], h->results_cap[0], res_cells))) return rc;
    }
    char *p = sg.p;
    std::memcpy(p, q_off, off_bytes);
    std::memcpy(p + off_bytes, e_off, off_bytes);
    std::memcpy(p + 2 * off_bytes, out_off, off_bytes);
    char *pv = p + 3 * off_bytes;
    std::memcpy(pv, q_vars, nq * 4);
    if (ne) { std::memcpy(pv + nq * 4, e_vars, ne * 4); std::memcpy(pv + (nq + ne) * 4, e_codes, ne * 4); }
    const int32_t none = 0x7fffffff;
    std::memcpy(p + bad_off, &none, 4);
    char *dbase = h->d_tiny_req;  // where the kernel finds the arrays
    if (zero_copy) {
        void *dp = nullptr;
        HIP_TRY(h, hipHostGetDevicePointer(&dp, sg.p, 0));
        dbase = static_cast<char *>(dp);
        std::memset(p + res_off, 0, res_cells * 8);
    } else {
        HIP_TRY(h, hipMemcpyAsync(h->d_tiny_req, p, req_bytes + 16, hipMemcpyHostToDevice, h->stream));
    }
    TinyArgs A;
    A.pool = h->d_pool;
    A.meta = h->d_tiny_meta;
    A.q_off = reinterpret_cast<const int64_t *>(dbase);
    A.e_off = reinterpret_cast<const int64_t *>(dbase + off_bytes);
    A.out_off = reinterpret_cast<const int64_t *>(dbase + 2 * off_bytes);
    A.q_vars = reinterpret_cast<const int32_t *>(dbase + 3 * off_bytes);
    A.e_vars = A.q_vars + nq;
    A.e_codes = A.e_vars + ne;
    A.out = zero_copy ? reinterpret_cast<double *>(dbase + res_off) : h->d_results[0];
    A.bad = reinterpret_cast<int32_t *>(dbase + bad_off);
    A.B = B;
    A.n_vars = h->net.n_vars;
    A.pool_cells = (int32_t)h->net.pool.size();
    A.meta_words = h->tiny_meta_words;
    A.flags = (flags & ~kTinyFlagHostBad) | (zero_copy ? kTinyFlagHostBad : 0u);
    const size_t lds = tiny_lds_bytes(A.pool_cells, A.meta_words);
    const unsigned grid = (unsigned)std::max<int64_t>(1, std::min<int64_t>((B + 63) / 64, (int64_t)h->n_cu * 16));
    mibn_ctx::Set &st = h->set[0];
    if ((rc = retire(h, st))) return rc;
    size_t e0 = 0, e1 = 0;
    if (!zero_copy && (rc = next_event(h, st, e0))) return rc;
    hipLaunchKernelGGL(tiny_kernel, dim3(grid), dim3(64), lds, h->stream, A);
    HIP_TRY(h, hipGetLastError());
    if (!zero_copy && (rc = next_event(h, st, e1))) return rc;
    int32_t bad = none;
    const double t_d2h = now_ms();
    float ms = 0;
    if (zero_copy) {
        HIP_TRY(h, hipStreamSynchronize(h->stream));
        std::memcpy(&bad, p + bad_off, 4);
        std::memcpy(out + out_off[0], p + res_off, res_cells * 8);
    } else {
        HIP_TRY(h, hipMemcpyAsync(p + bad_off, h->d_tiny_req + bad_off, 4, hipMemcpyDeviceToHost, h->stream));
        HIP_TRY(h, hipMemcpyAsync(out + out_off[0], h->d_results[0], res_cells * 8, hipMemcpyDeviceToHost, h->stream));
        HIP_TRY(h, hipStreamSynchronize(h->stream));
        std::memcpy(&bad, p + bad_off, 4);
        HIP_TRY(h, hipEventElapsedTime(&ms, st.ev[e0], st.ev[e1]));
    }
    h->stats.d2h_ms += now_ms() - t_d2h;
    st.ev_used = 0;
    if (bad != none) {  // the kernel skipped a malformed request: the message comes from the host-side checks
        Request rq;
        const int64_t b = bad;
        rq.nq = (int32_t)(q_off[b + 1] - q_off[b]);
        rq.qvars = q_vars + q_off[b];
        rq.ne = (int32_t)(e_off[b + 1] - e_off[b]);
        rq.evars = e_vars + e_off[b];
        std::string why = validate_request(h->net, rq);
        if (why.empty()) why = "out_off does not match the query table size";
        h->err = "request " + std::to_string(b) + ": " + why;
        return MIBN_E_ARG;
    }
    const double io_bytes = (double)bytes + 8.0 * (double)res_cells;
    h->stats.kernel_ms = ms;
    h->stats.n_launches = 1;
    h->stats.n_workgroups = grid;
    h->stats.alg_bytes = io_bytes;  // (request arrays in + posteriors out: nothing else touches HBM)
    h->stats.total_ms = now_ms() - t_start;
    h->total.kernel_ms += ms;
    h->total.n_launches += 1;
    h->total.alg_bytes += io_bytes;
    h->total.total_ms += h->stats.total_ms;
    h->total.d2h_ms += h->stats.d2h_ms;
    for (mibn_kernel_stat *ks : {&h->kstats[kNumKernels + 1], &h->ktotal[kNumKernels + 1]}) {
        std::snprintf(ks->name, sizeof(ks->name), "%s", "tiny_kernel");
        ks->launches += 1;
        ks->ms += ms;
        ks->alg_bytes += io_bytes;
        ks->items += grid;
    }
    return MIBN_OK;
}

// Plan, upload and launch a batch.  Synchronous (ticket == nullptr): waits and writes `out`.  Asynchronous: the
// results land in a pinned buffer, *ticket identifies the call for mibn_wait; up to two calls may be in flight, so
// the host plans call s+1 while the GPU still runs call s.
int run_batch_body(mibn_t *h, uint32_t flags, int64_t B, const int64_t *q_off, const int32_t *q_vars, const int64_t *e_off, const int32_t *e_vars,
              const int32_t *e_codes, const int64_t *out_off, double *out, int32_t *ticket) {
    if (!h || B < 0 || !q_off || !e_off || !out_off || (B && !out)) return MIBN_E_ARG;
    if (h->planner_only) { h->err = "planner-only context: no HIP device bound (there is no CPU fallback)"; return MIBN_E_NODEVICE; }
    if (!h->has_net) { h->err = "set_network first"; return MIBN_E_STATE; }
    const double t_start = now_ms();
    ++h->call_id;
    h->search_ms = 0;
    h->emit_ms = 0;
    h->stats = mibn_stats{};
    for (int k = 0; k <= kNumKernels + 6; ++k) {
        h->kstats[k] = mibn_kernel_stat{};
        std::snprintf(h->kstats[k].name, sizeof(h->kstats[k].name), "%s", stat_name(k));
    }
    const int slot = h->next_slot;
    if (h->pend[slot].active) { h->err = "two asynchronous calls are already in flight: mibn_wait the older one first"; return MIBN_E_STATE; }
    if (ticket) *ticket = slot;
    if (B == 0) return MIBN_OK;
    HIP_TRY(h, hipSetDevice(h->device));
    if (h->tiny && h->tiny_ok && !ticket && B < 0x7ffffff0 && out_off[B] - out_off[0] >= B) {
        const int rc_tiny = run_tiny(h, flags | (h->net.prune ? 0u : (uint32_t)MIBN_Q_NOPRUNE), B, q_off, q_vars, e_off, e_vars, e_codes, out_off, out, t_start);
        if (rc_tiny != 1) return rc_tiny;  // 1: a request does not fit the kernel (query table too large) - plan it
    }
    // validation (bayes_net.py:840-845) and the out-of-domain-evidence short cut
    std::vector<char> skip((size_t)B, 0);
    for (int64_t b = 0; b < B; ++b) {
        Request rq;
        rq.nq = (int32_t)(q_off[b + 1] - q_off[b]);
        rq.qvars = q_vars + q_off[b];
        rq.ne = (int32_t)(e_off[b + 1] - e_off[b]);
        rq.evars = e_vars + e_off[b];
        if (!request_is_valid(h->net, rq)) { h->err = "request " + std::to_string(b) + ": " + validate_request(h->net, rq); return MIBN_E_ARG; }
        int64_t cells = 1;
        for (int i = 0; i < rq.nq; ++i) cells *= h->net.card[rq.qvars[i]];
        if (out_off[b + 1] - out_off[b] != cells) { h->err = "request " + std::to_string(b) + ": out_off does not match the query table size"; return MIBN_E_ARG; }
        for (int i = 0; i < rq.ne; ++i) {
            int32_t c = e_codes[e_off[b] + i];
            if (c < 0 || c >= h->net.card[rq.evars[i]]) skip[b] = 1;  // label outside the domain -> empty posterior
        }
    }
    ensure_pool(h);
    if (h->trace) std::fprintf(stderr, "[mibn plan] validation of %lld requests %.2f ms\n", (long long)B, now_ms() - t_start);
    double call_fixed_ms = now_ms() - t_start;  // the host's side of this call besides planning (the share rule of wave_plan_kernel)
    double call_plan_ms = 0;  // the workers' planning of the host's shares and the waits for the device planner: what the call's time to the last launch is NOT fixed cost
    if (h->adaptive && !h->adaptive_seeded) {
        // A rank with a handful of planning threads (8 ranks on a 16-CPU quota: 2-4 each) cannot plan a stream like C3 at the rate
        // its GPU executes it (67 / 133 k queries/s at 2 / 4 threads against 280 k): it starts with the device planner instead of
        // finding that out over several host-bound calls; the share controller gives the planning back where the host keeps up.
        h->adaptive_seeded = true;
        if (h->pool->size() <= 4 && h->order_net_ok && h->emit_net_ok && !h->gpu_emit) { h->gpu_emit = 1; h->auto_emit = true; }
    }
    if (h->adaptive) {
        // over the calls since the last adjustment: host planning wall time against GPU kernel time (retired launches)
        const double dp = h->total.plan_ms - h->seen_plan_ms, dk = h->total.kernel_ms - h->seen_kernel_ms;
        if (h->call_id <= 2) {  // the first calls pay one-time costs (thread pool, pinned buffers, first kernel load): not a trend
            h->seen_plan_ms = h->total.plan_ms;
            h->seen_kernel_ms = h->total.kernel_ms;
            h->seen_requests = h->retired_requests;
        } else if (dp > 20.0 && dk > 20.0 && (h->pool->size() <= 4 || (dp > MIBN_POLICY_WINDOW_MS && dk > MIBN_POLICY_WINDOW_MS))) {
            // (a rank that is not starved judges windows of at least MIBN_POLICY_WINDOW_MS of planning AND of retired kernel time - two
            //  to three calls: the kernel time of a call is booked when its launches retire, up to two calls late, and a window of one
            //  call saw "100 ms of planning, 136 ms of kernels" as often as "100 against 68" on a steadily host-bound stream - n_evidence
            //  = 16 in round 5's session D: the two host-bound windows in a row the switch asks for came once in twelve calls)
            const double dreq = h->retired_requests - h->seen_requests;
            if (dreq > 0) h->kernel_ms_per_req = h->kernel_ms_per_req > 0 ? 0.5 * h->kernel_ms_per_req + 0.5 * dk / dreq : dk / dreq;
            // The device planner goes again where the host ALONE would keep up: the requests its workers plan per ms (measured
            // beside the device planner) x the kernel time per request must cover a request with a margin - the margin also
            // absorbs that the kernels of device-planned chunks run ~ 17 % longer than they would without the planner's kernels.
            // (Up to session S the rule was "the device's share has fallen to 0.3": a 6-thread rank - share 0.33 - oscillated
            // around it, 219 k queries/s; a fixed lower bar, 0.2, kept the device planner on a full-quota rank that had
            // switched it on during its first calls: 278 k instead of 300 k.  profiles/r04_s_policy.log, r04_t_policy.log)
            if (h->auto_emit && h->host_rate > 0 && dreq > 0 && h->host_rate * (dk / dreq) >= MIBN_HOST_KEEPS_UP) {
                h->gpu_emit = 0;
                h->auto_emit = false;
                h->host_bound_streak = 0;
            } else if (dp > MIBN_HOST_BOUND_RATIO * dk && ++h->host_bound_streak >= 2) {  // (round 6: twice in a row for every rank - a full-quota rank used to switch on one window, and the window behind a step's barrier, with the first call's planning exposed, tripped it for a per mille of the requests: profiles/r06_ce_ab.log, r06_cf_ab.log)  // (twice in a row: the kernel time of a call is booked when its
                                                                          // launches retire, up to two calls late - one window can mislead)
                // host-bound: first hand the order search to the device (same orders, no more bytes); networks it does
                // not cover give up the min-fill search for ever more expensive requests instead
                if (h->order_net_ok && h->emit_net_ok && !h->gpu_emit) {  // the whole planning, not only the search
                    h->gpu_emit = 1;
                    h->auto_emit = true;
                    h->host_rate = 0;  // (measured afresh beside the device planner: a rate left over from another workload - the line's
                                       //  n_evidence = 1 variant plans 1 000 requests per ms, n_evidence = 16 500 - made "the host alone would
                                       //  keep up" drop the device planner one window after every switch: round 5's session ZZ)
                }
                else if (h->order_net_ok && !h->emit_net_ok && !h->gpu_search) { h->gpu_search = 1; h->auto_search = true; }
                else if (!h->order_net_ok) h->net.minfill_above = std::min(h->net.minfill_above * 8.0, 1e18);
            } else if (dp <= MIBN_HOST_BOUND_RATIO * dk) {
                h->host_bound_streak = 0;
            }
            if (dp < 0.3 * dk) {
                if (h->net.minfill_above > h->base_minfill) h->net.minfill_above = std::max(h->net.minfill_above / 8.0, h->base_minfill);
                else if (h->auto_search) { h->gpu_search = 0; h->auto_search = false; }
            }
            h->seen_plan_ms = h->total.plan_ms;
            h->seen_kernel_ms = h->total.kernel_ms;
            h->seen_requests = h->retired_requests;
        }
    }
    int rc;
    const size_t res_cells = (size_t)(out_off[B] - out_off[0]);
    if ((rc = ensure(h, h->d_results[slot], h->results_cap[slot], res_cells))) return rc;
    double *const d_results = h->d_results[slot];
    const int n_lanes = h->n_streams > 1 ? 2 : 1;
    {
        bool idle = true;
        for (auto &st : h->set) idle = idle && !st.busy;
        if (!h->epoch) { HIP_TRY(h, hipEventCreate(&h->epoch)); idle = true; }
        if (idle) {  // (times since the epoch stay small: float milliseconds)
            HIP_TRY(h, hipEventRecord(h->epoch, h->stream));
            h->busy_until = 0;
        }
    }
    HIP_TRY(h, hipMemsetAsync(d_results, 0, res_cells * 8, h->stream));
    if (n_lanes > 1) {  // lane 1 starts this call's work after the results buffer is zeroed
        if (!h->zero_ev) HIP_TRY(h, hipEventCreateWithFlags(&h->zero_ev, hipEventDisableTiming));
        HIP_TRY(h, hipEventRecord(h->zero_ev, h->stream));
        HIP_TRY(h, hipStreamWaitEvent(h->stream2, h->zero_ev, 0));
    }
    size_t free_b = 0, total_b = 0;
    HIP_TRY(h, hipMemGetInfo(&free_b, &total_b));
    const int64_t budget_cells =
        (int64_t)(std::min(h->arena_gb * 1e9, 0.8 * (double)(free_b + h->arena_bytes[0] + h->arena_bytes[1])) / 8.0) / n_lanes;
    bool lane1_used = false;
    int64_t n_chunks = 0;
    // order_effort >= 1 (more candidate orders, the byte model's best two both emitted): the wave planner has it, order_kernel / emit_kernel do
    // not - where the wave planner does not cover the network the host plans (never two kinds of plans in one stream)
    bool effort_ok = true;
    if (h->net.order_effort >= 1 && (h->gpu_emit || h->gpu_search)) {
        std::unique_ptr<WNet> probe(new WNet);
        effort_ok = h->wave_plan && h->net.wave_view(*probe);
    }
    const bool emit_on = h->gpu_emit && h->order_net_ok && h->emit_net_ok && effort_ok;  // whole chunks planned on the device
    // A rank that needs the device planner has no planning time to spare - and on the device the second emission costs more GPU time than
    // the bytes it saves give back (a two-thread rank: 243 against 273 k queries/s, profiles/r06_bh_ab.log): its calls take the extra
    // candidates only, on the host's share and on the device's alike.  (Like minfill_above below: the search effort follows the load.)
    h->net.second_above = (emit_on && !h->second_on_device) ? 1e300 : h->base_second_above;
    const bool search_on = !emit_on && h->gpu_search && h->order_net_ok && h->net.order_effort < 1;
    int64_t search_b0 = -1;
    bool search_done = false;
    // a short first chunk gets an idle GPU going while the host plans the first full-size one (first_chunk = 2: also when
    // an earlier asynchronous call still keeps the GPU busy; 0: never)
    bool gpu_busy = false;
    for (auto &st : h->set) gpu_busy = gpu_busy || (st.busy && st.ev_used && hipEventQuery(st.ev[st.ev_used - 1]) == hipErrorNotReady);
    (void)hipGetLastError();  // (hipErrorNotReady is an answer, not an error: do not leave it for the check after the launches)
    const bool short_first = h->first_chunk == 2 || (h->first_chunk == 1 && !gpu_busy);
    for (int64_t b0 = 0, b1 = 0; b0 < B; b0 = b1, ++n_chunks) {
        // A call that finds the GPU idle (the first of a stream of calls, a blocking call) starts with a short chunk - a quarter -
        // that gets the GPU going while the host plans the rest; in steady state - the previous call still running - a call
        // is cut into full chunks.  (Ramping the chunks up from 1 024 requests, doubling, was measured and dropped: seven small
        // plan_batch calls cost 95 ms of host time instead of 63, and the chunk sets - their pinned buffers sized by the small
        // chunks they had seen - grew under the clock: profiles/r03_i_pipeline_start.log.)
        // With the device order search on, every call starts with an eighth of a chunk whose orders the host searches while
        // ONE launch searches the rest of the call (a call of a single chunk would otherwise be searched on the host whole).
        const bool dev_split = search_on && h->gpu_search == 1;
        int64_t size = h->chunk;
        if (b0 == 0 && dev_split) size = std::max<int64_t>(1024, h->chunk / 8);
        else if (emit_on) size = h->chunk;  // (the host has nothing to plan while a short first chunk runs)
        else if (b0 == 0 && short_first && (B > h->chunk || !gpu_busy)) size = std::max<int64_t>(1024, h->chunk / 4);
        b1 = std::min(B, b0 + size);
        const int64_t n = b1 - b0;
        mibn_ctx::Set &st = h->set[h->set_cursor];
        const int lane = n_lanes > 1 ? (h->set_cursor & 1) : 0;  // consecutive chunks alternate between the lanes
        hipStream_t S = lane ? h->stream2 : h->stream;
        lane1_used = lane1_used || lane == 1;
        h->set_cursor = (h->set_cursor + 1) % h->n_sets;
        if ((rc = retire(h, st))) return rc;  // its buffers are about to be rewritten
        double t0 = now_ms();
        // device order search: the first (short) chunk of a call is searched on the host while one launch searches the
        // rest of the call on the device; the later chunks only emit
        const uint8_t *orders = nullptr;
        const int32_t *order_len = nullptr;
        if (search_on && h->gpu_search == 2) {  // test mode: every chunk searched on the device, synchronously
            if ((rc = search_orders_async(h, flags, b0, b1, q_off, q_vars, e_off, e_vars))) return rc;
            if ((rc = search_wait(h))) return rc;
            orders = reinterpret_cast<const uint8_t *>(h->search_out.p);
            order_len = reinterpret_cast<const int32_t *>(h->search_out.p + (size_t)n * 128);
        } else if (search_on) {
            if (n_chunks == 0 && b1 < B) {
                if ((rc = search_orders_async(h, flags, b1, B, q_off, q_vars, e_off, e_vars))) return rc;
                search_b0 = b1;
            } else if (search_b0 >= 0) {
                if (!search_done) { if ((rc = search_wait(h))) return rc; search_done = true; h->search_ms += now_ms() - t0; }
                orders = reinterpret_cast<const uint8_t *>(h->search_out.p) + (size_t)(b0 - search_b0) * 128;
                order_len = reinterpret_cast<const int32_t *>(h->search_out.p + (size_t)(B - search_b0) * 128) + (b0 - search_b0);
            }
        }
        BatchPlan &ck = st.plan;
        // every error exit of the device-planner path: the planner's kernels (they read pinned request arrays and write pinned
        // results) and the chunks this call has already launched are drained before the caller sees the error - ADVICE r3
        auto bail = [&](int code) {
            (void)hipStreamSynchronize(h->search_stream);
            (void)hipStreamSynchronize(h->stream);
            (void)hipStreamSynchronize(h->stream2);
            for (auto &la : h->aux)
                for (hipStream_t a : la) (void)hipStreamSynchronize(a);
            return code;
        };
        bool on_device = false;
        std::vector<int64_t> beyond;  // requests of the device's part that exceed a device limit (kEmitErrDevice): the host plans them
        int64_t nd = 0;          // requests [b0, b0 + nd) planned by the device, the rest by the host's workers meanwhile
        size_t prog_base = 0;    // words of st.d_prog the device has written (the host's programs follow)
        if (emit_on) {
            nd = h->gpu_emit == 2 ? n : std::min<int64_t>(n, std::max<int64_t>(64, (int64_t)((double)n * std::min(h->emit_share, h->emit_share_opt > 0 ? 1.0 : (wave_mode(h) ? 0.99 : 0.95)) + 0.5)));
            if (n - nd < 256) nd = n;  // (without a pinned share the host keeps at least a twentieth: its rate stays measured)
            const size_t stride = h->emit_words;
            if ((rc = plan_on_device_launch(h, flags, b0, b0 + nd, q_off, q_vars, e_off, e_vars, e_codes, out_off, skip.data(), st, (size_t)n * stride))) return bail(rc);
            const double t_launched = now_ms();
            BatchPlan &hp = h->emit_host, &dp = h->emit_dev;
            double host_ms = 0, dev_ms = 0;
            if (nd < n) {
                const double th = now_ms();
                plan_batch(h->net, *h->pool, st.bufs, b0 + nd, b1, q_off, q_vars, e_off, e_vars, e_codes, out_off, skip.data(), hp,
                           (flags & MIBN_Q_NOPRUNE) != 0, nullptr, nullptr, b0);
                host_ms = now_ms() - th;
                call_plan_ms += host_ms;
                if (!hp.err.empty()) { h->err = hp.err; return bail(MIBN_E_LIMIT); }
            }
            const double tw = now_ms();
            rc = plan_on_device_collect(h, b0, nd, nd < n ? dp : ck, &dev_ms, &beyond);
            h->emit_ms += now_ms() - tw;
            call_plan_ms += now_ms() - tw;
            if (h->trace >= 2) std::fprintf(stderr, "[mibn plan] collect returned %d at %.2f ms of the block\n", rc, now_ms() - t0);
            for (mibn_kernel_stat *ks : {&h->kstats[kNumKernels + 3], &h->ktotal[kNumKernels + 3]}) {  // (beside the chunk in flight: their time is not GPU busy time of its own)
                if (!ks->name[0]) std::snprintf(ks->name, sizeof(ks->name), "%s", stat_name(kNumKernels + 3));
                ks->launches += 2;
                ks->ms += dev_ms;
                ks->items += (double)nd;
            }
#if defined(MIBN_EMIT_PROF)
            if (h->trace) {
                unsigned long long hp[12], z[12] = {0};
                HIP_TRY(h, hipMemcpyFromSymbol(hp, HIP_SYMBOL(g_emit_prof), sizeof(hp)));
                HIP_TRY(h, hipMemcpyToSymbol(HIP_SYMBOL(g_emit_prof), z, sizeof(z)));
                static const char *names[11] = {"begin (relevant set, CPT slices)", "key / pos / slot sets", "factors of x", "sweep candidates", "SWEEP 5 / 4", "CHAIN",
                                                "SWEEP 3 / 2", "pair", "single elimination", "final product", "work items"};
                double tot = 0;
                for (int k = 0; k < 11; ++k) tot += (double)hp[k];
                for (int k = 0; k < 11; ++k) std::fprintf(stderr, "[mibn emit phases] %-36s %6.1f us per request  %5.1f %%\n", names[k], (double)hp[k] / 100.0 / (double)nd, 100.0 * (double)hp[k] / tot);
            }
#endif
            if (h->trace) std::fprintf(stderr, "[mibn plan] chunk of %lld: device %lld requests (upload + launch %.2f ms, kernels %.2f ms, wait + collect %.2f ms), host %lld requests %.2f ms\n",
                                       (long long)n, (long long)nd, t_launched - t0, dev_ms, now_ms() - tw, (long long)(n - nd), host_ms);
            if (rc < 0 || rc > 1) return bail(rc);
            if (rc == 0 && nd < n && hp.total_words > (size_t)(n - nd) * stride) rc = 1;  // (the host's programs do not fit behind the device's)
            on_device = rc == 0;
            if (on_device && nd < n) {
                // join the two parts: the device's requests first
                const int T = (int)hp.tags.size();
                ck.st = dp.st;
                ck.st.alg_bytes += hp.st.alg_bytes; ck.st.alg_flops += hp.st.alg_flops; ck.st.n_steps += hp.st.n_steps;
                ck.st.max_step_cells = std::max(dp.st.max_step_cells, hp.st.max_step_cells);
                ck.arena_cells = std::max(dp.arena_cells, hp.arena_cells);
                ck.err.clear();
                ck.tags.resize((size_t)T + 1);
                for (int t = 0; t < T; ++t) ck.tags[(size_t)t].swap(hp.tags[(size_t)t]);
                ck.tags[(size_t)T].swap(dp.tags[0]);
                ck.thread_words = hp.thread_words;
                auto join = [&](auto &dst, const auto &a, const auto &b) { dst.assign(a.begin(), a.end()); dst.insert(dst.end(), b.begin(), b.end()); };
                join(ck.cost, dp.cost, hp.cost);
                join(ck.arena_need, dp.arena_need, hp.arena_need);
                join(ck.tag_first, dp.tag_first, hp.tag_first);
                join(ck.tag_count, dp.tag_count, hp.tag_count);
                join(ck.local_off, dp.local_off, hp.local_off);
                ck.thread_of.assign((size_t)nd, T);
                ck.thread_of.insert(ck.thread_of.end(), hp.thread_of.begin(), hp.thread_of.end());
                prog_base = (size_t)nd * stride;
                ck.prog_off = dp.prog_off;
                for (uint64_t o : hp.prog_off) ck.prog_off.push_back(prog_base + o);
                ck.total_words = prog_base + hp.total_words;
                // the share that would have let both finish together (the device's kernels ran beside the chunk in flight, like they will)
                // The planner's kernels are latency-bound - one request per lane, their duration hardly depends on how many
                // requests they plan - so the host's share is what its workers plan in that time, at the rate just measured.
                // (whole chunks only: the tail of a call - 20 624 requests behind seven chunks of 32 768 in bench.py's 250 000-request
                //  steps - gives the latency-bound device a smaller fraction than it takes of a full chunk; fed into the average,
                //  the tails pushed the share below the switch-off threshold of the policy above and the planning of a 4-thread
                //  rank oscillated between the device and the host alone: 187 k queries/s, profiles/r04_h_threads.log)
                // (host_ms > 0.02, not > 1: a stream answered from plan templates - n_evidence = 1 once its 9 900 shapes are stored - plans
                //  its 13 000 requests in 0.7 ms; with the old bar its rate was never measured, "the host alone would keep up" never
                //  fired and the device planner stayed on at three quarters of every chunk: 430 instead of 550 k queries/s, session N)
                if (host_ms > 0.02) h->host_rate = h->host_rate > 0 ? (wave_mode(h) ? 0.75 : 0.5) * h->host_rate + (wave_mode(h) ? 0.25 : 0.5) * (double)(n - nd) / host_ms : (double)(n - nd) / host_ms;
                if (h->emit_share_opt <= 0 && host_ms > 0.02 && dev_ms > 1.0 && 4 * n >= 3 * h->chunk) {
                    if (wave_mode(h) && h->kernel_ms_per_req > 0) {
                        // wave_plan_kernel's time grows with the requests it is given, and it is GPU time taken from the VE kernels: the
                        // device gets what the host cannot plan while the GPU works through the chunk -
                        //   fixed + (n - nd) ih  <=  0.85 (kv n + kp nd)     ih: ms per request of the host's workers, kv: kernel ms per
                        // request (retired launches, the adaptive policy's windows), kp: planner ms per request (just measured, beside
                        // the kernels), fixed: the host's side of a call besides planning (validation, the schedule).  (The planning
                        // rates alone left a two-thread rank host-bound - 223 against 262 k queries/s; a feedback rule on the policy's
                        // planner-wall / kernel-time ratio counted the waits for the device as host time and ended at a share of 0.99:
                        // profiles/r06_r_ab.log, r06_s_ab.log)
                        // (ih: the smoothed rate and the slower of it and this chunk's - the workers of a rank with the whole CPU quota plan in
                        //  bursts, 35 000 n_evidence = 16 requests in 38 ms one call and 74 ms the next: profiles/r06_x_share16.log)
                        const double ih = std::max(host_ms / (double)(n - nd), h->host_rate > 0 ? 1.0 / h->host_rate : 0.0), kv = h->kernel_ms_per_req, kp = dev_ms / (double)nd;
                        const double fixed = h->fixed_ms_per_req * (double)n;
                        const double target = std::max(0.03, std::min(0.99, ((double)n * ih + fixed - 0.85 * kv * (double)n) / ((ih + 0.85 * kp) * (double)n)));
                        h->emit_share = 0.5 * h->emit_share + 0.5 * target;
                        if (h->trace) std::fprintf(stderr, "[mibn share] host %.4f ms/request (%lld in %.1f ms), kernels %.5f ms/request, planner %.5f ms/request, fixed %.1f ms -> target %.3f, share %.3f\n",
                                                   ih, (long long)(n - nd), host_ms, kv, kp, fixed, target, h->emit_share);
                    } else {
                        const double host_n = (double)(n - nd) / host_ms * dev_ms;
                        h->emit_share = std::max(0.25, std::min(1.0, 0.5 * h->emit_share + 0.5 * (1.0 - host_n / (double)n)));
                    }
                }
            }
        }
        if (on_device && !beyond.empty()) {
            // The requests of the device's part that exceed one of wave_plan_kernel's limits: the host plans THEM (round 6: a refused
            // request used to send its whole chunk to the host - n_evidence = 16 met a limit five times in 200 000 requests and lost a
            // third of its rate) - one plan_batch over the device's range with everything else masked out; a program goes into its
            // request's own slot of the device program buffer, its work items and statistics take the place the kernel left empty.
            std::vector<char> mask(skip.begin(), skip.end());
            for (int64_t i = 0; i < nd; ++i) mask[(size_t)(b0 + i)] = 1;
            for (int64_t i : beyond) mask[(size_t)(b0 + i)] = skip[(size_t)(b0 + i)];
            BatchPlan &xp = st.xplan;
            plan_batch(h->net, *h->pool, st.xbufs, b0, b0 + nd, q_off, q_vars, e_off, e_vars, e_codes, out_off, mask.data(), xp,
                       (flags & MIBN_Q_NOPRUNE) != 0, nullptr, nullptr, b0);
            if (!xp.err.empty()) { h->err = xp.err; return bail(MIBN_E_LIMIT); }
            const size_t stride = h->emit_words;
            const size_t X = ck.tags.size();
            ck.tags.resize(X + xp.tags.size());
            for (size_t t = 0; t < xp.tags.size(); ++t) ck.tags[X + t] = xp.tags[t];
            bool fits = true;
            for (int64_t i : beyond) {
                const uint32_t *w = st.xbufs[(size_t)xp.thread_of[(size_t)i]].data + xp.local_off[(size_t)i];
                size_t words = 1;
                for (uint32_t k = 0; k < w[0]; ++k) words += w[words + 6];
                if (words + kMaxStepWords > stride) { fits = false; break; }
                HIP_TRY(h, hipMemcpyAsync(st.d_prog + (size_t)i * stride, w, words * 4, hipMemcpyHostToDevice, h->copy_stream));
                ck.prog_off[(size_t)i] = (uint64_t)i * stride;
                ck.local_off[(size_t)i] = ck.prog_off[(size_t)i];
                ck.thread_of[(size_t)i] = (int32_t)(X + (size_t)xp.thread_of[(size_t)i]);
                ck.tag_first[(size_t)i] = xp.tag_first[(size_t)i];
                ck.tag_count[(size_t)i] = xp.tag_count[(size_t)i];
                ck.cost[(size_t)i] = xp.cost[(size_t)i];
                ck.arena_need[(size_t)i] = xp.arena_need[(size_t)i];
            }
            ck.st.alg_bytes += xp.st.alg_bytes; ck.st.alg_flops += xp.st.alg_flops; ck.st.n_steps += xp.st.n_steps;
            ck.st.max_step_cells = std::max(ck.st.max_step_cells, xp.st.max_step_cells);
            ck.arena_cells = std::max(ck.arena_cells, xp.arena_cells);
            if (h->trace || h->emit_single < 4)
                std::fprintf(stderr, "[mibn plan] device planner: %zu of %lld requests beyond a device limit (first: request %lld): planned on the host\n", beyond.size(), (long long)nd,
                             (long long)(b0 + beyond[0]));
            h->emit_single += (int64_t)beyond.size();
            if (!fits) on_device = false;  // (a program longer than its slot: the host plans the chunk after all)
        }
        if (!on_device)
            plan_batch(h->net, *h->pool, st.bufs, b0, b1, q_off, q_vars, e_off, e_vars, e_codes, out_off, skip.data(), ck,
                       (flags & MIBN_Q_NOPRUNE) != 0, orders, order_len);
        if (on_device && h->gpu_emit == 2) {
            // test mode: the host plans the chunk too - programs, work items and statistics must agree exactly
            const size_t stride = h->emit_words;
            std::vector<uint32_t> dev((size_t)n * stride);
            HIP_TRY(h, hipStreamSynchronize(h->copy_stream));  // (the programs of requests beyond a device limit, planned on the host above)
            HIP_TRY(h, hipMemcpy(dev.data(), st.d_prog, dev.size() * 4, hipMemcpyDeviceToHost));
            BatchPlan ref;
            plan_batch(h->net, *h->pool, st.bufs, b0, b1, q_off, q_vars, e_off, e_vars, e_codes, out_off, skip.data(), ref,
                       (flags & MIBN_Q_NOPRUNE) != 0, nullptr, nullptr);
            if (!ref.err.empty()) { h->err = ref.err; return bail(MIBN_E_LIMIT); }
            for (int64_t i = 0; i < n; ++i) {
                const uint32_t *hw = st.bufs[ref.thread_of[i]].data + ref.local_off[i], *dw = dev.data() + (size_t)ck.prog_off[(size_t)i];  // (= i x stride + the words in front of the program)
                size_t words = 1;
                for (uint32_t k = 0; k < hw[0]; ++k) words += hw[words + 6];
                for (size_t k = 0; k < words; ++k)
                    if (hw[k] != dw[k]) {
                        size_t off = 1, step = 0;  // (which step holds the word)
                        while (step + 1 < hw[0] && off + hw[off + 6] <= k) { off += hw[off + 6]; ++step; }
                        h->err = "device planner: request " + std::to_string(b0 + i) + " word " + std::to_string(k) + " of " + std::to_string(words) + " (step " + std::to_string(step) + " of " +
                                 std::to_string(hw[0]) + ", kind " + std::to_string(hw[off] & 0xff) + ", word " + std::to_string(k - off) + " of the step's " + std::to_string(hw[off + 6]) + "): host " +
                                 std::to_string(hw[k]) + " device " + std::to_string(dw[k]) + "; device program " + std::to_string(dw[0]) + " steps";
                        return bail(MIBN_E_STATE);
                    }
                const Tag *ht = ref.tags[ref.thread_of[i]].data() + ref.tag_first[i], *dt = ck.tags[(size_t)ck.thread_of[i]].data() + ck.tag_first[i];
                bool same = ref.tag_count[i] == ck.tag_count[i] && ref.arena_need[i] == ck.arena_need[i] && ref.cost[i] == ck.cost[i];
                for (uint32_t k = 0; same && k < ref.tag_count[i]; ++k)
                    same = ht[k].rel_off == dt[k].rel_off && ht[k].a == dt[k].a && ht[k].wgs == dt[k].wgs && ht[k].level == dt[k].level &&
                           ht[k].kid == dt[k].kid && ht[k].bytes == dt[k].bytes;
                if (!same) { h->err = "device planner: work items / statistics of request " + std::to_string(b0 + i) + " differ from the host's"; return bail(MIBN_E_STATE); }
            }
        }
        if (!ck.err.empty()) { h->err = ck.err; (void)hipStreamSynchronize(h->stream); (void)hipStreamSynchronize(h->stream2); return MIBN_E_LIMIT; }
        for (auto &b : st.bufs)
            if (b.cap && !b.data) { h->err = "pinned host allocation failed"; return MIBN_E_HIP; }
        h->stats.plan_ms += now_ms() - t0;
        if (h->trace >= 2) std::fprintf(stderr, "[mibn plan] chunk planning block %.2f ms (on the device: %d)\n", now_ms() - t0, (int)on_device);
        if (!on_device && (rc = ensure(h, st.d_prog, st.prog_cap, ck.total_words + kMaxStepWords))) return rc;  // (slack: segment_wave prefetches whole descriptor slots)
        if ((rc = ensure(h, st.d_prog_off, st.prog_off_cap, (size_t)n))) return rc;
        if ((rc = ensure(h, st.d_arena_off, st.arena_off_cap, (size_t)n))) return rc;
        t0 = now_ms();
        size_t base = prog_base;
        for (size_t t = 0; t < ck.thread_words.size(); ++t) {
            if (ck.thread_words[t])
                HIP_TRY(h, hipMemcpyAsync(st.d_prog + base, st.bufs[t].data, ck.thread_words[t] * 4, hipMemcpyHostToDevice, h->copy_stream));
            base += ck.thread_words[t];
        }
        if ((rc = upload(h, st.stage[0], st.d_prog_off, ck.prog_off.data(), (size_t)n * 8))) return rc;
        h->stats.h2d_ms += now_ms() - t0;
        // waves: consecutive requests whose private arenas fit the scratch budget together
        // (a chunk that does not fit is cut into waves of about EQUAL scratch - ceil(total / budget) of them - not into full
        //  waves and a remainder: round 4 runs chunks of 52 429 requests = 224 GB, close to the budget)
        int64_t chunk_cells = 0;
        for (int64_t r = 0; r < n; ++r) chunk_cells += (ck.arena_need[r] + 15) & ~int64_t(15);
        const int64_t n_waves_min = std::max<int64_t>(1, (chunk_cells + budget_cells - 1) / std::max<int64_t>(1, budget_cells));
        const int64_t wave_target = std::min(budget_cells, chunk_cells / n_waves_min + (chunk_cells / n_waves_min) / 64 + 1);
        for (int64_t r0 = 0; r0 < n;) {
            int64_t r1 = r0, cells = 0;
            while (r1 < n) {
                const int64_t need = (ck.arena_need[r1] + 15) & ~int64_t(15);
                if (r1 > r0 && (cells + need > budget_cells || (n_waves_min > 1 && cells >= wave_target))) break;
                cells += need;
                ++r1;
            }
            if (cells > budget_cells) { h->err = "a request needs " + std::to_string(8.0 * cells / 1e9) + " GB of scratch, above the arena budget"; return MIBN_E_NOMEM; }
            t0 = now_ms();
            Schedule &sc = st.sched;
            build_schedule(h->net, ck, st.bufs, r0, r1, sc);
            h->stats.plan_ms += now_ms() - t0;
            call_fixed_ms += now_ms() - t0;
            if (h->trace) std::fprintf(stderr, "[mibn plan] build_schedule %.2f ms (%zu items, %zu workgroups, %zu launches)\n", now_ms() - t0, sc.items.size(), sc.wg_item.size(), sc.launches.size());
            const size_t need_bytes = (size_t)std::max<int64_t>(16, sc.arena_cells) * sizeof(double);
            if (need_bytes > h->arena_bytes[lane]) {
                // grow with headroom (chunks differ by ~10 %): re-allocating tens of GB costs hundreds of ms
                HIP_TRY(h, hipStreamSynchronize(S));  // earlier launches of the lane still use the old arena
                if (h->d_arena[lane]) { HIP_TRY(h, hipFree(h->d_arena[lane])); h->d_arena[lane] = nullptr; h->arena_bytes[lane] = 0; }
                const size_t want = std::max(need_bytes, std::min((size_t)((double)budget_cells * 8.0), need_bytes + need_bytes / 3));
                HIP_TRY(h, hipMalloc(&h->d_arena[lane], want));
                h->arena_bytes[lane] = want;
            }
            // items / arena offsets of this wave live until the wave's launches have run: one wave per set at a
            // time unless the chunk had to be split (then wait for the previous wave first)
            if (r0 > 0) { HIP_TRY(h, hipStreamSynchronize(S)); HIP_TRY(h, hipStreamSynchronize(h->copy_stream)); }
            if ((rc = ensure(h, st.d_items, st.items_cap, sc.items.size()))) return rc;
            if ((rc = ensure(h, st.d_wg_item, st.wg_item_cap, sc.wg_item.size()))) return rc;
            t0 = now_ms();
            if ((rc = upload(h, st.stage[1], st.d_arena_off, sc.arena_off.data(), (size_t)(r1 - r0) * 8))) return rc;
            if ((rc = upload(h, st.stage[2], st.d_items, sc.items.data(), sc.items.size() * sizeof(Item)))) return rc;
            if ((rc = upload(h, st.stage[3], st.d_wg_item, sc.wg_item.data(), sc.wg_item.size() * sizeof(uint32_t)))) return rc;
            if (!st.uploaded) HIP_TRY(h, hipEventCreateWithFlags(&st.uploaded, hipEventDisableTiming));
            HIP_TRY(h, hipEventRecord(st.uploaded, h->copy_stream));
            HIP_TRY(h, hipStreamWaitEvent(S, st.uploaded, 0));  // the kernels of this wave wait for its uploads only
            h->stats.h2d_ms += now_ms() - t0;
            LevelArgs A;
            A.prog = st.d_prog;
            A.prog_off = st.d_prog_off + r0;
            A.arena_off = st.d_arena_off;
            A.pool = h->d_pool;
            A.arena = h->d_arena[lane];
            A.results = d_results + (out_off[b0] - out_off[0]);
            double n_wg = 0;
            if (h->trace) st.gap_from = h->n_waves ? (int)((h->n_waves - 1) & 3) : -1;
            // Every launch is bracketed by its own pair of events on the lane's stream; the wave as a whole by the first and
            // the last one (retire() books the GPU's busy time from them).
            size_t e_first = 0;
            if ((rc = next_event(h, st, e_first, S))) return rc;
            A.items = st.d_items;
            // option overlap: the sweep launch of a level goes to a second stream, next to the level kernel's launch of the same
            // level (their items belong to different requests); level L + 1 starts on either stream when BOTH launches of level L
            // have finished.  The tail of one launch - a sweep workgroup lives ~100 us - then runs under the body of the other.
            const bool two = h->overlap != 0;
            // streams of a wave: 0 = S (the level kernel), 1 = S2 (the sweep kernel), 2 = S3 (the segment kernel, option seg_kernel).  The
            // launches of level L run side by side; a launch of level L + 1 waits for ALL launches of level L: the one on its own
            // stream by stream order, the others through their end events (`prev`: the ends of the previous level - not the running
            // `last`, or the launches of one level would serialise: ADVICE r3).
            constexpr int kNS = 4;  // (3 = S4: the MFMA kernel, option mfma_kernel)
            hipStream_t SS[kNS] = {S, h->aux[lane][0], h->aux[lane][1], h->aux[lane][2]};
            constexpr size_t kNone = mibn_ctx::Set::Timed::kNone;
            long last[kNS], prev[kNS], saw[kNS][kNS];
            for (int a = 0; a < kNS; ++a) { last[a] = prev[a] = -1; for (int b = 0; b < kNS; ++b) saw[a][b] = -1; }
            int cur_level = -1;
            mibn_ctx::Set::Timed pair{-2, kNone, kNone, 0.0, 0.0, h->call_id};  // the level in progress, as one concurrent launch group
            auto close_pair = [&]() {
                if (pair.e0 != kNone || pair.e2 != kNone || pair.e4 != kNone || pair.e6 != kNone) st.timed.push_back(pair);
                pair = mibn_ctx::Set::Timed{-2, kNone, kNone, 0.0, 0.0, h->call_id};
            };
            if (two) {  // (the uploads, the previous wave's kernels: the arena is theirs until then)
                for (int j = 1; j < kNS; ++j) HIP_TRY(h, hipStreamWaitEvent(SS[j], st.ev[e_first], 0));
            }
            for (size_t li = 0; li < sc.launches.size();) {
                // one launch of the level kernel per level (all its classes of work together) unless split_kinds, one of the
                // segment kernel for the level's segments (the first class of a level; option seg_kernel) and one of the sweep kernel
                // for its SWEEP items (the last class: its own LDS budget)
                size_t lj = li + 1;
                const bool sweep = sc.launches[li].kid == kKidSweep;
                const bool seg = h->seg_kernel && !h->split_kinds && sc.launches[li].kid == kKidSeg;
                // (the two classes of ve_mfma_kernel are the last ones of a level in front of the sweep class: planner.cpp ClassOrder)
                auto is_mfma1 = [&](int kid) { return h->mfma_kernel && !h->split_kinds && (kid == kKidFiber0 + 4 || kid == kKidFiber0 + 6 + 4); };
                const bool mf = is_mfma1(sc.launches[li].kid);
                double bytes = sc.launches[li].alg_bytes;
                size_t grid = sc.launches[li].grid;
                if (!h->split_kinds && !sweep && !seg)
                    for (; lj < sc.launches.size() && sc.launches[lj].level == sc.launches[li].level && sc.launches[lj].kid != kKidSweep &&
                           is_mfma1(sc.launches[lj].kid) == mf; ++lj) {
                        bytes += sc.launches[lj].alg_bytes;
                        grid += sc.launches[lj].grid;
                    }
                const Launch &L = sc.launches[li];
                A.wg_item = st.d_wg_item + L.wg_level;
                A.wg_base = (uint32_t)(L.wg_first - L.wg_level);
                size_t e0 = 0, e1 = 0;
                const int si = two ? (sweep ? 1 : (seg ? 2 : (mf ? 3 : 0))) : 0;
                hipStream_t Sx = SS[si];
                if (two) {
                    if (L.level != cur_level) { cur_level = L.level; for (int j = 0; j < kNS; ++j) prev[j] = last[j]; close_pair(); }
                    for (int j = 0; j < kNS; ++j)
                        if (j != si && prev[j] > saw[si][j]) { HIP_TRY(h, hipStreamWaitEvent(Sx, st.ev[(size_t)prev[j]], 0)); saw[si][j] = prev[j]; }
                }
                if ((rc = next_event(h, st, e0, Sx))) return rc;
                if (sweep && h->sweep_dma) hipLaunchKernelGGL(ve_sweep_dma_kernel, dim3((unsigned)grid), dim3(kSweepWG), kSweepLdsBytes, Sx, A);
                else if (sweep) hipLaunchKernelGGL(ve_sweep_kernel, dim3((unsigned)grid), dim3(kSweepWG), kSweepLdsBytes, Sx, A);
                else if (seg) hipLaunchKernelGGL(ve_segment_kernel, dim3((unsigned)grid), dim3(kWG), 0, Sx, A);
                else if (mf) hipLaunchKernelGGL(ve_mfma_kernel, dim3((unsigned)grid), dim3(kWG), 0, Sx, A);
                else hipLaunchKernelGGL(ve_level_kernel, dim3((unsigned)grid), dim3(kWG), 0, Sx, A);
                if ((rc = next_event(h, st, e1, Sx))) return rc;
                if (two) {
                    last[si] = (long)e1;
                    if (!h->split_kinds) {
                        if (si == 1) { pair.e2 = e0; pair.e3 = e1; } else if (si == 2) { pair.e4 = e0; pair.e5 = e1; } else if (si == 3) { pair.e6 = e0; pair.e7 = e1; } else { pair.e0 = e0; pair.e1 = e1; }
                        pair.bytes += bytes;
                        pair.items += (double)grid;
                    }
                }
                st.timed.push_back({sweep ? (h->sweep_dma ? kNumKernels + 2 : kKidSweep) : (seg ? kNumKernels + 5 : (mf ? kNumKernels + 6 : (h->split_kinds ? L.kid : kNumKernels))), e0, e1, bytes, (double)grid, h->call_id});
                n_wg += (double)grid;
                li = lj;
            }
            {
                if (two) {
                    close_pair();
                    for (int j = 1; j < kNS; ++j)  // the wave ends on S
                        if (last[j] > saw[0][j]) HIP_TRY(h, hipStreamWaitEvent(S, st.ev[(size_t)last[j]], 0));
                }
                size_t e_last = 0;
                if ((rc = next_event(h, st, e_last, S))) return rc;
                st.timed.push_back({-1, e_first, e_last, 0.0, (double)(r1 - r0), h->call_id});
            }
            HIP_TRY(h, hipGetLastError());
            if (h->trace) {
                hipEvent_t &ge = h->gap_ev[h->n_waves & 3];
                if (!ge) HIP_TRY(h, hipEventCreate(&ge));
                HIP_TRY(h, hipEventRecord(ge, S));
            }
            ++h->n_waves;
            st.busy = true;
            h->stats.arena_bytes = std::max(h->stats.arena_bytes, (double)need_bytes);
            h->stats.n_workgroups += n_wg;
            r0 = r1;
        }
        h->stats.alg_bytes += ck.st.alg_bytes;
        h->stats.alg_flops += ck.st.alg_flops;
        h->stats.n_steps += ck.st.n_steps;
        h->stats.max_step_cells = std::max(h->stats.max_step_cells, ck.st.max_step_cells);
    }
    auto fold = [&]() {  // the host-side counters of this call -> totals (kernel times are folded by retire())
        h->total.plan_ms += h->stats.plan_ms;
        h->total.h2d_ms += h->stats.h2d_ms;
        h->total.d2h_ms += h->stats.d2h_ms;
        h->total.total_ms += h->stats.total_ms;
        h->total.alg_bytes += h->stats.alg_bytes;
        h->total.alg_flops += h->stats.alg_flops;
        h->total.n_steps += h->stats.n_steps;
        h->total.n_workgroups += h->stats.n_workgroups;
        h->total.arena_bytes = std::max(h->total.arena_bytes, h->stats.arena_bytes);
        h->total.max_step_cells = std::max(h->total.max_step_cells, h->stats.max_step_cells);
    };
    if (lane1_used) {  // the download (main stream) follows the kernels of both lanes
        if (!h->lane_ev) HIP_TRY(h, hipEventCreateWithFlags(&h->lane_ev, hipEventDisableTiming));
        HIP_TRY(h, hipEventRecord(h->lane_ev, h->stream2));
        HIP_TRY(h, hipStreamWaitEvent(h->stream, h->lane_ev, 0));
    }
    // (the fixed cost of a device-planned call is everything up to its last launch but the planning itself - validation, the fallbacks, the schedule,
    //  the uploads and some 600 launches: counting validation and schedule alone - 7.7 ms of 20 - left a two-thread rank host-bound once order_effort 1
    //  made its planning dearer, profiles/r06_bi_share.log)
    if (call_plan_ms > 0) call_fixed_ms = std::max(call_fixed_ms, now_ms() - t_start - call_plan_ms);
    if (B > 0) h->fixed_ms_per_req = h->fixed_ms_per_req > 0 ? 0.5 * h->fixed_ms_per_req + 0.5 * call_fixed_ms / (double)B : call_fixed_ms / (double)B;
    if (h->trace) std::fprintf(stderr, "[mibn plan] call of %lld requests: %.2f ms to the last launch (plan %.2f, h2d %.2f)\n", (long long)B, now_ms() - t_start, h->stats.plan_ms, h->stats.h2d_ms);
    if (ticket) {
        mibn_ctx::Pending &pd = h->pend[slot];
        mibn_ctx::Staging &sg = h->res_stage[slot];
        const size_t bytes = res_cells * 8;
        if (bytes > sg.cap) {
            if (sg.p) { HIP_TRY(h, hipHostFree(sg.p)); sg.p = nullptr; sg.cap = 0; }
            HIP_TRY(h, hipHostMalloc((void **)&sg.p, bytes + bytes / 4 + 4096, hipHostMallocDefault));
            sg.cap = bytes + bytes / 4 + 4096;
        }
        if (!pd.done) HIP_TRY(h, hipEventCreateWithFlags(&pd.done, hipEventDisableTiming));
        HIP_TRY(h, hipMemcpyAsync(sg.p, d_results, bytes, hipMemcpyDeviceToHost, h->stream));
        HIP_TRY(h, hipEventRecord(pd.done, h->stream));
        pd.active = true;
        pd.out = out + out_off[0];
        pd.cells = res_cells;
        h->next_slot ^= 1;
        h->stats.total_ms = now_ms() - t_start;
        fold();
        return MIBN_OK;
    }
    if ((rc = retire_all(h))) return rc;
    double t0 = now_ms();
    HIP_TRY(h, hipMemcpyAsync(out + out_off[0], d_results, res_cells * 8, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    h->stats.d2h_ms += now_ms() - t0;
    h->stats.total_ms = now_ms() - t_start;
    fold();
    return MIBN_OK;
}
}  // namespace

// Every error exit of a call drains what the call (and earlier asynchronous calls) already put on the streams before the caller
// sees the error: the caller may free or reuse its request arrays and the pinned `out` the kernels and copies still touch
// (ADVICE r3 did this for the device-planner exits only; ADVICE r4: every exit behind the first launch).
static int run_batch(mibn_t *h, uint32_t flags, int64_t B, const int64_t *q_off, const int32_t *q_vars, const int64_t *e_off, const int32_t *e_vars,
              const int32_t *e_codes, const int64_t *out_off, double *out, int32_t *ticket) {
    const int rc = run_batch_body(h, flags, B, q_off, q_vars, e_off, e_vars, e_codes, out_off, out, ticket);
    if (rc != MIBN_OK && h && !h->planner_only) {
        const std::string keep = h->err;
        (void)hipSetDevice(h->device);
        for (hipStream_t q : {h->search_stream, h->copy_stream, h->stream, h->stream2})
            if (q) (void)hipStreamSynchronize(q);
        for (auto &la : h->aux)
            for (hipStream_t a : la)
                if (a) (void)hipStreamSynchronize(a);
        (void)hipGetLastError();
        h->err = keep;
    }
    return rc;
}

extern "C" int mibn_query_batch(mibn_t *h, int64_t B, const int64_t *q_off, const int32_t *q_vars,
                                const int64_t *e_off, const int32_t *e_vars, const int32_t *e_codes,
                                const int64_t *out_off, double *out) {
    if (h && (h->pend[0].active || h->pend[1].active)) { h->err = "asynchronous calls in flight: collect them with mibn_wait first"; return MIBN_E_STATE; }
    return run_batch(h, 0, B, q_off, q_vars, e_off, e_vars, e_codes, out_off, out, nullptr);
}

extern "C" int mibn_query_batch_ex(mibn_t *h, uint32_t flags, int64_t B, const int64_t *q_off, const int32_t *q_vars,
                                   const int64_t *e_off, const int32_t *e_vars, const int32_t *e_codes,
                                   const int64_t *out_off, double *out) {
    if (h && (h->pend[0].active || h->pend[1].active)) { h->err = "asynchronous calls in flight: collect them with mibn_wait first"; return MIBN_E_STATE; }
    if (flags & ~(uint32_t)MIBN_Q_NOPRUNE) { if (h) h->err = "unknown query flag"; return MIBN_E_ARG; }
    return run_batch(h, flags, B, q_off, q_vars, e_off, e_vars, e_codes, out_off, out, nullptr);
}

extern "C" int mibn_submit_batch(mibn_t *h, int64_t B, const int64_t *q_off, const int32_t *q_vars,
                                 const int64_t *e_off, const int32_t *e_vars, const int32_t *e_codes,
                                 const int64_t *out_off, double *out, int32_t *ticket) {
    if (!ticket) return MIBN_E_ARG;
    return run_batch(h, 0, B, q_off, q_vars, e_off, e_vars, e_codes, out_off, out, ticket);
}

extern "C" int mibn_wait(mibn_t *h, int32_t ticket) {
    if (!h || ticket < 0 || ticket > 1) return MIBN_E_ARG;
    mibn_ctx::Pending &pd = h->pend[ticket];
    if (!pd.active) return MIBN_OK;  // (an empty batch, or already collected)
    HIP_TRY(h, hipSetDevice(h->device));
    const double t0 = now_ms();
    HIP_TRY(h, hipEventSynchronize(pd.done));
    std::memcpy(pd.out, h->res_stage[ticket].p, pd.cells * 8);
    pd.active = false;
    h->stats.d2h_ms += now_ms() - t0;
    h->total.d2h_ms += now_ms() - t0;
    return MIBN_OK;
}

extern "C" int mibn_drain(mibn_t *h) {
    if (!h) return MIBN_E_ARG;
    if (h->planner_only) return MIBN_OK;
    HIP_TRY(h, hipSetDevice(h->device));
    return retire_all(h);
}

extern "C" int mibn_total_stats(const mibn_t *h, mibn_stats *out) {
    if (!h || !out) return MIBN_E_ARG;
    *out = h->total;
    return MIBN_OK;
}

extern "C" int mibn_total_kernel_stats(const mibn_t *h, int32_t cap, mibn_kernel_stat *out, int32_t *n) {
    if (!h || !out || !n) return MIBN_E_ARG;
    int k = 0;
    for (int i = 0; i <= kNumKernels + 6 && k < cap; ++i)
        if (h->ktotal[i].launches > 0) out[k++] = h->ktotal[i];
    *n = k;
    return MIBN_OK;
}

extern "C" int mibn_last_kernel_stats(const mibn_t *h, int32_t cap, mibn_kernel_stat *out, int32_t *n) {
    if (!h || !out || !n) return MIBN_E_ARG;
    int k = 0;
    for (int i = 0; i <= kNumKernels + 6 && k < cap; ++i)
        if (h->kstats[i].launches > 0) out[k++] = h->kstats[i];
    *n = k;
    return MIBN_OK;
}

extern "C" int mibn_gibbs(mibn_t *h, int32_t n_q, const int32_t *q_vars, int32_t n_e, const int32_t *e_vars,
                          const int32_t *e_codes, const int32_t *cycle, int64_t n_chains, int64_t n_iterations,
                          uint64_t seed, int64_t *counts) {
    return mibn_gibbs_shard(h, n_q, q_vars, n_e, e_vars, e_codes, cycle, 0, n_chains, n_iterations, seed, counts);
}

extern "C" int mibn_gibbs_shard(mibn_t *h, int32_t n_q, const int32_t *q_vars, int32_t n_e, const int32_t *e_vars,
                                const int32_t *e_codes, const int32_t *cycle, int64_t chain_first, int64_t n_chains,
                                int64_t n_iterations, uint64_t seed, int64_t *counts) {
    if (!h || !q_vars || !counts || n_chains < 0 || chain_first < 0 || n_iterations < 0) return MIBN_E_ARG;
    if (h->planner_only) { h->err = "planner-only context: no HIP device bound (there is no CPU fallback)"; return MIBN_E_NODEVICE; }
    if (!h->has_net) { h->err = "set_network first"; return MIBN_E_STATE; }
    Request rq;
    rq.nq = n_q; rq.qvars = q_vars; rq.ne = n_e; rq.evars = e_vars;
    std::string e = validate_request(h->net, rq);
    if (!e.empty()) { h->err = e; return MIBN_E_ARG; }
    if (n_chains == 0) {  // an empty shard (more ranks than chains)
        int64_t cells = 1;
        for (int k = 0; k < n_q; ++k) cells *= h->net.card[q_vars[k]];
        for (int64_t c = 0; c < cells; ++c) counts[c] = 0;
        return MIBN_OK;
    }
    HIP_TRY(h, hipSetDevice(h->device));
    return gibbs_run(h->net, h->d_pool, h->stream, h->gibbs_lds, n_q, q_vars, n_e, e_vars, e_codes, cycle, chain_first, n_chains,
                     n_iterations, seed, counts, h->err, h->stats.kernel_ms);
}

extern "C" int mibn_gibbs_conditional(mibn_t *h, int32_t n_e, const int32_t *e_vars, const int32_t *e_codes, const int32_t *cycle,
                                      int32_t var, int64_t n_rows, const uint8_t *states, double *out) {
    if (!h || n_rows < 0 || (n_rows && (!states || !out)) || (n_e && (!e_vars || !e_codes))) return MIBN_E_ARG;
    if (h->planner_only) { h->err = "planner-only context: no HIP device bound (there is no CPU fallback)"; return MIBN_E_NODEVICE; }
    if (!h->has_net) { h->err = "set_network first"; return MIBN_E_STATE; }
    if (var < 0 || var >= h->net.n_vars) { h->err = "gibbs conditional: unknown variable"; return MIBN_E_ARG; }
    for (int64_t r = 0; r < n_rows; ++r)
        for (int v = 0; v < h->net.n_vars; ++v)
            if (states[r * h->net.n_vars + v] >= h->net.card[v]) { h->err = "gibbs conditional: a state code is outside its variable's domain"; return MIBN_E_ARG; }
    if (n_rows == 0) return MIBN_OK;
    HIP_TRY(h, hipSetDevice(h->device));
    const int32_t q0 = var;  // (the histogram of the ordinary launch: one variable, unused)
    double ms = 0;
    return gibbs_run(h->net, h->d_pool, h->stream, h->gibbs_lds, 1, &q0, n_e, e_vars, e_codes, cycle, 0, n_rows, 1, 0, nullptr, h->err, ms,
                     var, states, out);
}

extern "C" int mibn_sample(mibn_t *h, int64_t n_samples, int32_t n_init, const int32_t *init_vars, const int32_t *init_codes,
                           uint64_t seed, uint8_t *states) {
    if (!h || n_samples < 0 || !states || (n_init && (!init_vars || !init_codes))) return MIBN_E_ARG;
    if (h->planner_only) { h->err = "planner-only context: no HIP device bound (there is no CPU fallback)"; return MIBN_E_NODEVICE; }
    if (!h->has_net) { h->err = "set_network first"; return MIBN_E_STATE; }
    if (n_samples == 0) return MIBN_OK;
    HIP_TRY(h, hipSetDevice(h->device));
    return sample_run(h->net, h->d_pool, h->stream, kSampleMode, 0, nullptr, n_init, init_vars, init_codes, 0, nullptr, nullptr,
                      n_samples, seed, states, nullptr, nullptr, h->err);
}

extern "C" int mibn_sample_probe(mibn_t *h, int64_t n_rows, const uint8_t *states, int32_t cdf_stride, double *likelihood, double *cdf) {
    if (!h || n_rows < 0 || (n_rows && (!states || !likelihood || !cdf))) return MIBN_E_ARG;
    if (h->planner_only) { h->err = "planner-only context: no HIP device bound (there is no CPU fallback)"; return MIBN_E_NODEVICE; }
    if (!h->has_net) { h->err = "set_network first"; return MIBN_E_STATE; }
    for (int v = 0; v < h->net.n_vars; ++v)
        if (h->net.card[v] > cdf_stride) { h->err = "sample probe: cdf_stride below a variable's cardinality"; return MIBN_E_ARG; }
    for (int64_t r = 0; r < n_rows; ++r)
        for (int v = 0; v < h->net.n_vars; ++v)
            if (states[r * h->net.n_vars + v] >= h->net.card[v]) { h->err = "sample probe: a state code is outside its variable's domain"; return MIBN_E_ARG; }
    if (n_rows == 0) return MIBN_OK;
    HIP_TRY(h, hipSetDevice(h->device));
    return sample_run(h->net, h->d_pool, h->stream, kProbeMode, 0, nullptr, 0, nullptr, nullptr, 0, nullptr, nullptr, n_rows, 0,
                      const_cast<uint8_t *>(states), nullptr, nullptr, h->err, cdf_stride, likelihood, cdf);
}

extern "C" int mibn_sampling_query(mibn_t *h, int32_t mode, int32_t n_q, const int32_t *q_vars, int32_t n_e, const int32_t *e_vars,
                                   const int32_t *e_codes, int64_t n_samples, uint64_t seed, double *weight_sum, int64_t *counts) {
    if (!h || !q_vars || !counts || n_samples < 0 || (mode != MIBN_REJECTION && mode != MIBN_LIKELIHOOD)) return MIBN_E_ARG;
    if (mode == MIBN_LIKELIHOOD && !weight_sum) return MIBN_E_ARG;
    if (h->planner_only) { h->err = "planner-only context: no HIP device bound (there is no CPU fallback)"; return MIBN_E_NODEVICE; }
    if (!h->has_net) { h->err = "set_network first"; return MIBN_E_STATE; }
    Request rq;
    rq.nq = n_q; rq.qvars = q_vars; rq.ne = n_e; rq.evars = e_vars;
    std::string e = validate_request(h->net, rq);
    if (!e.empty()) { h->err = e; return MIBN_E_ARG; }
    HIP_TRY(h, hipSetDevice(h->device));
    if (mode == MIBN_LIKELIHOOD)  // the event is clamped (bayes_net.py:646)
        return sample_run(h->net, h->d_pool, h->stream, kLikelihoodMode, n_q, q_vars, n_e, e_vars, e_codes, 0, nullptr, nullptr,
                          n_samples, seed, nullptr, weight_sum, counts, h->err);
    // rejection: nothing is clamped, samples that disagree with the event are dropped (bayes_net.py:604-612);
    // a label outside the domain can never be drawn
    for (int i = 0; i < n_e; ++i)
        if (e_codes[i] < 0 || e_codes[i] >= h->net.card[e_vars[i]]) {
            int64_t cells = 1;
            for (int k = 0; k < n_q; ++k) cells *= h->net.card[q_vars[k]];
            for (int64_t c = 0; c < cells; ++c) counts[c] = 0;
            return MIBN_OK;
        }
    return sample_run(h->net, h->d_pool, h->stream, kRejectionMode, n_q, q_vars, 0, nullptr, nullptr, n_e, e_vars, e_codes, n_samples,
                      seed, nullptr, nullptr, counts, h->err);
}

extern "C" int mibn_count_tables(mibn_t *h, int64_t n_rows, int32_t n_cols, const uint8_t *codes, int32_t row_major, const int32_t *card,
                                 int32_t n_tables, const int64_t *scope_off, const int32_t *scope_cols, const int64_t *counts_off,
                                 int64_t *counts) {
    if (!h || n_rows < 0 || n_cols < 0 || n_tables < 0 || (n_rows && n_cols && !codes) || !card || !scope_off || !scope_cols ||
        !counts_off || !counts || (row_major != 0 && row_major != 1))
        return MIBN_E_ARG;
    if (h->planner_only) { h->err = "planner-only context: no HIP device bound (there is no CPU fallback)"; return MIBN_E_NODEVICE; }
    HIP_TRY(h, hipSetDevice(h->device));
    return count_run(h->stream, n_rows, n_cols, codes, row_major != 0, card, n_tables, scope_off, scope_cols, counts_off, counts, h->err);
}

extern "C" int mibn_plan_order(mibn_t *h, int32_t n_q, const int32_t *q_vars, int32_t n_e, const int32_t *e_vars,
                               int32_t *order, int32_t *n) {
    if (!h || !order || !n) return MIBN_E_ARG;
    if (!h->has_net) { h->err = "set_network first"; return MIBN_E_STATE; }
    Request rq;
    rq.nq = n_q; rq.qvars = q_vars; rq.ne = n_e; rq.evars = e_vars;
    std::string e = validate_request(h->net, rq);
    if (!e.empty()) { h->err = e; return MIBN_E_ARG; }
    std::vector<uint32_t> prog;
    std::vector<int32_t> ord;
    PlanStats st;
    st.order = &ord;
    e = plan_request(h->net, rq, prog, st);
    if (!e.empty()) { h->err = e; return MIBN_E_LIMIT; }
    *n = (int32_t)ord.size();
    std::copy(ord.begin(), ord.end(), order);
    return MIBN_OK;
}

extern "C" int mibn_estimate_costs(mibn_t *h, int64_t B, const int64_t *q_off, const int32_t *q_vars, const int64_t *e_off,
                                   const int32_t *e_vars, double *cost) {
    if (!h || B < 0 || !q_off || !e_off || (B && (!q_vars || !cost))) return MIBN_E_ARG;
    if (!h->has_net) { h->err = "set_network first"; return MIBN_E_STATE; }
    ensure_pool(h);
    estimate_costs(h->net, *h->pool, B, q_off, q_vars, e_off, e_vars, cost);
    return MIBN_OK;
}

extern "C" int mibn_device_synchronize(mibn_t *h) {
    if (!h) return MIBN_E_ARG;
    if (h->planner_only) return MIBN_OK;
    HIP_TRY(h, hipSetDevice(h->device));
    HIP_TRY(h, hipDeviceSynchronize());
    return MIBN_OK;
}

// ---------------------------------------------------------------------------------------------------- RCCL
// One process per GPU; the data path has no collective (independent request / chain shards), these are the final
// gather / reduce over xGMI.  librccl.so (573 MB) is loaded on the first mibn_comm_* call only.
namespace {

int comm_load(mibn_ctx *h) {
    mibn_ctx::Comm &c = h->comm;
    if (c.dl) return MIBN_OK;
    const char *env = std::getenv("MIBN_RCCL_LIB");
    const char *cands[] = {env, "librccl.so.1", "/opt/rocm/lib/librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so"};
    for (const char *name : cands) {
        if (!name || !*name) continue;
        if ((c.dl = dlopen(name, RTLD_NOW | RTLD_LOCAL))) break;
    }
    if (!c.dl) { h->err = std::string("cannot load librccl.so: ") + dlerror(); return MIBN_E_COMM; }
#define MIBN_SYM(field, sym)                                                                  \
    c.field = reinterpret_cast<decltype(c.field)>(dlsym(c.dl, #sym));                         \
    if (!c.field) { h->err = "librccl.so lacks " #sym; dlclose(c.dl); c.dl = nullptr; return MIBN_E_COMM; }
    MIBN_SYM(GetUniqueId, ncclGetUniqueId)
    MIBN_SYM(CommInitRank, ncclCommInitRank)
    MIBN_SYM(CommDestroy, ncclCommDestroy)
    MIBN_SYM(CommCount, ncclCommCount)
    MIBN_SYM(CommUserRank, ncclCommUserRank)
    MIBN_SYM(AllGather, ncclAllGather)
    MIBN_SYM(Reduce, ncclReduce)
    MIBN_SYM(AllReduce, ncclAllReduce)
    MIBN_SYM(GetErrorString, ncclGetErrorString)
#undef MIBN_SYM
    return MIBN_OK;
}

#define NCCL_TRY(h, expr)                                                                             \
    do {                                                                                              \
        ncclResult_t r_ = (expr);                                                                     \
        if (r_ != ncclSuccess) {                                                                      \
            (h)->err = std::string(#expr) + ": " + (h)->comm.GetErrorString(r_);                      \
            return MIBN_E_COMM;                                                                       \
        }                                                                                             \
    } while (0)

int comm_ready(mibn_ctx *h) {
    if (!h) return MIBN_E_ARG;
    if (h->planner_only) { h->err = "planner-only context: no HIP device bound"; return MIBN_E_NODEVICE; }
    if (!h->comm.comm) { h->err = "mibn_comm_init first"; return MIBN_E_STATE; }
    HIP_TRY(h, hipSetDevice(h->device));
    return MIBN_OK;
}

int comm_buf(mibn_ctx *h, void *&ptr, size_t &cap, size_t bytes) {
    if (bytes <= cap) return MIBN_OK;
    if (ptr) { HIP_TRY(h, hipFree(ptr)); ptr = nullptr; cap = 0; }
    HIP_TRY(h, hipMalloc(&ptr, bytes + bytes / 4 + 256));
    cap = bytes + bytes / 4 + 256;
    return MIBN_OK;
}

}  // namespace

// dlopen + symbol resolution only: what can fail on ONE rank of a node, checked before anybody enters the collective init.
// (ncclGetUniqueId is for rank 0 alone: on any other rank it starts a bootstrap listener thread and a socket that wait for
// `world` connections which never arrive - ADVICE r3.)
extern "C" int mibn_comm_probe(mibn_t *h) {
    if (!h) return MIBN_E_ARG;
    if (h->planner_only) { h->err = "planner-only context: no HIP device bound"; return MIBN_E_NODEVICE; }
    return comm_load(h);
}

// One line about this context's device for the launch log: name, PCI bus id, and the link (type, hops) to every other visible
// device - hipExtGetLinkTypeAndHopCount; type 4 = xGMI (HSA_AMD_LINK_INFO_TYPE_XGMI), 2 = PCIe.
extern "C" int mibn_device_info(mibn_t *h, char *buf, int32_t cap) {
    if (!h || !buf || cap < 1) return MIBN_E_ARG;
    if (h->planner_only) { h->err = "planner-only context: no HIP device bound"; return MIBN_E_NODEVICE; }
    hipDeviceProp_t pr;
    HIP_TRY(h, hipGetDeviceProperties(&pr, h->device));
    char bus[64] = "?";
    (void)hipDeviceGetPCIBusId(bus, sizeof(bus), h->device);
    std::string out = "device " + std::to_string(h->device) + " " + pr.name + " (" + pr.gcnArchName + ") pci " + bus + " links:";
    int n = 0;
    (void)hipGetDeviceCount(&n);
    for (int d = 0; d < n; ++d) {
        if (d == h->device) continue;
        uint32_t type = 0, hops = 0;
        if (hipExtGetLinkTypeAndHopCount(h->device, d, &type, &hops) == hipSuccess)
            out += " ->" + std::to_string(d) + ":" + (type == 4 ? "xgmi" : type == 2 ? "pcie" : "type" + std::to_string(type)) + "/" + std::to_string(hops);
        else { (void)hipGetLastError(); out += " ->" + std::to_string(d) + ":?"; }
    }
    if (n <= 1) out += " none (one visible device)";
    std::snprintf(buf, (size_t)cap, "%s", out.c_str());
    return MIBN_OK;
}

extern "C" int mibn_comm_unique_id(mibn_t *h, void *id_out) {
    if (!h || !id_out) return MIBN_E_ARG;
    if (h->planner_only) { h->err = "planner-only context: no HIP device bound"; return MIBN_E_NODEVICE; }
    int rc;
    if ((rc = comm_load(h))) return rc;
    static_assert(sizeof(ncclUniqueId) == MIBN_COMM_ID_BYTES, "RCCL unique id size");
    ncclUniqueId id;
    NCCL_TRY(h, h->comm.GetUniqueId(&id));
    std::memcpy(id_out, &id, sizeof(id));
    return MIBN_OK;
}

extern "C" int mibn_comm_init(mibn_t *h, int32_t rank, int32_t world, const void *id_in) {
    if (!h || !id_in || world < 1 || rank < 0 || rank >= world) return MIBN_E_ARG;
    if (h->planner_only) { h->err = "planner-only context: no HIP device bound"; return MIBN_E_NODEVICE; }
    if (h->comm.comm) { h->err = "communicator already initialised"; return MIBN_E_STATE; }
    int rc;
    if ((rc = comm_load(h))) return rc;
    HIP_TRY(h, hipSetDevice(h->device));
    ncclUniqueId id;
    std::memcpy(&id, id_in, sizeof(id));
    if (!h->comm.stream) HIP_TRY(h, hipStreamCreateWithFlags(&h->comm.stream, hipStreamNonBlocking));
    // The collective init blocks until every rank of `world` has joined - for ever when one of them died or took another id.
    // Bounded: the call runs on a helper thread, and a rank that has waited MIBN_COMM_INIT_TIMEOUT_S seconds (default 180) gives
    // up with an error that names the likely causes instead of hanging the launch (the helper thread is abandoned: the process
    // is expected to exit on this error).
    double limit_s = 180.0;
    if (const char *e = std::getenv("MIBN_COMM_INIT_TIMEOUT_S")) { const double v = std::atof(e); if (v > 0) limit_s = v; }
    struct InitState { std::mutex m; std::condition_variable cv; bool done = false; ncclResult_t res = ncclSuccess; ncclComm_t comm = nullptr; };
    auto state = std::make_shared<InitState>();
    const int device = h->device;
    auto init_fn = h->comm.CommInitRank;
    std::thread([state, init_fn, device, world, id, rank]() {
        (void)hipSetDevice(device);
        ncclComm_t c = nullptr;
        const ncclResult_t r = init_fn(&c, world, id, rank);
        std::lock_guard<std::mutex> lk(state->m);
        state->res = r;
        state->comm = c;
        state->done = true;
        state->cv.notify_all();
    }).detach();
    {
        std::unique_lock<std::mutex> lk(state->m);
        if (!state->cv.wait_for(lk, std::chrono::duration<double>(limit_s), [&] { return state->done; })) {
            h->err = "ncclCommInitRank: rank " + std::to_string(rank) + " of " + std::to_string(world) + " on device " + std::to_string(device) +
                     " still waiting after " + std::to_string((int)limit_s) + " s (MIBN_COMM_INIT_TIMEOUT_S) - a rank died before the init, the ranks "
                     "hold different ids, two ranks share one device, or the GPUs cannot reach each other (check HSA_ENABLE_IPC_MODE_LEGACY=0)";
            return MIBN_E_COMM;
        }
        if (state->res != ncclSuccess) { h->err = std::string("ncclCommInitRank: ") + h->comm.GetErrorString(state->res); return MIBN_E_COMM; }
        h->comm.comm = state->comm;
    }
    h->comm.rank = rank;
    h->comm.world = world;
    return MIBN_OK;
}

// What RCCL itself says about the communicator (ncclCommCount / ncclCommUserRank) - not what the caller asked for: the bench
// line of an N > 1 run carries these, so that a transport that silently degraded cannot claim N ranks.
extern "C" int mibn_comm_count(mibn_t *h, int32_t *n_ranks, int32_t *my_rank) {
    int rc;
    if ((rc = comm_ready(h))) return rc;
    int n = 0, r = -1;
    NCCL_TRY(h, h->comm.CommCount(h->comm.comm, &n));
    NCCL_TRY(h, h->comm.CommUserRank(h->comm.comm, &r));
    if (n_ranks) *n_ranks = n;
    if (my_rank) *my_rank = r;
    return MIBN_OK;
}

extern "C" int mibn_comm_destroy(mibn_t *h) {
    if (!h) return MIBN_E_ARG;
    mibn_ctx::Comm &c = h->comm;
    if (c.comm) {
        (void)hipSetDevice(h->device);
        if (c.stream) (void)hipStreamSynchronize(c.stream);
        (void)c.CommDestroy(c.comm);
        c.comm = nullptr;
    }
    if (c.stream) { (void)hipStreamDestroy(c.stream); c.stream = nullptr; }
    if (c.d_send) { (void)hipFree(c.d_send); c.d_send = nullptr; c.send_cap = 0; }
    if (c.d_recv) { (void)hipFree(c.d_recv); c.d_recv = nullptr; c.recv_cap = 0; }
    return MIBN_OK;
}

extern "C" int mibn_comm_allgather_f64(mibn_t *h, const double *send, int64_t n, double *recv) {
    int rc;
    if ((rc = comm_ready(h))) return rc;
    if (n < 0 || (n && (!send || !recv))) return MIBN_E_ARG;
    if (n == 0) return MIBN_OK;
    mibn_ctx::Comm &c = h->comm;
    const size_t bytes = (size_t)n * 8;
    if ((rc = comm_buf(h, c.d_send, c.send_cap, bytes))) return rc;
    if ((rc = comm_buf(h, c.d_recv, c.recv_cap, bytes * (size_t)c.world))) return rc;
    HIP_TRY(h, hipMemcpyAsync(c.d_send, send, bytes, hipMemcpyHostToDevice, c.stream));
    NCCL_TRY(h, c.AllGather(c.d_send, c.d_recv, (size_t)n, ncclDouble, c.comm, c.stream));
    HIP_TRY(h, hipMemcpyAsync(recv, c.d_recv, bytes * (size_t)c.world, hipMemcpyDeviceToHost, c.stream));
    HIP_TRY(h, hipStreamSynchronize(c.stream));
    return MIBN_OK;
}

extern "C" int mibn_comm_reduce_i64(mibn_t *h, int64_t *buf, int64_t n, int32_t root) {
    int rc;
    if ((rc = comm_ready(h))) return rc;
    mibn_ctx::Comm &c = h->comm;
    if (n < 0 || (n && !buf) || root < 0 || root >= c.world) return MIBN_E_ARG;
    if (n == 0) return MIBN_OK;
    const size_t bytes = (size_t)n * 8;
    if ((rc = comm_buf(h, c.d_send, c.send_cap, bytes))) return rc;
    if ((rc = comm_buf(h, c.d_recv, c.recv_cap, bytes))) return rc;
    HIP_TRY(h, hipMemcpyAsync(c.d_send, buf, bytes, hipMemcpyHostToDevice, c.stream));
    NCCL_TRY(h, c.Reduce(c.d_send, c.d_recv, (size_t)n, ncclInt64, ncclSum, root, c.comm, c.stream));
    if (c.rank == root) HIP_TRY(h, hipMemcpyAsync(buf, c.d_recv, bytes, hipMemcpyDeviceToHost, c.stream));
    HIP_TRY(h, hipStreamSynchronize(c.stream));
    return MIBN_OK;
}

extern "C" int mibn_comm_allreduce_max_f64(mibn_t *h, double *buf, int64_t n) {
    int rc;
    if ((rc = comm_ready(h))) return rc;
    if (n < 0 || (n && !buf)) return MIBN_E_ARG;
    if (n == 0) return MIBN_OK;
    mibn_ctx::Comm &c = h->comm;
    const size_t bytes = (size_t)n * 8;
    if ((rc = comm_buf(h, c.d_send, c.send_cap, bytes))) return rc;
    if ((rc = comm_buf(h, c.d_recv, c.recv_cap, bytes))) return rc;
    HIP_TRY(h, hipMemcpyAsync(c.d_send, buf, bytes, hipMemcpyHostToDevice, c.stream));
    NCCL_TRY(h, c.AllReduce(c.d_send, c.d_recv, (size_t)n, ncclDouble, ncclMax, c.comm, c.stream));
    HIP_TRY(h, hipMemcpyAsync(buf, c.d_recv, bytes, hipMemcpyDeviceToHost, c.stream));
    HIP_TRY(h, hipStreamSynchronize(c.stream));
    return MIBN_OK;
}

extern "C" int mibn_comm_barrier(mibn_t *h) {
    int rc;
    if ((rc = comm_ready(h))) return rc;
    HIP_TRY(h, hipDeviceSynchronize());  // this rank's own device work first
    double one = 1.0;
    return mibn_comm_allreduce_max_f64(h, &one, 1);
}
