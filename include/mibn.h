/*
 * mibn - MI355X-native exact inference backend for sorobn-style Bayesian networks.
 *
 * C-ABI drop-in boundary for the hot path of MaxHalford/sorobn:
 *
 *   BayesNet.query(..., algorithm="exact")   sorobn/bayes_net.py:796-875
 *     -> BayesNet._variable_elimination      sorobn/bayes_net.py:739-794
 *          -> pointwise_mul / pointwise_mul_two   sorobn/bayes_net.py:233-256
 *          -> CDTAccessor.sum_out                 sorobn/bayes_net.py:100-103
 *   BayesNet.query(..., algorithm="gibbs")   sorobn/bayes_net.py:665-737
 *
 * The reference is pure Python with no FFI; the seam is the method boundary
 * `_variable_elimination(*query, event=...) -> pd.Series` (bayes_net.py:739, called from 848) and
 * `_gibbs_sampling(*query, event=..., n_iterations=...)` (bayes_net.py:665, called from 851-853).
 * A Python caller flattens the pandas CPTs once (mibn_set_network) and then answers batches of
 * requests through mibn_query_batch; labels <-> codes and pandas objects stay on the Python side
 * (sorobn_amd/bayes_net.py; binding shown in INTEGRATION.md).
 *
 * Conventions: plain C types only; every function returns 0 on success or a negative MIBN_E_*
 * code, with a message available from mibn_last_error(); the caller owns all host buffers, the
 * library owns all device memory; one host thread per handle (thread-compatible, not thread-safe).
 * There is NO CPU fallback: mibn_create fails with MIBN_E_NODEVICE when no gfx950 device is
 * visible.
 */
#ifndef MIBN_H
#define MIBN_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MIBN_OK 0
#define MIBN_E_ARG (-1)       /* invalid argument */
#define MIBN_E_NODEVICE (-2)  /* no HIP device / wrong architecture */
#define MIBN_E_HIP (-3)       /* HIP runtime error */
#define MIBN_E_NOMEM (-4)     /* plan needs more device memory than the arena budget */
#define MIBN_E_STATE (-5)     /* call order (e.g. query before set_network) */
#define MIBN_E_LIMIT (-6)     /* a compile-time limit (scope size, axes) was exceeded */
#define MIBN_E_COMM (-7)      /* RCCL missing or a collective failed */

typedef struct mibn_ctx mibn_t;

/* Number of visible HIP devices (0 when there is no GPU/driver). Never fails hard. */
int mibn_device_count(int *count);

/* Library version string (static storage). */
const char *mibn_version(void);

/* Create a context bound to HIP device `device`. */
int mibn_create(int device, mibn_t **out);
void mibn_destroy(mibn_t *h);
const char *mibn_last_error(const mibn_t *h);

/*
 * Install the network (replaces `self.P` of the reference, bayes_net.py:324, after prepare()).
 *   n_vars             variables 0..n_vars-1; variable v owns exactly one CPT, factor v
 *   card[n_vars]       cardinality of each variable (size of its sorted label domain)
 *   scope_off[n_vars+1], scope_vars[]   CSR: scope of factor v = scope_vars[scope_off[v]..]
 *                      = [*parents(v), v] - the level order of the reference's CPT Series
 *   value_off[n_vars+1], values[]       dense float64 table of factor v in C-order over its scope
 *                      (last scope entry fastest), absent rows = 0.0
 */
int mibn_set_network(mibn_t *h, int32_t n_vars, const int32_t *card, const int64_t *scope_off,
                     const int32_t *scope_vars, const int64_t *value_off, const double *values);

/*
 * Optional elimination-order hints: n_hints priority arrays of n_vars entries (lower = earlier).
 * The planner evaluates every hint next to its built-in candidate orders with the section 8(d)
 * cost model and executes the cheapest.  Orders change only floating-point rounding (~1e-16), not
 * the answer (the reference's own order is an arbitrary set-iteration order, bayes_net.py:766,779).
 */
int mibn_set_order_hints(mibn_t *h, int32_t n_hints, const int32_t *priorities);

/*
 * Answer a batch of exact posterior queries (= B calls of BayesNet._variable_elimination followed
 * by the normalisation of bayes_net.py:790).
 *   q_off[B+1], q_vars[]   CSR: query variables of request b, in the caller's argument order
 *   e_off[B+1], e_vars[], e_codes[]   CSR: evidence variables and their label codes; a code of -1
 *                          (label outside the domain) yields an all-zero posterior, like the
 *                          reference's empty Series
 *   out_off[B+1], out[]    dense posterior of request b at out[out_off[b]..out_off[b+1]) in C-order
 *                          over its query variables (last query variable fastest); all zeros when
 *                          the evidence has probability 0
 * Errors: MIBN_E_ARG for unknown variables, duplicates or query/evidence overlap.
 */
int mibn_query_batch(mibn_t *h, int64_t B, const int64_t *q_off, const int32_t *q_vars,
                     const int64_t *e_off, const int32_t *e_vars, const int32_t *e_codes,
                     const int64_t *out_off, double *out);

/*
 * mibn_query_batch with per-call flags (nothing engine-wide changes, so it is safe next to asynchronous calls in flight
 * on other engines of the same network):
 *   MIBN_Q_NOPRUNE   multiply every CPT instead of pruning to the ancestors of the query / evidence variables
 *                    (bayes_net.py:763-765): the semantics of full_joint_dist / predict_proba (bayes_net.py:460), where
 *                    with sparse or unnormalised CPTs a barren node does not sum to 1.
 */
#define MIBN_Q_NOPRUNE 1
int mibn_query_batch_ex(mibn_t *h, uint32_t flags, int64_t B, const int64_t *q_off, const int32_t *q_vars,
                        const int64_t *e_off, const int32_t *e_vars, const int32_t *e_codes,
                        const int64_t *out_off, double *out);

/* Statistics of the last mibn_query_batch call (for the roofline report). */
typedef struct mibn_stats {
    double alg_bytes;      /* sum over steps of 8*(sum input cells + output cells) */
    double alg_flops;      /* sum over steps of n_inputs * product-scope cells */
    double n_steps;        /* elimination + final product steps executed */
    double kernel_ms;      /* HIP-event time of all kernel launches (sum over launches) */
    double plan_ms;        /* host planning wall time */
    double h2d_ms;         /* program upload */
    double d2h_ms;         /* result download */
    double total_ms;       /* whole call, host wall clock */
    double n_launches;     /* kernel launches */
    double arena_bytes;    /* device scratch arena in use */
    double max_step_cells; /* largest product scope of any step */
    double n_workgroups;   /* work items (= workgroups) launched */
} mibn_stats;
int mibn_last_stats(const mibn_t *h, mibn_stats *out);

/* Per-kernel breakdown of the last mibn_query_batch call: HIP-event time on the library's stream and
 * the section-8(d) algorithmic bytes of the steps each kernel executed (tiles pro rata).  Fills at most
 * `cap` entries (one per kernel that ran), *n = number filled. */
typedef struct mibn_kernel_stat {
    char name[48];
    double launches, ms, alg_bytes, items;
} mibn_kernel_stat;
int mibn_last_kernel_stats(const mibn_t *h, int32_t cap, mibn_kernel_stat *out, int32_t *n);

/*
 * Asynchronous form of mibn_query_batch for streams of batches: mibn_submit_batch plans, uploads and launches and
 * returns at once with a ticket; mibn_wait(ticket) blocks until that call's posteriors are in `out` (which must
 * stay valid until then).  Up to two calls may be in flight, so the host plans call s+1 while the GPU still runs
 * call s.  mibn_drain waits for every launch and books its time; mibn_total_stats / mibn_total_kernel_stats are
 * the counters of mibn_last_stats / mibn_last_kernel_stats accumulated since the context was created (differences
 * over a region of calls are what bench.py reports).
 */
int mibn_submit_batch(mibn_t *h, int64_t B, const int64_t *q_off, const int32_t *q_vars,
                      const int64_t *e_off, const int32_t *e_vars, const int32_t *e_codes,
                      const int64_t *out_off, double *out, int32_t *ticket);
int mibn_wait(mibn_t *h, int32_t ticket);
int mibn_drain(mibn_t *h);
int mibn_total_stats(const mibn_t *h, mibn_stats *out);
int mibn_total_kernel_stats(const mibn_t *h, int32_t cap, mibn_kernel_stat *out, int32_t *n);

/* Plan only (no device work): fills alg_bytes / alg_flops / n_steps / max_step_cells for one
 * request.  Also usable without a device through a context created by mibn_create_planner(). */
int mibn_plan_stats(mibn_t *h, int32_t n_q, const int32_t *q_vars, int32_t n_e,
                    const int32_t *e_vars, mibn_stats *out);
int mibn_create_planner(mibn_t **out); /* host-only context: set_network/plan_stats work, queries fail */

/* The elimination order the planner executes for one request: order[0 .. *n) = hidden variables, first eliminated
 * first (`order` must hold n_vars entries).  The reference eliminates in Python-set iteration order (bayes_net.py:766,
 * 779); tests hand this order to the CPU oracle so that it can answer the wide requests in seconds instead of minutes. */
int mibn_plan_order(mibn_t *h, int32_t n_q, const int32_t *q_vars, int32_t n_e, const int32_t *e_vars,
                    int32_t *order, int32_t *n);

/* Cheap per-request cost estimate for shard balancing ("balance by the planner's bytes_query, not
 * by count - per-request cost varies 1000x"): cost[b] = section-8(d) bytes of the cheaper of the planner's two sweep
 * orders for request b (no program is emitted, no min-fill search; ~2 us per request and planner thread).  Works on a
 * planner-only context too.  `sorobn_amd.sharding.cost_balanced_ranges` turns the prefix sums into contiguous shards. */
int mibn_estimate_costs(mibn_t *h, int64_t B, const int64_t *q_off, const int32_t *q_vars, const int64_t *e_off,
                        const int32_t *e_vars, double *cost);

/* Wait for everything this context has submitted to its device (kernels, copies, collectives). */
int mibn_device_synchronize(mibn_t *h);

/* Tunables: "arena_gb" (scratch budget), "threads" (planner threads), "chunk" (requests per planning /
 * launch chunk), "plan_cache" (0: plan every request, no plan templates for repeated request shapes), "adaptive" (1: when host planning rather than the GPU bounds a stream
 * of calls, move the elimination-order search to the device - "gpu_search" - or, for networks of more than 128 variables,
 * reserve the min-fill search for ever more expensive requests; given back when the host has slack), "gpu_search" (1: the
 * order search of every chunk but the first of a call runs as a kernel, one request per lane, the same code as on the
 * host: identical orders and programs; 2: every chunk, synchronously - tests), "minfill_above" (bytes of the best sweep order above which the min-fill search runs), "stagger" (groups
 * of requests whose levels are staggered inside a chunk), "tiny" (0: never use the
 * small-network kernel - one lane per request, CPTs in LDS, no planning - that answers blocking calls on networks of at
 * most 32 variables / 4096 CPT cells / 65536 joint states), "sweep" (0..5: the most four-state variables of one big table
 * a single pass may eliminate with the tile resident in LDS - ve_sweep_kernel; below 3: off), "sweep_iters" (tiles per
 * workgroup of that kernel; "sweep_adapt": fewer - down to 2 - in launches of fewer workgroups than this, 0 = off),
 * "order_weights" (0: compare candidate elimination orders by plain section-8(d) bytes instead of the class-weighted
 * model), "order_effort" (default 1: more candidate orders, and the byte model's best two both emitted where the best costs
 * more than "second_above" modelled bytes - the program that moves fewer bytes stays; 0: round 5's search; calls the device
 * plans skip the second emission unless "second_on_device" is 1), "builtin_sweeps" (1: two depth-first topological orders as additional candidates), "streams" (2: consecutive chunks of a call run concurrently on two streams with an arena each), "first_chunk" (short first chunk of a call: 0 never, 1 when the GPU is idle, 2 always).
 * Test and profiling hooks: "sweep_canon" (0: the sweep kernel's general path for every step), "small_cells", "big_iters", "tile_h", "fuse", "chain", "outer" (force the
 * kernels' step forms onto small networks), "split_kinds" (one launch per class of work and level, so that
 * mibn_last_kernel_stats reports per-class rates), "trace" (one stderr line per launch), "gibbs_lds" (0: the Gibbs
 * kernel reads the CPTs through L2 even when they would fit in LDS). */
int mibn_set_option(mibn_t *h, const char *name, double value);

/*
 * Gibbs sampling (replaces BayesNet._gibbs_sampling, bayes_net.py:665-737): n_chains independent
 * chains of n_iterations single-site updates each (the reference's n_iterations counts single-site
 * updates, bayes_net.py:720-733; every iteration is recorded, no burn-in, no thinning).
 *   cycle     update order: every non-evidence variable exactly once (the reference cycles through
 *             sorted(nodes - event), bayes_net.py:697,718); NULL = ascending variable id
 *   counts[prod card(q_vars)]   int64 histogram over the joint query state (C-order over q_vars,
 *             last fastest), summed over all chains and iterations; the estimate of the reference
 *             (bayes_net.py:736-737) is counts / (n_chains * n_iterations)
 * Random stream: counter-based Philox4x32-10 keyed by (seed, chain); statistical parity only (the
 * reference's stream depends on the third-party `vose` sampler, see oracle/README.md).
 */
int mibn_gibbs(mibn_t *h, int32_t n_q, const int32_t *q_vars, int32_t n_e, const int32_t *e_vars,
               const int32_t *e_codes, const int32_t *cycle, int64_t n_chains, int64_t n_iterations,
               uint64_t seed, int64_t *counts);

/* A shard of the chains of one mibn_gibbs call: chains [chain_first, chain_first + n_chains) of the stream `seed` (the
 * Philox key is (seed, global chain index), so the union of disjoint shards - on any number of GPUs - gives bit for bit
 * the histogram of the unsharded call; mibn_gibbs == chain_first 0).  Multi-GPU: every rank runs its chain range and the
 * int64 histograms are summed with mibn_comm_reduce_i64. */
int mibn_gibbs_shard(mibn_t *h, int32_t n_q, const int32_t *q_vars, int32_t n_e, const int32_t *e_vars,
                     const int32_t *e_codes, const int32_t *cycle, int64_t chain_first, int64_t n_chains,
                     int64_t n_iterations, uint64_t seed, int64_t *counts);

/* Parity hook for the deterministic half of the Gibbs path (tests): the Markov-blanket conditionals the chain samples from,
 * bayes_net.py:699-710 - pointwise_mul of the CPTs of a node and its children, normalised per boundary configuration.  For each
 * of n_rows full joint states (states[row * n_vars + v] = label code; evidence entries are overwritten by e_codes) out[row *
 * card(var) + x] = P(var = x | the rest of the row), computed by gibbs_kernel ITSELF: the same update programs, the same
 * LDS-resident tables and the same one of its three update forms a mibn_gibbs call with this network / evidence / cycle and
 * n_rows chains takes - the weights are written out, normalised, where the chain would draw from them.  Only the random
 * stream of the Gibbs path stays unpinned (the reference's depends on the absent third-party `vose` sampler). */
int mibn_gibbs_conditional(mibn_t *h, int32_t n_e, const int32_t *e_vars, const int32_t *e_codes, const int32_t *cycle,
                           int32_t var, int64_t n_rows, const uint8_t *states, double *out);

/*
 * Forward (ancestral) sampling and the two approximate algorithms built on it.
 *   mibn_sample          BayesNet.sample(n, init) / _forward_sample (bayes_net.py:518-575): n_samples joint samples,
 *                        states[sample * n_vars + v] = label code of variable v; `init` variables are clamped
 *   mibn_sampling_query  mode MIBN_REJECTION = _rejection_sampling (577-619): counts[cell] = samples that agree with
 *                        the event, per joint query state; mode MIBN_LIKELIHOOD = _llh_weighting (621-663): the event
 *                        is clamped, weight_sum[cell] = sum of the sample likelihoods (the product of P(value |
 *                        parents) over ALL nodes, as the reference computes it) and counts[cell] = samples per state;
 *                        the reference's answer is (weight_sum / counts) normalised
 * Random stream: Philox4x32-10 keyed by (seed, sample, variable); statistical parity only (see mibn_gibbs).
 */
#define MIBN_REJECTION 1
#define MIBN_LIKELIHOOD 2
int mibn_sample(mibn_t *h, int64_t n_samples, int32_t n_init, const int32_t *init_vars, const int32_t *init_codes,
                uint64_t seed, uint8_t *states);
int mibn_sampling_query(mibn_t *h, int32_t mode, int32_t n_q, const int32_t *q_vars, int32_t n_e, const int32_t *e_vars,
                        const int32_t *e_codes, int64_t n_samples, uint64_t seed, double *weight_sum, int64_t *counts);

/* Parity hook for the deterministic half of the sampling paths (tests): sample_kernel ITSELF walks n_rows GIVEN joint states
 * (states[row * n_vars + v] = label code) exactly as it walks a sample it draws - same CPT offsets, same row sums, same running
 * sums, same likelihood product - and writes them out instead of drawing from them:
 *   likelihood[row]                        the product of P(state_v | parents in the state) over ALL nodes, the weight
 *                                          _forward_sample hands to _llh_weighting (bayes_net.py:541-546, 646-652)
 *   cdf[(row * n_vars + v) * cdf_stride + x]   x < card(v): the running sum of the conditional row P(v = . | parents in the state)
 *                                          the inverse-CDF draw of v compares u * total with (total = the entry at card(v) - 1) -
 *                                          the weights the reference hands to its alias sampler (bayes_net.py:28-42, 530-539)
 * Only the random stream of the sampling paths stays unpinned (the reference's depends on the absent third-party `vose`). */
int mibn_sample_probe(mibn_t *h, int64_t n_rows, const uint8_t *states, int32_t cdf_stride, double *likelihood, double *cdf);

/*
 * Grouped counting of label codes: the `X.groupby([*parents, node]).size()` of
 * BayesNet.partial_fit (bayes_net.py:467-510) and the pairwise `X.groupby([u, v]).size()` of structure.chow_liu
 * (structure.py:33-45).  No network needed.
 *   codes[n_cols * n_rows]    label codes, code < card[col] <= 256; row_major = 0: codes[col * n_rows + row],
 *                             row_major = 1: codes[row * n_cols + col] (what DataFrame.to_numpy() gives; transposed on
 *                             the device)
 *   scope_off[n_tables + 1], scope_cols[]   CSR: the columns of table t
 *   counts_off[n_tables + 1], counts[]      dense C-order contingency table of every table (last column fastest);
 *                             tables of up to 16384 cells are counted in LDS histograms, larger ones (up to 2^28
 *                             cells) with global atomics
 */
int mibn_count_tables(mibn_t *h, int64_t n_rows, int32_t n_cols, const uint8_t *codes, int32_t row_major, const int32_t *card,
                      int32_t n_tables, const int64_t *scope_off, const int32_t *scope_cols, const int64_t *counts_off,
                      int64_t *counts);

/*
 * Multi-GPU: one process per GPU, requests / chains sharded with no data-path collective; the
 * only communication is the final gather of the posteriors (exact path) or the sum of the histograms (Gibbs).  These
 * entry points sit directly on RCCL (librccl.so is dlopen'ed by mibn_comm_init - a single-GPU process never loads it) and
 * run on a stream of their own (the gather of batch s must not queue behind the kernels of batch s + 1), over xGMI between the GPUs of a node.  No PyTorch involved.
 *   mibn_comm_probe       loads librccl.so and resolves its entry points, nothing else: what can fail on one rank alone is
 *                         checked on EVERY rank before anybody enters the collective init
 *   mibn_comm_unique_id   rank 0 (only) creates the 128-byte RCCL id; the caller hands it to the other ranks out of band
 *                         (sorobn_amd/sharding.py: a file next to the rendezvous port, single node)
 *   mibn_comm_init        collective: every rank calls it with the same id.  Bounded: a rank that has waited
 *                         MIBN_COMM_INIT_TIMEOUT_S seconds (environment, default 180) for the others returns MIBN_E_COMM with
 *                         a message naming the likely causes instead of hanging the launch
 *   mibn_device_info      one text line about this context's device (name, PCI bus id, link type / hops to every other
 *                         visible device) for the launch log
 *   mibn_comm_count       ncclCommCount / ncclCommUserRank of the live communicator: what RCCL itself reports, for the
 *                         bench line of an N > 1 run (`config.rccl_ranks`)
 *   mibn_comm_allgather_f64   recv[r * n .. (r+1) * n) = rank r's send[0 .. n)   (host buffers, staged through HBM)
 *   mibn_comm_reduce_i64      sum over ranks of buf[0 .. n) -> buf on `root` (other ranks' buf unchanged)
 *   mibn_comm_allreduce_max_f64   element-wise max over ranks, in place (the bench's max-over-ranks step time)
 *   mibn_comm_barrier     all ranks have reached the call and their device work is complete
 */
#define MIBN_COMM_ID_BYTES 128
int mibn_comm_probe(mibn_t *h);
int mibn_device_info(mibn_t *h, char *buf, int32_t cap);
int mibn_comm_unique_id(mibn_t *h, void *id_out /* MIBN_COMM_ID_BYTES */);
int mibn_comm_init(mibn_t *h, int32_t rank, int32_t world, const void *id /* MIBN_COMM_ID_BYTES */);
int mibn_comm_count(mibn_t *h, int32_t *n_ranks, int32_t *my_rank);
int mibn_comm_destroy(mibn_t *h);
int mibn_comm_allgather_f64(mibn_t *h, const double *send, int64_t n, double *recv);
int mibn_comm_reduce_i64(mibn_t *h, int64_t *buf, int64_t n, int32_t root);
int mibn_comm_allreduce_max_f64(mibn_t *h, double *buf, int64_t n);
int mibn_comm_barrier(mibn_t *h);

#ifdef __cplusplus
}
#endif
#endif /* MIBN_H */
