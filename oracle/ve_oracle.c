/*
 * TEST INFRASTRUCTURE - CPU restatement ("oracle") of the reference's exact-inference hot path.
 *
 * This file restates, in plain C, the algorithm of MaxHalford/sorobn's
 *   CDTAccessor.sum_out            sorobn/bayes_net.py:100-103
 *   pointwise_mul_two              sorobn/bayes_net.py:233-250
 *   pointwise_mul                  sorobn/bayes_net.py:253-256
 *   BayesNet.ancestors             sorobn/bayes_net.py:373-378
 *   BayesNet._variable_elimination sorobn/bayes_net.py:739-794
 *   (result ordering of) query     sorobn/bayes_net.py:869-875
 * with the same data model as the reference: a factor is a SPARSE table of rows
 * (label codes..., value) - the counterpart of a pandas Series with a named MultiIndex - NOT the
 * dense strided tensors of the HIP product path.  It shares no code and no data layout with
 * sorobn_amd/csrc, which is the point: it is the independent checker.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library.
 * The product path never calls it.
 *
 * Pinning: tests/test_oracle.py checks this restatement against every golden vector generated
 * from the unmodified reference (tests/golden/make_golden.py) and against the factor-algebra
 * doctest vectors of bayes_net.py:62-97,114-229 (AIMA fig. 14.10).
 *
 * Elimination order: the reference iterates a Python set (bayes_net.py:766,779), i.e. an arbitrary
 * hash-dependent order; the oracle takes the order as an explicit priority array (default ascending
 * variable id, which is what the hash-ordered node names of oracle/refload.py force on the
 * reference).
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define VE_MAX_SCOPE 64
#define VE_ERR_ALLOC (-1)
#define VE_ERR_CAP (-2)
#define VE_ERR_KEYSPACE (-3)
#define VE_ERR_ARG (-4)

typedef struct {
    int nv;
    int32_t vars[VE_MAX_SCOPE];
    int64_t n;       /* rows */
    uint8_t *codes;  /* n * nv */
    double *vals;    /* n */
} factor;

typedef struct ve_net {
    int n_vars;
    int32_t *card;
    factor *cpt;      /* one per variable; scope = [*parents, var] */
    double last_product_rows; /* statistics of the last ve_query */
    double last_max_rows;
} ve_net;

static void factor_free(factor *f) {
    free(f->codes);
    free(f->vals);
    f->codes = NULL;
    f->vals = NULL;
    f->n = 0;
}

static int factor_alloc(factor *f, int nv, int64_t n) {
    f->nv = nv;
    f->n = n;
    f->codes = (uint8_t *)malloc((size_t)(n > 0 ? n : 1) * (size_t)(nv > 0 ? nv : 1));
    f->vals = (double *)malloc(sizeof(double) * (size_t)(n > 0 ? n : 1));
    return (f->codes && f->vals) ? 0 : VE_ERR_ALLOC;
}

static int find_var(const factor *f, int32_t v) {
    for (int i = 0; i < f->nv; ++i)
        if (f->vars[i] == v) return i;
    return -1;
}

/* cdt[cdt > 0] - bayes_net.py:255 */
static int drop_nonpositive(const factor *in, factor *out) {
    int64_t keep = 0;
    for (int64_t i = 0; i < in->n; ++i) keep += in->vals[i] > 0;
    if (factor_alloc(out, in->nv, keep)) return VE_ERR_ALLOC;
    memcpy(out->vars, in->vars, sizeof(int32_t) * in->nv);
    int64_t k = 0;
    for (int64_t i = 0; i < in->n; ++i)
        if (in->vals[i] > 0) {
            memcpy(out->codes + k * in->nv, in->codes + i * in->nv, in->nv);
            out->vals[k++] = in->vals[i];
        }
    return 0;
}

/* factor[level(var) == code] keeping the level - bayes_net.py:772-774 */
static int filter_level(factor *f, int pos, int code) {
    int64_t k = 0;
    for (int64_t i = 0; i < f->n; ++i)
        if (f->codes[i * f->nv + pos] == code) {
            if (k != i) {
                memmove(f->codes + k * f->nv, f->codes + i * f->nv, f->nv);
                f->vals[k] = f->vals[i];
            }
            ++k;
        }
    f->n = k;
    return 0;
}

typedef struct {
    int64_t key;
    int64_t row;
} keyrow;

static int cmp_keyrow(const void *a, const void *b) {
    const keyrow *x = (const keyrow *)a, *y = (const keyrow *)b;
    if (x->key != y->key) return x->key < y->key ? -1 : 1;
    return x->row < y->row ? -1 : (x->row > y->row);
}

/* pointwise_mul_two - bayes_net.py:233-250.
 * Disjoint scopes: outer product, left rows major (np.outer + stack).  Otherwise inner join on the
 * shared level names; result levels = left's levels followed by right's new ones. */
static int mul_two(const int32_t *card, const factor *L, const factor *R, factor *out) {
    int ls[VE_MAX_SCOPE], rs[VE_MAX_SCOPE], ns = 0; /* positions of shared vars */
    int rnew[VE_MAX_SCOPE], nnew = 0;
    for (int j = 0; j < R->nv; ++j) {
        int p = find_var(L, R->vars[j]);
        if (p >= 0) {
            ls[ns] = p;
            rs[ns] = j;
            ++ns;
        } else
            rnew[nnew++] = j;
    }
    int nv = L->nv + nnew;
    if (nv > VE_MAX_SCOPE) return VE_ERR_ARG;
    int64_t total = 0;
    int64_t *start = NULL;   /* per left row: first match in `order` */
    int64_t *count = NULL;
    keyrow *order = NULL;
    if (ns == 0) {
        total = L->n * R->n;
    } else {
        order = (keyrow *)malloc(sizeof(keyrow) * (size_t)(R->n > 0 ? R->n : 1));
        start = (int64_t *)malloc(sizeof(int64_t) * (size_t)(L->n > 0 ? L->n : 1));
        count = (int64_t *)malloc(sizeof(int64_t) * (size_t)(L->n > 0 ? L->n : 1));
        if (!order || !start || !count) return VE_ERR_ALLOC;
        for (int64_t j = 0; j < R->n; ++j) {
            int64_t key = 0;
            for (int s = 0; s < ns; ++s)
                key = key * card[R->vars[rs[s]]] + R->codes[j * R->nv + rs[s]];
            order[j].key = key;
            order[j].row = j;
        }
        qsort(order, (size_t)R->n, sizeof(keyrow), cmp_keyrow);
        for (int64_t i = 0; i < L->n; ++i) {
            int64_t key = 0;
            for (int s = 0; s < ns; ++s)
                key = key * card[L->vars[ls[s]]] + L->codes[i * L->nv + ls[s]];
            int64_t lo = 0, hi = R->n; /* lower bound */
            while (lo < hi) {
                int64_t mid = (lo + hi) >> 1;
                if (order[mid].key < key) lo = mid + 1; else hi = mid;
            }
            int64_t e = lo;
            while (e < R->n && order[e].key == key) ++e;
            start[i] = lo;
            count[i] = e - lo;
            total += e - lo;
        }
    }
    if (factor_alloc(out, nv, total)) return VE_ERR_ALLOC;
    memcpy(out->vars, L->vars, sizeof(int32_t) * L->nv);
    for (int k = 0; k < nnew; ++k) out->vars[L->nv + k] = R->vars[rnew[k]];
    int64_t o = 0;
    for (int64_t i = 0; i < L->n; ++i) {
        int64_t c = ns ? count[i] : R->n;
        for (int64_t m = 0; m < c; ++m) {
            int64_t j = ns ? order[start[i] + m].row : m;
            uint8_t *dst = out->codes + o * nv;
            memcpy(dst, L->codes + i * L->nv, L->nv);
            for (int k = 0; k < nnew; ++k) dst[L->nv + k] = R->codes[j * R->nv + rnew[k]];
            out->vals[o++] = L->vals[i] * R->vals[j];
        }
    }
    free(order);
    free(start);
    free(count);
    return 0;
}

/* sum_out - bayes_net.py:100-103: groupby(remaining levels).sum(); pandas' group_sum accumulates
 * with Kahan compensation in row order and returns groups sorted by key; only groups that occur
 * are present. */
static int sum_out(const int32_t *card, const factor *in, int32_t x, factor *out) {
    int px = find_var(in, x);
    if (px < 0) return VE_ERR_ARG;
    int nv = in->nv - 1;
    int pos[VE_MAX_SCOPE];
    /* group keys: mixed radix over the code range every remaining level actually takes in this table (an evidence level
     * holds a single value, bayes_net.py:772-774, and would otherwise multiply the key space by its full cardinality).
     * The mapping is monotone per level, so the groups still come out sorted by their level codes. */
    int lo[VE_MAX_SCOPE], radix[VE_MAX_SCOPE];
    for (int i = 0, k = 0; i < in->nv; ++i)
        if (i != px) pos[k++] = i;
    for (int k = 0; k < nv; ++k) {
        int mn = 255, mx = 0;
        for (int64_t i = 0; i < in->n; ++i) {
            const int c = in->codes[i * in->nv + pos[k]];
            if (c < mn) mn = c;
            if (c > mx) mx = c;
        }
        if (in->n == 0) mn = mx = 0;
        lo[k] = mn;
        radix[k] = mx - mn + 1;
    }
    int64_t keyspace = 1;
    for (int k = 0; k < nv; ++k) {
        keyspace *= radix[k];
        if (keyspace > ((int64_t)1 << 29)) return VE_ERR_KEYSPACE;
    }
    (void)card;
    double *sum = (double *)calloc((size_t)keyspace, sizeof(double));
    double *comp = (double *)calloc((size_t)keyspace, sizeof(double));
    uint8_t *seen = (uint8_t *)calloc((size_t)keyspace, 1);
    if (!sum || !comp || !seen) return VE_ERR_ALLOC;
    int64_t groups = 0;
    for (int64_t i = 0; i < in->n; ++i) {
        int64_t key = 0;
        for (int k = 0; k < nv; ++k)
            key = key * radix[k] + (in->codes[i * in->nv + pos[k]] - lo[k]);
        if (!seen[key]) {
            seen[key] = 1;
            ++groups;
        }
        double y = in->vals[i] - comp[key];
        double t = sum[key] + y;
        comp[key] = (t - sum[key]) - y;
        sum[key] = t;
    }
    if (factor_alloc(out, nv, groups)) return VE_ERR_ALLOC;
    for (int k = 0; k < nv; ++k) out->vars[k] = in->vars[pos[k]];
    int64_t o = 0;
    for (int64_t key = 0; key < keyspace; ++key)
        if (seen[key]) {
            int64_t r = key;
            for (int k = nv - 1; k >= 0; --k) {
                out->codes[o * nv + k] = (uint8_t)(lo[k] + r % radix[k]);
                r /= radix[k];
            }
            out->vals[o++] = sum[key];
        }
    free(sum);
    free(comp);
    free(seen);
    return 0;
}

/* ------------------------------------------------------------------------------ public API */

ve_net *ve_net_create(int n_vars, const int32_t *card) {
    ve_net *net = (ve_net *)calloc(1, sizeof(ve_net));
    if (!net) return NULL;
    net->n_vars = n_vars;
    net->card = (int32_t *)malloc(sizeof(int32_t) * (size_t)(n_vars > 0 ? n_vars : 1));
    net->cpt = (factor *)calloc((size_t)(n_vars > 0 ? n_vars : 1), sizeof(factor));
    memcpy(net->card, card, sizeof(int32_t) * (size_t)n_vars);
    return net;
}

void ve_net_destroy(ve_net *net) {
    if (!net) return;
    for (int i = 0; i < net->n_vars; ++i) factor_free(&net->cpt[i]);
    free(net->cpt);
    free(net->card);
    free(net);
}

/* CPT of `var` as sparse rows: scope = vars[0..nv-1] (parents..., var), codes[nrows*nv] */
int ve_net_set_cpt(ve_net *net, int var, int nv, const int32_t *vars, int64_t nrows,
                   const int32_t *codes, const double *vals) {
    if (var < 0 || var >= net->n_vars || nv < 1 || nv > VE_MAX_SCOPE || vars[nv - 1] != var)
        return VE_ERR_ARG;
    factor *f = &net->cpt[var];
    factor_free(f);
    if (factor_alloc(f, nv, nrows)) return VE_ERR_ALLOC;
    memcpy(f->vars, vars, sizeof(int32_t) * nv);
    for (int64_t i = 0; i < nrows * nv; ++i) f->codes[i] = (uint8_t)codes[i];
    memcpy(f->vals, vals, sizeof(double) * (size_t)nrows);
    return 0;
}

static void mark_ancestors(const ve_net *net, int v, uint8_t *rel) {
    /* bayes_net.py:373-378 (recursive union of parents), memoised through `rel` */
    const factor *f = &net->cpt[v];
    for (int i = 0; i + 1 < f->nv; ++i) {
        int p = f->vars[i];
        if (!rel[p]) {
            rel[p] = 1;
            mark_ancestors(net, p, rel);
        }
    }
}

static int factor_copy(const factor *in, factor *out) {
    if (factor_alloc(out, in->nv, in->n)) return VE_ERR_ALLOC;
    memcpy(out->vars, in->vars, sizeof(int32_t) * in->nv);
    memcpy(out->codes, in->codes, (size_t)in->n * in->nv);
    memcpy(out->vals, in->vals, sizeof(double) * (size_t)in->n);
    return 0;
}

/* pointwise_mul(list) = reduce(pointwise_mul_two, (f[f > 0] for f in list)) - bayes_net.py:253-256 */
static int mul_many(ve_net *net, factor *fs, int n, factor *out) {
    factor acc;
    int rc = drop_nonpositive(&fs[0], &acc);
    if (rc) return rc;
    for (int i = 1; i < n; ++i) {
        factor r, prod;
        if ((rc = drop_nonpositive(&fs[i], &r))) return rc;
        if ((rc = mul_two(net->card, &acc, &r, &prod))) return rc;
        net->last_product_rows += (double)prod.n;
        if ((double)prod.n > net->last_max_rows) net->last_max_rows = (double)prod.n;
        factor_free(&acc);
        factor_free(&r);
        acc = prod;
    }
    *out = acc;
    return 0;
}

/*
 * _variable_elimination + the ordering part of query() - bayes_net.py:739-794, 869-875.
 * order: priority per variable (lower = eliminated earlier) or NULL for ascending id.
 * Output rows: codes over qvars in the caller's qvars order, rows sorted lexicographically.
 * Returns the number of rows (0 = empty posterior) or a negative error.
 */
int64_t ve_query(ve_net *net, int nq, const int32_t *qvars, int ne, const int32_t *evars,
                 const int32_t *ecodes, const int32_t *order, int64_t cap, int32_t *out_codes,
                 double *out_vals) {
    int nvars = net->n_vars;
    net->last_product_rows = 0;
    net->last_max_rows = 0;
    uint8_t *rel = (uint8_t *)calloc((size_t)nvars, 1);
    uint8_t *special = (uint8_t *)calloc((size_t)nvars, 1);
    if (!rel || !special) return VE_ERR_ALLOC;
    for (int i = 0; i < nq; ++i) { rel[qvars[i]] = 1; special[qvars[i]] = 1; }
    for (int i = 0; i < ne; ++i) { rel[evars[i]] = 1; special[evars[i]] = 2; }
    for (int i = 0; i < nq; ++i) mark_ancestors(net, qvars[i], rel);
    for (int i = 0; i < ne; ++i) mark_ancestors(net, evars[i], rel);

    /* factors = evidence-filtered copies of the relevant CPTs (bayes_net.py:768-776) */
    factor *fs = (factor *)calloc((size_t)nvars + 1, sizeof(factor));
    int nf = 0, rc = 0;
    for (int v = 0; v < nvars; ++v) {
        if (!rel[v]) continue;
        if ((rc = factor_copy(&net->cpt[v], &fs[nf]))) return rc;
        for (int e = 0; e < ne; ++e) {
            int p = find_var(&fs[nf], evars[e]);
            if (p >= 0) filter_level(&fs[nf], p, ecodes[e]);
        }
        ++nf;
    }
    /* hidden variables in priority order (bayes_net.py:779) */
    int32_t *hid = (int32_t *)malloc(sizeof(int32_t) * (size_t)(nvars > 0 ? nvars : 1));
    int nh = 0;
    for (int v = 0; v < nvars; ++v)
        if (rel[v] && !special[v]) hid[nh++] = v;
    if (order)
        for (int i = 1; i < nh; ++i) { /* insertion sort by priority, stable */
            int32_t v = hid[i];
            int j = i - 1;
            while (j >= 0 && order[hid[j]] > order[v]) { hid[j + 1] = hid[j]; --j; }
            hid[j + 1] = v;
        }
    factor *tmp = (factor *)calloc((size_t)nvars + 1, sizeof(factor));
    for (int h = 0; h < nh; ++h) {
        int32_t x = hid[h];
        int nt = 0;
        /* pop factors mentioning x, last to first (bayes_net.py:780-784) */
        for (int i = nf - 1; i >= 0; --i)
            if (find_var(&fs[i], x) >= 0) {
                tmp[nt++] = fs[i];
                memmove(&fs[i], &fs[i + 1], sizeof(factor) * (size_t)(nf - i - 1));
                --nf;
            }
        factor prod, summed;
        if ((rc = mul_many(net, tmp, nt, &prod))) return rc;
        for (int i = 0; i < nt; ++i) factor_free(&tmp[i]);
        if ((rc = sum_out(net->card, &prod, x, &summed))) return rc;
        factor_free(&prod);
        fs[nf++] = summed;
    }
    /* posterior = pointwise_mul(factors); normalise; drop evidence levels (bayes_net.py:789-793) */
    factor post;
    if ((rc = mul_many(net, fs, nf, &post))) return rc;
    for (int i = 0; i < nf; ++i) factor_free(&fs[i]);
    double total = 0, comp = 0;
    for (int64_t i = 0; i < post.n; ++i) {
        double y = post.vals[i] - comp, t = total + y;
        comp = (t - total) - y;
        total = t;
    }
    int qpos[VE_MAX_SCOPE];
    for (int i = 0; i < nq; ++i) {
        qpos[i] = find_var(&post, qvars[i]);
        if (qpos[i] < 0 && post.n > 0) return VE_ERR_ARG;
    }
    int64_t n = post.n;
    if (n > cap) return VE_ERR_CAP;
    /* sort_index (bayes_net.py:875): rows sorted by the query levels in the caller's order */
    keyrow *kr = (keyrow *)malloc(sizeof(keyrow) * (size_t)(n > 0 ? n : 1));
    for (int64_t i = 0; i < n; ++i) {
        int64_t key = 0;
        for (int k = 0; k < nq; ++k)
            key = key * net->card[qvars[k]] + post.codes[i * post.nv + qpos[k]];
        kr[i].key = key;
        kr[i].row = i;
    }
    qsort(kr, (size_t)n, sizeof(keyrow), cmp_keyrow);
    for (int64_t o = 0; o < n; ++o) {
        int64_t i = kr[o].row;
        for (int k = 0; k < nq; ++k) out_codes[o * nq + k] = post.codes[i * post.nv + qpos[k]];
        out_vals[o] = post.vals[i] / total;
    }
    free(kr);
    factor_free(&post);
    free(fs);
    free(tmp);
    free(hid);
    free(rel);
    free(special);
    return n;
}

void ve_last_stats(const ve_net *net, double *product_rows, double *max_rows) {
    *product_rows = net->last_product_rows;
    *max_rows = net->last_max_rows;
}

/* --- raw factor algebra, exported so the doctest vectors of bayes_net.py:62-229 can be replayed */

static int load_factor(factor *f, int nv, const int32_t *vars, int64_t n, const int32_t *codes,
                       const double *vals) {
    if (factor_alloc(f, nv, n)) return VE_ERR_ALLOC;
    memcpy(f->vars, vars, sizeof(int32_t) * nv);
    for (int64_t i = 0; i < n * nv; ++i) f->codes[i] = (uint8_t)codes[i];
    memcpy(f->vals, vals, sizeof(double) * (size_t)n);
    return 0;
}

static int64_t store_factor(factor *f, int64_t cap, int32_t *out_nv, int32_t *out_vars,
                            int32_t *out_codes, double *out_vals) {
    if (f->n > cap) return VE_ERR_CAP;
    *out_nv = f->nv;
    for (int i = 0; i < f->nv; ++i) out_vars[i] = f->vars[i];
    for (int64_t i = 0; i < f->n * f->nv; ++i) out_codes[i] = f->codes[i];
    memcpy(out_vals, f->vals, sizeof(double) * (size_t)f->n);
    int64_t n = f->n;
    factor_free(f);
    return n;
}

int64_t ve_pointwise_mul_two(const int32_t *card, int nvL, const int32_t *varsL, int64_t nL,
                             const int32_t *codesL, const double *valsL, int nvR,
                             const int32_t *varsR, int64_t nR, const int32_t *codesR,
                             const double *valsR, int64_t cap, int32_t *out_nv, int32_t *out_vars,
                             int32_t *out_codes, double *out_vals) {
    factor L, R, P;
    if (load_factor(&L, nvL, varsL, nL, codesL, valsL)) return VE_ERR_ALLOC;
    if (load_factor(&R, nvR, varsR, nR, codesR, valsR)) return VE_ERR_ALLOC;
    int rc = mul_two(card, &L, &R, &P);
    factor_free(&L);
    factor_free(&R);
    if (rc) return rc;
    return store_factor(&P, cap, out_nv, out_vars, out_codes, out_vals);
}

int64_t ve_sum_out(const int32_t *card, int nv, const int32_t *vars, int64_t n,
                   const int32_t *codes, const double *vals, int32_t x, int64_t cap,
                   int32_t *out_nv, int32_t *out_vars, int32_t *out_codes, double *out_vals) {
    factor F, S;
    if (load_factor(&F, nv, vars, n, codes, vals)) return VE_ERR_ALLOC;
    int rc = sum_out(card, &F, x, &S);
    factor_free(&F);
    if (rc) return rc;
    return store_factor(&S, cap, out_nv, out_vars, out_codes, out_vals);
}
