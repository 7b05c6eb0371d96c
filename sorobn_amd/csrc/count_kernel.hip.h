// Grouped counting of label codes on gfx950 - the arithmetic under the parameter- and structure-learning rows of
// `X.groupby([*parents, node]).size()` of BayesNet.partial_fit (sorobn/bayes_net.py:467-510,
// rank 3) and the pairwise `X.groupby([u, v]).size()` / `value_counts` of structure.chow_liu (structure.py:33-45,
// rank 4).  Input: a column-major matrix of label codes (uint8, codes[col * n_rows + row]) and a list of *tables*,
// each a tuple of columns; output: for every table the dense C-order contingency table (last column fastest).
//
// The tables are packed into groups whose cells fit one workgroup's LDS histogram (64 KiB of uint32).  A workgroup
// (group g, row block b) walks its rows - consecutive lanes read consecutive rows of a column, 256 B per wave load,
// the columns stay in L1 across the tables of the group - adds into LDS with ds_add_u32, and flushes its non-zero
// cells with one 64-bit global atomic each.  Bound by LDS atomics (one per row and table), not by HBM: the code
// matrix is read once per group.
#pragma once
#include <hip/hip_runtime.h>

#include <string>
#include <vector>

#include "../../include/mibn.h"

namespace mibn {

constexpr int kCountLdsCells = 16384;  // 64 KiB of uint32 per workgroup

struct CountArgs {
    const uint8_t *codes;        // [n_cols][n_rows]
    const int32_t *tbl_begin;    // per table: first entry in tbl_col / tbl_stride
    const int32_t *tbl_col;      // flattened scopes
    const int32_t *tbl_stride;   // C-order strides (cells) of every scope entry
    const int32_t *tbl_lds;      // per table: offset inside its group's LDS histogram
    const int64_t *tbl_out;      // per table: offset in `counts`
    const int32_t *grp_begin;    // per group: [first table, last table)
    unsigned long long *counts;
    int64_t n_rows;
    int32_t n_groups;
};

__global__ __launch_bounds__(256) void count_kernel(const CountArgs A) {
    __shared__ unsigned int hist[kCountLdsCells];
    const int g = blockIdx.y;
    const int t0 = A.grp_begin[g], t1 = A.grp_begin[g + 1];
    const int cells = A.tbl_lds[t1] - A.tbl_lds[t0];  // (tbl_lds has one entry past the last table, cumulative per group)
    for (int i = threadIdx.x; i < cells; i += 256) hist[i] = 0u;
    __syncthreads();
    const int base = A.tbl_lds[t0];
    for (int64_t row = (int64_t)blockIdx.x * 256 + threadIdx.x; row < A.n_rows; row += (int64_t)gridDim.x * 256) {
        for (int t = t0; t < t1; ++t) {
            int cell = A.tbl_lds[t] - base;
            for (int k = A.tbl_begin[t]; k < A.tbl_begin[t + 1]; ++k)
                cell += (int)A.codes[(int64_t)A.tbl_col[k] * A.n_rows + row] * A.tbl_stride[k];
            atomicAdd(&hist[cell], 1u);
        }
    }
    __syncthreads();
    for (int t = t0; t < t1; ++t) {
        const int lo = A.tbl_lds[t] - base, n = A.tbl_lds[t + 1] - A.tbl_lds[t];
        for (int i = threadIdx.x; i < n; i += 256)
            if (hist[lo + i]) atomicAdd(&A.counts[A.tbl_out[t] + i], (unsigned long long)hist[lo + i]);
    }
}

// Tables above kCountLdsCells cells (a 4-state child with 7 parents, two 200-label columns, ...) do not fit one
// workgroup's LDS histogram: they are counted straight into HBM, one 64-bit global atomic per row and table
// (blockIdx.y = table).  Rare and atomic-bound; the LDS path above stays the one that matters.
struct CountBigArgs {
    const uint8_t *codes;
    const int32_t *tbl_begin, *tbl_col;
    const int64_t *tbl_stride, *tbl_out;
    unsigned long long *counts;
    int64_t n_rows;
};

__global__ __launch_bounds__(256) void count_big_kernel(const CountBigArgs A) {
    const int t = blockIdx.y;
    const int k0 = A.tbl_begin[t], k1 = A.tbl_begin[t + 1];
    for (int64_t row = (int64_t)blockIdx.x * 256 + threadIdx.x; row < A.n_rows; row += (int64_t)gridDim.x * 256) {
        int64_t cell = A.tbl_out[t];
        for (int k = k0; k < k1; ++k) cell += (int64_t)A.codes[(int64_t)A.tbl_col[k] * A.n_rows + row] * A.tbl_stride[k];
        atomicAdd(&A.counts[cell], 1ull);
    }
}

// row-major code matrix [n_rows][n_cols] -> the column-major one the count kernel reads, 64 x 64 tiles through LDS
// (a DataFrame's to_numpy() is row-major: transposing 100 MB here takes microseconds, on the host ~0.1 s)
__global__ __launch_bounds__(256) void transpose_codes_kernel(const uint8_t *__restrict__ in, uint8_t *__restrict__ out, int64_t n_rows,
                                                              int32_t n_cols) {
    __shared__ uint8_t tile[64][65];
    const int64_t r0 = (int64_t)blockIdx.x * 64;
    const int c0 = blockIdx.y * 64;
    for (int i = threadIdx.x; i < 64 * 64; i += 256) {
        const int r = i >> 6, c = i & 63;
        if (r0 + r < n_rows && c0 + c < n_cols) tile[r][c] = in[(r0 + r) * n_cols + c0 + c];
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 64 * 64; i += 256) {
        const int c = i >> 6, r = i & 63;
        if (r0 + r < n_rows && c0 + c < n_cols) out[(int64_t)(c0 + c) * n_rows + r0 + r] = tile[r][c];
    }
}

// host driver; returns MIBN_* code.  row_major: codes[row * n_cols + col] instead of codes[col * n_rows + row].  scope_off[n_tables + 1] / scope_cols: CSR list of the tables' columns;
// counts_off[n_tables + 1]: offsets of the dense tables in `counts` (must equal the running product of cards).
inline int count_run(hipStream_t stream, int64_t n_rows, int32_t n_cols, const uint8_t *codes, bool row_major, const int32_t *card, int32_t n_tables,
                     const int64_t *scope_off, const int32_t *scope_cols, const int64_t *counts_off, int64_t *counts,
                     std::string &err) {
    // small tables (<= kCountLdsCells cells) are packed into LDS groups, in their original order; the others take the
    // global-atomic kernel
    constexpr int64_t kCountMaxCells = 1ll << 28;  // 2 GiB of int64 counts per table
    std::vector<int32_t> small, bigs;
    std::vector<int32_t> big_begin{0}, big_col;
    std::vector<int64_t> big_stride, big_out;
    std::vector<std::vector<int64_t>> strides_of((size_t)n_tables);
    for (int t = 0; t < n_tables; ++t) {
        int64_t cells = 1;
        const int64_t a = scope_off[t], b = scope_off[t + 1];
        strides_of[(size_t)t].resize((size_t)(b - a));
        for (int64_t k = b - 1; k >= a; --k) {
            const int c = scope_cols[k];
            if (c < 0 || c >= n_cols) { err = "count: unknown column"; return MIBN_E_ARG; }
            if (card[c] < 1 || card[c] > 256) { err = "count: cardinality outside 1..256"; return MIBN_E_LIMIT; }
            strides_of[(size_t)t][(size_t)(k - a)] = cells;
            cells *= card[c];
            if (cells > kCountMaxCells) { err = "count: a table has more than " + std::to_string(kCountMaxCells) + " cells"; return MIBN_E_LIMIT; }
        }
        if (counts_off[t + 1] - counts_off[t] != cells) { err = "count: counts_off does not match the table sizes"; return MIBN_E_ARG; }
        if (cells <= kCountLdsCells) {
            small.push_back(t);
        } else {
            bigs.push_back(t);
            for (int64_t k = a; k < b; ++k) { big_col.push_back(scope_cols[k]); big_stride.push_back(strides_of[(size_t)t][(size_t)(k - a)]); }
            big_begin.push_back((int32_t)big_col.size());
            big_out.push_back(counts_off[t]);
        }
    }
    const int n_small = (int)small.size();
    std::vector<int32_t> tbl_begin(n_small + 1), tbl_col, tbl_stride, tbl_lds(n_small + 1), grp_begin{0};
    std::vector<int64_t> tbl_out((size_t)n_small);
    int32_t in_group = 0;
    for (int i = 0; i < n_small; ++i) {
        const int t = small[(size_t)i];
        tbl_begin[i] = (int32_t)tbl_col.size();
        const int64_t a = scope_off[t], b = scope_off[t + 1];
        const int64_t cells = counts_off[t + 1] - counts_off[t];
        for (int64_t k = a; k < b; ++k) { tbl_col.push_back(scope_cols[k]); tbl_stride.push_back((int32_t)strides_of[(size_t)t][(size_t)(k - a)]); }
        if (in_group + cells > kCountLdsCells) { grp_begin.push_back(i); in_group = 0; }
        in_group += (int32_t)cells;
        tbl_out[(size_t)i] = counts_off[t];
    }
    tbl_begin[n_small] = (int32_t)tbl_col.size();
    grp_begin.push_back(n_small);
    // cumulative LDS offsets: tbl_lds[i] grows monotonically across groups so that tbl_lds[i + 1] - tbl_lds[i] is
    // always the table's size; a group's base is the offset of its first table
    {
        int64_t run = 0;
        for (int i = 0; i < n_small; ++i) {
            const int t = small[(size_t)i];
            tbl_lds[i] = (int32_t)run;
            run += counts_off[t + 1] - counts_off[t];
            if (run >= (1ll << 31)) { err = "count: too many cells in total"; return MIBN_E_LIMIT; }
        }
        tbl_lds[n_small] = (int32_t)run;
    }
    const int n_groups = (int)grp_begin.size() - 1;
    const int64_t total = counts_off[n_tables] - counts_off[0];
    uint8_t *d_codes = nullptr, *d_rows = nullptr;
    int32_t *d_i32 = nullptr;
    int64_t *d_i64 = nullptr;
    unsigned long long *d_counts = nullptr;
    std::vector<int32_t> pack;
    auto put = [&](const std::vector<int32_t> &a) { size_t o = pack.size(); pack.insert(pack.end(), a.begin(), a.end()); return o; };
    const size_t o_tb = put(tbl_begin), o_tc = put(tbl_col), o_ts = put(tbl_stride), o_tl = put(tbl_lds), o_gb = put(grp_begin);
    const size_t o_bb = put(big_begin), o_bc = put(big_col);
    std::vector<int64_t> pack64(tbl_out);
    const size_t o_bs = pack64.size();
    pack64.insert(pack64.end(), big_stride.begin(), big_stride.end());
    const size_t o_bo = pack64.size();
    pack64.insert(pack64.end(), big_out.begin(), big_out.end());
    auto fail = [&](hipError_t e) { err = std::string("count: ") + hipGetErrorString(e); hipFree(d_codes); hipFree(d_rows); hipFree(d_i32); hipFree(d_i64); hipFree(d_counts); return MIBN_E_HIP; };
    hipError_t e;
    const size_t code_bytes = (size_t)n_rows * (size_t)n_cols;
    if ((e = hipMalloc(&d_codes, std::max<size_t>(16, code_bytes))) != hipSuccess) return fail(e);
    if ((e = hipMalloc(&d_i32, 4 * std::max<size_t>(1, pack.size()))) != hipSuccess) return fail(e);
    if ((e = hipMalloc(&d_i64, 8 * std::max<size_t>(1, pack64.size()))) != hipSuccess) return fail(e);
    if ((e = hipMalloc(&d_counts, 8 * std::max<int64_t>(1, total))) != hipSuccess) return fail(e);
    if (row_major && code_bytes) {
        if ((e = hipMalloc(&d_rows, code_bytes)) != hipSuccess) return fail(e);
        if ((e = hipMemcpyAsync(d_rows, codes, code_bytes, hipMemcpyHostToDevice, stream)) != hipSuccess) return fail(e);
        hipLaunchKernelGGL(transpose_codes_kernel, dim3((unsigned)((n_rows + 63) / 64), (unsigned)((n_cols + 63) / 64)), dim3(256), 0, stream,
                           d_rows, d_codes, n_rows, n_cols);
        if ((e = hipGetLastError()) != hipSuccess) return fail(e);
    } else if ((e = hipMemcpyAsync(d_codes, codes, code_bytes, hipMemcpyHostToDevice, stream)) != hipSuccess) {
        return fail(e);
    }
    if ((e = hipMemcpyAsync(d_i32, pack.data(), 4 * pack.size(), hipMemcpyHostToDevice, stream)) != hipSuccess) return fail(e);
    if ((e = hipMemcpyAsync(d_i64, pack64.data(), 8 * pack64.size(), hipMemcpyHostToDevice, stream)) != hipSuccess) return fail(e);
    if ((e = hipMemsetAsync(d_counts, 0, 8 * std::max<int64_t>(1, total), stream)) != hipSuccess) return fail(e);
    CountArgs A;
    A.codes = d_codes;
    A.tbl_begin = d_i32 + o_tb;
    A.tbl_col = d_i32 + o_tc;
    A.tbl_stride = d_i32 + o_ts;
    A.tbl_lds = d_i32 + o_tl;
    A.tbl_out = d_i64;
    A.grp_begin = d_i32 + o_gb;
    A.counts = d_counts - counts_off[0];
    A.n_rows = n_rows;
    A.n_groups = n_groups;
    if (n_rows > 0 && !bigs.empty()) {
        CountBigArgs Bg;
        Bg.codes = d_codes;
        Bg.tbl_begin = d_i32 + o_bb;
        Bg.tbl_col = d_i32 + o_bc;
        Bg.tbl_stride = d_i64 + o_bs;
        Bg.tbl_out = d_i64 + o_bo;
        Bg.counts = d_counts - counts_off[0];
        Bg.n_rows = n_rows;
        const unsigned bx = (unsigned)std::max<int64_t>(1, std::min<int64_t>((n_rows + 255) / 256, 1024));
        hipLaunchKernelGGL(count_big_kernel, dim3(bx, (unsigned)bigs.size()), dim3(256), 0, stream, Bg);
        if ((e = hipGetLastError()) != hipSuccess) return fail(e);
    }
    if (n_rows > 0 && n_small > 0) {
        const unsigned bx = (unsigned)std::max<int64_t>(1, std::min<int64_t>((n_rows + 255) / 256, std::max(1, 256 * 8 / std::max(1, n_groups))));
        hipLaunchKernelGGL(count_kernel, dim3(bx, (unsigned)n_groups), dim3(256), 0, stream, A);
        if ((e = hipGetLastError()) != hipSuccess) return fail(e);
    }
    std::vector<unsigned long long> host((size_t)std::max<int64_t>(1, total));
    if ((e = hipMemcpyAsync(host.data(), d_counts, 8 * (size_t)std::max<int64_t>(1, total), hipMemcpyDeviceToHost, stream)) != hipSuccess) return fail(e);
    if ((e = hipStreamSynchronize(stream)) != hipSuccess) return fail(e);
    for (int64_t i = 0; i < total; ++i) counts[i] = (int64_t)host[(size_t)i];
    hipFree(d_codes);
    hipFree(d_rows);
    hipFree(d_i32);
    hipFree(d_i64);
    hipFree(d_counts);
    return MIBN_OK;
}

}  // namespace mibn
