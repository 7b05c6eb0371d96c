// Micro-benchmark (GPU box): a five-variable elimination pass with the tile resident in LDS.
//   out[r, n1..n5] = sum_{x1..x5} F[r, x1..x5] * T1[n1,x1] * T2[n2,x2,n1] * ... * T5[n5,x5,n4]
// F: 4^10 cells (8 MiB) per request, x1 slowest ... x5, then the 4^5 R cells fastest; out: n fastest, then r.
// A workgroup owns tiles of 4^5 x-combinations x RT consecutive R cells (RT * 8 B runs in F), applies the five stages in
// place in LDS (stage j replaces the axis of x_j by the axis of n_j) and writes RT runs of 8 KiB.
//   hipcc --offload-arch=gfx950 -O3 -o sweep_lds sweep_lds.hip && ./sweep_lds
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cmath>

#define CHECK(e) do { hipError_t r_ = (e); if (r_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(r_), __LINE__); exit(1); } } while (0)

constexpr long kCells = 1 << 20;  // 4^10
constexpr int kR = 1024;          // R cells per table
constexpr int kXC = 1024;         // x combinations

// LDS address (in doubles) of logical cell (r, d0..d4), r fastest: d0 is xor-ed with the other digits and the upper r bits
// with d1, so that (r, any one digit) and (d0, d1, r bit 0) both spread over all 64 banks
template <int RB>
__device__ __forceinline__ int swz(int idx) {
    const int d1 = (idx >> (RB + 2)) & 3, d2 = (idx >> (RB + 4)) & 3, d3 = (idx >> (RB + 6)) & 3, d4 = (idx >> (RB + 8)) & 3;
    return idx ^ ((d1 ^ d2 ^ d3 ^ d4) << RB) ^ ((d1 << 1) & ((1 << RB) - 1));
}

template <int RB, int WG, int SWZ>
__global__ __launch_bounds__(WG) void sweep(const double *__restrict__ in, double *__restrict__ out, const double *__restrict__ Tg,
                                            int tiles_per_req, int iters, unsigned long long *prof) {
    constexpr int RT = 1 << RB;
    constexpr int TILE = kXC * RT;  // cells
    extern __shared__ double lds[];
    double *L = lds;            // TILE cells
    double *T = lds + TILE;     // 5 x 64 cells: T[j][ctrl][x][n]
    const int tid = threadIdx.x;
    for (int t = tid; t < 5 * 64; t += WG) T[t] = Tg[t];
    for (int it = 0; it < iters; ++it) {
        const int tile_id = blockIdx.x * iters + it;
        const int req = tile_id / tiles_per_req, tile = tile_id % tiles_per_req;
        const double *__restrict__ F = in + (long)req * kCells + tile * RT;
        double *__restrict__ O = out + (long)req * kCells + (long)tile * RT * kXC;
        __syncthreads();
        unsigned long long t0 = wall_clock64();
        // ---- load: 16-byte loads, RT/2 lanes per run
        constexpr int PAIRS = TILE / 2, PER = PAIRS / WG;
        double2 v[PER];
#pragma unroll
        for (int i = 0; i < PER; ++i) {
            const int c2 = i * WG + tid, rp = c2 % (RT / 2), xc = c2 / (RT / 2);
            v[i] = *reinterpret_cast<const double2 *>(F + (long)xc * kR + 2 * rp);
        }
#pragma unroll
        for (int i = 0; i < PER; ++i) {
            const int c2 = i * WG + tid, rp = c2 % (RT / 2), xc = c2 / (RT / 2);
            const int idx = xc * RT + 2 * rp;
            const int a = SWZ ? swz<RB>(idx) : idx;  // (the swizzle keeps r bit 0: a pair stays a pair)
            *reinterpret_cast<double2 *>(L + a) = v[i];
        }
        __syncthreads();
        unsigned long long t1 = wall_clock64();
        if (tid == 0 && prof) atomicAdd(prof + 0, t1 - t0);
        // ---- stages: x_j lives on digit 4 - j (x1 slowest); ctrl of stage j = n_{j-1} on digit 5 - j
#pragma unroll
        for (int j = 0; j < 5; ++j) {
            const int dig = 4 - j;                       // contracted digit
            const int sx = RT << (2 * dig);              // its stride
            const int cdig = dig + 1;                    // ctrl digit (j > 0)
            // fibers: TILE / 4.  thread digits: r, then the digits other than `dig` from the lowest; the loop runs over the
            // highest remaining one that is not the ctrl digit
            constexpr int FIB = TILE / 4, PERF = FIB / WG;
            // loop digit: the highest digit that is neither contracted nor the ctrl digit
            const int ldig = j == 0 ? 3 : (j == 1 ? 2 : 4);
            static_assert(PERF == 4 || PERF == 2, "unsupported");
            // decode the thread's digits (all but `dig` and `ldig`)
            int rem = tid, base = rem & (RT - 1);
            rem >>= RB;
            int ctrl = 0;
#pragma unroll
            for (int d = 0; d < 5; ++d) {
                if (d == dig || d == ldig) continue;
                const int val = rem & 3;
                rem >>= 2;
                base += val * (RT << (2 * d));
                if (d == cdig && j > 0) ctrl = val;
            }
            double t[16];
            {
                const double *Tp = T + j * 64 + ctrl * 16;
#pragma unroll
                for (int k = 0; k < 16; ++k) t[k] = Tp[k];
            }
#pragma unroll
            for (int l = 0; l < PERF; ++l) {
                const int b = base + (rem * PERF + l) * (RT << (2 * ldig));   // (rem: the thread's leftover bit when PERF = 2)
                double f[4];
#pragma unroll
                for (int x = 0; x < 4; ++x) { const int idx = b + x * sx; f[x] = L[SWZ ? swz<RB>(idx) : idx]; }
#pragma unroll
                for (int n = 0; n < 4; ++n) {
                    double s = 0;
#pragma unroll
                    for (int x = 0; x < 4; ++x) s += f[x] * t[x * 4 + n];
                    const int idx = b + n * sx;
                    L[SWZ ? swz<RB>(idx) : idx] = s;
                }
            }
            __syncthreads();
            { unsigned long long t2 = wall_clock64(); if (tid == 0 && prof) atomicAdd(prof + 1 + j, t2 - t1); t1 = t2; }
        }
        // ---- store: for every r a run of 1024 cells; lanes along (d0, d1) = 128 bytes, then r
        constexpr int PERS = TILE / WG;
#pragma unroll
        for (int i = 0; i < PERS; ++i) {
            const int c = i * WG + tid;
            const int lo = c & 15, r = (c >> 4) & (RT - 1), hi = c >> (4 + RB);   // hi = d2..d4
            const int idx = r + RT * (lo + 16 * hi);
            O[(long)r * kXC + lo + 16 * hi] = L[SWZ ? swz<RB>(idx) : idx];
        }
        { unsigned long long t2 = wall_clock64(); if (tid == 0 && prof) atomicAdd(prof + 6, t2 - t1); }
    }
}


// ---- version 2: padded linear LDS layout (no index arithmetic per access), next tile prefetched into registers ----
// idx = r + 8 d0 + 34 d1 + 136 d2 + 544 d3 + 2184 d4: (r, d0), (r, d4) and (d0, d1, r bit 0) each spread over all banks
template <int PAD> __device__ __forceinline__ constexpr int pstT(int d) { return PAD ? (d == 0 ? 8 : d == 1 ? 34 : d == 2 ? 136 : d == 3 ? 544 : 2184) : (8 << (2 * d)); }
#define pst(d) pstT<PAD>(d)
constexpr int kPadTile = 4 * 2184;

template <int WG, int PER>
__device__ __forceinline__ void load_tile(double (&v)[2 * PER], const double *__restrict__ in, int tile_id, int tiles_per_req, int tid) {
    const int req = tile_id / tiles_per_req, tile = tile_id % tiles_per_req;
    const double *__restrict__ F = in + (long)req * kCells + tile * 8;
#pragma unroll
    for (int i = 0; i < PER; ++i) {
        const int c2 = i * WG + tid;
        const double2 q = *reinterpret_cast<const double2 *>(F + (c2 >> 2) * kR + 2 * (c2 & 3));
        v[2 * i] = q.x; v[2 * i + 1] = q.y;
    }
}

template <int WG, int PF, int PAD>
__global__ __launch_bounds__(WG, 4) void sweep2(const double *__restrict__ in, double *__restrict__ out, const double *__restrict__ Tg,
                                             int tiles_per_req, int iters) {
    constexpr int RT = 8, TILE = kXC * RT;
    extern __shared__ double lds[];
    double *L = lds;
    double *T = lds + kPadTile;
    const int tid = threadIdx.x;
    for (int t = tid; t < 5 * 64; t += WG) T[t] = Tg[t];
    constexpr int PAIRS = TILE / 2, PER = PAIRS / WG;
    double v[2 * PER];
    const int first = blockIdx.x * iters;
    load_tile<WG, PER>(v, in, first, tiles_per_req, tid);
    for (int it = 0; it < iters; ++it) {
        const int tile_id = first + it;
        const int req = tile_id / tiles_per_req, tile = tile_id % tiles_per_req;
        double *__restrict__ O = out + (long)req * kCells + (long)tile * RT * kXC;
        __syncthreads();   // the previous tile's stores have read L
#pragma unroll
        for (int i = 0; i < PER; ++i) {
            const int c2 = i * WG + tid, rp = c2 & 3, xc = c2 >> 2;
            const int lo_ = 2 * rp + (xc & 3) * pst(0) + ((xc >> 2) & 3) * pst(1) + ((xc >> 4) & 3) * pst(2) + ((xc >> 6) & 3) * pst(3) + ((xc >> 8) & 3) * pst(4);
            *reinterpret_cast<double2 *>(L + lo_) = make_double2(v[2 * i], v[2 * i + 1]);
        }
        if (PF) load_tile<WG, PER>(v, in, first + min(it + 1, iters - 1), tiles_per_req, tid);
        __syncthreads();
#pragma unroll
        for (int j = 0; j < 5; ++j) {
            const int dig = 4 - j, cdig = dig + 1;
            // lane digits: r, then d0 (or d4 when d0 is contracted), then the others; loop digit = highest free non-ctrl digit
            const int ldig = j == 0 ? 3 : (j == 1 ? 2 : (j == 4 ? 3 : 4));
            constexpr int PERF = TILE / 4 / WG;
            int rem = tid, base = rem & 7;
            rem >>= 3;
            int ctrl = 0;
            // order of the thread digits: for dig == 0 start with d4 (conflict-free partner), else ascending
#pragma unroll
            for (int k = 0; k < 5; ++k) {
                const int d = dig == 0 ? (k == 0 ? 4 : k) : k;   // dig==0: 4,1,2,3,(4 again skipped below)
                if (dig == 0 && k == 4) continue;
                if (d == dig || d == ldig) continue;
                const int val = rem & 3;
                rem >>= 2;
                base += val * pst(d);
                if (d == cdig && j > 0) ctrl = val;
            }
            double t[16];
            const double *Tp = T + j * 64 + ctrl * 16;
#pragma unroll
            for (int k = 0; k < 16; ++k) t[k] = Tp[k];
#pragma unroll
            for (int l = 0; l < PERF; ++l) {
                const int b = base + (rem * PERF + l) * pst(ldig);
                double f[4];
#pragma unroll
                for (int x = 0; x < 4; ++x) f[x] = L[b + x * pst(dig)];
#pragma unroll
                for (int n = 0; n < 4; ++n) {
                    double s = 0;
#pragma unroll
                    for (int x = 0; x < 4; ++x) s += f[x] * t[x * 4 + n];
                    L[b + n * pst(dig)] = s;
                }
            }
            __syncthreads();
        }
        if (!PF) load_tile<WG, PER>(v, in, first + min(it + 1, iters - 1), tiles_per_req, tid);
        constexpr int PERS = TILE / WG;
#pragma unroll
        for (int i = 0; i < PERS; ++i) {
            const int c = i * WG + tid;
            const int d0 = c & 3, d1 = (c >> 2) & 3, r = (c >> 4) & 7, hi = c >> 7;
            O[(long)r * kXC + (c & 15) + 16 * hi] = L[r + d0 * pst(0) + d1 * pst(1) + (hi & 3) * pst(2) + ((hi >> 2) & 3) * pst(3) + ((hi >> 4) & 3) * pst(4)];
        }
    }
}

// ---- version 3: the loads of two adjacent tiles (r cells 0-7 and 8-15 of the same 128-byte lines) are issued together,
// as back-to-back 64-byte requests for the two halves of every line
template <int WG>
__global__ __launch_bounds__(WG, 4) void sweep3(const double *__restrict__ in, double *__restrict__ out, const double *__restrict__ Tg,
                                                int tiles_per_req, int iters) {
    constexpr int PAD = 0;
    constexpr int RT = 8, TILE = kXC * RT;
    extern __shared__ double lds[];
    double *L = lds;
    double *T = lds + kPadTile;
    const int tid = threadIdx.x;
    for (int t = tid; t < 5 * 64; t += WG) T[t] = Tg[t];
    constexpr int PER = TILE / 2 / WG;   // double2 per thread and tile: 8
    double va[2 * PER], vb[2 * PER];
    const int first = blockIdx.x * iters;    // iters = number of tile PAIRS
    const int rp = tid & 3;
#define LOADPAIR(pair_id_)                                                                           \
    {                                                                                                \
        const int req_ = (2 * (pair_id_)) / tiles_per_req, tile_ = (2 * (pair_id_)) % tiles_per_req; \
        const double *__restrict__ F_ = in + (long)req_ * kCells + tile_ * RT;                       \
        _Pragma("unroll") for (int i = 0; i < PER; ++i) {                                            \
            const int xc = (i * WG + tid) >> 2;                                                      \
            const double2 qa = *reinterpret_cast<const double2 *>(F_ + (long)xc * kR + 2 * rp);      \
            const double2 qb = *reinterpret_cast<const double2 *>(F_ + (long)xc * kR + 8 + 2 * rp);  \
            va[2 * i] = qa.x; va[2 * i + 1] = qa.y; vb[2 * i] = qb.x; vb[2 * i + 1] = qb.y;          \
        }                                                                                            \
    }
#define STAGES_AND_STORE(O_, AFTER_STAGES)                                                           \
    {                                                                                                \
        _Pragma("unroll") for (int j = 0; j < 5; ++j) {                                              \
            const int dig = 4 - j, cdig = dig + 1;                                                   \
            const int ldig = j == 0 ? 3 : (j == 1 ? 2 : (j == 4 ? 3 : 4));                           \
            constexpr int PERF = TILE / 4 / WG;                                                      \
            int rem = tid, base = rem & 7;                                                           \
            rem >>= 3;                                                                               \
            int ctrl = 0;                                                                            \
            _Pragma("unroll") for (int d = 0; d < 5; ++d) {                                          \
                if (d == dig || d == ldig) continue;                                                 \
                const int val = rem & 3;                                                             \
                rem >>= 2;                                                                           \
                base += val * pst(d);                                                                \
                if (d == cdig && j > 0) ctrl = val;                                                  \
            }                                                                                        \
            double t[16];                                                                            \
            const double *Tp = T + j * 64 + ctrl * 16;                                               \
            _Pragma("unroll") for (int q = 0; q < 16; ++q) t[q] = Tp[q];                             \
            _Pragma("unroll") for (int l = 0; l < PERF; ++l) {                                       \
                const int b = base + (rem * PERF + l) * pst(ldig);                                   \
                double f[4];                                                                         \
                _Pragma("unroll") for (int x = 0; x < 4; ++x) f[x] = L[b + x * pst(dig)];            \
                _Pragma("unroll") for (int n = 0; n < 4; ++n) {                                      \
                    double sacc = 0;                                                                 \
                    _Pragma("unroll") for (int x = 0; x < 4; ++x) sacc += f[x] * t[x * 4 + n];       \
                    L[b + n * pst(dig)] = sacc;                                                      \
                }                                                                                    \
            }                                                                                        \
            __syncthreads();                                                                         \
        }                                                                                            \
        AFTER_STAGES;                                                                                \
        constexpr int PERS = TILE / WG;                                                              \
        _Pragma("unroll") for (int i = 0; i < PERS; ++i) {                                           \
            const int c = i * WG + tid;                                                              \
            const int r = (c >> 10) & 7;                                                             \
            (O_)[c] = L[r + 8 * (c & 1023)];                                                         \
        }                                                                                            \
    }
    LOADPAIR(first);
    for (int it = 0; it < iters; ++it) {
        const int tile_id = 2 * (first + it);
        const int req = tile_id / tiles_per_req, tile = tile_id % tiles_per_req;
        double *__restrict__ O = out + (long)req * kCells + (long)tile * RT * kXC;
        __syncthreads();
#pragma unroll
        for (int i = 0; i < PER; ++i) *reinterpret_cast<double2 *>(L + 2 * (i * WG + tid)) = make_double2(va[2 * i], va[2 * i + 1]);
        __syncthreads();
        STAGES_AND_STORE(O, );
        __syncthreads();
#pragma unroll
        for (int i = 0; i < PER; ++i) *reinterpret_cast<double2 *>(L + 2 * (i * WG + tid)) = make_double2(vb[2 * i], vb[2 * i + 1]);
        __syncthreads();
        STAGES_AND_STORE(O + RT * kXC, LOADPAIR(first + min(it + 1, iters - 1)));
    }
#undef LOADPAIR
#undef STAGES_AND_STORE
}

template <class Fn>
static double time_ms(Fn f, int reps) {
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    f(); CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0, 0));
    for (int i = 0; i < reps; ++i) f();
    CHECK(hipEventRecord(e1, 0)); CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    return ms / reps;
}

int main(int argc, char **argv) {
    const int nreq = argc > 1 ? atoi(argv[1]) : 512;
    const long n = (long)nreq * kCells;
    double *in, *out, *T;
    unsigned long long *prof; CHECK(hipMalloc(&prof, 64)); 
    CHECK(hipMalloc(&in, n * 8)); CHECK(hipMalloc(&out, n * 8)); CHECK(hipMalloc(&T, 5 * 64 * 8));
    std::vector<double> h(kCells);
    for (long i = 0; i < kCells; ++i) h[i] = 1.0 + (double)((i * 2654435761u) % 1000) / 1000.0;
    for (int r = 0; r < nreq; ++r) CHECK(hipMemcpy(in + (long)r * kCells, h.data(), kCells * 8, hipMemcpyHostToDevice));
    std::vector<double> hT(5 * 64);
    for (int i = 0; i < 5 * 64; ++i) hT[i] = 0.1 + (double)((i * 40503u) % 97) / 97.0;
    CHECK(hipMemcpy(T, hT.data(), hT.size() * 8, hipMemcpyHostToDevice));
    const double gb = 2.0 * n * 8 / 1e9;
    for (int w = 0; w < 40; ++w) CHECK(hipMemcpy(out, in, n * 8, hipMemcpyDeviceToDevice));  // warm the clocks up
    CHECK(hipDeviceSynchronize());
    // reference for request 0, a few cells
    auto ref = [&](int r, int nc) {
        // nc = n5 + 4 n4 + 16 n3 + 64 n2 + 256 n1 (n5 on digit 0 ... n1 on digit 4)
        int nn[5]; for (int j = 0; j < 5; ++j) nn[j] = (nc >> (2 * (4 - j))) & 3;   // nn[0] = n1
        double s = 0;
        for (int xc = 0; xc < 1024; ++xc) {
            int xx[5]; for (int j = 0; j < 5; ++j) xx[j] = (xc >> (2 * (4 - j))) & 3;
            double p = h[(long)xc * kR + r];
            for (int j = 0; j < 5; ++j) p *= hT[j * 64 + (j ? nn[j - 1] : 0) * 16 + xx[j] * 4 + nn[j]];
            s += p;
        }
        return s;
    };
#define RUN(RB, WG, SWZ, ITERS, label)                                                                                       \
    {                                                                                                                        \
        const int RT = 1 << RB, tiles = kR / RT;                                                                             \
        const size_t lds = (size_t)(kXC * RT + 5 * 64) * 8;                                                                  \
        CHECK(hipFuncSetAttribute((const void *)sweep<RB, WG, SWZ>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));  \
        CHECK(hipMemset(out, 0, n * 8));                                                                                     \
        const double ms = time_ms([&] { hipLaunchKernelGGL((sweep<RB, WG, SWZ>), dim3(nreq * tiles / ITERS), dim3(WG), lds, 0, in, out, T, tiles, ITERS, (unsigned long long *)nullptr); }, 20); \
        CHECK(hipMemset(prof, 0, 64)); hipLaunchKernelGGL((sweep<RB, WG, SWZ>), dim3(nreq * tiles / ITERS), dim3(WG), lds, 0, in, out, T, tiles, ITERS, prof); \
        unsigned long long hp[8]; CHECK(hipMemcpy(hp, prof, 64, hipMemcpyDeviceToHost)); \
        CHECK(hipGetLastError());                                                                                            \
        std::vector<double> o(kCells);                                                                                       \
        CHECK(hipMemcpy(o.data(), out + (long)(nreq - 1) * kCells, kCells * 8, hipMemcpyDeviceToHost));                       \
        double err = 0;                                                                                                      \
        for (int k = 0; k < 40; ++k) { const int r = (k * 131) % kR, nc = (k * 577) % 1024; const double e = ref(r, nc); err = fmax(err, fabs(o[(long)r * kXC + nc] - e) / e); } \
        printf("%-60s %8.3f ms  %8.1f GB/s  max rel err %.1e\n", label, ms, gb / ms * 1e3, err);            \
        { const double nt = (double)nreq * tiles; printf("      per tile (100 MHz ticks -> us): load %.2f  stages %.2f %.2f %.2f %.2f %.2f  store %.2f\n", hp[0] / nt / 100, hp[1] / nt / 100, hp[2] / nt / 100, hp[3] / nt / 100, hp[4] / nt / 100, hp[5] / nt / 100, hp[6] / nt / 100); } fflush(stdout); \
    }

#define RUN2(WG, PF, PAD, ITERS, label)                                                                                           \
    {                                                                                                                        \
        const int tiles = kR / 8;                                                                                            \
        const size_t lds = (size_t)(kPadTile + 5 * 64) * 8;                                                                  \
        CHECK(hipFuncSetAttribute((const void *)sweep2<WG, PF, PAD>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));      \
        CHECK(hipMemset(out, 0, n * 8));                                                                                     \
        { int nb = 0; CHECK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, (const void *)sweep2<WG, PF, PAD>, WG, lds)); printf("[WGs/CU %d] ", nb); } \
        const double ms = time_ms([&] { hipLaunchKernelGGL((sweep2<WG, PF, PAD>), dim3(nreq * tiles / ITERS), dim3(WG), lds, 0, in, out, T, tiles, ITERS); }, 20); \
        CHECK(hipGetLastError());                                                                                            \
        std::vector<double> o(kCells);                                                                                       \
        CHECK(hipMemcpy(o.data(), out + (long)(nreq - 1) * kCells, kCells * 8, hipMemcpyDeviceToHost));                       \
        double err = 0;                                                                                                      \
        for (int k = 0; k < 40; ++k) { const int r = (k * 131) % kR, nc = (k * 577) % 1024; const double e = ref(r, nc); err = fmax(err, fabs(o[(long)r * kXC + nc] - e) / e); } \
        printf("%-60s %8.3f ms  %8.1f GB/s  max rel err %.1e\n", label, ms, gb / ms * 1e3, err); fflush(stdout);            \
    }

#define RUN3(WG, ITERS, label)                                                                                               \
    {                                                                                                                        \
        const int tiles = kR / 8;                                                                                            \
        const size_t lds = (size_t)(kPadTile + 5 * 64) * 8;                                                                  \
        CHECK(hipFuncSetAttribute((const void *)sweep3<WG>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));          \
        CHECK(hipMemset(out, 0, n * 8));                                                                                     \
        { int nb = 0; CHECK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, (const void *)sweep3<WG>, WG, lds)); printf("[WGs/CU %d] ", nb); } \
        const double ms = time_ms([&] { hipLaunchKernelGGL((sweep3<WG>), dim3(nreq * tiles / 2 / ITERS), dim3(WG), lds, 0, in, out, T, tiles, ITERS); }, 20); \
        CHECK(hipGetLastError());                                                                                            \
        std::vector<double> o(kCells);                                                                                       \
        CHECK(hipMemcpy(o.data(), out + (long)(nreq - 1) * kCells, kCells * 8, hipMemcpyDeviceToHost));                       \
        double err = 0;                                                                                                      \
        for (int k = 0; k < 40; ++k) { const int r = (k * 131) % kR, nc = (k * 577) % 1024; const double e = ref(r, nc); err = fmax(err, fabs(o[(long)r * kXC + nc] - e) / e); } \
        printf("%-60s %8.3f ms  %8.1f GB/s  max rel err %.1e\n", label, ms, gb / ms * 1e3, err); fflush(stdout);            \
    }
    RUN3(512, 4, "v3 128-byte runs for tile pairs, 512 thr, 4 pairs/WG");
    RUN3(512, 8, "v3 128-byte runs for tile pairs, 512 thr, 8 pairs/WG");
    RUN3(512, 2, "v3 128-byte runs for tile pairs, 512 thr, 2 pairs/WG");
    RUN2(512, 1, 1, 8, "v2 padded+prefetch, 512 thr, 8 tiles/WG");
    RUN2(512, 1, 0, 8, "v2 pow2 strides+prefetch, 512 thr, 8 tiles/WG");
    RUN2(512, 0, 0, 8, "v2 pow2 strides, no prefetch, 512 thr, 8 tiles/WG");
    RUN2(512, 0, 0, 1, "v2 pow2 strides, 512 thr, 1 tile/WG");
    RUN2(512, 0, 1, 1, "v2 padded, 512 thr, 1 tile/WG");
    RUN2(1024, 1, 1, 8, "v2 padded+prefetch, 1024 thr, 8 tiles/WG");
    RUN(3, 512, 1, 1, "5 vars, RT=8 (64 KiB tile), 512 thr, swizzled, 1 tile/WG");
    RUN(3, 512, 0, 1, "5 vars, RT=8 (64 KiB tile), 512 thr, plain LDS layout");
    RUN(3, 512, 1, 4, "5 vars, RT=8, 512 thr, swizzled, 4 tiles/WG");
    RUN(3, 1024, 1, 1, "5 vars, RT=8, 1024 thr, swizzled, 1 tile/WG");
    RUN(3, 1024, 1, 4, "5 vars, RT=8, 1024 thr, swizzled, 4 tiles/WG");
    RUN(2, 512, 1, 1, "5 vars, RT=4 (32 KiB tile, 32 B runs), 512 thr, swizzled");
    RUN(2, 256, 1, 2, "5 vars, RT=4 (32 KiB tile), 256 thr, swizzled, 2 tiles/WG");
    RUN(4, 1024, 1, 1, "5 vars, RT=16 (128 KiB tile), 1024 thr, swizzled");
    return 0;
}
