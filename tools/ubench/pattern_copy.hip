// Micro-benchmark (GPU box): the HBM ceiling of the sweep kernel's ACCESS PATTERN, without its arithmetic.
// A tile = 1024 runs of RUN bytes read at a stride of STRIDE bytes (the slowest axes of a 4^10-cell table) and
// RUN * 1024 bytes written as one contiguous block.  Variants:
//   copy16      plain grid-stride 16-byte copy (the box's streaming ceiling for a 1:1 read/write mix)
//   gather<RUN> the tile pattern through registers, 8 x 16 bytes in flight per lane, 256-lane workgroups, high occupancy
//   dma<RUN>    the tile pattern through LDS-DMA into a double-buffered 64 KiB tile + ds_read / global_store out,
//               one 512-lane workgroup per CU (the geometry of ve_sweep_dma_kernel with the stages taken out)
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o pattern_copy pattern_copy.hip && ./pattern_copy
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(e) do { hipError_t r_ = (e); if (r_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(r_), __LINE__); exit(1); } } while (0)

constexpr long kCells = 1 << 20;  // 8 MiB per table

__global__ __launch_bounds__(256) void copy16(const double2 *__restrict__ in, double2 *__restrict__ out, long n) {
    for (long i = blockIdx.x * 256l + threadIdx.x; i < n; i += (long)gridDim.x * 256) out[i] = in[i];
}

// RUNC = cells per run (8: 64 B, 16: 128 B, 32: 256 B); a tile = 1024 runs = RUNC * 1024 cells; tables of 2^20 cells
template <int RUNC>
__global__ __launch_bounds__(256) void gather(const double *__restrict__ in, double *__restrict__ out, int tiles_per_req) {
    constexpr int TILE = 1024 * RUNC;        // cells
    constexpr int PER = TILE / 2 / 256;      // 16-byte pieces per lane
    constexpr long RC = kCells / 1024;       // cells of the R axes per table = stride between runs
    const int tid = threadIdx.x;
    const int req = blockIdx.x / tiles_per_req, tile = blockIdx.x % tiles_per_req;
    const double *__restrict__ F = in + (long)req * kCells + (long)tile * RUNC;
    double *__restrict__ O = out + (long)req * kCells + (long)tile * TILE;
    double2 v[PER];
#pragma unroll
    for (int i = 0; i < PER; ++i) {
        const int c2 = i * 256 + tid, rp = c2 % (RUNC / 2), xc = c2 / (RUNC / 2);
        v[i] = *reinterpret_cast<const double2 *>(F + (long)xc * RC + 2 * rp);
    }
#pragma unroll
    for (int i = 0; i < PER; ++i) *reinterpret_cast<double2 *>(O + 2 * (i * 256 + tid)) = v[i];
}

__device__ __forceinline__ uint32_t lds_byte_addr(const void *p) {
    return (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const void *)p;
}
__device__ __forceinline__ void dma16(const double *gsrc, const uint32_t lds_base) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_base) : "memory");
}

// 64 KiB tiles (8192 cells) whatever the run length: 8192 / RUNC runs of RUNC cells
template <int RUNC, int WG, int NBUF>
__global__ __launch_bounds__(WG) void dma(const double *__restrict__ in, double *__restrict__ out, int tiles_per_req, int iters) {
    constexpr int TILE = 8192;
    constexpr int PER = TILE / 2 / WG;
    constexpr long RC = kCells / (TILE / RUNC);
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const int tid = threadIdx.x;
    const uint32_t lds0 = lds_byte_addr(lds) + 16u * (uint32_t)(tid & ~63);
    const int first = blockIdx.x * iters;
    auto issue = [&](int tile_id, int buf) {
        const int req = tile_id / tiles_per_req, tile = tile_id % tiles_per_req;
        const double *__restrict__ F = in + (long)req * kCells + (long)tile * RUNC;
        const uint32_t lb = (uint32_t)__builtin_amdgcn_readfirstlane((int)(lds0 + (uint32_t)buf * (TILE * 8)));
#pragma unroll
        for (int i = 0; i < PER; ++i) {
            const int c2 = i * WG + tid, rp = c2 % (RUNC / 2), xc = c2 / (RUNC / 2);
            dma16(F + (long)xc * RC + 2 * rp, lb + (uint32_t)(i * WG * 16));
        }
    };
    for (int b = 0; b < NBUF && b < iters; ++b) issue(first + b, b);
    for (int it = 0; it < iters; ++it) {
        const int tile_id = first + it;
        const int req = tile_id / tiles_per_req, tile = tile_id % tiles_per_req;
        double *__restrict__ O = out + (long)req * kCells + (long)tile * TILE;
        const double *L = lds + (it % NBUF) * TILE;
        // VMEM operations issued after this tile's DMA (they retire in order): the DMAs of the prologue behind it, then per
        // finished tile j its stores and the DMA of tile j + NBUF
        int n = it < NBUF ? (min(NBUF, iters) - 1 - it) * PER : 0;
        for (int j = max(it - NBUF + 1, 0); j < it; ++j) n += PER + (j + NBUF < iters ? PER : 0);
        switch (n) {
#define W(K) case K: asm volatile("s_waitcnt vmcnt(" #K ")" ::: "memory"); break;
            W(4) W(8) W(12) W(16) W(20) W(24) W(28) W(32) W(36) W(40) W(44) W(48) W(52) W(56) W(60)
#undef W
            default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
        }
        __syncthreads();
        double2 v[PER];
#pragma unroll
        for (int i = 0; i < PER; ++i) v[i] = *reinterpret_cast<const double2 *>(L + 2 * (i * WG + tid));
#pragma unroll
        for (int i = 0; i < PER; ++i) *reinterpret_cast<double2 *>(O + 2 * (i * WG + tid)) = v[i];
        if (it + NBUF < iters) {
            __syncthreads();
            issue(first + it + NBUF, it % NBUF);
        }
    }
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
    const int nreq = argc > 1 ? atoi(argv[1]) : 2048;
    const long n = (long)nreq * kCells;
    double *in, *out;
    CHECK(hipMalloc(&in, n * 8)); CHECK(hipMalloc(&out, n * 8));
    std::vector<double> h(kCells);
    for (long i = 0; i < kCells; ++i) h[i] = (double)i;
    for (int r = 0; r < nreq; ++r) CHECK(hipMemcpy(in + (long)r * kCells, h.data(), kCells * 8, hipMemcpyHostToDevice));
    const double gb = 2.0 * n * 8 / 1e9;
    for (int w = 0; w < 10; ++w) CHECK(hipMemcpy(out, in, n * 8, hipMemcpyDeviceToDevice));
    CHECK(hipDeviceSynchronize());
    {
        for (int g : {2048, 4096, 16384}) {
            const double ms = time_ms([&] { hipLaunchKernelGGL(copy16, dim3(g), dim3(256), 0, 0, (const double2 *)in, (double2 *)out, n / 2); }, 5);
            printf("copy16 grid %5d                                         %8.3f ms %8.1f GB/s\n", g, ms, gb / ms * 1e3);
        }
    }
    std::vector<double> o(kCells);
    auto check = [&](int runc, const char *label, double ms) {
        // out[(tile * 1024 + xc) * runc + r] == in[xc * (kCells / 1024) + tile * runc + r]   (gather: 1024 runs per tile)
        CHECK(hipMemcpy(o.data(), out + (long)(nreq - 1) * kCells, kCells * 8, hipMemcpyDeviceToHost));
        printf("%-56s %8.3f ms %8.1f GB/s", label, ms, gb / ms * 1e3);
        (void)runc;
    };
#define RUNG(RUNC, label)                                                                                                              \
    {                                                                                                                                  \
        const int tiles = (int)(kCells / (1024 * RUNC));                                                                               \
        CHECK(hipMemset(out, 0, n * 8));                                                                                               \
        const double ms = time_ms([&] { hipLaunchKernelGGL((gather<RUNC>), dim3(nreq * tiles), dim3(256), 0, 0, in, out, tiles); }, 5); \
        check(RUNC, label, ms);                                                                                                        \
        long bad = 0;                                                                                                                  \
        for (long t = 0; t < tiles; ++t) for (long xc = 0; xc < 1024; xc += 37) for (long r = 0; r < RUNC; ++r)                         \
            bad += o[(t * 1024 + xc) * RUNC + r] != h[xc * (kCells / 1024) + t * RUNC + r];                                            \
        printf("  %s\n", bad ? "WRONG" : "ok"); fflush(stdout);                                                                        \
    }
    RUNG(8, "gather  64-byte runs (regs, 256 lanes, 64 KiB tiles)");
    RUNG(16, "gather 128-byte runs (regs, 256 lanes, 128 KiB tiles)");
    RUNG(32, "gather 256-byte runs (regs, 256 lanes, 256 KiB tiles)");
#define RUND(RUNC, WG, NBUF, ITERS, label)                                                                                             \
    {                                                                                                                                  \
        const int tiles = (int)(kCells / 8192);                                                                                        \
        const size_t ldsb = (size_t)NBUF * 8192 * 8;                                                                                   \
        CHECK(hipFuncSetAttribute((const void *)dma<RUNC, WG, NBUF>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsb));          \
        CHECK(hipMemset(out, 0, n * 8));                                                                                               \
        const double ms = time_ms([&] { hipLaunchKernelGGL((dma<RUNC, WG, NBUF>), dim3(nreq * tiles / ITERS), dim3(WG), ldsb, 0, in, out, tiles, ITERS); }, 5); \
        CHECK(hipGetLastError());                                                                                                      \
        check(RUNC, label, ms);                                                                                                        \
        const long runs = 8192 / RUNC;                                                                                                 \
        long bad = 0;                                                                                                                  \
        for (long t = 0; t < tiles; ++t) for (long xc = 0; xc < runs; xc += 37) for (long r = 0; r < RUNC; ++r)                         \
            bad += o[(t * runs + xc) * RUNC + r] != h[xc * (kCells / runs) + t * RUNC + r];                                            \
        printf("  %s\n", bad ? "WRONG" : "ok"); fflush(stdout);                                                                        \
    }
    RUND(8, 512, 2, 8, "dma  64-byte runs, 512 lanes, 2 buffers, 8 tiles/WG");
    RUND(8, 512, 2, 16, "dma  64-byte runs, 512 lanes, 2 buffers, 16 tiles/WG");
    RUND(8, 1024, 2, 8, "dma  64-byte runs, 1024 lanes, 2 buffers, 8 tiles/WG");
    RUND(16, 512, 2, 8, "dma 128-byte runs, 512 lanes, 2 buffers, 8 tiles/WG");
    RUND(32, 512, 2, 8, "dma 256-byte runs, 512 lanes, 2 buffers, 8 tiles/WG");
    RUND(128, 512, 2, 8, "dma 1 KiB runs, 512 lanes, 2 buffers, 8 tiles/WG");
    RUND(8, 256, 2, 8, "dma  64-byte runs, 256 lanes, 2 buffers (1 WG/CU), 8 tiles/WG");
    return 0;
}
