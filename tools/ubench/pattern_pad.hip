// Micro-benchmark (GPU box, round 4): does the power-of-two stride between the runs of a sweep tile cost HBM bandwidth?
// The sweep kernel reads a tile as 1 024 runs of 64 bytes at a stride of Rcells * 8 bytes (8 KiB for a 4^10-cell table) - every
// run of a tile at the same offset modulo 8 KiB.  This is pattern_copy.hip's dma<> kernel (64 KiB tiles through LDS-DMA, contiguous
// 64 KiB blocks out) with the stride between runs padded by `pad` cells, i.e. what a padded pitch of the table's slowest axes
// would give.  NBUF = 2: one workgroup per CU (128 KiB of LDS); NBUF = 1: two per CU, the real kernel's geometry.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o pattern_pad pattern_pad.hip && ./pattern_pad [requests]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(e) do { hipError_t r_ = (e); if (r_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(r_), __LINE__); exit(1); } } while (0)

constexpr long kCells = 1 << 20;  // 8 MiB per table (unpadded)

__device__ __forceinline__ uint32_t lds_byte_addr(const void *p) {
    return (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const void *)p;
}
__device__ __forceinline__ void dma16(const double *gsrc, const uint32_t lds_base) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_base) : "memory");
}

// tables of 1024 rows (the 4^5 combinations of the eliminated variables) x rc cells, row pitch rc + pad cells; a tile = RUNC cells of
// every row; out: contiguous 64 KiB blocks (request stride out_req cells)
template <int RUNC, int WG, int NBUF>
__global__ __launch_bounds__(WG) void dma(const double *__restrict__ in, double *__restrict__ out, int tiles_per_req, int iters, long pitch,
                                          long in_req, long out_req) {
    constexpr int TILE = 8192;
    constexpr int PER = TILE / 2 / WG;
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const int tid = threadIdx.x;
    const uint32_t lds0 = lds_byte_addr(lds) + 16u * (uint32_t)(tid & ~63);
    const int first = blockIdx.x * iters;
    auto issue = [&](int tile_id, int buf) {
        const int req = tile_id / tiles_per_req, tile = tile_id % tiles_per_req;
        const double *__restrict__ F = in + (long)req * in_req + (long)tile * RUNC;
        const uint32_t lb = (uint32_t)__builtin_amdgcn_readfirstlane((int)(lds0 + (uint32_t)buf * (TILE * 8)));
#pragma unroll
        for (int i = 0; i < PER; ++i) {
            const int c2 = i * WG + tid, rp = c2 % (RUNC / 2), xc = c2 / (RUNC / 2);
            dma16(F + (long)xc * pitch + 2 * rp, lb + (uint32_t)(i * WG * 16));
        }
    };
    for (int b = 0; b < NBUF && b < iters; ++b) issue(first + b, b);
    for (int it = 0; it < iters; ++it) {
        const int tile_id = first + it;
        const int req = tile_id / tiles_per_req, tile = tile_id % tiles_per_req;
        double *__restrict__ O = out + (long)req * out_req + (long)tile * TILE;
        const double *L = lds + (it % NBUF) * TILE;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        double2 v[PER];
#pragma unroll
        for (int i = 0; i < PER; ++i) v[i] = *reinterpret_cast<const double2 *>(L + 2 * (i * WG + tid));
        if (NBUF == 1 && it + 1 < iters) {  // (single buffer: the tile is in registers, refill before the stores)
            __syncthreads();
            issue(first + it + 1, 0);
        }
#pragma unroll
        for (int i = 0; i < PER; ++i) *reinterpret_cast<double2 *>(O + 2 * (i * WG + tid)) = v[i];
        if (NBUF > 1 && it + NBUF < iters) {
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
    const long max_pad = 1024;
    const long in_cap = (long)nreq * (kCells + 1024 * max_pad + 4096);
    const long n = (long)nreq * kCells;
    double *in, *out;
    CHECK(hipMalloc(&in, in_cap * 8)); CHECK(hipMalloc(&out, (n + (long)nreq * 4096) * 8));
    CHECK(hipMemset(in, 0, in_cap * 8));
    const double gb = 2.0 * n * 8 / 1e9;
    for (int rep = 0; rep < 2; ++rep)
    for (long pad : {0l, 16l, 32l, 48l, 64l, 80l, 144l, 272l, 528l}) {
        for (long rq_pad : {0l, 2064l}) {  // (request tables not at a multiple of 8 MiB either)
#define RUND(RUNC, WG, NBUF, ITERS)                                                                                                    \
        {                                                                                                                              \
            const int tiles = (int)(kCells / 8192);                                                                                    \
            const long rc = kCells / (8192 / RUNC), pitch = rc + pad;                                                                  \
            const long in_req = (8192 / RUNC) * pitch + rq_pad, out_req = kCells + (rq_pad ? 2064 : 0);                                \
            const size_t ldsb = (size_t)NBUF * 8192 * 8 + (NBUF == 1 ? 9728 : 0);                                                      \
            CHECK(hipFuncSetAttribute((const void *)dma<RUNC, WG, NBUF>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsb));      \
            const double ms = time_ms([&] { hipLaunchKernelGGL((dma<RUNC, WG, NBUF>), dim3(nreq * tiles / ITERS), dim3(WG), ldsb, 0, in, out, tiles, ITERS, pitch, in_req, out_req); }, 5); \
            CHECK(hipGetLastError());                                                                                                  \
            printf("runs %4d B  pad %4ld cells  request pad %4ld  %d buffer(s)  %8.3f ms %8.1f GB/s\n", RUNC * 8, pad, rq_pad, NBUF, ms, gb / ms * 1e3); fflush(stdout); \
        }
        RUND(8, 512, 2, 8)
        RUND(8, 512, 1, 8)
        if (pad == 0 || pad == 16 || pad == 272) RUND(16, 512, 2, 8)
        }
    }
    return 0;
}
