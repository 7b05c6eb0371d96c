// Micro-benchmark (GPU box): level-synchronous launches vs ONE persistent dataflow kernel whose workgroups pull
// (level, request, tile) units from a global ticket and wait on per-request completion counters
// (agent-scope release/acquire).  Validates visibility: every level adds 1.0 to every cell.
//   hipcc --offload-arch=gfx950 -O3 -o dataflow dataflow.hip && ./dataflow [nreq] [levels] [tiles]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(e) do { hipError_t r_ = (e); if (r_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(r_), __LINE__); exit(1); } } while (0)

constexpr long kCells = 1 << 20;  // 8 MiB per table; each request ping-pongs between two tables

__device__ __forceinline__ void tile_body(const double *__restrict__ in, double *__restrict__ out, long begin, long end, int tid) {
    for (long i = begin + 2 * tid; i < end; i += 512) {
        const double2 v = *reinterpret_cast<const double2 *>(in + i);
        *reinterpret_cast<double2 *>(out + i) = make_double2(v.x + 1.0, v.y + 1.0);
    }
}

__global__ __launch_bounds__(256) void level_kernel(double *buf, int level, int tiles) {
    const int req = blockIdx.x / tiles, tile = blockIdx.x % tiles;
    const double *in = buf + ((long)req * 2 + (level & 1)) * kCells;
    double *out = buf + ((long)req * 2 + ((level + 1) & 1)) * kCells;
    const long per = kCells / tiles;
    tile_body(in, out, tile * per, (tile + 1) * per, threadIdx.x);
}

// MODE 0: fence + relaxed atomic; MODE 1: no dependency tracking at all (upper bound, results wrong)
template <int MODE>
__global__ __launch_bounds__(256) void dataflow_kernel(double *buf, int nreq, int levels, int tiles, unsigned *ticket, unsigned *done,
                                                       unsigned *err) {
    __shared__ unsigned sh_u;
    const int tid = threadIdx.x;
    const unsigned total = (unsigned)nreq * levels * tiles;
    for (;;) {
        __syncthreads();
        if (tid == 0) sh_u = __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) ? 0xffffffffu : atomicAdd(ticket, 1u);
        __syncthreads();
        const unsigned u = sh_u;
        if (u >= total) return;
        const int level = u / ((unsigned)nreq * tiles);
        const unsigned rem = u % ((unsigned)nreq * tiles);
        const int req = rem / tiles, tile = rem % tiles;
        if (MODE == 0 && level > 0) {
            if (tid == 0) {
                const unsigned need = (unsigned)level * tiles;
                long spins = 0;
                while (__hip_atomic_load(done + req, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < need) {
                    __builtin_amdgcn_s_sleep(8);
                    if (++spins > (1l << 22)) { atomicAdd(err, 1u); break; }
                }
            }
            __syncthreads();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        }
        const double *in = buf + ((long)req * 2 + (level & 1)) * kCells;
        double *out = buf + ((long)req * 2 + ((level + 1) & 1)) * kCells;
        const long per = kCells / tiles;
        tile_body(in, out, tile * per, (tile + 1) * per, tid);
        if (MODE == 0) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");  // every wave: its stores are visible device-wide
            __syncthreads();
            if (tid == 0) __hip_atomic_fetch_add(done + req, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

int main(int argc, char **argv) {
    const int nreq = argc > 1 ? atoi(argv[1]) : 512;
    const int levels = argc > 2 ? atoi(argv[2]) : 16;
    const int tiles = argc > 3 ? atoi(argv[3]) : 32;
    double *buf;
    unsigned *ctl;
    const long n = (long)nreq * 2 * kCells;
    CHECK(hipMalloc(&buf, n * 8));
    CHECK(hipMalloc(&ctl, (nreq + 16) * 4));
    std::vector<double> h(kCells);
    hipEvent_t a, b;
    CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
    const double gb = (double)nreq * levels * kCells * 16 / 1e9;
    auto verify = [&](const char *name, float ms) {
        // every cell of the final table of a few requests must equal `levels`
        long bad = 0;
        for (int r : {0, nreq / 2, nreq - 1}) {
            CHECK(hipMemcpy(h.data(), buf + ((long)r * 2 + (levels & 1)) * kCells, kCells * 8, hipMemcpyDeviceToHost));
            for (long i = 0; i < kCells; ++i) bad += h[i] != (double)levels;
        }
        unsigned herr = 0;
        CHECK(hipMemcpy(&herr, ctl + 1, 4, hipMemcpyDeviceToHost));
        printf("%-64s %8.3f ms %8.1f GB/s   wrong cells %ld  spin timeouts %u\n", name, ms, gb / ms * 1e3, bad, herr);
        fflush(stdout);
    };
    for (int rep = 0; rep < 2; ++rep) {
        CHECK(hipMemset(buf, 0, n * 8));
        CHECK(hipDeviceSynchronize());
        CHECK(hipEventRecord(a));
        for (int l = 0; l < levels; ++l) hipLaunchKernelGGL(level_kernel, dim3(nreq * tiles), dim3(256), 0, 0, buf, l, tiles);
        CHECK(hipEventRecord(b));
        CHECK(hipEventSynchronize(b));
        float ms;
        CHECK(hipEventElapsedTime(&ms, a, b));
        if (rep) verify("level-synchronous: one launch per level", ms);
    }
    for (int wgs : {2, 4, 8}) {
        for (int mode : {0, 1}) {
            CHECK(hipMemset(buf, 0, n * 8));
            CHECK(hipMemset(ctl, 0, (nreq + 16) * 4));
            CHECK(hipDeviceSynchronize());
            CHECK(hipEventRecord(a));
            if (mode == 0) hipLaunchKernelGGL(dataflow_kernel<0>, dim3(256 * wgs), dim3(256), 0, 0, buf, nreq, levels, tiles, ctl, ctl + 16, ctl + 1);
            else hipLaunchKernelGGL(dataflow_kernel<1>, dim3(256 * wgs), dim3(256), 0, 0, buf, nreq, levels, tiles, ctl, ctl + 16, ctl + 1);
            CHECK(hipEventRecord(b));
            CHECK(hipEventSynchronize(b));
            float ms;
            CHECK(hipEventElapsedTime(&ms, a, b));
            char name[128];
            snprintf(name, sizeof name, "dataflow persistent kernel, %d WGs/CU, %s", wgs, mode == 0 ? "release/acquire counters" : "NO dependency tracking (bound)");
            verify(name, ms);
        }
    }
    return 0;
}
