// Micro-benchmark (GPU box): the product's ve_sweep_kernel on a synthetic batch of identical five-variable SWEEP steps
// (4^10-cell tables, the stage pattern of a grid row sweep), to tune the kernel outside the engine.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I../../include -o sweep_real sweep_real.hip && ./sweep_real [requests] [tiles per workgroup]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cmath>

#ifdef SWEEP_PROF  // phase timing of the LDS-DMA kernel: 100 MHz ticks accumulated by lane 0 of every workgroup
__device__ unsigned long long g_prof[16];
#define MIBN_PROF_INIT unsigned long long prof_t_ = wall_clock64(), prof_a_[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
#define MIBN_PROF_TICK(k) { const unsigned long long t_ = wall_clock64(); prof_a_[k] += t_ - prof_t_; prof_t_ = t_; }
#define MIBN_PROF_END if (tid == 0) { for (int k_ = 0; k_ < 9; ++k_) atomicAdd(&g_prof[k_], prof_a_[k_]); }
#endif
#include "../../sorobn_amd/csrc/sweep_kernel.hip.h"

using namespace mibn;

#define CHECK(e) do { hipError_t r_ = (e); if (r_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(r_), __LINE__); exit(1); } } while (0)

int main(int argc, char **argv) {
    const int nreq = argc > 1 ? atoi(argv[1]) : 2048;
    const int iters = argc > 2 ? atoi(argv[2]) : 8;
    const int dead = argc > 3 ? atoi(argv[3]) : -1;  // stage whose digit dies (cout = 1): kout = 4, the readout path instead of the wave-owned tail
    const long kCells = 1 << 20;
    const int k = 5, rb = 3, tiles = 128;
    // constants pool: five CPT-like tables of 64 cells [n][x][ctrl]
    std::vector<double> pool(5 * 64);
    for (size_t i = 0; i < pool.size(); ++i) pool[i] = 0.1 + (double)((i * 40503u) % 97) / 97.0;
    // the step program: n_steps, then one SWEEP step
    std::vector<uint32_t> prog;
    prog.push_back(1);
    const int words = kHdrWords + 2 + k * kSweepStageWords + 5 * kSweepSmallWords;
    std::vector<uint32_t> w(words, 0);
    w[0] = kKindSweep | (6u << 8) | ((uint32_t)k << 16) | ((uint32_t)rb << 24);
    w[1] = 1024u | (kFlagSweepCanon << 16);
    w[2] = kSweepTileCells;
    w[3] = tiles;
    w[4] = (uint32_t)kCells;  // output right behind F in the request's arena
    w[5] = 0;
    w[6] = words;
    w[7] = dead < 0 ? (5u | (320u << 16)) : (4u | (272u << 16));
    w[8] = 0x43210;
    if (dead >= 0) {  // the surviving digits, ascending
        uint32_t sv = 0; int m = 0;
        for (int d = 0; d < 5; ++d) if (d != 4 - dead) sv |= (uint32_t)d << (4 * m++);
        w[8] = sv;
    }
    w[9] = (uint32_t)(((dead < 0 ? 2 : 1) * kCells + (dead < 0 ? 0 : kCells / 4)) >> 2);
    uint32_t *p = w.data() + kHdrWords;
    *p++ = 0; *p++ = 0;  // F at arena offset 0
    int t_off = 0;
    for (int j = 0; j < k; ++j) {
        const int dig = k - 1 - j, loop = sweep_loop_digit(k, dig);
        int f[3] = {7, 7, 7};
        for (int d = 0, m = 0; d < k; ++d)
            if (d != dig && d != loop) f[m++] = d;
        const int nctrl = 1;
        const int src = dig > 0 ? dig - 1 : 8 + 4;  // the lower neighbour; the last stage reads bits 4-5 of r
        const uint32_t cout = j == dead ? 1u : 4u;
        *p++ = (uint32_t)dig | (cout << 4) | (1u << 8) | ((uint32_t)nctrl << 12) | ((uint32_t)loop << 16) | ((uint32_t)f[0] << 20) | ((uint32_t)f[1] << 24) | ((uint32_t)f[2] << 28);
        *p++ = (uint32_t)t_off | ((16u * cout) << 16);
        *p++ = (uint32_t)src | ((4u * cout) << 8);
        *p++ = 0; *p++ = 0;
        t_off += 16 * (int)cout;
    }
    for (int j = 0; j < k; ++j) {
        const uint64_t off = kConstFlag | (uint64_t)(j * 64);
        *p++ = (uint32_t)(off & 0xffffffffu); *p++ = (uint32_t)(off >> 32);
        *p++ = 1; *p++ = 4; *p++ = 16; *p++ = 0; *p++ = 0;  // strides of n, x, ctrl in the 64-cell table (a dying digit reads n = 0)
    }
    prog.insert(prog.end(), w.begin(), w.end());
    const int wgs_per_req = (tiles + iters - 1) / iters;
    std::vector<uint64_t> prog_off(nreq, 0), arena_off(nreq);
    std::vector<Item> items(nreq);
    std::vector<uint32_t> wg_item((size_t)nreq * wgs_per_req);
    for (int r = 0; r < nreq; ++r) {
        arena_off[r] = (uint64_t)r * 2 * kCells;
        items[r] = Item{(uint32_t)r, 1u, (uint32_t)iters, (uint32_t)(r * wgs_per_req)};
        for (int q = 0; q < wgs_per_req; ++q) wg_item[(size_t)r * wgs_per_req + q] = (uint32_t)r;
    }
    LevelArgs A;
    uint32_t *d_prog; uint64_t *d_prog_off, *d_arena_off; double *d_pool, *d_arena; Item *d_items; uint32_t *d_wg;
    CHECK(hipMalloc(&d_prog, prog.size() * 4)); CHECK(hipMemcpy(d_prog, prog.data(), prog.size() * 4, hipMemcpyHostToDevice));
    CHECK(hipMalloc(&d_prog_off, nreq * 8)); CHECK(hipMemcpy(d_prog_off, prog_off.data(), nreq * 8, hipMemcpyHostToDevice));
    CHECK(hipMalloc(&d_arena_off, nreq * 8)); CHECK(hipMemcpy(d_arena_off, arena_off.data(), nreq * 8, hipMemcpyHostToDevice));
    CHECK(hipMalloc(&d_pool, pool.size() * 8)); CHECK(hipMemcpy(d_pool, pool.data(), pool.size() * 8, hipMemcpyHostToDevice));
    CHECK(hipMalloc(&d_arena, (size_t)nreq * 2 * kCells * 8));
    CHECK(hipMalloc(&d_items, nreq * sizeof(Item))); CHECK(hipMemcpy(d_items, items.data(), nreq * sizeof(Item), hipMemcpyHostToDevice));
    CHECK(hipMalloc(&d_wg, wg_item.size() * 4)); CHECK(hipMemcpy(d_wg, wg_item.data(), wg_item.size() * 4, hipMemcpyHostToDevice));
    std::vector<double> h(kCells);
    for (long i = 0; i < kCells; ++i) h[i] = 1.0 + (double)((i * 2654435761u) % 1000) / 1000.0;
    for (int r = 0; r < nreq; ++r) CHECK(hipMemcpy(d_arena + (size_t)r * 2 * kCells, h.data(), kCells * 8, hipMemcpyHostToDevice));
    A.prog = d_prog; A.prog_off = d_prog_off; A.arena_off = d_arena_off; A.pool = d_pool; A.arena = d_arena; A.results = nullptr;
    A.items = d_items; A.wg_item = d_wg; A.wg_base = 0;
    for (int wu = 0; wu < 30; ++wu) CHECK(hipMemcpy(d_arena + kCells, d_arena, kCells * 8 * 64, hipMemcpyDeviceToDevice));  // warm the clocks up
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    const unsigned grid = (unsigned)wg_item.size();
    auto run = [&](const char *name, auto kern, int wg, int lds) {
        CHECK(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
        for (int r = 0; r < nreq; ++r) CHECK(hipMemsetAsync(d_arena + (size_t)r * 2 * kCells + kCells, 0, kCells * 8, 0));
        hipLaunchKernelGGL(kern, dim3(grid), dim3(wg), lds, 0, A);
        CHECK(hipDeviceSynchronize());
        const int reps = 10;
        CHECK(hipEventRecord(e0, 0));
        for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(kern, dim3(grid), dim3(wg), lds, 0, A);
        CHECK(hipEventRecord(e1, 0)); CHECK(hipEventSynchronize(e1));
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
        ms /= reps;
        // check cells of the first, a middle and the last request against a direct evaluation
        double err = 0;
        std::vector<double> o(kCells);
        for (int rq : {0, nreq / 2, nreq - 1}) {
            CHECK(hipMemcpy(o.data(), d_arena + (size_t)rq * 2 * kCells + kCells, kCells * 8, hipMemcpyDeviceToHost));
            for (int q = 0; q < 48; ++q) {
                const int r = (q * 131 + rq) % 1024;
                int nc = (q * 577 + 3 * rq) % 1024;
                int nn[5]; for (int d = 0; d < 5; ++d) nn[d] = (nc >> (2 * d)) & 3;  // value on digit d after the pass
                if (dead >= 0) {  // the dead digit carries no value: output cell = the surviving digits packed in ascending order
                    nn[4 - dead] = 0;
                    nc = 0;
                    for (int d = 0, m = 0; d < 5; ++d) if (d != 4 - dead) nc |= nn[d] << (2 * m++);
                }
                double s = 0;
                for (int xc = 0; xc < 1024; ++xc) {
                    int xx[5]; for (int d = 0; d < 5; ++d) xx[d] = (xc >> (2 * d)) & 3;
                    double pr = h[(long)xc * 1024 + r];
                    for (int j = 0; j < 5; ++j) {
                        const int dig = 4 - j;
                        const int ctrl = dig > 0 ? xx[dig - 1] : (r >> 4) & 3;  // lower digits are still x when stage j runs
                        pr *= pool[j * 64 + nn[dig] + 4 * xx[dig] + 16 * ctrl];
                    }
                    s += pr;
                }
                err = fmax(err, fabs(o[(long)r * (dead < 0 ? 1024 : 256) + nc] - s) / s);
            }
        }
#ifdef SWEEP_PROF
        {
            unsigned long long hp[16], z[16] = {0};
            CHECK(hipMemcpyFromSymbol(hp, HIP_SYMBOL(g_prof), sizeof(hp)));
            CHECK(hipMemcpyToSymbol(HIP_SYMBOL(g_prof), z, sizeof(z)));
            const double nt = (double)nreq * tiles * (reps + 1) * 100.0;  // -> us per tile
            printf("      per tile, us: loop top %.2f  dma wait %.2f  barrier+stages %.2f %.2f %.2f %.2f %.2f  readout+stores %.2f  barrier+dma issue %.2f\n",
                   hp[0] / nt, hp[1] / nt, hp[2] / nt, hp[3] / nt, hp[4] / nt, hp[5] / nt, hp[6] / nt, hp[7] / nt, hp[8] / nt);
        }
#endif
        printf("%-26s %d requests x %d tiles, %d tiles per workgroup (%u workgroups): %.3f ms  %.1f GB/s  max rel err %.1e\n", name, nreq, tiles,
               iters, grid, ms, (dead < 0 ? 2.0 : 1.25) * nreq * kCells * 8 / ms / 1e6, err);
        fflush(stdout);
    };
    run("ve_sweep_kernel (r2)", ve_sweep_kernel, kSweepWG, kSweepLdsBytes);
    run("ve_sweep_dma_kernel", ve_sweep_dma_kernel, 512, kSweepLdsBytes);
    if (dead < 0) run("ve_sweep_loader_kernel", ve_sweep_loader_kernel, kSweepLoaderWG, kSweepLoaderLdsBytes);  // round 6: a loader wave + two tile slots
    return 0;
}
