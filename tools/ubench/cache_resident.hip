// Micro-benchmark (GPU box): what does a producer -> consumer chain of launches reach when its working set fits the
// 256 MiB Infinity Cache (or the 32 MiB of L2) instead of streaming through HBM?  Launch k reads the buffer launch k-1
// wrote and writes the other one (the level-synchronous schedule of ve_level_kernel, reduced to a copy).
//   hipcc --offload-arch=gfx950 -O3 -o cache_resident cache_resident.hip && ./cache_resident
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(e) do { hipError_t r_ = (e); if (r_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(r_), __LINE__); exit(1); } } while (0)

// one workgroup copies one contiguous tile of `per` double2 (the planner's tiles: contiguous slices of a table)
__global__ __launch_bounds__(256) void copy_tile(const double2 *__restrict__ in, double2 *__restrict__ out, long per) {
    const double2 *__restrict__ a = in + blockIdx.x * per;
    double2 *__restrict__ b = out + blockIdx.x * per;
    for (long i = threadIdx.x; i < per; i += 1024) {
        double2 v0 = a[i], v1, v2, v3;
        const bool p1 = i + 256 < per, p2 = i + 512 < per, p3 = i + 768 < per;
        if (p1) v1 = a[i + 256];
        if (p2) v2 = a[i + 512];
        if (p3) v3 = a[i + 768];
        b[i] = v0;
        if (p1) b[i + 256] = v1;
        if (p2) b[i + 512] = v2;
        if (p3) b[i + 768] = v3;
    }
}

int main(int argc, char **argv) {
    const long maxMB = 4096;
    double2 *A, *B;
    CHECK(hipMalloc(&A, maxMB << 20)); CHECK(hipMalloc(&B, maxMB << 20));
    CHECK(hipMemset(A, 0, maxMB << 20)); CHECK(hipMemset(B, 0, maxMB << 20));
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    const int nstreams = 4;
    hipStream_t st[nstreams];
    for (auto &s : st) CHECK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));

    printf("# ping-pong copy, one stream: buffer MB (working set = 2x), tile KB, launches, us/launch, GB/s (read+write)\n");
    for (int tileKB : {64, 128, 256}) {
        for (long mb : {4l, 8l, 16l, 32l, 48l, 64l, 96l, 128l, 192l, 256l, 512l, 2048l}) {
            const long bytes = mb << 20, per = (long)tileKB * 1024 / 16, tiles = bytes / (tileKB * 1024l);
            const int launches = (int)(mb <= 64 ? 400 : mb <= 256 ? 100 : 20);
            for (int rep = 0; rep < 2; ++rep) {
                CHECK(hipEventRecord(e0, st[0]));
                for (int k = 0; k < launches; ++k)
                    hipLaunchKernelGGL(copy_tile, dim3(tiles), dim3(256), 0, st[0], (k & 1) ? B : A, (k & 1) ? A : B, per);
                CHECK(hipEventRecord(e1, st[0])); CHECK(hipEventSynchronize(e1));
                float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
                if (rep) printf("1stream buf %5ld MB tile %3d KB launches %3d  %8.2f us/launch  %8.1f GB/s\n", mb, tileKB, launches,
                                ms * 1e3 / launches, 2.0 * bytes * launches / ms / 1e6);
            }
        }
    }
    printf("# the same on %d streams at once (each its own pair of buffers; total working set = %d x 2 x buffer)\n", nstreams, nstreams);
    for (long mb : {2l, 4l, 8l, 16l, 24l, 32l, 64l, 128l, 512l}) {
        const int tileKB = 128;
        const long bytes = mb << 20, per = (long)tileKB * 1024 / 16, tiles = bytes / (tileKB * 1024l);
        const int launches = (int)(mb <= 32 ? 400 : 50);
        for (int rep = 0; rep < 2; ++rep) {
            CHECK(hipDeviceSynchronize());
            CHECK(hipEventRecord(e0, 0));
            for (auto &s : st) CHECK(hipStreamWaitEvent(s, e0, 0));
            for (int k = 0; k < launches; ++k)
                for (int s = 0; s < nstreams; ++s) {
                    double2 *a = A + s * (bytes / 16), *b = B + s * (bytes / 16);
                    hipLaunchKernelGGL(copy_tile, dim3(tiles), dim3(256), 0, st[s], (k & 1) ? b : a, (k & 1) ? a : b, per);
                }
            hipEvent_t done[nstreams];
            for (int s = 0; s < nstreams; ++s) { CHECK(hipEventCreate(&done[s])); CHECK(hipEventRecord(done[s], st[s])); CHECK(hipStreamWaitEvent(0, done[s], 0)); }
            CHECK(hipEventRecord(e1, 0)); CHECK(hipEventSynchronize(e1));
            float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
            if (rep) printf("%dstreams buf %4ld MB each tile %3d KB launches %3d/stream  %8.2f us/launch-round  %8.1f GB/s\n", nstreams, mb, tileKB,
                            launches, ms * 1e3 / launches, 2.0 * bytes * launches * nstreams / ms / 1e6);
            for (auto &d : done) CHECK(hipEventDestroy(d));
        }
    }
    return 0;
}
