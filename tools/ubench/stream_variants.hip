// Micro-benchmark (GPU box): which load/store shape streams the elimination frontier fastest on gfx950?
// Emulates the dominant step  out[r, n] = sum_x F[x, r] * T[n, x, ctrl(r)]  over many private 8 MiB tables.
//   hipcc --offload-arch=gfx950 -O3 -o stream_variants stream_variants.hip && ./stream_variants
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(e) do { hipError_t r_ = (e); if (r_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(r_), __LINE__); exit(1); } } while (0)

constexpr int kWG = 256;
constexpr long kCells = 1 << 20;  // 4^10 cells per table

__global__ __launch_bounds__(256) void copy16(const double2 *__restrict__ in, double2 *__restrict__ out, long n) {
    for (long i = blockIdx.x * 256l + threadIdx.x; i < n; i += (long)gridDim.x * 256) out[i] = in[i];
}

// STORE: 0 = per-lane fiber (NC*8 contiguous bytes per lane, 16-byte pieces), 1 = transposed through LDS so every store
// instruction writes 1 KiB contiguous per wave, 2 = per-lane fiber with nontemporal stores + nontemporal loads
template <int CX, int NC, int STORE, int H, int LW>
__global__ __launch_bounds__(256) void step(const double *__restrict__ in, double *__restrict__ out, const double *__restrict__ Tg,
                                            int tiles_per_req) {
    constexpr long R = kCells / (CX > NC ? CX : NC);  // r cells per table (input R*CX cells, output R*NC cells)
    constexpr int PADW = NC * 2 + 4;    // dwords per lane row in the transpose buffer (16 B pad)
    __shared__ __attribute__((aligned(16))) double shT[NC * CX * 4];
    __shared__ __attribute__((aligned(16))) uint32_t shX[STORE == 1 ? 4 * 64 * PADW : (STORE >= 3 ? 4 * 64 * 20 : 4)];
    const int tid = threadIdx.x;
    const int req = blockIdx.x / tiles_per_req, tile = blockIdx.x % tiles_per_req;
    const double *__restrict__ F = in + (long)req * kCells;
    double *__restrict__ O = out + (long)req * kCells;
    for (int t = tid; t < NC * CX * 4; t += kWG) shT[t] = Tg[t];
    __syncthreads();
    const int wave = tid >> 6, lane = tid & 63;
    for (int hh = 0; hh < H; ++hh) {
        const long r0 = ((long)tile * H + hh) * (kWG * LW);   // first r cell of this iteration
        const int ctrl = (int)(r0 / (R / 4));
        const double *__restrict__ Tp = shT + ctrl * NC * CX;
        double f[LW][CX];
#pragma unroll
        for (int x = 0; x < CX; ++x) {
            if (LW == 1) {
                f[0][x] = STORE == 2 ? __builtin_nontemporal_load(F + x * R + r0 + tid) : F[x * R + r0 + tid];
            } else {
                const double2 v = *reinterpret_cast<const double2 *>(F + x * R + r0 + 2 * tid);
                f[0][x] = v.x; f[LW - 1][x] = v.y;
            }
        }
#pragma unroll
        for (int c = 0; c < LW; ++c) {
            double acc[NC];
#pragma unroll
            for (int n = 0; n < NC; ++n) {
                double s = 0;
#pragma unroll
                for (int x = 0; x < CX; ++x) s += f[c][x] * Tp[x * NC + n];
                acc[n] = s;
            }
            const long rr = LW == 1 ? r0 + tid : r0 + 2 * tid + c;
            if (NC == 1) {
                O[rr] = acc[0];
            } else if (STORE == 0) {
#pragma unroll
                for (int n = 0; n < NC; n += 2) *reinterpret_cast<double2 *>(O + rr * NC + n) = make_double2(acc[n], acc[n + 1]);
            } else if (STORE == 2) {
#pragma unroll
                for (int n = 0; n < NC; ++n) __builtin_nontemporal_store(acc[n], O + rr * NC + n);
            } else if (STORE == 3 && NC == 16) {
                // production pattern: two rounds of 8 cells per lane, store instructions write 64-byte segments
                uint32_t *X = shX + wave * 64 * 20;
                double *Ob = O + (r0 + wave * 64) * 16;
#pragma unroll
                for (int half = 0; half < 2; ++half) {
#pragma unroll
                    for (int n = 0; n < 8; n += 2) *reinterpret_cast<double2 *>(X + lane * 20 + 2 * n) = make_double2(acc[half * 8 + n], acc[half * 8 + n + 1]);
                    __builtin_amdgcn_wave_barrier();
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const int g = k * 64 + lane;
                        const int owner = g >> 2, piece = g & 3;
                        const double2 v = *reinterpret_cast<const double2 *>(X + owner * 20 + 4 * piece);
                        *reinterpret_cast<double2 *>(Ob + owner * 16 + half * 8 + 2 * piece) = v;
                    }
                    __builtin_amdgcn_wave_barrier();
                }
            } else if (STORE == 4 && NC == 16) {
                // two rounds of 32 lanes x 16 cells: every store instruction writes 1 KiB contiguous
                uint32_t *X = shX + wave * 64 * 20;   // 32 rows of 36 dwords fit in 64 * 20
                double *Ob = O + (r0 + wave * 64) * 16;
#pragma unroll
                for (int half = 0; half < 2; ++half) {
                    if ((lane >> 5) == half) {
#pragma unroll
                        for (int n = 0; n < 16; n += 2) *reinterpret_cast<double2 *>(X + (lane & 31) * 36 + 2 * n) = make_double2(acc[n], acc[n + 1]);
                    }
                    __builtin_amdgcn_wave_barrier();
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const int g = k * 64 + lane;             // chunk of this round's 4 KiB
                        const int owner = g >> 3, piece = g & 7;
                        const double2 v = *reinterpret_cast<const double2 *>(X + owner * 36 + 4 * piece);
                        *reinterpret_cast<double2 *>(Ob + half * 512 + 2 * g) = v;
                    }
                    __builtin_amdgcn_wave_barrier();
                }
            } else {
                // wave-private transpose: lane row -> 16-byte chunks in global order
                uint32_t *X = shX + wave * 64 * PADW;
#pragma unroll
                for (int n = 0; n < NC; n += 2) *reinterpret_cast<double2 *>(X + lane * PADW + 2 * n) = make_double2(acc[n], acc[n + 1]);
                __builtin_amdgcn_wave_barrier();
                double *Ob = O + (r0 + (LW == 1 ? wave * 64 : 0)) * NC;  // (LW == 2 not supported in this mode)
#pragma unroll
                for (int k = 0; k < NC / 2; ++k) {
                    const int g = k * 64 + lane;            // 16-byte chunk index inside the wave's region
                    const int owner = g / (NC / 2), piece = g % (NC / 2);
                    const double2 v = *reinterpret_cast<const double2 *>(X + owner * PADW + 4 * piece);
                    *reinterpret_cast<double2 *>(Ob + 2 * g) = v;
                }
                __builtin_amdgcn_wave_barrier();
            }
        }
    }
}

template <class K>
double time_ms(K launch, int reps) {
    hipEvent_t a, b;
    CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
    launch();
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(a));
    for (int i = 0; i < reps; ++i) launch();
    CHECK(hipEventRecord(b));
    CHECK(hipEventSynchronize(b));
    float ms;
    CHECK(hipEventElapsedTime(&ms, a, b));
    CHECK(hipGetLastError());
    return ms / reps;
}

int main(int argc, char **argv) {
    const int nreq = argc > 1 ? atoi(argv[1]) : 1024;  // 1024 x 8 MiB = 8 GiB in + 8 GiB out
    double *in, *out, *T;
    const long n = (long)nreq * kCells;
    CHECK(hipMalloc(&in, n * 8)); CHECK(hipMalloc(&out, n * 8)); CHECK(hipMalloc(&T, 16 * 16 * 4 * 8));
    CHECK(hipMemset(in, 0, n * 8)); CHECK(hipMemset(out, 0, n * 8));
    std::vector<double> hT(16 * 16 * 4, 0.25);
    CHECK(hipMemcpy(T, hT.data(), hT.size() * 8, hipMemcpyHostToDevice));
    const double gb = 2.0 * n * 8 / 1e9;
    auto report2 = [&](const char *name, double ms, double g) { printf("%-58s %8.3f ms  %8.1f GB/s\n", name, ms, g / ms * 1e3); fflush(stdout); };
    auto report = [&](const char *name, double ms) { report2(name, ms, gb); };
    report("copy 16 B/lane grid-stride (256 CUs x 8 WGs)", time_ms([&] { hipLaunchKernelGGL(copy16, dim3(256 * 8), dim3(256), 0, 0, (const double2 *)in, (double2 *)out, n / 2); }, 5));
    report("copy 16 B/lane grid-stride (256 CUs x 32 WGs)", time_ms([&] { hipLaunchKernelGGL(copy16, dim3(256 * 32), dim3(256), 0, 0, (const double2 *)in, (double2 *)out, n / 2); }, 5));
#define RUN(CX, NC, ST, H, LW, label)                                                                                              \
    {                                                                                                                              \
        const long R = kCells / (CX > NC ? CX : NC);                                                                               \
        const int tiles = (int)(R / ((long)H * kWG * LW));                                                                         \
        report2(label, time_ms([&] { hipLaunchKernelGGL((step<CX, NC, ST, H, LW>), dim3(nreq * tiles), dim3(256), 0, 0, in, out, T, tiles); }, 5), (double)nreq * R * (CX + NC) * 8 / 1e9); \
    }
    RUN(4, 4, 0, 128, 1, "cx4 nc4 fiber-store 8B loads H=128 (current shape)");
    RUN(4, 4, 0, 32, 1, "cx4 nc4 fiber-store 8B loads H=32");
    RUN(4, 4, 0, 512, 1, "cx4 nc4 fiber-store 8B loads H=512");
    RUN(4, 4, 2, 128, 1, "cx4 nc4 fiber-store nontemporal ld/st H=128");
    RUN(4, 4, 1, 128, 1, "cx4 nc4 LDS-transposed store 8B loads H=128");
    RUN(4, 4, 0, 64, 2, "cx4 nc4 fiber-store 16B loads (2 cells/lane) H=64");
    RUN(16, 16, 0, 32, 1, "cx16 nc16 fused pair, fiber-store (128 B/lane) H=32");
    RUN(16, 16, 1, 32, 1, "cx16 nc16 fused pair, LDS-transposed store H=32");
    RUN(16, 16, 2, 32, 1, "cx16 nc16 fused pair, nontemporal H=32");
    RUN(16, 16, 1, 8, 1, "cx16 nc16 fused pair, LDS-transposed store H=8");
    RUN(16, 16, 3, 8, 1, "cx16 nc16 fused pair, 2 rounds x 8 cells (64 B segments) H=8");
    RUN(16, 16, 4, 8, 1, "cx16 nc16 fused pair, 2 rounds x 32 lanes (1 KiB stores) H=8");
    RUN(16, 4, 1, 32, 1, "cx16 nc4 (2 eliminated, 1 new), LDS-transposed H=32");
    RUN(4, 16, 1, 32, 1, "cx4 nc16 (1 eliminated, 2 new), LDS-transposed H=32");
    RUN(4, 1, 0, 128, 1, "cx4 nc1 plain sum-out H=128");
    return 0;
}
