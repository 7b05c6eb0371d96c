// Wave-cooperative primitives of the device planner (wave_order.h, wave_emit.h): ONE request per 64-lane wave, the request's
// planning state in LDS, and every loop over axes / factors / vertices / candidate orders spread over the lanes.
//
// The planner code is written once against this small vocabulary and compiled two ways:
//   * for gfx950 (wave_plan_kernel in engine.hip): `for_n` deals the iterations to the lanes, the reductions are DPP / ballot
//     instructions, everything outside a `for_n` body is executed by all lanes alike (wave-uniform values and control flow);
//   * for the host (oracle/plan_sim.cpp, tests only): ONE lane runs every iteration in ascending order, the reductions are the
//     identity.  That is a legal execution of the same program whenever the bodies of a `for_n` are independent of each other -
//     which the reversed-order build (-DMIBN_WAVE_REVERSE: every for_n runs backwards) checks on the CPU - so the host build
//     pins the planner's LOGIC against emit_core.h / order_search.h word for word without a GPU, and the device build differs
//     only in the twenty lines below.
#pragma once
#include <cstdint>

#include "order_search.h"  // B2, MIBN_HD

namespace mibn {
namespace wv {

#if defined(__HIP_DEVICE_COMPILE__)

constexpr int kWidth = 64;
__device__ inline int lane() { return (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)); }
// LDS writes of every lane are visible to every lane behind this point (a wave's LDS operations retire in issue order: the
// fence only keeps the compiler from moving them)
__device__ inline void sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
// the same for GLOBAL memory the wave wrote and reads back itself (the runner-up's order and the first program's work items parked behind the
// program slot, wave_plan_request): the stores have left the wave before any lane loads
__device__ inline void gsync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}
// a value every lane holds alike, told to the compiler (scalar registers, scalar branches)
__device__ inline int uni(int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ inline uint32_t uni(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }
__device__ inline uint64_t uni(uint64_t v) {
    return (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)v) |
           ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(v >> 32)) << 32);
}
__device__ inline bool uni(bool v) { return __builtin_amdgcn_readfirstlane((int)v) != 0; }
__device__ inline int64_t uni(int64_t v) { return (int64_t)uni((uint64_t)v); }
__device__ inline B2 uni(const B2 &s) { B2 r; r.a = uni(s.a); r.b = uni(s.b); return r; }

template <class F> __device__ inline void for_n(int n, F f) {
    for (int i = lane(); i < n; i += kWidth) f(i);
}
// bit i = p(i), i < n <= 64
template <class P> __device__ inline uint64_t mask64(int n, P p) {
    const int i = lane();
    return __builtin_amdgcn_ballot_w64(i < n && p(i));
}
template <class P> __device__ inline B2 mask128(int n, P p) {
    const int i = lane();
    B2 m;
    m.a = __builtin_amdgcn_ballot_w64(i < n && p(i));
    m.b = n > 64 ? __builtin_amdgcn_ballot_w64(i + 64 < n && p(i + 64)) : 0ull;
    return m;
}

namespace detail {
template <int kCtrl, int kRowMask> __device__ inline int dpp(int old, int v) { return __builtin_amdgcn_update_dpp(old, v, kCtrl, kRowMask, 0xf, false); }
// inclusive reduction over the wave; the total ends up in lane 63
template <class Op> __device__ inline int reduce_i32(int v, int identity, Op op) {
    v = op(v, dpp<0x111, 0xf>(identity, v));  // row_shr:1
    v = op(v, dpp<0x112, 0xf>(identity, v));  // row_shr:2
    v = op(v, dpp<0x114, 0xf>(identity, v));  // row_shr:4
    v = op(v, dpp<0x118, 0xf>(identity, v));  // row_shr:8
    v = op(v, dpp<0x142, 0xa>(identity, v));  // row_bcast:15 into rows 1 and 3
    v = op(v, dpp<0x143, 0xc>(identity, v));  // row_bcast:31 into rows 2 and 3
    return __builtin_amdgcn_readlane(v, 63);
}
}  // namespace detail

template <class F> __device__ inline int sum_n(int n, F f) {
    int acc = 0;
    for (int i = lane(); i < n; i += kWidth) acc += f(i);
    return detail::reduce_i32(acc, 0, [](int a, int b) { return a + b; });
}
template <class F> __device__ inline int max_n(int n, F f, int identity) {
    int acc = identity;
    for (int i = lane(); i < n; i += kWidth) { const int v = f(i); acc = v > acc ? v : acc; }
    return detail::reduce_i32(acc, identity, [](int a, int b) { return a > b ? a : b; });
}
// min over i < n of f(i) (unsigned 64-bit keys); ~0 when n == 0
template <class F> __device__ inline uint64_t min_u64_n(int n, F f) {
    uint64_t acc = ~0ull;
    for (int i = lane(); i < n; i += kWidth) { const uint64_t v = f(i); acc = v < acc ? v : acc; }
    // the high words first, then the low words of the lanes that hold the smallest high word (unsigned compares on biased ints)
    auto umin = [](int a, int b) { return (uint32_t)a < (uint32_t)b ? a : b; };
    const uint32_t hi = (uint32_t)detail::reduce_i32((int)(uint32_t)(acc >> 32), -1, umin);
    const uint32_t lo_mine = (uint32_t)(acc >> 32) == hi ? (uint32_t)acc : 0xffffffffu;
    const uint32_t lo = (uint32_t)detail::reduce_i32((int)lo_mine, -1, umin);
    return ((uint64_t)hi << 32) | lo;
}
template <class P> __device__ inline bool any_n(int n, P p) {
    bool a = false;
    for (int i = lane(); i < n; i += kWidth) a = a || p(i);
    return __builtin_amdgcn_ballot_w64(a) != 0;
}
// compaction: emit(i, k) for the k-th i (ascending) with p(i); returns base + their number
template <class P, class E> __device__ inline int compact_n(int n, int base, P p, E emit) {
    const int l = lane();
    for (int i0 = 0; i0 < n; i0 += kWidth) {
        const int i = i0 + l;
        const bool t = i < n && p(i);
        const uint64_t m = __builtin_amdgcn_ballot_w64(t);
        if (t) emit(i, base + (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u)));
        base += __builtin_popcountll(m);
    }
    return base;
}

#else  // ------------------------------------------------------------------------------------------ host: one lane runs everything

constexpr int kWidth = 1;
inline int lane() { return 0; }
inline void sync() {}
inline void gsync() {}
template <class T> inline T uni(T v) { return v; }

template <class F> inline void for_n(int n, F f) {
#if defined(MIBN_WAVE_REVERSE)
    for (int i = n - 1; i >= 0; --i) f(i);
#else
    for (int i = 0; i < n; ++i) f(i);
#endif
}
template <class P> inline uint64_t mask64(int n, P p) {
    uint64_t m = 0;
    for_n(n, [&](int i) { if (p(i)) m |= 1ull << i; });
    return m;
}
template <class P> inline B2 mask128(int n, P p) {
    B2 m;
    for_n(n, [&](int i) { if (p(i)) m.set(i); });
    return m;
}
template <class F> inline int sum_n(int n, F f) {
    int acc = 0;
    for_n(n, [&](int i) { acc += f(i); });
    return acc;
}
template <class F> inline int max_n(int n, F f, int identity) {
    int acc = identity;
    for_n(n, [&](int i) { const int v = f(i); acc = v > acc ? v : acc; });
    return acc;
}
template <class F> inline uint64_t min_u64_n(int n, F f) {
    uint64_t acc = ~0ull;
    for_n(n, [&](int i) { const uint64_t v = f(i); acc = v < acc ? v : acc; });
    return acc;
}
template <class P> inline bool any_n(int n, P p) {
    bool a = false;
    for_n(n, [&](int i) { a = a || p(i); });
    return a;
}
template <class P, class E> inline int compact_n(int n, int base, P p, E emit) {
    for (int i = 0; i < n; ++i)  // (the positions ARE the ascending order: never reversed)
        if (p(i)) emit(i, base++);
    return base;
}

#endif

}  // namespace wv
}  // namespace mibn
