// Shared host/device helpers for libvings_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <unistd.h>
#include <cstdio>
#include <stdint.h>
#include <stdio.h>

#include "../../include/vings_hip.h"

namespace vings {

void set_error(const char* fmt, ...);

#define VINGS_CHECK_HIP(expr)                                                             \
    do {                                                                                  \
        hipError_t _e = (expr);                                                           \
        if (_e != hipSuccess) {                                                           \
            ::vings::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e),     \
                               __FILE__, __LINE__);                                       \
            return VINGS_ERR_HIP;                                                         \
        }                                                                                 \
    } while (0)

#define VINGS_REQUIRE(cond, msg)                                                          \
    do {                                                                                  \
        if (!(cond)) {                                                                    \
            ::vings::set_error("argument check failed: %s (%s)", #cond, msg);             \
            return VINGS_ERR_ARG;                                                         \
        }                                                                                 \
    } while (0)

// With VINGS_TRACE_SYNC=1 in the environment every launch is followed by a device
// synchronisation and logged to stderr before and after it: a kernel that never finishes is
// the last "queued" line without its "done".
bool trace_sync();
#define VINGS_LAUNCH_CHECK()                                                              \
    do {                                                                                  \
        VINGS_CHECK_HIP(hipGetLastError());                                               \
        if (::vings::trace_sync()) {                                                      \
            fprintf(stderr, "[vings %d] %s:%d queued\n", (int)getpid(), __FILE__, __LINE__); \
            VINGS_CHECK_HIP(hipDeviceSynchronize());                                      \
            fprintf(stderr, "[vings %d] %s:%d done\n", (int)getpid(), __FILE__, __LINE__); \
        }                                                                                 \
    } while (0)

static inline int cdiv(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }

// ---- optional per-kernel HIP-event timing (off by default) -------------------
enum ProfTag {
    PROF_PROJECT_FWD = 0, PROF_BIN, PROF_BLEND_FWD, PROF_BLEND_BWD, PROF_PROJECT_BWD, PROF_ADAM,
    PROF_DBA_JAC, PROF_DBA_SCHUR, PROF_DBA_SOLVE, PROF_DBA_RETRACT, PROF_AUX, PROF_LOSS, PROF_GLUE, PROF_NTAGS
};
bool prof_enabled();
// Record an event on `s`; the pair (begin,end) is attributed to `tag`.
void prof_begin(int tag, hipStream_t s);
void prof_end(int tag, hipStream_t s);
struct ProfScope {
    int tag; hipStream_t s; bool on;
    ProfScope(int t, hipStream_t st) : tag(t), s(st), on(prof_enabled()) { if (on) prof_begin(tag, s); }
    ~ProfScope() { if (on) prof_end(tag, s); }
};

// Workgroups are dealt round-robin over the 8 XCDs, each with its own L2, so blocks b and b+8
// share one.  This bijection hands every such group one contiguous run of the n work items
// (neighbouring image tiles, which read the same splat records, then meet in one L2).
__device__ __forceinline__ int xcd_contiguous(int b, int n) {
    const int q = n >> 3, r = n & 7, g = b & 7;
    return (g < r ? g * (q + 1) : r * (q + 1) + (g - r) * q) + (b >> 3);
}

// ---- wave64 primitives -----------------------------------------------------
constexpr int kWave = 64;

__device__ __forceinline__ int lane_id() {
    return __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
}

// v_rcp_f32 (1 ulp) in place of the ~10-instruction IEEE division sequence.
__device__ __forceinline__ float rcp_fast(float x) { return __builtin_amdgcn_rcpf(x); }

template <int CTRL>
__device__ __forceinline__ float dpp_f(float v) {
    return __builtin_bit_cast(
        float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, false));
}

// Sum over the 64 lanes of a wave; every lane receives the total.
// xor-1 / xor-2 via quad_perm, then row_half_mirror / row_mirror inside the
// 16-lane DPP row, then two cross-row exchanges.
__device__ __forceinline__ float wave_sum(float v) {
    v += dpp_f<0xB1>(v);   // quad_perm [1,0,3,2]
    v += dpp_f<0x4E>(v);   // quad_perm [2,3,0,1]
    v += dpp_f<0x141>(v);  // row_half_mirror
    v += dpp_f<0x140>(v);  // row_mirror
    v += __shfl_xor(v, 16, 64);
    v += __shfl_xor(v, 32, 64);
    return v;
}

// a + b after v_permlane32_swap: lanes 0-31 receive a[l] + a[l+32], lanes 32-63
// receive b[l-32] + b[l].  The two results of the swap builtin pass through an empty
// asm because hipcc (ROCm 7.2) otherwise folds r[0] + r[1] into 2 * r[0].
__device__ __forceinline__ float swap32_add(float a, float b) {
    const auto r = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(unsigned, a),
                                                    __builtin_bit_cast(unsigned, b), false, false);
    unsigned r0 = r[0], r1 = r[1];
    asm volatile("" : "+v"(r0), "+v"(r1));
    return __builtin_bit_cast(float, r0) + __builtin_bit_cast(float, r1);
}
// a + b after v_permlane16_swap: even 16-lane rows receive a[row] + a[row+1], odd rows
// receive b[row-1] + b[row].
__device__ __forceinline__ float swap16_add(float a, float b) {
    const auto r = __builtin_amdgcn_permlane16_swap(__builtin_bit_cast(unsigned, a),
                                                    __builtin_bit_cast(unsigned, b), false, false);
    unsigned r0 = r[0], r1 = r[1];
    asm volatile("" : "+v"(r0), "+v"(r1));
    return __builtin_bit_cast(float, r0) + __builtin_bit_cast(float, r1);
}

__device__ __forceinline__ uint32_t wave_sum_u32(uint32_t v) {
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// Inclusive prefix sum over the wave.
__device__ __forceinline__ uint32_t wave_scan_incl(uint32_t v) {
    const int l = lane_id();
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        uint32_t t = __shfl_up(v, o, 64);
        if (l >= o) v += t;
    }
    return v;
}

// One element of the row-masked Adam step (no bias correction, no weight decay); shared by
// every kernel that applies it so that they agree bit for bit.
__device__ __forceinline__ void adam_update(float& p, float g, float& m, float& v, float lr,
                                            float b1, float b2, float eps) {
    m = b1 * m + (1.0f - b1) * g;
    v = b2 * v + (1.0f - b2) * g * g;
    p += -lr * m / (sqrtf(v) + eps);
}

}  // namespace vings
