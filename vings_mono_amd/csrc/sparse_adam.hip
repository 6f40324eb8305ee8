// Row-masked Adam for splat parameters on gfx950 (HBM-bound elementwise op).
//
// Rows whose visibility flag is 0 keep their values: their moments do not decay.
// No bias correction, no weight decay.
// Serves SparseGaussianAdam.step as called at
// /root/reference/scripts/gaussian/gaussian_base.py:62,402 and
// /root/reference/scripts/gaussian/sky_utils.py:181.
#include "common.hpp"

namespace vings {

__global__ __launch_bounds__(256) void sparse_adam_kernel(
    float* __restrict__ param, const float* __restrict__ grad, float* __restrict__ exp_avg,
    float* __restrict__ exp_avg_sq, const uint8_t* __restrict__ visible, float lr, float b1,
    float b2, float eps, int64_t total, int M) {
    const int64_t stride = (int64_t)gridDim.x * 256;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += stride) {
        if (!visible[i / M]) continue;
        float p = param[i], m = exp_avg[i], v = exp_avg_sq[i];
        adam_update(p, grad[i], m, v, lr, b1, b2, eps);
        param[i] = p; exp_avg[i] = m; exp_avg_sq[i] = v;
    }
}

constexpr int kMaxGroups = 8;
struct AdamGroups {
    float* param[kMaxGroups];
    const float* grad[kMaxGroups];
    float* m[kMaxGroups];
    float* v[kMaxGroups];
    int width[kMaxGroups];
    float lr[kMaxGroups];
    int first_block[kMaxGroups + 1];   // blocks [first_block[g], first_block[g+1]) serve group g
    int n;
};

// Row of element e in an array of rows of M floats; the widths that occur are constants.
__device__ __forceinline__ int64_t row_of(int64_t e, int M) {
    switch (M) {
        case 1: return e;
        case 2: return e >> 1;
        case 3: return e / 3;
        case 4: return e >> 2;
        default: return e / M;
    }
}

// All groups in one launch, element-parallel: a lane owns four consecutive floats of one
// group's arrays (one 16-byte load / store per array: fully coalesced, where a lane per row
// would walk rows of 4 to 16 bytes with strided scalar accesses).  A quad none of whose rows
// is visible is not touched at all; in a quad that straddles a visible and an invisible row
// the invisible elements are written back unchanged.
__global__ __launch_bounds__(256) void sparse_adam_multi_kernel(AdamGroups G,
                                                                const uint8_t* __restrict__ visible,
                                                                float b1, float b2, float eps,
                                                                int N,
                                                                const uint32_t* __restrict__ guard_count,
                                                                uint32_t guard_limit) {
    // (see vings_sparse_adam_multi_guarded: a step whose gradients are incomplete is skipped)
    if (guard_count && *guard_count > guard_limit) return;
    int g = 0;
    while (g + 1 < G.n && (int)blockIdx.x >= G.first_block[g + 1]) ++g;
    const int M = G.width[g];
    const float lr = G.lr[g];
    const int64_t total = (int64_t)N * M;
    const int64_t e0 = ((int64_t)(blockIdx.x - G.first_block[g]) * 256 + threadIdx.x) * 4;
    if (e0 >= total) return;
    float* __restrict__ P = G.param[g];
    const float* __restrict__ Gr = G.grad[g];
    float* __restrict__ Mo = G.m[g];
    float* __restrict__ Vo = G.v[g];
    if (e0 + 4 <= total) {
        bool vis[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) vis[k] = visible[row_of(e0 + k, M)] != 0;
        if (!(vis[0] | vis[1] | vis[2] | vis[3])) return;
        float4 p4 = *reinterpret_cast<float4*>(P + e0);
        const float4 g4 = *reinterpret_cast<const float4*>(Gr + e0);
        float4 m4 = *reinterpret_cast<float4*>(Mo + e0);
        float4 v4 = *reinterpret_cast<float4*>(Vo + e0);
        if (vis[0]) adam_update(p4.x, g4.x, m4.x, v4.x, lr, b1, b2, eps);
        if (vis[1]) adam_update(p4.y, g4.y, m4.y, v4.y, lr, b1, b2, eps);
        if (vis[2]) adam_update(p4.z, g4.z, m4.z, v4.z, lr, b1, b2, eps);
        if (vis[3]) adam_update(p4.w, g4.w, m4.w, v4.w, lr, b1, b2, eps);
        *reinterpret_cast<float4*>(P + e0) = p4;
        *reinterpret_cast<float4*>(Mo + e0) = m4;
        *reinterpret_cast<float4*>(Vo + e0) = v4;
    } else {
        for (int64_t e = e0; e < total; ++e) {
            if (!visible[row_of(e, M)]) continue;
            float p = P[e], m = Mo[e], v = Vo[e];
            adam_update(p, Gr[e], m, v, lr, b1, b2, eps);
            P[e] = p; Mo[e] = m; Vo[e] = v;
        }
    }
}

}  // namespace vings

using namespace vings;

extern "C" int vings_sparse_adam(float* param, const float* grad, float* exp_avg,
                                 float* exp_avg_sq, const uint8_t* visible, float lr, float b1,
                                 float b2, float eps, int N, int M, void* stream) {
    VINGS_REQUIRE(N >= 0 && M > 0, "sizes");
    if (N == 0) return VINGS_OK;
    VINGS_REQUIRE(param && grad && exp_avg && exp_avg_sq && visible, "null pointer");
    const int64_t total = (int64_t)N * M;
    int blocks = cdiv(total, 256);
    if (blocks > 2048) blocks = 2048;
    ProfScope prof(PROF_ADAM, (hipStream_t)stream);
    sparse_adam_kernel<<<blocks, 256, 0, (hipStream_t)stream>>>(param, grad, exp_avg, exp_avg_sq,
                                                                visible, lr, b1, b2, eps, total, M);
    VINGS_LAUNCH_CHECK();
    return VINGS_OK;
}

static int adam_multi(int n_groups, float* const* params_host, const float* const* grads_host,
                      float* const* exp_avgs_host, float* const* exp_avg_sqs_host,
                      const int* widths_host, const float* lrs_host, const uint8_t* visible,
                      float b1, float b2, float eps, int N, void* stream,
                      const uint32_t* guard_count, uint32_t guard_limit) {
    VINGS_REQUIRE(n_groups > 0 && n_groups <= kMaxGroups, "1..8 groups");
    VINGS_REQUIRE(N >= 0, "sizes");
    if (N == 0) return VINGS_OK;
    VINGS_REQUIRE(params_host && grads_host && exp_avgs_host && exp_avg_sqs_host && widths_host &&
                      lrs_host && visible,
                  "null pointer");
    AdamGroups G;
    G.n = n_groups;
    for (int i = 0; i < n_groups; ++i) {
        VINGS_REQUIRE(params_host[i] && grads_host[i] && exp_avgs_host[i] && exp_avg_sqs_host[i] &&
                          widths_host[i] > 0,
                      "group pointers");
        G.param[i] = params_host[i];
        G.grad[i] = grads_host[i];
        G.m[i] = exp_avgs_host[i];
        G.v[i] = exp_avg_sqs_host[i];
        G.width[i] = widths_host[i];
        G.lr[i] = lrs_host[i];
    }
    int blocks = 0;
    for (int i = 0; i < n_groups; ++i) {
        VINGS_REQUIRE((((uintptr_t)G.param[i] | (uintptr_t)G.grad[i] | (uintptr_t)G.m[i] |
                        (uintptr_t)G.v[i]) & 15) == 0, "group arrays must be 16-byte aligned");
        G.first_block[i] = blocks;
        blocks += cdiv((int64_t)N * G.width[i], 1024);
    }
    G.first_block[n_groups] = blocks;
    ProfScope prof(PROF_ADAM, (hipStream_t)stream);
    sparse_adam_multi_kernel<<<blocks, 256, 0, (hipStream_t)stream>>>(G, visible, b1, b2, eps, N,
                                                                      guard_count, guard_limit);
    VINGS_LAUNCH_CHECK();
    return VINGS_OK;
}

extern "C" int vings_sparse_adam_multi(int n_groups, float* const* params_host,
                                       const float* const* grads_host,
                                       float* const* exp_avgs_host,
                                       float* const* exp_avg_sqs_host, const int* widths_host,
                                       const float* lrs_host, const uint8_t* visible, float b1,
                                       float b2, float eps, int N, void* stream) {
    return adam_multi(n_groups, params_host, grads_host, exp_avgs_host, exp_avg_sqs_host,
                      widths_host, lrs_host, visible, b1, b2, eps, N, stream, nullptr, 0u);
}

extern "C" int vings_sparse_adam_multi_guarded(int n_groups, float* const* params_host,
                                               const float* const* grads_host,
                                               float* const* exp_avgs_host,
                                               float* const* exp_avg_sqs_host,
                                               const int* widths_host, const float* lrs_host,
                                               const uint8_t* visible, float b1, float b2,
                                               float eps, int N, const uint32_t* instance_count,
                                               uint32_t capacity, void* stream) {
    VINGS_REQUIRE(instance_count != nullptr, "instance_count");
    return adam_multi(n_groups, params_host, grads_host, exp_avgs_host, exp_avg_sqs_host,
                      widths_host, lrs_host, visible, b1, b2, eps, N, stream, instance_count,
                      capacity);
}
