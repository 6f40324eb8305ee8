// Row-masked Adam for splat parameters on gfx950 (HBM-bound elementwise op).
//
// Rows whose visibility flag is 0 are not read beyond the flag and not written:
// their moments do not decay.  No bias correction, no weight decay.
// Serves SparseGaussianAdam.step as called at
// /root/reference/scripts/gaussian/gaussian_base.py:62,402 and
// /root/reference/scripts/gaussian/sky_utils.py:181.
#include "common.hpp"

namespace vings {

__device__ __forceinline__ void adam_update(float& p, float g, float& m, float& v, float lr,
                                            float b1, float b2, float eps) {
    m = b1 * m + (1.0f - b1) * g;
    v = b2 * v + (1.0f - b2) * g * g;
    p += -lr * m / (sqrtf(v) + eps);
}

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
    int n;
};

// One lane per splat row: the visibility byte is read once and all groups of
// that row are updated by the same lane.
__global__ __launch_bounds__(256) void sparse_adam_multi_kernel(AdamGroups G,
                                                                const uint8_t* __restrict__ visible,
                                                                float b1, float b2, float eps,
                                                                int N) {
    const int stride = gridDim.x * 256;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < N; i += stride) {
        if (!visible[i]) continue;
        for (int gidx = 0; gidx < G.n; ++gidx) {
            const int M = G.width[gidx];
            const float lr = G.lr[gidx];
            float* p = G.param[gidx] + (size_t)i * M;
            const float* g = G.grad[gidx] + (size_t)i * M;
            float* m = G.m[gidx] + (size_t)i * M;
            float* v = G.v[gidx] + (size_t)i * M;
            for (int k = 0; k < M; ++k) {
                float pp = p[k], mm = m[k], vv = v[k];
                adam_update(pp, g[k], mm, vv, lr, b1, b2, eps);
                p[k] = pp; m[k] = mm; v[k] = vv;
            }
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

extern "C" int vings_sparse_adam_multi(int n_groups, float* const* params_host,
                                       const float* const* grads_host,
                                       float* const* exp_avgs_host,
                                       float* const* exp_avg_sqs_host, const int* widths_host,
                                       const float* lrs_host, const uint8_t* visible, float b1,
                                       float b2, float eps, int N, void* stream) {
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
    int blocks = cdiv(N, 256);
    if (blocks > 4096) blocks = 4096;
    ProfScope prof(PROF_ADAM, (hipStream_t)stream);
    sparse_adam_multi_kernel<<<blocks, 256, 0, (hipStream_t)stream>>>(G, visible, b1, b2, eps, N);
    VINGS_LAUNCH_CHECK();
    return VINGS_OK;
}
