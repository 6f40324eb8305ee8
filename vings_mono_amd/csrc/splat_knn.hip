// Mean squared distance of every point to its three nearest neighbours (itself excluded),
// the quantity new splats take their initial scale from.  gfx950, brute force: one lane per
// query, candidates streamed through LDS in tiles of 256 (every lane reads the same LDS
// address: broadcast).  The reference computes it with a k-d tree on the host
// (distCUDA2, /root/reference/scripts/gaussian/gaussian_utils.py:12-16, called at
// gaussian_model.py:72,139) -- a device round trip per new keyframe; N^2 at the N <= 4e5 of
// that path is a few ms of ALU here.
#include "common.hpp"

namespace vings {

__global__ __launch_bounds__(256) void knn3_mean_dist2_kernel(int N, const float* __restrict__ xyz,
                                                              float* __restrict__ out) {
    __shared__ float4 tile[256];
    const int i = blockIdx.x * 256 + threadIdx.x;
    float qx = 0.f, qy = 0.f, qz = 0.f;
    if (i < N) { qx = xyz[3 * i]; qy = xyz[3 * i + 1]; qz = xyz[3 * i + 2]; }
    float b0 = INFINITY, b1 = INFINITY, b2 = INFINITY;   // b0 <= b1 <= b2
    for (int base = 0; base < N; base += 256) {
        const int j = base + threadIdx.x;
        __syncthreads();
        tile[threadIdx.x] = j < N ? make_float4(xyz[3 * j], xyz[3 * j + 1], xyz[3 * j + 2], 0.f)
                                  : make_float4(0.f, 0.f, 0.f, 0.f);
        __syncthreads();
        const int cnt = min(256, N - base);
        for (int t = 0; t < cnt; ++t) {
            const float4 p = tile[t];
            const float dx = p.x - qx, dy = p.y - qy, dz = p.z - qz;
            float d = dx * dx + dy * dy + dz * dz;
            if (base + t == i) d = INFINITY;               // not its own neighbour
            // insert into the sorted triple
            const float m0 = fminf(b0, d), x0 = fmaxf(b0, d);
            const float m1 = fminf(b1, x0), x1 = fmaxf(b1, x0);
            b0 = m0; b1 = m1; b2 = fminf(b2, x1);
        }
    }
    if (i < N) out[i] = (b0 + b1 + b2) / 3.0f;
}

}  // namespace vings

using namespace vings;

extern "C" int vings_knn3_mean_dist2(int N, const float* xyz, float* mean_dist2, void* stream) {
    VINGS_REQUIRE(N >= 0, "sizes");
    if (N == 0) return VINGS_OK;
    VINGS_REQUIRE(N >= 4, "needs at least 4 points (three neighbours each)");
    VINGS_REQUIRE(xyz && mean_dist2, "null pointer");
    ProfScope prof(PROF_GLUE, (hipStream_t)stream);
    knn3_mean_dist2_kernel<<<cdiv(N, 256), 256, 0, (hipStream_t)stream>>>(N, xyz, mean_dist2);
    VINGS_LAUNCH_CHECK();
    return VINGS_OK;
}
