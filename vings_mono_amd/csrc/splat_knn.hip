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

// World position and colour of the chosen pixels of a new keyframe: pixel (row u, column v)
// with depth z goes to the camera point ((v-cv)/fv*z, (u-cu)/fu*z, z) and through the
// camera-to-world matrix (tf.py:59-68,46-57 of the reference; gaussian_utils.py:39-46).
struct Pose34 { float m[12]; };

__global__ __launch_bounds__(256) void keyframe_backproject_kernel(
        int n, int W, int HW, const int64_t* __restrict__ pixel_ids, const float* __restrict__ depth,
        const float* __restrict__ rgb, Pose34 c2w, float fu, float fv, float cu, float cv,
        float* __restrict__ xyz, float* __restrict__ colors) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int64_t pix = pixel_ids[i];
    const int u = (int)(pix / W), v = (int)(pix % W);
    const float z = depth[pix];
    const float x = ((float)v - cv) / fv * z, y = ((float)u - cu) / fu * z;
    for (int r = 0; r < 3; ++r)
        xyz[3 * i + r] = c2w.m[4 * r] * x + c2w.m[4 * r + 1] * y + c2w.m[4 * r + 2] * z + c2w.m[4 * r + 3];
    for (int c = 0; c < 3; ++c) colors[3 * i + c] = rgb[(int64_t)c * HW + pix];
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

extern "C" int vings_keyframe_backproject(int n, int H, int W, const int64_t* pixel_ids,
                                          const float* depth, const float* rgb,
                                          const float* c2w_host, float fu, float fv, float cu,
                                          float cv, float* xyz, float* colors, void* stream) {
    VINGS_REQUIRE(n >= 0 && H > 0 && W > 0, "sizes");
    if (n == 0) return VINGS_OK;
    VINGS_REQUIRE(pixel_ids && depth && rgb && c2w_host && xyz && colors, "null pointer");
    VINGS_REQUIRE(fu != 0.f && fv != 0.f, "focal length");
    Pose34 pose;
    for (int k = 0; k < 12; ++k) pose.m[k] = c2w_host[k];
    ProfScope prof(PROF_GLUE, (hipStream_t)stream);
    keyframe_backproject_kernel<<<cdiv(n, 256), 256, 0, (hipStream_t)stream>>>(
        n, W, H * W, pixel_ids, depth, rgb, pose, fu, fv, cu, cv, xyz, colors);
    VINGS_LAUNCH_CHECK();
    return VINGS_OK;
}
