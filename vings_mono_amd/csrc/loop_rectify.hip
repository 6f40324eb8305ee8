// Per-splat stages of loop closure on gfx950 (one lane per splat, HBM-bound):
//  * top-3 keyframe record of every splat, updated with the importance of one more view
//    (calc_score_v1, /root/reference/scripts/loop/loop_model.py:169-214);
//  * rigid re-anchoring of the splats owned by the keyframes of the closed loop: position
//    through the old camera frame, scaled, out through the rectified pose; orientation by the
//    pose change (rectify_gaussians, /root/reference/scripts/loop/loop_rectify.py:322-367).
// The importance itself comes from vings_surfel_blend_importance, the deletion of weak
// splats (loop_rectify.py:325,370-380) from vings_splat_compact_*.
#include "common.hpp"

namespace vings {

__global__ __launch_bounds__(256) void splat_top3_update_kernel(int N, const float* __restrict__ cur,
                                                                int kf, float* __restrict__ scores3,
                                                                int32_t* __restrict__ ids3,
                                                                const float* __restrict__ gate,
                                                                float gate_min) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= N) return;
    if (gate && *gate < gate_min) return;     // the whole view is skipped (decided on the device)
    const float c = cur[i];
    float s0 = scores3[3 * i], s1 = scores3[3 * i + 1], s2 = scores3[3 * i + 2];
    int k0 = ids3[3 * i], k1 = ids3[3 * i + 1], k2 = ids3[3 * i + 2];
    // the three masks are taken from the record as it was, then applied in order
    const bool first = s0 < c;
    const bool second = (s0 > c) && (s1 < c);
    const bool third = (s1 > c) && (s2 < c);
    if (first) { s2 = s1; s1 = s0; s0 = c; k2 = k1; k1 = k0; k0 = kf; }
    if (second) { s2 = s1; s1 = c; k2 = k1; k1 = kf; }
    if (third) { s2 = c; k2 = kf; }
    if (first || second || third) {
        scores3[3 * i] = s0; scores3[3 * i + 1] = s1; scores3[3 * i + 2] = s2;
        ids3[3 * i] = k0; ids3[3 * i + 1] = k1; ids3[3 * i + 2] = k2;
    }
}

// Records of R ranks, each built over its own contiguous block of the keyframe sequence
// (rank 0 the earliest): the entries of rank r's record pass through the same strict update,
// best first, ranks in order.  Equal scores therefore keep the earlier keyframe and an empty
// slot (score 0) never enters, which is what the sequential pass over all keyframes leaves.
__global__ __launch_bounds__(256) void splat_top3_merge_kernel(int N, int R,
                                                               const float* __restrict__ scores_all,
                                                               const int32_t* __restrict__ ids_all,
                                                               float* __restrict__ scores3,
                                                               int32_t* __restrict__ ids3) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= N) return;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f;
    int k0 = 0, k1 = 0, k2 = 0;
    for (int r = 0; r < R; ++r) {
        const size_t base = 3 * ((size_t)r * N + i);
        for (int j = 0; j < 3; ++j) {
            const float c = scores_all[base + j];
            const int kf = ids_all[base + j];
            const bool first = s0 < c;
            const bool second = (s0 > c) && (s1 < c);
            const bool third = (s1 > c) && (s2 < c);
            if (first) { s2 = s1; s1 = s0; s0 = c; k2 = k1; k1 = k0; k0 = kf; }
            if (second) { s2 = s1; s1 = c; k2 = k1; k1 = kf; }
            if (third) { s2 = c; k2 = kf; }
        }
    }
    scores3[3 * i] = s0; scores3[3 * i + 1] = s1; scores3[3 * i + 2] = s2;
    ids3[3 * i] = k0; ids3[3 * i + 1] = k1; ids3[3 * i + 2] = k2;
}

// pose tables: old_w2c, new_c2w [K,12] (3x4 row major), dq [K,4] (w,x,y,z) = rotation of
// new_c2w * old_w2c, scale [K]
__global__ __launch_bounds__(256) void splat_reanchor_kernel(
    int N, float* __restrict__ xyz, float* __restrict__ rot, const int32_t* __restrict__ owner,
    int owner_stride, int kf_start, int kf_end, int K, const float* __restrict__ old_w2c,
    const float* __restrict__ new_c2w, const float* __restrict__ dq,
    const float* __restrict__ scale) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= N) return;
    const int kf = owner[(size_t)i * owner_stride];
    if (kf < kf_start || kf > kf_end || kf < 0 || kf >= K) return;
    const float* A = old_w2c + 12 * kf;
    const float* B = new_c2w + 12 * kf;
    const float x = xyz[3 * i], y = xyz[3 * i + 1], z = xyz[3 * i + 2];
    const float s = scale[kf];
    const float cx = (A[0] * x + A[1] * y + A[2] * z + A[3]) * s;
    const float cy = (A[4] * x + A[5] * y + A[6] * z + A[7]) * s;
    const float cz = (A[8] * x + A[9] * y + A[10] * z + A[11]) * s;
    xyz[3 * i] = B[0] * cx + B[1] * cy + B[2] * cz + B[3];
    xyz[3 * i + 1] = B[4] * cx + B[5] * cy + B[6] * cz + B[7];
    xyz[3 * i + 2] = B[8] * cx + B[9] * cy + B[10] * cz + B[11];
    // q' = dq * q / |q| (Hamilton product, w first); stored unit-length like the reference's
    // conversion back from a rotation matrix
    const float4 q = reinterpret_cast<const float4*>(rot)[i];   // w, x, y, z
    const float inv = 1.0f / fmaxf(sqrtf(q.x * q.x + q.y * q.y + q.z * q.z + q.w * q.w), 1e-12f);
    const float qw = q.x * inv, qx = q.y * inv, qy = q.z * inv, qz = q.w * inv;
    const float dw = dq[4 * kf], dx = dq[4 * kf + 1], dy = dq[4 * kf + 2], dz = dq[4 * kf + 3];
    reinterpret_cast<float4*>(rot)[i] =
        make_float4(dw * qw - dx * qx - dy * qy - dz * qz, dw * qx + dx * qw + dy * qz - dz * qy,
                    dw * qy - dx * qz + dy * qw + dz * qx, dw * qz + dx * qy - dy * qx + dz * qw);
}

__device__ __forceinline__ float refreshed_depth(float dsum, float a, float old) {
    float d = dsum / a;
    if (isnan(d) || d == INFINITY) d = 0.f;
    return d > 0.f ? d : old;
}

// One lane per pixel.  The lane at row/column 3 of an 8x8 block also forms the coarse cell:
// it recomputes its three neighbours from the inputs, so no lane waits for another's store.
__global__ __launch_bounds__(256) void depth_refresh_kernel(int H, int W,
                                                            const float* __restrict__ dsum,
                                                            const float* __restrict__ alpha,
                                                            float* __restrict__ dense,
                                                            float* __restrict__ coarse) {
    const int x = blockIdx.x * 64 + (threadIdx.x & 63);
    const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= W || y >= H) return;
    const int p = y * W + x;
    const float mine = refreshed_depth(dsum[p], alpha[p], dense[p]);
    if ((x & 7) == 3 && (y & 7) == 3) {
        const float b = refreshed_depth(dsum[p + 1], alpha[p + 1], dense[p + 1]);
        const float c = refreshed_depth(dsum[p + W], alpha[p + W], dense[p + W]);
        const float d = refreshed_depth(dsum[p + W + 1], alpha[p + W + 1], dense[p + W + 1]);
        coarse[(y >> 3) * (W >> 3) + (x >> 3)] = (mine + b + c + d) / 4.0f;
    }
    dense[p] = mine;
}

}  // namespace vings

using namespace vings;

extern "C" int vings_splat_top3_update(int N, const float* cur_importance, int kf_id,
                                       float* scores3, int32_t* kfids3, void* stream) {
    VINGS_REQUIRE(N >= 0, "sizes");
    if (N == 0) return VINGS_OK;
    VINGS_REQUIRE(cur_importance && scores3 && kfids3, "null pointer");
    ProfScope prof(PROF_GLUE, (hipStream_t)stream);
    splat_top3_update_kernel<<<cdiv(N, 256), 256, 0, (hipStream_t)stream>>>(N, cur_importance, kf_id,
                                                                            scores3, kfids3, nullptr, 0.f);
    VINGS_LAUNCH_CHECK();
    return VINGS_OK;
}

extern "C" int vings_splat_top3_update_gated(int N, const float* cur_importance, int kf_id,
                                             float* scores3, int32_t* kfids3, const float* gate,
                                             float gate_min, void* stream) {
    VINGS_REQUIRE(N >= 0, "sizes");
    if (N == 0) return VINGS_OK;
    VINGS_REQUIRE(cur_importance && scores3 && kfids3 && gate, "null pointer");
    ProfScope prof(PROF_GLUE, (hipStream_t)stream);
    splat_top3_update_kernel<<<cdiv(N, 256), 256, 0, (hipStream_t)stream>>>(N, cur_importance, kf_id,
                                                                            scores3, kfids3, gate, gate_min);
    VINGS_LAUNCH_CHECK();
    return VINGS_OK;
}

extern "C" int vings_splat_top3_merge(int N, int R, const float* scores_all, const int32_t* kfids_all,
                                      float* scores3, int32_t* kfids3, void* stream) {
    VINGS_REQUIRE(N >= 0 && R >= 1, "sizes");
    if (N == 0) return VINGS_OK;
    VINGS_REQUIRE(scores_all && kfids_all && scores3 && kfids3, "null pointer");
    ProfScope prof(PROF_GLUE, (hipStream_t)stream);
    splat_top3_merge_kernel<<<cdiv(N, 256), 256, 0, (hipStream_t)stream>>>(N, R, scores_all, kfids_all,
                                                                           scores3, kfids3);
    VINGS_LAUNCH_CHECK();
    return VINGS_OK;
}

extern "C" int vings_splat_reanchor(int N, float* xyz, float* rotation, const int32_t* owner_kf,
                                    int owner_stride, int kf_start, int kf_end, int K,
                                    const float* old_w2c, const float* new_c2w,
                                    const float* delta_quat, const float* scale, void* stream) {
    VINGS_REQUIRE(N >= 0 && K > 0 && owner_stride > 0 && kf_start <= kf_end, "sizes");
    if (N == 0) return VINGS_OK;
    VINGS_REQUIRE(xyz && rotation && owner_kf && old_w2c && new_c2w && delta_quat && scale,
                  "null pointer");
    VINGS_REQUIRE(((uintptr_t)rotation & 15) == 0, "rotation must be 16-byte aligned");
    ProfScope prof(PROF_GLUE, (hipStream_t)stream);
    splat_reanchor_kernel<<<cdiv(N, 256), 256, 0, (hipStream_t)stream>>>(
        N, xyz, rotation, owner_kf, owner_stride, kf_start, kf_end, K, old_w2c, new_c2w,
        delta_quat, scale);
    VINGS_LAUNCH_CHECK();
    return VINGS_OK;
}

extern "C" int vings_depth_refresh(int H, int W, const float* depth_sum, const float* alpha,
                                   float* dense_map, float* coarse_map, void* stream) {
    VINGS_REQUIRE(H > 0 && W > 0 && H % 8 == 0 && W % 8 == 0, "H and W must be multiples of 8");
    VINGS_REQUIRE(depth_sum && alpha && dense_map && coarse_map, "null pointer");
    ProfScope prof(PROF_GLUE, (hipStream_t)stream);
    depth_refresh_kernel<<<dim3(cdiv(W, 64), cdiv(H, 4)), 256, 0, (hipStream_t)stream>>>(
        H, W, depth_sum, alpha, dense_map, coarse_map);
    VINGS_LAUNCH_CHECK();
    return VINGS_OK;
}
