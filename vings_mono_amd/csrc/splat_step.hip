// Per-splat elementwise stages around the rasteriser in the mapper's view step, gfx950:
// parameter activations, and after the rasteriser backward the chain rule through those
// activations fused with the score-weighted gradient scaling and the score bookkeeping.
// One lane per splat, one launch each; all HBM-bound streaming.
//
// Replaces the per-view PyTorch ops of
//   get_property activations      /root/reference/scripts/gaussian/gaussian_base.py:44-48,64-72
//   add_records (score update)    /root/reference/scripts/gaussian/gaussian_model.py:163-168
//   weighting_grad                /root/reference/scripts/gaussian/gaussian_utils.py:56-62
//   visibility for the optimiser  /root/reference/scripts/gaussian/gaussian_base.py:385-402
// (about 60 small launches per view there).
#include "common.hpp"

namespace vings {

__global__ __launch_bounds__(256) void splat_activate_kernel(
    int N, const float* __restrict__ raw_opacity, const float* __restrict__ raw_scaling,
    const float* __restrict__ raw_rotation, float* __restrict__ opacities,
    float* __restrict__ scales, float* __restrict__ rotations) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= N) return;
    opacities[i] = 1.0f / (1.0f + expf(-raw_opacity[i]));
    const float2 ls = reinterpret_cast<const float2*>(raw_scaling)[i];
    reinterpret_cast<float2*>(scales)[i] = make_float2(expf(ls.x), expf(ls.y));
    const float4 q = reinterpret_cast<const float4*>(raw_rotation)[i];
    const float inv = 1.0f / fmaxf(sqrtf(q.x * q.x + q.y * q.y + q.z * q.z + q.w * q.w), 1e-12f);
    reinterpret_cast<float4*>(rotations)[i] = make_float4(q.x * inv, q.y * inv, q.z * inv, q.w * inv);
}

struct FinishArgs {
    const float* opacities; const float* scales; const float* raw_rotation;   // forward values
    const float* d_means3D; const float* d_colors; const float* d_opacities;  // w.r.t. activated
    const float* d_scales; const float* d_rotations;
    const float* cur_scores;      // [N,2] scores.grad of this view
    const int* radii; const uint8_t* stable_mask;
    float* global_scores; float* local_scores;   // [N,2]
    float* kf_max_scores; int64_t* kf_id;        // [N]
    float* g_xyz; float* g_rgb; float* g_opacity; float* g_scaling; float* g_rotation;
    uint8_t* visible;
    int64_t this_kf;
    int bookkeeping;
};

__global__ __launch_bounds__(256) void splat_grad_finish_kernel(int N, FinishArgs a) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= N) return;
    const float s0 = a.cur_scores[2 * i], s1 = a.cur_scores[2 * i + 1];
    const float g0_old = a.global_scores[2 * i];
    const float g0_new = fminf(fmaxf(g0_old + s0, 0.0f), 1e4f);
    // gradient weight: this view's share of the splat's accumulated importance
    const float w = s0 / (g0_new + 1e-6f + s0);
    if (a.bookkeeping) {
        a.global_scores[2 * i] = g0_new;
        a.local_scores[2 * i] += s0;
        a.local_scores[2 * i + 1] = fmaxf(a.local_scores[2 * i + 1], s1);
        if (a.kf_max_scores[i] < s0) { a.kf_max_scores[i] = s0; a.kf_id[i] = a.this_kf; }
    }
    const bool stable = a.stable_mask && a.stable_mask[i];
    a.visible[i] = (a.radii[i] > 0 && !stable) ? 1 : 0;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        a.g_xyz[3 * i + k] = a.d_means3D[3 * i + k] * w;
        a.g_rgb[3 * i + k] = a.d_colors[3 * i + k] * w;
    }
    const float o = a.opacities[i];
    a.g_opacity[i] = a.d_opacities[i] * (o * (1.0f - o)) * w;          // sigmoid'
    const float2 s = reinterpret_cast<const float2*>(a.scales)[i];
    const float2 ds = reinterpret_cast<const float2*>(a.d_scales)[i];
    reinterpret_cast<float2*>(a.g_scaling)[i] = make_float2(ds.x * s.x * w, ds.y * s.y * w);  // exp'
    // x / max(|x|, eps): d/dx = (g - n (n.g)) / |x|
    const float4 q = reinterpret_cast<const float4*>(a.raw_rotation)[i];
    const float4 g = reinterpret_cast<const float4*>(a.d_rotations)[i];
    const float len = sqrtf(q.x * q.x + q.y * q.y + q.z * q.z + q.w * q.w);
    float4 out;
    if (len > 1e-12f) {
        const float inv = 1.0f / len;
        const float nx = q.x * inv, ny = q.y * inv, nz = q.z * inv, nw = q.w * inv;
        const float ng = nx * g.x + ny * g.y + nz * g.z + nw * g.w;
        const float k = inv * w;
        out = make_float4((g.x - nx * ng) * k, (g.y - ny * ng) * k, (g.z - nz * ng) * k,
                          (g.w - nw * ng) * k);
    } else {
        const float k = 1e12f * w;
        out = make_float4(g.x * k, g.y * k, g.z * k, g.w * k);
    }
    reinterpret_cast<float4*>(a.g_rotation)[i] = out;
}

}  // namespace vings

using namespace vings;

extern "C" int vings_splat_activate(int N, const float* raw_opacity, const float* raw_scaling,
                                    const float* raw_rotation, float* opacities, float* scales,
                                    float* rotations, void* stream) {
    VINGS_REQUIRE(N >= 0, "sizes");
    if (N == 0) return VINGS_OK;
    VINGS_REQUIRE(raw_opacity && raw_scaling && raw_rotation && opacities && scales && rotations,
                  "null pointer");
    VINGS_REQUIRE((((uintptr_t)raw_rotation | (uintptr_t)rotations) & 15) == 0 &&
                      (((uintptr_t)raw_scaling | (uintptr_t)scales) & 7) == 0,
                  "rotation arrays must be 16-byte aligned, scaling arrays 8-byte aligned");
    ProfScope prof(PROF_GLUE, (hipStream_t)stream);
    splat_activate_kernel<<<cdiv(N, 256), 256, 0, (hipStream_t)stream>>>(
        N, raw_opacity, raw_scaling, raw_rotation, opacities, scales, rotations);
    VINGS_LAUNCH_CHECK();
    return VINGS_OK;
}

extern "C" int vings_splat_grad_finish(
    int N, const float* opacities, const float* scales, const float* raw_rotation,
    const float* d_means3D, const float* d_colors, const float* d_opacities, const float* d_scales,
    const float* d_rotations, const float* cur_scores, const int* radii,
    const uint8_t* stable_mask, float* global_scores, float* local_scores, float* kf_max_scores,
    int64_t* kf_id, int64_t this_kf, int bookkeeping, float* g_xyz, float* g_rgb, float* g_opacity,
    float* g_scaling, float* g_rotation, uint8_t* visible, void* stream) {
    VINGS_REQUIRE(N >= 0, "sizes");
    if (N == 0) return VINGS_OK;
    VINGS_REQUIRE(opacities && scales && raw_rotation && d_means3D && d_colors && d_opacities &&
                      d_scales && d_rotations && cur_scores && radii && global_scores,
                  "null input");
    VINGS_REQUIRE(g_xyz && g_rgb && g_opacity && g_scaling && g_rotation && visible, "null output");
    VINGS_REQUIRE(!bookkeeping || (local_scores && kf_max_scores && kf_id),
                  "bookkeeping needs the score arrays");
    VINGS_REQUIRE((((uintptr_t)raw_rotation | (uintptr_t)d_rotations | (uintptr_t)g_rotation) & 15) == 0 &&
                      (((uintptr_t)scales | (uintptr_t)d_scales | (uintptr_t)g_scaling) & 7) == 0,
                  "rotation arrays must be 16-byte aligned, scaling arrays 8-byte aligned");
    FinishArgs a;
    a.opacities = opacities; a.scales = scales; a.raw_rotation = raw_rotation;
    a.d_means3D = d_means3D; a.d_colors = d_colors; a.d_opacities = d_opacities;
    a.d_scales = d_scales; a.d_rotations = d_rotations; a.cur_scores = cur_scores;
    a.radii = radii; a.stable_mask = stable_mask; a.global_scores = global_scores;
    a.local_scores = local_scores; a.kf_max_scores = kf_max_scores; a.kf_id = kf_id;
    a.g_xyz = g_xyz; a.g_rgb = g_rgb; a.g_opacity = g_opacity; a.g_scaling = g_scaling;
    a.g_rotation = g_rotation; a.visible = visible; a.this_kf = this_kf;
    a.bookkeeping = bookkeeping;
    ProfScope prof(PROF_GLUE, (hipStream_t)stream);
    splat_grad_finish_kernel<<<cdiv(N, 256), 256, 0, (hipStream_t)stream>>>(N, a);
    VINGS_LAUNCH_CHECK();
    return VINGS_OK;
}
