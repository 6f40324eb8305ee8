// Chain rule through the parameter activations, score-weighted gradient scaling, score
// bookkeeping and the optimiser's visibility flag for ONE splat, shared by the stand-alone
// kernel (splat_step.hip) and the one fused behind the rasteriser's per-splat backward
// (surfel_project.hip).  Same float32 expression order in both, so they agree bit for bit.
//   add_records (score update)    /root/reference/scripts/gaussian/gaussian_model.py:163-168
//   weighting_grad                /root/reference/scripts/gaussian/gaussian_utils.py:56-62
//   visibility for the optimiser  /root/reference/scripts/gaussian/gaussian_base.py:385-402
#pragma once
#include "common.hpp"

namespace vings {

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
    // Multi-GPU: when `packed` is set the weighted gradients, s0 and the visibility flag go
    // straight into the planar exchange buffer (packed[c * N + i], 15 columns) instead of g_*
    // and `visible`, and the record keys are folded into record_keys [N,2] (see splat_step.hip).
    float* packed; long long* record_keys; int N;
    // Compact form of the same: only the rows some rank touched (slot_of_row[i] >= 0, the
    // order-preserving rank of row i among them) are exchanged.  Slot s lives in chunk
    // c = s / chunk_rows at packed[(c * 14 + column) * chunk_rows + (s - c * chunk_rows)]: 14
    // columns, planar inside a chunk, chunks back to back, so that each chunk is one
    // contiguous all-reduce; the visibility flag needs no column (a row is visible somewhere
    // exactly when it has a slot and is not stable).  rows_of_slot[s] = i is written here too.
    const int32_t* slot_of_row; int32_t* rows_of_slot; int chunk_rows;
};

// dm / dc / dop / ds / g: gradients w.r.t. the position, colour and the ACTIVATED opacity,
// scales and rotation of splat i; s0 / s1: the view's importance and error score.
__device__ __forceinline__ void splat_finish(int i, const FinishArgs& a, const float dm[3],
                                             const float dc[3], float dop, float2 ds, float4 g,
                                             float s0, float s1) {
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
    const bool vis = a.radii[i] > 0 && !stable;
    const float o = a.opacities[i];
    const float g_op = dop * (o * (1.0f - o)) * w;          // sigmoid'
    const float2 s = reinterpret_cast<const float2*>(a.scales)[i];
    const float2 g_sc = make_float2(ds.x * s.x * w, ds.y * s.y * w);  // exp'
    // x / max(|x|, eps): d/dx = (g - n (n.g)) / |x|
    const float4 q = reinterpret_cast<const float4*>(a.raw_rotation)[i];
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
    if (a.packed && a.slot_of_row) {
        const int slot = a.slot_of_row[i];
        if (slot < 0) return;                      // no rank touched this splat in this step
        const int c = slot / a.chunk_rows;
        float* __restrict__ dst = a.packed + ((size_t)c * 14 * a.chunk_rows + (slot - c * a.chunk_rows));
        const float v[14] = {dm[0] * w, dm[1] * w, dm[2] * w, dc[0] * w, dc[1] * w, dc[2] * w, g_op,
                             g_sc.x, g_sc.y, out.x, out.y, out.z, out.w, s0};
#pragma unroll
        for (int k = 0; k < 14; ++k) dst[(size_t)k * a.chunk_rows] = v[k];
        a.rows_of_slot[slot] = i;
        if (a.radii[i] > 0) {                      // a splat this rank did not touch has no scores
            const long long b0 = (long long)__float_as_uint(fmaxf(s0, 0.f));
            const long long b1 = (long long)__float_as_uint(fmaxf(s1, 0.f));
            const long long k0 = (b0 << 32) | ((long long)a.this_kf & 0xFFFFFFFFll);
            a.record_keys[2 * i] = max(a.record_keys[2 * i], k0);
            a.record_keys[2 * i + 1] = max(a.record_keys[2 * i + 1], b1);
        }
        return;
    }
    if (a.packed) {
        const float v[15] = {dm[0] * w, dm[1] * w, dm[2] * w, dc[0] * w, dc[1] * w, dc[2] * w, g_op,
                             g_sc.x, g_sc.y, out.x, out.y, out.z, out.w, s0, vis ? 1.f : 0.f};
#pragma unroll
        for (int c = 0; c < 15; ++c) a.packed[(size_t)c * a.N + i] = v[c];
        const long long b0 = (long long)__float_as_uint(fmaxf(s0, 0.f));
        const long long b1 = (long long)__float_as_uint(fmaxf(s1, 0.f));
        const long long k0 = (b0 << 32) | ((long long)a.this_kf & 0xFFFFFFFFll);
        a.record_keys[2 * i] = max(a.record_keys[2 * i], k0);
        a.record_keys[2 * i + 1] = max(a.record_keys[2 * i + 1], b1);
        return;
    }
    a.visible[i] = vis ? 1 : 0;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        a.g_xyz[3 * i + k] = dm[k] * w;
        a.g_rgb[3 * i + k] = dc[k] * w;
    }
    a.g_opacity[i] = g_op;
    reinterpret_cast<float2*>(a.g_scaling)[i] = g_sc;
    reinterpret_cast<float4*>(a.g_rotation)[i] = out;
}

}  // namespace vings
