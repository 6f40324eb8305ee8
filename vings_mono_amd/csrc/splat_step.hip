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
#include "splat_finish.hpp"

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

__global__ __launch_bounds__(256) void splat_grad_finish_kernel(int N, FinishArgs a) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= N) return;
    const float dm[3] = {a.d_means3D[3 * i], a.d_means3D[3 * i + 1], a.d_means3D[3 * i + 2]};
    const float dc[3] = {a.d_colors[3 * i], a.d_colors[3 * i + 1], a.d_colors[3 * i + 2]};
    splat_finish(i, a, dm, dc, a.d_opacities[i], reinterpret_cast<const float2*>(a.d_scales)[i],
                 reinterpret_cast<const float4*>(a.d_rotations)[i], a.cur_scores[2 * i],
                 a.cur_scores[2 * i + 1]);
}

// ---- exchange between GPU replicas ----
// packed is planar, 15 columns of N floats: {g_xyz 3, g_rgb 3, g_opacity 1, g_scaling 2,
// g_rotation 4, s0, visible}; it is summed over ranks after every view.  The two bookkeeping
// maxima no view step reads (best keyframe, error score) stay rank-local between merges:
// keys[i] = {bits(max(s0,0)) << 32 | keyframe, bits(max(s1,0))} holds their running maximum
// (non-negative floats order like their bit patterns), is maximised over ranks whenever the
// records are needed and then folded into the record arrays by splat_merge_records_kernel.
constexpr int kPackColumns = 15;

struct ExchangeArgs {
    float* g_xyz; float* g_rgb; float* g_opacity; float* g_scaling; float* g_rotation;
    uint8_t* visible;
    float* packed;
};

__global__ __launch_bounds__(256) void splat_pack_kernel(int N, ExchangeArgs a,
                                                         const float* __restrict__ cur_scores,
                                                         long long this_kf,
                                                         long long* __restrict__ keys) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= N) return;
    const float2 s = reinterpret_cast<const float2*>(cur_scores)[i];
    const float2 sc = reinterpret_cast<const float2*>(a.g_scaling)[i];
    const float4 q = reinterpret_cast<const float4*>(a.g_rotation)[i];
    const float v[kPackColumns] = {a.g_xyz[3 * i], a.g_xyz[3 * i + 1], a.g_xyz[3 * i + 2],
                                   a.g_rgb[3 * i], a.g_rgb[3 * i + 1], a.g_rgb[3 * i + 2],
                                   a.g_opacity[i], sc.x, sc.y, q.x, q.y, q.z, q.w, s.x,
                                   a.visible[i] ? 1.f : 0.f};
#pragma unroll
    for (int c = 0; c < kPackColumns; ++c) a.packed[(size_t)c * N + i] = v[c];
    const long long b0 = (long long)__float_as_uint(fmaxf(s.x, 0.f));
    const long long b1 = (long long)__float_as_uint(fmaxf(s.y, 0.f));
    const long long k0 = (b0 << 32) | (this_kf & 0xFFFFFFFFll);
    keys[2 * i] = max(keys[2 * i], k0);
    keys[2 * i + 1] = max(keys[2 * i + 1], b1);
}

__global__ __launch_bounds__(256) void splat_unpack_kernel(int N, ExchangeArgs a,
                                                           float* __restrict__ global_scores,
                                                           float* __restrict__ local_scores) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= N) return;
    float v[kPackColumns];
#pragma unroll
    for (int c = 0; c < kPackColumns; ++c) v[c] = a.packed[(size_t)c * N + i];
    a.g_xyz[3 * i] = v[0]; a.g_xyz[3 * i + 1] = v[1]; a.g_xyz[3 * i + 2] = v[2];
    a.g_rgb[3 * i] = v[3]; a.g_rgb[3 * i + 1] = v[4]; a.g_rgb[3 * i + 2] = v[5];
    a.g_opacity[i] = v[6];
    reinterpret_cast<float2*>(a.g_scaling)[i] = make_float2(v[7], v[8]);
    reinterpret_cast<float4*>(a.g_rotation)[i] = make_float4(v[9], v[10], v[11], v[12]);
    a.visible[i] = v[14] > 0.f ? 1 : 0;
    local_scores[2 * i] += v[13];
    global_scores[2 * i] = fminf(fmaxf(global_scores[2 * i] + v[13], 0.f), 1e4f);
}

__global__ __launch_bounds__(256) void splat_merge_records_kernel(int N, long long* __restrict__ keys,
                                                                  float* __restrict__ local_scores,
                                                                  float* __restrict__ kf_max_scores,
                                                                  int64_t* __restrict__ kf_id) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= N) return;
    const long long k0 = keys[2 * i], k1 = keys[2 * i + 1];
    keys[2 * i] = 0; keys[2 * i + 1] = 0;
    const float best = __uint_as_float((uint32_t)(k0 >> 32));
    local_scores[2 * i + 1] = fmaxf(local_scores[2 * i + 1], __uint_as_float((uint32_t)k1));
    if (kf_max_scores[i] < best) { kf_max_scores[i] = best; kf_id[i] = (int64_t)(k0 & 0xFFFFFFFFll); }
}


// ---- compact exchange: one chunk of summed rows -> score records + Adam ----
// After the all-reduce of a chunk (layout in splat_finish.hpp) one lane per slot adds the
// summed importance score to the records and, unless the splat is stable, applies the Adam
// step to its 13 parameters straight from the summed gradients: the unpack into the gradient
// arrays and the masked Adam pass over all N rows of the planar form fall away.  Slots are in
// row order, so the row accesses of a wave stay within a few cache lines.
struct StepGroups {
    float* p[5]; float* m[5]; float* v[5];   // xyz 3, rgb 3, opacity 1, scaling 2, rotation 4
    float lr[5];
};

template <int M>
__device__ __forceinline__ void adam_row(float* __restrict__ P, float* __restrict__ Mo,
                                         float* __restrict__ Vo, int i, const float* g, float lr,
                                         float b1, float b2, float eps) {
#pragma unroll
    for (int k = 0; k < M; ++k) {
        float p = P[(size_t)M * i + k], m = Mo[(size_t)M * i + k], v = Vo[(size_t)M * i + k];
        adam_update(p, g[k], m, v, lr, b1, b2, eps);
        P[(size_t)M * i + k] = p; Mo[(size_t)M * i + k] = m; Vo[(size_t)M * i + k] = v;
    }
}

__global__ __launch_bounds__(256) void splat_exchange_step_kernel(
    int n_slots, int chunk_rows, const float* __restrict__ chunk, const int32_t* __restrict__ rows,
    StepGroups G, float b1, float b2, float eps, const uint8_t* __restrict__ stable_mask,
    float* __restrict__ global_scores, float* __restrict__ local_scores) {
    const int j = blockIdx.x * 256 + threadIdx.x;
    if (j >= n_slots) return;
    float v[14];
#pragma unroll
    for (int c = 0; c < 14; ++c) v[c] = chunk[(size_t)c * chunk_rows + j];
    const int i = rows[j];
    local_scores[2 * i] += v[13];
    global_scores[2 * i] = fminf(fmaxf(global_scores[2 * i] + v[13], 0.f), 1e4f);
    if (stable_mask && stable_mask[i]) return;
    adam_row<3>(G.p[0], G.m[0], G.v[0], i, v + 0, G.lr[0], b1, b2, eps);
    adam_row<3>(G.p[1], G.m[1], G.v[1], i, v + 3, G.lr[1], b1, b2, eps);
    adam_row<1>(G.p[2], G.m[2], G.v[2], i, v + 6, G.lr[2], b1, b2, eps);
    adam_row<2>(G.p[3], G.m[3], G.v[3], i, v + 7, G.lr[3], b1, b2, eps);
    adam_row<4>(G.p[4], G.m[4], G.v[4], i, v + 9, G.lr[4], b1, b2, eps);
}

__global__ __launch_bounds__(256) void splat_touched_kernel(int N, const int* __restrict__ radii,
                                                            uint8_t* __restrict__ touched) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < N) touched[i] = radii[i] > 0 ? 1 : 0;
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
    FinishArgs a = {};
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

static int exchange_args(int N, float* g_xyz, float* g_rgb, float* g_opacity, float* g_scaling,
                         float* g_rotation, uint8_t* visible, float* packed, ExchangeArgs* a) {
    VINGS_REQUIRE(N >= 0, "sizes");
    if (N == 0) return VINGS_OK;
    VINGS_REQUIRE(g_xyz && g_rgb && g_opacity && g_scaling && g_rotation && visible && packed,
                  "null pointer");
    VINGS_REQUIRE(((uintptr_t)g_rotation & 15) == 0 && ((uintptr_t)g_scaling & 7) == 0,
                  "rotation arrays must be 16-byte aligned, scaling arrays 8-byte aligned");
    a->g_xyz = g_xyz; a->g_rgb = g_rgb; a->g_opacity = g_opacity; a->g_scaling = g_scaling;
    a->g_rotation = g_rotation; a->visible = visible; a->packed = packed;
    return VINGS_OK;
}

extern "C" int vings_splat_pack_columns(void) { return kPackColumns; }

extern "C" int vings_splat_pack_exchange(int N, const float* g_xyz, const float* g_rgb,
                                         const float* g_opacity, const float* g_scaling,
                                         const float* g_rotation, const uint8_t* visible,
                                         const float* cur_scores, int64_t this_kf, float* packed,
                                         int64_t* record_keys, void* stream) {
    ExchangeArgs a;
    const int rc = exchange_args(N, const_cast<float*>(g_xyz), const_cast<float*>(g_rgb),
                                 const_cast<float*>(g_opacity), const_cast<float*>(g_scaling),
                                 const_cast<float*>(g_rotation), const_cast<uint8_t*>(visible),
                                 packed, &a);
    if (rc != VINGS_OK || N == 0) return rc;
    VINGS_REQUIRE(cur_scores && record_keys, "null pointer");
    VINGS_REQUIRE((((uintptr_t)cur_scores | (uintptr_t)record_keys) & 7) == 0,
                  "score and key arrays must be 8-byte aligned");
    ProfScope prof(PROF_GLUE, (hipStream_t)stream);
    splat_pack_kernel<<<cdiv(N, 256), 256, 0, (hipStream_t)stream>>>(
        N, a, cur_scores, (long long)this_kf, reinterpret_cast<long long*>(record_keys));
    VINGS_LAUNCH_CHECK();
    return VINGS_OK;
}

extern "C" int vings_splat_unpack_exchange(int N, const float* packed, float* global_scores,
                                           float* local_scores, float* g_xyz, float* g_rgb,
                                           float* g_opacity, float* g_scaling, float* g_rotation,
                                           uint8_t* visible, void* stream) {
    ExchangeArgs a;
    const int rc = exchange_args(N, g_xyz, g_rgb, g_opacity, g_scaling, g_rotation, visible,
                                 const_cast<float*>(packed), &a);
    if (rc != VINGS_OK || N == 0) return rc;
    VINGS_REQUIRE(global_scores && local_scores, "null pointer");
    ProfScope prof(PROF_GLUE, (hipStream_t)stream);
    splat_unpack_kernel<<<cdiv(N, 256), 256, 0, (hipStream_t)stream>>>(N, a, global_scores,
                                                                       local_scores);
    VINGS_LAUNCH_CHECK();
    return VINGS_OK;
}

extern "C" int vings_splat_merge_records(int N, int64_t* record_keys, float* local_scores,
                                         float* kf_max_scores, int64_t* kf_id, void* stream) {
    VINGS_REQUIRE(N >= 0, "sizes");
    if (N == 0) return VINGS_OK;
    VINGS_REQUIRE(record_keys && local_scores && kf_max_scores && kf_id, "null pointer");
    ProfScope prof(PROF_GLUE, (hipStream_t)stream);
    splat_merge_records_kernel<<<cdiv(N, 256), 256, 0, (hipStream_t)stream>>>(
        N, reinterpret_cast<long long*>(record_keys), local_scores, kf_max_scores, kf_id);
    VINGS_LAUNCH_CHECK();
    return VINGS_OK;
}

extern "C" int vings_splat_exchange_columns(void) { return 14; }

extern "C" int vings_splat_touched(int N, const int32_t* radii, uint8_t* touched, void* stream) {
    VINGS_REQUIRE(N >= 0, "sizes");
    if (N == 0) return VINGS_OK;
    VINGS_REQUIRE(radii && touched, "null pointer");
    ProfScope prof(PROF_GLUE, (hipStream_t)stream);
    splat_touched_kernel<<<cdiv(N, 256), 256, 0, (hipStream_t)stream>>>(N, radii, touched);
    VINGS_LAUNCH_CHECK();
    return VINGS_OK;
}

extern "C" int vings_splat_exchange_step(int n_slots, int chunk_rows, const float* chunk,
                                         const int32_t* rows, float* const* params_host,
                                         float* const* exp_avgs_host,
                                         float* const* exp_avg_sqs_host, const float* lrs_host,
                                         float b1, float b2, float eps, const uint8_t* stable_mask,
                                         float* global_scores, float* local_scores, void* stream) {
    VINGS_REQUIRE(n_slots >= 0 && chunk_rows >= n_slots, "a chunk holds at most chunk_rows slots");
    if (n_slots == 0) return VINGS_OK;
    VINGS_REQUIRE(chunk && rows && params_host && exp_avgs_host && exp_avg_sqs_host && lrs_host &&
                      global_scores && local_scores,
                  "null pointer");
    StepGroups G;
    for (int g = 0; g < 5; ++g) {
        VINGS_REQUIRE(params_host[g] && exp_avgs_host[g] && exp_avg_sqs_host[g], "group pointers");
        G.p[g] = params_host[g]; G.m[g] = exp_avgs_host[g]; G.v[g] = exp_avg_sqs_host[g];
        G.lr[g] = lrs_host[g];
    }
    ProfScope prof(PROF_ADAM, (hipStream_t)stream);
    splat_exchange_step_kernel<<<cdiv(n_slots, 256), 256, 0, (hipStream_t)stream>>>(
        n_slots, chunk_rows, chunk, rows, G, b1, b2, eps, stable_mask, global_scores, local_scores);
    VINGS_LAUNCH_CHECK();
    return VINGS_OK;
}
