// Tracker -> mapper packaging on the device (gfx950): depth covariance from the reduced
// camera system, convex 8x upsampling of the coarse maps, and the per-keyframe masking /
// packaging step.  All three are bandwidth-bound streaming kernels.
//
// Counterparts in the reference (PyTorch + a NumPy median on the host there):
//   compute_covariances   /root/reference/scripts/frontend/depth_video.py:676-726
//   cvx_upsample          /root/reference/scripts/frontend/droid_net.py:17-38
//   DepthVideo.upsample   /root/reference/scripts/frontend/depth_video.py:236-243
//   judge_and_package_v3  /root/reference/scripts/vings_utils/middleware_utils.py:222-241
#include <algorithm>

#include "common.hpp"

namespace vings {

// --------------------------------------------------------------------------------------
// Depth covariance.  For depth variable k and pixel p:  e = the 6P-vector of coupling rows
// (block_of[k,pose] names the row block of E, -1 = zero),  F = (Q e)^T L_inv,
// out = (Q + |F|^2) / (disp^4 + 1e-6).  L_inv sits in LDS, padded to 8-column groups; every
// lane owns one pixel and walks the columns eight at a time, re-reading its (cached) e.
// --------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void depth_covariance_kernel(
        int HW, int P, int ld, const float* __restrict__ Q, const float* __restrict__ E,
        const int32_t* __restrict__ block_of, const float* __restrict__ L_inv,
        const float* __restrict__ disps, const int64_t* __restrict__ kx, float* __restrict__ out) {
    extern __shared__ float Ls[];                 // [6P][ld] then int block ids [P]
    int* blk = reinterpret_cast<int*>(Ls + 6 * P * ld);
    const int n = 6 * P, k = blockIdx.y;
    for (int t = threadIdx.x; t < n * ld; t += 256) {
        const int r = t / ld, c = t % ld;
        Ls[t] = c < n ? L_inv[r * n + c] : 0.f;
    }
    for (int t = threadIdx.x; t < P; t += 256) blk[t] = block_of[k * P + t];
    __syncthreads();
    const int pix = blockIdx.x * 256 + threadIdx.x;
    if (pix >= HW) return;
    const float q = Q[(int64_t)k * HW + pix];
    float delta = 0.f;
    for (int c0 = 0; c0 < ld; c0 += 8) {
        float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        for (int p = 0; p < P; ++p) {
            const int b = blk[p];
            if (b < 0) continue;
#pragma unroll
            for (int d = 0; d < 6; ++d) {
                const float e = q * E[((int64_t)b * 6 + d) * HW + pix];
                const float4 l0 = *reinterpret_cast<const float4*>(&Ls[(p * 6 + d) * ld + c0]);
                const float4 l1 = *reinterpret_cast<const float4*>(&Ls[(p * 6 + d) * ld + c0 + 4]);
                acc[0] = fmaf(e, l0.x, acc[0]); acc[1] = fmaf(e, l0.y, acc[1]);
                acc[2] = fmaf(e, l0.z, acc[2]); acc[3] = fmaf(e, l0.w, acc[3]);
                acc[4] = fmaf(e, l1.x, acc[4]); acc[5] = fmaf(e, l1.y, acc[5]);
                acc[6] = fmaf(e, l1.z, acc[6]); acc[7] = fmaf(e, l1.w, acc[7]);
            }
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) delta = fmaf(acc[j], acc[j], delta);
    }
    const int64_t o = kx[k] * (int64_t)HW + pix;
    const float d = disps[o], d2 = d * d;
    out[o] = (q + delta) / (d2 * d2 + 1e-6f);
}

// --------------------------------------------------------------------------------------
// Convex upsampling: every coarse cell (y,x) expands to an 8x8 patch; output pixel (i,j) of
// the patch is the softmax(mask[0..8, i, j, y, x])-weighted sum of the 3x3 coarse
// neighbourhood (zero outside the map).  One workgroup covers 32 cells of one coarse row;
// a lane owns one cell and one patch row i, reads its 72 mask values (consecutive lanes read
// consecutive cells of one mask channel) and writes 8 consecutive output pixels per field.
// Up to kMaxFields maps share one pass over the mask, which is 576/(64+9) of the traffic.
// --------------------------------------------------------------------------------------
constexpr int kMaxFields = 4;
struct UpsampleFields { const float* src[kMaxFields]; float* dst[kMaxFields]; };

template <int NF>
__global__ __launch_bounds__(256) void upsample_maps_kernel(
        int ht, int wd, const int64_t* __restrict__ ix, const float* __restrict__ mask,
        UpsampleFields f) {
    const int x = blockIdx.x * 32 + (threadIdx.x & 31);
    const int i = threadIdx.x >> 5, y = blockIdx.y, b = blockIdx.z;
    if (x >= wd) return;
    const int64_t frame = ix[b];
    const int64_t HW = (int64_t)ht * wd;
    float nb[NF][9];
#pragma unroll
    for (int k = 0; k < 9; ++k) {
        const int yy = y + k / 3 - 1, xx = x + k % 3 - 1;
        const bool in = yy >= 0 && yy < ht && xx >= 0 && xx < wd;
#pragma unroll
        for (int s = 0; s < NF; ++s) nb[s][k] = in ? f.src[s][frame * HW + (int64_t)yy * wd + xx] : 0.f;
    }
    const float* m = mask + ((int64_t)b * 576 + i * 8) * HW + (int64_t)y * wd + x;
    float res[NF][8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        float v[9], mx = -INFINITY;
#pragma unroll
        for (int k = 0; k < 9; ++k) { v[k] = m[(int64_t)(k * 64 + j) * HW]; mx = fmaxf(mx, v[k]); }
        float sum = 0.f;
#pragma unroll
        for (int k = 0; k < 9; ++k) { v[k] = expf(v[k] - mx); sum += v[k]; }
#pragma unroll
        for (int k = 0; k < 9; ++k) v[k] = v[k] / sum;
#pragma unroll
        for (int s = 0; s < NF; ++s) {
            float a = 0.f;
#pragma unroll
            for (int k = 0; k < 9; ++k) a += v[k] * nb[s][k];
            res[s][j] = a;
        }
    }
    const int64_t o = frame * HW * 64 + ((int64_t)(8 * y + i) * (8 * wd)) + 8 * x;
#pragma unroll
    for (int s = 0; s < NF; ++s) {
        float4* d = reinterpret_cast<float4*>(f.dst[s] + o);
        d[0] = make_float4(res[s][0], res[s][1], res[s][2], res[s][3]);
        d[1] = make_float4(res[s][4], res[s][5], res[s][6], res[s][7]);
    }
}

// --------------------------------------------------------------------------------------
// Keyframe packaging.  The per-frame median of the covariance map is an exact selection:
// three histogram passes over order-preserving keys (11 + 11 + 10 bits), tracking the two
// middle ranks at once (an even count averages them, as numpy.median does).
// --------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t float_key(float v) {
    const uint32_t b = __float_as_uint(v);
    return b ^ ((b >> 31) ? 0xFFFFFFFFu : 0x80000000u);
}
__device__ __forceinline__ float key_float(uint32_t k) {
    return __uint_as_float(k ^ ((k >> 31) ? 0x80000000u : 0xFFFFFFFFu));
}

struct SelectState { uint32_t prefix[2]; uint32_t rank[2]; };   // per frame

__global__ void select_init_kernel(int n_frames, int64_t count, SelectState* st, uint32_t* max_key) {
    const int f = blockIdx.x * blockDim.x + threadIdx.x;
    if (f == 0) *max_key = 0u;
    if (f >= n_frames) return;
    st[f].prefix[0] = st[f].prefix[1] = 0u;
    st[f].rank[0] = (uint32_t)((count - 1) / 2);
    st[f].rank[1] = (uint32_t)(count / 2);
}

// hist[frame][which][bin]: elements whose already-fixed high bits equal the state's prefix.
__global__ __launch_bounds__(256) void select_hist_kernel(
        int64_t count, int shift, int nbits, uint32_t fixed_mask, const int64_t* __restrict__ ids,
        const float* __restrict__ values, const SelectState* __restrict__ st,
        uint32_t* __restrict__ hist) {
    __shared__ uint32_t h[2][2048];
    const int f = blockIdx.y, bins = 1 << nbits;
    for (int t = threadIdx.x; t < 2 * 2048; t += 256) (&h[0][0])[t] = 0u;
    __syncthreads();
    const uint32_t p0 = st[f].prefix[0], p1 = st[f].prefix[1];
    const bool two = p0 != p1;
    const float* v = values + ids[f] * count;
    for (int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x; t < count; t += (int64_t)gridDim.x * 256) {
        const uint32_t key = float_key(v[t]);
        const uint32_t bin = (key >> shift) & (bins - 1);
        if ((key & fixed_mask) == p0) atomicAdd(&h[0][bin], 1u);
        if (two && (key & fixed_mask) == p1) atomicAdd(&h[1][bin], 1u);
    }
    __syncthreads();
    for (int t = threadIdx.x; t < bins; t += 256) {
        if (h[0][t]) atomicAdd(&hist[((int64_t)f * 2 + 0) * 2048 + t], h[0][t]);
        if (two && h[1][t]) atomicAdd(&hist[((int64_t)f * 2 + 1) * 2048 + t], h[1][t]);
    }
}

// One workgroup per frame: locates the bin holding each tracked rank and extends the prefix.
__global__ __launch_bounds__(256) void select_pick_kernel(int shift, int nbits, SelectState* st,
                                                          const uint32_t* __restrict__ hist,
                                                          float* __restrict__ median, int last) {
    __shared__ uint32_t chunk[2][256];
    const int f = blockIdx.x, bins = 1 << nbits, per = bins / 256;
    const bool two = st[f].prefix[0] != st[f].prefix[1];
    for (int s = 0; s < 2; ++s) {
        const uint32_t* h = hist + ((int64_t)f * 2 + ((s == 1 && two) ? 1 : 0)) * 2048;
        uint32_t sum = 0;
        for (int t = 0; t < per; ++t) sum += h[threadIdx.x * per + t];
        chunk[s][threadIdx.x] = sum;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        float val[2];
        for (int s = 0; s < 2; ++s) {
            const uint32_t* h = hist + ((int64_t)f * 2 + ((s == 1 && two) ? 1 : 0)) * 2048;
            uint32_t rank = st[f].rank[s], c = 0;
            while (c < 255 && rank >= chunk[s][c]) { rank -= chunk[s][c]; ++c; }
            int b = c * per;
            while (b < bins - 1 && rank >= h[b]) { rank -= h[b]; ++b; }
            st[f].rank[s] = rank;
            st[f].prefix[s] |= (uint32_t)b << shift;
            val[s] = key_float(st[f].prefix[s]);
        }
        if (last) median[f] = (val[0] + val[1]) / 2.0f;
    }
}

// depth = 1/(disp+1e-6), zeroed where too far or too uncertain; colour /255, zeroed with the
// depth; running maximum of the covariance over the pixels that keep their depth.
__global__ __launch_bounds__(256) void package_mask_kernel(
        int64_t HW, const int64_t* __restrict__ ids, const uint8_t* __restrict__ images,
        const float* __restrict__ disps_up, const float* __restrict__ cov_up,
        const float* __restrict__ median, float max_depth, float cov_times,
        float* __restrict__ rgbs, float* __restrict__ depths, uint32_t* __restrict__ max_key) {
    const int f = blockIdx.y;
    const int64_t frame = ids[f];
    const float limit = cov_times * median[f];
    // PyTorch divides a device tensor by a scalar as a multiplication by its reciprocal
    const float inv255 = 1.0f / 255.0f;
    uint32_t best = 0u;
    for (int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x; p < HW; p += (int64_t)gridDim.x * 256) {
        float d = 1.0f / (disps_up[frame * HW + p] + 1e-6f);
        const float c = cov_up[frame * HW + p];
        if (d > max_depth || c > limit) d = 0.f;
        depths[f * HW + p] = d;
        const bool keep = d != 0.f;
#pragma unroll
        for (int ch = 0; ch < 3; ++ch)
            rgbs[(f * HW + p) * 3 + ch] = keep ? (float)images[(frame * 3 + ch) * HW + p] * inv255 : 0.f;
        if (d > 0.f) best = max(best, float_key(c));
    }
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) best = max(best, (uint32_t)__shfl_xor((int)best, o, 64));
    if (lane_id() == 0 && best) atomicMax(max_key, best);
}

__global__ __launch_bounds__(256) void package_cov_kernel(
        int64_t HW, const int64_t* __restrict__ ids, const float* __restrict__ cov_up,
        const float* __restrict__ depths, const uint32_t* __restrict__ max_key,
        float* __restrict__ cov_out) {
    const int f = blockIdx.y;
    const int64_t frame = ids[f];
    const float top = key_float(*max_key);
    for (int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x; p < HW; p += (int64_t)gridDim.x * 256)
        cov_out[f * HW + p] = depths[f * HW + p] == 0.f ? top : cov_up[frame * HW + p];
}

}  // namespace vings

using namespace vings;

extern "C" int vings_depth_covariance(int ht, int wd, int P, int K, const float* Q, const float* E,
                                      const int32_t* block_of, const float* L_inv,
                                      const float* disps, const int64_t* kx, float* depths_cov,
                                      void* stream) {
    VINGS_REQUIRE(ht > 0 && wd > 0 && P > 0 && K > 0, "sizes");
    VINGS_REQUIRE(Q && E && block_of && L_inv && disps && kx && depths_cov, "null pointer");
    const int n = 6 * P, ld = (n + 7) / 8 * 8;
    const size_t lds = (size_t)n * ld * sizeof(float) + (size_t)P * sizeof(int);
    VINGS_REQUIRE(lds <= 64 * 1024, "at most 20 free poses (L_inv must fit 64 KB of LDS)");
    ProfScope prof(PROF_AUX, (hipStream_t)stream);
    depth_covariance_kernel<<<dim3(cdiv((int64_t)ht * wd, 256), K), 256, lds, (hipStream_t)stream>>>(
        ht * wd, P, ld, Q, E, block_of, L_inv, disps, kx, depths_cov);
    VINGS_LAUNCH_CHECK();
    return VINGS_OK;
}

extern "C" int vings_upsample_maps(int K, int ht, int wd, const int64_t* ix, const float* mask,
                                   int n_fields, const float* const* src_host,
                                   float* const* dst_host, void* stream) {
    VINGS_REQUIRE(K >= 0 && ht > 0 && wd > 0, "sizes");
    VINGS_REQUIRE(n_fields >= 1 && n_fields <= kMaxFields, "1 to 4 maps per call");
    if (K == 0) return VINGS_OK;
    VINGS_REQUIRE(ix && mask && src_host && dst_host, "null pointer");
    VINGS_REQUIRE(ht <= 65535 && K <= 65535, "grid limits");
    UpsampleFields f{};
    for (int s = 0; s < n_fields; ++s) {
        VINGS_REQUIRE(src_host[s] && dst_host[s], "null map pointer");
        VINGS_REQUIRE(((uintptr_t)dst_host[s] & 15) == 0, "output maps must be 16-byte aligned");
        f.src[s] = src_host[s]; f.dst[s] = dst_host[s];
    }
    ProfScope prof(PROF_AUX, (hipStream_t)stream);
    const dim3 grid(cdiv(wd, 32), ht, K);
    hipStream_t st = (hipStream_t)stream;
    switch (n_fields) {
        case 1: upsample_maps_kernel<1><<<grid, 256, 0, st>>>(ht, wd, ix, mask, f); break;
        case 2: upsample_maps_kernel<2><<<grid, 256, 0, st>>>(ht, wd, ix, mask, f); break;
        case 3: upsample_maps_kernel<3><<<grid, 256, 0, st>>>(ht, wd, ix, mask, f); break;
        default: upsample_maps_kernel<4><<<grid, 256, 0, st>>>(ht, wd, ix, mask, f); break;
    }
    VINGS_LAUNCH_CHECK();
    return VINGS_OK;
}

extern "C" size_t vings_package_workspace_bytes(int n_frames) {
    if (n_frames < 0) return 0;
    return (size_t)n_frames * (2 * 2048 * sizeof(uint32_t) + sizeof(SelectState)) + 256;
}

extern "C" int vings_package_keyframes(int n_frames, int H, int W, const int64_t* frame_ids,
                                       const uint8_t* images, const float* disps_up,
                                       const float* depths_cov_up, float max_depth,
                                       float cov_times, float* rgbs, float* depths,
                                       float* depths_cov, float* cov_median, void* workspace,
                                       size_t workspace_bytes, void* stream) {
    VINGS_REQUIRE(n_frames >= 0 && H > 0 && W > 0, "sizes");
    if (n_frames == 0) return VINGS_OK;
    VINGS_REQUIRE(n_frames <= 65535, "grid limits");
    VINGS_REQUIRE(frame_ids && images && disps_up && depths_cov_up && rgbs && depths &&
                  depths_cov && cov_median && workspace, "null pointer");
    VINGS_REQUIRE(workspace_bytes >= vings_package_workspace_bytes(n_frames), "workspace too small");
    VINGS_REQUIRE(((uintptr_t)workspace & 15) == 0, "workspace must be 16-byte aligned");
    hipStream_t st = (hipStream_t)stream;
    const int64_t HW = (int64_t)H * W;
    uint32_t* max_key = reinterpret_cast<uint32_t*>(workspace);
    uint32_t* hist = reinterpret_cast<uint32_t*>(static_cast<char*>(workspace) + 256);
    SelectState* state = reinterpret_cast<SelectState*>(hist + (size_t)n_frames * 2 * 2048);
    const size_t hist_bytes = (size_t)n_frames * 2 * 2048 * sizeof(uint32_t);
    ProfScope prof(PROF_AUX, st);
    select_init_kernel<<<cdiv(n_frames, 64), 64, 0, st>>>(n_frames, HW, state, max_key);
    VINGS_LAUNCH_CHECK();
    const int blocks = (int)std::min<int64_t>(cdiv(HW, 256 * 8), 1024);
    const int shifts[3] = {21, 10, 0}, nbits[3] = {11, 11, 10};
    const uint32_t fixed[3] = {0u, 0xFFE00000u, 0xFFFFFC00u};
    for (int pass = 0; pass < 3; ++pass) {
        VINGS_CHECK_HIP(hipMemsetAsync(hist, 0, hist_bytes, st));
        select_hist_kernel<<<dim3(blocks, n_frames), 256, 0, st>>>(
            HW, shifts[pass], nbits[pass], fixed[pass], frame_ids, depths_cov_up, state, hist);
        VINGS_LAUNCH_CHECK();
        select_pick_kernel<<<n_frames, 256, 0, st>>>(shifts[pass], nbits[pass], state, hist,
                                                     cov_median, pass == 2);
        VINGS_LAUNCH_CHECK();
    }
    package_mask_kernel<<<dim3(blocks, n_frames), 256, 0, st>>>(
        HW, frame_ids, images, disps_up, depths_cov_up, cov_median, max_depth, cov_times, rgbs,
        depths, max_key);
    VINGS_LAUNCH_CHECK();
    package_cov_kernel<<<dim3(blocks, n_frames), 256, 0, st>>>(HW, frame_ids, depths_cov_up, depths,
                                                               max_key, depths_cov);
    VINGS_LAUNCH_CHECK();
    return VINGS_OK;
}
