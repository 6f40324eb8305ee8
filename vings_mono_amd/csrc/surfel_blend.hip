// Front-to-back alpha blending of sorted surfels and its backward pass, gfx950.
//
// One 256-lane workgroup (4 wave64) per 16x16 pixel tile, one lane per pixel;
// each wave covers an 8x8 pixel block (the compact shape meets the fewest
// splats).  The tile's sorted splat list is streamed through LDS in batches of
// 256 packed 96-byte records (one gather per lane).  While staging, every lane
// decides for its splat which of the four blocks it can reach with
// alpha >= 1/255 (pixel box, then the exact ellipse / low-pass disc test of
// block_mask) and the waves publish one 64-bit ballot per (loader wave, block).
// A render wave then walks only the set bits of its own four ballots with
// scalar bit scans, so splats that cannot reach the block cost nothing, and
// reads each surviving record as an LDS broadcast (every lane the same
// address: no bank conflicts).  Both kernels are bound by vector-instruction
// issue (on gfx950, with two or more waves per SIMD: a plain VALU instruction
// 2.6 cycles, one with an SGPR operand or a DPP modifier 4, v_rcp / v_exp 8,
// v_permlane32_swap 10-12), so the design rule is "cheapest instruction stream
// per (wave, splat) pair"; this file is built without SLP vectorisation, whose
// v_pk_*_f32 pairs and register shuffles cost more than the scalar code.
//
// Backward walks the same bits back to front.  Per splat the 20 partial
// gradients of the contributing pixels are summed over the wave with a
// transposing tree (cross-half exchange through the LDS crossbar, row pairs
// with v_permlane16_swap, DPP inside a row) and added to the splat's gradient
// row with one coalesced 80-byte global float atomic per (wave, splat) pair.
//
// Serves GaussianRasterizer.forward / backward as called at
// /root/reference/scripts/gaussian/gaussian_base.py:193-203,382 and consumed
// at gaussian_base.py:205-219,387.
#include "common.hpp"
#include "blend_cull.hpp"

namespace vings {

constexpr float kNearN = 0.2f;
constexpr float kFarN = 100.0f;
constexpr float kFilterInvSq = 2.0f;
constexpr float kAlphaMax = 0.99f;
constexpr float kAlphaMin = 1.0f / 255.0f;
constexpr float kTMin = 0.0001f;
constexpr int kBatch = 256;
// backward batch: 21.5 KB of LDS for the staged records; the kernel is held to 7 waves per SIMD
// (72 VGPRs: the staging code spills 30 registers, the blend loop none; 0.5 % faster than 6
// waves at 80 VGPRs, measured by alternating the two builds on one box)
constexpr int kBatchBwd = 256;
constexpr int kRec4 = VINGS_GEOM_STRIDE / 4;  // float4 per record
constexpr int kGrad = VINGS_GRAD_STRIDE;      // gradient components per splat

struct Splat {  // unpacked view of one record
    float Tu[3], Tv[3], Tw[3];
    float cx, cy;
    float n[3];
    float opa;
    float rgb[3];
    float fl[2];
};

struct Rec5 { float4 a, b, c, d, e; };  // the five float4 of a record the blend needs

__device__ __forceinline__ Rec5 load_rec(const float4* __restrict__ r) {
    Rec5 q;
    q.a = r[0]; q.b = r[1]; q.c = r[2]; q.d = r[3]; q.e = r[4];
    return q;
}

__device__ __forceinline__ Splat unpack(const Rec5& q) {
    Splat s;
    s.Tu[0] = q.a.x; s.Tu[1] = q.a.y; s.Tu[2] = q.a.z; s.cx = q.a.w;
    s.Tv[0] = q.b.x; s.Tv[1] = q.b.y; s.Tv[2] = q.b.z; s.cy = q.b.w;
    s.Tw[0] = q.c.x; s.Tw[1] = q.c.y; s.Tw[2] = q.c.z;
    s.n[0] = q.c.w; s.n[1] = q.d.x; s.n[2] = q.d.y;
    s.opa = q.d.z;
    s.rgb[0] = q.d.w; s.rgb[1] = q.e.x; s.rgb[2] = q.e.y;
    s.fl[0] = q.e.z; s.fl[1] = q.e.w;
    return s;
}

struct Hit {
    float kx, ky, kz, lx, ly, lz, pz, rz, sx, sy, rho3d, rho2d, dx, dy, depth, G, alpha;
    bool on_splat;   // the ray/splat intersection (not the low-pass disc) drives G and the depth
};

// Ray / splat intersection and opacity at pixel (fx, fy); false = no contribution.
// kForGrad (the backward pass): sx, sy and rz come back as zeros wherever the low-pass disc
// drives the pixel instead of the intersection -- there the depth is the centre's and no
// gradient flows through the intersection -- and the depth is formed from those selected
// values.  That also makes them finite in EVERY lane (an intersection at infinity never beats
// the disc), which is what lets the caller run idle lanes through the gradient algebra with
// alpha = 0 and nothing else neutralised.
template <bool kForGrad = false>
__device__ __forceinline__ bool eval_hit(const Splat& s, float fx, float fy, Hit& h) {
    h.kx = fx * s.Tw[0] - s.Tu[0]; h.ky = fx * s.Tw[1] - s.Tu[1]; h.kz = fx * s.Tw[2] - s.Tu[2];
    h.lx = fy * s.Tw[0] - s.Tv[0]; h.ly = fy * s.Tw[1] - s.Tv[1]; h.lz = fy * s.Tw[2] - s.Tv[2];
    const float px = h.ky * h.lz - h.kz * h.ly;
    const float py = h.kz * h.lx - h.kx * h.lz;
    h.pz = h.kx * h.ly - h.ky * h.lx;
    // Straight-line on purpose: every lane evaluates everything and the rejections are
    // folded into one predicate.  Early returns here turn every accumulator of the caller
    // into a chain of phi copies (dozens of v_mov per splat); a lane that fails a test
    // may carry inf/NaN in h, which the callers only read under the returned predicate.
    h.rz = rcp_fast(h.pz);
    h.sx = px * h.rz; h.sy = py * h.rz;
    h.rho3d = h.sx * h.sx + h.sy * h.sy;
    h.dx = s.cx - fx; h.dy = s.cy - fy;
    h.rho2d = kFilterInvSq * (h.dx * h.dx + h.dy * h.dy);
    const float rho = fminf(h.rho3d, h.rho2d);
    h.on_splat = h.rho3d <= h.rho2d;
    if (kForGrad) {
        h.sx = h.on_splat ? h.sx : 0.f; h.sy = h.on_splat ? h.sy : 0.f; h.rz = h.on_splat ? h.rz : 0.f;
        h.depth = fmaf(h.sy, s.Tw[1], fmaf(h.sx, s.Tw[0], s.Tw[2]));
    } else {
        h.depth = h.on_splat ? (h.sx * s.Tw[0] + h.sy * s.Tw[1]) + s.Tw[2] : s.Tw[2];
    }
    // G = exp(-rho / 2) as one multiply and v_exp_f32: __expf(-0.5f * rho) is exp2 of
    // (-0.5f * rho) * log2(e), and halving is exact, so rho * (-log2(e) / 2) has the same bits
    h.G = __builtin_amdgcn_exp2f(rho * __uint_as_float(0xbf38aa3bu));
    h.alpha = fminf(kAlphaMax, s.opa * h.G);
    // (the reference's test rho < 0 can never fire: rho is a minimum of sums of squares, and a
    // NaN fails alpha >= 1/255 below)
    return (h.pz != 0.f) & !(h.depth < kNearN) & (h.alpha >= kAlphaMin);
}

// Gather this lane's record of the batch into LDS and publish, per 8x8 block of the
// tile, the ballot of lanes whose splat can reach that block.
__device__ __forceinline__ void stage_batch(float4* __restrict__ lds,
                                            unsigned long long* __restrict__ hit,
                                            const float* __restrict__ geom,
                                            const uint32_t* __restrict__ vals, uint32_t idx,
                                            uint32_t end, uint32_t* __restrict__ ids, float tx0,
                                            float ty0, const uint8_t* __restrict__ stable = nullptr) {
    uint32_t mask = 0;
    if (idx < end) {
        const uint32_t id = vals[idx];
        // (stable: the splat's "no optimiser step" flag rides in the top bit of its staged id)
        if (ids) ids[threadIdx.x] = (stable && stable[id]) ? (id | 0x80000000u) : id;
        const float4* src = reinterpret_cast<const float4*>(geom + (size_t)id * VINGS_GEOM_STRIDE);
        float4 r[kRec4];
#pragma unroll
        for (int k = 0; k < kRec4; ++k) r[k] = src[k];
#pragma unroll
        for (int k = 0; k < kRec4 - 1; ++k) lds[threadIdx.x * (kRec4 - 1) + k] = r[k];
        const float4 bx = r[kRec4 - 1];  // xmin, xmax, ymin, ymax
        if (!(bx.x > tx0 + 15.f || bx.y < tx0 || bx.z > ty0 + 15.f || bx.w < ty0))
            mask = block_mask(r, bx, tx0, ty0);
    }
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        const unsigned long long b = __ballot((mask >> s) & 1u);
        if (lane == 0) hit[wv * 4 + s] = b;
    }
}

__device__ __forceinline__ unsigned long long uniform_u64(unsigned long long v) {
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v);
    const uint32_t hi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
    return ((unsigned long long)hi << 32) | lo;
}

// Seven workgroups per CU (72 VGPRs, six of them spilled in the staging code, 20.4 KB of LDS):
// measured 0.494 ms against 0.504 ms at six; five or four are 5 % and 17 % slower (latency
// hiding matters more here than the partly filled last round of workgroups).
// kLean: the view step of the mapper.  Its loss has no term in the median depth nor in the
// blended flow (dL/dallmap[5], [7], [8] are identically zero, and no flow is attached to the
// splats), so those three planes are written as zeros and the contributor index of the median
// as 0, without the per-pair work behind them; every other output keeps its bits.
template <bool kLean>
__global__ __launch_bounds__(256, 7) void blend_fwd_kernel(
    int H, int W, int gx, const uint32_t* __restrict__ ranges, const uint32_t* __restrict__ vals,
    const float* __restrict__ geom, const float* __restrict__ bg,
    const uint8_t* __restrict__ pixel_mask, float* __restrict__ out_color,
    float* __restrict__ allmap, float* __restrict__ final_T, uint32_t* __restrict__ n_contrib,
    const uint32_t* __restrict__ tile_order) {
    __shared__ float4 rec[kBatch * (kRec4 - 1)];
    __shared__ unsigned long long hit[16];
    // (tile_order: a permutation of the tiles, longest list first; without it the tiles are
    // handed out XCD-contiguously)
    const int tile = tile_order ? (int)tile_order[blockIdx.x] : xcd_contiguous(blockIdx.x, gridDim.x);
    if ((unsigned)tile >= gridDim.x) return;   // a caller's order array with a stray entry
    const int tx = tile % gx, ty = tile / gx;
    // wave w renders block (w & 1, w >> 1) of the tile; lane l is pixel (l & 7, l >> 3) of it
    const int px = tx * 16 + kBlk * ((threadIdx.x >> 6) & 1) + (threadIdx.x & 7);
    const int py = ty * 16 + kBlk * (threadIdx.x >> 7) + ((threadIdx.x >> 3) & 7);
    const bool inside = px < W && py < H;
    const size_t HW = (size_t)H * W;
    const size_t pix = (size_t)py * W + px;
    const bool masked = inside && pixel_mask && pixel_mask[pix] == 0;
    bool done = !inside || masked;
    const float fx = (float)px, fy = (float)py;
    const uint32_t r0 = ranges[2 * tile], r1 = ranges[2 * tile + 1];
    const int wv = threadIdx.x >> 6;

    float C0 = 0, C1 = 0, C2 = 0, N0 = 0, N1 = 0, N2 = 0, F0 = 0, F1 = 0;
    float T = 1.f, Dd = 0, M1d = 0, M2 = 0, dist = 0, med = 0;   // M1d = 2 M1 (exactly: doubling does not round)
    uint32_t last = 0, medc = 0;
    const float c1 = kFarN / (kFarN - kNearN);

    for (uint32_t base = r0; base < r1; base += kBatch) {
        if (__syncthreads_and(done)) break;
        stage_batch(rec, hit, geom, vals, base + threadIdx.x, r1, nullptr, (float)(tx * 16),
                    (float)(ty * 16));
        __syncthreads();
        // the four ballot words of this wave's block (one per loader wave), front to back; two
        // plain nested loops keep the sixteen accumulators in place (an iterator object with
        // the word index in its state made the compiler copy all of them once per splat)
        for (int lw = 0; lw < 4; ++lw) {
            unsigned long long m = uniform_u64(hit[lw * 4 + wv]);
            while (m != 0ull) {
                if (__ballot(!done) == 0ull) break;  // every pixel of this wave has saturated
                const int j = lw * 64 + __builtin_ctzll(m);
                m &= m - 1ull;
                const uint32_t contributor = (base - r0) + (uint32_t)j + 1u;  // 1-based position
                const Splat s = unpack(load_rec(rec + j * (kRec4 - 1)));
                Hit h;
                bool c = eval_hit(s, fx, fy, h) & !done;
                const float test_T = T * (1.f - h.alpha);
                const bool sat = c & (test_T < kTMin);
                done |= sat;
                c &= !sat;
                // No branch from here on, not even a wave-uniform skip of splats that touch no
                // pixel: every lane runs the same instructions and idle lanes add exact zeros
                // through w = 0 (their depth is replaced by 1 so that every term stays finite).
                const float w = c ? h.alpha * T : 0.f;
                const float depth = c ? h.depth : 1.f;
                const float A = 1.f - T;
                const float m1 = fmaf(rcp_fast(depth), -c1 * kNearN, c1);   // c1 (1 - near / depth)
                // (m1^2 A + M2 - 2 m1 M1) w
                dist = fmaf(fmaf(m1, fmaf(m1, A, -M1d), M2), w, dist);
                Dd += depth * w;
                const float mw = m1 * w;
                M1d = fmaf(2.f, mw, M1d);
                M2 = fmaf(m1, mw, M2);
                if (!kLean) {
                    const bool md = c & (T > 0.5f);
                    med = md ? depth : med;
                    medc = md ? contributor : medc;
                    F0 += s.fl[0] * w; F1 += s.fl[1] * w;
                }
                N0 += s.n[0] * w; N1 += s.n[1] * w; N2 += s.n[2] * w;
                C0 += s.rgb[0] * w; C1 += s.rgb[1] * w; C2 += s.rgb[2] * w;
                T = c ? test_T : T;
                last = c ? contributor : last;
            }
        }
    }
    if (!inside) return;
    final_T[pix] = T; final_T[HW + pix] = 0.5f * M1d; final_T[2 * HW + pix] = M2;
    n_contrib[pix] = last; n_contrib[HW + pix] = medc;
    if (masked) {
        out_color[pix] = bg[0]; out_color[HW + pix] = bg[1]; out_color[2 * HW + pix] = bg[2];
#pragma unroll
        for (int c = 0; c < VINGS_AUX_CHANNELS; ++c) allmap[c * HW + pix] = 0.f;
        return;
    }
    out_color[pix] = C0 + T * bg[0];
    out_color[HW + pix] = C1 + T * bg[1];
    out_color[2 * HW + pix] = C2 + T * bg[2];
    allmap[0 * HW + pix] = Dd;
    allmap[1 * HW + pix] = 1.f - T;
    allmap[2 * HW + pix] = N0;
    allmap[3 * HW + pix] = N1;
    allmap[4 * HW + pix] = N2;
    allmap[5 * HW + pix] = med;
    allmap[6 * HW + pix] = dist;
    allmap[7 * HW + pix] = F0;
    allmap[8 * HW + pix] = F1;
}

// Sum over each 8-lane half of a DPP row; every lane of the half receives it.
__device__ __forceinline__ float half_row_sum(float v) {
    v += dpp_f<0xB1>(v);   // quad_perm [1,0,3,2]
    v += dpp_f<0x4E>(v);   // quad_perm [2,3,0,1]
    v += dpp_f<0x141>(v);  // row_half_mirror
    return v;
}

// Forward-only importance pass: importance[i] += sum over pixels of the blend weight
// alpha * T of splat i in this view -- the value the backward pass leaves in
// scores.grad[:,0], which does not depend on the upstream gradient.  The reference obtains
// it by rendering and back-propagating a throw-away loss
// (/root/reference/scripts/gaussian/gaussian_model.py:233-241,
//  /root/reference/scripts/loop/loop_model.py:185-186); here no image is written and no
// backward pass runs.  Same walk as blend_fwd_kernel; one wave sum per (wave, splat).
__global__ __launch_bounds__(256, 6) void blend_importance_kernel(
    int H, int W, int gx, const uint32_t* __restrict__ ranges, const uint32_t* __restrict__ vals,
    const float* __restrict__ geom, const uint8_t* __restrict__ pixel_mask,
    float* __restrict__ importance, const float* __restrict__ bg,
    uint32_t* __restrict__ rgb_max_bits, const uint32_t* __restrict__ tile_order) {
    __shared__ float4 rec[kBatch * (kRec4 - 1)];
    __shared__ unsigned long long hit[16];
    __shared__ float acc[kBatch];
    __shared__ uint32_t ids[kBatch];
    __shared__ uint32_t wmaxs[4];
    const int tile = tile_order ? (int)tile_order[blockIdx.x] : xcd_contiguous(blockIdx.x, gridDim.x);
    if ((unsigned)tile >= gridDim.x) return;   // a caller's order array with a stray entry
    const int tx = tile % gx, ty = tile / gx;
    const int px = tx * 16 + kBlk * ((threadIdx.x >> 6) & 1) + (threadIdx.x & 7);
    const int py = ty * 16 + kBlk * (threadIdx.x >> 7) + ((threadIdx.x >> 3) & 7);
    const bool inside = px < W && py < H;
    const bool masked = inside && pixel_mask && pixel_mask[(size_t)py * W + px] == 0;
    bool done = !inside || masked;
    const float fx = (float)px, fy = (float)py;
    const uint32_t r0 = ranges[2 * tile], r1 = ranges[2 * tile + 1];
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    float T = 1.f, C0 = 0.f, C1 = 0.f, C2 = 0.f;
    for (uint32_t base = r0; base < r1; base += kBatch) {
        if (__syncthreads_and(done)) break;   // also orders the previous flush before restaging
        stage_batch(rec, hit, geom, vals, base + threadIdx.x, r1, ids, (float)(tx * 16),
                    (float)(ty * 16));
        acc[threadIdx.x] = 0.f;
        __syncthreads();
        for (int lw = 0; lw < 4; ++lw) {
            unsigned long long m = uniform_u64(hit[lw * 4 + wv]);
            while (m != 0ull) {
                if (__ballot(!done) == 0ull) break;
                const int j = lw * 64 + __builtin_ctzll(m);
                m &= m - 1ull;
                const Splat s = unpack(load_rec(rec + j * (kRec4 - 1)));
                Hit h;
                bool c = eval_hit(s, fx, fy, h) & !done;
                const float test_T = T * (1.f - h.alpha);
                const bool sat = c & (test_T < kTMin);
                done |= sat;
                c &= !sat;
                if (__ballot(c) == 0ull) continue;
                float w = c ? h.alpha * T : 0.f;
                T = c ? test_T : T;
                if (rgb_max_bits) { C0 += s.rgb[0] * w; C1 += s.rgb[1] * w; C2 += s.rgb[2] * w; }
                w = half_row_sum(w);
                w += dpp_f<0x140>(w);          // row_mirror: sum of the 16-lane row
                w = swap16_add(w, w);
                w = swap32_add(w, w);
                if (lane == 0) atomicAdd(&acc[j], w);
            }
        }
        __syncthreads();
        const uint32_t n = min((uint32_t)kBatch, r1 - base);
        if (threadIdx.x < n) {
            const float v = acc[threadIdx.x];
            if (v != 0.f) atomicAdd(&importance[ids[threadIdx.x]], v);
        }
    }
    if (rgb_max_bits) {
        // largest colour value of the image this view would render (the reference skips views
        // whose render stays below 0.1); non-negative floats order like their bit patterns
        float m = 0.f;
        if (inside) {
            const float Tb = masked ? 1.f : T;
            m = fmaxf(fmaxf(C0 + Tb * bg[0], C1 + Tb * bg[1]), fmaxf(C2 + Tb * bg[2], 0.f));
        }
        uint32_t mb = __float_as_uint(m);
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) mb = max(mb, (uint32_t)__shfl_xor(mb, o, 64));
        if (lane == 0) wmaxs[wv] = mb;
        __syncthreads();
        if (threadIdx.x == 0) {
            mb = max(max(wmaxs[0], wmaxs[1]), max(wmaxs[2], wmaxs[3]));
            if (mb) atomicMax(rgb_max_bits, mb);
        }
    }
}

// ---------------------------------------------------------------------------
// backward
// ---------------------------------------------------------------------------

// kLean: dL/dallmap[5], [7], [8] are zero and not read (see blend_fwd_kernel).
// kStable: `stable` [N] flags the splats the optimiser will not step (the mapper's stable mask,
// /root/reference/scripts/gaussian/gaussian_base.py:400-402: their radii are zeroed before
// SparseGaussianAdam.step, so their gradients are never read).  Such a splat still takes part
// in the compositing recurrence of the pixels behind it and still collects its two scores
// (gaussian_model.py:163-168 records them for every splat), but the eighteen gradient
// components, their algebra and their reduction are skipped: a wave-uniform branch, since a
// splat is the same for every lane.
template <bool kLean, bool kStable = false>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(7, 7))) void blend_bwd_kernel(
    int H, int W, int gx, const uint32_t* __restrict__ ranges, const uint32_t* __restrict__ vals,
    const float* __restrict__ geom, const float* __restrict__ bg,
    const uint8_t* __restrict__ pixel_mask, const float* __restrict__ final_T,
    const uint32_t* __restrict__ n_contrib, const float* __restrict__ dL_dcolor,
    const float* __restrict__ dL_dallmap, float* __restrict__ dgeom,
    const uint32_t* __restrict__ tile_order, const uint8_t* __restrict__ stable,
    const uint32_t* __restrict__ gate = nullptr, uint32_t gate_want = 0u) {
    __shared__ float4 rec[kBatchBwd * (kRec4 - 1)];
    __shared__ uint32_t ids[kBatchBwd];
    __shared__ unsigned long long hit[16];
    __shared__ uint32_t wmax[4];
    // (gate: a device word decides whether this launch or its sibling of the other kLean does
    // the work, see vings_surfel_blend_backward_auto)
    if (gate && ((*gate != 0u) != (gate_want != 0u))) return;
    const int tile = tile_order ? (int)tile_order[blockIdx.x] : xcd_contiguous(blockIdx.x, gridDim.x);
    if ((unsigned)tile >= gridDim.x) return;   // a caller's order array with a stray entry
    const int tx = tile % gx, ty = tile / gx;
    // wave w renders block (w & 1, w >> 1) of the tile; lane l is pixel (l & 7, l >> 3) of it
    const int px = tx * 16 + kBlk * ((threadIdx.x >> 6) & 1) + (threadIdx.x & 7);
    const int py = ty * 16 + kBlk * (threadIdx.x >> 7) + ((threadIdx.x >> 3) & 7);
    const bool inside = px < W && py < H;
    const size_t HW = (size_t)H * W;
    const size_t pix = (size_t)py * W + px;
    const bool active = inside && !(pixel_mask && pixel_mask[pix] == 0);
    // Lane t < 20 adds component t of every (wave, splat) pair straight to the splat's row in
    // global memory: one 80-byte coalesced float atomic per pair.  (An LDS accumulator shared by
    // the four waves and flushed once per batch halves the global atomics, but its LDS float
    // atomics, five instructions per pair, cost more than that saves: 1.30 -> 1.275 ms without
    // it, and its 12.8 KB of LDS now hold a batch of 256 records instead of 160.)  Slot
    // 4k + c of the reduction sits, after the row sums, in register k of the row whose first
    // slot is c; one lane of that row presents it, and the lane whose component lives in that
    // slot fetches it from there.
    const int lane_ = threadIdx.x & 63;
    // slot of the reduction that holds component t of the gradient row (the pairs of the first
    // step are chosen for what they cost, see below)
    constexpr unsigned char kSlotOf[kGrad] = {0, 2, 4, 1, 3, 5, 6, 7, 8, 10, 11, 14, 12, 13, 9, 15,
                                              16, 17, 18, 19};
    const int slot_ = kSlotOf[lane_ < kGrad ? lane_ : 0];
    const int fetch_row = (((slot_ & 3) & 1) << 1) | ((slot_ & 3) >> 1);
    // lane of a row that ends up with register k of the row sums (see the reduction below)
    const int pos_of_k = (slot_ >> 2) == 0 ? 0 : (slot_ >> 2) == 1 ? 8 : (slot_ >> 2) == 2 ? 4
                         : (slot_ >> 2) == 3 ? 12 : 2;
    const int fetch_from = ((fetch_row << 4) | pos_of_k) << 2;         // byte address for ds_bpermute
    const float flush_sign = lane_ < 6 ? -1.f : 1.f;
    const bool low2 = (lane_ & 2) == 0;
    // The upper half of the wave works with the two pixel axes exchanged: it reads {Tv, y} where
    // the lower half reads {Tu, x} (twin slots of the record) and uses (py, px) for (px, py).
    // The ray/splat intersection changes sign as a whole (p = k x l becomes l x k), which
    // leaves sx, sy, rho, depth and alpha as they are; of the gradients, dL/dk and dL/dl change
    // places, so do the two components of dL/dxy, and dL/dTw is symmetric in the exchange.
    // The first step of the reduction sums exactly such pairs over lanes l / l+32 and hands one
    // of each pair to either half: with the exchange every lane already holds the component its
    // half keeps in the same register, and eight selects per (wave, splat) pair go away.
    const bool hi_half = (threadIdx.x & 32) != 0;
    const int twinA = hi_half ? 1 : 0, twinB = hi_half ? 0 : 1;
    const float fx = hi_half ? (float)py : (float)px, fy = hi_half ? (float)px : (float)py;
    const uint32_t r0 = ranges[2 * tile];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;

    const uint32_t last = active ? n_contrib[pix] : 0u;
    const uint32_t medc = (!kLean && active) ? n_contrib[HW + pix] : 0u;
    // the deepest contributor over the tile bounds the walk
    uint32_t mx = last;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) mx = max(mx, (uint32_t)__shfl_xor(mx, o, 64));
    if (lane == 0) wmax[wv] = mx;
    const uint32_t wave_last = __builtin_amdgcn_readfirstlane(mx);
    __syncthreads();
    const uint32_t maxlast = max(max(wmax[0], wmax[1]), max(wmax[2], wmax[3]));
    if (maxlast == 0) return;

    float T_final = 1.f, M1f = 0.f, M2f = 0.f;
    float gC0 = 0, gC1 = 0, gC2 = 0, gD = 0, gA = 0, gN0 = 0, gN1 = 0, gN2 = 0, gMed = 0, gDist = 0,
          gF0 = 0, gF1 = 0;
    if (last > 0) {
        T_final = final_T[pix]; M1f = final_T[HW + pix]; M2f = final_T[2 * HW + pix];
        gC0 = dL_dcolor[pix]; gC1 = dL_dcolor[HW + pix]; gC2 = dL_dcolor[2 * HW + pix];
        gD = dL_dallmap[pix];
        gA = dL_dallmap[HW + pix];
        gN0 = dL_dallmap[2 * HW + pix]; gN1 = dL_dallmap[3 * HW + pix]; gN2 = dL_dallmap[4 * HW + pix];
        gDist = dL_dallmap[6 * HW + pix];
        if (!kLean) {
            gMed = dL_dallmap[5 * HW + pix];
            gF0 = dL_dallmap[7 * HW + pix]; gF1 = dL_dallmap[8 * HW + pix];
        }
    }
    // The distortion term enters through the final moments of the pixel only:
    //   (M2 + m^2 A - 2 m M1) gDist + gA = P0 + m (m P1 + P2)       (share of S)
    //   2 (m A - M1) gDist               = m (2 P1) + P2            (share of dL/ddepth)
    const float P1 = (1.f - T_final) * gDist, P2 = -2.f * M1f * gDist, P0 = gA + M2f * gDist;
    const float P1x2 = P1 + P1;
    const float bg_dot = bg[0] * gC0 + bg[1] * gC1 + bg[2] * gC2;
    const float Tbg = T_final * bg_dot;          // background share of dL/dalpha, times (1 - alpha)
    const float gabs = fabsf(gC0) + fabsf(gC1) + fabsf(gC2);
    // Components 11/15, 12/13, 16/17 and 18/19 are the blend weight times a per-pixel constant,
    // and the first reduction step hands one of each pair to the other half of the wave: which
    // constant a lane keeps and which it sends is decided here, once, instead of by two selects
    // per pair and splat.
    const bool lo_half = (threadIdx.x & 32) == 0;
    const float keep14 = lo_half ? gN0 : gC0, send14 = lo_half ? gC0 : gN0;
    const float keep12 = lo_half ? gN1 : gN2, send12 = lo_half ? gN2 : gN1;
    const float keep16 = lo_half ? gC1 : gC2, send16 = lo_half ? gC2 : gC1;
    const float keep18 = lo_half ? 1.f : gabs, send18 = lo_half ? gabs : 1.f;
    const float c1 = kFarN / (kFarN - kNearN);
    const float c2 = (kFarN * kNearN) / (kFarN - kNearN);

    // Every blended channel c obeys the same back-to-front recurrence
    //   acc_c <- last_alpha * last_c + (1 - last_alpha) * acc_c
    // and enters dL/dalpha only through sum_c (value_c - acc_c) * dL/dchannel_c.  The
    // recurrence is linear with channel-independent coefficients, so the weighted sum
    // B = sum_c acc_c * dL/dchannel_c follows it too, fed by the previous contributor's
    // S = sum_c value_c * dL/dchannel_c: one scalar pair replaces 14 per-channel pairs.
    float T = T_final, last_alpha = 0.f, B = 0.f, S_last = 0.f;

    const int nbatch = (int)((maxlast + kBatchBwd - 1) / kBatchBwd);
    for (int b = nbatch - 1; b >= 0; --b) {
        const uint32_t bstart = (uint32_t)b * kBatchBwd;
        const int cnt = (int)min((uint32_t)kBatchBwd, maxlast - bstart);
        __syncthreads();  // previous batch fully flushed
        stage_batch(rec, hit, geom, vals, r0 + bstart + threadIdx.x, r0 + bstart + cnt, ids,
                    (float)(tx * 16), (float)(ty * 16), kStable ? stable : nullptr);
        __syncthreads();
        // the four ballot words of this wave's block, back to front (plain nested loops: see
        // blend_fwd_kernel)
#pragma unroll
        for (int lw = 3; lw >= 0; --lw) {
          unsigned long long wm = uniform_u64(hit[lw * 4 + wv]);
          // records behind the deepest contributor of this wave's own block touch nothing here
          const int lim = (int)wave_last - (int)bstart - lw * 64;
          if (lim <= 0) continue;
          if (lim < 64) wm &= (1ull << lim) - 1ull;
          while (wm != 0ull) {
            const int bit = 63 - __builtin_clzll(wm);
            const int j = lw * 64 + bit;
            wm &= ~(1ull << bit);
            const uint32_t pos = bstart + j;
            Rec5 q5;
            {
                const float4* rj = rec + j * (kRec4 - 1);
                q5.a = rj[twinA]; q5.b = rj[twinB]; q5.c = rj[2]; q5.d = rj[3]; q5.e = rj[4];
            }
            const Splat s = unpack(q5);
            Hit h;
            const bool c = eval_hit<true>(s, fx, fy, h) & (pos < last);
            if (__ballot(c) == 0ull) continue;  // wave-uniform: nobody in the strip is touched
            // No divergent region below: every lane runs the same instructions and idle lanes
            // are neutralised through two selected roots (alpha = 0, depth = 1): with them w,
            // dL/dalpha and dL/ddepth are exact zeros, and every other factor (G, sx, sy, rz) is
            // finite in every lane, so all 20 partial gradients of an idle lane are exact
            // zeros.  A branch here would cost more in phi copies and zero fills.
            const bool c3 = h.on_splat;                // ray-splat intersection drives G
            const float alpha = c ? h.alpha : 0.f;
            const float r_one_m = rcp_fast(1.f - alpha);  // exactly 1 for idle lanes
            T = T * r_one_m;
            const float w = alpha * T;
            const float depth = c ? h.depth : 1.f;
            const float G = h.G;                       // in [0, 1] in every lane
            // distortion weight of this splat (the per-pixel moments are final ones)
            const float rd = rcp_fast(depth);
            const float m = fmaf(rd, -c1 * kNearN, c1);       // c1 (1 - near / depth)
            const float dm_dd = c2 * rd * rd;
            // S = sum over channels of value * dL/dchannel (alpha channel value is 1)
            float S = P0 + m * (m * P1 + P2);
            S += s.rgb[0] * gC0 + s.rgb[1] * gC1 + s.rgb[2] * gC2;
            if (!kLean) S += s.fl[0] * gF0 + s.fl[1] * gF1;
            S += s.n[0] * gN0 + s.n[1] * gN1 + s.n[2] * gN2;
            S += depth * gD;
            const float Bn = fmaf(last_alpha, S_last - B, B);
            if (kStable && (__builtin_amdgcn_readfirstlane(ids[j]) >> 31)) {
                // no optimiser step for this splat: the recurrence goes on, the two scores are
                // summed over the wave (lower half: the blend weight, upper half: the weight
                // times |dL/dcolour|), and that is all
                B = c ? Bn : B;
                S_last = c ? S : S_last;
                last_alpha = c ? alpha : last_alpha;
                float v = swap32_add(w, w * gabs);
                v = half_row_sum(v);
                v += dpp_f<0x140>(v);            // row_mirror: sum of the 16-lane row
                v = swap16_add(v, v);
                if ((lane & 31) == 0)
                    atomicAdd(&dgeom[(size_t)(ids[j] & 0x7fffffffu) * kGrad + 18 + (lane >> 5)], v);
                continue;
            }
            const float dL_dalpha = c ? (S - Bn) * T - r_one_m * Tbg : 0.f;
            B = c ? Bn : B;
            S_last = c ? S : S_last;
            last_alpha = c ? alpha : last_alpha;
            float dL_dz = w * fmaf(fmaf(m, P1x2, P2), dm_dd, gD);
            if (!kLean) dL_dz += (c & (pos + 1 == medc)) ? gMed : 0.f;
            float g[kGrad];
            g[9] = G * dL_dalpha;      // dL/dopacity (slot 9)
            // (slots 12..19: formed in the reduction from the constants chosen above)
            const float a = -s.opa * g[9];             // dL/drho * -2, zero for idle lanes
            const float sx = h.sx, sy = h.sy, rz = h.rz;   // zeros off the splat (eval_hit<true>)
            const float dsx = a * sx + dL_dz * s.Tw[0];
            const float dsy = a * sy + dL_dz * s.Tw[1];
            const float dpx = dsx * rz, dpy = dsy * rz;  // rz = 0 switches the whole chain off
            const float dpz = -(dpx * sx + dpy * sy);
            const float dkx = h.ly * dpz - h.lz * dpy;
            const float dky = h.lz * dpx - h.lx * dpz;
            const float dkz = h.lx * dpy - h.ly * dpx;
            const float dlx = dpy * h.kz - dpz * h.ky;
            const float dly = dpz * h.kx - dpx * h.kz;
            const float dlz = dpx * h.ky - dpy * h.kx;
            // dL/dTu = -dk and dL/dTv = -dl: the sign is applied once per splat at the flush
            // (slots 2i / 2i+1 of a pair: in the upper half the two registers hold each other's
            // component, which is the one that half keeps)
            g[0] = dkx; g[1] = dlx; g[2] = dky; g[3] = dly; g[4] = dkz; g[5] = dlz;
            g[6] = fx * dkx + fy * dlx + dL_dz * sx;
            g[7] = fx * dky + fy * dly + dL_dz * sy;
            g[8] = fx * dkz + fy * dlz + dL_dz;
            const float a2 = c3 ? 0.f : a * kFilterInvSq;  // low-pass filter drives G instead
            g[10] = a2 * h.dx;
            g[11] = a2 * h.dy;
            {
                // Transposing reduction.  Step 1 sums a PAIR of components over lanes l / l+32
                // and leaves one component per half: each lane keeps the component its half
                // will own and fetches the partner's share of it.  Step 2 (v_permlane16_swap +
                // add) does the same for odd/even 16-lane rows.  After the two steps 20
                // components have become 5 registers whose four rows hold different components
                // (row r, register k -> slot 4k + {0,2,1,3}[r]), each already summed over
                // the four rows; step 3 sums inside the rows.
                // Step 1 goes through ds_bpermute: the LDS crossbar is idle here, while
                // v_permlane32_swap occupies the vector issue port for 10-12 cycles; the ten
                // exchanges are independent, so their latency overlaps.  (Step 2 through
                // the crossbar as well would put a second LDS round trip on the critical path
                // and measures slower.)
                float v10[kGrad / 2];
                {
                    const bool lo = lane < 32;
                    const int partner = (lane ^ 32) << 2;
                    float mine[kGrad / 2], theirs[kGrad / 2];
#pragma unroll
                    for (int i = 0; i < kGrad / 2; ++i) {
                        float send;
                        if (i == 7) { mine[i] = w * keep14; send = w * send14; }
                        else if (i == 6) { mine[i] = w * keep12; send = w * send12; }
                        else if (i == 8) { mine[i] = w * keep16; send = w * send16; }
                        else if (i == 9) { mine[i] = w * keep18; send = w * send18; }
                        else if (i <= 2 || i == 5) { mine[i] = g[2 * i]; send = g[2 * i + 1]; }
                        else { mine[i] = lo ? g[2 * i] : g[2 * i + 1]; send = lo ? g[2 * i + 1] : g[2 * i]; }
                        theirs[i] = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(
                            partner, __builtin_bit_cast(int, send)));
                    }
#pragma unroll
                    for (int i = 0; i < kGrad / 2; ++i) v10[i] = mine[i] + theirs[i];
                }
                float u5[kGrad / 4];
#pragma unroll
                for (int k = 0; k < kGrad / 4; ++k) u5[k] = swap16_add(v10[2 * k], v10[2 * k + 1]);
                // Step 3, inside a 16-lane row, keeps transposing: every level halves the
                // distance AND the number of registers (a pair of registers meets in one whose
                // lane groups own different components), 15 instructions where summing the five
                // registers separately takes 20.  Levels: row_mirror (8), row_half_mirror (4),
                // quad_perm 2301 (2), quad_perm 1032 (1).  The row sums of registers 0..4 end in
                // lanes 0, 8, 4, 12 and 2 of the row.
                float r01, r23, s03;
                // (the lanes of banks 2, 3 = the high half of a row take the second register:
                // a bank mask on the second add instead of two selects)
                asm volatile("s_nop 1\n\t"
                             "v_add_f32_dpp %0, %3, %3 row_mirror row_mask:0xf bank_mask:0xf\n\t"
                             "v_add_f32_dpp %1, %5, %5 row_mirror row_mask:0xf bank_mask:0xf\n\t"
                             "v_add_f32_dpp %0, %4, %4 row_mirror row_mask:0xf bank_mask:0xc\n\t"
                             "v_add_f32_dpp %1, %6, %6 row_mirror row_mask:0xf bank_mask:0xc\n\t"
                             "s_nop 1\n\t"
                             "v_add_f32_dpp %2, %0, %0 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
                             "s_nop 0\n\t"
                             "v_add_f32_dpp %2, %1, %1 row_half_mirror row_mask:0xf bank_mask:0xa"
                             : "=&v"(r01), "=&v"(r23), "=&v"(s03)
                             : "v"(u5[0]), "v"(u5[1]), "v"(u5[2]), "v"(u5[3]));
                const float r4 = u5[4] + dpp_f<0x140>(u5[4]);
                const float s4 = r4 + dpp_f<0x141>(r4);
                const float z = (low2 ? s03 : s4) + dpp_f<0x4E>(low2 ? s4 : s03);
                const float v = z + dpp_f<0xB1>(z);
                const float mine = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(
                    fetch_from, __builtin_bit_cast(int, v)));
                // (dL/dTu = -dk and dL/dTv = -dl: the sign of components 0..5)
                if (lane < kGrad)
                    atomicAdd(&dgeom[(size_t)(kStable ? ids[j] & 0x7fffffffu : ids[j]) * kGrad + lane],
                              mine * flush_sign);
            }
          }
        }
    }
}

// flag |= 1 if any of the n floats is not +0 / -0 (NaN counts as non-zero)
__global__ __launch_bounds__(256) void plane_nonzero_kernel(const float* __restrict__ p, size_t n,
                                                            uint32_t* __restrict__ flag) {
    bool any = false;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256)
        any |= !(p[i] == 0.f);
    if (__ballot(any) != 0ull && (threadIdx.x & 63) == 0) atomicOr(flag, 1u);
}

}  // namespace vings

using namespace vings;

static int blend_forward_impl(bool lean, int H, int W, const uint32_t* ranges,
                              const uint32_t* vals_sorted, const float* geom, const float* bg,
                              const uint8_t* pixel_mask, float* out_color, float* allmap,
                              float* final_T, uint32_t* n_contrib, void* stream,
                              const uint32_t* tile_order = nullptr) {
    VINGS_REQUIRE(H > 0 && W > 0, "sizes");
    VINGS_REQUIRE(ranges && bg && out_color && allmap && final_T && n_contrib, "null pointer");
    VINGS_REQUIRE(((uintptr_t)geom & 15) == 0, "geom must be 16-byte aligned");
    const int gx = (W + 15) / 16, gy = (H + 15) / 16;
    ProfScope prof(PROF_BLEND_FWD, (hipStream_t)stream);
    if (lean)
        blend_fwd_kernel<true><<<gx * gy, 256, 0, (hipStream_t)stream>>>(
            H, W, gx, ranges, vals_sorted, geom, bg, pixel_mask, out_color, allmap, final_T, n_contrib,
            tile_order);
    else
        blend_fwd_kernel<false><<<gx * gy, 256, 0, (hipStream_t)stream>>>(
            H, W, gx, ranges, vals_sorted, geom, bg, pixel_mask, out_color, allmap, final_T, n_contrib,
            tile_order);
    VINGS_LAUNCH_CHECK();
    return VINGS_OK;
}

extern "C" int vings_surfel_blend_forward(int H, int W, const uint32_t* ranges,
                                          const uint32_t* vals_sorted, const float* geom,
                                          const float* bg, const uint8_t* pixel_mask,
                                          float* out_color, float* allmap, float* final_T,
                                          uint32_t* n_contrib, void* stream) {
    return blend_forward_impl(false, H, W, ranges, vals_sorted, geom, bg, pixel_mask, out_color,
                              allmap, final_T, n_contrib, stream);
}

extern "C" int vings_surfel_blend_forward_lean(int H, int W, const uint32_t* ranges,
                                               const uint32_t* vals_sorted, const float* geom,
                                               const float* bg, const uint8_t* pixel_mask,
                                               float* out_color, float* allmap, float* final_T,
                                               uint32_t* n_contrib, void* stream) {
    return blend_forward_impl(true, H, W, ranges, vals_sorted, geom, bg, pixel_mask, out_color,
                              allmap, final_T, n_contrib, stream);
}

static int blend_backward_impl(bool clear, bool lean, int N, int H, int W, const uint32_t* ranges,
                                           const uint32_t* vals_sorted, const float* geom,
                                           const float* bg, const uint8_t* pixel_mask,
                                           const float* final_T, const uint32_t* n_contrib,
                                           const float* dL_dcolor, const float* dL_dallmap,
                                           float* dgeom, void* stream,
                                           const uint32_t* tile_order = nullptr,
                                           const uint8_t* stable = nullptr) {
    VINGS_REQUIRE(N >= 0 && H > 0 && W > 0, "sizes");
    if (N == 0) return VINGS_OK;
    VINGS_REQUIRE(ranges && bg && final_T && n_contrib && dL_dcolor && dL_dallmap && dgeom,
                  "null pointer");
    hipStream_t s = (hipStream_t)stream;
    if (clear)
        VINGS_CHECK_HIP(hipMemsetAsync(dgeom, 0, sizeof(float) * (size_t)N * VINGS_GRAD_STRIDE, s));
    const int gx = (W + 15) / 16, gy = (H + 15) / 16;
    ProfScope prof(PROF_BLEND_BWD, s);
    if (stable && lean)
        blend_bwd_kernel<true, true><<<gx * gy, 256, 0, s>>>(H, W, gx, ranges, vals_sorted, geom, bg,
                                                             pixel_mask, final_T, n_contrib, dL_dcolor,
                                                             dL_dallmap, dgeom, tile_order, stable);
    else if (stable)
        blend_bwd_kernel<false, true><<<gx * gy, 256, 0, s>>>(H, W, gx, ranges, vals_sorted, geom, bg,
                                                              pixel_mask, final_T, n_contrib, dL_dcolor,
                                                              dL_dallmap, dgeom, tile_order, stable);
    else if (lean)
        blend_bwd_kernel<true><<<gx * gy, 256, 0, s>>>(H, W, gx, ranges, vals_sorted, geom, bg,
                                                       pixel_mask, final_T, n_contrib, dL_dcolor,
                                                       dL_dallmap, dgeom, tile_order, nullptr);
    else
        blend_bwd_kernel<false><<<gx * gy, 256, 0, s>>>(H, W, gx, ranges, vals_sorted, geom, bg,
                                                        pixel_mask, final_T, n_contrib, dL_dcolor,
                                                        dL_dallmap, dgeom, tile_order, nullptr);
    VINGS_LAUNCH_CHECK();
    return VINGS_OK;
}

static int blend_importance_impl(int N, int H, int W, const uint32_t* ranges,
                                 const uint32_t* vals_sorted, const float* geom,
                                 const uint8_t* pixel_mask, float* importance, int accumulate,
                                 const float* bg, float* rgb_max, const uint32_t* tile_order,
                                 void* stream) {
    VINGS_REQUIRE(N >= 0 && H > 0 && W > 0, "sizes");
    if (N == 0) return VINGS_OK;
    VINGS_REQUIRE(ranges && importance, "null pointer");
    VINGS_REQUIRE(((uintptr_t)geom & 15) == 0, "geom must be 16-byte aligned");
    hipStream_t s = (hipStream_t)stream;
    VINGS_REQUIRE(!rgb_max || bg, "rgb_max needs the background colour");
    if (!accumulate) VINGS_CHECK_HIP(hipMemsetAsync(importance, 0, sizeof(float) * (size_t)N, s));
    if (rgb_max) VINGS_CHECK_HIP(hipMemsetAsync(rgb_max, 0, sizeof(float), s));
    const int gx = (W + 15) / 16, gy = (H + 15) / 16;
    ProfScope prof(PROF_BLEND_FWD, s);
    blend_importance_kernel<<<gx * gy, 256, 0, s>>>(H, W, gx, ranges, vals_sorted, geom,
                                                    pixel_mask, importance, bg,
                                                    reinterpret_cast<uint32_t*>(rgb_max), tile_order);
    VINGS_LAUNCH_CHECK();
    return VINGS_OK;
}

extern "C" int vings_surfel_blend_importance(int N, int H, int W, const uint32_t* ranges,
                                             const uint32_t* vals_sorted, const float* geom,
                                             const uint8_t* pixel_mask, float* importance,
                                             int accumulate, const float* bg, float* rgb_max,
                                             void* stream) {
    return blend_importance_impl(N, H, W, ranges, vals_sorted, geom, pixel_mask, importance,
                                 accumulate, bg, rgb_max, nullptr, stream);
}

extern "C" int vings_surfel_blend_importance_ordered(int N, int H, int W, const uint32_t* ranges,
                                                     const uint32_t* vals_sorted, const float* geom,
                                                     const uint8_t* pixel_mask, float* importance,
                                                     int accumulate, const float* bg, float* rgb_max,
                                                     const uint32_t* tile_order, void* stream) {
    return blend_importance_impl(N, H, W, ranges, vals_sorted, geom, pixel_mask, importance,
                                 accumulate, bg, rgb_max, tile_order, stream);
}

extern "C" int vings_surfel_blend_backward(int N, int H, int W, const uint32_t* ranges,
                                           const uint32_t* vals_sorted, const float* geom,
                                           const float* bg, const uint8_t* pixel_mask,
                                           const float* final_T, const uint32_t* n_contrib,
                                           const float* dL_dcolor, const float* dL_dallmap,
                                           float* dgeom, void* stream) {
    return blend_backward_impl(true, false, N, H, W, ranges, vals_sorted, geom, bg, pixel_mask,
                               final_T, n_contrib, dL_dcolor, dL_dallmap, dgeom, stream);
}

extern "C" int vings_surfel_blend_backward_accumulate(int N, int H, int W, const uint32_t* ranges,
                                           const uint32_t* vals_sorted, const float* geom,
                                           const float* bg, const uint8_t* pixel_mask,
                                           const float* final_T, const uint32_t* n_contrib,
                                           const float* dL_dcolor, const float* dL_dallmap,
                                           float* dgeom, void* stream) {
    return blend_backward_impl(false, false, N, H, W, ranges, vals_sorted, geom, bg, pixel_mask,
                               final_T, n_contrib, dL_dcolor, dL_dallmap, dgeom, stream);
}

extern "C" int vings_surfel_blend_backward_accumulate_lean(
    int N, int H, int W, const uint32_t* ranges, const uint32_t* vals_sorted, const float* geom,
    const float* bg, const uint8_t* pixel_mask, const float* final_T, const uint32_t* n_contrib,
    const float* dL_dcolor, const float* dL_dallmap, float* dgeom, void* stream) {
    return blend_backward_impl(false, true, N, H, W, ranges, vals_sorted, geom, bg, pixel_mask,
                               final_T, n_contrib, dL_dcolor, dL_dallmap, dgeom, stream);
}

extern "C" int vings_surfel_blend_forward_ordered(int lean, int H, int W, const uint32_t* ranges,
                                                  const uint32_t* vals_sorted, const float* geom,
                                                  const float* bg, const uint8_t* pixel_mask,
                                                  float* out_color, float* allmap, float* final_T,
                                                  uint32_t* n_contrib, const uint32_t* tile_order,
                                                  void* stream) {
    return blend_forward_impl(lean != 0, H, W, ranges, vals_sorted, geom, bg, pixel_mask, out_color,
                              allmap, final_T, n_contrib, stream, tile_order);
}

extern "C" int vings_surfel_blend_backward_accumulate_ordered(
    int lean, int N, int H, int W, const uint32_t* ranges, const uint32_t* vals_sorted,
    const float* geom, const float* bg, const uint8_t* pixel_mask, const float* final_T,
    const uint32_t* n_contrib, const float* dL_dcolor, const float* dL_dallmap, float* dgeom,
    const uint32_t* tile_order, void* stream) {
    return blend_backward_impl(false, lean != 0, N, H, W, ranges, vals_sorted, geom, bg, pixel_mask,
                               final_T, n_contrib, dL_dcolor, dL_dallmap, dgeom, stream, tile_order);
}

extern "C" int vings_surfel_blend_backward_accumulate_stable(
    int lean, int N, int H, int W, const uint32_t* ranges, const uint32_t* vals_sorted,
    const float* geom, const float* bg, const uint8_t* pixel_mask, const float* final_T,
    const uint32_t* n_contrib, const float* dL_dcolor, const float* dL_dallmap, float* dgeom,
    const uint32_t* tile_order, const uint8_t* stable_mask, void* stream) {
    return blend_backward_impl(false, lean != 0, N, H, W, ranges, vals_sorted, geom, bg, pixel_mask,
                               final_T, n_contrib, dL_dcolor, dL_dallmap, dgeom, stream, tile_order,
                               stable_mask);
}

extern "C" int vings_surfel_blend_backward_auto(
    int N, int H, int W, const uint32_t* ranges, const uint32_t* vals_sorted, const float* geom,
    const float* bg, const uint8_t* pixel_mask, const float* final_T, const uint32_t* n_contrib,
    const float* dL_dcolor, const float* dL_dallmap, float* dgeom, const uint32_t* tile_order,
    uint32_t* flag_word, void* stream) {
    VINGS_REQUIRE(N >= 0 && H > 0 && W > 0, "sizes");
    if (N == 0) return VINGS_OK;
    VINGS_REQUIRE(ranges && bg && final_T && n_contrib && dL_dcolor && dL_dallmap && dgeom && flag_word,
                  "null pointer");
    hipStream_t s = (hipStream_t)stream;
    const size_t HW = (size_t)H * W;
    VINGS_CHECK_HIP(hipMemsetAsync(flag_word, 0, sizeof(uint32_t), s));
    ProfScope prof(PROF_BLEND_BWD, s);
    plane_nonzero_kernel<<<cdiv((int64_t)HW, 256 * 8), 256, 0, s>>>(dL_dallmap + 5 * HW, HW, flag_word);
    VINGS_LAUNCH_CHECK();
    const int gx = (W + 15) / 16, gy = (H + 15) / 16;
    blend_bwd_kernel<true><<<gx * gy, 256, 0, s>>>(H, W, gx, ranges, vals_sorted, geom, bg, pixel_mask,
                                                   final_T, n_contrib, dL_dcolor, dL_dallmap, dgeom,
                                                   tile_order, nullptr, flag_word, 0u);
    VINGS_LAUNCH_CHECK();
    blend_bwd_kernel<false><<<gx * gy, 256, 0, s>>>(H, W, gx, ranges, vals_sorted, geom, bg, pixel_mask,
                                                    final_T, n_contrib, dL_dcolor, dL_dallmap, dgeom,
                                                    tile_order, nullptr, flag_word, 1u);
    VINGS_LAUNCH_CHECK();
    return VINGS_OK;
}
