// Surfel projection ("preprocess") forward and backward for gfx950.
//
// One lane per splat.  This translation unit is compiled with
// -ffp-contract=off: the tile rectangle, radius and depth key of every splat
// must round exactly like the float32 CPU oracle so that tile/sort indices
// are bit-exact; the arithmetic below is therefore written as plain
// left-to-right float expressions and must not be re-associated.
//
// Serves GaussianRasterizer.forward/backward as called at
// /root/reference/scripts/gaussian/gaussian_base.py:193-203,382.
#include <mutex>

#include <atomic>
#include "common.hpp"
#include "splat_finish.hpp"
#include "blend_cull.hpp"

namespace vings {

constexpr float kNear = 0.2f;
constexpr float kFilterSize = 0.707106f;

struct PixProj {  // B = ndc2pix * (P * w2c), rows x,y,w
    float b[3][4];
};

__device__ __forceinline__ PixProj pixel_projection(const float* __restrict__ proj, int W, int H) {
    PixProj B;
    const float hw = (float)W / 2, hh = (float)H / 2;
    const float ow = (float)(W - 1) / 2, oh = (float)(H - 1) / 2;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const float p0 = proj[c * 4 + 0], p1 = proj[c * 4 + 1], p3 = proj[c * 4 + 3];
        B.b[0][c] = hw * p0 + ow * p3;
        B.b[1][c] = hh * p1 + oh * p3;
        B.b[2][c] = p3;
    }
    return B;
}

__device__ __forceinline__ void quat_to_R(const float4 q, float R[3][3]) {
    const float r = q.x, x = q.y, y = q.z, z = q.w;
    R[0][0] = 1 - 2 * (y * y + z * z);
    R[0][1] = 2 * (x * y - r * z);
    R[0][2] = 2 * (x * z + r * y);
    R[1][0] = 2 * (x * y + r * z);
    R[1][1] = 1 - 2 * (x * x + z * z);
    R[1][2] = 2 * (y * z - r * x);
    R[2][0] = 2 * (x * z - r * y);
    R[2][1] = 2 * (y * z + r * x);
    R[2][2] = 1 - 2 * (x * x + y * y);
}

struct SplatFrame {
    float A[3][3];  // rows Tu, Tv, Tw
    float R[3][3];
    float n[3];     // view-space normal, camera facing
    float pv[3];
    float sx, sy, mult;
};

__device__ __forceinline__ void splat_frame_of(const float* __restrict__ means3D, float2 s,
                                               float4 q, const float* __restrict__ view,
                                               const PixProj& B, float scale_mod, int i,
                                               SplatFrame& f) {
    const float p0 = means3D[3 * i], p1 = means3D[3 * i + 1], p2 = means3D[3 * i + 2];
    f.pv[0] = view[0] * p0 + view[4] * p1 + view[8] * p2 + view[12];
    f.pv[1] = view[1] * p0 + view[5] * p1 + view[9] * p2 + view[13];
    f.pv[2] = view[2] * p0 + view[6] * p1 + view[10] * p2 + view[14];
    quat_to_R(q, f.R);
    f.sx = scale_mod * s.x;
    f.sy = scale_mod * s.y;
    const float L0[3] = {f.R[0][0] * f.sx, f.R[1][0] * f.sx, f.R[2][0] * f.sx};
    const float L1[3] = {f.R[0][1] * f.sy, f.R[1][1] * f.sy, f.R[2][1] * f.sy};
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        f.A[r][0] = B.b[r][0] * L0[0] + B.b[r][1] * L0[1] + B.b[r][2] * L0[2];
        f.A[r][1] = B.b[r][0] * L1[0] + B.b[r][1] * L1[1] + B.b[r][2] * L1[2];
        f.A[r][2] = B.b[r][0] * p0 + B.b[r][1] * p1 + B.b[r][2] * p2 + B.b[r][3];
    }
    const float nw[3] = {f.R[0][2], f.R[1][2], f.R[2][2]};
    f.n[0] = view[0] * nw[0] + view[4] * nw[1] + view[8] * nw[2];
    f.n[1] = view[1] * nw[0] + view[5] * nw[1] + view[9] * nw[2];
    f.n[2] = view[2] * nw[0] + view[6] * nw[1] + view[10] * nw[2];
    const float cosv = -(f.pv[0] * f.n[0] + f.pv[1] * f.n[1] + f.pv[2] * f.n[2]);
    f.mult = cosv == 0.f ? 0.f : (cosv > 0.f ? 1.f : -1.f);
}

__device__ __forceinline__ void splat_frame(const float* __restrict__ means3D,
                                            const float* __restrict__ scales,
                                            const float* __restrict__ rots,
                                            const float* __restrict__ view, const PixProj& B,
                                            float scale_mod, int i, SplatFrame& f) {
    splat_frame_of(means3D, reinterpret_cast<const float2*>(scales)[i],
                   reinterpret_cast<const float4*>(rots)[i], view, B, scale_mod, i, f);
}

// kRaw: scales / rots / opac hold the STORED parameters; the kernel applies the mapper's
// activations (exp, normalise, sigmoid: gaussian_base.py:44-48,64-72, the expressions of
// splat_activate_kernel) itself and leaves the activated values in act_* for the backward pass.
template <bool kRaw>
__global__ __launch_bounds__(256) void project_fwd_kernel(
    int N, int H, int W, float scale_mod, const float* __restrict__ means3D,
    const float* __restrict__ scales, const float* __restrict__ rots,
    const float* __restrict__ opac, float* __restrict__ act_scales, float* __restrict__ act_rots,
    float* __restrict__ act_opac, const float* __restrict__ colors,
    const float* __restrict__ flow, const float* __restrict__ view,
    const float* __restrict__ proj, float* __restrict__ geom, float* __restrict__ depth,
    int32_t* __restrict__ radii, int32_t* __restrict__ rect, uint32_t* __restrict__ tiles_touched,
    uint32_t* __restrict__ block_counts, uint32_t* __restrict__ block_dmin,
    uint32_t* __restrict__ block_dmax, int tight, float* __restrict__ dgeom_clear) {
    __shared__ uint32_t wsum[4], wmin[4], wmax[4];
    const int i = blockIdx.x * 256 + threadIdx.x;
    uint32_t touched = 0;
    uint32_t dlo = 0xFFFFFFFFu, dhi = 0u;   // float bits of the depth of a binned splat
    __shared__ float4 srec[256 * 6];        // records of the splats whose tiles get tested
    __shared__ int4 sgeo[256];
    uint32_t cand = 0;                      // tiles of this lane's splat to test (0: none)
    int x0 = 0, y0 = 0, x1 = 0, y1 = 0;
    int4 packed_rect = make_int4(0, 0, 0, 0);
    bool is_packed = false;
    if (i < N) {
        const int gx = (W + 15) / 16, gy = (H + 15) / 16;
        int32_t rad = 0;
        float dep = 0.f;
        const PixProj B = pixel_projection(proj, W, H);
        SplatFrame f;
        float2 s2 = reinterpret_cast<const float2*>(scales)[i];
        float4 q4 = reinterpret_cast<const float4*>(rots)[i];
        float o = opac[i];
        if (kRaw) {
            o = 1.0f / (1.0f + expf(-o));
            s2 = make_float2(expf(s2.x), expf(s2.y));
            const float inv = 1.0f / fmaxf(sqrtf(q4.x * q4.x + q4.y * q4.y + q4.z * q4.z + q4.w * q4.w), 1e-12f);
            q4 = make_float4(q4.x * inv, q4.y * inv, q4.z * inv, q4.w * inv);
            act_opac[i] = o;
            reinterpret_cast<float2*>(act_scales)[i] = s2;
            reinterpret_cast<float4*>(act_rots)[i] = q4;
        }
        splat_frame_of(means3D, s2, q4, view, B, scale_mod, i, f);
        bool ok = f.pv[2] > kNear && f.mult != 0.f;
        float cx = 0.f, cy = 0.f;
        if (ok) {
            const float* Tu = f.A[0];
            const float* Tv = f.A[1];
            const float* Tw = f.A[2];
            const float t0 = 9.f, t1 = 9.f, t2 = -1.f;
            const float d = t0 * Tw[0] * Tw[0] + t1 * Tw[1] * Tw[1] + t2 * Tw[2] * Tw[2];
            ok = d != 0.f;
            if (ok) {
                const float inv = 1.f / d;
                const float f0 = inv * t0, f1 = inv * t1, f2 = inv * t2;
                cx = f0 * Tu[0] * Tw[0] + f1 * Tu[1] * Tw[1] + f2 * Tu[2] * Tw[2];
                cy = f0 * Tv[0] * Tw[0] + f1 * Tv[1] * Tw[1] + f2 * Tv[2] * Tw[2];
                const float ex =
                    cx * cx - (f0 * Tu[0] * Tu[0] + f1 * Tu[1] * Tu[1] + f2 * Tu[2] * Tu[2]);
                const float ey =
                    cy * cy - (f0 * Tv[0] * Tv[0] + f1 * Tv[1] * Tv[1] + f2 * Tv[2] * Tv[2]);
                const float hx = sqrtf(fmaxf(1e-4f, ex)), hy = sqrtf(fmaxf(1e-4f, ey));
                const float radius = ceilf(fmaxf(fmaxf(hx, hy), 3.f * kFilterSize));
                ok = (radius < 1e7f) && (cx == cx) && (cy == cy) && (cx > -1e9f && cx < 1e9f) &&
                     (cy > -1e9f && cy < 1e9f);
                if (ok) {
                    x0 = min(gx, max(0, (int)((cx - radius) / 16)));
                    y0 = min(gy, max(0, (int)((cy - radius) / 16)));
                    x1 = min(gx, max(0, (int)((cx + radius + 15) / 16)));
                    y1 = min(gy, max(0, (int)((cy + radius + 15) / 16)));
                    touched = (uint32_t)((x1 - x0) * (y1 - y0));
                    ok = touched != 0;
                    if (ok) {
                        rad = (int32_t)radius;
                        dep = f.pv[2];
                    }
                }
            }
        }
        if (!ok) { touched = 0; x0 = y0 = x1 = y1 = 0; }
        else dlo = dhi = __float_as_uint(dep);
        radii[i] = rad;
        if (dgeom_clear && rad > 0) {   // the backward pass accumulates into this row
            float4* z = reinterpret_cast<float4*>(dgeom_clear + (size_t)i * VINGS_GRAD_STRIDE);
#pragma unroll
            for (int k = 0; k < VINGS_GRAD_STRIDE / 4; ++k) z[k] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
        depth[i] = dep;
        if (ok) {
            float4* g = reinterpret_cast<float4*>(geom + (size_t)i * VINGS_GEOM_STRIDE);
            const float c0 = colors[3 * i], c1 = colors[3 * i + 1], c2 = colors[3 * i + 2];
            float fl0 = 0.f, fl1 = 0.f;
            if (flow) { fl0 = flow[2 * i]; fl1 = flow[2 * i + 1]; }
            float4 rr[6];
            // {Tu, cx} {Tv, cy} {Tw, n0}: the two pixel axes sit in twin slots, so a lane of the
            // blend backward can read them exchanged (see blend_bwd_kernel)
            rr[0] = make_float4(f.A[0][0], f.A[0][1], f.A[0][2], cx);
            rr[1] = make_float4(f.A[1][0], f.A[1][1], f.A[1][2], cy);
            rr[2] = make_float4(f.A[2][0], f.A[2][1], f.A[2][2], f.mult * f.n[0]);
            rr[3] = make_float4(f.mult * f.n[1], f.mult * f.n[2], o, c0);
            rr[4] = make_float4(c1, c2, fl0, fl1);
            // Conservative pixel box of alpha >= 1/255: alpha = o * exp(-rho/2) with
            // rho = min(rho3d, rho2d), so a pixel contributes only where rho <= rc.
            // {rho3d <= rc} is the projected ellipse of radius sqrt(rc) (same tangent-line
            // bound as the 3-sigma box, exact while the ellipse stays in front of the
            // camera plane, d < 0); {rho2d <= rc} is a disc of radius sqrt(rc/2) about xy.
            float bx0 = 1e30f, bx1 = -1e30f, by0 = 1e30f, by1 = -1e30f;   // empty
            if (255.f * o >= 0.999f) {
                const float rc = 2.f * logf(255.f * o) * 1.002f + 0.02f;
                const float* Tu = f.A[0];
                const float* Tv = f.A[1];
                const float* Tw = f.A[2];
                const float dd = rc * Tw[0] * Tw[0] + rc * Tw[1] * Tw[1] - Tw[2] * Tw[2];
                if (dd < 0.f) {
                    const float fi = 1.f / dd;
                    const float ccx = fi * (rc * Tu[0] * Tw[0] + rc * Tu[1] * Tw[1] - Tu[2] * Tw[2]);
                    const float ccy = fi * (rc * Tv[0] * Tw[0] + rc * Tv[1] * Tw[1] - Tv[2] * Tw[2]);
                    const float hx2 = ccx * ccx - fi * (rc * Tu[0] * Tu[0] + rc * Tu[1] * Tu[1] - Tu[2] * Tu[2]);
                    const float hy2 = ccy * ccy - fi * (rc * Tv[0] * Tv[0] + rc * Tv[1] * Tv[1] - Tv[2] * Tv[2]);
                    const float hx = sqrtf(fmaxf(0.f, hx2)), hy = sqrtf(fmaxf(0.f, hy2));
                    const float rr = sqrtf(0.5f * rc);
                    const float slack = 0.5f + 1e-4f * (fabsf(ccx) + fabsf(ccy) + hx + hy);
                    bx0 = fminf(ccx - hx, cx - rr) - slack; bx1 = fmaxf(ccx + hx, cx + rr) + slack;
                    by0 = fminf(ccy - hy, cy - rr) - slack; by1 = fmaxf(ccy + hy, cy + rr) + slack;
                    if (!(bx0 == bx0 && bx1 == bx1 && by0 == by0 && by1 == by1)) {
                        bx0 = by0 = -1e30f; bx1 = by1 = 1e30f;
                    }
                } else {
                    bx0 = by0 = -1e30f; bx1 = by1 = 1e30f;   // crosses the camera plane: no bound
                }
            }
            rr[5] = make_float4(bx0, bx1, by0, by1);
#pragma unroll
            for (int k = 0; k < 6; ++k) g[k] = rr[k];
            if (tight) {
                // Drop the tiles the alpha box cannot reach: exactly those every blend pass
                // rejects for this splat with the same comparisons (stage_batch), so no
                // rendered value changes; only the instance list gets shorter.
                const float lo_x = fminf(fmaxf(bx0, -1e8f), 1e8f), hi_x = fminf(fmaxf(bx1, -1e8f), 1e8f);
                const float lo_y = fminf(fmaxf(by0, -1e8f), 1e8f), hi_y = fminf(fmaxf(by1, -1e8f), 1e8f);
                int a0 = max(x0, min(x1, (int)floorf((lo_x - 15.f) * (1.f / 16.f)) - 1));
                int a1 = max(x0, min(x1, (int)floorf(hi_x * (1.f / 16.f)) + 2));
                while (a0 < x1 && bx0 > 16.f * (float)a0 + 15.f) ++a0;
                while (a1 > a0 && bx1 < 16.f * (float)(a1 - 1)) --a1;
                int b0 = max(y0, min(y1, (int)floorf((lo_y - 15.f) * (1.f / 16.f)) - 1));
                int b1 = max(y0, min(y1, (int)floorf(hi_y * (1.f / 16.f)) + 2));
                while (b0 < y1 && by0 > 16.f * (float)b0 + 15.f) ++b0;
                while (b1 > b0 && by1 < 16.f * (float)(b1 - 1)) --b1;
                x0 = a0; x1 = max(a0, a1); y0 = b0; y1 = max(b0, b1);
                touched = (uint32_t)((x1 - x0) * (y1 - y0));
                // Inside that rectangle the box is only the hull of the footprint: a tile
                // none of whose four 8x8 blocks passes the blend passes' own conic test
                // contributes nothing either.  Rectangles of up to 64 tiles get one bit per
                // tile (row-major) from the workgroup step below; larger ones stay whole.
                if (touched > 0 && touched <= 64u && x0 < 32768 && y0 < 32768) {
                    cand = touched;
#pragma unroll
                    for (int k = 0; k < 6; ++k) srec[threadIdx.x * 6 + k] = rr[k];
                }
                if (touched == 0) { dlo = 0xFFFFFFFFu; dhi = 0u; }
            }
        }
    }
    if (tight) {
        // The candidate tiles of the block's 256 splats, spread evenly over the lanes (a lane
        // walking its own splat's rectangle would hold its whole wave for the largest one).
        __shared__ uint32_t cstart[257];
        __shared__ uint32_t cw[4];
        __shared__ unsigned long long sbits[256];
        const uint32_t inc = wave_scan_incl(cand);
        const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
        if (lane == 63) cw[wv] = inc;
        sbits[threadIdx.x] = 0ull;
        __syncthreads();
        uint32_t off = inc - cand;
        for (int k = 0; k < wv; ++k) off += cw[k];
        cstart[threadIdx.x] = off;
        if (threadIdx.x == 255) cstart[256] = off + cand;
        sgeo[threadIdx.x] = make_int4(x0, y0, x1 - x0, 0);
        __syncthreads();
        const uint32_t total = cstart[256];
        for (uint32_t q = threadIdx.x; q < total; q += 256) {
            int lo = 0;   // largest j with cstart[j] <= q; splats without candidates lose ties
#pragma unroll
            for (int step = 128; step > 0; step >>= 1)
                if (cstart[lo + step] <= q) lo += step;
            const int4 gq = sgeo[lo];
            const int t = (int)(q - cstart[lo]);
            const int ty = t / gq.z, tx = t - ty * gq.z;
            const float4* rec = srec + lo * 6;
            if (block_mask(rec, rec[5], (float)((gq.x + tx) * 16), (float)((gq.y + ty) * 16)) != 0u)
                atomicOr(&sbits[lo], 1ull << t);
        }
        __syncthreads();
        if (cand) {
            const unsigned long long bits = sbits[threadIdx.x];
            touched = (uint32_t)__popcll(bits);
            packed_rect = make_int4((int)(0x80000000u | (uint32_t)x0 | ((uint32_t)y0 << 15)), x1 - x0,
                                    (int)(uint32_t)bits, (int)(uint32_t)(bits >> 32));
            is_packed = true;
            if (touched == 0) { dlo = 0xFFFFFFFFu; dhi = 0u; }
        }
    }
    if (i < N) {
        tiles_touched[i] = touched;
        reinterpret_cast<int4*>(rect)[i] = is_packed ? packed_rect : make_int4(x0, y0, x1, y1);
    }
    // per-block instance count (feeds the exclusive scan over blocks)
    uint32_t s = wave_sum_u32(touched);
    // range of the depth bits (positive floats order like their bit patterns): the sort only
    // has to cover the bits in which the keys can differ
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        dlo = min(dlo, (uint32_t)__shfl_xor(dlo, o, 64));
        dhi = max(dhi, (uint32_t)__shfl_xor(dhi, o, 64));
    }
    if ((threadIdx.x & 63) == 0) {
        wsum[threadIdx.x >> 6] = s; wmin[threadIdx.x >> 6] = dlo; wmax[threadIdx.x >> 6] = dhi;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        block_counts[blockIdx.x] = wsum[0] + wsum[1] + wsum[2] + wsum[3];
        block_dmin[blockIdx.x] = min(min(wmin[0], wmin[1]), min(wmin[2], wmin[3]));
        block_dmax[blockIdx.x] = max(max(wmax[0], wmax[1]), max(wmax[2], wmax[3]));
    }
}

// Single workgroup: in-place exclusive scan of block_counts[0..nb), total in [nb].
__global__ __launch_bounds__(1024) void scan_blocks_kernel(uint32_t* __restrict__ counts, int nb,
                                                           uint32_t* __restrict__ host_tail,
                                                           uint32_t generation = 0u) {
    __shared__ uint32_t wtot[16];
    __shared__ uint32_t carry_s;
    if (threadIdx.x == 0) carry_s = 0;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    // depth-bit range over the blocks (independent of the scan: its loads are issued first);
    // results live behind the per-block arrays: counts[3nb+1] = min, [3nb+2] = max,
    // [3nb+3] = copy of the total
    uint32_t lo = 0xFFFFFFFFu, hi = 0u;
    for (int i = threadIdx.x; i < nb; i += 1024) {
        lo = min(lo, counts[nb + 1 + i]);
        hi = max(hi, counts[2 * nb + 1 + i]);
    }
    __syncthreads();
    // four consecutive entries per lane: up to 4096 blocks (1 M splats) in one round
    for (int base = 0; base < nb; base += 4096) {
        const int i0 = base + 4 * (int)threadIdx.x;
        uint32_t v[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) v[k] = i0 + k < nb ? counts[i0 + k] : 0u;
        const uint32_t mine = (v[0] + v[1]) + (v[2] + v[3]);
        const uint32_t inc = wave_scan_incl(mine);
        if (lane == 63) wtot[w] = inc;
        __syncthreads();
        uint32_t run = carry_s + inc - mine;
        for (int k = 0; k < w; ++k) run += wtot[k];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if (i0 + k < nb) counts[i0 + k] = run;
            run += v[k];
        }
        __syncthreads();
        if (threadIdx.x == 1023) carry_s = run;
        __syncthreads();
    }
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        lo = min(lo, (uint32_t)__shfl_xor(lo, o, 64));
        hi = max(hi, (uint32_t)__shfl_xor(hi, o, 64));
    }
    __shared__ uint32_t rlo[16], rhi[16];
    if (lane == 0) { rlo[w] = lo; rhi[w] = hi; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int k = 1; k < 16; ++k) { lo = min(lo, rlo[k]); hi = max(hi, rhi[k]); }
        counts[nb] = carry_s;
        counts[3 * nb + 1] = lo; counts[3 * nb + 2] = hi; counts[3 * nb + 3] = carry_s;
        // the same three words straight into pinned host memory: the host reads them after
        // the stream has drained, without a copy command in between
        host_tail[0] = lo; host_tail[1] = hi; host_tail[2] = carry_s;
        if (generation) host_tail[3] = generation;   // whose launch filled this mailbox
    }
}

// Gradients of one splat w.r.t. its position, activated scales / rotation / opacity, colour,
// its 2-D mean and the two score inputs, from the blend pass's per-splat accumulator row.
struct SplatGrad {
    float dm[3], ds[2], dq[4], dop, dc[3], dxy[2], sc[2];
};

__device__ __forceinline__ void splat_backward(
    int i, int H, int W, float scale_mod, const float* __restrict__ means3D,
    const float* __restrict__ scales, const float* __restrict__ rots,
    const float* __restrict__ view, const float* __restrict__ proj, int radii_i,
    const float* __restrict__ dgeom, SplatGrad& o, bool scores_only = false) {
    float dm[3] = {0, 0, 0}, ds[2] = {0, 0}, dq[4] = {0, 0, 0, 0};
    float dop = 0, dc[3] = {0, 0, 0}, dxy[2] = {0, 0}, sc[2] = {0, 0};
    if (radii_i > 0 && scores_only) {
        // a splat the optimiser will not step: its two scores are all anybody reads
        const float4 g4v = reinterpret_cast<const float4*>(dgeom + (size_t)i * VINGS_GRAD_STRIDE)[4];
        sc[0] = g4v.z; sc[1] = g4v.w;
    } else if (radii_i > 0) {
        const float4* g4 = reinterpret_cast<const float4*>(dgeom + (size_t)i * VINGS_GRAD_STRIDE);
        const float4 g0 = g4[0], g1 = g4[1], g2 = g4[2], g3 = g4[3], g4v = g4[4];
        float dA[3][3] = {{g0.x, g0.y, g0.z}, {g0.w, g1.x, g1.y}, {g1.z, g1.w, g2.x}};
        dxy[0] = g2.y; dxy[1] = g2.z;
        const float dnv[3] = {g2.w, g3.x, g3.y};
        dop = g3.z;
        dc[0] = g3.w; dc[1] = g4v.x; dc[2] = g4v.y;
        sc[0] = g4v.z; sc[1] = g4v.w;

        const PixProj B = pixel_projection(proj, W, H);
        SplatFrame f;
        splat_frame(means3D, scales, rots, view, B, scale_mod, i, f);
        if (dxy[0] != 0.f || dxy[1] != 0.f) {
            const float* Tu = f.A[0];
            const float* Tv = f.A[1];
            const float* Tw = f.A[2];
            const float t[3] = {9.f, 9.f, -1.f};
            const float d = t[0] * Tw[0] * Tw[0] + t[1] * Tw[1] * Tw[1] + t[2] * Tw[2] * Tw[2];
            const float inv = 1.f / d;
            const float ff[3] = {inv * t[0], inv * t[1], inv * t[2]};
            const float cx = ff[0] * Tu[0] * Tw[0] + ff[1] * Tu[1] * Tw[1] + ff[2] * Tu[2] * Tw[2];
            const float cy = ff[0] * Tv[0] * Tw[0] + ff[1] * Tv[1] * Tw[1] + ff[2] * Tv[2] * Tw[2];
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                dA[0][k] += dxy[0] * ff[k] * Tw[k];
                dA[1][k] += dxy[1] * ff[k] * Tw[k];
                dA[2][k] += dxy[0] * (ff[k] * Tu[k] - 2 * cx * ff[k] * Tw[k]) +
                            dxy[1] * (ff[k] * Tv[k] - 2 * cy * ff[k] * Tw[k]);
            }
        }
        float dL0[3], dL1[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            dL0[k] = B.b[0][k] * dA[0][0] + B.b[1][k] * dA[1][0] + B.b[2][k] * dA[2][0];
            dL1[k] = B.b[0][k] * dA[0][1] + B.b[1][k] * dA[1][1] + B.b[2][k] * dA[2][1];
            dm[k] = B.b[0][k] * dA[0][2] + B.b[1][k] * dA[1][2] + B.b[2][k] * dA[2][2];
        }
        const float dn[3] = {f.mult * dnv[0], f.mult * dnv[1], f.mult * dnv[2]};
        float dnw[3];
        dnw[0] = view[0] * dn[0] + view[1] * dn[1] + view[2] * dn[2];
        dnw[1] = view[4] * dn[0] + view[5] * dn[1] + view[6] * dn[2];
        dnw[2] = view[8] * dn[0] + view[9] * dn[1] + view[10] * dn[2];
        float g[3][3];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            g[k][0] = f.sx * dL0[k];
            g[k][1] = f.sy * dL1[k];
            g[k][2] = dnw[k];
        }
        ds[0] = scale_mod * (f.R[0][0] * dL0[0] + f.R[1][0] * dL0[1] + f.R[2][0] * dL0[2]);
        ds[1] = scale_mod * (f.R[0][1] * dL1[0] + f.R[1][1] * dL1[1] + f.R[2][1] * dL1[2]);
        const float4 q = reinterpret_cast<const float4*>(rots)[i];
        const float r = q.x, x = q.y, y = q.z, z = q.w;
        dq[0] = 2 * (-z * g[0][1] + y * g[0][2] + z * g[1][0] - x * g[1][2] - y * g[2][0] + x * g[2][1]);
        dq[1] = 2 * (y * g[0][1] + z * g[0][2] + y * g[1][0] - 2 * x * g[1][1] - r * g[1][2] +
                     z * g[2][0] + r * g[2][1] - 2 * x * g[2][2]);
        dq[2] = 2 * (-2 * y * g[0][0] + x * g[0][1] + r * g[0][2] + x * g[1][0] + z * g[1][2] -
                     r * g[2][0] + z * g[2][1] - 2 * y * g[2][2]);
        dq[3] = 2 * (-2 * z * g[0][0] - r * g[0][1] + x * g[0][2] + r * g[1][0] - 2 * z * g[1][1] +
                     y * g[1][2] + x * g[2][0] + y * g[2][1]);
    }
#pragma unroll
    for (int k = 0; k < 3; ++k) { o.dm[k] = dm[k]; o.dc[k] = dc[k]; }
#pragma unroll
    for (int k = 0; k < 4; ++k) o.dq[k] = dq[k];
    o.ds[0] = ds[0]; o.ds[1] = ds[1]; o.dop = dop;
    o.dxy[0] = dxy[0]; o.dxy[1] = dxy[1]; o.sc[0] = sc[0]; o.sc[1] = sc[1];
}

__global__ __launch_bounds__(256) void project_bwd_kernel(
    int N, int H, int W, float scale_mod, const float* __restrict__ means3D,
    const float* __restrict__ scales, const float* __restrict__ rots,
    const float* __restrict__ view, const float* __restrict__ proj,
    const int32_t* __restrict__ radii, const float* __restrict__ dgeom,
    float* __restrict__ d_means3D, float* __restrict__ d_scales, float* __restrict__ d_rots,
    float* __restrict__ d_opac, float* __restrict__ d_colors, float* __restrict__ d_means2D,
    float* __restrict__ scores) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= N) return;
    SplatGrad g;
    splat_backward(i, H, W, scale_mod, means3D, scales, rots, view, proj, radii[i], dgeom, g);
    d_means3D[3 * i] = g.dm[0]; d_means3D[3 * i + 1] = g.dm[1]; d_means3D[3 * i + 2] = g.dm[2];
    reinterpret_cast<float2*>(d_scales)[i] = make_float2(g.ds[0], g.ds[1]);
    reinterpret_cast<float4*>(d_rots)[i] = make_float4(g.dq[0], g.dq[1], g.dq[2], g.dq[3]);
    d_opac[i] = g.dop;
    d_colors[3 * i] = g.dc[0]; d_colors[3 * i + 1] = g.dc[1]; d_colors[3 * i + 2] = g.dc[2];
    if (d_means2D) {
        d_means2D[3 * i] = g.dxy[0]; d_means2D[3 * i + 1] = g.dxy[1]; d_means2D[3 * i + 2] = 0.f;
    }
    if (scores) reinterpret_cast<float2*>(scores)[i] = make_float2(g.sc[0], g.sc[1]);
}

// The same per-splat backward with the mapper's chain rule, score weighting and bookkeeping
// (splat_finish) applied in registers: the seven gradient arrays w.r.t. the activated
// parameters never reach memory.  `scores` (optional) still receives the view's two scores.
__global__ __launch_bounds__(256) void project_bwd_finish_kernel(
    int N, int H, int W, float scale_mod, const float* __restrict__ means3D,
    const float* __restrict__ rots, const float* __restrict__ view,
    const float* __restrict__ proj, const float* __restrict__ dgeom, FinishArgs a,
    float* __restrict__ scores, const uint32_t* __restrict__ guard_count, uint32_t guard_limit) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= N) return;
    // the view's instance list did not fit the buffers it was binned into: its gradients are
    // incomplete, so nothing of this step may reach the records or the parameters
    if (guard_count && *guard_count > guard_limit) return;
    SplatGrad g;
    // (stable splats: gaussian_base.py:400-402 keeps them out of the optimiser step, so their
    // gradients come out as zeros here without the chain behind them; the scores are recorded)
    splat_backward(i, H, W, scale_mod, means3D, a.scales, rots, view, proj, a.radii[i], dgeom, g,
                   a.stable_mask && a.stable_mask[i]);
    if (scores) reinterpret_cast<float2*>(scores)[i] = make_float2(g.sc[0], g.sc[1]);
    splat_finish(i, a, g.dm, g.dc, g.dop, make_float2(g.ds[0], g.ds[1]),
                 make_float4(g.dq[0], g.dq[1], g.dq[2], g.dq[3]), g.sc[0], g.sc[1]);
}

// The instance count of a view comes back through pinned, device-visible host memory: a pageable
// device-to-host copy costs ~50 us more per view.  Every GPU gets a ring of kMailSlots 64-byte
// mailboxes with an event each, created at the first projection call on that GPU and kept for
// the life of the process (the only memory this library owns besides the optional event pool).
// A host thread takes a slot of its GPU's ring at its first launch there and keeps it: its
// launches and its waits use that slot, so host threads (or streams driven from different
// threads) that render on one GPU at the same time do not see each other's counts, however many
// launches the others make in between.  Only when more than kMailSlots threads have rendered on
// one GPU do two of them share a slot; every launch therefore stamps the mailbox with its own
// generation number, and a wait() that finds another launch's stamp (or whose event was
// re-recorded by the other thread) falls back to draining its own stream and copying the three
// words from the device, where the launch left a second copy behind the block prefix.
// wait() returns the calling thread's LATEST launch on the GPU.
namespace {
constexpr int kMaxDevices = 64;
constexpr int kMailSlots = 16;
std::mutex g_mail_mu[kMaxDevices];
uint32_t* g_mail[kMaxDevices] = {};                      // kMailSlots * 16 words
hipEvent_t g_mail_evt[kMaxDevices][kMailSlots] = {};
std::atomic<unsigned> g_mail_next[kMaxDevices];          // slots handed to threads so far
std::atomic<unsigned> g_mail_gen[kMaxDevices];           // launches so far
thread_local signed char t_slot_of_dev[kMaxDevices];     // this thread's slot per GPU, +1 (0 = none yet)
// the calling thread's latest launch: GPU, slot, stamp, stream, device copy of the three words
thread_local int t_mail_dev = -1, t_mail_slot = -1;
thread_local uint32_t t_mail_gen = 0;
thread_local hipStream_t t_mail_stream = nullptr;
thread_local const uint32_t* t_mail_devcopy = nullptr;
// 1: a splat is binned only into the tiles its alpha >= 1/255 box reaches (the default);
// 0: into every tile of the reference's 3-sigma square (forward.cu preprocess / getRect).
std::atomic<int> g_tight_binning{1};   // process default
thread_local int t_tight_binning = -1;  // the calling thread's choice; -1: the process default
}  // namespace

}  // namespace vings

using namespace vings;

// Launch half: projection + block scan, the three result words into the GPU's mailbox, and an
// event behind them.
static int project_launch(int N, int H, int W, float scale_modifier, const float* means3D,
                          const float* scales, const float* rotations, const float* opacities,
                          const float* colors, const float* flow, const float* viewmatrix,
                          const float* projmatrix, float* geom, float* depth, int32_t* radii,
                          int32_t* rect, uint32_t* tiles_touched, uint32_t* block_prefix,
                          hipStream_t s, int* dev_out, float* act_scales = nullptr,
                          float* act_rotations = nullptr, float* act_opacities = nullptr,
                          float* dgeom_clear = nullptr, uint32_t* user_tail = nullptr) {
    VINGS_REQUIRE(N > 0 && H > 0 && W > 0, "sizes");
    VINGS_REQUIRE(means3D && scales && rotations && opacities && colors && viewmatrix &&
                      projmatrix && geom && depth && radii && rect && tiles_touched && block_prefix,
                  "null pointer");
    VINGS_REQUIRE((((uintptr_t)geom | (uintptr_t)rotations | (uintptr_t)rect) & 15) == 0,
                  "geom/rotations/rect must be 16-byte aligned");
    int dev = 0;
    VINGS_CHECK_HIP(hipGetDevice(&dev));
    VINGS_REQUIRE(dev >= 0 && dev < kMaxDevices, "device index");
    if (!user_tail) {
        std::lock_guard<std::mutex> lk(g_mail_mu[dev]);
        if (!g_mail[dev]) {
            uint32_t* m = nullptr;
            VINGS_CHECK_HIP(hipHostMalloc((void**)&m, 64 * kMailSlots, hipHostMallocDefault));
            for (int k = 0; k < kMailSlots; ++k)
                VINGS_CHECK_HIP(hipEventCreateWithFlags(&g_mail_evt[dev][k], hipEventDisableTiming));
            g_mail[dev] = m;
        }
    }
    int slot = -1;
    uint32_t gen = 0;
    if (!user_tail) {
        if (t_slot_of_dev[dev] == 0)
            t_slot_of_dev[dev] = (signed char)(1 + g_mail_next[dev].fetch_add(1u, std::memory_order_relaxed) % kMailSlots);
        slot = t_slot_of_dev[dev] - 1;
        do gen = g_mail_gen[dev].fetch_add(1u, std::memory_order_relaxed) + 1u; while (gen == 0u);
    }
    const int nb = cdiv(N, 256);
    const int tight = t_tight_binning >= 0 ? t_tight_binning
                                           : g_tight_binning.load(std::memory_order_relaxed);
    {
    ProfScope prof(PROF_PROJECT_FWD, s);
    if (act_scales)
        project_fwd_kernel<true><<<nb, 256, 0, s>>>(
            N, H, W, scale_modifier, means3D, scales, rotations, opacities, act_scales,
            act_rotations, act_opacities, colors, flow, viewmatrix, projmatrix, geom, depth, radii,
            rect, tiles_touched, block_prefix, block_prefix + nb + 1, block_prefix + 2 * nb + 1, tight, dgeom_clear);
    else
        project_fwd_kernel<false><<<nb, 256, 0, s>>>(
            N, H, W, scale_modifier, means3D, scales, rotations, opacities, nullptr, nullptr,
            nullptr, colors, flow, viewmatrix, projmatrix, geom, depth, radii, rect, tiles_touched,
            block_prefix, block_prefix + nb + 1, block_prefix + 2 * nb + 1, tight, nullptr);
    VINGS_LAUNCH_CHECK();
    }
    if (user_tail) {
        // the caller's own count words, no mailbox slot, no event: nothing here that a stream
        // capture could not record
        scan_blocks_kernel<<<1, 1024, 0, s>>>(block_prefix, nb, user_tail);
        VINGS_LAUNCH_CHECK();
        *dev_out = dev;
        return VINGS_OK;
    }
    scan_blocks_kernel<<<1, 1024, 0, s>>>(block_prefix, nb, g_mail[dev] + 16 * slot, gen);
    VINGS_LAUNCH_CHECK();
    VINGS_CHECK_HIP(hipEventRecord(g_mail_evt[dev][slot], s));
    t_mail_dev = dev;
    t_mail_slot = slot;
    t_mail_gen = gen;
    t_mail_stream = s;
    t_mail_devcopy = block_prefix + 3 * nb + 1;
    *dev_out = dev;
    return VINGS_OK;
}

static int project_wait(int dev, int64_t* num_instances_host, uint32_t* depth_bits_range_host) {
    VINGS_REQUIRE(t_mail_dev == dev && t_mail_slot >= 0,
                  "this thread has launched no projection on the current GPU");
    const int slot = t_mail_slot;
    VINGS_CHECK_HIP(hipEventSynchronize(g_mail_evt[dev][slot]));
    volatile uint32_t* tail = g_mail[dev] + 16 * slot;   // depth-bit min, max, instance total, stamp
    uint32_t w[3] = {tail[0], tail[1], tail[2]};
    if (tail[3] != t_mail_gen) {
        // a thread sharing this slot launched in between: the mailbox (and perhaps the event)
        // is theirs now; this thread's own words are still behind its block prefix
        VINGS_CHECK_HIP(hipStreamSynchronize(t_mail_stream));
        VINGS_CHECK_HIP(hipMemcpy(w, t_mail_devcopy, sizeof(w), hipMemcpyDeviceToHost));
    }
    *num_instances_host = (int64_t)w[2];
    if (depth_bits_range_host) {
        depth_bits_range_host[0] = w[2] > 0 ? w[0] : 0u;
        depth_bits_range_host[1] = w[2] > 0 ? w[1] : 0xFFFFFFFFu;
    }
    return VINGS_OK;
}

extern "C" int vings_surfel_project(int N, int H, int W, float scale_modifier,
                                    const float* means3D, const float* scales,
                                    const float* rotations, const float* opacities,
                                    const float* colors, const float* flow,
                                    const float* viewmatrix, const float* projmatrix,
                                    float* geom, float* depth, int32_t* radii, int32_t* rect,
                                    uint32_t* tiles_touched, uint32_t* block_prefix,
                                    int64_t* num_instances_host, uint32_t* depth_bits_range_host,
                                    void* stream) {
    VINGS_REQUIRE(N >= 0 && H > 0 && W > 0, "sizes");
    VINGS_REQUIRE(num_instances_host != nullptr, "num_instances_host");
    *num_instances_host = 0;
    if (depth_bits_range_host) { depth_bits_range_host[0] = 0u; depth_bits_range_host[1] = 0xFFFFFFFFu; }
    if (N == 0) return VINGS_OK;
    int cur = 0;
    VINGS_CHECK_HIP(hipGetDevice(&cur));
    VINGS_REQUIRE(cur >= 0 && cur < kMaxDevices, "device index");
    int dev = 0;
    const int rc = project_launch(N, H, W, scale_modifier, means3D, scales, rotations, opacities,
                                  colors, flow, viewmatrix, projmatrix, geom, depth, radii, rect,
                                  tiles_touched, block_prefix, (hipStream_t)stream, &dev);
    if (rc != VINGS_OK) return rc;
    return project_wait(dev, num_instances_host, depth_bits_range_host);
}

extern "C" int vings_surfel_project_launch(int N, int H, int W, float scale_modifier,
                                           const float* means3D, const float* scales,
                                           const float* rotations, const float* opacities,
                                           const float* colors, const float* flow,
                                           const float* viewmatrix, const float* projmatrix,
                                           float* geom, float* depth, int32_t* radii,
                                           int32_t* rect, uint32_t* tiles_touched,
                                           uint32_t* block_prefix, void* stream) {
    int dev = 0;
    return project_launch(N, H, W, scale_modifier, means3D, scales, rotations, opacities, colors,
                          flow, viewmatrix, projmatrix, geom, depth, radii, rect, tiles_touched,
                          block_prefix, (hipStream_t)stream, &dev);
}

extern "C" int vings_surfel_project_launch_activate(
    int N, int H, int W, float scale_modifier, const float* means3D, const float* raw_scaling,
    const float* raw_rotation, const float* raw_opacity, const float* colors, const float* flow,
    const float* viewmatrix, const float* projmatrix, float* scales, float* rotations,
    float* opacities, float* geom, float* depth, int32_t* radii, int32_t* rect,
    uint32_t* tiles_touched, uint32_t* block_prefix, float* dgeom_clear, void* stream) {
    VINGS_REQUIRE(scales && rotations && opacities, "null pointer");
    VINGS_REQUIRE(((uintptr_t)dgeom_clear & 15) == 0, "dgeom must be 16-byte aligned");
    VINGS_REQUIRE((((uintptr_t)rotations | (uintptr_t)raw_rotation) & 15) == 0 &&
                      (((uintptr_t)scales | (uintptr_t)raw_scaling) & 7) == 0,
                  "rotation arrays must be 16-byte aligned, scaling arrays 8-byte aligned");
    int dev = 0;
    return project_launch(N, H, W, scale_modifier, means3D, raw_scaling, raw_rotation, raw_opacity,
                          colors, flow, viewmatrix, projmatrix, geom, depth, radii, rect,
                          tiles_touched, block_prefix, (hipStream_t)stream, &dev, scales, rotations,
                          opacities, dgeom_clear);
}

extern "C" int vings_surfel_project_enqueue_activate(
    int N, int H, int W, float scale_modifier, const float* means3D, const float* raw_scaling,
    const float* raw_rotation, const float* raw_opacity, const float* colors, const float* flow,
    const float* viewmatrix, const float* projmatrix, float* scales, float* rotations,
    float* opacities, float* geom, float* depth, int32_t* radii, int32_t* rect,
    uint32_t* tiles_touched, uint32_t* block_prefix, float* dgeom_clear, uint32_t* count_words,
    void* stream) {
    VINGS_REQUIRE(scales && rotations && opacities && count_words, "null pointer");
    VINGS_REQUIRE(((uintptr_t)dgeom_clear & 15) == 0, "dgeom must be 16-byte aligned");
    VINGS_REQUIRE((((uintptr_t)rotations | (uintptr_t)raw_rotation) & 15) == 0 &&
                      (((uintptr_t)scales | (uintptr_t)raw_scaling) & 7) == 0,
                  "rotation arrays must be 16-byte aligned, scaling arrays 8-byte aligned");
    int dev = 0;
    return project_launch(N, H, W, scale_modifier, means3D, raw_scaling, raw_rotation, raw_opacity,
                          colors, flow, viewmatrix, projmatrix, geom, depth, radii, rect,
                          tiles_touched, block_prefix, (hipStream_t)stream, &dev, scales, rotations,
                          opacities, dgeom_clear, count_words);
}

extern "C" int vings_surfel_set_tight_binning(int on) {
    return g_tight_binning.exchange(on ? 1 : 0);
}

extern "C" int vings_surfel_set_tight_binning_thread(int mode) {
    const int prev = t_tight_binning;
    t_tight_binning = mode < 0 ? -1 : (mode ? 1 : 0);
    return prev;
}

extern "C" int vings_surfel_project_wait(int64_t* num_instances_host,
                                         uint32_t* depth_bits_range_host) {
    VINGS_REQUIRE(num_instances_host != nullptr, "num_instances_host");
    int dev = 0;
    VINGS_CHECK_HIP(hipGetDevice(&dev));
    VINGS_REQUIRE(dev >= 0 && dev < kMaxDevices && g_mail[dev], "no projection was launched on this GPU");
    return project_wait(dev, num_instances_host, depth_bits_range_host);
}

extern "C" int vings_surfel_project_backward(int N, int H, int W, float scale_modifier,
                                             const float* means3D, const float* scales,
                                             const float* rotations, const float* viewmatrix,
                                             const float* projmatrix, const int32_t* radii,
                                             const float* dgeom, float* dL_dmeans3D,
                                             float* dL_dscales, float* dL_drotations,
                                             float* dL_dopacities, float* dL_dcolors,
                                             float* dL_dmeans2D, float* scores_grad,
                                             void* stream) {
    VINGS_REQUIRE(N >= 0 && H > 0 && W > 0, "sizes");
    if (N == 0) return VINGS_OK;
    VINGS_REQUIRE(means3D && scales && rotations && viewmatrix && projmatrix && radii && dgeom &&
                      dL_dmeans3D && dL_dscales && dL_drotations && dL_dopacities && dL_dcolors,
                  "null pointer");
    hipStream_t s = (hipStream_t)stream;
    ProfScope prof(PROF_PROJECT_BWD, s);
    project_bwd_kernel<<<cdiv(N, 256), 256, 0, s>>>(
        N, H, W, scale_modifier, means3D, scales, rotations, viewmatrix, projmatrix, radii, dgeom,
        dL_dmeans3D, dL_dscales, dL_drotations, dL_dopacities, dL_dcolors, dL_dmeans2D,
        scores_grad);
    VINGS_LAUNCH_CHECK();
    return VINGS_OK;
}

static int project_backward_finish(
    int N, int H, int W, float scale_modifier, const float* means3D, const float* scales,
    const float* rotations, const float* viewmatrix, const float* projmatrix, const int32_t* radii,
    const float* dgeom, const float* opacities, const float* raw_rotation,
    const uint8_t* stable_mask, float* global_scores, float* local_scores, float* kf_max_scores,
    int64_t* kf_id, int64_t this_kf, int bookkeeping, float* g_xyz, float* g_rgb, float* g_opacity,
    float* g_scaling, float* g_rotation, uint8_t* visible, float* scores_grad, float* packed,
    int64_t* record_keys, const int32_t* slot_of_row, int32_t* rows_of_slot, int chunk_rows,
    void* stream, const uint32_t* guard_count = nullptr, uint32_t guard_limit = 0u) {
    VINGS_REQUIRE(N >= 0 && H > 0 && W > 0, "sizes");
    if (N == 0) return VINGS_OK;
    VINGS_REQUIRE(means3D && scales && rotations && viewmatrix && projmatrix && radii && dgeom &&
                      opacities && raw_rotation && global_scores,
                  "null input");
    VINGS_REQUIRE(packed ? (record_keys != nullptr && !bookkeeping)
                         : (g_xyz && g_rgb && g_opacity && g_scaling && g_rotation && visible),
                  "null output (packed needs record_keys and bookkeeping = 0)");
    VINGS_REQUIRE(!bookkeeping || (local_scores && kf_max_scores && kf_id),
                  "bookkeeping needs the score arrays");
    VINGS_REQUIRE((((uintptr_t)raw_rotation | (uintptr_t)rotations | (uintptr_t)g_rotation |
                    (uintptr_t)dgeom) & 15) == 0 &&
                      (((uintptr_t)scales | (uintptr_t)g_scaling | (uintptr_t)scores_grad) & 7) == 0,
                  "rotation arrays / dgeom must be 16-byte aligned, scaling / score arrays 8-byte aligned");
    FinishArgs a = {};
    a.opacities = opacities; a.scales = scales; a.raw_rotation = raw_rotation;
    a.radii = radii; a.stable_mask = stable_mask; a.global_scores = global_scores;
    a.local_scores = local_scores; a.kf_max_scores = kf_max_scores; a.kf_id = kf_id;
    a.g_xyz = g_xyz; a.g_rgb = g_rgb; a.g_opacity = g_opacity; a.g_scaling = g_scaling;
    a.g_rotation = g_rotation; a.visible = visible; a.this_kf = this_kf;
    a.bookkeeping = bookkeeping;
    a.packed = packed; a.record_keys = reinterpret_cast<long long*>(record_keys); a.N = N;
    a.slot_of_row = slot_of_row; a.rows_of_slot = rows_of_slot; a.chunk_rows = chunk_rows;
    hipStream_t s = (hipStream_t)stream;
    ProfScope prof(PROF_PROJECT_BWD, s);
    project_bwd_finish_kernel<<<cdiv(N, 256), 256, 0, s>>>(
        N, H, W, scale_modifier, means3D, rotations, viewmatrix, projmatrix, dgeom, a, scores_grad,
        guard_count, guard_limit);
    VINGS_LAUNCH_CHECK();
    return VINGS_OK;
}

extern "C" int vings_surfel_project_backward_finish(
    int N, int H, int W, float scale_modifier, const float* means3D, const float* scales,
    const float* rotations, const float* viewmatrix, const float* projmatrix, const int32_t* radii,
    const float* dgeom, const float* opacities, const float* raw_rotation,
    const uint8_t* stable_mask, float* global_scores, float* local_scores, float* kf_max_scores,
    int64_t* kf_id, int64_t this_kf, int bookkeeping, float* g_xyz, float* g_rgb, float* g_opacity,
    float* g_scaling, float* g_rotation, uint8_t* visible, float* scores_grad, float* packed,
    int64_t* record_keys, void* stream) {
    return project_backward_finish(N, H, W, scale_modifier, means3D, scales, rotations, viewmatrix,
                                   projmatrix, radii, dgeom, opacities, raw_rotation, stable_mask,
                                   global_scores, local_scores, kf_max_scores, kf_id, this_kf,
                                   bookkeeping, g_xyz, g_rgb, g_opacity, g_scaling, g_rotation,
                                   visible, scores_grad, packed, record_keys, nullptr, nullptr, 0,
                                   stream);
}

extern "C" int vings_surfel_project_backward_pack(
    int N, int H, int W, float scale_modifier, const float* means3D, const float* scales,
    const float* rotations, const float* viewmatrix, const float* projmatrix, const int32_t* radii,
    const float* dgeom, const float* opacities, const float* raw_rotation,
    const float* global_scores, int64_t this_kf, float* scores_grad, float* packed,
    int64_t* record_keys, const int32_t* slot_of_row, int32_t* rows_of_slot, int chunk_rows,
    void* stream) {
    VINGS_REQUIRE(packed && record_keys && slot_of_row && rows_of_slot && chunk_rows > 0,
                  "the compact exchange form needs its buffer, the slot tables and the chunk length");
    return project_backward_finish(N, H, W, scale_modifier, means3D, scales, rotations, viewmatrix,
                                   projmatrix, radii, dgeom, opacities, raw_rotation, nullptr,
                                   const_cast<float*>(global_scores), nullptr, nullptr, nullptr,
                                   this_kf, 0, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr,
                                   scores_grad, packed, record_keys, slot_of_row, rows_of_slot,
                                   chunk_rows, stream);
}

extern "C" int vings_surfel_project_backward_finish_guarded(
    int N, int H, int W, float scale_modifier, const float* means3D, const float* scales,
    const float* rotations, const float* viewmatrix, const float* projmatrix, const int32_t* radii,
    const float* dgeom, const float* opacities, const float* raw_rotation,
    const uint8_t* stable_mask, float* global_scores, float* local_scores, float* kf_max_scores,
    int64_t* kf_id, int64_t this_kf, float* g_xyz, float* g_rgb, float* g_opacity,
    float* g_scaling, float* g_rotation, uint8_t* visible, const uint32_t* instance_count,
    uint32_t capacity, void* stream) {
    VINGS_REQUIRE(instance_count != nullptr, "instance_count");
    return project_backward_finish(N, H, W, scale_modifier, means3D, scales, rotations, viewmatrix,
                                   projmatrix, radii, dgeom, opacities, raw_rotation, stable_mask,
                                   global_scores, local_scores, kf_max_scores, kf_id, this_kf, 1,
                                   g_xyz, g_rgb, g_opacity, g_scaling, g_rotation, visible, nullptr,
                                   nullptr, nullptr, nullptr, nullptr, 0, stream, instance_count,
                                   capacity);
}
