// Mapping loss of one rendered view and its gradient with respect to the
// rasteriser outputs, fused into five launches (gfx950).
//
// Replaces, for the hot loop, the PyTorch graph the reference builds per view:
//   render post-processing  /root/reference/scripts/gaussian/gaussian_base.py:205-224
//   depth -> normal         /root/reference/scripts/gaussian/normal_utils.py:6-21
//   get_loss (use_sky off)  /root/reference/scripts/gaussian/loss_utils.py:110-153
//   ssim                    /root/reference/scripts/gaussian/loss_utils.py:26-57
// and the autograd pass through it, i.e. everything between the rasteriser's
// forward outputs (rgb [3,H,W], allmap [9,H,W]) and the dL/drgb, dL/dallmap it
// is handed back in backward.  There the graph costs ~12 grouped 11x11
// convolutions and ~100 elementwise launches per view; here:
//
//   count     valid / sky pixel counts (they scale every gradient)
//   ssim_fwd  separable 11-tap Gaussian moments of (x, y, xx, yy, xy) through
//             LDS, SSIM value, and the three partial-derivative maps
//   ssim_bwd  the same separable filter over the derivative maps -> dL/drgb,
//             plus the L1 term
//   geom      depth / normal / alpha / distortion terms: depth = A0 / alpha,
//             central-difference surface normal with its 13-pixel adjoint
//             stencil evaluated from an LDS tile -> dL/dallmap
//   final     the eight loss scalars
//
// Everything is HBM-bound streaming work (about 40 floats per pixel in total);
// the LDS tiles exist so each input pixel is read once per kernel.
#include "common.hpp"

namespace vings {

constexpr int kWin = 11, kHalo = 5;
constexpr int kST = 32;                 // SSIM output tile (kST x kST), 256 lanes
constexpr int kSIn = kST + 2 * kHalo;   // 42
constexpr float kC1 = 0.01f * 0.01f, kC2 = 0.03f * 0.03f;

struct Gauss { float g[kWin]; };

struct LossSums {          // pixel counts; the loss sums go through per-block partials
    unsigned int n_valid, n_sky;
};
// Per-block partial sums (float64), reduced by loss_final_kernel.  Thousands of blocks
// adding to one address would serialise in L2 for longer than the kernels run.
constexpr int kSsimParts = 2;  // ssim, l1
constexpr int kGeomParts = 4;  // normal, alpha, depth, dist

__device__ __forceinline__ bool is_sky(const float* __restrict__ gt_rgb, size_t HW, size_t p) {
    return (gt_rgb[p] + gt_rgb[HW + p]) + gt_rgb[2 * HW + p] == 0.0f;
}

__device__ __forceinline__ double block_sum(double v, double* red) {
    // 256 lanes: DPP-free shuffle tree per wave, then one LDS hop
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    const int wv = threadIdx.x >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[wv] = v;
    __syncthreads();
    return red[0] + red[1] + red[2] + red[3];
}

// Pixel classes of the ground truth, one byte per pixel, written by ssim_fwd_kernel (every
// pixel belongs to exactly one of its tiles) together with the two class counts: the later
// kernels read these instead of re-deriving them from four float planes each.
constexpr uint8_t kPixValid = 1, kPixSky = 2;

// SSIM forward: per colour channel, moments through a separable Gaussian with zero
// padding.  maps[(3*ch + m) * HW + p], m = 0..2: valid(p) * d ssim / d {mu_x, E[xx], E[xy]}.
// Two of the five moments (mean and second moment of the ground-truth image), the pixel
// classes and their counts depend on the keyframe's ground truth alone, which does not change
// while the mapper iterates over it.  kMode 0: everything computed, nothing kept.  kMode 1:
// the same, and those parts are left in gt_moments [6,H,W] / cls / sums for later calls.
// kMode 2: they are read from there; three moments go through the filter instead of five
// (40 % fewer filter operations, 31 KB of LDS instead of 42 KB: five workgroups per CU
// instead of three).  Bit-identical results in all modes (every moment has its own chain of
// multiply-adds).
template <int kMode>
__global__ __launch_bounds__(256, kMode == 2 ? 4 : 3) void ssim_fwd_kernel(int H, int W, Gauss gw,
                                                       const float* __restrict__ rgb,
                                                       const float* __restrict__ gt_rgb,
                                                       const float* __restrict__ gt_depth,
                                                       uint8_t* __restrict__ cls,
                                                       LossSums* __restrict__ sums,
                                                       float* __restrict__ maps,
                                                       double* __restrict__ partials,
                                                       float* __restrict__ gt_moments) {
    constexpr int kM = kMode == 2 ? 3 : 5;      // moments through the filter
    __shared__ float sx[kSIn][kSIn + 1], sy[kSIn][kSIn + 1];
    __shared__ float hz[kM][kSIn][kST + 1];
    __shared__ double red[4];
    __shared__ unsigned cnt[2][4];
    const size_t HW = (size_t)H * W;
    const int tx0 = blockIdx.x * kST, ty0 = blockIdx.y * kST;
    const int c = threadIdx.x & 31, rq = threadIdx.x >> 5;  // output column, first output row
    double ssim_acc = 0.0, l1_acc = 0.0;
    // validity of this lane's four output pixels: rows 4 rq .. 4 rq + 3 of column c, so the
    // vertical filter reads 14 rows of the horizontal result for 4 outputs instead of 44
    bool vld[4];
    unsigned nv = 0, ns = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int gy = ty0 + 4 * rq + i, gx = tx0 + c;
        vld[i] = false;
        if (gy < H && gx < W) {
            const size_t p = (size_t)gy * W + gx;
            if (kMode == 2) {
                vld[i] = cls[p] == kPixValid;
            } else {
                const bool sky = is_sky(gt_rgb, HW, p);
                vld[i] = (!sky) && gt_depth[p] > 0.0f;
                cls[p] = sky ? kPixSky : (vld[i] ? kPixValid : 0);
                ns += sky;
                nv += vld[i];
            }
        }
    }
    // The tile of a channel (42 x 42 values of both images, seven per lane) is fetched into
    // registers while the previous channel is being filtered, so that only the first fetch of a
    // workgroup waits for memory.
    constexpr int kFetch = (kSIn * kSIn + 255) / 256;
    float fxv[kFetch], fyv[kFetch];
    auto fetch = [&](int ch) {
        const float* __restrict__ X = rgb + ch * HW;
        const float* __restrict__ Y = gt_rgb + ch * HW;
#pragma unroll
        for (int k = 0; k < kFetch; ++k) {
            const int idx = k * 256 + threadIdx.x;
            const int r = idx / kSIn, cc = idx - r * kSIn;
            const int gy = ty0 + r - kHalo, gx = tx0 + cc - kHalo;
            fxv[k] = 0.f; fyv[k] = 0.f;
            if (idx < kSIn * kSIn && gy >= 0 && gy < H && gx >= 0 && gx < W) {
                fxv[k] = X[(size_t)gy * W + gx];
                fyv[k] = Y[(size_t)gy * W + gx];
            }
        }
    };
    fetch(0);
    for (int ch = 0; ch < 3; ++ch) {
#pragma unroll
        for (int k = 0; k < kFetch; ++k) {
            const int idx = k * 256 + threadIdx.x;
            if (idx < kSIn * kSIn) {
                const int r = idx / kSIn, cc = idx - r * kSIn;
                sx[r][cc] = fxv[k]; sy[r][cc] = fyv[k];
            }
        }
        __syncthreads();
        if (ch < 2) fetch(ch + 1);
        // horizontal pass: a lane produces 4 neighbouring columns of one row from 14 reads
        for (int idx = threadIdx.x; idx < kSIn * (kST / 4); idx += 256) {
            const int r = idx >> 3, c0 = (idx & 7) * 4;
            // filter rows: 0 mu_x, 1 E[xx], 2 E[xy], and unless cached 3 mu_y, 4 E[yy]
            float hm[kM][4];
#pragma unroll
            for (int m = 0; m < kM; ++m)
#pragma unroll
                for (int i = 0; i < 4; ++i) hm[m][i] = 0.f;
#pragma unroll
            for (int kk = 0; kk < kWin + 3; ++kk) {
                const float xv = sx[r][c0 + kk], yv = sy[r][c0 + kk];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    if (kk - i >= 0 && kk - i < kWin) {
                        const float g = gw.g[kk - i];
                        const float gx_ = g * xv;
                        hm[0][i] += gx_; hm[1][i] += gx_ * xv; hm[2][i] += gx_ * yv;
                        if (kM == 5) {
                            const float gy_ = g * yv;
                            hm[3][i] += gy_; hm[4][i] += gy_ * yv;
                        }
                    }
                }
            }
#pragma unroll
            for (int m = 0; m < kM; ++m)
#pragma unroll
                for (int i = 0; i < 4; ++i) hz[m][r][c0 + i] = hm[m][i];
        }
        __syncthreads();
        float vm[kM][4];
#pragma unroll
        for (int m = 0; m < kM; ++m)
#pragma unroll
            for (int i = 0; i < 4; ++i) vm[m][i] = 0.f;
#pragma unroll
        for (int kk = 0; kk < kWin + 3; ++kk) {     // sliding window over 14 rows
            float hv[kM];
#pragma unroll
            for (int m = 0; m < kM; ++m) hv[m] = hz[m][4 * rq + kk][c];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                if (kk - i >= 0 && kk - i < kWin) {
                    const float g = gw.g[kk - i];
#pragma unroll
                    for (int m = 0; m < kM; ++m) vm[m][i] += g * hv[m];
                }
            }
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int r = 4 * rq + i;
            const int gy = ty0 + r, gx = tx0 + c;
            const float mu1 = vm[0][i], e11 = vm[1][i], e12 = vm[2][i];
            if (gy < H && gx < W) {
                const size_t p = (size_t)gy * W + gx;
                float mu2, e22;
                if (kMode == 2) {
                    mu2 = gt_moments[(size_t)ch * HW + p];
                    e22 = gt_moments[(size_t)(3 + ch) * HW + p];
                } else {
                    mu2 = vm[kM - 2][i]; e22 = vm[kM - 1][i];
                    if (kMode == 1) {
                        gt_moments[(size_t)ch * HW + p] = mu2;
                        gt_moments[(size_t)(3 + ch) * HW + p] = e22;
                    }
                }
                const float mu1_sq = mu1 * mu1, mu2_sq = mu2 * mu2, mu12 = mu1 * mu2;
                const float s1 = e11 - mu1_sq, s2 = e22 - mu2_sq, s12 = e12 - mu12;
                const float a = 2.f * mu12 + kC1, b = 2.f * s12 + kC2;
                const float cq = mu1_sq + mu2_sq + kC1, d = s1 + s2 + kC2;
                const float rcd = 1.0f / (cq * d);
                const float ssim = a * b * rcd;
                const float v = vld[i] ? 1.f : 0.f;
                // partial derivatives with (mu_x, E[xx], E[xy]) as the independent inputs
                const float d_mu1 = 2.f * mu2 * (b - a) * rcd - 2.f * mu1 * ssim * (d - cq) * rcd;
                const float d_e11 = -ssim / d;
                const float d_e12 = 2.f * a * rcd;
                maps[(size_t)(3 * ch + 0) * HW + p] = v * d_mu1;
                maps[(size_t)(3 * ch + 1) * HW + p] = v * d_e11;
                maps[(size_t)(3 * ch + 2) * HW + p] = v * d_e12;
                if (vld[i]) {
                    ssim_acc += (double)ssim;
                    l1_acc += (double)fabsf(sx[r + kHalo][c + kHalo] - sy[r + kHalo][c + kHalo]);
                }
            }
        }
        __syncthreads();
    }
    // class counts last: the loads behind them overlap the first tile instead of stalling it
    nv = wave_sum_u32(nv);
    ns = wave_sum_u32(ns);
    if ((threadIdx.x & 63) == 0) { cnt[0][threadIdx.x >> 6] = nv; cnt[1][threadIdx.x >> 6] = ns; }
    const double s_tot = block_sum(ssim_acc, red);     // (has the barriers cnt needs)
    const double l_tot = block_sum(l1_acc, red);
    if (threadIdx.x == 0) {
        nv = cnt[0][0] + cnt[0][1] + cnt[0][2] + cnt[0][3];
        ns = cnt[1][0] + cnt[1][1] + cnt[1][2] + cnt[1][3];
        if (kMode != 2) {
            if (nv) atomicAdd(&sums->n_valid, nv);
            if (ns) atomicAdd(&sums->n_sky, ns);
        }
        double* part = partials + (size_t)(blockIdx.y * gridDim.x + blockIdx.x) * kSsimParts;
        part[0] = s_tot; part[1] = l_tot;
    }
}

// SSIM backward + L1: dL/dx(p) = w_rgb / (3 n_valid) *
//   ( 0.8 sign(x - y) valid(p) - 0.2 [ F(M0) + 2 x F(M1) + y F(M2) ](p) ),  F = Gaussian filter.
__global__ __launch_bounds__(256) void ssim_bwd_kernel(int H, int W, Gauss gw, float w_rgb,
                                                       const float* __restrict__ rgb,
                                                       const float* __restrict__ gt_rgb,
                                                       const uint8_t* __restrict__ cls,
                                                       const float* __restrict__ maps,
                                                       const LossSums* __restrict__ sums,
                                                       float* __restrict__ dL_drgb) {
    __shared__ float sm[3][kSIn][kSIn + 1];
    __shared__ float hz[3][kSIn][kST + 1];
    const size_t HW = (size_t)H * W;
    const int tx0 = blockIdx.x * kST, ty0 = blockIdx.y * kST;
    const int c = threadIdx.x & 31, rq = threadIdx.x >> 5;
    const float nv = (float)max(sums->n_valid, 1u);
    const float k_pix = w_rgb / (3.f * nv);
    bool vld[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int gy = ty0 + 4 * rq + i, gx = tx0 + c;   // rows 4 rq .. 4 rq + 3 of column c
        vld[i] = false;
        if (gy < H && gx < W) {
            const size_t p = (size_t)gy * W + gx;
            vld[i] = cls[p] == kPixValid;
        }
    }
    // (tiles fetched one channel ahead, as in ssim_fwd_kernel)
    constexpr int kFetch = (kSIn * kSIn + 255) / 256;
    float fm[3][kFetch];
    auto fetch = [&](int ch) {
#pragma unroll
        for (int k = 0; k < kFetch; ++k) {
            const int idx = k * 256 + threadIdx.x;
            const int r = idx / kSIn, cc = idx - r * kSIn;
            const int gy = ty0 + r - kHalo, gx = tx0 + cc - kHalo;
            const bool in = idx < kSIn * kSIn && gy >= 0 && gy < H && gx >= 0 && gx < W;
            const size_t p = in ? (size_t)gy * W + gx : 0;
#pragma unroll
            for (int m = 0; m < 3; ++m) fm[m][k] = in ? maps[(size_t)(3 * ch + m) * HW + p] : 0.f;
        }
    };
    fetch(0);
    for (int ch = 0; ch < 3; ++ch) {
#pragma unroll
        for (int k = 0; k < kFetch; ++k) {
            const int idx = k * 256 + threadIdx.x;
            if (idx < kSIn * kSIn) {
                const int r = idx / kSIn, cc = idx - r * kSIn;
#pragma unroll
                for (int m = 0; m < 3; ++m) sm[m][r][cc] = fm[m][k];
            }
        }
        __syncthreads();
        if (ch < 2) fetch(ch + 1);
        // both passes as sliding windows: 14 reads feed 4 neighbouring outputs
        for (int idx = threadIdx.x; idx < kSIn * (kST / 4); idx += 256) {
            const int r = idx >> 3, c0 = (idx & 7) * 4;
            float hm[3][4];
#pragma unroll
            for (int m = 0; m < 3; ++m)
#pragma unroll
                for (int i = 0; i < 4; ++i) hm[m][i] = 0.f;
#pragma unroll
            for (int kk = 0; kk < kWin + 3; ++kk) {
                const float v0 = sm[0][r][c0 + kk], v1 = sm[1][r][c0 + kk], v2 = sm[2][r][c0 + kk];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    if (kk - i >= 0 && kk - i < kWin) {
                        const float g = gw.g[kk - i];
                        hm[0][i] += g * v0; hm[1][i] += g * v1; hm[2][i] += g * v2;
                    }
                }
            }
#pragma unroll
            for (int m = 0; m < 3; ++m)
#pragma unroll
                for (int i = 0; i < 4; ++i) hz[m][r][c0 + i] = hm[m][i];
        }
        __syncthreads();
        float vm[3][4];
#pragma unroll
        for (int m = 0; m < 3; ++m)
#pragma unroll
            for (int i = 0; i < 4; ++i) vm[m][i] = 0.f;
#pragma unroll
        for (int kk = 0; kk < kWin + 3; ++kk) {
            const float v0 = hz[0][4 * rq + kk][c], v1 = hz[1][4 * rq + kk][c],
                        v2 = hz[2][4 * rq + kk][c];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                if (kk - i >= 0 && kk - i < kWin) {
                    const float g = gw.g[kk - i];
                    vm[0][i] += g * v0; vm[1][i] += g * v1; vm[2][i] += g * v2;
                }
            }
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int r = 4 * rq + i;
            const int gy = ty0 + r, gx = tx0 + c;
            const float f0 = vm[0][i], f1 = vm[1][i], f2 = vm[2][i];
            if (gy < H && gx < W) {
                const size_t p = (size_t)gy * W + gx;
                const float x = rgb[ch * HW + p], y = gt_rgb[ch * HW + p];
                const float diff = x - y;
                const float sgn = vld[i] ? (float)((diff > 0.f) - (diff < 0.f)) : 0.f;
                dL_drgb[ch * HW + p] =
                    k_pix * (0.8f * sgn - 0.2f * (f0 + 2.f * x * f1 + y * f2));
            }
        }
        __syncthreads();
    }
}

struct GeomArgs {
    float R[9];                       // world-to-camera rotation, row major
    float fu, fv, cu, cv;             // u = row (vertical), v = column
    float w_normal, w_alpha, w_depth, w_dist;
};

__device__ __forceinline__ float depth_of(float a0, float alpha) {
    const float d = a0 / alpha;
    return (d == d && fabsf(d) != INFINITY) ? d : 0.f;   // nan / +-inf -> 0
}

// geometry tile: 32 pixels wide so that a row of the tile is one full 128-byte line of every
// plane (with 16 x 16 tiles every access touched half-used lines)
constexpr int kGW = 32, kGH = 8;

// Depth / normal / alpha / distortion terms -> dL/dallmap (all nine channels written).
__global__ __launch_bounds__(256) void geom_loss_kernel(int H, int W, GeomArgs ga,
                                                        const float* __restrict__ allmap,
                                                        const uint8_t* __restrict__ cls,
                                                        const float* __restrict__ gt_depth,
                                                        const float* __restrict__ gt_cov,
                                                        const LossSums* __restrict__ sums,
                                                        double* __restrict__ partials,
                                                        float* __restrict__ dL_dallmap) {
    __shared__ float dep[kGH + 4][kGW + 5];
    __shared__ float G[kGH + 2][kGW + 2][7];   // dL/d(dx), dL/d(dy) of the pixel's normal (+pad)
    __shared__ double red[4];
    const size_t HW = (size_t)H * W;
    const int tx0 = blockIdx.x * kGW, ty0 = blockIdx.y * kGH;
    const float nv = (float)max(sums->n_valid, 1u), ns = (float)max(sums->n_sky, 1u);
    for (int idx = threadIdx.x; idx < (kGH + 4) * (kGW + 4); idx += 256) {
        const int r = idx / (kGW + 4), cc = idx - r * (kGW + 4);
        const int gy = ty0 + r - 2, gx = tx0 + cc - 2;
        float d = 0.f;
        if (gy >= 0 && gy < H && gx >= 0 && gx < W) {
            const size_t p = (size_t)gy * W + gx;
            d = depth_of(allmap[p], allmap[HW + p]);
        }
        dep[r][cc] = d;
    }
    __syncthreads();
    // camera-space point of tile-local pixel (r, cc) in dep[] coordinates
    const float inv_fv = 1.0f / ga.fv, inv_fu = 1.0f / ga.fu;
    auto point = [&](int r, int cc, float& x, float& y, float& z) {
        const float d = dep[r][cc];
        const float v = (float)(tx0 + cc - 2), u = (float)(ty0 + r - 2);
        x = (v - ga.cv) * inv_fv * d; y = (u - ga.cu) * inv_fu * d; z = d;
    };
    double normal_acc = 0.0;
    for (int idx = threadIdx.x; idx < (kGH + 2) * (kGW + 2); idx += 256) {
        const int r = idx / (kGW + 2), cc = idx - r * (kGW + 2);  // G coordinates
        const int gy = ty0 + r - 1, gx = tx0 + cc - 1;
        float gdx[3] = {0, 0, 0}, gdy[3] = {0, 0, 0};
        const bool in = gy >= 0 && gy < H && gx >= 0 && gx < W;
        const bool owned = r >= 1 && r <= kGH && cc >= 1 && cc <= kGW;
        if (in) {
            const size_t p = (size_t)gy * W + gx;
            const bool valid = cls[p] == kPixValid;
            const bool interior = gy >= 1 && gy <= H - 2 && gx >= 1 && gx <= W - 2;
            float nh[3] = {0, 0, 0};  // surface normal (camera frame); zero on the border
            float dxv[3], dyv[3], inv_len = 0.f;
            if (interior) {
                float ax, ay, az, bx, by, bz;
                point(r + 1, cc + 2, ax, ay, az); point(r + 1, cc, bx, by, bz);  // right - left
                dxv[0] = ax - bx; dxv[1] = ay - by; dxv[2] = az - bz;
                point(r, cc + 1, ax, ay, az); point(r + 2, cc + 1, bx, by, bz);  // up - down
                dyv[0] = ax - bx; dyv[1] = ay - by; dyv[2] = az - bz;
                const float c0 = dxv[1] * dyv[2] - dxv[2] * dyv[1];
                const float c1 = dxv[2] * dyv[0] - dxv[0] * dyv[2];
                const float c2 = dxv[0] * dyv[1] - dxv[1] * dyv[0];
                const float len = sqrtf(c0 * c0 + c1 * c1 + c2 * c2);
                inv_len = 1.0f / fmaxf(len, 1e-12f);
                nh[0] = c0 * inv_len; nh[1] = c1 * inv_len; nh[2] = c2 * inv_len;
            }
            // rendered normal taken to the world frame and back: t = R (R^T n) as the
            // reference multiplies both normals by the rotation before the dot product
            const float n0 = allmap[2 * HW + p], n1 = allmap[3 * HW + p], n2 = allmap[4 * HW + p];
            float aw[3], t[3], sw[3], rs[3];
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                aw[j] = n0 * ga.R[j] + n1 * ga.R[3 + j] + n2 * ga.R[6 + j];
                sw[j] = nh[0] * ga.R[j] + nh[1] * ga.R[3 + j] + nh[2] * ga.R[6 + j];
            }
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                t[i] = ga.R[3 * i] * aw[0] + ga.R[3 * i + 1] * aw[1] + ga.R[3 * i + 2] * aw[2];
                rs[i] = ga.R[3 * i] * sw[0] + ga.R[3 * i + 1] * sw[1] + ga.R[3 * i + 2] * sw[2];
            }
            const float kq = valid ? ga.w_normal / nv : 0.f;
            if (owned) {
                if (valid) normal_acc += 1.0 - (double)(aw[0] * sw[0] + aw[1] * sw[1] + aw[2] * sw[2]);
                // d(1 - a.s)/dn = -R s: the rendered-normal channels of this pixel
                dL_dallmap[2 * HW + p] = -kq * rs[0];
                dL_dallmap[3 * HW + p] = -kq * rs[1];
                dL_dallmap[4 * HW + p] = -kq * rs[2];
            }
            if (interior && kq != 0.f) {
                // g = dL/dn_hat = -kq t ; through the normalisation, then the cross product
                const float g0 = -kq * t[0], g1 = -kq * t[1], g2 = -kq * t[2];
                const float ng = nh[0] * g0 + nh[1] * g1 + nh[2] * g2;
                const float dc0 = (g0 - nh[0] * ng) * inv_len, dc1 = (g1 - nh[1] * ng) * inv_len,
                            dc2 = (g2 - nh[2] * ng) * inv_len;
                gdx[0] = dyv[1] * dc2 - dyv[2] * dc1;   // dy x dc
                gdx[1] = dyv[2] * dc0 - dyv[0] * dc2;
                gdx[2] = dyv[0] * dc1 - dyv[1] * dc0;
                gdy[0] = dc1 * dxv[2] - dc2 * dxv[1];   // dc x dx
                gdy[1] = dc2 * dxv[0] - dc0 * dxv[2];
                gdy[2] = dc0 * dxv[1] - dc1 * dxv[0];
            }
        }
#pragma unroll
        for (int k = 0; k < 3; ++k) { G[r][cc][k] = gdx[k]; G[r][cc][3 + k] = gdy[k]; }
    }
    __syncthreads();
    // owned pixel of this lane
    const int lr = threadIdx.x >> 5, lc = threadIdx.x & 31;
    const int gy = ty0 + lr, gx = tx0 + lc;
    double alpha_acc = 0.0, depth_acc = 0.0, dist_acc = 0.0;
    if (gy < H && gx < W) {
        const size_t p = (size_t)gy * W + gx;
        const int r = lr + 1, cc = lc + 1;  // G coordinates
        const bool sky = cls[p] == kPixSky;
        const bool valid = cls[p] == kPixValid;
        // P(p) enters dx of the left neighbour (+), of the right neighbour (-),
        // dy of the lower neighbour (+) and of the upper neighbour (-)
        float dP[3];
#pragma unroll
        for (int k = 0; k < 3; ++k)
            dP[k] = G[r][cc - 1][k] - G[r][cc + 1][k] + G[r + 1][cc][3 + k] - G[r - 1][cc][3 + k];
        const float rx = ((float)gx - ga.cv) / ga.fv, ry = ((float)gy - ga.cu) / ga.fu;
        float dL_dd = rx * dP[0] + ry * dP[1] + dP[2];
        const float a0 = allmap[p], alpha = allmap[HW + p];
        const float q = a0 / alpha;
        const bool finite = (q == q) && fabsf(q) != INFINITY;
        const float d = finite ? q : 0.f;
        if (valid) {
            const float diff = d - gt_depth[p];
            const float wgt = 1.0f / gt_cov[p];
            depth_acc = (double)(fabsf(diff) * wgt);
            dL_dd += ga.w_depth / nv * wgt * (float)((diff > 0.f) - (diff < 0.f));
            dist_acc = (double)allmap[6 * HW + p];
        }
        if (sky) alpha_acc = (double)alpha;
        float dA0 = 0.f, dAlpha = sky ? ga.w_alpha / ns : 0.f;
        if (finite) {
            dA0 = dL_dd / alpha;
            dAlpha += -dL_dd * d / alpha;
        }
        dL_dallmap[p] = dA0;
        dL_dallmap[HW + p] = dAlpha;
        dL_dallmap[5 * HW + p] = 0.f;
        dL_dallmap[6 * HW + p] = valid ? ga.w_dist / nv : 0.f;
        dL_dallmap[7 * HW + p] = 0.f;
        dL_dallmap[8 * HW + p] = 0.f;
    }
    const double n_tot = block_sum(normal_acc, red);
    const double a_tot = block_sum(alpha_acc, red);
    const double d_tot = block_sum(depth_acc, red);
    const double s_tot = block_sum(dist_acc, red);
    if (threadIdx.x == 0) {
        double* part = partials + (size_t)(blockIdx.y * gridDim.x + blockIdx.x) * kGeomParts;
        part[0] = n_tot; part[1] = a_tot; part[2] = d_tot; part[3] = s_tot;
    }
}

// out[0..7] = total, rgb, normal, alpha, depth, dist loss terms (unweighted), n_valid, n_sky
__global__ __launch_bounds__(1024) void loss_final_kernel(
    const LossSums* __restrict__ sums, const double* __restrict__ ssim_parts, int n_ssim,
    const double* __restrict__ geom_parts, int n_geom, float w_rgb, float w_normal, float w_alpha,
    float w_depth, float w_dist, float* __restrict__ out) {
    // one workgroup of 1024 lanes: the kernel is a latency chain of strided loads, so the
    // wider the group the shorter the chain
    __shared__ double red[kSsimParts + kGeomParts][16];
    double acc[kSsimParts + kGeomParts] = {0, 0, 0, 0, 0, 0};
    for (int i = threadIdx.x; i < n_ssim; i += 1024)
        for (int k = 0; k < kSsimParts; ++k) acc[k] += ssim_parts[(size_t)i * kSsimParts + k];
    for (int i = threadIdx.x; i < n_geom; i += 1024)
        for (int k = 0; k < kGeomParts; ++k)
            acc[kSsimParts + k] += geom_parts[(size_t)i * kGeomParts + k];
    for (int k = 0; k < kSsimParts + kGeomParts; ++k) {
        double v = acc[k];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
        if ((threadIdx.x & 63) == 0) red[k][threadIdx.x >> 6] = v;
    }
    __syncthreads();
    if (threadIdx.x != 0) return;
    double tot[kSsimParts + kGeomParts];
    for (int k = 0; k < kSsimParts + kGeomParts; ++k) {
        double v = 0.0;
        for (int w = 0; w < 16; ++w) v += red[k][w];
        tot[k] = v;
    }
    const double nv = (double)max(sums->n_valid, 1u), ns = (double)max(sums->n_sky, 1u);
    const double ssim = tot[0] / (3.0 * nv), l1 = tot[1] / (3.0 * nv);
    const double rgb = 0.8 * l1 + 0.2 * (1.0 - ssim);
    const double normal = tot[2] / nv, alpha = tot[3] / ns, depth = tot[4] / nv, dist = tot[5] / nv;
    out[0] = (float)(w_rgb * rgb + w_normal * normal + w_alpha * alpha + w_depth * depth +
                     w_dist * dist);
    out[1] = (float)rgb; out[2] = (float)normal; out[3] = (float)alpha; out[4] = (float)depth;
    out[5] = (float)dist; out[6] = (float)sums->n_valid; out[7] = (float)sums->n_sky;
}

constexpr size_t kSumsBytes = 256;  // LossSums padded so the maps stay 256-byte aligned

}  // namespace vings

using namespace vings;

// workspace = [LossSums | ssim partials | geom partials | 9 derivative maps | pixel classes]
static size_t partial_bytes(int H, int W) {
    const size_t n = (size_t)cdiv(W, kST) * cdiv(H, kST) * kSsimParts +
                     (size_t)cdiv(W, kGW) * cdiv(H, kGH) * kGeomParts;
    return (n * sizeof(double) + 255) / 256 * 256;
}

extern "C" size_t vings_mapping_loss_workspace_bytes(int H, int W) {
    return kSumsBytes + partial_bytes(H, W) + (sizeof(float) * 9 + 1) * (size_t)H * (size_t)W;
}

// gt cache = [6 moment planes | LossSums (256 B) | pixel classes]
extern "C" size_t vings_mapping_loss_gt_cache_bytes(int H, int W) {
    return sizeof(float) * 6 * (size_t)H * (size_t)W + kSumsBytes + (size_t)H * (size_t)W;
}

static int mapping_loss(int H, int W, const float* rgb, const float* allmap, const float* gt_rgb,
                        const float* gt_depth, const float* gt_depth_cov, const float* Rw2c,
                        float fu, float fv, float cu, float cv, const float* weights,
                        float* dL_drgb, float* dL_dallmap, float* loss_out, void* workspace,
                        size_t workspace_bytes, void* gt_cache, int gt_cache_valid, void* stream) {
    VINGS_REQUIRE(H >= 3 && W >= 3, "image must be at least 3x3");
    VINGS_REQUIRE(rgb && allmap && gt_rgb && gt_depth && gt_depth_cov && Rw2c && weights,
                  "null input");
    VINGS_REQUIRE(dL_drgb && dL_dallmap && loss_out && workspace, "null output");
    VINGS_REQUIRE(workspace_bytes >= vings_mapping_loss_workspace_bytes(H, W), "workspace too small");
    VINGS_REQUIRE(((uintptr_t)workspace & 7) == 0, "workspace must be 8-byte aligned");
    VINGS_REQUIRE(fu != 0.f && fv != 0.f, "focal lengths");
    hipStream_t s = (hipStream_t)stream;
    LossSums* sums = reinterpret_cast<LossSums*>(workspace);
    const int n_ssim = cdiv(W, kST) * cdiv(H, kST), n_geom = cdiv(W, kGW) * cdiv(H, kGH);
    double* ssim_parts = reinterpret_cast<double*>(reinterpret_cast<char*>(workspace) + kSumsBytes);
    double* geom_parts = ssim_parts + (size_t)n_ssim * kSsimParts;
    float* maps = reinterpret_cast<float*>(reinterpret_cast<char*>(workspace) + kSumsBytes +
                                           partial_bytes(H, W));
    const size_t HW = (size_t)H * W;
    uint8_t* cls = reinterpret_cast<uint8_t*>(maps + 9 * HW);
    float* gt_moments = nullptr;
    if (gt_cache) {
        VINGS_REQUIRE(((uintptr_t)gt_cache & 7) == 0, "gt cache must be 8-byte aligned");
        gt_moments = reinterpret_cast<float*>(gt_cache);
        sums = reinterpret_cast<LossSums*>(gt_moments + 6 * HW);
        cls = reinterpret_cast<uint8_t*>(sums) + kSumsBytes;
    }
    const int mode = !gt_cache ? 0 : (gt_cache_valid ? 2 : 1);
    // the Gaussian window of loss_utils.py:26-28 (sigma 1.5, normalised), in float like there
    Gauss gw;
    {
        float sum = 0.f;
        for (int k = 0; k < kWin; ++k) {
            gw.g[k] = (float)exp(-(double)((k - kWin / 2) * (k - kWin / 2)) / (2.0 * 1.5 * 1.5));
            sum += gw.g[k];
        }
        for (int k = 0; k < kWin; ++k) gw.g[k] /= sum;
    }
    GeomArgs ga;
    for (int k = 0; k < 9; ++k) ga.R[k] = Rw2c[k];
    ga.fu = fu; ga.fv = fv; ga.cu = cu; ga.cv = cv;
    const float w_rgb = weights[0], w_depth = weights[1], w_alpha = weights[2],
                w_normal = weights[3], w_dist = weights[4];
    ga.w_normal = w_normal; ga.w_alpha = w_alpha; ga.w_depth = w_depth; ga.w_dist = w_dist;

    ProfScope prof(PROF_LOSS, s);
    if (mode != 2) VINGS_CHECK_HIP(hipMemsetAsync(sums, 0, sizeof(LossSums), s));
    const dim3 sgrid(cdiv(W, kST), cdiv(H, kST));
    if (mode == 0)
        ssim_fwd_kernel<0><<<sgrid, 256, 0, s>>>(H, W, gw, rgb, gt_rgb, gt_depth, cls, sums, maps,
                                                 ssim_parts, nullptr);
    else if (mode == 1)
        ssim_fwd_kernel<1><<<sgrid, 256, 0, s>>>(H, W, gw, rgb, gt_rgb, gt_depth, cls, sums, maps,
                                                 ssim_parts, gt_moments);
    else
        ssim_fwd_kernel<2><<<sgrid, 256, 0, s>>>(H, W, gw, rgb, gt_rgb, gt_depth, cls, sums, maps,
                                                 ssim_parts, gt_moments);
    VINGS_LAUNCH_CHECK();
    ssim_bwd_kernel<<<sgrid, 256, 0, s>>>(H, W, gw, w_rgb, rgb, gt_rgb, cls, maps, sums,
                                          dL_drgb);
    VINGS_LAUNCH_CHECK();
    const dim3 ggrid(cdiv(W, kGW), cdiv(H, kGH));
    geom_loss_kernel<<<ggrid, 256, 0, s>>>(H, W, ga, allmap, cls, gt_depth, gt_depth_cov, sums,
                                           geom_parts, dL_dallmap);
    VINGS_LAUNCH_CHECK();
    loss_final_kernel<<<1, 1024, 0, s>>>(sums, ssim_parts, n_ssim, geom_parts, n_geom, w_rgb,
                                        w_normal, w_alpha, w_depth, w_dist, loss_out);
    VINGS_LAUNCH_CHECK();
    return VINGS_OK;
}

extern "C" int vings_mapping_loss(int H, int W, const float* rgb, const float* allmap,
                                  const float* gt_rgb, const float* gt_depth,
                                  const float* gt_depth_cov, const float* Rw2c, float fu, float fv,
                                  float cu, float cv, const float* weights, float* dL_drgb,
                                  float* dL_dallmap, float* loss_out, void* workspace,
                                  size_t workspace_bytes, void* stream) {
    return mapping_loss(H, W, rgb, allmap, gt_rgb, gt_depth, gt_depth_cov, Rw2c, fu, fv, cu, cv,
                        weights, dL_drgb, dL_dallmap, loss_out, workspace, workspace_bytes, nullptr,
                        0, stream);
}

extern "C" int vings_mapping_loss_cached(int H, int W, const float* rgb, const float* allmap,
                                         const float* gt_rgb, const float* gt_depth,
                                         const float* gt_depth_cov, const float* Rw2c, float fu,
                                         float fv, float cu, float cv, const float* weights,
                                         float* dL_drgb, float* dL_dallmap, float* loss_out,
                                         void* workspace, size_t workspace_bytes, void* gt_cache,
                                         int gt_cache_valid, void* stream) {
    VINGS_REQUIRE(gt_cache != nullptr, "gt_cache");
    return mapping_loss(H, W, rgb, allmap, gt_rgb, gt_depth, gt_depth_cov, Rw2c, fu, fv, cu, cv,
                        weights, dL_drgb, dL_dallmap, loss_out, workspace, workspace_bytes, gt_cache,
                        gt_cache_valid, stream);
}
