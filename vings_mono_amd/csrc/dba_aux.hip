// Tracker-side helpers of the dense-BA path on gfx950: frame distance and the
// correlation-volume window lookup (forward and backward).
//
// Serves droid_backends.frame_distance
// (/root/reference/scripts/frontend/depth_video.py:289-299) and
// droid_backends.corr_index_forward/backward
// (/root/reference/scripts/frontend/modules/corr.py:12,19,40-50).
#include <hip/hip_fp16.h>

#include "common.hpp"

namespace vings {

constexpr float kMinDepthAux = 0.2f;

__device__ __forceinline__ void qrotf(const float* q, const float* v, float* o) {
    const float x = q[0], y = q[1], z = q[2], w = q[3];
    const float tx = 2 * (y * v[2] - z * v[1]), ty = 2 * (z * v[0] - x * v[2]), tz = 2 * (x * v[1] - y * v[0]);
    o[0] = v[0] + w * tx + (y * tz - z * ty);
    o[1] = v[1] + w * ty + (z * tx - x * tz);
    o[2] = v[2] + w * tz + (x * ty - y * tx);
}

// One workgroup per candidate edge; lanes stride over the pixel grid.
__global__ __launch_bounds__(256) void frame_distance_kernel(
    const float* __restrict__ poses, const float* __restrict__ disps,
    const float* __restrict__ intr, const int64_t* __restrict__ ii,
    const int64_t* __restrict__ jj, int B, int ht, int wd, float beta, float* __restrict__ out) {
    __shared__ float snum[4], sden[4];
    const int e = blockIdx.x;
    const long i = ii[e], j = jj[e];
    const int HW = ht * wd;
    if (i < 0 || i >= B || j < 0 || j >= B) { if (threadIdx.x == 0) out[e] = 1000.f; return; }
    float t[3], q[4];
    if (i == j) {
        t[0] = -0.1f; t[1] = t[2] = 0.f; q[0] = q[1] = q[2] = 0.f; q[3] = 1.f;
    } else {
        const float* pi = poses + 7 * i;
        const float* pj = poses + 7 * j;
        const float a[4] = {pj[3], pj[4], pj[5], pj[6]};
        const float b[4] = {-pi[3], -pi[4], -pi[5], pi[6]};
        q[0] = a[3] * b[0] + a[0] * b[3] + a[1] * b[2] - a[2] * b[1];
        q[1] = a[3] * b[1] - a[0] * b[2] + a[1] * b[3] + a[2] * b[0];
        q[2] = a[3] * b[2] + a[0] * b[1] - a[1] * b[0] + a[2] * b[3];
        q[3] = a[3] * b[3] - a[0] * b[0] - a[1] * b[1] - a[2] * b[2];
        float rt[3];
        const float ti[3] = {pi[0], pi[1], pi[2]};
        qrotf(q, ti, rt);
        t[0] = pj[0] - rt[0]; t[1] = pj[1] - rt[1]; t[2] = pj[2] - rt[2];
    }
    const float fx = intr[0], fy = intr[1], cx = intr[2], cy = intr[3];
    float num = 0.f, den = 0.f;
    for (int p = threadIdx.x; p < HW; p += 256) {
        const int u = p % wd, v = p / wd;
        const float d = disps[(size_t)i * HW + p];
        const float X0[3] = {((float)u - cx) / fx, ((float)v - cy) / fy, 1.f};
        float X1[3];
        qrotf(q, X0, X1);
        X1[0] += t[0] * d; X1[1] += t[1] * d; X1[2] += t[2] * d;
        if (X1[2] > kMinDepthAux) {
            const float du = fx * X1[0] / X1[2] + cx - (float)u, dv = fy * X1[1] / X1[2] + cy - (float)v;
            num += beta * sqrtf(du * du + dv * dv);
            den += beta;
        }
        const float Y[3] = {X0[0] + t[0] * d, X0[1] + t[1] * d, X0[2] + t[2] * d};
        if (Y[2] > kMinDepthAux) {
            const float du = fx * Y[0] / Y[2] + cx - (float)u, dv = fy * Y[1] / Y[2] + cy - (float)v;
            num += (1.f - beta) * sqrtf(du * du + dv * dv);
            den += (1.f - beta);
        }
    }
    num = wave_sum(num);
    den = wave_sum(den);
    if ((threadIdx.x & 63) == 0) { snum[threadIdx.x >> 6] = num; sden[threadIdx.x >> 6] = den; }
    __syncthreads();
    if (threadIdx.x == 0) {
        const float n = snum[0] + snum[1] + snum[2] + snum[3];
        const float dd = sden[0] + sden[1] + sden[2] + sden[3];
        out[e] = (dd / (float)HW < 0.75f) ? 1000.f : n / dd;
    }
}

__device__ __forceinline__ float to_f(float v) { return v; }
__device__ __forceinline__ float to_f(__half v) { return __half2float(v); }
__device__ __forceinline__ void from_f(float& o, float v) { o = v; }
__device__ __forceinline__ void from_f(__half& o, float v) { o = __float2half(v); }

// One lane per source pixel (e,y,x): it owns the h2 x w2 slice volume[e,y,x].
template <typename T>
__global__ __launch_bounds__(256) void corr_index_fwd_kernel(
    const T* __restrict__ volume, const float* __restrict__ coords, T* __restrict__ corr, int E,
    int h1, int w1, int h2, int w2, int r) {
    const int64_t n = (int64_t)E * h1 * w1;
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= n) return;
    const int x = (int)(idx % w1), y = (int)((idx / w1) % h1), e = (int)(idx / ((int64_t)w1 * h1));
    const size_t plane = (size_t)h1 * w1;
    const float cxv = coords[((size_t)e * 2 + 0) * plane + (size_t)y * w1 + x];
    const float cyv = coords[((size_t)e * 2 + 1) * plane + (size_t)y * w1 + x];
    const float fx0 = floorf(cxv), fy0 = floorf(cyv);
    const float dx = cxv - fx0, dy = cyv - fy0;
    const int x0 = (int)fx0, y0 = (int)fy0;
    const T* slice = volume + (size_t)idx * h2 * w2;
    const int rd = 2 * r + 1;
    auto fetch = [&](int yi, int xi) -> float {
        return (yi >= 0 && yi < h2 && xi >= 0 && xi < w2) ? to_f(slice[(size_t)yi * w2 + xi]) : 0.f;
    };
    for (int i = 0; i < rd; ++i) {
        const int yi = y0 + i - r;
        float left0 = fetch(yi, x0 - r), left1 = fetch(yi + 1, x0 - r);
        for (int j = 0; j < rd; ++j) {
            const int xi = x0 + j - r;
            const float right0 = fetch(yi, xi + 1), right1 = fetch(yi + 1, xi + 1);
            const float v = (1.f - dx) * (1.f - dy) * left0 + dx * (1.f - dy) * right0 +
                            (1.f - dx) * dy * left1 + dx * dy * right1;
            from_f(corr[(((size_t)e * rd + j) * rd + i) * plane + (size_t)y * w1 + x], v);   // x offset: the slow index
            left0 = right0; left1 = right1;
        }
    }
}

// The same lookup for r = 3, with a wave instead of a lane per source pixel while it READS:
// a lane that owns a pixel walks an 8 x 8 patch of its own h2 x w2 slice, so every load of the
// kernel above touches 64 cache lines for 64 two- or four-byte values.  Here the wave takes 64
// consecutive pixels one after the other, its 64 lanes fetch the 8 x 8 patch of that pixel in
// one load (8 segments of a row each), form the 7 x 7 bilinear values from the lane's right
// and lower neighbours and park them in LDS; at the end lane p writes the 49 values of pixel
// p, consecutive lanes to consecutive addresses as before.
template <typename T>
__global__ __launch_bounds__(256) void corr_index_fwd_r3_kernel(
    const T* __restrict__ volume, const float* __restrict__ coords, T* __restrict__ corr, int E,
    int h1, int w1, int h2, int w2) {
    constexpr int r = 3, rd = 7;
    __shared__ T out_s[4][rd * rd][65];           // [wave][output][pixel], padded against conflicts
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int plane = h1 * w1;
    const int groups = (plane + 63) / 64;
    const int gid = blockIdx.x * 4 + wv;
    if (gid >= E * groups) return;                // whole waves only; no workgroup barrier below
    const int e = gid / groups, yx0 = (gid - e * groups) * 64;
    const int yx = yx0 + lane;
    const bool live = yx < plane;
    const float cxv = live ? coords[((size_t)e * 2 + 0) * plane + yx] : 0.f;
    const float cyv = live ? coords[((size_t)e * 2 + 1) * plane + yx] : 0.f;
    const float fx0 = floorf(cxv), fy0 = floorf(cyv);
    const float dx = cxv - fx0, dy = cyv - fy0;
    const int x0 = (int)fx0, y0 = (int)fy0;
    const int pi = lane >> 3, pj = lane & 7;      // this lane's element of a pixel's patch
    const int npix = min(64, plane - yx0);
    const size_t slice_sz = (size_t)h2 * w2;
    const T* base = volume + ((size_t)e * plane + yx0) * slice_sz;
    // eight pixels per round, so that eight patch loads are in flight together
    for (int p0 = 0; p0 < npix; p0 += 8) {
        float v[8], dxs[8], dys[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int p = p0 + q;                  // < 64: readlane stays inside the wave
            const int x0p = __builtin_amdgcn_readlane(x0, p), y0p = __builtin_amdgcn_readlane(y0, p);
            dxs[q] = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, dx), p));
            dys[q] = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, dy), p));
            const int yi = y0p - r + pi, xi = x0p - r + pj;
            v[q] = (p < npix && yi >= 0 && yi < h2 && xi >= 0 && xi < w2)
                       ? to_f(base[(size_t)p * slice_sz + (size_t)yi * w2 + xi]) : 0.f;
        }
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const float right = __shfl_down(v[q], 1, 64);
            const float down = __shfl_down(v[q], 8, 64);
            const float dright = __shfl_down(right, 8, 64);
            const float dxp = dxs[q], dyp = dys[q];
            const float o = (1.f - dxp) * (1.f - dyp) * v[q] + dxp * (1.f - dyp) * right +
                            (1.f - dxp) * dyp * down + dxp * dyp * dright;
            if (pi < rd && pj < rd && p0 + q < npix) from_f(out_s[wv][pj * rd + pi][p0 + q], o);
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    if (live) {
#pragma unroll 7
        for (int k = 0; k < rd * rd; ++k)
            corr[((size_t)e * rd * rd + k) * plane + yx] = out_s[wv][k][lane];
    }
}

template <typename T>
__global__ __launch_bounds__(256) void corr_index_bwd_kernel(
    const float* __restrict__ coords, const T* __restrict__ corr_grad, T* __restrict__ volume_grad,
    int E, int h1, int w1, int h2, int w2, int r) {
    const int64_t n = (int64_t)E * h1 * w1;
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= n) return;
    const int x = (int)(idx % w1), y = (int)((idx / w1) % h1), e = (int)(idx / ((int64_t)w1 * h1));
    const size_t plane = (size_t)h1 * w1;
    const float cxv = coords[((size_t)e * 2 + 0) * plane + (size_t)y * w1 + x];
    const float cyv = coords[((size_t)e * 2 + 1) * plane + (size_t)y * w1 + x];
    const float fx0 = floorf(cxv), fy0 = floorf(cyv);
    const float dx = cxv - fx0, dy = cyv - fy0;
    const int x0 = (int)fx0, y0 = (int)fy0;
    T* slice = volume_grad + (size_t)idx * h2 * w2;
    const int rd = 2 * r + 1;
    auto push = [&](int yi, int xi, float v) {
        if (yi >= 0 && yi < h2 && xi >= 0 && xi < w2) {
            T& dst = slice[(size_t)yi * w2 + xi];
            from_f(dst, to_f(dst) + v);
        }
    };
    for (int i = 0; i < rd; ++i)
        for (int j = 0; j < rd; ++j) {
            const float g = to_f(corr_grad[(((size_t)e * rd + j) * rd + i) * plane + (size_t)y * w1 + x]);
            const int yi = y0 + i - r, xi = x0 + j - r;
            push(yi, xi, (1.f - dx) * (1.f - dy) * g);
            push(yi, xi + 1, dx * (1.f - dy) * g);
            push(yi + 1, xi, (1.f - dx) * dy * g);
            push(yi + 1, xi + 1, dx * dy * g);
        }
}

// ---------------------------------------------------------------------------
// droid_backends.altcorr_forward / altcorr_backward
// (/root/reference/scripts/frontend/modules/corr.py:79,86-87,106-125): the correlation window
// without the all-pairs volume.  corr[b,s,i*(2r+1)+j,h,w] is the bilinear sample at
// (x + j - r, y + i - r), (x, y) = coords[b,s,h,w,:], of the map
// (yy, xx) -> <fmap1[b,h,w,:], fmap2[b,yy,xx,:]>, zero outside fmap2: exactly what
// corr_index_forward returns for volume = fmap1 . fmap2^T, but at O(H W C) memory.  At 1080p / 8
// the volume of one edge would be 32 400^2 entries; this is the form a tracker grid of that
// size has to use.
// ---------------------------------------------------------------------------
constexpr int kAltPx = 32;   // consecutive pixels of one image row per workgroup

// One wave per pixel, one lane per integer position of the (2r+2)^2 neighbourhood: the lane
// walks its fmap2 row against the pixel's fmap1 row (broadcast from LDS).  The 32 pixels of a
// workgroup leave their windows in LDS and are stored with the pixel index fastest, so every
// store instruction covers full lines of one output plane.
__global__ __launch_bounds__(256) void altcorr_fwd_kernel(
    const float* __restrict__ fmap1, const float* __restrict__ fmap2,
    const float* __restrict__ coords, float* __restrict__ corr, int B, int S, int H, int W, int H2,
    int W2, int C, int r) {
    extern __shared__ float alt_lds[];
    const int n1 = 2 * r + 2, P = n1 * n1, rd = 2 * r + 1, K = rd * rd;
    float* f1 = alt_lds;                       // [kAltPx][C]
    float* dots = f1 + kAltPx * C;             // [4 waves][P]
    float* tile = dots + 4 * P;                // [K][kAltPx]
    const int chunks = (W + kAltPx - 1) / kAltPx;
    int blk = blockIdx.x;
    const int wc = blk % chunks; blk /= chunks;
    const int h = blk % H; blk /= H;
    const int s = blk % S;
    const int b = blk / S;
    const int w0 = wc * kAltPx, npx = min(kAltPx, W - w0);
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    // fmap1 rows of the chunk: contiguous in memory
    {
        const float* src = fmap1 + (((size_t)b * H + h) * W + w0) * C;
        for (int k = threadIdx.x; k < npx * C; k += 256) f1[k] = src[k];
    }
    __syncthreads();
    const int C4 = C >> 2;
    for (int p = wv; p < npx; p += 4) {
        const float* cp = coords + ((((size_t)b * S + s) * H + h) * W + (w0 + p)) * 2;
        const float x = cp[0], y = cp[1];
        // coordinates far outside (or not finite) see nothing; the bound keeps the casts defined
        const bool sane = fabsf(x) < 1e6f && fabsf(y) < 1e6f;
        const float xf = sane ? floorf(x) : -1e6f, yf = sane ? floorf(y) : -1e6f;
        const float dx = sane ? x - xf : 0.f, dy = sane ? y - yf : 0.f;
        const int x0 = (int)xf - r, y0 = (int)yf - r;
        const float4* a4 = reinterpret_cast<const float4*>(f1 + p * C);
        for (int q = lane; q < P; q += 64) {
            const int iy = q / n1, ix = q - iy * n1;
            const int yy = y0 + iy, xx = x0 + ix;
            float acc = 0.f;
            if (yy >= 0 && yy < H2 && xx >= 0 && xx < W2) {
                const float4* row = reinterpret_cast<const float4*>(
                    fmap2 + (((size_t)b * H2 + yy) * W2 + xx) * C);
                float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
#pragma unroll 8
                for (int c = 0; c < C4; ++c) {
                    const float4 u = row[c], v = a4[c];
                    a0 += u.x * v.x; a1 += u.y * v.y; a2 += u.z * v.z; a3 += u.w * v.w;
                }
                acc = (a0 + a1) + (a2 + a3);
            }
            dots[wv * P + q] = acc;
        }
        // same wave wrote and reads: LDS operations of a wave complete in order
        __builtin_amdgcn_wave_barrier();
        const float w00 = (1.f - dx) * (1.f - dy), w01 = dx * (1.f - dy), w10 = (1.f - dx) * dy,
                    w11 = dx * dy;
        const float* d = dots + wv * P;
        for (int k = lane; k < K; k += 64) {
            const int j = k / rd, i = k - j * rd;     // channel = x offset * rd + y offset
            const float* e = d + i * n1 + j;
            tile[k * kAltPx + p] = w00 * e[0] + w01 * e[1] + w10 * e[n1] + w11 * e[n1 + 1];
        }
    }
    __syncthreads();
    float* out = corr + ((((size_t)b * S + s) * K) * H + h) * W + w0;
    for (int k = threadIdx.x; k < K * kAltPx; k += 256) {
        const int kk = k / kAltPx, p = k - kk * kAltPx;
        if (p < npx) out[(size_t)kk * H * W + p] = tile[k];
    }
}

// Tiled form (r <= 3, C a multiple of 16): a workgroup takes a 16x4 block of query pixels of one
// (b, s), a wave one row of it.  The wave handles four pixels at a time, 16 lanes each, a lane a
// 2x2 patch of integer positions of the pixel's (2r+2)^2 neighbourhood (16 accumulators: four
// steps x four positions).  The targets of neighbouring pixels land close to each other (a
// tracker's flow field is smooth), so their neighbourhoods overlap: the workgroup takes the
// bounding rectangle of all of them inside fmap2 and, when it holds at most kAltCap positions,
// passes it through LDS in chunks of 16 channels -- every fmap2 row is then fetched once per
// workgroup with full-line loads instead of once per pixel by a lane of its own.  Scattered
// targets (rectangle too large) read the rows from global memory as before.
constexpr int kAltTW = 16, kAltTH = 4, kAltTile = kAltTW * kAltTH;
constexpr int kAltCap = 512;      // fmap2 positions a workgroup stages
constexpr int kAltCh = 16;        // channels per staged chunk
constexpr int kAltStride = 20;    // floats per staged position: 16 + 4 (16-byte aligned, and the
                                  // lanes of a 16-lane group fall into different banks)
constexpr int kAltStageIters = kAltCap * (kAltCh / 4) / 256;
constexpr size_t kAltTileLds =
    sizeof(float) * (kAltCap * kAltStride + kAltTile * kAltCh + 4 * 4 * 64 + 4 * kAltTile + 4);

// One step = four pixels of the wave's row (16 lanes each), a lane a 2x2 patch of positions:
// the pixel's 16 fmap1 values are read once for four dot products (20 LDS reads of 16 bytes per
// 64 multiply-adds of a lane instead of 32; the loop is bound by exactly these reads).
template <bool kStaged>
__device__ __forceinline__ void altcorr_row_dots(float (&acc)[kAltTW], const int (&off)[kAltTW],
                                                 const float* __restrict__ f1lane,
                                                 const float* __restrict__ src) {
#pragma unroll
    for (int st = 0; st < kAltTW / 4; ++st) {
        const float4* a4 = reinterpret_cast<const float4*>(f1lane + st * 4 * kAltCh);
        float4 v[kAltCh / 4];
#pragma unroll
        for (int c = 0; c < kAltCh / 4; ++c) v[c] = a4[c];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int o = off[st * 4 + e];
            const float4* u4 = reinterpret_cast<const float4*>(src + (o < 0 ? 0 : o));
            float d = 0.f;
#pragma unroll
            for (int c = 0; c < kAltCh / 4; ++c) {
                const float4 u = u4[c];
                d += u.x * v[c].x; d += u.y * v[c].y; d += u.z * v[c].z; d += u.w * v[c].w;
            }
            acc[st * 4 + e] += o < 0 ? 0.f : d;
        }
    }
}

__global__ __launch_bounds__(256) void altcorr_fwd_tile_kernel(
    const float* __restrict__ fmap1, const float* __restrict__ fmap2,
    const float* __restrict__ coords, float* __restrict__ corr, int B, int S, int H, int W, int H2,
    int W2, int C, int r) {
    extern __shared__ float alt_lds[];
    float* reg = alt_lds;                                  // [kAltCap][kAltStride]; at the end [K][64]
    float* f1s = reg + kAltCap * kAltStride;               // [64 pixels][16 channels]
    float* dots = f1s + kAltTile * kAltCh;                 // [4 waves][4 pixels][64]
    int* ix0 = reinterpret_cast<int*>(dots + 4 * 4 * 64);  // [64] window origin of every pixel
    int* iy0 = ix0 + kAltTile;
    float* fdx = reinterpret_cast<float*>(iy0 + kAltTile);
    float* fdy = fdx + kAltTile;
    int* red = reinterpret_cast<int*>(fdy + kAltTile);     // rectangle: x0, y0, width, height
    const int n1 = 2 * r + 2, rd = 2 * r + 1, K = rd * rd;
    const int tiles_x = (W + kAltTW - 1) / kAltTW, tiles_y = (H + kAltTH - 1) / kAltTH;
    int blk = blockIdx.x;
    const int tx = blk % tiles_x; blk /= tiles_x;
    const int ty = blk % tiles_y; blk /= tiles_y;
    const int s = blk % S;
    const int b = blk / S;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const float* fmap2b = fmap2 + (size_t)b * H2 * W2 * C;

    if (threadIdx.x < kAltTile) {
        const int pw = tx * kAltTW + (threadIdx.x & (kAltTW - 1)), ph = ty * kAltTH + threadIdx.x / kAltTW;
        float x = 2e6f, y = 2e6f;
        if (pw < W && ph < H) {
            const float* cp = coords + ((((size_t)b * S + s) * H + ph) * W + pw) * 2;
            x = cp[0]; y = cp[1];
        }
        // coordinates far outside (or not finite) see nothing; the bound keeps the casts defined
        const bool sane = fabsf(x) < 1e6f && fabsf(y) < 1e6f;
        const float xf = sane ? floorf(x) : -1e6f, yf = sane ? floorf(y) : -1e6f;
        const int x0 = (int)xf - r, y0 = (int)yf - r;
        ix0[threadIdx.x] = x0; iy0[threadIdx.x] = y0;
        fdx[threadIdx.x] = sane ? x - xf : 0.f; fdy[threadIdx.x] = sane ? y - yf : 0.f;
        const bool meets = x0 + n1 > 0 && x0 < W2 && y0 + n1 > 0 && y0 < H2;
        int xlo = meets ? x0 : INT_MAX, ylo = meets ? y0 : INT_MAX;
        int xhi = meets ? x0 + n1 : INT_MIN, yhi = meets ? y0 + n1 : INT_MIN;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            xlo = min(xlo, __shfl_xor(xlo, o, 64)); ylo = min(ylo, __shfl_xor(ylo, o, 64));
            xhi = max(xhi, __shfl_xor(xhi, o, 64)); yhi = max(yhi, __shfl_xor(yhi, o, 64));
        }
        if (threadIdx.x == 0) {
            const bool any = xlo <= xhi;
            const int ax = max(xlo, 0), ay = max(ylo, 0);
            red[0] = ax; red[1] = ay;
            red[2] = any ? min(xhi, W2) - ax : 0;
            red[3] = any ? min(yhi, H2) - ay : 0;
        }
    }
    __syncthreads();
    const int rx0 = red[0], ry0 = red[1], rw = red[2], rh = red[3];
    const int npos = rw * rh;
    const bool any = rw > 0 && rh > 0;
    const bool staged = any && npos <= kAltCap;

    // Lane -> (pixel of a step, 2x2 patch of the neighbourhood): ps = lane / 16 is the pixel
    // among the four of a step, the low four bits number the (n1/2)^2 patches.  off[4 st + e]:
    // where element e of the patch lives for the pixel of step st (-1: outside fmap2 or no patch).
    const int ps = lane >> 4, q16 = lane & 15, half = n1 >> 1;
    const int py2 = q16 / half, px2 = q16 - py2 * half;
    int off[kAltTW];
#pragma unroll
    for (int st = 0; st < kAltTW / 4; ++st) {
        const int pi = wv * kAltTW + st * 4 + ps;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int iy = 2 * py2 + (e >> 1), ix = 2 * px2 + (e & 1);
            const int yy = iy0[pi] + iy, xx = ix0[pi] + ix;
            const bool ok = q16 < half * half && yy >= 0 && yy < H2 && xx >= 0 && xx < W2;
            off[st * 4 + e] = !ok ? -1 : staged ? ((yy - ry0) * rw + (xx - rx0)) * kAltStride
                                                : (yy * W2 + xx) * C;
        }
    }
    // what this thread copies into LDS per chunk (the rectangle does not move between chunks)
    int goff[kAltStageIters];
#pragma unroll
    for (int t = 0; t < kAltStageIters; ++t) {
        const int idx = threadIdx.x + 256 * t, pos = idx >> 2;
        const int ry = pos / max(rw, 1), rx = pos - ry * max(rw, 1);
        goff[t] = (staged && pos < npos) ? ((ry0 + ry) * W2 + rx0 + rx) * C + (idx & 3) * 4 : -1;
    }
    // the f1 chunk: thread -> (pixel, four channels)
    const int fp = threadIdx.x >> 2;
    const int fpw = tx * kAltTW + (fp & (kAltTW - 1)), fph = ty * kAltTH + fp / kAltTW;
    const float* f1src = (fpw < W && fph < H)
        ? fmap1 + (((size_t)b * H + fph) * W + fpw) * C + (threadIdx.x & 3) * 4 : nullptr;

    float acc[kAltTW];
#pragma unroll
    for (int p = 0; p < kAltTW; ++p) acc[p] = 0.f;
    const int nch = any ? C / kAltCh : 0;
    // (Fetching the next chunk into registers while the current one is multiplied was measured
    // slower, 8.4 against 7.5 ms over four levels: the loop is bound by its LDS reads, not by the
    // fetch between the two barriers.)
    for (int ch = 0; ch < nch; ++ch) {
        __syncthreads();                       // the previous chunk has been consumed
        if (staged) {
#pragma unroll
            for (int t = 0; t < kAltStageIters; ++t)
                if (goff[t] >= 0) {
                    const int idx = threadIdx.x + 256 * t;
                    *reinterpret_cast<float4*>(reg + (idx >> 2) * kAltStride + (idx & 3) * 4) =
                        *reinterpret_cast<const float4*>(fmap2b + goff[t] + ch * kAltCh);
                }
        }
        *reinterpret_cast<float4*>(f1s + threadIdx.x * 4) =
            f1src ? *reinterpret_cast<const float4*>(f1src + ch * kAltCh) : make_float4(0.f, 0.f, 0.f, 0.f);
        __syncthreads();
        if (staged) altcorr_row_dots<true>(acc, off, f1s + (wv * kAltTW + ps) * kAltCh, reg);
        else altcorr_row_dots<false>(acc, off, f1s + (wv * kAltTW + ps) * kAltCh, fmap2b + ch * kAltCh);
    }
    __syncthreads();                           // reg becomes the output tile [K][64]
    float* tile = reg;
#pragma unroll
    for (int st = 0; st < kAltTW / 4; ++st) {
        float* d4 = dots + wv * 4 * 64;               // this wave's [4 pixels][64 positions]
        if (q16 < half * half) {
#pragma unroll
            for (int e = 0; e < 4; ++e)
                d4[ps * 64 + (2 * py2 + (e >> 1)) * n1 + 2 * px2 + (e & 1)] = acc[st * 4 + e];
        }
        // same wave wrote and reads: LDS operations of a wave complete in order
        __builtin_amdgcn_wave_barrier();
        for (int o = lane; o < 4 * K; o += 64) {      // the 4 K window values of the step's pixels
            const int sub = o / K, k = o - sub * K;
            const int pi = wv * kAltTW + st * 4 + sub;
            const float dx = fdx[pi], dy = fdy[pi];
            const int j = k / rd, i = k - j * rd;     // channel = x offset * rd + y offset
            const float* e = d4 + sub * 64 + i * n1 + j;
            tile[k * kAltTile + pi] = (1.f - dx) * (1.f - dy) * e[0] + dx * (1.f - dy) * e[1] +
                                      (1.f - dx) * dy * e[n1] + dx * dy * e[n1 + 1];
        }
        __builtin_amdgcn_wave_barrier();
    }
    __syncthreads();
    float* out = corr + (((size_t)b * S + s) * K) * H * W;
    for (int k = threadIdx.x; k < K * kAltTile; k += 256) {
        const int kk = k / kAltTile, pi = k - kk * kAltTile;
        const int pw = tx * kAltTW + (pi & (kAltTW - 1)), ph = ty * kAltTH + pi / kAltTW;
        if (pw < W && ph < H) out[((size_t)kk * H + ph) * W + pw] = tile[k];
    }
}

// Adjoint with respect to both feature maps.  One wave per (b, s, h, w); the lanes first fold
// corr_grad back onto the (2r+2)^2 integer positions (adjoint of the bilinear weights), then
// own channels: for every position with a non-zero weight the pixel's fmap1 row is added into
// that fmap2_grad row (coalesced float atomics, C contiguous floats) and the fmap2 row is
// accumulated into the pixel's fmap1_grad row, which is added once at the end (pixels of
// different s share it).
__global__ __launch_bounds__(256) void altcorr_bwd_kernel(
    const float* __restrict__ fmap1, const float* __restrict__ fmap2,
    const float* __restrict__ coords, const float* __restrict__ corr_grad,
    float* __restrict__ fmap1_grad, float* __restrict__ fmap2_grad, int B, int S, int H, int W,
    int H2, int W2, int C, int r, long long n_pix) {
    extern __shared__ float alt_lds[];
    const int n1 = 2 * r + 2, P = n1 * n1, rd = 2 * r + 1, K = rd * rd;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    float* gpos = alt_lds + wv * P;
    const long long pix = (long long)blockIdx.x * 4 + wv;
    if (pix >= n_pix) return;
    long long t = pix;
    const int w = (int)(t % W); t /= W;
    const int h = (int)(t % H); t /= H;
    const int s = (int)(t % S);
    const int b = (int)(t / S);
    const float* cp = coords + (size_t)pix * 2;
    const float x = cp[0], y = cp[1];
    const bool sane = fabsf(x) < 1e6f && fabsf(y) < 1e6f;
    const float xf = sane ? floorf(x) : -1e6f, yf = sane ? floorf(y) : -1e6f;
    const float dx = sane ? x - xf : 0.f, dy = sane ? y - yf : 0.f;
    const int x0 = (int)xf - r, y0 = (int)yf - r;
    const float w00 = (1.f - dx) * (1.f - dy), w01 = dx * (1.f - dy), w10 = (1.f - dx) * dy,
                w11 = dx * dy;
    const float* g = corr_grad + ((((size_t)b * S + s) * K) * H + h) * W + w;
    const size_t plane = (size_t)H * W;
    for (int q = lane; q < P; q += 64) {
        const int iy = q / n1, ix = q - iy * n1;
        float v = 0.f;
        if (iy < rd && ix < rd) v += w00 * g[(size_t)(ix * rd + iy) * plane];
        if (iy < rd && ix > 0) v += w01 * g[(size_t)((ix - 1) * rd + iy) * plane];
        if (iy > 0 && ix < rd) v += w10 * g[(size_t)(ix * rd + iy - 1) * plane];
        if (iy > 0 && ix > 0) v += w11 * g[(size_t)((ix - 1) * rd + iy - 1) * plane];
        gpos[q] = v;
    }
    __builtin_amdgcn_wave_barrier();     // the wave's own LDS writes, read back below in order
    const float* a = fmap1 + (((size_t)b * H + h) * W + w) * C;
    constexpr int kMaxPerLane = 8;                      // C <= 512
    float f1v[kMaxPerLane], acc[kMaxPerLane];
#pragma unroll
    for (int k = 0; k < kMaxPerLane; ++k) {
        const int c = lane + 64 * k;
        f1v[k] = c < C ? a[c] : 0.f;
        acc[k] = 0.f;
    }
    for (int q = 0; q < P; ++q) {
        const float gq = gpos[q];                       // broadcast; written by this wave
        if (gq == 0.f) continue;                        // wave-uniform
        const int iy = q / n1, ix = q - iy * n1;
        const int yy = y0 + iy, xx = x0 + ix;
        if (yy < 0 || yy >= H2 || xx < 0 || xx >= W2) continue;
        const size_t off = (((size_t)b * H2 + yy) * W2 + xx) * C;
#pragma unroll
        for (int k = 0; k < kMaxPerLane; ++k) {
            const int c = lane + 64 * k;
            if (c < C) {
                acc[k] += gq * fmap2[off + c];
                atomicAdd(fmap2_grad + off + c, gq * f1v[k]);
            }
        }
    }
    float* o = fmap1_grad + (((size_t)b * H + h) * W + w) * C;
#pragma unroll
    for (int k = 0; k < kMaxPerLane; ++k) {
        const int c = lane + 64 * k;
        if (c < C && acc[k] != 0.f) atomicAdd(o + c, acc[k]);
    }
}


}  // namespace vings

using namespace vings;

extern "C" int vings_frame_distance(const float* poses, const float* disps, const float* intr,
                                    const int64_t* ii, const int64_t* jj, int B, int ht, int wd,
                                    int E, float beta, float* out, void* stream) {
    VINGS_REQUIRE(B > 0 && ht > 0 && wd > 0 && E >= 0, "sizes");
    if (E == 0) return VINGS_OK;
    VINGS_REQUIRE(poses && disps && intr && ii && jj && out, "null pointer");
    ProfScope prof(PROF_AUX, (hipStream_t)stream);
    frame_distance_kernel<<<E, 256, 0, (hipStream_t)stream>>>(poses, disps, intr, ii, jj, B, ht, wd,
                                                              beta, out);
    VINGS_LAUNCH_CHECK();
    return VINGS_OK;
}

extern "C" int vings_corr_index_forward(const void* volume, const float* coords, void* corr, int E,
                                        int h1, int w1, int h2, int w2, int r, int half,
                                        void* stream) {
    VINGS_REQUIRE(E >= 0 && h1 > 0 && w1 > 0 && h2 > 0 && w2 > 0 && r >= 0, "sizes");
    if (E == 0) return VINGS_OK;
    VINGS_REQUIRE(volume && coords && corr, "null pointer");
    const int64_t n = (int64_t)E * h1 * w1;
    ProfScope prof(PROF_AUX, (hipStream_t)stream);
    if (r == 3) {
        const int groups = cdiv((int64_t)h1 * w1, 64);
        const int blocks = cdiv((int64_t)E * groups, 4);
        if (half)
            corr_index_fwd_r3_kernel<__half><<<blocks, 256, 0, (hipStream_t)stream>>>(
                (const __half*)volume, coords, (__half*)corr, E, h1, w1, h2, w2);
        else
            corr_index_fwd_r3_kernel<float><<<blocks, 256, 0, (hipStream_t)stream>>>(
                (const float*)volume, coords, (float*)corr, E, h1, w1, h2, w2);
        VINGS_LAUNCH_CHECK();
        return VINGS_OK;
    }
    if (half)
        corr_index_fwd_kernel<__half><<<cdiv(n, 256), 256, 0, (hipStream_t)stream>>>(
            (const __half*)volume, coords, (__half*)corr, E, h1, w1, h2, w2, r);
    else
        corr_index_fwd_kernel<float><<<cdiv(n, 256), 256, 0, (hipStream_t)stream>>>(
            (const float*)volume, coords, (float*)corr, E, h1, w1, h2, w2, r);
    VINGS_LAUNCH_CHECK();
    return VINGS_OK;
}

extern "C" int vings_corr_index_backward(const float* coords, const void* corr_grad,
                                         void* volume_grad, int E, int h1, int w1, int h2, int w2,
                                         int r, int half, void* stream) {
    VINGS_REQUIRE(E >= 0 && h1 > 0 && w1 > 0 && h2 > 0 && w2 > 0 && r >= 0, "sizes");
    if (E == 0) return VINGS_OK;
    VINGS_REQUIRE(coords && corr_grad && volume_grad, "null pointer");
    hipStream_t s = (hipStream_t)stream;
    const int64_t n = (int64_t)E * h1 * w1;
    const size_t bytes = (size_t)n * h2 * w2 * (half ? 2 : 4);
    VINGS_CHECK_HIP(hipMemsetAsync(volume_grad, 0, bytes, s));
    ProfScope prof(PROF_AUX, s);
    if (half)
        corr_index_bwd_kernel<__half><<<cdiv(n, 256), 256, 0, s>>>(
            coords, (const __half*)corr_grad, (__half*)volume_grad, E, h1, w1, h2, w2, r);
    else
        corr_index_bwd_kernel<float><<<cdiv(n, 256), 256, 0, s>>>(
            coords, (const float*)corr_grad, (float*)volume_grad, E, h1, w1, h2, w2, r);
    VINGS_LAUNCH_CHECK();
    return VINGS_OK;
}

static int altcorr_check(int B, int S, int H, int W, int H2, int W2, int C, int r) {
    VINGS_REQUIRE(B >= 0 && S > 0 && H > 0 && W > 0 && H2 > 0 && W2 > 0, "sizes");
    VINGS_REQUIRE(C > 0 && (C & 3) == 0 && C <= 512, "channels: a multiple of 4, at most 512");
    VINGS_REQUIRE(r >= 0 && r <= 7, "radius 0..7");
    return VINGS_OK;
}

extern "C" int vings_altcorr_forward(const float* fmap1, const float* fmap2, const float* coords,
                                     float* corr, int B, int S, int H, int W, int H2, int W2, int C,
                                     int r, void* stream) {
    const int rc = altcorr_check(B, S, H, W, H2, W2, C, r);
    if (rc != VINGS_OK) return rc;
    if (B == 0) return VINGS_OK;
    VINGS_REQUIRE(fmap1 && fmap2 && coords && corr, "null pointer");
    VINGS_REQUIRE((((uintptr_t)fmap1 | (uintptr_t)fmap2) & 15) == 0, "feature maps must be 16-byte aligned");
    if (r <= 3 && (C % kAltCh) == 0 && (long long)H2 * W2 * C < (1ll << 31)) {
        const long long tiles = (long long)B * S * cdiv(H, kAltTH) * cdiv(W, kAltTW);
        VINGS_REQUIRE(tiles < (1ll << 31), "grid too large");
        // (per device, so not cached in a process-wide flag: a few microseconds per call)
        VINGS_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(altcorr_fwd_tile_kernel),
                                            hipFuncAttributeMaxDynamicSharedMemorySize,
                                            (int)kAltTileLds));
        ProfScope prof(PROF_AUX, (hipStream_t)stream);
        altcorr_fwd_tile_kernel<<<(int)tiles, 256, kAltTileLds, (hipStream_t)stream>>>(
            fmap1, fmap2, coords, corr, B, S, H, W, H2, W2, C, r);
        VINGS_LAUNCH_CHECK();
        return VINGS_OK;
    }
    const int n1 = 2 * r + 2, rd = 2 * r + 1;
    const size_t lds = sizeof(float) * ((size_t)kAltPx * C + 4 * n1 * n1 + (size_t)rd * rd * kAltPx);
    VINGS_REQUIRE(lds <= 160 * 1024, "window and channel count exceed the LDS of a CU");
    const long long blocks = (long long)B * S * H * cdiv(W, kAltPx);
    VINGS_REQUIRE(blocks < (1ll << 31), "grid too large");
    if (lds > 64 * 1024)
        VINGS_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(altcorr_fwd_kernel),
                                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    ProfScope prof(PROF_AUX, (hipStream_t)stream);
    altcorr_fwd_kernel<<<(int)blocks, 256, lds, (hipStream_t)stream>>>(fmap1, fmap2, coords, corr, B, S,
                                                                      H, W, H2, W2, C, r);
    VINGS_LAUNCH_CHECK();
    return VINGS_OK;
}

extern "C" int vings_altcorr_backward(const float* fmap1, const float* fmap2, const float* coords,
                                      const float* corr_grad, float* fmap1_grad, float* fmap2_grad,
                                      int B, int S, int H, int W, int H2, int W2, int C, int r,
                                      void* stream) {
    const int rc = altcorr_check(B, S, H, W, H2, W2, C, r);
    if (rc != VINGS_OK) return rc;
    if (B == 0) return VINGS_OK;
    VINGS_REQUIRE(fmap1 && fmap2 && coords && corr_grad && fmap1_grad && fmap2_grad, "null pointer");
    hipStream_t s = (hipStream_t)stream;
    VINGS_CHECK_HIP(hipMemsetAsync(fmap1_grad, 0, sizeof(float) * (size_t)B * H * W * C, s));
    VINGS_CHECK_HIP(hipMemsetAsync(fmap2_grad, 0, sizeof(float) * (size_t)B * H2 * W2 * C, s));
    const long long n_pix = (long long)B * S * H * W;
    const long long blocks = (n_pix + 3) / 4;
    VINGS_REQUIRE(blocks < (1ll << 31), "grid too large");
    const int n1 = 2 * r + 2;
    ProfScope prof(PROF_AUX, s);
    altcorr_bwd_kernel<<<(int)blocks, 256, sizeof(float) * 4 * n1 * n1, s>>>(
        fmap1, fmap2, coords, corr_grad, fmap1_grad, fmap2_grad, B, S, H, W, H2, W2, C, r, n_pix);
    VINGS_LAUNCH_CHECK();
    return VINGS_OK;
}
