/*
 * surfel_oracle.c -- CPU restatement of the 2D-Gaussian (surfel) rasteriser.
 *
 * TEST INFRASTRUCTURE ONLY.  This file is the parity oracle and the reported CPU
 * baseline for the HIP rasteriser; nothing in the shipped package imports, links
 * or executes it.  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg load the shared object built from it.
 *
 * PARITY UNPINNED: the reference calls this algorithm through the Python
 * extension `diff_surfel_rasterization`
 * (/root/reference/scripts/gaussian/gaussian_base.py:5-6,160-203) whose CUDA
 * sources live in an un-vendored submodule (/root/reference/.gitmodules:1-4,
 * pip name diff-surfel-rasterization==0.0.1, /root/reference/backup/piplist.txt:42)
 * and the reference ships no tests or golden vectors for it.  What follows
 * restates the published 2D Gaussian Splatting rasteriser (ray/splat
 * intersection through a 3x3 homography, 3-sigma bounding box, low-pass
 * filter, front-to-back alpha blending over 16x16 tiles sorted by
 * tile<<32|depth keys) with the extensions the reference's call sites
 * require:
 *   - settings field `pixel_mask`           (gaussian_base.py:158,175)
 *   - `scores` gradient sink of shape [N,2] (gaussian_base.py:70,201,387)
 *   - allmap channels 7:9 = blended per-splat 2-vector `u2_minus_u1`
 *                                           (gaussian_model.py:268-286,427)
 * Output channel meaning follows the consumer at gaussian_base.py:205-219:
 * allmap[0]=sum w*depth, [1]=alpha, [2:5]=sum w*normal(view space),
 * [5]=median depth, [6]=distortion.
 *
 * The analytic backward is validated against central finite differences of
 * the forward in the float64 build of this same file (tests/test_oracle_surfel.py).
 *
 * Build: see oracle/Makefile (float32: libsurfel_oracle_f32.so,
 * float64: libsurfel_oracle_f64.so with -DORACLE_F64).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#ifdef ORACLE_F64
typedef double real;
#define R_EXP exp
#define R_SQRT sqrt
#define R_CEIL ceil
#else
typedef float real;
#define R_EXP expf
#define R_SQRT sqrtf
#define R_CEIL ceilf
#endif

#define TILE 16
#define NEAR_N ((real)0.2)
#define FAR_N ((real)100.0)
#define FILTER_SIZE ((real)0.707106)
#define FILTER_INV_SQUARE ((real)2.0)
#define ALPHA_MAX ((real)0.99)
#define ALPHA_MIN ((real)1.0 / (real)255.0)
#define T_MIN ((real)0.0001)
#define N_AUX 9 /* allmap channels */

static inline real rmax(real a, real b) { return a > b ? a : b; }
static inline real rmin(real a, real b) { return a < b ? a : b; }
static inline int imax(int a, int b) { return a > b ? a : b; }
static inline int imin(int a, int b) { return a < b ? a : b; }

/* Matrices arrive as the flat row-major storage of the reference's
 * `world_view_transform = w2c.T` and `full_proj_transform`
 * (/root/reference/scripts/gaussian/cameras.py:58-60): element [c*4+r] is
 * entry (r,c) of the column-vector matrix. */
static inline void xform_point(const real* m, const real* p, real* o) {
    o[0] = m[0] * p[0] + m[4] * p[1] + m[8] * p[2] + m[12];
    o[1] = m[1] * p[0] + m[5] * p[1] + m[9] * p[2] + m[13];
    o[2] = m[2] * p[0] + m[6] * p[1] + m[10] * p[2] + m[14];
}
static inline void xform_vec(const real* m, const real* v, real* o) {
    o[0] = m[0] * v[0] + m[4] * v[1] + m[8] * v[2];
    o[1] = m[1] * v[0] + m[5] * v[1] + m[9] * v[2];
    o[2] = m[2] * v[0] + m[6] * v[1] + m[10] * v[2];
}

static inline void quat_to_R(const real* q, real R[3][3]) {
    real r = q[0], x = q[1], y = q[2], z = q[3];
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

/* B = ndc2pix * PW : maps homogeneous world points to homogeneous pixels.
 * Pixel centres sit on integers: px = ((x_ndc+1)*W-1)/2. */
static inline void pixel_projection(const real* proj, int W, int H, real B[3][4]) {
    real hw = (real)W / 2, hh = (real)H / 2;
    real ow = (real)(W - 1) / 2, oh = (real)(H - 1) / 2;
    for (int c = 0; c < 4; ++c) {
        real p0 = proj[c * 4 + 0], p1 = proj[c * 4 + 1], p3 = proj[c * 4 + 3];
        B[0][c] = hw * p0 + ow * p3;
        B[1][c] = hh * p1 + oh * p3;
        B[2][c] = p3;
    }
}

/* ------------------------------------------------------------------ */
/* projection ("preprocess") forward                                   */
/* ------------------------------------------------------------------ */
int surfel_project(int N, int H, int W, real scale_mod, const real* means3D,
                   const real* scales, const real* rots, const real* opac,
                   const real* view, const real* proj,
                   real* xy, real* depth, real* transmat, real* normal_opacity,
                   int32_t* radii, int32_t* rect, uint32_t* tiles_touched) {
    const int gx = (W + TILE - 1) / TILE, gy = (H + TILE - 1) / TILE;
    real B[3][4];
    pixel_projection(proj, W, H, B);
#pragma omp parallel for schedule(static)
    for (int i = 0; i < N; ++i) {
        radii[i] = 0;
        tiles_touched[i] = 0;
        rect[4 * i + 0] = rect[4 * i + 1] = rect[4 * i + 2] = rect[4 * i + 3] = 0;
        xy[2 * i] = xy[2 * i + 1] = 0;
        depth[i] = 0;
        for (int k = 0; k < 9; ++k) transmat[9 * i + k] = 0;
        for (int k = 0; k < 4; ++k) normal_opacity[4 * i + k] = 0;

        const real* p = means3D + 3 * i;
        real pv[3];
        xform_point(view, p, pv);
        if (pv[2] <= NEAR_N) continue;

        real R[3][3];
        quat_to_R(rots + 4 * i, R);
        real sx = scale_mod * scales[2 * i], sy = scale_mod * scales[2 * i + 1];
        real L0[3] = {R[0][0] * sx, R[1][0] * sx, R[2][0] * sx};
        real L1[3] = {R[0][1] * sy, R[1][1] * sy, R[2][1] * sy};
        real nw[3] = {R[0][2], R[1][2], R[2][2]};
        real A[3][3];
        for (int r = 0; r < 3; ++r) {
            A[r][0] = B[r][0] * L0[0] + B[r][1] * L0[1] + B[r][2] * L0[2];
            A[r][1] = B[r][0] * L1[0] + B[r][1] * L1[1] + B[r][2] * L1[2];
            A[r][2] = B[r][0] * p[0] + B[r][1] * p[1] + B[r][2] * p[2] + B[r][3];
        }
        real n[3];
        xform_vec(view, nw, n);
        real cosv = -(pv[0] * n[0] + pv[1] * n[1] + pv[2] * n[2]);
        if (cosv == 0) continue;
        real mult = cosv > 0 ? (real)1 : (real)-1;
        n[0] *= mult; n[1] *= mult; n[2] *= mult;

        /* 3-sigma bounding box of the projected splat */
        const real t[3] = {9, 9, -1};
        const real *Tu = A[0], *Tv = A[1], *Tw = A[2];
        real d = t[0] * Tw[0] * Tw[0] + t[1] * Tw[1] * Tw[1] + t[2] * Tw[2] * Tw[2];
        if (d == 0) continue;
        real inv = (real)1 / d;
        real f[3] = {inv * t[0], inv * t[1], inv * t[2]};
        real cx = f[0] * Tu[0] * Tw[0] + f[1] * Tu[1] * Tw[1] + f[2] * Tu[2] * Tw[2];
        real cy = f[0] * Tv[0] * Tw[0] + f[1] * Tv[1] * Tw[1] + f[2] * Tv[2] * Tw[2];
        real ex = cx * cx - (f[0] * Tu[0] * Tu[0] + f[1] * Tu[1] * Tu[1] + f[2] * Tu[2] * Tu[2]);
        real ey = cy * cy - (f[0] * Tv[0] * Tv[0] + f[1] * Tv[1] * Tv[1] + f[2] * Tv[2] * Tv[2]);
        real hx = R_SQRT(rmax((real)1e-4, ex)), hy = R_SQRT(rmax((real)1e-4, ey));
        real radius = R_CEIL(rmax(rmax(hx, hy), (real)3 * FILTER_SIZE));
        /* guard: a non-finite or absurd radius is culled (cannot index tiles) */
        if (!(radius < (real)1e7) || !(cx == cx) || !(cy == cy) ||
            !(cx > (real)-1e9 && cx < (real)1e9) || !(cy > (real)-1e9 && cy < (real)1e9))
            continue;

        int x0 = imin(gx, imax(0, (int)((cx - radius) / TILE)));
        int y0 = imin(gy, imax(0, (int)((cy - radius) / TILE)));
        int x1 = imin(gx, imax(0, (int)((cx + radius + TILE - 1) / TILE)));
        int y1 = imin(gy, imax(0, (int)((cy + radius + TILE - 1) / TILE)));
        if ((x1 - x0) * (y1 - y0) == 0) continue;

        depth[i] = pv[2];
        radii[i] = (int32_t)radius;
        xy[2 * i] = cx; xy[2 * i + 1] = cy;
        for (int r = 0; r < 3; ++r)
            for (int c = 0; c < 3; ++c) transmat[9 * i + 3 * r + c] = A[r][c];
        normal_opacity[4 * i + 0] = n[0];
        normal_opacity[4 * i + 1] = n[1];
        normal_opacity[4 * i + 2] = n[2];
        normal_opacity[4 * i + 3] = opac[i];
        rect[4 * i + 0] = x0; rect[4 * i + 1] = y0; rect[4 * i + 2] = x1; rect[4 * i + 3] = y1;
        tiles_touched[i] = (uint32_t)((x1 - x0) * (y1 - y0));
    }
    return 0;
}

/* ------------------------------------------------------------------ */
/* binning: duplicate with keys, stable sort, tile ranges              */
/* ------------------------------------------------------------------ */
int64_t surfel_bin_count(int N, const uint32_t* tiles_touched) {
    int64_t D = 0;
    for (int i = 0; i < N; ++i) D += tiles_touched[i];
    return D;
}

static void radix_sort_pairs(uint64_t* keys, uint32_t* vals, int64_t n) {
    /* stable LSD radix sort, 8 bits per pass over all 64 key bits */
    uint64_t* k2 = (uint64_t*)malloc(sizeof(uint64_t) * (size_t)(n > 0 ? n : 1));
    uint32_t* v2 = (uint32_t*)malloc(sizeof(uint32_t) * (size_t)(n > 0 ? n : 1));
    uint64_t *ka = keys, *kb = k2;
    uint32_t *va = vals, *vb = v2;
    for (int pass = 0; pass < 8; ++pass) {
        int64_t cnt[257];
        memset(cnt, 0, sizeof(cnt));
        int sh = pass * 8;
        for (int64_t i = 0; i < n; ++i) cnt[((ka[i] >> sh) & 255) + 1]++;
        for (int b = 0; b < 256; ++b) cnt[b + 1] += cnt[b];
        for (int64_t i = 0; i < n; ++i) {
            int64_t dst = cnt[(ka[i] >> sh) & 255]++;
            kb[dst] = ka[i];
            vb[dst] = va[i];
        }
        uint64_t* tk = ka; ka = kb; kb = tk;
        uint32_t* tv = va; va = vb; vb = tv;
    }
    /* 8 passes: result is back in keys/vals */
    free(k2);
    free(v2);
}

int surfel_bin(int N, int H, int W, const real* depth, const int32_t* rect,
               const uint32_t* tiles_touched, int64_t D, uint64_t* keys,
               uint32_t* vals, uint32_t* ranges) {
    const int gx = (W + TILE - 1) / TILE, gy = (H + TILE - 1) / TILE;
    int64_t off = 0;
    for (int i = 0; i < N; ++i) {
        if (tiles_touched[i] == 0) continue;
        float df = (float)depth[i];
        uint32_t bits;
        memcpy(&bits, &df, 4);
        for (int y = rect[4 * i + 1]; y < rect[4 * i + 3]; ++y)
            for (int x = rect[4 * i + 0]; x < rect[4 * i + 2]; ++x) {
                uint64_t key = (uint64_t)(uint32_t)(y * gx + x);
                key = (key << 32) | bits;
                keys[off] = key;
                vals[off] = (uint32_t)i;
                ++off;
            }
    }
    if (off != D) return -1;
    radix_sort_pairs(keys, vals, D);
    memset(ranges, 0, sizeof(uint32_t) * 2 * (size_t)gx * gy);
    for (int64_t i = 0; i < D; ++i) {
        uint32_t t = (uint32_t)(keys[i] >> 32);
        if (i == 0 || (uint32_t)(keys[i - 1] >> 32) != t) ranges[2 * t] = (uint32_t)i;
        if (i == D - 1 || (uint32_t)(keys[i + 1] >> 32) != t) ranges[2 * t + 1] = (uint32_t)(i + 1);
    }
    return 0;
}

/* ------------------------------------------------------------------ */
/* per pixel / splat evaluation shared by forward and backward          */
/* ------------------------------------------------------------------ */
typedef struct {
    real kx, ky, kz, lx, ly, lz, px, py, pz, sx, sy, rho3d, rho2d, dx, dy, depth, G, alpha;
} Hit;

/* returns 0 when the splat does not contribute to the pixel */
static inline int eval_hit(const real* Tm, const real* cxy, real opa, real fx, real fy, Hit* h) {
    const real *Tu = Tm, *Tv = Tm + 3, *Tw = Tm + 6;
    h->kx = fx * Tw[0] - Tu[0]; h->ky = fx * Tw[1] - Tu[1]; h->kz = fx * Tw[2] - Tu[2];
    h->lx = fy * Tw[0] - Tv[0]; h->ly = fy * Tw[1] - Tv[1]; h->lz = fy * Tw[2] - Tv[2];
    h->px = h->ky * h->lz - h->kz * h->ly;
    h->py = h->kz * h->lx - h->kx * h->lz;
    h->pz = h->kx * h->ly - h->ky * h->lx;
    if (h->pz == 0) return 0;
    h->sx = h->px / h->pz; h->sy = h->py / h->pz;
    h->rho3d = h->sx * h->sx + h->sy * h->sy;
    h->dx = cxy[0] - fx; h->dy = cxy[1] - fy;
    h->rho2d = FILTER_INV_SQUARE * (h->dx * h->dx + h->dy * h->dy);
    real rho = rmin(h->rho3d, h->rho2d);
    h->depth = (h->rho3d <= h->rho2d) ? (h->sx * Tw[0] + h->sy * Tw[1]) + Tw[2] : Tw[2];
    if (h->depth < NEAR_N) return 0;
    real power = (real)-0.5 * rho;
    if (power > 0) return 0;
    h->G = R_EXP(power);
    h->alpha = rmin(ALPHA_MAX, opa * h->G);
    if (h->alpha < ALPHA_MIN) return 0;
    return 1;
}

/* ------------------------------------------------------------------ */
/* blend forward                                                       */
/* ------------------------------------------------------------------ */
int surfel_blend_forward(int H, int W, const uint32_t* ranges, const uint32_t* vals,
                         const real* xy, const real* transmat, const real* normal_opacity,
                         const real* colors, const real* flow, const real* bg,
                         const uint8_t* pixel_mask, real* out_color, real* allmap,
                         real* final_T, uint32_t* n_contrib) {
    const int gx = (W + TILE - 1) / TILE, gy = (H + TILE - 1) / TILE;
    const size_t HW = (size_t)H * W;
#pragma omp parallel for schedule(dynamic, 4)
    for (int tile = 0; tile < gx * gy; ++tile) {
        int tx = tile % gx, ty = tile / gx;
        uint32_t r0 = ranges[2 * tile], r1 = ranges[2 * tile + 1];
        for (int yy = 0; yy < TILE; ++yy)
            for (int xx = 0; xx < TILE; ++xx) {
                int x = tx * TILE + xx, y = ty * TILE + yy;
                if (x >= W || y >= H) continue;
                size_t pix = (size_t)y * W + x;
                real C[3] = {0, 0, 0}, Nn[3] = {0, 0, 0}, F[2] = {0, 0};
                real T = 1, Dd = 0, M1 = 0, M2 = 0, dist = 0, med = 0;
                uint32_t contributor = 0, last = 0, medc = 0;
                int masked = pixel_mask && !pixel_mask[pix];
                if (!masked) {
                    for (uint32_t j = r0; j < r1; ++j) {
                        ++contributor;
                        uint32_t g = vals[j];
                        Hit h;
                        if (!eval_hit(transmat + 9 * g, xy + 2 * g, normal_opacity[4 * g + 3],
                                      (real)x, (real)y, &h))
                            continue;
                        real test_T = T * (1 - h.alpha);
                        if (test_T < T_MIN) break;
                        real w = h.alpha * T;
                        real A = 1 - T;
                        real m = FAR_N / (FAR_N - NEAR_N) * (1 - NEAR_N / h.depth);
                        dist += (m * m * A + M2 - 2 * m * M1) * w;
                        Dd += h.depth * w;
                        M1 += m * w;
                        M2 += m * m * w;
                        if (T > (real)0.5) { med = h.depth; medc = contributor; }
                        for (int c = 0; c < 3; ++c) Nn[c] += normal_opacity[4 * g + c] * w;
                        for (int c = 0; c < 3; ++c) C[c] += colors[3 * g + c] * w;
                        if (flow) { F[0] += flow[2 * g] * w; F[1] += flow[2 * g + 1] * w; }
                        T = test_T;
                        last = contributor;
                    }
                }
                final_T[pix] = T; final_T[HW + pix] = M1; final_T[2 * HW + pix] = M2;
                n_contrib[pix] = last; n_contrib[HW + pix] = medc;
                if (masked) {
                    for (int c = 0; c < 3; ++c) out_color[c * HW + pix] = bg[c];
                    for (int c = 0; c < N_AUX; ++c) allmap[c * HW + pix] = 0;
                    continue;
                }
                for (int c = 0; c < 3; ++c) out_color[c * HW + pix] = C[c] + T * bg[c];
                allmap[0 * HW + pix] = Dd;
                allmap[1 * HW + pix] = 1 - T;
                allmap[2 * HW + pix] = Nn[0];
                allmap[3 * HW + pix] = Nn[1];
                allmap[4 * HW + pix] = Nn[2];
                allmap[5 * HW + pix] = med;
                allmap[6 * HW + pix] = dist;
                allmap[7 * HW + pix] = F[0];
                allmap[8 * HW + pix] = F[1];
            }
    }
    return 0;
}

/* ------------------------------------------------------------------ */
/* blend backward                                                      */
/* ------------------------------------------------------------------ */
static inline void atomic_add(real* p, real v) {
#pragma omp atomic
    *p += v;
}

int surfel_blend_backward(int H, int W, const uint32_t* ranges, const uint32_t* vals,
                          const real* xy, const real* transmat, const real* normal_opacity,
                          const real* colors, const real* flow, const real* bg,
                          const uint8_t* pixel_mask, const real* final_T,
                          const uint32_t* n_contrib, const real* dL_dcolor,
                          const real* dL_dallmap, real* dL_dtransmat, real* dL_dxy,
                          real* dL_dnormal, real* dL_dopacity, real* dL_dcolors,
                          real* scores) {
    const int gx = (W + TILE - 1) / TILE, gy = (H + TILE - 1) / TILE;
    const size_t HW = (size_t)H * W;
#pragma omp parallel for schedule(dynamic, 4)
    for (int tile = 0; tile < gx * gy; ++tile) {
        int tx = tile % gx, ty = tile / gx;
        uint32_t r0 = ranges[2 * tile];
        for (int yy = 0; yy < TILE; ++yy)
            for (int xx = 0; xx < TILE; ++xx) {
                int x = tx * TILE + xx, y = ty * TILE + yy;
                if (x >= W || y >= H) continue;
                size_t pix = (size_t)y * W + x;
                if (pixel_mask && !pixel_mask[pix]) continue;
                uint32_t last = n_contrib[pix], medc = n_contrib[HW + pix];
                if (last == 0) continue;
                const real T_final = final_T[pix];
                const real M1f = final_T[HW + pix], M2f = final_T[2 * HW + pix];
                const real A_final = 1 - T_final;
                real gC[3], gN[3], gF[2];
                for (int c = 0; c < 3; ++c) gC[c] = dL_dcolor[c * HW + pix];
                const real gD = dL_dallmap[0 * HW + pix];
                const real gA = dL_dallmap[1 * HW + pix];
                for (int c = 0; c < 3; ++c) gN[c] = dL_dallmap[(2 + c) * HW + pix];
                const real gMed = dL_dallmap[5 * HW + pix];
                const real gDist = dL_dallmap[6 * HW + pix];
                gF[0] = dL_dallmap[7 * HW + pix];
                gF[1] = dL_dallmap[8 * HW + pix];
                const real bg_dot = bg[0] * gC[0] + bg[1] * gC[1] + bg[2] * gC[2];
                const real gabs = (real)(fabs((double)gC[0]) + fabs((double)gC[1]) + fabs((double)gC[2]));

                real T = T_final;
                real last_alpha = 0;
                real accC[3] = {0, 0, 0}, lastC[3] = {0, 0, 0};
                real accN[3] = {0, 0, 0}, lastN[3] = {0, 0, 0};
                real accF[2] = {0, 0}, lastF[2] = {0, 0};
                real accD = 0, lastD = 0, accA = 0, accW = 0, lastW = 0;

                for (int64_t pos = (int64_t)last - 1; pos >= 0; --pos) {
                    uint32_t g = vals[r0 + pos];
                    const real* Tm = transmat + 9 * g;
                    real opa = normal_opacity[4 * g + 3];
                    Hit h;
                    if (!eval_hit(Tm, xy + 2 * g, opa, (real)x, (real)y, &h)) continue;
                    const real alpha = h.alpha;
                    T = T / (1 - alpha);
                    const real w = alpha * T;
                    real dL_dalpha = 0;
                    for (int c = 0; c < 3; ++c) {
                        real col = colors[3 * g + c];
                        accC[c] = last_alpha * lastC[c] + (1 - last_alpha) * accC[c];
                        lastC[c] = col;
                        dL_dalpha += (col - accC[c]) * gC[c];
                        atomic_add(&dL_dcolors[3 * g + c], w * gC[c]);
                    }
                    if (flow) {
                        for (int c = 0; c < 2; ++c) {
                            real fl = flow[2 * g + c];
                            accF[c] = last_alpha * lastF[c] + (1 - last_alpha) * accF[c];
                            lastF[c] = fl;
                            dL_dalpha += (fl - accF[c]) * gF[c];
                        }
                    }
                    /* distortion */
                    real dL_dz = 0;
                    const real m = FAR_N / (FAR_N - NEAR_N) * (1 - NEAR_N / h.depth);
                    const real dm_dd = (FAR_N * NEAR_N) / ((FAR_N - NEAR_N) * h.depth * h.depth);
                    if ((uint32_t)(pos + 1) == medc) dL_dz += gMed;
                    const real Wd = (M2f + m * m * A_final - 2 * m * M1f) * gDist;
                    accW = last_alpha * lastW + (1 - last_alpha) * accW;
                    lastW = Wd;
                    dL_dalpha += Wd - accW;
                    dL_dz += 2 * w * (m * A_final - M1f) * gDist * dm_dd;
                    /* alpha channel */
                    accA = last_alpha * 1 + (1 - last_alpha) * accA;
                    dL_dalpha += (1 - accA) * gA;
                    /* depth channel */
                    accD = last_alpha * lastD + (1 - last_alpha) * accD;
                    lastD = h.depth;
                    dL_dalpha += (h.depth - accD) * gD;
                    /* normal channels */
                    for (int c = 0; c < 3; ++c) {
                        real nn = normal_opacity[4 * g + c];
                        accN[c] = last_alpha * lastN[c] + (1 - last_alpha) * accN[c];
                        lastN[c] = nn;
                        dL_dalpha += (nn - accN[c]) * gN[c];
                        atomic_add(&dL_dnormal[3 * g + c], w * gN[c]);
                    }
                    dL_dalpha *= T;
                    last_alpha = alpha;
                    dL_dalpha += (-T_final / (1 - alpha)) * bg_dot;

                    atomic_add(&scores[2 * g + 0], w);
                    atomic_add(&scores[2 * g + 1], w * gabs);

                    const real dL_dG = opa * dL_dalpha;
                    atomic_add(&dL_dopacity[g], h.G * dL_dalpha);
                    dL_dz += w * gD;

                    if (h.rho3d <= h.rho2d) {
                        const real *Tw = Tm + 6;
                        real dsx = dL_dG * -h.G * h.sx + dL_dz * Tw[0];
                        real dsy = dL_dG * -h.G * h.sy + dL_dz * Tw[1];
                        real dsx_pz = dsx / h.pz, dsy_pz = dsy / h.pz;
                        real dpx = dsx_pz, dpy = dsy_pz, dpz = -(dsx_pz * h.sx + dsy_pz * h.sy);
                        /* dL/dk = l x dp ; dL/dl = dp x k */
                        real dkx = h.ly * dpz - h.lz * dpy;
                        real dky = h.lz * dpx - h.lx * dpz;
                        real dkz = h.lx * dpy - h.ly * dpx;
                        real dlx = dpy * h.kz - dpz * h.ky;
                        real dly = dpz * h.kx - dpx * h.kz;
                        real dlz = dpx * h.ky - dpy * h.kx;
                        /* Gradient w.r.t. the rows of T' = M T, M = [[1,0,-cx],[0,1,-cy],[0,0,1]]:
                         * the homography in pixel coordinates centred on the splat (cx, cy).
                         * k = (x-cx) Tw - Tu' and l = (y-cy) Tw - Tv' are the same vectors, so
                         * dL/dTw' = (x-cx) dk + (y-cy) dl + ...: pixel offsets of a few pixels
                         * instead of absolute coordinates of up to the image size, whose
                         * products cancel down to 1e-3 of their size in the chain rule that
                         * follows (surfel_project_backward applies M to the projection). */
                        /* (cx, cy) clamped into the image: any fixed point serves, and a
                         * centre far outside the image (a splat crossing the camera plane)
                         * would bring back offsets of 1e5 pixels */
                        real ccx = rmin(rmax(xy[2 * g], 0), (real)(W - 1));
                        real ccy = rmin(rmax(xy[2 * g + 1], 0), (real)(H - 1));
                        real ex = (real)x - ccx, ey = (real)y - ccy;
                        real* o = dL_dtransmat + 9 * g;
                        atomic_add(o + 0, -dkx); atomic_add(o + 1, -dky); atomic_add(o + 2, -dkz);
                        atomic_add(o + 3, -dlx); atomic_add(o + 4, -dly); atomic_add(o + 5, -dlz);
                        atomic_add(o + 6, ex * dkx + ey * dlx + dL_dz * h.sx);
                        atomic_add(o + 7, ex * dky + ey * dly + dL_dz * h.sy);
                        atomic_add(o + 8, ex * dkz + ey * dlz + dL_dz);
                    } else {
                        real dG_dx = -h.G * FILTER_INV_SQUARE * h.dx;
                        real dG_dy = -h.G * FILTER_INV_SQUARE * h.dy;
                        atomic_add(&dL_dxy[2 * g + 0], dL_dG * dG_dx);
                        atomic_add(&dL_dxy[2 * g + 1], dL_dG * dG_dy);
                        atomic_add(&dL_dtransmat[9 * g + 8], dL_dz);
                    }
                }
            }
    }
    return 0;
}

/* ------------------------------------------------------------------ */
/* projection backward                                                 */
/* ------------------------------------------------------------------ */
int surfel_project_backward(int N, int H, int W, real scale_mod, const real* means3D,
                            const real* scales, const real* rots, const real* view,
                            const real* proj, const int32_t* radii,
                            const real* dL_dtransmat, const real* dL_dxy,
                            const real* dL_dnormal, real* dL_dmeans3D, real* dL_dscales,
                            real* dL_drots) {
    real B[3][4];
    pixel_projection(proj, W, H, B);
#pragma omp parallel for schedule(static)
    for (int i = 0; i < N; ++i) {
        for (int k = 0; k < 3; ++k) dL_dmeans3D[3 * i + k] = 0;
        for (int k = 0; k < 2; ++k) dL_dscales[2 * i + k] = 0;
        for (int k = 0; k < 4; ++k) dL_drots[4 * i + k] = 0;
        if (radii[i] <= 0) continue;
        const real* p = means3D + 3 * i;
        real pv[3];
        xform_point(view, p, pv);
        real R[3][3];
        quat_to_R(rots + 4 * i, R);
        real sx = scale_mod * scales[2 * i], sy = scale_mod * scales[2 * i + 1];
        real L0[3] = {R[0][0] * sx, R[1][0] * sx, R[2][0] * sx};
        real L1[3] = {R[0][1] * sy, R[1][1] * sy, R[2][1] * sy};
        real nw[3] = {R[0][2], R[1][2], R[2][2]};
        real A[3][3];
        for (int r = 0; r < 3; ++r) {
            A[r][0] = B[r][0] * L0[0] + B[r][1] * L0[1] + B[r][2] * L0[2];
            A[r][1] = B[r][0] * L1[0] + B[r][1] * L1[1] + B[r][2] * L1[2];
            A[r][2] = B[r][0] * p[0] + B[r][1] * p[1] + B[r][2] * p[2] + B[r][3];
        }
        real n[3];
        xform_vec(view, nw, n);
        real cosv = -(pv[0] * n[0] + pv[1] * n[1] + pv[2] * n[2]);
        real mult = cosv > 0 ? (real)1 : (real)-1;

        /* dL_dtransmat holds the gradient w.r.t. T' = M T (pixel coordinates centred on the
         * splat, see surfel_blend_backward); T' = (M B) X with X = [L0 L1 p; 0 0 1], so the
         * chain rule uses the centred projection Bc = M B, formed from the projection matrix
         * itself: Bc[0] = (W/2) P0 + ((W-1)/2 - cx) P3, never B[0] - cx B[2]. */
        real dA[3][3], dAx[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}};
        for (int r = 0; r < 3; ++r)
            for (int c = 0; c < 3; ++c) dA[r][c] = dL_dtransmat[9 * i + 3 * r + c];
        real cx, cy;
        {
            const real t[3] = {9, 9, -1};
            const real *Tu = A[0], *Tv = A[1], *Tw = A[2];
            real d = t[0] * Tw[0] * Tw[0] + t[1] * Tw[1] * Tw[1] + t[2] * Tw[2] * Tw[2];
            real inv = (real)1 / d;
            real f[3] = {inv * t[0], inv * t[1], inv * t[2]};
            cx = f[0] * Tu[0] * Tw[0] + f[1] * Tu[1] * Tw[1] + f[2] * Tu[2] * Tw[2];
            cy = f[0] * Tv[0] * Tw[0] + f[1] * Tv[1] * Tw[1] + f[2] * Tv[2] * Tw[2];
            /* gradient through the bounding-box centre (low-pass-filter branch), w.r.t. T */
            real gx_ = dL_dxy[2 * i], gy_ = dL_dxy[2 * i + 1];
            if (gx_ != 0 || gy_ != 0) {
                for (int k = 0; k < 3; ++k) {
                    dAx[0][k] = gx_ * f[k] * Tw[k];
                    dAx[1][k] = gy_ * f[k] * Tw[k];
                    dAx[2][k] = gx_ * (f[k] * Tu[k] - 2 * cx * f[k] * Tw[k]) +
                                gy_ * (f[k] * Tv[k] - 2 * cy * f[k] * Tw[k]);
                }
            }
        }
        /* the point the blend pass centred on: (cx, cy) clamped into the image */
        const real ccx = rmin(rmax(cx, 0), (real)(W - 1)), ccy = rmin(rmax(cy, 0), (real)(H - 1));
        real Bc[3][4];
        {
            const real hw = (real)W / 2, hh = (real)H / 2;
            const real ow = (real)(W - 1) / 2, oh = (real)(H - 1) / 2;
            for (int c = 0; c < 4; ++c) {
                real p0 = proj[c * 4 + 0], p1 = proj[c * 4 + 1], p3 = proj[c * 4 + 3];
                Bc[0][c] = hw * p0 + (ow - ccx) * p3;
                Bc[1][c] = hh * p1 + (oh - ccy) * p3;
                Bc[2][c] = p3;
            }
        }
        /* A[:,0] = B3*L0, A[:,1] = B3*L1, A[:,2] = B3*p + b */
        real dL0[3], dL1[3], dp[3];
        for (int k = 0; k < 3; ++k) {
            dL0[k] = (Bc[0][k] * dA[0][0] + Bc[1][k] * dA[1][0] + Bc[2][k] * dA[2][0]) +
                     (B[0][k] * dAx[0][0] + B[1][k] * dAx[1][0] + B[2][k] * dAx[2][0]);
            dL1[k] = (Bc[0][k] * dA[0][1] + Bc[1][k] * dA[1][1] + Bc[2][k] * dA[2][1]) +
                     (B[0][k] * dAx[0][1] + B[1][k] * dAx[1][1] + B[2][k] * dAx[2][1]);
            dp[k] = (Bc[0][k] * dA[0][2] + Bc[1][k] * dA[1][2] + Bc[2][k] * dA[2][2]) +
                    (B[0][k] * dAx[0][2] + B[1][k] * dAx[1][2] + B[2][k] * dAx[2][2]);
        }
        /* n = mult * V3 * R[:,2]  ->  dR[:,2] = mult * V3^T dn */
        real dn[3] = {mult * dL_dnormal[3 * i], mult * dL_dnormal[3 * i + 1], mult * dL_dnormal[3 * i + 2]};
        real dnw[3];
        dnw[0] = view[0] * dn[0] + view[1] * dn[1] + view[2] * dn[2];
        dnw[1] = view[4] * dn[0] + view[5] * dn[1] + view[6] * dn[2];
        dnw[2] = view[8] * dn[0] + view[9] * dn[1] + view[10] * dn[2];
        real g[3][3];
        for (int k = 0; k < 3; ++k) {
            g[k][0] = sx * dL0[k];
            g[k][1] = sy * dL1[k];
            g[k][2] = dnw[k];
        }
        dL_dscales[2 * i + 0] = scale_mod * (R[0][0] * dL0[0] + R[1][0] * dL0[1] + R[2][0] * dL0[2]);
        dL_dscales[2 * i + 1] = scale_mod * (R[0][1] * dL1[0] + R[1][1] * dL1[1] + R[2][1] * dL1[2]);
        for (int k = 0; k < 3; ++k) dL_dmeans3D[3 * i + k] = dp[k];
        const real* q = rots + 4 * i;
        real r = q[0], x = q[1], y = q[2], z = q[3];
        dL_drots[4 * i + 0] = 2 * (-z * g[0][1] + y * g[0][2] + z * g[1][0] - x * g[1][2] - y * g[2][0] + x * g[2][1]);
        dL_drots[4 * i + 1] = 2 * (y * g[0][1] + z * g[0][2] + y * g[1][0] - 2 * x * g[1][1] - r * g[1][2] + z * g[2][0] + r * g[2][1] - 2 * x * g[2][2]);
        dL_drots[4 * i + 2] = 2 * (-2 * y * g[0][0] + x * g[0][1] + r * g[0][2] + x * g[1][0] + z * g[1][2] - r * g[2][0] + z * g[2][1] - 2 * y * g[2][2]);
        dL_drots[4 * i + 3] = 2 * (-2 * z * g[0][0] - r * g[0][1] + x * g[0][2] + r * g[1][0] - 2 * z * g[1][1] + y * g[1][2] + x * g[2][0] + y * g[2][1]);
    }
    return 0;
}

/* ------------------------------------------------------------------ */
/* sparse Adam (row-masked, no bias correction)                         */
/* Call site: /root/reference/scripts/gaussian/gaussian_base.py:62,402  */
/* ------------------------------------------------------------------ */
int sparse_adam(int N, int M, real* param, const real* grad, real* exp_avg,
                real* exp_avg_sq, const uint8_t* visible, real lr, real b1, real b2,
                real eps) {
#pragma omp parallel for schedule(static)
    for (int i = 0; i < N; ++i) {
        if (!visible[i]) continue;
        for (int k = 0; k < M; ++k) {
            size_t j = (size_t)i * M + k;
            real gr = grad[j];
            real m = b1 * exp_avg[j] + ((real)1 - b1) * gr;
            real v = b2 * exp_avg_sq[j] + ((real)1 - b2) * gr * gr;
            param[j] += -lr * m / (R_SQRT(v) + eps);
            exp_avg[j] = m;
            exp_avg_sq[j] = v;
        }
    }
    return 0;
}

int oracle_real_bytes(void) { return (int)sizeof(real); }
