// Which 8x8 pixel blocks of a 16x16 tile a splat can reach with alpha >= 1/255, from its packed
// record: shared by the blend passes (surfel_blend.hip, per staged instance) and by the
// projection (surfel_project.hip, which leaves tiles none of whose blocks is reachable out of
// the instance list).  The test is conservative by a margin far above float32 rounding, so the
// two compilations of it may round differently without either dropping a contributing pixel.
#pragma once
#include "common.hpp"

namespace vings {

constexpr int kBlk = 8;  // a wave renders an 8x8 pixel block; four blocks per tile

// Smallest value of the convex quadratic a u^2 + 2b uv + c v^2 + 2d u + 2e v + f over
// the square [0,S]^2: the interior stationary point if it lies inside, else the best of
// the four edge minima.
__device__ __forceinline__ float quad_min_square(float a, float b, float c, float d, float e,
                                                 float f, float inv_det, float inv_a, float inv_c,
                                                 float S) {
    const float us = (b * e - c * d) * inv_det, vs = (b * d - a * e) * inv_det;
    if (us >= 0.f && us <= S && vs >= 0.f && vs <= S) return f + d * us + e * vs;
    const float dS = b * S + d, eS = b * S + e;
    const float u1 = fminf(fmaxf(-d * inv_a, 0.f), S);            // edge v = 0
    const float u2 = fminf(fmaxf(-dS * inv_a, 0.f), S);           // edge v = S
    const float v3 = fminf(fmaxf(-e * inv_c, 0.f), S);            // edge u = 0
    const float v4 = fminf(fmaxf(-eS * inv_c, 0.f), S);           // edge u = S
    const float g1 = (a * u1 + 2.f * d) * u1 + f;
    const float g2 = (a * u2 + 2.f * dS) * u2 + ((c * S + 2.f * e) * S + f);
    const float g3 = (c * v3 + 2.f * e) * v3 + f;
    const float g4 = (c * v4 + 2.f * eS) * v4 + ((a * S + 2.f * d) * S + f);
    return fminf(fminf(g1, g2), fminf(g3, g4));
}

// Which of the tile's four 8x8 blocks can this splat reach with alpha >= 1/255?
// A pixel contributes only where min(rho3d, rho2d) <= rc = 2 ln(255 opacity).
//  * rho2d <= rc is a disc of radius sqrt(rc/2) about the projected centre;
//  * with k = x Tw - Tu, l = y Tw - Tv the ray/splat intersection is p = k x l =
//    x (Tv x Tw) + y (Tw x Tu) + Tu x Tv, LINEAR in the pixel, so rho3d <= rc reads
//    F(x,y) = px^2 + py^2 - rc pz^2 <= 0: a conic, an ellipse (convex F) whenever the
//    splat stays in front of the camera plane.  Its minimum over a block decides exactly.
// rc carries the same safety margin as the record's pixel box (0.2 % + 0.02), far above the
// float32 error of F near the boundary; anything not provably convex falls back to the box.
__device__ __forceinline__ uint32_t block_mask(const float4* r, const float4 bx, float tx0,
                                               float ty0) {
    uint32_t boxm = 0;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        const float x0 = tx0 + (float)(kBlk * (s & 1)), y0 = ty0 + (float)(kBlk * (s >> 1));
        if (!(bx.x > x0 + (kBlk - 1) || bx.y < x0 || bx.z > y0 + (kBlk - 1) || bx.w < y0))
            boxm |= 1u << s;
    }
    if (boxm == 0 || bx.x < -1e29f) return boxm;   // unbounded box: no conic bound either
    const float Tu0 = r[0].x, Tu1 = r[0].y, Tu2 = r[0].z, Tv0 = r[1].x, Tv1 = r[1].y, Tv2 = r[1].z;
    const float Tw0 = r[2].x, Tw1 = r[2].y, Tw2 = r[2].z, cx = r[0].w, cy = r[1].w, o = r[3].z;
    const float rc = 2.f * __logf(255.f * o) * 1.002f + 0.02f;
    const float Ax = Tv1 * Tw2 - Tv2 * Tw1, Ay = Tv2 * Tw0 - Tv0 * Tw2, Az = Tv0 * Tw1 - Tv1 * Tw0;
    const float Bx = Tw1 * Tu2 - Tw2 * Tu1, By = Tw2 * Tu0 - Tw0 * Tu2, Bz = Tw0 * Tu1 - Tw1 * Tu0;
    const float a = Ax * Ax + Ay * Ay - rc * Az * Az;
    const float b = Ax * Bx + Ay * By - rc * Az * Bz;
    const float c = Bx * Bx + By * By - rc * Bz * Bz;
    const float det = a * c - b * b;
    if (!(a > 0.f && c > 0.f && det > 1e-6f * a * c)) return boxm;   // not safely convex
    const float inv_det = 1.0f / det, inv_a = 1.0f / a, inv_c = 1.0f / c;
    const float r2 = 0.5f * rc;   // squared radius of the low-pass disc
    uint32_t m = 0;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        const float x0 = tx0 + (float)(kBlk * (s & 1)), y0 = ty0 + (float)(kBlk * (s >> 1));
        // p at the block origin, formed like the per-pixel evaluation
        const float kx = x0 * Tw0 - Tu0, ky = x0 * Tw1 - Tu1, kz = x0 * Tw2 - Tu2;
        const float lx = y0 * Tw0 - Tv0, ly = y0 * Tw1 - Tv1, lz = y0 * Tw2 - Tv2;
        const float Px = ky * lz - kz * ly, Py = kz * lx - kx * lz, Pz = kx * ly - ky * lx;
        const float d = Ax * Px + Ay * Py - rc * Az * Pz;
        const float e = Bx * Px + By * Py - rc * Bz * Pz;
        const float n2 = Px * Px + Py * Py, z2 = rc * Pz * Pz;
        const float fmin = quad_min_square(a, b, c, d, e, n2 - z2, inv_det, inv_a, inv_c,
                                           (float)(kBlk - 1));
        const float ddx = cx - fminf(fmaxf(cx, x0), x0 + (kBlk - 1));
        const float ddy = cy - fminf(fmaxf(cy, y0), y0 + (kBlk - 1));
        const bool ellipse = !(fmin > 1e-4f * (n2 + z2));   // a NaN never culls
        const bool disc = ddx * ddx + ddy * ddy <= r2;
        if (ellipse || disc) m |= 1u << s;
    }
    return m & boxm;
}

}  // namespace vings
