// Dense bundle adjustment (DBA) core for gfx950.
//
// One Gauss-Newton iteration =
//   index      : depth variables kx = unique(arange(t0,t1) U ii), per-variable
//                edge lists and (row, pose) entry lists          (1 workgroup)
//   linearize  : fused reprojection + Jacobians + accumulation BY SOURCE
//                FRAME: a workgroup owns 256 pixels of one depth variable and
//                walks all edges leaving that frame, so the depth block C, the
//                depth residual w and the pose/depth coupling of the source
//                pose are reduced in registers (no per-edge intermediates, no
//                scatter pass); the 6x6 pose blocks and right-hand sides of an
//                edge are J^T W J products on the matrix cores
//                (v_mfma_f64_16x16x4_f64 over the wave's 64 pixels), added to
//                a float64 accumulator
//   schur      : S = E Q E^T and u = E Q w as an MFMA block GEMM
//                (v_mfma_f64_16x16x4_f64) over the pixel axis, block sparsity
//                taken from the entry lists
//   solve      : damped Cholesky of H - S in one workgroup (LDS)
//   retract    : dz = Q (w - E^T dx), inverse depths and poses updated in place
//
// Everything after the float32 inputs is float64.  Monocular BA is gauge
// deficient (scale): with ep = 0.1 against pose-block entries of ~1e4 the
// reduced system amplifies evaluation noise by ~1e5, so float32 Jacobians
// cannot deliver pose updates to 1e-4 relative; CDNA4's full-rate float64
// vector and matrix units make the float64 path affordable (the whole step
// is a few hundred MB of traffic and < 1 GFLOP).
//
// Serves droid_backends.ba / BACore as called at
// /root/reference/scripts/frontend/depth_video.py:380-381,475-481,550-568,649-650;
// the mathematics follow /root/reference/scripts/frontend/geom/ba.py:29-104,
// chol.py:32-73 and projective_ops.py:18-125.
#include <cstdlib>

#include "common.hpp"

namespace vings {

constexpr double kMinDepth = 0.2;
// H, v, S and u are accumulated as kHCopies partial sums, a workgroup of the linearisation or
// of the Schur kernel adding to copy (its index mod kHCopies): every element is the target of
// one float64 atomic per workgroup and edge (or depth variable), a few hundred per element and
// launch, and atomics on one address are served one after the other (linearisation 85 -> 65 us
// when the copies came in).  The readers add the copies up (h_sum, v_sum, s_sum, u_sum).
constexpr int kHCopies = 8;
constexpr double kSensorAlpha = 0.05;
constexpr int kMaxFrames = 4096;

struct DbaWs {  // typed view of the caller's workspace
    double* E;       // [(P+E),6,HW]
    double* Q;       // [M,HW]
    double* w;       // [M,HW]
    double* H;       // [kHCopies][6P,6P] partial sums; lower triangle (row >= column), the rest stays zero
    double* v;       // [kHCopies][6P] partial sums
    double* S;       // [kHCopies][6P,6P] partial sums of the lower triangle, like H
    double* u;       // [kHCopies][6P] partial sums
    double* dx;      // [6P]
    double* chol;    // [6P,6P]
    int* kx;         // [M]
    int* kk;         // [E]
    int* kstart;     // [M+1]
    int* kedges;     // [E]
    int* ip;         // [E]
    int* jp;         // [E]
    int* ent_start;  // [M+1]
    int* ent_row;    // [P+E]
    int* ent_pose;   // [P+E]
    int* status;     // [4]
};

static size_t up256(size_t x) { return (x + 255) & ~(size_t)255; }

static void layout(int ht, int wd, int E, int P, int M, vings_dba_layout* L) {
    const size_t HW = (size_t)ht * wd;
    const size_t n = 6 * (size_t)P;
    size_t o = 0;
    auto take = [&](size_t bytes) { size_t r = o; o += up256(bytes); return r; };
    L->off_E = take(sizeof(double) * (size_t)(P + E) * 6 * HW);
    L->off_Q = take(sizeof(double) * (size_t)M * HW);
    L->off_w = take(sizeof(double) * (size_t)M * HW);
    L->off_H = take(sizeof(double) * n * n * kHCopies);
    L->off_v = take(sizeof(double) * n * kHCopies);
    L->off_S = take(sizeof(double) * n * n * kHCopies);
    L->off_u = take(sizeof(double) * n * kHCopies);
    L->off_dx = take(sizeof(double) * n);
    L->off_chol = take(sizeof(double) * n * n);
    L->off_kx = take(sizeof(int) * (size_t)(M + 1));
    L->off_kk = take(sizeof(int) * (size_t)(E + 1));
    L->off_kstart = take(sizeof(int) * (size_t)(M + 2));
    L->off_kedges = take(sizeof(int) * (size_t)(E + 1));
    L->off_ip = take(sizeof(int) * (size_t)(E + 1));
    L->off_jp = take(sizeof(int) * (size_t)(E + 1));
    L->off_ent_start = take(sizeof(int) * (size_t)(M + 2));
    L->off_ent_row = take(sizeof(int) * (size_t)(P + E + 1));
    L->off_ent_pose = take(sizeof(int) * (size_t)(P + E + 1));
    L->off_status = take(sizeof(int) * 4);
    L->total_bytes = o;
}

static DbaWs carve(void* ws, const vings_dba_layout& L) {
    char* p = (char*)ws;
    DbaWs W;
    W.E = (double*)(p + L.off_E); W.Q = (double*)(p + L.off_Q); W.w = (double*)(p + L.off_w);
    W.H = (double*)(p + L.off_H); W.v = (double*)(p + L.off_v); W.S = (double*)(p + L.off_S);
    W.u = (double*)(p + L.off_u); W.dx = (double*)(p + L.off_dx); W.chol = (double*)(p + L.off_chol);
    W.kx = (int*)(p + L.off_kx); W.kk = (int*)(p + L.off_kk); W.kstart = (int*)(p + L.off_kstart);
    W.kedges = (int*)(p + L.off_kedges); W.ip = (int*)(p + L.off_ip); W.jp = (int*)(p + L.off_jp);
    W.ent_start = (int*)(p + L.off_ent_start); W.ent_row = (int*)(p + L.off_ent_row);
    W.ent_pose = (int*)(p + L.off_ent_pose); W.status = (int*)(p + L.off_status);
    return W;
}

__device__ __forceinline__ double h_sum(const DbaWs& W, int n, int lo) {
    double a = 0.0;
#pragma unroll
    for (int c = 0; c < kHCopies; ++c) a += W.H[(size_t)c * n * n + lo];
    return a;
}
__device__ __forceinline__ double s_sum(const DbaWs& W, int n, int lo) {
    double a = 0.0;
#pragma unroll
    for (int c = 0; c < kHCopies; ++c) a += W.S[(size_t)c * n * n + lo];
    return a;
}
__device__ __forceinline__ double u_sum(const DbaWs& W, int n, int i) {
    double a = 0.0;
#pragma unroll
    for (int c = 0; c < kHCopies; ++c) a += W.u[(size_t)c * n + i];
    return a;
}
__device__ __forceinline__ double v_sum(const DbaWs& W, int n, int i) {
    double a = 0.0;
#pragma unroll
    for (int c = 0; c < kHCopies; ++c) a += W.v[(size_t)c * n + i];
    return a;
}

// ---------------------------------------------------------------------------
// SE3 helpers, quaternion (x,y,z,w)
// ---------------------------------------------------------------------------
struct Se3d { double t[3]; double q[4]; };

__device__ __forceinline__ void qrot(const double* q, const double* v, double* o) {
    const double x = q[0], y = q[1], z = q[2], w = q[3];
    const double tx = 2 * (y * v[2] - z * v[1]), ty = 2 * (z * v[0] - x * v[2]),
                 tz = 2 * (x * v[1] - y * v[0]);
    o[0] = v[0] + w * tx + (y * tz - z * ty);
    o[1] = v[1] + w * ty + (z * tx - x * tz);
    o[2] = v[2] + w * tz + (x * ty - y * tx);
}
__device__ __forceinline__ void qmul(const double* a, const double* b, double* o) {
    o[0] = a[3] * b[0] + a[0] * b[3] + a[1] * b[2] - a[2] * b[1];
    o[1] = a[3] * b[1] - a[0] * b[2] + a[1] * b[3] + a[2] * b[0];
    o[2] = a[3] * b[2] + a[0] * b[1] - a[1] * b[0] + a[2] * b[3];
    o[3] = a[3] * b[3] - a[0] * b[0] - a[1] * b[1] - a[2] * b[2];
}
// G_ij = G_j * G_i^-1 ; a self edge is the fixed stereo baseline (projective_ops.py:105)
__device__ __forceinline__ Se3d relative_pose(const float* __restrict__ poses, int i, int j) {
    Se3d G;
    if (i == j) {
        G.t[0] = -0.1; G.t[1] = 0.; G.t[2] = 0.;
        G.q[0] = G.q[1] = G.q[2] = 0.; G.q[3] = 1.;
        return G;
    }
    const float* pi = poses + 7 * i;
    const float* pj = poses + 7 * j;
    const double qi_inv[4] = {-(double)pi[3], -(double)pi[4], -(double)pi[5], (double)pi[6]};
    const double qj[4] = {pj[3], pj[4], pj[5], pj[6]};
    qmul(qj, qi_inv, G.q);
    double rt[3];
    const double ti[3] = {pi[0], pi[1], pi[2]};
    qrot(G.q, ti, rt);
    G.t[0] = (double)pj[0] - rt[0]; G.t[1] = (double)pj[1] - rt[1]; G.t[2] = (double)pj[2] - rt[2];
    return G;
}

// float64 wave sum without LDS traffic: DPP inside the 16-lane rows (each half of the double
// moved separately), v_permlane16_swap / v_permlane32_swap across rows.  The shuffle form
// (two ds_bpermute per step and value) kept the LDS pipe busier than the arithmetic did.
template <int CTRL>
__device__ __forceinline__ double dpp_d(double v) {
    const unsigned long long b = __builtin_bit_cast(unsigned long long, v);
    const int lo = __builtin_amdgcn_update_dpp(0, (int)(unsigned)b, CTRL, 0xf, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp(0, (int)(unsigned)(b >> 32), CTRL, 0xf, 0xf, false);
    return __builtin_bit_cast(double, ((unsigned long long)(unsigned)hi << 32) | (unsigned)lo);
}
// k32: lanes 0-31 receive a[l] + a[l+32], lanes 32-63 receive b[l-32] + b[l];
// otherwise the same for even / odd 16-lane rows.
template <bool k32>
__device__ __forceinline__ double swap_add_d2(double a, double b) {
    const unsigned long long ba = __builtin_bit_cast(unsigned long long, a);
    const unsigned long long bb = __builtin_bit_cast(unsigned long long, b);
    const unsigned alo = (unsigned)ba, ahi = (unsigned)(ba >> 32);
    const unsigned blo = (unsigned)bb, bhi = (unsigned)(bb >> 32);
    unsigned l0, l1, h0, h1;
    if (k32) {
        const auto rl = __builtin_amdgcn_permlane32_swap(alo, blo, false, false);
        const auto rh = __builtin_amdgcn_permlane32_swap(ahi, bhi, false, false);
        l0 = rl[0]; l1 = rl[1]; h0 = rh[0]; h1 = rh[1];
    } else {
        const auto rl = __builtin_amdgcn_permlane16_swap(alo, blo, false, false);
        const auto rh = __builtin_amdgcn_permlane16_swap(ahi, bhi, false, false);
        l0 = rl[0]; l1 = rl[1]; h0 = rh[0]; h1 = rh[1];
    }
    asm volatile("" : "+v"(l0), "+v"(l1), "+v"(h0), "+v"(h1));  // keep the two results apart
    const double x = __builtin_bit_cast(double, ((unsigned long long)h0 << 32) | l0);
    const double y = __builtin_bit_cast(double, ((unsigned long long)h1 << 32) | l1);
    return x + y;
}
__device__ __forceinline__ double row_sum_d(double v) {
    v += dpp_d<0xB1>(v);    // quad_perm [1,0,3,2]
    v += dpp_d<0x4E>(v);    // quad_perm [2,3,0,1]
    v += dpp_d<0x141>(v);   // row_half_mirror
    v += dpp_d<0x140>(v);   // row_mirror
    return v;
}
template <bool k32>
__device__ __forceinline__ double swap_add_d(double v) {
    const unsigned long long b = __builtin_bit_cast(unsigned long long, v);
    const unsigned lo = (unsigned)b, hi = (unsigned)(b >> 32);
    unsigned l0, l1, h0, h1;
    if (k32) {
        const auto rl = __builtin_amdgcn_permlane32_swap(lo, lo, false, false);
        const auto rh = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
        l0 = rl[0]; l1 = rl[1]; h0 = rh[0]; h1 = rh[1];
    } else {
        const auto rl = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
        const auto rh = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
        l0 = rl[0]; l1 = rl[1]; h0 = rh[0]; h1 = rh[1];
    }
    asm volatile("" : "+v"(l0), "+v"(l1), "+v"(h0), "+v"(h1));  // keep the two results apart
    const double x = __builtin_bit_cast(double, ((unsigned long long)h0 << 32) | l0);
    const double y = __builtin_bit_cast(double, ((unsigned long long)h1 << 32) | l1);
    return x + y;
}
__device__ __forceinline__ double wave_sum_f64(double v) {
    v += dpp_d<0xB1>(v);    // quad_perm [1,0,3,2]
    v += dpp_d<0x4E>(v);    // quad_perm [2,3,0,1]
    v += dpp_d<0x141>(v);   // row_half_mirror
    v += dpp_d<0x140>(v);   // row_mirror
    v = swap_add_d<false>(v);
    v = swap_add_d<true>(v);
    return v;
}

// ---------------------------------------------------------------------------
// index kernel
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void dba_index_kernel(const int64_t* __restrict__ ii,
                                                        const int64_t* __restrict__ jj, int B, int E,
                                                        int M, int t0, int t1, DbaWs W) {
    __shared__ int flag[kMaxFrames];
    __shared__ int pos[kMaxFrames];
    __shared__ int bad;
    // endpoints of the first kLdsEdges edges: the list building below is a serial double loop
    // of one lane, which out of global memory costs a cache round trip per step
    constexpr int kLdsEdges = 2048;
    __shared__ short sip[kLdsEdges], sjp[kLdsEdges];
    const int tid = threadIdx.x;
    if (tid == 0) bad = 0;
    for (int f = tid; f < B; f += 256) flag[f] = (f >= t0 && f < t1) ? 1 : 0;
    __syncthreads();
    for (int e = tid; e < E; e += 256) {
        const long a = ii[e], b = jj[e];
        if (a < 0 || a >= B || b < 0 || b >= B) { bad = 1; W.ip[e] = 0; W.jp[e] = 0; continue; }
        W.ip[e] = (int)a; W.jp[e] = (int)b;
        if (e < kLdsEdges) { sip[e] = (short)a; sjp[e] = (short)b; }
        flag[a] = 1;
    }
    __syncthreads();
    if (tid == 0) {
        int c = 0;
        for (int f = 0; f < B; ++f) {
            pos[f] = c;
            if (flag[f]) { if (c < M) W.kx[c] = f; ++c; }
        }
        W.status[0] = c;                 // number of depth variables found
        W.status[1] = 0;                 // cholesky failure flag
        W.status[2] = bad;               // index out of range / eta row count mismatch
        if (c != M) bad = 1;
    }
    __syncthreads();
    if (bad) {  // leave empty lists: every later kernel becomes a no-op
        for (int k = tid; k <= M; k += 256) { W.kstart[k] = 0; W.ent_start[k] = 0; }
        if (tid == 0) W.status[2] = 1;
        return;
    }
    for (int e = tid; e < E; e += 256) W.kk[e] = pos[W.ip[e]];
    __syncthreads();
    if (E <= kLdsEdges) {
        // The lists in parallel.  An edge's place inside its source frame's lists is its rank
        // among the earlier edges of that frame (all edges / those whose target pose is free),
        // counted by its own lane over the endpoints in LDS; the list starts are prefix sums
        // of the per-frame counts.  Same lists as the serial loop below, entry for entry.
        __shared__ int cnt_free[kMaxFrames];
        int* cnt_all = flag;                         // (the frame flags have been consumed)
        for (int k = tid; k < M; k += 256) { cnt_all[k] = 0; cnt_free[k] = 0; }
        __syncthreads();
        for (int e = tid; e < E; e += 256) {
            const int k = pos[sip[e]];
            atomicAdd(&cnt_all[k], 1);
            if (sjp[e] >= t0 && sjp[e] < t1) atomicAdd(&cnt_free[k], 1);
        }
        __syncthreads();
        if (tid == 0) {
            int ce = 0, cn = 0;
            for (int k = 0; k < M; ++k) {
                W.kstart[k] = ce;
                W.ent_start[k] = cn;
                const int f = W.kx[k];
                ce += cnt_all[k];
                cn += cnt_free[k] + ((f >= t0 && f < t1) ? 1 : 0);
            }
            W.kstart[M] = ce;
            W.ent_start[M] = cn;
        }
        __syncthreads();
        for (int k = tid; k < M; k += 256) {
            const int f = W.kx[k];
            if (f >= t0 && f < t1) { const int n = W.ent_start[k]; W.ent_row[n] = f - t0; W.ent_pose[n] = f - t0; }
        }
        for (int e = tid; e < E; e += 256) {
            const int src = sip[e], j = sjp[e];
            int r_all = 0, r_free = 0;
            for (int q = 0; q < e; ++q) {
                if (sip[q] != src) continue;
                ++r_all;
                r_free += (sjp[q] >= t0 && sjp[q] < t1) ? 1 : 0;
            }
            const int k = pos[src];
            W.kedges[W.kstart[k] + r_all] = e;
            if (j >= t0 && j < t1) {
                const int n = W.ent_start[k] + ((src >= t0 && src < t1) ? 1 : 0) + r_free;
                W.ent_row[n] = (t1 - t0) + e;
                W.ent_pose[n] = j - t0;
            }
        }
        return;
    }
    // per depth variable: edge list (in edge order) and entry list
    if (tid == 0) {
        int ce = 0, cn = 0;
        for (int k = 0; k < M; ++k) {
            W.kstart[k] = ce;
            W.ent_start[k] = cn;
            const int f = W.kx[k];
            if (f >= t0 && f < t1) { W.ent_row[cn] = f - t0; W.ent_pose[cn] = f - t0; ++cn; }
            for (int e = 0; e < E; ++e) {
                if ((e < kLdsEdges ? (int)sip[e] : W.ip[e]) != f) continue;
                W.kedges[ce++] = e;
                const int j = e < kLdsEdges ? (int)sjp[e] : W.jp[e];
                if (j >= t0 && j < t1) { W.ent_row[cn] = (t1 - t0) + e; W.ent_pose[cn] = j - t0; ++cn; }
            }
        }
        W.kstart[M] = ce;
        W.ent_start[M] = cn;
    }
}

// ---------------------------------------------------------------------------
// linearize
// ---------------------------------------------------------------------------
using f64x4 = __attribute__((ext_vector_type(4))) double;

__device__ __forceinline__ constexpr int tri(int a, int b) {  // (a<=b) in a 6x6 upper triangle
    return a * 6 - a * (a - 1) / 2 + (b - a);
}

__global__ __launch_bounds__(256) void dba_linearize_kernel(
    const float* __restrict__ poses, const float* __restrict__ disps,
    const float* __restrict__ intr, const float* __restrict__ disps_sens,
    const float* __restrict__ target, const float* __restrict__ weight,
    const float* __restrict__ eta, int ht, int wd, int E, int M, int t0, int t1, int motion_only,
    DbaWs W) {
    constexpr int kTileLd = 17;
    __shared__ double tile[4][64 * kTileLd];   // per wave: its pixels' 13-vectors (+ 3 zero rows)
    __shared__ double Cs[2][4][13 * 13];       // the waves' products, double-buffered over the edges
    const int k = blockIdx.y;
    const int HW = ht * wd;
    const int P = t1 - t0;
    const int p = blockIdx.x * 256 + threadIdx.x;
    const bool act = p < HW;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int e0 = W.kstart[k], e1 = W.kstart[k + 1];
    if (W.status[2]) return;
    if (motion_only && e0 == e1) return;
    const int i = W.kx[k];
    const double fx = intr[0], fy = intr[1], cx = intr[2], cy = intr[3];
    const int pu = act ? p % wd : 0, pvv = act ? p / wd : 0;
    const double d = act ? (double)disps[(size_t)i * HW + p] : 1.0;
    const double X0[3] = {((double)pu - cx) / fx, ((double)pvv - cy) / fy, 1.0};
    const bool i_free = i >= t0 && i < t1;
    double Cacc = 0., wacc = 0., Ei[6] = {0, 0, 0, 0, 0, 0};
    // the entry of the 13 x 13 product this lane adds up over the four waves (lanes < 169)
    const int ca = threadIdx.x / 13, cb = threadIdx.x - ca * 13;
    const bool own_block = threadIdx.x < 13 * 13 && ca < 6 && (cb < 6 || cb == 12);
    const int n6 = 6 * P, ib = (i - t0) * 6;
    double* const Hmine = W.H + (size_t)(blockIdx.x % kHCopies) * n6 * n6;
    double* const vmine = W.v + (size_t)(blockIdx.x % kHCopies) * n6;
    double carry = 0.0;
    int buf = 0;
    // rows 13..15 of the MFMA operands stay zero for the whole kernel
    tile[wv][lane * kTileLd + 13] = 0.; tile[wv][lane * kTileLd + 14] = 0.;
    tile[wv][lane * kTileLd + 15] = 0.; tile[wv][lane * kTileLd + 16] = 0.;

    // The edges leaving this frame, their target frames and relative poses, once per workgroup:
    // read per edge inside the loop they are three dependent memory round trips (edge list ->
    // target frame -> its pose) in front of every edge's arithmetic.
    constexpr int kEdgeCache = 32;
    __shared__ int s_e[kEdgeCache], s_j[kEdgeCache];
    __shared__ double s_G[kEdgeCache][8];
    if ((int)threadIdx.x < min(e1 - e0, kEdgeCache)) {
        const int e = W.kedges[e0 + threadIdx.x];
        const int j = W.jp[e];
        const Se3d G = relative_pose(poses, i, j);
        s_e[threadIdx.x] = e; s_j[threadIdx.x] = j;
#pragma unroll
        for (int c = 0; c < 3; ++c) s_G[threadIdx.x][c] = G.t[c];
#pragma unroll
        for (int c = 0; c < 4; ++c) s_G[threadIdx.x][3 + c] = G.q[c];
    }
    __syncthreads();
    // this pixel's flow target and weights of an edge; the next edge's are fetched while the
    // current edge is worked on
    auto fetch = [&](int idx, float (&tw)[4]) {
        tw[0] = tw[1] = tw[2] = tw[3] = 0.f;
        if (idx >= e1 || !act) return;
        const int e = idx - e0 < kEdgeCache ? s_e[idx - e0] : W.kedges[idx];
        const size_t o = ((size_t)e * 2) * HW + p;
        tw[0] = target[o]; tw[1] = target[o + HW]; tw[2] = weight[o]; tw[3] = weight[o + HW];
    };
    float tw_next[4];
    const int ne = e1 - e0;
    const int rot = ne > 0 ? (int)(blockIdx.x % (unsigned)ne) : 0;
    auto edge_at = [&](int t) { int q = t + rot; if (q >= ne) q -= ne; return e0 + q; };
    fetch(ne > 0 ? edge_at(0) : e1, tw_next);
    for (int t = 0; t < ne; ++t) {
        const int idx = edge_at(t);
        const bool cached = idx - e0 < kEdgeCache;
        const int e = cached ? s_e[idx - e0] : W.kedges[idx];
        const int j = cached ? s_j[idx - e0] : W.jp[e];
        const bool j_free = j >= t0 && j < t1;
        Se3d G;
        if (cached) {
#pragma unroll
            for (int c = 0; c < 3; ++c) G.t[c] = s_G[idx - e0][c];
#pragma unroll
            for (int c = 0; c < 4; ++c) G.q[c] = s_G[idx - e0][3 + c];
        } else {
            G = relative_pose(poses, i, j);
        }
        float tw[4] = {tw_next[0], tw_next[1], tw_next[2], tw_next[3]};
        fetch(t + 1 < ne ? edge_at(t + 1) : e1, tw_next);
        double X1[3];
        qrot(G.q, X0, X1);
        X1[0] += G.t[0] * d; X1[1] += G.t[1] * d; X1[2] += G.t[2] * d;
        const double Z = X1[2];
        const bool valid = act && Z > kMinDepth;
        const double dz = 1.0 / (Z < 0.5 * kMinDepth ? 1.0 : Z);
        const double xn = X1[0] * dz, yn = X1[1] * dz;
        double r[2] = {0, 0}, wt[2] = {0, 0};
        if (valid) {
            r[0] = (double)tw[0] - (fx * xn + cx);
            r[1] = (double)tw[1] - (fy * yn + cy);
            wt[0] = 0.001 * (double)tw[2];
            wt[1] = 0.001 * (double)tw[3];
        }
        // rows of d(coords)/d(xi_j), twist ordered (tau, phi)
        double Jj[2][6], Ji[2][6], Jz[2];
        const double h = d;
        Jj[0][0] = fx * h * dz; Jj[0][1] = 0.; Jj[0][2] = -fx * xn * h * dz;
        Jj[0][3] = -fx * xn * yn; Jj[0][4] = fx * (1. + xn * xn); Jj[0][5] = -fx * yn;
        Jj[1][0] = 0.; Jj[1][1] = fy * h * dz; Jj[1][2] = -fy * yn * h * dz;
        Jj[1][3] = -fy * (1. + yn * yn); Jj[1][4] = fy * xn * yn; Jj[1][5] = fy * xn;
        Jz[0] = fx * (G.t[0] * dz - G.t[2] * xn * dz);
        Jz[1] = fy * (G.t[1] * dz - G.t[2] * yn * dz);
        // Ji = -Jj * Ad(G): for a row (a | b): tau part R^T a, phi part R^T (b + a x t)
        const double qinv[4] = {-G.q[0], -G.q[1], -G.q[2], G.q[3]};
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            const double a[3] = {Jj[c][0], Jj[c][1], Jj[c][2]};
            const double b[3] = {Jj[c][3], Jj[c][4], Jj[c][5]};
            double ra[3], rb[3];
            qrot(qinv, a, ra);
            const double s[3] = {b[0] + (a[1] * G.t[2] - a[2] * G.t[1]),
                                 b[1] + (a[2] * G.t[0] - a[0] * G.t[2]),
                                 b[2] + (a[0] * G.t[1] - a[1] * G.t[0])};
            qrot(qinv, s, rb);
            Ji[c][0] = -ra[0]; Ji[c][1] = -ra[1]; Ji[c][2] = -ra[2];
            Ji[c][3] = -rb[0]; Ji[c][4] = -rb[1]; Ji[c][5] = -rb[2];
        }
        if (!motion_only) {
            Cacc += wt[0] * Jz[0] * Jz[0] + wt[1] * Jz[1] * Jz[1];
            wacc += wt[0] * r[0] * Jz[0] + wt[1] * r[1] * Jz[1];
#pragma unroll
            for (int m = 0; m < 6; ++m) {
                Ei[m] += wt[0] * Ji[0][m] * Jz[0] + wt[1] * Ji[1][m] * Jz[1];
                if (act && j_free)
                    W.E[((size_t)(P + e) * 6 + m) * HW + p] =
                        wt[0] * Jj[0][m] * Jz[0] + wt[1] * Jj[1][m] * Jz[1];
            }
        }
        if (!i_free && !j_free) continue;  // both poses fixed: nothing for the pose block
        // J^T W J of this edge on the matrix cores.  Per residual row c the lane's pixel has the
        // 13-vector a = (Ji[c][0..5], Jj[c][0..5], r[c]); the 13 x 13 matrix sum_pixels wt a a^T
        // holds Hii, Hij, Hjj, vi and vj.  v_mfma_f64_16x16x4_f64 takes four pixels at a time:
        // lane l supplies A[row l&15][pixel l>>4] = sqrt(wt) * a and the same element as B, so
        // every lane reads ONE value per instruction from a per-wave LDS tile that the lanes
        // filled with their own vectors (pixel-major rows of 17 doubles: the 16 rows of an operand are
        // consecutive words, the four pixels of it 34 banks apart).  16 instructions per
        // residual row replace 90 products and the 90-value wave reduction that followed them.
        {
            double* __restrict__ T = tile[wv];
            f64x4 acc = {0., 0., 0., 0.};
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                // (the weight goes in as its square root on both sides: one operand, one LDS
                // read per instruction, and no per-instruction multiply)
                const double sw = sqrt(wt[c]);
                double* mine = T + lane * kTileLd;
#pragma unroll
                for (int m = 0; m < 6; ++m) { mine[m] = sw * Ji[c][m]; mine[6 + m] = sw * Jj[c][m]; }
                mine[12] = sw * r[c];
                // same wave writes and reads: LDS operations of a wave complete in order
                __builtin_amdgcn_wave_barrier();
                const double* src = T + (lane >> 4) * kTileLd + (lane & 15);
#pragma unroll
                for (int m = 0; m < 16; ++m) {
                    const double a = src[4 * m * kTileLd];
                    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, a, acc, 0, 0, 0);
                }
                __builtin_amdgcn_wave_barrier();
            }
            // C[row (l>>4) + 4q][col l&15] of this wave's 64 pixels
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int row = (lane >> 4) + 4 * q, col = lane & 15;
                if (row < 13 && col < 13) Cs[buf][wv][row * 13 + col] = acc[q];
            }
        }
        // One barrier per edge, and one that waits for LDS only: __syncthreads also waits for
        // every outstanding global store and atomic of the wave (the E rows, the adds into H),
        // a memory round trip per edge.  The four waves' tiles are double-buffered, so one
        // barrier per edge is enough.
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        if (threadIdx.x < 13 * 13) {
            const int o = ca * 13 + cb;
            const double s = (Cs[buf][0][o] + Cs[buf][1][o]) + (Cs[buf][2][o] + Cs[buf][3][o]);
            if (own_block) {
                carry += s;                      // (i, i) block and v_i: the same for every edge
            } else if (s != 0.0 && ca < 12) {
                // rows / columns 0..5: pose i, 6..11: pose j, 12: the residual
                const int jb = (j - t0) * 6;
                const bool a_i = ca < 6, b_i = cb < 6;
                if ((a_i ? i_free : j_free) && (cb == 12 || (b_i ? i_free : j_free))) {
                    const int ra = a_i ? ib + ca : jb + ca - 6;
                    const int rb = b_i ? ib + cb : jb + cb - 6;
                    if (cb == 12) atomicAdd(&vmine[ra], s);
                    else if (ra >= rb) atomicAdd(&Hmine[(size_t)ra * n6 + rb], s);   // lower triangle only
                }
            }
        }
        buf ^= 1;
    }
    if (own_block && i_free && carry != 0.0) {
        if (cb == 12) atomicAdd(&vmine[ib + ca], carry);
        else if (ca >= cb) atomicAdd(&Hmine[(size_t)(ib + ca) * n6 + ib + cb], carry);
    }
    if (motion_only || !act) return;
    if (i_free) {
#pragma unroll
        for (int m = 0; m < 6; ++m) W.E[((size_t)(i - t0) * 6 + m) * HW + p] = Ei[m];
    }
    const size_t fo = (size_t)i * HW + p;
    const double ds = disps_sens ? (double)disps_sens[fo] : 0.;
    const bool sens = ds > 0.;
    const double C = Cacc + (sens ? kSensorAlpha : (double)eta[(size_t)k * HW + p]);
    const double wr = wacc - (sens ? kSensorAlpha * (d - ds) : 0.);
    W.Q[(size_t)k * HW + p] = 1.0 / C;
    W.w[(size_t)k * HW + p] = wr;
}

// ---------------------------------------------------------------------------
// Schur complement as an MFMA block GEMM.
// For depth variable k the rows of Y are its entries' 6 coupling rows plus the
// residual row w; G = (Y diag(Q)) Y^T is accumulated in 16x16 f64 MFMA tiles
// over the pixel axis and added into S and u.
// Operand map of v_mfma_f64_16x16x4_f64: lane l holds A[l&15][l>>4] and
// B[l>>4][l&15]; result reg r of lane l is C[(l>>4) + 4r][l&15].
// ---------------------------------------------------------------------------
constexpr int kSchurWaves = 8;                 // per workgroup
constexpr int kSchurThreads = 64 * kSchurWaves;

// Pixels per wave (a multiple of 16) and workgroups per depth variable: about one workgroup
// per CU (its eight waves are two per SIMD), fewer when that would leave a wave under two
// 16-pixel steps -- every further workgroup costs one more round of atomics on S.
struct SchurGrid { int pxw, gx; };
static SchurGrid schur_grid(int HW, int M) {
    // VINGS_SCHUR_WGS overrides the workgroup target (the CU count of the current device)
    static const int target = [] {
        if (const char* e = getenv("VINGS_SCHUR_WGS")) return atoi(e);
        int dev = 0, cus = 0;
        if (hipGetDevice(&dev) != hipSuccess ||
            hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0)
            cus = 256;
        return cus;
    }();
    int gx = target / (M > 0 ? M : 1);
    gx = min(gx, HW / (32 * kSchurWaves));
    if (gx < 1) gx = 1;
    int pxw = (cdiv(HW, gx * kSchurWaves) + 15) & ~15;
    if (pxw < 16) pxw = 16;
    return SchurGrid{pxw, cdiv(HW, kSchurWaves * pxw)};
}

// G[ra][rc] of depth variable k (rows counted over its entries, the last one w) into S and u;
// a tile above the diagonal also feeds its mirror image
__device__ __forceinline__ void schur_add(const DbaWs& W, int n0, int nrows, int n6, int ra, int rc,
                                          bool mirror, double val) {
    double* const S = W.S + (size_t)(blockIdx.x % kHCopies) * n6 * n6;
    double* const u = W.u + (size_t)(blockIdx.x % kHCopies) * n6;
    if (ra >= nrows - 1 || rc >= nrows || val == 0.0) return;
    const int ea = ra / 6, ca = ra - ea * 6;
    const int ia = W.ent_pose[n0 + ea] * 6 + ca;
    if (rc == nrows - 1) {
        atomicAdd(&u[ia], val);
    } else {
        const int eb = rc / 6, cc = rc - eb * 6;
        const int ibx = W.ent_pose[n0 + eb] * 6 + cc;
        // lower triangle only: of the ordered pair and (from a tile above the diagonal) its
        // mirror image, whichever lands on or below the diagonal; two rows of the same pose and
        // component (a frame that is its own target) put both on the diagonal element
        if (ia >= ibx) atomicAdd(&S[(size_t)ia * n6 + ibx], val);
        if (mirror && ibx >= ia) atomicAdd(&S[(size_t)ibx * n6 + ia], val);
    }
}

// Up to 64 rows per depth variable (R16 <= 4 panels of 16 rows, i.e. up to ten edges leaving a
// frame), no LDS staging and no barrier in the main loop: a wave owns a run of `pxw` pixels and
// ALL upper tiles over it.  Per 16 pixels ("step") lane (r, g) loads, for one row of every
// panel, the two pixel pairs 2g, 2g+1 and 8+2g, 8+2g+1 (16 bytes each; the four g lanes of a row
// cover 64 contiguous bytes per load, each element of E is read exactly once by exactly one
// lane) plus the same pixels of Q, and issues R16 (R16 + 1) / 2 x 4 matrix instructions into as
// many independent accumulators.  The registers of a step are reloaded with the step kDepth
// later as soon as its matrix instructions have issued, so the loads fly while the next steps
// run; the register budget (256) keeps two waves on a SIMD, which covers the rest of the
// latency.  Padding rows of the last panel read the w row: they only reach tile entries that
// are never added anywhere.  A run's ragged last step (the end of the image) goes through a
// guarded copy of the step.
// The eight waves of a workgroup add their tiles up in LDS (ds_add_f64) and the workgroup adds
// the sums into S and u once: a few thousand float64 atomics per workgroup, all workgroups on
// the same few hundred cache lines, which is why a workgroup is made as large as the
// registers allow and the grid is kept near one workgroup per CU.
typedef double f64x2u __attribute__((ext_vector_type(2), aligned(8)));

template <int R16>
__device__ __forceinline__ void schur_rows_body(int HW, int P, int pxw, int k, int n0, int nrows,
                                                const DbaWs& W, double* __restrict__ red) {
    constexpr int NT = R16 * (R16 + 1) / 2;
    constexpr int kDepth = R16 <= 3 ? 3 : 2;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int r = lane & 15, g = lane >> 4;
    const int w0 = min(HW, (blockIdx.x * kSchurWaves + wv) * pxw);
    const int w1 = min(HW, w0 + pxw);
    const int nfull = (w1 - w0) >> 4;      // whole steps; a remainder only at the image's end
    const double* Qk = W.Q + (size_t)k * HW + w0 + 2 * g;
    const double* rq[R16];
#pragma unroll
    for (int b = 0; b < R16; ++b) {
        const int rho = b * 16 + r;
        const double* base = W.w + (size_t)k * HW;
        if (rho < nrows - 1) {
            const int ent = rho / 6, c = rho - ent * 6;
            base = W.E + ((size_t)W.ent_row[n0 + ent] * 6 + c) * HW;
        }
        rq[b] = base + w0 + 2 * g;
    }
    for (int q = threadIdx.x; q < NT * 256; q += kSchurThreads) red[q] = 0.0;
    f64x4 acc[NT];
#pragma unroll
    for (int q = 0; q < NT; ++q) acc[q] = f64x4{0., 0., 0., 0.};
    f64x2u st[kDepth][R16][2], sq[kDepth][2];
    // step t of this run into ring slot u; past the last whole step the last one is read again
    // (never used), so that every load is unconditional and the compiler can count them
    auto fetch = [&](int u, int t) {
        const int off = 16 * min(t, nfull - 1);
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            sq[u][h] = *(const f64x2u*)(Qk + off + 8 * h);
#pragma unroll
            for (int b = 0; b < R16; ++b) st[u][b][h] = *(const f64x2u*)(rq[b] + off + 8 * h);
        }
    };
    auto products = [&](const f64x2u (&y)[R16][2], const f64x2u (&qv)[2]) {
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                double aq[R16];
#pragma unroll
                for (int b = 0; b < R16; ++b) aq[b] = y[b][h][e] * qv[h][e];
                int q = 0;
#pragma unroll
                for (int rb = 0; rb < R16; ++rb)
#pragma unroll
                    for (int cb = rb; cb < R16; ++cb, ++q)
                        acc[q] = __builtin_amdgcn_mfma_f64_16x16x4f64(aq[rb], y[cb][h][e], acc[q], 0, 0, 0);
            }
    };
    if (nfull > 0) {
#pragma unroll
        for (int u = 0; u < kDepth; ++u) fetch(u, u);
        for (int t = 0; t < nfull; t += kDepth) {
#pragma unroll
            for (int u = 0; u < kDepth; ++u) {
                if (t + u >= nfull) break;
                products(st[u], sq[u]);
                fetch(u, t + u + kDepth);
            }
        }
    }
    if (w0 + 16 * nfull < w1) {            // the ragged step
        f64x2u y[R16][2], qv[2];
        const int off = 16 * nfull;
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const int o = off + 8 * h + e;
                const bool in = w0 + 2 * g + o < w1;
                qv[h][e] = in ? Qk[o] : 0.0;
#pragma unroll
                for (int b = 0; b < R16; ++b) y[b][h][e] = in ? rq[b][o] : 0.0;
            }
        products(y, qv);
    }
    __syncthreads();                       // red[] cleared
#pragma unroll
    for (int q = 0; q < NT; ++q)
#pragma unroll
        for (int reg = 0; reg < 4; ++reg)
            if (acc[q][reg] != 0.0) atomicAdd(&red[(q * 4 + reg) * 64 + lane], acc[q][reg]);
    __syncthreads();
    // entry idx of the partial tiles: tile idx >> 8, result register (idx >> 6) & 3, lane idx & 63
    const int n6 = 6 * P;
    for (int idx = threadIdx.x; idx < NT * 256; idx += kSchurThreads) {
        const double val = red[idx];
        int q = idx >> 8, rb = 0;
        while (q >= R16 - rb) { q -= R16 - rb; ++rb; }
        const int cb = rb + q, sl = idx & 63, sreg = (idx >> 6) & 3;
        schur_add(W, n0, nrows, n6, rb * 16 + (sl >> 4) + 4 * sreg, cb * 16 + (sl & 15), rb != cb, val);
    }
}

// More than 64 rows (over ten edges leaving one frame): a wave takes tiles in turn over the
// workgroup's whole pixel span, panels straight from global memory, and adds them to S itself.
__device__ __forceinline__ void schur_wide_body(int HW, int P, int pxw, int k, int n0, int nrows,
                                                const DbaWs& W) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int r = lane & 15, g = lane >> 4;
    const int R16 = (nrows + 15) / 16;
    const int k0 = min(HW, (int)blockIdx.x * kSchurWaves * pxw);
    const int k1 = min(HW, k0 + kSchurWaves * pxw);
    const double* Qk = W.Q + (size_t)k * HW;
    const int n6 = 6 * P;
    auto row_ptr = [&](int rho) -> const double* {
        if (rho >= nrows) return nullptr;
        if (rho == nrows - 1) return W.w + (size_t)k * HW;
        const int ent = rho / 6, c = rho - ent * 6;
        return W.E + ((size_t)W.ent_row[n0 + ent] * 6 + c) * HW;
    };
    for (int t = wv; t < R16 * R16; t += kSchurWaves) {
        const int rb = t / R16, cb = t - rb * R16;
        if (cb < rb) continue;
        f64x4 acc = {0., 0., 0., 0.};
        const double* pa = row_ptr(rb * 16 + r);
        const double* pb = row_ptr(cb * 16 + r);
        for (int kk = k0; kk < k1; kk += 16) {
            const int px = kk + 4 * g;
            double a[4] = {0, 0, 0, 0}, b[4] = {0, 0, 0, 0};
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                if (px + s < k1) {
                    if (pa) a[s] = pa[px + s] * Qk[px + s];
                    if (pb) b[s] = pb[px + s];
                }
            }
#pragma unroll
            for (int s = 0; s < 4; ++s)
                acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a[s], b[s], acc, 0, 0, 0);
        }
#pragma unroll
        for (int reg = 0; reg < 4; ++reg)
            schur_add(W, n0, nrows, n6, rb * 16 + g + 4 * reg, cb * 16 + r, rb != cb, acc[reg]);
    }
}

__global__ __launch_bounds__(kSchurThreads) void dba_schur_kernel(int HW, int P, int pxw, DbaWs W) {
    __shared__ double red[10 * 256];
    const int k = blockIdx.y;
    if (W.status[2]) return;
    const int n0 = W.ent_start[k], n1 = W.ent_start[k + 1];
    const int nent = n1 - n0;
    if (nent == 0) return;
    const int nrows = 6 * nent + 1;
    switch ((nrows + 15) / 16) {
        case 1: schur_rows_body<1>(HW, P, pxw, k, n0, nrows, W, red); break;
        case 2: schur_rows_body<2>(HW, P, pxw, k, n0, nrows, W, red); break;
        case 3: schur_rows_body<3>(HW, P, pxw, k, n0, nrows, W, red); break;
        case 4: schur_rows_body<4>(HW, P, pxw, k, n0, nrows, W, red); break;
        default: schur_wide_body(HW, P, pxw, k, n0, nrows, W); break;
    }
}

// ---------------------------------------------------------------------------
// solve: (H + (ep + lm*diag H) I - S) dx = v - u   in float64, one workgroup.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void dba_solve_kernel(int n, double lm, double ep, int use_schur,
                                                         DbaWs W) {
    extern __shared__ double lds[];
    __shared__ int fail;
    double* A = (n <= 128) ? lds : W.chol;
    const int tid = threadIdx.x;
    if (tid == 0) fail = 0;
    for (int idx = tid; idx < n * n; idx += 1024) {
        const int i = idx / n, j = idx - i * n;
        const int lo = max(i, j) * n + min(i, j);      // H and S hold their lower triangles
        double a = h_sum(W, n, lo);
        if (i == j) a += ep + lm * a;
        if (use_schur) a -= s_sum(W, n, lo);
        A[idx] = a;
    }
    __syncthreads();
    for (int j = 0; j < n; ++j) {
        const double djj = A[j * n + j];
        if (!(djj > 0.0)) { if (tid == 0) fail = 1; }
        __syncthreads();
        if (fail) break;
        const double inv = 1.0 / sqrt(djj);
        __syncthreads();
        for (int i = j + tid; i < n; i += 1024) A[i * n + j] *= inv;   // column j (incl. diagonal)
        __syncthreads();
        const int m = n - j - 1;
        for (int idx = tid; idx < m * m; idx += 1024) {
            const int a = j + 1 + idx / m, b = j + 1 + idx % m;
            if (b <= a) A[a * n + b] -= A[a * n + j] * A[b * n + j];
        }
        __syncthreads();
    }
    if (fail) {  // chol.py:9-17: a failed factorisation yields a zero update
        for (int i = tid; i < n; i += 1024) W.dx[i] = 0.0;
        if (tid == 0) W.status[1] = 1;
        return;
    }
    // forward / backward substitution with the lower factor (serial over rows)
    double* y = W.dx;
    for (int i = tid; i < n; i += 1024) y[i] = v_sum(W, n, i) - (use_schur ? u_sum(W, n, i) : 0.0);
    __syncthreads();
    for (int j = 0; j < n; ++j) {
        if (tid == 0) y[j] /= A[j * n + j];
        __syncthreads();
        const double yj = y[j];
        for (int i = j + 1 + tid; i < n; i += 1024) y[i] -= A[i * n + j] * yj;
        __syncthreads();
    }
    for (int j = n - 1; j >= 0; --j) {
        if (tid == 0) y[j] /= A[j * n + j];
        __syncthreads();
        const double yj = y[j];
        for (int i = tid; i < j; i += 1024) y[i] -= A[j * n + i] * yj;
        __syncthreads();
    }
}

// The same solve for n <= 64 with lane i owning row i: the 1024-lane version spends its time
// in five workgroup barriers per column, which for the 6P = 48..78 unknowns of a real window is
// nearly all of it.  The factorisation runs in block columns of six, so the sums are grouped
// differently from the column-by-column version (results agree to rounding, not bit for bit).
// Rows are padded to an odd number of doubles so that the lanes' row accesses spread over the
// LDS banks.
// 1 / sqrt(d) for d > 0 from v_rsq_f64 (good to about 2^-23) and two Newton steps, a chain
// of 8 dependent instructions; sqrt() followed by a division is about 30, and the
// factorisation below is one such chain per column with little else beside it.
__device__ __forceinline__ double rsqrt_nr(double d) {
    double y = __builtin_amdgcn_rsq(d);
    const double h = 0.5 * d;
    y = y * fma(-h * y, y, 1.5);
    y = y * fma(-h * y, y, 1.5);
    return y;
}

// row r of element idx of a lower triangle stored row by row: the largest r with
// r (r + 1) / 2 <= idx (idx < 2^22: the float estimate is off by at most one)
__device__ __forceinline__ int tri_row(int idx) {
    int r = (int)((__builtin_sqrtf(8.f * (float)idx + 1.f) - 1.f) * 0.5f);
    r += ((r + 1) * (r + 2) / 2 <= idx) ? 1 : 0;
    r -= (r * (r + 1) / 2 > idx) ? 1 : 0;
    return r;
}

__device__ __forceinline__ double lane_bcast(double v, int l) {
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), l);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), l);
    return __hiloint2double(hi, lo);
}

__global__ __launch_bounds__(256) void dba_solve_wave_kernel(int n, double lm, double ep,
                                                            int use_schur, DbaWs W) {
    extern __shared__ double lds[];
    __shared__ double panel[64][8];     // L[i][j0 .. j0+5] of the current block column
    __shared__ double rdiag[64];        // 1 / L[i][i]
    __shared__ int fail_s;
    const int ld = n | 1;
    double* A = lds;
    const int i = threadIdx.x;
    if (i == 0) fail_s = 0;
    // four waves fetch the matrix (one wave alone would queue 36 dependent-latency rounds of
    // global loads)
    // H and S hold their lower triangles as kHCopies partial sums each: an element costs 16
    // loads, so a lane fetches three elements (48 loads in flight) before it adds anything up,
    // and writes each into both halves of A.
    const int nlow = n * (n + 1) / 2;
    for (int base = i; base < nlow; base += 3 * 256) {
        double hv[3][kHCopies], sv[3][kHCopies];
        int rr[3], cc[3];
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            const int idx = min(base + 256 * q, nlow - 1);
            const int r = tri_row(idx);
            rr[q] = r; cc[q] = idx - r * (r + 1) / 2;
            const int lo = r * n + cc[q];
#pragma unroll
            for (int c = 0; c < kHCopies; ++c) {
                hv[q][c] = W.H[(size_t)c * n * n + lo];
                sv[q][c] = use_schur ? W.S[(size_t)c * n * n + lo] : 0.0;
            }
        }
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            if (base + 256 * q >= nlow) break;
            double a = 0.0, sc = 0.0;
#pragma unroll
            for (int c = 0; c < kHCopies; ++c) { a += hv[q][c]; sc += sv[q][c]; }
            if (rr[q] == cc[q]) a += ep + lm * a;
            a -= sc;
            A[rr[q] * ld + cc[q]] = a;
            A[cc[q] * ld + rr[q]] = a;
        }
    }
    __syncthreads();
    // Right-looking factorisation in block columns of six (n = 6 P: one pose per block), two
    // barriers per BLOCK instead of two per column.  Step 1: lane i (row i >= j0) reads the
    // 6 x 6 diagonal block as LDS broadcasts, factors it in registers -- every lane the same
    // 56 flops, no exchange -- and solves its own row of the panel against it.  Step 2: all
    // 256 lanes subtract panel x panel^T from the trailing matrix, one element per lane and
    // round, each element touched once per block instead of once per column.
    for (int j0 = 0; j0 < n; j0 += 6) {
        if (i >= j0 && i < n) {
            double L[6][6], inv[6];
            bool bad = false;
#pragma unroll
            for (int c = 0; c < 6; ++c) {
                double d = A[(j0 + c) * ld + j0 + c];
#pragma unroll
                for (int k = 0; k < c; ++k) d -= L[c][k] * L[c][k];
                bad |= !(d > 0.0);
                inv[c] = rsqrt_nr(d);          // (garbage for d <= 0: the block is flagged bad)
                L[c][c] = d * inv[c];
#pragma unroll
                for (int r = c + 1; r < 6; ++r) {
                    double v = A[(j0 + r) * ld + j0 + c];
#pragma unroll
                    for (int k = 0; k < c; ++k) v -= L[r][k] * L[c][k];
                    L[r][c] = v * inv[c];
                }
            }
            if (bad) fail_s = 1;
            double x[6];
            if (i >= j0 + 6) {
#pragma unroll
                for (int c = 0; c < 6; ++c) {
                    double v = A[i * ld + j0 + c];
#pragma unroll
                    for (int k = 0; k < c; ++k) v -= x[k] * L[c][k];
                    x[c] = v * inv[c];
                }
            } else {
                const int r = i - j0;
                double iv = 0.0;
#pragma unroll
                for (int q = 0; q < 6; ++q) iv = (q == r) ? inv[q] : iv;
                rdiag[i] = iv;
#pragma unroll
                for (int c = 0; c < 6; ++c) {
                    double v = 0.0;
#pragma unroll
                    for (int q = c; q < 6; ++q) v = (q == r) ? L[q][c] : v;
                    x[c] = v;                  // zero right of the diagonal
                }
            }
#pragma unroll
            for (int c = 0; c < 6; ++c) { A[i * ld + j0 + c] = x[c]; panel[i][c] = x[c]; }
        }
        __syncthreads();
        if (fail_s) break;
        const int m0 = j0 + 6, m = n - m0;
        for (int idx = i; idx < m * (m + 1) / 2; idx += 256) {   // the lower triangle only
            const int r = tri_row(idx), c = idx - r * (r + 1) / 2;
            const double* pr = panel[m0 + r];
            const double* pc = panel[m0 + c];
            double acc = 0.0;
#pragma unroll
            for (int k = 0; k < 6; ++k) acc += pr[k] * pc[k];
            A[(m0 + r) * ld + m0 + c] -= acc;
        }
        __syncthreads();
    }
    const bool fail = fail_s != 0;
    if (i >= 64) return;
    const double rd = i < n ? rdiag[i] : 0.0;          // 1 / L[i][i], kept by lane i
    if (fail) {  // chol.py:9-17: a failed factorisation yields a zero update
        if (i < n) W.dx[i] = 0.0;
        if (i == 0) W.status[1] = 1;
        return;
    }
    // Substitutions with the right-hand side in registers (lane i holds y[i]); the pivot
    // element travels by v_readlane and is multiplied by the reciprocal pivot, so a step is
    // a few dependent instructions instead of two LDS round trips, a division and two barriers.
    const bool live = i < n;
    double yi = live ? v_sum(W, n, i) - (use_schur ? u_sum(W, n, i) : 0.0) : 0.0;
    const double* rowi = A + (live ? i : 0) * ld;
    for (int j = 0; j < n; ++j) {                      // L y = b
        const double yj = lane_bcast(yi, j) * lane_bcast(rd, j);
        if (i == j) yi = yj;
        if (live && i > j) yi -= rowi[j] * yj;
    }
    for (int j = n - 1; j >= 0; --j) {                 // L^T x = y
        const double yj = lane_bcast(yi, j) * lane_bcast(rd, j);
        if (i == j) yi = yj;
        if (i < j) yi -= A[j * ld + i] * yj;
    }
    if (live) W.dx[i] = yi;
}

// ---------------------------------------------------------------------------
// retraction
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void dba_disp_retract_kernel(float* __restrict__ disps, int HW,
                                                               DbaWs W) {
    const int k = blockIdx.y;
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p >= HW || W.status[2]) return;
    const int n0 = W.ent_start[k], n1 = W.ent_start[k + 1];
    double acc = 0.;
    for (int n = n0; n < n1; ++n) {
        const double* Er = W.E + (size_t)W.ent_row[n] * 6 * HW + p;
        const double* dxp = W.dx + W.ent_pose[n] * 6;
#pragma unroll
        for (int m = 0; m < 6; ++m) acc += Er[(size_t)m * HW] * dxp[m];
    }
    const size_t o = (size_t)k * HW + p;
    const double dzv = W.Q[o] * (W.w[o] - acc);
    float* dp = disps + (size_t)W.kx[k] * HW + p;
    double nd = (double)*dp + dzv;
    nd = nd > 10. ? 0. : nd;          // ba.py:101-102
    *dp = (float)fmax(nd, 0.);
}

__global__ void dba_pose_retract_kernel(float* __restrict__ poses, int t0, int P, DbaWs W) {
    const int pidx = blockIdx.x * blockDim.x + threadIdx.x;
    if (pidx >= P || W.status[2]) return;
    const double* xi = W.dx + 6 * pidx;
    const double tau[3] = {xi[0], xi[1], xi[2]}, phi[3] = {xi[3], xi[4], xi[5]};
    const double th2 = phi[0] * phi[0] + phi[1] * phi[1] + phi[2] * phi[2];
    const double th = sqrt(th2);
    double a, b, c;
    if (th < 1e-8) { a = 0.5 - th2 / 48.0; b = 0.5 - th2 / 24.0; c = 1.0 / 6 - th2 / 120.0; }
    else { a = sin(0.5 * th) / th; b = (1 - cos(th)) / th2; c = (th - sin(th)) / (th2 * th); }
    const double qe[4] = {a * phi[0], a * phi[1], a * phi[2], cos(0.5 * th)};
    // V tau = tau + b phi x tau + c phi x (phi x tau)
    const double pt[3] = {phi[1] * tau[2] - phi[2] * tau[1], phi[2] * tau[0] - phi[0] * tau[2],
                          phi[0] * tau[1] - phi[1] * tau[0]};
    const double ppt[3] = {phi[1] * pt[2] - phi[2] * pt[1], phi[2] * pt[0] - phi[0] * pt[2],
                           phi[0] * pt[1] - phi[1] * pt[0]};
    const double te[3] = {tau[0] + b * pt[0] + c * ppt[0], tau[1] + b * pt[1] + c * ppt[1],
                          tau[2] + b * pt[2] + c * ppt[2]};
    float* ps = poses + 7 * (t0 + pidx);
    const double t[3] = {ps[0], ps[1], ps[2]}, q[4] = {ps[3], ps[4], ps[5], ps[6]};
    double rt[3], qn[4];
    qrot(qe, t, rt);
    qmul(qe, q, qn);
    const double nrm = 1.0 / sqrt(qn[0] * qn[0] + qn[1] * qn[1] + qn[2] * qn[2] + qn[3] * qn[3]);
    ps[0] = (float)(rt[0] + te[0]); ps[1] = (float)(rt[1] + te[1]); ps[2] = (float)(rt[2] + te[2]);
    ps[3] = (float)(qn[0] * nrm); ps[4] = (float)(qn[1] * nrm); ps[5] = (float)(qn[2] * nrm);
    ps[6] = (float)(qn[3] * nrm);
}

__global__ void dba_reduced_kernel(int n, DbaWs W, double* __restrict__ Hout,
                                   double* __restrict__ vout) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx < n * n) {
        const int r = idx / n, c = idx - r * n;
        const int lo = max(r, c) * n + min(r, c);      // H and S hold their lower triangles
        Hout[idx] = h_sum(W, n, lo) - s_sum(W, n, lo);
    }
    if (idx < n) vout[idx] = v_sum(W, n, idx) - u_sum(W, n, idx);
}

// float32 copies of Q and E for the depth-covariance consumer; rows of E that
// belong to fixed poses / fixed targets were never written and export as zero.
__global__ void dba_export_kernel(int HW, int P, int E, int M, int t0, int t1, DbaWs W,
                                  float* __restrict__ Qout, float* __restrict__ Eout) {
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t nq = (size_t)M * HW, ne = (size_t)(P + E) * 6 * HW;
    if (idx < nq) Qout[idx] = (float)W.Q[idx];
    if (idx < ne) {
        const int row = (int)(idx / ((size_t)6 * HW));
        bool live;
        if (row < P) {
            live = false;
            for (int k = 0; k < M; ++k) live |= (W.kx[k] == row + t0);
        } else {
            const int j = W.jp[row - P];
            live = j >= t0 && j < t1;
        }
        Eout[idx] = live ? (float)W.E[idx] : 0.f;
    }
}

// ---------------------------------------------------------------------------
// host-side stages
// ---------------------------------------------------------------------------
struct DbaDims { int B, ht, wd, E, M, t0, t1; };

static int check_dims(const DbaDims& d) {
    VINGS_REQUIRE(d.B > 0 && d.B <= kMaxFrames, "1 <= frames <= 4096");
    VINGS_REQUIRE(d.ht > 0 && d.wd > 0 && d.E >= 0 && d.M >= 0, "sizes");
    VINGS_REQUIRE(d.t0 >= 0 && d.t1 >= d.t0 && d.t1 <= d.B, "0 <= t0 <= t1 <= frames");
    return VINGS_OK;
}

// build_index = false: the tables of dba_index_kernel (functions of ii, jj, t0, t1 only) are
// already in the workspace from an earlier iteration of the same call
static int stage_linearize(const float* poses, const float* disps, const float* intr,
                           const float* disps_sens, const float* target, const float* weight,
                           const float* eta, const int64_t* ii, const int64_t* jj,
                           const DbaDims& d, int motion_only, const DbaWs& W, hipStream_t s,
                           bool build_index = true) {
    const int P = d.t1 - d.t0, HW = d.ht * d.wd;
    const size_t n = 6 * (size_t)P;
    ProfScope prof(PROF_DBA_JAC, s);
    // H, v, S and u lie next to each other in the workspace: one fill command clears the
    // four accumulators of this iteration (the Schur stage always follows this one)
    VINGS_CHECK_HIP(hipMemsetAsync(W.H, 0, (size_t)((char*)(W.u + n * kHCopies) - (char*)W.H), s));
    if (build_index) {
        dba_index_kernel<<<1, 256, 0, s>>>(ii, jj, d.B, d.E, d.M, d.t0, d.t1, W);
        VINGS_LAUNCH_CHECK();
    }
    if (d.M > 0 && d.E > 0) {
        dim3 grid(cdiv(HW, 256), d.M);
        dba_linearize_kernel<<<grid, 256, 0, s>>>(poses, disps, intr, disps_sens, target, weight,
                                                  eta, d.ht, d.wd, d.E, d.M, d.t0, d.t1,
                                                  motion_only, W);
        VINGS_LAUNCH_CHECK();
    }
    return VINGS_OK;
}

static int stage_schur(const DbaDims& d, const DbaWs& W, hipStream_t s) {
    const int P = d.t1 - d.t0, HW = d.ht * d.wd;
    ProfScope prof(PROF_DBA_SCHUR, s);
    if (d.M > 0 && P > 0) {
        const SchurGrid sg = schur_grid(HW, d.M);
        dba_schur_kernel<<<dim3(sg.gx, d.M), kSchurThreads, 0, s>>>(HW, P, sg.pxw, W);
        VINGS_LAUNCH_CHECK();
    }
    return VINGS_OK;
}

static int stage_solve(const DbaDims& d, float lm, float ep, int motion_only, const DbaWs& W,
                       hipStream_t s) {
    const int n = 6 * (d.t1 - d.t0);
    if (n == 0) return VINGS_OK;
    ProfScope prof(PROF_DBA_SOLVE, s);
    if (n <= 64) {
        const size_t bytes = sizeof(double) * (size_t)n * (n | 1);
        dba_solve_wave_kernel<<<1, 256, bytes, s>>>(n, (double)lm, (double)ep, motion_only ? 0 : 1, W);
        VINGS_LAUNCH_CHECK();
        return VINGS_OK;
    }
    const size_t lds = n <= 128 ? sizeof(double) * (size_t)n * n : 0;
    if (lds > 48 * 1024)
        VINGS_CHECK_HIP(hipFuncSetAttribute((const void*)dba_solve_kernel,
                                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    dba_solve_kernel<<<1, 1024, lds, s>>>(n, (double)lm, (double)ep, motion_only ? 0 : 1, W);
    VINGS_LAUNCH_CHECK();
    return VINGS_OK;
}

static int stage_retract(float* poses, float* disps, const DbaDims& d, int motion_only,
                         const DbaWs& W, hipStream_t s) {
    const int P = d.t1 - d.t0, HW = d.ht * d.wd;
    ProfScope prof(PROF_DBA_RETRACT, s);
    if (!motion_only && d.M > 0) {
        dim3 grid(cdiv(HW, 256), d.M);
        dba_disp_retract_kernel<<<grid, 256, 0, s>>>(disps, HW, W);
        VINGS_LAUNCH_CHECK();
    }
    if (P > 0) {
        dba_pose_retract_kernel<<<cdiv(P, 64), 64, 0, s>>>(poses, d.t0, P, W);
        VINGS_LAUNCH_CHECK();
    }
    return VINGS_OK;
}

}  // namespace vings

using namespace vings;

extern "C" int vings_dba_layout_query(int ht, int wd, int E, int P, int M, vings_dba_layout* out) {
    VINGS_REQUIRE(out && ht > 0 && wd > 0 && E >= 0 && P >= 0 && M >= 0, "layout args");
    layout(ht, wd, E, P, M, out);
    return VINGS_OK;
}

#define DBA_COMMON_CHECKS()                                                                   \
    const DbaDims dims{B, ht, wd, E, M, t0, t1};                                              \
    { int rc_ = check_dims(dims); if (rc_ != VINGS_OK) return rc_; }                         \
    VINGS_REQUIRE(workspace && ((uintptr_t)workspace & 255) == 0, "workspace 256-byte aligned"); \
    vings_dba_layout L;                                                                       \
    layout(ht, wd, E, t1 - t0, M, &L);                                                        \
    if (workspace_bytes < L.total_bytes) {                                                    \
        set_error("dba workspace too small: %zu < %zu", workspace_bytes, L.total_bytes);      \
        return VINGS_ERR_CAPACITY;                                                            \
    }                                                                                         \
    const DbaWs W = carve(workspace, L);                                                      \
    hipStream_t s = (hipStream_t)stream;

extern "C" int vings_dba_linearize(const float* poses, const float* disps, const float* intr,
                                   const float* disps_sens, const float* target,
                                   const float* weight, const float* eta, const int64_t* ii,
                                   const int64_t* jj, int B, int ht, int wd, int E, int M, int t0,
                                   int t1, int motion_only, void* workspace,
                                   size_t workspace_bytes, void* stream) {
    DBA_COMMON_CHECKS();
    VINGS_REQUIRE(poses && disps && intr && (E == 0 || (target && weight && ii && jj)) &&
                      (motion_only || M == 0 || eta),
                  "null pointer");
    int rc = stage_linearize(poses, disps, intr, disps_sens, target, weight, eta, ii, jj, dims,
                             motion_only, W, s);
    if (rc != VINGS_OK) return rc;
    if (!motion_only) rc = stage_schur(dims, W, s);
    return rc;
}

extern "C" int vings_dba_reduced_system(int B, int ht, int wd, int E, int M, int t0, int t1,
                                        void* workspace, size_t workspace_bytes, double* H_out,
                                        double* v_out, void* stream) {
    DBA_COMMON_CHECKS();
    VINGS_REQUIRE(H_out && v_out, "null pointer");
    const int n = 6 * (t1 - t0);
    if (n == 0) return VINGS_OK;
    dba_reduced_kernel<<<cdiv((int64_t)n * n, 256), 256, 0, s>>>(n, W, H_out, v_out);
    VINGS_LAUNCH_CHECK();
    return VINGS_OK;
}

extern "C" int vings_dba_export(int B, int ht, int wd, int E, int M, int t0, int t1,
                                void* workspace, size_t workspace_bytes, float* Q_out,
                                float* E_out, void* stream) {
    DBA_COMMON_CHECKS();
    VINGS_REQUIRE(Q_out && E_out, "null pointer");
    const int HW = ht * wd, P = t1 - t0;
    const size_t n = (size_t)(P + E) * 6 * HW;
    const size_t nq = (size_t)M * HW;
    const size_t total = n > nq ? n : nq;
    if (total == 0) return VINGS_OK;
    dba_export_kernel<<<cdiv((int64_t)total, 256), 256, 0, s>>>(HW, P, E, M, t0, t1, W, Q_out, E_out);
    VINGS_LAUNCH_CHECK();
    return VINGS_OK;
}

extern "C" int vings_dba_retract(float* poses, float* disps, const double* dx, int B, int ht,
                                 int wd, int E, int M, int t0, int t1, int motion_only,
                                 void* workspace, size_t workspace_bytes, void* stream) {
    DBA_COMMON_CHECKS();
    VINGS_REQUIRE(poses && disps, "null pointer");
    if (dx)
        VINGS_CHECK_HIP(hipMemcpyAsync(W.dx, dx, sizeof(double) * 6 * (size_t)(t1 - t0),
                                       hipMemcpyDeviceToDevice, s));
    return stage_retract(poses, disps, dims, motion_only, W, s);
}

extern "C" int vings_dba_ba(float* poses, float* disps, const float* intr,
                            const float* disps_sens, const float* target, const float* weight,
                            const float* eta, const int64_t* ii, const int64_t* jj, int B, int ht,
                            int wd, int E, int M, int t0, int t1, int itrs, float lm, float ep,
                            int motion_only, void* workspace, size_t workspace_bytes,
                            void* stream) {
    DBA_COMMON_CHECKS();
    VINGS_REQUIRE(poses && disps && intr && (E == 0 || (target && weight && ii && jj)) &&
                      (motion_only || M == 0 || eta),
                  "null pointer");
    if (E == 0 || t1 == t0) return VINGS_OK;
    for (int it = 0; it < itrs; ++it) {
        int rc = stage_linearize(poses, disps, intr, disps_sens, target, weight, eta, ii, jj, dims,
                                 motion_only, W, s, it == 0);
        if (rc != VINGS_OK) return rc;
        if (!motion_only) { rc = stage_schur(dims, W, s); if (rc != VINGS_OK) return rc; }
        rc = stage_solve(dims, lm, ep, motion_only, W, s);
        if (rc != VINGS_OK) return rc;
        rc = stage_retract(poses, disps, dims, motion_only, W, s);
        if (rc != VINGS_OK) return rc;
    }
    return VINGS_OK;
}
