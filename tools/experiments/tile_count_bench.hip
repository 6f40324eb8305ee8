// Microbenchmark: what does routing the (splat, tile) instances of a 1080p view to per-tile
// counters cost on gfx950?  One million splats with 1..3 x 1..3 tile rectangles at random
// places (about 2.4 instances each, 8160 tiles); the kernel has the structure of the binning
// kernels (instances dealt out over the lanes of the block, owner found by bisection).
// Variants: decode only; device-scope atomics with 1 / 16 / 64 counter replicas per tile;
// workgroup-scope atomics on a replica chosen by the XCD the wave runs on; returning atomics
// plus an 8-byte store to the slot (the scatter).  Every index is in bounds by construction
// and nothing consumes the results except the checks printed at the end.
//   hipcc -O3 --offload-arch=gfx950 tools/experiments/tile_count_bench.hip -o tools/experiments/tile_count_bench
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

constexpr int GX = 120, GY = 68, T = GX * GY;

__device__ __forceinline__ uint32_t wave_scan_incl(uint32_t v) {
    const int l = __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
    for (int o = 1; o < 64; o <<= 1) { uint32_t t = __shfl_up(v, o, 64); if (l >= o) v += t; }
    return v;
}

__device__ __forceinline__ uint32_t xcc_id() {
    uint32_t v;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
    return v & 15u;
}

// mode 0 decode only | 1 device atomic | 2 workgroup-scope atomic, replica = XCD | 3 scatter
// (returning device atomic on the cursor + 8-byte store)
template <int kMode>
__global__ __launch_bounds__(256) void route_kernel(int N, const int4* __restrict__ rect,
                                                    uint32_t* __restrict__ counters, int replicas,
                                                    uint64_t* __restrict__ out, uint32_t cap,
                                                    uint32_t* __restrict__ sink) {
    __shared__ uint32_t wtot[4];
    __shared__ uint32_t start[257];
    __shared__ int4 srect[256];
    const int i = blockIdx.x * 256 + threadIdx.x;
    int4 r = make_int4(0, 0, 0, 0);
    if (i < N) r = rect[i];
    const uint32_t cnt = (uint32_t)((r.z - r.x) * (r.w - r.y));
    const uint32_t inc = wave_scan_incl(cnt);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    if (lane == 63) wtot[w] = inc;
    __syncthreads();
    uint32_t off = inc - cnt;
    for (int k = 0; k < w; ++k) off += wtot[k];
    start[threadIdx.x] = off;
    if (threadIdx.x == 255) start[256] = off + cnt;
    srect[threadIdx.x] = r;
    __syncthreads();
    const uint32_t total = start[256];
    const uint32_t rep = kMode == 2 ? xcc_id() % (uint32_t)replicas : blockIdx.x % (uint32_t)replicas;
    uint32_t acc = 0;
    for (uint32_t q = threadIdx.x; q < total; q += 256) {
        int lo = 0;
#pragma unroll
        for (int step = 128; step > 0; step >>= 1)
            if (start[lo + step] <= q) lo += step;
        const int4 rr = srect[lo];
        const uint32_t idx = q - start[lo];
        const uint32_t wd = (uint32_t)(rr.z - rr.x), row = idx / wd;
        const uint32_t tile = (uint32_t)(rr.y + (int)row) * GX + (uint32_t)rr.x + (idx - row * wd);
        uint32_t* c = counters + (size_t)rep * T + tile;
        if (kMode == 0) acc += tile;
        if (kMode == 1) atomicAdd(c, 1u);
        if (kMode == 2) __hip_atomic_fetch_add(c, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if (kMode == 3) {
            const uint32_t slot = atomicAdd(c, 1u);
            if (slot < cap) out[slot] = ((uint64_t)tile << 32) | (uint32_t)(blockIdx.x * 256 + lo);
        }
    }
    if (kMode == 0 && acc == 0xFFFFFFFFu) sink[0] = acc;
}

int main() {
    const int N = 1000000;
    std::vector<int4> rect(N);
    std::vector<uint32_t> want(T, 0);
    uint64_t D = 0;
    uint32_t s = 12345u;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return s >> 8; };
    auto extent = [&]() { const uint32_t u = rnd() % 100; return u < 55 ? 1 : (u < 90 ? 2 : 3); };
    for (int i = 0; i < N; ++i) {
        const int w = extent(), h = extent();
        const int x0 = (int)(rnd() % (GX - 2)), y0 = (int)(rnd() % (GY - 2));
        rect[i] = make_int4(x0, y0, x0 + w, y0 + h);
        for (int y = y0; y < y0 + h; ++y)
            for (int x = x0; x < x0 + w; ++x) { ++want[y * GX + x]; ++D; }
    }
    printf("splats %d instances %llu tiles %d\n", N, (unsigned long long)D, T);
    const int maxrep = 64;
    int4* d_rect; uint32_t* d_cnt; uint64_t* d_out; uint32_t* d_sink;
    CK(hipMalloc(&d_rect, sizeof(int4) * N));
    CK(hipMalloc(&d_cnt, sizeof(uint32_t) * (size_t)T * maxrep));
    CK(hipMalloc(&d_out, sizeof(uint64_t) * D));
    CK(hipMalloc(&d_sink, 64));
    CK(hipMemcpy(d_rect, rect.data(), sizeof(int4) * N, hipMemcpyHostToDevice));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    std::vector<uint32_t> got((size_t)T * maxrep);
    const int nb = (N + 255) / 256, reps = 20;
    struct Case { const char* name; int mode, replicas; };
    const Case cases[] = {{"decode only", 0, 1}, {"device atomics, 1 replica", 1, 1},
                          {"device atomics, 16 replicas", 1, 16}, {"device atomics, 64 replicas", 1, 64},
                          {"workgroup-scope atomics, replica = XCD (8)", 2, 8},
                          {"scatter: returning device atomic + 8 B store, 1 replica", 3, 1},
                          {"scatter: returning device atomic + 8 B store, 16 replicas", 3, 16}};
    for (const Case& c : cases) {
        float total_ms = 0.f;
        bool ok = true;
        for (int it = 0; it < reps; ++it) {
            // counters: zero for the counting variants; for the scatter the first slot of every
            // (tile, replica) share, computed from the exact counts so that every slot is < D
            if (c.mode == 3) {
                // exact per-replica counts need the block -> replica map: block b uses replica b % R
                std::vector<uint32_t> cnt((size_t)T * c.replicas, 0), st((size_t)T * c.replicas, 0);
                if (it == 0 || true) {
                    for (int i = 0; i < N; ++i) {
                        const int rp = (i / 256) % c.replicas;
                        for (int y = rect[i].y; y < rect[i].w; ++y)
                            for (int x = rect[i].x; x < rect[i].z; ++x) ++cnt[(size_t)rp * T + y * GX + x];
                    }
                    uint32_t run = 0;
                    for (int t = 0; t < T; ++t)
                        for (int rp = 0; rp < c.replicas; ++rp) { st[(size_t)rp * T + t] = run; run += cnt[(size_t)rp * T + t]; }
                }
                CK(hipMemcpy(d_cnt, st.data(), sizeof(uint32_t) * st.size(), hipMemcpyHostToDevice));
            } else {
                CK(hipMemset(d_cnt, 0, sizeof(uint32_t) * (size_t)T * maxrep));
            }
            CK(hipDeviceSynchronize());
            CK(hipEventRecord(e0, 0));
            if (c.mode == 0) route_kernel<0><<<nb, 256>>>(N, d_rect, d_cnt, c.replicas, d_out, (uint32_t)D, d_sink);
            if (c.mode == 1) route_kernel<1><<<nb, 256>>>(N, d_rect, d_cnt, c.replicas, d_out, (uint32_t)D, d_sink);
            if (c.mode == 2) route_kernel<2><<<nb, 256>>>(N, d_rect, d_cnt, c.replicas, d_out, (uint32_t)D, d_sink);
            if (c.mode == 3) route_kernel<3><<<nb, 256>>>(N, d_rect, d_cnt, c.replicas, d_out, (uint32_t)D, d_sink);
            CK(hipGetLastError());
            CK(hipEventRecord(e1, 0));
            CK(hipEventSynchronize(e1));
            float ms = 0.f;
            CK(hipEventElapsedTime(&ms, e0, e1));
            if (it >= 2) total_ms += ms;
            if (it == reps - 1 && (c.mode == 1 || c.mode == 2)) {
                CK(hipMemcpy(got.data(), d_cnt, sizeof(uint32_t) * (size_t)T * c.replicas, hipMemcpyDeviceToHost));
                for (int t = 0; t < T && ok; ++t) {
                    uint32_t sum = 0;
                    for (int rp = 0; rp < c.replicas; ++rp) sum += got[(size_t)rp * T + t];
                    ok = sum == want[t];
                }
            }
        }
        printf("%-62s %8.1f us  %s\n", c.name, total_ms / (reps - 2) * 1e3f,
               (c.mode == 1 || c.mode == 2) ? (ok ? "counts exact" : "COUNTS WRONG") : "");
    }
    return 0;
}
