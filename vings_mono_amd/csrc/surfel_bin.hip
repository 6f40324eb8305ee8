// Tile binning for the surfel rasteriser on gfx950: key emission
// (tile << 32 | float bits of depth), a stable least-significant-digit radix
// sort written for 64-lane wavefronts, and per-tile range extraction.
// While sorting, the depth bits are rebased to the smallest depth of the view
// and the tile index moved down next to them: at 1080p with depths spanning a
// factor of 64 that is 27 + 13 bits, five 8-bit passes instead of six over
// 32 + 13; the order is unchanged and the keys are handed out in the
// documented form.
//
// The sort order is fully determined by (tile, depth bits, splat index):
// each splat emits at most one instance per tile and instances are emitted
// in splat order, so a stable sort on the key reproduces the CPU oracle's
// order bit for bit.
//
// Serves the binning stage of GaussianRasterizer.forward
// (/root/reference/scripts/gaussian/gaussian_base.py:193-203).
#include "common.hpp"

namespace vings {

constexpr int kSortThreads = 256;
constexpr int kSortItems = 8;                        // keys per lane per tile
constexpr int kSortTile = kSortThreads * kSortItems;  // 2048 keys per workgroup
constexpr int kRadixBits = 8;
constexpr int kRadix = 1 << kRadixBits;
constexpr int kMaxPasses = 8;

// ---------------------------------------------------------------------------
// duplicate with keys: lane i owns splat i; its output offset is the
// exclusive prefix of tiles_touched, rebuilt here from the per-block prefix.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void emit_keys_kernel(
    int N, int gx, const float* __restrict__ depth, const int32_t* __restrict__ rect,
    const uint32_t* __restrict__ tiles_touched, const uint32_t* __restrict__ block_prefix,
    uint32_t dmin, int low_bits, uint64_t* __restrict__ keys, uint32_t* __restrict__ vals,
    uint32_t* __restrict__ ranges, int n_range_words, const uint32_t* __restrict__ tail,
    uint32_t cap) {
    // tail != NULL: the launch was sized from a guess; the true {smallest depth bits, largest,
    // instance count} are read here from where the projection left them, and nothing is
    // written beyond the `cap` entries the buffers hold
    if (tail) dmin = tail[0];
    __shared__ uint32_t wtot[4];
    __shared__ uint32_t start[257];     // first output slot of each splat, relative to the block
    __shared__ int4 srect[256];
    __shared__ uint32_t sdepth[256];
    const int i = blockIdx.x * 256 + threadIdx.x;
    // tiles without instances keep the empty range [0,0): cleared here, ahead of
    // tile_ranges_kernel on the same stream, instead of by a separate fill command
    for (int k = i; k < n_range_words; k += gridDim.x * 256) ranges[k] = 0u;
    const uint32_t cnt = i < N ? tiles_touched[i] : 0u;
    const uint32_t inc = wave_scan_incl(cnt);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    if (lane == 63) wtot[w] = inc;
    __syncthreads();
    uint32_t off = inc - cnt;
    for (int k = 0; k < w; ++k) off += wtot[k];
    start[threadIdx.x] = off;
    if (threadIdx.x == 255) start[256] = off + cnt;
    if (cnt) {
        srect[threadIdx.x] = reinterpret_cast<const int4*>(rect)[i];
        // sort key: tile above the depth bits rebased to the smallest depth of the view, so
        // the sort only runs over bits that can differ (same order as tile << 32 | depth bits)
        sdepth[threadIdx.x] = __float_as_uint(depth[i]) - dmin;
    }
    __syncthreads();
    // The block's instances form one contiguous output range; lanes take consecutive slots
    // (coalesced stores) and look up the owning splat by bisection over the 256 starts.
    const uint32_t total = start[256];
    const uint32_t out0 = block_prefix[blockIdx.x];
    for (uint32_t q = threadIdx.x; q < total && out0 + q < cap; q += 256) {
        int lo = 0;                       // largest j with start[j] <= q (splats with cnt = 0
#pragma unroll                            // share their start with the next one and lose)
        for (int step = 128; step > 0; step >>= 1)
            if (start[lo + step] <= q) lo += step;
        const int4 r = srect[lo];
        uint32_t idx = q - start[lo];
        uint32_t tile;
        if (r.x < 0) {
            // packed form: origin, width and one bit per tile of the rectangle (row-major);
            // slot idx is the idx-th set bit
            unsigned long long bits = ((unsigned long long)(uint32_t)r.w << 32) | (uint32_t)r.z;
            for (uint32_t k = 0; k < idx; ++k) bits &= bits - 1ull;
            const uint32_t t = (uint32_t)__builtin_ctzll(bits);
            const uint32_t wd = (uint32_t)r.y, row = t / wd;
            const uint32_t x0 = (uint32_t)r.x & 0x7FFFu, y0 = ((uint32_t)r.x >> 15) & 0xFFFFu;
            tile = (y0 + row) * gx + x0 + (t - row * wd);
        } else {
            const uint32_t wd = (uint32_t)(r.z - r.x);
            const uint32_t row = idx / wd;
            tile = (uint32_t)(r.y + (int)row) * gx + (uint32_t)r.x + (idx - row * wd);
        }
        keys[out0 + q] = ((uint64_t)tile << low_bits) | (uint64_t)sdepth[lo];
        vals[out0 + q] = (uint32_t)(blockIdx.x * 256 + lo);
    }
}

// ---------------------------------------------------------------------------
// radix sort
// ---------------------------------------------------------------------------
__device__ __forceinline__ uint32_t digit_of(uint64_t k, int shift) {
    return (uint32_t)(k >> shift) & (kRadix - 1);
}

// Per-tile digit histogram of one pass: thist[digit][tile].
__global__ __launch_bounds__(kSortThreads) void sort_tile_hist_kernel(
    const uint64_t* __restrict__ keys, int64_t D, int shift, int ntiles,
    uint32_t* __restrict__ thist, const uint32_t* __restrict__ tail) {
    __shared__ uint32_t h[kRadix];
    if (tail) D = min(D, (int64_t)tail[2]);
    h[threadIdx.x] = 0;
    __syncthreads();
    const int64_t base = (int64_t)blockIdx.x * kSortTile;
#pragma unroll
    for (int r = 0; r < kSortItems; ++r) {
        const int64_t i = base + r * kSortThreads + threadIdx.x;
        if (i < D) atomicAdd(&h[digit_of(keys[i], shift)], 1u);
    }
    __syncthreads();
    thist[(size_t)threadIdx.x * ntiles + blockIdx.x] = h[threadIdx.x];
}

// One 256-lane workgroup per digit: exclusive scan of that digit's row of thist (per sort
// tile: keys with this digit in earlier tiles) and the row total into totals[digit]; the
// scatter adds the number of keys with a smaller digit itself.  Each lane owns 8 consecutive
// entries, so rows of up to 2048 sort tiles (4 M keys) take one round.
constexpr int kScanItems = 8;
__global__ __launch_bounds__(256) void sort_scan_kernel(uint32_t* __restrict__ thist, int ntiles,
                                                        uint32_t* __restrict__ totals) {
    __shared__ uint32_t wtot[4];
    __shared__ uint32_t carry_s;
    const int digit = blockIdx.x;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    if (threadIdx.x == 0) carry_s = 0;
    __syncthreads();
    uint32_t* row = thist + (size_t)digit * ntiles;
    for (int t0 = 0; t0 < ntiles; t0 += 256 * kScanItems) {
        const int first = t0 + threadIdx.x * kScanItems;
        uint32_t v[kScanItems], sum = 0;
#pragma unroll
        for (int k = 0; k < kScanItems; ++k) {
            v[k] = first + k < ntiles ? row[first + k] : 0u;
            sum += v[k];
        }
        const uint32_t inc = wave_scan_incl(sum);
        if (lane == 63) wtot[w] = inc;
        __syncthreads();
        uint32_t run = carry_s + inc - sum;
        for (int k = 0; k < w; ++k) run += wtot[k];
#pragma unroll
        for (int k = 0; k < kScanItems; ++k) {
            if (first + k < ntiles) row[first + k] = run;
            run += v[k];
        }
        __syncthreads();
        if (threadIdx.x == 255) carry_s = run;
        __syncthreads();
    }
    if (threadIdx.x == 0) totals[digit] = carry_s;
}

// Stable scatter of one tile.  Keys are held in a wave-striped arrangement
// (item r of lane l of wave w sits at tile_base + w*512 + r*64 + l) so that
// walking r = 0..7 visits each wave's segment in memory order.  Within a
// round the rank of a key among equal digits is the number of lower lanes
// holding the same digit, found with one ballot per digit bit.
__global__ __launch_bounds__(kSortThreads) void sort_scatter_kernel(
    const uint64_t* __restrict__ keys_in, const uint32_t* __restrict__ vals_in,
    uint64_t* __restrict__ keys_out, uint32_t* __restrict__ vals_out, int64_t D, int shift,
    int ntiles, const uint32_t* __restrict__ thist, const uint32_t* __restrict__ totals,
    const uint32_t* __restrict__ tail) {
    __shared__ uint32_t wcount[4][kRadix];  // per-wave digit counts, then prefixes
    if (tail) D = min(D, (int64_t)tail[2]);
    __shared__ uint32_t gbase[kRadix];
    __shared__ uint32_t tile_off[kRadix];
    __shared__ uint32_t wsum4[4];
    __shared__ uint64_t lkey[kSortTile];
    __shared__ uint32_t lval[kSortTile];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    for (int k = threadIdx.x; k < 4 * kRadix; k += kSortThreads) (&wcount[0][0])[k] = 0;
    // first output slot of digit d in this tile = keys with a smaller digit anywhere
    // (exclusive prefix of the 256 digit totals, one digit per lane) + keys with digit d in
    // earlier tiles
    const uint32_t tot = totals[threadIdx.x];
    const uint32_t inc = wave_scan_incl(tot);
    if (lane == 63) wsum4[w] = inc;
    __syncthreads();
    uint32_t below = inc - tot;
    for (int k = 0; k < w; ++k) below += wsum4[k];
    gbase[threadIdx.x] = below + thist[(size_t)threadIdx.x * ntiles + blockIdx.x];
    __syncthreads();

    const int64_t wbase = (int64_t)blockIdx.x * kSortTile + w * (kSortItems * 64);
    uint64_t key[kSortItems];
    uint32_t val[kSortItems];
    uint32_t rank[kSortItems];
    const uint64_t lt_mask = lane == 0 ? 0ull : (~0ull >> (64 - lane));
#pragma unroll
    for (int r = 0; r < kSortItems; ++r) {
        const int64_t i = wbase + r * 64 + lane;
        const bool valid = i < D;
        key[r] = valid ? keys_in[i] : ~0ull;
        val[r] = valid ? vals_in[i] : 0u;
        const uint32_t d = digit_of(key[r], shift);
        uint64_t peers = __ballot(valid);
#pragma unroll
        for (int b = 0; b < kRadixBits; ++b) {
            const uint64_t m = __ballot((d >> b) & 1u);
            peers &= ((d >> b) & 1u) ? m : ~m;
        }
        // all lanes in `peers` share digit d (invalid lanes excluded)
        const uint32_t before = wcount[w][d];  // read before the leader updates it
        const uint32_t below = __popcll(peers & lt_mask);
        rank[r] = before + below;
        __builtin_amdgcn_wave_barrier();
        if (valid && below == 0) wcount[w][d] = before + __popcll(peers);
        __builtin_amdgcn_wave_barrier();
    }
    __syncthreads();
    // per digit: exclusive prefix over the 4 waves; then the digit's first slot inside the tile
    // (exclusive prefix of the tile's digit counts over the 256 digits, one per lane)
    {
        const int d = threadIdx.x;
        uint32_t run = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const uint32_t c = wcount[k][d];
            wcount[k][d] = run;
            run += c;
        }
        const uint32_t inc2 = wave_scan_incl(run);
        if (lane == 63) wsum4[w] = inc2;
        __syncthreads();
        uint32_t off = inc2 - run;
        for (int k = 0; k < w; ++k) off += wsum4[k];
        tile_off[d] = off;
        gbase[d] -= off;   // output slot of tile position p with digit d = gbase[d] + p (mod 2^32)
    }
    __syncthreads();
    // Sort the tile in LDS, then write it out in order: consecutive lanes then store to
    // consecutive addresses inside each digit's run (a wave's store touches the few runs it
    // spans instead of up to 64 scattered lines).
#pragma unroll
    for (int r = 0; r < kSortItems; ++r) {
        const int64_t i = wbase + r * 64 + lane;
        if (i < D) {
            const uint32_t d = digit_of(key[r], shift);
            const uint32_t p = tile_off[d] + wcount[w][d] + rank[r];
            lkey[p] = key[r];
            lval[p] = val[r];
        }
    }
    __syncthreads();
    const int64_t tile_base = (int64_t)blockIdx.x * kSortTile;
    const uint32_t n_valid = (uint32_t)max((int64_t)0, min((int64_t)kSortTile, D - tile_base));
#pragma unroll
    for (int r = 0; r < kSortItems; ++r) {
        const uint32_t p = r * kSortThreads + threadIdx.x;
        if (p < n_valid) {
            const uint64_t k = lkey[p];
            const uint32_t dst = gbase[digit_of(k, shift)] + p;
            keys_out[dst] = k;
            vals_out[dst] = lval[p];
        }
    }
}

// Per-tile ranges from the sorted compact keys, and the keys in their documented form
// (tile << 32 | float bits of depth) for the caller.
__global__ __launch_bounds__(256) void tile_ranges_kernel(const uint64_t* __restrict__ keys,
                                                          int64_t D, uint32_t dmin, int low_bits,
                                                          uint32_t* __restrict__ ranges,
                                                          uint64_t* __restrict__ keys_out,
                                                          const uint32_t* __restrict__ tail, int T) {
    if (tail) { D = min(D, (int64_t)tail[2]); dmin = tail[0]; }
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= D) return;
    const uint64_t k = keys[i];
    const uint32_t t = (uint32_t)(k >> low_bits);
    // t >= T only when a guessed key width was too narrow (the caller then repeats the call)
    if (t < (uint32_t)T) {
        if (i == 0 || (uint32_t)(keys[i - 1] >> low_bits) != t) ranges[2 * t] = (uint32_t)i;
        if (i == D - 1 || (uint32_t)(keys[i + 1] >> low_bits) != t) ranges[2 * t + 1] = (uint32_t)(i + 1);
    }
    const uint32_t low = (uint32_t)(k & ((1ull << low_bits) - 1ull));
    keys_out[i] = ((uint64_t)t << 32) | (uint64_t)(low + dmin);
}

static int bits_for(uint32_t span) {   // bits needed to hold values 0..span
    int bits = 0;
    while (bits < 32 && ((uint64_t)1 << bits) <= (uint64_t)span) ++bits;
    return bits;
}

static int sort_passes(int T, int low_bits) {
    const int tile_bits = bits_for((uint32_t)(T > 0 ? T - 1 : 0));
    const int p = (low_bits + tile_bits + kRadixBits - 1) / kRadixBits;
    return p > 0 ? p : 1;
}

struct BinWorkspace {
    uint64_t* keys_alt;
    uint32_t* vals_alt;
    uint32_t* ghist;  // [256] digit totals of the current pass
    uint32_t* thist;  // [256][ntiles]
    size_t bytes;
};

static size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }

static BinWorkspace carve(void* ws, int64_t D) {
    BinWorkspace b;
    const int ntiles = cdiv(D > 0 ? D : 1, kSortTile);
    size_t off = 0;
    char* p = (char*)ws;
    b.keys_alt = (uint64_t*)(p + off); off += align256(sizeof(uint64_t) * (size_t)(D > 0 ? D : 1));
    b.vals_alt = (uint32_t*)(p + off); off += align256(sizeof(uint32_t) * (size_t)(D > 0 ? D : 1));
    b.ghist = (uint32_t*)(p + off);    off += align256(sizeof(uint32_t) * kMaxPasses * kRadix);
    b.thist = (uint32_t*)(p + off);    off += align256(sizeof(uint32_t) * kRadix * (size_t)ntiles);
    b.bytes = off;
    return b;
}

}  // namespace vings

using namespace vings;

extern "C" size_t vings_surfel_bin_workspace_bytes(int64_t D) { return carve(nullptr, D).bytes; }

// tail == NULL: D, dmin and low_bits are the view's true values.  Otherwise D is the capacity
// of the buffers and low_bits a guess, and the kernels take the true values from tail[0..2].
static int bin_impl(int N, int H, int W, int64_t D, uint32_t dmin, int low_bits,
                    const uint32_t* tail, const float* depth, const int32_t* rect,
                    const uint32_t* tiles_touched, const uint32_t* block_prefix,
                    uint64_t* keys_sorted, uint32_t* vals_sorted, uint32_t* ranges,
                    void* workspace, size_t workspace_bytes, void* stream) {
    VINGS_REQUIRE(N >= 0 && H > 0 && W > 0 && D >= 0, "sizes");
    VINGS_REQUIRE(D < ((int64_t)1 << 32), "instance count must fit 32 bits");
    VINGS_REQUIRE(ranges != nullptr, "ranges");
    hipStream_t s = (hipStream_t)stream;
    const int gx = (W + 15) / 16, gy = (H + 15) / 16;
    const int T = gx * gy;
    if (D == 0 || N == 0) {
        VINGS_CHECK_HIP(hipMemsetAsync(ranges, 0, sizeof(uint32_t) * 2 * (size_t)T, s));
        return VINGS_OK;
    }
    VINGS_REQUIRE(depth && rect && tiles_touched && block_prefix && keys_sorted && vals_sorted &&
                      workspace,
                  "null pointer");
    VINGS_REQUIRE(((uintptr_t)workspace & 255) == 0, "workspace must be 256-byte aligned");
    const BinWorkspace b = carve(workspace, D);
    if (workspace_bytes < b.bytes) {
        set_error("bin workspace too small: %zu < %zu", workspace_bytes, b.bytes);
        return VINGS_ERR_CAPACITY;
    }
    const int passes = sort_passes(T, low_bits);
    VINGS_REQUIRE(passes <= kMaxPasses, "key too wide");
    ProfScope prof(PROF_BIN, s);
    // ping-pong so that the last pass leaves the (compact) keys in the workspace, from where
    // tile_ranges writes them decoded into keys_sorted, and the values in vals_sorted
    uint64_t* kbuf[2] = {keys_sorted, b.keys_alt};
    uint32_t* vbuf[2] = {vals_sorted, b.vals_alt};
    int cur = (passes % 2 == 0) ? 1 : 0;    // keys: end in kbuf[1]
    int vcur = (passes % 2 == 0) ? 0 : 1;   // values: end in vbuf[0]

    emit_keys_kernel<<<cdiv(N, 256), 256, 0, s>>>(N, gx, depth, rect, tiles_touched, block_prefix,
                                                  dmin, low_bits, kbuf[cur], vbuf[vcur], ranges,
                                                  2 * T, tail, (uint32_t)D);
    VINGS_LAUNCH_CHECK();

    const int ntiles = cdiv(D, kSortTile);
    for (int p = 0; p < passes; ++p) {
        const int shift = p * kRadixBits;
        sort_tile_hist_kernel<<<ntiles, kSortThreads, 0, s>>>(kbuf[cur], D, shift, ntiles, b.thist,
                                                              tail);
        VINGS_LAUNCH_CHECK();
        sort_scan_kernel<<<kRadix, 256, 0, s>>>(b.thist, ntiles, b.ghist);
        VINGS_LAUNCH_CHECK();
        sort_scatter_kernel<<<ntiles, kSortThreads, 0, s>>>(kbuf[cur], vbuf[vcur], kbuf[cur ^ 1],
                                                            vbuf[vcur ^ 1], D, shift, ntiles,
                                                            b.thist, b.ghist, tail);
        VINGS_LAUNCH_CHECK();
        cur ^= 1;
        vcur ^= 1;
    }
    // cur == 1, vcur == 0 here by construction
    tile_ranges_kernel<<<cdiv(D, 256), 256, 0, s>>>(b.keys_alt, D, dmin, low_bits, ranges,
                                                    keys_sorted, tail, T);
    VINGS_LAUNCH_CHECK();
    return VINGS_OK;
}

extern "C" int vings_surfel_bin(int N, int H, int W, int64_t D, uint32_t depth_bits_min,
                                uint32_t depth_bits_max, const float* depth,
                                const int32_t* rect, const uint32_t* tiles_touched,
                                const uint32_t* block_prefix, uint64_t* keys_sorted,
                                uint32_t* vals_sorted, uint32_t* ranges, void* workspace,
                                size_t workspace_bytes, void* stream) {
    // depth bits are rebased to the view's smallest depth; an unknown range (min > max) keeps
    // all 32 bits
    const bool ranged = depth_bits_min <= depth_bits_max;
    return bin_impl(N, H, W, D, ranged ? depth_bits_min : 0u,
                    ranged ? bits_for(depth_bits_max - depth_bits_min) : 32, nullptr, depth, rect,
                    tiles_touched, block_prefix, keys_sorted, vals_sorted, ranges, workspace,
                    workspace_bytes, stream);
}

extern "C" int vings_surfel_bin_low_bits(uint32_t depth_bits_min, uint32_t depth_bits_max) {
    return depth_bits_min <= depth_bits_max ? bits_for(depth_bits_max - depth_bits_min) : 32;
}

extern "C" int vings_surfel_bin_speculative(int N, int H, int W, int64_t D_capacity,
                                            int low_bits_guess, const uint32_t* tail,
                                            const float* depth, const int32_t* rect,
                                            const uint32_t* tiles_touched,
                                            const uint32_t* block_prefix, uint64_t* keys_sorted,
                                            uint32_t* vals_sorted, uint32_t* ranges,
                                            void* workspace, size_t workspace_bytes,
                                            void* stream) {
    VINGS_REQUIRE(tail != nullptr && D_capacity > 0 && N > 0, "speculative binning needs a guess");
    VINGS_REQUIRE(low_bits_guess >= 0 && low_bits_guess <= 32, "low_bits_guess");
    return bin_impl(N, H, W, D_capacity, 0u, low_bits_guess, tail, depth, rect, tiles_touched,
                    block_prefix, keys_sorted, vals_sorted, ranges, workspace, workspace_bytes,
                    stream);
}
