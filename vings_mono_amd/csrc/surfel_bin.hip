// Tile binning for the surfel rasteriser on gfx950: every (splat, tile) instance of a view
// into the list of its tile, every list ordered by depth.  Two levels:
//   1. emit      (tile, splat index) pairs as two 32-bit arrays, in splat order
//   2. tile sort stable least-significant-digit radix passes (8-bit digits, written for 64-lane
//                wavefronts) over the TILE numbers only: 2 passes at 1080p (13 tile bits) where
//                sorting the whole (tile, depth) key took 5 over its 40 differing bits.  The
//                depth does not travel through the passes (8 bytes per instance and pass
//                direction instead of 12): the last stage fetches it by splat index
//   3. ranges    per-tile ranges from the tile-ordered pairs
//   4. depth sort one workgroup per tile gathers the depths of its list, orders it in LDS by
//                (depth bits, splat index) and writes the keys (tile << 32 | depth bits) and
//                values of the caller's arrays:
//                a sample sort (bisection over 64 ordered sample keys, LDS counters, rank
//                among the few bucket mates), a bitonic network below 129 entries and for the
//                sample; lists beyond the LDS capacity: LDS-sorted runs + merge passes through
//                the free half of the radix sort's ping-pong buffers
// 9 launches instead of 17 for the 2.4 M instances of a 1080p view.
//
// The order is fully determined: a splat emits at most one instance per tile and instances
// are emitted in splat order, so a stable sort by (tile, depth) equals ordering every tile's
// list by (depth bits, splat index).  Keys, values and ranges equal the CPU oracle's bit for
// bit.
//
// Routing the instances to per-tile buckets with global atomics (count, scan, scatter)
// instead of the radix passes was measured and dropped: gfx950 serves scattered 4-byte atomics
// at about 44 G/s whatever their scope and however the counters are replicated
// (tools/experiments/tile_count_bench.hip: 2.4 M instances, decode alone 12 us, with the
// counting atomics 66 us, with returning atomics and the store 72 us), so the two routing
// kernels alone cost what three radix passes do.
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
constexpr int kMaxPasses = 4;
constexpr int kSortCap = 1024;          // composites one LDS depth sort holds (8 + 8 KB of LDS)

// ---------------------------------------------------------------------------
// duplicate with keys: lane i owns splat i; its output offset is the
// exclusive prefix of tiles_touched, rebuilt here from the per-block prefix.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void emit_keys_kernel(
    int N, int gx, const int32_t* __restrict__ rect,
    const uint32_t* __restrict__ tiles_touched, const uint32_t* __restrict__ block_prefix,
    uint32_t* __restrict__ tkeys, uint32_t* __restrict__ vals, uint32_t* __restrict__ ranges,
    int n_range_words, uint32_t cap) {
    // nothing is written beyond the `cap` entries the buffers hold (a speculative launch may
    // have guessed the instance count too small)
    __shared__ uint32_t wtot[4];
    __shared__ uint32_t start[257];     // first output slot of each splat, relative to the block
    __shared__ int4 srect[256];
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
    if (cnt) srect[threadIdx.x] = reinterpret_cast<const int4*>(rect)[i];
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
        tkeys[out0 + q] = tile;
        vals[out0 + q] = (uint32_t)(blockIdx.x * 256 + lo);
    }
}

// ---------------------------------------------------------------------------
// radix sort
// ---------------------------------------------------------------------------
__device__ __forceinline__ uint32_t digit_of(uint32_t k, int shift) {
    return (k >> shift) & (kRadix - 1);
}

// Per-tile digit histogram of one pass: thist[digit][tile].
__global__ __launch_bounds__(kSortThreads) void sort_tile_hist_kernel(
    const uint32_t* __restrict__ keys, int64_t D, int shift, int ntiles,
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
    const uint32_t* __restrict__ keys_in, const uint32_t* __restrict__ vals_in,
    uint32_t* __restrict__ keys_out, uint32_t* __restrict__ vals_out, int64_t D, int shift,
    int ntiles, const uint32_t* __restrict__ thist, const uint32_t* __restrict__ totals,
    const uint32_t* __restrict__ tail, int nbits) {
    // nbits: significant bits of this pass's digit (the last pass of a 13-bit tile number has 5:
    // the ballots of the bits above are all-zero and skipped)
    __shared__ uint32_t wcount[4][kRadix];  // per-wave digit counts, then prefixes
    if (tail) D = min(D, (int64_t)tail[2]);
    __shared__ uint32_t gbase[kRadix];
    __shared__ uint32_t tile_off[kRadix];
    __shared__ uint32_t wsum4[4];
    __shared__ uint32_t lkey[kSortTile];
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
    uint32_t key[kSortItems];
    uint32_t val[kSortItems];
    uint32_t rank[kSortItems];
    const uint64_t lt_mask = lane == 0 ? 0ull : (~0ull >> (64 - lane));
#pragma unroll
    for (int r = 0; r < kSortItems; ++r) {
        const int64_t i = wbase + r * 64 + lane;
        const bool valid = i < D;
        key[r] = valid ? keys_in[i] : ~0u;
        val[r] = valid ? vals_in[i] : 0u;
        const uint32_t d = digit_of(key[r], shift);
        uint64_t peers = __ballot(valid);
#pragma unroll
        for (int b = 0; b < kRadixBits; ++b) {
            if (b < nbits) {
                const uint64_t m = __ballot((d >> b) & 1u);
                peers &= ((d >> b) & 1u) ? m : ~m;
            }
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
            const uint32_t k = lkey[p];
            const uint32_t dst = gbase[digit_of(k, shift)] + p;
            keys_out[dst] = k;
            vals_out[dst] = lval[p];
        }
    }
}

// Per-tile ranges from the tile-ordered tile numbers.
__global__ __launch_bounds__(256) void tile_ranges_kernel(const uint32_t* __restrict__ tkeys,
                                                          int64_t D, uint32_t* __restrict__ ranges,
                                                          const uint32_t* __restrict__ tail, int T) {
    if (tail) D = min(D, (int64_t)tail[2]);
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= D) return;
    const uint32_t t = tkeys[i];
    if (t >= (uint32_t)T) return;
    if (i == 0 || tkeys[i - 1] != t) ranges[2 * t] = (uint32_t)i;
    if (i == D - 1 || tkeys[i + 1] != t) ranges[2 * t + 1] = (uint32_t)(i + 1);
}

// Ascending bitonic network over the first P (a power of two >= 64) entries of s, 256 lanes.
// Exchanges over a distance below 64 stay inside an aligned block of 64 entries: a wave takes
// such a block into registers (one entry per lane) and runs all of them as lane exchanges,
// without barriers; only the exchanges over 64 entries or more go through LDS.  A 512-entry
// list takes 6 LDS stages and 4 register phases instead of 45 LDS stages with a barrier each.
__device__ __forceinline__ uint64_t lane_xor64(uint64_t v, int j) {
    const uint32_t lo = __shfl_xor((uint32_t)v, j, 64), hi = __shfl_xor((uint32_t)(v >> 32), j, 64);
    return ((uint64_t)hi << 32) | lo;
}

// stages (k, j) for j = j0, j0/2, .., 1 on entry g of the network, held in v
__device__ __forceinline__ uint64_t bitonic_lane_stages(uint64_t v, uint32_t g, int k, int j0) {
    const bool asc = (g & (uint32_t)k) == 0;
    for (int j = j0; j > 0; j >>= 1) {
        const uint64_t o = lane_xor64(v, j);
        const bool keep_min = (((g & (uint32_t)j) == 0) == asc);   // the pair's lower entry, ascending
        v = (keep_min == (o < v)) ? o : v;
    }
    return v;
}

__device__ __forceinline__ void bitonic_lds(uint64_t* s, int P) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int nblocks = P >> 6;
    for (int blk = w; blk < nblocks; blk += 4) {
        const uint32_t g = (uint32_t)(blk * 64 + lane);
        uint64_t v = s[g];
#pragma unroll
        for (int k = 2; k <= 64; k <<= 1) v = bitonic_lane_stages(v, g, k, k >> 1);
        s[g] = v;
    }
    __syncthreads();
    for (int k = 128; k <= P; k <<= 1) {
        for (int j = k >> 1; j >= 64; j >>= 1) {
            for (int i = threadIdx.x; i < (P >> 1); i += 256) {
                const int lo = ((i & ~(j - 1)) << 1) | (i & (j - 1));
                const int hi = lo | j;
                const uint64_t a = s[lo], b = s[hi];
                if ((a > b) == ((lo & k) == 0)) { s[lo] = b; s[hi] = a; }
            }
            __syncthreads();
        }
        for (int blk = w; blk < nblocks; blk += 4) {
            const uint32_t g = (uint32_t)(blk * 64 + lane);
            s[g] = bitonic_lane_stages(s[g], g, k, 32);
        }
        __syncthreads();
    }
}

// Ascending order of the n <= kSortCap composites in s[0 .. n), 256 lanes; the result is left in
// s.  Lists of up to 128 entries go through the bitonic network.  Longer ones are sample-sorted,
// which costs a fifth of the network's compare-exchanges at the usual 300 entries: a strided
// sample of 64 (256 beyond 512 entries) keys is ordered by the network and cuts the key range
// into buckets of a few entries each; every entry finds its bucket by bisection over the
// sample, takes a place in it with an LDS counter, is moved next to its bucket mates, and its
// final position is the bucket's first slot plus the number of mates below it.  The composites
// are distinct, so buckets stay small whatever the depth distribution is (a strided sample of
// an already ordered list is the ideal sample); a bad sample costs time, never the order.
struct SortShared {
    uint64_t s[kSortCap];
    uint64_t grp[kSortCap];
    uint64_t spl[256];
    uint32_t cnt[256];
    uint32_t start[256];
    uint32_t wtot[4];
};

__device__ __forceinline__ void lds_sort(SortShared& sh, uint32_t n) {
    if (n <= 128u) {
        const int P = n <= 64u ? 64 : 128;
        for (uint32_t i = n + threadIdx.x; i < (uint32_t)P; i += 256) sh.s[i] = ~0ull;
        __syncthreads();
        bitonic_lds(sh.s, P);
        return;
    }
    const int NB = n <= 512u ? 64 : 256;
    if ((int)threadIdx.x < NB) {
        sh.spl[threadIdx.x] = sh.s[(threadIdx.x * n) / (uint32_t)NB];
        sh.cnt[threadIdx.x] = 0u;
    }
    __syncthreads();
    bitonic_lds(sh.spl, NB);
    constexpr int kRounds = kSortCap / 256;
    uint64_t e[kRounds];
    uint32_t bk[kRounds], pos[kRounds];
#pragma unroll
    for (int r = 0; r < kRounds; ++r) {
        const uint32_t i = r * 256 + threadIdx.x;
        if (i < n) {
            e[r] = sh.s[i];
            int lo = 0;                            // largest j with spl[j] <= e (0 if none)
            for (int step = NB >> 1; step > 0; step >>= 1)
                if (sh.spl[lo + step] <= e[r]) lo += step;
            bk[r] = (uint32_t)lo;
            pos[r] = atomicAdd(&sh.cnt[lo], 1u);
        }
    }
    __syncthreads();
    {   // exclusive scan of the bucket counts (one per lane)
        const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
        const uint32_t v = (int)threadIdx.x < NB ? sh.cnt[threadIdx.x] : 0u;
        const uint32_t inc = wave_scan_incl(v);
        if (lane == 63) sh.wtot[w] = inc;
        __syncthreads();
        uint32_t off = inc - v;
        for (int k = 0; k < w; ++k) off += sh.wtot[k];
        sh.start[threadIdx.x] = off;
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < kRounds; ++r)
        if (r * 256 + threadIdx.x < n) sh.grp[sh.start[bk[r]] + pos[r]] = e[r];
    __syncthreads();
#pragma unroll
    for (int r = 0; r < kRounds; ++r) {
        if (r * 256 + threadIdx.x < n) {
            const uint32_t b0 = sh.start[bk[r]], c = sh.cnt[bk[r]];
            uint32_t rank = b0;
            for (uint32_t q = b0; q < b0 + c; ++q) rank += sh.grp[q] < e[r] ? 1u : 0u;
            sh.s[rank] = e[r];
        }
    }
    __syncthreads();
}

// One workgroup per tile: gathers the depths of the tile's list (`ids`, in splat order: the radix
// passes are stable), orders it by (depth bits, splat index) and writes keys = tile << 32 |
// depth bits and vals = splat index into the caller's arrays.
__global__ __launch_bounds__(256) void tile_depth_sort_kernel(
    const uint32_t* __restrict__ ranges, const uint32_t* __restrict__ ids,
    const float* __restrict__ depth, uint64_t* __restrict__ keys, uint32_t* __restrict__ vals,
    uint64_t* __restrict__ scratch) {
    __shared__ SortShared sh;
    const uint32_t t = blockIdx.x;
    const uint2 rg = reinterpret_cast<const uint2*>(ranges)[t];
    const uint32_t b = rg.x, n = rg.y - rg.x;
    if (n == 0u) return;
    const uint64_t tile_hi = (uint64_t)t << 32;
    auto composite = [&](uint32_t i) {
        const uint32_t id = ids[b + i];
        return ((uint64_t)__float_as_uint(depth[id]) << 32) | (uint64_t)id;
    };
    if (n <= (uint32_t)kSortCap) {
        for (uint32_t i = threadIdx.x; i < n; i += 256) sh.s[i] = composite(i);
        __syncthreads();
        if (n > 1u) lds_sort(sh, n);
        for (uint32_t i = threadIdx.x; i < n; i += 256) {
            const uint64_t c = sh.s[i];
            keys[b + i] = tile_hi | (c >> 32);
            vals[b + i] = (uint32_t)c;
        }
        return;
    }
    // A list the LDS does not hold: runs of kSortCap composites sorted in LDS and parked in the
    // scratch buffer, then merged pairwise, alternating between the scratch buffer and the
    // tile's own stretch of the key array (nothing else lives there); every pair of runs is cut
    // into 256 equal output pieces along its merge path.
    uint64_t* A = scratch + b;
    uint64_t* B = keys + b;
    for (uint32_t r0 = 0; r0 < n; r0 += kSortCap) {
        const uint32_t len = min((uint32_t)kSortCap, n - r0);
        for (uint32_t i = threadIdx.x; i < len; i += 256) sh.s[i] = composite(r0 + i);
        __syncthreads();
        lds_sort(sh, len);
        for (uint32_t i = threadIdx.x; i < len; i += 256) A[r0 + i] = sh.s[i];
        __syncthreads();
    }
    uint64_t* src = A;
    uint64_t* dst = B;
    for (uint32_t R = kSortCap; R < n; R <<= 1) {
        for (uint32_t p0 = 0; p0 < n; p0 += 2 * R) {
            const uint32_t la = min(R, n - p0);
            const uint32_t lb = n - p0 - la < R ? n - p0 - la : R;
            const uint64_t* a = src + p0;
            const uint64_t* bb = a + la;
            uint64_t* out = dst + p0;
            const uint32_t len = la + lb;
            const uint32_t per = (len + 255) / 256;
            const uint32_t d0 = min(threadIdx.x * per, len), d1 = min(d0 + per, len);
            // merge path: of the first d0 outputs, i come from a and d0 - i from bb
            uint32_t lo = d0 > lb ? d0 - lb : 0, hi = min(d0, la);
            while (lo < hi) {
                const uint32_t mid = (lo + hi) >> 1;
                if (a[mid] < bb[d0 - 1 - mid]) lo = mid + 1; else hi = mid;
            }
            uint32_t i = lo, j = d0 - lo;
            for (uint32_t o = d0; o < d1; ++o) {
                const bool take_a = j >= lb || (i < la && a[i] < bb[j]);
                out[o] = take_a ? a[i++] : bb[j++];
            }
        }
        __syncthreads();
        uint64_t* tmp = src; src = dst; dst = tmp;
    }
    // (when the ordered list sits in the key array itself, every lane converts the entries it
    // reads in place)
    for (uint32_t i = threadIdx.x; i < n; i += 256) {
        const uint64_t c = src[i];
        keys[b + i] = tile_hi | (c >> 32);
        vals[b + i] = (uint32_t)c;
    }
}

// Tiles in the order the blend passes should take them: longest list first (counting sort over
// 256 length classes, one workgroup).  Workgroups are dispatched in index order, so the launch
// then ends on its cheapest tiles instead of on whatever the image's last rows hold, and
// consecutive workgroups -- dealt round-robin over the XCDs -- carry similar loads.
__global__ __launch_bounds__(1024) void tile_order_kernel(const uint32_t* __restrict__ ranges, int T,
                                                          uint32_t* __restrict__ order) {
    __shared__ uint32_t cnt[256];
    __shared__ uint32_t smax;
    if (threadIdx.x < 256) cnt[threadIdx.x] = 0;
    if (threadIdx.x == 0) smax = 1;
    __syncthreads();
    const uint2* rg = reinterpret_cast<const uint2*>(ranges);
    uint32_t mx = 0;
    for (int t = threadIdx.x; t < T; t += 1024) { const uint2 r = rg[t]; mx = max(mx, r.y - r.x); }
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) mx = max(mx, (uint32_t)__shfl_xor(mx, o, 64));
    if ((threadIdx.x & 63) == 0) atomicMax(&smax, mx);
    __syncthreads();
    const float scale = 255.f / (float)smax;
    auto cls = [&](int t) { const uint2 r = rg[t]; return 255u - min(255u, (uint32_t)((float)(r.y - r.x) * scale)); };
    for (int t = threadIdx.x; t < T; t += 1024) atomicAdd(&cnt[cls(t)], 1u);
    __syncthreads();
    if (threadIdx.x < 64) {      // exclusive scan of the 256 class counts by one wave, 4 per lane
        uint32_t c[4], sum = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) { c[k] = cnt[threadIdx.x * 4 + k]; sum += c[k]; }
        uint32_t inc = sum;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const uint32_t v = __shfl_up(inc, o, 64);
            if ((int)threadIdx.x >= o) inc += v;
        }
        uint32_t run = inc - sum;
#pragma unroll
        for (int k = 0; k < 4; ++k) { cnt[threadIdx.x * 4 + k] = run; run += c[k]; }
    }
    __syncthreads();
    for (int t = threadIdx.x; t < T; t += 1024) order[atomicAdd(&cnt[cls(t)], 1u)] = (uint32_t)t;
}

static int bits_for(uint32_t span) {   // bits needed to hold values 0..span
    int bits = 0;
    while (bits < 32 && ((uint64_t)1 << bits) <= (uint64_t)span) ++bits;
    return bits;
}

struct BinWorkspace {
    uint32_t* pairs;   // [2][D]: tile numbers, then splat indices (one side of the ping-pong; the
                       // other side borrows the caller's key array until the last stage fills it)
    uint64_t* merge;   // [D]: merge space of the depth sort for lists the LDS does not hold
    uint32_t* ghist;   // [256] digit totals of the current pass
    uint32_t* thist;   // [256][ntiles]
    size_t bytes;
};

static size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }

static BinWorkspace carve(void* ws, int64_t D) {
    BinWorkspace b;
    const int ntiles = cdiv(D > 0 ? D : 1, kSortTile);
    size_t off = 0;
    char* p = (char*)ws;
    b.pairs = (uint32_t*)(p + off); off += align256(2 * sizeof(uint32_t) * (size_t)(D > 0 ? D : 1));
    b.merge = (uint64_t*)(p + off); off += align256(sizeof(uint64_t) * (size_t)(D > 0 ? D : 1));
    b.ghist = (uint32_t*)(p + off);    off += align256(sizeof(uint32_t) * kMaxPasses * kRadix);
    b.thist = (uint32_t*)(p + off);    off += align256(sizeof(uint32_t) * kRadix * (size_t)ntiles);
    b.bytes = off;
    return b;
}

}  // namespace vings

using namespace vings;

extern "C" size_t vings_surfel_bin_workspace_bytes(int64_t D) { return carve(nullptr, D).bytes; }

// tail == NULL: D is the view's true instance count.  Otherwise D is the capacity of the
// buffers (a guess the caller checks afterwards) and the kernels take the true count from
// tail[2], where the projection left it.
static int bin_impl(int N, int H, int W, int64_t D, const uint32_t* tail, const float* depth,
                    const int32_t* rect, const uint32_t* tiles_touched,
                    const uint32_t* block_prefix, uint64_t* keys_sorted, uint32_t* vals_sorted,
                    uint32_t* ranges, void* workspace, size_t workspace_bytes, void* stream) {
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
    VINGS_REQUIRE(((uintptr_t)ranges & 7) == 0, "ranges must be 8-byte aligned");
    const BinWorkspace b = carve(workspace, D);
    if (workspace_bytes < b.bytes) {
        set_error("bin workspace too small: %zu < %zu", workspace_bytes, b.bytes);
        return VINGS_ERR_CAPACITY;
    }
    const int tile_bits = bits_for((uint32_t)(T - 1));
    const int passes = tile_bits > 0 ? (tile_bits + kRadixBits - 1) / kRadixBits : 1;
    VINGS_REQUIRE(passes <= kMaxPasses, "too many tiles");
    ProfScope prof(PROF_BIN, s);
    // (tile, splat) pairs as two 32-bit arrays of D entries each.  Ping-pong between the
    // caller's key array (its 8 D bytes hold one such pair of arrays) and the workspace, so that
    // the last pass leaves the pairs in the WORKSPACE: the depth sort then fills the caller's
    // keys and values from there.
    uint32_t* kbuf[2] = {reinterpret_cast<uint32_t*>(keys_sorted), b.pairs};
    uint32_t* vbuf[2] = {reinterpret_cast<uint32_t*>(keys_sorted) + D, b.pairs + D};
    int cur = (passes & 1) ? 0 : 1;

    emit_keys_kernel<<<cdiv(N, 256), 256, 0, s>>>(N, gx, rect, tiles_touched, block_prefix,
                                                  kbuf[cur], vbuf[cur], ranges, 2 * T, (uint32_t)D);
    VINGS_LAUNCH_CHECK();
    const int ntiles = cdiv(D, kSortTile);
    for (int p = 0; p < passes; ++p) {
        const int shift = p * kRadixBits;
        sort_tile_hist_kernel<<<ntiles, kSortThreads, 0, s>>>(kbuf[cur], D, shift, ntiles, b.thist,
                                                              tail);
        VINGS_LAUNCH_CHECK();
        sort_scan_kernel<<<kRadix, 256, 0, s>>>(b.thist, ntiles, b.ghist);
        VINGS_LAUNCH_CHECK();
        sort_scatter_kernel<<<ntiles, kSortThreads, 0, s>>>(kbuf[cur], vbuf[cur], kbuf[cur ^ 1],
                                                            vbuf[cur ^ 1], D, shift, ntiles,
                                                            b.thist, b.ghist, tail,
                                                            min(kRadixBits, max(1, tile_bits - shift)));
        VINGS_LAUNCH_CHECK();
        cur ^= 1;
    }
    // cur == 1 here: the tile-ordered pairs are in the workspace
    tile_ranges_kernel<<<cdiv(D, 256), 256, 0, s>>>(kbuf[1], D, ranges, tail, T);
    VINGS_LAUNCH_CHECK();
    tile_depth_sort_kernel<<<T, 256, 0, s>>>(ranges, vbuf[1], depth, keys_sorted, vals_sorted,
                                             b.merge);
    VINGS_LAUNCH_CHECK();
    return VINGS_OK;
}

extern "C" int vings_surfel_bin(int N, int H, int W, int64_t D, uint32_t depth_bits_min,
                                uint32_t depth_bits_max, const float* depth,
                                const int32_t* rect, const uint32_t* tiles_touched,
                                const uint32_t* block_prefix, uint64_t* keys_sorted,
                                uint32_t* vals_sorted, uint32_t* ranges, void* workspace,
                                size_t workspace_bytes, void* stream) {
    (void)depth_bits_min; (void)depth_bits_max;   // the depth order no longer depends on them
    return bin_impl(N, H, W, D, nullptr, depth, rect, tiles_touched, block_prefix, keys_sorted,
                    vals_sorted, ranges, workspace, workspace_bytes, stream);
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
    (void)low_bits_guess;
    VINGS_REQUIRE(tail != nullptr && D_capacity > 0 && N > 0, "speculative binning needs a guess");
    return bin_impl(N, H, W, D_capacity, tail, depth, rect, tiles_touched, block_prefix,
                    keys_sorted, vals_sorted, ranges, workspace, workspace_bytes, stream);
}

extern "C" int vings_surfel_tile_order(int H, int W, const uint32_t* ranges, uint32_t* tile_order,
                                       void* stream) {
    VINGS_REQUIRE(H > 0 && W > 0, "sizes");
    VINGS_REQUIRE(ranges && tile_order, "null pointer");
    VINGS_REQUIRE(((uintptr_t)ranges & 7) == 0, "ranges must be 8-byte aligned");
    const int T = ((W + 15) / 16) * ((H + 15) / 16);
    ProfScope prof(PROF_BIN, (hipStream_t)stream);
    tile_order_kernel<<<1, 1024, 0, (hipStream_t)stream>>>(ranges, T, tile_order);
    VINGS_LAUNCH_CHECK();
    return VINGS_OK;
}
