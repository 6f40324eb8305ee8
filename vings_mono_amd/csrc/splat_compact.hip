// Stream compaction of the splat set on gfx950: every per-splat array (parameters, Adam
// moments, score records) loses the rows of pruned splats in one plan + one apply launch,
// order preserved.
//
// Replaces the boolean-mask indexing the reference runs array by array when it prunes
//   prune_tensors_from_optimizer  /root/reference/scripts/gaussian/gaussian_base.py:94-110
//   update_records(mode="prune")  /root/reference/scripts/gaussian/gaussian_model.py:178-183
// (called from add_new_frame :121-123 and storage_control :246-248): 20 index_select-style
// launches plus 20 mask scans there.
#include <mutex>

#include "common.hpp"

namespace vings {

// Scratch of vings_splat_compact_plan_static: the per-block counts of the largest mask it
// accepts, in the library's own device memory (every GPU the library runs on holds one copy at
// its own address for the life of the process).  In host code this name is only a handle for
// hipGetSymbolAddress; kernels receive the address that call returns.
constexpr int kStaticMaxRows = 1 << 23;                       // 8 388 608 splats
constexpr int kStaticMaxBlocks = kStaticMaxRows / 256;        // 32 768 blocks -> 128 KiB + 4 B
static __device__ __align__(256) uint32_t g_compact_scratch[kStaticMaxBlocks + 1];

// keep[i] != 0 -> dst[i] = number of kept rows before i.  Block counts, a single-workgroup
// scan over them, then the per-row offsets.
__global__ __launch_bounds__(256) void compact_count_kernel(int N, const uint8_t* __restrict__ keep,
                                                            uint32_t* __restrict__ block_counts) {
    __shared__ uint32_t wsum[4];
    const int i = blockIdx.x * 256 + threadIdx.x;
    const uint32_t k = (i < N && keep[i]) ? 1u : 0u;
    const uint32_t s = wave_sum_u32(k);
    if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) block_counts[blockIdx.x] = wsum[0] + wsum[1] + wsum[2] + wsum[3];
}

__global__ __launch_bounds__(1024) void compact_scan_kernel(uint32_t* __restrict__ counts, int nb,
                                                            long long* __restrict__ total_out) {
    __shared__ uint32_t wtot[16];
    __shared__ uint32_t carry_s;
    if (threadIdx.x == 0) carry_s = 0;
    __syncthreads();
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    for (int base = 0; base < nb; base += 1024) {
        const int i = base + threadIdx.x;
        const uint32_t v = i < nb ? counts[i] : 0u;
        const uint32_t inc = wave_scan_incl(v);
        if (lane == 63) wtot[w] = inc;
        __syncthreads();
        uint32_t woff = 0;
        for (int k = 0; k < w; ++k) woff += wtot[k];
        const uint32_t carry = carry_s;
        if (i < nb) counts[i] = carry + woff + inc - v;
        __syncthreads();
        if (threadIdx.x == 1023) carry_s = carry + woff + inc;
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        counts[nb] = carry_s;
        if (total_out) *total_out = (long long)carry_s;
    }
}

__global__ __launch_bounds__(256) void compact_index_kernel(int N, const uint8_t* __restrict__ keep,
                                                            const uint32_t* __restrict__ block_prefix,
                                                            int32_t* __restrict__ dst) {
    __shared__ uint32_t wtot[4];
    const int i = blockIdx.x * 256 + threadIdx.x;
    const uint32_t k = (i < N && keep[i]) ? 1u : 0u;
    const uint32_t inc = wave_scan_incl(k);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    if (lane == 63) wtot[w] = inc;
    __syncthreads();
    uint32_t off = block_prefix[blockIdx.x] + inc - k;
    for (int q = 0; q < w; ++q) off += wtot[q];
    if (i < N) dst[i] = k ? (int32_t)off : -1;
}

constexpr int kMaxArrays = 24;
struct CompactArrays {
    const char* src[kMaxArrays];
    char* dst[kMaxArrays];
    int row_bytes[kMaxArrays];
    int n;
};

// One lane per source row; a kept row is copied in every array.  Rows are a few bytes
// (4 .. 16), so 4-byte words when the row allows it, bytes otherwise.
__global__ __launch_bounds__(256) void compact_apply_kernel(int N, const int32_t* __restrict__ dst_index,
                                                            CompactArrays A) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= N) return;
    const int d = dst_index[i];
    if (d < 0) return;
    for (int a = 0; a < A.n; ++a) {
        const int rb = A.row_bytes[a];
        const char* s = A.src[a] + (size_t)i * rb;
        char* t = A.dst[a] + (size_t)d * rb;
        if ((rb & 3) == 0) {
            for (int k = 0; k < rb; k += 4)
                *reinterpret_cast<uint32_t*>(t + k) = *reinterpret_cast<const uint32_t*>(s + k);
        } else {
            for (int k = 0; k < rb; ++k) t[k] = s[k];
        }
    }
}

}  // namespace vings

using namespace vings;

extern "C" size_t vings_splat_compact_workspace_bytes(int N) {
    return sizeof(uint32_t) * ((size_t)cdiv(N > 0 ? N : 1, 256) + 1);
}

extern "C" int vings_splat_compact_plan(int N, const uint8_t* keep, int32_t* dst_index,
                                        void* workspace, size_t workspace_bytes,
                                        int64_t* n_keep_host, void* stream) {
    VINGS_REQUIRE(N >= 0 && n_keep_host, "sizes");
    *n_keep_host = 0;
    if (N == 0) return VINGS_OK;
    VINGS_REQUIRE(keep && dst_index && workspace, "null pointer");
    VINGS_REQUIRE(workspace_bytes >= vings_splat_compact_workspace_bytes(N), "workspace too small");
    VINGS_REQUIRE(((uintptr_t)workspace & 3) == 0, "workspace must be 4-byte aligned");
    hipStream_t s = (hipStream_t)stream;
    uint32_t* counts = reinterpret_cast<uint32_t*>(workspace);
    const int nb = cdiv(N, 256);
    compact_count_kernel<<<nb, 256, 0, s>>>(N, keep, counts);
    VINGS_LAUNCH_CHECK();
    compact_scan_kernel<<<1, 1024, 0, s>>>(counts, nb, nullptr);
    VINGS_LAUNCH_CHECK();
    compact_index_kernel<<<nb, 256, 0, s>>>(N, keep, counts, dst_index);
    VINGS_LAUNCH_CHECK();
    uint32_t total = 0;
    VINGS_CHECK_HIP(hipMemcpyAsync(&total, counts + nb, sizeof(uint32_t), hipMemcpyDeviceToHost, s));
    VINGS_CHECK_HIP(hipStreamSynchronize(s));
    *n_keep_host = (int64_t)total;
    return VINGS_OK;
}

extern "C" int vings_splat_compact_plan_device(int N, const uint8_t* keep, int32_t* dst_index,
                                               void* workspace, size_t workspace_bytes,
                                               int64_t* n_keep_device, void* stream) {
    VINGS_REQUIRE(N >= 0 && n_keep_device, "sizes");
    hipStream_t s = (hipStream_t)stream;
    if (N == 0) {
        VINGS_CHECK_HIP(hipMemsetAsync(n_keep_device, 0, sizeof(int64_t), s));
        return VINGS_OK;
    }
    VINGS_REQUIRE(keep && dst_index && workspace, "null pointer");
    VINGS_REQUIRE(workspace_bytes >= vings_splat_compact_workspace_bytes(N), "workspace too small");
    VINGS_REQUIRE(((uintptr_t)workspace & 3) == 0, "workspace must be 4-byte aligned");
    uint32_t* counts = reinterpret_cast<uint32_t*>(workspace);
    const int nb = cdiv(N, 256);
    compact_count_kernel<<<nb, 256, 0, s>>>(N, keep, counts);
    VINGS_LAUNCH_CHECK();
    compact_scan_kernel<<<1, 1024, 0, s>>>(counts, nb, reinterpret_cast<long long*>(n_keep_device));
    VINGS_LAUNCH_CHECK();
    compact_index_kernel<<<nb, 256, 0, s>>>(N, keep, counts, dst_index);
    VINGS_LAUNCH_CHECK();
    return VINGS_OK;
}

namespace {
constexpr int kMaxDev = 64;
std::mutex g_scratch_mu;
uint32_t* g_scratch_ptr[kMaxDev] = {};
}  // namespace

extern "C" int vings_splat_compact_static_max_rows(void) { return kStaticMaxRows; }

// Address of the current GPU's copy of the scratch array, looked up once per device with that
// device current.
static int static_scratch(uint32_t** out) {
    int dev = 0;
    VINGS_CHECK_HIP(hipGetDevice(&dev));
    VINGS_REQUIRE(dev >= 0 && dev < kMaxDev, "device index");
    std::lock_guard<std::mutex> lk(g_scratch_mu);
    if (!g_scratch_ptr[dev])
        VINGS_CHECK_HIP(hipGetSymbolAddress((void**)&g_scratch_ptr[dev], HIP_SYMBOL(g_compact_scratch)));
    *out = g_scratch_ptr[dev];
    VINGS_REQUIRE(*out != nullptr, "scratch address");
    return VINGS_OK;
}

extern "C" int vings_splat_compact_static_prepare(void) {
    uint32_t* p = nullptr;
    return static_scratch(&p);
}

extern "C" int vings_splat_compact_plan_static(int N, const uint8_t* keep, int32_t* dst_index,
                                               int64_t* n_keep_device, void* stream) {
    VINGS_REQUIRE(N >= 0, "sizes");
    VINGS_REQUIRE(N <= kStaticMaxRows,
                  "more rows than the library's static scratch holds (vings_splat_compact_static_max_rows): "
                  "use vings_splat_compact_plan with a workspace");
    VINGS_REQUIRE(n_keep_device, "null pointer");
    hipStream_t s = (hipStream_t)stream;
    uint32_t* counts = nullptr;
    // (the first call on a GPU looks the address up, which is not a stream operation: a caller
    // that captures this call into a graph makes that first call, or
    // vings_splat_compact_static_prepare, before the capture begins)
    const int rc = static_scratch(&counts);
    if (rc != VINGS_OK) return rc;
    if (N == 0) {
        // nothing to scan: the count word is still written, on the stream
        VINGS_CHECK_HIP(hipMemsetAsync(n_keep_device, 0, sizeof(int64_t), s));
        return VINGS_OK;
    }
    VINGS_REQUIRE(keep && dst_index, "null pointer");
    const int nb = cdiv(N, 256);
    // every word the later kernels read (counts[0 .. nb]) is written by the two before them:
    // nothing of an earlier call's contents is assumed, nothing has to be cleared
    compact_count_kernel<<<nb, 256, 0, s>>>(N, keep, counts);
    VINGS_LAUNCH_CHECK();
    compact_scan_kernel<<<1, 1024, 0, s>>>(counts, nb, reinterpret_cast<long long*>(n_keep_device));
    VINGS_LAUNCH_CHECK();
    compact_index_kernel<<<nb, 256, 0, s>>>(N, keep, counts, dst_index);
    VINGS_LAUNCH_CHECK();
    return VINGS_OK;
}

extern "C" int vings_splat_compact_apply(int N, const int32_t* dst_index, int n_arrays,
                                         const void* const* src_host, void* const* dst_host,
                                         const int* row_bytes_host, void* stream) {
    VINGS_REQUIRE(N >= 0 && n_arrays > 0 && n_arrays <= kMaxArrays, "1..24 arrays");
    if (N == 0) return VINGS_OK;
    VINGS_REQUIRE(dst_index && src_host && dst_host && row_bytes_host, "null pointer");
    CompactArrays A;
    A.n = n_arrays;
    for (int a = 0; a < n_arrays; ++a) {
        VINGS_REQUIRE(src_host[a] && dst_host[a] && row_bytes_host[a] > 0, "array pointers");
        VINGS_REQUIRE(src_host[a] != dst_host[a], "compaction is out of place");
        if ((row_bytes_host[a] & 3) == 0)
            VINGS_REQUIRE((((uintptr_t)src_host[a] | (uintptr_t)dst_host[a]) & 3) == 0,
                          "word rows must be 4-byte aligned");
        A.src[a] = reinterpret_cast<const char*>(src_host[a]);
        A.dst[a] = reinterpret_cast<char*>(dst_host[a]);
        A.row_bytes[a] = row_bytes_host[a];
    }
    ProfScope prof(PROF_GLUE, (hipStream_t)stream);
    compact_apply_kernel<<<cdiv(N, 256), 256, 0, (hipStream_t)stream>>>(N, dst_index, A);
    VINGS_LAUNCH_CHECK();
    return VINGS_OK;
}
