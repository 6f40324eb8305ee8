/*
 * vings_hip.h -- C ABI of libvings_hip.so, the MI355X (gfx950) hot-path core
 * for VINGS-Mono: 2D-Gaussian surfel rasteriser forward/backward, row-masked
 * Adam, and the dense bundle-adjustment (DBA) core.
 *
 * The reference reaches this path through two Python extension modules
 * whose native sources are not vendored:
 *   diff_surfel_rasterization  (/root/reference/scripts/gaussian/gaussian_base.py:5-6)
 *   droid_backends             (/root/reference/scripts/frontend/depth_video.py:4)
 * Each entry point below names the reference call it serves.  The Python
 * drop-in modules in vings_mono_amd/dropin bind these symbols with ctypes.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer unless the name ends in _host
 *   - `stream` is a hipStream_t passed as void* (NULL = default stream)
 *   - tensors are dense, row-major, float32 unless stated
 *   - the caller owns all memory, including workspaces; no entry point
 *     allocates; only vings_surfel_project synchronises (it returns the
 *     instance count the caller needs to size the sort buffers)
 *   - return value 0 = success, negative = vings_status
 */
#ifndef VINGS_HIP_H
#define VINGS_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum vings_status {
    VINGS_OK = 0,
    VINGS_ERR_ARG = -1,      /* null pointer / bad size */
    VINGS_ERR_HIP = -2,      /* a HIP runtime call failed; see vings_last_error */
    VINGS_ERR_CAPACITY = -3, /* caller-provided buffer too small */
    VINGS_ERR_NUMERIC = -4   /* e.g. Cholesky breakdown */
} vings_status;

/* Human-readable text of the last failure on the calling thread. */
const char* vings_last_error(void);
/* ABI version; bumped when a signature changes. */
int vings_abi_version(void);

#define VINGS_GEOM_STRIDE 20 /* floats per packed splat record */
#define VINGS_AUX_CHANNELS 9 /* allmap channels */
#define VINGS_TILE 16

/* ------------------------------------------------------------------------
 * Surfel rasteriser forward, stage 1: projection + per-splat tile counts.
 * Serves GaussianRasterizer.forward (gaussian_base.py:193-203), settings
 * tuple built at gaussian_base.py:160-177.
 *
 * in : means3D[N,3] scales[N,2] rotations[N,4](w,x,y,z) opacities[N]
 *      colors[N,3] flow[N,2] or NULL; viewmatrix[16] projmatrix[16] as the
 *      flat storage of the reference's transposed matrices (cameras.py:58-60)
 * out: geom[N,20]  packed record {T[9], xy[2], normal[3], opacity, rgb[3], flow[2]}
 *                  (rows of culled splats are left untouched)
 *      depth[N] radii[N](int32) rect[N,4](int32 x0,y0,x1,y1 in tiles)
 *      tiles_touched[N](uint32)
 *      block_prefix[ceil(N/256)+1] (uint32) exclusive prefix of per-256-splat
 *                  instance counts; last element = total
 *      *num_instances_host  total number of (splat,tile) instances D
 * Synchronises `stream` once to return D.
 */
int vings_surfel_project(int N, int H, int W, float scale_modifier,
                         const float* means3D, const float* scales, const float* rotations,
                         const float* opacities, const float* colors, const float* flow,
                         const float* viewmatrix, const float* projmatrix,
                         float* geom, float* depth, int32_t* radii, int32_t* rect,
                         uint32_t* tiles_touched, uint32_t* block_prefix,
                         int64_t* num_instances_host, void* stream);

/* Bytes of scratch vings_surfel_bin needs for D instances. */
size_t vings_surfel_bin_workspace_bytes(int64_t D);

/* ------------------------------------------------------------------------
 * Stage 2: duplicate with keys (tile<<32 | depth bits), stable radix sort,
 * per-tile ranges.  Same reference call as above.
 * out: keys_sorted[D](uint64) vals_sorted[D](uint32 splat index)
 *      ranges[T,2](uint32 begin,end) with T = ceil(W/16)*ceil(H/16)
 * workspace: vings_surfel_bin_workspace_bytes(D) bytes, 256-byte aligned.
 */
int vings_surfel_bin(int N, int H, int W, int64_t D, const float* depth, const int32_t* rect,
                     const uint32_t* tiles_touched, const uint32_t* block_prefix,
                     uint64_t* keys_sorted, uint32_t* vals_sorted, uint32_t* ranges,
                     void* workspace, size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------
 * Stage 3: front-to-back alpha blend, one workgroup per 16x16 tile.
 * out: out_color[3,H,W]  allmap[9,H,W] (0 sum w*depth, 1 alpha, 2:5 sum w*normal,
 *      5 median depth, 6 distortion, 7:9 sum w*flow)  -- consumer:
 *      gaussian_base.py:205-219, gaussian_model.py:427
 *      final_T[3,H,W] (T, M1, M2) and n_contrib[2,H,W](uint32) are saved
 *      state for the backward pass.
 * pixel_mask[H*W] (uint8, 0 = skip pixel) may be NULL (gaussian_base.py:158).
 */
int vings_surfel_blend_forward(int H, int W, const uint32_t* ranges, const uint32_t* vals_sorted,
                               const float* geom, const float* bg, const uint8_t* pixel_mask,
                               float* out_color, float* allmap, float* final_T,
                               uint32_t* n_contrib, void* stream);

/* ------------------------------------------------------------------------
 * Backward of stage 3.  Triggered by total_loss.backward()
 * (gaussian_base.py:382).  dgeom[N,20] receives, per splat,
 * {dT[9], dxy[2], dnormal[3], dopacity, drgb[3], score_importance, score_error};
 * it is zeroed by this call.
 */
int vings_surfel_blend_backward(int N, int H, int W, const uint32_t* ranges,
                                const uint32_t* vals_sorted, const float* geom, const float* bg,
                                const uint8_t* pixel_mask, const float* final_T,
                                const uint32_t* n_contrib, const float* dL_dcolor,
                                const float* dL_dallmap, float* dgeom, void* stream);

/* ------------------------------------------------------------------------
 * Backward of stage 1: chains dgeom to the rasteriser inputs.
 * out: dL_dmeans3D[N,3] dL_dscales[N,2] dL_drotations[N,4] dL_dopacities[N]
 *      dL_dcolors[N,3]; optional dL_dmeans2D[N,3] (screen-space gradient,
 *      z = 0) and scores_grad[N,2] (gaussian_base.py:387) may be NULL.
 */
int vings_surfel_project_backward(int N, int H, int W, float scale_modifier,
                                  const float* means3D, const float* scales,
                                  const float* rotations, const float* viewmatrix,
                                  const float* projmatrix, const int32_t* radii,
                                  const float* dgeom, float* dL_dmeans3D, float* dL_dscales,
                                  float* dL_drotations, float* dL_dopacities, float* dL_dcolors,
                                  float* dL_dmeans2D, float* scores_grad, void* stream);

/* ------------------------------------------------------------------------
 * Row-masked Adam without bias correction:
 *   m = b1 m + (1-b1) g; v = b2 v + (1-b2) g^2; p -= lr m / (sqrt(v)+eps)
 * applied only to rows i with visible[i] != 0.  Serves
 * SparseGaussianAdam.step (gaussian_base.py:62,402).  `param` is [N,M].
 */
int vings_sparse_adam(float* param, const float* grad, float* exp_avg, float* exp_avg_sq,
                      const uint8_t* visible, float lr, float b1, float b2, float eps, int N,
                      int M, void* stream);

/* Same update for up to 8 parameter groups sharing one visibility mask in a
 * single launch (the five splat groups of gaussian_base.py:55-61). The
 * pointer/size arrays live in HOST memory. */
int vings_sparse_adam_multi(int n_groups, float* const* params_host,
                            const float* const* grads_host, float* const* exp_avgs_host,
                            float* const* exp_avg_sqs_host, const int* widths_host,
                            const float* lrs_host, const uint8_t* visible, float b1, float b2,
                            float eps, int N, void* stream);

/* ------------------------------------------------------------------------
 * Optional per-kernel timing with HIP events on the caller's stream (no
 * counterpart in the reference, whose only timers are the wandb brackets of
 * /root/reference/scripts/gaussian/wandb_utils.py:30-45).  enable(1) creates
 * an event pool once and resets the counters; read() synchronises on the
 * recorded events and returns the summed milliseconds / launch count of a tag.
 */
int vings_profile_enable(int on);
int vings_profile_num_tags(void);
const char* vings_profile_tag_name(int tag);
int vings_profile_read(int tag, double* total_ms, int64_t* count);

#ifdef __cplusplus
}
#endif
#endif /* VINGS_HIP_H */
