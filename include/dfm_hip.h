/*
 * dfm_hip.h -- C ABI of libdfm_hip.so: the MI355X (gfx950) plane-sweep
 * cost-volume path of Depth-from-Motion.
 *
 * Drop-in boundary.  The reference has no FFI of its own -- its hot path is
 * Python calling torch ops -- so each entry point below replaces the torch-op
 * sequence of one reference function and is what a maintainer binds (ctypes,
 * see INTEGRATION.md) from inside that function:
 *
 *   dfm_plane_sweep_fwd      <- build_dfm_cost
 *                               mmdet3d/models/backbones/dfm_backbone.py:217-314
 *   dfm_plane_sweep_bwd      <- autograd of the two F.grid_sample calls,
 *                               dfm_backbone.py:296-311
 *   dfm_point_sample_mv_fwd  <- point_sample x (frames x views) + view/frame
 *                               reduction, fusion_layers/point_fusion.py:14-106
 *                               and detectors/multiview_dfm.py:119-208
 *   dfm_frustum_to_voxel_fwd <- FrustumToVoxel.forward sampling stage
 *                               necks/feature_transformation.py:82-158
 *   dfm_depth_head_fwd       <- DepthHead.forward (upsample x4 trilinear,
 *                               softmax over depth, expectation)
 *                               dense_heads/depth_head.py:205-210
 *
 * Conventions
 *   - plain C: device pointers are `void*` / `float*` (hipMalloc'ed or a torch
 *     tensor's data_ptr()), sizes are explicit, no torch types.
 *   - every call is asynchronous on `stream` (a hipStream_t passed as void*;
 *     NULL = the null stream) and re-entrant: the library keeps no mutable
 *     launch state -- launch options travel with the call (dfm_sweep_opts),
 *     the only shared state is the mutex-guarded tuned-schedule cache and
 *     the bench-only dfm_profile_* event list.
 *   - return value: DFM_OK (0) or a negative dfm_status; never throws.
 *     dfm_last_error() returns a thread-local message for the last failure.
 *   - dtype: DFM_F32 or DFM_BF16 is the storage type of feature/volume
 *     tensors; coordinates, weights and accumulation are always fp32.
 *   - arithmetic: sampling coordinates follow the reference's fp32 operation
 *     order (one rounding per torch op, fma chains where torch.mm uses them)
 *     so DFM_F32 results are bit-identical to the reference's PyTorch-CPU
 *     output for finite coordinates (see DESIGN.md "Numerics").
 */
#ifndef DFM_HIP_H
#define DFM_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DFM_API __attribute__((visibility("default")))

typedef enum dfm_status {
    DFM_OK = 0,
    DFM_ERR_INVALID_ARG = -1,
    DFM_ERR_UNSUPPORTED = -2,
    DFM_ERR_WORKSPACE = -3,
    DFM_ERR_HIP = -4
} dfm_status;

typedef enum dfm_dtype { DFM_F32 = 0, DFM_BF16 = 1 } dfm_dtype;

/* ---------------------------------------------------------------------- */
/* library                                                                 */
/* ---------------------------------------------------------------------- */
DFM_API int dfm_version(void);            /* ABI version, currently 3     */
DFM_API const char *dfm_last_error(void); /* thread-local, never NULL     */

/* Per-call device timing of the plane-sweep build, measured with HIP events
 * recorded on the caller's stream around EVERYTHING a dfm_plane_sweep_fwd call
 * launches on the dense path: the re-blocking pack pass, the tile kernel and
 * the second-chance / direct / patch passes (rounds 1-4 left the pack pass out;
 * the strided-sweep kernels time their volume-writing kernel).  begin() arms up to
 * max_launches pairs; end() waits for them and returns the summed time and the count. */
DFM_API int dfm_profile_begin(int max_launches);
DFM_API int dfm_profile_end(double *total_ms, int *launches);

/* ---------------------------------------------------------------------- */
/* plane sweep (build_dfm_cost)                                            */
/* ---------------------------------------------------------------------- */

/* Geometry of one plane-sweep call.  Field meaning = the reference
 * arguments of build_dfm_cost (dfm_backbone.py:217-227). */
typedef struct dfm_sweep_desc {
    int32_t batch;      /* B                                               */
    int32_t channels;   /* C of cur/prev feats; output has 2C              */
    int32_t h_in, w_in; /* feature map size                                */
    int32_t num_depths; /* D                                               */
    int32_t h_out, w_out; /* round(h_in/csf), round(w_in/csf) (caller)     */
    float feat_sample_factor;
    float cost_sample_factor;
    float img_scale_factor;
    float crop_x, crop_y; /* img_crop_offset                               */
    float org_w;          /* img_shape[1], only used when flip             */
    int32_t flip;
    int32_t dtype;        /* dfm_dtype of cur/prev/out                     */
} dfm_sweep_desc;

/* Camera matrices on the device (no host round trip when the intrinsics are device tensors,
 * dfm_backbone.py:151-154): pads cam2img (B, rows, cols; rows, cols in {3,4}; row-major) to the
 * 4x4 that points_img2cam / points_cam2img build (rows :3 of a 4-row matrix, identity elsewhere;
 * utils.py:199-203, 239-240) and inverts it in fp32 (Gauss-Jordan, partial pivoting).
 * cam2img_4x4, cam2img_inv : (B, 16) fp32 -- the cam2img / cam2img_inv arguments below.
 * The inverse agrees with torch.inverse to fp32 rounding, not bit for bit (neither do two
 * LAPACK builds); callers that replay the reference's CPU result bit-exactly pass that inverse. */
DFM_API int dfm_camera_prepare(const float *cam2img, int32_t rows, int32_t cols, int32_t batch,
                               float *cam2img_4x4, float *cam2img_inv, void *stream);

/* Scratch the call needs (blocked copy of the two feature maps). */
DFM_API size_t dfm_plane_sweep_workspace_bytes(const dfm_sweep_desc *desc);

/*
 * cur, prev : (B, C, h_in, w_in) contiguous, desc->dtype          [device]
 * depths    : (D) fp32                                           [device]
 * cam2img   : (B, 16) fp32, ori_cam2img padded to 4x4, row major [device]
 * cam2img_inv: (B, 16) fp32, inverse of the above (the reference computes
 *             it with torch.inverse in fp32, utils.py:239-241)   [device]
 * cur2prev  : (B, 16) fp32 row major                              [device]
 * out       : (B, 2C, D, h_out, w_out) contiguous, desc->dtype    [device]
 * workspace : >= dfm_plane_sweep_workspace_bytes(desc), 256-B aligned
 *
 * Semantics for B > 1: sample b uses cam2img[b], cur2prev[b]; the
 * augmentation scalars in `desc` are shared (the reference's caller passes
 * img_metas[0]'s, dfm_backbone.py:169-172).  The reference loop itself is
 * only correct for B == 1 (:257-275); B > 1 here == the reference looped
 * over single-sample batches.
 */
DFM_API int dfm_plane_sweep_fwd(const dfm_sweep_desc *desc, const void *cur, const void *prev,
                                const float *depths, const float *cam2img,
                                const float *cam2img_inv, const float *cur2prev, void *out,
                                void *workspace, size_t workspace_bytes, void *stream);

/*
 * Same call, same values bit for bit, result written channels-last:
 *   out : (B, D, h_out, w_out, 2C) contiguous, i.e. a torch tensor of shape
 *         (B, 2C, D, h_out, w_out) in memory_format channels_last_3d -- what an
 *         NDHWC / implicit-GEMM Conv3d consumes, and one contiguous run of 2C values
 *         per lattice point for the HBM.  Needs channels % (16 / sizeof(T)) == 0.
 *   workspace : >= dfm_plane_sweep_cl_workspace_bytes(desc) (pixel-major copies of the
 *         two maps + one zero pixel), 256-B aligned.
 */
DFM_API size_t dfm_plane_sweep_cl_workspace_bytes(const dfm_sweep_desc *desc);
DFM_API int dfm_plane_sweep_fwd_channels_last(const dfm_sweep_desc *desc, const void *cur,
                                              const void *prev, const float *depths,
                                              const float *cam2img, const float *cam2img_inv,
                                              const float *cur2prev, void *out, void *workspace,
                                              size_t workspace_bytes, void *stream);
/* The same, with cur / prev ALSO channels-last: (B, h_in, w_in, C) contiguous (torch channels_last,
 * what an NHWC 2-D neck emits -- SURVEY.md 8f rank 3), 16-byte aligned.  They are the kernel's
 * pixel-major layout already and are sampled where they lie: no pack pass; the workspace only has to
 * hold one zero pixel (>= 256 + 4 * channels bytes is always enough). */
DFM_API int dfm_plane_sweep_fwd_nhwc(const dfm_sweep_desc *desc, const void *cur, const void *prev,
                                     const float *depths, const float *cam2img,
                                     const float *cam2img_inv, const float *cur2prev, void *out,
                                     void *workspace, size_t workspace_bytes, void *stream);
/* Channels-last (NHWC) cur / prev as above, out in the REFERENCE layout (B, 2C, D, h_out, w_out):
 * the strided-sweep kernel (pixel-major taps + LDS transpose) on the caller's maps, no pack pass.
 * Returns DFM_ERR_UNSUPPORTED for shapes that kernel does not cover (channels not whole 16-byte
 * blocks, h_out * w_out not a multiple of 16 / sizeof(T), out not 16-byte aligned): hand NCHW maps to
 * dfm_plane_sweep_fwd then. */
DFM_API int dfm_plane_sweep_fwd_from_nhwc(const dfm_sweep_desc *desc, const void *cur,
                                          const void *prev, const float *depths,
                                          const float *cam2img, const float *cam2img_inv,
                                          const float *cur2prev, void *out, void *workspace,
                                          size_t workspace_bytes, void *stream);

/*
 * Backward of the two bilinear samplings w.r.t. the feature maps.
 * grad_out : (B, 2C, D, h_out, w_out) desc->dtype
 * grad_cur, grad_prev : (B, C, h_in, w_in) FP32, must be zero-filled by the
 *   caller (the kernel accumulates with atomics); cast to bf16 by the caller
 *   if needed.
 */
DFM_API int dfm_plane_sweep_bwd(const dfm_sweep_desc *desc, const void *grad_out,
                                const float *depths, const float *cam2img,
                                const float *cam2img_inv, const float *cur2prev,
                                float *grad_cur, float *grad_prev, void *stream);

/* Debug/parity aid: the normalised sampling grids the reference hands to
 * F.grid_sample (dfm_backbone.py:291-294) for sample `b`:
 * cur_grid, prev_grid : (D*h_out*w_out, 2) fp32 [device]. */
DFM_API int dfm_plane_sweep_grid(const dfm_sweep_desc *desc, int32_t b, const float *depths,
                                 const float *cam2img, const float *cam2img_inv,
                                 const float *cur2prev, float *cur_grid, float *prev_grid,
                                 void *stream);

/* Which kernel the last dfm_plane_sweep_fwd on this thread dispatched:
 * 0 = none yet, 1 = lane-per-point gather kernel, 2 = LDS-staged tile kernel
 * (+ direct-tap pass over flagged tiles), 3 = tile kernel with direct taps, 4 = pixel-major
 * taps + LDS transpose (strided sweeps, one depth plane per workgroup), 5 = the same with the depth
 * axis walked per wave (what strided fp32 sweeps take by default; a shape it does not cover -- or a
 * build whose disassembly check failed, see build.py -- reports 4).  Thread-local. */
DFM_API int dfm_plane_sweep_last_kernel(void);
/* Which kernel the last plane-sweep BACKWARD call of this process launched (process-wide: autograd
 * runs backward functions on its own threads): 1 = lane-per-point scatter, 5 = LDS-atomic tile
 * kernel, 6 = matrix-product kernel (dense bf16 sweeps; the tile kernel keeps the prev map's
 * fast-moving near planes).  (dfm_plane_sweep_bwd_cur_nhwc does not change it.) */
DFM_API int dfm_plane_sweep_bwd_last_kernel(void);

/*
 * Launch options of ONE call (the library keeps no process-wide launch state).  Every field:
 * 0 = the library's default.
 *   kernel               1/2/3/4/5 = force that kernel (2 and 3 need D*h_out*w_out to be a multiple
 *                        of 16/sizeof(T), else the call uses 1; 4 needs whole 16-byte channel
 *                        blocks and h_out*w_out a multiple of 16/sizeof(T), else the default
 *                        dispatch applies).  default = 2 for dense sweeps (cost_sample_factor
 *                        < 1.5); strided ones (config K) take 4: taps from pixel-major maps
 *                        (one contiguous run of channels per tap), the workgroup's 256 points x
 *                        32/64 channels transposed through LDS and stored as 1 KiB runs per
 *                        channel plane -- or 3 where 4 does not apply; fp32 sweeps with whole
 *                        32-channel passes and 32-point tiles take 5 instead: a wave walks a run of
 *                        depth planes over its 32 points with each point's 2x2 footprint cached in
 *                        registers, and loads only the taps that changed since the previous plane
 *                        (config K: 2.3-2.8 of 8 per point and plane); 4 pins the per-plane kernel.
 *                        For the backward call: 1 = the lane-per-point scatter fallback.
 *   lanes_per_workgroup  128 | 256 | 512 | 1024 (tile kernels; default 256)
 *   lds_kib              LDS budget per workgroup, 4..160 (default: 52 serial body, 80 pipelined body
 *                        -- whose allocation is fixed at 80 KiB and whose two buffers get half the
 *                        budget each); a tile whose feature rows do not fit gets a second chance
 *                        with 144 KiB and is else redone with direct taps
 *   blocks_per_group     16-byte channel blocks one workgroup sweeps (default: all)
 *   planes_per_workgroup consecutive depth planes that share one workgroup's staged rows
 *                        (default 2; rounded down to a divisor of lanes/64)
 *   bands_per_chunk      workgroup order: adjacent bands of a depth plane scheduled back to
 *                        back before the next depth group.  1 (default) keeps all depth planes
 *                        of one band resident together (best L2 reuse of the staged rows);
 *                        larger values make the resident workgroups write longer contiguous
 *                        runs of every channel plane (profiles/archive/r01_store_microbench3.txt)
 *   points_per_lane      16/sizeof(T) (default: one 16-byte store per channel), or 4 with
 *                        DFM_BF16 and 512/1024 lanes: 8-byte stores, half the registers per
 *                        lane, twice the waves per CU for the same tile
 *   pipeline             body of the LDS tile kernel: 1 = serial (stage a channel block's rows,
 *                        barrier, blend + store, barrier), 2 (default) = pipelined: two LDS
 *                        buffers, the next block's rows fly under the blend, one barrier per
 *                        block, and no wait for the volume stores' acknowledgements
 *   store_align_points   8 | 16 | 32 | 64: tile boundaries of the LDS tile kernel are multiples of this
 *                        many lattice points of a channel plane's flat (d,h,w) index, i.e. every run of
 *                        stores starts and ends on a 16 / 32 / 64 / 128-byte boundary of bf16 data
 *                        (x2 for fp32) relative to the plane's first element (default 64)
 *   pair_stores          points_per_lane = 4 with DFM_BF16: 1 (default) = neighbouring lanes trade their
 *                        8-byte halves and store one 16-byte vector per channel pair, 2 = 8-byte stores
 *   unpack               DFM_BF16, LDS tile kernel: 1 (default) = the taps are unpacked to fp32 by the matrix
 *                        core (two selecting v_mfma_f32_16x16x32_bf16 per tap, exact), 2 = by VALU shifts.
 *                        Maps that hold a non-finite, denormal or -0 value always take the VALU form.
 */
typedef struct dfm_sweep_opts {
    int32_t kernel;
    int32_t lanes_per_workgroup;
    int32_t lds_kib;
    int32_t blocks_per_group;
    int32_t planes_per_workgroup;
    int32_t bands_per_chunk;
    int32_t points_per_lane;
    int32_t pipeline;
    int32_t store_align_points;
    int32_t pair_stores;
    int32_t unpack;
    int32_t reserved[1];
} dfm_sweep_opts;

/* dfm_plane_sweep_fwd with explicit launch options.  opts == NULL is dfm_plane_sweep_fwd: the
 * schedule cached for this (device, shape) by dfm_plane_sweep_autotune if there is one; else,
 * for a volume >= 1 GB that takes the LDS-staged kernel, the call runs the autotuner first
 * (once per device and shape; synchronous; `out` is valid afterwards; DFM_AUTOTUNE=0 in the
 * environment or an active stream capture disables it); else the defaults above. */
DFM_API int dfm_plane_sweep_fwd_opts(const dfm_sweep_desc *desc, const void *cur, const void *prev,
                                     const float *depths, const float *cam2img,
                                     const float *cam2img_inv, const float *cur2prev, void *out,
                                     void *workspace, size_t workspace_bytes, void *stream,
                                     const dfm_sweep_opts *opts);
/* Backward with options: only `kernel` is read -- 1 = lane-per-point scatter kernel, 5 = LDS-atomic
 * tile kernel for both maps, 0 / 6 = default (dense bf16 sweeps with w_out >= 32, h_out >= 2: the
 * matrix-product kernel of plane_sweep_bwd_mfma.hip, whose weights are rounded to bf16 like the
 * gradients; everything else: the tile kernel, fp32 weights). */
DFM_API int dfm_plane_sweep_bwd_opts(const dfm_sweep_desc *desc, const void *grad_out,
                                     const float *depths, const float *cam2img,
                                     const float *cam2img_inv, const float *cur2prev,
                                     float *grad_cur, float *grad_prev, void *stream,
                                     const dfm_sweep_opts *opts);
/* The same backward with the gradient volume stored channels-last, (B, D, h_out, w_out, 2C) in memory
 * (torch channels_last_3d: what the NDHWC aggregation stack's backward hands over).  With a workspace
 * of the volume's size (16-byte aligned) the gradient is first re-laid (B, 2C, D, h, w) by an LDS-tile
 * transpose at copy speed; workspace == NULL: read in place (no extra memory, slower: the lanes of the
 * backward are lattice points).  DFM_ERR_UNSUPPORTED when neither applies (convert and call
 * dfm_plane_sweep_bwd). */
DFM_API int dfm_plane_sweep_bwd_channels_last(const dfm_sweep_desc *desc, const void *grad_out,
                                              const float *depths, const float *cam2img,
                                              const float *cam2img_inv, const float *cur2prev,
                                              float *grad_cur, float *grad_prev, void *workspace,
                                              size_t workspace_bytes, void *stream);
/* Backward of a STRIDED fp32 sweep (cost_sample_factor >= 2: config K), CUR map only, into a PIXEL-MAJOR
 * gradient map: grad_cur is (B, H, W, C) fp32 in memory (torch: a channels_last tensor of shape
 * (B, C, H, W)), zero-initialised by the caller and accumulated into with atomics; grad_out is the reference
 * layout (B, 2C, D, h_out, w_out), fp32, of which the first C channels are read.  The cur map's sample
 * position does not move with depth (it is the lattice pixel up to rounding noise): a wave keeps the 3x3 pixel
 * window a point's taps stay in over ALL depth planes in registers and writes it out once (csrc/
 * plane_sweep_cl.hip: sweep_bwdc_kernel).  The prev map comes from dfm_plane_sweep_bwd_opts with
 * opts->kernel = 8 (the LDS-atomic tile kernel, prev map only).  DFM_ERR_UNSUPPORTED unless fp32,
 * channels % 32 == 0 and h_out * w_out % 16 == 0: the caller then uses dfm_plane_sweep_bwd for both maps.
 * Replaces autograd of the first F.grid_sample call of build_dfm_cost (reference dfm_backbone.py:296-303). */
DFM_API int dfm_plane_sweep_bwd_cur_nhwc(const dfm_sweep_desc *desc, const void *grad_out, const float *depths,
                                         const float *cam2img, const float *cam2img_inv,
                                         const float *cur2prev, float *grad_cur, void *stream);
/* Backward of a STRIDED fp32 sweep, PREV map only, as a gather (csrc/plane_sweep_bwd_gather.hip): a lane owns
 * a map pixel, walks the depth planes, finds through the plane's inverse homography (fitted on the device from
 * the forward map's own values at the lattice corners, fp64) the lattice points whose footprint can hold the
 * pixel, confirms each with the forward's fp32 arithmetic, gathers the 32 channels of the gradient volume and
 * STORES the sums: grad_prev is (B, C, H, W) fp32 in the reference layout, overwritten (planes the fit cannot
 * vouch for -- behind the camera, lattice step under ~1.1 pixels -- are added with atomics by a second kernel
 * of the same call).  grad_out: (B, 2C, D, h_out, w_out) fp32, of which the last C channels are read.
 * workspace: >= dfm_plane_sweep_bwd_prev_gather_workspace_bytes(desc) (12 floats per sample and plane).
 * DFM_ERR_UNSUPPORTED unless fp32, channels % 32 == 0 and cost_sample_factor >= 2.  Replaces autograd of the
 * second F.grid_sample call of build_dfm_cost (reference dfm_backbone.py:304-311); reports 9 through
 * dfm_plane_sweep_bwd_last_kernel. */
DFM_API size_t dfm_plane_sweep_bwd_prev_gather_workspace_bytes(const dfm_sweep_desc *desc);
/* The general form of the same kernel: half = 0 the CUR map (channels [0, C) of the volume), 1 the PREV map;
 * the gradient volume in desc->dtype (DFM_F32 | DFM_BF16), in the reference layout (grad_channels_last == 0) or
 * channels-last (B, D, h_out, w_out, 2C) -- what the NDHWC aggregation stack's backward hands over: a hit is then
 * one contiguous run of 32 channels, read in place, no re-layout pass and no scratch of the volume's size;
 * grad_map fp32, overwritten, in the reference layout (B, C, H, W) or pixel-major (B, H, W, C)
 * (map_pixel_major != 0: torch's channels_last, what the NHWC necks' backward wants).  Same workspace.
 * DFM_ERR_UNSUPPORTED unless channels % 32 == 0 and cost_sample_factor >= 2 (16-byte aligned channels-last
 * buffers). */
DFM_API int dfm_plane_sweep_bwd_gather(const dfm_sweep_desc *desc, int32_t half, const void *grad_out,
                                       int32_t grad_channels_last, const float *depths, const float *cam2img,
                                       const float *cam2img_inv, const float *cur2prev, float *grad_map,
                                       int32_t map_pixel_major, void *workspace, size_t workspace_bytes,
                                       void *stream);
DFM_API int dfm_plane_sweep_bwd_prev_gather(const dfm_sweep_desc *desc, const void *grad_out, const float *depths,
                                            const float *cam2img, const float *cam2img_inv,
                                            const float *cur2prev, float *grad_prev, void *workspace,
                                            size_t workspace_bytes, void *stream);
/* Times the candidate launch shapes / workgroup orders of the LDS-staged kernel with the caller's
 * own arguments (a few launches per candidate; SYNCHRONOUS, `out` is overwritten with valid
 * results) and caches the fastest for this (device, problem shape); later dfm_plane_sweep_fwd
 * calls of that shape use it.  *chosen (may be NULL) receives it.  Thread-safe. */
DFM_API int dfm_plane_sweep_autotune(const dfm_sweep_desc *desc, const void *cur, const void *prev,
                                     const float *depths, const float *cam2img,
                                     const float *cam2img_inv, const float *cur2prev, void *out,
                                     void *workspace, size_t workspace_bytes, void *stream,
                                     dfm_sweep_opts *chosen);
/* The cached choice for desc's shape on the current device: returns 1 and fills *opts, or 0
 * (nothing cached; *opts zeroed = defaults), or a negative dfm_status. */
DFM_API int dfm_plane_sweep_tuning(const dfm_sweep_desc *desc, dfm_sweep_opts *opts);
/* Measurement aid (bench.py): writes zeros into `out` (batch x planes x plane_bytes, 16-byte aligned) in
 * the tile kernel's store pattern -- one run of run_bytes (0 = the kernel's 4 KiB; multiples of 4096) per
 * plane per workgroup, a workgroup walking planes_per_workgroup planes (0 = all: the tile kernel) -- so a
 * bench line can record what this part sustains for that stream (parts differ by ~25 % on it while a
 * linear fill does not). */
DFM_API int dfm_store_probe(void *out, int32_t batch, int32_t planes, int64_t plane_bytes, int32_t run_bytes,
                            int32_t planes_per_workgroup, void *stream);

/* Part diagnostics (bench.py `part`): every CU runs `iterations` dependent FMAs per lane; out3[0] = shader
 * clock cycles (s_memtime) and out3[1] = 100 MHz reference ticks (s_memrealtime) that workgroup 0 saw go
 * by: cycles / ticks / 10 = the shader clock in GHz this part sustains under a VALU load.  out3: 3 x u64
 * on the device. */
DFM_API int dfm_clock_probe(void *out3, int32_t iterations, void *stream);
DFM_API void dfm_plane_sweep_reset_tuning(void);

/* ---------------------------------------------------------------------- */
/* multi-view voxel lifting (point_sample x views x frames + reduction)    */
/* ---------------------------------------------------------------------- */

/* Field meaning = the arguments of point_sample (point_fusion.py:14-27) as
 * MultiViewDfM.feature_transformation passes them (multiview_dfm.py:130-170). */
typedef struct dfm_mv_desc {
    int32_t num_views;   /* Nv                                                  */
    int32_t num_frames;  /* F (1 + num_ref_frames)                              */
    int32_t channels;    /* C of every view's feature map                       */
    int32_t feat_h, feat_w;
    int32_t nx, ny, nz;  /* volume output (C*F', nx, ny, nz); point index =
                          * (z*ny + y)*nx + x (AlignedAnchor3DRangeGenerator
                          * order).  nz == 0: flat output (num_points, C*F').   */
    int64_t num_points;
    float scale_x, scale_y; /* img_scale_factor (w, h)                          */
    float crop_x, crop_y;   /* img_crop_offset                                  */
    int32_t flip;
    float pad_h, pad_w;     /* img_pad_shape (img_meta['input_shape'])          */
    int32_t mode;           /* 0 nearest (aligned=False), 1 bilinear            */
    int32_t aggregate;      /* 0 'mean' over frames (F' = 1), 1 'concat' (F'=F) */
    int32_t valid_sample;   /* 1: valid-count semantics (MultiViewDfM default);
                             * 0: plain grid_sample output, single view only
                             * (point_sample(valid_flag=False))               */
    int32_t dtype;          /* dfm_dtype of feats and out                       */
    int32_t out_channels_last; /* forward only, nz > 0: out is (nx, ny, nz, C*F') in memory, a torch
                             * tensor (C*F', nx, ny, nz) in channels_last_3d -- the layout the
                             * NDHWC / MFMA neck convolutions read, no conversion copy         */
    int32_t feats_channels_last; /* forward only: feats is (F*Nv, feat_h, feat_w, C) in memory (a torch
                             * channels_last image backbone / neck): sampled in place, no pixel-major
                             * copy, no workspace; needs C to be whole 16-byte blocks               */
} dfm_mv_desc;

DFM_API size_t dfm_point_sample_mv_workspace_bytes(const dfm_mv_desc *desc);

/*
 * feats  : (F*Nv, C, feat_h, feat_w) contiguous, frame-major        [device]
 * points : (num_points, 3) fp32 LiDAR coordinates                    [device]
 * proj   : (F*Nv, 16) fp32 lidar2img per (frame, view), row major    [device]
 * ori_w  : (F*Nv) fp32, img_shape[1] of each view (used when flip)   [device]
 * out    : volume or flat, see desc                                  [device]
 * valid  : optional (num_points) bytes, 1 where any view sees the point
 * valid_sample=True semantics: a view contributes only where 0 < x < pad_w,
 * 0 < y < pad_h and depth > 0; sums are divided by the number of valid views
 * (clamped to 1).
 */
DFM_API int dfm_point_sample_mv_fwd(const dfm_mv_desc *desc, const void *feats,
                                    const float *points, const float *proj, const float *ori_w,
                                    void *out, unsigned char *valid, void *workspace,
                                    size_t workspace_bytes, void *stream);
/* The whole batch in ONE launch (the multi-view configs' channels-last path): descs[batch] differ only in
 * their image transform (scale / crop / flip / pad); feats (batch, F*Nv, feat_h, feat_w, C) channels-last
 * views sampled in place, proj (batch, F*Nv, 16), ori_w (batch, F*Nv), out (batch, num_points, C*F') rows
 * = a (batch, C*F', nx, ny, nz) channels_last_3d volume, valid (batch, num_points) or NULL;
 * points (num_points, 3) shared (points_per_sample = 0) or (batch, num_points, 3).  Several lanes per
 * voxel: contiguous tap loads and row stores.  Covers nearest sampling with valid_sample semantics and
 * 4 / 8 / 16 16-byte channel blocks (C = 64 bf16: 8); DFM_ERR_UNSUPPORTED otherwise -- call
 * dfm_point_sample_mv_fwd per sample.  Bit-identical results. */
DFM_API int dfm_point_sample_mv_fwd_batched(const dfm_mv_desc *descs, int32_t batch, const void *feats,
                                            const float *points, int32_t points_per_sample, const float *proj,
                                            const float *ori_w, void *out, unsigned char *valid, void *stream);
/* Backward w.r.t. the view features.  grad_out has the layout of `out`;
 * grad_feats (F*Nv, C, feat_h, feat_w) is FP32, zero-filled by the caller. */
/* workspace (optional): >= dfm_point_sample_mv_bwd_workspace_bytes(desc) bytes for the
 * pixel-major gradient accumulator (F*Nv, feat_h*feat_w, C) fp32; NULL selects the
 * slower lane-per-voxel scatter. */
DFM_API size_t dfm_point_sample_mv_bwd_workspace_bytes(const dfm_mv_desc *desc);
DFM_API int dfm_point_sample_mv_bwd(const dfm_mv_desc *desc, const void *grad_out,
                                    const float *points, const float *proj, const float *ori_w,
                                    float *grad_feats, void *workspace, size_t workspace_bytes,
                                    void *stream);

/* ---------------------------------------------------------------------- */
/* FrustumToVoxel sampling stage                                           */
/* ---------------------------------------------------------------------- */

/* necks/feature_transformation.py:82-158.  Sizes of the three sources and of
 * the voxel grid; pad_* = img_metas[0]['pad_shape'] (the reference uses sample
 * 0's for the whole batch, :101); depth_min / depth_span = depth_cfg
 * ['depth_min'] and fp32(depth_max - depth_min). */
typedef struct dfm_f2v_desc {
    int32_t batch;
    int32_t channels;      /* C of stereo_feat                               */
    int32_t d, h, w;       /* stereo_feat (B, C, d, h, w)                     */
    int32_t ds, hs, ws;    /* stereo_feat_softmax (B, 1, ds, hs, ws)          */
    int32_t sem_channels;  /* Cs of cur_sem_feats, 0 = cat_img_feature False  */
    int32_t hsem, wsem;    /* cur_sem_feats (B, Cs, hsem, wsem)               */
    int32_t nz, ny, nx;    /* coordinates_3d (nz, ny, nx, 3)                  */
    float pad_h, pad_w;
    float depth_min, depth_span;
    int32_t dtype;         /* dfm_dtype of the three sources and of out       */
    int32_t stereo_channels_last; /* 1: stereo is (B, d, h, w, C) in memory (torch
                            * channels_last_3d, what an NDHWC Conv3d stack hands over):
                            * it is sampled in place, no pixel-major copy is made   */
    int32_t out_channels_last; /* forward only: out is (B, nz, ny, nx, C + Cs) in memory (torch
                            * channels_last_3d), what voxel_convs' MFMA convolution reads;
                            * needs channel counts of whole 16-byte blocks                   */
    int32_t stereo_atten;  /* 1: stereo_atten_feat=True, Voxel *= pred_disp (feature_transformation.py:141) */
    int32_t no_sem_atten;  /* 1: sem_atten_feat=False, Voxel_2D is NOT weighted by pred_disp (:154);
                            * both 0 = the shipped config; when neither attention is on, softmax may
                            * be NULL (the reference never samples it, :133)                           */
    int32_t sem_channels_last; /* forward only, with the 16-byte-block kernel: cur_sem_feats is
                            * (B, hsem, wsem, Cs) in memory (torch channels_last, what an NHWC 2-D neck
                            * emits): sampled in place, no pixel-major copy                            */
} dfm_f2v_desc;

/*
 * stereo, softmax, sem : contiguous, desc->dtype                    [device]
 * coords   : (nz, ny, nx, 3) fp32 pseudo-LiDAR voxel centres        [device]
 * cam2img  : (B, 16) fp32, img_meta['cam2img'] as 4x4 (rows 0..2)   [device]
 * out      : (B, C + Cs, nz, ny, nx) = cat(Voxel, Voxel_2D) of the reference,
 *            the input of voxel_convs (sem_atten_feat=True,
 *            stereo_atten_feat=False: the shipped config)
 * workspace: >= dfm_frustum_to_voxel_workspace_bytes(desc) bytes, 256-byte
 *            aligned; holds stereo_feat and cur_sem_feats re-laid pixel-major
 *            ([d][h][w][C], [h][w][Cs]) for this call                  [device]
 */
DFM_API size_t dfm_frustum_to_voxel_workspace_bytes(const dfm_f2v_desc *desc);
DFM_API int dfm_frustum_to_voxel_fwd(const dfm_f2v_desc *desc, const void *stereo,
                                     const void *softmax, const void *sem, const float *coords,
                                     const float *cam2img, void *out, void *workspace,
                                     size_t workspace_bytes, void *stream);
/* DepthHead fused into the sampling (inference): instead of the materialised
 * stereo_feat_softmax = softmax(Upsample_x4(cost)) (dense_heads/depth_head.py:205-207) the kernel
 * takes the low-resolution cost (B, 1, ds/scale, hs/scale, ws/scale) [desc->dtype] and the per-column
 * softmax statistics col_max / col_sum (B, hs, ws) fp32 from dfm_depth_head_stats_fwd, and evaluates
 * the distribution at the corners each voxel touches with the depth-head kernel's own arithmetic:
 * out is bit-identical to dfm_depth_head_fwd followed by dfm_frustum_to_voxel_fwd, and none of the
 * three (B, 1, ds, hs, ws) tensors exists.  desc->ds/hs/ws = the (virtual) distribution's size. */
DFM_API int dfm_frustum_to_voxel_fused_fwd(const dfm_f2v_desc *desc, const void *stereo,
                                           const void *cost, const float *col_max,
                                           const float *col_sum, int32_t head_scale, const void *sem,
                                           const float *coords, const float *cam2img, void *out,
                                           void *workspace, size_t workspace_bytes, void *stream);
/* Backward w.r.t. stereo_feat and cur_sem_feats (the depth distribution is
 * detached in the reference, :136).  grad_out: dtype, in the layout the forward wrote
 * `out` in -- (B, C+Cs, nz, ny, nx), or channels-last (B, nz, ny, nx, C+Cs) when
 * desc->out_channels_last (since round 5: an NDHWC voxel_convs backward hands its gradient
 * over in that layout and it is read in place; rounds 1-4 took the planar form only);
 * grad_stereo (B,C,d,h,w) and grad_sem (B,Cs,hsem,wsem): FP32, zero-filled by
 * the caller, accumulated with atomics. */
/* workspace (optional): >= dfm_frustum_to_voxel_bwd_workspace_bytes(desc) bytes of
 * scratch for the pixel-major gradient accumulators ((B, d*h*w, C) and
 * (B, hsem*wsem, Cs) fp32); NULL selects the slower lane-per-voxel scatter. */
DFM_API size_t dfm_frustum_to_voxel_bwd_workspace_bytes(const dfm_f2v_desc *desc);
DFM_API int dfm_frustum_to_voxel_bwd(const dfm_f2v_desc *desc, const void *grad_out,
                                     const void *softmax, const float *coords,
                                     const float *cam2img, float *grad_stereo, float *grad_sem,
                                     void *workspace, size_t workspace_bytes, void *stream);
/* The backward as a GATHER (round 5; csrc/frustum_to_voxel.hip: f2v_bwd_gather_kernel): a lane owns a pixel of
 * the cost volume and a run of depth planes, finds through the REGULAR voxel grid -- grid6, HOST memory:
 * {x0, dx, y0, dy, z0, dz} with coords[(iz * ny + iy) * nx + ix] == (x0 + ix dx, y0 + iy dy, z0 + iz dz), checked
 * by the caller -- the voxels whose trilinear footprint holds each cell, confirms every one with the forward's own
 * arithmetic, gathers the voxel's gradient row and STORES the cell's sum: grad_stereo is OVERWRITTEN (no
 * zero fill, no atomics); grad_sem (zero-filled by the caller) receives 32 atomics per lane and depth chunk
 * instead of 128 per voxel.  grad_out in the forward output's layout (desc->out_channels_last).  The depth
 * distribution: `softmax` (materialised) or `cost` + column statistics + head_scale (fused head), whichever the
 * forward used; the other NULL.  DFM_ERR_UNSUPPORTED unless channels == 32, sem_channels in {0, 32} at the cost
 * volume's resolution with sem_atten (the scatter form then takes the call): config K's shapes. */
DFM_API size_t dfm_frustum_to_voxel_bwd_gather_workspace_bytes(const dfm_f2v_desc *desc);
DFM_API int dfm_frustum_to_voxel_bwd_gather(const dfm_f2v_desc *desc, const void *grad_out, const void *softmax,
                                            const void *cost, const float *col_max, const float *col_sum,
                                            int32_t head_scale, const float *coords, const float *grid6,
                                            const float *cam2img, float *grad_stereo, float *grad_sem,
                                            void *workspace, size_t workspace_bytes, void *stream);
/* The same gather with the stereo gradient in the layout and type of a channels-last cost volume: grad_stereo is
 * (B, d, h, w, C) in memory (torch channels_last_3d of (B, C, d, h, w)), desc->dtype, 16-byte aligned, OVERWRITTEN:
 * a lane stores its 32 fp32 sums as one contiguous row, rounded once -- the bits the planar fp32 form followed by a
 * conversion gives.  The NDHWC stack's autograd adds this gradient to the one the prediction convolution's backward
 * produces in that layout (the feature volume feeds pred_stereo, dfm_backbone.py:120-127, AND FrustumToVoxel, dfm.py:317-319); the planar fp32 result
 * cost a zero fill, a conversion and a strided addition there.  grad_sem as above (fp32, planar, accumulated). */
DFM_API int dfm_frustum_to_voxel_bwd_gather_cl(const dfm_f2v_desc *desc, const void *grad_out, const void *softmax,
                                               const void *cost, const float *col_max, const float *col_sum,
                                               int32_t head_scale, const float *coords, const float *grid6,
                                               const float *cam2img, void *grad_stereo, float *grad_sem,
                                               void *workspace, size_t workspace_bytes, void *stream);
/* The same backward with the depth head fused (training): pred_disp, which scales the gradients of the
 * attended branches, is evaluated from the low-resolution cost + column statistics exactly as
 * dfm_frustum_to_voxel_fused_fwd does; no (B, 1, ds, hs, ws) tensor is read. */
DFM_API int dfm_frustum_to_voxel_fused_bwd(const dfm_f2v_desc *desc, const void *grad_out, const void *cost,
                                           const float *col_max, const float *col_sum, int32_t head_scale,
                                           const float *coords, const float *cam2img, float *grad_stereo,
                                           float *grad_sem, void *workspace, size_t workspace_bytes,
                                           void *stream);

/* ---------------------------------------------------------------------- */
/* voxel_sample (voxel volume -> frustum), point_fusion.py:324-410          */
/* ---------------------------------------------------------------------- */
typedef struct dfm_vs_desc {
    int32_t channels;        /* C of voxel_features (1, C, nx, ny, nz)           */
    int32_t nx, ny, nz;
    int32_t num_depths;      /* len(depth_samples[::downsample_factor])         */
    int32_t h_out, w_out;    /* round(img_pad_shape / downsample_factor)        */
    float downsample_factor;
    float scale_x, scale_y, crop_x, crop_y;
    int32_t flip;
    float ori_w;             /* img_shape[1]                                    */
    float voxel_range[6];
    float voxel_size[3];
    float proj_inv[16];      /* fp32 inverse of the 4x4 lidar2img (utils.py:241) */
    int32_t mode;            /* 1 trilinear (aligned), 0 nearest                */
    int32_t dtype;
} dfm_vs_desc;

/* voxel_features (1,C,nx,ny,nz), depths (num_depths) fp32 -> out (1,C,D,h_out,w_out) */
DFM_API int dfm_voxel_sample_fwd(const dfm_vs_desc *desc, const void *voxel_features,
                                 const float *depths, void *out, void *stream);
/* Backward w.r.t. the voxel features (the reference op is differentiable through F.grid_sample,
 * point_fusion.py:396-410): grad_out (1, C, D, h_out, w_out) dtype -> grad_voxel_features
 * (1, C, nx, ny, nz) FP32, zero-filled by the caller (accumulated with atomics). */
DFM_API int dfm_voxel_sample_bwd(const dfm_vs_desc *desc, const void *grad_out, const float *depths,
                                 float *grad_voxel_features, void *stream);

/* ---------------------------------------------------------------------- */
/* DepthHead.forward (with_convs=False), dense_heads/depth_head.py:205-210  */
/* ---------------------------------------------------------------------- */
/*
 * cost          : (B, 1, d, h, w) mono/stereo cost, dtype            [device]
 * depth_samples : (scale*d) fp32 bin centres (DfM.prepare_depth)      [device]
 * depth_volumes : (B, 1, scale*d, scale*h, scale*w) trilinear x scale upsample
 *                 (align_corners=True), bit-exact vs torch CPU
 * softmax       : same shape, softmax over depth
 * depth_preds   : (B, 1, scale*h, scale*w) = sum(softmax * depth_samples)
 */
DFM_API int dfm_depth_head_fwd(int32_t batch, int32_t d, int32_t h, int32_t w, int32_t scale,
                               int32_t dtype, const void *cost, const float *depth_samples,
                               void *depth_volumes, void *softmax, void *depth_preds,
                               void *stream);
/* The statistics-only pass of the same kernel: no volume is written; col_max / col_sum
 * (B, scale*h, scale*w) fp32 receive the maximum of each upsampled depth column and the sum of
 * exp(logit - max) over it (what dfm_frustum_to_voxel_fused_fwd needs to evaluate the softmax at
 * any lattice point); depth_preds (B, 1, scale*h, scale*w) as above, or NULL. */
DFM_API int dfm_depth_head_stats_fwd(int32_t batch, int32_t d, int32_t h, int32_t w, int32_t scale,
                                     int32_t dtype, const void *cost, const float *depth_samples,
                                     float *col_max, float *col_sum, void *depth_preds,
                                     void *stream);
/* Backward: any of the three incoming gradients may be NULL; grad_cost
 * (B,1,d,h,w) is FP32, zero-filled by the caller. */
DFM_API int dfm_depth_head_bwd(int32_t batch, int32_t d, int32_t h, int32_t w, int32_t scale,
                               int32_t dtype, const void *cost, const float *depth_samples,
                               const void *grad_volumes, const void *grad_softmax,
                               const void *grad_preds, float *grad_cost, void *stream);

/* ---------------------------------------------------------------------- */
/* Plane sweep fused into dres0 / dres0_mono (csrc/sweep_conv.hip)            */
/* Replaces, for 32-channel bf16 feature maps, the sequence                   */
/*   cost_raw = build_dfm_cost(cur, prev, ...)          dfm_backbone.py:161-172 */
/*   dres0.conv(cost_raw)        Conv3d(64 -> 32, 3, 1, 1)   dfm_backbone.py:175  */
/*   dres0_mono.conv(cost_raw[:, :32])  Conv3d(32 -> 32)     dfm_backbone.py:189  */
/* without materialising the (B, 64, D, h_out, w_out) volume: the sampler fills */
/* the convolution's LDS block.                                                 */
/* ---------------------------------------------------------------------- */
/* Bytes of the packed-weight buffer (4 wave roles x 54 MFMA fragments + a zero page). */
DFM_API size_t dfm_sweep_conv_weight_bytes(void);
/* w_stereo : (32, 64, 3, 3, 3), w_mono : (32, 32, 3, 3, 3), contiguous, DFM_F32 or DFM_BF16 [device] */
DFM_API int dfm_sweep_conv_pack_weights(const void *w_stereo, const void *w_mono, int32_t weight_dtype,
                                        void *packed, void *stream);
/* statistics partials per (sample, channel) a forward call with this descriptor / depth_chunk emits */
DFM_API int dfm_sweep_conv_stats_splits(const dfm_sweep_desc *desc, int32_t depth_chunk);
/*
 * desc            : as dfm_plane_sweep_fwd; channels must be 32 and dtype DFM_BF16
 *                   (DFM_ERR_UNSUPPORTED otherwise: run the unfused sequence)
 * cur/prev_nhwc   : (B, h_in, w_in, 32) bf16, pixel-major (torch channels_last)     [device]
 * depths, cam2img, cam2img_inv, cur2prev : as dfm_plane_sweep_fwd                    [device]
 * y_stereo, y_mono: (B, D, h_out, w_out, 32) bf16 NDHWC: the two convolution outputs BEFORE
 *                   GroupNorm / ReLU (fp32 accumulation over 27 x 64 / 27 x 32 products of the
 *                   bf16-rounded samples -- the values the unfused bf16 volume would hold)
 * stats_*         : fp32 [B][32][splits][3]: count / mean / M2 of the stored values per (sample,
 *                   channel, workgroup), consumed by dfm_group_norm_apply_channels_last
 * depth_chunk     : output planes one workgroup walks (0 = chosen from the shape)
 */
DFM_API int dfm_sweep_conv_fwd(const dfm_sweep_desc *desc, const void *cur_nhwc, const void *prev_nhwc,
                               const float *depths, const float *cam2img, const float *cam2img_inv,
                               const float *cur2prev, const void *packed_weights, void *y_stereo,
                               void *y_mono, float *stats_stereo, float *stats_mono, int32_t depth_chunk,
                               void *stream);

/* ---------------------------------------------------------------------- */
/* The gate of DfMBackbone.forward (dfm_backbone.py:136-141)                   */
/* ---------------------------------------------------------------------- */
/*
 * out = g * stereo + (1 - g) * mono,  g = sigmoid(W . cat(stereo, mono)) per pixel -- the reference's
 * cat + Conv2d(2D -> D, kernel 1, bias=False) + sigmoid + blend as ONE launch (inference; fp32 arithmetic,
 * one rounding at the store).
 * stereo, mono, out : (batch, num_depths, hw) contiguous, DFM_F32 or DFM_BF16 (`dtype`)      [device]
 * weight            : (num_depths, 2 * num_depths) row-major, DFM_F32 or DFM_BF16, packed once per weight
 *                     version into dfm_cost_gate_weight_bytes(num_depths) bytes (fp32, plane-major) [device]
 * num_depths <= 96 (DFM_ERR_UNSUPPORTED / 0 bytes otherwise: run the torch sequence).
 */
DFM_API size_t dfm_cost_gate_weight_bytes(int32_t num_depths);
DFM_API int dfm_cost_gate_pack_weights(const void *weight, int32_t weight_dtype, int32_t num_depths,
                                       void *packed, void *stream);
DFM_API int dfm_cost_gate_fwd(int32_t batch, int32_t num_depths, int64_t hw, int32_t dtype,
                              const void *stereo, const void *mono, const void *packed_weights,
                              void *out, void *stream);
/* Round 6: the same gate on the matrix cores for bf16 costs and a bf16-exact weight (csrc/cost_gate.hip:
 * out[d][p] = sum_k W[d][k] x[k][p] as v_mfma_f32_32x32x16_bf16, a wave loads its 32 pixels' 2D values once; same
 * epilogue, same single rounding).  packed: dfm_cost_gate_mfma_weight_bytes(D) bytes of A-operand fragments written
 * by dfm_cost_gate_mfma_pack_weights (an fp32 weight is rounded to bf16 there: use the entry above for fp32
 * parameters).  stereo / mono / out: (batch, D, hw) bf16. */
DFM_API size_t dfm_cost_gate_mfma_weight_bytes(int32_t num_depths);
DFM_API int dfm_cost_gate_mfma_pack_weights(const void *weight, int32_t weight_dtype, int32_t num_depths,
                                            void *packed, void *stream);
DFM_API int dfm_cost_gate_mfma_fwd(int32_t batch, int32_t num_depths, int64_t hw, const void *stereo,
                                   const void *mono, const void *packed_weights, void *out, void *stream);

/* ---------------------------------------------------------------------- */
/* MFMA Conv3d 3x3x3, stride 1, pad 1, 32 -> 32 channels, NDHWC bf16         */
/* (ConvModule / convbn_3d of the aggregation stacks: dfm_backbone.py:50-128, */
/*  utils/conv_modules.py:27-43)                                              */
/* ---------------------------------------------------------------------- */
/* Bytes of the packed-weight buffer (54 MFMA fragments + a zero page). */
DFM_API size_t dfm_conv3d_k3_c32_weight_bytes(void);
/* weight : (32, cin_total, 3, 3, 3) contiguous, DFM_F32 or DFM_BF16 [device]; packs the 32 input
 * channels [cin_offset, cin_offset + 32) into `packed` (register-fragment order, bf16).
 * transposed != 0: the weights of the backward-data convolution instead (channels swapped, taps
 * mirrored): dfm_conv3d_k3_c32_fwd(grad_out, packed) is then grad_in of those 32 channels. */
DFM_API int dfm_conv3d_k3_c32_pack_weights(const void *weight, int32_t weight_dtype,
                                           int32_t cin_total, int32_t cin_offset,
                                           int32_t transposed, void *packed, void *stream);
/*
 * x       : (n, d, h, w, 32) bf16, channels-last                       [device]
 * acc_in  : NULL, or a fp32 partial (n, d, h, w, 32) to start from (a 64-channel input runs as
 *           two calls: the first with out_f32 = 1, the second with acc_in = that partial)
 * out     : (n, d, h, w, 32) bf16 -- or fp32 when out_f32 != 0
 * relu    : != 0 applies max(., 0) before the store
 * depth_chunk : output planes one workgroup walks (0 = chosen from the shape)
 * stats   : NULL, or fp32 [n][32][splits][3] (splits = dfm_conv3d_k3_c32_stats_splits(...) with the
 *           same sizes and depth_chunk): count / mean / M2 of the stored bf16 values per (sample,
 *           channel, producing wave) -- the per-channel GroupNorm statistics of the layer that
 *           follows, consumed by dfm_group_norm_apply_channels_last (bf16 output only)
 * fp32 accumulation over the 27 x 32 products of a voxel (v_mfma_f32_32x32x16_bf16).
 */
DFM_API int dfm_conv3d_k3_c32_stats_splits(int32_t n, int32_t d, int32_t h, int32_t w,
                                           int32_t depth_chunk);
DFM_API int dfm_conv3d_k3_c32_fwd(int32_t n, int32_t d, int32_t h, int32_t w, const void *x,
                                  const void *packed_weights, const float *acc_in, void *out,
                                  int32_t out_f32, int32_t relu, int32_t depth_chunk, float *stats,
                                  void *stream);

/* Same call on a 32-channel SLICE of a wider channels-last tensor: x points at the slice's first
 * channel, x_channel_stride = channels of the wide tensor (elements between consecutive pixels,
 * multiple of 8).  The stereo dres0 (64 -> 32) reads the two halves of the cost volume, the mono
 * dres0 its first half (dfm_backbone.py:175,189), in place. */
DFM_API int dfm_conv3d_k3_c32_fwd_strided(int32_t n, int32_t d, int32_t h, int32_t w, const void *x,
                                          int32_t x_channel_stride, const void *packed_weights,
                                          const float *acc_in, void *out, int32_t out_f32,
                                          int32_t relu, int32_t depth_chunk, float *stats,
                                          void *stream);
/* ... whose bf16 output is a 32-channel slice of a wider channels-last tensor too (round 6): `out` points at the
 * slice's first channel of the first pixel, out_channel_stride (>= 32, a multiple of 8) elements between pixels; no
 * fp32 partial, no statistics.  Backward-data of a 32 k -> 32 convolution (dfm_backbone.py:175: dres0 reads the 2 C
 * channels of the cost volume) writes its k halves straight into the (N, D, H, W, 32 k) gradient. */
DFM_API int dfm_conv3d_k3_c32_fwd_slices(int32_t n, int32_t d, int32_t h, int32_t w, const void *x,
                                         int32_t x_channel_stride, const void *packed_weights, void *out,
                                         int32_t out_channel_stride, int32_t relu, int32_t depth_chunk,
                                         void *stream);
/* The 32 -> 1 prediction convolutions (dfm_backbone.py:120-127, Conv3d(32, 1, 3, 1, 1)): the same
 * kernel with packed weights whose output rows 1..31 are zero (pack a (32, 32, 3, 3, 3) tensor with
 * the (1, 32, 3, 3, 3) weight in row 0); only channel 0 is stored: out = (n, d, h, w) bf16. */
DFM_API int dfm_conv3d_k3_c32_to1_fwd(int32_t n, int32_t d, int32_t h, int32_t w, const void *x,
                                      const void *packed_weights, void *out, int32_t relu,
                                      int32_t depth_chunk, void *stream);

/* Backward of the prediction heads' Conv3d(32 -> 1, 3, 1, 1) (mmdet3d/models/backbones/dfm_backbone.py:120-127), round 6
 * (csrc/conv3d_to1_bwd.hip): both gradients as matrix products with the 27 taps as a matrix dimension, no padded
 * tensors.
 *   grad_out : (n, d, h, w) bf16 -- the (n, 1, d, h, w) gradient of the convolution's output [device]
 *   weight   : (1, 32, 3, 3, 3) contiguous, weight_dtype DFM_F32 | DFM_BF16
 *   grad_x   : (n, d, h, w, 32) bf16 channels-last, 16-byte aligned, OVERWRITTEN
 *   x        : (n, d, h, w, 32) bf16 channels-last (the convolution's input)
 *   grad_weight : (1, 32, 3, 3, 3) in out_dtype, OVERWRITTEN (fp32 sums over per-wave partials added in a fixed order)
 *   workspace   : >= dfm_conv3d_to1_wgrad_workspace_bytes() */
DFM_API int dfm_conv3d_to1_bwd_data(int32_t n, int32_t d, int32_t h, int32_t w, const void *grad_out,
                                    const void *weight, int32_t weight_dtype, void *grad_x, void *stream);
DFM_API size_t dfm_conv3d_to1_wgrad_workspace_bytes(void);
DFM_API int dfm_conv3d_to1_wgrad(int32_t n, int32_t d, int32_t h, int32_t w, const void *x, const void *grad_out,
                                 void *grad_weight, int32_t out_dtype, void *workspace, size_t workspace_bytes,
                                 void *stream);

/* Round 6 -- the prediction head's tail as ONE pass (csrc/conv3d_to1n.hip): GroupNorm(+ReLU) of the 32-channel
 * volume applied ON LOAD inside the 32 -> 1 convolution that consumes it (dfm_backbone.py:120-127:
 * ConvModule(32 -> 32, GN, ReLU) -> Conv3d(32, 1, 3, 1, 1)); the normalised volume is never written.
 *   dfm_group_norm_coefficients: moment partials [n][groups][splits][3] (count / mean / M2, what
 *     dfm_conv3d_k3_c32_fwd emits as `stats`) -> coef fp32 [n][c][2] = (a, b) of y = x * a + b, with the merge and
 *     the arithmetic of dfm_group_norm_apply_channels_last (mean / rstd per group, a = rstd * gamma,
 *     b = beta - mean * a): a consumer that computes bf16(relu(x * a + b)) reproduces that pass bit for bit.
 *   dfm_conv3d_to1_norm_fwd: x (n, d, h, w, 32) bf16 RAW convolution output, coef as above, weight
 *     (1, 32, 3, 3, 3) fp32 / bf16 (unpacked: the 27 x 32 operand is built in registers), out (n, d, h, w) bf16;
 *     relu_in: ReLU behind the normalisation; relu_out: ReLU of the result; depth_chunk 0 = automatic.
 *     Zero padding pads the NORMALISED tensor, as the unfused sequence does. */
DFM_API int dfm_group_norm_coefficients(int32_t n, int32_t c, int32_t groups, float eps, const float *partials,
                                        int32_t splits, const float *gamma, const float *beta, float *coef,
                                        void *stream);
DFM_API int dfm_conv3d_to1_norm_fwd(int32_t n, int32_t d, int32_t h, int32_t w, const void *x, const float *coef,
                                    const void *weight, int32_t weight_dtype, int32_t relu_in, int32_t relu_out,
                                    void *out, int32_t depth_chunk, void *stream);

/* ---------------------------------------------------------------------- */
/* General MFMA Conv3d / ConvTranspose3d 3x3x3, NDHWC bf16, channels = 32 k    */
/* (hourglass conv1..conv6: utils/conv_modules.py:73-149; Conv3d+BN3d+ReLU     */
/*  stacks of the voxel necks: necks/imvoxel_neck.py:26-55,85-117,             */
/*  necks/dfm_neck.py:29-95)                                                   */
/* ---------------------------------------------------------------------- */
typedef struct dfm_conv3d_desc {
    int32_t n;              /* batch                                                         */
    int32_t cin, cout;      /* multiples of 32                                               */
    int32_t in_size[3];     /* (d, h, w) of the input                                        */
    int32_t out_size[3];    /* (d, h, w) of the output (checked against stride / padding)    */
    int32_t stride[3];      /* 1 | 2 per axis (ignored on transposed axes)                   */
    int32_t padding[3];     /* 0..2 per axis (ignored on transposed axes)                    */
    int32_t transposed[3];  /* != 0: this axis is the x2 transposed convolution (kernel 3,   */
                            /* stride 2, padding 1, output_padding 1: out = 2 in)            */
    int32_t relu;           /* != 0: max(., 0) before the store                              */
    int32_t in_channel_stride; /* elements between consecutive input pixels; 0 = cin.  > cin: x  */
                            /* is a channel slice of a wider channels-last tensor (pointer at   */
                            /* its first channel; multiple of 8 so pieces stay 16-byte aligned) */
    int32_t kernel1[3];     /* != 0: the kernel has extent 1 along this axis (padding 0, not      */
                            /* transposed; out = (in - 1) / stride + 1): kernel (1, 3, 3) is a 2-D  */
                            /* convolution of an NHWC tensor seen as a depth-1 volume.  The packed  */
                            /* weights keep 27 taps; the axis uses the centre index only            */
} dfm_conv3d_desc;
/* Bytes of the packed-weight buffer (27 * cin * cout bf16 in fragment order + a zero page). */
DFM_API size_t dfm_conv3d_g_weight_bytes(int32_t cin, int32_t cout);
/* weight : dim0 x dim1 x 27 contiguous, DFM_F32 or DFM_BF16 [device].  The kernel multiplies
 * A[row][k] (row = output channel, k = input channel) per tap t = (kd, kh, kw); this packs
 *   A[row][k](t) = swap ? weight[k][row][t'] : weight[row][k][t'],
 *   t' = t with the kernel index mirrored (k -> 2 - k) on the axes in the bit mask `flip`
 *   (bit 2 = d, bit 1 = h, bit 0 = w).
 *   nn.Conv3d forward            (weight (cout, cin, 27)):  swap 0, flip 0
 *   nn.ConvTranspose3d forward   (weight (cin, cout, 27)):  swap 1, flip 0, all axes transposed
 *   backward-data of nn.Conv3d   : swap 1; a stride-1 axis becomes a correlation axis with
 *                                  padding 2 - p and its flip bit set, a stride-2 axis (padding 1,
 *                                  even extent) a transposed axis without flip
 *   backward-data of nn.ConvTranspose3d: swap 0, flip 0, stride 2 / padding 1 on every axis */
DFM_API int dfm_conv3d_g_pack_weights(const void *weight, int32_t weight_dtype, int32_t cin,
                                      int32_t cout, int32_t swap, int32_t flip, void *packed,
                                      void *stream);
/* the same for a 2-D weight (dim0, dim1, 3, 3) (round 6): its 9 taps land in the centre depth slice of the 27, the
 * other two slices are zeros -- what a 2-D convolution run as a depth-1 volume (kernel1[0] = 1) reads, without the
 * caller embedding the weight into a (.., 3, 3, 3) tensor first (SPPUNetNeck / BEVHourglass: spp_unet_neck.py:93-119,
 * bev_hourglass.py:36-137); flip bits 2 (h) and 1 (w) */
DFM_API int dfm_conv3d_g_pack_weights_2d(const void *weight, int32_t weight_dtype, int32_t cin,
                                         int32_t cout, int32_t swap, int32_t flip, void *packed,
                                         void *stream);
/*
 * x        : (n, d, h, w, cin) bf16, channels-last (pixel stride desc->in_channel_stride) [device]
 * scale / shift : NULL, or fp32 [cout]: y = conv * scale[c] + shift[c] (a folded BatchNorm3d in
 *            eval mode, or a bias) applied to the fp32 accumulator
 * residual : NULL, or (n, od, oh, ow, cout) bf16 added after scale / shift (ResModule identity)
 * out      : (n, od, oh, ow, cout) bf16; order: scale/shift -> + residual -> ReLU -> round to bf16
 */
DFM_API int dfm_conv3d_g_fwd(const dfm_conv3d_desc *desc, const void *x, const void *packed_weights,
                             const float *scale, const float *shift, const void *residual,
                             void *out, void *stream);
/* Split-precision mode for fp32 models: the same kernel, but the fp32 accumulators are stored as they
 * are into `out` (fp32, (N, D', H', W', cout)), plus `acc_in` (same shape, may be NULL, may equal `out`).
 * A fp32 convolution y = conv(x, w) is then a few launches on bf16 operands accumulated in fp32.
 * The Python host's default (conv3d.set_fp32_mode('split'), three pieces per operand, all 24 significand
 * bits): x = x0 + x1 + x2, w = w0 + w1 + w2,  y = sum over i + j <= 2 of conv(x_i, w_j) -- SIX launches, the
 * dropped terms are 2^-27 of a product.  set_fp32_mode('split2'): two pieces, THREE launches,
 *   y = conv(x0, w0) + conv(x1, w0) + conv(x0, w1)   (dropped term 2^-18 of a product).
 * Either way: what nn.Conv3d / ConvTranspose3d (dfm_backbone.py:175-201, conv_modules.py:73-149,
 * imvoxel_neck.py:26-55) compute at the reference's default precision, without MIOpen.  A non-finite
 * operand value travels in the first piece only (its remainders are zero, not Inf - Inf).
 * desc->relu must be 0; no scale / shift / residual. */
DFM_API int dfm_conv3d_g_fwd_f32(const dfm_conv3d_desc *desc, const void *x, const void *packed_weights,
                                 const float *acc_in, float *out, void *stream);
/* The tiling dfm_conv3d_g_fwd uses for desc: {pixel fragments per wave, channel fragments per
 * wave, tile d, tile h, tile w, staged pixels, LDS bytes, workgroups}. */
DFM_API int dfm_conv3d_g_plan(const dfm_conv3d_desc *desc, int64_t *plan8);

/* Weight gradient of the same convolutions (backward-weight; MFMA, csrc/conv3d_wgrad.hip):
 *   out[a][b][kd][kh][kw] = sum over output positions o of g[o][a] * x[o * stride - padding + k][b]
 * g : (n, g_size, a) bf16, x : (n, x_size, b) bf16, both channels-last with explicit element strides
 * (n, d, h, w; multiples of 8; channels contiguous -- a channel slice of a wider tensor is fine);
 * out : (a, b, 27) fp32, overwritten.
 *   nn.Conv3d          : g = grad_output, x = input, the convolution's stride / padding
 *                        -> grad_weight (C_out, C_in, 3, 3, 3)
 *   nn.ConvTranspose3d (kernel 3, stride 2, padding 1, output_padding 1): g = input, x = grad_output,
 *                        stride 2, padding 1 -> grad_weight (C_in, C_out, 3, 3, 3)
 * workspace: >= dfm_conv3d_wgrad_workspace_bytes(desc) (per-workgroup partial sums, summed by a second
 * kernel: deterministic, no atomics). */
typedef struct dfm_conv3d_wgrad_desc {
    int32_t n;
    int32_t a, b;           /* channels of g (rows of out) and x (columns of out), multiples of 32 */
    int32_t g_size[3];      /* (d, h, w) of g                                                      */
    int32_t x_size[3];      /* (d, h, w) of x                                                      */
    int32_t stride[3];      /* 1 | 2 per axis                                                      */
    int32_t padding[3];     /* 0..2 per axis                                                       */
    int64_t g_stride[4];    /* element strides of g: n, d, h, w                                    */
    int64_t x_stride[4];    /* element strides of x: n, d, h, w                                    */
} dfm_conv3d_wgrad_desc;
DFM_API size_t dfm_conv3d_wgrad_workspace_bytes(const dfm_conv3d_wgrad_desc *desc);
DFM_API int dfm_conv3d_wgrad(const dfm_conv3d_wgrad_desc *desc, const void *g, const void *x, float *out,
                             void *workspace, size_t workspace_bytes, void *stream);
/* the same with `out` in out_dtype (DFM_F32 | DFM_BF16; (a, b, 27), overwritten): a bf16 parameter's gradient in the
 * parameter's own type, the fp32 sums rounded once in the reduction kernel (round 6: no conversion launch) */
DFM_API int dfm_conv3d_wgrad_to(const dfm_conv3d_wgrad_desc *desc, const void *g, const void *x, void *out,
                                int32_t out_dtype, void *workspace, size_t workspace_bytes, void *stream);

/* ---------------------------------------------------------------------- */
/* DepthHead.loss, dense_heads/depth_head.py:75-188 (called at dfm.py:348) */
/* ---------------------------------------------------------------------- */
typedef enum dfm_depth_loss_target {
    DFM_DL_LINEAR = 0,    /* ce, balanced_ce, focal, balanced_focal: 1 - min(|ds-gt|/interval, 1) */
    DFM_DL_HARD = 1,      /* hard_ce: the above thresholded at 0.5                               */
    DFM_DL_GAUSSIAN = 2,  /* gaussian_<sigma>: exp(-0.5 dist^2/sigma^2) / max(sum, 1)             */
    DFM_DL_LAPLACIAN = 3  /* laplacian_<sigma>: exp(-dist/sigma) / max(sum, 1)                    */
} dfm_depth_loss_target;

typedef struct dfm_depth_loss_desc {
    int32_t batch;       /* B*N images                                          */
    int32_t num_depths;  /* D bins of depth_volumes                             */
    int32_t h, w;
    int32_t target;      /* dfm_depth_loss_target                               */
    int32_t focal;       /* != 0: weight every bin by alpha*(1-p)^gamma (depth_head.py:131-139) */
    float min_depth, max_depth; /* valid gt: min_depth < gt < max_depth (:89)    */
    float interval;      /* depth_samples[1] - depth_samples[0] (:95)            */
    float sigma;         /* gaussian / laplacian                                 */
    float alpha, gamma;  /* focal                                                */
    int32_t dtype;       /* dfm_dtype of depth_volumes / grad_volumes            */
} dfm_depth_loss_desc;

/*
 * depth_volumes : (B, D, h, w) logits, dtype                          [device]
 * depth_img     : (B, h, w) fp32 ground-truth depth                   [device]
 * depth_samples : (D) fp32 bin centres                                [device]
 * pixel_loss    : (B, h, w) fp32: -sum_d p_d f(log_softmax_d) per valid pixel, 0 elsewhere --
 *                 the reference's `loss` before `.mean()` / the fg-bg weighted sum
 * valid         : (B, h, w) uint8: the reference's `mask`
 * The caller reduces: mean over valid pixels (ce, focal, hard_ce, gaussian, laplacian) or
 * (fg_weight*sum_fg + bg_weight*sum_bg)/n_valid (balanced_*), times loss_weight^2 (:186).
 */
DFM_API int dfm_depth_loss_fwd(const dfm_depth_loss_desc *desc, const void *depth_volumes,
                               const float *depth_img, const float *depth_samples,
                               float *pixel_loss, unsigned char *valid, void *stream);
/* grad_volumes (B, D, h, w) dtype = grad_pixel_loss[b,h,w] * d pixel_loss / d depth_volumes;
 * written for every element (zeros at invalid pixels). */
DFM_API int dfm_depth_loss_bwd(const dfm_depth_loss_desc *desc, const void *depth_volumes,
                               const float *depth_img, const float *depth_samples,
                               const float *grad_pixel_loss, void *grad_volumes, void *stream);
/* DepthHead.loss fused with the depth head (training): `cost` is the LOW-RESOLUTION
 * (B, 1, D/s, h/s, w/s) volume [desc->dtype] that DepthHead.forward upsamples (desc->num_depths / h / w are
 * the upsampled sizes, s = head_scale); the logits of a valid pixel's column are evaluated on the fly with
 * dfm_depth_head_fwd's arithmetic (pixel_loss is bit-identical to dfm_depth_loss_fwd on the materialised
 * depth_volumes), and the backward adds grad_pixel_loss * d pixel_loss / d logits through the transposed
 * upsample into grad_cost (B, 1, D/s, h/s, w/s) FP32, zero-filled (or pre-accumulated) by the caller --
 * what dfm_depth_loss_bwd + dfm_depth_head_bwd(grad_volumes) compute through two (B, D, h, w) tensors. */
DFM_API int dfm_depth_loss_fused_fwd(const dfm_depth_loss_desc *desc, const void *cost, int32_t head_scale,
                                     const float *depth_img, const float *depth_samples,
                                     float *pixel_loss, unsigned char *valid, void *stream);
DFM_API int dfm_depth_loss_fused_bwd(const dfm_depth_loss_desc *desc, const void *cost, int32_t head_scale,
                                     const float *depth_img, const float *depth_samples,
                                     const float *grad_pixel_loss, float *grad_cost, void *stream);

/* ---------------------------------------------------------------------- */
/* fused GroupNorm (+ReLU) of the aggregation stacks                        */
/* mmcv ConvModule(conv -> GN -> ReLU) at dfm_backbone.py:50-66,118-128,     */
/* feature_transformation.py:55-62; convbn_3d at utils/conv_modules.py:27-43 */
/* ---------------------------------------------------------------------- */
DFM_API size_t dfm_group_norm_workspace_bytes(int32_t n, int32_t c, int64_t spatial,
                                              int32_t groups);
/*
 * x, y      : (n, c, spatial) contiguous (NC(D)HW), dtype              [device]
 * gamma,beta: (c) fp32 affine parameters                               [device]
 * mean,rstd : (n*groups) fp32, written (saved for backward)            [device]
 * y = (x - mean_g) * rstd_g * gamma_c + beta_c, then max(y, 0) if relu != 0;
 * biased variance, rstd = 1/sqrt(var + eps) (torch.nn.GroupNorm semantics).
 */
DFM_API int dfm_group_norm_fwd(int32_t n, int32_t c, int64_t spatial, int32_t groups, float eps,
                               int32_t dtype, int32_t relu, const void *x, const float *gamma,
                               const float *beta, void *y, float *mean, float *rstd,
                               void *workspace, size_t workspace_bytes, void *stream);
/* Same op on channels-last data: x, y are (n, spatial, c) contiguous (a torch tensor in
 * memory_format channels_last_3d), the layout of the NDHWC convolutions around it.
 * Needs c = (16-byte vectors) x (a power of two), c <= 256. */
DFM_API int dfm_group_norm_fwd_channels_last(int32_t n, int32_t c, int64_t spatial, int32_t groups,
                                             float eps, int32_t dtype, int32_t relu, const void *x,
                                             const float *gamma, const float *beta, void *y,
                                             float *mean, float *rstd, void *workspace,
                                             size_t workspace_bytes, void *stream);
/* The normalisation pass alone, from statistics somebody else produced (the MFMA convolution's
 * epilogue): partials = fp32 [n][groups][splits][3] {count, mean, M2} (Chan-mergeable); they are
 * merged once (a small kernel, into the first n*groups*3 floats of `workspace`), then
 * y = (x - mean) * rstd * gamma + beta (+ReLU) like dfm_group_norm_fwd_channels_last, which also
 * fills mean / rstd for the backward.  One read and one write of the tensor instead of two reads. */
DFM_API int dfm_group_norm_apply_channels_last(int32_t n, int32_t c, int64_t spatial, int32_t groups,
                                               float eps, int32_t dtype, int32_t relu, const void *x,
                                               const float *gamma, const float *beta, void *y,
                                               float *mean, float *rstd, const float *partials,
                                               int32_t splits, void *workspace,
                                               size_t workspace_bytes, void *stream);
/* The two channels-last entry points with a fused residual: `residual` (NULL, or a tensor of x's
 * shape / dtype) is added after the affine map and before the ReLU,
 *   y = relu?((x - mean) * rstd * gamma + beta + residual)
 * -- the residual connections of the aggregation stacks (dfm_backbone.py:176,183,
 * utils/conv_modules.py:124-139) without a separate elementwise pass. */
DFM_API int dfm_group_norm_fwd_channels_last_res(int32_t n, int32_t c, int64_t spatial, int32_t groups,
                                                 float eps, int32_t dtype, int32_t relu, const void *x,
                                                 const float *gamma, const float *beta,
                                                 const void *residual, void *y, float *mean, float *rstd,
                                                 void *workspace, size_t workspace_bytes, void *stream);
DFM_API int dfm_group_norm_apply_channels_last_res(int32_t n, int32_t c, int64_t spatial, int32_t groups,
                                                   float eps, int32_t dtype, int32_t relu, const void *x,
                                                   const float *gamma, const float *beta,
                                                   const void *residual, void *y, float *mean, float *rstd,
                                                   const float *partials, int32_t splits, void *workspace,
                                                   size_t workspace_bytes, void *stream);
/* grad_gamma / grad_beta: (c) fp32, zero-filled by the caller; `y` is only
 * read when relu != 0 (mask y > 0). */
DFM_API int dfm_group_norm_bwd(int32_t n, int32_t c, int64_t spatial, int32_t groups,
                               int32_t dtype, int32_t relu, const void *grad_y, const void *x,
                               const void *y, const float *mean, const float *rstd,
                               const float *gamma, void *grad_x, float *grad_gamma,
                               float *grad_beta, void *workspace, size_t workspace_bytes,
                               void *stream);
/* The same backward on channels-last data (x, y, grad_y, grad_x: (n, spatial, c) contiguous): the
 * NDHWC stacks train without converting three tensors per layer to NC(D)HW and back.
 * grad_residual: NULL, or a tensor of grad_y's shape receiving grad_y behind the ReLU mask -- the
 * gradient of the residual input of dfm_group_norm_*_channels_last_res.  Same workspace size.
 * grad_gamma / grad_beta are OVERWRITTEN here (round 6: one workgroup per group sums over the batch and
 * stores; no zero fill by the caller, no atomics, a fixed order of additions). */
DFM_API int dfm_group_norm_bwd_channels_last(int32_t n, int32_t c, int64_t spatial, int32_t groups,
                                             int32_t dtype, int32_t relu, const void *grad_y,
                                             const void *x, const void *y, const float *mean,
                                             const float *rstd, const float *gamma, void *grad_x,
                                             void *grad_residual, float *grad_gamma, float *grad_beta,
                                             void *workspace, size_t workspace_bytes, void *stream);
/* dfm_group_norm_bwd_channels_last for y = relu(GroupNorm(x)) without a fused residual, y NOT kept (round 6): the ReLU
 * mask is recomputed from x with the forward's own expression (fma(x, rstd * gamma, beta - mean * rstd * gamma),
 * rounded through the storage type), so the statistics pass reads two tensors instead of three, the gradient pass
 * three instead of four, and the caller's autograd graph holds no second activation per layer
 * (mmdet3d/models/utils/conv_modules.py:27-43: every Conv3d + GN + ReLU block of the aggregation stacks).
 * beta: the norm's bias, fp32 [c] [device]; grad_gamma / grad_beta are OVERWRITTEN */
DFM_API int dfm_group_norm_bwd_channels_last_xmask(int32_t n, int32_t c, int64_t spatial, int32_t groups,
                                                   int32_t dtype, const void *grad_y, const void *x,
                                                   const float *mean, const float *rstd, const float *gamma,
                                                   const float *beta, void *grad_x, float *grad_gamma,
                                                   float *grad_beta, void *workspace, size_t workspace_bytes,
                                                   void *stream);

/* Backward of F.interpolate(mode='bilinear') on an NHWC map as a gather (round 6; the up-sampling steps of the 2-D
 * necks either side of the path: mmdet3d/models/necks/spp_unet_neck.py:60-70, 83-91, training only):
 *   gx[b, hi, wi, :] = sum_a sum_c row_w[hi][a] * col_w[wi][c] * gy[b, row_idx[hi][a], col_idx[wi][c], :]
 * gy : (n, h_out, w_out, c), gx : (n, h_in, w_in, c), `dtype`, 16-byte aligned, c whole 16-byte vectors; gx OVERWRITTEN
 * row_idx / row_w : [h_in][kh] the output rows interpolating from input row hi and their weights (padding: weight 0);
 * col_idx / col_w : [w_in][kw] likewise -- the non-zeros of the transposed 1-D interpolation matrices (the Python
 * host takes them from ATen's own forward, so the gradient is the adjoint of exactly what F.interpolate applied). */
DFM_API int dfm_bilinear_resize_bwd_nhwc(int32_t n, int32_t c, int32_t h_in, int32_t w_in, int32_t h_out,
                                         int32_t w_out, int32_t dtype, const void *gy, const int32_t *row_idx,
                                         const float *row_w, int32_t kh, const int32_t *col_idx, const float *col_w,
                                         int32_t kw, void *gx, void *stream);

/* AvgPool3d((k, 1, 1)) of FrustumToVoxel (necks/feature_transformation.py:167) on a channels-last volume, forward
 * and backward, one pass each (csrc/depth_pool.hip): the tensor as (outer, k, inner) contiguous -- for an
 * (N, C, D, H, W) channels_last_3d volume outer = N * D / k, inner = H * W * C -- y (outer, inner) the mean over
 * the middle axis in fp32, rounded once; grad_x[o][j][i] = grad_y[o][i] / k.  dtype DFM_F32 | DFM_BF16, inner in
 * whole 16-byte vectors, 16-byte aligned buffers. */
DFM_API int dfm_depth_pool_fwd(int64_t outer, int32_t k, int64_t inner, int32_t dtype, const void *x, void *y,
                               void *stream);
DFM_API int dfm_depth_pool_bwd(int64_t outer, int32_t k, int64_t inner, int32_t dtype, const void *grad_y,
                               void *grad_x, void *stream);

/* ---------------------------------------------------------------------- */
/* SPPUNetNeck: tail of the pyramid-pooling branches  */
/* ---------------------------------------------------------------------- */

/* necks/spp_unet_neck.py:60-70,97-106 (inference, bf16, NHWC): every branch's
 * ConvModule(in_channels -> spp_channels, 1x1, GroupNorm with one channel per group, ReLU) on its
 * pooled map, the bilinear up-sampling (align_corners=True) of the four results to (h, w) and the
 * concatenation behind the source maps -- two launches instead of ~25 on a few hundred pixels each. */
typedef struct dfm_spp_desc {
    int32_t batch;
    int32_t h, w;               /* size of the concatenated map (feats[start_level])            */
    int32_t num_sources;        /* <= 4 maps copied in front (feats[start_level:])               */
    int32_t source_channels[4]; /* multiples of 8                                                */
    int32_t num_branches;       /* <= 4                                                          */
    int32_t in_channels;        /* channels of the pooled maps (feats[-1])                       */
    int32_t spp_channels;       /* output channels of every branch: multiple of 8, <= 64         */
    int32_t pooled_h[4], pooled_w[4];
    float eps;                  /* GroupNorm eps                                                 */
} dfm_spp_desc;
DFM_API size_t dfm_spp_tail_workspace_bytes(const dfm_spp_desc *desc);
/* pooled[i]  : (batch, pooled_h[i], pooled_w[i], in_channels) bf16, the branch's window means   [device]
 * weight[i]  : (spp_channels, in_channels) fp32, the 1x1 convolution; gamma[i], beta[i]: (spp_channels) fp32
 * sources[i] : (batch, h, w, source_channels[i]) bf16 NHWC, 16-byte aligned
 * out        : (batch, h, w, sum(source_channels) + num_branches * spp_channels) bf16 NHWC =
 *              torch.cat((*sources, *upsampled_branches), 1) in channels_last
 * The four pointer arrays are HOST arrays of device pointers. */
DFM_API int dfm_spp_tail_fwd(const dfm_spp_desc *desc, const void *const *pooled, const float *const *weight,
                             const float *const *gamma, const float *const *beta,
                             const void *const *sources, void *out, void *workspace,
                             size_t workspace_bytes, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* DFM_HIP_H */
