/*
 * floodunet.h -- C ABI of libfloodunet.so: the MI355X (gfx950) UNet segmentation training
 * path of the FloodPlanet st_water_seg pipeline.
 *
 * The reference has no native layer: its hot path is torch.nn modules driven by Lightning
 * ("C ABI (new; nothing to mirror in the reference)").  Each entry
 * point below therefore cites the *Python* call it replaces (paths relative to the
 * reference checkout):
 *
 *   fu_create / fu_destroy      UNet(n_channels, n_classes, bilinear)        st_water_seg/models/unet.py:80-98
 *                               built by WaterSegmentationModel._build_model st_water_seg/models/water_seg_model.py:79-85
 *   fu_param_info / fu_bind_*   UNet.state_dict() / .parameters()            (state-dict keys of unet.py:6-98)
 *   fu_forward[_srcs]           UNet.forward (+ the early-fusion concat)     st_water_seg/models/unet.py:100-111, ef_model.py:28-44
 *                               (train: nn.BatchNorm2d batch statistics; eval: running statistics,
 *                                water_seg_model.py:92-96)
 *   fu_loss_ce                  nn.CrossEntropyLoss(ignore_index) + nan_to_num + argmax + metric counts
 *                                                                            st_water_seg/models/water_seg_model.py:40,103-113
 *   fu_backward[_block]         loss.backward() issued by Lightning's automatic optimisation
 *                                                                            st_water_seg/fit.py:95-97
 *   fu_adam_step                optim.Adam(self.parameters(), lr).step()     st_water_seg/models/water_seg_model.py:198-205
 *   fu_augment                  torchvision hflip / vflip / rotate of sample_transforms + apply_transforms
 *                                                                            st_water_seg/datasets/base_dataset.py:494-555
 *   fu_block_param_range        (new) gradient bucket of one backward block, for RCCL all-reduce overlap
 *   fu_dp_* / fu_allreduce_*    (new) the bucketed gradient all-reduce itself, RCCL resolved at run time
 *   fu_op_*                     single operators for per-op parity tests (conv2d, batch_norm, max_pool2d,
 *                               upsample, cross_entropy as used in unet.py / water_seg_model.py)
 *
 * Conventions
 *   - every pointer argument that carries tensor data is a DEVICE pointer owned by the caller
 *     (borrowed for the duration of the call); the context owns workspaces and saved activations.
 *   - all work is enqueued on the hipStream_t passed as `stream` (void* so that this header
 *     needs no HIP include); nothing synchronises the device unless documented.
 *   - every function returns FU_OK (0) or an error code; fu_last_error() returns a thread-local
 *     message.  No C++ exception crosses this boundary.
 *   - one context per device per process; calls on one context must be serialised by the caller.
 */
#ifndef FLOODUNET_H_
#define FLOODUNET_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FU_ABI_VERSION 5
#define FU_MAX_ENCODERS 6

typedef struct fu_ctx fu_ctx;
typedef void* fu_stream; /* hipStream_t */

enum fu_status {
  FU_OK = 0,
  FU_ERR_INVALID = 1,     /* bad argument / shape */
  FU_ERR_HIP = 2,         /* a HIP runtime call failed */
  FU_ERR_STATE = 3,       /* call order violated (e.g. backward before forward) */
  FU_ERR_UNSUPPORTED = 4  /* valid request this build does not implement */
};

enum fu_precision {
  FU_F32 = 0, /* fp32 storage, fp32 MFMA (exact fmaf chains): the 1e-4 parity mode */
  FU_BF16 = 1, /* bf16 activations/weights, fp32 accumulate, fp32 master weights / BN / loss */
  FU_F16 = 2   /* IEEE fp16 activations / weights / gradient maps on the matrix cores (v_mfma_f32_32x32x16_f16, the bf16
                * rate), fp32 accumulation, fp32 master weights, BN statistics, loss and Dice reductions (BASELINE configs[3]:
                * "mixed fp16 with fp32 Dice reduction").  The gradient maps carry a power-of-two loss scale chosen on the
                * device from max|dL/dlogits| at the start of every backward and removed where parameter gradients are
                * written: the flat gradient buffer holds true gradients, no caller-side GradScaler is needed (a step whose
                * gradients overflowed all the same is skipped on the device: fu_fp16_guard_state).  Inputs are
                * expected in fp16's range (the reference scales every sensor to [0, 1], datasets/floodplanet.py:347-525). */
};

enum fu_loss_kind {
  FU_LOSS_CE = 0,      /* softmax cross entropy with ignore_index (the reference's loss) */
  FU_LOSS_BCE_DICE = 1 /* north-star extension: per-pixel BCE + soft Dice on class 1 (parity unpinned) */
};

typedef struct fu_config {
  int32_t struct_size;   /* = sizeof(fu_config) */
  int32_t n_channels;    /* sum of in_channels.values() (water_seg_model.py:81-84) */
  int32_t n_classes;     /* 3 for FloodPlanet (datasets/floodplanet.py:63) */
  int32_t base_channels; /* 64 = UNet; other widths = UNetEncoder/Decoder(base_feat_channels) */
  int32_t bilinear;      /* 1 = nn.Upsample(bilinear, align_corners=True) (default), 0 = ConvTranspose2d */
  int32_t max_batch;     /* workspace is sized for this many tiles */
  int32_t height, width; /* tile size (any size >= 16; odd sizes take the F.pad path of unet.py:57-62) */
  int32_t precision;     /* enum fu_precision */
  int32_t device;        /* HIP device ordinal */
  /* Late fusion (lf_model.py:29-92, feat_fusion='concat_conv'): n_encoders >= 1 builds one UNetEncoder per input over
   * consecutive channel windows of the input (enc_channels[e] channels each, summing to n_channels, in the order
   * lf_model.py:58-76 gathers them), five 1x1 fusion convs (n_encoders*fs -> fs at fs = 64,128,256,512,512 for
   * base 64) and one UNetDecoder.  Parameter names: encoders.<e>.*, concat_convs.<level>.*, decoder.* (the host side
   * maps <e> to the reference's ModuleDict keys).  0 = the plain UNet.  Needs bilinear = 1. */
  int32_t n_encoders;
  int32_t enc_channels[FU_MAX_ENCODERS];
} fu_config;

/* ---- lifetime ---------------------------------------------------------------------------- */
int fu_abi_version(void);
const char* fu_last_error(void);
int fu_create(const fu_config* cfg, fu_ctx** out);
int fu_destroy(fu_ctx* ctx);

/* ---- parameter table (canonical = reference state_dict order) ----------------------------- */
int fu_num_params(const fu_ctx* ctx);        /* trainable tensors (74 for the bilinear net) */
int64_t fu_total_param_elems(const fu_ctx* ctx);
/* name: state-dict key without the Lightning "model." prefix; shape: up to 4 dims (OIHW for convs);
 * flat_offset: element offset inside the flat parameter / gradient buffers. */
int fu_param_info(const fu_ctx* ctx, int index, const char** name, int32_t* ndim, int64_t shape[4],
                  int64_t* flat_offset);
int fu_num_bn(const fu_ctx* ctx);            /* BatchNorm2d layers (18) */
int64_t fu_total_bn_channels(const fu_ctx* ctx);
/* name: key prefix of the BN module ("inc.double_conv.1"); offset into the flat running buffers */
int fu_bn_info(const fu_ctx* ctx, int index, const char** name, int32_t* channels, int64_t* flat_offset);

/* Bind caller-owned flat device buffers.  params/grads: fp32 [fu_total_param_elems];
 * running_mean/running_var: fp32 [fu_total_bn_channels]; num_batches_tracked: int64 [fu_num_bn].
 * grads may be NULL for inference-only use. */
int fu_bind_buffers(fu_ctx* ctx, float* params, float* grads, float* running_mean, float* running_var,
                    int64_t* num_batches_tracked);
/* Tell the context the bound parameters were modified by someone else (torch optimiser,
 * load_state_dict): device-side packed copies are refreshed on the next forward. */
int fu_params_changed(fu_ctx* ctx);

/* ---- the hot path -------------------------------------------------------------------------- */
/* x: fp32 NCHW [batch, n_channels, height, width].  logits_out: fp32 NCHW [batch, n_classes, H, W] or NULL.
 * training != 0: batch-statistics BN, running buffers updated, activations kept for backward. */
int fu_forward(fu_ctx* ctx, const float* x, int batch, int training, float* logits_out, fu_stream stream);

/* The same with the input given as n_src (<= 8) fp32 NCHW tensors [batch, src_channels[k], H, W] that the model sees
 * side by side along the channel axis (sum src_channels = n_channels): the reference's
 * `torch.concat([images, dem, slope, ...], dim=1)` (ef_model.py:28-44) / stacked sensors (BASELINE configs[4]) without the
 * concatenated copy -- the channels are gathered by the NCHW -> NHWC conversion the first conv needs anyway.  srcs /
 * src_channels: HOST arrays. */
int fu_forward_srcs(fu_ctx* ctx, const float* const* srcs, const int32_t* src_channels, int n_src, int batch,
                    int training, float* logits_out, fu_stream stream);

/* Loss on the logits of the last fu_forward.  target: int64 [batch, H, W].
 * loss_out: device fp32 scalar (mean over non-ignored pixels; 0 when every pixel is ignored).
 * confusion_out: optional device int64 [n_classes*n_classes], M[t*n_classes+p] over non-ignored pixels
 * (argmax prediction), ADDED to the existing contents.  n_valid_out: optional device int64 scalar. */
int fu_loss_ce(fu_ctx* ctx, const int64_t* target, int ignore_index, float* loss_out, int64_t* confusion_out,
               int64_t* n_valid_out, fu_stream stream);
/* North-star extension (no reference counterpart; specification = oracle/unet_oracle.py:bce_dice_loss):
 * BCE on p = softmax(z)[1] vs [target == 1] + dice_weight * soft Dice, over target != ignore_index, fp32 wave-shuffle
 * reductions; stores its logits gradient for fu_backward like fu_loss_ce. */
int fu_loss_bce_dice(fu_ctx* ctx, const int64_t* target, int ignore_index, float dice_weight, float* loss_out,
                     fu_stream stream);

/* Backward of everything enqueued by the last training fu_forward.  dlogits: fp32 NCHW gradient w.r.t.
 * the logits, or NULL to use the gradient of the last fu_loss_* call.  Gradients are written
 * (not accumulated) into the bound flat gradient buffer. */
int fu_backward(fu_ctx* ctx, const float* dlogits, fu_stream stream);
/* The weight-gradient chain of every conv (wgrad, slab reduce, transpose) runs on a context-owned side stream,
 * concurrently with the data-gradient / BatchNorm-backward chain on the caller's stream; fu_backward joins the two
 * before it returns its work to the caller's stream order, fu_backward_block at the end of every block (the block's
 * gradients are then final for a bucketed all-reduce).  mode 0 keeps everything on the caller's stream (the default
 * when the environment variable FU_NO_SIDE_STREAM is set at fu_create, for bilinear = 0 and in fp32 mode); mode 1
 * (default) as described; mode 2: fu_backward_block does not join -- the caller calls fu_backward_join(stream) before
 * it consumes gradients on `stream` (e.g. once per all-reduce bucket instead of once per block). */
int fu_set_side_stream(fu_ctx* ctx, int mode);
/* The upstream gradient autograd hands to `loss.backward()` (fit.py:95-97) as a device scalar: the next backward
 * multiplies the stored loss gradient (dL/dlogits of the last fu_loss_* call) by *scale_dev on its way into the head --
 * once, on the logits gradient, instead of on every parameter gradient, and without a host read.  The stored gradient is
 * not modified: repeating the call replaces the factor, repeating the backward (retain_graph) repeats the result; the
 * factor is dropped by the next fu_forward / fu_loss_* call. */
int fu_scale_loss_grad(fu_ctx* ctx, const float* scale_dev, fu_stream stream);
int fu_backward_join(fu_ctx* ctx, fu_stream stream);
/* Mode 2, without stalling the compute stream: `waiter` (any other stream, e.g. the one a collective is launched from) waits
 * for every gradient-producing launch enqueued so far -- the work on `stream` and the weight-gradient chain on the side
 * stream.  Neither `stream` nor the side stream waits for anything: the backward blocks that follow start at once, the
 * bucket's all-reduce starts when both chains have reached this point.  (fu_allreduce_begin does the same internally.)
 * The last bucket still needs fu_backward_join before the optimizer step reads the gradients on `stream`. */
int fu_backward_fence(fu_ctx* ctx, fu_stream stream, fu_stream waiter);

/* The same, one block at a time in backward order: block 0 = outc, 1..4 = up4..up1, 5..8 = down4..down1,
 * 9 = inc.  After block k returns, the gradient range fu_block_param_range(k) is final on `stream`. */
int fu_num_blocks(const fu_ctx* ctx);
int fu_backward_block(fu_ctx* ctx, int block, const float* dlogits, fu_stream stream);
int fu_block_param_range(const fu_ctx* ctx, int block, int64_t* flat_offset, int64_t* numel);

/* torch.optim.Adam semantics (water_seg_model.py:198-205) on the bound flat buffers.  The moments are CALLER-owned
 * flat fp32 device buffers [fu_total_param_elems] in parameter order, zero-initialised by the caller (the state of
 * torch.optim.Adam: exp_avg, exp_avg_sq); they outlive the context, so re-creating a context for another tile size or
 * a larger batch does not reset the optimiser (ABI 2 kept them inside the context).  fu_adam_step without bound moments
 * returns FU_ERR_STATE.  step is 1-based.  grad_scale multiplies the gradient first (1/world_size after a sum
 * all-reduce).  The scalars are doubles, formed and rounded to float exactly as torch.optim.Adam forms its Python
 * scalars; every operation of the update rounds on its own in ATen's order, so that with identical inputs the result
 * is torch's. */
int fu_bind_adam_state(fu_ctx* ctx, float* exp_avg, float* exp_avg_sq);
int fu_adam_step(fu_ctx* ctx, double lr, double beta1, double beta2, double eps, int64_t step, double grad_scale,
                 fu_stream stream);
int fu_adam_state(fu_ctx* ctx, float** exp_avg, float** exp_avg_sq); /* the bound pointers (NULL when none) */
/* FU_F16 only: the device-side loss scale follows max|dL/dlogits|, but a layer deep in the chain (a BatchNorm with a tiny
 * variance) can still overflow an fp16 gradient map.  fu_adam_step[_dev] therefore checks the gradient buffer on the device;
 * if a value is not finite the update of that step is left out entirely (parameters and moments untouched) and the next
 * backward scales the loss by a further 1/2 (taken back by one every 64 clean steps) -- what a GradScaler does, without a
 * host read.  This call reads the counters (it synchronises the device): steps skipped so far, current back-off exponent. */
int fu_fp16_guard_state(fu_ctx* ctx, int64_t* skipped_steps, int32_t* backoff_exponent);
/* The same update for a CAPTURED step (hipGraph): the seven float scalars of the kernel -- which depend on the step count --
 * are read from device memory, so one captured launch serves every replay.  fu_adam_scalars forms them on the host exactly
 * as fu_adam_step does (out[7]: 1-beta1, beta2, 1-beta2, sqrt(bias_correction2), eps, -lr/bias_correction1, grad_scale);
 * the caller copies them to scalars_dev before each replay.  All entry points of the hot path enqueue work only (no
 * allocation, no synchronisation, events and the side stream fork and re-join inside fu_backward), so a caller may capture
 * fu_forward + fu_loss_* + fu_backward + fu_adam_step_dev on one stream into a graph and replay it. */
int fu_adam_scalars(double lr, double beta1, double beta2, double eps, int64_t step, double grad_scale, float out[7]);
int fu_adam_step_dev(fu_ctx* ctx, const float* scalars_dev, fu_stream stream);
int fu_zero_grads(fu_ctx* ctx, fu_stream stream);

/* ---- exact data-parallel mode (SyncBN statistics + global N_valid) ----------------
 * The reference trains on one device; with tiles sharded over ranks the default (DDP) semantics use per-rank
 * BatchNorm statistics and a per-rank 1/N_valid.  After fu_set_exact_sync every training fu_forward / fu_loss_ce /
 * fu_backward* sums its statistics partials over the ranks before they are finalised -- BN forward (sum y, sum y^2),
 * BN backward (sum g, sum g*xhat), CE (sum loss, N_valid) -- so that W ranks x B tiles reproduce one device with W*B
 * tiles up to fp32 summation order.  At each of those points the library copies the partials into `exchange`
 * (caller-owned device memory, >= fu_exact_sync_bytes), calls hook(user, n_elems, is_double) -- which must sum
 * exchange[0..n_elems) (double or float) over the ranks in place, ordered on the stream of the running call -- and
 * copies the result back.  Parameter gradients then hold each rank's SHARE of the global gradient: all-reduce them with
 * SUM and call fu_adam_step with grad_scale 1.  world <= 1 or hook == NULL switches the mode off. */
typedef int (*fu_sync_hook)(void* user, int64_t n_elems, int is_double);
int fu_set_exact_sync(fu_ctx* ctx, fu_sync_hook hook, void* user, int world, void* exchange, int64_t exchange_bytes);
int64_t fu_exact_sync_bytes(const fu_ctx* ctx);

/* ---- gradient all-reduce behind the C ABI (one process per GPU, RCCL over xGMI) ---------------
 * For hosts without torch.distributed.  RCCL (librccl.so) is resolved at run time on the first of these calls; nothing
 * else in the library depends on it.  One communicator per context.  The semantics are DESIGN.md section 6's: every rank
 * runs the step on its own tiles; per backward bucket (a contiguous range of the flat gradient buffer,
 * fu_block_param_range) fu_allreduce_begin starts a SUM all-reduce on the context's communication stream, ordered behind
 * the work enqueued on `stream` so far and, in side-stream mode 2, behind the weight-gradient chain enqueued so far (the
 * communication stream waits for both; `stream` is not joined: call fu_backward_join once, after the last block, before the
 * optimizer step), while the remaining backward blocks keep `stream` busy; fu_allreduce_wait makes
 * `stream` wait for every all-reduce begun so far; then fu_adam_step(..., grad_scale = 1 / world, ...).
 *   rank 0: fu_dp_unique_id(&id); ship the 128 bytes to the other ranks by any means (file, socket, MPI, environment);
 *   every rank: fu_dp_init(ctx, &id, rank, world); fu_dp_broadcast_state(ctx, stream) once (rank 0's parameters and
 *   BatchNorm buffers to everyone); ... steps ...; fu_dp_destroy (fu_destroy calls it). */
typedef struct fu_dp_id { char bytes[128]; } fu_dp_id;   /* ncclUniqueId */
int fu_dp_unique_id(fu_dp_id* id);
int fu_dp_init(fu_ctx* ctx, const fu_dp_id* id, int rank, int world);
int fu_dp_broadcast_state(fu_ctx* ctx, fu_stream stream);
int fu_allreduce_begin(fu_ctx* ctx, int64_t flat_offset, int64_t numel, fu_stream stream);
int fu_allreduce_wait(fu_ctx* ctx, fu_stream stream);
int fu_dp_destroy(fu_ctx* ctx);

/* ---- on-GPU tile augmentation (datasets/base_dataset.py:494-555) ------- */
/* Per sample b: hflip (flags[b] & 1), then vflip (& 2), then rotate by angles_deg[b] (& 4) with torchvision's
 * tensor semantics (nearest, expand=False, centre = image centre, image fill 0), applied identically to
 * image fp32 NCHW [B,C,H,W] and target int64 [B,H,W] (fill = target_fill; the reference fills 0).  Out of place.
 * The random draws (which transforms, which angle) stay on the host, as in sample_transforms(). */
enum { FU_AUG_HFLIP = 1, FU_AUG_VFLIP = 2, FU_AUG_ROTATE = 4 };
int fu_augment(const float* image, const int64_t* target, float* image_out, int64_t* target_out, const int32_t* flags,
               const float* angles_deg, int B, int C, int H, int W, int64_t target_fill, fu_stream stream);

/* Tile assembly of a batch that is already in HBM (datasets/base_dataset.py:77-113 `normalize`, :271-325
 * `_add_buffer_to_image`; ef_model.py:28-44 / stacked sensors: channel concatenation).  srcs: HOST array of n_src (<= 8)
 * device pointers, source k fp32 NCHW [B, src_channels[k], H, W]; the valid crop of sample b is the top-left
 * valid_h[b] x valid_w[b] corner (device int32 [B]; NULL = the whole tile).  out: fp32 NCHW [B, sum C, H, W] =
 * (x - mean) / std inside the crop, pad_value outside (the reference pads the image with 0 AFTER normalising).
 * norm_mode 0 = None (mean 0, std 1), 1 = 'local' (per sample and channel mean / population std over the crop, numpy
 * semantics; written to mean_out / std_out fp32 [B, sum C]), 2 = 'global' (global_mean / global_std fp32 [sum C]). */
int fu_assemble_tiles(const float* const* srcs, const int32_t* src_channels, int n_src, int B, int H, int W,
                      const int32_t* valid_h, const int32_t* valid_w, int norm_mode, const float* global_mean,
                      const float* global_std, float pad_value, float* out, float* mean_out, float* std_out,
                      fu_stream stream);

/* Lanczos-4 resampling of a batch of tiles on the device (ABI 5).  Replaces the per-item WHOLE-RASTER resample of the reference's
 * loader (st_water_seg/datasets/floodplanet.py:338-340 -> utils/utils_image.py:11-54, cv2.INTER_LANCZOS4) by a per-tile one:
 * Lanczos is local, so the tile [Y0, Y0 + tile_h) x [X0, X0 + tile_w) of the resampled raster depends only on a window of the
 * source raster.  windows: fp32 [B][C][win_h][win_w] (the host's cut-outs, padded to a common size); iy / wy: int32 / fp32
 * [B][tile_h][8], ix / wx: [B][tile_w][8] -- per output row / column the 8 window-relative source indices (borders replicated)
 * and the normalised Lanczos weights (all-zero rows / columns beyond the raster's edge).  out: fp32 [B][C][tile_h][tile_w] =
 * the crop of the resampled raster, then the sensor scaling that follows the crop in the reference: scale_mode 0 none, 1 S1
 * clip((x + 50) / 100, 0, 1) with NaN -> 0 (:347), 2 S2 clip(x / 2^12, 0, 1) (:406), 3 L8 clip(x, 0, 18607.72) / 18607.72 (:525),
 * 4 PS stored as uint16 x / 2^16 (:467-468).  fp32, taps accumulated in order with separate multiply / add roundings: equal bit
 * for bit to the host restatement floodplanet_code_amd/datasets/resize.py (OpenCV itself is absent: parity unpinned vs cv2). */
int fu_resize_lanczos4_tiles(const float* windows, int B, int C, int win_h, int win_w, const int32_t* iy, const float* wy,
                             const int32_t* ix, const float* wx, int tile_h, int tile_w, int scale_mode, float* out,
                             fu_stream stream);

/* ---- inference stitching (ImageStitcher_v2, utils/utils_image.py:410-494) ------------- */
/* canvas[h0:hE, w0:wE, :] += softmax(logits of sample `sample` of the last fu_forward)[:hE-h0, :wE-w0, :];
 * weight[h0:hE, w0:wE] += 1.  canvas: fp32 [canvas_h, canvas_w, n_classes], weight: fp32 [canvas_h, canvas_w]. */
int fu_stitch_add(fu_ctx* ctx, int sample, float* canvas, float* weight, int canvas_h, int canvas_w, int h0, int w0,
                  int hE, int wE, fu_stream stream);
/* canvas /= (weight + 1e-5) in place (ImageStitcher_v2._combine_images); argmax_out: optional int64 [canvas_h, canvas_w] */
int fu_stitch_finalize(float* canvas, const float* weight, int n_classes, int canvas_h, int canvas_w,
                       int64_t* argmax_out, fu_stream stream);

/* ---- introspection ------------------------------------------------------------------------- */
int64_t fu_workspace_bytes(const fu_ctx* ctx);
/* algorithmic conv FLOPs of one tile: forward, and forward+backward */
int fu_flops_per_tile(const fu_ctx* ctx, double* fwd, double* train);

/* ---- profiling: HIP events around the convolution kernels (bench.py's roofline object) -------- */
enum fu_kernel_class {
  FU_K_CONV3X3 = 0, /* implicit-GEMM 3x3 conv kernels: forward and dgrad launches (16-bit: k_conv3x3_*_rs + k_conv3x3_*_fast) */
  FU_K_WGRAD = 1,   /* weight-gradient kernel (without its slab reduce) */
  FU_K_NUM = 2
};
/* enable == 1: allocate/reset the event pool and time every launch of the classes above on the stream it is
 * launched on; enable == 2: resume without dropping what was recorded (sampling some steps of a run: an event pair
 * around every launch costs ~2 us of serialisation each, 4 % of the bench step when every step is timed);
 * enable == 0: stop.  fu_profile_read synchronises the device and sums what was recorded. */
int fu_profile_enable(fu_ctx* ctx, int enable);
int fu_profile_read(fu_ctx* ctx, int kernel_class, int64_t* launches, double* total_ms, double* total_flops,
                    const char** kernel_name);

/* ---- TEST HOOKS: process-wide switches that let the parity tests run every shape on every kernel variant.  NOT part of
 * the stable ABI (no version bump when they change); never needed by a caller. -------------------------------------- */
/* Testing hook: on != 0 makes every bf16 3x3 convolution (forward / dgrad) run on the general kernel even when the
 * shape is eligible for the aligned-shape fast kernel, so that the parity tests can cover both.  Process-wide. */
void fu_test_force_general_conv(int on);
/* Testing hook: on = 1 runs the bf16 weight gradient of c_in > 64 on the lock-step kernel k_wgrad_bf16<4,8> instead of
 * the ping-pong kernel k_wgrad_bf16_pp, on = 2 keeps the ping-pong kernel but with its general (any-shape) staging where
 * the whole-tile staging would be chosen (same accumulation order: all three are bit-identical).  Process-wide. */
void fu_test_force_lockstep_wgrad(int on);
/* Testing hook: workgroup tile of the aligned-shape bf16 conv kernel at 64 output channels: 0 = heuristic (default),
 * 1 = never the tall 16x32-pixel tile (nor the row-stationary kernel), 2 = the tall tile wherever 64-channel tiles run,
 * 3 = the row-stationary kernel (fu_conv_rs.hip) wherever the shape is eligible.  Process-wide. */
void fu_test_conv_tile_mode(int mode);
/* Testing hook: on != 0 computes every BatchNorm-backward pair of sums with its own reduce pass; by default the 16-bit
 * modes take them from the kernel that produces the gradient where it can (row-stationary dgrad, head backward).
 * Process-wide. */
void fu_test_bnb_separate(int on);
/* Testing hook: on != 0 makes the head backward store its data gradient g = dlogits . W even where (16-bit modes, fused
 * sums) the BatchNorm-backward apply pass of the last conv would recompute it from dlogits and W.  Process-wide. */
void fu_test_head_store_g(int on);
/* Testing hook: on != 0 runs the late-fusion 1x1 convs (weights embedded as the centre tap of a 3x3) through all nine taps
 * instead of the 1-tap instantiation of the fast kernel; the other eight taps multiply exact zeros, so the results are
 * bit-identical.  Process-wide. */
void fu_test_force_full_taps(int on);

/* Testing hook: every BatchNorm-backward pair of sums that a producer kernel emitted (row-stationary dgrad epilogue, head
 * backward) is multiplied by `factor` before it is consumed -- the negative control of the parity tests (a wrong fused sum
 * must make them fail).  1 = off.  Process-wide. */
void fu_test_perturb_bnb_sums(float factor);
/* Testing hook: device pointer and element count (at max_batch) of one saved tensor of plan block `block` (0..4 = inc,
 * down1..4; 5..8 = up1..4 for the plain UNet): which 0 = y of the first conv, 1 = its gradient buffer, 2 / 3 = the same
 * for the second conv, 4 = dL/d(pooled input) (down blocks), 5 = dL/d(upsampled input) (up blocks).  Elements are of
 * the context's precision. */
int fu_test_get_buffer(fu_ctx* ctx, int block, int which, void** ptr, int64_t* elems);

/* ---- single operators (per-op parity tests; NHWC device buffers of the context's precision) -- */
/* element size of the activation type for `precision` */
int fu_elem_size(int precision);
/* fp32 NCHW -> NHWC (channels zero-padded to c_pad) and back (drops padding) */
int fu_op_nchw_to_nhwc(int precision, const float* src, void* dst, int B, int C, int H, int W, int c_pad,
                       fu_stream stream);
int fu_op_nhwc_to_nchw(int precision, const void* src, float* dst, int B, int C, int H, int W, int c_pad,
                       fu_stream stream);
/* y = conv3x3(cat(relu(a0*src0+b0) or src0, src1), w) + bias; w: fp32 OIHW [Cout, C0+C1, 3, 3].
 * stats_sum/stats_sqsum: optional fp32 [Cout] outputs (per-channel sum / sum of squares of y - bias). */
int fu_op_conv3x3_fwd(int precision, const void* src0, int C0, const float* bn_a0, const float* bn_b0,
                      const void* src1, int C1, const float* w_oihw, const float* bias, void* y, int Cout, int B,
                      int H, int W, float* stats_sum, float* stats_sqsum, fu_stream stream);
/* dx = conv3x3_transpose(dy, w): dx0 gets input channels [0,C0), dx1 gets [C0,C0+C1) */
int fu_op_conv3x3_dgrad(int precision, const void* dy, int Cout, const float* w_oihw, void* dx0, int C0, void* dx1,
                        int C1, int B, int H, int W, fu_stream stream);
/* dw (fp32 OIHW) = sum_p x[p+tap] * dy[p], x assembled exactly as in fu_op_conv3x3_fwd */
int fu_op_conv3x3_wgrad(int precision, const void* src0, int C0, const float* bn_a0, const float* bn_b0,
                        const void* src1, int C1, const void* dy, int Cout, float* dw_oihw, int B, int H, int W,
                        fu_stream stream);
int fu_op_maxpool2(int precision, const void* src, const float* bn_a, const float* bn_b, void* dst, int B, int H,
                   int W, int C, fu_stream stream);
/* bilinear x2, align_corners=True, result zero-padded (F.pad) to [outH, outW] */
int fu_op_upsample2(int precision, const void* src, const float* bn_a, const float* bn_b, void* dst, int B, int H,
                    int W, int C, int outH, int outW, fu_stream stream);

/* The operators below exist for the code that only runs in the benched 16-bit dispatch (they are test hooks like the
 * fu_test_* switches: no ABI promise).
 * dgrad as fu_op_conv3x3_dgrad with ONE destination that is the output gradient of a BatchNorm+ReLU whose raw input is y
 * [B,H,W,C0] and whose coefficients are bn_a / bn_b / mean / invstd: also returns that BatchNorm's backward sums
 * sum_gm[c] = sum_p g*m and sum_gmx[c] = sum_p g*m*xhat (m = [a*y+b > 0], xhat = (y-mean)*invstd), taken from the conv
 * kernel's epilogue (BnbFuse).  FU_ERR_UNSUPPORTED when the kernel that ran does not emit them (select the
 * row-stationary kernel with fu_test_conv_tile_mode(3)). */
int fu_op_conv3x3_dgrad_bnsums(int precision, const void* dy, int Cout, const float* w_oihw, void* dx, int C0,
                               const void* y, const float* bn_a, const float* bn_b, const float* mean,
                               const float* invstd, float* sum_gm, float* sum_gmx, int B, int H, int W,
                               fu_stream stream);
/* head backward (OutConv 1x1): g = dlogits * W (element type of `precision`), dw [ncls][C], db [ncls]; with mean / invstd /
 * sum_gm / sum_gmx non-null also the BatchNorm-backward sums of g as above (16-bit precisions). */
int fu_op_head_bwd(int precision, const float* dlogits_nhwc, const void* y, const float* bn_a, const float* bn_b,
                   const float* w, int C, int ncls, int64_t npix, void* g, float* dw, float* db, const float* mean,
                   const float* invstd, float* sum_gm, float* sum_gmx, fu_stream stream);
/* BatchNorm + ReLU backward in place: g holds dL/d relu(bn(y)) on entry and dL/dy on return; g_pool (optional,
 * [B,H/2,W/2,C]) = dL/d maxpool2(relu(bn(y))), whose backward is folded in (first-maximum tie rule). */
int fu_op_bn_bwd(int precision, void* g, const void* y, int C, int B, int H, int W, const float* bn_a,
                 const float* bn_b, const float* mean, const float* invstd, const void* g_pool, float* dgamma,
                 float* dbeta, fu_stream stream);

#ifdef __cplusplus
}
#endif
#endif /* FLOODUNET_H_ */
