/*
 * isdf_hip.h -- C ABI of the MI355X-native iSDF training hot path.
 *
 * The reference (facebookresearch/iSDF) is pure Python/PyTorch and has no FFI;
 * its seam for this path is the Python method `Trainer.step`
 * (isdf/modules/trainer.py:951-1016) and the two methods it calls,
 * `Trainer.sample_points` (:683-766) and `Trainer.sdf_eval_and_loss` (:768-868).
 * This header is what a ctypes binding inside those methods binds to; the
 * reference-side binding (graft(trainer)) is shown in INTEGRATION.md and lives in
 * isdf_amd/hot_path.py.
 *
 * Conventions
 *   - extern "C", plain pointers and sizes, no torch types.  All pointers are
 *     DEVICE pointers unless a parameter name ends in `_host`.
 *   - The caller owns every buffer (torch allocates them); the library keeps
 *     no state between calls and never allocates, frees or synchronises.
 *   - All work is enqueued on `stream` (a hipStream_t passed as void*).
 *   - Return value: 0 on success, a negative ISDF_E* code otherwise; never
 *     throws.  isdf_error_string() names the code.
 *   - Points are stored ray-major: point n = ray * S + s, S = n_strat + n_surf.
 */
#ifndef ISDF_HIP_H
#define ISDF_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ISDF_ABI_VERSION 7

enum {
  ISDF_OK = 0,
  ISDF_EINVAL = -1,       /* bad argument / null pointer / size mismatch      */
  ISDF_EUNSUPPORTED = -2, /* configuration the kernels are not built for      */
  ISDF_EWORKSPACE = -3,   /* caller workspace too small                       */
  ISDF_EHIP = -4,         /* a HIP runtime call failed (see hipGetLastError)  */
  ISDF_ECOLLECTIVE = -5   /* the caller's collective library refused the call */
};

/* ---- network description ------------------------------------------------
 * Mirrors SDFMap(PostionalEncoding) (isdf/modules/fc_map.py:63-111,
 * isdf/modules/embedding.py:24-111).  Parameters live in ONE flat fp32 buffer
 * in `named_parameters()` order:
 *   in_layer.0.{weight[Hd,E],bias[Hd]}, mid1.i.0.{weight[Hd,Hd],bias}, i<B,
 *   cat_layer.0.{weight[Hd,Hd+E],bias}, mid2.i.0.{...}, out_alpha.{weight[1,Hd],bias[1]}
 * (the host mirror re-points the nn.Module parameters at views of it).       */
typedef struct isdf_net_cfg {
  int32_t hidden;        /* Hd: hidden_feature_size (replicaCAD.json:58): any value <= 512 (widths other than 256 / 512 run zero-padded
                            on the 256- / 512-wide tile kernels; parameters, moments and gradients keep the reference's shapes) */
  int32_t blocks;        /* B : hidden_layers_block (replicaCAD.json:57)       */
  int32_t n_freqs;       /* n_embed_funcs + 1 (embedding.py:36); E = 42*n+3    */
  int32_t has_transform; /* 0: PE transform is None (live modes)    */
  float scale_input;     /* embedding.scale_input                              */
  float scale_output;    /* model.scale_output (fc_map.py:109)                 */
  float bounds_T[12];    /* rows of inv_bounds_transform[:3,:4] (row-major)    */
  int32_t fwd_operand;   /* MFMA operands of the forward and first-backward
                            GEMMs (the second-order passes: bwd_operand):
                            0 bf16; 1 fp16; 2 "fp16x2" = fp16 with a compensated
                            forward -- layers >= cat_layer also multiply the
                            fp16 residual of their weights, the layers past it
                            the fp16 residual of their input too -- which is
                            what meets the reference's fp32 sdf to 1e-3
                            (fc_map.py:94-111); 3 "fp16x2_full" =
                            the same compensation in EVERY forward layer, the
                            embedding included (exact-forward instrument: sdf
                            ~1e-6 and d sdf/dx < 1e-3 of the reference;
                            hidden 256 with a padded embedding of 256 only,
                            other shapes ISDF_EUNSUPPORTED).  isdf_shadow_bytes
                            grows by one forward set for modes 2 and 3.        */
  int32_t bwd_operand;   /* ABI 5: MFMA operands of the SECOND-order passes (adjoint of the input-gradient sweep, reverse sweep
                            with the injected term, the dW contraction) and storage type of the tensors spilled between them:
                            0 bf16 (range-safe for any loss-adjoint magnitude; rounds 1-3), 1 fp16 (needs fwd_operand >= 1;
                            11 instead of 8 significand bits in every weight-gradient operand: total_loss.backward(),
                            trainer.py:981, to ~1e-3 instead of ~4e-3; the adjoints of the shipped configs sit well inside
                            fp16's range)                                                                    */
  int32_t spill_operand; /* ABI 6: storage of two of the four tensor families the training step parks in HBM between its sweeps and for
                            the dW contraction -- P = d sdf / d z (below the top layer) and GB = the adjoint entering each layer in
                            the upward sweep, i.e. the operands of the SECOND-order product of total_loss.backward() (the eikonal /
                            normal terms' share of the gradient, trainer.py:814-830,981):
                            0 auto: e4m3 when n_freqs <= 6, hidden <= 256 and bwd_operand is fp16 (replicaCAD.json / scanNet.json), else 16-bit;
                            1 the 16-bit bwd_operand type (rounds 1-5); 2 OCP e4m3 (hidden <= 256 only), one byte per element: P / 2^-10, GB / s_G with a
                            per-POINT scale s_G (GB is linear in the point's loss adjoint); 3 e4m3 for GB only.  e4m3 halves 11 of the 23 tensors a step
                            stores and 21 of the 50 it reads back; at six octaves it moves the worst weight gradient by 2e-4 of the
                            reference's, with nine or more octaves the second-order term carries most of the first layers' gradient
                            and e4m3 costs ~1e-2 there -- which is why `auto` keeps those nets on 16 bits.            */
} isdf_net_cfg;

int isdf_abi_version(void);
const char* isdf_error_string(int code);

/* ISDF_OK if the tile kernels are instantiated for this network shape, ISDF_EUNSUPPORTED otherwise (call it when
 * the network is built -- trainer.py:419-439 -- rather than at the first step) */
int isdf_check_net(const isdf_net_cfg* net);

/* number of fp32 parameters (460033 for the default net) */
int64_t isdf_param_count(const isdf_net_cfg* net);
/* bytes of the packed 16-bit MFMA-operand copies of the weights ("shadow") */
int64_t isdf_shadow_bytes(const isdf_net_cfg* net);
/* bytes of scratch isdf_train_step / isdf_sdf_eval need for up to max_points */
int64_t isdf_workspace_bytes(const isdf_net_cfg* net, int64_t max_points, int32_t train);
/* number of floats in the flat reduction buffer written by isdf_train_step:
 *   [ grad(n_params) | loss_sums(8) | block_loss(F*64) | block_cnt(F*64) ]    */
int64_t isdf_reduce_floats(const isdf_net_cfg* net, int32_t n_frames);
/* first float of the reduction buffer that is complete when isdf_step_out.split_event is recorded: the message's SUFFIX
 * [isdf_reduce_split_floats, isdf_reduce_floats + extra_floats) = weight gradients of the layers from the cat layer up, the out
 * layer, loss sums, bins and the caller's tail; the PREFIX [0, isdf_reduce_split_floats) = layers below the cat layer
 * follows with the call's last launch.  (New: the reference has no collective, README.md:102; SURVEY 8e "overlap the
 * all-reduce of early-finished layers' dW".)                                                                       */
int64_t isdf_reduce_split_floats(const isdf_net_cfg* net);

/* Rebuild the packed operand copies from the fp32 parameters (after loading a
 * checkpoint; isdf_adamw does it itself after every update).                  */
int isdf_pack_weights(const isdf_net_cfg* net, const float* params, void* shadow, void* stream);

/* ---- K1: ray / point sampler ---------------------------------------------
 * Replaces sample.sample_pixels + get_batch_data + sample_along_rays
 * (isdf/modules/sample.py:11-178) and transform.origin_dirs_W
 * (isdf/geometry/transform.py:36-41).  dirs_C is computed from the intrinsics
 * (transform.py:13-33) instead of gathered from a [H,W,3] table.             */
typedef struct isdf_sample_args {
  /* keyframe store (data_util.FrameData, isdf/datasets/data_util.py:11-81)   */
  const float* depth_batch;   /* [K,H,W]                                       */
  const float* normal_batch;  /* [K,H,W,3] or NULL (do_normal false)           */
  const float* T_WC_batch;    /* [K,4,4]                                       */
  const int32_t* frame_idx;   /* [F] window -> keyframe index (trainer.py:965) */
  const int32_t* normal_idx;  /* [F] keyframe index used for normals; the
                                 reference passes the un-windowed normal_batch
                                 (trainer.py:956,969): pass 0..F-1
                                 to reproduce, frame_idx to fix                */
  int32_t n_frames;           /* F                                             */
  int32_t n_rays;             /* rays per frame (sample.n_rays)                */
  int32_t H, W;
  float fx, fy, cx, cy;
  int32_t n_strat, n_surf;    /* sample.n_strat_samples / n_surf_samples       */
  float min_depth;            /* sample.depth_range[0]                         */
  float dist_behind_surf;
  /* random inputs.  rng_mode 0 ("injected", parity): the caller supplies the
   * draws the reference would make, in the reference's own shapes:
   *   draw_h/draw_w [F*n_rays] int64  (torch.randint, sample.py:15-16)
   *   draw_u [R,n_strat] (torch.rand, sample.py:123), draw_n [R,n_surf-1]
   *   (torch.normal(0,0.1) on the CPU generator, sample.py:160-162), both
   *   indexed by COMPACTED ray.  rng_mode 1: in-kernel Philox4x32-10 keyed by
   *   (seed, offset): counter (drawn ray, 0) for the pixel, one call per four
   *   consecutive-by-64 output points for the along-ray draws (surface
   *   offsets: Box-Muller on the two 16-bit halves of a word); deterministic
   *   for a given (seed, offset, configuration), not stream-compatible with
   *   torch, and the mapping may change between ABI versions.                 */
  int32_t rng_mode;
  const int64_t* draw_h;
  const int64_t* draw_w;
  const float* draw_u;
  const float* draw_n;
  uint64_t seed, offset;
  /* ABI 5 -- the window INLINE: with n_inline == n_frames (1..ISDF_MAX_INLINE_FRAMES) the kernel takes the window's keyframe
   * indices from these kernel arguments and ignores frame_idx / normal_idx (which may then be NULL): no device copy of a window
   * that select_keyframes (trainer.py:652-674) re-draws on the host every step, and no dependent index load in front of the
   * depth gather.  0 = use the device arrays.                                                                            */
  int32_t n_inline;
  int32_t frame_idx_inline[8];
  int32_t normal_idx_inline[8];
  int32_t reserved_inline;
} isdf_sample_args;
#define ISDF_MAX_INLINE_FRAMES 8

typedef struct isdf_sample_out {
  int32_t* n_valid;      /* [1]  R = rays kept (depth != 0, normal not NaN)    */
  int64_t* indices_b;    /* [F*n_rays] first R entries valid (ordered)         */
  int64_t* indices_h;
  int64_t* indices_w;
  float* depth_sample;   /* [F*n_rays]                                         */
  float* dirs_C_sample;  /* [F*n_rays,3]                                       */
  float* norm_sample;    /* [F*n_rays,3] or NULL                               */
  float* T_WC_sample;    /* [F*n_rays,4,4] or NULL (reference materialises it) */
  float* dirs_W_sample;  /* [F*n_rays,3] world-frame ray direction             */
  float* z_vals;         /* [F*n_rays,S]                                       */
  float* pc;             /* [F*n_rays,S,3]                                     */
} isdf_sample_out;

/* ONE launch: pixel draw (or injected draws), gather of depth + normal, validity, ORDER-PRESERVING
 * compaction across workgroups (decoupled look-back), then per compacted ray the stratified + surface z
 * values and world points.  scan_ws: device scratch of isdf_sample_scan_bytes(F*n_rays) bytes that must be
 * ZERO when first used and is otherwise private to this entry point (it re-arms itself at the end of every
 * launch; launches sharing one scan_ws must be stream-ordered).                                            */
int64_t isdf_sample_scan_bytes(int64_t max_rays);
int isdf_sample_rays(const isdf_sample_args* a, const isdf_sample_out* o, void* scan_ws,
                     int64_t scan_ws_bytes, void* stream);

/* ---- fused PE + MLP (+ input gradient) inference --------------------------
 * Replaces SDFMap.forward and fc_map.gradient (fc_map.py:94-111,12-22) for
 * arbitrary points.  noise: pre-scaled additive term on the raw output
 * (fc_map.py:106-108) or NULL.  sdf_grad may be NULL.                          */
int isdf_sdf_eval(const isdf_net_cfg* net, const float* params, const void* shadow,
                  const float* pts, int64_t n_points, const float* noise,
                  float* sdf, float* sdf_grad, void* workspace, int64_t workspace_bytes,
                  void* stream);

/* ---- training step (everything between sampling and the optimiser) --------
 * Replaces Trainer.sdf_eval_and_loss + total_loss.backward()
 * (trainer.py:768-868,981; loss.py:13-240).                                   */
typedef struct isdf_loss_cfg {
  int32_t bounds_method;  /* 0 "ray" (loss.py:13-22), 1 "pc" (loss.py:56-89)   */
  int32_t loss_type;      /* 0 L1, 1 L2 (loss.py:138-143)                      */
  float trunc_weight, trunc_distance, eik_weight, eik_apply_dist, grad_weight;
  int32_t orien_loss;
} isdf_loss_cfg;

typedef struct isdf_step_args {
  const int32_t* n_valid;     /* [1] device: R (from the sampler)              */
  int32_t max_rays;           /* capacity of the per-ray arrays (F*n_rays)     */
  int32_t S;                  /* samples per ray                               */
  int32_t n_frames, H, W;     /* for the 8x8 block-loss bins (loss.py:208-240);
                                 H and W must be multiples of 8 (ISDF_EINVAL otherwise, as
                                 the reference's .view raises)                    */
  const float* pc;            /* [max_rays,S,3]                                */
  const float* z_vals;        /* [max_rays,S]                                  */
  const float* depth_sample;  /* [max_rays]                                    */
  const float* dirs_C_sample; /* [max_rays,3]                                  */
  const float* dirs_W_sample; /* [max_rays,3]                                  */
  const float* norm_sample;   /* [max_rays,3] (may be NULL when grad_weight=0) */
  const int64_t* indices_b;   /* [max_rays]                                    */
  const int64_t* indices_h;
  const int64_t* indices_w;
  const float* noise;         /* [max_rays,S] pre-scaled, or NULL              */
  float noise_std;            /* used when noise == NULL and noise_std != 0:   */
  uint32_t reserved0;         /* in-kernel Philox N(0,1) * noise_std keyed by  */
  uint64_t noise_seed;        /* (noise_seed, noise_offset, point)             */
  uint64_t noise_offset;      /* (fc_map.py:106-108)                            */
  const float* pc_bounds;     /* [max_rays,S]   bounds_method "pc" only        */
  const float* pc_grad_vec;   /* [max_rays,S,3] bounds_method "pc" only        */
  /* caller-owned tail of the reduction message (data parallel): reduce_buf may be
   * followed by `extra_floats` floats that ride in the same all-reduce; the step
   * writes extra_value into slot extra_slot and 0 into the others (one slot per
   * rank: after the SUM all-reduce every rank holds every rank's value -- the
   * host mirror carries the ranks' step times this way, no second collective) */
  int32_t extra_floats;
  int32_t extra_slot;
  float extra_value;
  int32_t reserved1;
} isdf_step_args;

typedef struct isdf_step_out {
  float* reduce_buf;    /* [isdf_reduce_floats]: SUMS (not means) so that ranks
                           can be all-reduced and scaled once      */
  float* sdf;           /* optional [max_rays,S]  debug / parity outputs       */
  float* sdf_grad;      /* optional [max_rays,S,3]                             */
  float* tot_loss_mat;  /* optional [max_rays,S]                               */
  void** prof_events;   /* optional HOST array of 4 hipEvent_t recorded on `stream`:
                           [0] before the chain kernel, [1] after it, [2] after
                           the dW kernel, [3] after the reductions (bench.py)    */
  float* host_mailbox;  /* optional PINNED HOST memory (device-mapped, >= 8 floats): the
                           step's last launch also stores loss_sums[8] there, so the
                           `losses` of Trainer.step (trainer.py:1016; loss.py:187-200)
                           need no device->host copy command -- they are valid after the
                           caller's closing stream synchronisation (metrics.py:27-30)   */
  void* split_event;    /* optional hipEvent_t, isdf_train_step only (data parallel): the closing reduction runs as TWO
                           launches and this event is recorded on `stream` between them -- the suffix of reduce_buf
                           (isdf_reduce_split_floats) is final at the event, so its all-reduce can run on a second stream
                           beside the second launch.  Same arithmetic, same values as the one-launch form.            */
} isdf_step_out;

/* layout of loss_sums inside reduce_buf (after the n_params gradient floats) */
enum { ISDF_LS_SDF = 0, ISDF_LS_GRAD = 1, ISDF_LS_EIK = 2, ISDF_LS_TOTAL = 3, ISDF_LS_COUNT = 4 };

int isdf_train_step(const isdf_net_cfg* net, const isdf_loss_cfg* loss, const float* params,
                    const void* shadow, const isdf_step_args* a, const isdf_step_out* o,
                    void* workspace, int64_t workspace_bytes, void* stream);

/* Single-GPU form of the same step with the optimiser fused in: what
 * `total_loss.backward(); self.optimiser.step(); self.optimiser.zero_grad()` do in
 * Trainer.step (isdf/modules/trainer.py:981-983) as ONE call.  After the dW kernel a single
 * launch sums the gradient, applies AdamW to params / exp_avg / exp_avg_sq in place,
 * refreshes the packed operand copies in `shadow`, and finalises the loss sums and block
 * bins.  reduce_buf is written exactly as by isdf_train_step (the gradient is the SUM over
 * points; AdamW divides by the point count).  Bit-identical to isdf_train_step followed by
 * isdf_adamw(count_ptr = &loss_sums[ISDF_LS_COUNT]).  Data-parallel runs use the two-call
 * form because the gradient all-reduce sits between the reduction and the update.          */
typedef struct isdf_optim_args {
  float* params;        /* [n_params] fp32 master weights, updated in place               */
  float* exp_avg;       /* [n_params] AdamW first moment                                   */
  float* exp_avg_sq;    /* [n_params] AdamW second moment                                  */
  void* shadow;         /* packed operand copies (isdf_shadow_bytes), refreshed in place   */
  float lr, beta1, beta2, eps, weight_decay;
  float grad_scale;     /* multiplies the mean gradient (1.0 for the reference's loss)     */
  int32_t step;         /* 1-based optimiser step (bias correction)                        */
  int32_t reserved;
  /* optional: loss.frame_avg (loss.py:208-240) of this step written by the same launch, i.e. what
   * isdf_frame_avg would return: loss_approx [F,8,8] and frame_avg[frame_avg_index ?
   * frame_avg_index[f] : f] -- pass the keyframe store's frame_avg_losses and the window's keyframe
   * indices to get `self.frames.frame_avg_losses[idxs] = frame_avg_loss` (trainer.py:979) for free.
   * Both or neither of loss_approx / frame_avg.                                                    */
  float* loss_approx;
  float* frame_avg;
  const int32_t* frame_avg_index;
  /* ABI 5: the same index list inline (frame_avg_inline_n == n_frames, 1..ISDF_MAX_INLINE_FRAMES; frame_avg_index is then
   * ignored and may be NULL), for the host-drawn window of every step (see isdf_sample_args.n_inline)                       */
  int32_t frame_avg_inline_n;
  int32_t frame_avg_index_inline[8];
  int32_t reserved_inline;
} isdf_optim_args;

int isdf_train_step_adamw(const isdf_net_cfg* net, const isdf_loss_cfg* loss, const isdf_step_args* a,
                          const isdf_step_out* o, const isdf_optim_args* opt, void* workspace,
                          int64_t workspace_bytes, void* stream);

/* Second half of the DATA-PARALLEL step: after isdf_train_step has written this rank's SUMS to reduce_buf
 * and ONE all-reduce(sum) over reduce_buf has made them global, this single launch does what
 * `self.optimiser.step()` and `self.frames.frame_avg_losses[idxs] = frame_avg_loss` do upstream (trainer.py:979-982):
 * AdamW on gradient = grad_sum * grad_scale / reduced count, refresh of the packed operand copies, and (when
 * opt->loss_approx / opt->frame_avg are given) loss.frame_avg (loss.py:208-240) from the reduced bins.
 * isdf_train_step + all-reduce + isdf_train_step_finish == isdf_train_step_adamw on the union batch
 * (tests/test_dp_gpu.py).                                                                                          */
int isdf_train_step_finish(const isdf_net_cfg* net, const isdf_optim_args* opt, const float* reduce_buf,
                           int32_t n_frames, int32_t extra_floats, float* host_mailbox, void* stream);

/* THE collective of the data-parallel step, enqueued on the step's OWN stream between isdf_train_step and
 * isdf_train_step_finish: in-place sum of buf[0..count) over the communicator's ranks.  The library does not link a
 * collective library: the caller passes the address of RCCL's `ncclAllReduce` (from the RCCL build its communicator was
 * created with -- torch ships its own) and the communicator (ncclComm_t).  Stream order replaces the two cross-stream
 * event hand-offs of a framework-issued collective (torch.distributed.all_reduce runs on a side stream: +36 us per step
 * at world size 1 in round 5, profiles/r05_bench_forced_dp_world1.json).  The caller must not have another collective of
 * the same communicator in flight on a different stream.  Returns ISDF_ECOLLECTIVE (with the library's code in
 * isdf_error_string's text) if RCCL refuses the call.  Reference: none (the reference is single-process).  */
typedef int (*isdf_nccl_allreduce_fn)(const void* sendbuf, void* recvbuf, size_t count, int datatype, int op,
                                      void* comm, void* stream);
int isdf_allreduce_sum_f32(isdf_nccl_allreduce_fn nccl_all_reduce, void* comm, float* buf, int64_t count, void* stream);
/* extra_floats / host_mailbox (optional, pinned host memory of 8 + extra_floats floats): the same launch stores the
 * REDUCED loss_sums[8] followed by the message's caller-owned tail there (isdf_step_args.extra_floats).
 * extra_floats > 0 without a host_mailbox is ISDF_EINVAL (the tail would be dropped silently).                    */

/* nearest-surface-point bounds (bounds_method "pc", loss.py:56-89): for every sample point the distance to
 * the nearest SURFACE sample (sign from z vs depth) and the unit vector from it.  surf_pts == NULL: the
 * surface set is this batch's own pc[:, 0, :] (what the single-process reference uses, loss.py:58-61).
 * Data parallel: surf_pts [n_surf,3] is the all-gathered surface set of all ranks' rays, so each
 * rank sees the surface the single-process run with the global batch would see; slots of dropped rays may
 * hold any far-away sentinel (e.g. 1e18).                                                                    */
int isdf_bounds_pc(const int32_t* n_valid, int32_t max_rays, int32_t S, const float* pc,
                   const float* z_vals, const float* depth_sample, const float* surf_pts,
                   int64_t n_surf, float* bounds, float* grad_vec, void* stream);

/* per-frame 8x8 block-loss averages from the (all-reduced) bins (loss.py:208-240): loss_approx [F,8,8] and
 * frame_avg_loss[frame_avg_index ? frame_avg_index[f] : f] -- pass the keyframe store's frame_avg_losses and
 * the window's keyframe ids to get `self.frames.frame_avg_losses[idxs] = frame_avg_loss` (trainer.py:979)
 * without the separate index_put; frame_avg_index NULL: dense [F] output.                                  */
int isdf_frame_avg(const float* reduce_buf, int64_t n_params, int32_t n_frames,
                   float* loss_approx, float* frame_avg_loss, const int32_t* frame_avg_index,
                   void* stream);

/* ---- per-frame ingest and keyframe test ("next" tier) ------------
 * isdf_estimate_normals: transform.pointcloud_from_depth_torch +
 * estimate_pointcloud_normals (transform.py:169-196,215-270) as run by
 * Trainer.get_data (trainer.py:553-557): depth [H,W] (0 = invalid) -> camera-frame
 * unit normals [H,W,3] (NaN where the 8-neighbour stencil has no valid pair).   */
int isdf_estimate_normals(const float* depth, int32_t H, int32_t W, float fx, float fy, float cx,
                          float cy, float* normals, void* stream);

/* isdf_render_depth: the keyframe test of Trainer.is_keyframe (trainer.py:597-609):
 * per ray sort the S samples by z, render.sdf_render_depth (render.py:12-35), and
 * count rays with |view - depth| / depth < kf_dist_th.  n_valid (device) or, when
 * NULL, n_rays_host rays; depth_sample / below_count may be NULL (render only).  */
int isdf_render_depth(const int32_t* n_valid, int64_t n_rays_host, int64_t max_rays, int32_t S,
                      const float* z_vals, const float* sdf, const float* depth_sample,
                      float kf_dist_th, float* view_depth, int32_t* below_count, void* stream);

/* ---- fused flat AdamW + operand-copy refresh -------------------------------
 * Replaces torch.optim.AdamW.step (trainer.py:435-439,982): decoupled weight
 * decay on all parameters, bias-corrected moments.  grad_scale multiplies the
 * summed gradient; count_ptr (device, optional) divides it further by
 * *count_ptr (the all-reduced number of loss elements).                        */
int isdf_adamw(const isdf_net_cfg* net, float* params, float* exp_avg, float* exp_avg_sq,
               const float* grad_sum, const float* count_ptr, float grad_scale,
               float lr, float beta1, float beta2, float eps, float weight_decay,
               int32_t step, void* shadow, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* ISDF_HIP_H */
