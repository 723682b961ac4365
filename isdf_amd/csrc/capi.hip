// extern "C" entry points declared in include/isdf_hip.h: argument validation,
// layout computation and kernel launches.  No allocation, no synchronisation,
// no global state.
#include <cstdio>
#include "isdf_common.h"
#include "chain_params.h"

using namespace isdf;

namespace isdf {
int launch_chain(const ChainParams& p, int mode, int64_t nTiles, hipStream_t st);
int launch_dw(const DwParams& p, hipStream_t st);
int launch_sample_rays(const isdf_sample_args& a, const isdf_sample_out& o, void* scan_ws, hipStream_t st);
int64_t sample_scan_bytes(int64_t max_rays);
int launch_adamw(float* p, float* m, float* v, const float* g, const float* cnt, float gs, float lr, float b1,
                 float b2, float eps, float wd, int step, int64_t n, hipStream_t st);
int launch_pack(const NetLayout& L, const float* params, uint16_t* shadow, hipStream_t st);
int launch_adamw_pack(const NetLayout& L, float* params, float* m, float* v, uint16_t* shadow, const float* grad,
                      const float* count_ptr, float grad_scale, float lr, float b1, float b2, float eps, float wd,
                      int step, hipStream_t st, int n_frames = 0, const float* bl = nullptr, const float* bc = nullptr,
                      float* la = nullptr, float* fa = nullptr, const int32_t* fa_index = nullptr,
                      const float* loss_sums = nullptr, const float* extra = nullptr, int n_extra = 0, float* mailbox = nullptr,
                      int fa_inline_n = 0, const int32_t* fa_inline = nullptr);
int launch_step_tail(int phase, const NetLayout& L, const float* dwPart, const float* vecPart, int vecStride, float* grad,
                     float* params, float* m, float* v, uint16_t* shadow, float grad_scale, float lr, float b1, float b2,
                     float eps, float wd, int step, const float* wg_loss, int64_t maxTiles, const int32_t* n_valid, int S,
                     const float* tot_ws, const int64_t* ib, const int64_t* ih, const int64_t* iw, int F, int H, int W,
                     float* loss_sums, float* bl, float* bc, float* la_out, float* fa_out, const int32_t* fa_index,
                     hipStream_t st, float* mailbox, float* extra, int n_extra, int extra_slot, float extra_value, int part = 0,
                     int fa_inline_n = 0, const int32_t* fa_inline = nullptr);
int launch_frame_avg(const float* bl, const float* bc, int F, float* la, float* fa, const int32_t* fa_index,
                     hipStream_t st);
int launch_bounds_pc(const int32_t* n_valid, int max_rays, int S, const float* pc, const float* z, const float* depth,
                     const float* surf, int64_t n_surf, float* bounds, float* gv, hipStream_t st);
int launch_normals(const float* depth, int H, int W, float fx, float fy, float cx, float cy, float* normals,
                   hipStream_t st);
int launch_render_depth(const int32_t* n_valid, int64_t n_host, int64_t max_rays, int S, const float* z,
                        const float* sdf, const float* depth_sample, float th, float* view, int32_t* below,
                        hipStream_t st);
}  // namespace isdf

namespace isdf { thread_local int g_isdf_last_hip_error = 0; }

extern "C" {

int isdf_abi_version(void) { return ISDF_ABI_VERSION; }

static thread_local int g_isdf_last_collective_error = 0;

const char* isdf_error_string(int code) {
  switch (code) {
    case ISDF_OK: return "ok";
    case ISDF_EINVAL: return "invalid argument";
    case ISDF_EUNSUPPORTED: return "unsupported configuration (the tile kernels take hidden_feature_size <= 512 (zero-padded to 256 / 512), n_freqs = n_embed_funcs+1 in 1..12, any hidden_layers_block up to 7; bounds_method ray|pc)";
    case ISDF_EWORKSPACE: return "workspace too small";
    case ISDF_EHIP: {
      static thread_local char buf[160];
      snprintf(buf, sizeof(buf), "HIP runtime error: %s", g_isdf_last_hip_error ? hipGetErrorString((hipError_t)g_isdf_last_hip_error) : "(no launch status recorded)");
      return buf;
    }
    case ISDF_ECOLLECTIVE: {
      static thread_local char buf[96];
      snprintf(buf, sizeof(buf), "the collective library refused the all-reduce (ncclResult_t %d)", g_isdf_last_collective_error);
      return buf;
    }
  }
  return "unknown error";
}

int isdf_check_net(const isdf_net_cfg* net) {
  NetLayout l; int rc = make_layout(net, &l);
  if (rc) return rc;
  return layout_supported(l) ? ISDF_OK : ISDF_EUNSUPPORTED;
}

int64_t isdf_param_count(const isdf_net_cfg* net) {
  NetLayout l; int rc = make_layout(net, &l);
  return rc ? rc : l.n_params;
}

int64_t isdf_shadow_bytes(const isdf_net_cfg* net) {
  NetLayout l; int rc = make_layout(net, &l);
  return rc ? rc : l.shadowElems * 2;
}

int64_t isdf_workspace_bytes(const isdf_net_cfg* net, int64_t max_points, int32_t train) {
  NetLayout l; int rc = make_layout(net, &l);
  if (rc) return rc;
  if (max_points < 0) return ISDF_EINVAL;
  WorkspaceLayout w; make_workspace(l, max_points, max_points, train != 0, &w);
  return w.totalBytes;
}

int64_t isdf_reduce_floats(const isdf_net_cfg* net, int32_t n_frames) {
  NetLayout l; int rc = make_layout(net, &l);
  if (rc) return rc;
  return l.n_params + 8 + 2 * (int64_t)n_frames * 64;
}

int64_t isdf_reduce_split_floats(const isdf_net_cfg* net) {
  NetLayout l; int rc = make_layout(net, &l);
  if (rc) return rc;
  return l.offW[l.cat];    // cat_layer.0.weight: everything from here on is final at isdf_step_out.split_event
}

int isdf_pack_weights(const isdf_net_cfg* net, const float* params, void* shadow, void* stream) {
  isdf_clear_stale_hip_error();
  NetLayout l; int rc = make_layout(net, &l);
  if (rc) return rc;
  if (!params || !shadow) return ISDF_EINVAL;
  return launch_pack(l, params, (uint16_t*)shadow, (hipStream_t)stream);
}

int64_t isdf_sample_scan_bytes(int64_t max_rays) { return max_rays < 1 ? ISDF_EINVAL : sample_scan_bytes(max_rays); }

int isdf_sample_rays(const isdf_sample_args* a, const isdf_sample_out* o, void* scan_ws, int64_t scan_ws_bytes,
                     void* stream) {
  isdf_clear_stale_hip_error();
  if (!a || !o || !a->depth_batch || !a->T_WC_batch || !o->n_valid || !o->indices_b ||
      !o->indices_h || !o->indices_w || !o->depth_sample || !o->dirs_C_sample || !o->dirs_W_sample || !o->z_vals ||
      !o->pc)
    return ISDF_EINVAL;
  if (a->n_inline != 0 && (a->n_inline != a->n_frames || a->n_inline > ISDF_MAX_INLINE_FRAMES)) return ISDF_EINVAL;
  for (int f = 0; f < a->n_inline; ++f)   // inline window indices are host values: reject what would gather in front of the keyframe buffers
    if (a->frame_idx_inline[f] < 0 || (a->normal_batch && a->normal_idx_inline[f] < 0)) return ISDF_EINVAL;
  if (a->n_inline == 0 && (!a->frame_idx || (a->normal_batch && !a->normal_idx))) return ISDF_EINVAL;
  if (a->n_frames < 1 || a->n_rays < 1 || a->H < 1 || a->W < 1 || a->n_strat < 1 || a->n_surf < 0) return ISDF_EINVAL;
  if ((int64_t)a->n_frames * a->n_rays > 0x7fffffff / 64) return ISDF_EINVAL;
  if (a->rng_mode == 0 && (!a->draw_h || !a->draw_w || !a->draw_u || (a->n_surf > 1 && !a->draw_n))) return ISDF_EINVAL;
  if (!scan_ws || scan_ws_bytes < sample_scan_bytes((int64_t)a->n_frames * a->n_rays)) return ISDF_EWORKSPACE;
  return launch_sample_rays(*a, *o, scan_ws, (hipStream_t)stream);
}

int isdf_sdf_eval(const isdf_net_cfg* net, const float* params, const void* shadow, const float* pts,
                  int64_t n_points, const float* noise, float* sdf, float* sdf_grad, void* workspace,
                  int64_t workspace_bytes, void* stream) {
  isdf_clear_stale_hip_error();
  NetLayout l; int rc = make_layout(net, &l);
  if (rc) return rc;
  if (!layout_supported(l)) return ISDF_EUNSUPPORTED;
  if (!params || !shadow || !pts || !sdf || n_points < 0) return ISDF_EINVAL;
  if (n_points == 0) return ISDF_OK;
  ChainParams p = {};
  p.lay = l; p.params = params; p.shadow = (const uint16_t*)shadow; p.pts = pts; p.noise = noise;
  p.n_points_host = n_points; p.S = 1; p.sdf = sdf; p.sdf_grad = sdf_grad;
  const int mode = sdf_grad ? 1 : 0;
  WorkspaceLayout w; make_workspace(l, n_points, 0, false, &w);
  if (mode == 1) {
    if (!workspace || workspace_bytes < w.totalBytes) return ISDF_EWORKSPACE;
    p.spill = (uint16_t*)((char*)workspace + w.offSpill); p.sp = w.sp;
  }
  // (development build only: phase stamps go to the last 4 KB of a caller-provided workspace; no-op in the shipped build)
  chain_debug_from_env(p.dbg, workspace && workspace_bytes >= 4096 ? (char*)workspace + workspace_bytes - 4096 : nullptr);
  return launch_chain(p, mode, w.nTiles, (hipStream_t)stream);
}

static int train_step_impl(const isdf_net_cfg* net, const isdf_loss_cfg* loss, const float* params, const void* shadow,
                           const isdf_step_args* a, const isdf_step_out* o, void* workspace, int64_t workspace_bytes,
                           void* stream, const isdf_optim_args* opt) {
  isdf_clear_stale_hip_error();
  NetLayout l; int rc = make_layout(net, &l);
  if (rc) return rc;
  if (!layout_supported(l)) return ISDF_EUNSUPPORTED;
  if (!loss || !params || !shadow || !a || !o || !workspace || !o->reduce_buf) return ISDF_EINVAL;
  if (!a->pc || !a->z_vals || !a->depth_sample || !a->dirs_C_sample || !a->dirs_W_sample || !a->indices_b ||
      !a->indices_h || !a->indices_w || !a->n_valid)
    return ISDF_EINVAL;
  if (a->max_rays < 1 || a->S < 1 || a->n_frames < 1 || a->H < 8 || a->W < 8) return ISDF_EINVAL;
  // the 8x8 block bins tile the image exactly (the reference's .view(-1, 8, H/8, 8, W/8) raises otherwise, loss.py:208-219)
  if (a->H % 8 != 0 || a->W % 8 != 0 || a->H >= 65536 || a->W >= 65536) return ISDF_EINVAL;
  if (loss->bounds_method != 0 && loss->bounds_method != 1) return ISDF_EUNSUPPORTED;  // "normal" is broken upstream (loss.py:29)
  if (loss->bounds_method == 1 && (!a->pc_bounds || !a->pc_grad_vec)) return ISDF_EINVAL;
  if (loss->loss_type != 0 && loss->loss_type != 1) return ISDF_EINVAL;
  if (loss->grad_weight != 0.f && !a->norm_sample) return ISDF_EINVAL;
  const int64_t maxPts = (int64_t)a->max_rays * a->S;
  if (maxPts > 0x7fffffff) return ISDF_EINVAL;   // the loss stage indexes points with 32-bit arithmetic
  // every argument check sits in front of the first launch: a rejected call leaves workspace and spills untouched
  if (a->extra_floats < 0 || a->extra_floats > 1016 || (a->extra_floats > 0 && (a->extra_slot < 0 || a->extra_slot >= a->extra_floats)))
    return ISDF_EINVAL;
  WorkspaceLayout w; make_workspace(l, maxPts, a->max_rays, true, &w);
  if (workspace_bytes < w.totalBytes) return ISDF_EWORKSPACE;
  hipStream_t st = (hipStream_t)stream;
  char* ws = (char*)workspace;
  float* wgLoss = (float*)(ws + w.offWgLoss);
  float* dwPart = (float*)(ws + w.offDwPart);
  float* vecPart = (float*)(ws + w.offVecPart);
  float* totLoss = (float*)(ws + w.offTotLoss);
  // every element of reduce_buf is written exactly once below (no memset, no atomics)

  ChainParams p = {};
  p.lay = l; p.loss = *loss; p.params = params; p.shadow = (const uint16_t*)shadow;
  p.pts = a->pc; p.noise = a->noise; p.n_valid = a->n_valid; p.S = a->S;
  p.noise_std = a->noise_std; p.noise_seed = a->noise_seed; p.noise_off = a->noise_offset;
  p.z_vals = a->z_vals; p.depth = a->depth_sample; p.dirsC = a->dirs_C_sample; p.dirsW = a->dirs_W_sample;
  p.normals = a->norm_sample; p.pc_bounds = a->pc_bounds; p.pc_grad_vec = a->pc_grad_vec;
  p.sdf = o->sdf; p.sdf_grad = o->sdf_grad; p.tot_loss_mat = o->tot_loss_mat;
  p.tot_ws = totLoss; p.wg_loss = wgLoss; p.vec_part = vecPart; p.vecStride = w.vecStride;
  p.spill = (uint16_t*)(ws + w.offSpill); p.sp = w.sp;
  p.pe_aux = (float*)(ws + w.offPeAux);
  chain_debug_from_env(p.dbg, ws + w.totalBytes - 4096);   // no-op in the shipped build (chain_debug.h)
  hipEvent_t* ev = (hipEvent_t*)o->prof_events;
  if (ev && hipEventRecord(ev[0], st) != hipSuccess) return ISDF_EHIP;
  rc = launch_chain(p, 2, w.nTiles, st);
  if (rc) return rc;
  if (ev && hipEventRecord(ev[1], st) != hipSuccess) return ISDF_EHIP;

  DwParams d = {};
  d.lay = l; d.sp = w.sp; d.spill = p.spill; d.pe_aux = p.pe_aux; d.n_valid = a->n_valid; d.S = a->S; d.dwPart = dwPart;
  rc = launch_dw(d, st);
  if (rc) return rc;
  if (ev && hipEventRecord(ev[2], st) != hipSuccess) return ISDF_EHIP;
  float* lossSums = o->reduce_buf + l.n_params;
  float* blockLoss = lossSums + 8;
  float* blockCnt = blockLoss + (int64_t)a->n_frames * 64;
  float* extra = blockCnt + (int64_t)a->n_frames * 64;   // caller-owned tail (extra_floats), right behind isdf_reduce_floats
  if (opt) {   // single-GPU tail: slab reduction + AdamW + operand repack + loss/bin finalisation in one launch
    rc = launch_step_tail(0, l, dwPart, vecPart, w.vecStride, o->reduce_buf, opt->params, opt->exp_avg, opt->exp_avg_sq,
                          (uint16_t*)opt->shadow, opt->grad_scale, opt->lr, opt->beta1, opt->beta2, opt->eps,
                          opt->weight_decay, opt->step, wgLoss, w.nTiles, a->n_valid, a->S, totLoss, a->indices_b,
                          a->indices_h, a->indices_w, a->n_frames, a->H, a->W, lossSums, blockLoss, blockCnt,
                          opt->loss_approx, opt->frame_avg, opt->frame_avg_index, st, o->host_mailbox, extra, a->extra_floats,
                          a->extra_slot, a->extra_value, 0, opt->frame_avg_inline_n, opt->frame_avg_index_inline);
    if (rc) return rc;
    if (ev && hipEventRecord(ev[3], st) != hipSuccess) return ISDF_EHIP;
    return ISDF_OK;
  }
  // two-call / data-parallel form: slab + partial reduction and loss/bin finalisation in ONE launch; the summed
  // gradient then goes to the all-reduce and isdf_adamw.  With o->split_event: TWO launches, the event between them -- the
  // message's suffix (layers from the cat layer up, out layer, loss sums, bins) is final at the event.
  const int parts = o->split_event ? 2 : 1;
  for (int part = 1; part <= parts; ++part) {
    rc = launch_step_tail(1, l, dwPart, vecPart, w.vecStride, o->reduce_buf, nullptr, nullptr, nullptr, nullptr, 1.f, 0.f,
                          0.f, 0.f, 0.f, 0.f, 1, wgLoss, w.nTiles, a->n_valid, a->S, totLoss, a->indices_b, a->indices_h,
                          a->indices_w, a->n_frames, a->H, a->W, lossSums, blockLoss, blockCnt, nullptr, nullptr, nullptr,
                          st, o->host_mailbox, extra, a->extra_floats, a->extra_slot, a->extra_value, parts == 1 ? 0 : part);
    if (rc) return rc;
    if (parts == 2 && part == 1 && hipEventRecord((hipEvent_t)o->split_event, st) != hipSuccess) return ISDF_EHIP;
  }
  if (ev && hipEventRecord(ev[3], st) != hipSuccess) return ISDF_EHIP;
  return ISDF_OK;
}

int isdf_train_step(const isdf_net_cfg* net, const isdf_loss_cfg* loss, const float* params,
                    const void* shadow, const isdf_step_args* a, const isdf_step_out* o,
                    void* workspace, int64_t workspace_bytes, void* stream) {
  return train_step_impl(net, loss, params, shadow, a, o, workspace, workspace_bytes, stream, nullptr);
}

int isdf_train_step_adamw(const isdf_net_cfg* net, const isdf_loss_cfg* loss, const isdf_step_args* a,
                          const isdf_step_out* o, const isdf_optim_args* opt, void* workspace,
                          int64_t workspace_bytes, void* stream) {
  if (!a || !opt || !opt->params || !opt->exp_avg || !opt->exp_avg_sq || !opt->shadow || opt->step < 1) return ISDF_EINVAL;
  if ((opt->loss_approx == nullptr) != (opt->frame_avg == nullptr)) return ISDF_EINVAL;   // both or neither
  if (opt->frame_avg_inline_n != 0 && (opt->frame_avg_inline_n != a->n_frames || opt->frame_avg_inline_n > ISDF_MAX_INLINE_FRAMES))
    return ISDF_EINVAL;
  for (int f = 0; f < opt->frame_avg_inline_n; ++f)   // inline indices are host values: a negative one would write in front of frame_avg
    if (opt->frame_avg_index_inline[f] < 0) return ISDF_EINVAL;
  if (o && o->split_event) return ISDF_EINVAL;   // the fused form has no message to split
  return train_step_impl(net, loss, opt->params, opt->shadow, a, o, workspace, workspace_bytes, stream, opt);
}

int isdf_train_step_finish(const isdf_net_cfg* net, const isdf_optim_args* opt, const float* reduce_buf, int32_t n_frames,
                           int32_t extra_floats, float* host_mailbox, void* stream) {
  isdf_clear_stale_hip_error();
  NetLayout l; int rc = make_layout(net, &l);
  if (rc) return rc;
  if (!layout_supported(l)) return ISDF_EUNSUPPORTED;
  if (!opt || !opt->params || !opt->exp_avg || !opt->exp_avg_sq || !opt->shadow || opt->step < 1 || !reduce_buf) return ISDF_EINVAL;
  if ((opt->loss_approx == nullptr) != (opt->frame_avg == nullptr)) return ISDF_EINVAL;   // both or neither
  if (opt->loss_approx && n_frames < 1) return ISDF_EINVAL;
  if (opt->frame_avg_inline_n != 0 && (opt->frame_avg_inline_n != n_frames || opt->frame_avg_inline_n > ISDF_MAX_INLINE_FRAMES))
    return ISDF_EINVAL;
  for (int f = 0; f < opt->frame_avg_inline_n; ++f)
    if (opt->frame_avg_index_inline[f] < 0) return ISDF_EINVAL;
  if (extra_floats < 0 || extra_floats > 1016 || n_frames < 0) return ISDF_EINVAL;
  if (extra_floats > 0 && !host_mailbox) return ISDF_EINVAL;   // the reduced tail has nowhere to go: say so instead of dropping it
  const float* lossSums = reduce_buf + l.n_params;
  const float* bl = lossSums + 8;
  const int F = opt->loss_approx ? n_frames : 0;
  return launch_adamw_pack(l, opt->params, opt->exp_avg, opt->exp_avg_sq, (uint16_t*)opt->shadow, reduce_buf,
                           lossSums + ISDF_LS_COUNT, opt->grad_scale, opt->lr, opt->beta1, opt->beta2, opt->eps,
                           opt->weight_decay, opt->step, (hipStream_t)stream, F, bl, bl + (int64_t)n_frames * 64,
                           opt->loss_approx, opt->frame_avg, opt->frame_avg_index, lossSums, bl + (int64_t)n_frames * 128,
                           extra_floats, host_mailbox, opt->frame_avg_inline_n, opt->frame_avg_index_inline);
}

int isdf_allreduce_sum_f32(isdf_nccl_allreduce_fn nccl_all_reduce, void* comm, float* buf, int64_t count, void* stream) {
  if (!nccl_all_reduce || !comm || !buf || count < 1) return ISDF_EINVAL;
  constexpr int kNcclFloat32 = 7, kNcclSum = 0;            // ncclDataType_t / ncclRedOp_t (rccl.h)
  const int rc = nccl_all_reduce(buf, buf, (size_t)count, kNcclFloat32, kNcclSum, comm, stream);
  if (rc != 0) { g_isdf_last_collective_error = rc; return ISDF_ECOLLECTIVE; }
  return ISDF_OK;
}

int isdf_bounds_pc(const int32_t* n_valid, int32_t max_rays, int32_t S, const float* pc, const float* z_vals,
                   const float* depth_sample, const float* surf_pts, int64_t n_surf, float* bounds, float* grad_vec,
                   void* stream) {
  isdf_clear_stale_hip_error();
  if (!n_valid || !pc || !z_vals || !depth_sample || !bounds || !grad_vec || max_rays < 1 || S < 1) return ISDF_EINVAL;
  if (surf_pts && n_surf < 1) return ISDF_EINVAL;
  return launch_bounds_pc(n_valid, max_rays, S, pc, z_vals, depth_sample, surf_pts, n_surf, bounds, grad_vec,
                          (hipStream_t)stream);
}

int isdf_frame_avg(const float* reduce_buf, int64_t n_params, int32_t n_frames, float* loss_approx,
                   float* frame_avg_loss, const int32_t* frame_avg_index, void* stream) {
  isdf_clear_stale_hip_error();
  if (!reduce_buf || !loss_approx || !frame_avg_loss || n_frames < 1 || n_params < 0) return ISDF_EINVAL;
  const float* bl = reduce_buf + n_params + 8;
  return launch_frame_avg(bl, bl + (int64_t)n_frames * 64, n_frames, loss_approx, frame_avg_loss, frame_avg_index,
                          (hipStream_t)stream);
}

int isdf_estimate_normals(const float* depth, int32_t H, int32_t W, float fx, float fy, float cx, float cy,
                          float* normals, void* stream) {
  isdf_clear_stale_hip_error();
  if (!depth || !normals || H < 1 || W < 1 || fx == 0.f || fy == 0.f) return ISDF_EINVAL;
  return launch_normals(depth, H, W, fx, fy, cx, cy, normals, (hipStream_t)stream);
}

int isdf_render_depth(const int32_t* n_valid, int64_t n_rays_host, int64_t max_rays, int32_t S, const float* z_vals,
                      const float* sdf, const float* depth_sample, float kf_dist_th, float* view_depth,
                      int32_t* below_count, void* stream) {
  isdf_clear_stale_hip_error();
  if (!z_vals || !sdf || !view_depth || S < 1 || max_rays < 1) return ISDF_EINVAL;
  if (!n_valid && (n_rays_host < 0 || n_rays_host > max_rays)) return ISDF_EINVAL;
  return launch_render_depth(n_valid, n_rays_host, max_rays, S, z_vals, sdf, depth_sample, kf_dist_th, view_depth,
                             below_count, (hipStream_t)stream);
}

int isdf_adamw(const isdf_net_cfg* net, float* params, float* exp_avg, float* exp_avg_sq, const float* grad_sum,
               const float* count_ptr, float grad_scale, float lr, float beta1, float beta2, float eps,
               float weight_decay, int32_t step, void* shadow, void* stream) {
  isdf_clear_stale_hip_error();
  NetLayout l; int rc = make_layout(net, &l);
  if (rc) return rc;
  if (!params || !exp_avg || !exp_avg_sq || !grad_sum || step < 1) return ISDF_EINVAL;
  if (shadow && layout_supported(l))   // update + the four packed operand copies in one launch
    return launch_adamw_pack(l, params, exp_avg, exp_avg_sq, (uint16_t*)shadow, grad_sum, count_ptr, grad_scale, lr,
                             beta1, beta2, eps, weight_decay, step, (hipStream_t)stream);
  rc = launch_adamw(params, exp_avg, exp_avg_sq, grad_sum, count_ptr, grad_scale, lr, beta1, beta2, eps, weight_decay,
                    step, l.n_params, (hipStream_t)stream);
  if (rc || !shadow) return rc;
  return launch_pack(l, params, (uint16_t*)shadow, (hipStream_t)stream);
}

}  // extern "C"
