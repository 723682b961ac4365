// Flat AdamW, MFMA-operand repacking, loss/bin finalisation, bounds_pc.
//   torch.optim.AdamW.step            isdf/modules/trainer.py:435-439,982
//   loss.frame_avg / approx_loss      isdf/modules/loss.py:208-240
//   loss.tot_loss means               isdf/modules/loss.py:187-202
//   loss.bounds_pc                    isdf/modules/loss.py:56-89
// All HBM-/latency-bound element-wise work: 16-B coalesced accesses, one pass.
#include "isdf_common.h"

namespace isdf {

// ---- AdamW (decoupled weight decay, bias-corrected) --------------------------
struct AdamwCoef { float lr, b1, b2, eps, wd, bc1, bc2_sqrt; };
__device__ __forceinline__ float adamw_update(float* __restrict__ p, float* __restrict__ m, float* __restrict__ v,
                                              int64_t i, float gsum, float gs, const AdamwCoef& c) {
  // no FMA contraction: the stand-alone kernel and the fused step tail must round identically
#pragma clang fp contract(off)
  const float gi = gsum * gs;
  float pi = p[i] * (1.f - c.lr * c.wd);
  const float m0 = m[i], v0 = v[i];   // (non-temporal moments measured: no change)
  const float mi = c.b1 * m0 + (1.f - c.b1) * gi;
  const float vi = c.b2 * v0 + (1.f - c.b2) * gi * gi;
  const float denom = sqrtf(vi) / c.bc2_sqrt + c.eps;
  pi -= (c.lr / c.bc1) * (mi / denom);
  p[i] = pi;
  m[i] = mi; v[i] = vi;
  return pi;
}
__global__ void adamw_kernel(float* __restrict__ p, float* __restrict__ m, float* __restrict__ v,
                             const float* __restrict__ g, const float* __restrict__ count_ptr,
                             float grad_scale, AdamwCoef c, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float gs = grad_scale;
  if (count_ptr) { if (*count_ptr == 0.f) return; gs /= *count_ptr; }   // empty batch: no update (see step_tail_kernel)
  adamw_update(p, m, v, i, g[i], gs, c);
}

// ---- packed MFMA-operand copies ------------------------------------------------
// every packed matrix is [rows/32][K/16][64 lanes][8]: lane l holds row
// (l&31), k = ks*16 + 8*(l>>5) .. +8  -- exactly the A fragment of
// v_mfma_f32_32x32x16, so a wave's fragment load is one contiguous 1 KB.
// (row, k) are PADDED coordinates (HD hidden units, EP embedding features); the parameters have the reference's shapes [H x K]:
// everything that touches a padding unit H..HD-1 or a padding feature E..EP-1 is 0
__device__ __forceinline__ float fwd_src(const NetLayout& L, const float* P, int li, int row, int k) {
  const int HD = L.HD, H = L.H;
  if (row >= H) return 0.f;
  if (li == 0) return k < L.E ? P[L.offW[0] + (int64_t)row * L.K[0] + k] : 0.f;
  if (li == L.cat) {
    if (k < HD) return k < H ? P[L.offW[li] + (int64_t)row * L.K[li] + k] : 0.f;
    const int e = k - HD;
    return e < L.E ? P[L.offW[li] + (int64_t)row * L.K[li] + H + e] : 0.f;
  }
  return k < H ? P[L.offW[li] + (int64_t)row * L.K[li] + k] : 0.f;
}
__device__ __forceinline__ float bwd_src(const NetLayout& L, const float* P, int li, int row, int k) {
  // W_li^T restricted to the first HD inputs: [row = input i][k = output o]
  return (row < L.H && k < L.H) ? P[L.offW[li] + (int64_t)k * L.K[li] + row] : 0.f;
}
__device__ __forceinline__ float g_src(const NetLayout& L, const float* P, int row, int k) {
  if (row >= L.E) return 0.f;
  if (k < L.HD) return k < L.H ? P[L.offW[0] + (int64_t)k * L.K[0] + row] : 0.f;
  return k - L.HD < L.H ? P[L.offW[L.cat] + (int64_t)(k - L.HD) * L.K[L.cat] + L.H + row] : 0.f;
}

__global__ void pack_kernel(NetLayout L, const float* __restrict__ P, uint16_t* __restrict__ shadow) {
  const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;  // one 8-element group
  const int64_t perFwd = L.fwdSetElems / 8, perBwd = L.bwdSetElems / 8;
  if (gid >= perFwd + perBwd) return;
  const bool isFwd = gid < perFwd;
  const int64_t e0 = (isFwd ? gid : gid - perFwd) * 8;  // element offset inside the set
  float v[8];
  if (isFwd) {
    int li = 0;
    for (int k = 1; k < L.L; ++k) if (e0 >= L.fwdMat[k]) li = k;
    const int Kp = li == 0 ? L.EP : (li == L.cat ? L.HD + L.EP : L.HD);
    int64_t r = (e0 - L.fwdMat[li]) / 8;
    const int lane = (int)(r & 63); r >>= 6;
    const int ks = (int)(r % (Kp / 16)); const int rb = (int)(r / (Kp / 16));
    const int row = rb * 32 + (lane & 31), k0 = ks * 16 + 8 * (lane >> 5);
#pragma unroll
    for (int t = 0; t < 8; ++t) v[t] = fwd_src(L, P, li, row, k0 + t);
  } else {
    const bool isG = e0 >= L.bwdG;
    int li = 1;
    if (!isG) for (int k = 2; k < L.L; ++k) if (e0 >= L.bwdMat[k]) li = k;
    const int Kp = isG ? 2 * L.HD : L.HD;
    int64_t r = (e0 - (isG ? L.bwdG : L.bwdMat[li])) / 8;
    const int lane = (int)(r & 63); r >>= 6;
    const int ks = (int)(r % (Kp / 16)); const int rb = (int)(r / (Kp / 16));
    const int row = rb * 32 + (lane & 31), k0 = ks * 16 + 8 * (lane >> 5);
#pragma unroll
    for (int t = 0; t < 8; ++t) v[t] = isG ? g_src(L, P, row, k0 + t) : bwd_src(L, P, li, row, k0 + t);
  }
  const uint2 a16 = pack4<true>(v[0], v[1], v[2], v[3]), b16 = pack4<true>(v[4], v[5], v[6], v[7]);
  const uint2 abf = pack4<false>(v[0], v[1], v[2], v[3]), bbf = pack4<false>(v[4], v[5], v[6], v[7]);
  const uint4 hA = L.fwd_f16 ? make_uint4(a16.x, a16.y, b16.x, b16.y) : make_uint4(abf.x, abf.y, bbf.x, bbf.y);
  const uint4 hB = make_uint4(abf.x, abf.y, bbf.x, bbf.y);
  *(uint4*)(shadow + (isFwd ? L.setFwdA : L.setBwdA) + e0) = hA;
  if (!L.bwd_f16) *(uint4*)(shadow + (isFwd ? L.setFwdB : L.setBwdB) + e0) = hB;
  if (L.fwd_x2 && isFwd && (L.fwd_x2_all || e0 >= L.fwdMat[L.cat])) {   // fp16 residuals of the compensated layers' weights
    float r[8];
#pragma unroll
    for (int t = 0; t < 8; ++t) r[t] = f16_residual(v[t]);
    const uint2 al = pack4<true>(r[0], r[1], r[2], r[3]), bl = pack4<true>(r[4], r[5], r[6], r[7]);
    *(uint4*)(shadow + L.setFwdLo + e0) = make_uint4(al.x, al.y, bl.x, bl.y);
  }
}

// ---- loss sums + 8x8 block-loss bins ------------------------------------------
// Block 0: deterministic sum of the per-tile loss partials.  Block 1+f: frame f.
// The reference scatters per-ray loss sums into a dense [F,H,W] image and a
// 0/1 mask (loss.py:225-229, sample.py:58-61): duplicate pixels -> the LAST ray
// wins and the pixel counts once.  Here the frame's rays (contiguous in the
// compacted, frame-sorted ray list) are staged in LDS as packed pixel keys, each
// ray is dropped if a later ray has its key, and the survivors are binned with
// LDS atomics; each frame's 64 bins are then written once (no global atomics).
constexpr int FIN_CAP = 12288;   // rays per frame staged in LDS (48 KB)
struct FinalizeArgs {
  const float* wg_loss; int64_t maxTiles; const int32_t* n_valid; int S; const float* tot_ws;
  const int64_t *ib, *ih, *iw; int n_frames, H, W;
  float *loss_sums, *block_loss, *block_cnt;
  // optional (single-GPU tail only): the per-frame averages of loss.frame_avg written straight away --
  // loss_approx [F,8,8] and frame_avg[fa_index ? fa_index[f] : f] (the keyframe store's frame_avg_losses)
  float *la_out, *fa_out; const int32_t* fa_index;
  int fa_inline_n; int32_t fa_inline[8];     // the same index list as kernel arguments (isdf_optim_args.frame_avg_index_inline)
  // optional: loss sums mirrored into pinned host memory; caller-owned tail of the reduction message
  float* mailbox; float* extra; int n_extra, extra_slot; float extra_value;
};
// binS: 32.32 fixed point -- integer LDS atomics are order-independent, so the bins (hence frame_avg_losses and
// the keyframe-selection probabilities built from them) are bit-reproducible; float atomics are not
struct FinalizeLds { float sh[16][8]; unsigned long long binS[64]; float binC[64]; int range[2]; int cnt[16][2]; uint32_t keys[FIN_CAP]; };

__device__ __forceinline__ void finalize_block(int block, const FinalizeArgs& a, FinalizeLds& lds) {
  const float* __restrict__ wg_loss = a.wg_loss; const int64_t maxTiles = a.maxTiles; const int S = a.S;
  const float* __restrict__ tot_ws = a.tot_ws;
  const int64_t* __restrict__ ib = a.ib; const int64_t* __restrict__ ih = a.ih; const int64_t* __restrict__ iw = a.iw;
  const int H = a.H, W = a.W;
  float* __restrict__ loss_sums = a.loss_sums; float* __restrict__ block_loss = a.block_loss;
  float* __restrict__ block_cnt = a.block_cnt;
  auto& sh = lds.sh; auto& binS = lds.binS; auto& binC = lds.binC; auto& range = lds.range; auto& keys = lds.keys;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int64_t R = *a.n_valid;
  const int64_t P = R * S;
  if (block == 0) {
    const int64_t nTiles = (P + TILE_PTS - 1) / TILE_PTS;
    float acc[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
    for (int64_t t = tid; t < nTiles && t < maxTiles; t += 1024)
#pragma unroll
      for (int k = 0; k < 5; ++k) acc[k] += wg_loss[t * 8 + k];
#pragma unroll
    for (int k = 0; k < 5; ++k) {
      float v = acc[k];
#pragma unroll
      for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
      if (lane == 0) sh[wv][k] = v;
    }
    __syncthreads();
    if (tid < 8) {
      float v = 0.f;
      if (tid < 5) for (int k = 0; k < 16; ++k) v += sh[k][tid];
      loss_sums[tid] = v;
      if (a.mailbox) a.mailbox[tid] = v;   // pinned host memory: valid after the caller's closing stream synchronisation
    }
    if (tid < a.n_extra) a.extra[tid] = tid == a.extra_slot ? a.extra_value : 0.f;
    return;
  }
  const int f = block - 1;
  // lower_bound(indices_b, f) and lower_bound(indices_b, f + 1): the rays are sorted by frame, so both are COUNTS
  // (rays of earlier frames / of frames <= f), taken by all threads in one memory round trip.  (A two-thread binary search
  // here was ten dependent round trips, ~10 us, and set the duration of the whole step-tail launch.)
  {
    int c0 = 0, c1 = 0;
    for (int64_t r = tid; r < R; r += 1024) { const int64_t b = ib[r]; c0 += b < f; c1 += b <= f; }
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) { c0 += __shfl_xor(c0, m, 64); c1 += __shfl_xor(c1, m, 64); }
    if (lane == 0) { lds.cnt[wv][0] = c0; lds.cnt[wv][1] = c1; }
  }
  if (tid < 64) { binS[tid] = 0ull; binC[tid] = 0.f; }
  __syncthreads();
  if (tid < 2) {
    int c = 0;
    for (int k = 0; k < 16; ++k) c += lds.cnt[k][tid];
    range[tid] = c;
  }
  __syncthreads();
  const int lo = range[0], n = range[1] - range[0];
  const bool staged = n <= FIN_CAP;
  if (staged) for (int r = tid; r < n; r += 1024) keys[r] = ((uint32_t)ih[lo + r] << 16) | (uint32_t)iw[lo + r];
  __syncthreads();
  const int hb = H / 8, wb = W / 8;
  for (int r = tid; r < n; r += 1024) {
    const uint32_t h = (uint32_t)ih[lo + r], w = (uint32_t)iw[lo + r];
    const uint32_t key = (h << 16) | w;
    bool dup = false;
    if (staged) { for (int q = r + 1; q < n; ++q) dup |= keys[q] == key; }
    else { for (int q = r + 1; q < n; ++q) dup |= ((((uint32_t)ih[lo + q]) << 16) | (uint32_t)iw[lo + q]) == key; }
    if (dup) continue;
    float s = 0.f;
    const float* tp = tot_ws + (int64_t)(lo + r) * S;
    for (int k = 0; k < S; ++k) s += tp[k];               // total_loss_mat.sum(-1), loss.py:229
    const int bin = (int)((h / hb) * 8 + (w / wb));
    atomicAdd(&binS[bin], (unsigned long long)(long long)llrint((double)s * 4294967296.0));
    atomicAdd(&binC[bin], 1.f);
  }
  __syncthreads();
  if (tid < 64) {
    const float bs = (float)((double)(long long)binS[tid] * (1.0 / 4294967296.0));
    block_loss[f * 64 + tid] = bs; block_cnt[f * 64 + tid] = binC[tid];
    if (a.la_out) {   // frame_avg_kernel's arithmetic, on the bins just built (loss.py:208-240)
      float c = binC[tid];
      c = c == 0.f ? 1.f : c;
      float v = bs / c;
      a.la_out[f * 64 + tid] = v;
#pragma unroll
      for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
      if (tid == 0) a.fa_out[a.fa_inline_n ? a.fa_inline[f] : (a.fa_index ? a.fa_index[f] : f)] = v / 64.f;
    }
  }
}

// ---- fused step tail (single-GPU path) ------------------------------------------
// One launch after the dW kernel: [weights: sum the K-split dW slabs -> gradient -> AdamW -> the four packed
// 16-bit operand copies] | [biases / out layer: sum the per-tile partials -> gradient -> AdamW] | [loss sums and
// per-frame block bins].  Replaces dw_reduce + vec_reduce + finalize + adamw + pack (5 launches, ~36 us of a
// 347 us step) for world_size 1; the data-parallel path keeps them apart because the all-reduce of the
// gradient sits between the reduction and the update.  Same summation order and AdamW arithmetic as the
// separate kernels, so both paths produce bit-identical parameters.
struct TailParams {
  NetLayout lay;
  const float* dwPart; const float* vecPart; int32_t vecStride;
  float* grad;                       // [n_params] summed gradient (still written: reduce_buf contract)
  float *params, *m, *v; uint16_t* shadow;
  AdamwCoef c; float grad_scale;     // gradient = sum * grad_scale / (n_valid * S)   [PHASE 0]
  const float* count_ptr;            // PHASE 2: gradient = grad[] * grad_scale / *count_ptr (reduced count)
  FinalizeArgs fin;
  int nW, nV;                        // blocks of the weight and the vector sections
  int wBlock0;                       // first weight block of this launch (a split tail runs the weight section in two launches)
};

__device__ __forceinline__ void shadow_put(const NetLayout& L, uint16_t* sh, bool fwdSet, int64_t elem, float val,
                                           bool withResidual = false) {
  const uint32_t h = pack4<true>(val, 0.f, 0.f, 0.f).x & 0xffffu, b = pack4<false>(val, 0.f, 0.f, 0.f).x & 0xffffu;
  sh[(fwdSet ? L.setFwdA : L.setBwdA) + elem] = (uint16_t)(L.fwd_f16 ? h : b);
  if (!L.bwd_f16) sh[(fwdSet ? L.setFwdB : L.setBwdB) + elem] = (uint16_t)b;   // the bf16 copies serve only the bf16 second-order sweeps
  if (withResidual) sh[L.setFwdLo + elem] = (uint16_t)(pack4<true>(f16_residual(val), 0.f, 0.f, 0.f).x & 0xffffu);
}
// element offset of (row, k) inside a packed [rows/32][Kp/16][64][8] matrix (see pack_kernel)
__device__ __forceinline__ int64_t packed_elem(int row, int k, int Kp) {
  return ((((int64_t)(row >> 5) * (Kp >> 4) + (k >> 4)) * 64 + (row & 31) + 32 * ((k & 15) >> 3)) << 3) + (k & 7);
}

// PHASE 0: everything (single GPU).  PHASE 1: reduction + finalisation only (isdf_train_step: the gradient
// sums go to the all-reduce).  PHASE 2: AdamW + operand repack from an already reduced gradient (isdf_adamw).
// The 66 MB of K-split slabs are read exactly once: non-temporal, so they do not evict the packed weight copies this kernel
// writes for the next step (measured: next chain kernel -3 %, dW -5 %, step +3.5 %).
// loss.frame_avg (loss.py:208-240) of frame f from the (all-reduced) bins: 64 threads, one per 8x8 block
__device__ __forceinline__ void frame_avg_block(int f, int t, const float* __restrict__ block_loss,
                                                const float* __restrict__ block_cnt, float* __restrict__ loss_approx,
                                                float* __restrict__ frame_avg, const int32_t* __restrict__ fa_index,
                                                int dst_inline = -1) {
  if (t >= 64) return;
  float c = block_cnt[f * 64 + t];
  c = c == 0.f ? 1.f : c;                      // loss.py:215
  float v = block_loss[f * 64 + t] / c;
  loss_approx[f * 64 + t] = v;
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
  if (t == 0) frame_avg[dst_inline >= 0 ? dst_inline : (fa_index ? fa_index[f] : f)] = v / 64.f;         // loss.py:236-238
}

template <int PHASE>
__global__ __launch_bounds__(1024) void step_tail_kernel(const TailParams p) {
  __shared__ FinalizeLds lds;
  const NetLayout& L = p.lay;
  const int HD = L.HD;
  const int b = blockIdx.x;
  if (b >= p.nW + p.nV) {
    if (PHASE != 2) finalize_block(b - p.nW - p.nV, p.fin, lds);
    else if (b - p.nW - p.nV < p.fin.n_frames)
      frame_avg_block(b - p.nW - p.nV, threadIdx.x, p.fin.block_loss, p.fin.block_cnt, p.fin.la_out, p.fin.fa_out, p.fin.fa_index,
                      p.fin.fa_inline_n ? p.fin.fa_inline[b - p.nW - p.nV] : -1);
    else if ((int)threadIdx.x < 8 + p.fin.n_extra)   // last block of isdf_train_step_finish: the host's view of the reduced message
      p.fin.mailbox[threadIdx.x] = threadIdx.x < 8 ? p.fin.loss_sums[threadIdx.x] : p.fin.extra[threadIdx.x - 8];
    return;
  }
  const int64_t P = PHASE == 2 ? 0 : (int64_t)(*p.fin.n_valid) * p.fin.S;
  float gs = p.grad_scale;
  if (PHASE == 0) gs /= (float)P;
  if (PHASE == 2 && p.count_ptr) gs /= *p.count_ptr;
  // A batch without a single valid ray (all sampled depths 0): upstream the means over empty tensors are NaN and AdamW poisons
  // every weight (loss.py:187-202).  Here the update is SKIPPED -- sums and the count (0) are still reported, so the caller sees
  // what happened -- the one deliberate deviation from the reference's arithmetic on this path.
  const bool emptyBatch = (PHASE == 0 && P == 0) || (PHASE == 2 && p.count_ptr && *p.count_ptr == 0.f);
  if (b < p.nW) {
    // ---- weights (dw_reduce_kernel's mapping: one thread per element of a 256x256 dW unit)
    const int64_t idx = (int64_t)(b + p.wBlock0) * 1024 + threadIdx.x;
    constexpr int64_t perUnit = (int64_t)DW_BLK * DW_BLK;
    const int unit = (int)(idx / perUnit);
    if (unit >= dw_units(L)) return;
    const DwUnit du = dw_unit(L, unit);
    const int rem = (int)(idx - unit * perUnit);
    const int o = du.ob * DW_BLK + rem / DW_BLK;
    const int ip = du.ib * DW_BLK + rem % DW_BLK;          // padded input column
    const int li = du.li;
    if (o >= L.H) return;                                  // padding unit (NetLayout::H): no parameter behind it
    if (li == 0 && ip >= L.E) return;
    if (li == L.cat && ip >= HD && ip - HD >= L.E) return;
    if ((li != 0 && ip < HD && ip >= L.H)) return;         // padding input unit
    // column in the fp32 weight [H x K_li]: the cat layer's embedding columns follow its H hidden ones
    const int col = (li == L.cat && ip >= HD) ? L.H + (ip - HD) : ip;
    const int64_t pi = L.offW[li] + (int64_t)o * L.K[li] + col;
    float s = 0.f;
    if (PHASE == 2) s = p.grad[pi];
    else {
      const slab_t* src = (const slab_t*)p.dwPart + (int64_t)dw_slab_base(L, unit) * perUnit + rem;
      // all K-split slabs of this element in flight at once: with 4 at a time the kernel was nine dependent HBM round
      // trips long (23-26 us for 88 MB).  A unit has DW_SPLIT_REG slabs, or DW_SPLIT_PE when the dW kernel rebuilds its input operand (isdf_common.h).
#pragma unroll
      for (int k = 0; k < DW_SPLIT_REG; ++k) s += __builtin_nontemporal_load(src + (int64_t)k * perUnit);
      if (dw_unit_from_emb(L, du)) {
#pragma unroll
        for (int k = DW_SPLIT_REG; k < DW_SPLIT_PE; ++k) s += __builtin_nontemporal_load(src + (int64_t)k * perUnit);
      }
      p.grad[pi] = s;
      if (PHASE == 1) return;
    }
    if (emptyBatch) return;
    const float w = adamw_update(p.params, p.m, p.v, pi, s, gs, p.c);
    // packed operand copies (pack_kernel's sources, inverted): forward orientation ...
    const int KpF = li == 0 ? L.EP : (li == L.cat ? HD + L.EP : HD);
    shadow_put(L, p.shadow, true, L.fwdMat[li] + packed_elem(o, ip, KpF), w, L.fwd_x2 && (L.fwd_x2_all || li >= L.cat));
    // ... W^T restricted to the first HD inputs (layers >= 1) ...
    if (li >= 1 && ip < HD) shadow_put(L, p.shadow, false, L.bwdMat[li] + packed_elem(ip, o, HD), w);
    // ... and the embedding-gradient matrix [W_in^T | W_cat[:, HD:]^T]
    if (li == 0) shadow_put(L, p.shadow, false, L.bwdG + packed_elem(ip, o, 2 * HD), w);
    else if (li == L.cat && ip >= HD) shadow_put(L, p.shadow, false, L.bwdG + packed_elem(ip - HD, HD + o, 2 * HD), w);
    return;
  }
  // ---- biases (L*HD), w_out (HD), b_out (1): vec_reduce_kernel's mapping, 64 parameters x 16 tile groups
  float (*sh)[64] = (float (*)[64])lds.keys;
  const int pi = threadIdx.x & 63, g = threadIdx.x >> 6;
  const int v = (b - p.nW) * 64 + pi;
  const int nVec = L.L * HD + HD + 1;
  const int nTiles = (int)((P + TILE_PTS - 1) / TILE_PTS);
  int slotA = 0, slotB = -1, dst = -1;
  if (v < L.L * HD) { slotA = v; if (v % HD < L.H) dst = L.offB[v / HD] + v % HD; }          // (padding units: no parameter)
  else if (v < L.L * HD + HD) { slotA = v; slotB = v + HD; if (v - L.L * HD < L.H) dst = L.offWout + (v - L.L * HD); }
  else if (v < nVec) { slotA = L.L * HD + 2 * HD; dst = L.offBout; }
  if (PHASE == 2) {
    if (g == 0 && dst >= 0 && !emptyBatch) adamw_update(p.params, p.m, p.v, dst, p.grad[dst], gs, p.c);
    return;
  }
  float s = 0.f, s2 = 0.f;
  if (dst >= 0) {
    int t = g;
    for (; t + 112 < nTiles; t += 128) {    // 8 (16 with slotB) independent loads in flight per thread
      const float* r0 = p.vecPart + (int64_t)t * p.vecStride;
      float a[8], bb[8];
#pragma unroll
      for (int q = 0; q < 8; ++q) a[q] = r0[(int64_t)(16 * q) * p.vecStride + slotA];
      if (slotB >= 0) {
#pragma unroll
        for (int q = 0; q < 8; ++q) bb[q] = r0[(int64_t)(16 * q) * p.vecStride + slotB];
        s2 += ((bb[0] + bb[1]) + (bb[2] + bb[3])) + ((bb[4] + bb[5]) + (bb[6] + bb[7]));
      }
      s += ((a[0] + a[1]) + (a[2] + a[3])) + ((a[4] + a[5]) + (a[6] + a[7]));
    }
    for (; t + 48 < nTiles; t += 64) {      // 4 independent loads in flight per thread
      const float* r0 = p.vecPart + (int64_t)t * p.vecStride;
      const float a0 = r0[slotA], a1 = r0[(int64_t)16 * p.vecStride + slotA],
                  a2 = r0[(int64_t)32 * p.vecStride + slotA], a3 = r0[(int64_t)48 * p.vecStride + slotA];
      s += (a0 + a1) + (a2 + a3);
      if (slotB >= 0) {
        const float b0 = r0[slotB], b1 = r0[(int64_t)16 * p.vecStride + slotB],
                    b2 = r0[(int64_t)32 * p.vecStride + slotB], b3 = r0[(int64_t)48 * p.vecStride + slotB];
        s2 += (b0 + b1) + (b2 + b3);
      }
    }
    for (; t < nTiles; t += 16) {
      const float* row = p.vecPart + (int64_t)t * p.vecStride;
      s += row[slotA];
      if (slotB >= 0) s2 += row[slotB];
    }
    s += s2;
  }
  sh[g][pi] = s;
  __syncthreads();
  if (g == 0 && dst >= 0) {
    float t = 0.f;
#pragma unroll
    for (int k = 0; k < 16; ++k) t += sh[k][pi];
    p.grad[dst] = t;
    if (PHASE == 0 && !emptyBatch) adamw_update(p.params, p.m, p.v, dst, t, gs, p.c);
  }
}

__global__ void frame_avg_kernel(const float* __restrict__ block_loss, const float* __restrict__ block_cnt,
                                 int n_frames, float* __restrict__ loss_approx, float* __restrict__ frame_avg,
                                 const int32_t* __restrict__ fa_index) {
  frame_avg_block(blockIdx.x, threadIdx.x, block_loss, block_cnt, loss_approx, frame_avg, fa_index);   // 64 threads
}

// ---- bounds_pc: brute-force nearest surface point, LDS-tiled -------------------
// surf == nullptr: the surface set is this batch's own surface samples pc[:, 0, :] (single process,
// loss.py:58-61).  Data parallel: surf[n_surf][3] is the all-gathered surface set of every rank's rays
// (SURVEY 8e); slots of invalid rays hold +inf-like sentinels and are never the nearest point.
__global__ __launch_bounds__(256) void bounds_pc_kernel(const int32_t* __restrict__ n_valid, int S,
                                                        const float* __restrict__ pc, const float* __restrict__ z_vals,
                                                        const float* __restrict__ depth, const float* __restrict__ surf,
                                                        int64_t n_surf, float* __restrict__ bounds,
                                                        float* __restrict__ grad_vec) {
  __shared__ float sx[256], sy[256], sz[256];
  const int64_t Rl = *n_valid, P = Rl * S;
  const int64_t R = surf ? n_surf : Rl;                  // size of the surface set
  const int64_t sstride = surf ? 3 : (int64_t)S * 3;
  const float* sp = surf ? surf : pc;
  const int64_t n = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if ((int64_t)blockIdx.x * 256 >= P) return;
  float px = 0.f, py = 0.f, pz = 0.f;
  if (n < P) { px = pc[n * 3]; py = pc[n * 3 + 1]; pz = pc[n * 3 + 2]; }
  float best = INFINITY; int64_t bi = 0;
  for (int64_t r0 = 0; r0 < R; r0 += 256) {
    const int64_t r = r0 + threadIdx.x;
    __syncthreads();
    if (r < R) { sx[threadIdx.x] = sp[r * sstride]; sy[threadIdx.x] = sp[r * sstride + 1]; sz[threadIdx.x] = sp[r * sstride + 2]; }
    __syncthreads();
    const int cnt = (int)((R - r0) < 256 ? (R - r0) : 256);
    for (int k = 0; k < cnt; ++k) {
      const float dx = px - sx[k], dy = py - sy[k], dz = pz - sz[k];
      const float d2 = __fadd_rn(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)), __fmul_rn(dz, dz));
      if (d2 < best) { best = d2; bi = r0 + k; }
    }
  }
  if (n >= P) return;
  const int64_t ray = n / S;
  const float dist = sqrtf(best);
  const bool behind = z_vals[n] > depth[ray];
  bounds[n] = behind ? -dist : dist;
  float gx = px - sp[bi * sstride], gy = py - sp[bi * sstride + 1], gz = pz - sp[bi * sstride + 2];
  const float nn = sqrtf(gx * gx + gy * gy + gz * gz);
  gx /= nn; gy /= nn; gz /= nn;            // 0/0 -> NaN like the reference (trainer.py:823-824 handles it)
  if (behind) { gx = -gx; gy = -gy; gz = -gz; }
  grad_vec[n * 3] = gx; grad_vec[n * 3 + 1] = gy; grad_vec[n * 3 + 2] = gz;
}

// ---- launchers -------------------------------------------------------------------
int launch_adamw(float* p, float* m, float* v, const float* g, const float* cnt, float gs, float lr,
                 float b1, float b2, float eps, float wd, int step, int64_t n, hipStream_t st) {
  const AdamwCoef c = {lr, b1, b2, eps, wd, 1.f - powf(b1, (float)step), sqrtf(1.f - powf(b2, (float)step))};
  hipLaunchKernelGGL(adamw_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, p, m, v, g, cnt, gs, c, n);
  return isdf_launch_status();
}
int launch_pack(const NetLayout& L, const float* params, uint16_t* shadow, hipStream_t st) {
  const int64_t groups = (L.fwdSetElems + L.bwdSetElems) / 8;
  hipLaunchKernelGGL(pack_kernel, dim3((unsigned)((groups + 255) / 256)), dim3(256), 0, st, L, params, shadow);
  return isdf_launch_status();
}
static FinalizeArgs finalize_args(const float* wg_loss, int64_t maxTiles, const int32_t* n_valid, int S, const float* tot_ws,
                                  const int64_t* ib, const int64_t* ih, const int64_t* iw, int F, int H, int W,
                                  float* loss_sums, float* bl, float* bc) {
  FinalizeArgs a = {};
  a.wg_loss = wg_loss; a.maxTiles = maxTiles; a.n_valid = n_valid; a.S = S; a.tot_ws = tot_ws; a.ib = ib; a.ih = ih; a.iw = iw;
  a.n_frames = F; a.H = H; a.W = W; a.loss_sums = loss_sums; a.block_loss = bl; a.block_cnt = bc;
  return a;
}
// phase 0: params/m/v/shadow + optim scalars + finalize args; phase 1: grad + finalize args only;
// phase 2 (launch_adamw_pack): params/m/v/shadow + optim scalars + count_ptr, grad = reduced gradient
int launch_step_tail(int phase, const NetLayout& L, const float* dwPart, const float* vecPart, int vecStride, float* grad,
                     float* params, float* m, float* v, uint16_t* shadow, float grad_scale, float lr, float b1, float b2,
                     float eps, float wd, int step, const float* wg_loss, int64_t maxTiles, const int32_t* n_valid, int S,
                     const float* tot_ws, const int64_t* ib, const int64_t* ih, const int64_t* iw, int F, int H, int W,
                     float* loss_sums, float* bl, float* bc, float* la_out, float* fa_out, const int32_t* fa_index,
                     hipStream_t st, float* mailbox, float* extra, int n_extra, int extra_slot, float extra_value, int part,
                     int fa_inline_n, const int32_t* fa_inline) {
  // part 0: everything in one launch.  Split tail (phase 1 only): part 1 = weight blocks of the dW units from the cat layer up +
  // vector section + finalisation (the message's suffix, isdf_reduce_split_floats), part 2 = the weight blocks below.
  TailParams p = {};
  p.lay = L; p.dwPart = dwPart; p.vecPart = vecPart; p.vecStride = vecStride; p.grad = grad;
  p.params = params; p.m = m; p.v = v; p.shadow = shadow; p.grad_scale = grad_scale;
  if (phase == 0)
    p.c = AdamwCoef{lr, b1, b2, eps, wd, 1.f - powf(b1, (float)step), sqrtf(1.f - powf(b2, (float)step))};
  p.fin = finalize_args(wg_loss, maxTiles, n_valid, S, tot_ws, ib, ih, iw, F, H, W, loss_sums, bl, bc);
  if (la_out && fa_out) {
    p.fin.la_out = la_out; p.fin.fa_out = fa_out; p.fin.fa_index = fa_index;
    p.fin.fa_inline_n = fa_inline_n;
    for (int k = 0; k < fa_inline_n && k < 8; ++k) p.fin.fa_inline[k] = fa_inline[k];
  }
  p.fin.mailbox = mailbox; p.fin.extra = extra; p.fin.n_extra = n_extra; p.fin.extra_slot = extra_slot; p.fin.extra_value = extra_value;
  const int64_t total = (int64_t)dw_units(L) * DW_BLK * DW_BLK;
  p.nW = (int)((total + 1023) / 1024);
  p.nV = (L.L * L.HD + L.HD + 1 + 63) / 64;
  int tailBlocks = p.nV + 1 + F;
  if (part != 0) {
    if (phase != 1) return ISDF_EINVAL;
    int unitsBelow = 0;                     // dW units of the layers below the cat layer (units are numbered layer by layer)
    for (int li = 0; li < L.cat; ++li) unitsBelow += (L.HD / DW_BLK) * (dw_kpad(L, li) / DW_BLK);
    const int split = unitsBelow * (DW_BLK * DW_BLK / 1024);
    if (part == 1) { p.wBlock0 = split; p.nW -= split; }
    else { p.nW = split; p.nV = 0; tailBlocks = 0; }
  }
  const dim3 grid((unsigned)(p.nW + tailBlocks));
  if (grid.x == 0) return ISDF_OK;
  if (phase == 0) hipLaunchKernelGGL(step_tail_kernel<0>, grid, dim3(1024), 0, st, p);
  else hipLaunchKernelGGL(step_tail_kernel<1>, grid, dim3(1024), 0, st, p);
  return isdf_launch_status();
}
// n_frames > 0: the same launch also turns the (all-reduced) bins into loss_approx / frame averages (the data-parallel
// step's closing launch, isdf_train_step_finish)
int launch_adamw_pack(const NetLayout& L, float* params, float* m, float* v, uint16_t* shadow, const float* grad,
                      const float* count_ptr, float grad_scale, float lr, float b1, float b2, float eps, float wd,
                      int step, hipStream_t st, int n_frames = 0, const float* bl = nullptr, const float* bc = nullptr,
                      float* la = nullptr, float* fa = nullptr, const int32_t* fa_index = nullptr,
                      const float* loss_sums = nullptr, const float* extra = nullptr, int n_extra = 0, float* mailbox = nullptr,
                      int fa_inline_n = 0, const int32_t* fa_inline = nullptr) {
  TailParams p = {};
  p.fin.fa_inline_n = fa_inline_n;
  for (int k = 0; k < fa_inline_n && k < 8; ++k) p.fin.fa_inline[k] = fa_inline[k];
  p.fin.block_loss = const_cast<float*>(bl); p.fin.block_cnt = const_cast<float*>(bc);
  p.fin.la_out = la; p.fin.fa_out = fa; p.fin.fa_index = fa_index; p.fin.n_frames = n_frames;
  p.fin.loss_sums = const_cast<float*>(loss_sums); p.fin.extra = const_cast<float*>(extra); p.fin.n_extra = n_extra;
  p.fin.mailbox = mailbox;
  p.lay = L; p.grad = const_cast<float*>(grad); p.params = params; p.m = m; p.v = v; p.shadow = shadow;
  p.grad_scale = grad_scale; p.count_ptr = count_ptr;
  p.c = AdamwCoef{lr, b1, b2, eps, wd, 1.f - powf(b1, (float)step), sqrtf(1.f - powf(b2, (float)step))};
  const int64_t total = (int64_t)dw_units(L) * DW_BLK * DW_BLK;
  p.nW = (int)((total + 1023) / 1024);
  p.nV = (L.L * L.HD + L.HD + 1 + 63) / 64;
  hipLaunchKernelGGL(step_tail_kernel<2>, dim3((unsigned)(p.nW + p.nV + n_frames + (mailbox ? 1 : 0))), dim3(1024), 0, st, p);
  return isdf_launch_status();
}
int launch_frame_avg(const float* bl, const float* bc, int F, float* la, float* fa, const int32_t* fa_index,
                     hipStream_t st) {
  hipLaunchKernelGGL(frame_avg_kernel, dim3(F), dim3(64), 0, st, bl, bc, F, la, fa, fa_index);
  return isdf_launch_status();
}
int launch_bounds_pc(const int32_t* n_valid, int max_rays, int S, const float* pc, const float* z, const float* depth,
                     const float* surf, int64_t n_surf, float* bounds, float* gv, hipStream_t st) {
  const int64_t maxPts = (int64_t)max_rays * S;
  hipLaunchKernelGGL(bounds_pc_kernel, dim3((unsigned)((maxPts + 255) / 256)), dim3(256), 0, st, n_valid, S, pc, z,
                     depth, surf, n_surf, bounds, gv);
  return isdf_launch_status();
}

}  // namespace isdf
