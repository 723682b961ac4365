// Flat AdamW, MFMA-operand repacking, loss/bin finalisation, bounds_pc.
//   torch.optim.AdamW.step            isdf/modules/trainer.py:435-439,982
//   loss.frame_avg / approx_loss      isdf/modules/loss.py:208-240
//   loss.tot_loss means               isdf/modules/loss.py:187-202
//   loss.bounds_pc                    isdf/modules/loss.py:56-89
// All HBM-/latency-bound element-wise work: 16-B coalesced accesses, one pass.
#include "isdf_common.h"

namespace isdf {

// ---- AdamW (decoupled weight decay, bias-corrected) --------------------------
__global__ void adamw_kernel(float* __restrict__ p, float* __restrict__ m, float* __restrict__ v,
                             const float* __restrict__ g, const float* __restrict__ count_ptr,
                             float grad_scale, float lr, float b1, float b2, float eps, float wd,
                             float bc1, float bc2_sqrt, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float gs = grad_scale;
  if (count_ptr) gs /= *count_ptr;
  const float gi = g[i] * gs;
  float pi = p[i] * (1.f - lr * wd);
  const float mi = b1 * m[i] + (1.f - b1) * gi;
  const float vi = b2 * v[i] + (1.f - b2) * gi * gi;
  const float denom = sqrtf(vi) / bc2_sqrt + eps;
  pi -= (lr / bc1) * (mi / denom);
  p[i] = pi; m[i] = mi; v[i] = vi;
}

// ---- packed MFMA-operand copies ------------------------------------------------
// every packed matrix is [rows/32][K/16][64 lanes][8]: lane l holds row
// (l&31), k = ks*16 + 8*(l>>5) .. +8  -- exactly the A fragment of
// v_mfma_f32_32x32x16, so a wave's fragment load is one contiguous 1 KB.
__device__ __forceinline__ float fwd_src(const NetLayout& L, const float* P, int li, int row, int k) {
  const int HD = L.HD;
  if (li == 0) return k < L.E ? P[L.offW[0] + (int64_t)row * L.K[0] + k] : 0.f;
  if (li == L.cat) {
    if (k < HD) return P[L.offW[li] + (int64_t)row * L.K[li] + k];
    const int e = k - HD;
    return e < L.E ? P[L.offW[li] + (int64_t)row * L.K[li] + HD + e] : 0.f;
  }
  return P[L.offW[li] + (int64_t)row * L.K[li] + k];
}
__device__ __forceinline__ float bwd_src(const NetLayout& L, const float* P, int li, int row, int k) {
  // W_li^T restricted to the first HD inputs: [row = input i][k = output o]
  return P[L.offW[li] + (int64_t)k * L.K[li] + row];
}
__device__ __forceinline__ float g_src(const NetLayout& L, const float* P, int row, int k) {
  if (row >= L.E) return 0.f;
  if (k < L.HD) return P[L.offW[0] + (int64_t)k * L.K[0] + row];
  return P[L.offW[L.cat] + (int64_t)(k - L.HD) * L.K[L.cat] + L.HD + row];
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
  *(uint4*)(shadow + (isFwd ? L.setFwdB : L.setBwdB) + e0) = hB;
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
__global__ __launch_bounds__(1024) void finalize_kernel(const float* __restrict__ wg_loss, int64_t maxTiles,
                                                        const int32_t* __restrict__ n_valid, int S,
                                                        const float* __restrict__ tot_ws,
                                                        const int64_t* __restrict__ ib, const int64_t* __restrict__ ih,
                                                        const int64_t* __restrict__ iw, int n_frames, int H, int W,
                                                        float* __restrict__ loss_sums, float* __restrict__ block_loss,
                                                        float* __restrict__ block_cnt) {
  __shared__ float sh[16][8];
  __shared__ float binS[64], binC[64];
  __shared__ int range[2];
  __shared__ uint32_t keys[FIN_CAP];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int64_t R = *n_valid;
  const int64_t P = R * S;
  if (blockIdx.x == 0) {
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
    }
    return;
  }
  const int f = blockIdx.x - 1;
  if (tid < 2) {  // lower_bound(indices_b, f + tid): rays are sorted by frame
    int64_t lo = 0, hi = R;
    const int64_t key = f + tid;
    while (lo < hi) { const int64_t mid = (lo + hi) >> 1; if (ib[mid] < key) lo = mid + 1; else hi = mid; }
    range[tid] = (int)lo;
  }
  if (tid < 64) { binS[tid] = 0.f; binC[tid] = 0.f; }
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
    atomicAdd(&binS[bin], s);
    atomicAdd(&binC[bin], 1.f);
  }
  __syncthreads();
  if (tid < 64) { block_loss[f * 64 + tid] = binS[tid]; block_cnt[f * 64 + tid] = binC[tid]; }
}

__global__ void frame_avg_kernel(const float* __restrict__ block_loss, const float* __restrict__ block_cnt,
                                 int n_frames, float* __restrict__ loss_approx, float* __restrict__ frame_avg) {
  const int f = blockIdx.x, t = threadIdx.x;  // 64 threads
  float c = block_cnt[f * 64 + t];
  c = c == 0.f ? 1.f : c;                      // loss.py:215
  float v = block_loss[f * 64 + t] / c;
  loss_approx[f * 64 + t] = v;
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
  if (t == 0) frame_avg[f] = v / 64.f;         // loss.py:236-238
}

// ---- bounds_pc: brute-force nearest surface point, LDS-tiled -------------------
__global__ __launch_bounds__(256) void bounds_pc_kernel(const int32_t* __restrict__ n_valid, int S,
                                                        const float* __restrict__ pc, const float* __restrict__ z_vals,
                                                        const float* __restrict__ depth, float* __restrict__ bounds,
                                                        float* __restrict__ grad_vec) {
  __shared__ float sx[256], sy[256], sz[256];
  const int64_t R = *n_valid, P = R * S;
  const int64_t n = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if ((int64_t)blockIdx.x * 256 >= P) return;
  float px = 0.f, py = 0.f, pz = 0.f;
  if (n < P) { px = pc[n * 3]; py = pc[n * 3 + 1]; pz = pc[n * 3 + 2]; }
  float best = INFINITY; int64_t bi = 0;
  for (int64_t r0 = 0; r0 < R; r0 += 256) {
    const int64_t r = r0 + threadIdx.x;
    __syncthreads();
    if (r < R) { sx[threadIdx.x] = pc[r * S * 3]; sy[threadIdx.x] = pc[r * S * 3 + 1]; sz[threadIdx.x] = pc[r * S * 3 + 2]; }
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
  float gx = px - pc[bi * S * 3], gy = py - pc[bi * S * 3 + 1], gz = pz - pc[bi * S * 3 + 2];
  const float nn = sqrtf(gx * gx + gy * gy + gz * gz);
  gx /= nn; gy /= nn; gz /= nn;            // 0/0 -> NaN like the reference (trainer.py:823-824 handles it)
  if (behind) { gx = -gx; gy = -gy; gz = -gz; }
  grad_vec[n * 3] = gx; grad_vec[n * 3 + 1] = gy; grad_vec[n * 3 + 2] = gz;
}

// ---- launchers -------------------------------------------------------------------
int launch_adamw(float* p, float* m, float* v, const float* g, const float* cnt, float gs, float lr,
                 float b1, float b2, float eps, float wd, int step, int64_t n, hipStream_t st) {
  const float bc1 = 1.f - powf(b1, (float)step);
  const float bc2s = sqrtf(1.f - powf(b2, (float)step));
  hipLaunchKernelGGL(adamw_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, p, m, v, g, cnt, gs, lr,
                     b1, b2, eps, wd, bc1, bc2s, n);
  return hipGetLastError() == hipSuccess ? ISDF_OK : ISDF_EHIP;
}
int launch_pack(const NetLayout& L, const float* params, uint16_t* shadow, hipStream_t st) {
  const int64_t groups = (L.fwdSetElems + L.bwdSetElems) / 8;
  hipLaunchKernelGGL(pack_kernel, dim3((unsigned)((groups + 255) / 256)), dim3(256), 0, st, L, params, shadow);
  return hipGetLastError() == hipSuccess ? ISDF_OK : ISDF_EHIP;
}
int launch_finalize(const float* wg_loss, int64_t maxTiles, const int32_t* n_valid, int S, const float* tot_ws,
                    const int64_t* ib, const int64_t* ih, const int64_t* iw, int F, int H, int W, float* loss_sums,
                    float* bl, float* bc, hipStream_t st) {
  hipLaunchKernelGGL(finalize_kernel, dim3(1 + F), dim3(1024), 0, st, wg_loss, maxTiles, n_valid, S, tot_ws, ib, ih,
                     iw, F, H, W, loss_sums, bl, bc);
  return hipGetLastError() == hipSuccess ? ISDF_OK : ISDF_EHIP;
}
int launch_frame_avg(const float* bl, const float* bc, int F, float* la, float* fa, hipStream_t st) {
  hipLaunchKernelGGL(frame_avg_kernel, dim3(F), dim3(64), 0, st, bl, bc, F, la, fa);
  return hipGetLastError() == hipSuccess ? ISDF_OK : ISDF_EHIP;
}
int launch_bounds_pc(const int32_t* n_valid, int max_rays, int S, const float* pc, const float* z, const float* depth,
                     float* bounds, float* gv, hipStream_t st) {
  const int64_t maxPts = (int64_t)max_rays * S;
  hipLaunchKernelGGL(bounds_pc_kernel, dim3((unsigned)((maxPts + 255) / 256)), dim3(256), 0, st, n_valid, S, pc, z,
                     depth, bounds, gv);
  return hipGetLastError() == hipSuccess ? ISDF_OK : ISDF_EHIP;
}

}  // namespace isdf
