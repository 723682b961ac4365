// Ray / point sampler (K1).  Replaces, per training step,
//   sample.sample_pixels      isdf/modules/sample.py:11-21
//   sample.get_batch_data     isdf/modules/sample.py:24-74   (gather + validity + compaction)
//   transform.origin_dirs_W   isdf/geometry/transform.py:36-41
//   sample.stratified_sample  isdf/modules/sample.py:77-128
//   sample.sample_along_rays  isdf/modules/sample.py:131-178
// and computes dirs_C from the intrinsics (transform.py:13-33) instead of
// gathering it from a [H,W,3] table.  No [F,H,W] mask image is materialised
// (sample.py:58-61): the block-loss bins are built from the ray list instead
// (optim.hip).
//
// The work per step is tiny (F*n_rays rays), i.e. latency- not bandwidth-bound:
// pass 1 is ONE persistent workgroup that walks the drawn rays in order, so the
// compaction is order-preserving by construction and needs no inter-workgroup
// protocol; pass 2 is one thread per (ray, sample).
#include "isdf_common.h"

namespace isdf {

__device__ __forceinline__ uint4 ray_random(const isdf_sample_args& a, uint32_t ray, uint32_t slot) {
  return philox4x32_10(make_uint4(ray, slot, (uint32_t)a.offset, (uint32_t)(a.offset >> 32)),
                       make_uint2((uint32_t)a.seed, (uint32_t)(a.seed >> 32)));
}

__global__ __launch_bounds__(1024) void sample_pixels_kernel(const isdf_sample_args a, const isdf_sample_out o) {
  __shared__ int waveCnt[16];
  __shared__ int baseSh;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int total = a.n_frames * a.n_rays;
  if (tid == 0) baseSh = 0;
  __syncthreads();
  for (int r0 = 0; r0 < total; r0 += 1024) {
    const int r = r0 + tid;
    bool valid = false;
    int b = 0, h = 0, wq = 0; float d = 0.f, n0 = 0.f, n1 = 0.f, n2 = 0.f;
    if (r < total) {
      b = r / a.n_rays;  // indices_b = arange(F).repeat_interleave(n_rays), sample.py:18-19
      if (a.rng_mode == 0) { h = (int)a.draw_h[r]; wq = (int)a.draw_w[r]; }
      else { const uint4 u = ray_random(a, (uint32_t)r, 0u); h = (int)(u.x % (uint32_t)a.H); wq = (int)(u.y % (uint32_t)a.W); }
      const int64_t pix = (int64_t)h * a.W + wq;
      d = a.depth_batch[(int64_t)a.frame_idx[b] * a.H * a.W + pix];
      valid = d != 0.f;                                       // sample.py:39-40
      if (a.normal_batch) {
        const float* np = a.normal_batch + ((int64_t)a.normal_idx[b] * a.H * a.W + pix) * 3;
        n0 = np[0]; n1 = np[1]; n2 = np[2];
        valid = valid && !(n0 != n0);                         // sample.py:47-49
      }
    }
    // ordered compaction: wave ballot + prefix over the 16 waves
    const unsigned long long m = __ballot(valid);
    const int before = __popcll(m & ((1ull << lane) - 1ull));
    if (lane == 0) waveCnt[wv] = __popcll(m);
    __syncthreads();
    int wbase = baseSh;
    for (int k = 0; k < wv; ++k) wbase += waveCnt[k];
    if (valid) {
      const int q = wbase + before;
      o.indices_b[q] = b; o.indices_h[q] = h; o.indices_w[q] = wq;
      o.depth_sample[q] = d;
      if (o.norm_sample) { o.norm_sample[q * 3] = n0; o.norm_sample[q * 3 + 1] = n1; o.norm_sample[q * 3 + 2] = n2; }
      // ray_dirs_C, transform.py:13-33 ('z' depth)
      const float dx = ((float)wq - a.cx) / a.fx, dy = ((float)h - a.cy) / a.fy, dz = 1.f;
      o.dirs_C_sample[q * 3] = dx; o.dirs_C_sample[q * 3 + 1] = dy; o.dirs_C_sample[q * 3 + 2] = dz;
      const float* T = a.T_WC_batch + (int64_t)a.frame_idx[b] * 16;
      if (o.T_WC_sample) {
#pragma unroll
        for (int k = 0; k < 16; ++k) o.T_WC_sample[(int64_t)q * 16 + k] = T[k];
      }
      // origin_dirs_W, transform.py:36-41: (R * d).sum(-1), no fused multiply-add
#pragma unroll
      for (int i = 0; i < 3; ++i) {
        const float s = __fadd_rn(__fadd_rn(__fmul_rn(T[i * 4], dx), __fmul_rn(T[i * 4 + 1], dy)), __fmul_rn(T[i * 4 + 2], dz));
        o.dirs_W_sample[q * 3 + i] = s;
      }
    }
    __syncthreads();
    if (tid == 0) { int s = baseSh; for (int k = 0; k < 16; ++k) s += waveCnt[k]; baseSh = s; }
    __syncthreads();
  }
  if (tid == 0) *o.n_valid = baseSh;
}

__global__ void sample_along_rays_kernel(const isdf_sample_args a, const isdf_sample_out o) {
  const int S = a.n_strat + a.n_surf;
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t R = *o.n_valid;
  if (idx >= R * S) return;
  const int r = (int)(idx / S), s = (int)(idx - (int64_t)r * S);
  const float depth = o.depth_sample[r];
  const float maxd = __fadd_rn(depth, a.dist_behind_surf);   // trainer.py:741
  float z;
  if (s < a.n_surf) {
    if (s == 0) z = depth;                                     // sample.py:158
    else {
      float off;
      if (a.rng_mode == 0) off = a.draw_n[(int64_t)r * (a.n_surf - 1) + (s - 1)];
      else {  // Box-Muller, sigma 0.1 (sample.py:160-162)
        const uint4 u = ray_random(a, (uint32_t)r, 1u + (uint32_t)s);
        const float u1 = fmaxf(u01(u.x), 1e-7f), u2 = u01(u.y);
        off = 0.1f * sqrtf(-2.f * __logf(u1)) * __cosf(6.2831853f * u2);
      }
      z = fminf(fmaxf(__fadd_rn(depth, off), a.min_depth), maxd);  // clamp, sample.py:167-171
    }
  } else {
    const int k = s - a.n_surf, nb = a.n_strat;
    float U;
    if (a.rng_mode == 0) U = a.draw_u[(int64_t)r * nb + k];
    else U = u01(ray_random(a, (uint32_t)r, 64u + (uint32_t)k).x);
    // torch.linspace(0, 1, nb+1)[k] in fp32 (symmetric evaluation), sample.py:96-98
    const float step = 1.f / (float)nb;
    const float lin = k < (nb + 1) / 2 ? __fmul_rn(step, (float)k) : __fadd_rn(1.f, -__fmul_rn(step, (float)(nb - k)));
    const float range = __fadd_rn(maxd, -a.min_depth);
    const float lim = __fadd_rn(__fmul_rn(lin, range), a.min_depth);
    const float blen = range / (float)nb;
    z = __fadd_rn(lim, __fmul_rn(U, blen));                       // sample.py:123-126
  }
  o.z_vals[idx] = z;
  const int b = (int)o.indices_b[r];
  const float* T = a.T_WC_batch + (int64_t)a.frame_idx[b] * 16;
#pragma unroll
  for (int i = 0; i < 3; ++i)  // pc = origins + dirs_W * z, sample.py:176
    o.pc[idx * 3 + i] = __fadd_rn(T[i * 4 + 3], __fmul_rn(o.dirs_W_sample[r * 3 + i], z));
}

int launch_sample_pixels(const isdf_sample_args& a, const isdf_sample_out& o, hipStream_t st) {
  hipLaunchKernelGGL(sample_pixels_kernel, dim3(1), dim3(1024), 0, st, a, o);
  return isdf_launch_status();
}
int launch_sample_along_rays(const isdf_sample_args& a, const isdf_sample_out& o, hipStream_t st) {
  const int64_t maxPts = (int64_t)a.n_frames * a.n_rays * (a.n_strat + a.n_surf);
  hipLaunchKernelGGL(sample_along_rays_kernel, dim3((unsigned)((maxPts + 255) / 256)), dim3(256), 0, st, a, o);
  return isdf_launch_status();
}

}  // namespace isdf
