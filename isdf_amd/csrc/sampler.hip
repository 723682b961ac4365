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
// ONE launch of 1024-thread workgroups (16 waves), two instantiations of one kernel body:
//   reference batch (<= 4096 drawn rays; 1000 in the BASELINE workload): a workgroup takes a chunk of 64 rays.  Wave 0
//     draws / reads the pixel and gathers depth + normal + pose (one 4-B and one 12-B random read per ray -- the only
//     reads of the keyframe buffers) while ALL 16 waves re-count the valid rays BEFORE the chunk themselves: every gather
//     of the workgroup is in flight together, one memory round trip, no inter-workgroup traffic.
//   streaming sizes: a chunk is 1024 rays (one per thread); chunk ids come from an atomic ticket (a chunk never waits
//     for one that has not started) and the chunk's offset in the ORDERED output from a decoupled look-back over
//     8-byte {epoch, flag, count} granules (agent-scope stores / polls, 64 predecessors per round).
// All 16 waves then expand the chunk's kept rays into their S samples (z values + world points): a wave takes 256
// consecutive points, lane l the points l, l+64, l+128, l+192 of them -- every store instruction is fully coalesced
// (4-B / 12-B per lane), which is where 86 % of the bytes go (432 of 500 B per ray), and in Philox mode ONE
// Philox4x32-10 call per lane feeds its four points (7.75 calls per ray instead of 28).
#include "isdf_common.h"

namespace isdf {

constexpr int SMP_NT = 1024;            // threads per workgroup (both modes)
constexpr int SMP_CHUNK_SMALL = 64;     // rays per workgroup at the reference batch size (16 workgroups for 1000 rays)
constexpr int SMP_CHUNK_STREAM = 1024;  // rays per workgroup at streaming sizes (one ray per thread)
constexpr int SMP_SMALL_SLOTS = 4;      // rays per thread in the small mode's predecessor count => up to 4096 rays

__device__ __forceinline__ uint4 ray_random(const isdf_sample_args& a, uint32_t ray, uint32_t slot) {
  return philox4x32_10(make_uint4(ray, slot, (uint32_t)a.offset, (uint32_t)(a.offset >> 32)),
                       make_uint2((uint32_t)a.seed, (uint32_t)(a.seed >> 32)));
}

// scan workspace: [0] ticket, [1] finished chunks, [2] launch epoch, [3] pad (uint32), then one 64-bit state
// word per chunk.  The last chunk to finish resets ticket/finished and bumps the epoch, so the buffer only
// has to be zero when it is allocated.
constexpr unsigned long long ST_AGG = 1ull, ST_INC = 2ull;
__device__ __forceinline__ unsigned long long st_pack(uint32_t epoch, unsigned long long flag, uint32_t v) {
  return ((unsigned long long)(epoch & 0x3fffffffu) << 34) | (flag << 32) | v;
}

// window entry b of an index list: from the kernel arguments (a select chain over <= 8 values -- a per-lane index into a
// kernel-argument array would compile to vector loads of the kernarg segment) or from the device array
__device__ __forceinline__ int pick8(const int32_t (&v)[8], int b) {
  int r = v[0];
#pragma unroll
  for (int k = 1; k < 8; ++k) r = b == k ? v[k] : r;
  return r;
}
__device__ __forceinline__ int window_frame(const isdf_sample_args& a, int b) { return a.n_inline ? pick8(a.frame_idx_inline, b) : a.frame_idx[b]; }
__device__ __forceinline__ int window_normal(const isdf_sample_args& a, int b) { return a.n_inline ? pick8(a.normal_idx_inline, b) : a.normal_idx[b]; }

// validity of drawn ray r (the pixel draw and the two gathers of sample.py:11-55) -- what decides the compaction
__device__ __forceinline__ bool ray_valid(const isdf_sample_args& a, int r) {
  const int b = r / a.n_rays;
  int h, wq;
  if (a.rng_mode == 0) { h = (int)a.draw_h[r]; wq = (int)a.draw_w[r]; }
  else { const uint4 u = ray_random(a, (uint32_t)r, 0u); h = (int)(u.x % (uint32_t)a.H); wq = (int)(u.y % (uint32_t)a.W); }
  const int64_t pix = (int64_t)h * a.W + wq;
  const float d = a.depth_batch[(int64_t)window_frame(a, b) * a.H * a.W + pix];
  bool valid = d != 0.f;
  if (a.normal_batch) {
    const float n0 = a.normal_batch[((int64_t)window_normal(a, b) * a.H * a.W + pix) * 3];
    valid = valid && !(n0 != n0);
  }
  return valid;
}

// One kernel body, two instantiations (1024 threads each):
//   CHUNK = 64   the reference batch (<= 4096 rays, latency-bound): wave 0 gathers the chunk's rays while ALL 16 waves
//                re-count the valid rays BEFORE the chunk themselves (fixed slots, straight-line code: every gather of
//                the workgroup is in flight together) -- one memory round trip, no inter-workgroup traffic, no workspace.
//   CHUNK = 1024 streaming sizes: every thread gathers one ray; chunk ids come from an atomic ticket (a chunk never
//                waits for one that has not started) and the chunk's offset from a decoupled look-back over the
//                aggregates of the chunks before it (64 predecessors per poll round).
template <int CHUNK>
__global__ __launch_bounds__(SMP_NT) void sample_rays_kernel(const isdf_sample_args a, const isdf_sample_out o,
                                                              uint32_t* __restrict__ ws, int nChunks) {
  constexpr bool SMALL = CHUNK == SMP_CHUNK_SMALL;
  constexpr int NT = SMP_NT, NW = NT / 64, GW = CHUNK / 64;   // gather waves
  __shared__ int sChunk, sBase;
  __shared__ uint32_t sEpoch;
  __shared__ int sPre[NW], sWaveCnt[NW];
  // (the pad keeps the streaming instantiation at 82 KB of LDS = ONE 16-wave workgroup per CU; seven floats per ray fit two, which
  //  measured SLOWER: 1e6 rays 0.203 vs 0.198 ms, 1e7 rays 1.78 vs 1.61 ms -- more gathers and look-back polls in flight, same HBM)
  __shared__ float sRay[CHUNK][8];   // depth, origin xyz, dirs_W xyz, pad  (compacted order within the chunk)
  __shared__ __attribute__((aligned(16))) float sPc[NW][768];   // a wave's 256 world points, staged for 16-byte stores
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  unsigned long long* state = (unsigned long long*)(ws + 4);
  const int total = a.n_frames * a.n_rays;
  const int S = a.n_strat + a.n_surf;
  int c = blockIdx.x;
  uint32_t epoch = 0;
  if (!SMALL) {
    if (tid == 0) {
      sEpoch = __hip_atomic_load(ws + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1u;
      sChunk = (int)atomicAdd(ws, 1u);      // dynamic chunk id: logical order == start order
    }
    __syncthreads();
    c = sChunk;
    epoch = sEpoch;
  }

  // ---- gather waves: the chunk's own rays, one per lane
  bool valid = false;
  int b = 0, h = 0, wq = 0, before = 0; float d = 0.f, n0 = 0.f, n1 = 0.f, n2 = 0.f;
  const float* T = a.T_WC_batch;
  float Tm[12];
#pragma unroll
  for (int k = 0; k < 12; ++k) Tm[k] = 0.f;
  if (wv < GW) {
    const int r = c * CHUNK + wv * 64 + lane;
    if (r < total) {
      b = r / a.n_rays;  // indices_b = arange(F).repeat_interleave(n_rays), sample.py:18-19
      if (a.rng_mode == 0) { h = (int)a.draw_h[r]; wq = (int)a.draw_w[r]; }
      else { const uint4 u = ray_random(a, (uint32_t)r, 0u); h = (int)(u.x % (uint32_t)a.H); wq = (int)(u.y % (uint32_t)a.W); }
      const int64_t pix = (int64_t)h * a.W + wq;
      const int fi = window_frame(a, b);
      const float* np = a.normal_batch ? a.normal_batch + ((int64_t)window_normal(a, b) * a.H * a.W + pix) * 3 : nullptr;
      d = a.depth_batch[(int64_t)fi * a.H * a.W + pix];       // both gathers in flight together ...
      if (np) { n0 = np[0]; n1 = np[1]; n2 = np[2]; }
      T += (int64_t)fi * 16;
#pragma unroll
      for (int k = 0; k < 12; ++k) Tm[k] = T[k];              // ... with the pose (L2-resident), not behind them
      valid = d != 0.f;                                       // sample.py:39-40
      if (np) valid = valid && !(n0 != n0);                   // sample.py:47-49
    }
    // ordered compaction inside the wave
    const unsigned long long m = __ballot(valid);
    before = __popcll(m & ((1ull << lane) - 1ull));
    if (lane == 0) sWaveCnt[wv] = __popcll(m);
  }
  if (SMALL) {
    // ---- all 16 waves: how many of the c*64 rays before this chunk are valid
    int n = 0;
#pragma unroll
    for (int k = 0; k < SMP_SMALL_SLOTS; ++k) {
      const int r = tid + k * NT;
      if (r < c * CHUNK) n += ray_valid(a, r) ? 1 : 0;
    }
#pragma unroll
    for (int k = 32; k >= 1; k >>= 1) n += __shfl_xor(n, k, 64);
    if (lane == 0) sPre[wv] = n;
  }
  __syncthreads();

  // ---- wave 0: the chunk's offset in the ordered output
  if (wv == 0) {
    int cnt = lane < GW ? sWaveCnt[lane] : 0;
#pragma unroll
    for (int k = 8; k >= 1; k >>= 1) cnt += __shfl_xor(cnt, k, 64);
    cnt = __shfl(cnt, 0, 64);
    int base = 0;
    if (SMALL) {
      int v = lane < NW ? sPre[lane] : 0;
#pragma unroll
      for (int k = 8; k >= 1; k >>= 1) v += __shfl_xor(v, k, 64);
      base = __shfl(v, 0, 64);
    } else if (c == 0) {
      if (lane == 0) __hip_atomic_store(state, st_pack(epoch, ST_INC, (uint32_t)cnt), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else {   // decoupled look-back over the chunks before this one
      if (lane == 0) __hip_atomic_store(state + c, st_pack(epoch, ST_AGG, (uint32_t)cnt), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      int pos = c - 1;
      while (true) {
        const int idx = pos - lane;
        unsigned long long w = 0ull;
        if (idx >= 0) w = __hip_atomic_load(state + idx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const bool ready = idx >= 0 && (uint32_t)(w >> 34) == (epoch & 0x3fffffffu) && ((w >> 32) & 3ull) != 0ull;
        const bool inc = ready && ((w >> 32) & 3ull) == ST_INC;
        const unsigned long long rm = __ballot(ready), im = __ballot(inc), vm = __ballot(idx >= 0);
        const int firstInc = im ? __ffsll((long long)im) - 1 : 63;
        const unsigned long long need = (firstInc == 63 ? ~0ull : ((2ull << firstInc) - 1ull)) & vm;
        if ((rm & need) == need) {
          int v = (need >> lane) & 1ull ? (int)(uint32_t)w : 0;
#pragma unroll
          for (int k = 32; k >= 1; k >>= 1) v += __shfl_xor(v, k, 64);
          base += v;
          if (im) break;
          pos -= 64;          // 64 aggregates, no inclusive prefix among them: keep looking back
        } else {
          __builtin_amdgcn_s_sleep(2);
        }
      }
      if (lane == 0) __hip_atomic_store(state + c, st_pack(epoch, ST_INC, (uint32_t)(base + cnt)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (lane == 0) sBase = base;
    if (c == nChunks - 1 && lane == 0) *o.n_valid = base + cnt;
  }
  __syncthreads();

  // ---- gather waves: per-ray outputs at the compacted position
  int wbase = 0, cnt_ = 0;
#pragma unroll
  for (int k = 0; k < GW; ++k) { const int n = sWaveCnt[k]; if (k < wv) wbase += n; cnt_ += n; }
  const int base = sBase;
  if (wv < GW && valid) {
    const int jl = wbase + before;
    const int64_t q = (int64_t)base + jl;
    o.indices_b[q] = b; o.indices_h[q] = h; o.indices_w[q] = wq;
    o.depth_sample[q] = d;
    if (o.norm_sample) { o.norm_sample[q * 3] = n0; o.norm_sample[q * 3 + 1] = n1; o.norm_sample[q * 3 + 2] = n2; }
    // ray_dirs_C, transform.py:13-33 ('z' depth)
    const float dx = ((float)wq - a.cx) / a.fx, dy = ((float)h - a.cy) / a.fy, dz = 1.f;
    o.dirs_C_sample[q * 3] = dx; o.dirs_C_sample[q * 3 + 1] = dy; o.dirs_C_sample[q * 3 + 2] = dz;
    if (o.T_WC_sample) {
      float4* dst = (float4*)(o.T_WC_sample + q * 16);
      dst[0] = make_float4(Tm[0], Tm[1], Tm[2], Tm[3]); dst[1] = make_float4(Tm[4], Tm[5], Tm[6], Tm[7]);
      dst[2] = make_float4(Tm[8], Tm[9], Tm[10], Tm[11]); dst[3] = make_float4(T[12], T[13], T[14], T[15]);
    }
    // origin_dirs_W, transform.py:36-41: (R * d).sum(-1), no fused multiply-add
    float* sr = sRay[jl];
    sr[0] = d;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      const float s = __fadd_rn(__fadd_rn(__fmul_rn(Tm[i * 4], dx), __fmul_rn(Tm[i * 4 + 1], dy)), __fmul_rn(Tm[i * 4 + 2], dz));
      o.dirs_W_sample[q * 3 + i] = s;
      sr[1 + i] = Tm[i * 4 + 3];
      sr[4 + i] = s;
    }
  }
  __syncthreads();

  // ---- along-ray samples of the chunk's valid rays: point p of the chunk -> (ray j, sample s); consecutive
  // threads write consecutive points (sample.py:131-178)
  const int npts = cnt_ * S;
  struct __attribute__((packed, aligned(4))) f4u { float x, y, z, w; };   // 16-byte store at 4-byte alignment
  for (int G = wv; G * 256 < npts; G += NW) {
    // full groups write their 256 x 3 floats of `pc` as three 16-byte stores per lane (1 KB per wave instruction) through a
    // per-wave LDS transpose instead of twelve 4-byte stores at a 12-byte lane stride (a third of every line per instruction)
    const bool fullGroup = G * 256 + 256 <= npts;
    // in-kernel draws: one Philox call per lane and 256-point group, word m for the lane's m-th point; counter
    // (chunk, 1 + group*64 + lane) never collides with the pixel draws' (ray, 0)
    uint4 rnd = make_uint4(0u, 0u, 0u, 0u);
    if (a.rng_mode != 0) rnd = ray_random(a, (uint32_t)c, 1u + (uint32_t)(G * 64 + lane));
#pragma unroll
   for (int m = 0; m < 4; ++m) {
    const int p = G * 256 + m * 64 + lane;
    if (p >= npts) continue;
    const uint32_t word = m == 0 ? rnd.x : (m == 1 ? rnd.y : (m == 2 ? rnd.z : rnd.w));
    const int j = p / S, s = p - j * S;
    const int64_t r = (int64_t)base + j;             // compacted ray index: the draws are indexed by it
    const float* sr = sRay[j];
    const float depth = sr[0];
    const float maxd = __fadd_rn(depth, a.dist_behind_surf);   // trainer.py:741
    float z;
    if (s < a.n_surf) {
      if (s == 0) z = depth;                                     // sample.py:158
      else {
        float off;
        if (a.rng_mode == 0) off = a.draw_n[r * (a.n_surf - 1) + (s - 1)];
        else {  // Box-Muller, sigma 0.1 (sample.py:160-162), from the two 16-bit halves of the point's word: radius
          // resolved to 2^-16 (tail to 4.7 sigma; the offset is clamped to [min_depth, depth + dist_behind_surf] anyway)
          const float u1 = ((float)(word >> 16) + 0.5f) * (1.f / 65536.f), u2 = (float)(word & 0xffffu) * (1.f / 65536.f);
          off = 0.1f * sqrtf(-2.f * __logf(u1)) * __cosf(6.2831853f * u2);
        }
        z = fminf(fmaxf(__fadd_rn(depth, off), a.min_depth), maxd);  // clamp, sample.py:167-171
      }
    } else {
      const int k = s - a.n_surf, nb = a.n_strat;
      float U;
      if (a.rng_mode == 0) U = a.draw_u[r * nb + k];
      else U = u01(word);
      // torch.linspace(0, 1, nb+1)[k] in fp32 (symmetric evaluation), sample.py:96-98
      const float step = 1.f / (float)nb;
      const float lin = k < (nb + 1) / 2 ? __fmul_rn(step, (float)k) : __fadd_rn(1.f, -__fmul_rn(step, (float)(nb - k)));
      const float range = __fadd_rn(maxd, -a.min_depth);
      const float lim = __fadd_rn(__fmul_rn(lin, range), a.min_depth);
      const float blen = range / (float)nb;
      z = __fadd_rn(lim, __fmul_rn(U, blen));                       // sample.py:123-126
    }
    const int64_t n = r * S + s;
    o.z_vals[n] = z;      // (four 4-byte stores per lane; one 16-byte store through an LDS stage measured SLOWER: r04_sampler_z_stores.txt)
#pragma unroll
    for (int i = 0; i < 3; ++i) {  // pc = origins + dirs_W * z, sample.py:176
      const float v = __fadd_rn(sr[1 + i], __fmul_rn(sr[4 + i], z));
      if (fullGroup) sPc[wv][(m * 64 + lane) * 3 + i] = v;
      else o.pc[n * 3 + i] = v;
    }
   }
    if (fullGroup) {   // wave-uniform
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      float* dst = o.pc + ((int64_t)base * S + (int64_t)G * 256) * 3;
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        const float4 v = *(const float4*)&sPc[wv][(k * 64 + lane) * 4];
        *(f4u*)(dst + (k * 64 + lane) * 4) = f4u{v.x, v.y, v.z, v.w};
      }
      __builtin_amdgcn_wave_barrier();   // the next group of this wave overwrites sPc
    }
  }

  if (SMALL) return;
  // ---- the last chunk to finish re-arms the workspace for the next launch
  __syncthreads();
  if (tid == 0) {
    const uint32_t done = atomicAdd(ws + 1, 1u);
    if ((int)done == nChunks - 1) {
      __hip_atomic_store(ws, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(ws + 1, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(ws + 2, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}

constexpr int SMP_SMALL_MAX_RAYS = SMP_NT * SMP_SMALL_SLOTS;   // 4096

int64_t sample_scan_bytes(int64_t max_rays) { return 16 + 8 * ((max_rays + SMP_CHUNK_STREAM - 1) / SMP_CHUNK_STREAM); }

int launch_sample_rays(const isdf_sample_args& a, const isdf_sample_out& o, void* scan_ws, hipStream_t st) {
  const int total = a.n_frames * a.n_rays;
  if (total <= SMP_SMALL_MAX_RAYS) {
    const int nChunks = (total + SMP_CHUNK_SMALL - 1) / SMP_CHUNK_SMALL;
    hipLaunchKernelGGL(sample_rays_kernel<SMP_CHUNK_SMALL>, dim3((unsigned)nChunks), dim3(SMP_NT), 0, st, a, o, (uint32_t*)scan_ws, nChunks);
  } else {
    const int nChunks = (total + SMP_CHUNK_STREAM - 1) / SMP_CHUNK_STREAM;
    hipLaunchKernelGGL(sample_rays_kernel<SMP_CHUNK_STREAM>, dim3((unsigned)nChunks), dim3(SMP_NT), 0, st, a, o, (uint32_t*)scan_ws, nChunks);
  }
  return isdf_launch_status();
}

}  // namespace isdf
