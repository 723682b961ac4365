// Fused per-tile chain kernel (K2): positional encoding -> SDF MLP forward ->
// input gradient (first reverse sweep) -> loss + loss adjoints -> adjoint of
// the first reverse sweep (runs upward like a JVP) -> ordinary reverse sweep
// with the injected sigma'' term.  Replaces, for one tile of TILE_PTS points,
//   embedding.PostionalEncoding.forward   isdf/modules/embedding.py:95-111
//   SDFMap.forward                        isdf/modules/fc_map.py:94-111
//   fc_map.gradient (autograd.grad)       isdf/modules/fc_map.py:12-22
//   loss.bounds_ray / sdf_loss / tot_loss isdf/modules/loss.py:13-22,122-205
//   eikonal + normal terms                isdf/modules/trainer.py:814-830
//   the activation side of total_loss.backward()   trainer.py:981
// The weight-gradient contractions over points are done by dw.hip from the
// bf16 operand tiles this kernel spills.
//
// Mapping to CDNA4: one workgroup = 4 waves = 64 points, two workgroups per CU
// (64 KB LDS each) so one does MFMA while the other is in an elementwise
// epilogue.  Every GEMM is C[feature][point] = W[feature][k] * X[k][point] on
// v_mfma_f32_32x32x16_{f16,bf16}: A = packed weights streamed straight from L2
// in fragment order (1 KB contiguous per wave-load, each weight is used by
// exactly one wave of the workgroup so LDS staging would add nothing),
// B = the activation tile in LDS ([point][k], 16-B XOR swizzle, ds_read_b128).
// In the C layout a lane owns one point and 4 consecutive features per
// register quad, so epilogues write 8-byte packed pieces back to the LDS tile.
#include "isdf_common.h"
#include "chain_params.h"

namespace isdf {

constexpr float kHalfPi = 1.5707963267948966f;
constexpr float kBeta = 100.f;

template <int HD, int EP>
struct Tile {
  static constexpr int BM = TILE_PTS;
  static constexpr int NW = CHAIN_NW;
  static constexpr int FB = HD / (NW * 32);   // 32-row feature blocks per wave
  static constexpr int PB = BM / 32;          // 32-point blocks
  static constexpr int R2 = (EP > HD ? EP : HD);
  static constexpr int XK = HD + R2;          // elements per LDS row
  static constexpr int ROWB = XK * 2;
  static constexpr int XBYTES = BM * ROWB;
  // small fp32 arrays after the X tile
  static constexpr int OFF_XS = XBYTES;                   // [BM][4] x' (scaled/transformed point)
  static constexpr int OFF_PART = OFF_XS + BM * 16;       // [NW][BM][4] partial sums (raw / g)
  static constexpr int OFF_GB = OFF_PART + NW * BM * 16;   // [BM][4] gbar in x' space, [3] = sbar*so
  static constexpr int OFF_RED = OFF_GB + BM * 16;        // [BM/64][8] per-wave loss sums
  static constexpr int LDS_BYTES = OFF_RED + (BM / 64) * 32 + 32;
};

// Workgroup barrier that only waits for this wave's LDS traffic.  The global
// spill tiles are thread-private (the lane that stores a piece is the lane that
// re-reads it), so global stores/loads may stay in flight across the barrier;
// __syncthreads() would drain them (s_waitcnt vmcnt(0)) at every layer.
__device__ __forceinline__ void lds_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
}

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ uint4 nt_load16(const uint4* p) {
  const u32x4 v = __builtin_nontemporal_load((const u32x4*)p);
  return make_uint4(v[0], v[1], v[2], v[3]);
}
__device__ __forceinline__ void nt_store16(uint4 x, uint4* p) {
  u32x4 v; v[0] = x.x; v[1] = x.y; v[2] = x.z; v[3] = x.w;
  __builtin_nontemporal_store(v, (u32x4*)p);
}

__device__ __forceinline__ int swz(int row, int colbytes) { return colbytes ^ ((row & 15) << 4); }

// C[FB*32 feats][PB*32 pts] += Wpacked[feat][k] * X[pt][k]  over KSTEPS*16 k.
// Measured: a workgroup's time is a serial latency chain, and a staged
// weight loop pays one dependent L2/MALL round trip per stage (8 per K=256 unit).  So the
// wave requests its WHOLE weight slice for a chunk of CK k-steps up front (64 VGPRs:
// one round trip per chunk; K=256 is one chunk for 256-wide nets), one scheduling
// barrier keeps the load block ahead of the MFMAs, and the k-steps are fully unrolled.
// `lateHook` (the epilogue's spill prefetch) is issued right after the last MFMA: issuing it
// mid-GEMM overlapped more latency but pushed the train kernel 130 VGPRs over its budget.
template <bool F16, int KSTEPS, int FBN, int PBN, int ROWB, typename Hook>
__device__ __forceinline__ void gemm(f32x16 (&acc)[FBN][PBN], const uint4* __restrict__ wp,
                                     int rbStride, const char* xl, int colByteBase, int lane, Hook&& lateHook) {
  constexpr int CK = 8 / FBN;                  // k-steps per chunk: CK*FBN uint4 = 32 VGPRs
  static_assert(KSTEPS % CK == 0, "K must be a multiple of the chunk");
  constexpr int NCH = KSTEPS / CK;
  const int j = lane & 31, hi = lane >> 5;
  const int sw = (j & 15) << 4;
  const uint4* wl = wp + lane;
#pragma unroll 1
  for (int ch = 0; ch < NCH; ++ch) {
    uint4 wb[CK][FBN];
#pragma unroll
    for (int s = 0; s < CK; ++s)
#pragma unroll
      for (int fb = 0; fb < FBN; ++fb) wb[s][fb] = wl[fb * rbStride + (ch * CK + s) * 64];
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int s = 0; s < CK; ++s) {
      typename Op<F16>::v8 b[PBN];
#pragma unroll
      for (int pb = 0; pb < PBN; ++pb) {
        const int cb = (colByteBase + (ch * CK + s) * 32 + hi * 16) ^ sw;
        b[pb] = __builtin_bit_cast(typename Op<F16>::v8, *(const uint4*)(xl + (pb * 32 + j) * ROWB + cb));
      }
#pragma unroll
      for (int fb = 0; fb < FBN; ++fb)
#pragma unroll
        for (int pb = 0; pb < PBN; ++pb)
          acc[fb][pb] = Op<F16>::mfma(__builtin_bit_cast(typename Op<F16>::v8, wb[s][fb]), b[pb], acc[fb][pb]);
    }
  }
  lateHook();   // after the last MFMA: the weight registers are dead, so the prefetch adds no pressure
}

template <int FBN, int PBN> __device__ __forceinline__ void zero_acc(f32x16 (&acc)[FBN][PBN]) {
#pragma unroll
  for (int fb = 0; fb < FBN; ++fb)
#pragma unroll
    for (int pb = 0; pb < PBN; ++pb)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[fb][pb][r] = 0.f;
}

// frag16 spill order: element offset inside a tile for (wave, fb, pb, qp, lane)
template <int FBN, int PBN>
__device__ __forceinline__ int frag16_off(int w, int fb, int pb, int qp, int lane) {
  return ((((w * FBN + fb) * PBN + pb) * 2 + qp) * 64 + lane) * 8;
}

// sum v over the 32 lanes that share `hi`; lane j==0 of each half stores it.  Each
// (layer, feature) has exactly ONE owner half-wave per workgroup, so the
// per-workgroup partial needs no atomics (same-address global atomics from 422
// workgroups serialise at ~12 ns each and dominated the first version).
template <int CTRL> __device__ __forceinline__ float dpp_mov(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, true));
}
__device__ __forceinline__ float half_wave_sum(float v) {
  // butterfly over 32 lanes on the VALU: quad_perm xor1, xor2, row_half_mirror (8), row_mirror (16 lanes),
  // then one swizzle for the 16<->16 exchange (__shfl_xor = 5 dependent ds_bpermute round trips:
  // the reverse-sweep epilogues took ~15 k cycles with it, ~8 k with this)
  v += dpp_mov<0xB1>(v);      // quad_perm [1,0,3,2]
  v += dpp_mov<0x4E>(v);      // quad_perm [2,3,0,1]
  v += dpp_mov<0x141>(v);     // row_half_mirror: quads 0<->1, 2<->3 (values are quad-uniform)
  v += dpp_mov<0x140>(v);     // row_mirror: lower 8 <-> upper 8 of each 16-lane row
  v += __int_as_float(__builtin_amdgcn_ds_swizzle(__float_as_int(v), 0x401F));   // xor 16 within 32 lanes
  return v;
}
__device__ __forceinline__ void half_wave_store(float v, float* dst, int lane) {
  v = half_wave_sum(v);
  if ((lane & 31) == 0) *dst = v;
}

// Softplus(beta=100, threshold=20) on the hardware's base-2 transcendental units:
//   a = max(z, ln2/beta * log2(1 + 2^(beta*log2e*z)))
// (softplus(z) > z always, and torch's threshold branch returns z where the
// two differ by < 2e-11, so max() reproduces it without a select).
constexpr float kC1 = kBeta * 1.4426950408889634f;   // beta * log2(e)
constexpr float kC2 = 0.6931471805599453f / kBeta;   // ln2 / beta
__device__ __forceinline__ float softplus_f(float z) {
  const float t = __builtin_amdgcn_exp2f(fminf(kC1 * z, 30.f));
  return fmaxf(z, kC2 * __builtin_amdgcn_logf(1.f + t));
}
__device__ __forceinline__ float softplus_s1(float z, float& s1) {   // also sigma'(z) = t/(1+t)
  const float t = __builtin_amdgcn_exp2f(fminf(kC1 * z, 30.f));
  const float u = 1.f + t;
  s1 = t * __builtin_amdgcn_rcpf(u);
  return fmaxf(z, kC2 * __builtin_amdgcn_logf(u));
}
// sigma'(z) recovered from the stored activation: 1 - exp(-beta a)
__device__ __forceinline__ float s1_from_a(float a) { return 1.f - __builtin_amdgcn_exp2f(-kC1 * a); }

template <int HD, int EP, bool F16, int MODE>
__global__ __launch_bounds__(CHAIN_NW * 64, (HD <= 256 ? 4 : 2)) void chain_kernel(const ChainParams p) {
  typedef Tile<HD, EP> T;
  static_assert(HD == EP, "tile kernels assume padded embedding width == hidden width");
  constexpr int BM = T::BM, FB = T::FB, PB = T::PB, ROWB = T::ROWB;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* X = smem;
  float* xs = (float*)(smem + T::OFF_XS);
  float* part = (float*)(smem + T::OFF_PART);
  float* gbs = (float*)(smem + T::OFF_GB);
  float* red = (float*)(smem + T::OFF_RED);

  const NetLayout& L = p.lay;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int j = lane & 31, hi = lane >> 5;
  const int64_t P = p.n_valid ? (int64_t)(*p.n_valid) * p.S : p.n_points_host;
  const int64_t n0 = (int64_t)blockIdx.x * BM;
  if (n0 >= P) return;
  const int nf = L.n_freqs;
  const float so = L.scale_output;
  if (p.dbg_stagger && (blockIdx.x & 1)) {   // experiment: de-phase odd tiles (L2-bound vs HBM-bound sweeps)
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    while (__builtin_amdgcn_s_memtime() - t0 < (unsigned long long)p.dbg_stagger * 1000ull) __builtin_amdgcn_s_sleep(64);
  }
  int tsn = 0;
  auto TS = [&]() {   // debug timeline: wave 0 of workgroup 100 stamps phase boundaries
    if (p.dbg_times && blockIdx.x == 100 && tid == 0) p.dbg_times[tsn] = __builtin_amdgcn_s_memtime();
    ++tsn;
  };
  TS();

  const uint16_t* setFwdA = p.shadow + L.setFwdA;
  const uint16_t* setFwdB = p.shadow + L.setFwdB;
  const uint16_t* setBwdA = p.shadow + L.setBwdA;
  const uint16_t* setBwdB = p.shadow + L.setBwdB;
  uint16_t* spillTile = p.spill + (int64_t)(p.dbg_alias ? (blockIdx.x % p.dbg_alias) : blockIdx.x) * BM * HD;
  float* vecTile = MODE == 2 ? p.vec_part + (int64_t)blockIdx.x * p.vecStride : nullptr;
  (void)setFwdB; (void)setBwdB; (void)spillTile;

  // ------------------------------------------------------------------ PE stage
  // thread (pt, part): embedding.py:95-111.  Region 2 of X (cols HD..) gets the
  // embedding in the forward operand type; region 1 a bf16 copy staged for the
  // spill (dW operand A_0).
  {
    const int pt = tid & (BM - 1), prt = tid / BM;
    constexpr int NPART = (T::NW * 64) / BM;
    const int64_t n = n0 + pt;
    float x0 = 0.f, x1 = 0.f, x2 = 0.f;
    if (n < P) { x0 = p.pts[n * 3]; x1 = p.pts[n * 3 + 1]; x2 = p.pts[n * 3 + 2]; }
    // transform_3D_grid (transform.py:287-304) then * scale (embedding.py:12-22)
    float y0 = (L.T[0] * x0 + L.T[1] * x1 + L.T[2] * x2 + L.T[3]) * L.scale_input;
    float y1 = (L.T[4] * x0 + L.T[5] * x1 + L.T[6] * x2 + L.T[7]) * L.scale_input;
    float y2 = (L.T[8] * x0 + L.T[9] * x1 + L.T[10] * x2 + L.T[11]) * L.scale_input;
    typedef typename Op<F16>::e opT;
    char* row = X + pt * ROWB;
    auto put = [&](int feat, float v) {
      *(opT*)(row + swz(pt, (HD + feat) * 2)) = (opT)v;
      if (MODE == 2) *(__bf16*)(row + swz(pt, feat * 2)) = (__bf16)v;
    };
    if (prt == 0) {
      xs[pt * 4] = y0; xs[pt * 4 + 1] = y1; xs[pt * 4 + 2] = y2;
      put(0, y0); put(1, y1); put(2, y2);
      for (int f = L.E; f < EP; ++f) put(f, 0.f);
    }
    for (int d = prt; d < N_DIRS; d += NPART) {
      const float proj = y0 * kDirs[0][d] + y1 * kDirs[1][d] + y2 * kDirs[2][d];
      float fr = 1.f;
      for (int f = 0; f < nf; ++f) {
        const float xb = proj * fr;
        put(3 + d * nf + f, __sinf(xb));
        put(3 + N_DIRS * nf + d * nf + f, __sinf(xb + kHalfPi));
        fr *= 2.f;
      }
    }
  }
  lds_barrier();
  auto spill_region = [&](int colElemBase, uint16_t* dstTile) {
    // copy a bf16 [BM][HD] region of X to global in frag16 order (16 B per lane)
#pragma unroll
    for (int fb = 0; fb < FB; ++fb)
#pragma unroll
      for (int pb = 0; pb < PB; ++pb)
#pragma unroll
        for (int qp = 0; qp < 2; ++qp) {
          const int rowi = pb * 32 + j;
          const int f0 = w * (FB * 32) + fb * 32 + 16 * qp + 4 * hi;
          uint2 lo = *(const uint2*)(X + rowi * ROWB + swz(rowi, (colElemBase + f0) * 2));
          uint2 hi2 = *(const uint2*)(X + rowi * ROWB + swz(rowi, (colElemBase + f0 + 8) * 2));
          nt_store16(make_uint4(lo.x, lo.y, hi2.x, hi2.y), (uint4*)(dstTile + frag16_off<FB, PB>(w, fb, pb, qp, lane)));
        }
  };
  if (MODE == 2) {
    spill_region(0, spillTile + p.sp.A[0]);
    lds_barrier();
  }

  // ------------------------------------------------------------------ forward
  f32x16 acc[FB][PB];
  const int rbW = w * FB;  // first 32-row block of this wave
  auto wptr = [&](const uint16_t* set, int64_t matOff, int kp) {
    return (const uint4*)(set + matOff) + (int64_t)rbW * (kp / 16) * 64;
  };
  // iterate the wave's accumulator as (fb, pb, qp) blocks of 8 values:
  // values v[0..3] -> features f0..f0+3, v[4..7] -> f0+8..f0+11, point row = pb*32+j
  auto for_blocks2 = [&](auto&& fn, auto&& tail) {
#pragma unroll
    for (int fb = 0; fb < FB; ++fb)
#pragma unroll
      for (int qp = 0; qp < 2; ++qp) {
        const int f0 = w * (FB * 32) + fb * 32 + 16 * qp + 4 * hi;
#pragma unroll
        for (int pb = 0; pb < PB; ++pb) fn(fb, pb, qp, f0, pb * 32 + j);
        tail(f0);
      }
  };
  auto for_blocks = [&](auto&& fn) { for_blocks2(fn, [](int) {}); };
  // A spilled tile is re-read by the same lanes that wrote it; the reads are
  // issued BEFORE the layer's GEMM (prefetch) and consumed in its epilogue.
  struct Pre { uint4 v[FB][2][PB]; };
  const int laneChunk = frag16_off<FB, PB>(w, 0, 0, 0, lane) / 8;   // uint4 index of this lane's first piece
  auto chunkOf = [](int fb, int pb, int qp) { return ((fb * PB + pb) * 2 + qp) * 64; };   // compile-time
  auto prefetch = [&](int64_t tensorOff, Pre& pr) {
    const uint4* base = (const uint4*)(spillTile + tensorOff) + laneChunk;
#pragma unroll
    for (int fb = 0; fb < FB; ++fb)
#pragma unroll
      for (int qp = 0; qp < 2; ++qp)
#pragma unroll
        for (int pb = 0; pb < PB; ++pb) pr.v[fb][qp][pb] = nt_load16(base + chunkOf(fb, pb, qp));
  };
  auto load_tile8 = [&](const Pre& pr, int fb, int pb, int qp, float (&o)[8]) {
    const uint4 u = pr.v[fb][qp][pb];
    float a[4], b[4];
    unpack4_bf16(make_uint2(u.x, u.y), a); unpack4_bf16(make_uint2(u.z, u.w), b);
#pragma unroll
    for (int e = 0; e < 4; ++e) { o[e] = a[e]; o[4 + e] = b[e]; }
  };
  auto store_tile8 = [&](int64_t tensorOff, int fb, int pb, int qp, const float (&v)[8]) {
    const uint2 a = pack4<false>(v[0], v[1], v[2], v[3]), b = pack4<false>(v[4], v[5], v[6], v[7]);
    // streamed once: non-temporal so the spill stream does not evict the L2-resident weight copies
    nt_store16(make_uint4(a.x, a.y, b.x, b.y), (uint4*)(spillTile + tensorOff) + laneChunk + chunkOf(fb, pb, qp));
  };
  auto put_x = [&](bool f16, int row, int f0, const float (&v)[8], int colElemBase) {
    uint2 a, b;
    if (f16) { a = pack4<true>(v[0], v[1], v[2], v[3]); b = pack4<true>(v[4], v[5], v[6], v[7]); }
    else { a = pack4<false>(v[0], v[1], v[2], v[3]); b = pack4<false>(v[4], v[5], v[6], v[7]); }
    *(uint2*)(X + row * ROWB + swz(row, (colElemBase + f0) * 2)) = a;
    *(uint2*)(X + row * ROWB + swz(row, (colElemBase + f0 + 8) * 2)) = b;
  };

  float rawp[PB];
#pragma unroll
  for (int pb = 0; pb < PB; ++pb) rawp[pb] = 0.f;

  for (int li = 0; li < L.L; ++li) {
    zero_acc(acc);
    if (li == 0)
      gemm<F16, EP / 16, FB, PB, ROWB>(acc, wptr(setFwdA, L.fwdMat[0], EP), (EP / 16) * 64, X, HD * 2, lane, [] {});
    else if (li == L.cat)
      gemm<F16, (HD + EP) / 16, FB, PB, ROWB>(acc, wptr(setFwdA, L.fwdMat[li], HD + EP), ((HD + EP) / 16) * 64, X, 0, lane, [] {});
    else
      gemm<F16, HD / 16, FB, PB, ROWB>(acc, wptr(setFwdA, L.fwdMat[li], HD), (HD / 16) * 64, X, 0, lane, [] {});
    TS();
    lds_barrier();  // all waves finished reading region 1
    TS();
    const float* bias = p.params + L.offB[li];
    const bool last = li == L.L - 1;
    const float* wout = p.params + L.offWout;
    if (!last) {
      float bv[8];
      for_blocks2([&](int fb, int pb, int qp, int f0, int row) {
        if (pb == 0) {
          const float4 b0 = *(const float4*)(bias + f0), b1 = *(const float4*)(bias + f0 + 8);
          bv[0] = b0.x; bv[1] = b0.y; bv[2] = b0.z; bv[3] = b0.w; bv[4] = b1.x; bv[5] = b1.y; bv[6] = b1.z; bv[7] = b1.w;
        }
        float a[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) a[e] = softplus_f(acc[fb][pb][8 * qp + e] + bv[e]);
        if (MODE >= 1) store_tile8(p.sp.A[li + 1], fb, pb, qp, a);
        put_x(F16, row, f0, a, 0);
      }, [](int) {});
    } else {
      float bv[8], wv[8];
      for_blocks2([&](int fb, int pb, int qp, int f0, int row) {
        if (pb == 0) {
          const float4 b0 = *(const float4*)(bias + f0), b1 = *(const float4*)(bias + f0 + 8);
          const float4 w0 = *(const float4*)(wout + f0), w1 = *(const float4*)(wout + f0 + 8);
          bv[0] = b0.x; bv[1] = b0.y; bv[2] = b0.z; bv[3] = b0.w; bv[4] = b1.x; bv[5] = b1.y; bv[6] = b1.z; bv[7] = b1.w;
          wv[0] = w0.x; wv[1] = w0.y; wv[2] = w0.z; wv[3] = w0.w; wv[4] = w1.x; wv[5] = w1.y; wv[6] = w1.z; wv[7] = w1.w;
        }
        float a[8], pl[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          float s1;
          a[e] = softplus_s1(acc[fb][pb][8 * qp + e] + bv[e], s1);
          rawp[pb] += wv[e] * a[e];
          pl[e] = so * wv[e] * s1;   // p_L = q_L * sigma'(z_L), q_L = so * w_out
        }
        if (MODE >= 1) {
          store_tile8(p.sp.A[li + 1], fb, pb, qp, a);
          put_x(F16, row, f0, pl, 0);
          if (MODE == 2) store_tile8(p.sp.P[li], fb, pb, qp, pl);
        }
      }, [](int) {});
    }
    if (last) {
#pragma unroll
      for (int pb = 0; pb < PB; ++pb) {
        const float v = rawp[pb] + __shfl_xor(rawp[pb], 32, 64);   // add the two feature halves
        if (hi == 0) part[(w * BM + pb * 32 + j) * 4] = v;
      }
    }
    TS();
    lds_barrier();
    TS();
  }
  // sdf = (raw + noise) * so   (fc_map.py:104-109)
  float my_sdf = 0.f;
  if (tid < BM) {
    float r = p.params[L.offBout];
#pragma unroll
    for (int k = 0; k < T::NW; ++k) r += part[(k * BM + tid) * 4];
    const int64_t n = n0 + tid;
    if (p.noise) { if (n < P) r += p.noise[n]; }
    else if (p.noise_std != 0.f) {   // Box-Muller on Philox4x32-10 keyed by (seed, offset, point)
      const uint4 u = philox4x32_10(make_uint4((uint32_t)n, (uint32_t)(n >> 32), (uint32_t)p.noise_off, (uint32_t)(p.noise_off >> 32)),
                                    make_uint2((uint32_t)p.noise_seed, (uint32_t)(p.noise_seed >> 32) ^ 0x5eedu));
      const float u1 = fmaxf(u01(u.x), 1e-7f), u2 = u01(u.y);
      r += p.noise_std * sqrtf(-2.f * __logf(u1)) * __cosf(6.2831853f * u2);
    }
    my_sdf = r * so;
    if (p.sdf && n < P) p.sdf[n] = my_sdf;
  }
  if (MODE == 0) return;

  // ------------------------------------------------------------------ first reverse sweep
  for (int li = L.L - 1; li >= 1; --li) {
    Pre preA;
    zero_acc(acc);
    gemm<F16, HD / 16, FB, PB, ROWB>(acc, wptr(setBwdA, L.bwdMat[li], HD), (HD / 16) * 64, X, 0, lane,
                                     [&] { prefetch(p.sp.A[li], preA); });
    TS();
    lds_barrier();
    TS();
    const bool toR2 = (li - 1 == L.cat);
    for_blocks([&](int fb, int pb, int qp, int f0, int row) {
      float a[8], pv[8];
      load_tile8(preA, fb, pb, qp, a);
#pragma unroll
      for (int e = 0; e < 8; ++e) pv[e] = acc[fb][pb][8 * qp + e] * s1_from_a(a[e]);
      put_x(F16, row, f0, pv, 0);
      if (toR2) put_x(F16, row, f0, pv, HD);
      if (MODE == 2) store_tile8(p.sp.P[li - 1], fb, pb, qp, pv);
    });
    TS();
    lds_barrier();
    TS();
  }
  // Eg = [W_in^T | W_cat[:,HD:]^T] [p_0 ; p_cat]   (rows = embedding features)
  zero_acc(acc);
  gemm<F16, (2 * HD) / 16, FB, PB, ROWB>(acc, wptr(setBwdA, L.bwdG, 2 * HD), ((2 * HD) / 16) * 64, X, 0, lane, [] {});
  // g_x' = J_pe^T Eg : contract the wave's 64 embedding rows against d emb / d x'
  {
    float g0[PB], g1[PB], g2[PB];
#pragma unroll
    for (int pb = 0; pb < PB; ++pb) { g0[pb] = g1[pb] = g2[pb] = 0.f; }
    const int half = N_DIRS * nf;
#pragma unroll
    for (int fb = 0; fb < FB; ++fb)
#pragma unroll
      for (int pb = 0; pb < PB; ++pb) {
        const int row = pb * 32 + j;
        const float y0 = xs[row * 4], y1 = xs[row * 4 + 1], y2 = xs[row * 4 + 2];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int f = w * (FB * 32) + fb * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
          const float v = acc[fb][pb][r];
          if (f < 3) {
            g0[pb] += f == 0 ? v : 0.f; g1[pb] += f == 1 ? v : 0.f; g2[pb] += f == 2 ? v : 0.f;
          } else if (f < L.E) {
            const int t = f - 3;
            const bool isCos = t >= half;
            const int tt = isCos ? t - half : t;
            const int d = tt / nf, fq = tt - d * nf;
            const float fr = (float)(1 << fq);
            const float dx = kDirs[0][d], dy = kDirs[1][d], dz = kDirs[2][d];
            const float xb = (y0 * dx + y1 * dy + y2 * dz) * fr;
            const float c = __cosf(isCos ? xb + kHalfPi : xb) * fr * v;
            g0[pb] += c * dx; g1[pb] += c * dy; g2[pb] += c * dz;
          }
        }
      }
    lds_barrier();  // part[] reuse; X reads of the G gemm complete
#pragma unroll
    for (int pb = 0; pb < PB; ++pb) {
      const float a0 = g0[pb] + __shfl_xor(g0[pb], 32, 64), a1 = g1[pb] + __shfl_xor(g1[pb], 32, 64),
                  a2 = g2[pb] + __shfl_xor(g2[pb], 32, 64);
      if (hi == 0) {
        float* d = part + (w * BM + pb * 32 + j) * 4;
        d[0] = a0; d[1] = a1; d[2] = a2;
      }
    }
  }
  lds_barrier();

  // ------------------------------------------------------------------ loss + adjoints (one thread per point)
  float lsum[4] = {0.f, 0.f, 0.f, 0.f};
  if (tid < BM) {
    const int64_t n = n0 + tid;
    float e0 = 0.f, e1 = 0.f, e2 = 0.f;
#pragma unroll
    for (int k = 0; k < T::NW; ++k) {
      const float* s = part + (k * BM + tid) * 4;
      e0 += s[0]; e1 += s[1]; e2 += s[2];
    }
    // g_x = scale_input * R^T g_x'
    const float si = L.scale_input;
    const float gx = si * (L.T[0] * e0 + L.T[4] * e1 + L.T[8] * e2);
    const float gy = si * (L.T[1] * e0 + L.T[5] * e1 + L.T[9] * e2);
    const float gz = si * (L.T[2] * e0 + L.T[6] * e1 + L.T[10] * e2);
    if (p.sdf_grad && n < P) { p.sdf_grad[n * 3] = gx; p.sdf_grad[n * 3 + 1] = gy; p.sdf_grad[n * 3 + 2] = gz; }
    float sbar = 0.f, bx = 0.f, by = 0.f, bz = 0.f;
    if (MODE == 2 && n < P) {
      const isdf_loss_cfg& lc = p.loss;
      const int64_t ray = n / p.S;
      const int s = (int)(n - ray * p.S);
      float bnd, tx, ty, tz;  // bound and target gradient direction
      if (lc.bounds_method == 0) {  // loss.py:13-22
        const float cx = p.dirsC[ray * 3], cy = p.dirsC[ray * 3 + 1], cz = p.dirsC[ray * 3 + 2];
        bnd = sqrtf(cx * cx + cy * cy + cz * cz) * (p.depth[ray] - p.z_vals[n]);
        tx = -p.dirsW[ray * 3]; ty = -p.dirsW[ray * 3 + 1]; tz = -p.dirsW[ray * 3 + 2];
      } else {
        bnd = p.pc_bounds[n];
        tx = p.pc_grad_vec[n * 3]; ty = p.pc_grad_vec[n * 3 + 1]; tz = p.pc_grad_vec[n * 3 + 2];
      }
      if (p.normals && (s == 0 || tx != tx)) {  // surface sample, or NaN target (trainer.py:823-824)
        tx = p.normals[ray * 3]; ty = p.normals[ray * 3 + 1]; tz = p.normals[ray * 3 + 2];
      }
      // sdf loss (loss.py:122-164)
      const bool freeSp = bnd > lc.trunc_distance;
      const float sd = my_sdf;
      float v, dv;
      if (freeSp) {
        const float m1 = fmaxf(sd - bnd, 0.f), ex = __expf(-5.f * sd), m2 = ex - 1.f;
        v = fmaxf(m1, m2);
        dv = m1 >= m2 ? (sd > bnd ? 1.f : 0.f) : -5.f * ex;
      } else { v = sd - bnd; dv = 1.f; }
      float sl, ds;
      if (lc.loss_type == 0) { sl = fabsf(v); ds = (v > 0.f ? 1.f : (v < 0.f ? -1.f : 0.f)) * dv; }
      else { sl = v * v; ds = 2.f * v * dv; }
      if (!freeSp) { sl *= lc.trunc_weight; ds *= lc.trunc_weight; }
      float tot = sl;
      lsum[0] = sl;
      sbar = ds;
      const float gn = sqrtf(gx * gx + gy * gy + gz * gz);
      const float inv = gn > 0.f ? 1.f / gn : 0.f;
      const float nx = gx * inv, ny = gy * inv, nz = gz * inv;
      if (lc.grad_weight != 0.f) {  // trainer.py:818-830, CosineSimilarity eps 1e-6
        const float tn = fmaxf(sqrtf(tx * tx + ty * ty + tz * tz), 1e-6f);
        const float hx = tx / tn, hy = ty / tn, hz = tz / tn;
        const float gc = fmaxf(gn, 1e-6f);
        const float cs = (gx * hx + gy * hy + gz * hz) / gc;
        float gl = 1.f - cs;
        if (lc.orien_loss) gl = gl > 1.f ? 1.f : 0.f;
        else {
          const float k = lc.grad_weight / gc;
          if (gn > 1e-6f) { bx -= k * (hx - cs * nx); by -= k * (hy - cs * ny); bz -= k * (hz - cs * nz); }
          else { bx -= k * hx; by -= k * hy; bz -= k * hz; }
        }
        lsum[1] = gl;
        tot += lc.grad_weight * gl;
      }
      if (lc.eik_weight != 0.f) {  // trainer.py:814-816, loss.py:196-199
        float ek = fabsf(gn - 1.f);
        if (bnd < lc.eik_apply_dist) ek = 0.f;
        else {
          const float sg = gn > 1.f ? 1.f : (gn < 1.f ? -1.f : 0.f);
          bx += lc.eik_weight * sg * nx; by += lc.eik_weight * sg * ny; bz += lc.eik_weight * sg * nz;
        }
        ek *= lc.eik_weight;
        lsum[2] = ek;
        tot += ek;
      }
      lsum[3] = tot;
      if (p.tot_loss_mat) p.tot_loss_mat[n] = tot;
      p.tot_ws[n] = tot;
    }
    if (MODE == 2) {
      // gbar in x' space: x' = si (R x + t)  =>  gbar_x' = si * R gbar_x
      gbs[tid * 4] = si * (L.T[0] * bx + L.T[1] * by + L.T[2] * bz);
      gbs[tid * 4 + 1] = si * (L.T[4] * bx + L.T[5] * by + L.T[6] * bz);
      gbs[tid * 4 + 2] = si * (L.T[8] * bx + L.T[9] * by + L.T[10] * bz);
      gbs[tid * 4 + 3] = sbar * so;
    }
  }
  if (MODE != 2) return;
  if (tid < BM) {  // per-wave loss / sbar sums -> LDS, one thread combines (deterministic)
    float v5[5] = {lsum[0], lsum[1], lsum[2], lsum[3], gbs[tid * 4 + 3]};
#pragma unroll
    for (int k = 0; k < 5; ++k) {
      float v = v5[k];
#pragma unroll
      for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
      if (lane == 0) red[w * 8 + k] = v;
    }
  }
  lds_barrier();
  if (tid == 0) {
    float s[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
    for (int q = 0; q < BM / 64; ++q)
#pragma unroll
      for (int k = 0; k < 5; ++k) s[k] += red[q * 8 + k];
#pragma unroll
    for (int k = 0; k < 4; ++k) p.wg_loss[(int64_t)blockIdx.x * 8 + k] = s[k];
    const int64_t rem = P - n0;
    p.wg_loss[(int64_t)blockIdx.x * 8 + 4] = (float)(rem < BM ? rem : BM);
    vecTile[L.L * HD + 2 * HD] = s[4];   // d b_out = sum sbar*so
  }

  // ------------------------------------------------------------------ Ebar = J_pe gbar  -> region 2 (bf16)
  {
    const int pt = tid & (BM - 1), prt = tid / BM;
    constexpr int NPART = (T::NW * 64) / BM;
    const float y0 = xs[pt * 4], y1 = xs[pt * 4 + 1], y2 = xs[pt * 4 + 2];
    const float b0 = gbs[pt * 4], b1 = gbs[pt * 4 + 1], b2 = gbs[pt * 4 + 2];
    char* row = X + pt * ROWB;
    auto put = [&](int feat, float v) { *(__bf16*)(row + swz(pt, (HD + feat) * 2)) = (__bf16)v; };
    if (prt == 0) {
      put(0, b0); put(1, b1); put(2, b2);
      for (int f = L.E; f < EP; ++f) put(f, 0.f);
    }
    for (int d = prt; d < N_DIRS; d += NPART) {
      const float dx = kDirs[0][d], dy = kDirs[1][d], dz = kDirs[2][d];
      const float proj = y0 * dx + y1 * dy + y2 * dz;
      const float c = b0 * dx + b1 * dy + b2 * dz;
      float fr = 1.f;
      for (int f = 0; f < nf; ++f) {
        const float xb = proj * fr;
        put(3 + d * nf + f, __cosf(xb) * fr * c);
        put(3 + N_DIRS * nf + d * nf + f, __cosf(xb + kHalfPi) * fr * c);
        fr *= 2.f;
      }
    }
  }
  lds_barrier();
  spill_region(HD, spillTile + p.sp.GB[0]);

  // ------------------------------------------------------------------ adjoint of the first reverse sweep (upward)
  for (int li = 0; li < L.L; ++li) {
    Pre preA, preP;
    auto pf = [&] { prefetch(p.sp.A[li + 1], preA); prefetch(p.sp.P[li], preP); };
    zero_acc(acc);
    if (li == 0)
      gemm<false, EP / 16, FB, PB, ROWB>(acc, wptr(setFwdB, L.fwdMat[0], EP), (EP / 16) * 64, X, HD * 2, lane, pf);
    else if (li == L.cat)
      gemm<false, (HD + EP) / 16, FB, PB, ROWB>(acc, wptr(setFwdB, L.fwdMat[li], HD + EP), ((HD + EP) / 16) * 64, X, 0, lane, pf);
    else
      gemm<false, HD / 16, FB, PB, ROWB>(acc, wptr(setFwdB, L.fwdMat[li], HD), (HD / 16) * 64, X, 0, lane, pf);
    TS();
    lds_barrier();
    TS();
    const bool last = li == L.L - 1;
    float qsum[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) qsum[e] = 0.f;
    for_blocks2([&](int fb, int pb, int qp, int f0, int row) {
      float a[8], pv[8], qb[8], inj[8];
      load_tile8(preA, fb, pb, qp, a);
      load_tile8(preP, fb, pb, qp, pv);
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float u = acc[fb][pb][8 * qp + e];
        const float s1 = s1_from_a(a[e]);
        qb[e] = u * s1;
        inj[e] = kBeta * u * pv[e] * (1.f - s1);   // u * q * sigma''(z),  q*sigma' = p
        if (last) qsum[e] += qb[e];
      }
      store_tile8(p.sp.INJ[li], fb, pb, qp, inj);
      if (!last) {
        put_x(false, row, f0, qb, 0);
        store_tile8(p.sp.GB[li + 1], fb, pb, qp, qb);
      }
    }, [&](int f0) {
      if (last) {  // d w_out += so * sum_pts qbar_L
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          half_wave_store(so * qsum[e], vecTile + L.L * HD + f0 + (e & 3) + 8 * (e >> 2), lane);
          qsum[e] = 0.f;
        }
      }
    });
    lds_barrier();
  }

  // ------------------------------------------------------------------ ordinary reverse sweep with injection
  {
    const float* wout = p.params + L.offWout;
    for (int li = L.L - 1; li >= 0; --li) {
      const bool top = li == L.L - 1;
      Pre preA, preI;
      auto pf = [&] { prefetch(p.sp.A[li + 1], preA); prefetch(p.sp.INJ[li], preI); };
      if (!top) {
        zero_acc(acc);
        gemm<false, HD / 16, FB, PB, ROWB>(acc, wptr(setBwdB, L.bwdMat[li + 1], HD), (HD / 16) * 64, X, 0, lane, pf);
        TS();
        lds_barrier();
        TS();
      } else {
        pf();
      }
      float bsum[8], wsum[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) { bsum[e] = 0.f; wsum[e] = 0.f; }
      for_blocks2([&](int fb, int pb, int qp, int f0, int row) {
        float a[8], inj[8], zb[8];
        load_tile8(preA, fb, pb, qp, a);
        load_tile8(preI, fb, pb, qp, inj);
        const float sb = gbs[row * 4 + 3];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float ab = top ? sb * wout[f0 + (e & 3) + 8 * (e >> 2)] : acc[fb][pb][8 * qp + e];
          zb[e] = ab * s1_from_a(a[e]) + inj[e];
          bsum[e] += zb[e];
          if (top) wsum[e] += sb * a[e];
        }
        store_tile8(p.sp.ZB[li], fb, pb, qp, zb);
        if (li > 0) put_x(false, row, f0, zb, 0);
      }, [&](int f0) {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const int f = f0 + (e & 3) + 8 * (e >> 2);
          half_wave_store(bsum[e], vecTile + li * HD + f, lane);
          if (top) half_wave_store(wsum[e], vecTile + L.L * HD + HD + f, lane);
          bsum[e] = 0.f; wsum[e] = 0.f;
        }
      });
      TS();
      lds_barrier();
      TS();
    }
  }
}

// ---------------------------------------------------------------------------
template <int HD, bool F16, int MODE>
static int launch_one(const ChainParams& p, int64_t nTiles, hipStream_t st) {
  typedef Tile<HD, HD> T;
  auto k = chain_kernel<HD, HD, F16, MODE>;
  if (hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, T::LDS_BYTES) != hipSuccess) return ISDF_EHIP;
  hipLaunchKernelGGL(k, dim3((unsigned)nTiles), dim3(CHAIN_NW * 64), T::LDS_BYTES, st, p);
  return hipGetLastError() == hipSuccess ? ISDF_OK : ISDF_EHIP;
}

template <int MODE>
static int launch_mode(const ChainParams& p, int64_t nTiles, hipStream_t st) {
  if (p.lay.HD == 256) return p.lay.fwd_f16 ? launch_one<256, true, MODE>(p, nTiles, st) : launch_one<256, false, MODE>(p, nTiles, st);
  return p.lay.fwd_f16 ? launch_one<512, true, MODE>(p, nTiles, st) : launch_one<512, false, MODE>(p, nTiles, st);
}

int launch_chain(const ChainParams& p, int mode, int64_t nTiles, hipStream_t st) {
  if (!layout_supported(p.lay)) return ISDF_EUNSUPPORTED;
  if (nTiles <= 0) return ISDF_OK;
  switch (mode) {
    case 0: return launch_mode<0>(p, nTiles, st);
    case 1: return launch_mode<1>(p, nTiles, st);
    case 2: return launch_mode<2>(p, nTiles, st);
  }
  return ISDF_EINVAL;
}

}  // namespace isdf
