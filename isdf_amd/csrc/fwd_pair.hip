// Pair-tile forward kernel (K2, MODE 0, <HD = 256, EP = 256>): positional encoding -> SDF MLP forward for pairs of 64-point
// tiles ("halves"), ONE persistent workgroup per CU walking pairs b, b + grid, ..., software-pipelined so that inside every wave the
// MFMAs of one half's GEMM issue between the Softplus / pack / LDS-write instructions of the other half's epilogue.  Replaces, per
// 128 points,
//   embedding.PostionalEncoding.forward   isdf/modules/embedding.py:95-111
//   SDFMap.forward                        isdf/modules/fc_map.py:94-111
// (the same operand types and accumulation order as chain_kernel<256, 256, OPER, 0>; Softplus as chain_dev.h's softplus_x(): the first
// two versions of this kernel, with softplus_f(), were bit-identical to the one-tile kernel on every ragged size and operand mode --
// profiles/r05_fwd_pair_v1_ab.txt -- this one sits <= 3e-5 of the output scale from it, fp32 rounding of the reformulated epilogue, and
// as close to the fp32 oracle as the one-tile kernel: tests/test_gpu_parity.py.)
//
// Why: chain.hip gets its MFMA / VALU overlap only from the hardware scheduler picking between two independent
// workgroups per CU; all eight waves of a workgroup are in a GEMM or in an epilogue at the same time, every GEMM starts with an
// exposed L2 round trip for its weight fragments and every tile streams its own copy of every matrix through the CU's 64 B/clk
// vector-memory path (as busy as the matrix pipe would be at 100 %).  Here, per wave (229 of 256 VGPRs):
//
//      stage 2l   :  GEMM of half B, layer l      interleaved with   epilogue of half A, layer l          (barrier)
//      stage 2l+1 :  GEMM of half A, layer l+1    interleaved with   epilogue of half B, layer l          (barrier)
//      (past the last layer: GEMM of half A, layer 0 of the NEXT pair beside the epilogue of half B's last layer -- the loop over
//       pairs is rotated by one stage, so only a workgroup's first and last stage are one-sided)
//
//   * "interleaved" is by construction, not by the scheduler's choice: the stage is one unrolled instruction stream of
//     [LDS operand read, MFMA, one epilogue PHASE (one Softplus step of four elements), MFMA, phase, weight request] groups separated
//     by sched_barriers, so VALU and matrix pipe are busy in the same cycles of the same wave;
//   * the weight fragments of a layer sit in a 16-fragment window (64 VGPRs) and serve BOTH halves: half B re-requests every
//     register with the next layer's fragment right after its last use, a full stage before half A needs it -- no GEMM waits for
//     L2, and a K = 256 layer is fetched once per 128 points (K = 512 and the compensated layers stream through the window once
//     per half); past the last layer the "next" fragments are layer 0's, for the next pair;
//   * one barrier per GEMM (the stage's GEMM reads one half of the tile, its epilogue writes the other);
//   * per workgroup, not per pair: the biases / w_out staged in LDS, the first fill of the window; a pair's coordinates are
//     requested a pair ahead.
// What bounds it: the SIMD issue port (0.88 utilised: a Softplus element is 38 issue cycles, a K = 256 stage has four per MFMA), not
// the matrix pipe (0.50); packed fp32 instructions are no way around it (they occupy the matrix pipe), nor is wave priority.
#include "chain_dev.h"

namespace isdf {

struct FwdPairTile {
  static constexpr int HD = 256, HB = TILE_PTS, BM = 2 * TILE_PTS, NW = 8;
  static constexpr int ROWB = 4 * HD;                  // bytes per LDS row: [a | emb] 16-bit elements (emb -> a_lo past the cat layer, fp16x2)
  static constexpr int HALFB = HB * ROWB;              // one half of the tile (64 KB)
  static constexpr int OFF_RAW = 2 * HALFB;            // float [NW][BM]: per-wave partial of the output layer
  static constexpr int OFF_BIAS = OFF_RAW + NW * BM * 4;   // float [MAXL][HD] hidden biases, then [HD] w_out (zero beyond unit H)
  static constexpr int LDS_BYTES = OFF_BIAS + (MAXL + 1) * HD * 4;
};
static_assert(FwdPairTile::LDS_BYTES <= 160 * 1024, "one workgroup per CU");
static_assert(TILE_PTS == 64, "a half is one 64-point tile");

// unit kinds: the fragment stream a layer's GEMM consumes per half (16 fragments = K 256 of one packed matrix slice)
enum { UK_NONE = 0, UK_16 = 1, UK_32 = 2, UK_16T = 3, UK_32L = 4 };
//   UK_16   one matrix, K = 256                                   (input layer, hidden layers)
//   UK_32   one matrix, K = 512                                   (cat layer)
//   UK_16T  W (a + a_lo) in one pass over W, then W_lo a          (fp16x2: hidden layers past the cat layer)
//   UK_32L  W [a | emb], then W_lo[:, HD:] emb                    (fp16x2: cat layer)
enum { EK_NONE = 0, EK_HID = 1, EK_HILO = 2, EK_LAST = 3 };
constexpr int uk_frags(int uk) { return uk == UK_16 ? 16 : uk == UK_32 ? 32 : uk == UK_16T ? 32 : uk == UK_32L ? 48 : 0; }

template <typename F, int... I>
__device__ __forceinline__ void static_layers(F&& f, std::integer_sequence<int, I...>) { (f(std::integral_constant<int, I>{}), ...); }

struct FwdUnit { int kind, soff0, soff1, col0; };   // byte offsets of the wave's slice of the unit's matrices; LDS column (bytes) of its first operand

// NL / CAT: the number of hidden layers and the index of the cat layer as COMPILE-TIME constants: the layer loop unrolls into one
// straight-line stream of stages (no stage dispatch at run time, no 128-register PHIs at loop joins -- with a run-time loop the
// register allocator spilled a dozen window fragments right behind their loads).
template <int OPER, int NL, int CAT, int NF>
__global__ __launch_bounds__(FwdPairTile::NW * 64, 2) void fwd_pair_kernel(const ChainParams p) {
  typedef FwdPairTile T;
  constexpr bool F16 = OPER >= 1, X2 = OPER >= 2;
  constexpr int HD = T::HD, EP = T::HD, HB = T::HB, BM = T::BM, ROWB = T::ROWB;
  typedef typename Op<F16>::v8 v8;
  typedef typename Op<F16>::e opT;
  extern __shared__ __attribute__((aligned(1024))) char smem[];
  float* rawL = (float*)(smem + T::OFF_RAW);
  float* biasL = (float*)(smem + T::OFF_BIAS);

  const NetLayout& L = p.lay;
  const int tid = threadIdx.x;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lane = tid & 63, j = lane & 31, hi = lane >> 5, lane16 = lane * 16;
  const int64_t P = p.n_valid ? (int64_t)(*p.n_valid) * p.S : p.n_points_host;
  const int64_t nPairs = (P + BM - 1) / BM;
  if ((int64_t)blockIdx.x >= nPairs) return;
  const float so = L.scale_output;
  ChainStamps TS(p.dbg);
  TS();
  TS.wall(0);

  const rsrc_t rsW = make_rsrc(p.shadow, 0x7fffffffu);
  auto unit_of = [&](int li) __attribute__((always_inline)) {
    FwdUnit u;
    const int kp = li == 0 ? EP : (li == CAT ? HD + EP : HD);
    u.soff0 = (int)((L.setFwdA + L.fwdMat[li]) * 2) + w * (kp / 16) * 1024;
    u.soff1 = (int)((L.setFwdLo + L.fwdMat[li]) * 2) + w * (kp / 16) * 1024 + (li == CAT ? (HD / 16) * 1024 : 0);
    u.col0 = li == 0 ? HD * 2 : 0;
    const bool comp = X2 && li >= CAT;
    u.kind = li == CAT ? (comp ? UK_32L : UK_32) : (comp ? UK_16T : UK_16);
    return u;
  };
  // fragment t of a unit's stream: one 1 KB wave-load (the wave's 32 rows x 16 k)
  auto frag = [&](int soff, int ks) __attribute__((always_inline)) { return bload16<0>(rsW, lane16 + (ks & 3) * 1024, soff + (ks >> 2) * 4096); };
  uint4 W[16];
  {
    const FwdUnit u0 = unit_of(0);
#pragma unroll
    for (int r = 0; r < 16; ++r) W[r] = frag(u0.soff0, r);
  }

  // ------------------------------------------------------------------ biases / w_out -> LDS (zero beyond unit H), PE of both halves
  {
    // all of a thread's loads in flight together: a load -> store loop pays one dependent L2 / HBM round trip per iteration
    constexpr int NB = ((NL + 1) * HD + T::NW * 64 - 1) / (T::NW * 64);
    float bvv[NB];
#pragma unroll
    for (int i = 0; i < NB; ++i) {
      const int q = tid + i * T::NW * 64, li = q / HD, u = q % HD;
      bvv[i] = 0.f;
      if (q < (NL + 1) * HD && u < L.H) bvv[i] = p.params[(li < NL ? L.offB[li] : L.offWout) + u];
    }
#pragma unroll
    for (int i = 0; i < NB; ++i) {
      const int q = tid + i * T::NW * 64, li = q / HD, u = q % HD;
      if (q < (NL + 1) * HD)   // hidden biases on the base-2 scale of softplus_x() (chain_dev.h)
        biasL[(li < NL ? li : MAXL) * HD + u] = li < NL ? bvv[i] * kC1 : bvv[i];
    }
  }
  // The workgroup is PERSISTENT: it walks pairs blockIdx, blockIdx + gridDim, ... (one workgroup per CU).  Per pair that saves the bias
  // staging, the first fill of the weight window (the last stage of a pair has already requested layer 0's fragments: the "next unit"
  // of the last layer IS the next pair's first) and the latency of the coordinate loads (requested one pair ahead).
  auto load_xyz = [&](int64_t pair, float (&o)[2][3]) __attribute__((always_inline)) {
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int64_t n = pair * BM + h * HB + lane;
      o[h][0] = 0.f; o[h][1] = 0.f; o[h][2] = 0.f;
      if (pair < nPairs && n < P) { o[h][0] = p.pts[n * 3]; o[h][1] = p.pts[n * 3 + 1]; o[h][2] = p.pts[n * 3 + 2]; }
    }
  };
  float pxyz[2][3];
  load_xyz(blockIdx.x, pxyz);
  const int sw = (lane & 15) << 4;
  auto pe_stage = [&](const float (&xyz)[2][3]) __attribute__((always_inline)) {
    // PE (embedding.py:95-111).  A lane is a POINT of a half, a wave takes (direction, half) items w, w + 8, ... of the 42: the
    // direction is wave-uniform, so a feature's column is a scalar and its LDS address one v_xad of the lane's row base and swizzle;
    // the octaves are unrolled (NF is a template constant), and octave pairs (1,2), (3,4) of a direction -- 4-byte aligned in the
    // row, never across a 16-byte swizzle slot -- leave as one packed store.  (chain.hip's mapping, 8 points x 8 direction slices
    // per wave with run-time octave loops, took 12.9 k of this workgroup's 70 k cycles: profiles/r05_fwd_pair_v1_timeline_fp16.txt.)
    // Octaves 0 and 3 of a direction are evaluated as chain.hip does (xb = proj 2^f, sin(xb), sin(xb + pi/2)), the two octaves above
    // each by angle doubling (s' = 2 s c, c' = 1 - 2 s^2: three plain operations instead of two transcendental ones and their range
    // scaling; two doublings from an exact start: <= 1e-6 from the direct value, a 250th of the operand's fp16 rounding step).
    float ys[2][3];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const float x0 = xyz[h][0], x1 = xyz[h][1], x2 = xyz[h][2];
      // transform_3D_grid (transform.py:287-304) then * scale (embedding.py:12-22)
      ys[h][0] = (L.T[0] * x0 + L.T[1] * x1 + L.T[2] * x2 + L.T[3]) * L.scale_input;
      ys[h][1] = (L.T[4] * x0 + L.T[5] * x1 + L.T[6] * x2 + L.T[7]) * L.scale_input;
      ys[h][2] = (L.T[8] * x0 + L.T[9] * x1 + L.T[10] * x2 + L.T[11]) * L.scale_input;
    }
    auto put1 = [&](int rowb, int feat, float v) __attribute__((always_inline)) { *(opT*)(smem + rowb + (((HD + feat) * 2) ^ sw)) = (opT)v; };
    auto put2 = [&](int rowb, int feat, float v0, float v1) __attribute__((always_inline)) {   // features feat, feat + 1 at a 4-byte aligned column
      typedef opT op2 __attribute__((ext_vector_type(2)));
      const op2 q = {(opT)v0, (opT)v1};
      *(uint32_t*)(smem + rowb + (((HD + feat) * 2) ^ sw)) = __builtin_bit_cast(uint32_t, q);
    };
    if (w == T::NW - 1) {   // the identity features and the padding column, on the wave with the fewest items
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int rowb = h * T::HALFB + lane * ROWB;
        put1(rowb, 0, ys[h][0]); put1(rowb, 1, ys[h][1]); put1(rowb, 2, ys[h][2]);
        for (int f = 3 + 2 * N_DIRS * NF; f < EP; ++f) put1(rowb, f, 0.f);
      }
    }
#pragma unroll 1
    for (int it = w; it < 2 * N_DIRS; it += T::NW) {
      const int h = it >= N_DIRS, d = h ? it - N_DIRS : it;
      const float y0 = h ? ys[1][0] : ys[0][0], y1 = h ? ys[1][1] : ys[0][1], y2 = h ? ys[1][2] : ys[0][2];
      const int rowb = h * T::HALFB + lane * ROWB;
      const float proj = y0 * kDirs[0][d] + y1 * kDirs[1][d] + y2 * kDirs[2][d];
      float sv[NF], cv[NF];
      float fr = 1.f;
#pragma unroll
      for (int f = 0; f < NF; ++f) {
        if (f % 3 == 0) {
          const float xb = proj * fr;
          sv[f] = __sinf(xb);
          cv[f] = __sinf(xb + kHalfPi);
        } else {
          const float t2 = sv[f - 1] + sv[f - 1];
          sv[f] = t2 * cv[f - 1];
          cv[f] = __builtin_fmaf(-t2, sv[f - 1], 1.f);
        }
        fr *= 2.f;
      }
      // column of (d, f) is 3 + d NF + f (+ N_DIRS NF for the shifted block): odd for f = 0 when NF is even, so the pairs start at f = 1
      const int fs = 3 + d * NF, fc = 3 + N_DIRS * NF + d * NF;
      static_assert(NF % 2 == 0, "the pairing below assumes an even number of octaves (odd first column of every direction)");
      put1(rowb, fs, sv[0]); put1(rowb, fc, cv[0]);
#pragma unroll
      for (int f = 1; f + 1 < NF; f += 2) { put2(rowb, fs + f, sv[f], sv[f + 1]); put2(rowb, fc + f, cv[f], cv[f + 1]); }
      put1(rowb, fs + NF - 1, sv[NF - 1]); put1(rowb, fc + NF - 1, cv[NF - 1]);
    }
  };

  // ------------------------------------------------------------------ the stage
  // LDS byte offsets (within a half) of this lane's operand reads and epilogue writes (chain.hip: gemm() / put_x())
  const int xlane = j * ROWB + ((hi * 16) ^ ((j & 15) << 4));
  const int xw = j * ROWB + 8 * hi + (((j & 15) << 4) ^ ((w & 3) * 64)) + (w >> 2) * 256;
  f32x16 accA[2], accB[2];
  float rawp[2];

  // GEMM of half `xg` (unit `u`, kind UK) into accG, interleaved with the epilogue EK of accE into half `xe`.
  // IS_B: the second half to use the unit: every window register is re-requested with the NEXT unit's fragment after its last use.
  auto stage = [&](auto ukc, auto ekc, auto isb, f32x16 (&accG)[2], f32x16 (&accE)[2], const FwdUnit u, const int nxtSoff0,
                   const char* xg, char* xe, const int eli, const int eh) __attribute__((always_inline)) {
    constexpr int UK = decltype(ukc)::value, EK = decltype(ekc)::value;
    constexpr bool IS_B = decltype(isb)::value;
    constexpr bool TWO = UK == UK_16T;
    // per-stage lane bases, opaque to the optimiser: otherwise every (k-step, half) address is hoisted out of the layer loop as a
    // lane constant (two dozen VGPRs at a budget that has none to spare) instead of one v_xor next to its ds_read
    int xgl = xlane + (int)(xg - smem), xel = xw + (int)(xe - smem);
    asm volatile("" : "+v"(xgl), "+v"(xel));
    // ---- the epilogue as a software pipeline of PHASES over groups of four elements (one 8-byte piece of the tile):
    //   phase 0  x = fma(beta log2(e), acc, bias')       phase 1  y = 2^min(x, 30)      phase 2  y = log2(1 + y)
    //   phase 3  a = ln2/beta max(x, y), pack, store     (phase 4, fp16x2: the residual piece a - fp16(a))
    // One phase runs behind one MFMA: its four elements are independent (the dependent steps of Softplus -- two of them
    // transcendental -- are an MFMA apart instead of back to back: the first version, one element at a time, was bound by that
    // latency chain: profiles/r05_fwd_pair_v1_timeline_fp16.txt), and the phases are spread evenly over the stage's MFMA slots.
    // The same operations on the same values as chain_dev.h's softplus_x().
    constexpr int PH = EK == EK_HILO ? 5 : 4;
    constexpr int NQ = EK == EK_NONE ? 0 : 8 * PH;                                        // phases of the stage's epilogue
    constexpr int NS = UK == UK_NONE ? NQ : (UK == UK_16 ? 32 : UK == UK_32 ? 64 : 96);   // MFMA slots of the stage's GEMM
    static_assert(NS >= NQ, "at most one phase per MFMA slot");
    float bv[2][8], wv[2][8], z[4], y[4];
    uint2 hp = make_uint2(0u, 0u);
    f32x2 r2 = {0.f, 0.f};
    auto load_bv = [&](int qp) __attribute__((always_inline)) {   // features 32 w + 16 qp + 4 hi + {0..3, 8..11}
      if constexpr (EK != EK_NONE) {
        const float* bsrc = biasL + eli * HD + w * 32 + 16 * qp + 4 * hi;
        const float4 b0 = *(const float4*)bsrc, b1 = *(const float4*)(bsrc + 8);
        bv[qp][0] = b0.x; bv[qp][1] = b0.y; bv[qp][2] = b0.z; bv[qp][3] = b0.w; bv[qp][4] = b1.x; bv[qp][5] = b1.y; bv[qp][6] = b1.z; bv[qp][7] = b1.w;
        if constexpr (EK == EK_LAST) {
          const float* wsrc = biasL + MAXL * HD + w * 32 + 16 * qp + 4 * hi;
          const float4 w0 = *(const float4*)wsrc, w1 = *(const float4*)(wsrc + 8);
          wv[qp][0] = w0.x; wv[qp][1] = w0.y; wv[qp][2] = w0.z; wv[qp][3] = w0.w; wv[qp][4] = w1.x; wv[qp][5] = w1.y; wv[qp][6] = w1.z; wv[qp][7] = w1.w;
        }
      }
    };
    load_bv(0);
    if constexpr (EK == EK_LAST) { rawp[0] = 0.f; rawp[1] = 0.f; }
    (void)wv; (void)r2; (void)z; (void)y; (void)bv; (void)hp;
    auto phase = [&](int q) __attribute__((always_inline)) {
      if constexpr (EK != EK_NONE) {
        const int g = q / PH, ph = q % PH;
        const int qp = g >> 2, pb = (g >> 1) & 1, hb = g & 1;     // elements 8 qp + 4 hb + {0..3} of accE[pb]
        if (ph == 0) {
          if (g == 2) load_bv(1);                                 // the second block's parameters, well ahead of group 4
#pragma unroll
          for (int i = 0; i < 4; ++i) z[i] = __builtin_fmaf(kC1, accE[pb][8 * qp + 4 * hb + i], bv[qp][4 * hb + i]);
        } else if (ph == 1) {
#pragma unroll
          for (int i = 0; i < 4; ++i) y[i] = __builtin_amdgcn_exp2f(fminf(z[i], 30.f));
        } else if (ph == 2) {
#pragma unroll
          for (int i = 0; i < 4; ++i) y[i] = __builtin_amdgcn_logf(1.f + y[i]);
        } else if (ph == 3) {
#pragma unroll
          for (int i = 0; i < 4; ++i) z[i] = kC2 * fmaxf(z[i], y[i]);          // z <- the activation
          if constexpr (EK == EK_LAST) {
            // w_out . a as packed FMAs of ELEMENT PAIRS (never a broadcast operand: isa_lint rule 1), same grouping as chain.hip
            r2 += f32x2{wv[qp][4 * hb], wv[qp][4 * hb + 1]} * f32x2{z[0], z[1]};
            r2 += f32x2{wv[qp][4 * hb + 2], wv[qp][4 * hb + 3]} * f32x2{z[2], z[3]};
            if (hb == 1) { rawp[pb] += r2[0] + r2[1]; r2 = f32x2{0.f, 0.f}; }
          } else {   // four values -> one 8-byte piece of the tile (features f0 .. f0+3 | f0+8 .. f0+11)
            const int lb = ((xel ^ (32 * qp)) + pb * 32 * ROWB) ^ (hb ? 16 : 0);
            hp = pack4<EK == EK_HILO ? true : F16>(z[0], z[1], z[2], z[3]);
            *(uint2*)(smem + lb) = hp;
          }
        } else {     // fp16x2: region 2 <- fp16(a - fp16(a)), the second operand of the next layer's compensated GEMM.
          // The residual is formed against the STORED halves (hp), not against a second conversion of a: the compiler contracts the
          // k ln2 / beta product into whichever conversion it meets (v_fma_mix*: one rounding) or not (two), and a residual formed
          // against the other kind of half is off by an fp16 ulp of a whenever the two roundings disagree (2^-13 of the elements;
          // 1.5e-5 of sdf at the worst of 300 k points between two builds of this file: profiles/r05_fwd_pair_v5_ab.txt).
          const int lb = ((xel ^ (32 * qp)) + pb * 32 * ROWB) ^ (hb ? 16 : 0);
          typedef _Float16 h4 __attribute__((ext_vector_type(4)));
          const h4 hv = __builtin_bit_cast(h4, hp);
          *(uint2*)(smem + lb + HD * 2) = pack4<true>(z[0] - (float)hv[0], z[1] - (float)hv[1], z[2] - (float)hv[2], z[3] - (float)hv[3]);
        }
      }
    };
    // the phases that run behind MFMA slot m: phase q sits in slot q NS / NQ
    auto slot_done = [&](int m) __attribute__((always_inline)) {
      if constexpr (EK != EK_NONE) {
        const int q = (m * NQ + NS - 1) / NS;          // the only candidate (NS >= NQ): the smallest q with q NS / NQ >= m
        if (q < NQ && q * NS / NQ == m) phase(q);
      }
    };
    // operand reads of step t: segment / k-step / LDS column.  The first operand is read one step ahead (double-buffered); the second
    // operand of a two-operand step (a_lo) at the start of its own step, two MFMAs ahead of its use.
    v8 bq[2][2], bl[2];
    auto opaddr = [&](int t) __attribute__((always_inline)) {
      int ks, col;
      if (UK == UK_16 || UK == UK_32) { ks = t; col = u.col0; }
      else if (UK == UK_16T) { ks = t & 15; col = 0; }
      else { ks = t < 32 ? t : t - 32; col = t < 32 ? 0 : HD * 2; }
      return (xgl ^ ((ks & 7) * 32)) + (ks >> 3) * 256 + col;
    };
    auto readb = [&](int t, v8 (&b)[2]) __attribute__((always_inline)) {
      const int a = opaddr(t);
#pragma unroll
      for (int pb = 0; pb < 2; ++pb) b[pb] = __builtin_bit_cast(v8, *(const uint4*)(smem + a + pb * 32 * ROWB));
    };
    auto readlo = [&](int t) __attribute__((always_inline)) {
      const int a = opaddr(t) + HD * 2;
#pragma unroll
      for (int pb = 0; pb < 2; ++pb) bl[pb] = __builtin_bit_cast(v8, *(const uint4*)(smem + a + pb * 32 * ROWB));
    };
    (void)bl; (void)bq;
    if constexpr (UK == UK_NONE) {
      // epilogue-only stage (the last layer of the second half): no MFMA to hide behind, the scheduler may interleave the groups freely
#pragma unroll
      for (int q = 0; q < NQ; ++q) phase(q);
    } else {
      constexpr int NT = uk_frags(UK);
      readb(0, bq[0]);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        const int r = t & 15;
        const bool two = TWO && t < 16;
        const int m0 = TWO ? (t < 16 ? 4 * t : 64 + 2 * (t - 16)) : 2 * t;   // first MFMA slot of the step
        if (t + 1 < NT) readb(t + 1, bq[(t + 1) & 1]);
        if constexpr (TWO) { if (t < 16) readlo(t); }
        __builtin_amdgcn_sched_barrier(0);
        accG[0] = Op<F16>::mfma(__builtin_bit_cast(v8, W[r]), bq[t & 1][0], t == 0 ? f32x16(0.f) : accG[0]);
        __builtin_amdgcn_sched_barrier(0);
        slot_done(m0);
        __builtin_amdgcn_sched_barrier(0);
        accG[1] = Op<F16>::mfma(__builtin_bit_cast(v8, W[r]), bq[t & 1][1], t == 0 ? f32x16(0.f) : accG[1]);
        __builtin_amdgcn_sched_barrier(0);
        slot_done(m0 + 1);
        if constexpr (TWO) {
          if (two) {
            __builtin_amdgcn_sched_barrier(0);
            accG[0] = Op<F16>::mfma(__builtin_bit_cast(v8, W[r]), bl[0], accG[0]);
            __builtin_amdgcn_sched_barrier(0);
            slot_done(m0 + 2);
            __builtin_amdgcn_sched_barrier(0);
            accG[1] = Op<F16>::mfma(__builtin_bit_cast(v8, W[r]), bl[1], accG[1]);
            __builtin_amdgcn_sched_barrier(0);
            slot_done(m0 + 3);
          }
        }
        // the window register's next tenant
        constexpr int NF_ = uk_frags(UK);
        if (t + 16 < NF_) {   // this unit's fragment t + 16
          const int tt = t + 16;
          if (UK == UK_32) W[r] = frag(u.soff0, tt);
          else if (UK == UK_16T) W[r] = frag(u.soff1, tt - 16);
          else W[r] = tt < 32 ? frag(u.soff0, tt) : frag(u.soff1, tt - 32);
        } else if (IS_B) {
          W[r] = frag(nxtSoff0, r);              // the next unit's fragment r, a stage ahead of half A
        } else if (NF_ > 16) {
          W[r] = frag(u.soff0, r);               // half B starts the unit over
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    if constexpr (EK == EK_LAST) {
#pragma unroll
      for (int pb = 0; pb < 2; ++pb) {
        const float v = rawp[pb] + __shfl_xor(rawp[pb], 32, 64);   // add the two feature halves
        if (hi == 0) rawL[w * BM + eh * HB + pb * 32 + j] = v;
      }
    }
  };

  // the stages, unrolled at compile time: unit kind and epilogue kind of every stage are constants
  char* const XA = smem;
  char* const XB = smem + T::HALFB;
  auto soff0_of = [&](int li) __attribute__((always_inline)) { return (int)((L.setFwdA + L.fwdMat[li]) * 2) + w * ((li == 0 ? EP : (li == CAT ? HD + EP : HD)) / 16) * 1024; };
  // The loop over pairs is ROTATED by one stage: the epilogue of half B's last layer (no GEMM of its own pair left to hide behind) runs
  // behind the NEXT pair's first GEMM (half A, layer 0), whose operand -- the embedding -- the next pair's PE writes once the
  // last GEMM of this pair has read its own.  A pair is then 12 two-sided stages and the PE; only the first pair of a workgroup
  // pays a one-sided stage (before the rotation: every pair two of them -- 2.6 k + 3.2 k cycles of a pair's 27 k,
  // profiles/r05_fwd_pair_v4_timeline_fp16.txt).
  auto layer = [&](auto lic) __attribute__((always_inline)) {
    constexpr int li = decltype(lic)::value;
    constexpr bool comp = X2 && li >= CAT;
    constexpr int UK = li == CAT ? (comp ? UK_32L : UK_32) : (comp ? UK_16T : UK_16);
    constexpr int EKprev = li == 0 ? EK_NONE : (X2 && li > CAT ? EK_HILO : EK_HID);                   // epilogue of layer li - 1 (never the last)
    constexpr int EKcur = li == NL - 1 ? EK_LAST : (X2 && li + 1 > CAT ? EK_HILO : EK_HID);
    const FwdUnit u = unit_of(li);
    const int nxt = soff0_of(li + 1 < NL ? li + 1 : 0);   // (past the last layer: layer 0 again, for the next pair)
    if constexpr (li > 0) {
      // GEMM of half A, layer li || epilogue of half B, layer li - 1   (layer 0's: the last stage of the previous pair / the prologue)
      stage(std::integral_constant<int, UK>{}, std::integral_constant<int, EKprev>{}, std::false_type{}, accA, accB, u, nxt, XA, XB, li - 1, 1);
      TS();
      lds_barrier();
      TS();
    }
    // GEMM of half B, layer li || epilogue of half A, layer li
    stage(std::integral_constant<int, UK>{}, std::integral_constant<int, EKcur>{}, std::true_type{}, accB, accA, u, nxt, XB, XA, li, 0);
    TS();
    lds_barrier();
    TS();
  };
  static_assert(CAT != 0, "layer 0 is a one-operand 16-fragment unit here");
  const FwdUnit u0 = unit_of(0);
  pe_stage(pxyz);
  TS();
  lds_barrier();
  TS();
  stage(std::integral_constant<int, UK_16>{}, std::integral_constant<int, EK_NONE>{}, std::false_type{}, accA, accB, u0, soff0_of(1), XA, XB, -1, 1);
  TS();
  lds_barrier();
  TS();
  for (int64_t pair = blockIdx.x; pair < nPairs; pair += gridDim.x) {
  const int64_t n0 = pair * BM;
  const bool has_next = pair + gridDim.x < nPairs;
  load_xyz(pair + gridDim.x, pxyz);   // the next pair's coordinates: in flight behind this pair's stages
  static_layers(layer, std::make_integer_sequence<int, NL>{});
  if (has_next) pe_stage(pxyz);
  TS();
  lds_barrier();
  TS();
  // GEMM of half A, layer 0 of the NEXT pair (after the last pair: of stale operands, into an accumulator nobody reads) || epilogue of
  // half B, last layer
  stage(std::integral_constant<int, UK_16>{}, std::integral_constant<int, EK_LAST>{}, std::false_type{}, accA, accB, u0, soff0_of(1), XA, XB, NL - 1, 1);
  TS();
  lds_barrier();
  TS();

  // sdf = (raw + noise) * so   (fc_map.py:104-109)
  if (tid < BM) {
    float r = p.params[L.offBout];
#pragma unroll
    for (int k = 0; k < T::NW; ++k) r += rawL[k * BM + tid];
    const int64_t n = n0 + tid;
    if (p.noise) { if (n < P) r += p.noise[n]; }
    else if (p.noise_std != 0.f) {   // Box-Muller on Philox4x32-10 keyed by (seed, offset, point) -- chain.hip's tail
      const uint4 u = philox4x32_10(make_uint4((uint32_t)n, (uint32_t)(n >> 32), (uint32_t)p.noise_off, (uint32_t)(p.noise_off >> 32)),
                                    make_uint2((uint32_t)p.noise_seed, (uint32_t)(p.noise_seed >> 32) ^ 0x5eedu));
      const float u1 = fmaxf(u01(u.x), 1e-7f), u2 = u01(u.y);
      r += p.noise_std * sqrtf(-2.f * __logf(u1)) * __cosf(6.2831853f * u2);
    }
    if (p.sdf && n < P) p.sdf[n] = r * so;
  }
  // (rawL is rewritten eleven barriers from here)
  }
  TS.wall(1);
}

template <int OPER>
static int launch_fwd_pair_oper(const ChainParams& p, int64_t nPairs, hipStream_t st) {
  auto k = fwd_pair_kernel<OPER, 6, 3, 6>;
  if (hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, FwdPairTile::LDS_BYTES) != hipSuccess) return ISDF_EHIP;
  // one persistent workgroup per CU (152 KB of LDS each)
  const int64_t grid = p.n_cu > 0 && nPairs > p.n_cu ? p.n_cu : nPairs;
  hipLaunchKernelGGL(k, dim3((unsigned)grid), dim3(FwdPairTile::NW * 64), FwdPairTile::LDS_BYTES, st, p);
  return isdf_launch_status();
}

// MODE 0 of the <256, 256> tile with hidden_layers_block = 2 and 6 PE octaves (replicaCAD.json / scanNet.json: 6 hidden layers, cat
// layer 3, E = 255) for the bf16 /
// fp16 / fp16x2 operand modes; other depths and fp16x2_full (four operand regions) stay on chain.hip's one-tile kernel
bool fwd_pair_supported(const NetLayout& l) { return l.HD == 256 && l.EP == 256 && !l.fwd_x2_all && l.L == 6 && l.cat == 3 && l.n_freqs == 6; }

int launch_fwd_pair(const ChainParams& p, int64_t nTiles, hipStream_t st) {
  const int64_t nPairs = (nTiles + 1) / 2;
  if (nPairs <= 0) return ISDF_OK;
  if (p.lay.fwd_x2) return launch_fwd_pair_oper<2>(p, nPairs, st);
  return p.lay.fwd_f16 ? launch_fwd_pair_oper<1>(p, nPairs, st) : launch_fwd_pair_oper<0>(p, nPairs, st);
}

}  // namespace isdf
