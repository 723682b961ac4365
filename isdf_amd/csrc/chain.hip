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
// Mapping to CDNA4: one workgroup = 8 waves = 64 points, two workgroups per CU
// (74 KB LDS each; 128 VGPRs per wave) so one does MFMA while the other is in an
// elementwise epilogue.  Every GEMM is C[feature][point] = W[feature][k] * X[k][point] on
// v_mfma_f32_32x32x16_{f16,bf16}: A = packed weights streamed straight from L2
// in fragment order (1 KB contiguous per wave-load, each weight is used by
// exactly one wave of the workgroup so LDS staging would add nothing),
// B = the activation tile in LDS ([point][k], 16-B XOR swizzle, ds_read_b128).
// In the C layout a lane owns one point and 4 consecutive features per
// register quad, so epilogues write 8-byte packed pieces back to the LDS tile.
#include <atomic>
#include "chain_dev.h"

namespace isdf {

// MFMA operand type of the forward / first-reverse GEMMs: 0 bf16, 1 fp16, 2 fp16 with the compensated forward
// ("fp16x2", NetLayout::fwd_x2): layers >= cat also multiply the fp16 RESIDUAL of their weights, and the layers past
// the cat layer the fp16 residual of their input activation (kept in region 2 of the tile, idle there), i.e.
// W x ~= Wh xh + Wl xh + Wh xl -- what brings sdf within 1e-3 of the fp32 reference at BASELINE size.
// 3 "fp16x2_full": the same three products in EVERY forward layer, embedding included -- the exact-forward instrument
// (sdf ~1e-6 of the fp32 reference, d sdf/dx within 1e-3).  It keeps four operand regions in the tile (a, emb, a_lo,
// emb_lo: 128 KB at <256, 256>), so one workgroup per CU; only instantiated for <256, 256>.
constexpr bool oper_f16(int oper) { return oper >= 1; }
constexpr bool oper_x2(int oper) { return oper >= 2; }
constexpr bool oper_x2_all(int oper) { return oper == 3; }

template <int HD, int EP, bool LO_REGIONS = false>
struct Tile {
  static constexpr int BM = TILE_PTS;
  static constexpr int NW = CHAIN_NW;
  static constexpr int CKF = CHAIN_CHUNK_FRAGS;       // weight fragments per chunk
  static constexpr int FB = HD / (NW * 32);   // 32-row feature blocks per wave
  static constexpr int PB = BM / 32;          // 32-point blocks
  static constexpr int R2 = (EP > HD ? EP : HD);
  static constexpr int XK = (HD + R2) * (LO_REGIONS ? 2 : 1);   // elements per LDS row: [a | emb] (+ [a_lo | emb_lo])
  static constexpr int ROWB = XK * 2;
  static constexpr int XBYTES = BM * ROWB;
  // small fp32 arrays after the X tile
  static constexpr int OFF_XS = XBYTES;                   // [BM][4] x' (scaled/transformed point)
  // EP > HD (realsense*.json nets): the embedding gradient goes through a separate fp32 [32 points][HD] buffer in
  // (row half, point block) sub-passes, contracted by 32 points x NPART direction slices
  static constexpr bool WIDE_E = EP > HD;
  static constexpr int NPART = WIDE_E ? (NW * 64) / 32 : (NW * 64) / BM;
  static constexpr int OFF_PART = OFF_XS + BM * 16;       // [NPART][BM][4] partial sums (raw / g)
  static constexpr int OFF_GB = OFF_PART + NPART * BM * 16;   // [BM][4] gbar in x' space, [3] = sbar*so
  static constexpr int OFF_RED = OFF_GB + BM * 16;        // [BM/64][8] per-wave loss sums
  static constexpr int OFF_EG = (OFF_RED + (BM / 64) * 32 + 32 + 1023) / 1024 * 1024;   // fp32 [32][HD], 1 KB-aligned rows
  static constexpr int LDS_BYTES = WIDE_E ? OFF_EG + 32 * HD * 4 : OFF_RED + (BM / 64) * 32 + 32;
};

// C[FB*32 feats][PB*32 pts] += Wpacked[feat][k] * X[pt][k]  over KSTEPS*16 k.
// Measured: a workgroup's time is a serial latency chain, and a staged
// weight loop pays one dependent L2/MALL round trip per stage (8 per K=256 unit).  So the
// wave requests its WHOLE weight slice for a chunk of CK k-steps up front (32 VGPRs:
// one round trip per chunk), one scheduling barrier keeps the load block ahead of the MFMAs,
// and the k-steps are fully unrolled.  `postHook` (the epilogue's spill-tile requests) is issued right
// after the last MFMA, when the weight registers are dead: vmcnt retires in order, so a request in front
// of a weight load would put its HBM round trip into the MFMA stream, and requesting earlier costs
// registers the 128-VGPR budget does not have (measured variants: profiles/r02_ab_chain_variants.txt; issuing the
// requests behind the LAST weight request in the middle of the GEMM measured +2 .. +9 %: profiles/r03_whatif_spill_traffic.txt).
template <int FBN, int CKF> struct WChunk { uint4 v[CKF / FBN][FBN]; };   // one chunk of packed weight fragments (32 VGPRs)
struct WRef { int soff; int rb; };   // byte offset of a wave's slice of a packed matrix in the shadow buffer + row-block stride (bytes)

template <int FBN, int CKF>
__device__ __forceinline__ void load_w(WChunk<FBN, CKF>& wq, rsrc_t rw, WRef r, int lane16, int chunk) {
  constexpr int CK = CKF / FBN;
#pragma unroll
  for (int s = 0; s < CK; ++s)
#pragma unroll
    for (int fb = 0; fb < FBN; ++fb)
      wq.v[s][fb] = bload16<0>(rw, lane16 + (s & 3) * 1024, r.soff + fb * r.rb + chunk * CK * 1024 + (s >> 2) * 4096);
}

// ZERO: the accumulators start from zero -- the first k-step's MFMAs take the inline constant 0 as their C operand instead of
// 16 * FBN * PBN v_mov_b32 zeroing the registers beforehand (8 % of the kernel's VALU instructions).
// TWO: every weight fragment also multiplies a SECOND activation operand at column colByteBase2 (W a_hi + W a_lo of a compensated
// layer in one pass over W: the vector-memory path is the kernel's busiest unit, a second pass would fetch the matrix again).
template <bool F16, int KSTEPS, int FBN, int PBN, int ROWB, int CKF, bool ZERO = false, bool TWO = false, typename Hook>
__device__ __forceinline__ void gemm(f32x16 (&acc)[FBN][PBN], WChunk<FBN, CKF>& wq, rsrc_t rw, WRef wr, const char* xl,
                                     int colByteBase, int lane, Hook&& postHook, int colByteBase2 = 0) {
  constexpr int CK = CKF / FBN;                  // k-steps per chunk: CK*FBN uint4 = 32 VGPRs
  static_assert(KSTEPS % CK == 0, "K must be a multiple of the chunk");
  static_assert((ROWB & 255) == 0, "row base must leave the swizzle bits (4-7) clear");
  constexpr int NCH = KSTEPS / CK;
  typedef typename Op<F16>::v8 v8;
  const int j = lane & 31, hi = lane >> 5;
  // LDS address of (row j, k-step ks) = rowbase + ((ks*32 + hi*16) ^ sw): the swizzle only touches bits 4-7 and
  // the row base has them clear, so it is ONE xor of a per-lane base with the constant (ks&7)*32, the rest
  // (point block, region, ks>>3) being immediate offsets.
  const int xlane = j * ROWB + ((hi * 16) ^ ((j & 15) << 4));   // byte offset from the tile base (kept an offset so the LDS address space survives)
  const int lane16 = lane * 16;
  load_w(wq, rw, wr, lane16, 0);
  // One chunk: the activation operand is read one k-step ahead of the MFMAs that consume it; the weight registers are
  // re-requested for the next chunk after the chunk's last MFMA.
  auto chunk = [&](int ch, auto refill, auto zero) {
    const int k0 = ch * CK;
    int xch = (xlane ^ ((k0 & 7) * 32)) + (k0 >> 3) * 256 + colByteBase;
    // opaque to the optimiser: otherwise the 8 per-k-step addresses (xch ^ s*32) are hoisted out of the layer loops as
    // lane constants, spilled, and reloaded in the middle of the MFMA stream behind an s_waitcnt vmcnt(0)
    asm volatile("" : "+v"(xch));
    constexpr int NB = TWO ? 2 * PBN : PBN;
    auto readb = [&](int s, v8 (&b)[NB]) {
#pragma unroll
      for (int pb = 0; pb < PBN; ++pb)
        b[pb] = __builtin_bit_cast(v8, *(const uint4*)(xl + ((xch ^ (s * 32)) + pb * 32 * ROWB)));
      if constexpr (TWO) {
#pragma unroll
        for (int pb = 0; pb < PBN; ++pb)
          b[PBN + pb] = __builtin_bit_cast(v8, *(const uint4*)(xl + ((xch ^ (s * 32)) + pb * 32 * ROWB + (colByteBase2 - colByteBase))));
      }
    };
    v8 b[2][NB];
    readb(0, b[0]);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int s = 0; s < CK; ++s) {
      if (s + 1 < CK) readb(s + 1, b[(s + 1) & 1]);
#pragma unroll
      for (int fb = 0; fb < FBN; ++fb)
#pragma unroll
        for (int pb = 0; pb < PBN; ++pb)
          acc[fb][pb] = Op<F16>::mfma(__builtin_bit_cast(v8, wq.v[s][fb]), b[s & 1][pb],
                                      (decltype(zero)::value && s == 0) ? f32x16(0.f) : acc[fb][pb]);
      if constexpr (TWO) {
#pragma unroll
        for (int fb = 0; fb < FBN; ++fb)
#pragma unroll
          for (int pb = 0; pb < PBN; ++pb)
            acc[fb][pb] = Op<F16>::mfma(__builtin_bit_cast(v8, wq.v[s][fb]), b[s & 1][PBN + pb], acc[fb][pb]);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    if (decltype(refill)::value) {
      load_w(wq, rw, wr, lane16, ch + 1);
      __builtin_amdgcn_sched_barrier(0);
    }
  };
  static_assert(NCH >= 2, "the first chunk is peeled");
  chunk(0, std::true_type{}, std::integral_constant<bool, ZERO>{});
#pragma unroll 1
  for (int ch = 1; ch < NCH - 1; ++ch) chunk(ch, std::true_type{}, std::false_type{});
  __builtin_amdgcn_sched_barrier(0);
  chunk(NCH - 1, std::false_type{}, std::false_type{});
  postHook();
}

// frag16 spill order: element offset inside a tile for (wave, fb, pb, qp, lane)
template <int FBN, int PBN>
__device__ __forceinline__ int frag16_off(int w, int fb, int pb, int qp, int lane) {
  return ((((w * FBN + fb) * PBN + pb) * 2 + qp) * 64 + lane) * 8;
}

// BW: operand type of the SECOND-order sweeps (adjoint, reverse) and of every spilled tensor (the dW kernel's operands) in train mode:
// false = bf16 (rounds 1-3: range-safe whatever the loss adjoints' magnitude), true = fp16 (NetLayout::bwd_f16: the same packed weight
// copies as the forward / first reverse sweeps, 11 instead of 8 significand bits in every dW operand -- DESIGN 5, round 4).
// SP8 (train mode with BW): P below the top layer and GB are spilled as e4m3 bytes instead of 16-bit values (NetLayout::sp8, SpillLayout).
template <int HD, int EP, int OPER, int MODE, bool BW = false, int SP8 = 0>
__global__ __launch_bounds__(CHAIN_NW * 64, (HD <= 256 && EP == HD && !oper_x2_all(OPER) ? 4 : 2)) void chain_kernel(const ChainParams p) {
  typedef Tile<HD, EP, oper_x2_all(OPER)> T;
  constexpr bool F16 = oper_f16(OPER), X2 = oper_x2(OPER), X2ALL = oper_x2_all(OPER);
  static_assert(!BW || (F16 && MODE == 2), "fp16 second-order sweeps go with fp16 forward operands, train mode only");
  static_assert(!SP8 || BW, "the e4m3 spill format is instantiated for the fp16 second-order sweeps");
  constexpr bool G8 = SP8 & 1, P8 = SP8 & 2;     // GB / P (below the top layer) spilled as e4m3
  typedef typename Op<BW>::e spillT;   // element type of the spilled tensors
  static_assert(!X2ALL || (HD == 256 && EP == 256), "fp16x2_full: four operand regions only fit the <256, 256> tile");
  constexpr int LO = X2ALL ? (HD + EP) : HD;   // first column of the residual of the running activation (a_lo)
  static_assert(EP == HD || EP == 2 * HD, "padded embedding width is one or two hidden widths");
  constexpr bool WIDE_E = T::WIDE_E;
  constexpr int BM = T::BM, FB = T::FB, PB = T::PB, ROWB = T::ROWB;
  extern __shared__ __attribute__((aligned(1024))) char smem[];   // gemm() needs bits 4-9 of row bases clear
  char* X = smem;
  float* xs = (float*)(smem + T::OFF_XS);
  float* part = (float*)(smem + T::OFF_PART);
  float* gbs = (float*)(smem + T::OFF_GB);
  float* red = (float*)(smem + T::OFF_RED);

  const NetLayout& L = p.lay;
  const int tid = threadIdx.x;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);   // wave-uniform: weight/spill offsets stay in SGPRs
  static_assert(FB <= 2, "the LDS write addressing below assumes one or two feature blocks per wave");
  // Lane ids and the LDS write base.  They are RE-DERIVED from an opaque copy of threadIdx at every phase
  // (refresh()): derived lane constants otherwise stay live across the GEMMs, get spilled there and are
  // reloaded behind s_waitcnt vmcnt(0) in the middle of the MFMA stream (the kernel runs at a 128-VGPR budget).
  int lane, j, hi, lane16, xw;
  auto refresh = [&] {
    int t = tid;
    asm volatile("" : "+v"(t));
    lane = t & 63; j = lane & 31; hi = lane >> 5; lane16 = lane * 16;
    // byte offset of (row j, feature block w*FB, qp 0, first half) in the swizzled X tile: the swizzle, the
    // block-in-4 selector, qp (32 B) and the half (16 B) only meet in bits 4-7, so blocks differ by one XOR
    xw = j * ROWB + 8 * hi + (((j & 15) << 4) ^ (((w * FB) & 3) * 64)) + ((w * FB) >> 2) * 256;
  };
  refresh();
  const int64_t P = p.n_valid ? (int64_t)(*p.n_valid) * p.S : p.n_points_host;
  const int64_t n0 = (int64_t)blockIdx.x * BM;
  if (n0 >= P) return;
  const int nf = L.n_freqs;
  const float so = L.scale_output;
  ChainStamps TS(p.dbg, p.n_cu);   // phase time stamps of the -DISDF_DEBUG_HOOKS=1 build; empty inlines in the shipped kernel
  TS();
  // MODE.FP16_OVFL (hwreg 1, bit 23): float -> e4m3 / fp16 conversions SATURATE instead of producing NaN / inf.  Without it
  // v_cvt_scalef32_pk_fp8_f32 turns anything above 464 into NaN (tools/probes/fp8_cvt.hip); the e4m3 spill of P and GB below is
  // scaled so that this does not happen at any magnitude measured (isdf_common.h), and an outlier then clips instead of poisoning
  // the step.
  if (SP8) __builtin_amdgcn_s_setreg(1 | (23 << 6), 1);
  // Issue priority between the two workgroups of a CU: with equal priority the OLDER workgroup's waves win
  // VALU/MFMA arbitration all the way, finish ~33 us early and leave the younger one to run the rest alone at the poor
  // single-workgroup rate.  The second workgroup of a CU (dispatch round blockIdx / #CUs odd -- the CU count of the device the
  // launch goes to, not a constant: a partitioned (CPX) device has 32) therefore raises its priority for the two middle sweeps and
  // the pair finishes together.  Speed only: nothing depends on the dispatch order actually being round-robin.
  const bool genOdd = ((int)blockIdx.x / p.n_cu) & 1;
  auto PRIO = [&](int phase) {   // phase 0 fwd, 1 first reverse, 2 adjoint, 3 reverse
    if (genOdd) { if (phase == 1 || phase == 2) __builtin_amdgcn_s_setprio(1); else __builtin_amdgcn_s_setprio(0); }
  };
  PRIO(0);
  TS.wall(0);

  // element offsets of the four shadow weight sets inside the shadow buffer
  const int64_t setFwdA = L.setFwdA, setFwdB = L.setFwdB, setBwdA = L.setBwdA, setBwdB = L.setBwdB, setFwdLo = L.setFwdLo;
  // this tile's block of the spill buffer ([tile][tensor][BM*HD] bf16)
  uint16_t* spillTile = p.spill + TS.spill_tile() * p.sp.tileStride;
  const rsrc_t rsW = make_rsrc(p.shadow, 0x7fffffffu);
  const rsrc_t rsS = make_rsrc(spillTile, (uint32_t)(p.sp.tileStride * 2));   // loads (compiler-tracked)
  const i32x4 srdS = make_srd(spillTile, (uint32_t)(p.sp.tileStride * 2));     // stores (bstore16_nt)
  // byte offset of this wave's first piece of a spilled tensor (frag16 order, see frag16_off)
  auto sbase = [&](int64_t tensorOff) { return (int)(tensorOff * 2) + w * (FB * PB * 2) * 1024; };
  constexpr auto cidx = [](int fb, int pb, int qp) { return (fb * PB + pb) * 2 + qp; };   // piece-of-64-lanes index within the wave
  float* vecTile = MODE == 2 ? p.vec_part + (int64_t)blockIdx.x * p.vecStride : nullptr;
  const rsrc_t rsV = make_rsrc(vecTile, MODE == 2 ? (uint32_t)p.vecStride * 4u : 0u);
  (void)setFwdB; (void)setBwdB; (void)setFwdLo; (void)rsS; (void)rsV; (void)srdS;
  // feature index of (fb, qp) blocks: f0 = ubase(fb, qp) + 4*hi
  auto ubase = [&](int fb, int qp) { return w * (FB * 32) + fb * 32 + 16 * qp; };
  // 8 fp32 parameters of a per-unit vector (a bias, w_out) at flat offset vecBase: units ub + 4*hi + {0..3, 8..11}.  Units >= L.H are
  // padding (a hidden width below the tile width, NetLayout::H): they read as 0 and nothing past the real vector is touched.
  auto ld_params8 = [&](int vecBase, int ub, float (&o)[8]) {
    // a buffer descriptor over exactly this vector: every dword at or beyond unit L.H is out of range and reads as 0 (the range check of
    // a raw buffer load is per dword), so a narrower net needs no mask and no branch here
    const rsrc_t rv = make_rsrc(p.params + vecBase, (uint32_t)L.H * 4u);
    const u32x4 a = __builtin_amdgcn_raw_buffer_load_b128(rv, 16 * hi, ub * 4, 0);
    const u32x4 b = __builtin_amdgcn_raw_buffer_load_b128(rv, 16 * hi + 32, ub * 4, 0);
#pragma unroll
    for (int e = 0; e < 4; ++e) { o[e] = __uint_as_float(a[e]); o[4 + e] = __uint_as_float(b[e]); }
  };
  // per-workgroup partial of a bias / out-layer gradient entry: sum over the half-wave's 32 points
  // The 8 values of an accumulator block (features elemUniform + 4*hi + {0..3, 8..11}) go out in ONE store: after the butterflies
  // every lane holds all eight sums, lane j < 8 of each half keeps sum j and writes it to its feature.  (One store per VALUE made
  // these two-dword partials half of the kernel's store instructions, and the vector-memory path is its busiest unit.)
  auto vec_store8 = [&](float (&v)[8], int elemUniform) {
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = half_wave_sum(v[e]);
    const bool b0 = j & 1, b1 = j & 2, b2 = j & 4;
    const float t0 = b0 ? v[1] : v[0], t1 = b0 ? v[3] : v[2], t2 = b0 ? v[5] : v[4], t3 = b0 ? v[7] : v[6];
    const float u0 = b1 ? t1 : t0, u1 = b1 ? t3 : t2;
    const float r = b2 ? u1 : u0;   // = v[j & 7]
    if (j < 8) __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(r), rsV, 16 * hi + 4 * (j & 3) + 32 * (j >> 2), elemUniform * 4, 0);
  };

  // ------------------------------------------------------------------ PE stage
  // thread (pt, part): embedding.py:95-111.  Region 2 of X (cols HD..) gets the
  // embedding in the forward operand type.  (Until round 5 region 1 also took a copy in the spill type, staged for the spill
  // of the dW operand A_0: the dW kernel now rebuilds the embedding from x' itself, `pe_aux` below.)
  if constexpr (!WIDE_E && BM == 64) {
    // Round 5: a lane is a POINT, a wave takes directions w, w + 8, w + 16: the direction is wave-uniform, so a feature's column is
    // a scalar and its LDS address one v_xad of the lane's row base and swizzle, instead of a handful of per-lane integer operations
    // per 2-byte store.  Same arithmetic per value, so the same bits (sdf, d sdf / d x and the losses of a step came out bit-identical to
    // the previous library's, tools/train_ab_check.py).  Measured effect on this kernel: none (chain 184.9 -> 184.1 us same-box,
    // profiles/r05_ab_chain_pe_stages.txt) -- the 14 k cycles in front of the first GEMM are the two DEPENDENT memory round trips of
    // the prologue (n_valid, then the points), not this stage's instructions; kept because it is the simpler addressing.
    const int ln = tid & 63;
    const int64_t n = n0 + ln;
    float x0 = 0.f, x1 = 0.f, x2 = 0.f;
    if (n < P) { x0 = p.pts[n * 3]; x1 = p.pts[n * 3 + 1]; x2 = p.pts[n * 3 + 2]; }
    // transform_3D_grid (transform.py:287-304) then * scale (embedding.py:12-22)
    const float y0 = (L.T[0] * x0 + L.T[1] * x1 + L.T[2] * x2 + L.T[3]) * L.scale_input;
    const float y1 = (L.T[4] * x0 + L.T[5] * x1 + L.T[6] * x2 + L.T[7]) * L.scale_input;
    const float y2 = (L.T[8] * x0 + L.T[9] * x1 + L.T[10] * x2 + L.T[11]) * L.scale_input;
    typedef typename Op<F16>::e opT;
    const int sw = (ln & 15) << 4;
    char* row = X + ln * ROWB;
    auto put = [&](int feat, float v) {
      *(opT*)(row + (((HD + feat) * 2) ^ sw)) = (opT)v;
      if (X2ALL) *(opT*)(row + (((LO + HD + feat) * 2) ^ sw)) = (opT)(v - (float)(opT)v);   // emb_lo
    };
    if (w == T::NW - 1) {   // (the wave with the fewest directions)
      xs[ln * 4] = y0; xs[ln * 4 + 1] = y1; xs[ln * 4 + 2] = y2;
      put(0, y0); put(1, y1); put(2, y2);
      for (int f = L.E; f < EP; ++f) put(f, 0.f);
    }
    for (int d = w; d < N_DIRS; d += T::NW) {
      const float proj = y0 * kDirs[0][d] + y1 * kDirs[1][d] + y2 * kDirs[2][d];
      float fr = 1.f;
      for (int f = 0; f < nf; ++f) {
        const float xb = proj * fr;
        put(3 + d * nf + f, __sinf(xb));
        put(3 + N_DIRS * nf + d * nf + f, __sinf(xb + kHalfPi));
        fr *= 2.f;
      }
    }
  } else
  {
    // a wave = (BM / NW points) x (direction slices): rows are 1 KB apart, i.e. 8 banks -- 64 points per wave was 8-way conflicted
    const int pt = (tid % (BM / T::NW)) + (BM / T::NW) * (tid / 64), prt = (tid % 64) / (BM / T::NW);
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
      if (X2ALL) *(opT*)(row + swz(pt, (LO + HD + feat) * 2)) = (opT)(v - (float)(opT)v);   // emb_lo
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
  refresh();

  // ------------------------------------------------------------------ forward
  f32x16 acc[FB][PB];
  const int rbW = w * FB;  // first 32-row block of this wave
  auto wptr = [&](int64_t set, int64_t matOff, int kp) {
    return WRef{(int)((set + matOff) * 2) + rbW * (kp / 16) * 1024, (kp / 16) * 1024};
  };
  auto fwdW = [&](int64_t set, int li) {   // forward-orientation matrix of layer li (K = EP | HD+EP | HD)
    return wptr(set, L.fwdMat[li], li == 0 ? EP : (li == L.cat ? HD + EP : HD));
  };
  WChunk<FB, T::CKF> wq;   // the weight-fragment registers of the running gemm
  WChunk<FB, T::CKF / 2> wq2;   // ... of a two-operand gemm: half the k-steps per chunk, twice the activation registers
  // iterate the wave's accumulator as (fb, pb, qp) blocks of 8 values:
  // values v[0..3] -> features f0..f0+3, v[4..7] -> f0+8..f0+11, point row = pb*32+j
  auto for_blocks2 = [&](auto&& fn, auto&& tail) {
#pragma unroll
    for (int fb = 0; fb < FB; ++fb)
#pragma unroll
      for (int qp = 0; qp < 2; ++qp) {
#pragma unroll
        for (int pb = 0; pb < PB; ++pb) fn(fb, pb, qp, pb * 32 + j);
        tail(fb, qp);
      }
  };
  auto for_blocks = [&](auto&& fn) { for_blocks2(fn, [](int, int) {}); };
  // A spilled tile is re-read by the same lanes that wrote it; the reads are
  // issued BEFORE the layer's GEMM (prefetch) and consumed in its epilogue.
  struct Pre { uint4 v[FB][2][PB]; };
  auto prefetch = [&](int64_t tensorOff, Pre& pr) {
    const int sb = sbase(tensorOff);
#pragma unroll
    for (int fb = 0; fb < FB; ++fb)
#pragma unroll
      for (int qp = 0; qp < 2; ++qp)
#pragma unroll
        for (int pb = 0; pb < PB; ++pb)
          pr.v[fb][qp][pb] = bload16<spill_load_aux(HD)>(rsS, lane16 + (cidx(fb, pb, qp) & 3) * 1024, sb + (cidx(fb, pb, qp) >> 2) * 4096);
  };
  auto load_tile8 = [&](const Pre& pr, int fb, int pb, int qp, float (&o)[8]) {
    const uint4 u = pr.v[fb][qp][pb];
    if constexpr (BW) {
      const f16x4 a = __builtin_bit_cast(f16x4, make_uint2(u.x, u.y)), b = __builtin_bit_cast(f16x4, make_uint2(u.z, u.w));
#pragma unroll
      for (int e = 0; e < 4; ++e) { o[e] = (float)a[e]; o[4 + e] = (float)b[e]; }
    } else {
      float a[4], b[4];
      unpack4_bf16(make_uint2(u.x, u.y), a); unpack4_bf16(make_uint2(u.z, u.w), b);
#pragma unroll
      for (int e = 0; e < 4; ++e) { o[e] = a[e]; o[4 + e] = b[e]; }
    }
  };
  auto store_tile8 = [&](int64_t tensorOff, int fb, int pb, int qp, const float (&v)[8]) {
    const uint2 a = pack4<BW>(v[0], v[1], v[2], v[3]), b = pack4<BW>(v[4], v[5], v[6], v[7]);
    bstore16_nt<spill_store_nt(HD)>(make_uint4(a.x, a.y, b.x, b.y), srdS, lane16, sbase(tensorOff), cidx(fb, pb, qp));
  };
  // tensors that only the dW kernel re-reads (GB, ZB): default cache policy since round 4 (chain -2.3 %, dW unchanged); A and P,
  // which the chain kernel itself re-reads, follow spill_store_nt(HD) (chain_dev.h: default policy too for the 256-wide nets, whose stack fits the cache)
  auto store_tile8_dw = [&](int64_t tensorOff, int fb, int pb, int qp, const float (&v)[8]) {
    const uint2 a = pack4<BW>(v[0], v[1], v[2], v[3]), b = pack4<BW>(v[4], v[5], v[6], v[7]);
    bstore16_nt<false>(make_uint4(a.x, a.y, b.x, b.y), srdS, lane16, sbase(tensorOff), cidx(fb, pb, qp));
  };
  // ZB is written by the LAST sweep and read only by the dW kernel: for the 256-wide nets it goes out non-temporal, which leaves the
  // Infinity Cache to the tensors the sweeps re-read (chain -1.7 us, dW +0.9: profiles/r06_cache_policy.txt)
  auto store_tile8_zb = [&](int64_t tensorOff, int fb, int pb, int qp, const float (&v)[8]) {
    const uint2 a = pack4<BW>(v[0], v[1], v[2], v[3]), b = pack4<BW>(v[4], v[5], v[6], v[7]);
    bstore16_nt<spill_zb_nt(HD)>(make_uint4(a.x, a.y, b.x, b.y), srdS, lane16, sbase(tensorOff), cidx(fb, pb, qp));
  };
  // ---- e4m3 spills of P and GB (SpillLayout): 16 bytes per lane = its 8 values of point block 0, then of point block 1.  Encode
  // divides by `scale`, decode multiplies (v_cvt_scalef32_pk_fp8_f32 / _f32_fp8: one instruction per two values, the scale is free).
  static_assert(PB == 2, "a frag8 piece holds the two point blocks of a tile");
  struct Pre8 { uint4 v[FB][2]; };
  auto sbase8 = [&](int64_t tensorOff) { return (int)(tensorOff * 2) + w * (FB * 2) * 1024; };
  auto prefetch8 = [&](int64_t tensorOff, Pre8& pr) {
    const int sb = sbase8(tensorOff);
#pragma unroll
    for (int fb = 0; fb < FB; ++fb)
#pragma unroll
      for (int qp = 0; qp < 2; ++qp)
        pr.v[fb][qp] = bload16<spill_load_aux(HD)>(rsS, lane16 + ((fb * 2 + qp) & 3) * 1024, sb + ((fb * 2 + qp) >> 2) * 4096);
  };
  auto load_f8 = [&](const Pre8& pr, int fb, int pb, int qp, float scale, float (&o)[8]) {
    const uint4 u = pr.v[fb][qp];
    const int lo = (int)(pb ? u.z : u.x), hi2 = (int)(pb ? u.w : u.y);
    const f32x2 a = __builtin_amdgcn_cvt_scalef32_pk_f32_fp8(lo, scale, false), b = __builtin_amdgcn_cvt_scalef32_pk_f32_fp8(lo, scale, true);
    const f32x2 c = __builtin_amdgcn_cvt_scalef32_pk_f32_fp8(hi2, scale, false), d = __builtin_amdgcn_cvt_scalef32_pk_f32_fp8(hi2, scale, true);
    o[0] = a[0]; o[1] = a[1]; o[2] = b[0]; o[3] = b[1]; o[4] = c[0]; o[5] = c[1]; o[6] = d[0]; o[7] = d[1];
  };
  uint2 held8 = make_uint2(0u, 0u);   // point block 0's bytes of the piece being built (for_blocks visits pb = 0, 1 back to back)
  auto store_f8 = [&](int64_t tensorOff, int fb, int pb, int qp, const float (&v)[8], float scale, auto nt) {
    typedef short s16x2 __attribute__((ext_vector_type(2)));
    s16x2 w0 = {0, 0}, w1 = {0, 0};
    w0 = __builtin_amdgcn_cvt_scalef32_pk_fp8_f32(w0, v[0], v[1], scale, false);
    w0 = __builtin_amdgcn_cvt_scalef32_pk_fp8_f32(w0, v[2], v[3], scale, true);
    w1 = __builtin_amdgcn_cvt_scalef32_pk_fp8_f32(w1, v[4], v[5], scale, false);
    w1 = __builtin_amdgcn_cvt_scalef32_pk_fp8_f32(w1, v[6], v[7], scale, true);
    const uint32_t u0 = __builtin_bit_cast(uint32_t, w0), u1 = __builtin_bit_cast(uint32_t, w1);
    if (pb == 0) held8 = make_uint2(u0, u1);
    else bstore16_nt<decltype(nt)::value>(make_uint4(held8.x, held8.y, u0, u1), srdS, lane16, sbase8(tensorOff), fb * 2 + qp);
  };
  // sigma'(z) of a layer for the backward epilogues, re-derived from the bf16 activation tile (no sigma' tensor is stored)
  auto load_s1 = [&](const Pre& pr, int fb, int pb, int qp, float (&o)[8]) {
    load_tile8(pr, fb, pb, qp, o);
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = s1_from_a(o[e]);
  };
  // ... and t = 1 - sigma' = exp(-beta a) with it: the injected second-order term needs (1 - sigma') / sigma', and 1 - (1 - t) is the
  // less accurate (and one instruction longer) way to get t back
  auto load_s1t = [&](const Pre& pr, int fb, int pb, int qp, float (&o)[8], float (&t)[8]) {
    load_tile8(pr, fb, pb, qp, o);
#pragma unroll
    for (int e = 0; e < 8; ++e) { t[e] = __builtin_amdgcn_exp2f(-kC1 * o[e]); o[e] = 1.f - t[e]; }
  };
  auto put_x = [&](bool f16, int fb, int pb, int qp, const float (&v)[8], int colElemBase) {
    uint2 a, b;
    if (f16) { a = pack4<true>(v[0], v[1], v[2], v[3]); b = pack4<true>(v[4], v[5], v[6], v[7]); }
    else { a = pack4<false>(v[0], v[1], v[2], v[3]); b = pack4<false>(v[4], v[5], v[6], v[7]); }
    const int lb = (xw ^ (64 * fb + 32 * qp)) + colElemBase * 2 + pb * 32 * ROWB;   // see refresh()
    *(uint2*)(X + lb) = a;           // features f0 .. f0+3
    *(uint2*)(X + (lb ^ 16)) = b;    // features f0+8 .. f0+11
  };

  // A wave's own pieces of one tensor, parked in region 2 of the tile across a few units when that region is idle (16 B per lane,
  // the lane that parks a piece is the lane that takes it back: no barrier needed for it).  Saves the HBM re-read of a tensor whose
  // consumer is only a few units away.
  auto park_addr = [&](int c) {
    const int off = (c * 64 + lane) * 16;                  // byte offset inside the wave's share of region 2
    constexpr int RB = T::R2 * 2;                          // bytes of region 2 in one tile row
    return (w * (BM / T::NW) + off / RB) * ROWB + HD * 2 + off % RB;
  };
  static_assert(FB * PB * 2 * 1024 <= (BM / T::NW) * T::R2 * 2, "a wave's pieces of one tensor fit its rows of region 2");
  auto park_tile8 = [&](int fb, int pb, int qp, const float (&v)[8]) {
    const uint2 a = pack4<BW>(v[0], v[1], v[2], v[3]), b = pack4<BW>(v[4], v[5], v[6], v[7]);
    *(uint4*)(X + park_addr(cidx(fb, pb, qp))) = make_uint4(a.x, a.y, b.x, b.y);
  };
  float rawp[PB];
#pragma unroll
  for (int pb = 0; pb < PB; ++pb) rawp[pb] = 0.f;

  // region 1 <- fp16(v), region 2 <- fp16(v - fp16(v)): the operand pair of a compensated layer (OPER 2).  The residual is formed against
  // the STORED half (round 6; until then the value was converted twice, once for the store and once for the residual, and the compiler
  // is free to round the two differently -- one rounding through v_fma_mix*, two through v_mul + v_cvt: a residual against a
  // differently rounded half is off by an fp16 ulp of the value; fwd_pair.hip has done it this way since round 5)
  auto put_x_hilo = [&](int fb, int pb, int qp, const float (&v)[8]) {
    f16x4 ha, hb;
    float r[8];
#pragma unroll
    for (int e = 0; e < 4; ++e) { ha[e] = (_Float16)v[e]; hb[e] = (_Float16)v[4 + e]; }
#pragma unroll
    for (int e = 0; e < 4; ++e) { r[e] = v[e] - (float)ha[e]; r[4 + e] = v[4 + e] - (float)hb[e]; }
    const int lb = (xw ^ (64 * fb + 32 * qp)) + pb * 32 * ROWB;   // see put_x
    *(uint2*)(X + lb) = __builtin_bit_cast(uint2, ha);
    *(uint2*)(X + (lb ^ 16)) = __builtin_bit_cast(uint2, hb);
    put_x(true, fb, pb, qp, r, LO);
  };
  for (int li = 0; li < L.L; ++li) {
    refresh();
    // compensated layers (OPER 2): the cat layer adds W_lo[:, HD:] emb (the residual of its embedding columns; the one of
    // its hidden columns moves sdf by 3e-5 and is skipped), the layers past it W_lo a and W a_lo (a_lo sits in region 2,
    // which the forward pass no longer needs once the cat layer has consumed the embedding).  Numpy model of these
    // numerics vs the reference at BASELINE size: tests/precision_model.py, tools/studies/split_precision_study.py.
    const bool comp = X2 && (X2ALL || li >= L.cat);
    if (li == 0) {
      if (X2ALL) {   // W (emb + emb_lo) in one pass over W, then W_lo emb
        gemm<F16, EP / 16, FB, PB, ROWB, T::CKF / 2, true, true>(acc, wq2, rsW, fwdW(setFwdA, li), X, HD * 2, lane, [] {}, (LO + HD) * 2);
        gemm<F16, EP / 16, FB, PB, ROWB, T::CKF>(acc, wq, rsW, fwdW(setFwdLo, li), X, HD * 2, lane, [] {});
      } else {
        gemm<F16, EP / 16, FB, PB, ROWB, T::CKF, true>(acc, wq, rsW, fwdW(setFwdA, li), X, HD * 2, lane, [] {});
      }
    } else if (li == L.cat) {
      if (X2ALL) {   // W ([a | emb] + [a_lo | emb_lo]) in one pass over W, then W_lo [a | emb]
        gemm<F16, (HD + EP) / 16, FB, PB, ROWB, T::CKF / 2, true, true>(acc, wq2, rsW, fwdW(setFwdA, li), X, 0, lane, [] {}, LO * 2);
        gemm<F16, (HD + EP) / 16, FB, PB, ROWB, T::CKF>(acc, wq, rsW, fwdW(setFwdLo, li), X, 0, lane, [] {});
      } else {
        gemm<F16, (HD + EP) / 16, FB, PB, ROWB, T::CKF, true>(acc, wq, rsW, fwdW(setFwdA, li), X, 0, lane, [] {});
      }
      if (!X2ALL && comp) {
        WRef wl = fwdW(setFwdLo, li);
        wl.soff += (HD / 16) * 1024;   // k-steps HD/16 .. of every row block: the embedding columns
        gemm<F16, EP / 16, FB, PB, ROWB, T::CKF>(acc, wq, rsW, wl, X, HD * 2, lane, [] {});
      }
    } else {
      if (comp) {   // W (a + a_lo) in one pass over W, then W_lo a
        gemm<F16, HD / 16, FB, PB, ROWB, T::CKF / 2, true, true>(acc, wq2, rsW, fwdW(setFwdA, li), X, 0, lane, [] {}, LO * 2);
        gemm<F16, HD / 16, FB, PB, ROWB, T::CKF>(acc, wq, rsW, fwdW(setFwdLo, li), X, 0, lane, [] {});
      } else {
        gemm<F16, HD / 16, FB, PB, ROWB, T::CKF, true>(acc, wq, rsW, fwdW(setFwdA, li), X, 0, lane, [] {});
      }
    }
    TS();
    lds_barrier();  // all waves finished reading the tile
    TS();
    refresh();
    const bool last = li == L.L - 1;
    if (!last) {
      float bv[8];
      for_blocks([&](int fb, int pb, int qp, int row) {
        if (pb == 0) ld_params8(L.offB[li], ubase(fb, qp), bv);
        float a[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) a[e] = softplus_f(acc[fb][pb][8 * qp + e] + bv[e]);
        if (MODE >= 1) store_tile8(p.sp.A[li + 1], fb, pb, qp, a);
        if (X2 && (X2ALL || li + 1 > L.cat)) put_x_hilo(fb, pb, qp, a);   // the NEXT layer reads a_lo
        else put_x(F16, fb, pb, qp, a, 0);
      });
    } else {
      float bv[8], wv[8];
      for_blocks([&](int fb, int pb, int qp, int row) {
        if (pb == 0) {
          ld_params8(L.offB[li], ubase(fb, qp), bv);
          ld_params8(L.offWout, ubase(fb, qp), wv);
        }
        float a[8], pl[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          float s1;
          a[e] = softplus_s1(acc[fb][pb][8 * qp + e] + bv[e], s1);
          pl[e] = so * wv[e] * s1;   // p_L = q_L * sigma'(z_L), q_L = so * w_out
        }
        // w_out . a over the block's 8 units, as four packed FMAs of ELEMENT PAIRS (wv[e], wv[e+1]) * (a[e], a[e+1]).  Written out
        // because of what the vectoriser makes of the scalar `rawp[pb] += wv[e] * a[e]`: it pairs the two point blocks and broadcasts
        // wv[e], i.e. v_pk_fma_f32 with op_sel:[0,1,0] for the odd e -- a form that, with two workgroups on a CU, intermittently
        // dropped the low half's product in lanes 48-63 on MI355X (isdf_amd/isa_lint.py rule 1 has the measurements; the build refuses
        // a library that contains it).
        f32x2 r2 = {0.f, 0.f};
#pragma unroll
        for (int e = 0; e < 8; e += 2) r2 += f32x2{wv[e], wv[e + 1]} * f32x2{a[e], a[e + 1]};
        rawp[pb] += r2[0] + r2[1];
        if (MODE >= 1) {
          store_tile8(p.sp.A[li + 1], fb, pb, qp, a);
          put_x(F16, fb, pb, qp, pl, 0);
          if (MODE == 2) store_tile8(p.sp.P[li], fb, pb, qp, pl);   // (the top layer's P stays 16-bit in either spill format: SpillLayout)
        }
      });
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
  PRIO(1);
  for (int li = L.L - 1; li >= 1; --li) {
    Pre preA;   // a_{li-1}: sigma'(z_{li-1}) is re-derived from it
    refresh();
    gemm<F16, HD / 16, FB, PB, ROWB, T::CKF, true>(acc, wq, rsW, wptr(setBwdA, L.bwdMat[li], HD), X, 0, lane,
                                             [&] { prefetch(p.sp.A[li], preA); });
    TS();
    lds_barrier();
    TS();
    refresh();
    const bool toR2 = (li - 1 == L.cat);
    for_blocks([&](int fb, int pb, int qp, int row) {
      float a[8], pv[8];
      load_s1(preA, fb, pb, qp, a);   // a[] = sigma'
#pragma unroll
      for (int e = 0; e < 8; ++e) pv[e] = acc[fb][pb][8 * qp + e] * a[e];
      put_x(F16, fb, pb, qp, pv, 0);
      if (toR2) put_x(F16, fb, pb, qp, pv, HD);
      if (MODE == 2) {
        if constexpr (P8) store_f8(p.sp.P[li - 1], fb, pb, qp, pv, kSpillPScale, std::integral_constant<bool, spill_store_nt(HD)>{});
        else store_tile8(p.sp.P[li - 1], fb, pb, qp, pv);
      }
    });
    TS();
    lds_barrier();
    TS();
  }
  // Eg = [W_in^T | W_cat[:,HD:]^T] [p_0 ; p_cat]   (rows = embedding features)
  // the loss stage's per-ray inputs are requested behind the G gemm so that its single working wave
  // does not start with a dependent HBM round trip
  float li_bnd = 0.f, li_c[3] = {0.f, 0.f, 0.f}, li_dz[2] = {0.f, 0.f}, li_t[3] = {0.f, 0.f, 0.f}, li_n[3] = {0.f, 0.f, 0.f};
  auto loss_inputs = [&] {
    if (MODE != 2 || tid >= BM) return;
    const int64_t n = n0 + tid;
    if (n >= P) return;
    const int64_t ray = (int64_t)((uint32_t)n / (uint32_t)p.S);   // 32-bit division: max_rays * S < 2^31 (capi.hip), and this wave
    if (p.loss.bounds_method == 0) {  // loss.py:13-22                //   sits on the tile's critical path between the sweeps
      li_c[0] = p.dirsC[ray * 3]; li_c[1] = p.dirsC[ray * 3 + 1]; li_c[2] = p.dirsC[ray * 3 + 2];
      li_dz[0] = p.depth[ray]; li_dz[1] = p.z_vals[n];
      li_t[0] = p.dirsW[ray * 3]; li_t[1] = p.dirsW[ray * 3 + 1]; li_t[2] = p.dirsW[ray * 3 + 2];
    } else {
      li_bnd = p.pc_bounds[n];
      li_t[0] = p.pc_grad_vec[n * 3]; li_t[1] = p.pc_grad_vec[n * 3 + 1]; li_t[2] = p.pc_grad_vec[n * 3 + 2];
    }
    if (p.normals) { li_n[0] = p.normals[ray * 3]; li_n[1] = p.normals[ray * 3 + 1]; li_n[2] = p.normals[ray * 3 + 2]; }
  };
  refresh();
  if constexpr (!WIDE_E) {
  gemm<F16, (2 * HD) / 16, FB, PB, ROWB, T::CKF, true>(acc, wq, rsW, wptr(setBwdA, L.bwdG, 2 * HD), X, 0, lane, loss_inputs);
  TS();   // (development build: G gemm done)
  // g_x' = J_pe^T Eg.  Eg goes through the (now idle) X tile as fp32 [BM][HD] so the contraction can run in
  // the PE stage's (point, direction-slice) mapping: 2*nf sin/cos per direction per thread and wave-uniform
  // direction constants, instead of one cos + index arithmetic per accumulator element (which took
  // ~45 k cycles per tile, profiles/r01_chain_timeline_final.txt stamp 45).
  lds_barrier();   // every wave finished reading X for the G gemm
  refresh();
#pragma unroll
  for (int fb = 0; fb < FB; ++fb)
#pragma unroll
    for (int pb = 0; pb < PB; ++pb) {
      const int row = pb * 32 + j;
#pragma unroll
      for (int rq = 0; rq < 4; ++rq) {
        const int f0 = w * (FB * 32) + fb * 32 + 8 * rq + 4 * hi;
        *(float4*)(X + row * ROWB + swz(row, f0 * 4)) =
            make_float4(acc[fb][pb][4 * rq], acc[fb][pb][4 * rq + 1], acc[fb][pb][4 * rq + 2], acc[fb][pb][4 * rq + 3]);
      }
    }
  lds_barrier();
  TS();   // (Eg staged in LDS)
  if constexpr (BM == 64) {
    // lane = point, wave = direction slice w (the partial sum of slice w lands where slice `prt` of the mapping below put it: same
    // partial sums, same summation order in the loss stage)
    static_assert(T::NPART == T::NW, "one partial g per wave");
    const int ln = tid & 63, sw = (ln & 15) << 4;
    const float y0 = xs[ln * 4], y1 = xs[ln * 4 + 1], y2 = xs[ln * 4 + 2];
    const char* row = X + ln * ROWB;
    auto eg = [&](int feat) { return *(const float*)(row + ((feat * 4) ^ sw)); };
    float g0 = 0.f, g1 = 0.f, g2 = 0.f;
    if (w == 0) { g0 = eg(0); g1 = eg(1); g2 = eg(2); }
    const int half = N_DIRS * nf;
    for (int d = w; d < N_DIRS; d += T::NW) {
      const float dx = kDirs[0][d], dy = kDirs[1][d], dz = kDirs[2][d];
      const float proj = y0 * dx + y1 * dy + y2 * dz;
      float fr = 1.f, c = 0.f;
      for (int f = 0; f < nf; ++f) {
        const float xb = proj * fr;
        // d sin(xb)/d proj = cos(xb) fr ;  d sin(xb + pi/2)/d proj = cos(xb + pi/2) fr
        c += (__cosf(xb) * eg(3 + d * nf + f) + __cosf(xb + kHalfPi) * eg(3 + half + d * nf + f)) * fr;
        fr *= 2.f;
      }
      g0 += c * dx; g1 += c * dy; g2 += c * dz;
    }
    float* dst = part + (w * BM + ln) * 4;
    dst[0] = g0; dst[1] = g1; dst[2] = g2;
  } else
  {
    // a wave = (BM / NW points) x (direction slices): rows are 1 KB apart, i.e. 8 banks -- 64 points per wave was 8-way conflicted
    const int pt = (tid % (BM / T::NW)) + (BM / T::NW) * (tid / 64), prt = (tid % 64) / (BM / T::NW);
    constexpr int NPART = T::NPART;
    const float y0 = xs[pt * 4], y1 = xs[pt * 4 + 1], y2 = xs[pt * 4 + 2];
    const char* row = X + pt * ROWB;
    auto eg = [&](int feat) { return *(const float*)(row + swz(pt, feat * 4)); };
    float g0 = 0.f, g1 = 0.f, g2 = 0.f;
    if (prt == 0) { g0 = eg(0); g1 = eg(1); g2 = eg(2); }
    const int half = N_DIRS * nf;
    for (int d = prt; d < N_DIRS; d += NPART) {
      const float dx = kDirs[0][d], dy = kDirs[1][d], dz = kDirs[2][d];
      const float proj = y0 * dx + y1 * dy + y2 * dz;
      float fr = 1.f, c = 0.f;
      for (int f = 0; f < nf; ++f) {
        const float xb = proj * fr;
        // d sin(xb)/d proj = cos(xb) fr ;  d sin(xb + pi/2)/d proj = cos(xb + pi/2) fr
        c += (__cosf(xb) * eg(3 + d * nf + f) + __cosf(xb + kHalfPi) * eg(3 + half + d * nf + f)) * fr;
        fr *= 2.f;
      }
      g0 += c * dx; g1 += c * dy; g2 += c * dz;
    }
    float* dst = part + (prt * BM + pt) * 4;
    dst[0] = g0; dst[1] = g1; dst[2] = g2;
  }
  lds_barrier();
  TS();   // (J_pe^T contraction done)
  } else {
    // EP = 2 HD: the X tile cannot hold fp32 [BM][EP] next to nothing, and the G operands (p_0 | p_cat in the
    // first 2 HD columns) must survive the first row half.  So: one GEMM per row half of G (HD embedding
    // features each), and per 32-point block the accumulator goes through a separate fp32 [32][HD] LDS buffer
    // and is contracted by 32 points x NPART direction slices; partial g sums accumulate in part[].
    char* EG = smem + T::OFF_EG;
    constexpr int NPART = T::NPART;
    for (int q = tid; q < NPART * BM; q += T::NW * 64) ((float4*)part)[q] = make_float4(0.f, 0.f, 0.f, 0.f);
    const int half = N_DIRS * nf;
#pragma unroll
    for (int rh = 0; rh < EP / HD; ++rh) {
      if (rh > 0) refresh();
      WRef wg = wptr(setBwdA, L.bwdG, 2 * HD);
      wg.soff += rh * (HD / 32) * ((2 * HD) / 16) * 1024;
      if (rh == 0) gemm<F16, (2 * HD) / 16, FB, PB, ROWB, T::CKF, true>(acc, wq, rsW, wg, X, 0, lane, loss_inputs);
      else gemm<F16, (2 * HD) / 16, FB, PB, ROWB, T::CKF, true>(acc, wq, rsW, wg, X, 0, lane, [] {});
      refresh();
#pragma unroll
      for (int pb = 0; pb < PB; ++pb) {
        lds_barrier();   // previous contraction finished reading EG
#pragma unroll
        for (int fb = 0; fb < FB; ++fb)
#pragma unroll
          for (int rq = 0; rq < 4; ++rq) {
            const int f0 = w * (FB * 32) + fb * 32 + 8 * rq + 4 * hi;
            *(float4*)(EG + j * (HD * 4) + swz(j, f0 * 4)) =
                make_float4(acc[fb][pb][4 * rq], acc[fb][pb][4 * rq + 1], acc[fb][pb][4 * rq + 2], acc[fb][pb][4 * rq + 3]);
          }
        lds_barrier();
        const int pl = tid & 31, prt = tid >> 5, pt = pb * 32 + pl;
        const float y0 = xs[pt * 4], y1 = xs[pt * 4 + 1], y2 = xs[pt * 4 + 2];
        const char* row = EG + pl * (HD * 4);
        const int lo = rh * HD;
        // feature `feat` of this row half (0 outside it)
        auto eg = [&](int feat) { return (unsigned)(feat - lo) < (unsigned)HD ? *(const float*)(row + swz(pl, (feat - lo) * 4)) : 0.f; };
        float g0 = 0.f, g1 = 0.f, g2 = 0.f;
        if (prt == 0 && rh == 0) { g0 = eg(0); g1 = eg(1); g2 = eg(2); }
        for (int d = prt; d < N_DIRS; d += NPART) {
          const float dx = kDirs[0][d], dy = kDirs[1][d], dz = kDirs[2][d];
          const float proj = y0 * dx + y1 * dy + y2 * dz;
          float fr = 1.f, c = 0.f;
          for (int f = 0; f < nf; ++f) {
            const float xb = proj * fr;
            c += (__cosf(xb) * eg(3 + d * nf + f) + __cosf(xb + kHalfPi) * eg(3 + half + d * nf + f)) * fr;
            fr *= 2.f;
          }
          g0 += c * dx; g1 += c * dy; g2 += c * dz;
        }
        float* dst = part + (prt * BM + pt) * 4;   // (prt, pt) is owned by this thread in every sub-pass
        dst[0] += g0; dst[1] += g1; dst[2] += g2;
      }
    }
    lds_barrier();
  }

  // ------------------------------------------------------------------ loss + adjoints (one thread per point)
  float lsum[4] = {0.f, 0.f, 0.f, 0.f};
  if (tid < BM) {
    const int64_t n = n0 + tid;
    float e0 = 0.f, e1 = 0.f, e2 = 0.f;
#pragma unroll
    for (int k = 0; k < T::NPART; ++k) {
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
      const int64_t ray = (int64_t)((uint32_t)n / (uint32_t)p.S);
      const int s = (int)(n - ray * p.S);
      float bnd = li_bnd, tx = li_t[0], ty = li_t[1], tz = li_t[2];   // bound and target gradient direction
      if (lc.bounds_method == 0) {
        bnd = sqrtf(li_c[0] * li_c[0] + li_c[1] * li_c[1] + li_c[2] * li_c[2]) * (li_dz[0] - li_dz[1]);
        tx = -tx; ty = -ty; tz = -tz;
      }
      if (p.normals && (s == 0 || tx != tx)) {  // surface sample, or NaN target (trainer.py:823-824)
        tx = li_n[0]; ty = li_n[1]; tz = li_n[2];
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
      // what the dW kernel rebuilds its two embedding-shaped operands from (the embedding and Ebar = J_pe gbar): 6 floats per point
      // instead of 2 x EP 16-bit values.  Every row of the tile is written (rows past the last point: x' of the origin, gbar = 0).
      // ... and the point's GB spill scale s_G = 2^k * (power of two above |gbar'|_inf) (SpillLayout): the adjoint sweep below
      // encodes with it, the reverse sweep and the dW kernel decode with it
      const float gmax = fmaxf(fmaxf(fabsf(gbs[tid * 4]), fabsf(gbs[tid * 4 + 1])), fabsf(gbs[tid * 4 + 2]));
      const uint32_t ge = __float_as_uint(gmax) & 0x7f800000u;
      // gbar = 0 (no eikonal / normal term at this point, or a padding row): GB is exactly 0 whatever the scale; subnormal or
      // non-finite maxima get a harmless one as well
      const float sG = (ge < (27u << 23) || ge >= (200u << 23)) ? 1.f : __uint_as_float(ge + ((uint32_t)(1 + spill_gb_shift(nf)) << 23));
      xs[tid * 4 + 3] = sG;
      float4* aux = (float4*)(p.pe_aux + (n0 + tid) * 8);
      aux[0] = make_float4(xs[tid * 4], xs[tid * 4 + 1], xs[tid * 4 + 2], 0.f);
      aux[1] = make_float4(gbs[tid * 4], gbs[tid * 4 + 1], gbs[tid * 4 + 2], sG);
    }
  }
  if (MODE != 2) return;
  TS();   // (loss + adjoints done)
  if (tid < BM) {  // per-wave loss / sbar sums -> LDS, one thread combines (deterministic)
    float v5[5] = {lsum[0], lsum[1], lsum[2], lsum[3], gbs[tid * 4 + 3]};
#pragma unroll
    for (int k = 0; k < 5; ++k) {
      float v = half_wave_sum(v5[k]);
      v += __shfl_xor(v, 32, 64);
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

  TS();   // (loss sums written)
  // ------------------------------------------------------------------ Ebar = J_pe gbar  -> region 2 (bf16)
  if (tid < HD / 4) {   // w_out for the top epilogue (0 for the padding units of a narrower net: out of the descriptor's range)
    const u32x4 w4 = __builtin_amdgcn_raw_buffer_load_b128(make_rsrc(p.params + L.offWout, (uint32_t)L.H * 4u), tid * 16, 0, 0);
    const float4 wv4 = make_float4(__uint_as_float(w4[0]), __uint_as_float(w4[1]), __uint_as_float(w4[2]), __uint_as_float(w4[3]));
    ((float4*)part)[tid] = wv4;
  }
  if constexpr (!WIDE_E && BM == 64) {
    // lane = point, wave = direction slice (see the PE stage)
    const int ln = tid & 63, sw = (ln & 15) << 4;
    const float y0 = xs[ln * 4], y1 = xs[ln * 4 + 1], y2 = xs[ln * 4 + 2];
    const float b0 = gbs[ln * 4], b1 = gbs[ln * 4 + 1], b2 = gbs[ln * 4 + 2];
    char* row = X + ln * ROWB;
    auto put = [&](int feat, float v) { *(spillT*)(row + (((HD + feat) * 2) ^ sw)) = (spillT)v; };
    if (w == T::NW - 1) {
      put(0, b0); put(1, b1); put(2, b2);
      for (int f = L.E; f < EP; ++f) put(f, 0.f);
    }
    for (int d = w; d < N_DIRS; d += T::NW) {
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
  } else
  {
    // a wave = (BM / NW points) x (direction slices): rows are 1 KB apart, i.e. 8 banks -- 64 points per wave was 8-way conflicted
    const int pt = (tid % (BM / T::NW)) + (BM / T::NW) * (tid / 64), prt = (tid % 64) / (BM / T::NW);
    constexpr int NPART = (T::NW * 64) / BM;
    const float y0 = xs[pt * 4], y1 = xs[pt * 4 + 1], y2 = xs[pt * 4 + 2];
    const float b0 = gbs[pt * 4], b1 = gbs[pt * 4 + 1], b2 = gbs[pt * 4 + 2];
    char* row = X + pt * ROWB;
    auto put = [&](int feat, float v) { *(spillT*)(row + swz(pt, (HD + feat) * 2)) = (spillT)v; };
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
  TS();   // (Ebar in region 2)
  TS();   // (development build: the stamp that closed the Ebar spill until round 5 -- the dW kernel rebuilds Ebar from pe_aux)

  // ------------------------------------------------------------------ adjoint of the first reverse sweep (upward)
  // The top layer's epilogue also IS the top of the ordinary reverse sweep (zbar_L needs only a_L, the
  // injection just computed and sbar*w_out), so it never leaves registers and the reverse sweep
  // starts at layer L-2 with its operand already in the X tile.
  PRIO(2);
  auto adj_gemm = [&](int li, auto&& pf) {
    refresh();
    if (li == 0)
      gemm<BW, EP / 16, FB, PB, ROWB, T::CKF, true>(acc, wq, rsW, fwdW(BW ? setFwdA : setFwdB, li), X, HD * 2, lane, pf);
    else if (li == L.cat)
      gemm<BW, (HD + EP) / 16, FB, PB, ROWB, T::CKF, true>(acc, wq, rsW, fwdW(BW ? setFwdA : setFwdB, li), X, 0, lane, pf);
    else
      gemm<BW, HD / 16, FB, PB, ROWB, T::CKF, true>(acc, wq, rsW, fwdW(BW ? setFwdA : setFwdB, li), X, 0, lane, pf);
  };
  // The injected second-order term of layer li, u q sigma''(z) = beta (u sigma') (q sigma') (1 - sigma') / sigma', is NOT spilled:
  // the reverse sweep rebuilds it from GB[li+1] = u sigma' and P[li] = q sigma', which the dW kernel needs in memory anyway --
  // one re-read moves from this sweep to the reverse sweep and five tensor stores per tile disappear.
  for (int li = 0; li < L.L - 1; ++li) {
    Pre preA;
    adj_gemm(li, [&] { prefetch(p.sp.A[li + 1], preA); });
    TS();
    lds_barrier();
    TS();
    refresh();
    for_blocks([&](int fb, int pb, int qp, int row) {
      float a[8], qb[8];
      load_s1(preA, fb, pb, qp, a);   // a[] = sigma'
#pragma unroll
      for (int e = 0; e < 8; ++e) qb[e] = acc[fb][pb][8 * qp + e] * a[e];
      put_x(BW, fb, pb, qp, qb, 0);
      if constexpr (G8) store_f8(p.sp.GB[li + 1], fb, pb, qp, qb, xs[row * 4 + 3], std::false_type{});
      else store_tile8_dw(p.sp.GB[li + 1], fb, pb, qp, qb);
      if (li == L.L - 2) park_tile8(fb, pb, qp, qb);   // the reverse sweep's FIRST unit needs it back three units from here
    });
    TS();
    lds_barrier();
    TS();
  }
  {   // top layer (peeled: its three partial-sum streams must not raise the register pressure of the loop above)
    const int li = L.L - 1;
    Pre preA;
    adj_gemm(li, [&] { prefetch(p.sp.A[li + 1], preA); });
    TS();
    lds_barrier();
    TS();
    refresh();
    float qsum[8], bsum[8], wsum[8], wv[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { qsum[e] = 0.f; bsum[e] = 0.f; wsum[e] = 0.f; }
    for_blocks2([&](int fb, int pb, int qp, int row) {
      if (pb == 0) {   // w_out staged in LDS (part[]) by the Ebar stage
        const int f0 = ubase(fb, qp) + 4 * hi;
        const float4 w0 = *(const float4*)(part + f0), w1 = *(const float4*)(part + f0 + 8);
        wv[0] = w0.x; wv[1] = w0.y; wv[2] = w0.z; wv[3] = w0.w; wv[4] = w1.x; wv[5] = w1.y; wv[6] = w1.z; wv[7] = w1.w;
      }
      float a[8], zb[8];
      load_tile8(preA, fb, pb, qp, a);
      const float sb = gbs[row * 4 + 3];
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float u = acc[fb][pb][8 * qp + e];
        const float t1 = __builtin_amdgcn_exp2f(-kC1 * a[e]), s1 = 1.f - t1;    // sigma' and 1 - sigma' = exp(-beta a)
        qsum[e] += u * s1;
        // p_L = q_L sigma' = so w_out sigma' rebuilt here (until round 5 it was re-read: it is an e4m3 tensor now, and this is exact)
        zb[e] = sb * wv[e] * s1 + kBeta * u * (so * wv[e] * s1) * t1;
        bsum[e] += zb[e];
        wsum[e] += sb * a[e];
      }
      store_tile8_zb(p.sp.ZB[li], fb, pb, qp, zb);
      if (li > 0) put_x(BW, fb, pb, qp, zb, 0);
    }, [&](int fb, int qp) {
#pragma unroll
      for (int e = 0; e < 8; ++e) qsum[e] *= so;
      vec_store8(qsum, L.L * HD + ubase(fb, qp));        // d w_out += so * sum_pts qbar_L
      vec_store8(wsum, L.L * HD + HD + ubase(fb, qp));   // d w_out += sum_pts sbar*so * a_L
      vec_store8(bsum, li * HD + ubase(fb, qp));
#pragma unroll
      for (int e = 0; e < 8; ++e) { qsum[e] = 0.f; bsum[e] = 0.f; wsum[e] = 0.f; }
    });
    TS();
    lds_barrier();
    TS();
  }

  // ------------------------------------------------------------------ ordinary reverse sweep with injection
  PRIO(3);
  for (int li = L.L - 2; li >= 0; --li) {
    Pre preA;
    typename std::conditional<G8, Pre8, Pre>::type preG;
    typename std::conditional<P8, Pre8, Pre>::type preP;
    refresh();
    gemm<BW, HD / 16, FB, PB, ROWB, T::CKF, true>(acc, wq, rsW, wptr(BW ? setBwdA : setBwdB, L.bwdMat[li + 1], HD), X, 0, lane,
                                               [&] {
                                                 prefetch(p.sp.A[li + 1], preA);
                                                 if (li != L.L - 2) {                     // the top one is parked in the tile
                                                   if constexpr (G8) prefetch8(p.sp.GB[li + 1], preG); else prefetch(p.sp.GB[li + 1], preG);
                                                 }
                                                 if constexpr (P8) prefetch8(p.sp.P[li], preP); else prefetch(p.sp.P[li], preP);
                                               });
    TS();
    lds_barrier();
    TS();
    refresh();
    float bsum[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) bsum[e] = 0.f;
    for_blocks2([&](int fb, int pb, int qp, int row) {
      float a[8], t1[8], gq[8], pv[8], zb[8];
      load_s1t(preA, fb, pb, qp, a, t1);   // a[] = sigma', t1[] = 1 - sigma' = exp(-beta a)
      if constexpr (G8) {
        if (li == L.L - 2) {               // parked by the adjoint sweep in the 16-bit spill type
          Pre pk;
          pk.v[fb][qp][pb] = *(const uint4*)(X + park_addr(cidx(fb, pb, qp)));
          load_tile8(pk, fb, pb, qp, gq);
        } else {
          load_f8(preG, fb, pb, qp, xs[row * 4 + 3], gq);
        }
      } else {
        if (li == L.L - 2) preG.v[fb][qp][pb] = *(const uint4*)(X + park_addr(cidx(fb, pb, qp)));
        load_tile8(preG, fb, pb, qp, gq);
      }
      if constexpr (P8) load_f8(preP, fb, pb, qp, kSpillPScale, pv); else load_tile8(preP, fb, pb, qp, pv);
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float s1 = a[e];
        // injected term beta (u sigma') (q sigma') (1 - sigma') / sigma'.  sigma' -> 0 takes u sigma' and q sigma' with it (a stored
        // activation of 0 gives sigma' = 0 exactly, and GB / P were formed with that same sigma': exact zeros), so a floor under the
        // divisor is all the guard it needs: 0 * t / 1e-30 = 0 -- one v_max instead of a compare and a select
        const float inj = (kBeta * gq[e]) * (pv[e] * t1[e]) * __builtin_amdgcn_rcpf(fmaxf(s1, 1e-30f));
        zb[e] = acc[fb][pb][8 * qp + e] * s1 + inj;
        bsum[e] += zb[e];
      }
      store_tile8_zb(p.sp.ZB[li], fb, pb, qp, zb);
      if (li > 0) put_x(BW, fb, pb, qp, zb, 0);
    }, [&](int fb, int qp) {
      vec_store8(bsum, li * HD + ubase(fb, qp));
#pragma unroll
      for (int e = 0; e < 8; ++e) bsum[e] = 0.f;
    });
    TS();
    lds_barrier();
    TS();
  }
  TS.wall(1);
}

// ---------------------------------------------------------------------------
template <int HD, int EP, int OPER, int MODE, bool BW = false, int SP8 = 0>
static int launch_one(const ChainParams& p, int64_t nTiles, hipStream_t st) {
  typedef Tile<HD, EP, oper_x2_all(OPER)> T;
  if constexpr (!BW && MODE == 2 && oper_f16(OPER))
    if (p.lay.bwd_f16) return launch_one<HD, EP, OPER, MODE, true>(p, nTiles, st);
  if constexpr (BW && !SP8 && HD == 256) {       // (make_layout: e4m3 spills with the 256-wide tiles only)
    if (p.lay.sp8 == 3) return launch_one<HD, EP, OPER, MODE, true, 3>(p, nTiles, st);
    if (p.lay.sp8 == 1) return launch_one<HD, EP, OPER, MODE, true, 1>(p, nTiles, st);
  }
  if (MODE == 2 && p.lay.sp8 != SP8) return ISDF_EUNSUPPORTED;
  auto k = chain_kernel<HD, EP, OPER, MODE, BW, SP8>;
  if (hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, T::LDS_BYTES) != hipSuccess) return ISDF_EHIP;
  hipLaunchKernelGGL(k, dim3((unsigned)nTiles), dim3(T::NW * 64), T::LDS_BYTES, st, p);
  return isdf_launch_status();
}

template <int HD, int EP, int MODE>
static int launch_oper(const ChainParams& p, int64_t nTiles, hipStream_t st) {
  if (p.lay.fwd_x2_all) {
    if constexpr (HD == 256 && EP == 256) return launch_one<HD, EP, 3, MODE>(p, nTiles, st);
    else return ISDF_EUNSUPPORTED;
  }
  if (p.lay.fwd_x2) return launch_one<HD, EP, 2, MODE>(p, nTiles, st);
  return p.lay.fwd_f16 ? launch_one<HD, EP, 1, MODE>(p, nTiles, st) : launch_one<HD, EP, 0, MODE>(p, nTiles, st);
}

template <int MODE>
static int launch_mode(const ChainParams& p, int64_t nTiles, hipStream_t st) {
  if (p.lay.HD == 256 && p.lay.EP == 256) return launch_oper<256, 256, MODE>(p, nTiles, st);   // replicaCAD / scanNet
  if (p.lay.HD == 256) return launch_oper<256, 512, MODE>(p, nTiles, st);   // realsense*.json: hidden 256, E = 381 / 465
  return launch_oper<512, 512, MODE>(p, nTiles, st);                         // BASELINE configs[4]
}

bool fwd_pair_supported(const NetLayout& l);                                        // fwd_pair.hip
int launch_fwd_pair(const ChainParams& p, int64_t nTiles, hipStream_t st);

// compute units of the device the launch goes to -- the STREAM's device, not the thread's current one (a caller may hold a stream of
// another device); cached per ordinal in atomics (concurrent host threads).  256 on an unpartitioned MI355X.  Speed only: the
// issue-priority heuristic of the chain kernel and the persistent grid of the forward kernel.
static int device_cu_count(hipStream_t st) {
  static std::atomic<int> cached[64];
  int dev = -1;
  if (st == nullptr || hipStreamGetDevice(st, &dev) != hipSuccess) {     // the null stream belongs to the current device
    if (hipGetDevice(&dev) != hipSuccess) return 256;
  }
  if (dev < 0 || dev >= 64) return 256;
  int n = cached[dev].load(std::memory_order_relaxed);
  if (!n) {
    n = (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && n > 0) ? n : 256;
    cached[dev].store(n, std::memory_order_relaxed);
  }
  return n;
}

int launch_chain(const ChainParams& p0, int mode, int64_t nTiles, hipStream_t st) {
  if (!layout_supported(p0.lay)) return ISDF_EUNSUPPORTED;
  if (nTiles <= 0) return ISDF_OK;
  ChainParams p = p0;
  p.n_cu = device_cu_count(st);
  switch (mode) {
    case 0:   // forward only: the pair-tile kernel where it exists (<256, 256>), the one-tile kernel elsewhere
      if (fwd_pair_supported(p.lay)) return launch_fwd_pair(p, nTiles, st);
      return launch_mode<0>(p, nTiles, st);
    case 1: return launch_mode<1>(p, nTiles, st);
    case 2: return launch_mode<2>(p, nTiles, st);
  }
  return ISDF_EINVAL;
}

}  // namespace isdf
