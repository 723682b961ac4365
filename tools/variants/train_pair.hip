// Pair-tile TRAIN kernel (K2, MODE 2, <HD = 256, EP = 256>, hidden_layers_block = 2, six octaves, fp16 operand family with fp16
// second-order sweeps -- the replicaCAD.json / scanNet.json configuration in the default operand modes; everything else stays on
// chain.hip): the structure of fwd_pair.hip carried through all four sweeps.  ONE 8-wave workgroup per CU (256 VGPRs per wave) owns
// TWO 64-point tiles ("halves") and runs them one stage out of phase,
//
//      stage 2k+1 :  GEMM of half A, unit k   interleaved with   epilogue of half B, unit k-1      (barrier)
//      stage 2k+2 :  GEMM of half B, unit k   interleaved with   epilogue of half A, unit k        (barrier)
//
// over the unit sequence  forward 0..5 | first reverse 5..1 | G | (PE-shaped middle, both halves together) | adjoint 0..5 | reverse 4..0.
// Same mathematics, operand types and HBM formats as chain.hip (a half is one tile of the spill buffer, one row of vec_part / wg_loss:
// the dW kernel and the step tail are unchanged); replaces, for two tiles of 64 points (same reference lines as chain.hip):
//   embedding.PostionalEncoding.forward   isdf/modules/embedding.py:95-111
//   SDFMap.forward                        isdf/modules/fc_map.py:94-111
//   fc_map.gradient (autograd.grad)       isdf/modules/fc_map.py:12-22
//   loss.bounds_ray / sdf_loss / tot_loss isdf/modules/loss.py:13-22,122-205
//   eikonal + normal terms                isdf/modules/trainer.py:814-830
//   the activation side of total_loss.backward()   trainer.py:981
//
// Why: the one-tile kernel's SIMD issue port is half idle (0.50) and its busiest unit is the vector-memory path (7 800
// vector-memory instructions per tile through a 64 B/clk path).  Here (i) a K = 256 weight matrix is fetched ONCE for 128 points
// (16-fragment window, re-requested a full stage ahead), (ii) every spilled tensor an epilogue re-reads sits in a ROLLING prefetch
// window: the register an epilogue group has just consumed is re-requested with the piece the OTHER half's next epilogue needs, a
// full stage before its use -- one set of prefetch registers instead of two, and nothing a stage waits for was requested inside it,
// (iii) the epilogue's VALU work issues in the shadow of the other half's MFMAs, one group of four elements behind an MFMA.
#include "chain_dev.h"

namespace isdf {

struct TrainPairTile {
  static constexpr int HD = 256, HB = TILE_PTS, BM = 2 * TILE_PTS, NW = 8, NPART = 8;
  static constexpr int ROWB = 4 * HD;                  // bytes per LDS row of a half: [region 1 | region 2] 16-bit elements
  static constexpr int HALFB = HB * ROWB;              // 64 KB
  static constexpr int OFF_XS = 2 * HALFB;             // float [BM][4]  x' (scaled / transformed point)
  static constexpr int OFF_PART = OFF_XS + BM * 16;    // float [2][NPART][HB][4]  partial sums (raw / g), per half
  static constexpr int OFF_GB = OFF_PART + 2 * NPART * HB * 16;   // float [BM][4]  gbar in x' space, [3] = sbar * so
  static constexpr int OFF_RED = OFF_GB + BM * 16;     // float [2][8]  per-half loss sums
  static constexpr int OFF_BIAS = OFF_RED + 64;        // float [6][HD] hidden biases, then [HD] w_out (zero beyond unit H)
  static constexpr int LDS_BYTES = OFF_BIAS + 7 * HD * 4;
};
static_assert(TrainPairTile::LDS_BYTES <= 160 * 1024, "one workgroup per CU");
static_assert(TILE_PTS == 64, "a half is one 64-point tile of the spill buffer");

namespace tp {
enum { UK_NONE = 0, UK_16 = 1, UK_32 = 2, UK_16T = 3, UK_32L = 4 };                          // GEMM kinds (fwd_pair.hip)
enum { EK_NONE = 0, EK_F_HID, EK_F_HILO, EK_F_LAST, EK_R1, EK_GG, EK_ADJ, EK_ADJ_TOP, EK_REV };   // epilogue kinds
constexpr int uk_frags(int uk) { return uk == UK_16 ? 16 : uk == UK_32 ? 32 : uk == UK_16T ? 32 : uk == UK_32L ? 48 : 0; }
constexpr int ek_pre(int ek) { return ek == EK_R1 ? 1 : (ek == EK_ADJ || ek == EK_ADJ_TOP || ek == EK_REV) ? 2 : 0; }   // spilled tensors an epilogue re-reads
struct Unit { int soff0, soff1, col0; };              // byte offsets of the wave's slice of the unit's matrices; LDS column (bytes) of its first operand
struct EArgs {                                        // one epilogue's parameters
  int li;        // layer index of the unit
  int h;         // half it belongs to
  int sA, sP, sGB, sZB;   // byte offsets (within a tile's spill block, this wave's slice) of the tensors it STORES
  int toR2;      // first reverse: also write region 2 (p_cat)
  int putx;      // reverse: write the tile for the next GEMM (li > 0)
};
struct PArgs { int n, h, t0, t1; };               // what to prefetch for the NEXT stage's epilogue: tensor count, half, byte offsets
// the unit sequence of a half (6 hidden layers, cat layer 3):
//   k = 0..5 forward l = k | 6..10 first reverse li = 11 - k | 11 the G GEMM | 12..17 adjoint li = k - 12 | 18..22 reverse li = 22 - k
constexpr int NL = 6, CAT = 3, NU = 23, KMID = 12;
constexpr int uk_of(bool x2, int k) {
  if (k < 6) { const bool comp = x2 && k >= CAT; return k == CAT ? (comp ? (int)UK_32L : (int)UK_32) : (comp ? (int)UK_16T : (int)UK_16); }
  if (k < 11) return UK_16;
  if (k == 11) return UK_32;
  if (k < 18) return k - 12 == CAT ? (int)UK_32 : (int)UK_16;
  return UK_16;
}
constexpr int ek_of(bool x2, int k) {
  if (k < 0) return EK_NONE;
  if (k < 6) return k == NL - 1 ? (int)EK_F_LAST : (x2 && k + 1 > CAT ? (int)EK_F_HILO : (int)EK_F_HID);
  if (k < 11) return EK_R1;
  if (k == 11) return EK_GG;
  if (k < 17) return EK_ADJ;
  if (k == 17) return EK_ADJ_TOP;
  return EK_REV;
}
}  // namespace tp

template <typename F, int... I>
__device__ __forceinline__ void tp_static_for(F&& f, std::integer_sequence<int, I...>) { (f(std::integral_constant<int, I>{}), ...); }

template <int OPER>
__global__ __launch_bounds__(TrainPairTile::NW * 64, 2) void train_pair_kernel(const ChainParams p) {
  using namespace tp;
  typedef TrainPairTile T;
  constexpr bool X2 = OPER >= 2;
  constexpr int NF = 6;
  constexpr int HD = T::HD, EP = T::HD, HB = T::HB, BM = T::BM, ROWB = T::ROWB;
  typedef Op<true>::v8 v8;
  typedef _Float16 hT;
  extern __shared__ __attribute__((aligned(1024))) char smem[];
  float* xs = (float*)(smem + T::OFF_XS);
  float* part = (float*)(smem + T::OFF_PART);
  float* gbs = (float*)(smem + T::OFF_GB);
  float* red = (float*)(smem + T::OFF_RED);
  float* biasL = (float*)(smem + T::OFF_BIAS);
  float* woutL = biasL + NL * HD;

  const NetLayout& L = p.lay;
  const int tid = threadIdx.x;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lane = tid & 63, j = lane & 31, hi = lane >> 5, lane16 = lane * 16;
  const int64_t P = p.n_valid ? (int64_t)(*p.n_valid) * p.S : p.n_points_host;
  const int64_t n0 = (int64_t)blockIdx.x * BM;
  if (n0 >= P) return;
  const bool validB = n0 + HB < P;              // the second half holds points (otherwise it runs on zeros and writes nothing)
  const float so = L.scale_output;
  ChainStamps TS(p.dbg);
  TS();
  TS.wall(0);

  // ---- global-memory descriptors: weights, and per half the tile's spill block / vec_part row (an empty half gets empty ranges)
  const rsrc_t rsW = make_rsrc(p.shadow, 0x7fffffffu);
  const int64_t tile0 = (int64_t)blockIdx.x * 2;
  uint16_t* spill0 = p.spill + tile0 * p.sp.tileStride;
  uint16_t* spill1 = validB ? spill0 + p.sp.tileStride : spill0;
  const uint32_t spBytes = (uint32_t)(p.sp.tileStride * 2);
  const rsrc_t rsS0 = make_rsrc(spill0, spBytes), rsS1 = make_rsrc(spill1, validB ? spBytes : 0u);
  const i32x4 srd0 = make_srd(spill0, spBytes), srd1 = make_srd(spill1, validB ? spBytes : 0u);
  float* vec0 = p.vec_part + tile0 * p.vecStride;
  float* vec1 = validB ? vec0 + p.vecStride : vec0;
  const rsrc_t rsV0 = make_rsrc(vec0, (uint32_t)p.vecStride * 4u), rsV1 = make_rsrc(vec1, validB ? (uint32_t)p.vecStride * 4u : 0u);
  auto sbase = [&](int64_t tensorOff) __attribute__((always_inline)) { return (int)(tensorOff * 2) + w * 4096; };   // this wave's first piece of a spilled tensor

  // ---- weights: unit descriptors and the 16-fragment window
  auto frag16 = [&](int l16, int soff, int ks) __attribute__((always_inline)) { return bload16<0>(rsW, l16 + (ks & 3) * 1024, soff + (ks >> 2) * 4096); };
  auto frag = [&](int soff, int ks) __attribute__((always_inline)) { return frag16(lane16, soff, ks); };
  auto wslice = [&](int64_t set, int64_t matOff, int kp) __attribute__((always_inline)) { return (int)((set + matOff) * 2) + w * (kp / 16) * 1024; };
  auto fwd_unit = [&](int li, bool lo) __attribute__((always_inline)) {   // forward-orientation matrix of layer li (forward and adjoint sweeps)
    const int kp = li == 0 ? EP : (li == CAT ? HD + EP : HD);
    Unit u;
    u.soff0 = wslice(L.setFwdA, L.fwdMat[li], kp);
    u.soff1 = lo ? wslice(L.setFwdLo, L.fwdMat[li], kp) + (li == CAT ? (HD / 16) * 1024 : 0) : 0;
    u.col0 = li == 0 ? HD * 2 : 0;
    return u;
  };
  auto bwd_unit = [&](int li) __attribute__((always_inline)) { Unit u; u.soff0 = wslice(L.setBwdA, L.bwdMat[li], HD); u.soff1 = 0; u.col0 = 0; return u; };
  auto gg_unit = [&]() __attribute__((always_inline)) { Unit u; u.soff0 = wslice(L.setBwdA, L.bwdG, 2 * HD); u.soff1 = 0; u.col0 = 0; return u; };
  uint4 W[16];
  {
    const Unit u0 = fwd_unit(0, false);
#pragma unroll
    for (int r = 0; r < 16; ++r) W[r] = frag(u0.soff0, r);
  }

  // ------------------------------------------------------------------ biases / w_out -> LDS (zero beyond unit H)
  {
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
      const int q = tid + i * T::NW * 64;
      if (q < (NL + 1) * HD) biasL[q] = bvv[i];
    }
  }
  // ------------------------------------------------------------------ PE of both halves (lane = point, wave = direction slice)
  const int sw = (lane & 15) << 4;
  {
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int64_t n = n0 + h * HB + lane;
      float x0 = 0.f, x1 = 0.f, x2 = 0.f;
      if (n < P) { x0 = p.pts[n * 3]; x1 = p.pts[n * 3 + 1]; x2 = p.pts[n * 3 + 2]; }
      // transform_3D_grid (transform.py:287-304) then * scale (embedding.py:12-22)
      const float y0 = (L.T[0] * x0 + L.T[1] * x1 + L.T[2] * x2 + L.T[3]) * L.scale_input;
      const float y1 = (L.T[4] * x0 + L.T[5] * x1 + L.T[6] * x2 + L.T[7]) * L.scale_input;
      const float y2 = (L.T[8] * x0 + L.T[9] * x1 + L.T[10] * x2 + L.T[11]) * L.scale_input;
      char* row = smem + h * T::HALFB + lane * ROWB;
      auto put = [&](int feat, float v) __attribute__((always_inline)) {
        *(hT*)(row + (((HD + feat) * 2) ^ sw)) = (hT)v;      // region 2: the forward operand
        *(hT*)(row + ((feat * 2) ^ sw)) = (hT)v;             // region 1: staged for the spill (dW operand A_0)
      };
      if (w == T::NW - 1) {
        xs[(h * HB + lane) * 4] = y0; xs[(h * HB + lane) * 4 + 1] = y1; xs[(h * HB + lane) * 4 + 2] = y2;
        put(0, y0); put(1, y1); put(2, y2);
        for (int f = L.E; f < EP; ++f) put(f, 0.f);
      }
      for (int d = w; d < N_DIRS; d += T::NW) {
        const float proj = y0 * kDirs[0][d] + y1 * kDirs[1][d] + y2 * kDirs[2][d];
        float fr = 1.f;
#pragma unroll
        for (int f = 0; f < NF; ++f) {
          const float xb = proj * fr;
          put(3 + d * NF + f, __sinf(xb));
          put(3 + N_DIRS * NF + d * NF + f, __sinf(xb + kHalfPi));
          fr *= 2.f;
        }
      }
    }
  }
  TS();
  lds_barrier();
  TS();

  // LDS byte offsets (within a half) of this lane's operand reads and epilogue writes (chain.hip: gemm() / put_x())
  const int xlane = j * ROWB + ((hi * 16) ^ ((j & 15) << 4));
  const int xw = j * ROWB + 8 * hi + (((j & 15) << 4) ^ ((w & 3) * 64)) + (w >> 2) * 256;
  // spill a [HB][HD] 16-bit region of a half to global in frag16 order (16 B per lane and piece)
  auto spill_region = [&](int h, int colElemBase, int64_t tensorOff) __attribute__((always_inline)) {
    const int sb = sbase(tensorOff);
#pragma unroll
    for (int pb = 0; pb < 2; ++pb)
#pragma unroll
      for (int qp = 0; qp < 2; ++qp) {
        const int lb = h * T::HALFB + (xw ^ (32 * qp)) + colElemBase * 2 + pb * 32 * ROWB;
        const uint2 lo = *(const uint2*)(smem + lb), hi2 = *(const uint2*)(smem + (lb ^ 16));
        if (h) bstore16_nt<true>(make_uint4(lo.x, lo.y, hi2.x, hi2.y), srd1, lane16, sb, pb * 2 + qp);
        else bstore16_nt<true>(make_uint4(lo.x, lo.y, hi2.x, hi2.y), srd0, lane16, sb, pb * 2 + qp);
      }
  };
  spill_region(0, 0, p.sp.A[0]);
  spill_region(1, 0, p.sp.A[0]);
  lds_barrier();

  // per-workgroup partial of a bias / out-layer gradient entry (chain.hip vec_store8): the 8 values of a block in ONE store
  auto vec_store8 = [&](float (&v)[8], int elemUniform, int h, int j, int hi) __attribute__((always_inline)) {
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = half_wave_sum(v[e]);
    const bool b0 = j & 1, b1 = j & 2, b2 = j & 4;
    const float t0 = b0 ? v[1] : v[0], t1 = b0 ? v[3] : v[2], t2 = b0 ? v[5] : v[4], t3 = b0 ? v[7] : v[6];
    const float u0 = b1 ? t1 : t0, u1 = b1 ? t3 : t2;
    const float r = b2 ? u1 : u0;   // = v[j & 7]
    if (j < 8) {
      if (h) __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(r), rsV1, 16 * hi + 4 * (j & 3) + 32 * (j >> 2), elemUniform * 4, 0);
      else __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(r), rsV0, 16 * hi + 4 * (j & 3) + 32 * (j >> 2), elemUniform * 4, 0);
    }
  };

  f32x16 accA[2], accB[2];
  // the rolling prefetch windows, one per half: tensor k, piece c = pb * 2 + qp of the half's NEXT epilogue.  A half's epilogue
  // re-requests every register it has just consumed with the piece its OWN next epilogue needs -- two stages ahead (with one window
  // shared by both halves the distance is half a unit, and a unit then cannot be shorter than twice the loaded HBM latency:
  // profiles/r05_train_pair_timeline.txt, stages alternating 5.9 k / 2.8 k cycles)
  uint4 PreA[2][4], PreB[2][4];
  float rawp[2];
#pragma unroll
  for (int k = 0; k < 2; ++k)
#pragma unroll
    for (int c = 0; c < 4; ++c) { PreA[k][c] = make_uint4(0, 0, 0, 0); PreB[k][c] = make_uint4(0, 0, 0, 0); }

  // ================================================================== the stage
  // GEMM of half `goff` (unit u, kind UK) into accG, interleaved with the epilogue EK of accE (parameters e), whose groups also
  // re-request the prefetch window for the NEXT stage's epilogue (pf).  IS_B: second half to use the unit (window -> next unit).
  auto stage = [&](auto ukc, auto ekc, auto isb, f32x16 (&accG)[2], f32x16 (&accE)[2], uint4 (&Pre)[2][4], const Unit u, const int nxtSoff0,
                   const int gh, const EArgs e, const PArgs pf) __attribute__((always_inline)) {
    constexpr int UK = decltype(ukc)::value, EK = decltype(ekc)::value;
    constexpr bool IS_B = decltype(isb)::value;
    constexpr bool TWO = UK == UK_16T;
    constexpr int NPRE = ek_pre(EK);
    // Lane ids and lane bases are RE-DERIVED from an opaque copy of threadIdx in every stage (chain.hip's refresh()): as kernel-wide
    // lane constants they (and every address and predicate built from them) stay live across all 46 stages, and at this register
    // budget the allocator spills them -- and reloads them behind an s_waitcnt vmcnt(0), i.e. behind the stage's own prefetches.
    int t_ = tid;
    asm volatile("" : "+v"(t_));
    const int lane = t_ & 63, j = lane & 31, hi = lane >> 5, lane16 = lane * 16;
    const int xlane = j * ROWB + ((hi * 16) ^ ((j & 15) << 4));
    const int xw = j * ROWB + 8 * hi + (((j & 15) << 4) ^ ((w & 3) * 64)) + (w >> 2) * 256;
    const int xgl = xlane + gh * T::HALFB, xel = xw + e.h * T::HALFB;
    constexpr int NG = 8;                                                                  // epilogue groups of four elements
    constexpr int NS = UK == UK_NONE ? NG : (UK == UK_16 ? 32 : UK == UK_32 ? 64 : 96);   // MFMA slots of the stage's GEMM
    // ---- epilogue state
    float bv[8], wv[8], bsum[8], csum[8];
    uint2 keep[3];                 // first halves of the 16-byte pieces that leave when the block is complete
    f32x2 r2 = {0.f, 0.f};
    (void)bv; (void)wv; (void)bsum; (void)csum; (void)keep; (void)r2;
    if constexpr (EK == EK_F_LAST) { rawp[0] = 0.f; rawp[1] = 0.f; }
    auto load8 = [&](const float* src, float (&o)[8]) __attribute__((always_inline)) {
      const float4 b0 = *(const float4*)src, b1 = *(const float4*)(src + 8);
      o[0] = b0.x; o[1] = b0.y; o[2] = b0.z; o[3] = b0.w; o[4] = b1.x; o[5] = b1.y; o[6] = b1.z; o[7] = b1.w;
    };
    // Spill stores the COMPILER counts: vmcnt covers stores too on this ISA and retires in order, and every store the waitcnt pass
    // does not know about (chain.hip's hand-issued asm store) makes each later counted wait stricter by one -- with 8 .. 16 of them per
    // stage a wait for a weight fragment turned into a wait for the prefetch burst issued 16 requests later, i.e. for HBM.  The store
    // goes out with soffset = 0 and the tensor offset in the VGPR: for THAT form the hazard recogniser inserts the store-data wait
    // state itself (the SGPR-soffset form is the one it wrongly believes exempt: DESIGN 4).
    auto store16 = [&](uint2 a, uint2 b, int sb, int c, bool nt) __attribute__((always_inline)) {
      u32x4 v; v[0] = a.x; v[1] = a.y; v[2] = b.x; v[3] = b.y;
      const int voff = lane16 + sb + c * 1024;
      if (e.h) { if (nt) __builtin_amdgcn_raw_buffer_store_b128(v, rsS1, voff, 0, kAuxNT); else __builtin_amdgcn_raw_buffer_store_b128(v, rsS1, voff, 0, 0); }
      else { if (nt) __builtin_amdgcn_raw_buffer_store_b128(v, rsS0, voff, 0, kAuxNT); else __builtin_amdgcn_raw_buffer_store_b128(v, rsS0, voff, 0, 0); }
    };
    auto unpack4 = [&](const uint4& u, int hb, float (&o)[4]) __attribute__((always_inline)) {
      const f16x4 a = __builtin_bit_cast(f16x4, hb ? make_uint2(u.z, u.w) : make_uint2(u.x, u.y));
#pragma unroll
      for (int i = 0; i < 4; ++i) o[i] = (float)a[i];
    };
    // one group: elements 8 qp + 4 hb + {0..3} of accE[pb]; block piece c = pb * 2 + qp
    auto group = [&](int g) __attribute__((always_inline)) {
      if constexpr (EK != EK_NONE) {
        const int qp = g >> 2, pb = (g >> 1) & 1, hb = g & 1, c = pb * 2 + qp;
        const int f0 = w * 32 + 16 * qp + 4 * hi;                         // features f0 .. f0+3 (hb 0) / f0+8 .. f0+11 (hb 1)
        const int lb = ((xel ^ (32 * qp)) + pb * 32 * ROWB) ^ (hb ? 16 : 0);   // this group's 8-byte piece of the tile (region 1)
        float a4[4];
        (void)f0; (void)lb;
        if constexpr (EK == EK_F_HID || EK == EK_F_HILO || EK == EK_F_LAST) {
          if (pb == 0 && hb == 0) { load8(biasL + e.li * HD + f0, bv); if (EK == EK_F_LAST) load8(woutL + f0, wv); }
          float pl[4];
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const float z = accE[pb][8 * qp + 4 * hb + i] + bv[4 * hb + i];
            if constexpr (EK == EK_F_LAST) { float s1; a4[i] = softplus_s1(z, s1); pl[i] = so * wv[4 * hb + i] * s1; }   // p_L = q_L sigma'(z_L), q_L = so w_out
            else a4[i] = softplus_f(z);
          }
          const uint2 pa = pack4<true>(a4[0], a4[1], a4[2], a4[3]);
          if constexpr (EK == EK_F_LAST) {
            // w_out . a as packed FMAs of ELEMENT PAIRS (isa_lint rule 1), same grouping as chain.hip
            r2 += f32x2{wv[4 * hb], wv[4 * hb + 1]} * f32x2{a4[0], a4[1]};
            r2 += f32x2{wv[4 * hb + 2], wv[4 * hb + 3]} * f32x2{a4[2], a4[3]};
            if (hb == 1) { rawp[pb] += r2[0] + r2[1]; r2 = f32x2{0.f, 0.f}; }
            const uint2 pp = pack4<true>(pl[0], pl[1], pl[2], pl[3]);
            *(uint2*)(smem + lb) = pp;                                     // p_L: the first reverse sweep's operand
            if (hb == 0) { keep[0] = pa; keep[1] = pp; }
            else { store16(keep[0], pa, e.sA, c, true); store16(keep[1], pp, e.sP, c, true); }
          } else {
            *(uint2*)(smem + lb) = pa;
            if constexpr (EK == EK_F_HILO)   // region 2 <- fp16(a - fp16(a)): the next layer's second operand
              *(uint2*)(smem + lb + HD * 2) = pack4<true>(f16_residual(a4[0]), f16_residual(a4[1]), f16_residual(a4[2]), f16_residual(a4[3]));
            if (hb == 0) keep[0] = pa; else store16(keep[0], pa, e.sA, c, true);
          }
        } else if constexpr (EK == EK_R1) {
          unpack4(Pre[0][c], hb, a4);                                      // a_l: sigma'(z_l) is re-derived from it
          float v4[4];
#pragma unroll
          for (int i = 0; i < 4; ++i) v4[i] = accE[pb][8 * qp + 4 * hb + i] * s1_from_a(a4[i]);
          const uint2 pv = pack4<true>(v4[0], v4[1], v4[2], v4[3]);
          *(uint2*)(smem + lb) = pv;
          if (e.toR2) *(uint2*)(smem + lb + HD * 2) = pv;
          if (hb == 0) keep[0] = pv; else store16(keep[0], pv, e.sP, c, true);
        } else if constexpr (EK == EK_ADJ) {
          // qb = u sigma' (the adjoint entering the next layer, and a dW operand), and with it the injected second-order term of this
          // layer for the reverse sweep, beta (u sigma') (q sigma') (1 - sigma') / sigma' -- q sigma' = P_l is re-read here (it is two
          // sweeps old), so that the reverse sweep re-reads TWO tensors per unit (a, the injection) instead of three (a, GB, P): two
          // tensors per half are what two-stage-ahead prefetch windows for both halves cost in registers (2 x 2 x 16 VGPRs).  The
          // injection travels in the ZB slot of its layer, which the reverse epilogue overwrites with zb after reading it.
          float pv4[4], v4[4], in4[4];
          unpack4(Pre[0][c], hb, a4);
          unpack4(Pre[1][c], hb, pv4);
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const float t1 = __builtin_amdgcn_exp2f(-kC1 * a4[i]), s1 = 1.f - t1;    // sigma' and 1 - sigma' = exp(-beta a)
            const float uu = accE[pb][8 * qp + 4 * hb + i];
            v4[i] = uu * s1;
            in4[i] = kBeta * uu * pv4[i] * t1;      // sigma' cancels: beta (u sigma') (q sigma') (1 - sigma') / sigma' = beta u (q sigma') (1 - sigma'), u at hand here
          }
          const uint2 pq = pack4<true>(v4[0], v4[1], v4[2], v4[3]), pi = pack4<true>(in4[0], in4[1], in4[2], in4[3]);
          *(uint2*)(smem + lb) = pq;
          if (hb == 0) { keep[0] = pq; keep[1] = pi; }
          else { store16(keep[0], pq, e.sGB, c, false); store16(keep[1], pi, e.sZB, c, true); }
        } else if constexpr (EK == EK_GG) {
          // Eg as fp32 into the (idle) tile of the half: [HB][HD] floats, 16-byte pieces of 4 consecutive features
          // accumulator registers 8 qp + 4 hb + {0..3} of block pb are features 32 w + 16 qp + 8 hb' ... : register r = 4 rq + i holds
          // feature 32 w + 8 rq + 4 hi + i (chain.hip staging loop), and 8 qp + 4 hb = 4 rq  =>  rq = 2 qp + hb
          const int rq = 2 * qp + hb, row = pb * 32 + j;
          const int ff = w * 32 + 8 * rq + 4 * hi;
          *(float4*)(smem + e.h * T::HALFB + row * ROWB + swz(row, ff * 4)) =
              make_float4(accE[pb][4 * rq], accE[pb][4 * rq + 1], accE[pb][4 * rq + 2], accE[pb][4 * rq + 3]);
        } else if constexpr (EK == EK_ADJ_TOP) {
          if (pb == 0 && hb == 0) {
            load8(woutL + f0, wv);
#pragma unroll
            for (int q = 0; q < 8; ++q) { csum[q] = 0.f; bsum[q] = 0.f; }
          }
          float pv4[4], zb[4];
          unpack4(Pre[0][c], hb, a4);
          unpack4(Pre[1][c], hb, pv4);
          const float sb = gbs[(e.h * HB + pb * 32 + j) * 4 + 3];
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const float uu = accE[pb][8 * qp + 4 * hb + i];
            const float t1 = __builtin_amdgcn_exp2f(-kC1 * a4[i]), s1 = 1.f - t1;    // sigma' and 1 - sigma' = exp(-beta a)
            // d w_out = so sum_pts qbar_L (adjoint path) + sum_pts sbar so a_L (reverse path): ONE running sum (chain.hip keeps the two
            // apart and the step tail adds them; here eight registers matter)
            csum[4 * hb + i] += so * (uu * s1) + sb * a4[i];
            zb[i] = sb * wv[4 * hb + i] * s1 + kBeta * uu * pv4[i] * t1;
            bsum[4 * hb + i] += zb[i];
          }
          const uint2 pz = pack4<true>(zb[0], zb[1], zb[2], zb[3]);
          *(uint2*)(smem + lb) = pz;
          if (hb == 0) keep[0] = pz; else store16(keep[0], pz, e.sZB, c, false);
          if (pb == 1 && hb == 1) {
            const int ub = w * 32 + 16 * qp;
            vec_store8(csum, NL * HD + ub, e.h, j, hi);            // both parts of d w_out in the first of the two slots the step tail adds
            if (j < 8) {                                           // ... and zeros in the second
              if (e.h) __builtin_amdgcn_raw_buffer_store_b32(0u, rsV1, 16 * hi + 4 * (j & 3) + 32 * (j >> 2), (NL * HD + HD + ub) * 4, 0);
              else __builtin_amdgcn_raw_buffer_store_b32(0u, rsV0, 16 * hi + 4 * (j & 3) + 32 * (j >> 2), (NL * HD + HD + ub) * 4, 0);
            }
            vec_store8(bsum, e.li * HD + ub, e.h, j, hi);
#pragma unroll
            for (int q = 0; q < 8; ++q) { csum[q] = 0.f; bsum[q] = 0.f; }
          }
        } else {   // EK_REV
          if (pb == 0 && hb == 0) {
#pragma unroll
            for (int q = 0; q < 8; ++q) bsum[q] = 0.f;
          }
          float in4[4], zb[4];
          unpack4(Pre[0][c], hb, a4);
          unpack4(Pre[1][c], hb, in4);                                     // the injected term, formed by the adjoint sweep
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            zb[i] = accE[pb][8 * qp + 4 * hb + i] * s1_from_a(a4[i]) + in4[i];
            bsum[4 * hb + i] += zb[i];
          }
          const uint2 pz = pack4<true>(zb[0], zb[1], zb[2], zb[3]);
          if (e.putx) *(uint2*)(smem + lb) = pz;
          if (hb == 0) keep[0] = pz; else store16(keep[0], pz, e.sZB, c, false);
          if (pb == 1 && hb == 1) vec_store8(bsum, e.li * HD + w * 32 + 16 * qp, e.h, j, hi);
        }
      }
    };
    // The window's re-requests leave as ONE burst at the end of the stage, behind the stage's last weight request: vmcnt retires in
    // order, so a re-read requested BETWEEN two weight fragments is waited for -- with its full HBM latency -- by the next stage's
    // GEMM when it waits for the later fragment (first version: re-requests right behind each consumed block; stages that had
    // weight requests queued behind them took 6 k cycles, the others 3 k: profiles/r05_train_pair_timeline.txt).
    auto refill = [&]() __attribute__((always_inline)) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int c = (i & 1) * 2 + (i >> 1);      // the order the next epilogue consumes them: c = 0, 2, 1, 3
        if (pf.n >= 1) Pre[0][c] = pf.h ? bload16<kAuxNT>(rsS1, lane16 + c * 1024, pf.t0) : bload16<kAuxNT>(rsS0, lane16 + c * 1024, pf.t0);
        if (pf.n >= 2) Pre[1][c] = pf.h ? bload16<kAuxNT>(rsS1, lane16 + c * 1024, pf.t1) : bload16<kAuxNT>(rsS0, lane16 + c * 1024, pf.t1);
      }
      (void)NPRE;
    };
    auto slot_done = [&](int m) __attribute__((always_inline)) {
      const int q = (m * NG + NS - 1) / NS;
      if (q < NG && q * NS / NG == m) group(q);
    };
    // ---- the GEMM (fwd_pair.hip)
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
#pragma unroll
      for (int g = 0; g < NG; ++g) group(g);
      refill();
    } else {
      constexpr int NT = uk_frags(UK);
      readb(0, bq[0]);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        const int r = t & 15;
        const int m0 = TWO ? (t < 16 ? 4 * t : 64 + 2 * (t - 16)) : 2 * t;   // first MFMA slot of the step
        if (t + 1 < NT) readb(t + 1, bq[(t + 1) & 1]);
        if constexpr (TWO) { if (t < 16) readlo(t); }
        __builtin_amdgcn_sched_barrier(0);
        accG[0] = Op<true>::mfma(__builtin_bit_cast(v8, W[r]), bq[t & 1][0], t == 0 ? f32x16(0.f) : accG[0]);
        __builtin_amdgcn_sched_barrier(0);
        slot_done(m0);
        __builtin_amdgcn_sched_barrier(0);
        accG[1] = Op<true>::mfma(__builtin_bit_cast(v8, W[r]), bq[t & 1][1], t == 0 ? f32x16(0.f) : accG[1]);
        __builtin_amdgcn_sched_barrier(0);
        slot_done(m0 + 1);
        if constexpr (TWO) {
          if (t < 16) {
            __builtin_amdgcn_sched_barrier(0);
            accG[0] = Op<true>::mfma(__builtin_bit_cast(v8, W[r]), bl[0], accG[0]);
            __builtin_amdgcn_sched_barrier(0);
            slot_done(m0 + 2);
            __builtin_amdgcn_sched_barrier(0);
            accG[1] = Op<true>::mfma(__builtin_bit_cast(v8, W[r]), bl[1], accG[1]);
            __builtin_amdgcn_sched_barrier(0);
            slot_done(m0 + 3);
          }
        }
        constexpr int NFR = uk_frags(UK);      // the window register's next tenant
        if (t + 16 < NFR) {
          const int tt = t + 16;
          if (UK == UK_32) W[r] = frag16(lane16, u.soff0, tt);
          else if (UK == UK_16T) W[r] = frag16(lane16, u.soff1, tt - 16);
          else W[r] = tt < 32 ? frag16(lane16, u.soff0, tt) : frag16(lane16, u.soff1, tt - 32);
        } else if (IS_B) {
          W[r] = frag16(lane16, nxtSoff0, r);              // the next unit's fragment r, a stage ahead of half A
        } else if (NFR > 16) {
          W[r] = frag16(lane16, u.soff0, r);               // half B starts the unit over
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      refill();
    }
    if constexpr (EK == EK_F_LAST) {
#pragma unroll
      for (int pb = 0; pb < 2; ++pb) {
        const float v = rawp[pb] + __shfl_xor(rawp[pb], 32, 64);   // add the two feature halves
        if (hi == 0) part[((e.h * T::NPART + w) * HB + pb * 32 + j) * 4] = v;
      }
    }
  };

  // ================================================================== the unit sequence (tp::uk_of / ek_of), unrolled at compile time
  auto unit_k = [&](int k) __attribute__((always_inline)) {
    if (k < 6) return fwd_unit(k, X2 && k >= CAT);
    if (k < 11) return bwd_unit(11 - k);
    if (k == 11) return gg_unit();
    if (k < 18) return fwd_unit(k - 12, false);
    return bwd_unit(22 - k + 1);
  };
  auto eargs_k = [&](int k, int h) __attribute__((always_inline)) {
    EArgs e = {0, h, 0, 0, 0, 0, 0, 0};
    if (k < 0) return e;
    if (k < 6) { e.li = k; e.sA = sbase(p.sp.A[k + 1]); e.sP = sbase(p.sp.P[k]); }
    else if (k < 11) { const int li = 11 - k; e.li = li; e.sP = sbase(p.sp.P[li - 1]); e.toR2 = (li - 1 == CAT); }
    else if (k == 11) { e.li = 0; }
    else if (k < 18) { const int li = k - 12; e.li = li; e.sGB = sbase(p.sp.GB[li + 1]); e.sZB = sbase(p.sp.ZB[li]); }
    else { const int li = 22 - k; e.li = li; e.sZB = sbase(p.sp.ZB[li]); e.putx = li > 0; }
    return e;
  };
  // what unit k's epilogue re-reads (requested by the same half's PREVIOUS epilogue: two stages ahead)
  auto pargs_k = [&](int k, int h) __attribute__((always_inline)) {
    PArgs f = {0, h, 0, 0};
    if (k < 6 || k == 11 || k >= NU) return f;
    if (k < 11) { f.n = 1; f.t0 = sbase(p.sp.A[11 - k]); }
    else if (k < 17) { f.n = 2; f.t0 = sbase(p.sp.A[k - 12 + 1]); f.t1 = sbase(p.sp.P[k - 12]); }
    else if (k == 17) { f.n = 2; f.t0 = sbase(p.sp.A[NL]); f.t1 = sbase(p.sp.P[NL - 1]); }
    else { const int li = 22 - k; f.n = 2; f.t0 = sbase(p.sp.A[li + 1]); f.t1 = sbase(p.sp.ZB[li]); }   // a, and the injection parked in the ZB slot
    return f;
  };

  float my_sdf = 0.f;
  float li_bnd = 0.f, li_c[3] = {0.f, 0.f, 0.f}, li_dz[2] = {0.f, 0.f}, li_t[3] = {0.f, 0.f, 0.f}, li_n[3] = {0.f, 0.f, 0.f};

  // one pair of stages for unit k; first/last flags drop the epilogue of the previous sweep / the GEMM of the next
  auto unit_pair = [&](auto kc) __attribute__((always_inline)) {
    constexpr int k = decltype(kc)::value;
    constexpr int UK = uk_of(X2, k);
    constexpr int EKprev = k == KMID ? (int)EK_NONE : ek_of(X2, k - 1);      // the middle section drains the pipeline: no epilogue pending at k = 12
    constexpr int EKcur = ek_of(X2, k);
    const Unit u = unit_k(k);
    const int nxt = unit_k(k + 1 < NU ? k + 1 : 0).soff0;               // (past the last unit: a harmless re-request)
    // stage A: GEMM of half A, unit k || epilogue of half B, unit k - 1; its epilogue slots request what E_B(k) re-reads
    stage(std::integral_constant<int, UK>{}, std::integral_constant<int, EKprev>{}, std::false_type{}, accA, accB, PreB, u, nxt, 0,
          eargs_k(k == KMID ? -1 : k - 1, 1), pargs_k(k, 1));
    TS();
    lds_barrier();
    TS();
    // stage B: GEMM of half B, unit k || epilogue of half A, unit k; its epilogue slots request what E_A(k + 1) re-reads
    stage(std::integral_constant<int, UK>{}, std::integral_constant<int, EKcur>{}, std::true_type{}, accB, accA, PreA, u, nxt, 1,
          eargs_k(k, 0), pargs_k(k + 1, 0));
    TS();
    lds_barrier();
    TS();
  };
  // the forward, first-reverse and G units
  tp_static_for(unit_pair, std::make_integer_sequence<int, KMID>{});
  {   // epilogue of half B, the G unit (staging Eg of half B); nothing to prefetch: the adjoint sweep's first epilogue is requested below
    const Unit u = gg_unit();
    stage(std::integral_constant<int, UK_NONE>{}, std::integral_constant<int, EK_GG>{}, std::false_type{}, accA, accB, PreB, u, u.soff0, 0,
          eargs_k(11, 1), pargs_k(KMID, 1));
  }
  // ------------------------------------------------------------------ sdf = (raw + noise) * so   (fc_map.py:104-109); loss inputs
  // (part[] still holds the per-wave partials of the output layer: the G epilogue wrote the tiles, not part[])
  const int lh = tid >> 6, lpt = tid & 63;     // (threads 0..127: one per point of the pair)
  if (tid < BM) {
    float r = p.params[L.offBout];
#pragma unroll
    for (int k = 0; k < T::NPART; ++k) r += part[((lh * T::NPART + k) * HB + lpt) * 4];
    const int64_t n = n0 + tid;
    if (p.noise) { if (n < P) r += p.noise[n]; }
    else if (p.noise_std != 0.f) {   // Box-Muller on Philox4x32-10 keyed by (seed, offset, point) -- chain.hip
      const uint4 u = philox4x32_10(make_uint4((uint32_t)n, (uint32_t)(n >> 32), (uint32_t)p.noise_off, (uint32_t)(p.noise_off >> 32)),
                                    make_uint2((uint32_t)p.noise_seed, (uint32_t)(p.noise_seed >> 32) ^ 0x5eedu));
      const float u1 = fmaxf(u01(u.x), 1e-7f), u2 = u01(u.y);
      r += p.noise_std * sqrtf(-2.f * __logf(u1)) * __cosf(6.2831853f * u2);
    }
    my_sdf = r * so;
    if (p.sdf && n < P) p.sdf[n] = my_sdf;
    if (n < P) {
      const int64_t ray = (int64_t)((uint32_t)n / (uint32_t)p.S);
      if (p.loss.bounds_method == 0) {  // loss.py:13-22
        li_c[0] = p.dirsC[ray * 3]; li_c[1] = p.dirsC[ray * 3 + 1]; li_c[2] = p.dirsC[ray * 3 + 2];
        li_dz[0] = p.depth[ray]; li_dz[1] = p.z_vals[n];
        li_t[0] = p.dirsW[ray * 3]; li_t[1] = p.dirsW[ray * 3 + 1]; li_t[2] = p.dirsW[ray * 3 + 2];
      } else {
        li_bnd = p.pc_bounds[n];
        li_t[0] = p.pc_grad_vec[n * 3]; li_t[1] = p.pc_grad_vec[n * 3 + 1]; li_t[2] = p.pc_grad_vec[n * 3 + 2];
      }
      if (p.normals) { li_n[0] = p.normals[ray * 3]; li_n[1] = p.normals[ray * 3 + 1]; li_n[2] = p.normals[ray * 3 + 2]; }
    }
  }
  lds_barrier();   // Eg of both halves staged; part[] read
  TS();
  // ------------------------------------------------------------------ g_x' = J_pe^T Eg, both halves (lane = point, wave = direction slice)
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const float y0 = xs[(h * HB + lane) * 4], y1 = xs[(h * HB + lane) * 4 + 1], y2 = xs[(h * HB + lane) * 4 + 2];
    const char* row = smem + h * T::HALFB + lane * ROWB;
    auto eg = [&](int feat) __attribute__((always_inline)) { return *(const float*)(row + ((feat * 4) ^ sw)); };
    float g0 = 0.f, g1 = 0.f, g2 = 0.f;
    if (w == 0) { g0 = eg(0); g1 = eg(1); g2 = eg(2); }
    constexpr int half = N_DIRS * NF;
    for (int d = w; d < N_DIRS; d += T::NW) {
      const float dx = kDirs[0][d], dy = kDirs[1][d], dz = kDirs[2][d];
      const float proj = y0 * dx + y1 * dy + y2 * dz;
      float fr = 1.f, c = 0.f;
#pragma unroll
      for (int f = 0; f < NF; ++f) {
        const float xb = proj * fr;
        // d sin(xb)/d proj = cos(xb) fr ;  d sin(xb + pi/2)/d proj = cos(xb + pi/2) fr
        c += (__cosf(xb) * eg(3 + d * NF + f) + __cosf(xb + kHalfPi) * eg(3 + half + d * NF + f)) * fr;
        fr *= 2.f;
      }
      g0 += c * dx; g1 += c * dy; g2 += c * dz;
    }
    float* dst = part + ((h * T::NPART + w) * HB + lane) * 4;
    dst[0] = g0; dst[1] = g1; dst[2] = g2;
  }
  lds_barrier();
  TS();
  // ------------------------------------------------------------------ loss + adjoints (one thread per point; chain.hip's loss stage)
  float lsum[4] = {0.f, 0.f, 0.f, 0.f};
  if (tid < BM) {
    const int64_t n = n0 + tid;
    float e0 = 0.f, e1 = 0.f, e2 = 0.f;
#pragma unroll
    for (int k = 0; k < T::NPART; ++k) {
      const float* s = part + ((lh * T::NPART + k) * HB + lpt) * 4;
      e0 += s[0]; e1 += s[1]; e2 += s[2];
    }
    // g_x = scale_input * R^T g_x'
    const float si = L.scale_input;
    const float gx = si * (L.T[0] * e0 + L.T[4] * e1 + L.T[8] * e2);
    const float gy = si * (L.T[1] * e0 + L.T[5] * e1 + L.T[9] * e2);
    const float gz = si * (L.T[2] * e0 + L.T[6] * e1 + L.T[10] * e2);
    if (p.sdf_grad && n < P) { p.sdf_grad[n * 3] = gx; p.sdf_grad[n * 3 + 1] = gy; p.sdf_grad[n * 3 + 2] = gz; }
    float sbar = 0.f, bx = 0.f, by = 0.f, bz = 0.f;
    if (n < P) {
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
    // gbar in x' space: x' = si (R x + t)  =>  gbar_x' = si * R gbar_x
    gbs[tid * 4] = si * (L.T[0] * bx + L.T[1] * by + L.T[2] * bz);
    gbs[tid * 4 + 1] = si * (L.T[4] * bx + L.T[5] * by + L.T[6] * bz);
    gbs[tid * 4 + 2] = si * (L.T[8] * bx + L.T[9] * by + L.T[10] * bz);
    gbs[tid * 4 + 3] = sbar * so;
    // per-half loss / sbar sums (waves 0 and 1 hold the two halves): deterministic butterfly, lane 0 writes
    float v5[5] = {lsum[0], lsum[1], lsum[2], lsum[3], sbar * so};
#pragma unroll
    for (int k = 0; k < 5; ++k) {
      float v = half_wave_sum(v5[k]);
      v += __shfl_xor(v, 32, 64);
      if (lane == 0) red[lh * 8 + k] = v;
    }
  }
  lds_barrier();
  if (tid < 2 && (tid == 0 || validB)) {   // thread h writes half h's row
    const int h = tid;
    float* wl = p.wg_loss + (tile0 + h) * 8;
#pragma unroll
    for (int k = 0; k < 4; ++k) wl[k] = red[h * 8 + k];
    const int64_t rem = P - (n0 + h * HB);
    wl[4] = (float)(rem < HB ? rem : HB);
    (h ? vec1 : vec0)[NL * HD + 2 * HD] = red[h * 8 + 4];   // d b_out = sum sbar*so
  }
  TS();
  // ------------------------------------------------------------------ Ebar = J_pe gbar  -> region 2 of both halves (fp16)
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const int q = h * HB + lane;
    const float y0 = xs[q * 4], y1 = xs[q * 4 + 1], y2 = xs[q * 4 + 2];
    const float b0 = gbs[q * 4], b1 = gbs[q * 4 + 1], b2 = gbs[q * 4 + 2];
    char* row = smem + h * T::HALFB + lane * ROWB;
    auto put = [&](int feat, float v) __attribute__((always_inline)) { *(hT*)(row + (((HD + feat) * 2) ^ sw)) = (hT)v; };
    if (w == T::NW - 1) {
      put(0, b0); put(1, b1); put(2, b2);
      for (int f = L.E; f < EP; ++f) put(f, 0.f);
    }
    for (int d = w; d < N_DIRS; d += T::NW) {
      const float dx = kDirs[0][d], dy = kDirs[1][d], dz = kDirs[2][d];
      const float proj = y0 * dx + y1 * dy + y2 * dz;
      const float c = b0 * dx + b1 * dy + b2 * dz;
      float fr = 1.f;
#pragma unroll
      for (int f = 0; f < NF; ++f) {
        const float xb = proj * fr;
        put(3 + d * NF + f, __cosf(xb) * fr * c);
        put(3 + N_DIRS * NF + d * NF + f, __cosf(xb + kHalfPi) * fr * c);
        fr *= 2.f;
      }
    }
  }
  lds_barrier();
  TS();
  spill_region(0, HD, p.sp.GB[0]);
  spill_region(1, HD, p.sp.GB[0]);
  TS();
  // ------------------------------------------------------------------ adjoint and reverse sweeps
  tp_static_for([&](auto kc) __attribute__((always_inline)) { unit_pair(std::integral_constant<int, decltype(kc)::value + KMID>{}); },
                std::make_integer_sequence<int, NU - KMID>{});
  {   // epilogue of half B, the last unit
    const Unit u = bwd_unit(1);
    stage(std::integral_constant<int, UK_NONE>{}, std::integral_constant<int, EK_REV>{}, std::false_type{}, accA, accB, PreB, u, u.soff0, 0,
          eargs_k(NU - 1, 1), pargs_k(NU, 1));
  }
  TS.wall(1);
}

bool train_pair_supported(const NetLayout& l) {
  return l.HD == 256 && l.EP == 256 && l.L == 6 && l.cat == 3 && l.n_freqs == 6 && l.fwd_f16 && !l.fwd_x2_all && l.bwd_f16;
}

int launch_train_pair(const ChainParams& p, int64_t nTiles, hipStream_t st) {
  const int64_t nPairs = (nTiles + 1) / 2;
  if (nPairs <= 0) return ISDF_OK;
  typedef TrainPairTile T;
  auto k = p.lay.fwd_x2 ? train_pair_kernel<2> : train_pair_kernel<1>;
  if (hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, T::LDS_BYTES) != hipSuccess) return ISDF_EHIP;
  hipLaunchKernelGGL(k, dim3((unsigned)nPairs), dim3(T::NW * 64), T::LDS_BYTES, st, p);
  return isdf_launch_status();
}

}  // namespace isdf
