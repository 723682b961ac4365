// Pair-tile training kernel (K2, <HD = 256, EP = 256>, train mode): the same mathematics and the same HBM formats as
// chain.hip (one 64-point tile = one block of the spill buffer, one row of vec_part / wg_loss), but ONE workgroup per CU
// that owns TWO adjacent tiles ("halves") and software-pipelines them against each other:
//
//      stage k   :  epilogue of half 0, unit u   then  GEMM of half 1, unit u        (barrier)
//      stage k+1 :  epilogue of half 1, unit u   then  GEMM of half 0, unit u + 1    (barrier)
//
// Replaces, for two tiles of 64 points (same reference lines as chain.hip):
//   embedding.PostionalEncoding.forward   isdf/modules/embedding.py:95-111
//   SDFMap.forward                        isdf/modules/fc_map.py:94-111
//   fc_map.gradient (autograd.grad)       isdf/modules/fc_map.py:12-22
//   loss.bounds_ray / sdf_loss / tot_loss isdf/modules/loss.py:13-22,122-205
//   eikonal + normal terms                isdf/modules/trainer.py:814-830
//   the activation side of total_loss.backward()   trainer.py:981
//
// Why: with two independent 8-wave workgroups per CU (chain.hip) every wave has 128 VGPRs, each
// workgroup streams its own copy of every weight matrix from L2 (the CU's 64 B/clk vector-memory path is ~47 % busy
// with weights + spills), and nothing can be requested ahead of a GEMM (no registers), so every epilogue starts with an
// exposed HBM round trip and every GEMM with an exposed L2 round trip.  Here a wave has 256 VGPRs:
//   * the weight fragments of a layer are fetched ONCE for both halves and sit in a 64-VGPR window (gemm_half); the
//     second half re-requests every register with the NEXT unit's fragment as soon as its MFMAs are issued: no GEMM
//     starts with an L2 round trip, and (vmcnt retires in order) no GEMM waits behind its own stage's spill traffic;
//   * the spilled tensors an epilogue re-reads are requested a GEMM and a barrier ahead (32 VGPRs);
//   * a stage's epilogue and GEMM work on different halves of the tile: one barrier per GEMM instead of two, and one
//     accumulator for both halves (the epilogue drains it before the GEMM refills it).
#include "chain_dev.h"

namespace isdf {

#ifndef ISDF_DEBUG_HOOKS
#define ISDF_DEBUG_HOOKS 0
#endif
#ifndef ISDF_PAIR_INTERLEAVE
#define ISDF_PAIR_INTERLEAVE 0   // 1: a stage's epilogue (VALU, other half) and GEMM (MFMA) are ONE scheduling region, interleaved by
#endif                           //    sched_group_barrier: the epilogue's VALU issues in the shadow of the MFMAs of the same wave
#if ISDF_PAIR_INTERLEAVE
#define PAIR_SB() ((void)0)
// a VOLATILE asm is ordered against every memory operation: one in front of the GEMM would pin the whole epilogue (its
// stores, hence its arithmetic) ahead of the GEMM's first LDS read
#define PAIR_OPAQUE(x) asm("" : "+v"(x))
#else
#define PAIR_SB() __builtin_amdgcn_sched_barrier(0)
#define PAIR_OPAQUE(x) asm volatile("" : "+v"(x))
#endif
#ifndef ISDF_PAIR_VALU_PER_MFMA
#define ISDF_PAIR_VALU_PER_MFMA 12
#endif
#ifndef GEMM_PAIR_LDS_DEPTH
#define GEMM_PAIR_LDS_DEPTH 2
#endif
#ifndef ISDF_NT_DW_TENSORS
#define ISDF_NT_DW_TENSORS 0
#endif
#ifndef ISDF_NT_P
#define ISDF_NT_P 1
#endif

template <int HD> struct PairTile {
  static constexpr int BM = 128;               // points per workgroup = 2 tiles of TILE_PTS
  static constexpr int PB = 2;                 // 32-point blocks per half
  static constexpr int NW = 8;                 // waves; wave w owns features 32w .. 32w+31 of every GEMM
  static constexpr int ROWB = 4 * HD;          // bytes per LDS row: [HD | HD] 16-bit elements
  static constexpr int XBYTES = BM * ROWB;
  static constexpr int NPART = (NW * 64) / BM; // direction slices of the PE-shaped stages
  static constexpr int MAXLP = 8;              // hidden layers whose biases are staged in LDS
  static constexpr int OFF_XS = XBYTES;                        // [BM][4] x'
  static constexpr int OFF_PART = OFF_XS + BM * 16;            // [NPART][BM][4] partial g
  static constexpr int OFF_GB = OFF_PART + NPART * BM * 16;    // [BM][4] gbar (x' space), [3] = sbar*so
  static constexpr int OFF_RAW = OFF_GB + BM * 16;             // [NW][BM] per-wave partial of the output layer
  static constexpr int OFF_BIAS = OFF_RAW + NW * BM * 4;       // [MAXLP][HD] biases, then [HD] w_out
  static constexpr int LDS_BYTES = OFF_BIAS + (MAXLP + 1) * HD * 4;
};

struct PRef { int soff; };   // byte offset of a wave's 32-row slice of a packed matrix in the shadow buffer

// request k-step `ks` of the wave's slice into one fragment register
__device__ __forceinline__ uint4 load_wfrag(rsrc_t rw, PRef r, int lane16, int ks) {
  return bload16<0>(rw, lane16 + (ks & 3) * 1024, r.soff + (ks >> 2) * 4096);
}

// acc[2 point blocks of a half] = W[32 feats][K] * X[64 pts][K]^T.  W[] is the wave's fragment window of 16 k-steps
// (64 VGPRs = a whole K = 256 slice): on entry it holds k-steps 0..15 of `cur`, requested during an EARLIER stage.
// vmcnt retires in order, so a fragment requested inside a stage could only be waited for behind that stage's spill
// stores and re-read requests; with the whole slice resident a K = 256 GEMM waits for nothing that was issued in its
// own stage.  STAGE 0 (first half to use the matrix): the window stays (K = 512: k-steps 16..31 stream through it and
// 0..15 are re-requested).  STAGE 1 (second half): registers 0..7 are re-requested with the NEXT unit's fragments as soon
// as their last MFMAs are issued, registers 8..15 by the caller at the start of the next stage (16 loads per wave inside
// one GEMM = 128 KB per CU through the 64 B/clk vector-memory path: the GEMM took 3-6 k cycles instead of 1.4 k).
// STAGE 2: second half, nothing follows.
// Both halves accumulate in the same k order: a point's result does not depend on the tile or half it lands in.
template <bool F16, int KSTEPS, int STAGE, int ROWB>
__device__ __forceinline__ void gemm_half(f32x16 (&acc)[2], uint4 (&W)[16], rsrc_t rw, PRef cur, PRef nxt, const char* xh,
                                          int colByteBase, int lane) {
  typedef typename Op<F16>::v8 v8;
  static_assert(KSTEPS == 16 || KSTEPS == 32, "K = 256 or 512");
  const int j = lane & 31, hi = lane >> 5;
  const int xlane = j * ROWB + ((hi * 16) ^ ((j & 15) << 4));
  const int lane16 = lane * 16;
#pragma unroll
  for (int pb = 0; pb < 2; ++pb)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[pb][r] = 0.f;
  constexpr int D = GEMM_PAIR_LDS_DEPTH;   // k-steps of activation-operand reads in flight ahead of the MFMAs
  v8 b[D + 1][2];
  int xch[KSTEPS / 8];
#pragma unroll
  for (int ch = 0; ch < KSTEPS / 8; ++ch) {
    xch[ch] = xlane + ch * 256 + colByteBase;
    PAIR_OPAQUE(xch[ch]);   // keep the per-k-step addresses from being hoisted out of the layer loops
  }
  auto readb = [&](int ks, v8 (&o)[2]) {
#pragma unroll
    for (int pb = 0; pb < 2; ++pb) o[pb] = __builtin_bit_cast(v8, *(const uint4*)(xh + ((xch[ks >> 3] ^ ((ks & 7) * 32)) + pb * 32 * ROWB)));
  };
#pragma unroll
  for (int ks = 0; ks < D; ++ks) readb(ks, b[ks % (D + 1)]);
  PAIR_SB();
#pragma unroll
  for (int ks = 0; ks < KSTEPS; ++ks) {
    const int r = ks & 15;
    if (ks + D < KSTEPS) readb(ks + D, b[(ks + D) % (D + 1)]);
#pragma unroll
    for (int pb = 0; pb < 2; ++pb) acc[pb] = Op<F16>::mfma(__builtin_bit_cast(v8, W[r]), b[ks % (D + 1)][pb], acc[pb]);
    const int tgt = ks + 16;
    if (tgt < KSTEPS) W[r] = load_wfrag(rw, cur, lane16, tgt);
    else if (STAGE == 1) { if (r < 8) W[r] = load_wfrag(rw, nxt, lane16, r); }   // upper half: loadW_hi at the start of the next stage
    else if (STAGE == 0 && KSTEPS > 16) W[r] = load_wfrag(rw, cur, lane16, tgt - KSTEPS);
    PAIR_SB();
  }
}

// NL / CAT: hidden layers and the index of the cat layer as COMPILE-TIME constants (0: read them from the layout).  With
// constants the layer loops unroll, every stage is one basic block (no per-layer branches) and ISDF_PAIR_INTERLEAVE can
// schedule a stage's epilogue and GEMM as one region.
template <int HD, bool F16, int NL, int CAT>
__global__ __launch_bounds__(512, 2) void chain_pair_kernel(const ChainParams p) {
  typedef PairTile<HD> T;
  constexpr int BM = T::BM, PB = T::PB, ROWB = T::ROWB, NPART = T::NPART, HB = TILE_PTS;
  static_assert(TILE_PTS == 64, "a half is one 64-point tile of the spill buffer");
  // The two halves of the activation tile are SEPARATE LDS objects: a stage's epilogue writes one and its GEMM reads the
  // other, and only distinct objects let the compiler see that those accesses cannot alias (ISDF_PAIR_INTERLEAVE moves
  // the GEMM's reads above the epilogue's writes).
  __shared__ __attribute__((aligned(1024))) char X0s[TILE_PTS * T::ROWB];
  __shared__ __attribute__((aligned(1024))) char X1s[TILE_PTS * T::ROWB];
  __shared__ __attribute__((aligned(16))) char smallL[T::LDS_BYTES - T::XBYTES];
  auto Xh = [&](int h) -> char* { return h ? X1s : X0s; };
  char* const sm0 = smallL - T::XBYTES;   // the OFF_* constants count from the start of the (former single) array
  float* xs = (float*)(sm0 + T::OFF_XS);
  float* part = (float*)(sm0 + T::OFF_PART);
  float* gbs = (float*)(sm0 + T::OFF_GB);
  float* rawL = (float*)(sm0 + T::OFF_RAW);
  float* biasL = (float*)(sm0 + T::OFF_BIAS);
  float* woutL = biasL + T::MAXLP * HD;

  const NetLayout& L = p.lay;
  const int tid = threadIdx.x;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  int lane, j, hi, lane16, xw;
  auto refresh = [&] {   // lane constants are re-derived per phase (see chain.hip)
    int t = tid;
    PAIR_OPAQUE(t);
    lane = t & 63; j = lane & 31; hi = lane >> 5; lane16 = lane * 16;
    xw = j * ROWB + 8 * hi + (((j & 15) << 4) ^ ((w & 3) * 64)) + (w >> 2) * 256;
  };
  refresh();
  const int64_t P = p.n_valid ? (int64_t)(*p.n_valid) * p.S : p.n_points_host;
  const int64_t n0 = (int64_t)blockIdx.x * BM;
  if (n0 >= P) return;
  const int nf = L.n_freqs;
  const float so = L.scale_output;
  const int nL = NL ? NL : L.L;
  const int catL = NL ? CAT : L.cat;
#if ISDF_DEBUG_HOOKS   // development build: wave 0 of workgroup 100 stamps every stage end (tools/timeline.py)
  int tsn = 0;
  auto TS = [&]() {   // 32-bit stamps: slots 0..255 of the 1 KB stamp area
    if (p.dbg_times && blockIdx.x == 100 && tid == 0 && tsn < 256) ((unsigned*)p.dbg_times)[tsn] = (unsigned)__builtin_amdgcn_s_memtime();
    ++tsn;
  };
#else
  auto TS = [] {};
#endif
  TS();

  const int64_t setFwdA = L.setFwdA, setFwdB = L.setFwdB, setBwdA = L.setBwdA, setBwdB = L.setBwdB;
  uint16_t* spillPair = p.spill + (int64_t)blockIdx.x * 2 * p.sp.tileStride;
  const int halfSpillBytes = (int)(p.sp.tileStride * 2);
  const rsrc_t rsW = make_rsrc(p.shadow, 0x7fffffffu);
  const rsrc_t rsS = make_rsrc(spillPair, (uint32_t)(2 * halfSpillBytes));
  float* vecPair = p.vec_part + (int64_t)blockIdx.x * 2 * p.vecStride;
  const rsrc_t rsV = make_rsrc(vecPair, (uint32_t)p.vecStride * 8u);
  // byte offset of this wave's first piece of a spilled tensor of half h (frag16 order, chain.hip)
  auto sbase = [&](int64_t tensorOff, int h) {
    return __builtin_amdgcn_readfirstlane((int)(tensorOff * 2) + w * (PB * 2) * 1024 + h * halfSpillBytes);   // wave-uniform: the store's SGPR offset
  };
  auto vec_store = [&](float v, int elemUniform, int h) {   // per-tile partial of a bias / out-layer gradient entry
    v = half_wave_sum(v);
    // branch-free: only lane j == 0 of a half-wave owns the entry, the other lanes' stores fall outside the descriptor
    // and are dropped by the bounds check (an exec-masked branch per entry makes the compiler's vmcnt bookkeeping
    // pessimistic at every join: the next GEMM then waited for this stage's re-read requests)
    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), rsV, j == 0 ? 16 * hi : 0x40000000, (h * p.vecStride + elemUniform) * 4, 0);
  };

  // biases and w_out -> LDS (the epilogues read them with two ds_read_b128 instead of an L2 round trip)
  for (int li = 0; li < nL; ++li)
    for (int q = tid; q < HD; q += T::NW * 64) biasL[li * HD + q] = p.params[L.offB[li] + q];
  for (int q = tid; q < HD; q += T::NW * 64) woutL[q] = p.params[L.offWout + q];

  // weight fragment window; first unit = forward layer 0
  auto wref = [&](int64_t set, int64_t matOff, int kp) { return PRef{(int)((set + matOff) * 2) + w * (kp / 16) * 1024}; };
  auto fwdRef = [&](int64_t set, int li) { return wref(set, L.fwdMat[li], li == catL ? 2 * HD : HD); };
  auto r1Ref = [&](int li) { return wref(setBwdA, L.bwdMat[li], HD); };
  auto r2Ref = [&](int li) { return wref(setBwdB, L.bwdMat[li + 1], HD); };
  const PRef gRef = wref(setBwdA, L.bwdG, 2 * HD);
  uint4 W[16];
  {
    const PRef r0 = fwdRef(setFwdA, 0);
#pragma unroll
    for (int s = 0; s < 16; ++s) W[s] = load_wfrag(rsW, r0, lane16, s);
  }

  // ------------------------------------------------------------------ PE stage (both halves; embedding.py:95-111)
  {
#if ISDF_PE_MAP   // a wave = (BM / NW points) x (direction slices): rows 1 KB apart land on 8 banks, so 64 points per wave was 8-way conflicted
    const int pt = (tid % (BM / T::NW)) + (BM / T::NW) * (tid / 64), prt = (tid % 64) / (BM / T::NW);
#else
    const int pt = tid & (BM - 1), prt = tid / BM;
#endif
    const int64_t n = n0 + pt;
    float x0 = 0.f, x1 = 0.f, x2 = 0.f;
    if (n < P) { x0 = p.pts[n * 3]; x1 = p.pts[n * 3 + 1]; x2 = p.pts[n * 3 + 2]; }
    const float y0 = (L.T[0] * x0 + L.T[1] * x1 + L.T[2] * x2 + L.T[3]) * L.scale_input;
    const float y1 = (L.T[4] * x0 + L.T[5] * x1 + L.T[6] * x2 + L.T[7]) * L.scale_input;
    const float y2 = (L.T[8] * x0 + L.T[9] * x1 + L.T[10] * x2 + L.T[11]) * L.scale_input;
    typedef typename Op<F16>::e opT;
    char* row = Xh(pt >> 6) + (pt & 63) * ROWB;
    auto put = [&](int feat, float v) {
      *(opT*)(row + swz(pt, (HD + feat) * 2)) = (opT)v;        // region 2: forward operand
      *(__bf16*)(row + swz(pt, feat * 2)) = (__bf16)v;         // region 1: bf16 copy staged for the spill (dW operand A_0)
    };
    if (prt == 0) {
      xs[pt * 4] = y0; xs[pt * 4 + 1] = y1; xs[pt * 4 + 2] = y2;
      put(0, y0); put(1, y1); put(2, y2);
      for (int f = L.E; f < HD; ++f) put(f, 0.f);
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
  // copy a [64][HD] 16-bit region of both halves to their spill tiles in frag16 order
  auto spill_region = [&](int colElemBase, int64_t tensorOff) {
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int pb = 0; pb < PB; ++pb)
#pragma unroll
        for (int qp = 0; qp < 2; ++qp) {
          const int lb = (xw ^ (32 * qp)) + colElemBase * 2 + pb * 32 * ROWB;
          const uint2 lo = *(const uint2*)(Xh(h) + lb);
          const uint2 hi2 = *(const uint2*)(Xh(h) + (lb ^ 16));
          u32x4 v; v[0] = lo.x; v[1] = lo.y; v[2] = hi2.x; v[3] = hi2.y;
          __builtin_amdgcn_raw_buffer_store_b128(v, rsS, lane16 + sbase(tensorOff, h) + (pb * 2 + qp) * 1024, 0, kAuxNT);
        }
  };
  refresh();
  TS();
  spill_region(0, p.sp.A[0]);
  TS();

  // ------------------------------------------------------------------ building blocks
  f32x16 acc[PB];   // ONE accumulator: a stage's epilogue consumes it before the stage's GEMM (other half) refills it
  struct Pre { uint4 v[2][PB]; };   // one re-read tensor of a half: [qp][pb], 16 VGPRs
  Pre preA, preB;   // requested between a stage's epilogue and its GEMM, read by the NEXT stage's epilogue
  auto prefetch = [&](int64_t tensorOff, int h, Pre& pr) {
    const int sb = sbase(tensorOff, h);
#pragma unroll
    for (int qp = 0; qp < 2; ++qp)
#pragma unroll
      for (int pb = 0; pb < PB; ++pb) pr.v[qp][pb] = bload16<kAuxNT>(rsS, lane16 + (pb * 2 + qp) * 1024, sb);
  };
  // (The NEXT stage's re-read tiles are requested between a stage's epilogue and its GEMM.  Spreading the requests over
  // the epilogue, one piece behind each unpacked piece, was measured: 231 vs 208 us.)
  // An epilogue pins its re-read tiles first: arithmetic that depends only on them (unpack, -beta*a) must not be
  // scheduled ahead of the stage (the compiler moved it in front of the previous GEMM, spilled the results and re-read
  // them from scratch behind s_waitcnt vmcnt(0), which also drains the requests that were just issued).
  auto pin = [&](Pre& pr) {
#pragma unroll
    for (int qp = 0; qp < 2; ++qp)
#pragma unroll
      for (int pb = 0; pb < PB; ++pb)
        asm volatile("" : "+v"(pr.v[qp][pb].x), "+v"(pr.v[qp][pb].y), "+v"(pr.v[qp][pb].z), "+v"(pr.v[qp][pb].w));
  };
  auto load_tile8 = [&](const Pre& pr, int pb, int qp, float (&o)[8]) {
    const uint4 u = pr.v[qp][pb];
    float a[4], b[4];
    unpack4_bf16(make_uint2(u.x, u.y), a); unpack4_bf16(make_uint2(u.z, u.w), b);
#pragma unroll
    for (int e = 0; e < 4; ++e) { o[e] = a[e]; o[4 + e] = b[e]; }
  };
  // Spill stores are COMPILER-VISIBLE here (chain.hip hand-issues them): every s_waitcnt vmcnt(N) the compiler derives is
  // then exact -- an uncounted store in the queue makes each later wait stricter by one, i.e. a wait for an OLD weight
  // fragment would also wait for the stores and re-read requests issued after it.  The store-data hazard of chain_dev.h
  // belongs to the SGPR-soffset encoding (which the compiler believes exempt); with soffset = 0 and the tensor offset in
  // the VGPR offset the hazard recogniser inserts the wait state itself.
  auto vstore16 = [&](uint4 x, int voff, auto ntc) {
    u32x4 v; v[0] = x.x; v[1] = x.y; v[2] = x.z; v[3] = x.w;
    __builtin_amdgcn_raw_buffer_store_b128(v, rsS, voff, 0, decltype(ntc)::value ? kAuxNT : 0);
  };
  auto store_tile8 = [&](int64_t tensorOff, int h, int pb, int qp, const float (&v)[8]) {
    const uint2 a = pack4<false>(v[0], v[1], v[2], v[3]), b = pack4<false>(v[4], v[5], v[6], v[7]);
    vstore16(make_uint4(a.x, a.y, b.x, b.y), lane16 + sbase(tensorOff, h) + (pb * 2 + qp) * 1024, std::true_type{});
  };
  auto store_tile8_p = [&](int64_t tensorOff, int h, int pb, int qp, const float (&v)[8]) {
    const uint2 a = pack4<false>(v[0], v[1], v[2], v[3]), b = pack4<false>(v[4], v[5], v[6], v[7]);
    vstore16(make_uint4(a.x, a.y, b.x, b.y), lane16 + sbase(tensorOff, h) + (pb * 2 + qp) * 1024, std::integral_constant<bool, ISDF_NT_P != 0>{});
  };
  auto store_tile8_dw = [&](int64_t tensorOff, int h, int pb, int qp, const float (&v)[8]) {   // tensors only dw.hip re-reads
    const uint2 a = pack4<false>(v[0], v[1], v[2], v[3]), b = pack4<false>(v[4], v[5], v[6], v[7]);
    vstore16(make_uint4(a.x, a.y, b.x, b.y), lane16 + sbase(tensorOff, h) + (pb * 2 + qp) * 1024, std::integral_constant<bool, ISDF_NT_DW_TENSORS != 0>{});
  };
  auto put_x = [&](bool f16, int h, int pb, int qp, const float (&v)[8], int colElemBase) {
    uint2 a, b;
    if (f16) { a = pack4<true>(v[0], v[1], v[2], v[3]); b = pack4<true>(v[4], v[5], v[6], v[7]); }
    else { a = pack4<false>(v[0], v[1], v[2], v[3]); b = pack4<false>(v[4], v[5], v[6], v[7]); }
    const int lb = (xw ^ (32 * qp)) + colElemBase * 2 + pb * 32 * ROWB;
    *(uint2*)(Xh(h) + lb) = a;           // features f0 .. f0+3,  f0 = 32 w + 16 qp + 4 hi
    *(uint2*)(Xh(h) + (lb ^ 16)) = b;    // features f0+8 .. f0+11
  };
  // 8 fp32 values vec[f0 + {0..3, 8..11}] of an LDS-staged parameter vector
  auto ld_vec8 = [&](const float* vec, int qp, float (&o)[8]) {
    const float4 a = *(const float4*)(vec + w * 32 + 16 * qp + 4 * hi), b = *(const float4*)(vec + w * 32 + 16 * qp + 4 * hi + 8);
    o[0] = a.x; o[1] = a.y; o[2] = a.z; o[3] = a.w; o[4] = b.x; o[5] = b.y; o[6] = b.z; o[7] = b.w;
  };
  auto xhalf = [&](int h) { return (const char*)Xh(h); };

  // GEMMs (half H; STAGE_B: this is the second half to use the matrix, refill the window from `nxt`)
  auto G_fwd = [&](int H, auto Bc, auto F16c, int64_t set, int li, PRef nxt) {   // forward-orientation matrix of layer li
    constexpr int SB = decltype(Bc)::value; constexpr bool OPF = decltype(F16c)::value;
    refresh();
    const PRef cur = fwdRef(set, li);
    if (li == catL) gemm_half<OPF, 32, SB, ROWB>(acc, W, rsW, cur, nxt, xhalf(H), 0, lane);            // [a | emb], K = 2 HD
    else gemm_half<OPF, 16, SB, ROWB>(acc, W, rsW, cur, nxt, xhalf(H), li == 0 ? HD * 2 : 0, lane);     // layer 0 reads region 2
  };
  auto G_sq = [&](int H, auto Bc, auto F16c, PRef cur, PRef nxt) {   // K = HD from region 1
    constexpr int SB = decltype(Bc)::value;
    refresh();
    gemm_half<decltype(F16c)::value, 16, SB, ROWB>(acc, W, rsW, cur, nxt, xhalf(H), 0, lane);
  };
  auto G_g = [&](int H, auto Bc, PRef nxt) {   // Eg = [W_in^T | W_cat[:,HD:]^T] [p_0 ; p_cat]
    constexpr int SB = decltype(Bc)::value;
    refresh();
    gemm_half<F16, 32, SB, ROWB>(acc, W, rsW, gRef, nxt, xhalf(H), 0, lane);
  };
  const std::integral_constant<int, 0> stA{}; const std::integral_constant<int, 1> stB{}; const std::integral_constant<int, 2> stBlast{};
  const std::integral_constant<bool, F16> opF{}; const std::false_type opB{};

  // ---- epilogues
  auto E_fwd = [&](int H, int li) {
    refresh();
    if (li != nL - 1) {
#pragma unroll
      for (int qp = 0; qp < 2; ++qp) {
        float bv[8];
        ld_vec8(biasL + li * HD, qp, bv);
#pragma unroll
        for (int pb = 0; pb < PB; ++pb) {
          float a[8];
#pragma unroll
          for (int e = 0; e < 8; ++e) a[e] = softplus_f(acc[pb][8 * qp + e] + bv[e]);
          store_tile8(p.sp.A[li + 1], H, pb, qp, a);
          put_x(F16, H, pb, qp, a, 0);
          PAIR_SB();
        }
      }
    } else {
      float rawp[PB];
#pragma unroll
      for (int pb = 0; pb < PB; ++pb) rawp[pb] = 0.f;
#pragma unroll
      for (int qp = 0; qp < 2; ++qp) {
        float bv[8], wv[8];
        ld_vec8(biasL + li * HD, qp, bv);
        ld_vec8(woutL, qp, wv);
#pragma unroll
        for (int pb = 0; pb < PB; ++pb) {
          float a[8], pl[8];
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            float s1;
            a[e] = softplus_s1(acc[pb][8 * qp + e] + bv[e], s1);
            rawp[pb] += wv[e] * a[e];
            pl[e] = so * wv[e] * s1;   // p_L = q_L * sigma'(z_L), q_L = so * w_out
          }
          store_tile8(p.sp.A[li + 1], H, pb, qp, a);
          put_x(F16, H, pb, qp, pl, 0);
          store_tile8_p(p.sp.P[li], H, pb, qp, pl);
          PAIR_SB();
        }
      }
#pragma unroll
      for (int pb = 0; pb < PB; ++pb) {
        const float v = rawp[pb] + __shfl_xor(rawp[pb], 32, 64);   // add the two feature halves
        if (hi == 0) rawL[w * BM + H * HB + pb * 32 + j] = v;
      }
    }
  };
  auto E_r1 = [&](int H, int li, Pre& pA) {   // p_{li-1} = (p_li W_li) * sigma'(z_{li-1});  pA = A[li]
    refresh();
    pin(pA);
    const bool toR2 = (li - 1 == catL);
#pragma unroll
    for (int qp = 0; qp < 2; ++qp)
#pragma unroll
      for (int pb = 0; pb < PB; ++pb) {
        float a[8], pv[8];
        load_tile8(pA, pb, qp, a);
#pragma unroll
        for (int e = 0; e < 8; ++e) pv[e] = acc[pb][8 * qp + e] * s1_from_a(a[e]);
        put_x(F16, H, pb, qp, pv, 0);
        if (toR2) put_x(F16, H, pb, qp, pv, HD);
        store_tile8_p(p.sp.P[li - 1], H, pb, qp, pv);
        PAIR_SB();
      }
  };
  auto E_g = [&](int H) {   // Eg -> fp32 [64][HD] over the half's (now idle) rows
    refresh();
#pragma unroll
    for (int pb = 0; pb < PB; ++pb) {
      const int row = pb * 32 + j;
#pragma unroll
      for (int rq = 0; rq < 4; ++rq) {
        const int f0 = w * 32 + 8 * rq + 4 * hi;
        *(float4*)(Xh(H) + row * ROWB + swz(row, f0 * 4)) =
            make_float4(acc[pb][4 * rq], acc[pb][4 * rq + 1], acc[pb][4 * rq + 2], acc[pb][4 * rq + 3]);
      }
    }
  };
  auto E_adj = [&](int H, int li, Pre& pA, Pre& pB) {   // pA = A[li+1], pB = P[li]
    refresh();
    pin(pA); pin(pB);
#pragma unroll
    for (int qp = 0; qp < 2; ++qp)
#pragma unroll
      for (int pb = 0; pb < PB; ++pb) {
        float a[8], pv[8], qb[8], inj[8];
        load_tile8(pA, pb, qp, a);
        load_tile8(pB, pb, qp, pv);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float u = acc[pb][8 * qp + e];
          const float s1 = s1_from_a(a[e]);
          qb[e] = u * s1;
          inj[e] = kBeta * u * pv[e] * (1.f - s1);   // u * q * sigma''(z),  q*sigma' = p
        }
        store_tile8(p.sp.INJ[li], H, pb, qp, inj);
        put_x(false, H, pb, qp, qb, 0);
        store_tile8_dw(p.sp.GB[li + 1], H, pb, qp, qb);
        PAIR_SB();
      }
  };
  auto E_adj_top = [&](int H, Pre& pA, Pre& pB) {   // top layer: also the top of the ordinary reverse sweep (chain.hip)
    refresh();
    pin(pA); pin(pB);
    const int li = nL - 1;
#pragma unroll
    for (int qp = 0; qp < 2; ++qp) {
      float qsum[8], bsum[8], wsum[8], wv[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) { qsum[e] = 0.f; bsum[e] = 0.f; wsum[e] = 0.f; }
      ld_vec8(woutL, qp, wv);
#pragma unroll
      for (int pb = 0; pb < PB; ++pb) {
        float a[8], pv[8], zb[8];
        load_tile8(pA, pb, qp, a);
        load_tile8(pB, pb, qp, pv);
        const float sb = gbs[(H * HB + pb * 32 + j) * 4 + 3];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float u = acc[pb][8 * qp + e];
          const float s1 = s1_from_a(a[e]);
          qsum[e] += u * s1;
          zb[e] = sb * wv[e] * s1 + kBeta * u * pv[e] * (1.f - s1);
          bsum[e] += zb[e];
          wsum[e] += sb * a[e];
        }
        store_tile8_dw(p.sp.ZB[li], H, pb, qp, zb);
        put_x(false, H, pb, qp, zb, 0);
        PAIR_SB();
      }
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int f = w * 32 + 16 * qp + (e & 3) + 8 * (e >> 2);   // + 4*hi in the lane offset
        vec_store(so * qsum[e], nL * HD + f, H);        // d w_out += so * sum_pts qbar_L
        vec_store(wsum[e], nL * HD + HD + f, H);        // d w_out += sum_pts sbar*so * a_L
        vec_store(bsum[e], li * HD + f, H);
      }
    }
  };
  auto E_r2 = [&](int H, int li, Pre& pA, Pre& pB) {   // pA = A[li+1], pB = INJ[li]
    refresh();
    pin(pA); pin(pB);
#pragma unroll
    for (int qp = 0; qp < 2; ++qp) {
      float bsum[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) bsum[e] = 0.f;
#pragma unroll
      for (int pb = 0; pb < PB; ++pb) {
        float a[8], inj[8], zb[8];
        load_tile8(pA, pb, qp, a);
        load_tile8(pB, pb, qp, inj);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          zb[e] = acc[pb][8 * qp + e] * s1_from_a(a[e]) + inj[e];
          bsum[e] += zb[e];
        }
        store_tile8_dw(p.sp.ZB[li], H, pb, qp, zb);
        if (li > 0) put_x(false, H, pb, qp, zb, 0);
        PAIR_SB();
      }
#pragma unroll
      for (int e = 0; e < 8; ++e) vec_store(bsum[e], li * HD + w * 32 + 16 * qp + (e & 3) + 8 * (e >> 2), H);
    }
  };

  // A stage = [epilogue of the previous GEMM] [requests for the NEXT stage's epilogue] [GEMM of the other half] barrier.
  // The epilogue and the GEMM of a stage touch different halves of the tile, so ONE barrier per GEMM is enough (chain.hip
  // needs two) and one accumulator serves both halves; the requests have the GEMM and the barrier to land.
#if ISDF_PAIR_INTERLEAVE
  // per MFMA: its operand read, then a slice of the epilogue; every 4th MFMA also one LDS write and one memory instruction
  auto stage_pattern = [] {
#pragma unroll
    for (int i = 0; i < 64; ++i) {
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                        // MFMA
      __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                        // DS read
      __builtin_amdgcn_sched_group_barrier(0x002, ISDF_PAIR_VALU_PER_MFMA, 0);  // VALU
      if ((i & 3) == 3) {
        __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);                      // DS write
        __builtin_amdgcn_sched_group_barrier(0x010, 1, 0);                      // VMEM
      }
    }
  };
#define PAIR_STAGE(PRE, EPI, PF, GEMM) do { PRE; __builtin_amdgcn_sched_barrier(0); EPI; PF; GEMM; stage_pattern(); TS(); __builtin_amdgcn_sched_barrier(0); lds_barrier(); TS(); } while (0)
#else
#define PAIR_STAGE(PRE, EPI, PF, GEMM) do { PRE; EPI; TS(); PF; GEMM; TS(); lds_barrier(); TS(); } while (0)
#endif
  auto loadW_hi = [&](PRef r) {   // upper half of the fragment window (see gemm_half STAGE 1)
#pragma unroll
    for (int k = 8; k < 16; ++k) W[k] = load_wfrag(rsW, r, lane16, k);
  };
#define NOP (void)0
  const PRef none{0};

  // ------------------------------------------------------------------ forward (fc_map.py:94-111)
  G_fwd(0, stA, opF, setFwdA, 0, none);
  TS();
  lds_barrier();
  TS();
  // the loss stage's per-ray inputs (requested at the end of the first reverse sweep, consumed long after)
  float li_bnd = 0.f, li_c[3] = {0.f, 0.f, 0.f}, li_dz[2] = {0.f, 0.f}, li_t[3] = {0.f, 0.f, 0.f}, li_n[3] = {0.f, 0.f, 0.f};
  auto loss_inputs = [&] {
    if (tid >= BM) return;
    const int64_t n = n0 + tid;
    if (n >= P) return;
    const int64_t ray = n / p.S;
    if (p.loss.bounds_method == 0) {  // loss.py:13-22
      li_c[0] = p.dirsC[ray * 3]; li_c[1] = p.dirsC[ray * 3 + 1]; li_c[2] = p.dirsC[ray * 3 + 2];
      li_dz[0] = p.depth[ray]; li_dz[1] = p.z_vals[n];
      li_t[0] = p.dirsW[ray * 3]; li_t[1] = p.dirsW[ray * 3 + 1]; li_t[2] = p.dirsW[ray * 3 + 2];
    } else {
      li_bnd = p.pc_bounds[n];
      li_t[0] = p.pc_grad_vec[n * 3]; li_t[1] = p.pc_grad_vec[n * 3 + 1]; li_t[2] = p.pc_grad_vec[n * 3 + 2];
    }
    if (p.normals) { li_n[0] = p.normals[ray * 3]; li_n[1] = p.normals[ray * 3 + 1]; li_n[2] = p.normals[ray * 3 + 2]; }
  };
#pragma unroll
  for (int li = 0; li < nL; ++li) {
    const bool last = li == nL - 1;
    PAIR_STAGE(NOP, E_fwd(0, li), NOP, G_fwd(1, stB, opF, setFwdA, li, last ? r1Ref(nL - 1) : fwdRef(setFwdA, li + 1)));
    if (!last) PAIR_STAGE(loadW_hi(fwdRef(setFwdA, li + 1)), E_fwd(1, li), NOP, G_fwd(0, stA, opF, setFwdA, li + 1, none));
    else PAIR_STAGE(loadW_hi(r1Ref(nL - 1)), E_fwd(1, li), prefetch(p.sp.A[nL - 1], 0, preA), G_sq(0, stA, opF, r1Ref(nL - 1), none));   // first unit of the first reverse sweep
  }

  // ------------------------------------------------------------------ first reverse sweep (fc_map.py:12-22)
#pragma unroll
  for (int li = nL - 1; li >= 1; --li) {
    PAIR_STAGE(NOP, E_r1(0, li, preA), prefetch(p.sp.A[li], 1, preA), G_sq(1, stB, opF, r1Ref(li), li > 1 ? r1Ref(li - 1) : gRef));
    if (li > 1) PAIR_STAGE(loadW_hi(r1Ref(li - 1)), E_r1(1, li, preA), prefetch(p.sp.A[li - 1], 0, preA), G_sq(0, stA, opF, r1Ref(li - 1), none));
    else PAIR_STAGE(loadW_hi(gRef), E_r1(1, li, preA), loss_inputs(), G_g(0, stA, none));
  }
  PAIR_STAGE(NOP, E_g(0), NOP, G_g(1, stB, fwdRef(setFwdB, 0)));
  loadW_hi(fwdRef(setFwdB, 0));   // upper half of the adjoint sweep's first matrix (lands during the loss stage)
  E_g(1);
  lds_barrier();
  TS();

  // ------------------------------------------------------------------ g_x' = J_pe^T Eg (both halves, PE-stage mapping)
  {
#if ISDF_PE_MAP   // a wave = (BM / NW points) x (direction slices): rows 1 KB apart land on 8 banks, so 64 points per wave was 8-way conflicted
    const int pt = (tid % (BM / T::NW)) + (BM / T::NW) * (tid / 64), prt = (tid % 64) / (BM / T::NW);
#else
    const int pt = tid & (BM - 1), prt = tid / BM;
#endif
    const float y0 = xs[pt * 4], y1 = xs[pt * 4 + 1], y2 = xs[pt * 4 + 2];
    const char* row = Xh(pt >> 6) + (pt & 63) * ROWB;
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
        c += (__cosf(xb) * eg(3 + d * nf + f) + __cosf(xb + kHalfPi) * eg(3 + half + d * nf + f)) * fr;
        fr *= 2.f;
      }
      g0 += c * dx; g1 += c * dy; g2 += c * dz;
    }
    float* dst = part + (prt * BM + pt) * 4;
    dst[0] = g0; dst[1] = g1; dst[2] = g2;
  }
  lds_barrier();
  TS();

  // ------------------------------------------------------------------ sdf, loss + adjoints (one thread per point)
  if (tid < BM) {
    const int64_t n = n0 + tid;
    // sdf = (raw + noise) * so   (fc_map.py:104-109)
    float r = p.params[L.offBout];
#pragma unroll
    for (int k = 0; k < T::NW; ++k) r += rawL[k * BM + tid];
    if (p.noise) { if (n < P) r += p.noise[n]; }
    else if (p.noise_std != 0.f) {   // Box-Muller on Philox4x32-10 keyed by (seed, offset, point)
      const uint4 u = philox4x32_10(make_uint4((uint32_t)n, (uint32_t)(n >> 32), (uint32_t)p.noise_off, (uint32_t)(p.noise_off >> 32)),
                                    make_uint2((uint32_t)p.noise_seed, (uint32_t)(p.noise_seed >> 32) ^ 0x5eedu));
      const float u1 = fmaxf(u01(u.x), 1e-7f), u2 = u01(u.y);
      r += p.noise_std * sqrtf(-2.f * __logf(u1)) * __cosf(6.2831853f * u2);
    }
    const float my_sdf = r * so;
    if (p.sdf && n < P) p.sdf[n] = my_sdf;

    float e0 = 0.f, e1 = 0.f, e2 = 0.f;
#pragma unroll
    for (int k = 0; k < NPART; ++k) {
      const float* s = part + (k * BM + tid) * 4;
      e0 += s[0]; e1 += s[1]; e2 += s[2];
    }
    // g_x = scale_input * R^T g_x'
    const float si = L.scale_input;
    const float gx = si * (L.T[0] * e0 + L.T[4] * e1 + L.T[8] * e2);
    const float gy = si * (L.T[1] * e0 + L.T[5] * e1 + L.T[9] * e2);
    const float gz = si * (L.T[2] * e0 + L.T[6] * e1 + L.T[10] * e2);
    if (p.sdf_grad && n < P) { p.sdf_grad[n * 3] = gx; p.sdf_grad[n * 3 + 1] = gy; p.sdf_grad[n * 3 + 2] = gz; }
    float lsum[4] = {0.f, 0.f, 0.f, 0.f};
    float sbar = 0.f, bx = 0.f, by = 0.f, bz = 0.f;
    if (n < P) {
      const isdf_loss_cfg& lc = p.loss;
      const int64_t ray = n / p.S;
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
    // per-tile loss sums: wave w (0 or 1) holds exactly the 64 points of half w
    float v5[5] = {lsum[0], lsum[1], lsum[2], lsum[3], sbar * so};
#pragma unroll
    for (int k = 0; k < 5; ++k) {
      float v = half_wave_sum(v5[k]);
      v += __shfl_xor(v, 32, 64);
      v5[k] = v;
    }
    if ((tid & 63) == 0) {
      const int64_t tile = (int64_t)blockIdx.x * 2 + w;
#pragma unroll
      for (int k = 0; k < 4; ++k) p.wg_loss[tile * 8 + k] = v5[k];
      const int64_t rem = P - (n0 + (int64_t)w * HB);
      p.wg_loss[tile * 8 + 4] = (float)(rem < 0 ? 0 : (rem < HB ? rem : HB));
      vecPair[(int64_t)w * p.vecStride + nL * HD + 2 * HD] = v5[4];   // d b_out = sum sbar*so
    }
  }
  lds_barrier();
  TS();

  // ------------------------------------------------------------------ Ebar = J_pe gbar  -> region 2 (bf16), both halves
  {
#if ISDF_PE_MAP   // a wave = (BM / NW points) x (direction slices): rows 1 KB apart land on 8 banks, so 64 points per wave was 8-way conflicted
    const int pt = (tid % (BM / T::NW)) + (BM / T::NW) * (tid / 64), prt = (tid % 64) / (BM / T::NW);
#else
    const int pt = tid & (BM - 1), prt = tid / BM;
#endif
    const float y0 = xs[pt * 4], y1 = xs[pt * 4 + 1], y2 = xs[pt * 4 + 2];
    const float b0 = gbs[pt * 4], b1 = gbs[pt * 4 + 1], b2 = gbs[pt * 4 + 2];
    char* row = Xh(pt >> 6) + (pt & 63) * ROWB;
    auto put = [&](int feat, float v) { *(__bf16*)(row + swz(pt, (HD + feat) * 2)) = (__bf16)v; };
    if (prt == 0) {
      put(0, b0); put(1, b1); put(2, b2);
      for (int f = L.E; f < HD; ++f) put(f, 0.f);
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
  refresh();
  TS();
  spill_region(HD, p.sp.GB[0]);
  TS();

  // ------------------------------------------------------------------ adjoint of the first reverse sweep (upward)
  prefetch(p.sp.A[1], 0, preA); prefetch(p.sp.P[0], 0, preB);
  G_fwd(0, stA, opB, setFwdB, 0, none);
  TS();
  lds_barrier();
  TS();
#pragma unroll
  for (int li = 0; li < nL - 1; ++li) {
    PAIR_STAGE(NOP, E_adj(0, li, preA, preB), (prefetch(p.sp.A[li + 1], 1, preA), prefetch(p.sp.P[li], 1, preB)),
               G_fwd(1, stB, opB, setFwdB, li, fwdRef(setFwdB, li + 1)));
    PAIR_STAGE(loadW_hi(fwdRef(setFwdB, li + 1)), E_adj(1, li, preA, preB), (prefetch(p.sp.A[li + 2], 0, preA), prefetch(p.sp.P[li + 1], 0, preB)),
               G_fwd(0, stA, opB, setFwdB, li + 1, none));
  }
  // From here on the epilogues are the heaviest of the kernel (three partial-sum streams at the top layer, bias sums
  // below) and a fragment window that is live ACROSS them costs spilled VGPRs.  So the second-half GEMMs stop
  // re-requesting (STAGE 2) and the next unit's slice is requested in one burst after the following epilogue: one exposed
  // L2 round trip per unit for the last six units.
  auto loadW = [&](PRef r) {
#pragma unroll
    for (int k = 0; k < 16; ++k) W[k] = load_wfrag(rsW, r, lane16, k);
  };
  PAIR_STAGE(NOP, E_adj_top(0, preA, preB), (prefetch(p.sp.A[nL], 1, preA), prefetch(p.sp.P[nL - 1], 1, preB)), G_fwd(1, stBlast, opB, setFwdB, nL - 1, none));
  PAIR_STAGE(NOP, E_adj_top(1, preA, preB), (loadW(r2Ref(nL - 2)), (prefetch(p.sp.A[nL - 1], 0, preA), prefetch(p.sp.INJ[nL - 2], 0, preB))), G_sq(0, stA, opB, r2Ref(nL - 2), none));

  // ------------------------------------------------------------------ ordinary reverse sweep with injection
#pragma unroll
  for (int li = nL - 2; li >= 1; --li) {
    PAIR_STAGE(NOP, E_r2(0, li, preA, preB), (prefetch(p.sp.A[li + 1], 1, preA), prefetch(p.sp.INJ[li], 1, preB)), G_sq(1, stBlast, opB, r2Ref(li), none));
    PAIR_STAGE(NOP, E_r2(1, li, preA, preB), (loadW(r2Ref(li - 1)), (prefetch(p.sp.A[li], 0, preA), prefetch(p.sp.INJ[li - 1], 0, preB))), G_sq(0, stA, opB, r2Ref(li - 1), none));
  }
  PAIR_STAGE(NOP, E_r2(0, 0, preA, preB), (prefetch(p.sp.A[1], 1, preA), prefetch(p.sp.INJ[0], 1, preB)), G_sq(1, stBlast, opB, r2Ref(0), none));
  E_r2(1, 0, preA, preB);
  TS();
#undef PAIR_STAGE
#undef NOP
}

// ---------------------------------------------------------------------------
#ifndef ISDF_PAIR_UNROLL_LAYERS
#define ISDF_PAIR_UNROLL_LAYERS ISDF_PAIR_INTERLEAVE   // instantiate <NL, CAT> = <6, 3> / <8, 4> (the shipped 256-wide nets) with unrolled layer loops
#endif
template <int HD, bool F16, int NL, int CAT>
static int launch_pair(const ChainParams& p, int64_t nTiles, hipStream_t st) {
  typedef PairTile<HD> T;
  auto k = chain_pair_kernel<HD, F16, NL, CAT>;
  hipLaunchKernelGGL(k, dim3((unsigned)((nTiles + 1) / 2)), dim3(T::NW * 64), 0, st, p);   // LDS is static (three objects)
  return isdf_launch_status();
}

// train-mode launcher for the nets this kernel covers (hidden 256, padded embedding 256, <= 8 hidden layers);
// the caller (launch_chain) falls back to chain_kernel for everything else.  The spill buffer, vec_part and wg_loss
// must hold an EVEN number of tiles (make_workspace rounds up).
bool pair_supported(const NetLayout& l) { return l.HD == 256 && l.EP == 256 && l.L <= PairTile<256>::MAXLP && l.L >= 3; }
int launch_chain_pair(const ChainParams& p, int64_t nTiles, hipStream_t st) {
  if (nTiles <= 0) return ISDF_OK;
#if ISDF_PAIR_UNROLL_LAYERS
  if (p.lay.L == 6 && p.lay.cat == 3)
    return p.lay.fwd_f16 ? launch_pair<256, true, 6, 3>(p, nTiles, st) : launch_pair<256, false, 6, 3>(p, nTiles, st);
#endif
  return p.lay.fwd_f16 ? launch_pair<256, true, 0, 0>(p, nTiles, st) : launch_pair<256, false, 0, 0>(p, nTiles, st);
}

}  // namespace isdf
