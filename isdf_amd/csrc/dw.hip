// Weight-gradient kernel (K3): for every hidden layer
//   dW_l[o][i] = sum_pts  zbar_l[pt][o] * I_l[pt][i]  +  p_l[pt][o] * Gb_l[pt][i]
// i.e. the two outer-product sums autograd accumulates in total_loss.backward()
// (isdf/modules/trainer.py:981): the ordinary reverse sweep and the reverse of
// the input-gradient graph.  Operands are the bf16 tiles chain.hip spilled.
//
// Mapping: one workgroup = one 256x256 dW "unit" (a layer; the cat layer is two
// units) x one K-split over point tiles.  8 waves as 4(o) x 2(i), each wave a
// 64x128 fp32 accumulator (128 VGPRs).  The contraction index is the POINT, but
// the spilled tiles are [point][feature], so both MFMA operands are fetched
// with ds_read_b64_tr_b16 (LDS transpose read, gfx950) from row-major LDS
// tiles; rows are padded by 64 B so the 4-row x 32-B footprints of the two
// 16-lane groups of a half-wave fall in disjoint bank windows.  Global loads of
// the next stage are issued into registers before the MFMAs of the current one
// and written to LDS after the barrier (issue-early / write-late).
#include "isdf_common.h"
#include "chain_params.h"

namespace isdf {

// Read-once / write-once streams are non-temporal: the operand tiles (353 MB) must not evict the packed weight copies the
// NEXT step's chain kernel streams from L2 (measured: that kernel 185.6 -> 182.8 us), and the K-split slabs are written
// once and read once by the step tail (dW 75.9 -> 74.9 us).

template <int HD> struct DwTile {
  static constexpr int BM = DW_PTS;
  static constexpr int ROWB = DW_BLK * 2 + 64;    // padded LDS row (bytes) of a 256-column operand slice
  static constexpr int TEN = BM * ROWB;           // one operand slice in LDS
  static constexpr int AUXB = BM * 32;            // one tile's pe_aux rows (8 floats per point)
  static constexpr int LDS_BYTES = 4 * TEN + 2 * AUXB;   // two operand slices x two stage buffers + two tiles of pe_aux
  static constexpr int CH = (BM * DW_BLK * 2) / (512 * 16);  // uint4 per thread per tensor
};

// Piece c (16 B = 8 elems) of the 64-point half `half`, 256-feature slice `sl` of a chain tile
// stored in frag16 order (chain.hip).  A slice is 8 consecutive 32-feature blocks; block g of the
// tile is (wave g / FB, fb g % FB), so the tile piece index is ((g*PB + pb)*2 + qp)*64 + lane.
// Returns that index; (pt, f0) = point in the half and first feature in the slice:
// elems 0..3 at features f0.., elems 4..7 at f0+8..
__device__ __forceinline__ int frag16_piece(int c, int half, int sl, int& pt, int& f0) {
  constexpr int PB = TILE_PTS / 32, HB = DW_PTS / 32;
  const int lane = c & 63; int r = c >> 6;
  const int qp = r & 1; r >>= 1;
  const int pbh = r % HB; const int blk = r / HB;   // blk 0..7
  pt = pbh * 32 + (lane & 31);
  f0 = blk * 32 + 16 * qp + 4 * (lane >> 5);
  return ((((sl * 8 + blk) * PB + half * HB + pbh) * 2 + qp) * 64 + lane);
}

// F16: the spilled operands are fp16 (NetLayout::bwd_f16) instead of bf16 -- the same bits through the same transpose reads,
// v_mfma_f32_32x32x16_f16 instead of _bf16.
//
// Units whose input-side operand is embedding-shaped (layer 0, and the embedding columns of the cat layer) do not READ it
// (round 6): both operands of such a unit's input side -- the embedding of stage q = 0 and Ebar = J_pe gbar of stage q = 1 -- are
// functions of six floats per point (x' and gbar in x' space, `pe_aux`, written by the chain kernel's loss stage), so the
// workgroup rebuilds the 64 x 256 slice in LDS: a thread owns two adjacent columns (its direction, octave and phase are
// fixed for the whole kernel) and walks 16 points.  The arithmetic is the chain kernel's PE / Ebar stage value for value
// (same expression shapes, `fr` a power of two), so the operand bits equal what that kernel used to spill.  4 of the 28 tensor
// reads of a step and 2 of the chain kernel's 25 tensor stores are gone (82 MB of 1.245 GB per 27 k-point step).
// SP8: NetLayout::sp8 (bit 0: GB spilled as e4m3, bit 1: P below the top layer) -- a launch constant, so the format of every stage is
// known at compile time and the register sets are sized for it (the kernel sits at the 256-register limit).
template <int HD, bool F16, int SP8>
__global__ __launch_bounds__(512, 2) void dw_kernel(const DwParams p) {
  typedef DwTile<HD> T;
  constexpr int BM = T::BM, ROWB = T::ROWB, CH = T::CH;
  static_assert(TILE_PTS == DW_PTS, "pe_aux staging below assumes one dW stage pair per chain tile");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const NetLayout& L = p.lay;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int wo = w >> 1, wi = w & 1;
  // block -> (unit, K-split): units have 32 or 48 splits (isdf_common.h)
  int unit = 0, split = blockIdx.x;
  for (int n; split >= (n = dw_unit_splits(L, unit)); ++unit) split -= n;
  const int DW_SPLITK = dw_unit_splits(L, unit);
  const DwUnit du = dw_unit(L, unit);
  const int li = du.li;
  // input-side operand: columns [256*ib, +256) of the padded layer input.  Layer 0 reads the
  // embedding; the cat layer reads [a | emb]; the rest read the previous activation.
  const bool fromEmb = dw_unit_from_emb(L, du);
  const int slBfull = (li == L.cat && fromEmb) ? du.ib - HD / DW_BLK : du.ib;   // 256-column slice of the operand
  const int slB = slBfull % (HD / DW_BLK);
  const int slA = du.ob;

  const int64_t P = p.n_valid ? (int64_t)(*p.n_valid) * p.S : p.n_points_host;
  constexpr int HALVES = TILE_PTS / DW_PTS;
  const int nTiles = (int)((P + TILE_PTS - 1) / TILE_PTS) * HALVES;   // 64-point half tiles

  // stage q of tile t: q=0 -> (ZB[li], I), q=1 -> (P[li], GB).  Stage parity = q = register set = LDS buffer: the even stages'
  // operands are 16-bit tiles (4 x 16 B per thread and tensor), the odd stages' are e4m3 (2 x 16 B; SpillLayout), converted to the
  // 16-bit MFMA operand type on their way into LDS -- P times 2^-10, GB times the point's scale s_G (pe_aux[7]).
  const int64_t offZ = p.sp.ZB[li], offP = p.sp.P[li];
  const int64_t offI = fromEmb ? 0 : p.sp.A[li];
  const int64_t offG = fromEmb ? 0 : p.sp.GB[li];
  constexpr int CH8 = CH / 2;

  f32x16 acc[2][4];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

  const int nStages = split < nTiles ? 2 * ((nTiles - split + DW_SPLITK - 1) / DW_SPLITK) : 0;
  // a PE unit of a six-octave net builds its operand in the ALIGNED column order (see run()); others in the reference's order
  const bool peAligned = fromEmb && L.EP == DW_BLK && L.n_freqs == 6;

  // per-lane transpose-read addressing: in each 16-lane group source lane s
  // supplies row (s>>2), 4-element column chunk (s&3); destination lane i gets
  // column i of the 4x16 block (measured on gfx950: tools/probes/probe_layouts.hip)
  const int s16 = lane & 15, cg = (lane >> 4) & 1, hi = lane >> 5;
  const int trRow = 8 * hi + (s16 >> 2);
  const int trColB = (16 * cg + 4 * (s16 & 3)) * 2;
  typedef bf16x4 __attribute__((address_space(3))) * lds4;
  auto trload = [&](const char* base, int ptBase, int colElem) -> bf16x8 {
    const char* a0 = base + (ptBase + trRow) * ROWB + colElem * 2 + trColB;
    bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds4)(a0));
    bf16x4 hi4 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds4)(a0 + 4 * ROWB));
    bf16x8 r;
    r[0] = lo[0]; r[1] = lo[1]; r[2] = lo[2]; r[3] = lo[3];
    r[4] = hi4[0]; r[5] = hi4[1]; r[6] = hi4[2]; r[7] = hi4[3];
    return r;
  };
  auto compute = [&](int buf) {
    const char* sb = smem + buf * 2 * T::TEN;
#pragma unroll
    for (int ks = 0; ks < BM / 16; ++ks) {
      bf16x8 a[2], b[4];
#pragma unroll
      for (int ob = 0; ob < 2; ++ob) a[ob] = trload(sb, ks * 16, wo * 64 + ob * 32);
#pragma unroll
      for (int ib = 0; ib < 4; ++ib) b[ib] = trload(sb + T::TEN, ks * 16, wi * 128 + ib * 32);
#pragma unroll
      for (int ob = 0; ob < 2; ++ob)
#pragma unroll
        for (int ib = 0; ib < 4; ++ib)
          acc[ob][ib] = Op<F16>::mfma(__builtin_bit_cast(typename Op<F16>::v8, a[ob]), __builtin_bit_cast(typename Op<F16>::v8, b[ib]),
                                      acc[ob][ib]);
    }
  };

  // The stage pipeline, once per kind of unit (PE: the input-side operand is rebuilt from pe_aux, not loaded).
  auto run = [&](auto pe_tag, auto p8_tag) {
    constexpr bool PE = decltype(pe_tag)::value;
    constexpr bool p8 = decltype(p8_tag)::value, g8 = SP8 & 1;     // this unit's P / GB are e4m3 tensors
    constexpr int CHA1 = p8 ? CH8 : CH, CHB1 = g8 ? CH8 : CH;
    // Two register sets so TWO stages of global loads are in flight while one is computed
    // (HBM-bound kernel: 64 KB per stage and CU; one stage in flight left ~40 % of the bandwidth unused).
    uint4 regA0[CH], regB0[CH], regA1[CHA1], regB1[CHB1];
    float sG1[2] = {1.f, 1.f};        // s_G of this thread's two points (rows lane & 31 and 32 + (lane & 31)) of the odd stage in flight
    typedef unsigned int u32x4n __attribute__((ext_vector_type(4)));
    auto ntload = [](const uint4* q) {
      const u32x4n v = __builtin_nontemporal_load((const u32x4n*)q);
      return make_uint4(v[0], v[1], v[2], v[3]);
    };
    auto issue0 = [&](int st) {        // even stage: ZB and the layer input, 16-bit
      const int t = split + (st >> 1) * DW_SPLITK;
      const uint4* ta = (const uint4*)(p.spill + offZ + (int64_t)t * p.sp.tileStride);
      const uint4* tb = (const uint4*)(p.spill + offI + (int64_t)t * p.sp.tileStride);
#pragma unroll
      for (int c = 0; c < CH; ++c) {
        int pt, f0;
        regA0[c] = ntload(ta + frag16_piece(c * 512 + tid, 0, slA, pt, f0));
        if constexpr (!PE) regB0[c] = ntload(tb + frag16_piece(c * 512 + tid, 0, slB, pt, f0));
      }
    };
    auto issue1 = [&](int st) {        // odd stage: P and GB -- e4m3 (frag8 pieces: 16 B = a lane's 8 values of both point blocks) or 16-bit
      const int t = split + (st >> 1) * DW_SPLITK;
      const uint4* ta = (const uint4*)(p.spill + offP + (int64_t)t * p.sp.tileStride);
      const uint4* tb = (const uint4*)(p.spill + offG + (int64_t)t * p.sp.tileStride);
      if constexpr (p8) {
#pragma unroll
        for (int c = 0; c < CH8; ++c) regA1[c] = ntload(ta + (slA * 16 + ((c * 512 + tid) >> 6)) * 64 + (tid & 63));
      } else {
#pragma unroll
        for (int c = 0; c < CH; ++c) { int pt, f0; regA1[c] = ntload(ta + frag16_piece(c * 512 + tid, 0, slA, pt, f0)); }
      }
      if constexpr (!PE) {
        if constexpr (g8) {
#pragma unroll
          for (int c = 0; c < CH8; ++c) regB1[c] = ntload(tb + (slB * 16 + ((c * 512 + tid) >> 6)) * 64 + (tid & 63));
          const float* aux = p.pe_aux + ((int64_t)t * BM + (tid & 31)) * 8 + 7;
          sG1[0] = aux[0]; sG1[1] = aux[32 * 8];
        } else {
#pragma unroll
          for (int c = 0; c < CH; ++c) { int pt, f0; regB1[c] = ntload(tb + frag16_piece(c * 512 + tid, 0, slB, pt, f0)); }
        }
      }
    };
    // two e4m3 values of a dword -> two 16-bit operand values, times `scale`
    auto cvt2 = [](uint32_t w8, float scale, auto hiTag) -> uint32_t {
      if constexpr (F16) return __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_scalef32_pk_f16_fp8((int)w8, scale, decltype(hiTag)::value));
      else return __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8((int)w8, scale, decltype(hiTag)::value));
    };
    // ---- PE units: a thread owns up to three (point, direction) ITEMS of a 64-point tile -- item = tid + 512 r, point = item / 21,
    // direction = item % 21, the same for every stage -- and writes the direction's 2 n_freqs columns: sin / cos of octave 0 from the
    // hardware (in revolutions: the 1 / 2 pi lives in the direction constants), the higher octaves by angle doubling
    // (s' = 2 s c, c' = 1 - 2 s^2: 4 plain VALU operations per octave and pair instead of two projections and two transcendentals;
    // <= 3e-6 from the direct value after 11 doublings, against an fp16 ulp of 5e-4).  Threads 320..383 of round r = 2 (which has
    // only 320 items) write x' / gbar (features 0..2) and the zero padding of their point.
    //
    // Column order of the rebuilt operand.  ALIGNED (the embedding fits one 256-column slice and n_freqs is even: replicaCAD.json /
    // scanNet.json): [sin groups | cos groups | x' | padding] -- direction d's n_freqs sines start at column d n_freqs, a 4-byte
    // boundary, so a group leaves as n_freqs / 2 aligned 4-byte stores (in the reference's order, [x' | sin | cos], every group starts
    // 2 mod 4: twelve 2-byte stores per item and direction).  The dW columns come out in that order too; the slab store below puts
    // column i where the reference's column lives.  Otherwise (EP = 512: nine to eleven octaves) the reference's order, 2-byte stores.
    constexpr float kInv2Pi = 0.15915494309189535f, k2Pi = 6.283185307179586f;
    float drx[3] = {0.f, 0.f, 0.f}, dry[3] = {0.f, 0.f, 0.f}, drz[3] = {0.f, 0.f, 0.f};
    int itemOff[3] = {0, 0, 0};       // aligned order: (point * 32) << 16 | byte offset of the item's first column in the operand tile;
                                      // reference order: point << 16 | (first column relative to the unit's slice) + 1024
    float4 auxr = make_float4(0.f, 0.f, 0.f, 0.f);
    if constexpr (PE) {
      const int nf = L.n_freqs;
#pragma unroll
      for (int r = 0; r < 3; ++r) {
        const int item = tid + 512 * r;
        const bool dirItem = item < BM * N_DIRS;
        const int pt = dirItem ? item / N_DIRS : (item - BM * N_DIRS) & (BM - 1), d = dirItem ? item - pt * N_DIRS : 0;
        drx[r] = kDirs[0][d] * kInv2Pi; dry[r] = kDirs[1][d] * kInv2Pi; drz[r] = kDirs[2][d] * kInv2Pi;
        itemOff[r] = peAligned ? ((pt * 32) << 16) | (pt * ROWB + (dirItem ? d * nf * 2 : 0))
                               : (pt << 16) | ((dirItem ? 3 + d * nf - slBfull * DW_BLK : 0) + 1024);
      }
    }
    // pe_aux of the k-th tile of this workgroup: 64 points x 2 float4, one float4 per thread of the first two waves
    auto load_aux = [&](int k) {
      if (tid < 128 && 2 * k < nStages)
        auxr = ((const float4*)p.pe_aux)[((int64_t)(split + k * DW_SPLITK)) * (BM * 2) + tid];
    };
    auto store_aux = [&](int k) {
      if (tid < 128) *(float4*)(smem + 4 * T::TEN + (k & 1) * T::AUXB + tid * 16) = auxr;
    };
    auto put16 = [&](char* tile, const auto& r) {      // a 16-bit operand slice: frag16 pieces -> row-major LDS tile
#pragma unroll
      for (int c = 0; c < CH; ++c) {
        int pt, f0;
        frag16_piece(c * 512 + tid, 0, 0, pt, f0);
        char* pa = tile + pt * ROWB + f0 * 2;
        *(uint2*)(pa) = make_uint2(r[c].x, r[c].y);
        *(uint2*)(pa + 16) = make_uint2(r[c].z, r[c].w);
      }
    };
    auto put8 = [&](char* tile, const auto& r, float s0, float s1) {   // an e4m3 slice, converted; s0 / s1: the scales of the lane's two points
#pragma unroll
      for (int c = 0; c < CH8; ++c) {
        const int idx = c * 512 + tid, ln = idx & 63, rr = idx >> 6;
        const int f0 = (rr >> 1) * 32 + 16 * (rr & 1) + 4 * (ln >> 5);       // elems 0..3 at features f0.., 4..7 at f0 + 8..
        char* pa = tile + (ln & 31) * ROWB + f0 * 2;
        const uint4 a = r[c];
        *(uint2*)(pa) = make_uint2(cvt2(a.x, s0, std::false_type{}), cvt2(a.x, s0, std::true_type{}));
        *(uint2*)(pa + 16) = make_uint2(cvt2(a.y, s0, std::false_type{}), cvt2(a.y, s0, std::true_type{}));
        *(uint2*)(pa + 32 * ROWB) = make_uint2(cvt2(a.z, s1, std::false_type{}), cvt2(a.z, s1, std::true_type{}));
        *(uint2*)(pa + 32 * ROWB + 16) = make_uint2(cvt2(a.w, s1, std::false_type{}), cvt2(a.w, s1, std::true_type{}));
      }
    };
    auto commit = [&](int st, int buf) {
      char* sb = smem + buf * 2 * T::TEN;
      if ((st & 1) == 0) {
        put16(sb, regA0);
        if constexpr (!PE) put16(sb + T::TEN, regB0);
      } else {
        if constexpr (p8) put8(sb, regA1, kSpillPScale, kSpillPScale); else put16(sb, regA1);
        if constexpr (!PE) { if constexpr (g8) put8(sb + T::TEN, regB1, sG1[0], sG1[1]); else put16(sb + T::TEN, regB1); }
      }
      if constexpr (PE) {
        // q = 0: the embedding (embedding.py:95-111): [x' | sin(xb_df) | cos(xb_df)], xb_df = (x' . dir_d) 2^f
        // q = 1: Ebar = J_pe gbar (chain.hip's Ebar stage): [gbar | cos(xb_df) k_df | -sin(xb_df) k_df], k_df = (gbar . dir_d) 2^f
        const char* auxl = smem + 4 * T::TEN + ((st >> 1) & 1) * T::AUXB;
        char* tb = sb + T::TEN;
        const int nf = L.n_freqs, halfE = N_DIRS * nf, colBase = slBfull * DW_BLK;
        typedef typename Op<F16>::e eT;
        typedef eT e2 __attribute__((ext_vector_type(2)));
        auto st1 = [&](char* row, int col, float v) {            // one column; `col` relative to this unit's 256-column slice
          if ((unsigned)col < (unsigned)DW_BLK) *(eT*)(row + col * 2) = (eT)v;
        };
        auto fill = [&](auto nfc, auto q1c) {
          constexpr int NFT = decltype(nfc)::value;               // n_freqs at compile time: the ALIGNED column order (0: run-time loop, 2-byte stores)
          constexpr bool Q1 = decltype(q1c)::value;
#pragma unroll
          for (int r = 0; r < 3; ++r) {
            // (opaque to the optimiser: derived per-item addresses hoisted out of the stage loop would sit in registers next to the
            // 128 of the accumulator for the whole kernel -- one packed word per round does)
            int io = itemOff[r];
            asm volatile("" : "+v"(io));
            const char* ax = auxl + (NFT ? (io >> 16) : (io >> 16) * 32);
            if (r < 2 || tid < BM * N_DIRS - 1024) {
              const float4 y = *(const float4*)ax;
              const float r0 = y.x * drx[r] + y.y * dry[r] + y.z * drz[r];
              float sn = __builtin_amdgcn_sinf(r0), cs = __builtin_amdgcn_cosf(r0), kf = 0.f;
              if constexpr (Q1) {
                const float4 g = *(const float4*)(ax + 16);
                kf = (g.x * drx[r] + g.y * dry[r] + g.z * drz[r]) * k2Pi;
              }
              auto vals = [&](float& a, float& b) {        // this octave's two values, then on to the next octave
                a = Q1 ? cs * kf : sn; b = Q1 ? -sn * kf : cs;
                const float t = sn * cs;
                cs = __builtin_fmaf(-2.f * sn, sn, 1.f); sn = t + t; kf += kf;
              };
              if constexpr (NFT == 0) {
                char* row = tb + (io >> 16) * ROWB;
                const int cS = (io & 0xffff) - 1024, cC = cS + halfE;      // first column of the sine / cosine group in this slice
                for (int f = 0; f < nf; ++f) { float a, b; vals(a, b); st1(row, cS + f, a); st1(row, cC + f, b); }
              } else {
                static_assert(NFT % 2 == 0, "paired stores assume an even octave count");
                char* ps = tb + (io & 0xffff);                       // 4-byte aligned: column d * NFT of row pt
                char* pc = ps + N_DIRS * NFT * 2;
#pragma unroll
                for (int f = 0; f < NFT; f += 2) {
                  float a, b, a2, b2;
                  vals(a, b); vals(a2, b2);
                  e2 vs, vc; vs[0] = (eT)a; vs[1] = (eT)a2; vc[0] = (eT)b; vc[1] = (eT)b2;
                  *(e2*)(ps + f * 2) = vs; *(e2*)(pc + f * 2) = vc;
                }
              }
            } else if (tid < BM * N_DIRS - 1024 + BM) {
              char* row = tb + (NFT ? (io >> 16) >> 5 : (io >> 16)) * ROWB;
              const float4 v = *(const float4*)(ax + (Q1 ? 16 : 0));
              if constexpr (NFT != 0) {      // aligned order: x' behind the 42 n_freqs sine / cosine columns, then the padding
                e2 v01, v2z; v01[0] = (eT)v.x; v01[1] = (eT)v.y; v2z[0] = (eT)v.z; v2z[1] = (eT)0.f;
                *(e2*)(row + 2 * N_DIRS * NFT * 2) = v01; *(e2*)(row + 2 * N_DIRS * NFT * 2 + 4) = v2z;
                for (int f = 2 * N_DIRS * NFT + 4; f < DW_BLK; f += 2) { e2 z; z[0] = (eT)0.f; z[1] = (eT)0.f; *(e2*)(row + f * 2) = z; }
              } else {
                st1(row, 0 - colBase, v.x); st1(row, 1 - colBase, v.y); st1(row, 2 - colBase, v.z);
                for (int f = L.E; f < L.EP; ++f) st1(row, f - colBase, 0.f);
              }
            }
          }
        };
        const bool q1 = st & 1;
        if (peAligned && nf == 6) { if (q1) fill(std::integral_constant<int, 6>{}, std::true_type{}); else fill(std::integral_constant<int, 6>{}, std::false_type{}); }
        else { if (q1) fill(std::integral_constant<int, 0>{}, std::true_type{}); else fill(std::integral_constant<int, 0>{}, std::false_type{}); }
      }
    };

    if constexpr (PE) {
      load_aux(0); store_aux(0); load_aux(1);
    }
    if (nStages > 0) { issue0(0); }
    if (nStages > 1) { issue1(1); }
    if constexpr (PE) __syncthreads();          // tile 0's pe_aux rows are in LDS
    if (nStages > 0) commit(0, 0);
    if (nStages > 2) issue0(2);
    __syncthreads();

    // Stage st sits in LDS buffer (st&1).  Each iteration first writes stage st+1 (loads issued two
    // stages ago) into the OTHER LDS buffer, re-issues that register set for stage st+3, then runs the
    // MFMAs of stage st: the LDS commit of one wave overlaps the MFMAs of the others, one barrier per stage.
    // (PE units: the pe_aux rows of the NEXT tile go to LDS in the first half, a barrier ahead of the commit that reads them.)
    for (int st = 0; st < nStages; st += 2) {
      if (st + 1 < nStages) {
        commit(st + 1, 1);
        if constexpr (PE) { store_aux((st >> 1) + 1); load_aux((st >> 1) + 2); }
        if (st + 3 < nStages) issue1(st + 3);
      }
      compute(0);
      __syncthreads();
      if (st + 1 < nStages) {
        if (st + 2 < nStages) { commit(st + 2, 0); if (st + 4 < nStages) issue0(st + 4); }
        compute(1);
        __syncthreads();
      }
    }
  };
  typedef std::integral_constant<bool, (SP8 & 2) != 0> P8T;
  if (fromEmb) run(std::true_type{}, P8T{});
  else if ((SP8 & 2) && li == L.L - 1) run(std::false_type{}, std::false_type{});      // the top layer's P stays 16-bit (SpillLayout)
  else run(std::false_type{}, P8T{});

  // partial slab [o][i]
  slab_t* slab = (slab_t*)p.dwPart + ((int64_t)dw_slab_base(L, unit) + split) * DW_BLK * DW_BLK;
#pragma unroll
  for (int ob = 0; ob < 2; ++ob)
#pragma unroll
    for (int ib = 0; ib < 4; ++ib)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int o = wo * 64 + ob * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
        int i = wi * 128 + ib * 32 + (lane & 31);
        if (peAligned) {     // aligned operand order [sin | cos | x' | pad] -> the reference's [x' | sin | cos | pad]
          const int nsc = 2 * N_DIRS * L.n_freqs;
          i = i < nsc ? i + 3 : (i < nsc + 3 ? i - nsc : i);
        }
        __builtin_nontemporal_store(acc[ob][ib][r], slab + o * DW_BLK + i);
      }
}

// The K-split slabs and the per-workgroup bias / out-layer partials are summed by the step-tail kernel (optim.hip).

int launch_dw(const DwParams& p, hipStream_t st) {
  if (!layout_supported(p.lay)) return ISDF_EUNSUPPORTED;
  typedef DwTile<256> T;
  const dim3 grid(dw_total_slabs(p.lay)), block(512);
  auto go = [&](auto k) {
    if (hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, T::LDS_BYTES) != hipSuccess) return (int)ISDF_EHIP;
    hipLaunchKernelGGL(k, grid, block, T::LDS_BYTES, st, p);
    return isdf_launch_status();
  };
  if (p.lay.sp8 && !p.lay.bwd_f16) return ISDF_EUNSUPPORTED;
  if (p.lay.HD == 256) {
    if (p.lay.sp8 == 3) return go(dw_kernel<256, true, 3>);
    if (p.lay.sp8 == 1) return go(dw_kernel<256, true, 1>);
    return p.lay.bwd_f16 ? go(dw_kernel<256, true, 0>) : go(dw_kernel<256, false, 0>);
  }
  if (p.lay.sp8) return ISDF_EUNSUPPORTED;      // (make_layout: e4m3 spills with the 256-wide tiles only)
  return p.lay.bwd_f16 ? go(dw_kernel<512, true, 0>) : go(dw_kernel<512, false, 0>);
}

}  // namespace isdf
