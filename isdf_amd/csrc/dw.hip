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
  static constexpr int LDS_BYTES = 4 * TEN;       // two operand slices x two stage buffers
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
template <int HD, bool F16>
__global__ __launch_bounds__(512, 2) void dw_kernel(const DwParams p) {
  typedef DwTile<HD> T;
  constexpr int BM = T::BM, ROWB = T::ROWB, CH = T::CH;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const NetLayout& L = p.lay;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int wo = w >> 1, wi = w & 1;
  const int unit = blockIdx.x / DW_SPLITK, split = blockIdx.x % DW_SPLITK;
  const DwUnit du = dw_unit(L, unit);
  const int li = du.li;
  // input-side operand: columns [256*ib, +256) of the padded layer input.  Layer 0 reads the
  // embedding; the cat layer reads [a | emb]; the rest read the previous activation.
  const bool fromEmb = li == 0 || (li == L.cat && du.ib * DW_BLK >= HD);
  const int slBfull = (li == L.cat && fromEmb) ? du.ib - HD / DW_BLK : du.ib;   // 256-column slice of the operand
  // an embedding-shaped operand is stored as EP/HD consecutive HD-wide tensors of HD/256 slices each
  const int slT = slBfull / (HD / DW_BLK), slB = slBfull % (HD / DW_BLK);
  const int slA = du.ob;

  const int64_t P = p.n_valid ? (int64_t)(*p.n_valid) * p.S : p.n_points_host;
  constexpr int HALVES = TILE_PTS / DW_PTS;
  const int nTiles = (int)((P + TILE_PTS - 1) / TILE_PTS) * HALVES;   // 64-point half tiles

  // stage q of tile t: q=0 -> (ZB[li], I), q=1 -> (P[li], GB)
  const int64_t offZ = p.sp.ZB[li], offP = p.sp.P[li];
  const int64_t offI = fromEmb ? p.sp.A[0] + slT * p.sp.tensorElems : p.sp.A[li];
  const int64_t offG = fromEmb ? p.sp.GB[0] + slT * p.sp.tensorElems : p.sp.GB[li];

  f32x16 acc[2][4];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

  const int nStages = split < nTiles ? 2 * ((nTiles - split + DW_SPLITK - 1) / DW_SPLITK) : 0;
  // Two register sets so TWO stages of global loads are in flight while one is computed
  // (HBM-bound kernel: 64 KB per stage and CU; one stage in flight left ~40 % of the bandwidth unused).
  uint4 regA0[CH], regB0[CH], regA1[CH], regB1[CH];
  auto issue = [&](int st, uint4 (&ra)[CH], uint4 (&rb)[CH]) {
    const int u = split + (st >> 1) * DW_SPLITK;      // half-tile index
    const int t = u / HALVES, half = u % HALVES;
    const uint4* ta = (const uint4*)(p.spill + ((st & 1) ? offP : offZ) + (int64_t)t * p.sp.tileStride);
    const uint4* tb = (const uint4*)(p.spill + ((st & 1) ? offG : offI) + (int64_t)t * p.sp.tileStride);
#pragma unroll
    for (int c = 0; c < CH; ++c) {
      int pt, f0;
      typedef unsigned int u32x4n __attribute__((ext_vector_type(4)));
      const u32x4n va = __builtin_nontemporal_load((const u32x4n*)(ta + frag16_piece(c * 512 + tid, half, slA, pt, f0)));
      const u32x4n vb = __builtin_nontemporal_load((const u32x4n*)(tb + frag16_piece(c * 512 + tid, half, slB, pt, f0)));
      ra[c] = make_uint4(va[0], va[1], va[2], va[3]); rb[c] = make_uint4(vb[0], vb[1], vb[2], vb[3]);
    }
  };
  auto commit = [&](const uint4 (&ra)[CH], const uint4 (&rb)[CH], int buf) {
    char* sb = smem + buf * 2 * T::TEN;
#pragma unroll
    for (int c = 0; c < CH; ++c) {
      int pt, f0;
      frag16_piece(c * 512 + tid, 0, 0, pt, f0);
      char* pa = sb + pt * ROWB + f0 * 2;
      *(uint2*)(pa) = make_uint2(ra[c].x, ra[c].y);
      *(uint2*)(pa + 16) = make_uint2(ra[c].z, ra[c].w);
      char* pb = sb + T::TEN + pt * ROWB + f0 * 2;
      *(uint2*)(pb) = make_uint2(rb[c].x, rb[c].y);
      *(uint2*)(pb + 16) = make_uint2(rb[c].z, rb[c].w);
    }
  };

  if (nStages > 0) { issue(0, regA0, regB0); }
  if (nStages > 1) { issue(1, regA1, regB1); }
  if (nStages > 0) commit(regA0, regB0, 0);
  if (nStages > 2) issue(2, regA0, regB0);
  __syncthreads();

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

  // Stage st sits in LDS buffer (st&1).  Each iteration first writes stage st+1 (loads issued two
  // stages ago) into the OTHER LDS buffer, re-issues that register set for stage st+3, then runs the
  // MFMAs of stage st: the LDS commit of one wave overlaps the MFMAs of the others, one barrier per stage.
  for (int st = 0; st < nStages; st += 2) {
    if (st + 1 < nStages) { commit(regA1, regB1, 1); if (st + 3 < nStages) issue(st + 3, regA1, regB1); }
    compute(0);
    __syncthreads();
    if (st + 1 < nStages) {
      if (st + 2 < nStages) { commit(regA0, regB0, 0); if (st + 4 < nStages) issue(st + 4, regA0, regB0); }
      compute(1);
      __syncthreads();
    }
  }

  // partial slab [o][i]
  slab_t* slab = (slab_t*)p.dwPart + ((int64_t)unit * DW_SPLITK + split) * DW_BLK * DW_BLK;
#pragma unroll
  for (int ob = 0; ob < 2; ++ob)
#pragma unroll
    for (int ib = 0; ib < 4; ++ib)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int o = wo * 64 + ob * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
        const int i = wi * 128 + ib * 32 + (lane & 31);
        __builtin_nontemporal_store(acc[ob][ib][r], slab + o * DW_BLK + i);
      }
}

// The K-split slabs and the per-workgroup bias / out-layer partials are summed by the step-tail kernel (optim.hip).

int launch_dw(const DwParams& p, hipStream_t st) {
  if (!layout_supported(p.lay)) return ISDF_EUNSUPPORTED;
  typedef DwTile<256> T;
  const dim3 grid(dw_units(p.lay) * DW_SPLITK), block(512);
  auto go = [&](auto k) {
    if (hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, T::LDS_BYTES) != hipSuccess) return (int)ISDF_EHIP;
    hipLaunchKernelGGL(k, grid, block, T::LDS_BYTES, st, p);
    return isdf_launch_status();
  };
  if (p.lay.HD == 256) return p.lay.bwd_f16 ? go(dw_kernel<256, true>) : go(dw_kernel<256, false>);
  return p.lay.bwd_f16 ? go(dw_kernel<512, true>) : go(dw_kernel<512, false>);
}

}  // namespace isdf
