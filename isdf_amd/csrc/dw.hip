// Weight-gradient kernel (K3): for every hidden layer
//   dW_l[o][i] = sum_pts  zbar_l[pt][o] * I_l[pt][i]  +  p_l[pt][o] * Gb_l[pt][i]
// i.e. the two outer-product sums autograd accumulates in total_loss.backward()
// (isdf/modules/trainer.py:981): the ordinary reverse sweep and the reverse of
// the input-gradient graph.  Operands are the bf16 tiles chain.hip spilled.
//
// Mapping: one workgroup = one 256x256 dW "unit" (a layer; the cat layer is two
// units) x one K-split over point tiles.  FOUR waves, one per SIMD, as 2(o) x 2(i),
// each wave a 128x128 fp32 accumulator quadrant (256 of its 512 registers).  The
// contraction index is the POINT, but the spilled tiles are [point][feature], so
// both MFMA operands are fetched with ds_read_b64_tr_b16 (LDS transpose read,
// gfx950) from row-major LDS tiles; rows are padded by 64 B so the four rows of a
// transpose read fall in disjoint bank windows, and their 8-byte chunks are
// XOR-swizzled by the row so that the commit's stores do too.  Two stages of global
// loads are in flight; the next stage's way into LDS rides behind the MFMAs of the
// running one, slice by slice (dw_kernel's header).
#include "isdf_common.h"
#include "chain_params.h"
#include "chain_dev.h"

namespace isdf {

// Read-once / write-once streams are non-temporal: the operand tiles (353 MB) must not evict the packed weight copies the
// NEXT step's chain kernel streams from L2 (measured: that kernel 185.6 -> 182.8 us), and the K-split slabs are written
// once and read once by the step tail (dW 75.9 -> 74.9 us).

template <int HD> struct DwTile {
  static constexpr int BM = DW_PTS;
  static constexpr int NT = 256;                  // threads: four waves, ONE per SIMD, up to 512 registers each (see dw_kernel)
  static constexpr int ROWB = DW_BLK * 2 + 64;    // padded LDS row (bytes) of a 256-column operand slice
  static constexpr int TEN = BM * ROWB;           // one operand slice in LDS
  static constexpr int AUXB = BM * 32;            // one tile's pe_aux rows (8 floats per point)
  static constexpr int DIRTAB = 4 * TEN + 2 * AUXB;      // the 21 directions / 2 pi, 16 bytes each (PE units)
  static constexpr int DUMMY = DIRTAB + 512;             // a spare row: where the stores of threads without work go (PE fill, last round)
  static constexpr int LDS_BYTES = DUMMY + ROWB;          // two operand slices x two stage buffers + two tiles of pe_aux + the directions + the spare row
  static constexpr int CH = (BM * DW_BLK * 2) / (NT * 16);  // uint4 per thread per 16-bit tensor (8); an e4m3 tensor has half
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

// F16: the spilled 16-bit operands are fp16 (NetLayout::bwd_f16) instead of bf16.  SP8: NetLayout::sp8 (bit 0: GB spilled as e4m3,
// bit 1: P below the top layer) -- a launch constant, so every stage's format is known at compile time.
//
// Round 6, second version: FOUR waves, one per SIMD.  The first version (eight waves, two per SIMD, 256 registers each) spent
// ~3.1 us per 64-point stage whatever the bytes -- the top layer's 32 workgroups alone took as long as all 255 together
// (profiles/r06_dw_unit_kinds.txt): every wave ran 4 x [12 transpose reads -> wait -> 8 MFMAs] with no read-ahead across k-steps (its
// 24 operand registers were single-buffered beside a 128-register accumulator), all eight committed the next stage's 64 KB to LDS in
// front of the first MFMA, and the matrix pipe was 39 % busy.  With one wave per SIMD a wave owns a 128 x 128 quadrant (256 accumulator
// registers), reads the operands of the next HALF-step while the current one's 8 MFMAs run (48 registers), feeds every A fragment to four
// MFMAs instead of two (a third fewer LDS reads), and has the registers to hang the next stage's commit, its reload and -- for the
// units that rebuild their embedding-shaped operand -- the PE fill behind the MFMAs of the running stage, a slice per half-step.
// What the PMC counters then showed the first version had really been bound by: its LDS commit (8-way bank conflicts, 70 % of its LDS
// cycles -- see the swizzle below).  76.9 -> 58 us; the kernel now runs at the rate its bytes arrive (330 MB at 5.6 TB/s).
//
// Units whose input-side operand is embedding-shaped (layer 0, and the embedding columns of the cat layer) do not READ it: the
// embedding of the even stage and Ebar = J_pe gbar of the odd stage are functions of six floats per point (x' and gbar in x' space,
// `pe_aux`, written by the chain kernel's loss stage), so the workgroup rebuilds the 64 x 256 slice in LDS (see run()).
template <int HD, bool F16, int SP8>
__global__ __launch_bounds__(256, 1) void dw_kernel(const DwParams p) {
  typedef DwTile<HD> T;
  constexpr int BM = T::BM, ROWB = T::ROWB, CH = T::CH, NT = T::NT;
  static_assert(TILE_PTS == DW_PTS, "pe_aux staging below assumes one dW stage pair per chain tile");
  static_assert(CH == 8 && BM / 16 == 4, "one 16-bit chunk rides behind each of the eight half-steps (an e4m3 chunk behind every other)");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const NetLayout& L = p.lay;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int wo = w >> 1, wi = w & 1;               // this wave's 128 x 128 quadrant of the unit
  // block -> (unit, K-split): units have 35 or 40 splits (isdf_common.h)
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
  // A slice of the padded embedding that lies entirely behind its last feature (the 512-wide net of BASELINE configs[4]: E = 255 in an
  // EP = 512 operand) is all zeros: no parameter sits behind any of its dW columns and the step tail never reads its slab.
  if (fromEmb && slBfull * DW_BLK >= L.E) return;

  const int64_t P = p.n_valid ? (int64_t)(*p.n_valid) * p.S : p.n_points_host;
  const int nTiles = (int)((P + TILE_PTS - 1) / TILE_PTS);

  // stage q of tile t: q=0 -> (ZB[li], I), q=1 -> (P[li], GB).  Stage parity = q = register set = LDS buffer: the even stages'
  // operands are 16-bit tiles (8 x 16 B per thread and tensor), the odd stages' are e4m3 where the format says so (4 x 16 B;
  // SpillLayout), converted to the 16-bit MFMA operand type on their way into LDS -- P times 2^-10, GB times the point's scale s_G
  // (pe_aux[7]).
  const int64_t offZ = p.sp.ZB[li], offP = p.sp.P[li];
  const int64_t offI = fromEmb ? 0 : p.sp.A[li];
  const int64_t offG = fromEmb ? 0 : p.sp.GB[li];
  constexpr int CH8 = CH / 2;

  f32x16 acc[4][4];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

  const int nStages = split < nTiles ? 2 * ((nTiles - split + DW_SPLITK - 1) / DW_SPLITK) : 0;
  // a PE unit of a six-octave net builds its operand in the ALIGNED column order (see run()); others in the reference's order
  const bool peAligned = fromEmb && slBfull == 0 && L.E <= DW_BLK && L.n_freqs == 6;

  // per-lane transpose-read addressing: in each 16-lane group source lane s
  // supplies row (s>>2), 4-element column chunk (s&3); destination lane i gets
  // column i of the 4x16 block (measured on gfx950: tools/probes/probe_layouts.hip)
  const int s16 = lane & 15, cg = (lane >> 4) & 1, hi = lane >> 5;
  const int trRow = 8 * hi + (s16 >> 2);
  const int trColB = (16 * cg + 4 * (s16 & 3)) * 2;
  // LDS layout of an operand slice: row-major [point][256 features], rows padded to ROWB, and the 8-byte chunk index of a row XORed
  // with swz(row) = (row >> 1) & 7.  The padding (16 banks per row) spreads the four rows of a transpose read over the 64 banks; the
  // XOR does the same for the COMMIT, whose ds_write_b64 serves 16 consecutive lanes = 16 consecutive points at one column per LDS
  // cycle group: without it those land on two bank pairs of the stores' 32 (8-way: 32 LDS cycles per store instead of 4, ~4 000 of
  // them per 64-point stage -- what the first version of this kernel was actually bound by).  A transpose read covers whole aligned
  // groups of eight chunks per row, so the XOR only changes which lane reads which chunk of the same bank set.
  const int trX = ((trRow >> 1) & 7) << 3;
  const int trOffA = trRow * ROWB + (trColB ^ trX), trOffB = (trRow + 4) * ROWB + (trColB ^ trX ^ 16);
  // (the operand slice a PE unit REBUILDS is written by its fill, not by the commit: it stays unswizzled -- the fill's 4-byte stores
  // then take immediate offsets instead of an XOR and an add each)
  const int trOffA_in = fromEmb ? trRow * ROWB + trColB : trOffA, trOffB_in = fromEmb ? (trRow + 4) * ROWB + trColB : trOffB;
  // where this thread's pieces go (chunk 0; see put16 / put8 in run()): a frag16 piece c NT + tid is point pt16, features f16.. and
  // f16 + 8..; a frag8 piece is points ln & 31 and 32 + (ln & 31), features f8.. and f8 + 8.. (rows pt and pt + 32 share swz)
  int pt16, f16;
  const int vo16 = frag16_piece(tid, 0, 0, pt16, f16) * 16, vo8 = tid * 16;      // ... and where they come from (byte offsets in a tile slice)
  const int wr16A = pt16 * ROWB + ((f16 * 2) ^ (((pt16 >> 1) & 7) << 3)), wr16B = wr16A ^ 16;
  const int f8 = (w >> 1) * 32 + 16 * (w & 1) + 4 * (lane >> 5);
  const int wr8A = (lane & 31) * ROWB + ((f8 * 2) ^ ((((lane & 31) >> 1) & 7) << 3)), wr8B = wr8A ^ 16;
  typedef bf16x4 __attribute__((address_space(3))) * lds4;
  auto trload = [&](const char* base, int ptBase, int colElem, int offA, int offB) -> bf16x8 {
    const char* a0 = base + ptBase * ROWB + colElem * 2;      // (ptBase: a multiple of 16, colElem of 32: above the swizzled bits)
    bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds4)(a0 + offA));
    bf16x4 hi4 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds4)(a0 + offB));
    bf16x8 r;
    r[0] = lo[0]; r[1] = lo[1]; r[2] = lo[2]; r[3] = lo[3];
    r[4] = hi4[0]; r[5] = hi4[1]; r[6] = hi4[2]; r[7] = hi4[3];
    return r;
  };
  // One stage = four k-steps of 16 points = eight HALF-steps of 8 MFMAs (two of the wave's four 32-row blocks x its four 32-column
  // blocks).  Operand fragments are read one half-step ahead -- the two A fragments of the next half-step, and during the second
  // half of a k-step the four B fragments of the next k-step -- 48 registers instead of the 64 of a whole k-step of read-ahead.
  // `between(ks, h)` is the slice of the NEXT stage's way into the other LDS buffer (and of its registers' reload) that rides
  // behind the MFMAs of half-step (ks, h).
  typedef typename Op<F16>::v8 opv8;
  auto loadA = [&](const char* sb, int ks, int h, bf16x8 (&a)[2]) {
    a[0] = trload(sb, ks * 16, wo * 128 + (2 * h) * 32, trOffA, trOffB); a[1] = trload(sb, ks * 16, wo * 128 + (2 * h + 1) * 32, trOffA, trOffB);
  };
  auto loadB = [&](const char* sb, int ks, bf16x8 (&b)[4]) {
#pragma unroll
    for (int ib = 0; ib < 4; ++ib) b[ib] = trload(sb + T::TEN, ks * 16, wi * 128 + ib * 32, trOffA_in, trOffB_in);
  };
  auto mfmas = [&](auto hTag, const bf16x8 (&a)[2], const bf16x8 (&b)[4]) {
    constexpr int h = decltype(hTag)::value;
#pragma unroll
    for (int o2 = 0; o2 < 2; ++o2)
#pragma unroll
      for (int ib = 0; ib < 4; ++ib)
        acc[2 * h + o2][ib] = Op<F16>::mfma(__builtin_bit_cast(opv8, a[o2]), __builtin_bit_cast(opv8, b[ib]), acc[2 * h + o2][ib]);
  };
  // ONE wave per SIMD: nothing but the wave's own instruction order overlaps its MFMAs with anything.  Left to itself the compiler
  // emits a half-step as [8 MFMAs][everything else] -- each MFMA waits 32 cycles for the pipe behind the previous one, nothing else
  // issues meanwhile, and the rest then runs beside an idle matrix pipe (PMC: issue + issue-wait = twice the MFMA time).  So every
  // half-step is its own scheduling region, laid out as 8 x [1 MFMA | <= 2 LDS reads | <= NW LDS writes | <= NV VALU | <= 1 load].
  auto pipeline = [&](auto nvTag, auto nwTag) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
      __builtin_amdgcn_sched_group_barrier(0x200, decltype(nwTag)::value, 0);
      __builtin_amdgcn_sched_group_barrier(0x002, decltype(nvTag)::value, 0);
      __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
  };
  auto compute = [&](int buf, auto nvTag, auto nwTag, auto&& pre, auto&& between0) {
    auto between = [&](auto ks, auto h) { between0(ks, h); pipeline(nvTag, nwTag); };
    const char* sb = smem + buf * 2 * T::TEN;
    typedef std::integral_constant<int, 0> I0; typedef std::integral_constant<int, 1> I1;
    typedef std::integral_constant<int, 2> I2; typedef std::integral_constant<int, 3> I3;
    bf16x8 a0[2], a1[2], b0[4], b1[4];
    pre();                                      // (PE units: the LDS reads of the first fill round, a half-step ahead of their use)
    loadB(sb, 0, b0); loadA(sb, 0, 0, a0);
    loadA(sb, 0, 1, a1);                        mfmas(I0{}, a0, b0); between(I0{}, I0{});
    loadB(sb, 1, b1); loadA(sb, 1, 0, a0);      mfmas(I1{}, a1, b0); between(I0{}, I1{});
    loadA(sb, 1, 1, a1);                        mfmas(I0{}, a0, b1); between(I1{}, I0{});
    loadB(sb, 2, b0); loadA(sb, 2, 0, a0);      mfmas(I1{}, a1, b1); between(I1{}, I1{});
    loadA(sb, 2, 1, a1);                        mfmas(I0{}, a0, b0); between(I2{}, I0{});
    loadB(sb, 3, b1); loadA(sb, 3, 0, a0);      mfmas(I1{}, a1, b0); between(I2{}, I1{});
    loadA(sb, 3, 1, a1);                        mfmas(I0{}, a0, b1); between(I3{}, I0{});
                                                mfmas(I1{}, a1, b1); between(I3{}, I1{});
  };

  // The stage pipeline, once per kind of unit (PE: the input-side operand is rebuilt from pe_aux, not loaded; p8: this unit's P is e4m3).
  // (NFT: a PE unit's octave count at compile time = the ALIGNED operand order, or 0 = the reference's order with a run-time octave loop.
  // A launch-uniform choice, but made per run() instantiation and not inside the fill: a branch there would cut every half-step's
  // scheduling region in two and put the fill BEHIND the MFMAs instead of between them.)
  // (SLT: for the nets whose embedding spans two 256-column slices -- nine / eleven octaves: realsense*.json -- which slice this unit
  // rebuilds, again per instantiation: slice 0 holds x', every sine and the first cosines, slice 1 the other cosines and the padding.)
  auto run = [&](auto pe_tag, auto p8_tag, auto nft_tag, auto slice_tag) {
    constexpr bool PE = decltype(pe_tag)::value;
    constexpr int NFT = decltype(nft_tag)::value, SLT = decltype(slice_tag)::value;
    constexpr bool ALIGNED = NFT == 6;          // (the aligned column order exists for the six-octave nets only)
    constexpr bool p8 = decltype(p8_tag)::value, g8 = SP8 & 1;     // this unit's P / GB are e4m3 tensors
    constexpr int CHA1 = p8 ? CH8 : CH, CHB1 = g8 ? CH8 : CH;
    // Two register sets so TWO stages of global loads are in flight while one is computed.
    uint4 regA0[CH], regB0[CH], regA1[CHA1], regB1[CHB1];
    float sG1[2] = {1.f, 1.f};        // s_G of this thread's two points (rows lane & 31 and 32 + (lane & 31)) of the odd stage in flight
    // Every global load goes through a buffer descriptor (chain_dev.h): address = the TILE's descriptor (SGPRs, rebuilt per slice by
    // the scalar unit) + ONE per-thread VGPR (vo16 / vo8: the thread's piece of chunk 0) + a scalar offset for slice and chunk.  With
    // flat addresses the compiler keeps a 64-bit VGPR pair per chunk and tensor alive across the loop -- 48 registers this kernel
    // does not have.  A 16-bit tile slice: 8 chunks of 4 KB (frag16_piece: chunk c is 32 features = 256 pieces behind chunk 0), slices
    // 32 KB apart; an e4m3 slice: 4 chunks of 4 KB, slices 16 KB apart.
    const int tileBytes = (int)p.sp.tileStride * 2;
    auto tile_rsrc = [&](int64_t off, int t) { return make_rsrc(p.spill + off + (int64_t)t * p.sp.tileStride, (uint32_t)tileBytes); };
    auto issue0 = [&](int st, auto cTag) {        // even stage: ZB and the layer input, 16-bit
      constexpr int c = decltype(cTag)::value;
      const int t = split + (st >> 1) * DW_SPLITK;
      regA0[c] = bload16<kAuxDwLoad>(tile_rsrc(offZ, t), vo16, slA * 32768 + c * 4096);
      if constexpr (!PE) regB0[c] = bload16<kAuxDwLoad>(tile_rsrc(offI, t), vo16, slB * 32768 + c * 4096);
    };
    auto issue1 = [&](int st, auto ksTag, auto hTag) {       // odd stage, slice (ks, h): P and GB -- e4m3 (frag8 pieces: 16 B = a lane's 8 values of both point blocks; chunk ks, P at h = 0, GB at h = 1) or 16-bit (chunk 2 ks + h)
      constexpr int ks = decltype(ksTag)::value, h = decltype(hTag)::value, c = 2 * ks + h;
      const int t = split + (st >> 1) * DW_SPLITK;
      if constexpr (p8) { if constexpr (h == 0) regA1[ks] = bload16<kAuxDwLoad>(tile_rsrc(offP, t), vo8, slA * 16384 + ks * 4096); }
      else regA1[c] = bload16<kAuxDwLoad>(tile_rsrc(offP, t), vo16, slA * 32768 + c * 4096);
      if constexpr (!PE) {
        if constexpr (g8) {
          if constexpr (h == 1) regB1[ks] = bload16<kAuxDwLoad>(tile_rsrc(offG, t), vo8, slB * 16384 + ks * 4096);
          if constexpr (ks == 3 && h == 1) {       // behind the LAST slice: the commit of the previous odd stage reads the old scales until then
            const rsrc_t rx = make_rsrc(p.pe_aux + (int64_t)t * BM * 8, BM * 32);
            sG1[0] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rx, (tid & 31) * 32 + 28, 0, 0));
            sG1[1] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rx, (tid & 31) * 32 + 28, 32 * 32, 0));
          }
        } else {
          regB1[c] = bload16<kAuxDwLoad>(tile_rsrc(offG, t), vo16, slB * 32768 + c * 4096);
        }
      }
    };
    // two e4m3 values of a dword -> two 16-bit operand values, times `scale`
    auto cvt2 = [](uint32_t w8, float scale, auto hiTag) -> uint32_t {
      if constexpr (F16) return __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_scalef32_pk_f16_fp8((int)w8, scale, decltype(hiTag)::value));
      else return __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8((int)w8, scale, decltype(hiTag)::value));
    };
    // ---- PE units: a thread owns up to six (point, direction) ITEMS of a 64-point tile -- item = tid + 256 r, point = item / 21,
    // direction = item % 21, the same for every stage -- and writes the direction's 2 n_freqs columns: sin / cos of octave 0 from the
    // hardware (in revolutions: the 1 / 2 pi lives in the direction constants), the higher octaves by angle doubling
    // (s' = 2 s c, c' = 1 - 2 s^2: 4 plain VALU operations per octave and pair instead of two projections and two transcendentals;
    // <= 3e-6 from the direct value after 11 doublings, against an fp16 ulp of 5e-4).  Threads 64..127 of round r = 5 (which has
    // only 64 items) write x' / gbar (features 0..2) and the zero padding of their point.
    //
    // Column order of the rebuilt operand.  ALIGNED (the embedding fits one 256-column slice and n_freqs is even: replicaCAD.json /
    // scanNet.json): [sin groups | cos groups | x' | padding] -- direction d's n_freqs sines start at column d n_freqs, a 4-byte
    // boundary, so a group leaves as n_freqs / 2 aligned 4-byte stores (in the reference's order, [x' | sin | cos], every group starts
    // 2 mod 4: twelve 2-byte stores per item and direction).  The dW columns come out in that order too; the slab store below puts
    // column i where the reference's column lives.  Otherwise (EP = 512: nine to eleven octaves) the reference's order, 2-byte stores.
    constexpr float kInv2Pi = 0.15915494309189535f, k2Pi = 6.283185307179586f;
    constexpr int NR = (BM * N_DIRS + BM + NT - 1) / NT;      // rounds (6)
    // (a round's point, direction and direction constants are re-derived where they are used -- two integer operations and one
    // 16-byte LDS read of a 21-entry table -- instead of living in 24 registers across the stage loop)
    float4 auxr = make_float4(0.f, 0.f, 0.f, 0.f);
    if constexpr (PE) {
      if (tid < N_DIRS)
        *(float4*)(smem + T::DIRTAB + tid * 16) = make_float4(kDirs[0][tid] * kInv2Pi, kDirs[1][tid] * kInv2Pi, kDirs[2][tid] * kInv2Pi, 0.f);
    }
    // pe_aux of the k-th tile of this workgroup: 64 points x 2 float4, one float4 per thread of the first two waves
    auto load_aux = [&](int k) {          // (unconditional, as every load of the pipeline: threads 128.. load a row they do not store)
      k = min(k, nStages / 2 - 1);
      const uint4 v = bload16<0>(make_rsrc(p.pe_aux + (int64_t)(split + k * DW_SPLITK) * BM * 8, BM * 32), (tid & 127) * 16, 0);
      auxr = make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
    };
    auto store_aux = [&](int k) {         // (threads 128.. hold and store the same rows again: no branch in the stage loop)
      *(float4*)(smem + 4 * T::TEN + (k & 1) * T::AUXB + (tid & 127) * 16) = auxr;
    };
    auto put16 = [&](char* tile, const auto& r, auto cTag) {      // chunk c of a 16-bit operand slice: a frag16 piece -> row-major LDS tile
      constexpr int c = decltype(cTag)::value;       // chunk c sits 32 features = 64 bytes (above the swizzled bits) behind chunk 0
      *(uint2*)(tile + wr16A + c * 64) = make_uint2(r[c].x, r[c].y);
      *(uint2*)(tile + wr16B + c * 64) = make_uint2(r[c].z, r[c].w);
    };
    auto put8 = [&](char* tile, const auto& r, float s0, float s1, auto cTag) {   // chunk c of an e4m3 slice, converted; s0 / s1: the scales of the lane's two points
      constexpr int c = decltype(cTag)::value;       // chunk c: 64 features = 128 bytes behind chunk 0
      const uint4 a = r[c];
      *(uint2*)(tile + wr8A + c * 128) = make_uint2(cvt2(a.x, s0, std::false_type{}), cvt2(a.x, s0, std::true_type{}));
      *(uint2*)(tile + wr8B + c * 128) = make_uint2(cvt2(a.y, s0, std::false_type{}), cvt2(a.y, s0, std::true_type{}));
      *(uint2*)(tile + wr8A + 32 * ROWB + c * 128) = make_uint2(cvt2(a.z, s1, std::false_type{}), cvt2(a.z, s1, std::true_type{}));
      *(uint2*)(tile + wr8B + 32 * ROWB + c * 128) = make_uint2(cvt2(a.w, s1, std::false_type{}), cvt2(a.w, s1, std::true_type{}));
    };
    // round r of the PE fill of stage st (parity ODD) into its LDS buffer
    // The three LDS reads of a round (the point's x' and gbar rows of pe_aux, the direction) are issued ONE ROUND AHEAD of the VALU work
    // that consumes them: with one wave per SIMD a wait for an LDS read stalls everything behind it, the MFMAs included.
    float4 fy = make_float4(0.f, 0.f, 0.f, 0.f), fg = fy, fdr = fy;
    // (point, direction) of the round in flight: item = tid + 256 r, 256 = 12 x 21 + 4 -- each round is twelve points and four
    // directions past the previous one; round 0 starts from the thread's constants behind an opaque copy (or the compiler hoists the
    // six rounds' points / directions / addresses out of the stage loop and spills them).  Last round: threads 64.. own a point's x' row.
    int fpt = 0, fd = 0;
    const int pt0 = tid / N_DIRS, d0 = tid - pt0 * N_DIRS;
    static_assert(NT == 12 * N_DIRS + 4, "incremental item update");
    auto fill_fetch = [&](int st, auto oddTag, auto rTag) {
      constexpr int r = decltype(rTag)::value;
      if constexpr (r == 0) { fpt = pt0; fd = d0; asm volatile("" : "+v"(fpt), "+v"(fd)); }
      else {
        fd += 4; fpt += 12;
        if (fd >= N_DIRS) { fd -= N_DIRS; fpt += 1; }
        if constexpr (r == NR - 1) { if (tid >= BM * N_DIRS - NT * (NR - 1)) { fpt = (tid - (BM * N_DIRS - NT * (NR - 1))) & (BM - 1); fd = 0; } }
      }
      const char* ax = smem + 4 * T::TEN + ((st >> 1) & 1) * T::AUXB + fpt * 32;
      fy = *(const float4*)ax;
      if constexpr (decltype(oddTag)::value) fg = *(const float4*)(ax + 16);
      fdr = *(const float4*)(smem + T::DIRTAB + fd * 16);
    };
    auto fill_round = [&](int st, auto oddTag, auto rTag) {
      constexpr bool Q1 = decltype(oddTag)::value;
      constexpr int r = decltype(rTag)::value;
      const float4 y = fy, g = fg, dr = fdr;                      // this round's rows, fetched a round ago ...
      const int pt = fpt, d = fd;
      if constexpr (r + 1 < NR) fill_fetch(st, oddTag, std::integral_constant<int, r + 1>{});      // ... and the next round's on their way
      // q = 0: the embedding (embedding.py:95-111): [x' | sin(xb_df) | cos(xb_df)], xb_df = (x' . dir_d) 2^f
      // q = 1: Ebar = J_pe gbar (chain.hip's Ebar stage): [gbar | cos(xb_df) k_df | -sin(xb_df) k_df], k_df = (gbar . dir_d) 2^f
      char* tb = smem + (Q1 ? 1 : 0) * 2 * T::TEN + T::TEN;
      const int nf = L.n_freqs, halfE = N_DIRS * nf, colBase = slBfull * DW_BLK;
      typedef typename Op<F16>::e eT;
      typedef eT e2 __attribute__((ext_vector_type(2)));
      auto st1 = [&](char* row, int col, float v) {            // one column; `col` relative to this unit's 256-column slice
        if ((unsigned)col < (unsigned)DW_BLK) *(eT*)(row + col * 2) = (eT)v;
      };
      constexpr bool lastRound = r == NR - 1;
      char* row = tb + pt * ROWB;
      constexpr int lastItems = BM * N_DIRS - NT * (NR - 1);      // direction items of the last round (64)
      const float r0 = y.x * dr.x + y.y * dr.y + y.z * dr.z;
      float sn = __builtin_amdgcn_sinf(r0), cs = __builtin_amdgcn_cosf(r0), kf = 0.f;
      if constexpr (Q1) kf = (g.x * dr.x + g.y * dr.y + g.z * dr.z) * k2Pi;
      auto vals = [&](float& a, float& b) {        // this octave's two values, then on to the next octave
        a = Q1 ? cs * kf : sn; b = Q1 ? -sn * kf : cs;
        const float u = sn + sn;                   // (2 s) c and 1 - (2 s) s: the same values as 2 (s c) and fma(-2 s, s, 1), one operation less
        const float c2 = __builtin_fmaf(-u, sn, 1.f);
        sn = u * cs; cs = c2; kf += kf;
      };
      if constexpr (ALIGNED) {
        // STRAIGHT-LINE code: in the last round, which has 64 direction items and 64 x' rows for 256 threads, every thread runs both
        // store sequences and the ones without work aim at a spare LDS row (a select, not a branch)
        static_assert(NFT % 2 == 0 && 2 * N_DIRS * NFT + 4 == DW_BLK, "paired stores; x' and one zero fill the slice behind the 42 NFT sine / cosine columns");
        char* dummy = smem + T::DUMMY;
        char* irow = !lastRound || tid < lastItems ? row : dummy;
        const int cs0 = d * NFT * 2, cc0 = cs0 + N_DIRS * NFT * 2;    // 4-byte aligned: column d * NFT of row pt
#pragma unroll
        for (int f = 0; f < NFT; f += 2) {
          float a, b, a2, b2;
          vals(a, b); vals(a2, b2);
          e2 vs, vc; vs[0] = (eT)a; vs[1] = (eT)a2; vc[0] = (eT)b; vc[1] = (eT)b2;
          *(e2*)(irow + cs0 + f * 2) = vs; *(e2*)(irow + cc0 + f * 2) = vc;
        }
        if constexpr (lastRound) {      // aligned order: x' behind the sine / cosine columns
          char* xrow = tid >= lastItems && tid < lastItems + BM ? row : dummy;
          const float4 v = Q1 ? g : y;
          e2 v01, v2z; v01[0] = (eT)v.x; v01[1] = (eT)v.y; v2z[0] = (eT)v.z; v2z[1] = (eT)0.f;
          *(e2*)(xrow + 2 * N_DIRS * NFT * 2) = v01; *(e2*)(xrow + 2 * N_DIRS * NFT * 2 + 4) = v2z;
        }
      } else if constexpr (NFT != 0) {
        // The reference's column order [x' | sin | cos | padding] at a compile-time octave count, STRAIGHT-LINE as well: a store whose
        // column falls outside this unit's slice -- or whose thread has no item in the last round -- aims at the spare row (a select per
        // store instead of a branch per store: the branchy loop below runs ~10 us per 64-point stage on its own).  2-byte stores: with an
        // odd octave count the groups start at either parity.  Every sine lives in slice 0 (3 + 21 NFT <= 256 up to twelve octaves).
        static_assert(3 + N_DIRS * NFT <= DW_BLK && 3 + 2 * N_DIRS * NFT > DW_BLK && 3 + 2 * N_DIRS * NFT <= 2 * DW_BLK, "two slices; the sines in the first");
        char* dummy = smem + T::DUMMY;
        char* irow = !lastRound || tid < lastItems ? row : dummy;
        const int cS = 3 + d * NFT, cC = cS + N_DIRS * NFT - SLT * DW_BLK;      // first sine / cosine column of the direction, relative to this slice
#pragma unroll
        for (int f = 0; f < NFT; ++f) {
          float a, b;
          vals(a, b);
          if constexpr (SLT == 0) *(eT*)(irow + (cS + f) * 2) = (eT)a;
          const int cc = cC + f;
          char* pc = (unsigned)cc < (unsigned)DW_BLK ? irow : dummy;
          *(eT*)(pc + (cc & (DW_BLK - 1)) * 2) = (eT)b;
        }
        if constexpr (lastRound && SLT == 0) {      // x' in front of the sines; the padding behind the cosines (slice 1) is zeroed once, ahead of the stage loop
          char* xrow = tid >= lastItems && tid < lastItems + BM ? row : dummy;
          const float4 v = Q1 ? g : y;
          *(eT*)(xrow) = (eT)v.x; *(eT*)(xrow + 2) = (eT)v.y; *(eT*)(xrow + 4) = (eT)v.z;
        }
      } else if (!lastRound || tid < lastItems) {
        const int cS = 3 + d * nf - colBase, cC = cS + halfE;      // first column of the sine / cosine group in this slice
        for (int f = 0; f < nf; ++f) { float a, b; vals(a, b); st1(row, cS + f, a); st1(row, cC + f, b); }
      } else if (tid < lastItems + BM) {
        const float4 v = Q1 ? g : y;
        st1(row, 0 - colBase, v.x); st1(row, 1 - colBase, v.y); st1(row, 2 - colBase, v.z);
        for (int f = L.E; f < L.EP; ++f) st1(row, f - colBase, 0.f);
      }
    };
    // slice (ks, h) of stage st's way into its LDS buffer: chunk 2 ks + h of every 16-bit tensor, chunk ks of an e4m3 tensor (P at
    // h = 0, GB at h = 1) and, for a PE unit, its share of the six fill rounds
    auto commit = [&](int st, auto oddTag, auto ksTag, auto hTag) {
      constexpr bool ODD = decltype(oddTag)::value;
      constexpr int ks = decltype(ksTag)::value, h = decltype(hTag)::value;
      typedef std::integral_constant<int, 2 * ks + h> C; typedef std::integral_constant<int, ks> C8;
      char* sb = smem + (ODD ? 1 : 0) * 2 * T::TEN;
      if constexpr (!ODD) {
        put16(sb, regA0, C{});
        if constexpr (!PE) put16(sb + T::TEN, regB0, C{});
      } else {
        if constexpr (p8) { if constexpr (h == 0) put8(sb, regA1, kSpillPScale, kSpillPScale, C8{}); } else put16(sb, regA1, C{});
        if constexpr (!PE) {
          if constexpr (g8) { if constexpr (h == 1) put8(sb + T::TEN, regB1, sG1[0], sG1[1], C8{}); } else put16(sb + T::TEN, regB1, C{});
        }
      }
      if constexpr (PE) {
        static_assert(NR == 6, "fill rounds 0, 1 | 2, 3 | 4 | 5 ride behind k-steps 0..3");
        constexpr int r = ks < 2 ? 2 * ks + h : (h == 0 ? ks + 2 : -1);
        if constexpr (r >= 0) fill_round(st, oddTag, std::integral_constant<int, r>{});
      }
    };
    // the reload of the registers slice (ks, h) has just committed, for the stage two ahead.  EVERY load of the pipeline is
    // unconditional (past the last stage the workgroup re-reads its last tile, an L2 hit): a load under a branch makes the compiler's
    // s_waitcnt insertion assume the path without it, and every commit would wait for the loads issued a k-step ago instead of two
    // stages ago.
    auto issue = [&](int st, auto oddTag, auto ksTag, auto hTag) {
      constexpr int ks = decltype(ksTag)::value, h = decltype(hTag)::value;
      constexpr bool ODD = decltype(oddTag)::value;
      st = min(st, nStages - (ODD ? 1 : 2));
      if constexpr (ODD) issue1(st, ksTag, hTag);
      else issue0(st, std::integral_constant<int, 2 * ks + h>{});
    };
    auto each_slice = [&](auto&& fn) {
      typedef std::integral_constant<int, 0> I0; typedef std::integral_constant<int, 1> I1;
      fn(I0{}, I0{}); fn(I0{}, I1{}); fn(I1{}, I0{}); fn(I1{}, I1{});
      fn(std::integral_constant<int, 2>{}, I0{}); fn(std::integral_constant<int, 2>{}, I1{});
      fn(std::integral_constant<int, 3>{}, I0{}); fn(std::integral_constant<int, 3>{}, I1{});
    };

    if constexpr (PE) {
      load_aux(0); store_aux(0); load_aux(1);
      if constexpr (NFT != 0 && !ALIGNED && SLT == 1) {      // the zero padding behind the last cosine column: the same in every stage, written once into both buffers
        constexpr int c0 = 3 + 2 * N_DIRS * NFT - DW_BLK, npad = DW_BLK - c0;
        typedef typename Op<F16>::e eT0;
        for (int i = tid; i < BM * npad; i += NT) {
          const int r = i / npad, c = c0 + i - r * npad;
          *(eT0*)(smem + T::TEN + r * ROWB + c * 2) = (eT0)0.f;
          *(eT0*)(smem + 3 * T::TEN + r * ROWB + c * 2) = (eT0)0.f;
        }
      }
    }
    each_slice([&](auto k, auto h) { issue(0, std::false_type{}, k, h); });
    each_slice([&](auto k, auto h) { issue(1, std::true_type{}, k, h); });
    if constexpr (PE) __syncthreads();          // tile 0's pe_aux rows are in LDS
    if constexpr (PE) fill_fetch(0, std::false_type{}, std::integral_constant<int, 0>{});
    each_slice([&](auto k, auto h) { commit(0, std::false_type{}, k, h); issue(2, std::false_type{}, k, h); });
    __syncthreads();

    // Stage st sits in LDS buffer (st & 1).  While its MFMAs run, stage st + 1 (loads issued two stages ago) goes into the OTHER
    // buffer slice by slice -- one slice behind the MFMAs of each half-step -- and each committed register is re-issued for stage
    // st + 3 on the spot; one barrier per stage.  (PE units: the pe_aux rows of the NEXT tile go to LDS behind the last half-step of
    // the even stages, a barrier ahead of the fill that reads them.)  Behind the last stage the commit writes a buffer nobody reads.
    typedef std::integral_constant<int, PE ? 12 : 4> NV;      // VALU / LDS-store instructions per MFMA slot (the PE fill is ~90 VALU a round)
    typedef std::integral_constant<int, PE ? 2 : 1> NW;
    for (int st = 0; st < nStages; st += 2) {       // (nStages is even: two stages per tile)
      compute(0, NV{}, NW{}, [&] { if constexpr (PE) fill_fetch(st + 1, std::true_type{}, std::integral_constant<int, 0>{}); },
              [&](auto ks, auto h) {                            // stage st (even) computes; stage st + 1 (odd) goes into buffer 1
        commit(st + 1, std::true_type{}, ks, h);
        issue(st + 3, std::true_type{}, ks, h);
        if constexpr (PE && decltype(ks)::value == 3 && decltype(h)::value == 1) { store_aux((st >> 1) + 1); load_aux((st >> 1) + 2); }
      });
      __syncthreads();
      compute(1, NV{}, NW{}, [&] { if constexpr (PE) fill_fetch(st + 2, std::false_type{}, std::integral_constant<int, 0>{}); },
              [&](auto ks, auto h) {                            // stage st + 1 computes; stage st + 2 (even) goes into buffer 0
        commit(st + 2, std::false_type{}, ks, h);
        issue(st + 4, std::false_type{}, ks, h);
      });
      __syncthreads();
    }
  };
  typedef std::integral_constant<bool, (SP8 & 2) != 0> P8T;
  typedef std::integral_constant<int, 0> I0_; typedef std::integral_constant<int, 1> I1_;
  // the reference-order fill at a compile-time octave count: the nine / eleven-octave nets (16-bit spills: SP8 = 0 instantiations only)
  const bool straightRef = SP8 == 0 && fromEmb && L.EP == 2 * DW_BLK && (L.n_freqs == 9 || L.n_freqs == 11);
  if (nStages == 0) {}      // (more K-splits than tiles: a zero slab)
  else if (fromEmb && peAligned) run(std::true_type{}, P8T{}, std::integral_constant<int, 6>{}, I0_{});
  else if (straightRef) {
    if constexpr (SP8 == 0) {
      if (L.n_freqs == 9) { if (slBfull == 0) run(std::true_type{}, P8T{}, std::integral_constant<int, 9>{}, I0_{}); else run(std::true_type{}, P8T{}, std::integral_constant<int, 9>{}, I1_{}); }
      else { if (slBfull == 0) run(std::true_type{}, P8T{}, std::integral_constant<int, 11>{}, I0_{}); else run(std::true_type{}, P8T{}, std::integral_constant<int, 11>{}, I1_{}); }
    }
  }
  else if (fromEmb) run(std::true_type{}, P8T{}, std::integral_constant<int, 0>{}, I0_{});
  else if ((SP8 & 2) && li == L.L - 1) run(std::false_type{}, std::false_type{}, std::integral_constant<int, 0>{}, I0_{});      // the top layer's P stays 16-bit (SpillLayout)
  else run(std::false_type{}, P8T{}, std::integral_constant<int, 0>{}, I0_{});

  // partial slab [o][i]
  slab_t* slab = (slab_t*)p.dwPart + ((int64_t)dw_slab_base(L, unit) + split) * DW_BLK * DW_BLK;
#pragma unroll
  for (int ob = 0; ob < 4; ++ob)
#pragma unroll
    for (int ib = 0; ib < 4; ++ib)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int o = wo * 128 + ob * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
        int i = wi * 128 + ib * 32 + (lane & 31);
        if (peAligned) {     // aligned operand order [sin | cos | x' | pad] -> the reference's [x' | sin | cos | pad]
          const int nsc = 2 * N_DIRS * L.n_freqs;
          i = i < nsc ? i + 3 : (i < nsc + 3 ? i - nsc : i);
        }
        slab[o * DW_BLK + i] = acc[ob][ib][r];      // default cache policy: the step tail reads the slabs next (non-temporal stores: tail +1.8 us)
      }
}

// The K-split slabs and the per-workgroup bias / out-layer partials are summed by the step-tail kernel (optim.hip).

int launch_dw(const DwParams& p, hipStream_t st) {
  if (!layout_supported(p.lay)) return ISDF_EUNSUPPORTED;
  typedef DwTile<256> T;
  const dim3 grid(dw_total_slabs(p.lay)), block(T::NT);
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
