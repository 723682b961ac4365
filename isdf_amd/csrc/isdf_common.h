// Internal layout definitions shared by the host entry points and the kernels.
// (Nothing here is part of the C ABI -- that is include/isdf_hip.h.)
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/isdf_hip.h"

namespace isdf {

constexpr int MAXL = 16;       // max hidden layers (2B+2)
constexpr int N_DIRS = 21;     // icosahedron directions, embedding.py:40-62
constexpr int TILE_PTS = 64;   // points per chain-kernel workgroup (two workgroups per CU)
constexpr int DW_PTS = 64;     // points per dW-kernel stage
constexpr int CHAIN_NW = 8;    // waves per chain-kernel workgroup (each owns HD/CHAIN_NW features)
constexpr int CHAIN_CHUNK_FRAGS = 8;   // weight fragments a wave requests at once (32 VGPRs at the 128-VGPR budget)
// K-splits per dW unit: 5 x 32 + 2 x 48 = 256 workgroups for the default net, one per CU.  With the whole chip at work the kernel
// runs at the rate its bytes arrive (~5.6 TB/s of operand reads and slab writes), so the split only has to keep every CU busy to the
// end: the units that REBUILD their embedding-shaped operand (dw.hip) do VALU work on top of their MFMAs (55 us on their own at 48
// splits against 50 us for the others at 32) and get the finer split.  Same-box A/Bs of 30/53 .. 36/38, twice: all within 1 us of each
// other now (profiles/r06_dw_v2.txt; the first version of the kernel, bound by its LDS commit instead, liked 35/40:
// profiles/r06_dw_splits.txt).
constexpr int DW_SPLIT_REG = 32, DW_SPLIT_PE = 48, DW_SPLIT_MAX = 48;
typedef float slab_t;          // K-split partial slabs (bf16 slabs measured: parity unchanged, -2 us only)

// Vector types for the 16-bit MFMA operands.
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(4))) _Float16 f16x4;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(2))) float f32x2;

// Parameter / packed-weight / workspace layout, computed on the host by
// make_layout() and passed to kernels by value.
struct NetLayout {
  int HD, EP, E, B, L, cat, n_freqs;   // HD: tile width of the hidden layers (256 or 512), EP: padded embedding width
  int H;                       // hidden_feature_size itself (<= HD): the fp32 parameters have the reference's shapes [H x K]; units
                               // H..HD-1 of every layer are padding -- zero weights and biases in the packed copies, so their
                               // activation softplus(0) feeds nothing and every adjoint entering them is exactly zero
  int fwd_f16;                 // 1: fp16 operands in forward/first-backward GEMMs
  int fwd_x2;                  // 1: compensated forward ("fp16x2"): layers >= cat add W_lo * x (and, past the cat layer,
                               //    W * x_lo) so that sdf meets the reference to 1e-3
  int fwd_x2_all;             // 1: "fp16x2_full" -- EVERY forward layer is compensated (weights and inputs, embedding included): the
                               //    exact-forward instrument, sdf ~1e-6 of the fp32 reference; <256, 256> nets only
  int bwd_f16;                // 1: the second-order sweeps and every spilled dW operand in fp16 instead of bf16 (needs fwd_f16)
  int sp8;                    // e4m3 spills (isdf_net_cfg.spill_operand; SpillLayout): bit 0 = GB, bit 1 = P below the top layer
  int has_transform;
  float scale_input, scale_output;
  float T[12];
  int64_t n_params;
  int32_t offW[MAXL], offB[MAXL], K[MAXL];  // fp32 flat offsets; K = fan-in (unpadded)
  int32_t offWout, offBout;
  // packed 16-bit matrices: offsets (elements) inside one "fwd set" / "bwd set"
  int64_t fwdMat[MAXL];        // [HD x Kpad(li)]  = W_li          (A operand, M = out feature)
  int64_t bwdMat[MAXL];        // [HD x HD]        = W_li^T[:HD]   (li >= 1)
  int64_t bwdG;                // [EP x 2HD]       = [W_in^T | W_cat[:, HD:]^T]
  int64_t fwdSetElems, bwdSetElems;
  // the four sets inside the shadow buffer (element offsets)
  int64_t setFwdA, setFwdB, setBwdA, setBwdB;  // A: fwd_operand type, B: bf16
  int64_t setFwdLo;            // fp16 residuals W - fp16(W) of the forward matrices of layers >= cat (fwd_x2; all layers with fwd_x2_all)
  int64_t shadowElems;
};

// Spill tensors written by the chain kernel and read by the dW kernel.  A[] and ZB[]: nTiles * TILE_PTS * HD 16-bit elements
// each in "frag16" order (see chain.hip): bf16, or fp16 with NetLayout::bwd_f16.  P[] (below the top layer) and GB[] likewise, or --
// with NetLayout::sp8, round 6 -- ONE BYTE per element, OCP e4m3: P / 2^-10, and GB / s_G[point] with s_G = 2^k * (the power of two above the point's |gbar|_inf in x' space),
// k = spill_gb_shift(): GB is linear in the point's loss adjoint gbar, so the per-point scale takes the loss's magnitude out of the
// format and what is left is bounded by the network (measured: |GB| / 2^ceil(log2 |gbar|) <= 153 at six octaves, 1 230 at eleven).
// A tile's e4m3 tensor is TILE_PTS * HD bytes in "frag8" order: piece ((w * FB + fb) * 2 + qp) * 64 + lane = 16 bytes = the lane's
// 8 values of point block 0, then of point block 1.  The second-order product P^T GB carries the eikonal / normal terms' share of the
// gradient (weights 0.268 / 0.018 upstream): e4m3's 2^-4 on it moves the worst weight gradient from 1.29e-3 to 1.46e-3 of the
// reference's (tools/studies/spill_format_study.py), whereas A or ZB in e4m3 would cost 3.4e-3 / 3.9e-3.  All offsets below are in
// 16-bit units (an e4m3 tensor takes tensorElems / 2 of them).
constexpr float kSpillPScale = 1.f / 1024.f;      // P is stored as e4m3(P / kSpillPScale): |P| <= 0.2 measured (trained nets), e4m3 holds 448
struct SpillLayout {
  int64_t tensorElems;   // per tensor per tile (TILE_PTS * HD); the buffer is [tile][tensor][tensorElems]
  int64_t tileStride;    // elements between consecutive tiles (= tensor count * tensorElems)
  int64_t A[MAXL + 1];   // A[li+1] = activation after layer li.  A[0] (the embedding) is NOT stored: -1 (round 6; see GB)
  int64_t P[MAXL];       // d sdf / d z_li
  int64_t GB[MAXL];      // GB[li] = adjoint entering layer li (li >= 1).  GB[0] (Ebar = J_pe gbar) is NOT stored: -1.  The two
                         // embedding-shaped dW operands are functions of 6 floats per point (x' and gbar, WorkspaceLayout::offPeAux): the dW
                         // kernel rebuilds them in LDS instead of reading 2 x 512 B per point back (twice: layer 0 and the cat layer)
  int64_t ZB[MAXL];      // d loss / d z_li
  int64_t totalElems;
};

struct WorkspaceLayout {
  int64_t nTiles;
  SpillLayout sp;
  int64_t offSpill;      // bytes
  int64_t offWgLoss;     // float [nTiles][8]
  int64_t offDwPart;     // float [dw_total_slabs][256*256]: unit u's dw_unit_splits(u) K-split slabs start at slab dw_slab_base(u)
  int64_t offVecPart;    // float [nTiles][vecStride]: per-workgroup bias / out-layer gradient partials
  int64_t offTotLoss;    // float [maxPts] per-point total loss
  int64_t offPeAux;      // float [nTiles*TILE_PTS][8]: x' (3), 0, gbar in x' space (3), 0 per point -- chain -> dW
  int32_t vecStride;
  int64_t totalBytes;
};

// ---- launch status ------------------------------------------------------------
// hipGetLastError() is sticky per host thread: an error left behind by ANOTHER library's HIP call (torch probes
// produce benign ones) would otherwise be reported by our next launcher.  Entry points clear it first
// (isdf_clear_stale_hip_error), launchers report their own status (isdf_launch_status) and remember the HIP
// code for isdf_error_string(ISDF_EHIP).
#if defined(__HIPCC__)
extern thread_local int g_isdf_last_hip_error;
inline void isdf_clear_stale_hip_error() { (void)hipGetLastError(); }
inline int isdf_launch_status() {
  const hipError_t e = hipGetLastError();
  if (e == hipSuccess) return ISDF_OK;
  g_isdf_last_hip_error = (int)e;
  return ISDF_EHIP;
}
#endif

inline int round_up(int x, int m) { return (x + m - 1) / m * m; }

// k of the GB spill scale s_G = 2^k * pow2ceil(|gbar'|_inf): the positional encoding amplifies the adjoint by up to 2^(n_freqs - 1).
// Measured |GB| / pow2ceil(|gbar'|_inf): <= 40 (random init) .. 153 (trained) at six octaves, 1 230 at eleven: k puts that at ~300 of
// e4m3's 448 -- the large entries carry the contraction, and every binade of unused headroom pushes the small ones into e4m3's
// subnormals (the first version ran three binades lower and perturbed the gradients 2.4 x more than the format has to).
__host__ __device__ inline int spill_gb_shift(int n_freqs) { return n_freqs > 6 ? n_freqs - 7 : -1; }

inline int make_layout(const isdf_net_cfg* c, NetLayout* l) {
  if (!c || !l) return ISDF_EINVAL;
  if (c->blocks < 1 || 2 * c->blocks + 2 > MAXL || c->n_freqs < 1 || c->hidden < 1) return ISDF_EINVAL;
  l->H = c->hidden; l->HD = c->hidden <= 256 ? 256 : 512; l->B = c->blocks; l->n_freqs = c->n_freqs;
  if (c->hidden > 512) return ISDF_EUNSUPPORTED;
  l->E = 2 * N_DIRS * c->n_freqs + 3;
  // padded embedding width: a multiple of the 256-wide dW unit and at least the hidden width (the embedding
  // shares region 2 of the activation tile with hidden-width operands)
  l->EP = round_up(l->E, 256) > l->HD ? round_up(l->E, 256) : l->HD;
  l->L = 2 * c->blocks + 2; l->cat = c->blocks + 1;
  if (c->fwd_operand < 0 || c->fwd_operand > 3) return ISDF_EINVAL;
  l->fwd_f16 = c->fwd_operand ? 1 : 0;
  l->fwd_x2 = c->fwd_operand >= 2 ? 1 : 0;
  l->fwd_x2_all = c->fwd_operand == 3 ? 1 : 0;
  if (c->bwd_operand < 0 || c->bwd_operand > 1 || (c->bwd_operand == 1 && !l->fwd_f16)) return ISDF_EINVAL;
  l->bwd_f16 = c->bwd_operand;
  if (c->spill_operand < 0 || c->spill_operand > 3 || (c->spill_operand >= 2 && !l->bwd_f16)) return ISDF_EINVAL;
  if (c->spill_operand >= 2 && l->HD != 256) return ISDF_EUNSUPPORTED;     // the e4m3 formats are instantiated for the 256-wide tiles
  // auto: e4m3 where it is (nearly) free -- up to six octaves the second-order product P^T GB is a small share of every layer's
  // gradient; the positional encoding amplifies it by 2^(n_freqs - 1), and from nine octaves on e4m3's 2^-4 shows
  l->sp8 = c->spill_operand == 2 ? 3 : c->spill_operand == 3 ? 1 : (c->spill_operand == 0 && l->bwd_f16 && c->n_freqs <= 6 && l->HD == 256) ? 3 : 0;
  l->has_transform = c->has_transform;
  l->scale_input = c->scale_input; l->scale_output = c->scale_output;
  for (int i = 0; i < 12; ++i) l->T[i] = c->has_transform ? c->bounds_T[i] : (i % 5 == 0 ? 1.f : 0.f);
  int64_t off = 0;
  for (int li = 0; li < l->L; ++li) {
    int K = li == 0 ? l->E : (li == l->cat ? l->H + l->E : l->H);
    l->K[li] = K;
    l->offW[li] = (int32_t)off; off += (int64_t)l->H * K;
    l->offB[li] = (int32_t)off; off += l->H;
  }
  l->offWout = (int32_t)off; off += l->H;
  l->offBout = (int32_t)off; off += 1;
  l->n_params = off;
  int64_t f = 0, b = 0;
  for (int li = 0; li < l->L; ++li) {
    int Kp = li == 0 ? l->EP : (li == l->cat ? l->HD + l->EP : l->HD);
    l->fwdMat[li] = f; f += (int64_t)l->HD * Kp;
    l->bwdMat[li] = b; if (li >= 1) b += (int64_t)l->HD * l->HD;
  }
  l->bwdG = b; b += (int64_t)l->EP * 2 * l->HD;
  l->fwdSetElems = f; l->bwdSetElems = b;
  l->setFwdA = 0; l->setFwdB = f; l->setBwdA = 2 * f; l->setBwdB = 2 * f + b;
  l->setFwdLo = 2 * f + 2 * b;
  l->shadowElems = 2 * f + 2 * b + (l->fwd_x2 ? f : 0);
  return ISDF_OK;
}

// The tile kernels are instantiated for <HD, EP> = <256, 256> (replicaCAD.json / scanNet.json: hidden 256,
// n_freqs 6 -> E 255), <256, 512> (realsense*.json: hidden 256, n_freqs 9 / 11 -> E 381 / 465) and <512, 512>
// (BASELINE configs[4]); any hidden_feature_size up to 512 runs on them zero-padded (NetLayout::H).  Other shapes
// (hidden > 512, hidden <= 256 with more than 12 embedding octaves): ISDF_EUNSUPPORTED.
inline bool layout_supported(const NetLayout& l) {
  // "fp16x2_full" keeps four operand regions (a, emb and their residuals) in the LDS tile: 128 KB at <256, 256>, too much beyond
  if (l.fwd_x2_all) return l.HD == 256 && l.EP == 256;
  return (l.HD == 256 && (l.EP == 256 || l.EP == 512)) || (l.HD == 512 && l.EP == 512);
}

// dW is computed in 256x256 "units": (layer li, output block ob, padded-input block ib).
constexpr int DW_BLK = 256;
struct DwUnit { int li, ob, ib; };
__host__ __device__ inline int dw_kpad(const NetLayout& l, int li) {
  return li == 0 ? l.EP : (li == l.cat ? l.HD + l.EP : l.HD);
}
__host__ __device__ inline int dw_units(const NetLayout& l) {
  int n = 0;
  for (int li = 0; li < l.L; ++li) n += (l.HD / DW_BLK) * (dw_kpad(l, li) / DW_BLK);
  return n;
}
__host__ __device__ inline DwUnit dw_unit(const NetLayout& l, int u) {
  const int nOb = l.HD / DW_BLK;
  for (int li = 0; li < l.L; ++li) {
    const int n = nOb * (dw_kpad(l, li) / DW_BLK);
    if (u < n) return DwUnit{li, u % nOb, u / nOb};
    u -= n;
  }
  return DwUnit{0, 0, 0};
}

__host__ __device__ inline bool dw_unit_from_emb(const NetLayout& l, const DwUnit& du) {
  return du.li == 0 || (du.li == l.cat && du.ib * DW_BLK >= l.HD);
}
__host__ __device__ inline int dw_unit_splits(const NetLayout& l, int u) {
  return dw_unit_from_emb(l, dw_unit(l, u)) ? DW_SPLIT_PE : DW_SPLIT_REG;
}
__host__ __device__ inline int dw_slab_base(const NetLayout& l, int u) {   // K-split slabs in front of unit u's ([slab][256 x 256] fp32)
  int n = 0;
  for (int v = 0; v < u; ++v) n += dw_unit_splits(l, v);
  return n;
}
__host__ __device__ inline int dw_total_slabs(const NetLayout& l) { return dw_slab_base(l, dw_units(l)); }

inline void make_workspace(const NetLayout& l, int64_t maxPts, int64_t maxRays, bool train, WorkspaceLayout* w) {
  w->nTiles = (maxPts + TILE_PTS - 1) / TILE_PTS;
  SpillLayout& s = w->sp;
  s.tensorElems = TILE_PTS * (int64_t)l.HD;   // offsets below are WITHIN a tile's block
  int64_t o = 0;
  s.A[0] = -1;                                // embedding-shaped operands are rebuilt by the dW kernel (offPeAux), not stored
  for (int i = 1; i <= l.L; ++i) { s.A[i] = o; o += s.tensorElems; }
  if (train) {
    // e4m3 (NetLayout::sp8): one byte per element; the TOP layer's P stays 16-bit (p_L = so w_out[o] sigma' is a per-column constant
    // times sigma': its e4m3 rounding error is the same for every saturated point and does not average out over the batch)
    for (int i = 0; i < l.L; ++i) { s.P[i] = o; o += ((l.sp8 & 2) && i < l.L - 1) ? s.tensorElems / 2 : s.tensorElems; }
    s.GB[0] = -1;
    for (int i = 1; i < l.L; ++i) { s.GB[i] = o; o += (l.sp8 & 1) ? s.tensorElems / 2 : s.tensorElems; }
    for (int i = 0; i < l.L; ++i) { s.ZB[i] = o; o += s.tensorElems; }
  }
  s.tileStride = o;
  s.totalElems = o * w->nTiles;
  o = s.totalElems;
  int64_t b = 0;
  w->offSpill = b; b += o * 2; b = (b + 255) / 256 * 256;
  (void)maxRays;
  w->offWgLoss = b; b += (train ? w->nTiles * 8 : 0) * 4; b = (b + 255) / 256 * 256;
  w->offDwPart = b; b += train ? (int64_t)dw_total_slabs(l) * DW_BLK * DW_BLK * 4 : 0; b = (b + 255) / 256 * 256;
  w->vecStride = round_up(l.L * l.HD + 2 * l.HD + 8, 64);   // [db_0..db_{L-1} | dwout(adjoint) | dwout(reverse) | dbout]
  w->offVecPart = b; b += train ? w->nTiles * (int64_t)w->vecStride * 4 : 0; b = (b + 255) / 256 * 256;
  w->offTotLoss = b; b += train ? maxPts * 4 : 0; b = (b + 255) / 256 * 256;
  w->offPeAux = b; b += train ? w->nTiles * TILE_PTS * 8 * 4 : 0; b = (b + 255) / 256 * 256;
  w->totalBytes = b + 256 + 4096;   // last 4 KB: timeline stamps of the -DISDF_DEBUG_HOOKS=1 development build
}

// ---- device helpers ---------------------------------------------------------
#if defined(__HIPCC__)

static __constant__ float kDirs[3][N_DIRS] = {
    {0.8506508f, 0.809017f, 0.5257311f, 1.f, 0.809017f, 0.8506508f, 0.309017f, 0.f, 0.5f, 0.f, -0.5257311f,
     -0.309017f, 0.f, -0.309017f, 0.309017f, 0.5f, 0.5f, 0.f, -0.5f, -0.809017f, -0.809017f},
    {0.f, 0.5f, 0.8506508f, 0.f, 0.5f, 0.f, 0.809017f, 0.5257311f, 0.309017f, 1.f, 0.8506508f, 0.809017f,
     0.5257311f, 0.809017f, 0.809017f, 0.309017f, -0.309017f, 0.f, 0.309017f, 0.5f, 0.5f},
    {0.5257311f, 0.309017f, 0.f, 0.f, -0.309017f, -0.5257311f, -0.5f, -0.8506508f, -0.809017f, 0.f, 0.f,
     -0.5f, 0.8506508f, 0.5f, 0.5f, 0.809017f, 0.809017f, 1.f, 0.809017f, 0.309017f, -0.309017f}};

template <bool F16> struct Op;
template <> struct Op<true> {
  typedef f16x8 v8; typedef f16x4 v4; typedef _Float16 e;
  static __device__ __forceinline__ f32x16 mfma(v8 a, v8 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
  }
  static __device__ __forceinline__ float clampf(float x) { return x; }
};
template <> struct Op<false> {
  typedef bf16x8 v8; typedef bf16x4 v4; typedef __bf16 e;
  static __device__ __forceinline__ f32x16 mfma(v8 a, v8 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
  }
  static __device__ __forceinline__ float clampf(float x) { return x; }
};

template <bool F16> __device__ __forceinline__ uint2 pack4(float a, float b, float c, float d) {
  typename Op<F16>::v4 v;
  v[0] = (typename Op<F16>::e)Op<F16>::clampf(a); v[1] = (typename Op<F16>::e)Op<F16>::clampf(b);
  v[2] = (typename Op<F16>::e)Op<F16>::clampf(c); v[3] = (typename Op<F16>::e)Op<F16>::clampf(d);
  return __builtin_bit_cast(uint2, v);
}
// v - fp16(v): the second operand of the compensated ("fp16x2") forward GEMMs
__device__ __forceinline__ float f16_residual(float v) { return v - (float)(_Float16)v; }
__device__ __forceinline__ void unpack4_bf16(uint2 u, float (&o)[4]) {
  o[0] = __uint_as_float(u.x << 16); o[1] = __uint_as_float(u.x & 0xffff0000u);
  o[2] = __uint_as_float(u.y << 16); o[3] = __uint_as_float(u.y & 0xffff0000u);
}

// ---- Philox4x32-10 (in-kernel RNG; not stream-compatible with torch) ----------
__device__ __forceinline__ uint4 philox4x32_10(uint4 ctr, uint2 key) {
  constexpr uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u, W0 = 0x9E3779B9u, W1 = 0xBB67AE85u;
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const uint32_t hi0 = __umulhi(M0, ctr.x), lo0 = M0 * ctr.x;
    const uint32_t hi1 = __umulhi(M1, ctr.z), lo1 = M1 * ctr.z;
    ctr = make_uint4(hi1 ^ ctr.y ^ key.x, lo1, hi0 ^ ctr.w ^ key.y, lo0);
    key.x += W0; key.y += W1;
  }
  return ctr;
}
__device__ __forceinline__ float u01(uint32_t x) { return (x >> 8) * (1.0f / 16777216.0f); }  // [0,1)


#endif  // __HIPCC__
}  // namespace isdf
