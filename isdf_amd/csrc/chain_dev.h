// Device helpers of the tile kernel (chain.hip: one 64-point tile per workgroup, two workgroups per CU): LDS-only
// barrier, buffer-descriptor loads and the hand-issued spill store, swizzle, DPP half-wave reductions,
// Softplus(beta = 100) on the base-2 units.
#pragma once
#include "isdf_common.h"
#include "chain_params.h"

namespace isdf {

constexpr float kHalfPi = 1.5707963267948966f;
constexpr float kBeta = 100.f;

// Workgroup barrier that only waits for this wave's LDS traffic.  The global
// spill tiles are thread-private (the lane that stores a piece is the lane that
// re-reads it), so global stores/loads may stay in flight across the barrier;
// __syncthreads() would drain them (s_waitcnt vmcnt(0)) at every layer.
__device__ __forceinline__ void lds_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
}

// All global traffic of the hot loops goes through buffer descriptors: address = SGPR descriptor + SGPR offset +
// ONE per-lane VGPR (lane*16) + immediate.  With flat 64-bit addresses the compiler kept a VGPR pair per matrix /
// spill tensor alive across the layer loops, spilled them, and reloaded them inside the MFMA stream behind
// s_waitcnt vmcnt(0).
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef __amdgpu_buffer_rsrc_t rsrc_t;
__device__ __forceinline__ rsrc_t make_rsrc(const void* base, uint32_t bytes) {
  return __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, bytes, 0x00020000);
}
constexpr int kAuxNT = 2;   // the non-temporal hint of a buffer load (aux bit 1)
// Cache policy of the spill traffic.  Rounds 2 - 5: everything the chain kernel re-reads itself (A, P) went out and came back
// non-temporal -- the stack (1.0 GB per step then) streamed through the 256 MB Infinity Cache anyway and only evicted the weight
// copies on its way.  Round 6: with e4m3 P / GB and no stored embedding the stack is 230 MB per step and FITS; stored and re-read
// with the default policy it is served from the cache: chain kernel 169.5 -> 154.9 us same-box, either hint alone: nothing
// (profiles/r06_cache_policy.txt).  The dW kernel's operand loads are each tensor's LAST use: they stay non-temporal (dW +2.5 us
// otherwise in the first A/B, nothing in the second), and so does the step tail's one read of the K-split slabs (chain +5 us
// with the default policy there); the slabs themselves are STORED with the default policy (tail -1.8 us).
// The 512-wide nets keep the non-temporal stream: their packed weight copies (10 MB) do not fit an XCD's L2 and live in the
// Infinity Cache themselves -- with default-policy spills the chain kernel re-fetches them from HBM (5.4 -> 7.8 ms at the --wide
// bench workload).  Larger batches of the 256-wide net (54 k .. 729 k points) are neutral to 3 % faster with the default policy.
constexpr bool spill_store_nt(int hd) { return hd > 256; }            // the chain kernel's spill stores
constexpr int spill_load_aux(int hd) { return hd > 256 ? kAuxNT : 0; }   // ... and its re-reads
constexpr bool spill_zb_nt(int hd) { return hd <= 256; }               // ZB (last sweep -> dW only): around the cache where the cache holds the stack
constexpr int kAuxDwLoad = kAuxNT;      // the dW kernel's operand loads
template <int AUX> __device__ __forceinline__ uint4 bload16(rsrc_t r, int voff, int soff) {
  const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, AUX);
  return make_uint4(v[0], v[1], v[2], v[3]);
}
// 16-byte non-temporal store of frag16 piece `c` (byte c*1024 past soff; c&3 goes into the immediate).
// Hand-issued: with an SGPR soffset the compiler inserts NO wait state between a buffer_store_dwordx4 and a
// following VALU write of its data registers (it assumes that form is exempt from the >64-bit store-data hazard).
// On gfx950 it is not: a v_pk_mul_f32 scheduled right behind the store corrupted bytes 4-5 of lanes 12-15 of
// every 16 in memory (found as NaN weight gradients; the same code through global_store_dwordx4 was clean).
// srd = {base_lo, base_hi, bytes, 0x00020000} in SGPRs.  The compiler's vmcnt bookkeeping does not see this
// store; an uncounted store can only make its later counted waits stricter, never too weak.
typedef int i32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ i32x4 make_srd(const void* base, uint32_t bytes) {
  const unsigned long long b = (unsigned long long)base;
  i32x4 d;
  d[0] = __builtin_amdgcn_readfirstlane((int)(uint32_t)b);
  d[1] = __builtin_amdgcn_readfirstlane((int)(uint32_t)(b >> 32)) & 0xffff;
  d[2] = (int)bytes; d[3] = 0x00020000;
  return d;
}
// (write-through `sc1` instead of `nt` measured the same: profiles/r02_ab_chain_variants.txt)
#define ISDF_BSTORE16_NT(IMM) \
  asm volatile("buffer_store_dwordx4 %0, %1, %2, %3 offen offset:" #IMM " nt\n\ts_nop 1" ::"v"(v), "v"(voff), "s"(srd), "s"(soff) : "memory")
#define ISDF_BSTORE16_DF(IMM) \
  asm volatile("buffer_store_dwordx4 %0, %1, %2, %3 offen offset:" #IMM "\n\ts_nop 1" ::"v"(v), "v"(voff), "s"(srd), "s"(soff) : "memory")
template <bool NT>
__device__ __forceinline__ void bstore16_nt(uint4 x, i32x4 srd, int voff, int soff, int c) {
  u32x4 v; v[0] = x.x; v[1] = x.y; v[2] = x.z; v[3] = x.w;
  soff += (c >> 2) * 4096;
  if (NT) {
    switch (c & 3) {
      case 0: ISDF_BSTORE16_NT(0); break;
      case 1: ISDF_BSTORE16_NT(1024); break;
      case 2: ISDF_BSTORE16_NT(2048); break;
      default: ISDF_BSTORE16_NT(3072); break;
    }
  } else {
    switch (c & 3) {
      case 0: ISDF_BSTORE16_DF(0); break;
      case 1: ISDF_BSTORE16_DF(1024); break;
      case 2: ISDF_BSTORE16_DF(2048); break;
      default: ISDF_BSTORE16_DF(3072); break;
    }
  }
}

__device__ __forceinline__ int swz(int row, int colbytes) { return colbytes ^ ((row & 15) << 4); }

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
// The pair-tile forward kernel (fwd_pair.hip) works on the base-2 image of the pre-activation, x = beta log2(e) (acc + b) =
// fma(kC1, acc, kC1 b) with the bias pre-scaled when it is staged in LDS, and scales back after the max: a = kC2 max(x, log2(1 + 2^x))
// -- five plain VALU operations and two transcendental ones per element instead of six and two; the same function to fp32 rounding
// (its outputs sit <= 3e-5 of the output scale from the one-tile kernel's: profiles/r05_fwd_pair_v3_ab.txt).  Why an operation
// matters: on this chip every VALU wave-instruction next to MFMAs costs ~4 issue cycles of its SIMD, a transcendental 8, an MFMA 8
// of its 32 (tools/probes/valu_rate.hip, profiles/r05_probe_valu_rate.txt), and a K = 256 layer's Softplus epilogue then needs MORE
// issue cycles than its 64 MFMAs take to execute: the instruction count of the epilogues, not the matrix pipe, bounds these kernels.
__device__ __forceinline__ float softplus_x(float acc, float bs) {   // bs = kC1 * bias
  const float x = __builtin_fmaf(kC1, acc, bs);
  const float t = __builtin_amdgcn_exp2f(fminf(x, 30.f));
  return kC2 * fmaxf(x, __builtin_amdgcn_logf(1.f + t));
}
// sigma'(z) recovered from the stored activation: 1 - exp(-beta a)
__device__ __forceinline__ float s1_from_a(float a) {
  return 1.f - __builtin_amdgcn_exp2f(-kC1 * a);
}

}  // namespace isdf
