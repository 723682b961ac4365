// Development instruments of the chain kernel, compiled in ONLY with -DISDF_DEBUG_HOOKS=1
// (tools/build_variants.py dbg=-DISDF_DEBUG_HOOKS=1, loaded by tools/timeline.py through ISDF_HIP_LIB).
// The shipped library carries none of it: ChainDebug is empty and every hook below is an empty inline.
#pragma once
#include <stdint.h>
#include <stdlib.h>
#include <hip/hip_runtime.h>

namespace isdf {

#ifndef ISDF_DEBUG_HOOKS
#define ISDF_DEBUG_HOOKS 0
#endif

#if ISDF_DEBUG_HOOKS
struct ChainDebug {
  int32_t alias;              // ISDF_DEBUG_ALIAS_SPILL=n: tiles share n spill regions (timing only, results garbage)
  int32_t stagger;            // ISDF_DEBUG_STAGGER=k: odd tiles start k kilo-cycles late
  int32_t stagger_gen;        // ISDF_DEBUG_STAGGER_GEN=1: ... the SECOND workgroup of every CU (dispatch round) instead of odd tiles
  int32_t all_waves;          // ISDF_DEBUG_TIMELINE=2: EVERY wave of workgroup 100 stamps, 64 slots each ([64 w + n]; no wall clocks)
  unsigned long long* times;  // ISDF_DEBUG_TIMELINE: [0..127] s_memtime stamps of workgroup 100, [128..] wall clocks
};
inline void chain_debug_from_env(ChainDebug& d, void* stamp_area) {
  if (const char* e = getenv("ISDF_DEBUG_ALIAS_SPILL")) d.alias = atoi(e);
  if (const char* e = getenv("ISDF_DEBUG_STAGGER")) d.stagger = atoi(e);
  if (const char* e = getenv("ISDF_DEBUG_STAGGER_GEN")) d.stagger_gen = atoi(e);
  if (const char* e = getenv("ISDF_DEBUG_TIMELINE")) { d.times = (unsigned long long*)stamp_area; d.all_waves = atoi(e) == 2; }
}
#if defined(__HIPCC__)
struct ChainStamps {
  const ChainDebug& d; int n = 0;
  __device__ explicit ChainStamps(const ChainDebug& dd, int n_cu = 256) : d(dd) {
    if (d.stagger && ((d.stagger_gen ? ((int)blockIdx.x / n_cu) : blockIdx.x) & 1)) {   // de-phase odd tiles / second-round workgroups (dispatch round = blockIdx / #CUs, as the kernel's genOdd)
      const unsigned long long t0 = __builtin_amdgcn_s_memtime();
      while (__builtin_amdgcn_s_memtime() - t0 < (unsigned long long)d.stagger * 1000ull) __builtin_amdgcn_s_sleep(64);
    }
  }
  __device__ void operator()() {   // phase boundary: wave 0 of workgroup 100 (slots 0..127), and its LAST wave (slots 384..511)
    if (d.times && d.all_waves) {
      if (blockIdx.x == 100 && (threadIdx.x & 63) == 0 && n < 64) d.times[(threadIdx.x >> 6) * 64 + n] = __builtin_amdgcn_s_memtime();
      ++n;
      return;
    }
    if (d.times && blockIdx.x == 100 && threadIdx.x == 0 && n < 128) d.times[n] = __builtin_amdgcn_s_memtime();   // (a persistent workgroup: its first pass)
    if (d.times && blockIdx.x == 100 && threadIdx.x == blockDim.x - 64 && n < 128) d.times[384 + n] = __builtin_amdgcn_s_memtime();
    ++n;
  }
  __device__ void wall(int slot) const {   // wall clock (s_memrealtime, 100 MHz) of every 4th workgroup: slots 128 .. 383 of the 512-entry
    // (4 KB) stamp area -- [128 + 2 k + slot] for workgroup 4 k, k < 128 -- below the last-wave stamps at 384..511
    if (d.times && !d.all_waves && threadIdx.x == 0 && (blockIdx.x & 3) == 0 && blockIdx.x < 4 * 128)
      d.times[128 + slot + blockIdx.x / 2] = __builtin_amdgcn_s_memrealtime();
  }
  __device__ int64_t spill_tile() const { return d.alias ? (int64_t)(blockIdx.x % d.alias) : (int64_t)blockIdx.x; }
};
#endif
#else
struct ChainDebug {};
inline void chain_debug_from_env(ChainDebug&, void*) {}
#if defined(__HIPCC__)
struct ChainStamps {
  __device__ explicit ChainStamps(const ChainDebug&, int = 0) {}
  __device__ void operator()() const {}
  __device__ void wall(int) const {}
  __device__ int64_t spill_tile() const { return (int64_t)blockIdx.x; }
};
#endif
#endif

}  // namespace isdf
