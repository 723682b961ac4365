// Dev probe (not product code): what does one VALU / transcendental wave-instruction cost on a gfx950 SIMD, alone and in the shadow
// of MFMAs, at 1 / 2 / 4 waves per SIMD?  (Round 5: the pair-tile forward kernel's K = 256 stages take ~4.2 k cycles per SIMD for
// 64 MFMAs + 64 x 4 Softplus elements; is that the VALU pipe or missing overlap?)
//   mode 0  v_fma_f32 only (8 independent chains)              mode 1  v_exp_f32 only            mode 2  v_log_f32 only
//   mode 3  v_mfma_f32_32x32x16_f16 only (2 accumulators)      mode 4  MFMA + 8 v_fma per MFMA   mode 5  MFMA + 2 v_exp + 6 v_fma per MFMA
//   mode 6  MFMA + 4 v_fma per MFMA                            mode 7  v_cvt_pk_f16_f32          mode 8  v_pk_fma_f32 (8 chains)
//   mode 9  MFMA + 4 v_pk_fma_f32 per MFMA    mode 10  MFMA + 4 v_pk_add_f32    mode 11  MFMA + 4 ds_read_b128    mode 12  MFMA + 2 v_exp + 2 v_log
//   mode 13 MFMA + 4 v_add_f32 with a DPP-free VOP2 encoding (v_add vs v_fma: does the encoding matter?)
// Prints elapsed shader cycles (s_memtime) of the slowest wave of workgroup 0 and cycles per wave-instruction per SIMD.
// Build: hipcc --offload-arch=gfx950 -O2 valu_rate.hip -o valu_rate        Run: ./valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(2))) float f32x2;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

#define FMA8 asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n" \
                          "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9" \
                          : "+v"(f[0]), "+v"(f[1]), "+v"(f[2]), "+v"(f[3]), "+v"(f[4]), "+v"(f[5]), "+v"(f[6]), "+v"(f[7]) : "v"(c0), "v"(c1))
#define FMA4 asm volatile("v_fma_f32 %0, %0, %4, %5\n v_fma_f32 %1, %1, %4, %5\n v_fma_f32 %2, %2, %4, %5\n v_fma_f32 %3, %3, %4, %5" \
                          : "+v"(f[0]), "+v"(f[1]), "+v"(f[2]), "+v"(f[3]) : "v"(c0), "v"(c1))
#define FMA6 asm volatile("v_fma_f32 %0, %0, %6, %7\n v_fma_f32 %1, %1, %6, %7\n v_fma_f32 %2, %2, %6, %7\n v_fma_f32 %3, %3, %6, %7\n" \
                          "v_fma_f32 %4, %4, %6, %7\n v_fma_f32 %5, %5, %6, %7" \
                          : "+v"(f[0]), "+v"(f[1]), "+v"(f[2]), "+v"(f[3]), "+v"(f[4]), "+v"(f[5]) : "v"(c0), "v"(c1))
#define EXP8 asm volatile("v_exp_f32 %0, %0\n v_exp_f32 %1, %1\n v_exp_f32 %2, %2\n v_exp_f32 %3, %3\n v_exp_f32 %4, %4\n v_exp_f32 %5, %5\n" \
                          "v_exp_f32 %6, %6\n v_exp_f32 %7, %7" : "+v"(f[0]), "+v"(f[1]), "+v"(f[2]), "+v"(f[3]), "+v"(f[4]), "+v"(f[5]), "+v"(f[6]), "+v"(f[7]))
#define LOG8 asm volatile("v_log_f32 %0, %0\n v_log_f32 %1, %1\n v_log_f32 %2, %2\n v_log_f32 %3, %3\n v_log_f32 %4, %4\n v_log_f32 %5, %5\n" \
                          "v_log_f32 %6, %6\n v_log_f32 %7, %7" : "+v"(f[0]), "+v"(f[1]), "+v"(f[2]), "+v"(f[3]), "+v"(f[4]), "+v"(f[5]), "+v"(f[6]), "+v"(f[7]))
#define EXP2 asm volatile("v_exp_f32 %0, %0\n v_exp_f32 %1, %1" : "+v"(f[6]), "+v"(f[7]))
#define CVT8 asm volatile("v_cvt_pk_f16_f32 %0, %0, %1\n v_cvt_pk_f16_f32 %1, %1, %2\n v_cvt_pk_f16_f32 %2, %2, %3\n v_cvt_pk_f16_f32 %3, %3, %4\n" \
                          "v_cvt_pk_f16_f32 %4, %4, %5\n v_cvt_pk_f16_f32 %5, %5, %6\n v_cvt_pk_f16_f32 %6, %6, %7\n v_cvt_pk_f16_f32 %7, %7, %0" \
                          : "+v"(f[0]), "+v"(f[1]), "+v"(f[2]), "+v"(f[3]), "+v"(f[4]), "+v"(f[5]), "+v"(f[6]), "+v"(f[7]))
#define PK8 asm volatile("v_pk_fma_f32 %0, %0, %8, %8\n v_pk_fma_f32 %1, %1, %8, %8\n v_pk_fma_f32 %2, %2, %8, %8\n v_pk_fma_f32 %3, %3, %8, %8\n" \
                         "v_pk_fma_f32 %4, %4, %8, %8\n v_pk_fma_f32 %5, %5, %8, %8\n v_pk_fma_f32 %6, %6, %8, %8\n v_pk_fma_f32 %7, %7, %8, %8" \
                         : "+v"(g[0]), "+v"(g[1]), "+v"(g[2]), "+v"(g[3]), "+v"(g[4]), "+v"(g[5]), "+v"(g[6]), "+v"(g[7]) : "v"(gc))

#define PK4(op) asm volatile(op " %0, %0, %4, %4\n " op " %1, %1, %4, %4\n " op " %2, %2, %4, %4\n " op " %3, %3, %4, %4" \
                         : "+v"(g[0]), "+v"(g[1]), "+v"(g[2]), "+v"(g[3]) : "v"(gc))
#define PKA4 asm volatile("v_pk_add_f32 %0, %0, %4\n v_pk_add_f32 %1, %1, %4\n v_pk_add_f32 %2, %2, %4\n v_pk_add_f32 %3, %3, %4" \
                         : "+v"(g[0]), "+v"(g[1]), "+v"(g[2]), "+v"(g[3]) : "v"(gc))
#define ADD4 asm volatile("v_add_f32 %0, %0, %4\n v_add_f32 %1, %1, %4\n v_add_f32 %2, %2, %4\n v_add_f32 %3, %3, %4" \
                         : "+v"(f[0]), "+v"(f[1]), "+v"(f[2]), "+v"(f[3]) : "v"(c1))
#define EL4 asm volatile("v_exp_f32 %0, %0\n v_exp_f32 %1, %1\n v_log_f32 %2, %2\n v_log_f32 %3, %3" : "+v"(f[0]), "+v"(f[1]), "+v"(f[2]), "+v"(f[3]))
#define DSR4 asm volatile("ds_read_b128 %0, %4\n ds_read_b128 %1, %4 offset:1024\n ds_read_b128 %2, %4 offset:2048\n ds_read_b128 %3, %4 offset:3072\n s_waitcnt lgkmcnt(2)" \
                         : "=v"(d4[0]), "=v"(d4[1]), "=v"(d4[2]), "=v"(d4[3]) : "v"(laddr) : "memory")
#define MF0 acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc0, 0, 0, 0); __builtin_amdgcn_sched_barrier(0)
#define MF1 acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(b, a, acc1, 0, 0, 0); __builtin_amdgcn_sched_barrier(0)
#define SB __builtin_amdgcn_sched_barrier(0)

template <int mode>
__global__ __launch_bounds__(1024) void probe(int iters, unsigned long long* out, float* sink) {
  extern __shared__ char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  float f[8], c0 = 0.999f, c1 = 0.001f;
  f32x2 g[8], gc = {0.999f, 0.001f};
  for (int i = 0; i < 8; ++i) { f[i] = 0.5f + 0.01f * i + 0.001f * lane; g[i] = f32x2{f[i], f[i] + 1.f}; }
  f16x8 a, b;
  for (int t = 0; t < 8; ++t) { a[t] = (_Float16)(0.001f * (lane + t)); b[t] = (_Float16)(0.002f * (lane - t)); }
  f32x16 acc0 = {0}, acc1 = {0};
  uint4 d4[4] = {}; const int laddr = (tid & 63) * 16 + (tid >> 6) * 4096;
  __syncthreads();
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      if constexpr (mode == 0) { FMA8; }
      else if constexpr (mode == 1) { EXP8; }
      else if constexpr (mode == 2) { LOG8; }
      else if constexpr (mode == 3) { acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc0, 0, 0, 0); acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(b, a, acc1, 0, 0, 0); }
      else if constexpr (mode == 4) { acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc0, 0, 0, 0); __builtin_amdgcn_sched_barrier(0); FMA8; __builtin_amdgcn_sched_barrier(0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(b, a, acc1, 0, 0, 0); __builtin_amdgcn_sched_barrier(0); FMA8; __builtin_amdgcn_sched_barrier(0); }
      else if constexpr (mode == 5) { acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc0, 0, 0, 0); __builtin_amdgcn_sched_barrier(0); FMA6; EXP2; __builtin_amdgcn_sched_barrier(0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(b, a, acc1, 0, 0, 0); __builtin_amdgcn_sched_barrier(0); FMA6; EXP2; __builtin_amdgcn_sched_barrier(0); }
      else if constexpr (mode == 6) { acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc0, 0, 0, 0); __builtin_amdgcn_sched_barrier(0); FMA4; __builtin_amdgcn_sched_barrier(0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(b, a, acc1, 0, 0, 0); __builtin_amdgcn_sched_barrier(0); FMA4; __builtin_amdgcn_sched_barrier(0); }
      else if constexpr (mode == 7) { CVT8; }
      else if constexpr (mode == 9) { MF0; PK4("v_pk_fma_f32"); SB; MF1; PK4("v_pk_fma_f32"); SB; }
      else if constexpr (mode == 10) { MF0; PKA4; SB; MF1; PKA4; SB; }
      else if constexpr (mode == 11) { MF0; DSR4; SB; MF1; DSR4; SB; }
      else if constexpr (mode == 12) { MF0; EL4; SB; MF1; EL4; SB; }
      else if constexpr (mode == 13) { MF0; ADD4; SB; MF1; ADD4; SB; }
      else { PK8; }
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0.f;
  for (int i = 0; i < 8; ++i) s += f[i] + g[i][0] + g[i][1];
  s += acc0[0] + acc1[1] + (float)(d4[0].x ^ d4[1].y ^ d4[2].z ^ d4[3].w);
  if (s == 123.456f) sink[tid] = s;
  unsigned long long* ldsT = (unsigned long long*)smem;
  if (lane == 0) ldsT[tid >> 6] = t1 - t0;
  __syncthreads();
  if (blockIdx.x == 0 && tid == 0) {
    unsigned long long mx = 0;
    for (int w = 0; w < (int)(blockDim.x >> 6); ++w) mx = ldsT[w] > mx ? ldsT[w] : mx;
    out[0] = mx;
  }
}

int main() {
  unsigned long long* out; float* sink;
  CK(hipMalloc(&out, 64)); CK(hipMalloc(&sink, 1 << 16));
  typedef void (*kern_t)(int, unsigned long long*, float*);
  const int NM = 14;
  kern_t kerns[NM] = {probe<0>, probe<1>, probe<2>, probe<3>, probe<4>, probe<5>, probe<6>, probe<7>, probe<8>, probe<9>, probe<10>, probe<11>, probe<12>, probe<13>};
  for (int m = 0; m < NM; ++m) CK(hipFuncSetAttribute((const void*)kerns[m], hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024));
  const char* names[NM] = {"v_fma_f32", "v_exp_f32", "v_log_f32", "mfma 32x32x16 f16", "mfma + 8 v_fma", "mfma + 6 v_fma + 2 v_exp", "mfma + 4 v_fma", "v_cvt_pk_f16_f32", "v_pk_fma_f32",
                           "mfma + 4 v_pk_fma_f32", "mfma + 4 v_pk_add_f32", "mfma + 4 ds_read_b128", "mfma + 2 v_exp + 2 v_log", "mfma + 4 v_add_f32"};
  const int valu_per_k[NM] = {8, 8, 8, 0, 16, 16, 8, 8, 8, 8, 8, 8, 8, 8}, mfma_per_k[NM] = {0, 0, 0, 2, 2, 2, 2, 0, 0, 2, 2, 2, 2, 2};
  const int iters = 200;
  for (int wpb = 256; wpb <= 1024; wpb *= 2)
    for (int mode = 0; mode < NM; ++mode) {
      for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL(kerns[mode], dim3(256), dim3(wpb), 100 * 1024, 0, iters, out, sink);
        CK(hipDeviceSynchronize());
      }
      unsigned long long cyc; CK(hipMemcpy(&cyc, out, 8, hipMemcpyDeviceToHost));
      const int wps = wpb / 256;                      // waves per SIMD
      const double nv = (double)valu_per_k[mode] * 8 * iters * wps, nm = (double)mfma_per_k[mode] * 8 * iters * wps;
      printf("%d wave(s)/SIMD  %-26s %9llu cycles", wps, names[mode], cyc);
      if (nv > 0) printf("  %.2f cyc per VALU wave-instruction per SIMD", cyc / nv);
      if (nm > 0) printf("  %.2f cyc per MFMA per SIMD", cyc / nm);
      printf("\n");
    }
  return 0;
}
