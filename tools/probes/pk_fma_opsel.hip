// Dev probe (not product code): does `v_pk_fma_f32 ... op_sel:[0,1,0]` (low result <- HIGH dword of a VGPR src1 pair) misbehave
// in ISOLATION on gfx950?  Round 4 found it dropping the low half's product in lanes 48-63 inside the chain kernel when two
// workgroups shared a CU (isdf_amd/isa_lint.py, profiles/r04_pk_fma_opsel_erratum.txt).  Here the instruction runs in a loop against
// two v_fma_f32, next to the form with the selector on src0, under several kinds of company on the CU:
//   hog kind (workgroups with bit 8 of the block index set): 0 none, 1 MFMA only, 2 LDS traffic only, 3 16-byte global stores only,
//   4 plain v_fma_f32 only, 5 MFMA + LDS + stores;   mix: the checkers also issue transcendental ops, LDS writes and 16-byte
//   global stores between the FMAs (the chain kernel epilogue's company).
// Result on MI355X (profiles/r04_pk_fma_opsel_probe.txt): wrong LOW results in lanes 48-63 as soon as MFMAs of another workgroup run
// on the CU; never for the form with the selector on src0, never in the high half, never without co-resident MFMA work.
// Build: hipcc --offload-arch=gfx950 -O2 pk_fma_opsel.hip -o pk_fma_opsel        Run: ./pk_fma_opsel [iters]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <vector>
#include <cmath>

typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

constexpr int NFORM = 6;   // 0: v_pk_fma_f32 op_sel:[0,1,0] (src1 high)   1: the same product with the selector on src0   2: form 0 with vdst == src1
                           // 3: v_pk_mul_f32 op_sel:[0,1]   4: v_pk_add_f32 op_sel:[0,1]   5: v_pk_fma_f32 op_sel:[0,0,1] (src2 high)

__device__ __forceinline__ float rnd(uint32_t& s) {   // [0.5, 1.5)
  s = s * 1664525u + 1013904223u;
  return __uint_as_float(0x3f000000u | (s >> 9)) ;   // [0.5, 1.0)
}

__global__ __launch_bounds__(512, 4) void probe(int iters, int hogKind, int mix, unsigned* bad, float* sink, float* samples) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  // kind 6 / 7: the MFMA company sits in the SAME workgroup -- waves 4-7 (one per SIMD, next to checker waves 0-3) / odd waves
  const bool hog = hogKind == 6 ? (tid >> 6) >= 4 : hogKind == 7 ? ((tid >> 6) & 1) : (hogKind >= 1 && ((blockIdx.x >> 8) & 1));
  float* lds = (float*)smem;
  if (hog) {
    f16x8 a, b;
    for (int t = 0; t < 8; ++t) { a[t] = (_Float16)(0.001f * (lane + t)); b[t] = (_Float16)(0.002f * (lane - t)); }
    f32x16 acc0 = {0}, acc1 = {0};
    float f0 = 0.5f + lane * 0.001f, f1 = 1.0f;
    for (int it = 0; it < iters; ++it) {
      if (hogKind == 1 || hogKind >= 5)
        for (int k = 0; k < 8; ++k) {
          acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc0, 0, 0, 0);
          acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(b, a, acc1, 0, 0, 0);
        }
      if (hogKind == 2 || hogKind == 5) {
        for (int k = 0; k < (hogKind == 2 ? 8 : 1); ++k) {
          f32x4 v = *(f32x4*)(lds + ((tid * 4 + (it + k) * 64) & 8191));
          a[0] += (_Float16)v[0];
          lds[(tid * 4 + 1 + k) & 8191] = acc0[0] + f0;
        }
      }
      if ((hogKind == 3) || (hogKind == 5 && (it & 15) == 0))
        *(f32x4*)(sink + 4096 + ((size_t)blockIdx.x * 512 + tid) * 4) = f32x4{acc0[1] + f0, acc1[2], acc0[3], acc1[4]};
      if (hogKind == 4)
        for (int k = 0; k < 64; ++k) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(f0) : "v"(f1));
    }
    acc0[0] += f0;
    sink[(size_t)blockIdx.x * 512 + tid] = acc0[0] + acc1[5];
    return;
  }
  uint32_t s = 0x9e3779b9u * (blockIdx.x * 512 + tid + 1);
  unsigned nbad[NFORM][2] = {};
  float keep = 0.f;
  for (int it = 0; it < iters; ++it) {
    f32x2 a = {rnd(s), rnd(s)}, b = {rnd(s), rnd(s)}, c = {rnd(s), rnd(s)};
    float t0 = 0.f, t1 = 0.f;
    if (mix) {   // the epilogue's company: transcendental ops, an LDS write, a 16-byte store
      t0 = __builtin_amdgcn_exp2f(a[0]); t1 = __builtin_amdgcn_logf(b[0] + 1.f);
      lds[(tid * 2 + it) & 8191] = t0;
      if ((it & 7) == 0) *(f32x4*)(sink + 4096 + ((size_t)blockIdx.x * 512 + tid) * 4) = f32x4{t0, t1, a[1], b[1]};
    }
    float r0, r1;
    asm volatile("v_fma_f32 %0, %2, %4, %5\n\tv_fma_f32 %1, %3, %4, %6"
                 : "=&v"(r0), "=&v"(r1) : "v"(a[0]), "v"(a[1]), "v"(b[1]), "v"(c[0]), "v"(c[1]));
    f32x2 d0, d1, d2 = b;
    asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,1,0]" : "=&v"(d0) : "v"(a), "v"(b), "v"(c));
    asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,0,0]" : "=&v"(d1) : "v"(b), "v"(a), "v"(c));
    asm volatile("v_pk_fma_f32 %0, %1, %0, %2 op_sel:[0,1,0]" : "+v"(d2) : "v"(a), "v"(c));
    // a dependent chain like the kernel's: 4 more of form 0, checked as a whole
    f32x2 ch = c, ch_ref = c;
    for (int k = 0; k < 4; ++k) {
      asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,1,0]" : "+v"(ch) : "v"(a), "v"(b));
      float x0, x1;
      asm volatile("v_fma_f32 %0, %2, %4, %5\n\tv_fma_f32 %1, %3, %4, %6"
                   : "=&v"(x0), "=&v"(x1) : "v"(a[0]), "v"(a[1]), "v"(b[1]), "v"(ch_ref[0]), "v"(ch_ref[1]));
      ch_ref = f32x2{x0, x1};
    }
    {
      f32x2 m, ad, s2;
      float m0, m1, a0_, a1_, s0, s1;
      asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[0,1]" : "=&v"(m) : "v"(a), "v"(b));
      asm volatile("v_pk_add_f32 %0, %1, %2 op_sel:[0,1]" : "=&v"(ad) : "v"(a), "v"(b));
      asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,0,1]" : "=&v"(s2) : "v"(a), "v"(b), "v"(c));
      asm volatile("v_mul_f32 %0, %2, %4\n\tv_mul_f32 %1, %3, %4" : "=&v"(m0), "=&v"(m1) : "v"(a[0]), "v"(a[1]), "v"(b[1]));
      asm volatile("v_add_f32 %0, %2, %4\n\tv_add_f32 %1, %3, %4" : "=&v"(a0_), "=&v"(a1_) : "v"(a[0]), "v"(a[1]), "v"(b[1]));
      asm volatile("v_fma_f32 %0, %2, %4, %6\n\tv_fma_f32 %1, %3, %5, %6" : "=&v"(s0), "=&v"(s1) : "v"(a[0]), "v"(a[1]), "v"(b[0]), "v"(b[1]), "v"(c[1]));
      nbad[3][0] += __float_as_uint(m[0]) != __float_as_uint(m0);   nbad[3][1] += __float_as_uint(m[1]) != __float_as_uint(m1);
      nbad[4][0] += __float_as_uint(ad[0]) != __float_as_uint(a0_); nbad[4][1] += __float_as_uint(ad[1]) != __float_as_uint(a1_);
      nbad[5][0] += __float_as_uint(s2[0]) != __float_as_uint(s0);  nbad[5][1] += __float_as_uint(s2[1]) != __float_as_uint(s1);
    }
    if (__float_as_uint(d0[0]) != __float_as_uint(r0) && atomicAdd(&bad[NFORM * 2 * 64], 1u) < 8) {   // what a wrong result looks like
      float* sp = samples + 12 * (atomicAdd(&bad[NFORM * 2 * 64 + 1], 1u) & 7);
      sp[0] = a[0]; sp[1] = a[1]; sp[2] = b[0]; sp[3] = b[1]; sp[4] = c[0]; sp[5] = c[1]; sp[6] = d0[0]; sp[7] = d0[1]; sp[8] = r0; sp[9] = r1;
      sp[10] = (float)lane; sp[11] = (float)blockIdx.x;
    }
    nbad[0][0] += __float_as_uint(d0[0]) != __float_as_uint(r0) || __float_as_uint(ch[0]) != __float_as_uint(ch_ref[0]);
    nbad[0][1] += __float_as_uint(d0[1]) != __float_as_uint(r1) || __float_as_uint(ch[1]) != __float_as_uint(ch_ref[1]);
    nbad[1][0] += __float_as_uint(d1[0]) != __float_as_uint(r0);
    nbad[1][1] += __float_as_uint(d1[1]) != __float_as_uint(r1);
    nbad[2][0] += __float_as_uint(d2[0]) != __float_as_uint(r0);
    nbad[2][1] += __float_as_uint(d2[1]) != __float_as_uint(r1);
    keep += t0 + t1;
  }
  for (int f = 0; f < NFORM; ++f)
    for (int h = 0; h < 2; ++h)
      if (nbad[f][h]) atomicAdd(&bad[(f * 2 + h) * 64 + lane], nbad[f][h]);
  if (keep == 123.456f) sink[0] = keep;
}

int main(int argc, char** argv) {
  const int iters = argc > 1 ? atoi(argv[1]) : 4000;
  unsigned* bad; float* sink;
  float* samples;
  CK(hipMalloc(&bad, (NFORM * 2 * 64 + 2) * sizeof(unsigned)));
  CK(hipMalloc(&samples, 8 * 12 * sizeof(float)));
  CK(hipMalloc(&sink, (4096 + (size_t)1024 * 512 * 4) * sizeof(float)));
  CK(hipFuncSetAttribute((const void*)probe, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024));
  const char* forms[NFORM] = {"v_pk_fma_f32 op_sel:[0,1,0] (src1 high->low)", "v_pk_fma_f32 op_sel:[1,0,0] (selector on src0)", "v_pk_fma_f32 op_sel:[0,1,0], vdst == src1",
                              "v_pk_mul_f32 op_sel:[0,1]", "v_pk_add_f32 op_sel:[0,1]", "v_pk_fma_f32 op_sel:[0,0,1] (src2 high->low)"};
  struct Cfg { int hog, mix, grid, lds; const char* what; } cfgs[] = {
    {0, 0, 512, 74 * 1024, "checkers only, two workgroups per CU"},
    {0, 1, 512, 74 * 1024, "checkers only (with trans ops, LDS writes, 16-byte stores), two per CU"},
    {1, 0, 512, 74 * 1024, "checker + MFMA-only hog"},
    {2, 0, 512, 74 * 1024, "checker + LDS-only hog"},
    {3, 0, 512, 74 * 1024, "checker + 16-byte-store-only hog"},
    {4, 0, 512, 74 * 1024, "checker + v_fma_f32-only hog"},
    {5, 0, 512, 74 * 1024, "checker + MFMA/LDS/store hog"},
    {5, 1, 512, 74 * 1024, "checker with trans ops, LDS writes, 16-byte stores + MFMA/LDS/store hog"},
    {5, 1, 1024, 36 * 1024, "the same, four workgroups per CU"},
    {6, 0, 256, 100 * 1024, "ONE workgroup per CU: waves 0-3 check, waves 4-7 run MFMAs (one of each per SIMD)"},
    {6, 1, 256, 100 * 1024, "the same, checkers with trans ops, LDS writes, 16-byte stores"},
    {7, 0, 256, 100 * 1024, "ONE workgroup per CU: even waves check, odd waves run MFMAs (SIMDs 0, 2 check only; 1, 3 MFMA only)"},
  };
  for (const Cfg& c : cfgs) {
    CK(hipMemset(bad, 0, (NFORM * 2 * 64 + 2) * sizeof(unsigned)));
    hipLaunchKernelGGL(probe, dim3(c.grid), dim3(512), c.lds, 0, iters, c.hog, c.mix, bad, sink, samples);
    CK(hipDeviceSynchronize());
    std::vector<unsigned> h(NFORM * 2 * 64 + 2);
    CK(hipMemcpy(h.data(), bad, h.size() * sizeof(unsigned), hipMemcpyDeviceToHost));
    const double checkers = (c.hog == 0 ? c.grid : c.grid / 2.0) * 512.0 * iters;   // (kinds 6, 7: half of every workgroup)
    printf("%-72s (%.2e lane-iterations per form)\n", c.what, checkers);
    for (int f = 0; f < NFORM; ++f) {
      unsigned long long q[2][4] = {};
      for (int hh = 0; hh < 2; ++hh) for (int l = 0; l < 64; ++l) q[hh][l / 16] += h[(f * 2 + hh) * 64 + l];
      printf("   %-48s wrong low results by lane quarter [%llu %llu %llu %llu], wrong high [%llu %llu %llu %llu]\n", forms[f],
             q[0][0], q[0][1], q[0][2], q[0][3], q[1][0], q[1][1], q[1][2], q[1][3]);
    }
    if (h[NFORM * 2 * 64]) {
      float sp[8 * 12];
      CK(hipMemcpy(sp, samples, sizeof(sp), hipMemcpyDeviceToHost));
      const int n = h[NFORM * 2 * 64] < 3 ? h[NFORM * 2 * 64] : 3;
      for (int i = 0; i < n; ++i) {
        const float* q = sp + 12 * i;
        printf("      e.g. lane %2.0f block %3.0f: a = (%.7f, %.7f) b = (%.7f, %.7f) c = (%.7f, %.7f): got low %.7f, a0*b1+c0 = %.7f, c0 = %.7f, a0*b0+c0 = %.7f\n",
               q[10], q[11], q[0], q[1], q[2], q[3], q[4], q[5], q[6], q[8], q[4], fmaf(q[0], q[2], q[4]));
      }
    }
  }
  return 0;
}
