// Probe (round 6): semantics of the gfx950 fp8 (OCP e4m3) conversions the 8-bit spill format of P / GB relies on.
//   v_cvt_pk_fp8_f32            two floats -> two e4m3 bytes (rounding? overflow: saturate or NaN?)
//   v_cvt_pk_f32_fp8            back
//   v_cvt_scalef32_pk_f16_fp8   two e4m3 bytes -> two fp16, times a per-lane scale (multiplies or divides?)
// build: hipcc --offload-arch=gfx950 -O2 tools/probes/fp8_cvt.hip -o tools/probes/fp8_cvt ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
typedef float f2 __attribute__((ext_vector_type(2)));
template <bool OVFL>
__global__ void k(const float* in, int* out, float* back, float* hb, float* encs, float* decs, float sc, int n) {
  // MODE.FP16_OVFL (bit 23 of hwreg 1): does it turn the conversions' overflow -> NaN into saturation?
  if (OVFL) __builtin_amdgcn_s_setreg(1 | (23 << 6) | (0 << 11), 1);
  int i = threadIdx.x;
  if (i >= n) return;
  int w = __builtin_amdgcn_cvt_pk_fp8_f32(in[2 * i], in[2 * i + 1], 0, false);
  out[i] = w;
  f2 r = __builtin_amdgcn_cvt_pk_f32_fp8(w, false);
  back[2 * i] = r[0]; back[2 * i + 1] = r[1];
  h2 h = __builtin_amdgcn_cvt_scalef32_pk_f16_fp8(w, sc, false);
  hb[2 * i] = (float)h[0]; hb[2 * i + 1] = (float)h[1];
  // scaled encode (does it divide or multiply by `sc`?) and scaled decode to f32
  typedef short s2 __attribute__((ext_vector_type(2)));
  s2 old = {0, 0};
  const s2 ws = __builtin_amdgcn_cvt_scalef32_pk_fp8_f32(old, in[2 * i], in[2 * i + 1], sc, false);
  const int wsi = __builtin_bit_cast(int, ws);
  f2 r1 = __builtin_amdgcn_cvt_pk_f32_fp8(wsi, false);          // what the scaled encode stored, decoded plainly
  f2 r2 = __builtin_amdgcn_cvt_scalef32_pk_f32_fp8(w, sc, false);   // the plain encode, decoded with the scale
  encs[2 * i] = r1[0]; encs[2 * i + 1] = r1[1]; decs[2 * i] = r2[0]; decs[2 * i + 1] = r2[1];
}
int main() {
  const float vals[] = {0.f, 1.f, -1.f, 1.0625f, 1.1875f, 1.125f, 1.375f, 0.0156f, 0.01f, 0.002f, 0.001f, 0.0005f, 447.f, 448.f, 460.f, 480.f, 500.f, 1000.f, 1e6f,
                        -1e6f, INFINITY, NAN, 3.3f, 7.7f, 100.f, 240.f, 17.f, 0.3f, 0.07f, 1e-5f, 65504.f, -0.75f};
  const int n2 = sizeof(vals) / sizeof(float), n = n2 / 2;
  float *din, *dback, *dhb, *denc, *ddec; int* dout;
  hipMalloc(&din, n2 * 4); hipMalloc(&dback, n2 * 4); hipMalloc(&dhb, n2 * 4); hipMalloc(&dout, n * 4); hipMalloc(&denc, n2 * 4); hipMalloc(&ddec, n2 * 4);
  hipMemcpy(din, vals, n2 * 4, hipMemcpyHostToDevice);
  for (int pass = 0; pass < 2; ++pass) {
  printf("---- MODE.FP16_OVFL = %d\n", pass);
  if (pass == 0) hipLaunchKernelGGL(k<false>, dim3(1), dim3(64), 0, 0, din, dout, dback, dhb, denc, ddec, 0.25f, n);
  else hipLaunchKernelGGL(k<true>, dim3(1), dim3(64), 0, 0, din, dout, dback, dhb, denc, ddec, 0.25f, n);
  float back[64], hb[64], encs[64], decs[64]; int out[32];
  hipMemcpy(encs, denc, n2 * 4, hipMemcpyDeviceToHost); hipMemcpy(decs, ddec, n2 * 4, hipMemcpyDeviceToHost);
  hipMemcpy(back, dback, n2 * 4, hipMemcpyDeviceToHost); hipMemcpy(hb, dhb, n2 * 4, hipMemcpyDeviceToHost); hipMemcpy(out, dout, n * 4, hipMemcpyDeviceToHost);
  printf("%14s %6s %14s %18s %22s %22s\n", "input", "byte", "cvt_pk_f32_fp8", "scalef32_f16(sc=.25)", "scaled-encode(sc=.25)", "scaled-decode(sc=.25)");
  for (int i = 0; i < n2; ++i) printf("%14.6g   0x%02x %14.6g %18.6g %22.6g %22.6g\n", vals[i], (out[i / 2] >> (8 * (i & 1))) & 0xff, back[i], hb[i], encs[i], decs[i]);
  }
  return 0;
}
