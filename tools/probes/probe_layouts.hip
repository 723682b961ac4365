// Dev probe (not product code): pins down, on real gfx950 hardware,
//  (1) the operand/accumulator lane layouts of the bf16 MFMAs the kernels use,
//  (2) what ds_read_b64_tr_b16 delivers to each lane,
// so that isdf_amd/csrc kernels are written against measured facts, not guesses.
// Build: hipcc --offload-arch=gfx950 -O2 probe_layouts.hip -o probe_layouts
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <cstring>
#include <vector>

typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

// D[32x32] = A[32x16] * B[16x32]; A row-major [i][k], B row-major [k][j]
__global__ void mfma32(const float* A, const float* B, float* D) {
  int l = threadIdx.x;
  bf16x8 a, b;
  for (int t = 0; t < 8; ++t) {
    a[t] = (__bf16)A[(l & 31) * 16 + 8 * (l >> 5) + t];
    b[t] = (__bf16)B[(8 * (l >> 5) + t) * 32 + (l & 31)];
  }
  f32x16 acc = {0};
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
  for (int r = 0; r < 16; ++r) {
    int row = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5);
    int col = l & 31;
    D[row * 32 + col] = acc[r];
  }
}

// D[16x16] = A[16x32] * B[32x16]
__global__ void mfma16(const float* A, const float* B, float* D) {
  int l = threadIdx.x;
  bf16x8 a, b;
  for (int t = 0; t < 8; ++t) {
    a[t] = (__bf16)A[(l & 15) * 32 + 8 * (l >> 4) + t];
    b[t] = (__bf16)B[(8 * (l >> 4) + t) * 16 + (l & 15)];
  }
  f32x4 acc = {0};
  acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc, 0, 0, 0);
  for (int r = 0; r < 4; ++r) {
    int row = (l >> 4) * 4 + r;
    int col = l & 15;
    D[row * 16 + col] = acc[r];
  }
}

// LDS holds element e = its own index (as bf16-exact small ints via a table);
// lane l supplies the address of chunk perm(l) (4 contiguous elements).
__global__ void trprobe(int mode, float* out) {
  __shared__ __attribute__((aligned(16))) __bf16 lds[1024];
  int l = threadIdx.x;
  for (int e = l; e < 1024; e += 64) lds[e] = (__bf16)(float)(e & 255);  // exact in bf16
  __syncthreads();
  int chunk;
  if (mode == 0) chunk = l;                                   // linear
  else if (mode == 1) chunk = (l & 3) * 4 + ((l >> 2) & 3) + (l & 48);  // swapped in 16-group
  else chunk = l;
  bf16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(
      (bf16x4 __attribute__((address_space(3)))*)(lds + 4 * chunk));
  for (int j = 0; j < 4; ++j) out[l * 4 + j] = (float)v[j];
}

static float bf16r(float x) {  // round-to-nearest-even to bf16
  unsigned u; memcpy(&u, &x, 4);
  unsigned r = u + 0x7fff + ((u >> 16) & 1);
  r &= 0xffff0000u; float y; memcpy(&y, &r, 4); return y;
}

int main() {
  // ---- MFMA 32x32x16
  {
    std::vector<float> A(32 * 16), B(16 * 32), D(32 * 32), R(32 * 32, 0.f);
    for (int i = 0; i < 32; ++i) for (int k = 0; k < 16; ++k) A[i * 16 + k] = bf16r(0.01f * (i * 3 + 1) + 0.1f * k);
    for (int k = 0; k < 16; ++k) for (int j = 0; j < 32; ++j) B[k * 32 + j] = bf16r(0.02f * (j * 5 + 2) - 0.07f * k * k);
    for (int i = 0; i < 32; ++i) for (int j = 0; j < 32; ++j) { double s = 0; for (int k = 0; k < 16; ++k) s += (double)A[i * 16 + k] * B[k * 32 + j]; R[i * 32 + j] = (float)s; }
    float *dA, *dB, *dD;
    CK(hipMalloc(&dA, A.size() * 4)); CK(hipMalloc(&dB, B.size() * 4)); CK(hipMalloc(&dD, D.size() * 4));
    CK(hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice));
    mfma32<<<1, 64>>>(dA, dB, dD); CK(hipDeviceSynchronize());
    CK(hipMemcpy(D.data(), dD, D.size() * 4, hipMemcpyDeviceToHost));
    double me = 0, mr = 0; for (size_t i = 0; i < D.size(); ++i) { me = fmax(me, fabs(D[i] - R[i])); mr = fmax(mr, fabs(R[i])); }
    printf("MFMA32x32x16 bf16 layout check: max|err|=%g (max|ref|=%g) -> %s\n", me, mr, me < 1e-3 * mr ? "OK" : "MISMATCH");
  }
  {
    std::vector<float> A(16 * 32), B(32 * 16), D(16 * 16), R(16 * 16, 0.f);
    for (int i = 0; i < 16; ++i) for (int k = 0; k < 32; ++k) A[i * 32 + k] = bf16r(0.01f * (i * 3 + 1) + 0.1f * k);
    for (int k = 0; k < 32; ++k) for (int j = 0; j < 16; ++j) B[k * 16 + j] = bf16r(0.02f * (j * 5 + 2) - 0.03f * k * k);
    for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) { double s = 0; for (int k = 0; k < 32; ++k) s += (double)A[i * 32 + k] * B[k * 16 + j]; R[i * 16 + j] = (float)s; }
    float *dA, *dB, *dD;
    CK(hipMalloc(&dA, A.size() * 4)); CK(hipMalloc(&dB, B.size() * 4)); CK(hipMalloc(&dD, D.size() * 4));
    CK(hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice));
    mfma16<<<1, 64>>>(dA, dB, dD); CK(hipDeviceSynchronize());
    CK(hipMemcpy(D.data(), dD, D.size() * 4, hipMemcpyDeviceToHost));
    double me = 0, mr = 0; for (size_t i = 0; i < D.size(); ++i) { me = fmax(me, fabs(D[i] - R[i])); mr = fmax(mr, fabs(R[i])); }
    printf("MFMA16x16x32 bf16 layout check: max|err|=%g (max|ref|=%g) -> %s\n", me, mr, me < 1e-3 * mr ? "OK" : "MISMATCH");
  }
  // ---- ds_read_b64_tr_b16
  for (int mode = 0; mode < 2; ++mode) {
    float* dO; CK(hipMalloc(&dO, 256 * 4));
    trprobe<<<1, 64>>>(mode, dO); CK(hipDeviceSynchronize());
    float O[256]; CK(hipMemcpy(O, dO, sizeof(O), hipMemcpyDeviceToHost));
    printf("tr16 mode %d (lane: elems = source element index):\n", mode);
    for (int l = 0; l < 64; ++l) printf("  lane %2d: %3d %3d %3d %3d\n", l, (int)O[l * 4], (int)O[l * 4 + 1], (int)O[l * 4 + 2], (int)O[l * 4 + 3]);
  }
  hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
  printf("device: %s CUs=%d clock=%d kHz L2=%d\n", p.name, p.multiProcessorCount, p.clockRate, p.l2CacheSize);
  return 0;
}
