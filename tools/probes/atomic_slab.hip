// Dev probe (not product code): what would accumulating the dW K-split partials with fp32 atomics cost, instead of writing 36 slabs per unit
// (66 MB) and summing them in the step tail?  252 workgroups x 512 threads, each adds its 256 x 256 fp32 tile into ONE of
// 7 shared 256 KB tiles (36 workgroups per tile, spread over all XCDs), element (thread, k) -> tile[k * 512 + thread] (coalesced).
//   mode 0: agent-scope atomicAdd (the only form that is correct across XCDs)      mode 1: plain 16-byte stores of the same bytes into private
//   slabs (what dw.hip does today)      mode 2: unsafe (no-return, relaxed) atomic add via __builtin_amdgcn_global_atomic_fadd_f32
// Build: hipcc --offload-arch=gfx950 -O2 atomic_slab.hip -o atomic_slab        Run: ./atomic_slab
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

template <int MODE>
__global__ __launch_bounds__(512) void k(float* acc, float* slabs) {
  const int unit = blockIdx.x / 36, tid = threadIdx.x;
  float v[128];
#pragma unroll
  for (int i = 0; i < 128; ++i) v[i] = 1.0f + 0.001f * (float)((tid + i + blockIdx.x) & 7);
  if (MODE == 1) {
    typedef float f4 __attribute__((ext_vector_type(4)));
    f4* dst = (f4*)(slabs + (size_t)blockIdx.x * 65536);
#pragma unroll
    for (int i = 0; i < 32; ++i) { f4 q = {v[4 * i], v[4 * i + 1], v[4 * i + 2], v[4 * i + 3]}; __builtin_nontemporal_store(q, dst + i * 512 + tid); }
  } else {
    float* dst = acc + (size_t)unit * 65536;
#pragma unroll
    for (int i = 0; i < 128; ++i) {
      if (MODE == 0) atomicAdd(dst + i * 512 + tid, v[i]);
      else __hip_atomic_fetch_add(dst + i * 512 + tid, v[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}

int main() {
  float *acc, *slabs;
  CK(hipMalloc(&acc, 7 * 65536 * 4)); CK(hipMalloc(&slabs, (size_t)252 * 65536 * 4));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const char* names[3] = {"atomicAdd (agent scope, returns)", "16-byte non-temporal stores into private slabs", "relaxed agent-scope fetch_add, result unused"};
  for (int mode = 0; mode < 3; ++mode) {
    float best = 1e9f;
    for (int rep = 0; rep < 6; ++rep) {
      CK(hipMemset(acc, 0, 7 * 65536 * 4));
      CK(hipDeviceSynchronize());
      CK(hipEventRecord(e0));
      if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(252), dim3(512), 0, 0, acc, slabs);
      else if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(252), dim3(512), 0, 0, acc, slabs);
      else hipLaunchKernelGGL(k<2>, dim3(252), dim3(512), 0, 0, acc, slabs);
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      if (rep > 0 && ms < best) best = ms;
    }
    float h[4]; CK(hipMemcpy(h, acc, 16, hipMemcpyDeviceToHost));
    printf("%-52s %8.1f us   (acc[0] = %.3f, expected ~36 x 1.00x for the atomic modes)\n", names[mode], best * 1e3f, h[0]);
  }
  return 0;
}
