// Per-frame ingest and keyframe-test kernels ("next" tier).
//   transform.pointcloud_from_depth_torch   isdf/geometry/transform.py:169-196
//   transform.estimate_pointcloud_normals   isdf/geometry/transform.py:215-270
//   render.sdf_render_depth                 isdf/modules/render.py:12-35
//   Trainer.is_keyframe (sort + ratio)      isdf/modules/trainer.py:597-609
// The reference runs the normal estimation as ~10 eager ops with a 70 MB index
// tensor per 680x1200 frame (0.37 s on 8 CPU threads).  Here it is one stencil
// pass: 4 B read + 12 B written per pixel, 13 MB per 680x1200 frame (2.5 us of HBM
// time).  The pass is VALU-bound (nine square roots, five divisions and the argmin with the
// reference's NaN rule per pixel): 13.1 us per frame with correctly rounded sqrt / division, 9.0 us
// with the hardware-rate v_sqrt_f32 / v_rcp_f32 (1 ulp) it ships with -- the reference fixture
// (99.9 % of the pixels within 1e-4) passes either way: profiles/r03_ingest_fastmath.txt.
#include "isdf_common.h"

namespace isdf {

// hardware-rate square root and reciprocal (1 ulp each): the 1e-4 bar on the normals does not need correct rounding
__device__ __forceinline__ float n_sqrt(float x) { return __builtin_amdgcn_sqrtf(x); }
__device__ __forceinline__ float n_div(float a, float b) { return a * __builtin_amdgcn_rcpf(b); }

__device__ __forceinline__ void pix_point(const float* __restrict__ depth, int H, int W, int i, int j, float fx,
                                          float fy, float cx, float cy, float& x, float& y, float& z) {
  if (i < 0 || i >= H || j < 0 || j >= W) { x = y = z = __int_as_float(0x7fc00000); return; }   // NaN padding
  z = depth[(int64_t)i * W + j];
  x = n_div(__fmul_rn(z, (float)j - cx), fx);     // transform.py:190-191
  y = n_div(__fmul_rn(z, (float)i - cy), fy);
}

// One block = 32 x 16 pixels, two per thread (1 594 blocks for a 680x1200 frame: ONE round on 256 CUs x 8 blocks).  The
// camera-frame points of the block and its 2-pixel halo (36 x 20) are computed ONCE into LDS (two IEEE divisions per
// point); the 8 neighbours of a pixel are then three LDS reads each.  (The first version recomputed the nine points per
// pixel -- 18 divisions and 9 global loads per thread -- in 3 188 blocks, i.e. two rounds: 17.0 us per frame.)
constexpr int NRM_BH = 16;
__global__ __launch_bounds__(256) void normals_kernel(const float* __restrict__ depth, int H, int W, float fx,
                                                      float fy, float cx, float cy, float* __restrict__ normals) {
  __shared__ float px[NRM_BH + 4][37], py[NRM_BH + 4][37], pz[NRM_BH + 4][37];
  const int tx = threadIdx.x & 31, ty0 = threadIdx.x >> 5;
  const int j0 = blockIdx.x * 32 - 2, i0 = blockIdx.y * NRM_BH - 2;
  for (int t = threadIdx.x; t < (NRM_BH + 4) * 36; t += 256) {
    const int li = t / 36, lj = t - li * 36;
    float x, y, z;
    pix_point(depth, H, W, i0 + li, j0 + lj, fx, fy, cx, cy, x, y, z);
    px[li][lj] = x; py[li][lj] = y; pz[li][lj] = z;
  }
  __syncthreads();
  const int j = blockIdx.x * 32 + tx;
  constexpr int d = 2;
  const int ly[8] = {-d, -d, 0, d, d, d, 0, -d}, lx[8] = {0, d, d, d, 0, -d, -d, -d};
#pragma unroll
  for (int half = 0; half < NRM_BH / 8; ++half) {
    const int ty = ty0 + 8 * half;
    const int i = blockIdx.y * NRM_BH + ty;
    if (i >= H || j >= W) continue;
    const float p1x = px[ty + 2][tx + 2], p1y = py[ty + 2][tx + 2], p1z = pz[ty + 2][tx + 2];
    float qx[8], qy[8], qz[8], len[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const float x = px[ty + 2 + ly[k]][tx + 2 + lx[k]], y = py[ty + 2 + ly[k]][tx + 2 + lx[k]], z = pz[ty + 2 + ly[k]][tx + 2 + lx[k]];
      qx[k] = x - p1x; qy[k] = y - p1y; qz[k] = z - p1z;
      len[k] = n_sqrt(__fadd_rn(__fadd_rn(__fmul_rn(qx[k], qx[k]), __fmul_rn(qy[k], qy[k])), __fmul_rn(qz[k], qz[k])));
    }
    float best = 0.f; int bk = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      float s = __fadd_rn(len[k], len[(k + 2) & 7]);
      if (s != s) s = INFINITY;                       // diff[isnan] = inf, transform.py:259
      if (k == 0 || s < best) { best = s; bk = k; }   // argmin keeps the first minimum
    }
    float ax = 0, ay = 0, az = 0, bx = 0, by = 0, bz = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      if (k == bk) { ax = qx[k]; ay = qy[k]; az = qz[k]; bx = qx[(k + 2) & 7]; by = qy[(k + 2) & 7]; bz = qz[(k + 2) & 7]; }
    }
    // torch.cross then normalise (transform.py:262-267)
    const float nx = __fadd_rn(__fmul_rn(ay, bz), -__fmul_rn(az, by));
    const float ny = __fadd_rn(__fmul_rn(az, bx), -__fmul_rn(ax, bz));
    const float nz = __fadd_rn(__fmul_rn(ax, by), -__fmul_rn(ay, bx));
    const float nn = n_sqrt(__fadd_rn(__fadd_rn(__fmul_rn(nx, nx), __fmul_rn(ny, ny)), __fmul_rn(nz, nz)));
    float* o = normals + ((int64_t)i * W + j) * 3;
    o[0] = n_div(nx, nn); o[1] = n_div(ny, nn); o[2] = n_div(nz, nn);
  }
}

// One thread per ray.  The reference sorts the ray's samples by z (stable) and takes the FIRST one with a negative sdf, with two quirks
// kept: no negative sample -> sorted sample 0; a crossing at the LAST sorted sample -> depth 0 (render.py:19-31).  No sort is needed for
// that (round 6; the insertion sort of rounds 1-5 kept two dynamically indexed 64-entry arrays in scratch memory, the only scratch in the
// library): the first negative sample of the sorted order is the negative sample with the smallest (z, original index), found in one pass
// -- the smallest over ALL samples if none is negative -- and it is the last sorted sample exactly when S - 1 samples precede it, a second
// pass over the ray's z values (27 floats, L1-resident).  Same comparisons as the stable sort, so the same sample, bit for bit.
__global__ void render_depth_kernel(const int32_t* __restrict__ n_valid, int64_t n_host, int S,
                                    const float* __restrict__ z_vals, const float* __restrict__ sdf,
                                    const float* __restrict__ depth_sample, float kf_dist_th,
                                    float* __restrict__ view_depth, int32_t* __restrict__ below_count) {
  const int64_t R = n_valid ? (int64_t)*n_valid : n_host;
  const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  bool below = false;
  if (r < R) {
    const float* z = z_vals + r * S;
    const float* s = sdf + r * S;
    int cn = -1, ca = 0;                 // first-in-sorted-order among the negative samples / among all samples
    float zn = 0.f, za = z[0];
    for (int k = 0; k < S; ++k) {
      const float zk = z[k];
      if (zk < za) { za = zk; ca = k; }                          // strict: of equal z the earlier sample stays first (stable sort)
      if (s[k] < 0.f && (cn < 0 || zk < zn)) { zn = zk; cn = k; }
    }
    const int c = cn >= 0 ? cn : ca;
    const float zc = cn >= 0 ? zn : za, sc = s[c];
    int before = 0;                      // samples in front of c in the sorted order
    for (int k = 0; k < S; ++k) before += (z[k] < zc || (z[k] == zc && k < c)) ? 1 : 0;
    float dpt = __fadd_rn(zc, sc);
    if (before == S - 1) dpt = 0.f;
    view_depth[r] = dpt;
    if (depth_sample) {
      const float ds = depth_sample[r];
      below = fabsf(dpt - ds) / ds < kf_dist_th;        // trainer.py:604-606
    }
  }
  if (below_count) {
    const unsigned long long m = __ballot(below);
    if ((threadIdx.x & 63) == 0 && m) atomicAdd(below_count, (int)__popcll(m));
  }
}

int launch_normals(const float* depth, int H, int W, float fx, float fy, float cx, float cy, float* normals,
                   hipStream_t st) {
  hipLaunchKernelGGL(normals_kernel, dim3((W + 31) / 32, (H + NRM_BH - 1) / NRM_BH), dim3(256), 0, st, depth, H, W, fx, fy, cx, cy,
                     normals);
  return isdf_launch_status();
}

int launch_render_depth(const int32_t* n_valid, int64_t n_host, int64_t max_rays, int S, const float* z,
                        const float* sdf, const float* depth_sample, float th, float* view, int32_t* below,
                        hipStream_t st) {
  if (below && hipMemsetAsync(below, 0, 4, st) != hipSuccess) return ISDF_EHIP;
  hipLaunchKernelGGL(render_depth_kernel, dim3((unsigned)((max_rays + 127) / 128)), dim3(128), 0, st, n_valid,
                     n_host, S, z, sdf, depth_sample, th, view, below);
  return isdf_launch_status();
}

}  // namespace isdf
