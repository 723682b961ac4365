// Kernel-argument structs shared by the launchers (capi.hip) and the kernels.
#pragma once
#include "isdf_common.h"
#include "chain_debug.h"

namespace isdf {

struct ChainParams {
  NetLayout lay;
  isdf_loss_cfg loss;
  const float* params;
  const uint16_t* shadow;
  const float* pts;           // [P,3]
  const float* noise;         // [P] or null
  float noise_std; uint64_t noise_seed, noise_off;   // in-kernel Philox noise when noise == null
  const int32_t* n_valid;     // device: rays (points = n_valid*S); null -> n_points_host
  int64_t n_points_host;
  int32_t S;
  // per-ray inputs (train)
  const float* z_vals; const float* depth; const float* dirsC; const float* dirsW; const float* normals;
  const float* pc_bounds; const float* pc_grad_vec;
  // outputs
  float* sdf; float* sdf_grad; float* tot_loss_mat;
  float* tot_ws;              // [maxPts] per-point total loss (train)
  float* wg_loss;             // [nTiles][8]
  float* vec_part;            // [nTiles][vecStride] bias / out-layer gradient partials (no atomics)
  int32_t vecStride;
  uint16_t* spill; SpillLayout sp;
  float* pe_aux;              // [nTiles*TILE_PTS][8] (train): x' (3), pad, gbar in x' space (3), pad -- what the dW kernel rebuilds the two
                              // embedding-shaped operands (the embedding itself and Ebar = J_pe gbar) from instead of reading them back
  int32_t n_cu;               // compute units of the device the launch goes to (set by launch_chain): dispatch round of a workgroup = blockIdx / n_cu
  ChainDebug dbg;             // empty in the shipped build (chain_debug.h)
};

struct DwParams {
  NetLayout lay;
  SpillLayout sp;
  const uint16_t* spill;
  const float* pe_aux;   // [nTiles*TILE_PTS][8], written by the chain kernel (ChainParams::pe_aux)
  const int32_t* n_valid; int64_t n_points_host; int32_t S;
  float* dwPart;   // [dw_total_slabs][256*256] (isdf_common.h)
};


}  // namespace isdf
