#!/bin/bash
# Builds variants/lib_trainpair.so: the tree's library with the ARCHIVED pair-tile train kernel (tools/variants/train_pair.hip) compiled in
# and dispatched for MODE 2 of the replicaCAD.json / scanNet.json network (fp16 operand family, fp16 second-order sweeps).  Round 5's
# experiment (profiles/r05_train_pair_experiment.txt): functional (sdf and losses bit-identical to the one-tile kernel, gradients
# to 7e-6), at parity in speed, not shipped.     usage: bash tools/variants/train_pair_build.sh [extra hipcc flags, e.g. -DISDF_DEBUG_HOOKS=1]
set -e
cd "$(dirname "$0")/../.."
T=$(mktemp -d /tmp/isdf_trainpair_XXXX)
mkdir -p $T/isdf_amd variants; cp -r isdf_amd/csrc $T/isdf_amd/csrc; cp -r include $T/include
cp tools/variants/train_pair.hip $T/isdf_amd/csrc/
python - "$T/isdf_amd/csrc/chain.hip" <<'PY'
import sys
p = sys.argv[1]; s = open(p).read()
s = s.replace("int launch_chain(const ChainParams& p0, int mode, int64_t nTiles, hipStream_t st) {",
              "bool train_pair_supported(const NetLayout& l);\nint launch_train_pair(const ChainParams& p, int64_t nTiles, hipStream_t st);\n\n"
              "int launch_chain(const ChainParams& p0, int mode, int64_t nTiles, hipStream_t st) {")
s = s.replace("    case 2: return launch_mode<2>(p, nTiles, st);",
              "    case 2: if (train_pair_supported(p.lay)) return launch_train_pair(p, nTiles, st);\n            return launch_mode<2>(p, nTiles, st);")
open(p, "w").write(s)
PY
FL="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-command-line-argument -fno-gpu-rdc $*"
UN="-fno-slp-vectorize -mllvm -pragma-unroll-threshold=1000000"
for s in chain dw sampler optim ingest capi; do /opt/rocm/bin/hipcc $FL -c $T/isdf_amd/csrc/$s.hip -o $T/$s.o & done
/opt/rocm/bin/hipcc $FL $UN -c $T/isdf_amd/csrc/fwd_pair.hip -o $T/fwd_pair.o &
/opt/rocm/bin/hipcc $FL $UN -c $T/isdf_amd/csrc/train_pair.hip -o $T/train_pair.o &
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o variants/lib_trainpair.so $T/chain.o $T/fwd_pair.o $T/train_pair.o $T/dw.o $T/sampler.o $T/optim.o $T/ingest.o $T/capi.o
echo built variants/lib_trainpair.so
