#!/bin/bash
# usage: reasm.sh edited.s out_lib.so   (after setup.sh; the other objects come from isdf_amd/build/)
set -e
ROOT=$(cd "$(dirname "$0")/../.." && pwd)
OUT=$(realpath -m "$2")
cd /tmp/asmlab/out
cp "$1" chain-hip-amdgcn-amd-amdhsa-gfx950.s
for i in 1 2 3 5 6 7; do eval "$(sed -n "${i}p" ../cmds.txt)" > /dev/null 2>&1; done
objs=""
for s in capi dw ingest optim sampler; do objs="$objs $ROOT/isdf_amd/build/$s.o"; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$OUT" chain.o $objs
echo built $OUT
