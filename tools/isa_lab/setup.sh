#!/bin/bash
# Scratch copy + save-temps compile of chain.hip; leaves /tmp/asmlab/{orig.s,cmds.txt,out/}.  Optional $1: a patch to apply first
# (e.g. tools/variants/pk_fma_opsel_repro.patch).
set -e
ROOT=$(cd "$(dirname "$0")/../.." && pwd)
rm -rf /tmp/asmlab && mkdir -p /tmp/asmlab/isdf_amd /tmp/asmlab/out
cp -r $ROOT/isdf_amd/csrc /tmp/asmlab/isdf_amd/ && cp -r $ROOT/include /tmp/asmlab/
[ -n "${1:-}" ] && (cd /tmp/asmlab && patch -p1 -s -i "$(realpath $1)")
cd /tmp/asmlab/out
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-gpu-rdc -c ../isdf_amd/csrc/chain.hip -o chain.o -save-temps -v > ../v.log 2>&1
grep -E '^ "' ../v.log | sed -n '4,10p' > ../cmds.txt        # 1 assemble, 2 lld, 3 bundle, 4 host preprocess, 5-7 host compile
cp chain-hip-amdgcn-amd-amdhsa-gfx950.s ../orig.s
echo "/tmp/asmlab/orig.s: $(wc -l < ../orig.s) lines"
