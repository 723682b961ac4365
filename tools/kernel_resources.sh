#!/bin/bash
# usage: tools/kernel_resources.sh file.hip [extra flags]  -> VGPRs / spills / scratch / occupancy of every kernel in the file
F=$1; shift
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-gpu-rdc -S --cuda-device-only "$@" $F -o /tmp/kr.s -Rpass-analysis=kernel-resource-usage 2>&1 |
  sed -n 's/.*remark: *//p' | sed 's/ *\[-Rpass-analysis.*//' |
  awk -F': ' '/^Function Name/{n=$2} /^VGPRs:/{v=$2} /^VGPRs Spill/{sp=$2} /^ScratchSize/{sc=$2} /^Occupancy/{o=$2} /^LDS Size/{print n, "vgpr", v, "spill", sp, "scratch", sc, "occ", o}' |
  c++filt | sed 's/isdf:://g; s/(ChainParams)//; s/void //'
