#!/bin/bash
# usage: spill_lines.sh file.hip [extra flags]: source lines (of the 256/f16/train chain kernel) that carry scratch spill traffic
F=$1; shift
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-gpu-rdc -I/root/repo/isdf_amd/csrc -I/root/repo/include -S --cuda-device-only -g1 "$@" $F -o /tmp/sl.s 2>/dev/null
awk '/^_ZN4isdf12chain_kernelILi256ELi256ELb1ELi2EEEvNS_11ChainParamsE:/{f=1} f{ if($1==".loc") loc=$2":"$3; if ($1 ~ /^scratch_(store|load)/) print loc, $1 } f&&/s_endpgm/{exit}' /tmp/sl.s | sort | uniq -c | sort -k2,2 -t' ' | sort -t: -k2 -n | awk '{print}' 
