#!/bin/bash
# round 3: 128-point tiles (one 8-wave workgroup per CU, 216 VGPRs, every weight fragment feeds four MFMAs): parity subset on the variant, A/B
O=gpurun_out/r03u; mkdir -p $O
ISDF_HIP_LIB=$PWD/variants/lib_t128.so python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "base_size and not full and not exact" 2>&1 | tail -4
bash tools/ab_bench.sh > $O/ab.txt 2>&1; cat $O/ab.txt
