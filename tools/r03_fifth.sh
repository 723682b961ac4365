#!/bin/bash
# round 3, fifth GPU call: same-box A/B of the chain-kernel variants (whole-K weight requests, unrolled PE-shaped stages)
O=gpurun_out/r03e; mkdir -p $O
bash tools/ab_bench.sh > $O/ab.txt 2>&1
cat $O/ab.txt
ISDF_HIP_LIB=$PWD/variants/lib_c16u1.so python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "base_size or wide or shapes or realsense or full_steps or x3 or forward" 2>&1 | tail -3
