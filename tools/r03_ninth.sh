#!/bin/bash
# round 3, ninth GPU call: 32-point x 4-wave tiles (four workgroups per CU) for the <256,256> shape: parity, then A/B
O=gpurun_out/r03i; mkdir -p $O
ISDF_HIP_LIB=$PWD/variants/lib_t32.so timeout 600 python -m pytest tests -m gpu -x -q 2>&1 | tail -12 > $O/gpu_tests_t32.log; tail -5 $O/gpu_tests_t32.log
bash tools/ab_bench.sh > $O/ab.txt 2>&1
cat $O/ab.txt
