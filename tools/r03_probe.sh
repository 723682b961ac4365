#!/bin/bash
# round 3: one-off A/B
O=gpurun_out/r03y; mkdir -p $O
ISDF_HIP_LIB=$PWD/variants/lib_nomem.so python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "base_size_train or step_full" 2>&1 | tail -2
bash tools/ab_bench.sh > $O/ab.txt 2>&1; cat $O/ab.txt
