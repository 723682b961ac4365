#!/bin/bash
# round 3: gradient parity numbers at BASELINE size for two builds
for f in variants/lib_cfea06f.so variants/lib_head.so; do
  echo "== $f"
  ISDF_HIP_LIB=$PWD/$f python -m pytest tests/test_gpu_parity.py -m gpu -x -q -s -k "base_size_train_step and fp16x2" 2>&1 | grep -i "rel-L2\|passed\|failed"
done
