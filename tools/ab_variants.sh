#!/bin/bash
# A/B harness for kernel variants: for each variants/lib_*.so run the GPU parity subset + a short bench
for f in variants/lib_*.so; do
  cp $f isdf_amd/libisdf_hip.so
  echo "== $f"
  python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "train or forward or full_size or fused" 2>&1 | grep -E "^E  |^FAILED|passed|failed" | cut -c1-240 | tail -6
  python bench.py --steps 300 --warmup 30 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; j=json.loads(sys.stdin.read()); print(j['value'], j['ms_per_step'], j['kernel_ms'])"
done
