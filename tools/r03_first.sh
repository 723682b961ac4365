#!/bin/bash
# round 3, first GPU call: parity suite on the cleaned tree + same-box A/B of the two fp16 operand modes
O=gpurun_out/r03a; mkdir -p $O
python -m pytest tests -m gpu -x -q -s 2>&1 | tail -60 > $O/gpu_tests.log
for rep in 1 2; do
for op in fp16x2 fp16; do
  python bench.py --steps 300 --warmup 30 --no-cpu-baseline --fwd-operand $op 2>/dev/null | tail -1 > $O/bench_${op}_$rep.json
done
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r03a/bench_*.json')):
    j=json.load(open(f)); print(f, j['value'], j['kernel_ms'], j['synchronised_step']['ms_per_step'], j['roofline']['frac'])
PY
tail -15 $O/gpu_tests.log
