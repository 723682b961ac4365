#!/bin/bash
# round 3: two-operand GEMM (W a_hi + W a_lo in one pass over W): parity, A/B, exact-forward mode
O=gpurun_out/r03s; mkdir -p $O
python -m pytest tests -m gpu -x -q 2>&1 | tail -3
bash tools/ab_bench.sh > $O/ab.txt 2>&1; cat $O/ab.txt
for f in variants/lib_old.so variants/lib_w1.so; do ISDF_HIP_LIB=$PWD/$f python bench.py --fwd-operand fp16x2_full --steps 200 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import sys,json; j=json.loads(sys.stdin.read()); print('$f exact mode', j['value'], j['kernel_ms'], j['synchronised_step']['ms_per_step'])"; done
ISDF_HIP_LIB=$PWD/variants/lib_w1.so python bench.py --infer-points 8000000 2>/dev/null | tail -1 | python -c "
import sys,json; j=json.loads(sys.stdin.read()); print({k:(v['ms'],v['points_per_s']) for k,v in j['modes'].items()})"
