#!/bin/bash
# round 3, thirteenth GPU call: the exact-forward operand mode "fp16x2_full" (OPER 3): parity suite, its bench line, inference
O=gpurun_out/r03m; mkdir -p $O
python -m pytest tests/test_gpu_parity.py -m gpu -x -q -s -k "base_size_forward or exact_forward" 2>&1 | grep -v "^$" | tail -25 > $O/exact_tests.txt; cat $O/exact_tests.txt
python -m pytest tests -m gpu -x -q 2>&1 | tail -4
python bench.py --fwd-operand fp16x2_full --steps 200 --no-cpu-baseline > $O/bench_exact.json 2> $O/bench_exact.err; tail -c 1500 $O/bench_exact.json
python bench.py --infer-points 8000000 > $O/bench_infer.json 2> $O/bench_infer.err; tail -c 2500 $O/bench_infer.json
