#!/bin/bash
# round 3, fourteenth GPU call: accumulators start from the MFMA's inline-constant C = 0 instead of 32 v_mov per GEMM: parity, A/B
O=gpurun_out/r03n; mkdir -p $O
python -m pytest tests -m gpu -x -q 2>&1 | tail -4
bash tools/ab_bench.sh > $O/ab.txt 2>&1; cat $O/ab.txt
