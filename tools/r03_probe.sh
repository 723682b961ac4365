#!/bin/bash
# round 3: one-off A/B (parity suite first)
O=gpurun_out/r03y; mkdir -p $O
python -m pytest tests -m gpu -x -q 2>&1 | tail -2
bash tools/ab_bench.sh > $O/ab.txt 2>&1; cat $O/ab.txt
