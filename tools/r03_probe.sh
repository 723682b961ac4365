#!/bin/bash
# round 3: one-off A/B
O=gpurun_out/r03y; mkdir -p $O
bash tools/ab_bench.sh > $O/ab.txt 2>&1; cat $O/ab.txt
