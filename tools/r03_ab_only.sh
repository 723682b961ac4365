#!/bin/bash
# round 3, sixteenth GPU call: re-read requests issued mid-GEMM (behind the last weight request) instead of after the last MFMA
O=gpurun_out/r03q; mkdir -p $O
bash tools/ab_bench.sh > $O/ab.txt 2>&1; cat $O/ab.txt
