#!/bin/bash
# round 3, twelfth GPU call: deferred spill stores (issued behind the next GEMM's first weight request): parity, A/B
O=gpurun_out/r03l; mkdir -p $O
python -m pytest tests -m gpu -x -q 2>&1 | tail -4
bash tools/ab_bench.sh > $O/ab.txt 2>&1; cat $O/ab.txt
