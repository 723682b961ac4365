#!/bin/bash
# Same-box A/B of the pair-tile chain kernel (chain_pair.hip) against the one-tile kernel (default):
# GPU parity subset with the pair kernel, then bench.py with both.  usage: tools/pair_ab.sh [outdir]
out=${1:-gpurun_out/pair}
mkdir -p $out
ISDF_CHAIN_PAIR=1 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "train or full_size or fused or step" > $out/tests.log 2>&1
grep -E "^E  |^FAILED|passed|failed" $out/tests.log | cut -c1-300 | tail -12
for rep in 1 2; do
  ISDF_CHAIN_PAIR=1 python bench.py --steps 300 --warmup 50 --no-cpu-baseline 2>/dev/null | tail -1 > $out/bench_pair_$rep.json
  python bench.py --steps 300 --warmup 50 --no-cpu-baseline 2>/dev/null | tail -1 > $out/bench_single_$rep.json
  for k in pair single; do python - <<PY
import json
j=json.load(open("$out/bench_${k}_$rep.json")); print("%-7s rep$rep %8.1f steps/s %.4f ms %s sync %.4f loss %.5f" % ("$k", j["value"], j["ms_per_step"], j["kernel_ms"], j["trainer_step_sync_ms"], j["final_total_loss"]))
PY
  done
done
