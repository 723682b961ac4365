#!/bin/bash
# A/B harness: run bench.py with every variants/lib_*.so (tools/build_variants.py) on this box, back to back, twice (order
# effects), and print steps/s + per-kernel times of the default operand mode and of the fp16 fast mode.
# usage: tools/ab_bench.sh [extra bench args]
for rep in 1 2; do
for f in variants/lib_*.so; do
  ISDF_HIP_LIB=$PWD/$f python bench.py --steps 300 --warmup 50 --no-cpu-baseline "$@" 2>/dev/null | tail -1 | python -c "
import sys,json
j=json.loads(sys.stdin.read()); fm=j.get('fast_mode_fp16') or {}
print('%-10s rep$rep  %8.1f steps/s  %.4f ms  chain %.4f dw %.4f tail %.4f  sync %.4f | fp16: %8.1f steps/s chain %.4f | loss %.5f' % ('$f'.split('lib_')[1][:-3], j['value'], j['ms_per_step'], *list(j['kernel_ms'].values())[:3], j['trainer_step_sync_ms'], fm.get('steps_per_s', 0), fm.get('chain_ms', 0), j['final_total_loss']))"
done
done
