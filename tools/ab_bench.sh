#!/bin/bash
# A/B harness: run bench.py with every variants/lib_*.so on this box, back to back, twice (order effects), and print
# steps/s + per-kernel times.  usage: tools/ab_bench.sh [extra bench args]
cp isdf_amd/libisdf_hip.so /tmp/lib_keep.so
for rep in 1 2; do
for f in variants/lib_*.so; do
  cp $f isdf_amd/libisdf_hip.so
  python bench.py --steps 300 --warmup 50 --no-cpu-baseline "$@" 2>/dev/null | tail -1 | python -c "
import sys,json
j=json.loads(sys.stdin.read()); print('%-28s rep$rep  %8.1f steps/s  %.4f ms  %s  sync %.4f loss %.5f' % ('$f'.split('lib_')[1][:-3], j['value'], j['ms_per_step'], j['kernel_ms'], j['trainer_step_sync_ms'], j['final_total_loss']))"

done
done
cp /tmp/lib_keep.so isdf_amd/libisdf_hip.so
