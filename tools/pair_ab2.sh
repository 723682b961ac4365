#!/bin/bash
# A/B on one box: default lib (pair + single via env) and variants/lib_pe0.so (old PE mapping), bench only
for rep in 1 2; do
for cfg in "pair:ISDF_CHAIN_PAIR=1" "single:ISDF_X=0" "pair_pe0:ISDF_HIP_LIB=variants/lib_pe0.so ISDF_CHAIN_PAIR=1" "single_pe0:ISDF_HIP_LIB=variants/lib_pe0.so"; do
  n=${cfg%%:*}; e=${cfg#*:}
  env $e python bench.py --steps 300 --warmup 50 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import sys,json
j=json.loads(sys.stdin.read()); print('%-12s rep$rep %8.1f steps/s %.4f ms %s sync %.4f loss %.5f' % ('$n', j['value'], j['ms_per_step'], j['kernel_ms'], j['trainer_step_sync_ms'], j['final_total_loss']))"
done; done
