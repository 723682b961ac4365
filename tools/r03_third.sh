#!/bin/bash
# round 3, third GPU call: parity suite on the mailbox / plan-cache host path, bench (fused + forced data-parallel),
# fine-grained phase timeline of the chain kernel (development build) for both fp16 modes
O=gpurun_out/r03c; mkdir -p $O
python -m pytest tests -m gpu -x -q -s 2>&1 | tail -80 > $O/gpu_tests.log
tail -3 $O/gpu_tests.log
python bench.py --steps 300 --warmup 30 --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench.json
ISDF_BENCH_FORCE_DP=1 python bench.py --steps 300 --warmup 30 --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_forced_dp.json
for op in fp16 fp16x2; do
  ISDF_FWD_OPERAND=$op python tools/timeline.py > $O/timeline_$op.txt 2>&1
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r03c/*.json')):
    try: j=json.load(open(f))
    except Exception as e: print(f, 'ERR', e); continue
    print(f, j['value'], j.get('ms_per_step'), j.get('kernel_ms'), 'sync', j.get('synchronised_step',{}).get('ms_per_step'), j.get('roofline',{}).get('frac'), (j.get('fast_mode_fp16') or {}).get('steps_per_s'))
PY
grep -E "SUMMARY|n stamps" $O/timeline_*.txt
