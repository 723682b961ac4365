#!/bin/bash
# round 3: one-off probes
O=gpurun_out/r03r; mkdir -p $O
python bench.py --no-cpu-baseline > $O/bench.json 2> $O/bench.err; python - <<'PY'
import json
j=json.loads(open('gpurun_out/r03r/bench.json').read().strip().splitlines()[-1])
print(j['value'], j['ms_per_step'], j['kernel_ms'], j['synchronised_step']['ms_per_step'], j['roofline']['frac'], j.get('fast_mode_fp16'))
print(j['chain_us_per_step'])
PY
