#!/bin/bash
# round 3, seventh GPU call: full bench line as the FIRST command of a fresh box (does the synchronised-step figure hold?),
# sampler with 16-byte pc stores: parity tests + 1e6-ray bench + kernel stats at both sizes
O=gpurun_out/r03g; mkdir -p $O
python bench.py 2>/dev/null | tail -1 > $O/bench_first.json
python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "sampler or sample or base_size_sampler or shapes or step_contract" 2>&1 | tail -4
python bench.py --sampler-scale 200000 --steps 300 2>/dev/null | tail -1 > $O/sampler_1M.json
python bench.py --sampler-scale 2000000 --steps 100 2>/dev/null | tail -1 > $O/sampler_10M.json
python bench.py --steps 300 --warmup 30 --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_second.json
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r03g/*.json')):
    j=json.load(open(f)); print(f, j['value'], j.get('ms_per_step'), j.get('kernel_ms'), j.get('synchronised_step'), (j.get('roofline') or {}).get('frac'))
PY
