#!/bin/bash
# round 3: streaming sampler with two workgroups per CU (LDS 81 996 -> 77 900 B per block)
O=gpurun_out/r03w; mkdir -p $O
python -m pytest tests -m gpu -x -q -k "sampl or philox or compaction or million" 2>&1 | tail -3
python bench.py --sampler-scale 200000 --steps 300 2>/dev/null | tail -1 > $O/bench_sampler_1M.json
python bench.py --sampler-scale 2000000 --steps 50 2>/dev/null | tail -1 > $O/bench_sampler_10M.json
python - <<'PY'
import json
for n in ("1M","10M"):
    j=json.load(open("gpurun_out/r03w/bench_sampler_%s.json"%n)); print(n, j["value"], j["ms_per_step"], j["roofline"])
PY
