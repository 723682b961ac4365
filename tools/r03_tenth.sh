#!/bin/bash
# round 3, tenth GPU call: final tree -- full parity suite (incl. the empty-batch edge case), 2-rank functional run of the
# data-parallel bench path (gloo, one GPU), forced-DP line
O=gpurun_out/r03j; mkdir -p $O
python -m pytest tests -m gpu -x -q 2>&1 | tail -4
ISDF_BENCH_BACKEND=gloo timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29613 bench.py --gpus 2 --steps 100 --warmup 10 --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_dp2_gloo_functional.json
ISDF_BENCH_FORCE_DP=1 python bench.py --steps 300 --warmup 30 --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_forced_dp_world1.json
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r03j/*.json')):
    try: j=json.load(open(f))
    except Exception as e: print(f,'ERR',e, open(f).read()[:300]); continue
    print(f, j['value'], j.get('n_gpus'), j.get('ms_per_step'), j.get('kernel_ms'), j.get('synchronised_step'), j['config']['parallelism'])
PY
