#!/bin/bash
# round 3, closing GPU call on the final tree: the bench line as the first command of a fresh box, full parity suite, smoke, kernel stats
O=gpurun_out/r03z; mkdir -p $O
python bench.py 2>/dev/null | tail -1 > $O/bench.json
python -m pytest tests -m gpu -x -q 2>&1 | tail -5 > $O/gpu_tests.log; tail -3 $O/gpu_tests.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
python bench.py --infer-points 8000000 2>/dev/null | tail -1 > $O/bench_inference_8M.json
python bench.py --gpus 1 --steps 20 --warmup 5 2>/dev/null | tail -1 > $O/bench_driver_args.json
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/stats -- python $GRAFT_REPO_ROOT/bench.py --steps 100 --warmup 10 --no-cpu-baseline > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
python - <<'PY'
import json,glob,csv
for f in sorted(glob.glob('gpurun_out/r03z/*.json')):
    j=json.load(open(f)); print(f, j['value'], j.get('ms_per_step'), j.get('kernel_ms'), j.get('synchronised_step'), (j.get('roofline') or {}).get('frac'), j.get('fast_mode_fp16'), j.get('modes'), (j.get('cpu_baseline') or {}).get('value'), (j.get('gpu_eager_baseline') or {}).get('value'))
for f in glob.glob('gpurun_out/r03z/stats/**/*kernel_stats.csv', recursive=True):
    for r in list(csv.DictReader(open(f)))[:5]: print('  ', r['Name'][:80], r['Calls'], r['AverageNs'])
PY
