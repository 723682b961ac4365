#!/bin/bash
# round 3, second GPU call: parity suite (new reference fixtures, Philox mapping, one-collective DP), ingest fast-math A/B,
# sampler at both sizes, forced-DP step, full bench line
O=gpurun_out/r03b; mkdir -p $O
python -m pytest tests -m gpu -x -q -s 2>&1 | tail -80 > $O/gpu_tests.log
tail -3 $O/gpu_tests.log
# ingest: reference-fixture parity + time, correctly rounded vs hardware-rate sqrt / rcp
for v in base nfast; do
  ISDF_HIP_LIB=$PWD/variants/lib_$v.so python -m pytest tests/test_gpu_parity.py -m gpu -q -s -k "ingest_normals" 2>&1 | tail -4 > $O/ingest_test_$v.log
  ISDF_HIP_LIB=$PWD/variants/lib_$v.so python bench.py --ingest --steps 300 2>/dev/null | tail -1 > $O/ingest_$v.json
done
python bench.py --sampler-scale 200000 --steps 300 2>/dev/null | tail -1 > $O/sampler_1M.json
ISDF_BENCH_FORCE_DP=1 python bench.py --steps 300 --warmup 30 --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_forced_dp.json
python bench.py --steps 300 --warmup 30 2>/dev/null | tail -1 > $O/bench.json
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/prof -o r03b -- python $GRAFT_REPO_ROOT/bench.py --steps 300 --warmup 30 --no-cpu-baseline > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
python - <<'PY'
import json,glob,csv
for f in sorted(glob.glob('gpurun_out/r03b/*.json')):
    try: j=json.load(open(f))
    except Exception as e: print(f, 'ERR', e); continue
    print(f, j['value'], j.get('ms_per_step'), j.get('kernel_ms'), j.get('synchronised_step',{}).get('ms_per_step'), j.get('roofline',{}).get('frac'), j.get('fast_mode_fp16'), j.get('gpu_eager_baseline'), (j.get('cpu_baseline') or {}).get('value'))
for f in glob.glob('gpurun_out/r03b/prof/**/*kernel_stats.csv', recursive=True):
    for r in list(csv.DictReader(open(f)))[:8]: print(r['Name'][:70], r['Calls'], r['AverageNs'])
PY
for v in base nfast; do echo $v; cat $O/ingest_test_$v.log; done
