#!/bin/bash
# round 3, eighth GPU call: accuracy runs with the default (fp16x2) operands, 2-rank functional run of the data-parallel bench
# path over gloo on one GPU, refreshed bench lines after the timing-event fix
O=gpurun_out/r03h; mkdir -p $O
python bench.py 2>/dev/null | tail -1 > $O/bench.json
python bench.py --gpus 1 --steps 20 --warmup 5 2>/dev/null | tail -1 > $O/bench_driver_args.json
ISDF_BENCH_FORCE_DP=1 python bench.py --steps 300 --warmup 30 --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_forced_dp_world1.json
ISDF_BENCH_BACKEND=gloo python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29611 bench.py --gpus 2 --steps 100 --warmup 10 --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_dp2_gloo_functional.json
python tests/accuracy_experiment.py --backend hip --seeds 1 2 3 4 5 --keyframes 24 --steps-per-kf 100 --out $O/accuracy_hip_24kf_x100.json > $O/acc1.log 2>&1
python tests/accuracy_experiment.py --backend hip --seeds 1 2 3 4 5 --reference-schedule --steps 1000 --out $O/accuracy_hip_reference_schedule_1000steps.json > $O/acc2.log 2>&1
python bench.py --sampler-scale 200000 --steps 300 2>/dev/null | tail -1 > $O/bench_sampler_1M.json
python bench.py --sampler-scale 2000000 --steps 100 2>/dev/null | tail -1 > $O/bench_sampler_10M.json
cd /tmp && export TMPDIR=/tmp
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/stats_sampler -- python $GRAFT_REPO_ROOT/bench.py --sampler-scale 200000 --steps 100 > /dev/null 2>&1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/stats -- python $GRAFT_REPO_ROOT/bench.py --steps 100 --warmup 10 --no-cpu-baseline > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
python - <<'PY'
import json,glob,csv
for f in sorted(glob.glob('gpurun_out/r03h/*.json')):
    try: j=json.load(open(f))
    except Exception as e: print(f,'ERR',e); continue
    if 'value' in j: print(f, j['value'], j.get('ms_per_step'), j.get('kernel_ms'), j.get('synchronised_step'), (j.get('roofline') or {}).get('frac'))
    elif 'runs' in j:
        import numpy as np
        v=[r['l1_visible_m'] for r in j['runs']]; s=[r['l1_surface_m'] for r in j['runs']]
        print(f, 'visible L1 %.4f +- %.4f surface %.4f'%(np.mean(v),np.std(v),np.mean(s)))
for f in glob.glob('gpurun_out/r03h/stats*/**/*kernel_stats.csv', recursive=True):
    print(f)
    for r in list(csv.DictReader(open(f)))[:5]: print('  ', r['Name'][:80], r['Calls'], r['AverageNs'])
PY
tail -3 $O/acc1.log $O/acc2.log
