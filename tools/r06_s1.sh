cd $GRAFT_REPO_ROOT; O=gpurun_out/r06_bench; mkdir -p $O
( time python bench.py --gpus 1 --steps 20 --warmup 5 2>$O/bench_driver.err | tail -1 > $O/bench_driver.json ) 2> $O/time.txt
tail -3 $O/time.txt; tail -5 $O/bench_driver.err
python -c "
import json; j=json.load(open('$O/bench_driver.json')); print(j['value'], j['ms_per_step'], j['kernel_ms'], j['pipelined'], j['synchronised_step']['ms_per_step'], j['roofline']['frac'], j['roofline']['bound']); print(j.get('sdf_l1_vs_gt')); print(j['cpu_baseline']['value'], j.get('gpu_eager_baseline'))"
python -m pytest tests/test_bench_launch.py -m gpu -q 2>&1 | tail -3
