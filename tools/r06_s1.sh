cd $GRAFT_REPO_ROOT; O=gpurun_out/r06_host; mkdir -p $O
python -m pytest tests -m gpu -q --tb=short 2>&1 | grep -v "^  File\|^    \|amdgpu.ids" | tail -40 > $O/tests.log
grep "^E  \|^tests/\|^___\|passed\|failed" $O/tests.log | cut -c1-300
python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-accuracy 2>/dev/null | tail -1 > $O/bench_driver.json
python -c "
import json; j=json.load(open('$O/bench_driver.json')); print(j['value'], j['ms_per_step'], j['kernel_ms'], j['pipelined']['ms_per_step'], j['synchronised_step']['ms_per_step'], j['synchronised_step']['timed_region'])"
