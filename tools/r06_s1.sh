cd $GRAFT_REPO_ROOT; O=gpurun_out/r06_f8g; mkdir -p $O
python -m pytest tests -m gpu -q --tb=short 2>&1 | grep -v "^  File\|^    \|amdgpu.ids" | tail -30 > $O/tests.log
grep "^E  \|^tests/\|^___\|passed\|failed" $O/tests.log | cut -c1-300
python bench.py --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench.json
python -c "
import json; j=json.load(open('$O/bench.json')); print(j['value'], j['ms_per_step'], j['kernel_ms'], j['synchronised_step']['ms_per_step'], j['roofline']['frac'])"
