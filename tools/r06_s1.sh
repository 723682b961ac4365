cd $GRAFT_REPO_ROOT; O=gpurun_out/r06_f8d; mkdir -p $O
python -m pytest tests -m gpu -q --tb=short 2>&1 | grep -v "^  File\|^    \|amdgpu.ids" | tail -60 > $O/tests.log
grep "^E  \|^tests/\|^___\|passed\|failed" $O/tests.log | cut -c1-300
for rep in 1 2; do for m in 16bit e4m3_gb e4m3; do
python bench.py --steps 300 --warmup 50 --no-cpu-baseline --spill-operand $m 2>/dev/null | tail -1 | python -c "
import sys,json
j=json.loads(sys.stdin.read()); fm=j.get('fast_mode_fp16') or {}
print('%-10s rep$rep  %8.1f steps/s  %.4f ms  chain %.4f dw %.4f tail %.4f  sync %.4f | loss %.5f' % ('$m', j['value'], j['ms_per_step'], *list(j['kernel_ms'].values())[:3], j['trainer_step_sync_ms'], j['final_total_loss']))"
done; done > $O/modes.txt 2>&1
cat $O/modes.txt
