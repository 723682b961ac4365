cd $GRAFT_REPO_ROOT; O=gpurun_out/r06_host; mkdir -p $O
python tools/host_timeline.py 2>&1 | grep -v amdgpu.ids > $O/host_timeline2.txt; cat $O/host_timeline2.txt
timeout 900 python -m pytest tests -x -q -m gpu -k "not parity" 2>&1 | tail -2
for rep in 1 2; do python bench.py --steps 200 --warmup 30 --no-cpu-baseline --no-accuracy 2>/dev/null | tail -1 | python -c "
import sys,json
j=json.loads(sys.stdin.read()); print('rep$rep  sync %.4f ms  pipelined %.4f ms  chain %.4f dw %.4f tail %.4f' % (j['ms_per_step'], j['pipelined']['ms_per_step'], *list(j['kernel_ms'].values())[:3]))"; done
