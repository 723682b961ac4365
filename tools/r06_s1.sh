cd $GRAFT_REPO_ROOT; O=gpurun_out/r06_cachepol; mkdir -p $O
run() { name=$1; shift; python bench.py "$@" --no-cpu-baseline --no-accuracy 2>/dev/null | tail -1 | python -c "
import sys,json
j=json.loads(sys.stdin.read()); print('%-12s sync %.4f ms  pipelined %.4f ms  chain %.4f dw %.4f tail %.4f' % ('$name', j['ms_per_step'], j['pipelined']['ms_per_step'], *list(j['kernel_ms'].values())[:3]))"; }
{ run default --steps 200 --warmup 30
  run spill16 --spill-operand 16bit --steps 200 --warmup 30
  run bf16 --fwd-operand bf16 --steps 200 --warmup 30
  run 729k --rays-per-frame 5400 --steps 30 --warmup 5
  run rays400 --rays-per-frame 400 --steps 100 --warmup 10
  run rays1000 --rays-per-frame 1000 --steps 50 --warmup 10
  run wide --wide --steps 30 --warmup 5
} > $O/other_workloads3.txt 2>&1; cat $O/other_workloads3.txt
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu 2>&1 | tail -1
