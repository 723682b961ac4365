cd $GRAFT_REPO_ROOT; O=gpurun_out/r06_cachepol; mkdir -p $O
run() { name=$1; shift; for f in variants/lib_prev.so variants/lib_new.so; do
  ISDF_HIP_LIB=$PWD/$f python bench.py "$@" --no-cpu-baseline --no-accuracy 2>/dev/null | tail -1 | python -c "
import sys,json
j=json.loads(sys.stdin.read()); print('%-12s %-6s sync %.4f ms  pipelined %.4f ms  chain %.4f dw %.4f tail %.4f' % ('$name', '$f'.split('lib_')[1][:-3], j['ms_per_step'], j['pipelined']['ms_per_step'], *list(j['kernel_ms'].values())[:3]))"
done; }
{ run default --steps 200 --warmup 30
  run wide --wide --steps 30 --warmup 5
} > $O/other_workloads2.txt 2>&1; cat $O/other_workloads2.txt
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu 2>&1 | tail -2
