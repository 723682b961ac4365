cd $GRAFT_REPO_ROOT; O=gpurun_out/r06_dp; mkdir -p $O
timeout 600 python -m pytest tests/test_dp_rccl_direct_gpu.py tests/test_dp_gpu.py tests/test_bench_launch.py -x -q -m gpu 2>&1 | tail -15 > $O/tests.txt; cat $O/tests.txt
for rep in 1 2; do
for mode in none direct torch; do
  if [ $mode = none ]; then export ISDF_BENCH_FORCE_DP=0; else export ISDF_BENCH_FORCE_DP=1; fi
  if [ $mode = torch ]; then export ISDF_DP_COLLECTIVE=torch; else export ISDF_DP_COLLECTIVE=; fi
  python bench.py --steps 200 --warmup 30 --no-cpu-baseline --no-accuracy 2>/dev/null | tail -1 | python -c "
import sys,json
j=json.loads(sys.stdin.read()); print('%-8s rep$rep  sync %.4f ms  pipelined %.4f ms  collective %s' % ('$mode', j['ms_per_step'], j['pipelined']['ms_per_step'], j['distributed'].get('collective')))"
done; done > $O/ab.txt 2>&1; cat $O/ab.txt
