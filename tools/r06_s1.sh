cd $GRAFT_REPO_ROOT; O=gpurun_out/r06_dp; mkdir -p $O
timeout 600 python -m pytest tests/test_dp_rccl_direct_gpu.py tests/test_dp_gpu.py tests/test_bench_launch.py -x -q -m gpu 2>&1 | tail -3
ISDF_BENCH_FORCE_DP=1 python bench.py --steps 100 --warmup 30 --no-cpu-baseline --no-accuracy 2>/dev/null | tail -1 | python -c "
import sys,json
j=json.loads(sys.stdin.read()); print('forced dp  sync %.4f ms  pipelined %.4f ms  collective %s' % (j['ms_per_step'], j['pipelined']['ms_per_step'], j['distributed'].get('collective')))"
