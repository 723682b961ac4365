# Round 6's scratch gpurun session (rewritten per experiment during the round; left in its most-used form):
#   every variants/lib_*.so -- built by tools/build_variants.py / tools/build_rev_lib.sh -- through the same bench, alternating, on ONE box.
#   /usr/local/graft/bin/gpurun --timeout 900 -- 'timeout 850 bash tools/r06_s1.sh'
cd $GRAFT_REPO_ROOT; O=gpurun_out/r06_ab; mkdir -p $O
for rep in 1 2 3; do for f in variants/lib_*.so; do
  ISDF_HIP_LIB=$PWD/$f python bench.py --steps 200 --warmup 30 --no-cpu-baseline --no-accuracy 2>/dev/null | tail -1 | python -c "
import sys,json
j=json.loads(sys.stdin.read()); print('%-10s rep$rep  sync %.4f ms  pipelined %.4f ms  chain %.4f dw %.4f tail %.4f' % ('$f'.split('lib_')[1][:-3], j['ms_per_step'], j['pipelined']['ms_per_step'], *list(j['kernel_ms'].values())[:3]))"
done; done > $O/ab.txt 2>&1; cat $O/ab.txt
