cd $GRAFT_REPO_ROOT; O=gpurun_out/r06_cachepol; mkdir -p $O
for a in 0 32 512; do
  ISDF_DEBUG_ALIAS_SPILL=$a ISDF_HIP_LIB=$PWD/variants/lib_dbg.so python bench.py --steps 100 --warmup 30 --no-cpu-baseline --no-accuracy 2>/dev/null | tail -1 | python -c "
import sys,json
j=json.loads(sys.stdin.read()); print('alias %-4s  sync %.4f ms  pipelined %.4f ms  chain %.4f dw %.4f tail %.4f' % ('$a', j['ms_per_step'], j['pipelined']['ms_per_step'], *list(j['kernel_ms'].values())[:3]))"
done > $O/whatif_alias.txt 2>&1; cat $O/whatif_alias.txt
