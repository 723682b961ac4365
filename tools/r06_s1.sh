cd $GRAFT_REPO_ROOT; O=gpurun_out/r06_preload; mkdir -p $O
ISDF_HIP_LIB=$PWD/variants/lib_prev.so python tools/train_ab_check.py --dump /tmp/a.npz > $O/ab_check.txt 2>&1
ISDF_HIP_LIB=$PWD/variants/lib_pw1.so python tools/train_ab_check.py --dump /tmp/b.npz >> $O/ab_check.txt 2>&1
python tools/train_ab_check.py --compare /tmp/a.npz /tmp/b.npz 2>&1 | grep -c "bit-identical"; python tools/train_ab_check.py --compare /tmp/a.npz /tmp/b.npz 2>&1 | grep -v "bit-identical" | head -5
for rep in 1 2 3; do for f in variants/lib_*.so; do
  ISDF_HIP_LIB=$PWD/$f python bench.py --steps 200 --warmup 30 --no-cpu-baseline --no-accuracy 2>/dev/null | tail -1 | python -c "
import sys,json
j=json.loads(sys.stdin.read()); print('%-10s rep$rep  sync %.4f ms  pipelined %.4f ms  chain %.4f dw %.4f tail %.4f' % ('$f'.split('lib_')[1][:-3], j['ms_per_step'], j['pipelined']['ms_per_step'], *list(j['kernel_ms'].values())[:3]))"
done; done > $O/ab.txt 2>&1; cat $O/ab.txt
