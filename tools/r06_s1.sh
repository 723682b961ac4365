cd $GRAFT_REPO_ROOT; O=gpurun_out/r06_hilo; mkdir -p $O
for tag in before after; do
  if [ $tag = before ]; then export ISDF_HIP_LIB=$PWD/variants/lib_before_hilo.so; else unset ISDF_HIP_LIB; fi
  echo "== $tag put_x_hilo fix"
  python -m pytest tests/test_gpu_parity.py -m gpu -q -s -k "base_size_forward_and_input_gradient and fp16x2 and not full" 2>&1 | grep "sdf rel-L2\|passed\|failed"
done > $O/hilo.txt 2>&1
cat $O/hilo.txt
unset ISDF_HIP_LIB
python -m pytest tests -m gpu -q --tb=short 2>&1 | grep "^E  \|^tests/\|^___\|passed\|failed" | cut -c1-300
