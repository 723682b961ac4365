#!/bin/bash
# round 3, sixth GPU call: balanced tiles -- parity suite, then same-box A/B against the previous library
O=gpurun_out/r03f; mkdir -p $O
python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > $O/gpu_tests.log
tail -4 $O/gpu_tests.log
bash tools/ab_bench.sh > $O/ab.txt 2>&1
cat $O/ab.txt
python tools/build_variants.py dbg="-DISDF_DEBUG_HOOKS=1" > /dev/null 2>&1
for op in fp16 fp16x2; do ISDF_FWD_OPERAND=$op TIMELINE_BRIEF=1 python tools/timeline.py 2>&1 | grep -E "SUMMARY|workgroups sampled"; done
