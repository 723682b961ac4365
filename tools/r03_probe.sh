#!/bin/bash
# round 3: accuracy control per forward operand mode (final kernels, pinned schedule 24 keyframes x 100 steps, seeds 1..5)
O=gpurun_out/r03acc; mkdir -p $O
for m in fp16 fp16x2_full bf16; do
  python tests/accuracy_experiment.py --backend hip --seeds 1 2 3 4 5 --keyframes 24 --steps-per-kf 100 --fwd-operand $m --out $O/mode_$m.json > $O/mode_$m.log 2>&1
  echo $m $(tail -1 $O/mode_$m.log)
done
