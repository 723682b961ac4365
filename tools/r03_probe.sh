#!/bin/bash
# round 3: accuracy control on the final kernels
O=gpurun_out/r03acc; mkdir -p $O
python tests/accuracy_experiment.py --backend hip --seeds 1 2 3 4 5 --keyframes 24 --steps-per-kf 100 --out $O/final_24kf_x100.json > $O/final1.log 2>&1; tail -1 $O/final1.log
python tests/accuracy_experiment.py --backend hip --seeds 1 2 3 4 5 --reference-schedule --steps 1000 --out $O/final_refsched.json > $O/final2.log 2>&1; python -c "
import json,numpy as np; j=json.load(open('$O/final_refsched.json')); v=[r['l1_visible_m'] for r in j['runs']]; s=[r['l1_surface_m'] for r in j['runs']]; print('refsched', np.mean(v), np.std(v, ddof=1), np.mean(s))"
