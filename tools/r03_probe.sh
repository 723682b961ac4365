#!/bin/bash
# round 3: functional check of bench.py's N > 1 path after the timed-loop changes (two ranks share the one GPU over gloo; not a number)
O=gpurun_out/r03y; mkdir -p $O
ISDF_BENCH_BACKEND=gloo timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29617 bench.py --gpus 2 --steps 100 --warmup 10 --no-cpu-baseline 2>$O/dp2.err | tail -1 > $O/bench_dp2_gloo_functional.json
tail -3 $O/dp2.err; python -c "
import json; j=json.load(open('$O/bench_dp2_gloo_functional.json')); print(j['value'], j['n_gpus'], j['ms_per_step'], j['kernel_ms'], j['synchronised_step']['ms_per_step'], j['config']['parallelism'])"
