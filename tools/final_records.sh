#!/bin/bash
# End-of-round records (run on the GPU box through gpurun): bench line, rocprofv3 kernel stats, HBM PMC passes,
# phase timeline, throughput-scaling point.  Outputs under gpurun_out/final/.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/final; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
python $R/bench.py 2>&1 | tail -1 > $O/bench.json
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python $R/bench.py --steps 100 --warmup 10 --no-cpu-baseline > $O/stats.log 2>&1
for set in "FETCH_SIZE" "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum"; do
  n=$(echo $set | cut -d' ' -f1)
  timeout 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $O/pmc_$n -- python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/pmc_$n.log 2>&1
done
python $R/tools/timeline.py > $O/timeline.txt 2>&1
python $R/bench.py --rays-per-frame 5400 --steps 30 --warmup 5 --no-cpu-baseline 2>&1 | tail -1 > $O/bench_729k.json
ls $O
