#!/bin/bash
# Copy the judged outputs of tools/round_records.sh (gpurun_out/r<NN>final/, scratch) to profiles/r<NN>_* (tracked).
#   tools/collect_records.sh 06
set -e
NN=${1:?round number}; O=gpurun_out/r${NN}final; P=profiles
cp $O/bench.json $P/r${NN}_bench.json
cp $O/bench_729k.json $P/r${NN}_bench_729k_points.json
cp $O/bench_wide.json $P/r${NN}_bench_wide_c5.json
for n in driver_args driver_args_noramp forced_dp_world1 inference_8M ingest sampler_1M stream_480x640; do
  cp $O/bench_$n.json $P/r${NN}_bench_$n.json
done
cp $O/hbm_traffic.json $P/r${NN}_hbm_traffic.json
cp $O/pmc_summary.csv $P/r${NN}_pmc_bench.csv
for s in "" _infer _ingest _sampler; do
  f=$(ls -t $(find $O/stats$s -name '*kernel_stats.csv') 2>/dev/null | head -1)      # (a re-run leaves the earlier run's files beside the new ones)
  case "$s" in "") d=$P/r${NN}_kernel_stats.csv;; _infer) d=$P/r${NN}_kernel_stats_inference_8M.csv;; _ingest) d=$P/r${NN}_kernel_stats_ingest.csv;; _sampler) d=$P/r${NN}_kernel_stats_sampler_1M.csv;; esac
  [ -n "$f" ] && cp "$f" "$d"
done
ls -la $P/r${NN}_bench*.json $P/r${NN}_kernel_stats*.csv $P/r${NN}_hbm_traffic.json $P/r${NN}_pmc_bench.csv
