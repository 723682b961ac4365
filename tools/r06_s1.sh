cd $GRAFT_REPO_ROOT; O=gpurun_out/r06_acc; mkdir -p $O
A="--paired-draws --keyframes 24 --steps-per-kf 100"
timeout 1700 python tests/accuracy_experiment.py $A --backend hip --seeds $(seq 21 120) --out $O/paired_hip_21_120.json > $O/paired_hip_21_120.log 2>&1; echo rc=$?
tail -n 2 $O/paired_hip_21_120.log
python tools/accuracy_stats.py profiles/r05_accuracy_paired_draws_control_fp32_gpu_seeds21_120.json $O/paired_hip_21_120.json > $O/stats_vs_control.json 2>&1; head -20 $O/stats_vs_control.json
python tools/accuracy_stats.py profiles/r05_accuracy_paired_draws_hip_seeds21_120.json $O/paired_hip_21_120.json > $O/stats_vs_r05_hip.json 2>&1; head -20 $O/stats_vs_r05_hip.json
