#!/bin/bash
# Round-4 GPU sessions (one gpurun call each):  tools/r04_session.sh <stage>
#   ab1        GPU tests (new trained-state tests print their measures), same-box A/B of variants/lib_*.so, step-time
#              distribution of a native-clock run, 30 more seeds of the accuracy control
#   ab2        full GPU suite on the ABI-5 tree, the fixed bias-LDS variant (parity subset + same-box A/B), PAIRED accuracy runs
#              (identical random streams and initial network, HIP vs fp32 eager), data-parallel bench modes on one GPU
#   final      full GPU suite, end-of-round records (tools/round_records.sh 04), native-clock run of the final host path,
#              accuracy by forward operand mode (40 seeds each)
#   bwd16      fp16 second-order sweeps / dW operands (bwd_operand): GPU suite (both types parametrised), same-box bench A/B,
#              accuracy control 40 seeds each, window-transport A/B inside the bench line
#   accuracy   fp32 eager-GPU control vs the HIP path (HEAD and the reverted d490710 variant) on 10 shared seeds, the
#              trained-weights gradient-bias probe for both libraries, reference-driver schedule and native-clock runs
#   hidden     zero-padded narrow nets (the 64-wide reference fixtures on the GPU), fwd_race_probe on the in-tree library, full GPU
#              suite, a bench line
#   race       tests/fwd_race_probe.py over variants/lib_*.so (source-patch and ISA-edit variants: the v_pk_fma_f32 op_sel hunt)
#   pinned     tools/probes/pk_fma_opsel (the instruction form in isolation), frame_avg_losses in pinned host memory: test + bench A/B
# Outputs land in gpurun_out/r04*/ (scratch); what is judged is copied to profiles/ by hand.
set -u
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
O=gpurun_out/r04
mkdir -p $O
SEEDS10="1 2 3 4 5 6 7 8 9 10"
SEEDS5="1 2 3 4 5"
VAR=variants/lib_pl_from_a.so
stage=${1:-accuracy}
t0=$(date +%s)
lap() { echo "[$(( $(date +%s) - t0 )) s] $*"; }

if [ "$stage" = accuracy ]; then
  python tests/accuracy_experiment.py --backend port --device cuda --seeds $SEEDS10 --keyframes 24 --steps-per-kf 100 \
      --out $O/acc_control_fp32_gpu_24x100.json > $O/acc_control_fp32_gpu_24x100.log 2>&1; lap control 24x100
  python tests/accuracy_experiment.py --backend hip --seeds $SEEDS10 --keyframes 24 --steps-per-kf 100 \
      --out $O/acc_hip_24x100.json > $O/acc_hip_24x100.log 2>&1; lap hip 24x100
  ISDF_HIP_LIB=$VAR python tests/accuracy_experiment.py --backend hip --seeds $SEEDS10 --keyframes 24 --steps-per-kf 100 \
      --out $O/acc_hip_pl_from_a_24x100.json > $O/acc_hip_pl_from_a_24x100.log 2>&1; lap variant 24x100
  python tests/grad_bias_probe.py --make-weights /tmp/probe_w.pt > $O/probe_make.log 2>&1; lap probe weights
  python tests/grad_bias_probe.py --weights /tmp/probe_w.pt --out $O/probe_head.json > $O/probe_head.log 2>&1; lap probe head
  ISDF_HIP_LIB=$VAR python tests/grad_bias_probe.py --weights /tmp/probe_w.pt --out $O/probe_pl_from_a.json > $O/probe_pl_from_a.log 2>&1; lap probe variant
  python tests/accuracy_experiment.py --native-clock --backend hip --seeds $SEEDS5 --out $O/native_clock_hip.json > $O/native_clock_hip.log 2>&1; lap native hip
  python tests/accuracy_experiment.py --native-clock --backend port --device cuda --seeds $SEEDS5 --out $O/native_clock_port_cuda.json > $O/native_clock_port_cuda.log 2>&1; lap native port
  python tests/accuracy_experiment.py --native-clock --backend hip --seeds $SEEDS5 --max-steps 20000 --out $O/native_clock_hip_cap20000.json > $O/native_clock_hip_cap20000.log 2>&1; lap native hip capped
  python tests/accuracy_experiment.py --reference-schedule --steps 1000 --backend hip --seeds $SEEDS10 --out $O/refsched_hip.json > $O/refsched_hip.log 2>&1; lap refsched hip
  python tests/accuracy_experiment.py --reference-schedule --steps 1000 --backend port --device cuda --seeds $SEEDS10 --out $O/refsched_port_cuda.json > $O/refsched_port_cuda.log 2>&1; lap refsched port
  tail -n 3 $O/*.log
fi

if [ "$stage" = ab1 ]; then
  python -m pytest tests/test_gpu_parity.py -q -s -m gpu -k "trained" > $O/pytest_trained.log 2>&1; lap pytest trained
  python -m pytest tests -q -m gpu > $O/pytest_gpu.log 2>&1; lap pytest gpu
  for rep in 1 2; do for f in variants/lib_*.so; do
    ISDF_HIP_LIB=$PWD/$f python bench.py --steps 300 --warmup 50 --no-cpu-baseline > $O/ab1_$(basename $f .so)_$rep.json 2> /dev/null; lap bench $f $rep
  done; done
  python tests/accuracy_experiment.py --native-clock --backend hip --seeds 1 2 --out $O/native_clock_hip_diag.json > $O/native_clock_hip_diag.log 2>&1; lap native diag
  python tests/accuracy_experiment.py --backend hip --seeds $(seq 11 40) --keyframes 24 --steps-per-kf 100 \
      --out $O/acc_hip_24x100_seeds11_40.json > $O/acc_hip_24x100_seeds11_40.log 2>&1; lap hip 30 seeds
  python tests/accuracy_experiment.py --backend port --device cuda --seeds $(seq 11 20) --keyframes 24 --steps-per-kf 100 \
      --out $O/acc_control_fp32_gpu_24x100_seeds11_20.json > $O/acc_control_seeds11_20.log 2>&1; lap control 10 seeds
  tail -n 25 $O/pytest_trained.log; tail -n 8 $O/pytest_gpu.log
  for f in $O/ab1_*.json; do python - "$f" <<'PY'
import json, sys
j = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); fm = j.get("fast_mode_fp16") or {}
print("%-28s %8.1f steps/s %.4f ms chain %.4f dw %.4f tail %.4f sync %.4f | fp16 %8.1f chain %.4f | loss %.5f" % (
    sys.argv[1].split("ab1_")[1][:-5], j["value"], j["ms_per_step"], *list(j["kernel_ms"].values())[:3], j["trainer_step_sync_ms"],
    fm.get("steps_per_s", 0), fm.get("chain_ms", 0), j["final_total_loss"]))
PY
  done
  tail -n 3 $O/native_clock_hip_diag.log $O/acc_hip_24x100_seeds11_40.log $O/acc_control_seeds11_20.log
fi

if [ "$stage" = ab2 ]; then
  python -m pytest tests -q -m gpu -s > $O/pytest_gpu2.log 2>&1; lap pytest gpu
  ISDF_HIP_LIB=$PWD/variants/lib_biaslds2.so python -m pytest tests/test_gpu_parity.py -q -m gpu \
      -k "ragged or base_size_forward or public_seams or determinism or checkpoint or trained" > $O/pytest_biaslds2.log 2>&1; lap pytest biaslds2
  for rep in 1 2; do for f in variants/lib_*.so; do
    ISDF_HIP_LIB=$PWD/$f python bench.py --steps 300 --warmup 50 --no-cpu-baseline > $O/ab2_$(basename $f .so)_$rep.json 2> /dev/null; lap bench $f $rep
  done; done
  S20="$(seq 1 20)"
  python tests/accuracy_experiment.py --paired-draws --backend hip --seeds $S20 --keyframes 24 --steps-per-kf 100 \
      --out $O/acc_paired_hip.json > $O/acc_paired_hip.log 2>&1; lap paired hip
  python tests/accuracy_experiment.py --paired-draws --backend port --device cuda --seeds $S20 --keyframes 24 --steps-per-kf 100 \
      --out $O/acc_paired_control.json > $O/acc_paired_control.log 2>&1; lap paired control
  ISDF_BENCH_FORCE_DP=1 python bench.py --steps 300 --warmup 50 --no-cpu-baseline > $O/bench_forced_dp_world1.json 2> $O/bench_forced_dp_world1.err; lap dp1
  ISDF_BENCH_FORCE_DP=1 python bench.py --steps 300 --warmup 50 --no-cpu-baseline --overlap-allreduce > $O/bench_forced_dp_world1_overlap.json 2> $O/bench_forced_dp_world1_overlap.err; lap dp1 overlap
  for mode in "--scaling weak" "--scaling strong" "--scaling weak --overlap-allreduce"; do
    tag=$(echo $mode | tr -d ' -')
    ISDF_BENCH_BACKEND=gloo python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29611 \
        bench.py --gpus 2 --steps 100 --warmup 20 --no-cpu-baseline $mode > $O/bench_dp2_gloo_$tag.json 2> $O/bench_dp2_gloo_$tag.err; lap dp2 $tag
  done
  tail -n 12 $O/pytest_gpu2.log; tail -n 6 $O/pytest_biaslds2.log
  grep -h "trained-weights eval\|worst tensor\|trained-state" $O/pytest_gpu2.log | cut -c1-900
  for f in $O/ab2_*.json $O/bench_forced_dp_world1*.json $O/bench_dp2_gloo_*.json; do python - "$f" <<'PY'
import json, sys
try:
    j = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); fm = j.get("fast_mode_fp16") or {}
    print("%-40s %8.1f /s %.4f ms chain %.4f dw %.4f tail %.4f sync %.4f | %s | loss %.5f" % (
        sys.argv[1].split("/")[-1][:-5], j["value"], j["ms_per_step"], *list(j["kernel_ms"].values())[:3], j["trainer_step_sync_ms"],
        json.dumps(j.get("distributed")), j["final_total_loss"]))
except Exception as e:
    print(sys.argv[1], "unreadable:", e)
PY
  done
  tail -n 2 $O/acc_paired_hip.log $O/acc_paired_control.log
fi

if [ "$stage" = final ]; then
  python -m pytest tests -q -m gpu -s > $O/pytest_gpu_final.log 2>&1; lap pytest gpu
  bash tools/round_records.sh 04 > $O/round_records.log 2>&1; lap records
  cd $GRAFT_REPO_ROOT
  python tests/accuracy_experiment.py --native-clock --backend hip --seeds $SEEDS5 --out $O/native_clock_hip_final.json > $O/native_clock_hip_final.log 2>&1; lap native hip
  for m in fp16x2_full bf16; do
    python tests/accuracy_experiment.py --backend hip --seeds $(seq 1 40) --keyframes 24 --steps-per-kf 100 --fwd-operand $m \
        --out $O/acc_hip_24x100_$m.json > $O/acc_hip_24x100_$m.log 2>&1; lap accuracy $m
  done
  tail -n 8 $O/pytest_gpu_final.log
  grep -h "trained-weights eval\|worst tensor\|trained-state" $O/pytest_gpu_final.log | cut -c1-1100
  tail -n 40 $O/round_records.log
  tail -n 2 $O/native_clock_hip_final.log $O/acc_hip_24x100_fp16x2_full.log $O/acc_hip_24x100_bf16.log
fi

if [ "$stage" = bwd16 ]; then
  python -m pytest tests -q -m gpu -s > $O/pytest_gpu_bwd16.log 2>&1; lap pytest gpu
  for rep in 1 2; do for b in bf16 fp16; do
    python bench.py --steps 300 --warmup 50 --no-cpu-baseline --bwd-operand $b > $O/ab3_bwd_${b}_$rep.json 2> /dev/null; lap bench $b $rep
  done; done
  for b in fp16 bf16; do
    python tests/accuracy_experiment.py --backend hip --seeds $(seq 1 40) --keyframes 24 --steps-per-kf 100 --bwd-operand $b \
        --out $O/acc_hip_24x100_bwd_$b.json > $O/acc_hip_24x100_bwd_$b.log 2>&1; lap accuracy bwd $b
  done
  tail -n 8 $O/pytest_gpu_bwd16.log
  grep -h "trained-weights eval\|per-tensor rel-L2\|trained-state\|weight gradients vs oracle\|fwd .* bwd\|worst gradient deviation" $O/pytest_gpu_bwd16.log | cut -c1-700
  for f in $O/ab3_*.json; do python - "$f" <<'PY'
import json, sys
j = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("%-22s %8.1f /s %.4f ms chain %.4f dw %.4f tail %.4f sync %.4f | windowed %s | loss %.5f" % (
    sys.argv[1].split("ab3_")[1][:-5], j["value"], j["ms_per_step"], *list(j["kernel_ms"].values())[:3], j["trainer_step_sync_ms"],
    json.dumps((j.get("synchronised_step_windowed") or {}).get("window_transport_ab")), j["final_total_loss"]))
PY
  done
  tail -n 1 $O/acc_hip_24x100_bwd_fp16.log $O/acc_hip_24x100_bwd_bf16.log
fi

if [ "$stage" = final2 ]; then      # the records of the FINAL tree (fp16 second-order sweeps by default, window inline)
  python -m pytest tests -q -m gpu -s > $O/pytest_gpu_final2.log 2>&1; lap pytest gpu
  bash tools/round_records.sh 04 > $O/round_records2.log 2>&1; lap records
  cd $GRAFT_REPO_ROOT
  python tests/accuracy_experiment.py --native-clock --backend hip --seeds $SEEDS5 --out $O/native_clock_hip_final2.json > $O/native_clock_hip_final2.log 2>&1; lap native hip
  ISDF_BENCH_FORCE_DP=1 python bench.py --steps 300 --warmup 50 --no-cpu-baseline --overlap-allreduce 2> /dev/null | tail -1 > $O/bench_forced_dp_world1_overlap_final.json; lap dp1 overlap
  for mode in "--scaling weak" "--scaling strong" "--scaling weak --overlap-allreduce"; do
    tag=$(echo $mode | tr -d ' -')
    ISDF_BENCH_BACKEND=gloo python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29611 \
        bench.py --gpus 2 --steps 100 --warmup 20 --no-cpu-baseline $mode 2> /dev/null | tail -1 > $O/bench_dp2_gloo_final_$tag.json; lap dp2 $tag
  done
  tail -n 6 $O/pytest_gpu_final2.log
  tail -n 45 $O/round_records2.log
  tail -n 1 $O/native_clock_hip_final2.log
fi

#   hidden     zero-padded narrow nets (NetLayout::H): the 64-wide reference fixtures as GPU parity cases, then the full GPU suite and
#              a bench line (the 256-wide default must not have moved)
if [ "$stage" = hidden ]; then
  O=gpurun_out/r04hidden; mkdir -p $O
  timeout 120 python tests/fwd_race_probe.py --reps 20 > $O/race_probe.log 2>&1; lap "race probe rc=$?"
  grep -v "^JSON" $O/race_probe.log | tail -4
  timeout 300 python -m pytest tests/test_gpu_parity.py -q -m gpu -k narrow_nets -s > $O/pytest_narrow.log 2>&1; lap "narrow rc=$?"
  grep "sdf (scaled\|passed\|failed" $O/pytest_narrow.log
  timeout 600 python -m pytest tests -q -m gpu > $O/pytest_gpu.log 2>&1; lap "pytest gpu rc=$?"
  tail -5 $O/pytest_gpu.log
  timeout 300 python bench.py --no-cpu-baseline > $O/bench.json 2> $O/bench.err; lap bench
  cat $O/bench.json
fi

#   race       tests/fwd_race_probe.py over variants/lib_*.so: which points of a 27 000-point sdf_eval a misbehaving chain-kernel
#              build gets wrong (the padded-width change and, earlier, the bias-from-LDS experiment both broke exactly the cases with
#              more than 256 tiles)
if [ "$stage" = race ]; then
  O=gpurun_out/r04race; mkdir -p $O
  for f in ${RACE_LIBS:-variants/lib_*.so}; do
    name=$(basename $f .so)
    ISDF_HIP_LIB=$PWD/$f timeout 120 python tests/fwd_race_probe.py --reps ${RACE_REPS:-15} $([ $name = lib_partk ] && echo --partk) > $O/$name.log 2>&1; lap "$name rc=$?"
    grep -v "^JSON" $O/$name.log | tail -${RACE_TAIL:-14}
  done
fi

#   pinned     tools/probes/pk_fma_opsel (the instruction form of isa_lint rule 1 in isolation), the pinned-host frame_avg_losses
#              test, a bench line with the placement A/B
if [ "$stage" = pinned ]; then
  O=gpurun_out/r04pinned; mkdir -p $O
  timeout 120 tools/probes/pk_fma_opsel ${PK_ITERS:-4000} > $O/pk_fma_opsel.txt 2>&1; lap "pk probe rc=$?"
  cat $O/pk_fma_opsel.txt
  timeout 300 python -m pytest tests/test_gpu_parity.py -q -m gpu -k "pinned_host or trainer_step_contract or checkpoint or driver_schedule" > $O/pytest_pinned.log 2>&1; lap "pytest rc=$?"
  tail -4 $O/pytest_pinned.log
  timeout 300 python bench.py --no-cpu-baseline > $O/bench.json 2> $O/bench.err; lap bench
  python - <<'PY'
import json
j = json.loads(open("gpurun_out/r04pinned/bench.json").read().strip().split("\n")[-1])
print("value", j["value"], "sync ms", j["trainer_step_sync_ms"], "windowed", json.dumps(j["synchronised_step_windowed"])[:900])
PY
fi

if [ "$stage" = final3 ]; then      # the records of the FINAL tree of round 4 (padded widths, element-pair reduction, pinned frame losses)
  python -m pytest tests -q -m gpu -s > $O/pytest_gpu_final3.log 2>&1; lap pytest gpu
  bash tools/round_records.sh 04 > $O/round_records3.log 2>&1; lap records
  cd $GRAFT_REPO_ROOT
  python tests/accuracy_experiment.py --native-clock --backend hip --seeds $SEEDS5 --out $O/native_clock_hip_final3.json > $O/native_clock_hip_final3.log 2>&1; lap native hip
  tail -n 4 $O/pytest_gpu_final3.log
  tail -n 45 $O/round_records3.log
  tail -n 1 $O/native_clock_hip_final3.log
fi
