#!/bin/bash
# Round-5 GPU sessions (one gpurun call each):  tools/r05_session.sh <stage>
#   fwd1     pair-tile forward kernel: bit-identity against the one-tile kernel (variants/lib_onetile.so), timing at 8 M points,
#            stage timelines (variants/lib_dbg.so), the GPU suite, inference + step bench lines
#   records  the GPU suite and the end-of-round records (tools/round_records.sh 05)
#   chain1   chain-kernel changes (lane = point PE-shaped stages, reverse-epilogue trim) against variants/lib_prev.so: outputs of one
#            step compared array by array, same-box bench A/B (two repetitions), MODE 2 timeline, the whole GPU suite
#   gap      VERDICT r4 item 4: is the +0.30 cm of the 16-bit path against the fp32 control real?  PAIRED draws (same initial network and
#            random streams), seeds 21..120 for the HIP path and the fp32 eager control (five control processes side by side: eager is
#            launch-bound), and the HIP path with library-accurate transcendentals (variants/lib_libm.so) on seeds 1..60
#   fwd3     fwd2 + the GPU suite (the Softplus reformulation touches every mode of the chain kernel) + bench lines
#   fwd2     the same check / timing / timelines after a kernel change (no test suite, no step bench): the quick iteration stage
# Outputs land in gpurun_out/r05*/ (scratch); what is judged is copied to profiles/ by hand.
set -u
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
stage=${1:-fwd1}
O=gpurun_out/r05_$stage
mkdir -p $O
t0=$(date +%s)
lap() { echo "[$(( $(date +%s) - t0 )) s] $*"; }

if [ "$stage" = fwd1 ]; then
  timeout 600 python tools/fwd_pair_check.py --dump /tmp/a.npz --time 8000000 > $O/check_pair.log 2>&1; lap pair dump rc=$?
  ISDF_HIP_LIB=$PWD/variants/lib_onetile.so timeout 600 python tools/fwd_pair_check.py --dump /tmp/b.npz --time 8000000 > $O/check_onetile.log 2>&1; lap onetile dump rc=$?
  python tools/fwd_pair_check.py --compare /tmp/a.npz /tmp/b.npz > $O/compare.log 2>&1; lap compare rc=$?
  ISDF_FWD_OPERAND=fp16 timeout 300 python tools/timeline_fwd.py > $O/timeline_fp16.txt 2>&1; lap timeline fp16
  ISDF_FWD_OPERAND=fp16x2 timeout 300 python tools/timeline_fwd.py > $O/timeline_fp16x2.txt 2>&1; lap timeline fp16x2
  timeout 900 python -m pytest tests -q -m gpu -x > $O/pytest_gpu.log 2>&1; lap pytest rc=$?
  timeout 600 python bench.py --infer-points 8000000 > $O/bench_infer.json 2> $O/bench_infer.err; lap bench infer
  ISDF_HIP_LIB=$PWD/variants/lib_onetile.so timeout 600 python bench.py --infer-points 8000000 > $O/bench_infer_onetile.json 2> /dev/null; lap bench infer onetile
  timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; lap bench
  tail -n 4 $O/check_pair.log $O/check_onetile.log; tail -n 40 $O/compare.log; tail -n 5 $O/pytest_gpu.log
  head -c 1500 $O/bench_infer.json; echo; head -c 600 $O/bench_infer_onetile.json; echo
  cat $O/timeline_fp16.txt | tail -n 45
fi

if [ "$stage" = fwd2 ]; then
  timeout 600 python tools/fwd_pair_check.py --dump /tmp/a.npz --time 8000000 > $O/check_pair.log 2>&1; lap pair dump rc=$?
  ISDF_HIP_LIB=$PWD/variants/lib_onetile.so timeout 600 python tools/fwd_pair_check.py --dump /tmp/b.npz --time 8000000 > $O/check_onetile.log 2>&1; lap onetile dump rc=$?
  if [ -f variants/lib_prevpair.so ]; then   # the pair kernel before the change under test, on the same box
    ISDF_HIP_LIB=$PWD/variants/lib_prevpair.so timeout 600 python tools/fwd_pair_check.py --dump /tmp/c.npz --time 8000000 > $O/check_prevpair.log 2>&1; lap prevpair dump rc=$?
    timeout 600 python tools/fwd_pair_check.py --dump /tmp/a2.npz --time 8000000 > $O/check_pair_again.log 2>&1; lap pair again rc=$?
    tail -n 4 $O/check_prevpair.log $O/check_pair_again.log
  fi
  python tools/fwd_pair_check.py --compare /tmp/a.npz /tmp/b.npz > $O/compare.log 2>&1; lap compare rc=$?
  ISDF_FWD_OPERAND=fp16 timeout 300 python tools/timeline_fwd.py > $O/timeline_fp16.txt 2>&1; lap timeline fp16
  ISDF_FWD_OPERAND=fp16x2 timeout 300 python tools/timeline_fwd.py > $O/timeline_fp16x2.txt 2>&1; lap timeline fp16x2
  tail -n 4 $O/check_pair.log $O/check_onetile.log; grep -c "bit-identical" $O/compare.log; grep "DIFFERENT" $O/compare.log | head -20; tail -n 1 $O/compare.log
  tail -n 32 $O/timeline_fp16.txt; tail -n 32 $O/timeline_fp16x2.txt
fi

if [ "$stage" = fwd3 ]; then
  timeout 600 python tools/fwd_pair_check.py --dump /tmp/a.npz --time 8000000 > $O/check_pair.log 2>&1; lap pair dump rc=$?
  ISDF_HIP_LIB=$PWD/variants/lib_onetile.so timeout 600 python tools/fwd_pair_check.py --dump /tmp/b.npz --time 8000000 > $O/check_onetile.log 2>&1; lap onetile dump rc=$?
  python tools/fwd_pair_check.py --compare /tmp/a.npz /tmp/b.npz > $O/compare.log 2>&1; lap compare rc=$?
  ISDF_FWD_OPERAND=fp16 timeout 300 python tools/timeline_fwd.py > $O/timeline_fp16.txt 2>&1; lap timeline fp16
  ISDF_FWD_OPERAND=fp16x2 timeout 300 python tools/timeline_fwd.py > $O/timeline_fp16x2.txt 2>&1; lap timeline fp16x2
  timeout 900 python -m pytest tests -q -m gpu > $O/pytest_gpu.log 2>&1; lap pytest rc=$?
  timeout 600 python bench.py --infer-points 8000000 > $O/bench_infer.json 2> $O/bench_infer.err; lap bench infer
  timeout 900 python bench.py --no-cpu-baseline > $O/bench.json 2> $O/bench.err; lap bench
  tail -n 4 $O/check_pair.log $O/check_onetile.log; grep -c "bit-identical" $O/compare.log; grep "DIFFERENT" $O/compare.log | head -20; tail -n 1 $O/compare.log
  tail -n 32 $O/timeline_fp16.txt; tail -n 15 $O/timeline_fp16x2.txt; tail -n 12 $O/pytest_gpu.log
  python - "$O/bench.json" <<'PY'
import json, sys
j = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("bench: %.1f steps/s %.4f ms kernels %s frac %.4f sync %.4f" % (j["value"], j["ms_per_step"], j["kernel_ms"], j["roofline"]["frac"], j["trainer_step_sync_ms"]))
PY
fi

if [ "$stage" = gap ]; then
  A="--paired-draws --keyframes 24 --steps-per-kf 100"
  for k in 0 1 2 3 4; do
    lo=$((21 + 20 * k)); hi=$((40 + 20 * k))
    timeout 1500 python tests/accuracy_experiment.py $A --backend port --device cuda --seeds $(seq $lo $hi) --out $O/paired_control_${lo}_${hi}.json > $O/paired_control_${lo}_${hi}.log 2>&1 &
  done
  timeout 1200 python tests/accuracy_experiment.py $A --backend hip --seeds $(seq 21 120) --out $O/paired_hip_21_120.json > $O/paired_hip_21_120.log 2>&1; lap hip 21..120 rc=$?
  ISDF_HIP_LIB=$PWD/variants/lib_libm.so timeout 1200 python tests/accuracy_experiment.py $A --backend hip --seeds $(seq 1 60) --out $O/paired_hip_libm_1_60.json > $O/paired_hip_libm_1_60.log 2>&1; lap hip libm 1..60 rc=$?
  wait; lap controls done
  tail -n 2 $O/*.log
fi

if [ "$stage" = chain1 ]; then
  timeout 300 python tools/train_ab_check.py --dump /tmp/ta.npz > $O/ab_dump_new.log 2>&1; lap dump new rc=$?
  ISDF_HIP_LIB=$PWD/variants/lib_prev.so timeout 300 python tools/train_ab_check.py --dump /tmp/tb.npz > $O/ab_dump_prev.log 2>&1; lap dump prev rc=$?
  python tools/train_ab_check.py --compare /tmp/ta.npz /tmp/tb.npz > $O/ab_compare.log 2>&1; lap compare
  for rep in 1 2; do for f in variants/lib_prev.so isdf_amd/libisdf_hip.so; do
    ISDF_HIP_LIB=$PWD/$f timeout 300 python bench.py --steps 300 --warmup 50 --no-cpu-baseline > $O/ab_$(basename $f .so)_$rep.json 2> /dev/null; lap bench $f $rep
  done; done
  timeout 300 python tools/timeline.py > $O/timeline_train.txt 2>&1; lap timeline
  timeout 900 python -m pytest tests -q -m gpu -s > $O/pytest_gpu.log 2>&1; lap pytest rc=$?
  cat $O/ab_compare.log
  for f in $O/ab_*.json; do python - "$f" <<'PY'
import json, sys
j = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); fm = j.get("fast_mode_fp16") or {}
print("%-28s %8.1f steps/s %.4f ms chain %.4f dw %.4f tail %.4f sync %.4f | fp16 %8.1f chain %.4f | loss %.5f" % (
    sys.argv[1].split("ab_")[1][:-5], j["value"], j["ms_per_step"], *list(j["kernel_ms"].values())[:3], j["trainer_step_sync_ms"],
    fm.get("steps_per_s", 0), fm.get("chain_ms", 0), j["final_total_loss"]))
PY
  done
  head -n 40 $O/timeline_train.txt; tail -n 3 $O/timeline_train.txt
  grep -E "passed|failed" $O/pytest_gpu.log | tail -n 3; grep -E "^(trained franka|  worst tensor|loss weights x|fp16x2: .sdf|fp16: .sdf|bf16: .sdf)" $O/pytest_gpu.log
fi

if [ "$stage" = records ]; then
  timeout 900 python -m pytest tests -q -m gpu -s > $O/pytest_gpu.log 2>&1; lap pytest rc=$?
  grep -E "passed|failed" $O/pytest_gpu.log | tail -n 3; grep -E "^(trained franka|  worst tensor|loss weights x|fp16x2: .sdf|fp16: .sdf|bf16: .sdf)" $O/pytest_gpu.log
  bash tools/round_records.sh 05 > $O/round_records.log 2>&1; lap records
  tail -n 60 $O/round_records.log
fi

if [ "$stage" = tp1 ]; then      # pair-tile train kernel: outputs of one step against variants/lib_prev.so (one-tile kernel), array by array
  timeout 300 python tools/train_ab_check.py --dump /tmp/ta.npz > $O/ab_dump_new.log 2>&1; lap dump new rc=$?
  ISDF_HIP_LIB=$PWD/variants/lib_prev.so timeout 300 python tools/train_ab_check.py --dump /tmp/tb.npz > $O/ab_dump_prev.log 2>&1; lap dump prev rc=$?
  python tools/train_ab_check.py --compare /tmp/ta.npz /tmp/tb.npz > $O/ab_compare.log 2>&1; lap compare
  cat $O/ab_compare.log; tail -n 3 $O/ab_dump_new.log
  for f in variants/lib_prev.so isdf_amd/libisdf_hip.so; do
    ISDF_HIP_LIB=$PWD/$f timeout 300 python bench.py --steps 300 --warmup 50 --no-cpu-baseline > $O/ab_$(basename $f .so).json 2> $O/ab_$(basename $f .so).err; lap bench $f
  done
  for f in $O/ab_lib*.json; do python - "$f" <<'PY'
import json, sys
try:
    j = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); fm = j.get("fast_mode_fp16") or {}
    print("%-28s %8.1f steps/s %.4f ms chain %.4f dw %.4f tail %.4f sync %.4f | fp16 %8.1f chain %.4f | loss %.5f" % (
        sys.argv[1].split("ab_")[1][:-5], j["value"], j["ms_per_step"], *list(j["kernel_ms"].values())[:3], j["trainer_step_sync_ms"],
        fm.get("steps_per_s", 0), fm.get("chain_ms", 0), j["final_total_loss"]))
except Exception as e:
    print(sys.argv[1], "ERR", e)
PY
  done
fi

if [ "$stage" = tp2 ]; then      # pair-tile train kernel: stage timeline (variants/lib_dbg.so) + bench
  timeout 300 python tools/timeline.py > $O/timeline_train.txt 2>&1; lap timeline
  ISDF_FWD_OPERAND=fp16 timeout 300 python tools/timeline.py > $O/timeline_train_fp16.txt 2>&1; lap timeline fp16
  timeout 300 python bench.py --steps 300 --warmup 50 --no-cpu-baseline > $O/bench.json 2> $O/bench.err; lap bench
  cat $O/timeline_train.txt | head -130; tail -n 8 $O/timeline_train.txt
  python - "$O/bench.json" <<'PY'
import json, sys
j = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); fm = j.get("fast_mode_fp16") or {}
print("bench %8.1f steps/s %.4f ms chain %.4f dw %.4f tail %.4f sync %.4f | fp16 %8.1f chain %.4f" % (j["value"], j["ms_per_step"], *list(j["kernel_ms"].values())[:3], j["trainer_step_sync_ms"], fm.get("steps_per_s", 0), fm.get("chain_ms", 0)))
PY
fi

if [ "$stage" = tp3 ]; then      # pair-tile train kernel variants: cache policy of the spill traffic, epilogue groups late in the stage
  for rep in 1 2; do for f in variants/lib_prev.so isdf_amd/libisdf_hip.so variants/lib_tpdef.so variants/lib_tplate.so; do
    ISDF_HIP_LIB=$PWD/$f timeout 300 python bench.py --steps 300 --warmup 50 --no-cpu-baseline > $O/ab_$(basename $f .so)_$rep.json 2> /dev/null; lap bench $f $rep
  done; done
  for f in $O/ab_lib*.json; do python - "$f" <<'PY'
import json, sys
try:
    j = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); fm = j.get("fast_mode_fp16") or {}
    print("%-28s %8.1f steps/s %.4f ms chain %.4f dw %.4f tail %.4f sync %.4f | fp16 %8.1f chain %.4f | loss %.5f" % (
        sys.argv[1].split("ab_")[1][:-5], j["value"], j["ms_per_step"], *list(j["kernel_ms"].values())[:3], j["trainer_step_sync_ms"],
        fm.get("steps_per_s", 0), fm.get("chain_ms", 0), j["final_total_loss"]))
except Exception as e:
    print(sys.argv[1], "ERR", e)
PY
  done
fi

if [ "$stage" = tp4 ]; then      # pair-tile train kernel: check vs prev + bench A/B + timeline
  timeout 300 python tools/train_ab_check.py --dump /tmp/ta.npz > $O/ab_dump_new.log 2>&1; lap dump new rc=$?
  ISDF_HIP_LIB=$PWD/variants/lib_prev.so timeout 300 python tools/train_ab_check.py --dump /tmp/tb.npz > $O/ab_dump_prev.log 2>&1; lap dump prev rc=$?
  python tools/train_ab_check.py --compare /tmp/ta.npz /tmp/tb.npz > $O/ab_compare.log 2>&1; head -n 8 $O/ab_compare.log
  for rep in 1 2; do for f in variants/lib_prev.so isdf_amd/libisdf_hip.so; do
    ISDF_HIP_LIB=$PWD/$f timeout 300 python bench.py --steps 300 --warmup 50 --no-cpu-baseline > $O/ab_$(basename $f .so)_$rep.json 2> /dev/null; lap bench $f $rep
  done; done
  for f in $O/ab_lib*.json; do python - "$f" <<'PY'
import json, sys
try:
    j = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); fm = j.get("fast_mode_fp16") or {}
    print("%-28s %8.1f steps/s %.4f ms chain %.4f dw %.4f tail %.4f sync %.4f | fp16 %8.1f chain %.4f | loss %.5f" % (
        sys.argv[1].split("ab_")[1][:-5], j["value"], j["ms_per_step"], *list(j["kernel_ms"].values())[:3], j["trainer_step_sync_ms"],
        fm.get("steps_per_s", 0), fm.get("chain_ms", 0), j["final_total_loss"]))
except Exception as e:
    print(sys.argv[1], "ERR", e)
PY
  done
  timeout 300 python tools/timeline.py > $O/timeline_train.txt 2>&1; lap timeline
  grep -v "^  wg" $O/timeline_train.txt | tail -n 108
fi

if [ "$stage" = det ]; then    # is the pair-tile forward kernel deterministic (run to run), and do the variants named in ISDF_DET_VARIANTS compute the same bits?
  for v in base ${ISDF_DET_VARIANTS:-}; do
    ISDF_HIP_LIB=$PWD/variants/lib_$v.so timeout 300 python tools/fwd_pair_check.py --dump /tmp/$v.npz > $O/check_$v.log 2>&1; lap $v rc=$?
  done
  ISDF_HIP_LIB=$PWD/variants/lib_base.so timeout 300 python tools/fwd_pair_check.py --dump /tmp/base2.npz > $O/check_base2.log 2>&1; lap base2 rc=$?
  for v in base2 ${ISDF_DET_VARIANTS:-}; do echo "== base vs $v"; python tools/fwd_pair_check.py --compare /tmp/base.npz /tmp/$v.npz > $O/compare_$v.log 2>&1; grep -v "bit-identical" $O/compare_$v.log | head -n 40; done
fi
