#!/bin/bash
# Round-5 GPU sessions (one gpurun call each):  tools/r05_session.sh <stage>
#   fwd1     pair-tile forward kernel: bit-identity against the one-tile kernel (variants/lib_onetile.so), timing at 8 M points,
#            stage timelines (variants/lib_dbg.so), the GPU suite, inference + step bench lines
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
