#!/bin/bash
# round 3, eleventh GPU call: VALU diet (sigma' spilled as fp16 by the forward epilogue instead of re-derived with an exp in three
# sweeps): parity suite, same-box A/B, instruction / wait counters before and after
O=gpurun_out/r03k; mkdir -p $O
python -m pytest tests -m gpu -x -q 2>&1 | tail -4
bash tools/ab_bench.sh > $O/ab.txt 2>&1; cat $O/ab.txt
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for v in old s1h; do
for set in "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU"; do
  n=$(echo $set | cut -d' ' -f1)
  ISDF_HIP_LIB=$R/variants/lib_$v.so timeout 120 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $R/$O/pmc_${v}_$n -- python $R/tools/train_only.py 6 > /dev/null 2>&1
done
done
cd $R
python - <<'PY'
import csv, glob, collections
for v in ("old", "s1h"):
    acc = collections.defaultdict(list)
    for f in sorted(glob.glob("gpurun_out/r03k/pmc_%s_*/**/*_counter_collection.csv" % v, recursive=True)):
        for r in csv.DictReader(open(f)):
            if "chain_kernel" in r["Kernel_Name"]: acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    print(v, {k: "%.4g" % (sum(x)/len(x)) for k, x in sorted(acc.items())})
PY
