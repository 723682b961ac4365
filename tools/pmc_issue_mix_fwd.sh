#!/bin/bash
# The forward kernel's part of tools/pmc_issue_mix.sh only (two rocprofv3 --pmc passes: the counters tools/issue_model.py needs).
#   tools/pmc_issue_mix_fwd.sh   -> gpurun_out/pmc_issue_fwd/summary.csv   (fwd_pair_kernel, fp16 operands, 2 M points)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/pmc_issue_fwd; mkdir -p $O
for set in "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_INSTS_VALU_TRANS GRBM_GUI_ACTIVE"; do
  n=$(echo $set | cut -d' ' -f1)
  timeout 100 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $O/fwd_$n -- env ISDF_FWD_POINTS=2000000 ISDF_FWD_OPERAND=fp16 python $R/tools/fwd_only.py > $O/fwd_$n.log 2>&1
done
python - <<'PY'
import csv, glob, collections, os
R = os.environ["GRAFT_REPO_ROOT"]; O = R + "/gpurun_out/pmc_issue_fwd"
rows = []
for f in sorted(glob.glob(O + "/*/**/*_counter_collection.csv", recursive=True)):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if "fwd_pair_kernel" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for c, v in acc.items(): rows.append(("fwd_pair_kernel", c, sum(v) / len(v), len(v)))
with open(O + "/summary.csv", "w") as f:
    f.write("kernel,counter,avg_per_dispatch,dispatches\n")
    for k, c, v, n in sorted(set(rows)): f.write("%s,%s,%.6g,%d\n" % (k, c, v, n))
print(open(O + "/summary.csv").read())
PY
