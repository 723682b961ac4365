#!/bin/bash
# Instruction mix + wave-state counters of the tile kernels (run on the GPU box through gpurun): the inputs of the SIMD
# issue-port model (tools/probes/valu_rate.hip: next to MFMAs a plain VALU wave-instruction costs ~4 issue cycles of its SIMD, a
# transcendental 8, an MFMA 8 of its 32).  Separate rocprofv3 --pmc passes with --kernel-trace only.
#   tools/pmc_issue_mix.sh           -> gpurun_out/pmc_issue/summary.csv  (chain_kernel of the training step, fwd_pair_kernel at 2 M points)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/pmc_issue; mkdir -p $O
for set in "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU" \
           "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_INSTS_VALU_TRANS GRBM_GUI_ACTIVE"; do
  n=$(echo $set | cut -d' ' -f1)
  timeout 150 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $O/train_$n -- python $R/tools/train_only.py 6 > $O/train_$n.log 2>&1
  timeout 150 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $O/fwd_$n -- env ISDF_FWD_POINTS=2000000 ISDF_FWD_OPERAND=fp16 python $R/tools/fwd_only.py > $O/fwd_$n.log 2>&1
done
python - <<'PY'
import csv, glob, collections, os
R = os.environ["GRAFT_REPO_ROOT"]; O = R + "/gpurun_out/pmc_issue"
rows = []
for f in sorted(glob.glob(O + "/*/**/*_counter_collection.csv", recursive=True)):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        k = "fwd_pair_kernel" if "fwd_pair_kernel" in k else ("chain_kernel" if "chain_kernel" in k else ("dw_kernel" if "dw_kernel" in k else None))
        if k: acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k in acc:
        for c, v in acc[k].items(): rows.append((k, c, sum(v) / len(v), len(v)))
with open(O + "/summary.csv", "w") as f:
    f.write("kernel,counter,avg_per_dispatch,dispatches\n")
    for k, c, v, n in sorted(set(rows)): f.write("%s,%s,%.6g,%d\n" % (k, c, v, n))
print(open(O + "/summary.csv").read())
PY
tail -n 3 $O/*.log | tail -n 30
