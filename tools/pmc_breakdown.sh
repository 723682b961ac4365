#!/bin/bash
# Instruction-mix and wave-state counters of the chain / dW kernels (run on the GPU box through gpurun).
# ISDF_CHAIN_PAIR=1 tools/pmc_breakdown.sh collects them for the pair-tile kernel instead of the one-tile kernel.
# Separate rocprofv3 --pmc passes (with --kernel-trace only); summaries go to gpurun_out/pmc2/.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/pmc2
for set in "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS"; do
  n=$(echo $set | cut -d' ' -f1)
  timeout 120 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $R/gpurun_out/pmc2/$n -- python $R/tools/train_only.py 6 > $R/gpurun_out/pmc2/$n.log 2>&1
done
python - <<'PY'
import csv, glob, collections, os
R = os.environ["GRAFT_REPO_ROOT"]
rows = []
for f in sorted(glob.glob(R + "/gpurun_out/pmc2/*/**/*_counter_collection.csv", recursive=True)):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        k = "chain_pair_kernel" if "chain_pair_kernel" in k else ("chain_kernel" if "chain_kernel" in k else ("dw_kernel" if "dw_kernel" in k else None))
        if k: acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k in acc:
        for c, v in acc[k].items(): rows.append((k, c, sum(v) / len(v)))
with open(R + "/gpurun_out/pmc2/summary.csv", "w") as f:
    f.write("kernel,counter,avg_per_dispatch\n")
    for k, c, v in rows: f.write("%s,%s,%.6g\n" % (k, c, v))
print(open(R + "/gpurun_out/pmc2/summary.csv").read())
PY
