#!/bin/bash
# Vector-memory-path counters of the chain / dW kernels (run on the GPU box through gpurun): TA / TCP (L1) / TCC (L2) busy, stall and
# latency counters, separate rocprofv3 --pmc passes (--kernel-trace only).  Summary -> gpurun_out/pmc3/summary.csv
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/pmc3; mkdir -p $O
i=0
for set in "GRBM_GUI_ACTIVE TA_TA_BUSY_sum TA_BUSY_avr TA_BUSY_max" \
           "TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_ADDR_STALLED_BY_TD_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_BUFFER_TOTAL_CYCLES_sum" \
           "TA_BUFFER_LOAD_WAVEFRONTS_sum TA_BUFFER_STORE_WAVEFRONTS_sum TA_BUFFER_COALESCED_READ_CYCLES_sum TA_BUFFER_COALESCED_WRITE_CYCLES_sum" \
           "TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_TD_TCP_STALL_CYCLES_sum TCP_GATE_EN1_sum" \
           "TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_WRITE_REQ_sum TCP_TCC_WRITE_REQ_LATENCY_sum" \
           "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TOTAL_READ_sum TCP_TOTAL_WRITE_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum" \
           "TCC_BUSY_sum TCC_CYCLE_sum TCC_REQ_sum TCC_TAG_STALL_sum" \
           "TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_WRREQ_LEVEL_sum TCC_EA0_WRREQ_STALL_sum TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum" \
           "MemUnitBusy MemUnitStalled" "SQ_INST_CYCLES_VMEM SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR"; do
  i=$((i+1))
  timeout 120 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $O/p$i -- python $R/tools/train_only.py 6 > $O/p$i.log 2>&1 || echo "pass $i failed: $set"
done
python - <<'PY'
import csv, glob, collections, os
R = os.environ["GRAFT_REPO_ROOT"]; O = R + "/gpurun_out/pmc3"
rows = []
for f in sorted(glob.glob(O + "/p*/**/*_counter_collection.csv", recursive=True)):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        k = "chain_kernel" if "chain_kernel" in k else ("dw_kernel" if "dw_kernel" in k else ("step_tail" if "step_tail" in k else None))
        if k: acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k in acc:
        for c, v in acc[k].items(): rows.append((k, c, sum(v) / len(v)))
with open(O + "/summary.csv", "w") as f:
    f.write("kernel,counter,avg_per_dispatch\n")
    for k, c, v in sorted(rows): f.write("%s,%s,%.6g\n" % (k, c, v))
print(open(O + "/summary.csv").read())
PY
