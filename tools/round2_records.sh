#!/bin/bash
# Round-2 records (run on the GPU box through gpurun): bench lines, rocprofv3 kernel stats, HBM PMC passes (separate
# passes, --kernel-trace only), sampler at 1e6 rays, phase timeline.  Outputs under gpurun_out/r2final/.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r2final; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
python $R/bench.py 2>/dev/null | tail -1 > $O/bench.json
python $R/bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_driver_args.json
python $R/bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --ramp-seconds 0 2>/dev/null | tail -1 > $O/bench_driver_args_noramp.json
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python $R/bench.py --steps 100 --warmup 10 --no-cpu-baseline > $O/stats.log 2>&1
for set in "FETCH_SIZE" "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum"; do
  n=$(echo $set | cut -d' ' -f1)
  timeout 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $O/pmc_$n -- python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --ramp-seconds 0 > $O/pmc_$n.log 2>&1
  timeout 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $O/pmcs_$n -- python $R/bench.py --sampler-scale 200000 --steps 100 --no-cpu-baseline > $O/pmcs_$n.log 2>&1
done
python $R/bench.py --sampler-scale 200000 --steps 300 2>/dev/null | tail -1 > $O/bench_sampler_1M.json
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_sampler -- python $R/bench.py --sampler-scale 200000 --steps 100 > $O/stats_sampler.log 2>&1
python $R/tools/timeline.py > $O/timeline.txt 2>&1
ISDF_CHAIN_PAIR=1 python $R/bench.py --steps 300 --warmup 50 --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_pair_kernel.json
python $R/tools/timeline_pair.py > $O/timeline_pair.txt 2>&1
python $R/bench.py --rays-per-frame 5400 --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_729k.json
python $R/bench.py --wide --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_wide.json
python $R/bench.py --infer-points 8000000 2>/dev/null | tail -1 > $O/bench_inference_8M.json
python $R/bench.py --ingest 2>/dev/null | tail -1 > $O/bench_ingest.json
timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_ingest -- python $R/bench.py --ingest > $O/stats_ingest.log 2>&1
python - <<'PY'
import csv, glob, collections, os, json
R = os.environ["GRAFT_REPO_ROOT"]; O = R + "/gpurun_out/r2final"
def collect(pat):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in sorted(glob.glob(O + "/" + pat + "/**/*counter_collection.csv", recursive=True)):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            for tag in ("chain_kernel", "dw_kernel", "step_tail_kernel", "sample_rays_kernel"):
                if tag in k:
                    acc[tag][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return {k: {c: sum(v) / len(v) for c, v in d.items()} for k, d in acc.items()}
a, b = collect("pmc_*"), collect("pmcs_*")
rows = ["# rocprofv3 --pmc <counters> --kernel-trace --output-format csv -- python bench.py --steps 20 --warmup 5 --no-cpu-baseline --ramp-seconds 0",
        "# (sampler_scale rows: ... bench.py --sampler-scale 200000 --steps 100); two separate passes (FETCH_SIZE | WRITE_SIZE TCC_HIT_sum TCC_MISS_sum)",
        "# per-dispatch averages, MI355X; FETCH_SIZE/WRITE_SIZE in KB; FETCH_SIZE counts 64 B per 128-B request for wide reads on gfx950 -> x2",
        "kernel,counter,avg_per_dispatch"]
for tag, d in list(a.items()) + [("sampler_scale:" + k, v) for k, v in b.items()]:
    for c, v in sorted(d.items()):
        rows.append("%s,%s,%.6g" % (tag, c, v))
open(O + "/pmc_summary.csv", "w").write("\n".join(rows) + "\n")
out = {"source": "profiles/r02_pmc_bench.csv",
       "note": "FETCH_SIZE*1024*2 (gfx950 wide-read correction, MI355X_MICROARCH.md HBM section) + WRITE_SIZE*1024, per dispatch"}
for tag, d in a.items():
    if "FETCH_SIZE" in d and "WRITE_SIZE" in d:
        out[tag] = {"fetch_bytes_x2": d["FETCH_SIZE"] * 2048, "write_bytes": d["WRITE_SIZE"] * 1024,
                    "hbm_bytes": d["FETCH_SIZE"] * 2048 + d["WRITE_SIZE"] * 1024}
d = b.get("sample_rays_kernel", {})
if "FETCH_SIZE" in d and "WRITE_SIZE" in d:
    # random 4-B / 12-B gathers are NOT wide reads: the x2 correction is calibrated for 16 B/lane streaming only, so both
    # readings are recorded
    out["sampler_scale"] = {"rays": 1000000, "fetch_bytes_raw": d["FETCH_SIZE"] * 1024, "fetch_bytes_x2": d["FETCH_SIZE"] * 2048,
                            "write_bytes": d["WRITE_SIZE"] * 1024, "hbm_bytes": d["FETCH_SIZE"] * 2048 + d["WRITE_SIZE"] * 1024}
json.dump(out, open(O + "/hbm_traffic.json", "w"), indent=1)
print(open(O + "/pmc_summary.csv").read())
PY
ls $O
