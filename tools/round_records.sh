#!/bin/bash
# End-of-round records (run on the GPU box through gpurun):  ISDF_COMMIT=$(git rev-parse --short HEAD) tools/round_records.sh <NN>   e.g. 06
# bench lines, rocprofv3 kernel stats, HBM PMC passes (separate passes, --kernel-trace only), sampler at 1e6 rays, the other
# workloads.  Outputs under gpurun_out/r<NN>final/ (scratch); copy what is judged to profiles/r<NN>_*.
NN=${1:-04}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r${NN}final; mkdir -p $O; export ISDF_RECORDS_DIR=$O ISDF_RECORDS_ROUND=$NN
cd /tmp && export TMPDIR=/tmp
python $R/bench.py 2>/dev/null | tail -1 > $O/bench.json                                   # defaults: --steps 300 --warmup 30
python $R/bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-accuracy 2>/dev/null | tail -1 > $O/bench_driver_args.json
python $R/bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-accuracy --ramp-seconds 0 2>/dev/null | tail -1 > $O/bench_driver_args_noramp.json
ISDF_BENCH_FORCE_DP=1 python $R/bench.py --steps 300 --warmup 30 --no-cpu-baseline --no-accuracy 2>/dev/null | tail -1 > $O/bench_forced_dp_world1.json
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python $R/bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-accuracy > $O/stats.log 2>&1
for set in "FETCH_SIZE" "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum"; do
  n=$(echo $set | cut -d' ' -f1)
  timeout 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $O/pmc_$n -- python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-accuracy --ramp-seconds 0 > $O/pmc_$n.log 2>&1
  timeout 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $O/pmcs_$n -- python $R/bench.py --sampler-scale 200000 --steps 100 --no-cpu-baseline --no-accuracy > $O/pmcs_$n.log 2>&1
done
python $R/bench.py --sampler-scale 200000 --steps 300 2>/dev/null | tail -1 > $O/bench_sampler_1M.json
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_sampler -- python $R/bench.py --sampler-scale 200000 --steps 100 > $O/stats_sampler.log 2>&1
python $R/bench.py --rays-per-frame 5400 --steps 30 --warmup 5 --no-cpu-baseline --no-accuracy 2>/dev/null | tail -1 > $O/bench_729k.json
python $R/bench.py --wide --steps 30 --warmup 5 --no-cpu-baseline --no-accuracy 2>/dev/null | tail -1 > $O/bench_wide.json
python $R/bench.py --infer-points 8000000 2>/dev/null | tail -1 > $O/bench_inference_8M.json
python $R/bench.py --ingest 2>/dev/null | tail -1 > $O/bench_ingest.json
python $R/bench.py --stream 480x640 --no-cpu-baseline --no-accuracy 2>/dev/null | tail -1 > $O/bench_stream_480x640.json          # the north star's synthetic 640x480 stream
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_infer -- python $R/bench.py --infer-points 8000000 > $O/stats_infer.log 2>&1
timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_ingest -- python $R/bench.py --ingest > $O/stats_ingest.log 2>&1
python - <<'PY'
import csv, glob, collections, os, json
R = os.environ["GRAFT_REPO_ROOT"]; O = os.environ["ISDF_RECORDS_DIR"]; NN = os.environ["ISDF_RECORDS_ROUND"]
def collect(pat):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in sorted(glob.glob(O + "/" + pat + "/**/*counter_collection.csv", recursive=True)):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            for tag in ("chain_kernel", "fwd_pair_kernel", "dw_kernel", "step_tail_kernel", "sample_rays_kernel"):
                if tag in k:
                    acc[tag][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return {k: {c: sum(v) / len(v) for c, v in d.items()} for k, d in acc.items()}
a, b = collect("pmc_*"), collect("pmcs_*")
rows = ["# rocprofv3 --pmc <counters> --kernel-trace --output-format csv -- python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-accuracy --ramp-seconds 0",
        "# (sampler_scale rows: ... bench.py --sampler-scale 200000 --steps 100); separate passes per counter set",
        "# per-dispatch averages, MI355X; FETCH_SIZE/WRITE_SIZE in KB; FETCH_SIZE counts 64 B per 128-B request for wide reads on gfx950 -> x2",
        "kernel,counter,avg_per_dispatch"]
for tag, d in list(a.items()) + [("sampler_scale:" + k, v) for k, v in b.items()]:
    for c, v in sorted(d.items()):
        rows.append("%s,%s,%.6g" % (tag, c, v))
open(O + "/pmc_summary.csv", "w").write("\n".join(rows) + "\n")
out = {"source": "profiles/r%s_pmc_bench.csv" % NN, "commit": os.environ.get("ISDF_COMMIT"),      # the tree these passes ran on (no .git on the GPU box: passed in)
       "note": "FETCH_SIZE*1024*2 (gfx950 wide-read correction, MI355X_MICROARCH.md HBM section) + WRITE_SIZE*1024, per dispatch"}
for tag, d in a.items():
    if "FETCH_SIZE" in d and "WRITE_SIZE" in d:
        out[tag] = {"fetch_bytes_x2": d["FETCH_SIZE"] * 2048, "write_bytes": d["WRITE_SIZE"] * 1024,
                    "hbm_bytes": d["FETCH_SIZE"] * 2048 + d["WRITE_SIZE"] * 1024}
d = b.get("sample_rays_kernel", {})
if "FETCH_SIZE" in d and "WRITE_SIZE" in d:
    out["sampler_scale"] = {"rays": 1000000, "fetch_bytes_raw": d["FETCH_SIZE"] * 1024, "fetch_bytes_x2": d["FETCH_SIZE"] * 2048,
                            "write_bytes": d["WRITE_SIZE"] * 1024, "hbm_bytes": d["FETCH_SIZE"] * 2048 + d["WRITE_SIZE"] * 1024}
json.dump(out, open(O + "/hbm_traffic.json", "w"), indent=1)
print(open(O + "/pmc_summary.csv").read())
for f in sorted(glob.glob(O + "/*.json")):
    try: j = json.load(open(f))
    except Exception as e: print(f, "ERR", e); continue
    if "value" in j: print(os.path.basename(f), j["value"], j.get("ms_per_step"), j.get("kernel_ms"), (j.get("synchronised_step") or {}).get("ms_per_step"), (j.get("roofline") or {}).get("frac"))
for f in glob.glob(O + "/stats*/**/*kernel_stats.csv", recursive=True):
    print(f)
    for r in list(csv.DictReader(open(f)))[:6]: print("  ", r["Name"][:80], r["Calls"], r["AverageNs"])
PY
ls $O
