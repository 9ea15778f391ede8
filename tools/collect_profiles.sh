#!/bin/bash
# Collect the rocprofv3 summaries committed under profiles/ (run on the GPU box through gpurun; writes gpurun_out/prof_TAG/,
# keeping only the per-kernel statistics and an aggregate of the counters -- the raw traces are too large to bring back).
# usage: tools/collect_profiles.sh TAG      e.g. r01_v3
tag=${1:-run}
repo=$(pwd)
export TMPDIR=/tmp
out=$repo/gpurun_out/prof_$tag
raw=/tmp/prof_raw_$tag
mkdir -p $out $raw
# 1. the bench command itself: per-kernel durations must agree with bench.py's roofline.kernel_ms
rocprofv3 --kernel-trace --stats --output-format csv -d $raw/bench -o bench -- python3 bench.py --steps 50 --warmup 5 --no-cpu-baseline --headline-only > $out/bench.json 2> $out/bench.err
# 2. the optimiser's inner iteration: forward with ReLU bits + backward
rocprofv3 --kernel-trace --stats --output-format csv -d $raw/fwd_bwd -o fwd_bwd -- python3 tools/prof_fwd.py bf16x3 20 bwd > $out/fwd_bwd.log 2>&1
# 3. counters of the dominant kernel, separate passes (no trace domains mixed in)
rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_LDS -d $raw/pmc_sq -o pmc -- python3 tools/prof_fwd.py bf16x3 6 > $out/pmc_sq.log 2>&1
rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE -d $raw/pmc_fetch -o pmc -- python3 tools/prof_fwd.py bf16x3 6 > $out/pmc_fetch.log 2>&1
rocprofv3 --kernel-trace --output-format csv --pmc WRITE_SIZE -d $raw/pmc_write -o pmc -- python3 tools/prof_fwd.py bf16x3 6 > $out/pmc_write.log 2>&1
find $raw -name "*kernel_stats.csv" | while read f; do cp "$f" $out/$(basename $(dirname "$f"))_$(basename "$f"); done
find $raw -type f | head -40 > $out/raw_files.txt
python3 - "$raw" "$out" <<'PY'
import csv, glob, json, os, sys
raw, out = sys.argv[1], sys.argv[2]
agg = {}
for f in glob.glob(os.path.join(raw, "pmc_*", "**", "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        k = row.get("Kernel_Name", "")
        if "bf16_fwd_kernel" not in k:
            continue
        c = row["Counter_Name"]
        a = agg.setdefault(c, {"sum": 0.0, "dispatches": set()})
        a["sum"] += float(row["Counter_Value"]); a["dispatches"].add(row["Dispatch_Id"])
res = {c: {"per_dispatch_mean": a["sum"] / max(len(a["dispatches"]), 1), "dispatches": len(a["dispatches"])} for c, a in agg.items()}
json.dump({"kernel": "snr::bf::bf16_fwd_kernel<1,false>, 4096 rays x 64 samples", "counters": res}, open(os.path.join(out, "fwd_pmc.json"), "w"), indent=1)
print(json.dumps(res, indent=1))
PY
ls -la $out
