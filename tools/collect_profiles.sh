#!/bin/bash
# Collect the rocprofv3 summaries committed under profiles/ (run on the GPU box through gpurun; writes gpurun_out/prof_TAG/,
# keeping only the per-kernel statistics and an aggregate of the counters -- the raw traces are too large to bring back).
# usage: tools/collect_profiles.sh TAG      e.g. r02_v1
# Copy afterwards:  gpurun_out/prof_TAG/*  ->  profiles/TAG_*   (bench.py reads profiles/r*_{fwd,bwd}_pmc_{fp32,bf16x3}.json for roofline.traffic)
tag=${1:-run}
repo=$(pwd)
export TMPDIR=/tmp
out=$repo/gpurun_out/prof_$tag
raw=/tmp/prof_raw_$tag
mkdir -p $out $raw
# 1. the bench command itself, headline launches only: per-kernel durations must agree with bench.py's roofline.kernel_ms
for prec in fp32 bf16x3; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $raw/bench_$prec -o bench -- python3 bench.py --steps 100 --warmup 10 --no-cpu-baseline --headline-only --precision $prec > $out/bench_${prec}.json 2> $out/bench_${prec}.err
done
# 2. the optimiser's inner iteration: forward with ReLU bits + backward
for prec in fp32 bf16x3; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $raw/fwd_bwd_$prec -o fwd_bwd -- python3 tools/prof_fwd.py $prec 20 bwd > $out/fwd_bwd_${prec}.log 2>&1
done
# 3. the one-object fused optimise loop (how many launches an iteration is, and what they cost)
rocprofv3 --kernel-trace --stats --output-format csv -d $raw/loop -o loop -- python3 tools/prof_loop.py 30 > $out/loop.log 2>&1
# 2b. family B (NeRFRenderer.render_rays, SNR_Z_BOX): the same kernels with the box bounds / per-ray depths / Philox jitter in the prologue
for prec in fp32 bf16x3; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $raw/box_fwd_$prec -o box_fwd -- python3 tools/prof_fwd.py $prec 20 fwd box > $out/box_fwd_${prec}.log 2>&1
  rocprofv3 --kernel-trace --stats --output-format csv -d $raw/box_fwd_bwd_$prec -o box_fwd_bwd -- python3 tools/prof_fwd.py $prec 20 bwd box > $out/box_fwd_bwd_${prec}.log 2>&1
done
# 3b. the training step (BASELINE config 5's per-GPU shape): chains with dumps + the weight-gradient kernels, both arithmetics
rocprofv3 --kernel-trace --stats --output-format csv -d $raw/train -o train -- python3 tools/train_bench.py 8 bf16x3 > $out/train.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $raw/train_fp32 -o train_fp32 -- python3 tools/train_bench.py 8 fp32 > $out/train_fp32.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $raw/train_auto -o train_auto -- python3 tools/train_bench.py 8 auto > $out/train_auto.log 2>&1
# 3b'. BASELINE config 3's iteration: 64 objects x 4096 rays x 64 samples per launch (what the 1 -> 8 GPU curve shards)
rocprofv3 --kernel-trace --stats --output-format csv -d $raw/c3 -o c3 -- python3 tools/prof_c3.py 64 8 > $out/c3.log 2>&1
# 3b". the weight-gradient products alone (256 x 256 layer, 524 288 points): per-kernel time, then matrix-pipe and clock counters
for prec in fp32 bf16x3; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $raw/wgrad_$prec -o wgrad_$prec -- python3 tools/prof_wgrad.py $prec 8 > $out/wgrad_${prec}.log 2>&1
  rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE -d $raw/wgradpmc_$prec -o pmc -- python3 tools/prof_wgrad.py $prec 8 > $out/wgrad_pmc_${prec}.log 2>&1
done
# 3c. the HBM-bound stand-alone kernels (encode with PE output, composite forward, scene composite)
rocprofv3 --kernel-trace --stats --output-format csv -d $raw/hbm -o hbm -- python3 tools/prof_hbm.py > $out/hbm.log 2>&1
# 4. counters of the dominant kernels, separate passes (no trace domains mixed in).  "fwd" = the headline forward (no ReLU bits saved),
#    "bwd" = the optimiser's pair (forward that saves the bits + backward); the aggregation below takes the forward kernel's counters from
#    the first and the backward kernel's from the second
for prec in fp32 bf16x3; do
  for which in fwd bwd; do
    mode=""; [ $which = bwd ] && mode="bwd"
    rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_LDS -d $raw/pmc_sq_${prec}_$which -o pmc -- python3 tools/prof_fwd.py $prec 6 $mode > $out/pmc_sq_${prec}_$which.log 2>&1
    rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE -d $raw/pmc_fetch_${prec}_$which -o pmc -- python3 tools/prof_fwd.py $prec 6 $mode > $out/pmc_fetch_${prec}_$which.log 2>&1
    rocprofv3 --kernel-trace --output-format csv --pmc WRITE_SIZE -d $raw/pmc_write_${prec}_$which -o pmc -- python3 tools/prof_fwd.py $prec 6 $mode > $out/pmc_write_${prec}_$which.log 2>&1
  done
done
find $raw -name "*kernel_stats.csv" | while read f; do cp "$f" $out/$(basename $(dirname "$f"))_$(basename "$f"); done
find $raw -type f | head -60 > $out/raw_files.txt
python3 - "$raw" "$out" <<'PY'
import csv, glob, json, os, sys
raw, out = sys.argv[1], sys.argv[2]
for prec, fwd_name, bwd_name in (("fp32", "decoder_fwd", "decoder_bwd"), ("bf16x3", "bf16_fwd_kernel", "bf16_bwd")):
    for which, kname in (("fwd", fwd_name), ("bwd", bwd_name)):
        agg = {}
        for f in glob.glob(os.path.join(raw, f"pmc_*_{prec}_{which}", "**", "*counter_collection.csv"), recursive=True):
            for row in csv.DictReader(open(f)):
                if kname not in row.get("Kernel_Name", ""):
                    continue
                c = row["Counter_Name"]
                a = agg.setdefault(c, {"sum": 0.0, "dispatches": set(), "name": row["Kernel_Name"]})
                a["sum"] += float(row["Counter_Value"]); a["dispatches"].add(row["Dispatch_Id"])
        if not agg:
            continue
        res = {c: {"per_dispatch_mean": a["sum"] / max(len(a["dispatches"]), 1), "dispatches": len(a["dispatches"])} for c, a in agg.items()}
        name = next(iter(agg.values()))["name"]
        json.dump({"kernel": name[:160] + ", 4096 rays x 64 samples", "precision": prec, "counters": res,
                   "units": "FETCH_SIZE / WRITE_SIZE in KiB as rocprofv3 reports them (FETCH_SIZE x2 on gfx950, MI355X_MICROARCH.md)"},
                  open(os.path.join(out, f"{which}_pmc_{prec}.json"), "w"), indent=1)
        print(prec, which, json.dumps(res))
# the weight-gradient products: counters per dispatch + duration -> matrix-pipe busy fraction and effective clock
for prec, kname in (("fp32", "wgrad_mfma_kernel"), ("bf16x3", "wgrad_bf16x3_kernel")):
    agg, dur = {}, []
    for f in glob.glob(os.path.join(raw, f"wgradpmc_{prec}", "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            if kname in row.get("Kernel_Name", ""):
                agg.setdefault(row["Counter_Name"], []).append(float(row["Counter_Value"]))
    for f in glob.glob(os.path.join(raw, f"wgradpmc_{prec}", "**", "*kernel_trace.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            if kname in row.get("Kernel_Name", ""):
                dur.append((int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e3)
    if agg and dur:
        m = {c: sum(v) / len(v) for c, v in agg.items()}
        us = sum(dur) / len(dur)
        json.dump({"kernel": kname + ", 256 x 256 layer, 524 288 points", "mean_us_under_counters": us, "counters_per_dispatch": m,
                   "mfma_busy_frac": m.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / (4 * m["SQ_WAVE_CYCLES"]) if m.get("SQ_WAVE_CYCLES") else None,
                   "effective_clock_ghz": m.get("GRBM_GUI_ACTIVE", 0) / 8 / us / 1e3 if m.get("GRBM_GUI_ACTIVE") else None},
                  open(os.path.join(out, f"wgrad_pmc_{prec}.json"), "w"), indent=1)
PY
ls -la $out
