#!/bin/bash
# Counters of the fused fp32 forward kernels, separate passes (no trace domains mixed in): tools/pmc_fwd.sh TAG [fwd|bwd]
# Prints, per kernel name: mean per dispatch of each counter, MFMA busy fraction = SQ_VALU_MFMA_BUSY_CYCLES / (4 x SQ_WAVE_CYCLES-normalised), etc.
tag=${1:-x}; which=${2:-fwd}
export TMPDIR=/tmp
raw=/tmp/pmc_$tag
rm -rf $raw; mkdir -p $raw
mode=""; [ "$which" = bwd ] && mode="bwd"
rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_LDS -d $raw/a -o pmc -- python3 tools/prof_fwd.py fp32 6 $mode > $raw/a.log 2>&1
rocprofv3 --kernel-trace --output-format csv --pmc SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_INSTS_MFMA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE -d $raw/b -o pmc -- python3 tools/prof_fwd.py fp32 6 $mode > $raw/b.log 2>&1
python3 - "$raw" <<'PY'
import csv, glob, os, sys
raw = sys.argv[1]
agg = {}
for f in glob.glob(os.path.join(raw, "**", "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"]
        if "decoder_" not in k:
            continue
        k = k.split("(")[0][:60]
        a = agg.setdefault(k, {}).setdefault(row["Counter_Name"], [0.0, set()])
        a[0] += float(row["Counter_Value"]); a[1].add(row["Dispatch_Id"])
for k, cs in agg.items():
    m = {c: v[0] / max(len(v[1]), 1) for c, v in cs.items()}
    print(k)
    for c in sorted(m):
        print(f"   {c:28s} {m[c]:.4g}")
    if "SQ_WAVE_CYCLES" in m and "SQ_VALU_MFMA_BUSY_CYCLES" in m:
        print(f"   MFMA busy / (4 x WAVE_CYCLES quad-cycles -> cycles) = {m['SQ_VALU_MFMA_BUSY_CYCLES'] / (4 * m['SQ_WAVE_CYCLES']):.3f}  (one wave per SIMD: fraction of the pipe; two waves per SIMD: x2)")
        print(f"   MFMA busy / (4 x BUSY_CYCLES) = {m['SQ_VALU_MFMA_BUSY_CYCLES'] / (4 * m['SQ_BUSY_CYCLES']):.3f}")
PY
