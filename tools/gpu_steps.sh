#!/bin/bash
# usage: tools/gpu_steps.sh OUTDIR 'cmd1' 'cmd2' ...   -- run GPU steps in order under gpurun; a step that was killed / timed out / died of a signal
# (exit >= 124) stops the sequence (no further GPU step after a possible fault); ordinary failures (pytest's 1) do not.
out=$1; shift
mkdir -p "$out"
i=0
for cmd in "$@"; do
    i=$((i + 1))
    echo "=== step $i: $cmd" | tee -a "$out/steps.log"
    bash -c "$cmd" > "$out/step$i.log" 2>&1
    rc=$?
    echo "=== step $i rc=$rc" | tee -a "$out/steps.log"
    tail -n 6 "$out/step$i.log"
    if [ $rc -ge 124 ]; then echo "stopping: step $i ended with $rc" | tee -a "$out/steps.log"; exit $rc; fi
done
exit 0
