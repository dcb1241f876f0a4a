#!/bin/bash
# quick look on the GPU box: SQ counters of one one-lane step, then a short one-lane bench; output under gpurun_out/quick_<tag>/
set -e -o pipefail
tag=${1:-q}
out=gpurun_out/quick_$tag
rm -rf $out && mkdir -p $out
export TMPDIR=/tmp
extra="${@:2}"
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_BUSY_CYCLES --output-format csv -d $out/sq -- python3 bench.py --lanes 1 --steps 1 --warmup 0 --cpu-sample 0 --holdout 0 $extra > $out/sq.log 2>&1
python3 tools/pmc_sq_summary.py $out/sq > $out/sq_summary.txt 2>&1 || true
python3 bench.py --lanes 1 --steps 5 --warmup 1 --cpu-sample 0 --holdout 0 $extra > $out/b1.json 2> $out/b1.err
python3 - <<P
import json
d=json.load(open("$out/b1.json"))
print(d["value"], d["ms_per_step"])
for k,v in d["kernels"].items(): print("  %-16s %7.3f"%(k,v["ms_per_step"]))
P
find $out -name "*kernel_trace.csv" -delete || true
find $out -name "*counter_collection.csv" -size +30M -delete || true
