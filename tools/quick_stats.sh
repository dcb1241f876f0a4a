#!/bin/bash
# per-kernel durations of two one-lane bench steps (rocprofv3 --kernel-trace --stats); output under gpurun_out/qstats_<tag>/
set -e -o pipefail
tag=${1:-q}
out=gpurun_out/qstats_$tag
rm -rf $out && mkdir -p $out
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out/st -- python3 bench.py --lanes 1 --steps 2 --warmup 0 --cpu-sample 0 --holdout 0 ${@:2} > $out/log 2>&1
f=$(find $out -name "*kernel_stats.csv" | head -1)
python3 - "$f" <<'P'
import csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:45]:
    name = re.sub(r"\(.*", "", re.sub(r"^void ", "", r["Name"].replace("(anonymous namespace)::", "")))
    print("%-34s calls %5s total %9.3f ms avg %8.3f ms  %5s%%" % (name[:34], r["Calls"], int(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e6, r["Percentage"]))
P
find $out -name "*kernel_trace.csv" -delete || true
