#!/bin/bash
# Run on the GPU box (through gpurun) from the repository root: the counter passes first (each on its own: --pmc never together
# with other trace domains), folded into profiles/<tag>_pmc_* on the box so that the bench lines that follow carry traffic and SQ
# figures measured on the very sources they run; then the bench line, the one-lane line and the rocprofv3 kernel statistics of the
# default command.  Everything lands in gpurun_out/prof_<tag>/; afterwards, anywhere:
#   python tools/fold_profiles.py gpurun_out/prof_<tag> <tag>
set -e -o pipefail
tag=${1:-r03_x}
out=gpurun_out/prof_$tag
rm -rf $out && mkdir -p $out
export TMPDIR=/tmp
one="python3 bench.py --lanes 1 --steps 1 --warmup 0 --cpu-sample 0 --holdout 0"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $out/fetch -- $one > $out/fetch.log 2>&1
echo "fetch done"
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $out/write -- $one > $out/write.log 2>&1
echo "write done"
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_BUSY_CYCLES --output-format csv -d $out/sq -- $one > $out/sq.log 2>&1
echo "sq done"
python3 tools/fold_profiles.py $out $tag > $out/fold.log 2>&1
python3 bench.py > $out/bench.json 2> $out/bench.err
echo "bench done"; tail -c 300 $out/bench.json; echo
python3 bench.py --lanes 1 > $out/bench_1lane.json 2> $out/bench_1lane.err
echo "1-lane bench done"
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -- python3 bench.py --cpu-sample 0 --holdout 0 > $out/stats.log 2>&1
echo "stats done"
# keep what is merged back small: the per-dispatch traces are large, the folded tables are what gets committed
find $out -name "*kernel_trace.csv" -size +20M -delete || true
du -sh $out
