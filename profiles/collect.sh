#!/bin/bash
# Runs on the GPU box (via gpurun) from the repo root: kernel-trace stats + two separate PMC passes
# (FETCH_SIZE, WRITE_SIZE) around the default bench workload, as MI355X_MICROARCH.md prescribes.
# Usage: bash profiles/collect.sh <tag>   -> gpurun_out/<tag>_{stats,fetch,write}/ + gpurun_out/<tag>_bench_under_rocprof.json
set -e
TAG=${1:-r01_x}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
BENCH="$ROOT/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-pmc --no-strong"   # (the bench line printed under the profiler goes to <tag>_bench_under_rocprof.json)
rocprofv3 --kernel-trace --stats -d $OUT/${TAG}_stats -o run --output-format csv -- python3 $BENCH > $OUT/${TAG}_bench_under_rocprof.json 2> $OUT/${TAG}_stats.err
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/${TAG}_fetch -o run --output-format csv -- python3 $BENCH > /dev/null 2> $OUT/${TAG}_fetch.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/${TAG}_write -o run --output-format csv -- python3 $BENCH > /dev/null 2> $OUT/${TAG}_write.err
cd $ROOT
python3 profiles/summarize.py $TAG
