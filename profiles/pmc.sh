#!/bin/bash
# Runs on the GPU box from the repo root: one rocprofv3 --pmc pass per counter group (kernel-trace only, as gpurun
# requires) around a python script. Usage: bash profiles/pmc.sh <tag> <script.py> "<CTR CTR ...>" ["<CTR ...>" ...]
set -e
TAG=$1; SCRIPT=$2; shift 2
ROOT=$(pwd); OUT=$ROOT/gpurun_out; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "$@"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $grp -d $OUT/${TAG}_pmc$i -o run --output-format csv -- python3 $ROOT/$SCRIPT > $OUT/${TAG}_pmc$i.log 2>&1
done
cd $ROOT
python3 profiles/pmc_summary.py $TAG
