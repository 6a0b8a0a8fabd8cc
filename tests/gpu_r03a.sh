#!/bin/bash
# round-3 GPU session A: wavefront v2 parity + A/B + per-kernel times
set -o pipefail
mkdir -p gpurun_out
python3 -m pytest tests/test_gpu_render_parity.py -x -q -k "wavefront" > gpurun_out/r03a_parity.log 2>&1 || { tail -30 gpurun_out/r03a_parity.log; exit 1; }
tail -3 gpurun_out/r03a_parity.log
python3 tests/prof_wavefront.py sponza disney_metal > gpurun_out/r03a_ab.log 2>&1 || { tail -30 gpurun_out/r03a_ab.log; exit 1; }
cat gpurun_out/r03a_ab.log
ROOT=$(pwd); cd /tmp && export TMPDIR=/tmp
WF_ONLY=octant-major rocprofv3 --kernel-trace --stats -d $ROOT/gpurun_out/r03a_wf_stats -o run --output-format csv -- python3 $ROOT/tests/prof_wavefront.py "sponza 1280x720x8" > $ROOT/gpurun_out/r03a_wf_prof.log 2>&1
cd $ROOT; find gpurun_out/r03a_wf_stats -name "*kernel_stats.csv" | head -1 | xargs head -12
