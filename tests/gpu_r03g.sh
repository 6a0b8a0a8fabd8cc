#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
python3 -m pytest tests/test_gpu_poisson_and_pipeline.py -x -q -m gpu > gpurun_out/r03g_gputests.log 2>&1 || { tail -40 gpurun_out/r03g_gputests.log; exit 1; }
tail -2 gpurun_out/r03g_gputests.log
python3 tests/time_poisson.py > gpurun_out/r03g_time_poisson.txt 2>&1 || { tail -20 gpurun_out/r03g_time_poisson.txt; exit 1; }
cat gpurun_out/r03g_time_poisson.txt
