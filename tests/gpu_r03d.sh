#!/bin/bash
# round-3 GPU session D: folded Poisson GEMMs, plan_rows + cost-balanced bands, multi-device failure tests
set -o pipefail
mkdir -p gpurun_out
python3 -m pytest tests/test_gpu_poisson_and_pipeline.py tests/test_gpu_multi.py tests/test_gpu_configs.py -x -q -m gpu > gpurun_out/r03d_gputests.log 2>&1 || { tail -40 gpurun_out/r03d_gputests.log; exit 1; }
tail -3 gpurun_out/r03d_gputests.log
python3 tests/time_poisson.py > gpurun_out/r03d_time_poisson.txt 2>&1 || { tail -20 gpurun_out/r03d_time_poisson.txt; exit 1; }
cat gpurun_out/r03d_time_poisson.txt
python3 tests/time_bands.py > gpurun_out/r03d_band_costs.txt 2>&1 || { tail -20 gpurun_out/r03d_band_costs.txt; exit 1; }
cat gpurun_out/r03d_band_costs.txt
