#!/bin/bash
# round-3 GPU session N: spatial-split BVH build as the default for meshes walked from HBM
set -o pipefail
mkdir -p gpurun_out
python3 -m pytest tests/test_gpu_render_parity.py tests/test_gpu_bvh8_variant.py tests/test_path_integrator.py tests/test_gpu_configs.py -x -q -m gpu > gpurun_out/r03n_gputests.log 2>&1 || { tail -40 gpurun_out/r03n_gputests.log; exit 1; }
tail -3 gpurun_out/r03n_gputests.log
python3 tests/time_configs.py > gpurun_out/r03n_time_configs.txt 2>&1 || { tail -20 gpurun_out/r03n_time_configs.txt; exit 1; }
cat gpurun_out/r03n_time_configs.txt
python3 tests/ab_upload_knob.py sbvh 1.0 > gpurun_out/r03n_ab_sbvh.txt 2>&1 || { tail -20 gpurun_out/r03n_ab_sbvh.txt; exit 1; }
cat gpurun_out/r03n_ab_sbvh.txt
