#!/bin/bash
# round-3 GPU session F: Q4 nodes as the default, folded MFMA DCT with deeper pipelining as the default solver
set -o pipefail
mkdir -p gpurun_out
python3 -m pytest tests/test_gpu_poisson_and_pipeline.py tests/test_gpu_render_parity.py tests/test_gpu_bvh8_variant.py tests/test_path_integrator.py -x -q -m gpu > gpurun_out/r03f_gputests.log 2>&1 || { tail -40 gpurun_out/r03f_gputests.log; exit 1; }
tail -3 gpurun_out/r03f_gputests.log
python3 tests/time_poisson.py > gpurun_out/r03f_time_poisson.txt 2>&1 || { tail -20 gpurun_out/r03f_time_poisson.txt; exit 1; }
cat gpurun_out/r03f_time_poisson.txt
python3 tests/time_configs.py > gpurun_out/r03f_time_configs.txt 2>&1 || { tail -20 gpurun_out/r03f_time_configs.txt; exit 1; }
cat gpurun_out/r03f_time_configs.txt
python3 bench.py --no-cpu-baseline > gpurun_out/r03f_bench.json 2> gpurun_out/r03f_bench.err || { tail -20 gpurun_out/r03f_bench.err; exit 1; }
python3 -c "
import json; d=json.loads(open('gpurun_out/r03f_bench.json').read().strip().splitlines()[-1])
print({k: d[k] for k in ('value','ms_per_step','render_ms','poisson_ms')}); print(d['scaling_strong']); print(d['pipelined'])
for k in d['kernels'] or []: print(k['kernel'][:60], round(k['avg_us'],1), k['unit'], round(k['achieved'],1), round(k['frac'],3))"
