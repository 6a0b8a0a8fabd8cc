#!/bin/bash
# round-3 GPU session I: non-constant texture lookups out of line in the general material switch + fused eval/pdf of the two-sided lobes
set -o pipefail
mkdir -p gpurun_out
V=gradient-based-path-tracing_amd/csrc/build_texinline/libgdpt_texinline.so
python3 tests/_lib_child.py - > gpurun_out/r03i_hash_main.txt 2>&1 || { tail gpurun_out/r03i_hash_main.txt; exit 1; }
python3 tests/_lib_child.py $V > gpurun_out/r03i_hash_var.txt 2>&1 || { tail gpurun_out/r03i_hash_var.txt; exit 1; }
python3 - <<'PY'
a = [l for l in open("gpurun_out/r03i_hash_main.txt") if l.startswith("RESULT")][0]
b = [l for l in open("gpurun_out/r03i_hash_var.txt") if l.startswith("RESULT")][0]
print("buffers and counters identical between the out-of-line and the inline texture builds:", a == b)
PY
python3 -m pytest tests/test_gpu_render_parity.py tests/test_path_integrator.py tests/test_reconnect_shift.py -x -q -m gpu > gpurun_out/r03i_gputests.log 2>&1 || { tail -40 gpurun_out/r03i_gputests.log; exit 1; }
tail -2 gpurun_out/r03i_gputests.log
python3 tests/ab_lib.py $V > gpurun_out/r03i_ab_tex.log 2>&1; cat gpurun_out/r03i_ab_tex.log
