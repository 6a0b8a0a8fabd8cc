#!/bin/bash
# round-3 GPU session E: quantised 64-byte BVH4 nodes (A/B library build_q4/libgdpt_q4.so) against the fp32 nodes
set -o pipefail
mkdir -p gpurun_out
Q4=gradient-based-path-tracing_amd/csrc/build_q4/libgdpt_q4.so
python3 tests/_lib_child.py - > gpurun_out/r03e_hash_main.txt 2>&1 || { tail gpurun_out/r03e_hash_main.txt; exit 1; }
python3 tests/_lib_child.py $Q4 > gpurun_out/r03e_hash_q4.txt 2>&1 || { tail gpurun_out/r03e_hash_q4.txt; exit 1; }
python3 - <<'PY'
a = [l for l in open("gpurun_out/r03e_hash_main.txt") if l.startswith("RESULT")][0]
b = [l for l in open("gpurun_out/r03e_hash_q4.txt") if l.startswith("RESULT")][0]
print("q4 buffers and counters identical to the fp32-node build:", a == b)
PY
python3 tests/ab_lib.py $Q4 > gpurun_out/r03e_ab_q4.log 2>&1; cat gpurun_out/r03e_ab_q4.log
