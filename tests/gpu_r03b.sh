#!/bin/bash
# round-3 GPU session B: wavefront after the atomics fix: parity, A/B, kernel stats, PMC of the trace and step kernels
set -o pipefail
mkdir -p gpurun_out
python3 -m pytest tests/test_gpu_render_parity.py -x -q -k "wavefront" > gpurun_out/r03b_parity.log 2>&1 || { tail -30 gpurun_out/r03b_parity.log; exit 1; }
tail -2 gpurun_out/r03b_parity.log
python3 tests/prof_wavefront.py "sponza 1280x720x8" "sponza 1280x720x64" disney_metal > gpurun_out/r03b_ab.log 2>&1 || { tail -30 gpurun_out/r03b_ab.log; exit 1; }
cat gpurun_out/r03b_ab.log
ROOT=$(pwd); cd /tmp && export TMPDIR=/tmp
rocprofv3 -L > $ROOT/gpurun_out/r03b_counters_list.txt 2>&1
rocprofv3 --kernel-trace --stats -d $ROOT/gpurun_out/r03b_wf_stats -o run --output-format csv -- python3 $ROOT/tests/prof_wf_once.py sponza 8 > $ROOT/gpurun_out/r03b_wf_prof.log 2>&1
cd $ROOT; find gpurun_out/r03b_wf_stats -name "*kernel_stats.csv" | head -1 | xargs head -8
bash profiles/pmc.sh r03b_wf "tests/prof_wf_once.py sponza 8" "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum" "SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT" "TA_BUSY_avr TA_TA_BUSY_sum TCP_PENDING_STALL_CYCLES_sum GRBM_GUI_ACTIVE" > gpurun_out/r03b_pmc.log 2>&1; tail -5 gpurun_out/r03b_pmc.log
python3 - <<'PY'
import json
d = json.load(open("gpurun_out/r03b_wf_pmc_summary.json"))
for k, v in d.items():
    print(k); [print(f"   {c:34s} {x:18.1f}") for c, x in sorted(v.items())]
PY
