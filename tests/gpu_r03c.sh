#!/bin/bash
# round-3 GPU session C: whole GPU suite on the new build, band costs, bench, TA counters of the lane machine
set -o pipefail
mkdir -p gpurun_out
python3 -m pytest tests -x -q -m gpu > gpurun_out/r03c_gputests.log 2>&1 || { tail -40 gpurun_out/r03c_gputests.log; exit 1; }
tail -3 gpurun_out/r03c_gputests.log
python3 tests/time_bands.py > gpurun_out/r03c_band_costs.txt 2>&1 || { tail -20 gpurun_out/r03c_band_costs.txt; exit 1; }
cat gpurun_out/r03c_band_costs.txt
python3 bench.py > gpurun_out/r03c_bench.json 2> gpurun_out/r03c_bench.err || { tail -20 gpurun_out/r03c_bench.err; exit 1; }
python3 -c "
import json; d=json.loads(open('gpurun_out/r03c_bench.json').read().strip().splitlines()[-1])
print({k: d[k] for k in ('value','ms_per_step','render_ms','poisson_ms')}); print(d['roofline']); print(d['cpu_baseline'])"
bash profiles/pmc.sh r03c_lane "tests/prof_wf_once.py sponza 8 wavefront=0" "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum" "TA_BUSY_avr TA_TA_BUSY_sum TCP_PENDING_STALL_CYCLES_sum GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU" > gpurun_out/r03c_pmc.log 2>&1; tail -3 gpurun_out/r03c_pmc.log
python3 - <<'PY'
import json
d = json.load(open("gpurun_out/r03c_lane_pmc_summary.json"))
for k, v in d.items():
    print(k); [print(f"   {c:34s} {x:18.1f}") for c, x in sorted(v.items())]
PY
