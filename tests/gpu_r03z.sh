#!/bin/bash
# round-3 final measurements: GPU suite, bench line, rocprofv3 kernel stats + FETCH / WRITE passes, SQ / TA counters of the cbox and
# sponza render kernels, every configuration at its own size, Poisson timing. Summaries are copied to profiles/r03_* afterwards.
set -o pipefail
mkdir -p gpurun_out
python3 -m pytest tests -x -q -m gpu > gpurun_out/r03z_gputests.log 2>&1 || { tail -40 gpurun_out/r03z_gputests.log; exit 1; }
tail -3 gpurun_out/r03z_gputests.log
python3 bench.py > gpurun_out/r03z_bench.json 2> gpurun_out/r03z_bench.err || { tail -20 gpurun_out/r03z_bench.err; exit 1; }
python3 -c "
import json; d=json.loads(open('gpurun_out/r03z_bench.json').read().strip().splitlines()[-1])
print({k: d[k] for k in ('value','ms_per_step','render_ms','poisson_ms')}); r=d['roofline']; print({k: r.get(k) for k in ('achieved','frac','lane_util','valu_busy_share_of_launch','issue_slot_frac','launch_us_rocprof','launch_us_first','traffic')}); print(d['cpu_baseline']['value'], d['cpu_baseline']['oracle_build'])"
bash profiles/collect.sh r03_z > gpurun_out/r03z_collect.log 2>&1 || { tail -20 gpurun_out/r03z_collect.log; exit 1; }
tail -3 gpurun_out/r03z_collect.log
SQ="SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU"
TA="TA_TA_BUSY_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum"
TCC="TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum"
COUNT=0 bash profiles/pmc.sh r03z_cbox tests/prof_cbox.py "$SQ" "$TA" "$TCC" "SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU" > gpurun_out/r03z_pmc_cbox.log 2>&1 || { tail gpurun_out/r03z_pmc_cbox.log; exit 1; }
bash profiles/pmc.sh r03z_sponza tests/prof_sponza.py "$SQ" "$TA" "$TCC" "SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU" > gpurun_out/r03z_pmc_sponza.log 2>&1 || { tail gpurun_out/r03z_pmc_sponza.log; exit 1; }
python3 tests/time_configs.py > gpurun_out/r03z_time_configs.txt 2>&1 || { tail -20 gpurun_out/r03z_time_configs.txt; exit 1; }
grep -v Warning gpurun_out/r03z_time_configs.txt
python3 tests/time_poisson.py > gpurun_out/r03z_time_poisson.txt 2>&1; grep -v amdgpu.ids gpurun_out/r03z_time_poisson.txt
python3 tests/prof_stamps.py > gpurun_out/r03z_stamps.txt 2>&1; head -12 gpurun_out/r03z_stamps.txt
