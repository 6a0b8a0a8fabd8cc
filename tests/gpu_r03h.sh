#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
for sz in "512 512" "1024 1024"; do
tag=r03h_$(echo $sz | tr ' ' 'x')
bash profiles/pmc.sh $tag "tests/prof_poisson.py $sz" "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_F64" "SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_INSTS_VALU GRBM_GUI_ACTIVE" "TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum" > gpurun_out/${tag}.log 2>&1
python3 - $tag <<'PY'
import json, sys
d = json.load(open(f"gpurun_out/{sys.argv[1]}_pmc_summary.json"))
for k, v in d.items():
    if "fold" in k:
        print(sys.argv[1], k[:50]); [print(f"   {c:34s} {x:16.1f}") for c, x in sorted(v.items())]
PY
done
