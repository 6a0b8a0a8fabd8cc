#!/bin/bash
# round-3 GPU session AQ: last verification of the tree as committed — whole GPU suite, smoke(), bench line
set -o pipefail
mkdir -p gpurun_out
python3 -m pytest tests -x -q -m gpu > gpurun_out/r03aq_gputests.log 2>&1 || { tail -60 gpurun_out/r03aq_gputests.log; exit 1; }
tail -3 gpurun_out/r03aq_gputests.log
python3 -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/r03aq_smoke.log 2>&1 || { tail -20 gpurun_out/r03aq_smoke.log; exit 1; }
tail -1 gpurun_out/r03aq_smoke.log
python3 bench.py --no-cpu-baseline > gpurun_out/r03aq_bench.json 2> gpurun_out/r03aq_bench.err || { tail -20 gpurun_out/r03aq_bench.err; exit 1; }
python3 -c "
import json; d=json.loads(open('gpurun_out/r03aq_bench.json').read().strip().splitlines()[-1])
print({k: d[k] for k in ('value','ms_per_step','render_ms','poisson_ms')}); print(d['scaling_strong']['value'])"
