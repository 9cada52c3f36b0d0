#!/bin/bash
# The round-end sequence the driver runs, on the GPU box from the repo root (one gpurun call):  bash tools/final_check.sh r06
# Every step is joined with && -- a step that fails or is killed at its limit ends the call, no further GPU step behind it.
R=${1:-r06}
mkdir -p gpurun_out/$R &&
timeout -k 10 850 python -m pytest tests -m gpu -q -x --durations=8 > gpurun_out/$R/final_pytest.log 2>&1 && tail -2 gpurun_out/$R/final_pytest.log &&
timeout -k 10 120 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 &&
timeout -k 10 300 python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/$R/final_bench.json 2> gpurun_out/$R/final_bench.err &&
python3 -c "
import json; o = json.loads(open('gpurun_out/$R/final_bench.json').read().strip().splitlines()[-1])
print(o['value'], o['unit'], o['ms_per_step'], 'roofline', o['roofline']['frac'], 'traffic', o['roofline']['traffic'], 'sub-record errors', o['config'].get('sub_record_errors'))"
