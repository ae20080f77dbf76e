#!/bin/bash
# experiment: step time with the SGD beside the insert chain (timing only)
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 200 python -u bench.py --no-cpu-baseline > gpurun_out/x_base.json 2> gpurun_out/x_base.err || exit 1
CAFE_X_EARLY_SGD=1 timeout -k 10 200 python -u bench.py --no-cpu-baseline > gpurun_out/x_early.json 2> gpurun_out/x_early.err || exit 1
python - <<PY
import json
for n in ("x_base","x_early"):
    j=json.loads(open(f"gpurun_out/{n}.json").read().strip().splitlines()[-1])
    print(n, j["ms_per_step"], j["roofline"]["whole_step_frac"], j["sketch"]["inorder_calls"])
PY
