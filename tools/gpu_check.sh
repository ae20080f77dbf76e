#!/bin/bash
# tools/gpu_check.sh <tag>: GPU tests, bench line and kernel trace of the default workload, on the GPU box.
set -o pipefail
tag=${1:-x}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread > gpurun_out/t_$tag.log 2>&1
rc=$?
tail -5 gpurun_out/t_$tag.log
[ $rc -ne 0 ] && exit $rc
timeout -k 10 300 python -u bench.py --no-cpu-baseline > gpurun_out/bench_$tag.json 2> gpurun_out/bench_$tag.err || exit 1
python - <<PY
import json
j=json.loads(open("gpurun_out/bench_$tag.json").read().strip().splitlines()[-1])
print("BENCH", j["value"]/1e6, "M/s", j["ms_per_step"], "ms", j["roofline"]["whole_step_frac"], j["sketch"])
PY
rm -rf gpurun_out/prof_$tag
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_$tag -o $tag -- python3 bench.py --steps 60 --warmup 5 --no-cpu-baseline > gpurun_out/prof_$tag.log 2>&1 || exit 1
echo PROF_OK
