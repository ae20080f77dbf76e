#!/bin/bash
# on the GPU box: the driver's bench command, the drifting-id variant and the configuration sweep -> gpurun_out/
tag=${1:-r03}
cd $GRAFT_REPO_ROOT
timeout -k 10 300 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/${tag}_steps20.json 2> gpurun_out/${tag}_steps20.err || exit 1
timeout -k 10 300 python bench.py --steps 20 --warmup 5 --drift 7919 --no-cpu-baseline > gpurun_out/${tag}_drift.json 2> gpurun_out/${tag}_drift.err || exit 1
bash tools/sweep_configs.sh $tag > gpurun_out/${tag}_sweep.txt 2>&1 || exit 1
echo ARTIFACTS_OK
