#!/bin/bash
# tools/profile_round.sh <tag>: on the GPU box -- kernel trace + stats of the bench command, then FETCH_SIZE and WRITE_SIZE
# in passes of their own (MI355X_MICROARCH.md, HBM section).  Outputs under gpurun_out/<tag>_*; tools/profile_collect.py
# turns them into profiles/<tag>_* and profiles/traffic.json here.
set -o pipefail
tag=${1:-r03}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/${tag}_stats gpurun_out/${tag}_fetch gpurun_out/${tag}_write
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${tag}_stats -o ${tag} -- python3 bench.py --steps 100 --warmup 5 --repetitions 3 --no-cpu-baseline --no-extras > gpurun_out/${tag}_stats.log 2>&1 || exit 1
# the counter runs start where the bench's timed windows start: behind 1000 untimed steps (the sketch's hot set is built,
# the library's own rule picks the form of the update); profile_collect.py averages the last 100 launches of each kernel
timeout -s KILL 400 rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/${tag}_fetch -o ${tag} -- python3 bench.py --steps 6 --warmup 2 --preheat 1000 --repetitions 1 --no-cpu-baseline --no-extras > gpurun_out/${tag}_fetch.log 2>&1 || exit 1
timeout -s KILL 400 rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/${tag}_write -o ${tag} -- python3 bench.py --steps 6 --warmup 2 --preheat 1000 --repetitions 1 --no-cpu-baseline --no-extras > gpurun_out/${tag}_write.log 2>&1 || exit 1
find gpurun_out/${tag}_stats gpurun_out/${tag}_fetch gpurun_out/${tag}_write -type f | head -20
echo PROFILE_OK
