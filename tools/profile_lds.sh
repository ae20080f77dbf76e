#!/bin/bash
# tools/profile_lds.sh <tag>: LDS bank-conflict cycles against LDS-active cycles per kernel (a PMC pass of its own)
set -o pipefail
tag=${1:-r03}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/${tag}_lds
timeout -s KILL 400 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d gpurun_out/${tag}_lds -o ${tag} -- python3 bench.py --steps 6 --warmup 2 --preheat 1000 --repetitions 1 --no-cpu-baseline --no-extras > gpurun_out/${tag}_lds.log 2>&1 || exit 1
echo LDS_OK
