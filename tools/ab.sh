#!/bin/bash
# tools/ab.sh <tag> <bench args A> -- <bench args B>: two bench lines on the same box (no CPU leg), one summary line each
set -o pipefail
tag=$1; shift
cd $GRAFT_REPO_ROOT
i=0
args=()
run() {
  # arguments of the form NAME=VALUE go to the environment of that run
  local envs=() rest=()
  for a in "${args[@]}"; do if [[ "$a" =~ ^[A-Z_0-9]+=.*$ ]]; then envs+=("$a"); else rest+=("$a"); fi; done
  env "${envs[@]}" timeout -k 10 300 python -u bench.py --no-cpu-baseline "${rest[@]}" > gpurun_out/ab_${tag}_$i.json 2> gpurun_out/ab_${tag}_$i.err || { tail -5 gpurun_out/ab_${tag}_$i.err; exit 1; }
  python - <<PY
import json
j=json.loads(open("gpurun_out/ab_${tag}_$i.json").read().strip().splitlines()[-1])
r=j["roofline"]
print("AB$i", "${args[*]}", "|", round(j["value"]/1e6,1), "M/s", j["ms_per_step"], "ms", r["whole_step_frac"], j["window_ms"], "fwd", r["other_kernel"]["avg_launch_us"] if "other_kernel" in r else None, "chain", r["insert_chain"]["avg_launch_us"], "gn", r["grad_norm"]["avg_launch_us"], j["sketch"]["left_parallel_path"], "rounds", j["sketch"]["rounds_total"])
PY
  i=$((i+1))
}
for a in "$@"; do
  if [ "$a" == "--" ]; then run; args=(); else args+=("$a"); fi
done
run
