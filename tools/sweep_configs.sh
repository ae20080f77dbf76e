#!/bin/bash
# one bench line per configuration of BASELINE.json (and two larger batches of cfg3), no CPU leg -> gpurun_out/sweep_<tag>.jsonl
tag=${1:-r03}
cd $GRAFT_REPO_ROOT
out=gpurun_out/sweep_$tag.jsonl
: > $out
for args in "--config cfg3" "--config cfg3 --batch 16384" "--config cfg3 --batch 65536 --steps 50 --repetitions 5" "--config cfg4 --preheat 300" "--config cfg2 --preheat 300" "--config cfg5 --preheat 300" "--config cfg5 --preheat 300 --one-call-per-step" "--config cfg1 --preheat 300"; do
  timeout -k 10 400 python -u bench.py --no-cpu-baseline --no-extras $args 2>/dev/null | tail -1 >> $out || exit 1
done
python - <<PY
import json
for ln in open("$out"):
    j = json.loads(ln)
    ch = j["roofline"]["insert_chain"]
    ids = ch["algorithmic_bytes_per_launch"] / 104.0          # 104 B per id
    print(j["config"]["workload"][:60], "|", j["config"]["sketch_calls"][:22], "|", round(j["value"] / 1e6, 1), "M/s", j["ms_per_step"], "ms", j["roofline"]["whole_step_frac"],
          "| chain", ch["avg_launch_us"], "us =", round(ch["avg_launch_us"] * 212992.0 / ids, 1), "us per 213k ids |",
          "inorder", j["sketch"]["inorder_calls"], j["sketch"]["left_parallel_path"], "inner", j["sketch"]["inner_reset_rejects"], "rounds", round(j["sketch"]["rounds_total"] / max(1, j["sketch"]["insert_calls"]), 2))
PY
