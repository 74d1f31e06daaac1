#!/bin/bash
# The evidence pass behind profiles/rNN_*: rocprofv3 trace + PMC passes of C5 (14-kL shard) and C4 (tools/prof_passes.sh), the driver's
# command line, and one bench line per BASELINE config.  Run on the GPU box:   gpurun -- bash tools/evidence_pass.sh r05
set -u
TAG="${1:?usage: evidence_pass.sh <tag>}"
R="${GRAFT_REPO_ROOT:?GRAFT_REPO_ROOT is not set (run through gpurun)}"
O="$R/gpurun_out/$TAG"
mkdir -p "$O"
bash "$R/tools/prof_passes.sh" "$TAG" > "$O/prof_passes.log" 2>&1
cd "$R" || exit 1
timeout 1700 python bench.py --steps 20 --warmup 5 > "$O/bench_default.json" 2> "$O/bench_default.err"; echo "default rc=$?"
for WL in C1 C2 C3 C4; do
  timeout 900 python bench.py --workload "$WL" > "$O/bench_$WL.json" 2> "$O/bench_$WL.err"; echo "$WL rc=$?"
done
tail -c 600 "$O/bench_default.json"
