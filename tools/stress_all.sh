#!/bin/bash
# Every randomised differential campaign against the oracle with one seed:   gpurun -- bash tools/stress_all.sh <seed> <tag>
set -u
SEED="${1:?usage: stress_all.sh <seed> <tag>}"; TAG="${2:?usage: stress_all.sh <seed> <tag>}"
R="${GRAFT_REPO_ROOT:?GRAFT_REPO_ROOT is not set (run through gpurun)}"
O="$R/gpurun_out/$TAG"; mkdir -p "$O"; cd "$R" || exit 1
for T in meanfield iteration fit eri hot bcs gso small ham algebra eigh twins; do
  echo "== tools/${T}_stress.py (seed $SEED)"
  STRESS_SEED="$SEED" timeout 900 python "tools/${T}_stress.py" 2>&1 | grep -v "WARNING" | tail -4
done | tee "$O/stress_campaigns.txt"
