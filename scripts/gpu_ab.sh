#!/bin/bash
# A/B of two environments on ONE GPU box (box-to-box spread is +-3%: only same-box comparisons mean anything).
# usage: scripts/gpu_ab.sh <tag> "<env A, e.g. X=1 Y=2 or ->" "<env B>" [bench args...]   (runs A B A B)
set -u
TAG=${1:-ab}; A=${2:--}; B=${3:--}
shift 3 || true
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/$TAG
mkdir -p $OUT
cd $REPO
export PYTHONDONTWRITEBYTECODE=1
for round in 1 2; do
  for which in A B; do
    if [ $which = A ]; then E=$A; else E=$B; fi
    [ "$E" = "-" ] && E=""
    (env $E timeout 600 python bench.py --no-cpu-baseline --no-api-path --repeats 3 "$@" 2>>$OUT/bench.stderr | tail -1) > $OUT/bench_$which$round.json
    python - <<PY
import json
try:
    d = json.load(open("$OUT/bench_$which$round.json"))
    print("$which$round [$E]:", round(d["value"], 1), "it/s  ms/step", round(d["ms_per_step"], 3), "| dom ms", round(d["roofline"]["avg_launch_ms"], 4))
except Exception as e:
    print("$which$round unreadable:", e)
PY
  done
done
tail -3 $OUT/bench.stderr
