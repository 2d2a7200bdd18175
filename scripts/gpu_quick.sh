#!/bin/bash
# A short GPU-box visit (through gpurun): selected test files, then bench lines.  usage: scripts/gpu_quick.sh <tag> "<pytest args>" [bench args...]
set -u
TAG=${1:-r03q}
TESTS=${2:-}
shift 2 || true
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/$TAG
mkdir -p $OUT
cd $REPO
export PYTHONDONTWRITEBYTECODE=1
if [ -n "$TESTS" ]; then
  (timeout 900 python -m pytest $TESTS -q -m gpu -p no:cacheprovider -x 2>&1 | tail -60) > $OUT/tests.log
  tail -25 $OUT/tests.log
fi
(timeout 600 python bench.py --no-cpu-baseline "$@" 2>$OUT/bench.stderr | tail -1) > $OUT/bench.json
tail -5 $OUT/bench.stderr
python - <<PY
import json
try:
    d = json.load(open("$OUT/bench.json"))
    print("bench:", round(d["value"], 1), "it/s  ms/step", round(d["ms_per_step"], 3), "min/max", round(d.get("ms_per_step_min", 0), 3), round(d.get("ms_per_step_max", 0), 3),
          "| dom ms", round(d["roofline"]["avg_launch_ms"], 4), "|", d["config"]["step_driver"][:40])
except Exception as e:
    print("bench line unreadable:", e)
PY
