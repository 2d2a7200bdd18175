#!/bin/bash
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r06f
mkdir -p $OUT
cd $REPO
export PYTHONDONTWRITEBYTECODE=1
bench() {
  local name=$1; shift
  (env "$@" BDS_BENCH_OVERLAP_TABLE=1 timeout 300 python bench.py --no-cpu-baseline --no-api-path --no-random-views --no-pair-stats --repeats 3 2>$OUT/bench_$name.stderr | tail -1) > $OUT/bench_$name.json
  python - <<PY
import json
try:
    d = json.load(open("$OUT/bench_$name.json"))
    print("$name:", round(d["value"], 1), "it/s | selfcheck", (d.get("selfcheck") or {}).get("ok"))
except Exception as e:
    print("$name unreadable:", e)
PY
  grep "operator ms" $OUT/bench_$name.stderr | head -1 | cut -c1-900
}
bench base
bench b4 BDS_SLOTS_BWD=4
bench b5 BDS_SLOTS_BWD=5
bench b4f4 BDS_SLOTS_BWD=4 BDS_SLOTS_FWD=4
bench b4f3 BDS_SLOTS_BWD=4 BDS_SLOTS_FWD=3
bench b3f3 BDS_SLOTS_BWD=3 BDS_SLOTS_FWD=3
bench f4 BDS_SLOTS_FWD=4
bench base2
