#!/bin/bash
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r06c
mkdir -p $OUT
cd $REPO
export PYTHONDONTWRITEBYTECODE=1
bench() {
  local name=$1; shift
  (env "$@" BDS_BENCH_OVERLAP_TABLE=1 BDS_BENCH_NO_SELFCHECK=1 timeout 300 python bench.py --no-cpu-baseline --no-api-path --no-random-views --no-pair-stats --repeats 3 2>$OUT/bench_$name.stderr | tail -1) > $OUT/bench_$name.json
  python - <<PY
import json
try:
    d = json.load(open("$OUT/bench_$name.json"))
    print("$name:", round(d["value"], 1), "it/s")
except Exception as e:
    print("$name unreadable:", e)
PY
  grep "operator ms" $OUT/bench_$name.stderr | head -1 | cut -c1-900
}
bench base
bench padb9 BDS_PAD_BWD_KB=9
bench padb12 BDS_PAD_BWD_KB=12
bench padf6 BDS_PAD_FWD_KB=6
bench padb9f6 BDS_PAD_BWD_KB=9 BDS_PAD_FWD_KB=6
bench base2
