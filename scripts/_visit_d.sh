#!/bin/bash
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r06d
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
for r in 1 2; do
  bench prio0_$r BDS_LIB=$REPO/bilateral_driving_amd/libbds_prio0.so
  bench prio2_$r
done
