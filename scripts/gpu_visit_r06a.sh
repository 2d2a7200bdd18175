#!/bin/bash
# Round-5 visit A: tile-stage launch forms (capacity-sized launches, persistent launch) -- correctness first, then same-box A/B;
# golden worst cases; the lidar-initialised scene.     usage (through gpurun): scripts/gpu_visit_r06a.sh [tag]
set -u
TAG=${1:-r06a}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/$TAG
mkdir -p $OUT
cd $REPO
export PYTHONDONTWRITEBYTECODE=1
T0=$(date +%s)
# 1. the new launch forms: bit-exact lists, overflow protocol, replayed frame
(timeout 420 python -m pytest tests/test_gpu_21_tile_stage_forms.py -x -q -p no:cacheprovider 2>&1 | tail -25) > $OUT/t21.log
tail -4 $OUT/t21.log
(timeout 600 python -m pytest tests/test_gpu_09_graph_frame.py tests/test_gpu_08_coarse_lists.py -x -q -p no:cacheprovider 2>&1 | tail -15) > $OUT/t09.log
tail -3 $OUT/t09.log
(timeout 300 python -m pytest tests/test_gpu_01_gs_parity.py -x -q -k "isect" -p no:cacheprovider 2>&1 | tail -8) > $OUT/t01_isect.log
tail -2 $OUT/t01_isect.log
echo "tests done at $(( $(date +%s) - T0 )) s"
# 2. golden worst cases (and whether north_star's own bounds hold as written)
rm -f $OUT/golden_worst.txt
(BDS_GOLDEN_WORST_LOG=$OUT/golden_worst.txt timeout 300 python -m pytest tests/test_gpu_00_bilagrid_parity.py -q -p no:cacheprovider 2>&1 | tail -4) > $OUT/t00.log
tail -2 $OUT/t00.log
(BDS_GOLDEN_STRICT=1 timeout 300 python -m pytest tests/test_gpu_00_bilagrid_parity.py -q -p no:cacheprovider 2>&1 | tail -30) > $OUT/t00_strict.log
tail -3 $OUT/t00_strict.log
echo "goldens done at $(( $(date +%s) - T0 )) s"
# 3. A/B on this box: 13 launches sized by N (round 4) | sized by the capacity | persistent launch of 64 / 128 / 256 workgroups
bench() {   # name, env...
  local name=$1; shift
  (env "$@" timeout 300 python bench.py --no-cpu-baseline --no-api-path --no-random-views --no-pair-stats --repeats 3 2>$OUT/bench_$name.stderr | tail -1) > $OUT/bench_$name.json
  python - <<PY
import json
try:
    d = json.load(open("$OUT/bench_$name.json"))
    pk = d["per_kernel"]
    print("$name:", round(d["value"], 1), "it/s | ms/step", round(d["ms_per_step"], 3), "| selfcheck", (d.get("selfcheck") or {}).get("ok"),
          "| prepare", pk.get("isect_prepare", {}).get("ms"), "build", pk.get("isect_build", {}).get("ms"))
except Exception as e:
    print("$name unreadable:", e)
PY
  grep "operator ms" $OUT/bench_$name.stderr | head -1 | cut -c1-1500
}
for round in 1 2; do
  bench base$round BDS_CAP_LAUNCH=0 BDS_TILE_PERSIST=0
  bench cap$round BDS_CAP_LAUNCH=1 BDS_TILE_PERSIST=0
  bench p128_$round BDS_TILE_PERSIST=128
  if [ $round = 1 ]; then
    bench p64 BDS_TILE_PERSIST=64
    bench p256 BDS_TILE_PERSIST=256
  fi
done
BDS_BENCH_OVERLAP_TABLE=1 bench base_tab BDS_CAP_LAUNCH=0 BDS_TILE_PERSIST=0
BDS_BENCH_OVERLAP_TABLE=1 bench cap_tab BDS_CAP_LAUNCH=1 BDS_TILE_PERSIST=0
BDS_BENCH_OVERLAP_TABLE=1 bench p128_tab BDS_TILE_PERSIST=128
echo "A/B done at $(( $(date +%s) - T0 )) s"
# 4. the lidar-initialised scene: default lists, then 16 / 32-px list tiles
lidar() {
  local name=$1; shift
  (env "$@" timeout 400 python bench.py --scene lidar ${LARGS:-} --no-cpu-baseline --no-api-path --no-random-views --repeats 3 2>$OUT/lidar_$name.stderr | tail -1) > $OUT/lidar_$name.json
  python - <<PY
import json
try:
    d = json.load(open("$OUT/lidar_$name.json"))
    c, v = d["config"], d.get("valu") or {}
    print("lidar $name:", round(d["value"], 1), "it/s | n_vis", round(c["n_visible_mean"]), "M16", round(c["isects_mean"]), "list pairs", round(c["list_pairs_mean"]),
          "| visited/listed", v.get("visited_over_listed"), "px/visited", v.get("pixels_per_visited_pair"), "| selfcheck", (d.get("selfcheck") or {}).get("ok"))
    print("   per_kernel:", {k: e["ms"] for k, e in d["per_kernel"].items()})
except Exception as e:
    print("lidar $name unreadable:", e)
PY
}
lidar lt64
lidar lt32 BDS_LIST_TILE=32
lidar lt16 BDS_LIST_TILE=16
LARGS="--lidar-opacity init" lidar init64
echo "lidar done at $(( $(date +%s) - T0 )) s"
du -sh $OUT
