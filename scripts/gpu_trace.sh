#!/bin/bash
# One bench line + rocprofv3 kernel trace of the graph-replayed frame (timeline of one view + per-kernel stats).  usage: scripts/gpu_trace.sh <tag> [bench args]
set -u
TAG=${1:-r03t}; shift || true
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/$TAG
mkdir -p $OUT
cd $REPO
export PYTHONDONTWRITEBYTECODE=1
(timeout 600 python bench.py --no-cpu-baseline "$@" 2>$OUT/bench.stderr | tail -1) > $OUT/bench.json
python - <<PY
import json
try:
    d = json.load(open("$OUT/bench.json"))
    print("bench:", round(d["value"], 1), "it/s  ms/step", round(d["ms_per_step"], 3), "min/max", round(d.get("ms_per_step_min", 0), 3), round(d.get("ms_per_step_max", 0), 3),
          "| dom ms", round(d["roofline"]["avg_launch_ms"], 4))
except Exception as e:
    print("bench line unreadable:", e); print(open("$OUT/bench.stderr").read()[-1500:])
PY
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o bench -- python $REPO/bench.py --steps 6 --warmup 1 --repeats 1 --no-cpu-baseline --no-pair-stats "$@" > $OUT/trace_bench.json 2>$OUT/trace.stderr
python $REPO/scripts/step_timeline.py $OUT/trace/bench_kernel_trace.csv 20 > $OUT/step_timeline.txt 2>&1
python $REPO/scripts/frame_overlap_report.py $OUT/trace/bench_kernel_trace.csv > $OUT/frame_overlap.txt 2>&1
find $OUT -name "*kernel_trace.csv" -delete
find $OUT -type f -size +8M -delete
cat $OUT/frame_overlap.txt
