#!/bin/bash
# Run on the GPU box (through gpurun): bench line + rocprofv3 kernel stats + HBM counters.
# usage: scripts/profile_round.sh <tag>      (writes gpurun_out/<tag>/...)
set -u
TAG=${1:-r01}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/$TAG
mkdir -p $OUT
cd $REPO
(timeout 500 python bench.py 2>$OUT/bench.stderr | tail -1) > $OUT/bench.json
cat $OUT/bench.json
cd /tmp && export TMPDIR=/tmp
# 1) kernel trace + stats (CSV)
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o bench -- \
    python $REPO/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $OUT/trace_bench.json 2>$OUT/trace.stderr
# 2) HBM traffic counters, separate passes (FETCH_SIZE and WRITE_SIZE do not fit one pass)
timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o bench -- \
    python $REPO/bench.py --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2>$OUT/pmc_fetch.stderr
timeout 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o bench -- \
    python $REPO/bench.py --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2>$OUT/pmc_write.stderr
find $OUT -name "*.csv" | head -20
du -sh $OUT
