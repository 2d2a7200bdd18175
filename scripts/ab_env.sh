#!/bin/bash
# same-box A/B of an environment switch on the bench line: scripts/ab_env.sh <tag> <VAR> [bench args...]   (VAR=0 / VAR=1, three interleaved pairs)
set -u
TAG=$1; VAR=$2; shift 2
OUT=gpurun_out/$TAG; mkdir -p $OUT
Q="--no-cpu-baseline --no-pair-stats --no-api-path --no-random-views $*"
for i in 1 2 3; do for val in 0 1; do
  (env $VAR=$val python bench.py $Q 2>$OUT/${VAR}${val}_$i.err | tail -1) > $OUT/${VAR}${val}_$i.json
done; done
python - $OUT <<'PY'
import json,sys,glob
for f in sorted(glob.glob(sys.argv[1]+"/*.json")):
    try:
        d=json.load(open(f)); pk=d["per_kernel"]
        print(f.split("/")[-1][:-5], round(d["value"],1), d["selfcheck"]["ok"], {k: round(v["ms"]*1e3,1) for k,v in pk.items() if k in ("project_bwd","sh_bwd","isect_prepare","rasterize_fwd")})
    except Exception as e: print(f, "ERR", e)
PY
