#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r02g
mkdir -p $O
run() { # name, env, args
  echo "== $1"; ( export $2; timeout 90 python bench.py --steps 3 --warmup 1 --no-cpu-baseline $3 > $O/$1.json 2> $O/$1.err; echo "rc=$?" )
  python - <<PY
import json
try:
    j=json.loads(open('$O/$1.json').read()); print('$1', round(j['value'],2), 'layers/s', round(j['ms_per_step'],2), 'ms/step frac', round(j['roofline']['frac'],3))
except Exception as e: print('$1 no result', e)
PY
}
run ov4_side1 X=1 "--overlap 4"
run ov3_side1 X=1 "--overlap 3"
run ov2_side1 X=1 "--overlap 2"
run ov4_side1_b X=1 "--overlap 4"
