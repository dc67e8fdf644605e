#!/bin/bash
# round 5, call F: the three short chains on fewer streams, with / without internal helpers for down_proj's chain
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export PYTHONPATH=$GRAFT_REPO_ROOT
O=gpurun_out/r05f; mkdir -p $O
run_bench() {  # name, extra args
  n=$1; shift
  timeout 400 python bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-extras "$@" > $O/bench_$n.json 2> $O/bench_$n.err
  python - <<PY
import json
try:
    d = json.loads(open('$O/bench_$n.json').read().strip().splitlines()[-1])
    print('bench $n: %.2f layers/s  %.2f ms/step (median %.2f)  k_syrk4 %.3f of peak' % (d['value'], d['ms_per_step'], d['ms_per_step_median'], d['roofline']['frac']))
except Exception as e:
    print('bench $n failed', e); print(open('$O/bench_$n.err').read()[-800:])
PY
}
run_bench default
run_bench small1 --small-streams 1
run_bench small2 --small-streams 2
run_bench small3 --small-streams 3
run_bench small1_wide --small-streams 1 --helpers wide
run_bench small2_wide --small-streams 2 --helpers wide
run_bench default_again
timeout 400 rocprofv3 --kernel-trace --output-format csv -d $O/kt -o kt -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras --small-streams 1 > $O/kt.log 2>&1
F=$(ls $O/kt/*/*kernel_trace.csv $O/kt/*kernel_trace.csv 2>/dev/null | head -1)
python tools/step_timeline.py $F > $O/step_timeline_small1.txt 2>&1; head -16 $O/step_timeline_small1.txt
rm -rf $O/kt
