#!/bin/bash
# round 5, call T: K4's far update of a group as one launch (no helper streams) vs three
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export PYTHONPATH=$GRAFT_REPO_ROOT
O=gpurun_out/r05t; mkdir -p $O
timeout 300 python -m pytest tests/test_gptq_gpu.py -m gpu -q -p no:cacheprovider -x 2>&1 | tail -3
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
LLMC_K4_SPLIT_FAR=1 run_bench three_launches
run_bench one_launch
LLMC_K4_SPLIT_FAR=1 run_bench three_launches_again
run_bench one_launch_again
