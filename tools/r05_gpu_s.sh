#!/bin/bash
# round 5, call S: one far-update launch per outer block (no helper stream) vs two
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export PYTHONPATH=$GRAFT_REPO_ROOT
O=gpurun_out/r05s; mkdir -p $O
timeout 400 python -m pytest tests/test_gptq_gpu.py -m gpu -q -p no:cacheprovider -k "planes or chol" 2>&1 | tail -4
run_bench() {  # name, extra args
  n=$1; shift
  timeout 400 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras "$@" > $O/bench_$n.json 2> $O/bench_$n.err
  python - <<PY
import json
try:
    d = json.loads(open('$O/bench_$n.json').read().strip().splitlines()[-1])
    print('bench $n: %.2f layers/s  %.2f ms/step (median %.2f)  k_syrk4 %.3f of peak' % (d['value'], d['ms_per_step'], d['ms_per_step_median'], d['roofline']['frac']))
except Exception as e:
    print('bench $n failed', e); print(open('$O/bench_$n.err').read()[-800:])
PY
}
LLMC_K3_SPLIT_FAR=1 run_bench two_launches
run_bench one_launch
LLMC_K3_SPLIT_FAR=1 run_bench two_launches_again
run_bench one_launch_again
