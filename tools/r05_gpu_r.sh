#!/bin/bash
# round 5, call R: the tile-count threshold of the planes form in the bench
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export PYTHONPATH=$GRAFT_REPO_ROOT
O=gpurun_out/r05r; mkdir -p $O
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
run_bench default160
LLMC_GEMM3S_MIN_TILES=48 run_bench min48
LLMC_GEMM3S_MIN_TILES=600 run_bench min600
LLMC_GEMM3S_MIN_TILES=1500 run_bench min1500
LLMC_K3_NO_PLANES=1 run_bench noplanes
run_bench default160_again
