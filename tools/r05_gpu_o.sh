#!/bin/bash
# round 5, call O: k_gemm3s (producer / MFMA wave specialisation) vs k_gemm3 (LLMC_GEMM3_NOSPEC=1): tests, stage times, bench, PMC
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export PYTHONPATH=$GRAFT_REPO_ROOT
O=gpurun_out/r05o; mkdir -p $O
( time timeout 900 python -m pytest tests/test_gptq_gpu.py -m gpu -q -p no:cacheprovider ) > $O/tests.log 2>&1
tail -6 $O/tests.log
LLMC_GEMM3_NOSPEC=1 timeout 300 python tools/bench_stages.py > $O/stage_times_nospec.txt 2>&1; grep down $O/stage_times_nospec.txt
timeout 300 python tools/bench_stages.py > $O/stage_times_spec.txt 2>&1; grep down $O/stage_times_spec.txt
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
LLMC_GEMM3_NOSPEC=1 run_bench nospec
run_bench spec
LLMC_GEMM3_NOSPEC=1 run_bench nospec_again
run_bench spec_again





