#!/bin/bash
# round 5, call L: K4's error columns k-major (TA = true operands for the near / far updates): tests, stage times, bench A/B
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export PYTHONPATH=$GRAFT_REPO_ROOT
O=gpurun_out/r05l; mkdir -p $O
( time timeout 900 python -m pytest tests/test_gptq_gpu.py tests/test_bench_shapes_gpu.py tests/test_gptq_deploy_gpu.py tests/test_e2e_gpu.py -m gpu -q -p no:cacheprovider ) > $O/tests.log 2>&1
tail -6 $O/tests.log
LLMC_K4_ERR_ROWMAJOR=1 timeout 300 python tools/bench_stages.py > $O/stage_times_v1.txt 2>&1; cat $O/stage_times_v1.txt | grep -v amdgpu
timeout 300 python tools/bench_stages.py > $O/stage_times_v2.txt 2>&1; cat $O/stage_times_v2.txt | grep -v amdgpu
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
LLMC_K4_ERR_ROWMAJOR=1 run_bench v1
run_bench v2
LLMC_K4_ERR_ROWMAJOR=1 run_bench v1_again
run_bench v2_again
LLMC_K4_ERR_ROWMAJOR=1 run_bench v1_70b --model llama3-70b --steps 3 --warmup 1
run_bench v2_70b --model llama3-70b --steps 3 --warmup 1
