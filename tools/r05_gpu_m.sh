#!/bin/bash
# round 5, call M: runtime knobs of the HIP runtime on the default bench (kernel-argument placement, direct dispatch)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r05m; mkdir -p $O
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
run_bench default
HIP_FORCE_DEV_KERNARG=1 run_bench devkernarg1
HIP_FORCE_DEV_KERNARG=0 run_bench devkernarg0
run_bench default_again
AMD_DIRECT_DISPATCH=0 run_bench directdispatch0
HIP_FORCE_DEV_KERNARG=1 run_bench devkernarg1_again
GPU_MAX_HW_QUEUES=8 run_bench q8
HSA_ENABLE_INTERRUPT=0 run_bench nointerrupt
run_bench default_3
