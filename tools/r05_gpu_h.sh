#!/bin/bash
# round 5, call H: exact diagonal on a second stream under the MFMA kernel (test, cost, does it really overlap?), ADVICE-low tests
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export PYTHONPATH=$GRAFT_REPO_ROOT
O=gpurun_out/r05h; mkdir -p $O
( time timeout 900 python -m pytest tests/test_hessian_gpu.py tests/test_fp8_block_gpu.py tests/test_clip_v2.py tests/test_clip_wide_gpu.py -m gpu -q -p no:cacheprovider ) > $O/tests.log 2>&1
tail -6 $O/tests.log
run_bench() {  # name, extra args
  n=$1; shift
  timeout 400 python bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-extras "$@" > $O/bench_$n.json 2> $O/bench_$n.err
  python - <<PY
import json
try:
    d = json.loads(open('$O/bench_$n.json').read().strip().splitlines()[-1])
    print('bench $n: %.2f layers/s  %.2f ms/step (median %.2f)  k_syrk4 %.3f of peak  timeouts %s' % (d['value'], d['ms_per_step'], d['ms_per_step_median'], d['roofline']['frac'], d['roofline']['round_barrier_timeouts_last_step']))
except Exception as e:
    print('bench $n failed', e); print(open('$O/bench_$n.err').read()[-800:])
PY
}
run_bench plain
run_bench exactdiag --exact-diag 1
run_bench plain_again
run_bench exactdiag_again --exact-diag 1
timeout 400 rocprofv3 --kernel-trace --output-format csv -d $O/kt -o kt -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras --exact-diag 1 > $O/kt.log 2>&1
F=$(ls $O/kt/*/*kernel_trace.csv $O/kt/*kernel_trace.csv 2>/dev/null | head -1)
python tools/step_timeline.py $F --full > $O/step_timeline_exactdiag.txt 2>&1; head -14 $O/step_timeline_exactdiag.txt; grep -n "k_syrk4\|k_diag_sumsq\|k_diag_apply\|k_syrk_fixup" $O/step_timeline_exactdiag.txt | head -30
rm -rf $O/kt
