#!/bin/bash
# round 5, call G: exact diagonal (test, cost, parity envelope with both arms), the default bench line end to end
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export PYTHONPATH=$GRAFT_REPO_ROOT
O=gpurun_out/r05g; mkdir -p $O
( time timeout 600 python -m pytest tests/test_hessian_gpu.py -m gpu -q -p no:cacheprovider ) > $O/tests.log 2>&1
tail -6 $O/tests.log
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
run_bench plain
run_bench exactdiag --exact-diag 1
run_bench plain_again
( time timeout 1500 python tools/parity_envelope.py --full-down --out $O/parity_envelope_full_down ) > $O/envelope.log 2>&1; tail -5 $O/envelope.log
grep -E "^pair|vs|diag\(H\)|scales within" $O/parity_envelope_full_down.txt | head -60
( time timeout 1200 python bench.py ) > $O/bench_full.json 2> $O/bench_full.err; tail -3 $O/bench_full.err
python - <<PY
import json
try:
    j = json.loads(open('$O/bench_full.json').read().strip().splitlines()[0])
    print('bench', j['value'], j['ms_per_step'], j['ms_per_step_median'], j['roofline']['frac'], j.get('cpu_baseline', {}).get('value'), j.get('cpu_baseline', {}).get('cores'), j.get('cpu_baseline', {}).get('hessian_gflops'))
    for k, v in j.get('extra', {}).items(): print(' ', k, v.get('value'), v.get('ms_per_step'), (v.get('roofline') or {}).get('frac'), v.get('error'))
except Exception as e: print('bench failed', e); print(open('$O/bench_full.err').read()[-1500:])
PY
