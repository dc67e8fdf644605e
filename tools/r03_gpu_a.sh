#!/bin/bash
# round 3, call A: GPU tests with the sample-table Hessian kernel, bench (reference calling pattern vs one tensor), the
# end-to-end parity envelope (reference on the host, reference on ROCm, ours), Triton goldens for the FP8 block ops
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r03a; mkdir -p $O
timeout 600 python -m pytest tests -x -q -m gpu > $O/tests.log 2>&1; tail -5 $O/tests.log
timeout 200 python bench.py --steps 5 --warmup 2 --no-extras --no-cpu-baseline > $O/bench_bs1.json 2> $O/bench_bs1.err
timeout 200 python bench.py --steps 5 --warmup 2 --calib-bs 128 --no-extras --no-cpu-baseline > $O/bench_bs128.json 2> $O/bench_bs128.err
python - <<PY
import json
for n in ('bs1','bs128'):
    try:
        j=json.load(open('$O/bench_%s.json'%n)); print(n, j['value'], j['ms_per_step'], j['roofline']['frac'], j['roofline']['avg_launch_ms'], j['roofline']['launches'])
    except Exception as e: print(n, 'failed', e)
PY
timeout 300 python tools/fp8_triton_golden.py $O/fp8_triton.npz > $O/fp8_triton.log 2>&1; tail -2 $O/fp8_triton.log
timeout 900 python tools/parity_envelope.py --out $O/envelope > $O/envelope.log 2>&1; tail -60 $O/envelope.log
