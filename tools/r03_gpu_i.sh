#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r03i; mkdir -p $O
timeout 300 python -m pytest tests/test_gptq_gpu.py tests/test_bench_shapes_gpu.py tests/test_spqr_gpu.py -x -q -m gpu > $O/tests.log 2>&1; tail -3 $O/tests.log
timeout 300 python bench.py --steps 10 --warmup 3 --no-extras --no-cpu-baseline > $O/bench.json 2> $O/bench.err
timeout 300 python bench.py --steps 10 --warmup 3 --no-extras --no-cpu-baseline --wide-helper 1 > $O/bench_wh.json 2> $O/bench_wh.err
python - <<PY
import json
for n in ('bench','bench_wh'):
    try:
        j=json.load(open('$O/%s.json'%n)); print(n, j['value'], j['ms_per_step'], j['roofline']['frac'], j['roofline']['avg_launch_ms'])
    except Exception as e: print(n,'failed',e)
PY
timeout 100 python tools/bench_stages.py 2>&1 | tail -1
