#!/bin/bash
# round 3, call G: K3 with the far panel work of every step on the helper stream: K3 / GPTQ tests, stage times, bench with and
# without the helper for the widest chain
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r03g; mkdir -p $O
timeout 300 python -m pytest tests/test_gptq_gpu.py tests/test_bench_shapes_gpu.py tests/test_awq_gpu.py -q -m gpu > $O/tests.log 2>&1; tail -5 $O/tests.log
timeout 100 python tools/bench_stages.py > $O/stage_times.txt 2>&1; tail -4 $O/stage_times.txt
LLMC_NO_SIDE_STREAM=1 timeout 100 python tools/bench_stages.py > $O/stage_times_noside.txt 2>&1; tail -1 $O/stage_times_noside.txt
timeout 200 python bench.py --steps 8 --warmup 3 --no-extras --no-cpu-baseline > $O/bench_helper.json 2> $O/bench_helper.err
timeout 200 python bench.py --steps 8 --warmup 3 --no-extras --no-cpu-baseline --wide-helper 0 > $O/bench_nohelper.json 2> $O/bench_nohelper.err
python - <<PY
import json
for n in ('helper','nohelper'):
    try:
        j=json.load(open('$O/bench_%s.json'%n)); print(n, j['value'], j['ms_per_step'], j['roofline']['frac'])
    except Exception as e: print(n,'failed',e)
PY
