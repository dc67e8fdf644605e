#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r02f
mkdir -p $O
timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-overlap > $O/bench_noov.json 2> $O/bench_noov.err
cat $O/bench_noov.json | python -c "import json,sys; j=json.loads(sys.stdin.read()); print('no-overlap', j['value'], j['ms_per_step'], j['roofline']['frac'], j['roofline']['avg_fixup_ms'])"
timeout 600 python bench.py --steps 3 --warmup 1 > $O/bench.json 2> $O/bench.err
cat $O/bench.json | python -c "import json,sys; j=json.loads(sys.stdin.read()); print('overlap', j['value'], j['ms_per_step'], j['roofline']['frac'], j.get('cpu_baseline'))"
tail -3 $O/bench.err
timeout 600 python -m pytest tests/test_e2e_gpu.py tests/test_gptq_gpu.py -m gpu -x -q > $O/pytest.log 2>&1; tail -3 $O/pytest.log
