#!/bin/bash
# round-3 evidence: full GPU tests (-x like the driver), smoke, the default bench line, kernel stats, stage times
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r03final; mkdir -p $O
rm -f $O/actuals.jsonl
LLMC_TEST_ACTUALS=$PWD/$O/actuals.jsonl timeout 900 python -m pytest tests -x -q -m gpu > $O/tests.log 2>&1; tail -6 $O/tests.log
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
timeout 900 python bench.py --steps 10 --warmup 3 > $O/bench.json 2> $O/bench.err; python - <<PY
import json
try:
    j=json.load(open('$O/bench.json')); print(j['value'], j['ms_per_step'], j['roofline']['frac'], j.get('cpu_baseline',{}).get('value'))
    for k,v in j.get('extra',{}).items(): print(k, v.get('value'), v.get('ms_per_step'), (v.get('roofline') or {}).get('frac'), v.get('error'))
except Exception as e: print('bench failed', e); print(open('$O/bench.err').read()[-1500:])
PY
timeout 100 python bench.py --steps 2 --warmup 1 --no-extras --no-cpu-baseline --mode handoff > $O/bench_handoff_1gpu.json 2>&1; tail -c 300 $O/bench_handoff_1gpu.json
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -o kt -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras > $O/kt.log 2>&1
python tools/kernel_stats_csv.py $O/kt/kt_kernel_trace.csv 32 > $O/kernel_stats.txt 2>&1; rm -rf $O/kt; head -8 $O/kernel_stats.txt
timeout 200 python tools/bench_stages.py > $O/stage_times.txt 2>&1; tail -4 $O/stage_times.txt
timeout 200 python tools/bench_fp8_block.py --more > $O/fp8_block_rates.txt 2>&1; tail -3 $O/fp8_block_rates.txt
