#!/bin/bash
# round-4 evidence: full GPU tests, smoke, the default bench line (with extras + the reference on the host cores), kernel
# stats, PMC traffic of the dominant kernel, stage times, AWQ kernel stats, the full-size down_proj parity envelope.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r04final2; mkdir -p $O
rm -f $O/actuals.jsonl
( time LLMC_TEST_ACTUALS=$PWD/$O/actuals.jsonl timeout 1500 python -m pytest tests -q -m gpu -p no:cacheprovider ) > $O/tests.log 2>&1; tail -8 $O/tests.log
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
timeout 1200 python bench.py --steps 10 --warmup 3 > $O/bench.json 2> $O/bench.err; python - <<PY
import json
try:
    j=json.load(open('$O/bench.json')); print('bench', j['value'], j['ms_per_step'], j['roofline']['frac'], j.get('cpu_baseline',{}).get('value'))
    for k,v in j.get('extra',{}).items(): print(' ', k, v.get('value'), v.get('ms_per_step'), (v.get('roofline') or {}).get('frac'), v.get('error'))
except Exception as e: print('bench failed', e); print(open('$O/bench.err').read()[-1500:])
PY
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -o kt -- python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-extras > $O/kt.log 2>&1
F=$(ls $O/kt/*/*kernel_trace.csv $O/kt/*kernel_trace.csv 2>/dev/null | head -1)
python tools/kernel_stats_csv.py $F 34 > $O/kernel_stats.txt 2>&1; rm -rf $O/kt; head -12 $O/kernel_stats.txt
timeout 200 python tools/bench_stages.py 2>&1 | grep -v amdgpu.ids > $O/stage_times.txt; cat $O/stage_times.txt
timeout 200 python tools/probes/k3_time.py 14336 4096 2>&1 | grep -v amdgpu.ids > $O/k3_time.txt; head -5 $O/k3_time.txt
bash tools/pmc_bench.sh $O/pmc > $O/pmc.log 2>&1; tail -3 $O/pmc.log; cp $O/pmc/pmc_traffic.json $O/pmc_traffic.json 2>/dev/null; rm -rf $O/pmc/f $O/pmc/w
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kta -o kta -- python bench.py --workload awq --steps 2 --warmup 1 --no-cpu-baseline > $O/kta.log 2>&1
F=$(ls $O/kta/*/*kernel_trace.csv $O/kta/*kernel_trace.csv 2>/dev/null | head -1)
python tools/kernel_stats_csv.py $F 16 > $O/awq_kernel_stats.txt 2>&1; rm -rf $O/kta; head -8 $O/awq_kernel_stats.txt
[ -n "$SKIP_ENVELOPE" ] || ( time timeout 1500 python tools/parity_envelope.py --full-down --out $O/parity_envelope_full_down ) > $O/envelope.log 2>&1; tail -30 $O/envelope.log
