#!/bin/bash
# end of round 5, after K4's merged far-update launch: smoke, the default bench line, kernel stats (the GPU tests of this build: tools/r05_gpu_t.sh)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r05final3; mkdir -p $O
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
timeout 600 python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; python - <<PY
import json
try:
    j=json.loads(open('$O/bench.json').read().strip().splitlines()[0]); print('bench', j['value'], j['ms_per_step'], j['ms_per_step_median'], j['roofline']['frac'], j.get('cpu_baseline',{}).get('value'))
    for k,v in j.get('extra',{}).items(): print(' ', k, v.get('value'), v.get('ms_per_step'), (v.get('roofline') or {}).get('frac'), v.get('error'))
except Exception as e: print('bench failed', e); print(open('$O/bench.err').read()[-1500:])
PY
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/ks -o ks -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extras > $O/ks.log 2>&1
F=$(ls $O/ks/*/*kernel_trace.csv $O/ks/*kernel_trace.csv 2>/dev/null | head -1)
python tools/kernel_stats_csv.py $F > $O/kernel_stats.txt 2>&1; head -8 $O/kernel_stats.txt
rm -rf $O/ks
