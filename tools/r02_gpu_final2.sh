#!/bin/bash
# round-2 evidence refresh after gemm6: GPTQ bench line, kernel stats, stage times, full GPU tests
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r02final; mkdir -p $O
timeout 900 python -m pytest tests -x -q -m gpu > $O/tests.log 2>&1; tail -2 $O/tests.log
timeout 400 python bench.py --steps 3 --warmup 1 > $O/bench.json 2> $O/bench.err
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -o kt -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline > $O/kt.log 2>&1
python tools/kernel_stats_csv.py $O/kt/kt_kernel_trace.csv 32 > $O/kernel_stats.txt 2>&1
rm -rf $O/kt
timeout 200 python tools/bench_stages.py > $O/stage_times.txt 2>&1
python -c "
import json; j=json.load(open('$O/bench.json')); print(j['value'], j['ms_per_step'], j['roofline']['frac'], j['cpu_baseline']['value'])"
head -10 $O/kernel_stats.txt; tail -4 $O/stage_times.txt
