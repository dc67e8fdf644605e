#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r04i; mkdir -p $O
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/ktf -o ktf -- python bench.py --workload fp8 --steps 10 --warmup 2 --no-cpu-baseline > $O/ktf.log 2>&1
F=$(ls $O/ktf/*/*kernel_trace.csv $O/ktf/*kernel_trace.csv 2>/dev/null | head -1)
python tools/kernel_stats_csv.py $F 12 | tee $O/fp8_kernel_stats.txt; rm -rf $O/ktf
tail -1 $O/ktf.log | python -c "import json,sys; j=json.loads(sys.stdin.read()); print(j['value'], j['ms_per_step'])"
