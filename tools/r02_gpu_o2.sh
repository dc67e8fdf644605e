#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/o2; mkdir -p $O
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -o kt -- python tools/bench_stages.py > $O/kt.log 2>&1
python tools/kernel_stats_csv.py $O/kt/kt_kernel_trace.csv 40 > $O/kernel_stats.txt 2>&1
rm -rf $O/kt
grep -i "linear_eval4\|split6\|gemm3\|zero_sub\|kernel " $O/kernel_stats.txt
