#!/bin/bash
# round 5, call J: rocprofv3 --kernel-trace --stats summary of the default bench command
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r05j; mkdir -p $O
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/ks -o ks -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extras > $O/ks.log 2>&1
tail -2 $O/ks.log
F=$(ls $O/ks/*/*kernel_trace.csv $O/ks/*kernel_trace.csv 2>/dev/null | head -1)
python tools/kernel_stats_csv.py $F > $O/kernel_stats.txt 2>&1; head -34 $O/kernel_stats.txt
S=$(ls $O/ks/*/*kernel_stats.csv $O/ks/*kernel_stats.csv 2>/dev/null | head -1); head -12 $S > $O/rocprof_kernel_stats_head.csv
rm -rf $O/ks
