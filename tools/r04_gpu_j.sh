#!/bin/bash
O=gpurun_out/r04j
mkdir -p $O
cd /root/repo
timeout 300 python tools/bench_stages.py --70b 2>&1 | grep -v amdgpu.ids | tee $O/stage_times_70b.txt
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/$O/kt -o kt -- python /root/repo/tools/bench_stages.py --70b > /root/repo/$O/kt.log 2>&1
cd /root/repo
F=$(find $O/kt -name "*kernel_stats.csv" | head -1)
python tools/kernel_stats_csv.py $F 24 > $O/kernel_stats_70b.txt 2>&1; rm -rf $O/kt; head -26 $O/kernel_stats_70b.txt
