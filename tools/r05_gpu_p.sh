#!/bin/bash
# round 5, call P: per-stream timeline of the default bench with the planes form of the far updates
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export PYTHONPATH=$GRAFT_REPO_ROOT
O=gpurun_out/r05p; mkdir -p $O
timeout 400 rocprofv3 --kernel-trace --output-format csv -d $O/kt -o kt -- python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-extras > $O/kt.log 2>&1
F=$(find $O/kt -name "*kernel_trace.csv" | head -1)
python tools/step_timeline.py $F --full > $O/step_timeline_planes.txt 2>&1; head -50 $O/step_timeline_planes.txt
rm -rf $O/kt
