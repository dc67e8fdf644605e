#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r02d
mkdir -p $O
timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline > $O/bench.json 2> $O/bench.err
cat $O/bench.json
timeout 300 python tools/bench_stages.py > $O/stages.txt 2>&1
cat $O/stages.txt
