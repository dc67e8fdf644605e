#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r02e
mkdir -p $O
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/kt -o kt -- python tools/probes/k3k4_trace.py > $O/trace.log 2>&1
tail -3 $O/trace.log
ls -la $O/kt
