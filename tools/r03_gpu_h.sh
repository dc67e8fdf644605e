#!/bin/bash
# round 3, call H: kernel timeline of one K = 14336 factorisation (where does the chain of a factor step spend its time?)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r03h; mkdir -p $O
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/kt -o kt -- python tools/probes/k3k4_trace.py 14336x4096 > $O/trace.log 2>&1
F=$(ls $O/kt/*kernel_trace.csv | head -1)
# second iteration's factorisation: skip the first 112 potrf calls; window from the 113th (first step) and from the 160th (middle)
python tools/probes/trace_window.py $F k_potrf_inv 112 70 > $O/k3_window_early.txt 2>&1
python tools/probes/trace_window.py $F k_potrf_inv 168 70 > $O/k3_window_mid.txt 2>&1
python tools/probes/trace_window.py $F k_gptq_block 140 50 > $O/k4_window.txt 2>&1
rm -rf $O/kt
head -75 $O/k3_window_early.txt
