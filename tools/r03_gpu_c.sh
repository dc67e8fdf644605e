#!/bin/bash
# round 3, call C: Hessian kernel with the bookkeeping moved into MFMA gaps (tests + A/B against the round-2 build), the clip
# kernel's error table against the oracle, measured values behind the end-to-end bounds, HBM-bound kernel rates
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r03c; mkdir -p $O
timeout 300 python -m pytest tests/test_hessian_gpu.py tests/test_bench_shapes_gpu.py -x -q -m gpu > $O/tests_hessian.log 2>&1; tail -4 $O/tests_hessian.log
timeout 200 python tools/probes/ab_syrk_libs.py tools/probes/libllmc_hip_r02.so llmc_amd/csrc/libllmc_hip.so 262144 4096 262144 14336 > $O/ab_syrk.txt 2>&1; cat $O/ab_syrk.txt
timeout 100 python tools/probes/clip_agree.py > $O/clip_agree.txt 2>&1; cat $O/clip_agree.txt
rm -f $O/actuals.jsonl
LLMC_TEST_ACTUALS=$PWD/$O/actuals.jsonl timeout 400 python -m pytest tests/test_e2e_gpu.py tests/test_spqr_gpu.py tests/test_envelope_gpu.py -q -m gpu > $O/tests_e2e.log 2>&1; tail -6 $O/tests_e2e.log; cat $O/actuals.jsonl
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -o kt -- python tools/bench_elementwise.py > $O/elem.log 2>&1
python tools/kernel_stats_csv.py $O/kt/kt_kernel_trace.csv 24 > $O/elementwise_kernel_stats.txt 2>&1; rm -rf $O/kt; cat $O/elementwise_kernel_stats.txt
