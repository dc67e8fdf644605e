#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r03x; mkdir -p $O
timeout 900 python -m pytest tests/test_quant_gpu.py tests/test_fp8_block_gpu.py tests/test_export_gpu.py tests/test_gptq_deploy_gpu.py -q -m gpu -x > $O/tests.log 2>&1; tail -3 $O/tests.log
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -o kt -- python tools/bench_elementwise.py > $O/kt.log 2>&1
python tools/kernel_stats_csv.py $O/kt/kt_kernel_trace.csv 24 > $O/elementwise_kernel_stats.txt 2>&1; rm -rf $O/kt; head -20 $O/elementwise_kernel_stats.txt
