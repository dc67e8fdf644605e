#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r03m; mkdir -p $O; rm -f $O/actuals.jsonl
LLMC_TEST_ACTUALS=$PWD/$O/actuals.jsonl timeout 600 python -m pytest tests/test_fp8_block_gpu.py -q -m gpu -x > $O/tests.log 2>&1; tail -3 $O/tests.log; cat $O/actuals.jsonl
timeout 300 python tools/bench_fp8_block.py --more > $O/fp8_block_rates.txt 2>&1; tail -3 $O/fp8_block_rates.txt
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/p1 -o p -- python tools/probes/fp8_gemm_only.py > $O/p1.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM --output-format csv -d $O/p2 -o p -- python tools/probes/fp8_gemm_only.py > $O/p2.log 2>&1
python tools/pmc_summary.py $O gemm256 > $O/pmc.txt 2>&1; cat $O/pmc.txt; rm -rf $O/p1 $O/p2
