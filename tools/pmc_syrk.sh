#!/bin/bash
# PMC passes for k_syrk (one pass per counter group; gpurun forbids mixing --pmc with sys traces).
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/pmc_syrk
mkdir -p $OUT
CMD="python tools/bench_syrk.py 262144x4096"
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES --output-format csv -d $OUT/p1 -o p1 -- $CMD > $OUT/p1.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_INSTS_MFMA --output-format csv -d $OUT/p2 -o p2 -- $CMD > $OUT/p2.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE TCC_HIT_sum --output-format csv -d $OUT/p3 -o p3 -- $CMD > $OUT/p3.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE TCC_MISS_sum --output-format csv -d $OUT/p4 -o p4 -- $CMD > $OUT/p4.log 2>&1
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE TCC_REQ_sum TCC_EA0_RDREQ_sum --output-format csv -d $OUT/p5 -o p5 -- $CMD > $OUT/p5.log 2>&1
find $OUT -name "*.csv" | head -20
tail -2 $OUT/p1.log
