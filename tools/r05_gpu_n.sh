#!/bin/bash
# round 5, call N: where k_gemm3's time goes — L2 hit rate, memory-side requests, wave-state split, LDS conflicts (K = 14336 factorisation)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export PYTHONPATH=$GRAFT_REPO_ROOT
O=gpurun_out/r05n; mkdir -p $O
rocprofv3 -L 2>/dev/null | grep -o "TCC_[A-Z0-9_]*\|TCP_[A-Z0-9_]*\|SQ_[A-Z0-9_]*" | sort -u > $O/counters.txt; wc -l $O/counters.txt
pass() {  # name, counters...
  n=$1; shift
  timeout 300 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $O/$n -o p -- python tools/probes/k3_time.py 14336 4096 > $O/$n.log 2>&1 || { echo "pass $n failed"; tail -5 $O/$n.log; }
}
pass p1 GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES
pass p2 GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_REQ_sum
pass p3 GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_INSTS_SALU
pass p4 GRBM_GUI_ACTIVE TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCC_EA0_WRREQ_sum TCC_EA0_RDREQ_32B_sum
python tools/probes/pmc_kernels.py $O/p1 $O/p2 $O/p3 $O/p4 -- k_gemm3 k_sgemm k_linear_eval4 > $O/gemm3_pmc.txt 2>&1; cat $O/gemm3_pmc.txt | cut -c1-600
rm -rf $O/p1 $O/p2 $O/p3 $O/p4
