#!/bin/bash
O=gpurun_out/r04j
mkdir -p $O
cd /root/repo
export TMPDIR=/tmp
rocprofv3 -L 2>/dev/null | grep -o "SQ_[A-Z_0-9]*" | sort -u | tr '\n' ' ' > $O/sq_counters.txt
wc -w $O/sq_counters.txt
run() { # name counters...
  n=$1; shift
  timeout 300 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $O/pmc_$n -o p -- python tools/probes/chain_diag.py > $O/pmc_$n.log 2>&1
  python tools/probes/pmc_table.py $O/pmc_$n "k_sgemm<false, false, true" | sort -t' ' -k6 | awk '{ if ($0 ~ /dur_us +[5-9][0-9][0-9]\./) print }' | head -6 > $O/sgemm_$n.txt
  rm -rf $O/pmc_$n
  cat $O/sgemm_$n.txt | cut -c1-400
}
run a GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES
run b GRBM_GUI_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU
