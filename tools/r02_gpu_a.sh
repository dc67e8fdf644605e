#!/bin/bash
# round-2 GPU call A: full GPU test suite (new default K1 kernel), K1 variant lab, PMC passes on the lab
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r02a
mkdir -p $O
timeout 600 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1
echo "pytest rc=$?" >> $O/pytest.log
tail -5 $O/pytest.log
timeout 300 tools/probes/syrk_lab > $O/lab.txt 2>&1
cat $O/lab.txt
export LAB_REPS=2 LAB_QUICK=1
timeout 200 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/p1 -o p1 -- tools/probes/syrk_lab 262144 4096 > $O/p1.log 2>&1
timeout 200 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU --output-format csv -d $O/p2 -o p2 -- tools/probes/syrk_lab 262144 4096 > $O/p2.log 2>&1
timeout 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE TCC_HIT_sum --output-format csv -d $O/p3 -o p3 -- tools/probes/syrk_lab 262144 4096 65536 14336 > $O/p3.log 2>&1
timeout 200 rocprofv3 --kernel-trace --pmc WRITE_SIZE TCC_MISS_sum --output-format csv -d $O/p4 -o p4 -- tools/probes/syrk_lab 262144 4096 65536 14336 > $O/p4.log 2>&1
rocprofv3 -L 2>/dev/null | grep -i "name" | grep -i "dram\|mall\|EA0\|HBM\|TCC_EA" | head -60 > $O/counters.txt
python tools/pmc_summary.py $O syrk > $O/pmc_summary.txt 2>&1
cat $O/pmc_summary.txt | head -150
