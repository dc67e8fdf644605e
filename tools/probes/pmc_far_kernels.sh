# PMC passes (busy / clock / LDS / L2) over the two far-update kernels on their largest bench shape: k_gemm3w (K3) and k_sgemm_wide (K4).
# gpurun -- 'bash tools/probes/pmc_far_kernels.sh'   -> gpurun_out/pmc_far/summary.txt
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/pmc_far; mkdir -p $OUT
cat > $OUT/run.py <<'PY'
import os, sys
sys.path.insert(0, os.getcwd())
import torch
from llmc_amd import _ffi
L = _ffi.lib()
Kd, n = 512, 13312
P = torch.randn(Kd, n, device='cuda'); C = torch.randn(n, n, device='cuda'); ws = torch.empty(6 * Kd * n, dtype=torch.int16, device='cuda')
for it in range(6):
    _ffi.check(L.llmc_test_gemm3_planes(P.data_ptr(), P.data_ptr(), C.data_ptr(), n, n, n, n, n, Kd, 0, 1, ws.data_ptr(), _ffi.stream()), 'planes')
M, N, K = 4096, 13824, 14336
A = torch.randn(Kd, M, device='cuda') * 0.01; B = torch.randn(Kd, K, device='cuda'); W = torch.randn(M, K, device='cuda')
for it in range(6):
    _ffi.check(L.llmc_test_sgemm_phased(A.data_ptr(), B[:, K - N:].data_ptr(), W[:, K - N:].data_ptr(), A.stride(0), B.stride(0), W.stride(0), M, N, Kd, 1, 128, _ffi.stream()), 'far')
torch.cuda.synchronize()
PY
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_BUSY_CYCLES --output-format csv -d $OUT/p1 -o p1 -- python $OUT/run.py > $OUT/p1.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE --output-format csv -d $OUT/p2 -o p2 -- python $OUT/run.py > $OUT/p2.log 2>&1
rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum GRBM_GUI_ACTIVE --output-format csv -d $OUT/p3 -o p3 -- python $OUT/run.py > $OUT/p3.log 2>&1
rocprofv3 --kernel-trace --pmc TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum GRBM_GUI_ACTIVE --output-format csv -d $OUT/p4 -o p4 -- python $OUT/run.py > $OUT/p4.log 2>&1
python tools/probes/pmc_kernels.py $OUT/p1 $OUT/p2 $OUT/p3 $OUT/p4 -- k_gemm3w k_sgemm_wide > $OUT/summary.txt 2>&1
cat $OUT/summary.txt | cut -c1-220
rm -rf $OUT/p1 $OUT/p2 $OUT/p3 $OUT/p4
