"""Time k_gemm3w's largest bench launch (n = 13312, Kd = 512, upper) with the library named by LLMC_PROBE_LIB (tools/probes/g3w_lab.sh)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from llmc_amd import _ffi
if os.environ.get('LLMC_PROBE_LIB'):
    _ffi.LIB_PATH = os.environ['LLMC_PROBE_LIB']
L = _ffi.lib()
Kd, n = 512, 13312
P = torch.randn(Kd, n, device='cuda'); C = torch.randn(n, n, device='cuda'); ws = torch.empty(6 * Kd * n, dtype=torch.int16, device='cuda')
ts = []
for it in range(10):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    _ffi.check(L.llmc_test_gemm3_planes(P.data_ptr(), P.data_ptr(), C.data_ptr(), n, n, n, n, n, Kd, 0, 1, ws.data_ptr(), _ffi.stream()), 'planes')
    e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1) * 1e3)
print(os.environ.get('LLMC_PROBE_LIB', 'shipped'), f'split + k_gemm3w: {sorted(ts[3:])[3]:.1f} us', flush=True)
