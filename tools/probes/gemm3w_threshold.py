import os, sys
sys.path.insert(0, os.getcwd())
import torch
from llmc_amd import _ffi
L = _ffi.lib()
Kd = 512
for n in (2048, 3072, 4096, 5120, 6144, 7168):
    P = torch.randn(Kd, n, device='cuda'); C0 = torch.randn(n, n, device='cuda'); ws = torch.empty(6 * Kd * n, dtype=torch.int16, device='cuda')
    res = {}
    for arm, opts in (('w', dict(gemm3s_min_tiles=1)), ('s', dict(gemm3s_min_tiles=1, gemm3_no_wide=1)), ('w', dict(gemm3s_min_tiles=1)), ('s', dict(gemm3s_min_tiles=1, gemm3_no_wide=1))):
        with _ffi.option(**opts):
            ts = []
            for it in range(12):
                C = C0.clone()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                _ffi.check(L.llmc_test_gemm3_planes(P.data_ptr(), P.data_ptr(), C.data_ptr(), n, n, n, n, n, Kd, 0, 1, ws.data_ptr(), _ffi.stream()), 'planes')
                e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1) * 1e3)
            res.setdefault(arm, []).append(sorted(ts[3:])[4])
    t = n // 128
    print(f'n = {n:5d} ({t * (t + 1) // 2:5d} working tiles): k_gemm3w {res["w"][0]:7.1f} {res["w"][1]:7.1f} us | k_gemm3s {res["s"][0]:7.1f} {res["s"][1]:7.1f} us', flush=True)
