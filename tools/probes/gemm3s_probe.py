"""(Round 5 probe; LLMC_GEMM3S_DBG needs a -DLLMC_LAB build of the library, gemm3_nospec is an option since round 6.)
Time one far update (C -= P^T P, upper only, Kd = 512) on k_gemm3 and on k_gemm3s with parts switched off (LLMC_GEMM3S_DBG).
usage: python tools/probes/gemm3s_probe.py [n]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

from llmc_amd import _ffi

n = int(sys.argv[1]) if len(sys.argv) > 1 else 13312
Kd = 512
L = _ffi.lib()
P = torch.randn(Kd, n, device='cuda')
C = torch.randn(n, n, device='cuda')


def run(label, env):
    for k in ('LLMC_GEMM3_NOSPEC', 'LLMC_GEMM3S_DBG', 'LLMC_GEMM3S_PRIO'):
        os.environ.pop(k, None)
    os.environ.update(env)
    ts = []
    for it in range(8):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        _ffi.check(L.llmc_test_gemm3(P.data_ptr(), P.data_ptr(), C.data_ptr(), n, n, n, n, n, Kd, 1, 0, 0, 0, 1, _ffi.stream()), 'gemm3')
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    ts = sorted(ts[3:])
    print(f'{label:44s} {ts[len(ts) // 2]:8.1f} us', flush=True)


ws = torch.empty(6 * Kd * n, dtype=torch.int16, device='cuda')


def run_planes(label, env, split=True):
    for k in ('LLMC_GEMM3_NOSPEC', 'LLMC_GEMM3S_DBG'):
        os.environ.pop(k, None)
    os.environ.update(env)
    ts = []
    for it in range(8):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        _ffi.check(L.llmc_test_gemm3_planes(P.data_ptr(), P.data_ptr(), C.data_ptr(), n, n, n, n, n, Kd, 0, 1, ws.data_ptr(), _ffi.stream()), 'planes')
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    ts = sorted(ts[3:])
    print(f'{label:44s} {ts[len(ts) // 2]:8.1f} us', flush=True)


run('k_gemm3 (128 x 128, all waves do both)', {'LLMC_GEMM3_NOSPEC': '1'})
run('k_gemm3s, split by the producer waves', {})
run('k_gemm3s, producers split, no MFMAs', {'LLMC_GEMM3S_DBG': '2'})
run_planes('2 x k_split3_planes + k_gemm3s on planes', {})
run_planes('the same without MFMAs', {'LLMC_GEMM3S_DBG': '2'})

# stamps of one workgroup (LLMC_GEMM3S_DBG=4): s_memtime at the phase boundaries of every wave
import ctypes
import numpy as np
os.environ['LLMC_GEMM3S_DBG'] = '4'
for _ in range(2):
    _ffi.check(L.llmc_test_gemm3_planes(P.data_ptr(), P.data_ptr(), C.data_ptr(), n, n, n, n, n, Kd, 0, 1, ws.data_ptr(), _ffi.stream()), 'planes')
torch.cuda.synchronize()
st = np.zeros(8 * 128, dtype=np.int64)
assert L.llmc_test_gemm3s_stamps(st.ctypes.data) == 0
st = st.reshape(8, 128)
t0 = st[:, 0].min()
tick = 1.0   # s_memtime counts shader clocks here (a 78-us tile = 160k ticks)
print('shader clocks since the first wave of the stamped workgroup started (planes form)')
print('MFMA wave 0: step start (after barrier) / MFMAs issued')
for s in range(16):
    print(f'  step {s:2d}: start {(st[0, 1 + 4 * s] - t0) * tick:8.2f}  issued {(st[0, 2 + 4 * s] - t0) * tick:8.2f}')
print(f'  loop end {(st[0, 120] - t0) * tick:8.2f}   epilogue end {(st[0, 121] - t0) * tick:8.2f}')
print(f'  producer wave 4: tile in LDS {(st[4, 120] - t0) * tick:8.2f}   C written {(st[4, 121] - t0) * tick:8.2f}')
print('producer wave 4: loads there / planes written + next loads issued / LDS writes done / after barrier')
for j in range(16):
    print(f'  step {j:2d}: {(st[4, 1 + 4 * j] - t0) * tick:8.2f} {(st[4, 2 + 4 * j] - t0) * tick:8.2f} {(st[4, 3 + 4 * j] - t0) * tick:8.2f} {(st[4, 4 + 4 * j] - t0) * tick:8.2f}')
print('step 14 barrier: MFMA waves issued step 13 at', [int(st[w, 2 + 4 * 13] - t0) for w in range(4)])
print('  producers: planes of step 14 written', [int(st[w, 3 + 4 * 14] - t0) for w in range(4, 8)], 'old loads issued', [int(st[w, 100] - t0) for w in range(4, 8)],
      'lgkmcnt(0)', [int(st[w, 101] - t0) for w in range(4, 8)], 'after barrier', [int(st[w, 4 + 4 * 14] - t0) for w in range(4, 8)])
