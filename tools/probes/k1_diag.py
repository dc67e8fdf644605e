"""K1 diagnostics (round 4): why does k_syrk4 run at ~0.62 of peak on K = 4096 launches and ~0.54 on K = 14336?
  (a) K = 4096 columns read out of a matrix with the K = 14336 ROW STRIDE (ldx = 14336): same tiles, same flops, the
      down_proj memory pattern (512-B segments 28 KB apart)  -> is it the layout?
  (b) the K = 4096 launch repeated back to back for ~40 ms                     -> is it sustained power / clocks?
  (c) K = 14336 itself, and K = 14336 right after 20 ms of idle               -> burst vs sustained
Prints one line per measurement: ms per launch and contract PFLOP/s (T*K*(K+1))."""
import ctypes as C
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

from llmc_amd import _ffi

L = _ffi.lib()
T = 262144


def synth(Tn, K, seed):
    g = torch.Generator(device='cuda').manual_seed(seed)
    c = torch.exp(0.5 * torch.randn(K, generator=g, device='cuda'))
    c[torch.randperm(K, generator=g, device='cuda')[:8]] *= 100.0
    x = torch.empty((Tn, K), device='cuda', dtype=torch.bfloat16)
    step = max(1, (1 << 27) // K)
    for i in range(0, Tn, step):
        n = min(step, Tn - i)
        x[i:i + n] = (torch.randn((n, K), generator=g, device='cuda') * c).to(torch.bfloat16)
    return x


def launcher(x, K, ldx):
    """returns a closure that launches the partials kernel once (no reduction) on the current stream"""
    need = L.llmc_hessian_accum_ws_bytes(T, K, ldx)
    ws = _ffi.workspace(need, x.device)
    st = _ffi.stream()

    def go():
        _ffi.check(L.llmc_hessian_accum_partials(_ffi.ptr(x), 1, T, K, ldx, _ffi.ptr(ws), st), 'partials')
    return go, ws


def timed(go, reps, label, K, per_launch=False):
    evs = [torch.cuda.Event(enable_timing=True) for _ in range(reps + 1)]
    torch.cuda.synchronize()
    evs[0].record()
    for i in range(reps):
        go()
        evs[i + 1].record()
    torch.cuda.synchronize()
    ts = [evs[i].elapsed_time(evs[i + 1]) for i in range(reps)]
    fl = T * K * (K + 1)
    tot = sum(ts) / reps
    print(f'{label:58s} {tot:8.3f} ms/launch  {fl / tot / 1e12:6.3f} PFLOP/s = {fl / tot / 1e12 / 2.5:5.3f} of peak', flush=True)
    if per_launch:
        print('    per launch ms: ' + ' '.join(f'{t:.3f}' for t in ts), flush=True)
    return ts


def pmc_mode():
    """two launches per width after one warm-up each, for rocprofv3 --kernel-trace --pmc (tools/r04_gpu_a.sh)"""
    x14 = synth(T, 14336, 1)
    x4 = synth(T, 4096, 2)
    go4, w4 = launcher(x4, 4096, 4096)
    go14, w14 = launcher(x14, 14336, 14336)
    for g in (go4, go14, go4, go4, go14, go14):
        g()
        torch.cuda.synchronize()


def main():
    if len(sys.argv) > 1 and sys.argv[1] == '--pmc':
        return pmc_mode()
    x14 = synth(T, 14336, 1)
    x4 = synth(T, 4096, 2)
    go4, w4 = launcher(x4, 4096, 4096)
    go4s, w4s = launcher(x14, 4096, 14336)        # the first 4096 channels of the wide matrix: stride 28 KB
    go14, w14 = launcher(x14, 14336, 14336)
    for g in (go4, go4s, go14):
        g()
    torch.cuda.synchronize()
    timed(go4, 3, 'K=4096 ldx=4096, 3 launches (the bench pattern)', 4096, True)
    time.sleep(0.05)
    timed(go4s, 3, 'K=4096 ldx=14336 (down_proj row stride), 3 launches', 4096, True)
    time.sleep(0.05)
    timed(go4, 14, 'K=4096 ldx=4096, 14 launches back to back (~40 ms)', 4096, True)
    time.sleep(0.05)
    timed(go4s, 14, 'K=4096 ldx=14336, 14 launches back to back', 4096, True)
    time.sleep(0.05)
    timed(go14, 1, 'K=14336 after 50 ms idle', 14336)
    timed(go14, 3, 'K=14336 x3 back to back', 14336, True)
    # the bench order: 3 x K=4096 then K=14336
    time.sleep(0.05)
    e = [torch.cuda.Event(enable_timing=True) for _ in range(5)]
    e[0].record()
    for i in range(3):
        go4(); e[i + 1].record()
    go14(); e[4].record()
    torch.cuda.synchronize()
    print('bench order (3 x K=4096, then K=14336): ' + ' '.join(f'{e[i].elapsed_time(e[i + 1]):.3f}' for i in range(4)) + ' ms', flush=True)


if __name__ == '__main__':
    main()
