"""A/B of the Hessian kernel between two builds of the library on the SAME box, alternating launches:
   python tools/probes/ab_syrk_libs.py libA.so libB.so [T K]...
Times llmc_hessian_accum_partials with HIP events (median of reps), prints contract TFLOP/s (T*K*(K+1)) and checks that both
produce the same Hessian up to fp32 summation order. Measurement infrastructure (round 3: the round-2 kernel,
tools/probes/libllmc_hip_r02.so built from git history, against the sample-table kernel)."""
import ctypes as C
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def load(path):
    L = C.CDLL(path)
    L.llmc_hessian_accum_ws_bytes.restype = C.c_size_t
    L.llmc_hessian_accum_ws_bytes.argtypes = [C.c_int64] * 3
    L.llmc_hessian_accum_partials.restype = C.c_int
    L.llmc_hessian_accum_partials.argtypes = [C.c_void_p, C.c_int, C.c_int64, C.c_int64, C.c_int64, C.c_void_p, C.c_void_p]
    L.llmc_hessian_accum_reduce.restype = C.c_int
    L.llmc_hessian_accum_reduce.argtypes = [C.c_void_p, C.c_int64, C.c_int64, C.c_int64, C.c_double, C.c_double, C.c_void_p, C.c_void_p]
    return L


def main():
    libs = [(os.path.basename(p), load(os.path.abspath(p))) for p in sys.argv[1:3]]
    shapes = [(int(sys.argv[i]), int(sys.argv[i + 1])) for i in range(3, len(sys.argv) - 1, 2)] or [(262144, 4096), (131072, 14336)]
    dev = torch.device('cuda', 0)
    st = torch.cuda.current_stream().cuda_stream
    for T, K in shapes:
        g = torch.Generator(device=dev).manual_seed(1)
        c = torch.exp(0.5 * torch.randn(K, generator=g, device=dev))
        c[torch.randperm(K, generator=g, device=dev)[:8]] *= 100
        x = torch.empty((T, K), dtype=torch.bfloat16, device=dev)
        for i in range(0, T, 16384):
            x[i:i + 16384] = (torch.randn((min(16384, T - i), K), generator=g, device=dev) * c).to(torch.bfloat16)
        ws = [torch.empty(L.llmc_hessian_accum_ws_bytes(T, K, K), dtype=torch.uint8, device=dev) for _, L in libs]
        Hs = [torch.empty((K, K), dtype=torch.float32, device=dev) for _ in libs]
        times = [[] for _ in libs]
        for rep in range(7):
            for li, (name, L) in enumerate(libs):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                rc = L.llmc_hessian_accum_partials(x.data_ptr(), 1, T, K, K, ws[li].data_ptr(), st)
                e1.record()
                assert rc == 0, (name, rc)
                L.llmc_hessian_accum_reduce(Hs[li].data_ptr(), T, K, K, 0.0, 1.0, ws[li].data_ptr(), st)
                torch.cuda.synchronize()
                if rep >= 2:
                    times[li].append(e0.elapsed_time(e1))
        fl = T * K * (K + 1)
        d = (Hs[0] - Hs[1]).abs().max().item() / Hs[0].abs().max().item()
        for li, (name, _) in enumerate(libs):
            t = sorted(times[li])[len(times[li]) // 2]
            print(f'T={T} K={K} {name:<28} median {t:8.3f} ms  {fl / t / 1e9:8.1f} TFLOP/s  frac {fl / t / 1e9 / 2500:.4f}  all {[round(v, 2) for v in times[li]]}')
        print(f'   max |H_a - H_b| / max|H| = {d:.3g}', flush=True)


if __name__ == '__main__':
    main()
