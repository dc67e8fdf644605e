"""Run-to-run and side-stream-on/off bit equality of the GPTQ column loop (llmc_gptq_quantize)."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

if len(sys.argv) > 1 and sys.argv[1] == 'child':
    import torch
    from llmc_amd.compression.quantization.gptq_ops import chol_inv_upper, gptq_quantize
    out = {}
    for (R, K) in ((4096, 4096), (1024, 14336)):
        g = torch.Generator(device='cuda').manual_seed(R + K)
        W = (torch.randn(R, K, generator=g, device='cuda') * 0.02)
        X = torch.randn(2 * K, K, generator=g, device='cuda')
        H = X.T @ X / K
        H += 0.01 * torch.diagonal(H).mean() * torch.eye(K, device='cuda')
        U = chol_inv_upper(H.clone(), check=False)
        res = [gptq_quantize(W.clone(), U, False, 0.0, 15.0, 128) for _ in range(5)]
        torch.cuda.synchronize()
        print(R, K, 'runs differ:', [int((res[0][0] != r[0]).sum().item()) for r in res[1:]], flush=True)
        out[f'{R}x{K}'] = res[0][0].cpu()
    torch.save(out, sys.argv[2])
else:
    import torch
    env = dict(os.environ)
    subprocess.run([sys.executable, __file__, 'child', '/tmp/k4_side.pt'], env=env, check=True)
    env['LLMC_OPTIONS'] = ''      # (helper streams are switched with llmc_hip_set_helper_streams since round 6)
    subprocess.run([sys.executable, __file__, 'child', '/tmp/k4_noside.pt'], env=env, check=True)
    a, b = torch.load('/tmp/k4_side.pt'), torch.load('/tmp/k4_noside.pt')
    for k in a:
        print(k, 'side vs no-side differing elements:', int((a[k] != b[k]).sum().item()))
