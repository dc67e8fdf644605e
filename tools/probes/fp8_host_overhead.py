"""FloatQuantizer.real_quant_weight_dynamic on the 28 weights of a Mixtral-8x7B block: host time per call (no sync) against GPU time."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from llmc_amd.compression.quantization import FloatQuantizer
h, kv, ffn = 4096, 1024, 14336
shapes = [(h, h), (kv, h), (kv, h), (h, h)] + [(ffn, h), (ffn, h), (h, ffn)] * 8
ws = [(torch.randn(r, k, device='cuda') * 0.02).to(torch.bfloat16) for r, k in shapes]
q = FloatQuantizer('e4m3', True, 'per_tensor', use_qtorch=True)
for _ in range(3):
    out = [q.real_quant_weight_dynamic(w) for w in ws]
torch.cuda.synchronize()
for rep in range(3):
    t0 = time.perf_counter()
    out = [q.real_quant_weight_dynamic(w) for w in ws]
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f'28 weights: host issue {1e3 * (t1 - t0):.2f} ms ({1e6 * (t1 - t0) / 28:.0f} us per call), until the GPU is done {1e3 * (t2 - t0):.2f} ms')
from llmc_amd.compression.quantization.hist_range import sample_minmax
n_seq, seq = 128, 512
acts = {'attn_in': (n_seq, h), 'o_in': (n_seq, h)}
for e in range(8):
    acts[f'e{e}_in'] = (n_seq // 4, h)
    acts[f'e{e}_mid'] = (n_seq // 4, ffn)
samples = {k: [(torch.randn(seq, kk, device='cuda')).to(torch.bfloat16) for _ in range(n)] for k, (n, kk) in acts.items()}
nbytes = sum(sum(t.numel() * 2 for t in v) for v in samples.values())
def act_part():
    o = []
    for k in samples:
        mn, mx = sample_minmax(samples[k])
        o.append(torch.max(mx.mean().abs(), mn.mean().abs()).clamp(min=1e-5) / 448.0)
    return o
for _ in range(2):
    act_part()
torch.cuda.synchronize()
for rep in range(3):
    t0 = time.perf_counter(); act_part(); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f'18 inputs ({nbytes / 1e9:.2f} GB): host issue {1e3 * (t1 - t0):.2f} ms, until the GPU is done {1e3 * (t2 - t0):.2f} ms = {nbytes / (t2 - t0) / 1e12:.2f} TB/s')
import cProfile, pstats

