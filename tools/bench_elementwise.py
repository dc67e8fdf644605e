"""HBM-bound quantizer kernels on Llama shapes: run each a few times (use with rocprofv3 --kernel-trace for kernel-only times)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from llmc_amd.compression.quantization import FloatQuantizer, IntegerQuantizer, pack_lsb

g = torch.Generator(device='cuda').manual_seed(0)
R, K = 14336, 4096
w = (torch.randn(R, K, generator=g, device='cuda') * 0.02).to(torch.bfloat16)
q4 = IntegerQuantizer(4, False, 'per_group', group_size=128)
q4s = IntegerQuantizer(4, True, 'per_group', group_size=128)
q8 = IntegerQuantizer(8, True, 'per_channel')
f8 = FloatQuantizer('e4m3', True, 'per_tensor', use_qtorch=True)
for _ in range(5):
    q4.get_tensor_qparams(w)
    q4.fake_quant_weight_dynamic(w)
    codes, s, z = q4s.real_quant_weight_dynamic(w)
    pack_lsb(codes, 4)
    q8.fake_quant_weight_dynamic(w)
    f8.real_quant_weight_dynamic(w)
    f8.fake_quant_weight_dynamic(w)
torch.cuda.synchronize()
print('done', R, K)
