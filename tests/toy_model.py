"""A tiny model adapter with the methods llmc's algorithms call on `BaseModel` (llmc/models/base_model.py:22-481):
two gated-MLP blocks whose subset table has the same structure as Llama's (a subset of two Linears sharing the
LayerNorm output, then a Linear fed by the product). Used to drive the host classes end to end."""
from types import SimpleNamespace

import torch
import torch.nn as nn


class ToyBlock(nn.Module):
    def __init__(self, hidden, inner):
        super().__init__()
        self.ln = nn.LayerNorm(hidden)
        self.gate_proj = nn.Linear(hidden, inner, bias=False)
        self.up_proj = nn.Linear(hidden, inner, bias=False)
        self.down_proj = nn.Linear(inner, hidden, bias=False)

    def forward(self, x, **kwargs):
        h = self.ln(x)
        return x + self.down_proj(torch.nn.functional.silu(self.gate_proj(h)) * self.up_proj(h))


class Stacked(nn.Module):
    def __init__(self, layers):
        super().__init__()
        self.layers = nn.ModuleList(layers)

    def forward(self, x, **kw):
        return torch.cat([l(x) for l in self.layers], dim=-1)


class ToyModel:
    block_name_prefix = 'blocks'

    def __init__(self, hidden=256, inner=384, n_blocks=2, dtype=torch.bfloat16, seed=0):
        g = torch.Generator().manual_seed(seed)
        self.model = nn.Module()
        self.model.blocks = nn.ModuleList([ToyBlock(hidden, inner) for _ in range(n_blocks)])
        for p in self.model.parameters():
            if p.dim() == 2:
                p.data = torch.randn(p.shape, generator=g) * 0.05
        self.model = self.model.to(dtype)
        self.torch_dtype = dtype
        self.model_config = SimpleNamespace(hidden_size=hidden, num_attention_heads=4, intermediate_size=inner)
        self.tokenizer = None
        self.mm_model = None
        self.kvcache_buffer = []

    def get_blocks(self):
        return list(self.model.blocks)

    def get_model(self):
        return self.model

    def get_block_linears(self, block):
        return {n: m for n, m in block.named_modules() if isinstance(m, nn.Linear)}

    def get_extra_modules(self, block):
        return {}

    def get_subsets_in_block(self, block):
        return [
            {'layers': {'gate_proj': block.gate_proj, 'up_proj': block.up_proj}, 'prev_op': [block.ln],
             'input': ['gate_proj'], 'inspect': Stacked([block.gate_proj, block.up_proj]), 'has_kwargs': False,
             'is_mlp': True},
            {'layers': {'down_proj': block.down_proj}, 'prev_op': [block.up_proj], 'input': ['down_proj'],
             'inspect': block.down_proj, 'has_kwargs': False, 'is_mlp': True},
        ]

    def _replace(self, parent, cls, params, names=None):
        for n, m in list(parent.named_children()):
            if names is not None and n not in names:
                continue
            if isinstance(m, nn.Linear) or type(m).__name__.endswith('Linear'):
                setattr(parent, n, cls.new(m, **params))

    def replace_module_block(self, cls, block, block_idx, params):
        self._replace(block, cls, params)

    def replace_module_subset(self, cls, block, subset, block_idx, params):
        self._replace(block, cls, params, names=set(subset['layers']))

    def replace_language_module_all(self, cls, params, keep_device=False):
        for b in self.model.blocks:
            b.cuda()
            self._replace(b, cls, params)
            if not keep_device:
                b.cpu()

    def convert_dtype(self, dtype):
        for b in self.model.blocks:
            for n, m in b.named_modules():
                if hasattr(m, 'weight') and torch.is_tensor(getattr(m, 'weight', None)) and m.weight.is_floating_point():
                    m.weight.data = m.weight.data.to(dtype)


def calib_input(model, n_seq=6, seq=64, seed=1):
    g = torch.Generator().manual_seed(seed)
    hidden = model.model_config.hidden_size
    c = torch.exp(0.5 * torch.randn(hidden, generator=g))
    c[torch.randperm(hidden, generator=g)[:4]] *= 30
    data = [(torch.randn(1, seq, hidden, generator=g) * c).to(model.torch_dtype) for _ in range(n_seq)]
    return {'data': data, 'kwargs': [{} for _ in range(n_seq)]}
