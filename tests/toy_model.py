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


# ---- a sparse mixture-of-experts block in the style llmc's Mixtral adapter expects (llmc/models/mixtral.py:43-86: per-expert
# nn.Linear w1 / w3 / w2 plus the router `gate`; transformers 5 fuses the experts into 3-D parameters, so the structure is
# restated here). Experts that receive no token in a forward pass are SKIPPED, like the Hugging Face / DeepSeek forwards.
class ToyExpert(nn.Module):
    def __init__(self, hidden, inner):
        super().__init__()
        self.w1 = nn.Linear(hidden, inner, bias=False)
        self.w3 = nn.Linear(hidden, inner, bias=False)
        self.w2 = nn.Linear(inner, hidden, bias=False)

    def forward(self, x):
        return self.w2(torch.nn.functional.silu(self.w1(x)) * self.w3(x))


class ToySparseMoe(nn.Module):
    def __init__(self, hidden, inner, n_experts, top_k=2):
        super().__init__()
        self.gate = nn.Linear(hidden, n_experts, bias=False)
        self.experts = nn.ModuleList([ToyExpert(hidden, inner) for _ in range(n_experts)])
        self.top_k = top_k

    def forward(self, x):
        shp = x.shape
        h = x.reshape(-1, shp[-1])
        w = torch.softmax(self.gate(h).float(), dim=-1)
        topw, sel = torch.topk(w, self.top_k, dim=-1)
        topw = (topw / topw.sum(-1, keepdim=True)).to(h.dtype)
        out = torch.zeros_like(h)
        for e, expert in enumerate(self.experts):
            tok, k = (sel == e).nonzero(as_tuple=True)
            if tok.numel() == 0:
                continue                                    # no token routed here: the expert's Linears are not called
            out.index_add_(0, tok, expert(h[tok]) * topw[tok, k, None])
        return out.reshape(shp)


class ToyMoeBlock(nn.Module):
    def __init__(self, hidden, inner, n_experts):
        super().__init__()
        self.ln = nn.LayerNorm(hidden)
        self.block_sparse_moe = ToySparseMoe(hidden, inner, n_experts)

    def forward(self, x, **kwargs):
        return x + self.block_sparse_moe(self.ln(x))


class ToyMoeModel(ToyModel):
    def __init__(self, hidden=128, inner=256, n_experts=4, n_blocks=1, dtype=torch.bfloat16, seed=0):
        g = torch.Generator().manual_seed(seed)
        self.model = nn.Module()
        self.model.blocks = nn.ModuleList([ToyMoeBlock(hidden, inner, n_experts) for _ in range(n_blocks)])
        for p in self.model.parameters():
            if p.dim() == 2:
                p.data = torch.randn(p.shape, generator=g) * 0.05
        self.model = self.model.to(dtype)
        self.torch_dtype = dtype
        self.model_config = SimpleNamespace(hidden_size=hidden, num_attention_heads=4, intermediate_size=inner)
        self.tokenizer = None
        self.mm_model = None
        self.kvcache_buffer = []

    def get_extra_modules(self, block):
        return {'block_sparse_moe': block.block_sparse_moe}

    def get_subsets_in_block(self, block):      # llmc/models/mixtral.py:62-86 (without the attention subsets)
        moe = block.block_sparse_moe
        n = len(moe.experts)
        return [
            {'layers': {**{f'block_sparse_moe.experts.{i}.w1': moe.experts[i].w1 for i in range(n)},
                        **{f'block_sparse_moe.experts.{i}.w3': moe.experts[i].w3 for i in range(n)},
                        'block_sparse_moe.gate': moe.gate},
             'prev_op': [block.ln], 'input': ['block_sparse_moe'], 'inspect': moe, 'has_kwargs': False, 'is_mlp': True},
            *[{'layers': {f'block_sparse_moe.experts.{i}.w2': moe.experts[i].w2}, 'prev_op': [moe.experts[i].w3],
               'input': [f'block_sparse_moe.experts.{i}.w2'], 'inspect': moe.experts[i].w2, 'has_kwargs': False,
               'is_mlp': True} for i in range(n)],
        ]

    def _replace(self, parent, cls, params, names=None):
        # nested names: walk the block's Linear modules by qualified name
        for qn, m in list(parent.named_modules()):
            if not (isinstance(m, nn.Linear) or type(m).__name__.endswith('Linear')) or qn == '':
                continue
            if names is not None and qn not in names:
                continue
            owner = parent.get_submodule(qn.rsplit('.', 1)[0]) if '.' in qn else parent
            setattr(owner, qn.rsplit('.', 1)[-1], cls.new(m, **params))
