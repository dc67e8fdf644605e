"""Generate tests/golden/*.npz by running the REFERENCE's own code (llmc @ /root/reference) on CPU.

Run in the build container only (the GPU box has no /root/reference):
    python oracle/make_golden.py            # all suites
    python oracle/make_golden.py quant pack # selected suites
The reference is imported read-only with two import shims (oracle/_shims: loguru, easydict) and three
monkeypatches that do what the reference's own CI rewrite does (ci_check/change_files.py:34-179):
`.cuda()` -> no-op, torch.cuda.synchronize/empty_cache -> no-op, device='cuda' -> 'cpu'.  Its CI
work-shrinkers (AWQ n_grid=1, nsamples=1) are NOT applied.  Outputs are small seeded fixtures; 16-bit
tensors are stored as float32 (exact).  Test infrastructure only.
"""
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
GOLD = os.path.join(ROOT, 'tests', 'golden')
sys.path.insert(0, os.path.join(HERE, '_shims'))
sys.path.insert(0, '/root/reference')

import torch  # noqa: E402

torch.Tensor.cuda = lambda self, *a, **k: self
torch.cuda.synchronize = lambda *a, **k: None
torch.cuda.empty_cache = lambda *a, **k: None
_orig_tensor, _orig_zeros_like, _orig_zeros = torch.tensor, torch.zeros_like, torch.zeros


def _cpu_dev(fn):
    def wrap(*a, **k):
        if str(k.get('device', '')).startswith('cuda'):
            k['device'] = 'cpu'
        return fn(*a, **k)
    return wrap


_orig_to = torch.Tensor.to


def _to(self, *a, **k):
    if str(k.get('device', '')).startswith('cuda'):
        k['device'] = 'cpu'
    a = tuple('cpu' if ((isinstance(x, str) and x.startswith('cuda')) or
                        (isinstance(x, torch.device) and x.type == 'cuda')) else x for x in a)
    if isinstance(k.get('device'), torch.device) and k['device'].type == 'cuda':
        k['device'] = 'cpu'
    return _orig_to(self, *a, **k)


torch.Tensor.to = _to
torch.tensor = _cpu_dev(_orig_tensor)
torch.zeros_like = _cpu_dev(_orig_zeros_like)
torch.zeros = _cpu_dev(_orig_zeros)

os.environ.setdefault('WORLD_SIZE', '1')
os.environ.setdefault('RANK', '0')

from llmc.compression.quantization.quant import IntegerQuantizer  # noqa: E402

DT = {'f16': torch.float16, 'bf16': torch.bfloat16, 'f32': torch.float32}


def f32(t):
    return t.detach().float().numpy().copy()


def save(name, **arrs):
    os.makedirs(GOLD, exist_ok=True)
    path = os.path.join(GOLD, name + '.npz')
    np.savez_compressed(path, **arrs)
    print(f'wrote {path} ({os.path.getsize(path) / 1024:.1f} KiB)')


def rand_weight(gen, R, K, dt, outliers=True):
    w = torch.randn(R, K, generator=gen) * 0.02
    if outliers:
        idx = torch.randperm(K, generator=gen)[: max(1, K // 64)]
        w[:, idx] *= 20
    # a constant group (max == min -> clamp(1e-5) path), an all-zero group and a huge group
    w[0, :128] = 0.0
    if R > 1:
        w[1, :128] = 0.0173
    if R > 2:
        w[2, :128] *= 3000.0
    return w.to(DT[dt])


# ------------------------------------------------------------------------------------------------
def suite_quant():
    """IntegerQuantizer on weights: qparams, fake dynamic, real dynamic, static with mixed dtypes."""
    gen = torch.Generator().manual_seed(1234)
    out = {}
    cases = []
    for dt in ('f16', 'bf16', 'f32'):
        for (bit, sym, gran, gs) in [(4, False, 'per_group', 128), (4, True, 'per_group', 128),
                                     (8, True, 'per_channel', None), (8, False, 'per_channel', None),
                                     (4, True, 'per_group', 64), (8, True, 'per_tensor', None),
                                     (3, False, 'per_group', 32)]:
            cases.append((dt, bit, sym, gran, gs))
    for ci, (dt, bit, sym, gran, gs) in enumerate(cases):
        kw = dict(group_size=gs) if gs else {}
        q = IntegerQuantizer(bit, sym, gran, **kw)
        w = rand_weight(gen, 16, 384, dt)
        t, s, z, qmax, qmin = q.get_tensor_qparams(w)
        fq = q.fake_quant_weight_dynamic(w)
        rw, rs, rz = q.real_quant_weight_dynamic(w)
        p = f'c{ci}_'
        out[p + 'w'] = f32(w)
        out[p + 'scales'] = f32(s).reshape(-1)
        out[p + 'zeros'] = f32(z).reshape(-1) if z.dim() > 0 else np.zeros(0, np.float32)
        out[p + 'scales_dtype'] = np.array(str(s.dtype))
        out[p + 'fake'] = f32(fq)
        out[p + 'codes'] = rw.numpy().astype(np.int32)
        out[p + 'codes_dtype'] = np.array(str(rw.dtype))
        out[p + 'rscales'] = f32(rs)
        out[p + 'rzeros'] = rz.numpy().astype(np.int32) if rz is not None else np.zeros(0, np.int32)
        out[p + 'meta'] = np.array([bit, int(sym), gs or 0, float(qmin), float(qmax)], dtype=np.float64)
        out[p + 'gran'] = np.array(gran)
        out[p + 'dt'] = np.array(dt)
    out['n_cases'] = np.array(len(cases))

    # static: fp32 weights (GPTQ leaves layer.weight fp32) with model-dtype / fp32 qparams
    sc = []
    for (wdt, sdt, zdt, sym) in [('f32', 'f16', 'f16', False), ('f32', 'bf16', 'bf16', False),
                                 ('f32', 'f32', 'f32', False), ('f32', 'f16', None, True),
                                 ('f16', 'f16', 'f32', False), ('bf16', 'f32', 'f32', False),
                                 ('f32', 'f16', 'f32', False)]:
        sc.append((wdt, sdt, zdt, sym))
    for ci, (wdt, sdt, zdt, sym) in enumerate(sc):
        q = IntegerQuantizer(4, sym, 'per_group', group_size=128)
        w0 = rand_weight(gen, 16, 256, 'f16' if sdt == 'f32' else sdt)
        _, s, z, qmax, qmin = q.get_tensor_qparams(w0)
        s = s.to(DT[sdt])
        z = z.to(DT[zdt]) if zdt is not None and z.dim() > 0 else z
        w = (w0.float() + 0.003 * torch.randn(16, 256, generator=gen)).to(DT[wdt])
        args = {'scales': s, 'zeros': z, 'qmax': qmax, 'qmin': qmin}
        fq = q.fake_quant_weight_static(w, dict(args))
        rw, rs, rz = q.real_quant_weight_static(w, dict(args))
        p = f's{ci}_'
        out[p + 'w'] = f32(w)
        out[p + 'scales'] = f32(s).reshape(-1)
        out[p + 'zeros'] = f32(z).reshape(-1) if z.dim() > 0 else np.zeros(0, np.float32)
        out[p + 'fake'] = f32(fq)
        out[p + 'fake_dtype'] = np.array(str(fq.dtype))
        out[p + 'codes'] = rw.numpy().astype(np.int32)
        out[p + 'meta'] = np.array([4, int(sym), 128, float(qmin), float(qmax)], dtype=np.float64)
        out[p + 'dts'] = np.array([wdt, sdt, zdt or 'none'])
    out['n_static'] = np.array(len(sc))
    save('quant', **out)


def suite_quant_pt():
    """per_tensor ASYMMETRIC IntegerQuantizer: min/max are 0-dim tensors and (qmax - qmin) is a 0-dim fp32 tensor, so
    type promotion makes scales / zeros fp32 whatever the tensor dtype (quant.py:132-136,555-556)."""
    gen = torch.Generator().manual_seed(4321)
    out = {}
    cases = [(dt, bit) for dt in ('f16', 'bf16', 'f32') for bit in (8, 4)]
    for ci, (dt, bit) in enumerate(cases):
        q = IntegerQuantizer(bit, False, 'per_tensor')
        w = rand_weight(gen, 16, 384, dt)
        t, s, z, qmax, qmin = q.get_tensor_qparams(w)
        fq = q.fake_quant_weight_dynamic(w)
        rw, rs, rz = q.real_quant_weight_dynamic(w)
        p = f'c{ci}_'
        out[p + 'w'] = f32(w)
        out[p + 'scales'], out[p + 'zeros'] = f32(s).reshape(-1), f32(z).reshape(-1)
        out[p + 'scales_dtype'], out[p + 'zeros_dtype'] = np.array(str(s.dtype)), np.array(str(z.dtype))
        out[p + 'fake'], out[p + 'fake_dtype'] = f32(fq), np.array(str(fq.dtype))
        out[p + 'codes'], out[p + 'codes_dtype'] = rw.numpy().astype(np.int32), np.array(str(rw.dtype))
        out[p + 'rscales'], out[p + 'rzeros'] = f32(rs).reshape(-1), rz.numpy().astype(np.int32).reshape(-1)
        out[p + 'meta'] = np.array([bit, float(qmin), float(qmax)], dtype=np.float64)
        out[p + 'dt'] = np.array(dt)
    out['n'] = np.array(len(cases))
    save('quant_pt', **out)


def suite_pack():
    """VllmRealQuantLinear.pack and AutoawqRealQuantLinear.gemm_pack (never run by the reference's CI)."""
    from easydict import EasyDict
    from llmc.compression.quantization.module_utils import (AutoawqRealQuantLinear,
                                                            VllmRealQuantLinear)
    gen = torch.Generator().manual_seed(77)
    out = {}
    for ci, (bit, K) in enumerate([(4, 256), (8, 256), (4, 200), (8, 130)]):
        q = IntegerQuantizer(bit, True, 'per_channel')
        w = rand_weight(gen, 24, K, 'f16', outliers=False)
        codes, scales, _ = q.real_quant_weight_dynamic(w)
        cfg = EasyDict({'weight': {'bit': bit}})
        packed, ps = VllmRealQuantLinear.pack(codes, scales, cfg)
        out[f'v{ci}_codes'] = codes.numpy().astype(np.int32)
        out[f'v{ci}_packed'] = packed.numpy()
        out[f'v{ci}_bit'] = np.array(bit)
    out['n_vllm'] = np.array(4)

    for ci, (R, K, g) in enumerate([(64, 256, 128), (32, 384, 64)]):
        q = IntegerQuantizer(4, False, 'per_group', group_size=g)
        lin = torch.nn.Linear(K, R, bias=False).half()
        lin.weight.data = rand_weight(gen, R, K, 'f16')
        _, scales, zeros = q.real_quant_weight_dynamic(lin.weight.data)
        cfg = EasyDict({'weight': {'bit': 4, 'group_size': g, 'pack_version': 'gemm_pack'}})
        qw, sc, qz = AutoawqRealQuantLinear.gemm_pack(lin, lin.weight.data, scales, zeros, cfg)
        out[f'a{ci}_w'] = f32(lin.weight.data)
        out[f'a{ci}_scales'] = f32(scales)
        out[f'a{ci}_zeros'] = zeros.numpy().astype(np.int32)
        out[f'a{ci}_qweight'] = qw.numpy()
        out[f'a{ci}_qscales'] = f32(sc)
        out[f'a{ci}_qzeros'] = qz.numpy()
        out[f'a{ci}_g'] = np.array(g)
    out['n_awq'] = np.array(2)
    save('pack', **out)


def _gptq_instance(wq, actorder, static_groups, percdamp=0.01, blocksize=128, dtype=torch.float16):
    """A GPTQ object without the model plumbing: only what the numeric methods read."""
    from llmc.compression.quantization.gptq import GPTQ
    g = GPTQ.__new__(GPTQ)
    g.dev = torch.device('cpu')
    g.wquantizer = wq
    g.actorder = actorder
    g.static_groups = static_groups
    g.percdamp = percdamp
    g.blocksize = blocksize
    g.chunk_num = 1
    g.owq = False
    g.layers_cache = {}
    g.model_dtype = dtype
    g.need_perm = (wq.granularity == 'per_group' and not static_groups and actorder)
    g.act_static = False
    return g


GPTQ_CFGS = [
    # (name, bit, sym, gran, gs, actorder, static_groups, dtype, R, K, dead)
    ('asym_g128_act_dyn', 4, False, 'per_group', 128, True, False, 'bf16', 24, 384, True),
    ('sym_g128_act_static', 4, True, 'per_group', 128, True, True, 'f16', 32, 256, False),
    ('sym_pc_noact', 8, True, 'per_channel', None, False, False, 'f16', 32, 256, False),
    ('asym_g64_act_dyn', 4, False, 'per_group', 64, True, False, 'f16', 32, 256, False),
    ('asym_g128_noact_static', 4, False, 'per_group', 128, False, True, 'bf16', 32, 256, False),
]
# second file (round 3): other bit widths and group sizes, per-channel with actorder, 8-bit static groups
GPTQ_MORE_CFGS = [
    ('asym_g32_w3_act_dyn', 3, False, 'per_group', 32, True, False, 'bf16', 32, 256, False),
    ('sym_g128_w2_noact_dyn', 2, True, 'per_group', 128, False, False, 'f16', 32, 256, False),
    ('asym_pc_w4_act', 4, False, 'per_channel', None, True, False, 'bf16', 32, 384, True),
    ('sym_g64_w8_act_static', 8, True, 'per_group', 64, True, True, 'f16', 24, 256, False),
    ('asym_g16_w4_noact_dyn', 4, False, 'per_group', 16, False, False, 'bf16', 16, 256, False),
]


def suite_gptq():
    _gptq_suite(GPTQ_CFGS, 2024, 'gptq', True)


def suite_gptq_more():
    _gptq_suite(GPTQ_MORE_CFGS, 4048, 'gptq_more', False)


def _gptq_suite(cfgs, seed, fname, with_mm):
    """GPTQ.add_batch / process_hessian_and_weights / weight_transform / update_model_qparams / w_q / w_qdq
    on small seeded layers, several configurations."""
    import torch.distributed as dist
    if not dist.is_initialized():
        dist.init_process_group('gloo', init_method='tcp://127.0.0.1:29591', rank=0, world_size=1)
    out = {}
    gen = torch.Generator().manual_seed(seed)
    for (name, bit, sym, gran, gs, actorder, static_groups, dt, R, K, dead) in cfgs:
        kw = dict(group_size=gs) if gs else {}
        wq = IntegerQuantizer(bit, sym, gran, **kw)
        g = _gptq_instance(wq, actorder, static_groups, dtype=DT[dt])
        layer = torch.nn.Linear(K, R, bias=False).to(DT[dt])
        layer.weight.data = rand_weight(gen, R, K, dt)
        # collect_block_qparams (base_blockwise_quantization.py:338-365)
        _, s0, z0, qmax, qmin = wq.get_tensor_qparams(layer.weight.data)
        layer.register_buffer('buf_scales', s0.detach())
        layer.register_buffer('buf_zeros', z0.detach())
        layer.register_buffer('buf_qmax', torch.tensor(qmax))
        layer.register_buffer('buf_qmin', torch.tensor(qmin))
        lname = 'fc'
        g.layers_cache[lname] = {}
        g.layer_init(layer, lname)
        nb, T = 2, 96
        xs = []
        for b in range(nb):
            z = torch.randn(1, T, K, generator=gen)
            c = torch.exp(0.5 * torch.randn(K, generator=gen))
            x = z * c
            x[..., 5] *= 30
            if dead:
                x[..., 17] = 0
                x[..., 200] = 0
            x = x.to(DT[dt])
            xs.append(x)
            g.add_batch(layer, lname, x, None)
        H = g.layers_cache[lname]['H'].clone()
        g.initialize_qparams_and_prepare_weights(layer, lname)
        perm = g.perm.clone() if actorder else None
        W0 = layer.weight.data.clone()
        Wp, U = g.process_hessian_and_weights(layer, lname)
        Wp_in = Wp.clone()
        Losses = torch.zeros_like(Wp)
        tmp = torch.zeros_like(Wp)
        Wrun = Wp.clone()
        if wq.granularity == 'per_group' and not static_groups:
            pass
        g.weight_transform(Wrun, U, Losses, tmp)
        p = name + '/'
        if name in ('asym_g128_act_dyn', 'sym_g128_act_static'):
            out[p + 'x'] = np.stack([f32(x[0]) for x in xs])
            out[p + 'H'] = f32(H)
        out[p + 'W0'] = f32(W0)
        out[p + 'perm'] = perm.numpy().astype(np.int64) if perm is not None else np.zeros(0, np.int64)
        out[p + 'Wp'] = f32(Wp_in)
        out[p + 'U'] = f32(U)
        out[p + 'tmp'] = f32(tmp)
        out[p + 'losses'] = f32(Losses)
        if wq.granularity == 'per_group':
            out[p + 'g_scales'] = np.stack([f32(q['scale']).reshape(-1) for q in g.groups], axis=1)
            if not sym:
                out[p + 'g_zeros'] = np.stack([f32(q['zero']).reshape(-1) for q in g.groups], axis=1)
        else:
            out[p + 'g_scales'] = f32(g.qparams['scale']).reshape(-1, 1)
        # finish the layer the way update_layer_with_transformed_weights does (gptq.py:186-196)
        t2 = tmp.clone()
        if actorder:
            g.invperm = torch.argsort(g.perm)
            t2 = t2[:, g.invperm]
        layer.weight.data = t2.reshape(layer.weight.shape)
        if wq.granularity == 'per_group' and not static_groups:
            g.update_model_qparams(layer)
        out[p + 'final_w'] = f32(layer.weight.data)
        out[p + 'buf_scales'] = f32(layer.buf_scales).reshape(-1)
        out[p + 'buf_scales_dtype'] = np.array(str(layer.buf_scales.dtype))
        bz = layer.buf_zeros
        out[p + 'buf_zeros'] = f32(bz).reshape(-1) if bz.dim() > 0 else np.zeros(0, np.float32)
        fq = g.w_qdq(layer, wq)
        out[p + 'w_qdq'] = f32(fq)
        out[p + 'w_qdq_dtype'] = np.array(str(fq.dtype))
        if not g.need_perm:
            cw, cs, cz = g.w_q(layer, wq)
            out[p + 'w_q_codes'] = cw.numpy().astype(np.int32)
            out[p + 'w_q_scales'] = f32(cs)
            out[p + 'w_q_zeros'] = cz.numpy().astype(np.int32) if cz is not None else np.zeros(0, np.int32)
        out[p + 'meta'] = np.array([bit, int(sym), gs or 0, int(actorder), int(static_groups), R, K,
                                    float(qmin), float(qmax)], dtype=np.float64)
        out[p + 'dt'] = np.array(dt)
        out[p + 'gran'] = np.array(gran)
    out['names'] = np.array([c[0] for c in cfgs])

    if with_mm:
        # sgemm order pin: MKL result of Err1.matmul(Hinv[i1:i2, i2:]) on a realistic block
        a = torch.randn(96, 128, generator=gen) * 0.01
        b = torch.randn(128, 200, generator=gen)
        out['mm_a'], out['mm_b'], out['mm_out'] = f32(a), f32(b), f32(a.matmul(b))
    save(fname, **out)


def suite_gptq_owq():
    """GPTQ with OWQ (gptq.py:44-56, 66-83, 199-244): the n_out columns with the largest Hessian diagonal go last, stay
    in floating point and only receive the error feedback; groups are clipped at columns - n_out."""
    import torch.distributed as dist
    if not dist.is_initialized():
        dist.init_process_group('gloo', init_method='tcp://127.0.0.1:29594', rank=0, world_size=1)
    out = {}
    cfgs = [('asym_g128_owq6', 4, False, 'per_group', 128, 'bf16', 24, 256, 6),
            ('sym_g64_owq16', 4, True, 'per_group', 64, 'f16', 32, 256, 16),
            ('asym_pc_owq8', 4, False, 'per_channel', None, 'f16', 16, 256, 8)]
    gen = torch.Generator().manual_seed(31337)
    for (name, bit, sym, gran, gs, dt, R, K, n_out) in cfgs:
        kw = dict(group_size=gs) if gs else {}
        wq = IntegerQuantizer(bit, sym, gran, **kw)
        g = _gptq_instance(wq, False, False, dtype=DT[dt])
        g.owq, g.need_perm, g.n_out_dict = True, True, {'fc': n_out}
        layer = torch.nn.Linear(K, R, bias=False).to(DT[dt])
        layer.weight.data = rand_weight(gen, R, K, dt)
        _, s0, z0, qmax, qmin = wq.get_tensor_qparams(layer.weight.data)
        layer.register_buffer('buf_scales', s0.detach())
        layer.register_buffer('buf_zeros', z0.detach())
        layer.register_buffer('buf_qmax', torch.tensor(qmax))
        layer.register_buffer('buf_qmin', torch.tensor(qmin))
        g.layers_cache['fc'] = {}
        g.layer_init(layer, 'fc')
        xs = []
        for b in range(2):
            x = (torch.randn(1, 96, K, generator=gen) * torch.exp(0.5 * torch.randn(K, generator=gen)))
            x[..., 5] *= 30
            x[..., 77] *= 12
            x = x.to(DT[dt])
            xs.append(x)
            g.add_batch(layer, 'fc', x, None)
        H = g.layers_cache['fc']['H'].clone()
        rtn_s, rtn_z = layer.buf_scales.clone(), layer.buf_zeros.clone()
        g.initialize_qparams_and_prepare_weights(layer, 'fc')
        W0 = layer.weight.data.clone()
        Wp, U = g.process_hessian_and_weights(layer, 'fc')
        Wp_in = Wp.clone()
        Losses, tmp, Wrun = torch.zeros_like(Wp), torch.zeros_like(Wp), Wp.clone()
        g.weight_transform(Wrun, U, Losses, tmp)
        p = name + '/'
        out[p + 'W0'] = f32(W0)
        out[p + 'Hdiag'] = f32(torch.diag(H))
        out[p + 'perm'] = g.perm.numpy().astype(np.int64)
        out[p + 'Wp'], out[p + 'U'] = f32(Wp_in), f32(U)
        out[p + 'tmp'], out[p + 'losses'], out[p + 'W_after'] = f32(tmp), f32(Losses), f32(Wrun)
        out[p + 'rtn_scales'] = f32(rtn_s).reshape(-1)
        out[p + 'rtn_zeros'] = f32(rtn_z).reshape(-1) if rtn_z.dim() > 0 else np.zeros(0, np.float32)
        # finish the layer like update_layer_with_transformed_weights (gptq.py:186-196)
        t2 = tmp.clone()
        t2[:, g.n_nonout:] = Wrun[:, g.n_nonout:]
        t2 = t2[:, g.invperm]
        layer.weight.data = t2.reshape(layer.weight.shape)
        if gran == 'per_group':
            g.update_model_qparams(layer)
        out[p + 'final_w'] = f32(layer.weight.data)
        out[p + 'buf_scales'] = f32(layer.buf_scales).reshape(-1)
        out[p + 'buf_scales_dtype'] = np.array(str(layer.buf_scales.dtype))
        bz = layer.buf_zeros
        out[p + 'buf_zeros'] = f32(bz).reshape(-1) if bz.dim() > 0 else np.zeros(0, np.float32)
        fq = g.w_qdq(layer, wq)
        out[p + 'w_qdq'], out[p + 'w_qdq_dtype'] = f32(fq), np.array(str(fq.dtype))
        out[p + 'meta'] = np.array([bit, int(sym), gs or 0, R, K, n_out, float(qmin), float(qmax)], dtype=np.float64)
        out[p + 'dt'], out[p + 'gran'] = np.array(dt), np.array(gran)
    out['names'] = np.array([c[0] for c in cfgs])
    save('gptq_owq', **out)


AWQ_CFGS = [('bf16_sym_g128_v2', 'bf16', True, 128, 'v2', [64, 32], 4), ('f16_asym_g128_v2', 'f16', False, 128, 'v2', [48], 4),
            ('bf16_sym_g64_v1', 'bf16', True, 64, 'v1', [32, 32], 4)]
# second file (round 3): other bit widths, per-channel (gs = 0), three stacked layers
AWQ_MORE_CFGS = [('f16_asym_g32_w3_v2', 'f16', False, 32, 'v2', [32, 32], 3), ('bf16_sym_pc_w8_v2', 'bf16', True, 0, 'v2', [64], 8),
                 ('f16_sym_g128_w4_v1_3l', 'f16', True, 128, 'v1', [32, 16, 16], 4), ('bf16_asym_g64_w2_v2', 'bf16', False, 64, 'v2', [48], 2)]


def suite_awq():
    _awq_suite(AWQ_CFGS, 4242, 'awq')


def suite_awq_more():
    _awq_suite(AWQ_MORE_CFGS, 8484, 'awq_more')


def _awq_suite(cfgs, seed, fname):
    """Awq.search_scale_subset (20-point grid, one batch) with inspect = the stacked Linear layers."""
    import torch.distributed as dist
    import types
    from llmc.compression.quantization.awq import Awq
    from llmc.compression.quantization.base_blockwise_quantization import BaseBlockwiseQuantization
    if not dist.is_initialized():
        dist.init_process_group('gloo', init_method='tcp://127.0.0.1:29592', rank=0, world_size=1)

    class Stacked(torch.nn.Module):
        def __init__(self, layers):
            super().__init__()
            self.layers = torch.nn.ModuleList(layers)

        def forward(self, x):
            return torch.cat([l(x) for l in self.layers], dim=-1)

    # On a GPU `org_sd = {k: v.cpu() ...}` (awq.py:199) is a COPY of the weights; on CPU `.cpu()` aliases them and
    # the in-place `mul_` of grid step 0 corrupts the saved originals whenever scales(ratio=0) != 1 (trans v1).
    # Keep the GPU semantics the reference is written for: make .cpu() copy while this suite runs.
    torch.Tensor.cpu = lambda self, *a, **k: self.clone()
    out = {}
    gen = torch.Generator().manual_seed(seed)
    for (name, dt, sym, gs, ver, Rs, bit) in cfgs:
        K, N = 256, 192
        wq = IntegerQuantizer(bit, sym, 'per_group', group_size=gs) if gs else IntegerQuantizer(bit, sym, 'per_channel')
        a = Awq.__new__(Awq)
        a.wquantizer = wq
        a.aquantizer = None
        a.w_only = True
        a.awq_bs = None
        a.save_mem = False
        a.padding_mask = None
        a.trans_version = ver
        a.n_samples = 2
        a.has_gqa = False
        a.do_gqa_trans = False
        layers = []
        for R in Rs:
            l = torch.nn.Linear(K, R, bias=False).to(DT[dt])
            wt = torch.randn(R, K, generator=gen) * 0.02
            wt[:, torch.randperm(K, generator=gen)[:4]] *= 20
            l.weight.data = wt.to(DT[dt])
            layers.append(l)
        z = torch.randn(2, N // 2, K, generator=gen)
        c = torch.exp(0.5 * torch.randn(K, generator=gen))
        idx = torch.randperm(K, generator=gen)[:8]
        c[idx] *= 100.0
        x = (z * c).to(DT[dt])
        losses = []
        orig = a.calculate_loss

        def rec(org_out, o, _orig=orig):
            v = _orig(org_out, o)
            losses.append(v)
            return v
        a.calculate_loss = rec
        w0 = [l.weight.data.clone() for l in layers]
        layers_dict = {f'l{i}': l for i, l in enumerate(layers)}
        w_max = a.get_weight_scale(layers_dict)
        a._bs = x.shape[0]
        x_mean = a.get_act_scale(x)
        s10 = a.get_scales(None, x, w_max, False, 0.5)
        best = a.search_scale_subset(None, layers_dict, [x], Stacked(layers), False, {})
        for l, w in zip(layers, w0):
            assert torch.equal(l.weight.data, w), 'reference must restore the weights'
        p = name + '/'
        for i, w in enumerate(w0):
            out[p + f'w{i}'] = f32(w)
        out[p + 'x'] = f32(x)
        out[p + 'w_max'] = f32(w_max)
        out[p + 'x_mean'] = f32(x_mean)
        out[p + 'scales_r050'] = f32(s10)
        out[p + 'best_scales'] = f32(best)
        out[p + 'losses'] = np.array(losses, dtype=np.float64)
        # one explicit grid point for the elementwise chain
        s = a.get_scales(None, x, w_max, False, 0.35)
        wqs = torch.cat([wq.fake_quant_weight_dynamic(w.clone().mul_(s.view(1, -1))) for w in w0], dim=0)
        xs = x / s.view(1, -1)
        out[p + 'scales_r035'] = f32(s)
        out[p + 'wq_r035'] = f32(wqs)
        out[p + 'xs_r035'] = f32(xs)
        out[p + 'meta'] = np.array([int(sym), gs, len(Rs), K] + ([bit] if fname != 'awq' else []), dtype=np.int64)
        out[p + 'dt'] = np.array(dt)
        out[p + 'ver'] = np.array(ver)
    out['names'] = np.array([c[0] for c in cfgs])
    save(fname, **out)


def suite_awq_flat():
    """Awq.search_scale_subset on data whose loss curve has TWO NEAR-EQUAL MINIMA (second best within ~1e-3 of the best):
    the argmin has to come out of loss values that agree with the reference's far better than the 2 % tolerance the
    round-2 tests allowed. The data is found by scanning seeds with the oracle's restatement; the golden values are the
    reference's own."""
    import torch.distributed as dist
    from llmc.compression.quantization.awq import Awq
    sys.path.insert(0, ROOT)
    from oracle import awq_ref as A
    from oracle import quant_ref as Q
    if not dist.is_initialized():
        dist.init_process_group('gloo', init_method='tcp://127.0.0.1:29597', rank=0, world_size=1)

    class Stacked(torch.nn.Module):
        def __init__(self, layers):
            super().__init__()
            self.layers = torch.nn.ModuleList(layers)

        def forward(self, x):
            return torch.cat([l(x) for l in self.layers], dim=-1)

    torch.Tensor.cpu = lambda self, *a, **k: self.clone()      # GPU semantics of `org_sd = {k: v.cpu()}` (see suite_awq)
    dt, sym, gs, ver, Rs, K, N = 'bf16', True, 128, 'v2', [64, 32], 256, 192
    qmin, qmax = Q.int_range(4, sym)
    found = None
    for seed in range(2000):
        gen = torch.Generator().manual_seed(90000 + seed)
        ws = []
        for R in Rs:
            wt = torch.randn(R, K, generator=gen) * 0.02
            wt[:, torch.randperm(K, generator=gen)[:4]] *= 6
            ws.append(wt.to(DT[dt]))
        c = torch.exp(0.5 * torch.randn(K, generator=gen))
        c[torch.randperm(K, generator=gen)[:8]] *= 12.0
        x = (torch.randn(2, N // 2, K, generator=gen) * c).to(DT[dt])
        _, losses, n = A.search_scale([f32(w) for w in ws], f32(x), dt, sym, qmin, qmax, gs, ver)
        srt = np.sort(losses)
        gap = (srt[1] - srt[0]) / srt[0]
        if 2e-4 < gap < 1.5e-3 and 0 < n < 19:
            found = (seed, ws, x, gap, n)
            break
    assert found is not None, 'no seed with two near-equal minima'
    seed, ws, x, gap, n = found
    wq = IntegerQuantizer(4, sym, 'per_group', group_size=gs)
    a = Awq.__new__(Awq)
    a.wquantizer, a.aquantizer, a.w_only, a.awq_bs, a.save_mem, a.padding_mask = wq, None, True, None, False, None
    a.trans_version, a.n_samples, a.has_gqa, a.do_gqa_trans = ver, 2, False, False
    layers = []
    for w in ws:
        l = torch.nn.Linear(K, w.shape[0], bias=False).to(DT[dt])
        l.weight.data = w.clone()
        layers.append(l)
    losses = []
    orig = a.calculate_loss

    def rec(org_out, o, _orig=orig):
        v = _orig(org_out, o)
        losses.append(v)
        return v
    a.calculate_loss = rec
    a._bs = x.shape[0]
    best = a.search_scale_subset(None, {f'l{i}': l for i, l in enumerate(layers)}, [x], Stacked(layers), False, {})
    ref = np.array(losses, dtype=np.float64)
    srt = np.sort(ref)
    out = {'names': np.array(['bf16_sym_g128_v2_flat'])}
    p = 'bf16_sym_g128_v2_flat/'
    for i, w in enumerate(ws):
        out[p + f'w{i}'] = f32(w)
    out[p + 'x'] = f32(x)
    out[p + 'best_scales'] = f32(best)
    out[p + 'losses'] = ref
    out[p + 'gap'] = np.array((srt[1] - srt[0]) / srt[0])
    out[p + 'meta'] = np.array([int(sym), gs, len(Rs), K], dtype=np.int64)
    out[p + 'dt'] = np.array(dt)
    out[p + 'ver'] = np.array(ver)
    print('seed', seed, 'oracle gap', gap, 'reference gap', float(out[p + 'gap']), 'argmin', int(np.argmin(ref)))
    save('awq_flat', **out)


def suite_awq_inspect():
    """Awq.search_scale_subset with an inspected module that is NOT the Linear layers themselves (the Llama gate/up
    subset inspects the whole MLP, llmc/models/llama.py:79), two calibration batches (per-batch best bookkeeping,
    awq.py:242-248) and a padding mask (awq.py:229-231)."""
    import torch.distributed as dist
    from llmc.compression.quantization.awq import Awq
    if not dist.is_initialized():
        dist.init_process_group('gloo', init_method='tcp://127.0.0.1:29593', rank=0, world_size=1)

    class MLP(torch.nn.Module):
        def __init__(self, K, R, dt):
            super().__init__()
            self.gate_proj = torch.nn.Linear(K, R, bias=False).to(dt)
            self.up_proj = torch.nn.Linear(K, R, bias=False).to(dt)
            self.down_proj = torch.nn.Linear(R, K, bias=False).to(dt)

        def forward(self, x):
            return self.down_proj(torch.nn.functional.silu(self.gate_proj(x)) * self.up_proj(x))

    torch.Tensor.cpu = lambda self, *a, **k: self.clone()      # GPU semantics of `org_sd = {k: v.cpu()}` (see suite_awq)
    out = {}
    gen = torch.Generator().manual_seed(777)
    cfgs = [('bf16_sym_g128_mlp_2batch_mask', 'bf16', True, 128, 'v2', True),
            ('f16_asym_g64_mlp_2batch', 'f16', False, 64, 'v2', False)]
    for name, dt, sym, gs, ver, use_mask in cfgs:
        K, R, B, S = 256, 192, 2, 48
        wq = IntegerQuantizer(4, sym, 'per_group', group_size=gs)
        a = Awq.__new__(Awq)
        a.wquantizer, a.aquantizer, a.w_only, a.awq_bs, a.save_mem = wq, None, True, None, False
        a.trans_version, a.n_samples, a.has_gqa, a.do_gqa_trans = ver, 2 * B, False, False
        mlp = MLP(K, R, DT[dt])
        for l in (mlp.gate_proj, mlp.up_proj, mlp.down_proj):
            wt = torch.randn(l.weight.shape, generator=gen) * 0.05
            if l is not mlp.down_proj:
                wt[:, torch.randperm(K, generator=gen)[:4]] *= 10
            l.weight.data = wt.to(DT[dt])
        c = torch.exp(0.5 * torch.randn(K, generator=gen))
        c[torch.randperm(K, generator=gen)[:8]] *= 30.0
        xs = [(torch.randn(B, S, K, generator=gen) * c).to(DT[dt]) for _ in range(2)]
        masks = None
        if use_mask:
            masks = [(torch.rand(B, S, generator=gen) > 0.2).to(torch.int64) for _ in range(2)]
        a.padding_mask = masks
        losses = []
        orig = a.calculate_loss

        def rec(org_out, o, _orig=orig):
            v = _orig(org_out, o)
            losses.append(v)
            return v
        a.calculate_loss = rec
        layers_dict = {'gate_proj': mlp.gate_proj, 'up_proj': mlp.up_proj}
        w0 = {n: l.weight.data.clone() for n, l in mlp.named_modules() if isinstance(l, torch.nn.Linear)}
        best = a.search_scale_subset(None, layers_dict, [x.clone() for x in xs], mlp, False, {})
        for n, l in mlp.named_modules():
            if isinstance(l, torch.nn.Linear):
                assert torch.equal(l.weight.data, w0[n]), 'reference must restore the weights'
        p = name + '/'
        for n, w in w0.items():
            out[p + 'w_' + n] = f32(w)
        for i, x in enumerate(xs):
            out[p + f'x{i}'] = f32(x)
            if masks is not None:
                out[p + f'mask{i}'] = masks[i].numpy()
        out[p + 'best_scales'] = f32(best)
        out[p + 'losses'] = np.array(losses, dtype=np.float64)          # order: grid point major, batch minor
        out[p + 'meta'] = np.array([int(sym), gs, K, R, int(use_mask)], dtype=np.int64)
        out[p + 'dt'] = np.array(dt)
        out[p + 'ver'] = np.array(ver)
    out['names'] = np.array([c[0] for c in cfgs])
    save('awq_inspect', **out)


def suite_awq_wa():
    """Awq.search_scale_subset with ACTIVATION quantization (`not self.w_only`: fake_quantize_input, awq.py:166-177,
    223-224) and with weight quantizers the W4A16 goldens do not touch: integer per_channel / per_tensor, FP8 e4m3 / e5m2
    with float_quantize = the restated qtorch (suite_fp8_qtorch). The configuration awq_fp8_static.yml describes
    (BASELINE configs[4]'s parent: FP8 per_tensor weights and activations) is the third case. awq_bs = 1 on a two-sample
    batch takes the per-sample branch of fake_quantize_input, where a per_tensor activation range is per sample."""
    import torch.distributed as dist
    import llmc.compression.quantization.quant as qmod
    from llmc.compression.quantization.awq import Awq
    qmod.float_quantize = _qtorch_stub
    if not dist.is_initialized():
        dist.init_process_group('gloo', init_method='tcp://127.0.0.1:29594', rank=0, world_size=1)

    class MLP(torch.nn.Module):
        def __init__(self, K, R, dt):
            super().__init__()
            self.gate_proj = torch.nn.Linear(K, R, bias=False).to(dt)
            self.up_proj = torch.nn.Linear(K, R, bias=False).to(dt)
            self.down_proj = torch.nn.Linear(R, K, bias=False).to(dt)

        def forward(self, x):
            return self.down_proj(torch.nn.functional.silu(self.gate_proj(x)) * self.up_proj(x))

    torch.Tensor.cpu = lambda self, *a, **k: self.clone()      # GPU semantics of `org_sd = {k: v.cpu()}` (see suite_awq)
    out = {}
    gen = torch.Generator().manual_seed(4711)
    # name, dtype, weight quantizer (kind, bit, sym, granularity, group), act quantizer (kind, bit, sym, granularity), inspect, awq_bs, batches
    cfgs = [('w8a8_int_pc_pt_mlp', 'bf16', ('int', 8, True, 'per_channel', 0), ('int', 8, True, 'per_token'), 'mlp', None, 1),
            ('w4a8_int_g64_asym_ptensor_bs1', 'f16', ('int', 4, False, 'per_group', 64), ('int', 8, True, 'per_tensor'), 'mlp', 1, 2),
            ('fp8_e4m3_ptensor_static_yml', 'bf16', ('float', 'e4m3', True, 'per_tensor', 0), ('float', 'e4m3', True, 'per_tensor'), 'linear', None, 1),
            ('fp8_e5m2_pc_pt', 'bf16', ('float', 'e5m2', True, 'per_channel', 0), ('float', 'e5m2', True, 'per_token'), 'mlp', None, 1),
            ('w8a8_int_ptensor_weights', 'bf16', ('int', 8, True, 'per_tensor', 0), ('int', 8, False, 'per_token'), 'linear', None, 1)]

    def make(kind, bit, sym, gran, gs=0):
        if kind == 'int':
            return IntegerQuantizer(bit, sym, gran, group_size=gs) if gs else IntegerQuantizer(bit, sym, gran)
        return qmod.FloatQuantizer(bit, sym, gran, use_qtorch=True)

    for name, dt, wcfg, acfg, inspect, awq_bs, nb in cfgs:
        K, R, B, S = 256, 192, 2, 48
        a = Awq.__new__(Awq)
        a.wquantizer, a.aquantizer, a.w_only, a.awq_bs, a.save_mem = make(*wcfg), make(*acfg), False, awq_bs, False
        a.trans_version, a.n_samples, a.has_gqa, a.do_gqa_trans, a.padding_mask = 'v2', nb * B, False, False, None
        mlp = MLP(K, R, DT[dt])
        for l in (mlp.gate_proj, mlp.up_proj, mlp.down_proj):
            wt = torch.randn(l.weight.shape, generator=gen) * 0.05
            if l is not mlp.down_proj:
                wt[:, torch.randperm(K, generator=gen)[:4]] *= 10
            l.weight.data = wt.to(DT[dt])
        c = torch.exp(0.5 * torch.randn(K, generator=gen))
        c[torch.randperm(K, generator=gen)[:8]] *= 30.0
        xs = [(torch.randn(B, S, K, generator=gen) * c).to(DT[dt]) for _ in range(nb)]
        losses = []
        orig = a.calculate_loss

        def rec(org_out, o, _orig=orig):
            v = _orig(org_out, o)
            losses.append(v)
            return v
        a.calculate_loss = rec
        if inspect == 'mlp':
            layers_dict, module = {'gate_proj': mlp.gate_proj, 'up_proj': mlp.up_proj}, mlp
        else:
            layers_dict, module = {'gate_proj': mlp.gate_proj}, mlp.gate_proj
        w0 = {n: l.weight.data.clone() for n, l in mlp.named_modules() if isinstance(l, torch.nn.Linear)}
        best = a.search_scale_subset(None, layers_dict, [x.clone() for x in xs], module, False, {})
        for n, l in mlp.named_modules():
            if isinstance(l, torch.nn.Linear):
                assert torch.equal(l.weight.data, w0[n]), 'reference must restore the weights'
        # one explicit grid point of the chain: scaled + fake-quantized weight and input
        a._bs = xs[0].shape[0] if awq_bs is None else awq_bs
        sc = a.get_scales(None, xs[0], a.get_weight_scale(layers_dict), False, 0.4)
        wq04 = a.wquantizer.fake_quant_weight_dynamic(w0['gate_proj'].clone().mul_(sc.view(1, -1)))
        xq04 = a.fake_quantize_input(xs[0] / sc.view(1, -1), layers_dict)
        p = name + '/'
        for n, w in w0.items():
            out[p + 'w_' + n] = f32(w)
        for i, x in enumerate(xs):
            out[p + f'x{i}'] = f32(x)
        out[p + 'best_scales'] = f32(best)
        out[p + 'scales_r040'] = f32(sc)
        out[p + 'wq_r040'] = f32(wq04)
        out[p + 'xq_r040'] = f32(xq04)
        out[p + 'losses'] = np.array(losses, dtype=np.float64)
        out[p + 'wcfg'] = np.array([str(v) for v in wcfg])
        out[p + 'acfg'] = np.array([str(v) for v in acfg])
        out[p + 'meta'] = np.array([K, R, nb, 0 if awq_bs is None else awq_bs], dtype=np.int64)
        out[p + 'inspect'] = np.array(inspect)
        out[p + 'dt'] = np.array(dt)
    out['names'] = np.array([c[0] for c in cfgs])
    save('awq_wa', **out)


def suite_awq_gqa():
    """special.do_gqa_trans (awq.py:88-108, 338-365; base_blockwise_quantization.py:591-595, 678-685, 877-897): the v_proj -> o_proj
    subset of a GQA attention (fewer key/value heads than query heads). The scales live on v_proj's output channels and are
    repeated per query-head group for o_proj; the search runs on the inputs of the PREVIOUS subset (q/k/v's), as the reference
    does. Goldens: the 20 x batches losses, the best scales, both weights after apply_scale, o_proj's inputs after
    update_input_feat."""
    import torch.distributed as dist
    from llmc.compression.quantization.awq import Awq
    if not dist.is_initialized():
        dist.init_process_group('gloo', init_method='tcp://127.0.0.1:29594', rank=0, world_size=1)
    torch.Tensor.cpu = lambda self, *a, **k: self.clone()
    out = {}
    gen = torch.Generator().manual_seed(4242)
    cfgs = [('bf16_sym_g128', 'bf16', True, 128, True), ('f16_asym_g64_2batch', 'f16', False, 64, False)]
    for name, dt, sym, gs, one_batch in cfgs:
        H, NH, NKV, HD, B, S = 256, 8, 2, 32, 2, 40
        wq = IntegerQuantizer(4, sym, 'per_group', group_size=gs)
        a = Awq.__new__(Awq)
        a.wquantizer, a.aquantizer, a.w_only, a.awq_bs, a.save_mem = wq, None, True, None, False
        a.trans_version, a.has_gqa, a.do_gqa_trans, a.padding_mask = 'v2', True, True, None
        a.num_key_value_heads, a.head_dim, a.num_key_value_groups = NKV, HD, NH // NKV
        a.fp8_block_size = 128
        nb = 1 if one_batch else 2
        a.n_samples = nb * B
        v_proj = torch.nn.Linear(H, NKV * HD, bias=True).to(DT[dt])
        o_proj = torch.nn.Linear(NH * HD, H, bias=False).to(DT[dt])
        v_proj.weight.data = (torch.randn(NKV * HD, H, generator=gen) * 0.05 * torch.exp(0.7 * torch.randn(NKV * HD, 1, generator=gen))).to(DT[dt])
        v_proj.bias.data = (torch.randn(NKV * HD, generator=gen) * 0.02).to(DT[dt])
        wo = torch.randn(H, NH * HD, generator=gen) * 0.05
        wo[:, torch.randperm(NH * HD, generator=gen)[:6]] *= 8
        o_proj.weight.data = wo.to(DT[dt])
        c = torch.exp(0.5 * torch.randn(H, generator=gen))
        xs = [(torch.randn(B, S, H, generator=gen) * c).to(DT[dt]) for _ in range(nb)]                 # q/k/v's input
        xo = [(torch.randn(B, S, NH * HD, generator=gen)).to(DT[dt]) for _ in range(nb)]             # o_proj's input
        losses = []
        orig = a.calculate_loss

        def rec(org_out, o, _orig=orig):
            v = _orig(org_out, o)
            losses.append(v)
            return v
        a.calculate_loss = rec
        p = name + '/'
        out[p + 'v_w'], out[p + 'v_b'], out[p + 'o_w'] = f32(v_proj.weight.data), f32(v_proj.bias.data), f32(o_proj.weight.data)
        for i in range(nb):
            out[p + f'x{i}'], out[p + f'xo{i}'] = f32(xs[i]), f32(xo[i])
        scale = a.search_scale_subset(v_proj, {'o_proj': o_proj}, [x.clone() for x in xs], o_proj, True, {})
        a.apply_scale(scale, [v_proj], [o_proj])
        feat = {'o_proj': [x.clone() for x in xo]}
        a.update_input_feat(scale, feat, {'o_proj': o_proj}, True)
        out[p + 'best_scales'] = f32(scale)
        out[p + 'losses'] = np.array(losses, dtype=np.float64)
        out[p + 'v_w_after'], out[p + 'v_b_after'], out[p + 'o_w_after'] = f32(v_proj.weight.data), f32(v_proj.bias.data), f32(o_proj.weight.data)
        for i in range(nb):
            out[p + f'xo{i}_after'] = f32(feat['o_proj'][i])
        out[p + 'meta'] = np.array([int(sym), gs, H, NH, NKV, HD, nb], dtype=np.int64)
        out[p + 'dt'] = np.array(dt)
    out['names'] = np.array([c[0] for c in cfgs])
    save('awq_gqa', **out)


CLIP_CFGS = [('bf16_sym_g128_clipsym', 'bf16', True, 128, True, 4, 96, 32), ('f16_asym_g128_noclipsym', 'f16', False, 128, False, 4, 96, 32),
             ('f16_sym_g64_clipsym', 'f16', True, 64, True, 4, 96, 32)]
# second file (round 3): other bit widths, group 32, more sampled tokens (several 16-token chunks of ATen's cascade sum),
# an asymmetric quantizer with symmetric clipping
CLIP_MORE_CFGS = [('bf16_asym_g32_w3_noclipsym', 'bf16', False, 32, False, 3, 96, 32), ('f16_sym_g128_w2_clipsym', 'f16', True, 128, True, 2, 96, 32),
                  ('bf16_sym_g64_w8_clipsym_200tok', 'bf16', True, 64, True, 8, 400, 200), ('f16_asym_g128_w4_clipsym_77tok', 'f16', False, 128, True, 4, 154, 77)]


def suite_clip():
    _clip_suite(CLIP_CFGS, 99, 'clip')


def suite_clip_more():
    _clip_suite(CLIP_MORE_CFGS, 1999, 'clip_more')


def _clip_suite(cfgs, seed, fname):
    """AutoClipper.auto_clip_layer / apply_clip (clip_version v1, w_only)."""
    from llmc.compression.quantization.auto_clip import AutoClipper
    out = {}
    gen = torch.Generator().manual_seed(seed)
    for name, dt, sym, gs, clip_sym, bit, T, nst in cfgs:
        R, K = 64, 256
        wq = IntegerQuantizer(bit, sym, 'per_group', group_size=gs)
        ac = AutoClipper(w_only=True, wquantizer=wq, aquantizer=None, clip_version='v1', clip_sym=clip_sym,
                         save_clip=False, padding_mask=None)
        wt = torch.randn(R, K, generator=gen) * 0.02
        wt[torch.rand(R, K, generator=gen) < 0.01] *= 8       # weight outliers make clipping worthwhile
        w = wt.to(DT[dt])
        x = (torch.randn(2, T // 2, K, generator=gen) * torch.exp(0.5 * torch.randn(K, generator=gen))).to(DT[dt])
        mx, mn = ac.auto_clip_layer(0, 'fc', w, [x.clone()], n_sample_token=nst)
        layer = torch.nn.Linear(K, R, bias=False).to(DT[dt])
        layer.weight.data = w.clone()
        ac.apply_clip(0, layer, mn, mx, 'fc')
        p = name + '/'
        out[p + 'w'], out[p + 'x'] = f32(w), f32(x)
        out[p + 'best_max'], out[p + 'best_min'] = f32(mx), f32(mn)
        out[p + 'clipped'] = f32(layer.weight.data)
        out[p + 'meta'] = np.array([int(sym), gs, int(clip_sym), nst] + ([bit] if fname != 'clip' else []), dtype=np.int64)
        out[p + 'dt'] = np.array(dt)
    out['names'] = np.array([c[0] for c in cfgs])
    save(fname, **out)


def suite_clip_mb():
    """AutoClipper.auto_clip_layer called with SEVERAL batches (auto_clip.py:130-184: err_mean over the list) — run() itself
    always concatenates, so this pins the list form of the method's own surface."""
    from llmc.compression.quantization.auto_clip import AutoClipper
    out = {}
    gen = torch.Generator().manual_seed(123)
    cfgs = [('bf16_sym_g128_clipsym_3b', 'bf16', True, 128, True, 3), ('f16_asym_g64_noclipsym_2b', 'f16', False, 64, False, 2)]
    for name, dt, sym, gs, clip_sym, nb in cfgs:
        R, K, T = 64, 256, 80
        wq = IntegerQuantizer(4, sym, 'per_group', group_size=gs)
        ac = AutoClipper(w_only=True, wquantizer=wq, aquantizer=None, clip_version='v1', clip_sym=clip_sym,
                         save_clip=False, padding_mask=None)
        wt = torch.randn(R, K, generator=gen) * 0.02
        wt[torch.rand(R, K, generator=gen) < 0.01] *= 8
        w = wt.to(DT[dt])
        c = torch.exp(0.5 * torch.randn(K, generator=gen))
        xs = [(torch.randn(2, T // 2, K, generator=gen) * c).to(DT[dt]) for _ in range(nb)]
        mx, mn = ac.auto_clip_layer(0, 'fc', w, [x.clone() for x in xs], n_sample_token=20)
        p = name + '/'
        out[p + 'w'] = f32(w)
        for i, x in enumerate(xs):
            out[p + f'x{i}'] = f32(x)
        out[p + 'best_max'], out[p + 'best_min'] = f32(mx), f32(mn)
        out[p + 'meta'] = np.array([int(sym), gs, int(clip_sym), 20, nb], dtype=np.int64)
        out[p + 'dt'] = np.array(dt)
    out['names'] = np.array([c[0] for c in cfgs])
    save('clip_mb', **out)


def suite_clip_wide():
    """AutoClipper.auto_clip_layer beyond W4A16 per_group: per_channel and per_tensor ranges (one group as wide as the
    row), activation quantization (`fake_quantize_input`, auto_clip.py:276-281: the [1, tok, ng, g] VIEW is what the
    activation quantizer sees, so per_token ranges are per token AND group), FP8 quantizers (float_quantize = the restated
    qtorch) and clip_version v2 (learnable range by logit factors, auto_clip.py:262-272). Besides the searched ranges the
    candidates the reference formed are recorded (fake-quantized weights of every shrink level, quantized input): they pin
    the error-table kernel separately from the host's candidate construction. K = 1152 makes the per-stream cascade of
    ATen's row sum flush (18 vectors per stream); K = 488 has vectors past the last four and trailing elements."""
    import llmc.compression.quantization.quant as qmod
    from llmc.compression.quantization.auto_clip import AutoClipper
    qmod.float_quantize = _qtorch_stub
    out = {}
    gen = torch.Generator().manual_seed(2024)
    # name, dtype, R, K, weight (kind, bit, sym, gran, gs, calib), act (kind, bit, sym, gran) | None, version, clip_sym, tokens, n_sample_token
    cfgs = [('bf16_w4_pc_sym_v1_k1152', 'bf16', 64, 1152, ('int', 4, True, 'per_channel', 0, 'minmax'), None, 'v1', True, 154, 77),
            ('f16_w8a8_pc_pt_v1', 'f16', 64, 640, ('int', 8, True, 'per_channel', 0, 'minmax'), ('int', 8, True, 'per_token'), 'v1', True, 96, 32),
            ('bf16_fp8_e4m3_ptensor_wa_v1', 'bf16', 128, 512, ('float', 'e4m3', True, 'per_tensor', 0, 'minmax'), ('float', 'e4m3', True, 'per_tensor'), 'v1', True, 96, 32),
            ('bf16_w4a8_pc_learnable_v2_sym', 'bf16', 64, 512, ('int', 4, True, 'per_channel', 0, 'learnable'), ('int', 8, True, 'per_token'), 'v2', True, 96, 32),
            ('f16_w4_pc_learnable_v2_asym', 'f16', 64, 256, ('int', 4, False, 'per_channel', 0, 'learnable'), None, 'v2', False, 96, 32),
            ('bf16_w4a8_g64_pt_v1', 'bf16', 64, 256, ('int', 4, False, 'per_group', 64, 'minmax'), ('int', 8, True, 'per_token'), 'v1', False, 96, 32),
            ('f16_w3_pc_asym_v1_k488', 'f16', 64, 488, ('int', 3, False, 'per_channel', 0, 'minmax'), None, 'v1', False, 200, 100)]

    def make(kind, bit, sym, gran, gs=0, calib='minmax'):
        if kind == 'int':
            kw = dict(calib_algo=calib)
            if gs:
                kw['group_size'] = gs
            return IntegerQuantizer(bit, sym, gran, **kw)
        return qmod.FloatQuantizer(bit, sym, gran, use_qtorch=True)

    for name, dt, R, K, wcfg, acfg, ver, clip_sym, T, nst in cfgs:
        wq = make(*wcfg)
        aq = make(*acfg) if acfg else None
        ac = AutoClipper(w_only=aq is None, wquantizer=wq, aquantizer=aq, clip_version=ver, clip_sym=clip_sym,
                         save_clip=False, padding_mask=None)
        wt = torch.randn(R, K, generator=gen) * 0.02
        wt[torch.rand(R, K, generator=gen) < 0.01] *= 8
        w = wt.to(DT[dt])
        x = (torch.randn(2, T // 2, K, generator=gen) * torch.exp(0.5 * torch.randn(K, generator=gen))).to(DT[dt])
        cands, qxs = [], []
        fw, fx = ac.fake_quantize_weight, ac.fake_quantize_input

        def rec_w(*a, _f=fw, **k):
            q = _f(*a, **k)
            cands.append(q.clone())
            return q

        def rec_x(*a, _f=fx, **k):
            q = _f(*a, **k)
            qxs.append(q.clone())
            return q
        ac.fake_quantize_weight, ac.fake_quantize_input = rec_w, rec_x
        mx, mn = ac.auto_clip_layer(0, 'fc', w, [x.clone()], n_sample_token=nst)
        oc = 256 if R % 256 == 0 else 64
        nb, nsh = R // oc, 10
        assert len(cands) == nb * nsh
        g = wq.group_size if wcfg[3] == 'per_group' else K
        # [oc batch][level] -> [level, R, K]
        Q = torch.stack([torch.cat([cands[b * nsh + s_].reshape(oc, K) for b in range(nb)], dim=0) for s_ in range(nsh)])
        for q in qxs[1:]:
            assert torch.equal(q, qxs[0])
        p = name + '/'
        out[p + 'w'], out[p + 'x'] = f32(w), f32(x)
        out[p + 'cands_bits'] = Q.contiguous().view(torch.int16).numpy().view(np.uint16)     # 16-bit patterns of dtype dt
        out[p + 'qx_bits'] = qxs[0].reshape(-1, K).contiguous().view(torch.int16).numpy().view(np.uint16)   # the sampled tokens, quantized: [n_tok, K]
        out[p + 'best_max'], out[p + 'best_min'] = f32(mx), f32(mn)
        if ver == 'v1':
            layer = torch.nn.Linear(K, R, bias=False).to(DT[dt])
            layer.weight.data = w.clone()
            ac.apply_clip(0, layer, mn, mx, 'fc')
            out[p + 'clipped'] = f32(layer.weight.data)
        out[p + 'wcfg'] = np.array([str(v) for v in wcfg])
        out[p + 'acfg'] = np.array([str(v) for v in acfg]) if acfg else np.array([], dtype=str)
        out[p + 'meta'] = np.array([R, K, g, int(clip_sym), nst], dtype=np.int64)
        out[p + 'ver'], out[p + 'dt'] = np.array(ver), np.array(dt)
    out['names'] = np.array([c[0] for c in cfgs])
    save('clip_wide', **out)


def suite_clip_v2():
    """What AutoClipper clip_version v2 leaves behind (auto_clip.py:213-256) with a `calib_algo: learnable` weight
    quantizer (quant.py:205-224): apply_clip stores a range as logit factors, the quantizer scales its min / max range by
    their sigmoid. The v2 SEARCH itself only runs per_channel in the reference (per_group raises inside quant.py:701), so the
    ranges here come from the v1 search on the same data; goldens: the factors and the fake-quantized weights."""
    from llmc.compression.quantization.auto_clip import AutoClipper
    out = {}
    gen = torch.Generator().manual_seed(321)
    cfgs = [('bf16_sym_pc_clipsym', 'bf16', True, 'per_channel', None, True), ('f16_asym_pc_noclipsym', 'f16', False, 'per_channel', None, False),
            ('bf16_asym_g128', 'bf16', False, 'per_group', 128, False)]
    for name, dt, sym, gran, gs, clip_sym in cfgs:
        R, K = 64, 256
        kw = dict(group_size=gs) if gs else {}
        wq = IntegerQuantizer(4, sym, gran, calib_algo='learnable', **kw)
        ac = AutoClipper(w_only=True, wquantizer=wq, aquantizer=None, clip_version='v2', clip_sym=clip_sym,
                         save_clip=False, padding_mask=None)
        wt = torch.randn(R, K, generator=gen) * 0.02
        wt[torch.rand(R, K, generator=gen) < 0.01] *= 8
        w = wt.to(DT[dt])
        g = gs or K
        wg = w.reshape(R, K // g, g)
        org_max = (wg.abs() if clip_sym else wg).amax(dim=-1, keepdim=True)
        org_min = wg.amin(dim=-1, keepdim=True)
        lev = torch.randint(0, 10, (R, K // g, 1), generator=gen)
        mx = org_max * (1 - lev / 20)
        mn = -mx if clip_sym else org_min * (1 - lev / 20)
        mx, mn = mx.to(DT[dt]), mn.to(DT[dt])
        layer = torch.nn.Linear(K, R, bias=False).to(DT[dt])
        layer.weight.data = w.clone()
        ac.apply_clip(0, layer, mn, mx, 'fc')
        args = {'upbound_factor': layer.buf_upbound_factor, 'lowbound_factor': layer.buf_lowbound_factor}
        fq = wq.fake_quant_weight_dynamic(layer.weight, args)
        fq0 = wq.fake_quant_weight_dynamic(layer.weight, {'upbound_factor': None, 'lowbound_factor': None})
        p = name + '/'
        out[p + 'w'] = f32(w)
        out[p + 'max'], out[p + 'min'] = f32(mx), f32(mn)
        out[p + 'up_factor'] = f32(layer.buf_upbound_factor)
        out[p + 'low_factor'] = f32(layer.buf_lowbound_factor) if layer.buf_lowbound_factor is not None else np.zeros(0, np.float32)
        out[p + 'w_qdq'], out[p + 'w_qdq_nofactor'] = f32(fq), f32(fq0)
        out[p + 'meta'] = np.array([int(sym), gs or 0, int(clip_sym)], dtype=np.int64)
        out[p + 'dt'] = np.array(dt)
    out['names'] = np.array([c[0] for c in cfgs])
    save('clip_v2', **out)


def suite_fp8():
    """FloatQuantizer e4m3 weight path. qtorch is not installed/vendored: float_quantize is bound to torch's own
    float8_e4m3fn round trip (the cast the reference's real-quant path ends in), which is what the oracle pins."""
    import llmc.compression.quantization.quant as qmod

    def fq(x, e, m, rounding='nearest'):
        assert (e, m) == (4, 3)
        return x.to(torch.float8_e4m3fn).float()
    qmod.float_quantize = fq
    out = {}
    gen = torch.Generator().manual_seed(7)
    ci = 0
    for dt in ('bf16', 'f16'):
        for gran in ('per_tensor', 'per_channel'):
            q = qmod.FloatQuantizer('e4m3', True, gran, use_qtorch=True)
            w = (torch.randn(24, 160, generator=gen) * 0.05).to(DT[dt])
            w[3, 5] = 0.0
            w[4, 6] = -0.0
            rw, rs, _ = q.real_quant_weight_dynamic(w)
            fk = q.fake_quant_weight_dynamic(w)
            p = f'c{ci}_'
            out[p + 'w'] = f32(w)
            out[p + 'bits'] = rw.view(torch.uint8).numpy()
            out[p + 'scales'] = f32(rs).reshape(-1)
            out[p + 'fake'] = f32(fk)
            out[p + 'dt'] = np.array(dt)
            out[p + 'gran'] = np.array(gran)
            ci += 1
    out['n'] = np.array(ci)
    save('fp8', **out)


def _qtorch_stub(x, e, m, rounding='nearest'):
    """float_quantize bound to the RESTATEMENT of qtorch's published algorithm (oracle/quant_ref.py:qtorch_float_quantize):
    qtorch itself is a third-party dependency that is neither vendored by the reference nor installed here."""
    if ROOT not in sys.path:
        sys.path.insert(0, ROOT)
    from oracle import quant_ref as QR
    assert rounding == 'nearest'
    return torch.from_numpy(QR.qtorch_float_quantize(x.detach().float().numpy(), e, m)).reshape(x.shape)


def suite_fp8_qtorch():
    """FloatQuantizer e4m3 AND e5m2 (quant.py:983-984, 1162-1163) with float_quantize = the restated qtorch: the reference's
    own class code around it (scales from finfo.max, the division in the tensor dtype, `.to(float8 type)` of the quantized
    values, the fp32 dequantisation product) runs unchanged. Weights are drawn so that a good part of |x / scale| lies above
    240, where qtorch's e4m3 saturates and the OCP cast of the older goldens (fp8.npz) does not."""
    import llmc.compression.quantization.quant as qmod
    qmod.float_quantize = _qtorch_stub
    out = {}
    gen = torch.Generator().manual_seed(70)
    ci = 0
    for bit in ('e4m3', 'e5m2'):
        for dt in ('bf16', 'f16'):
            for gran in ('per_tensor', 'per_channel'):
                q = qmod.FloatQuantizer(bit, True, gran, use_qtorch=True)
                w = (torch.randn(24, 160, generator=gen) * 0.05).to(DT[dt])
                w[3, 5] = 0.0
                w[4, 6] = -0.0
                w[5, :8] = w.abs().max() * torch.tensor([1.0, -0.99, 0.75, -0.6, 0.56, -0.5536, 0.53, 0.25]).to(DT[dt])
                rw, rs, _ = q.real_quant_weight_dynamic(w)
                fk = q.fake_quant_weight_dynamic(w)
                p = f'c{ci}_'
                out[p + 'w'] = f32(w)
                out[p + 'bits'] = rw.view(torch.uint8).numpy()
                out[p + 'scales'] = f32(rs).reshape(-1)
                out[p + 'fake'] = f32(fk)
                out[p + 'dt'] = np.array(dt)
                out[p + 'gran'] = np.array(gran)
                out[p + 'bit'] = np.array(bit)
                ci += 1
    out['n'] = np.array(ci)
    save('fp8_qtorch', **out)


def suite_fp8_group_qtorch():
    """FloatQuantizer per_group (rtn_w_a_block.yml quantizes activations as FP8 e4m3 in groups of 128; quant.py:612-618 reshapes to
    [-1, g], the scale of a group stays in the tensor dtype): fake_quant_act_dynamic and fake_quant_weight_dynamic, float_quantize =
    the restated qtorch."""
    import llmc.compression.quantization.quant as qmod
    qmod.float_quantize = _qtorch_stub
    out = {}
    gen = torch.Generator().manual_seed(1207)
    ci = 0
    for bit in ('e4m3', 'e5m2'):
        for dt in ('bf16', 'f16'):
            for gs in (128, 32):
                q = qmod.FloatQuantizer(bit, True, 'per_group', group_size=gs, use_qtorch=True)
                x = (torch.randn(3, 10, 256, generator=gen) * torch.exp(0.7 * torch.randn(256, generator=gen))).to(DT[dt])
                x[0, 0, :gs] = 0.0
                fa = q.fake_quant_act_dynamic(x)
                w = (torch.randn(24, 256, generator=gen) * 0.05).to(DT[dt])
                fw = q.fake_quant_weight_dynamic(w)
                p = f'c{ci}_'
                out[p + 'x'], out[p + 'fake_x'] = f32(x), f32(fa)
                out[p + 'w'], out[p + 'fake_w'] = f32(w), f32(fw)
                out[p + 'dt'], out[p + 'bit'], out[p + 'gs'] = np.array(dt), np.array(bit), np.array(gs)
                ci += 1
    out['n'] = np.array(ci)
    save('fp8_group_qtorch', **out)


def suite_fp8_block_qtorch():
    """suite_fp8_block with float_quantize = the restated qtorch (see suite_fp8_qtorch): FloatQuantizer e4m3 per_block and
    the reference's non-Triton weight_cast_to_fp8 / weight_cast_to_bf16 (quant.py:18-43), which go through the same class."""
    _fp8_block(_qtorch_stub, 'fp8_block_qtorch', 78)


def suite_fp8_block():
    """FloatQuantizer e4m3 `per_block` (128 x 128 tiles, DeepSeek-V3 layout) and the reference's non-Triton
    weight_cast_to_fp8 / weight_cast_to_bf16 (quant.py:18-43). float_quantize is bound to torch's e4m3fn cast (the
    arithmetic of the reference's Triton kernels, kernel.py; `fp8_semantics='cast'` here). The Triton kernels cannot run in
    this container: their goldens come from the MI355X (tools/fp8_triton_golden.py)."""
    def fq(x, e, m, rounding='nearest'):
        assert (e, m) == (4, 3)
        return x.to(torch.float8_e4m3fn).float()
    _fp8_block(fq, 'fp8_block', 77)


def _fp8_block(fq, fname, seed):
    import llmc.compression.quantization.quant as qmod
    qmod.float_quantize = fq
    out = {}
    gen = torch.Generator().manual_seed(seed)
    ci = 0
    # N is kept a multiple of the block: the reference's restore_tensor (quant.py:647-651) scrambles a padded N
    for dt, (M, N), bsz in (('bf16', (200, 384), 128), ('f16', (128, 256), 128), ('bf16', (100, 192), 64)):
        q = qmod.FloatQuantizer('e4m3', True, 'per_block', block_size=bsz, use_qtorch=True)
        w = (torch.randn(M, N, generator=gen) * 0.05).to(DT[dt])
        w[:, 7] *= 25
        w[3, 5] = 0.0
        if bsz == 64:
            w[:64, :64] = 0.0                       # an all-zero block: scale clamps to 1e-5 / 448
        rw, rs, _ = q.real_quant_weight_dynamic(w.clone())
        fk = q.fake_quant_weight_dynamic(w.clone())
        w8, s8 = qmod.weight_cast_to_fp8(w.clone(), bsz)
        back = qmod.weight_cast_to_bf16(w8, s8, bsz)
        p = f'c{ci}_'
        out[p + 'w'] = f32(w)
        out[p + 'bits'] = rw.view(torch.uint8).numpy()
        out[p + 'scales'] = f32(rs)
        out[p + 'fake'] = f32(fk)
        out[p + 'cast_bits'] = w8.view(torch.uint8).numpy()
        out[p + 'cast_scales'] = f32(s8)
        out[p + 'cast_back'] = f32(back)
        out[p + 'dt'] = np.array(dt)
        out[p + 'block'] = np.array(bsz)
        ci += 1
    out['n'] = np.array(ci)
    save(fname, **out)


def suite_e2e():
    """The reference's OWN algorithm classes (RTN / GPTQ / Awq, constructed and driven exactly as llmc/__main__.py
    does: ctor -> run_block_loop() -> deploy()) on the toy model adapter of tests/toy_model.py, on CPU.
    tests/test_e2e_gpu.py runs llmc_amd's classes on the same adapter / config / seeds and compares."""
    import copy
    import torch.distributed as dist
    from easydict import EasyDict
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    from toy_model import ToyModel, calib_input
    from llmc.compression.quantization import GPTQ, RTN, Awq
    if not dist.is_initialized():
        dist.init_process_group('gloo', init_method='tcp://127.0.0.1:29593', rank=0, world_size=1)
    torch.nn.Module.cuda = lambda self, *a, **k: self
    torch.Tensor.cpu = lambda self, *a, **k: self.clone()      # GPU semantics of .cpu(): a copy (see suite_awq)
    out = {}

    def linears(model):
        return {f'{i}.{n}': m for i, b in enumerate(model.get_blocks()) for n, m in b.named_modules()
                if hasattr(m, 'weight') and m.weight is not None and m.weight.dim() == 2}

    config = EasyDict(calib={'seq_len': 64}, model={'type': 'Toy'}, eval={})
    # RTN W4 sym g128 -> fake quant
    def ToyModel_():
        return ToyModel(hidden=128, inner=256, seed=3)
    model = ToyModel_()
    qc = EasyDict(weight={'bit': 4, 'symmetric': True, 'granularity': 'per_group', 'group_size': 128}, modality='language')
    algo = RTN(model, qc, calib_input(model), None, config)
    algo.run_block_loop()
    algo.deploy('fake_quant')
    for n, m in linears(model).items():
        out['rtn/' + n] = f32(m.weight.data)
    # GPTQ, two configurations
    for tag, sym, static in (('gptq_dyn', False, False), ('gptq_static', True, True)):
        model = ToyModel_()
        qc = EasyDict(weight={'bit': 4, 'symmetric': sym, 'granularity': 'per_group', 'group_size': 128},
                      special={'actorder': True, 'static_groups': static, 'percdamp': 0.01, 'blocksize': 128,
                               'true_sequential': True}, quant_out=True, modality='language')
        algo = GPTQ(model, qc, calib_input(model), None, config)
        algo.dev = torch.device('cpu')          # the CI rewrite's `self.dev = 'cpu'` (ci_check/change_files.py)
        algo.run_block_loop()
        for n, m in linears(model).items():
            out[f'{tag}/w/{n}'] = f32(m.weight.data)
            out[f'{tag}/scales/{n}'] = f32(m.buf_scales).reshape(-1)
        algo.deploy('fake_quant')
        out[f'{tag}/fake/0.down_proj'] = f32(model.get_blocks()[0].down_proj.weight.data)
    # Awq: trans v2 + clip v1, one calibration batch
    model = ToyModel_()
    inp = calib_input(model)
    inp1 = {'data': [torch.cat(inp['data'], dim=0)], 'kwargs': [{}]}
    qc = EasyDict(weight={'bit': 4, 'symmetric': True, 'granularity': 'per_group', 'group_size': 128},
                  special={'trans': True, 'trans_version': 'v2', 'weight_clip': True, 'clip_sym': True}, modality='language')
    algo = Awq(model, qc, inp1, None, config)
    algo.run_block_loop()
    for i, b in enumerate(model.get_blocks()):
        out[f'awq/ln/{i}'] = f32(b.ln.weight.data)
    for n, m in linears(model).items():
        out['awq/w/' + n] = f32(m.weight.data)
    save('e2e', **out)



def suite_e2e_spqr():
    """The reference's SpQR class (ctor -> run_block_loop() -> deploy('fake_quant')) on the toy adapter, CPU."""
    import torch.distributed as dist
    from easydict import EasyDict
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    from toy_model import ToyModel, calib_input
    from llmc.compression.quantization.spqr import SpQR
    if not dist.is_initialized():
        dist.init_process_group('gloo', init_method='tcp://127.0.0.1:29594', rank=0, world_size=1)
    torch.nn.Module.cuda = lambda self, *a, **k: self
    torch.Tensor.cpu = lambda self, *a, **k: self.clone()
    out = {}

    def linears(model):
        return {f'{i}.{n}': m for i, b in enumerate(model.get_blocks()) for n, m in b.named_modules()
                if hasattr(m, 'weight') and m.weight is not None and m.weight.dim() == 2}

    config = EasyDict(calib={'seq_len': 64}, model={'type': 'Toy'}, eval={})
    model = ToyModel(hidden=128, inner=256, seed=3)
    q2 = {'bit': 3, 'symmetric': False, 'granularity': 'per_group', 'group_size': 16, 'round_zp': False}
    qc = EasyDict(weight={'bit': 4, 'symmetric': False, 'granularity': 'per_group', 'group_size': 16, 'round_zp': False},
                  special={'actorder': True, 'percdamp': 1, 'blocksize': 128, 'true_sequential': True,
                           'relative_threshold': 0.2, 'simplified_outliers': False, 'scale': dict(q2), 'zero': dict(q2)},
                  quant_out=True, modality='language')
    algo = SpQR(model, qc, calib_input(model), None, config)
    algo.dev = torch.device('cpu')
    algo.run_block_loop()
    for n, m in linears(model).items():
        out[f'w/{n}'] = f32(m.weight.data)
        out[f'scales/{n}'] = f32(m.buf_scales).reshape(-1)
        out[f'zeros/{n}'] = f32(m.buf_zeros).reshape(-1)
        out[f'nout/{n}'] = np.int64(m.buf_mask.to_dense().sum().item())
    algo.deploy('fake_quant')
    for n, m in linears(model).items():
        if n in ('0.gate_proj', '1.down_proj'):
            out[f'fake/{n}'] = f32(m.weight.data)
    save('e2e_spqr', **out)


def suite_hist():
    """IntegerQuantizer(calib_algo='static_hist').get_static_hist_range + get_qparams (quant.py:462-512, 545-559)."""
    out = {}
    gen = torch.Generator().manual_seed(11)
    cases = {
        'growing': [torch.randn(1, 64, 256, generator=gen) * (1 + i) for i in range(4)],          # range grows: re-binning
        'outlier_tail': [torch.cat([torch.randn(1, 128, 128, generator=gen), torch.tensor([[[40.0] * 128]])], 1) for _ in range(3)],
        'same_range': [torch.randn(1, 32, 64, generator=gen).clamp(-2, 2).index_put((torch.tensor([0]), torch.tensor([0]), torch.tensor([0, 1])), torch.tensor([-2.0, 2.0])) for _ in range(3)],
        'skewed': [torch.rand(1, 200, 100, generator=gen) ** 4 * 9 - 0.5 for _ in range(2)],
    }
    for name, acts in cases.items():
        for dt in ('bf16',):
            q = IntegerQuantizer(8, True, 'per_tensor', calib_algo='static_hist')
            xs = [a.to(DT[dt]) for a in acts]
            rec = {}
            org = q.get_hist_threshold

            def thr(h, mn, mx):
                rec['hist'], rec['min'], rec['max'] = h.clone(), mn.clone(), mx.clone()
                return org(h, mn, mx)
            q.get_hist_threshold = thr
            mins, maxs = q.get_static_hist_range(list(xs))
            sl, zl, _, _ = IntegerQuantizer(8, True, 'per_tensor', calib_algo='static_hist').get_batch_tensors_qparams(list(xs))
            p = f'{name}/'
            out[p + 'x'] = np.stack([f32(x[0]) for x in xs])
            out[p + 'hist'] = f32(rec['hist'])
            out[p + 'min'] = f32(rec['min'])
            out[p + 'max'] = f32(rec['max'])
            out[p + 'new_min'] = f32(mins[0])
            out[p + 'new_max'] = f32(maxs[0])
            out[p + 'scale'] = f32(sl[0])
            print(name, float(rec['min']), float(rec['max']), '->', float(mins[0]), float(maxs[0]))
    save('hist', **out)


def suite_spqr():
    """SpQR.add_batch / layer_transform (Hessian prep, weight_transform with leave-one-out outlier detection and the
    second-level scale / zero quantizers) / set_model_qparams / w_qdq (spqr.py:116-380) on small seeded layers."""
    import math
    from llmc.compression.quantization.spqr import SpQR
    out = {}
    cfgs = [
        # (name, bit, gs, actorder, percdamp, rel_threshold, simplified, dtype, R, K, dead)
        ('g16_act_thr02', 4, 16, True, 1.0, 0.2, False, 'f16', 32, 256, True),
        ('g32_noact_thr01', 3, 32, False, 0.01, 0.1, False, 'bf16', 24, 256, False),
        ('g16_act_inf', 4, 16, True, 1.0, 'inf', False, 'f16', 16, 128, False),
        ('g64_act_simplified', 4, 64, True, 1.0, 0.2, True, 'f16', 16, 256, False),
    ]
    gen = torch.Generator().manual_seed(77)
    for (name, bit, gs, actorder, percdamp, thr, simplified, dt, R, K, dead) in cfgs:
        sp = SpQR.__new__(SpQR)
        sp.dev = torch.device('cpu')
        sp.model_dtype = DT[dt]
        sp.wquantizer = IntegerQuantizer(bit, False, 'per_group', group_size=gs, round_zp=False)
        sp.actorder, sp.percdamp, sp.blocksize = actorder, percdamp, 128
        sp.relative_threshold = math.inf if thr == 'inf' else thr
        sp.simplified_outliers = simplified
        if actorder:
            sp.need_perm = True
        qc = dict(bit=3, symmetric=False, granularity='per_group', group_size=16, round_zp=False)
        sp.scale_quantizer = IntegerQuantizer(**qc)
        sp.zero_quantizer = IntegerQuantizer(**qc)
        sp.Q = IntegerQuantizer(bit, False, 'per_channel', round_zp=False)
        sp.layers_cache = {}
        layer = torch.nn.Linear(K, R, bias=False).to(DT[dt])
        layer.weight.data = rand_weight(gen, R, K, dt)
        lname = 'fc'
        sp.layers_cache[lname] = {}
        sp.layer_init(layer, lname)
        xs = []
        for b in range(2):
            x = torch.randn(1, 96, K, generator=gen) * torch.exp(0.5 * torch.randn(K, generator=gen))
            x[..., 5] *= 30
            if dead:
                x[..., 17] = 0
                x[..., 200] = 0
            x = x.to(DT[dt])
            xs.append(x)
            sp.add_batch(layer, lname, x, None)
        H = sp.layers_cache[lname]['H'].clone()
        W0 = layer.weight.data.clone()
        rec = {}
        org_wt = sp.weight_transform

        def wt(W, Hinv, Losses, tmp, mask):
            rec['Wp'], rec['U'] = W.clone(), Hinv.clone()
            org_wt(W, Hinv, Losses, tmp, mask)
            rec['tmp'], rec['losses'], rec['mask'] = tmp.clone(), Losses.clone(), mask.clone()
            rec['threshold'] = sp.relative_threshold * (rec['Wp'].var(dim=0) / torch.diag(Hinv).square()).mean().item()
        sp.weight_transform = wt
        sp.layer_transform(layer, lname)
        wqdq = sp.w_qdq(layer, sp.wquantizer)
        p = name + '/'
        if name == 'g16_act_thr02':
            out[p + 'x'] = np.stack([f32(x[0]) for x in xs])
        out[p + 'H'] = f32(H)
        out[p + 'W0'] = f32(W0)
        out[p + 'perm'] = layer.buf_perm.numpy().astype(np.int64) if actorder else np.zeros(0, np.int64)
        out[p + 'Wp'] = f32(rec['Wp'])
        out[p + 'U'] = f32(rec['U'])
        out[p + 'tmp'] = f32(rec['tmp'])
        out[p + 'losses'] = f32(rec['losses'])
        out[p + 'mask'] = rec['mask'].numpy().astype(np.uint8)
        out[p + 'threshold'] = np.float64(rec['threshold'])
        out[p + 'weight'] = f32(layer.weight.data)
        out[p + 'buf_scales'] = f32(layer.buf_scales)
        out[p + 'buf_zeros'] = f32(layer.buf_zeros)
        out[p + 'buf_mask'] = layer.buf_mask.to_dense().numpy().astype(np.uint8)
        out[p + 'w_qdq'] = f32(wqdq)
        out[p + 'cfg'] = np.array([bit, gs, int(actorder), R, K, int(simplified)], np.int64)
        out[p + 'percdamp'] = np.float64(percdamp)
        out[p + 'rel_threshold'] = np.float64(math.inf if thr == 'inf' else thr)
        print(name, 'outliers', int(rec['mask'].sum()), '/', rec['mask'].numel(), 'thr', rec['threshold'])
    save('spqr', **out)


def suite_mse():
    """calib_algo='mse' weight ranges (quant.py:145-203): the searched (min, max) per row/group and the qparams and
    fake-quantized weights that follow from them."""
    gen = torch.Generator().manual_seed(4321)
    out = {}
    cases = []
    for dt in ('f16', 'bf16', 'f32'):
        for (bit, sym, gran, gs) in [(4, False, 'per_group', 128), (4, True, 'per_group', 128),
                                     (8, True, 'per_channel', None), (3, False, 'per_group', 64)]:
            cases.append((dt, bit, sym, gran, gs))
    for ci, (dt, bit, sym, gran, gs) in enumerate(cases):
        kw = dict(group_size=gs) if gs else {}
        q = IntegerQuantizer(bit, sym, gran, calib_algo='mse', **kw)
        w = rand_weight(gen, 24, 512, dt)
        t = q.reshape_tensor(w)
        mn0, mx0 = q.get_minmax_range(t.float())
        mn, mx = q.get_tensor_range(t.clone())
        _, s, z, qmax, qmin = q.get_tensor_qparams(w)
        fq = q.fake_quant_weight_dynamic(w)
        p = f'c{ci}_'
        out[p + 'w'] = f32(w)
        out[p + 'min0'] = f32(mn0).reshape(-1)
        out[p + 'max0'] = f32(mx0).reshape(-1)
        out[p + 'min'] = f32(mn).reshape(-1)
        out[p + 'max'] = f32(mx).reshape(-1)
        out[p + 'scales'] = f32(s).reshape(-1)
        out[p + 'scales_dtype'] = np.array(str(s.dtype))
        out[p + 'zeros'] = f32(z).reshape(-1) if z.dim() > 0 else np.zeros(0, np.float32)
        out[p + 'fake'] = f32(fq)
        out[p + 'fake_dtype'] = np.array(str(fq.dtype))
        out[p + 'meta'] = np.array([bit, int(sym), gs or 0, float(qmin), float(qmax)], dtype=np.float64)
    out['cases'] = np.array(['|'.join(map(str, c)) for c in cases])
    save('mse', **out)

def suite_awq_fp8ckpt():
    """The FP8-checkpoint branches of the AWQ path (DeepSeek-V3 layout: `weight` float8_e4m3fn + `weight_scale_inv` per
    128 x 128 block): Awq.get_weight_scale (awq.py:48-72), Awq.fake_quantize_weight (awq.py:147-164),
    scale_ln_fcs / scale_fc_fc (base_blockwise_quantization.py:655-700, 750-775) and w_qdq (base_…:46-68), run from the
    reference's own class code on CPU. There the reference binds its NON-Triton weight_cast_to_bf16 / weight_cast_to_fp8
    (quant.py:18-43, FloatQuantizer e4m3 per_block with use_qtorch) — float_quantize = the restated qtorch, as in
    suite_fp8_block_qtorch. (The search loop itself is not recorded: the reference's non-Triton LlmcFp8Linear.forward
    replaces the layer's weight by a bf16 tensor, module_utils.py:160-164, so its first grid point turns the module
    into a bf16 one.)"""
    import llmc.compression.quantization.quant as qmod
    qmod.float_quantize = _qtorch_stub
    import llmc.compression.quantization.awq as awq_mod
    import llmc.compression.quantization.base_blockwise_quantization as bb_mod
    from llmc.compression.quantization.awq import Awq
    from llmc.compression.quantization.module_utils import LlmcFp8Linear
    for m in (awq_mod, bb_mod):            # the CPU binding (awq.py:17-20): the quantizer spelling of the two casts
        m.weight_cast_to_bf16, m.weight_cast_to_fp8 = qmod.weight_cast_to_bf16, qmod.weight_cast_to_fp8
    out = {}
    gen = torch.Generator().manual_seed(20250926)
    cfgs = [('w4g128_asym', 4, False, 128, 128), ('w4g64_sym', 4, True, 64, 128), ('w8ch_sym', 8, True, 0, 128),
            ('w4g32_asym_b64', 4, False, 32, 64)]
    for name, bit, sym, gs, bsz in cfgs:
        K, Rs = 256, (192, 128)
        wq = IntegerQuantizer(bit, sym, 'per_group', group_size=gs) if gs else IntegerQuantizer(bit, sym, 'per_channel')
        a = Awq.__new__(Awq)
        a.wquantizer = wq
        a.fp8_block_size = bsz
        a.has_gqa = False
        a.do_gqa_trans = False
        layers, p = [], name + '/'
        for li, R in enumerate(Rs):
            l = LlmcFp8Linear(K, R, None, bsz)
            wt = torch.randn(R, K, generator=gen) * 0.02
            wt[:, torch.randperm(K, generator=gen)[:4]] *= 20
            w8, s8 = qmod.weight_cast_to_fp8(wt.to(torch.bfloat16), bsz)
            l.weight.data, l.weight_scale_inv.data = w8, s8
            layers.append(l)
            out[p + f'w8_{li}'] = w8.view(torch.uint8).numpy().copy()
            out[p + f's8_{li}'] = f32(s8)
        layers_dict = {f'l{i}': l for i, l in enumerate(layers)}
        w_max = a.get_weight_scale(layers_dict)
        out[p + 'w_max'] = f32(w_max)
        scales = (torch.exp(0.4 * torch.randn(K, generator=gen))).to(torch.bfloat16)
        out[p + 'scales'] = f32(scales)
        for li, l in enumerate(layers):
            w_keep, s_keep = l.weight.data.clone(), l.weight_scale_inv.data.clone()
            a.fake_quantize_weight(l, scales, False, f'l{li}')
            out[p + f'fq_w8_{li}'] = l.weight.data.view(torch.uint8).numpy().copy()
            out[p + f'fq_s8_{li}'] = f32(l.weight_scale_inv.data)
            l.weight.data, l.weight_scale_inv.data = w_keep.clone(), s_keep.clone()
            # w_qdq (deploy-time fake quant of an FP8 module): returns the fp8 tensor, replaces the block scales
            r = a.w_qdq(l, wq)
            out[p + f'qdq_w8_{li}'] = r.view(torch.uint8).numpy().copy()
            out[p + f'qdq_s8_{li}'] = f32(l.weight_scale_inv.data)
            l.weight.data, l.weight_scale_inv.data = w_keep, s_keep
        # scale folding: LayerNorm -> the two FP8 layers, then fc -> fc (v_proj -> o_proj shape: out == in)
        ln = torch.nn.LayerNorm(K).to(torch.bfloat16)
        ln.weight.data = (1.0 + 0.1 * torch.randn(K, generator=gen)).to(torch.bfloat16)
        ln.bias.data = (0.1 * torch.randn(K, generator=gen)).to(torch.bfloat16)
        out[p + 'ln_w'], out[p + 'ln_b'] = f32(ln.weight.data), f32(ln.bias.data)
        a.scale_ln_fcs(ln, layers, scales)
        out[p + 'ln_w_after'], out[p + 'ln_b_after'] = f32(ln.weight.data), f32(ln.bias.data)
        for li, l in enumerate(layers):
            out[p + f'ln_w8_{li}'] = l.weight.data.view(torch.uint8).numpy().copy()
            out[p + f'ln_s8_{li}'] = f32(l.weight_scale_inv.data)
        fc1, fc2 = LlmcFp8Linear(K, K, None, bsz), LlmcFp8Linear(K, 128, None, bsz)
        for li, l in enumerate((fc1, fc2)):
            wt = torch.randn(l.out_features, K, generator=gen) * 0.03
            l.weight.data, l.weight_scale_inv.data = qmod.weight_cast_to_fp8(wt.to(torch.bfloat16), bsz)
            out[p + f'fc{li + 1}_w8'] = l.weight.data.view(torch.uint8).numpy().copy()
            out[p + f'fc{li + 1}_s8'] = f32(l.weight_scale_inv.data)
        a.scale_fc_fc(fc1, fc2, scales)
        for li, l in enumerate((fc1, fc2)):
            out[p + f'fc{li + 1}_w8_after'] = l.weight.data.view(torch.uint8).numpy().copy()
            out[p + f'fc{li + 1}_s8_after'] = f32(l.weight_scale_inv.data)
        out[p + 'meta'] = np.array([bit, int(sym), gs, bsz, K], dtype=np.int64)
    out['names'] = np.array([c[0] for c in cfgs])
    save('awq_fp8ckpt', **out)


SUITES = {'awq_fp8ckpt': suite_awq_fp8ckpt, 'fp8_group_qtorch': suite_fp8_group_qtorch, 'awq_wa': suite_awq_wa, 'clip_wide': suite_clip_wide, 'clip_more': suite_clip_more, 'awq_more': suite_awq_more, 'gptq_more': suite_gptq_more, 'awq_gqa': suite_awq_gqa, 'clip_v2': suite_clip_v2, 'awq_flat': suite_awq_flat, 'clip_mb': suite_clip_mb, 'mse': suite_mse, 'quant': suite_quant, 'pack': suite_pack, 'gptq': suite_gptq, 'awq': suite_awq, 'clip': suite_clip,
          'fp8': suite_fp8, 'e2e': suite_e2e, 'awq_inspect': suite_awq_inspect, 'quant_pt': suite_quant_pt, 'gptq_owq': suite_gptq_owq, 'fp8_block': suite_fp8_block, 'fp8_qtorch': suite_fp8_qtorch, 'fp8_block_qtorch': suite_fp8_block_qtorch, 'spqr': suite_spqr, 'e2e_spqr': suite_e2e_spqr, 'hist': suite_hist}

if __name__ == '__main__':
    which = sys.argv[1:] or list(SUITES)
    for s in which:
        SUITES[s]()
