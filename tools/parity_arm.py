"""One ARM of the end-to-end GPTQ parity envelope (tools/parity_envelope.py): quantize ONE Linear layer from its saved
weights and calibration samples and write what the layer ends up with.

Arms:
  ref_cpu   llmc's own GPTQ class methods (oracle/_ref = /root/reference after its own ci_check CPU rewrite) on the host
            cores, `--threads` MKL threads
  ref_rocm  the UNMODIFIED reference (oracle/_ref_gpu = a plain copy of /root/reference/llmc) on the GPU through
            PyTorch-ROCm (rocBLAS sgemm, hipSOLVER/MAGMA potrf)
  ours      llmc_amd: HessianAccumulator (per-sample hook calls) -> quantize_stacked
Each arm runs the reference's call sequence (layer_init, add_batch per sample, layer_transform; gptq.py:97-196,
254-322) for both bench variants on the same Hessian: w_only (asym g128, actorder, dynamic groups) and vllm (sym g128,
actorder, static groups). Output npz: per variant W' (fp32, original column order), scales, zeros, perm, sum(Losses);
plus the Hessian's diagonal and a 256 x 256 corner. Test / measurement infrastructure; never imported by the product."""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
VARIANTS = {'w_only': dict(sym=False, static_groups=False), 'vllm': dict(sym=True, static_groups=True)}


def run_reference(a, data, dev):
    import torch
    import torch.distributed as dist
    os.environ.setdefault('WORLD_SIZE', '1')
    os.environ.setdefault('RANK', '0')
    dist.all_reduce = lambda *x, **k: None          # one rank: the reference's per-batch all_reduce is the identity
    dist.get_world_size = lambda *x, **k: 1
    from llmc.compression.quantization.gptq import GPTQ
    from llmc.compression.quantization.quant import IntegerQuantizer
    W, X = data['W'], data['X']
    R, K = W.shape
    out = {}
    Hsave = None
    for vname in a.variants:
        v = VARIANTS[vname]
        wq = IntegerQuantizer(4, v['sym'], 'per_group', group_size=128)
        g = GPTQ.__new__(GPTQ)
        g.dev = dev
        g.wquantizer, g.actorder, g.static_groups, g.percdamp, g.blocksize = wq, True, v['static_groups'], 0.01, 128
        g.chunk_num, g.owq, g.layers_cache, g.model_dtype, g.act_static = 1, False, {}, W.dtype, False
        g.need_perm = (not v['static_groups'])
        layer = torch.nn.Linear(K, R, bias=False).to(W.dtype)
        layer.weight.data = W.clone()
        layer = layer.to(dev)
        _, s0, z0, qmax, qmin = wq.get_tensor_qparams(layer.weight.data)   # collect_block_qparams (base_...:338-365)
        for n, t in (('buf_scales', s0), ('buf_zeros', z0), ('buf_qmax', torch.as_tensor(qmax)), ('buf_qmin', torch.as_tensor(qmin))):
            layer.register_buffer(n, (t.detach() if torch.is_tensor(t) else t).to(dev))
        g.layers_cache['fc'] = {}
        g.layer_init(layer, 'fc')
        t0 = time.perf_counter()
        if Hsave is None:
            for i in range(X.shape[0]):
                g.add_batch(layer, 'fc', X[i:i + 1].to(dev), None)
            Hsave = g.layers_cache['fc']['H'].clone()
            out['t_hessian'] = time.perf_counter() - t0
            out['H_diag'] = torch.diagonal(Hsave).float().cpu().numpy()
            out['H_corner'] = Hsave[:256, :256].float().cpu().numpy()
        else:
            g.layers_cache['fc']['H'] = Hsave.clone()
            g.layers_cache['fc']['nsamples'] = X.shape[0]
        rtn_scales = layer.buf_scales.clone()
        losses = {}
        wt = g.weight_transform

        def spy(Wm, Hinv, Losses, tmp, wt=wt):
            r = wt(Wm, Hinv, Losses, tmp)
            losses['sum'] = float(Losses.double().sum().item())
            return r
        g.weight_transform = spy
        t0 = time.perf_counter()
        g.layer_transform(layer, 'fc')
        if dev.type == 'cuda':
            torch.cuda.synchronize()
        out[vname + '/t_transform'] = time.perf_counter() - t0
        out[vname + '/W'] = layer.weight.data.float().cpu().numpy()
        out[vname + '/scales'] = layer.buf_scales.float().reshape(R, -1).cpu().numpy()
        z = layer.buf_zeros
        out[vname + '/zeros'] = z.float().reshape(R, -1).cpu().numpy() if (torch.is_tensor(z) and z.dim() > 0 and z.numel() > 1) \
            else __import__('numpy').zeros((R, K // 128), 'float32')
        out[vname + '/perm'] = layer.buf_perm.cpu().numpy()
        out[vname + '/loss'] = losses.get('sum', float('nan'))
        out[vname + '/rtn_scales'] = rtn_scales.float().reshape(R, -1).cpu().numpy()
    return out


def run_ours(a, data):
    import numpy as np
    import torch
    sys.path.insert(0, ROOT)
    from llmc_amd.compression.quantization import IntegerQuantizer
    from llmc_amd.compression.quantization.gptq_pipeline import GptqConfig, quantize_stacked
    from llmc_amd.compression.quantization.hessian import HessianAccumulator
    dev = torch.device('cuda', 0)
    W, X = data['W'].to(dev), data['X'].to(dev)
    R, K = W.shape
    out = {}
    acc = HessianAccumulator(K, dev, exact_diag=(a.arm != 'ours_fp32diag'))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(X.shape[0]):
        acc.add(X[i:i + 1])
    H = acc.H
    torch.cuda.synchronize()
    out['t_hessian'] = time.perf_counter() - t0
    out['H_diag'] = torch.diagonal(H).cpu().numpy()
    out['H_corner'] = H[:256, :256].cpu().numpy()
    for vname in a.variants:
        v = VARIANTS[vname]
        cfg = GptqConfig(bit=4, symmetric=v['sym'], group_size=128, actorder=True, static_groups=v['static_groups'])
        wq = IntegerQuantizer(4, v['sym'], 'per_group', group_size=128)
        _, s0, z0, _, _ = wq.get_tensor_qparams(W)
        static = [(s0, None if v['sym'] else z0)] if v['static_groups'] else None
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        r = quantize_stacked([W], H.clone(), cfg, static_qparams=static)[0]
        torch.cuda.synchronize()
        out[vname + '/t_transform'] = time.perf_counter() - t0
        assert int(r.info.item()) == 0, 'Hessian not positive definite'
        out[vname + '/W'] = r.weight.float().cpu().numpy()
        sc = r.scales if not v['static_groups'] else s0
        out[vname + '/scales'] = sc.float().reshape(R, -1).cpu().numpy()
        zz = r.zeros if not v['static_groups'] else (None if v['sym'] else z0)
        out[vname + '/zeros'] = zz.float().reshape(R, -1).cpu().numpy() if zz is not None else np.zeros((R, K // 128), 'float32')
        out[vname + '/perm'] = r.perm.cpu().numpy()
        out[vname + '/loss'] = float(r.loss.double().item())
        out[vname + '/rtn_scales'] = s0.float().reshape(R, -1).cpu().numpy()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--arm', required=True, choices=['ref_cpu', 'ref_rocm', 'ours', 'ours_fp32diag'])
    ap.add_argument('--data', required=True)
    ap.add_argument('--out', required=True)
    ap.add_argument('--threads', type=int, default=0)
    ap.add_argument('--variants', nargs='+', default=['w_only', 'vllm'])
    a = ap.parse_args()
    if a.arm == 'ref_cpu':
        sys.path.insert(0, os.path.join(ROOT, 'oracle', '_shims'))
        sys.path.insert(0, os.path.join(ROOT, 'oracle', '_ref'))
    elif a.arm == 'ref_rocm':
        sys.path.insert(0, os.path.join(ROOT, 'oracle', '_shims'))
        sys.path.insert(0, os.path.join(ROOT, 'oracle', '_ref_gpu'))
    import numpy as np
    import torch
    if a.threads:
        torch.set_num_threads(a.threads)
    data = torch.load(a.data)
    t0 = time.perf_counter()
    if a.arm in ('ours', 'ours_fp32diag'):
        out = run_ours(a, data)
    else:
        out = run_reference(a, data, torch.device('cpu') if a.arm == 'ref_cpu' else torch.device('cuda', 0))
    out['t_total'] = time.perf_counter() - t0
    out['threads'] = torch.get_num_threads()
    np.savez(a.out, **out)
    print(a.arm, a.threads, 'done in %.1f s' % out['t_total'], flush=True)


if __name__ == '__main__':
    main()
