"""Time the REFERENCE's GPTQ path (oracle/_ref, built by oracle/build_ref.py) on the host cores; print one JSON line.

What is timed (SURVEY.md §8d "CPU reference timed beside it"), with torch.set_num_threads(os.cpu_count()):
  * GPTQ.add_batch (gptq.py:254-295) on `--batches` sequences of [1, seq, K] (the op is a fixed-shape GEMM per batch:
    scaled linearly to the calibration set),
  * GPTQ.process_hessian_and_weights (:128-176) and GPTQ.weight_transform (:199-244) of one K x K layer in full.
Run as a subprocess by bench.py's cpu_baseline leg; test infrastructure, never imported by the product."""
import argparse
import json
import os
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, '_shims'))
sys.path.insert(0, os.path.join(HERE, '_ref'))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--K', type=int, default=4096)
    ap.add_argument('--seq', type=int, default=2048)
    ap.add_argument('--batches', type=int, default=4)
    ap.add_argument('--threads', type=int, default=0)
    ap.add_argument('--workload', default='gptq')
    ap.add_argument('--tokens', type=int, default=2048)
    a = ap.parse_args()
    if a.workload == 'awq':
        return main_awq(a)
    import torch
    import torch.distributed as dist
    # the Hessian GEMM scales with the cores; the factorisations and the column loop (thousands of small ATen ops) get
    # SLOWER beyond a handful of threads (256 threads: > 3 minutes for what 16 do in seconds) — each stage runs with the
    # thread count that suits it best, capped at the box's cores
    cores = a.threads or os.cpu_count() or 1
    small = min(cores, 16)
    torch.set_num_threads(cores)
    os.environ.setdefault('WORLD_SIZE', '1')
    os.environ.setdefault('RANK', '0')
    # add_batch all-reduces H across ranks (gptq.py:292-295). One rank here: the collective is the identity, and a Gloo
    # rendezvous would depend on the box's hostname resolving — stub the two calls instead of opening a process group.
    dist.all_reduce = lambda *a, **k: None
    dist.get_world_size = lambda *a, **k: 1
    from llmc.compression.quantization.gptq import GPTQ
    from llmc.compression.quantization.quant import IntegerQuantizer

    K = a.K
    wq = IntegerQuantizer(4, False, 'per_group', group_size=128)
    g = GPTQ.__new__(GPTQ)                                         # numeric methods only
    g.dev = torch.device('cpu')
    g.wquantizer, g.actorder, g.static_groups, g.percdamp, g.blocksize = wq, True, False, 0.01, 128
    g.chunk_num, g.owq, g.layers_cache, g.model_dtype, g.act_static = 1, False, {}, torch.bfloat16, False
    g.need_perm = True
    gen = torch.Generator().manual_seed(0)
    layer = torch.nn.Linear(K, K, bias=False).to(torch.bfloat16)
    layer.weight.data = (torch.randn(K, K, generator=gen) * 0.02).to(torch.bfloat16)
    _, s0, z0, qmax, qmin = wq.get_tensor_qparams(layer.weight.data)
    for n, v in (('buf_scales', s0), ('buf_zeros', z0), ('buf_qmax', torch.tensor(qmax)), ('buf_qmin', torch.tensor(qmin))):
        layer.register_buffer(n, v.detach() if torch.is_tensor(v) else v)
    g.layers_cache['fc'] = {}
    g.layer_init(layer, 'fc')
    xs = [(torch.randn(1, a.seq, K, generator=gen) * torch.exp(0.5 * torch.randn(K, generator=gen))).to(torch.bfloat16)
          for _ in range(a.batches)]
    g.add_batch(layer, 'fc', xs[0], None)                          # warm-up (thread pool, allocator)
    # thread sweep for the Hessian GEMM (VERDICT r04 #9a: all 256 threads of the GPU box gave 69 GFLOP/s where 8 threads of
    # the survey box gave 319): one sequence per count, the best count times the sample
    sweep = {}
    if not a.threads:
        for n in sorted({c for c in (8, 16, 32, 64, 128, 256, cores) if c <= cores}):
            torch.set_num_threads(n)
            g.add_batch(layer, 'fc', xs[0], None)
            t0 = time.perf_counter()
            g.add_batch(layer, 'fc', xs[0], None)
            sweep[n] = time.perf_counter() - t0
        cores = min(sweep, key=sweep.get)
    torch.set_num_threads(cores)
    t0 = time.perf_counter()
    for x in xs:
        g.add_batch(layer, 'fc', x, None)
    t_h = (time.perf_counter() - t0) / a.batches
    torch.set_num_threads(small)
    g.layers_cache['fc']['H'] += 0.1 * torch.eye(K)
    g.initialize_qparams_and_prepare_weights(layer, 'fc')
    t0 = time.perf_counter()
    W, Hinv = g.process_hessian_and_weights(layer, 'fc')
    t_c = time.perf_counter() - t0
    Losses, tmp = torch.zeros_like(W), torch.zeros_like(W)
    t0 = time.perf_counter()
    g.weight_transform(W, Hinv, Losses, tmp)
    t_l = time.perf_counter() - t0
    print(json.dumps({'K': K, 'seq': a.seq, 'batches': a.batches, 'threads': cores, 'threads_small_ops': small, 't_hessian_per_seq': t_h,
                      'hessian_thread_sweep_s': {str(k): round(v, 4) for k, v in sweep.items()},
                      'hessian_gflops': 2.0 * a.seq * K * K / t_h / 1e9, 'host_cores': os.cpu_count(),
                      't_factor': t_c, 't_loop': t_l, 'blas': torch.__config__.parallel_info().split('\n')[0:3],
                      'finite': bool(torch.isfinite(tmp).all())}), flush=True)


def main_awq(a):
    """The reference's AWQ grid step (awq.py:229-236) through its own methods, inspect = the Linear itself: per ratio
    get_scales -> scaling_input -> fake_quantize_weight -> inspect_module_forward -> calculate_loss, plus the state-dict
    restore the reference does per step, on `--tokens` tokens of one K x K layer."""
    import copy
    import torch
    cores = a.threads or os.cpu_count() or 1
    torch.set_num_threads(cores)
    from llmc.compression.quantization.awq import Awq
    from llmc.compression.quantization.quant import IntegerQuantizer
    K = a.K
    aw = Awq.__new__(Awq)
    aw.wquantizer = IntegerQuantizer(4, True, 'per_group', group_size=128)
    aw.trans_version = 'v2'
    aw._bs = 1
    gen = torch.Generator().manual_seed(0)
    fc = torch.nn.Linear(K, K, bias=False).to(torch.bfloat16)
    fc.weight.data = (torch.randn(K, K, generator=gen) * 0.02).to(torch.bfloat16)
    x = (torch.randn(1, a.tokens, K, generator=gen) * torch.exp(0.5 * torch.randn(K, generator=gen))).to(torch.bfloat16)
    w_max = aw.get_weight_scale({'fc': fc})
    org_sd = {k: v.clone() for k, v in fc.state_dict().items()}
    org_out = aw.get_original_out(x, fc, {})
    n_grid = 3                                   # of 20: every step does the same work
    t0 = time.perf_counter()
    for n in range(n_grid):
        ratio = n / 20
        scales = aw.get_scales(None, x, w_max, False, ratio)
        xs = x / scales.view(1, -1)              # scaling_input (base_blockwise_quantization.py:877-889)
        aw.fake_quantize_weight(fc, scales, False, 'fc')
        out = aw.inspect_module_forward(xs, fc, {})
        loss = aw.calculate_loss(org_out, out)
        fc.load_state_dict(org_sd)
    t_step = (time.perf_counter() - t0) / n_grid
    print(json.dumps({'K': K, 'tokens': a.tokens, 'threads': cores, 't_grid_step': t_step, 'loss': loss}), flush=True)


if __name__ == '__main__':
    main()
