#!/usr/bin/env python
"""bench.py — layers/sec of the GPTQ W4A16 hot path on Llama-3-8B Linear shapes (BASELINE.json metric).

One "step" = GPTQ of one transformer block's seven Linear layers (q,k,v,o,gate,up,down; four distinct inputs) from 128 x 2048
resident calibration tokens: Hessian (MFMA, exact fp64 diagonal) -> actorder/dead/damp -> Cholesky + inverse -> blocked column
loop -> scales/zeros/compensated weights.  Synthetic inputs.

  python bench.py --gpus N --steps K --warmup W
N > 1 without a launcher: bench.py re-executes itself under `python -m torch.distributed.run` with N ranks (one per GPU, RCCL);
under a launcher (WORLD_SIZE set) it is a rank.
  no --mode, N > 1: ONE hand-off step is tried first (every rank, agreement over a Gloo control group). If it works the timed
          region runs `handoff` — north_star's partition: every rank owns its blocks, a block's calibration activations arrive
          from the previous owner over RCCL send/recv (xGMI) — and `independent_value` / `cooperative_value` follow outside it;
          if it fails the timed region runs `independent` and the line says why (`handoff_error`).
  --mode handoff | independent | cooperative: that mode only (cooperative: all ranks share ONE block per step, strong scaling:
          Hessian / activation broadcast, sample-sharded all_reduce for the wide subset, llmc_amd/dist/layer_shard.py).
  --dry   GPU-less plumbing check (gloo, CPU stand-ins instead of kernels, tiny shapes): tests/test_bench_spawn.py.
  --workload awq | fp8: BASELINE configs[2] / configs[4] (tools/bench_extras.py), also reported under "extra" of the default line.

Prints ONE JSON line on rank 0 (contract in the task statement): metric/value + roofline + cpu_baseline.
"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_MFMA_16BIT = 2.5e15  # dense bf16/f16 MFMA, /opt/skills/guides/MI355X_MICROARCH.md
PARITY_FILE = 'profiles/r06_parity_envelope_full_down.json'
TRAFFIC_FILES = ('r06_pmc_traffic.json', 'r04_pmc_traffic.json')

MODELS = {
    # name: (hidden, kv_out, ffn, n_blocks)
    'llama3-8b': (4096, 1024, 14336, 32),
    'llama3-70b': (8192, 1024, 28672, 80),
    'tiny': (512, 128, 1024, 2),
}


def block_groups(model):
    h, kv, ffn, _ = MODELS[model]
    # (input name, K, [(layer name, R)]) — llmc/models/llama.py:52-91 subsets
    return [('attn_in', h, [('q_proj', h), ('k_proj', kv), ('v_proj', kv)]),
            ('o_in', h, [('o_proj', h)]),
            ('mlp_in', h, [('gate_proj', ffn), ('up_proj', ffn)]),
            ('down_in', ffn, [('down_proj', h)])]


def parity_envelope_summary(args):
    """End-to-end agreement with the reference on the configuration the metric is quoted on, from the committed measurement
    (tools/parity_envelope.py --full-down / --down-70b on an MI355X: the reference's own GPTQ class on the host cores [8B only] /
    on ROCm / llmc_amd on identical weights and the full 128 x 2048 calibration set, the model's down_proj). `parity_live` of
    the same line is what THIS run measured itself."""
    src = {'llama3-8b': PARITY_FILE, 'llama3-70b': PARITY_FILE.replace('full_down', 'down_70b')}.get(args.model)
    try:
        j = json.load(open(os.path.join(ROOT, src)))
        pr = j['shapes'][0]['pairs']['w_only' if args.variant == 'w_only' else 'vllm']
        pick = lambda m: {k: m[k] for k in ('codes_equal', 'scales_within_1e-4', 'scales_within_1e-2', 'zeros_equal', 'perm_equal',
                                            'perm_diff_within_4x_noise', 'H_diag_rel_max') if k in m}
        ours = 'ours' if args.exact_diag else 'ours_fp32diag'
        out = {'layer': j['shapes'][0]['title'], 'source': src, 'precomputed': 'read from the committed file (round 6)',
               'arm': ours + (' (default: diag(H) folded into fp64 inside the Hessian kernel)' if args.exact_diag else ' (--exact-diag 0)')}
        for ref in ('ref_cpu_32t', 'ref_rocm'):
            if f'{ref} vs {ours}' in pr:
                out[f'ours_vs_{ref}'] = pick(pr[f'{ref} vs {ours}'])
        if 'ref_cpu_32t vs ref_rocm' in pr:
            out['reference_cpu_vs_reference_rocm'] = pick(pr['ref_cpu_32t vs ref_rocm'])
        return out
    except Exception:       # noqa: BLE001
        return None


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=3)
    ap.add_argument('--warmup', type=int, default=1)
    ap.add_argument('--model', default='llama3-8b', choices=list(MODELS))
    ap.add_argument('--n-seq', type=int, default=128)
    ap.add_argument('--seq-len', type=int, default=2048)
    ap.add_argument('--calib-bs', type=int, default=1,
                    help='sequences per add_batch call (reference calib.bs; 1 = gptq_w_only.yml:12: 128 hook calls of [1, seq, K] per input, '
                         'each its own allocation, walked by ONE launch through the sample table; n_seq = one call on one tensor)')
    ap.add_argument('--dtype', default='bf16', choices=['bf16', 'f16'])
    ap.add_argument('--variant', default='w_only', choices=['w_only', 'vllm'],
                    help='w_only: asym g128 actorder dynamic groups (configs/quantization/methods/GPTQ/gptq_w_only.yml); '
                         'vllm: sym g128 static groups + INT4 pack (configs/quantization/backend/vllm/gptq_w4a16.yml)')
    ap.add_argument('--workload', default='gptq', choices=['gptq', 'awq', 'fp8'])
    ap.add_argument('--mode', default=None, choices=['independent', 'handoff', 'cooperative'])
    ap.add_argument('--exact-diag', type=int, default=1, help='0: keep the MFMA kernel\'s fp32 diagonal of H (A/B; rounds 1-5\'s default)')
    ap.add_argument('--merge-k1', type=int, default=1, help='0: one Hessian launch per input instead of one per width')
    ap.add_argument('--overlap', type=int, default=3,
                    help='streams for the subsets\' chains, widest first, the last takes the rest (3 beats 2 and 4); 0 = serial')
    ap.add_argument('--dry', action='store_true', help='GPU-less plumbing check (gloo + CPU stand-ins)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-extras', action='store_true', help='skip the secondary workloads reported under "extra" (N = 1 only)')
    return ap.parse_args(argv)


def maybe_spawn(args):
    """`python bench.py --gpus N` on a bare shell launches N ranks of itself."""
    if args.gpus <= 1 or 'WORLD_SIZE' in os.environ:
        return None
    import socket
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', f'--nproc-per-node={args.gpus}',
           '--master-addr', '127.0.0.1', '--master-port', str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    return subprocess.call(cmd, env=env)


# ---- synthetic inputs ------------------------------------------------------------------------------------
def synth_weight(R, K, seed, device, dtype):
    import torch
    g = torch.Generator(device=device).manual_seed(1000 + seed)
    w = torch.randn(R, K, generator=g, device=device, dtype=torch.float32) * 0.02
    n_out = max(1, K // 1000)
    idx = torch.randperm(K, generator=g, device=device)[:n_out]
    w[:, idx] *= 20.0
    return w.to(dtype)


def synth_acts(n_seq, seq, K, seed, device, dtype):
    import torch
    g = torch.Generator(device=device).manual_seed(2000 + seed)
    c = torch.exp(0.5 * torch.randn(K, generator=g, device=device))
    idx = torch.randperm(K, generator=g, device=device)[:8]
    c[idx] *= 100.0
    x = torch.empty((n_seq, seq, K), device=device, dtype=dtype)
    step = max(1, min(n_seq, (1 << 28) // (seq * K)))
    for i in range(0, n_seq, step):
        z = torch.randn((min(step, n_seq - i), seq, K), generator=g, device=device, dtype=torch.float32)
        x[i:i + step] = (z * c).to(dtype)
    return x


# ---- CPU baseline: the reference itself (oracle/_ref, built by __graft_entry__.build()) or, failing that, the port ---------
def _block_model(model, n_seq, t_h_seq, t_chol, t_loop, K):
    """One block as the reference executes it (7 Hessians / factorisations / loops), pieces scaled by flop count."""
    t_block, layers = 0.0, 0
    for _, k, ls in block_groups(model):
        for _, r in ls:
            t_block += t_h_seq * n_seq * (k / K) ** 2 + t_chol * (k / K) ** 3 + t_loop * (r / K) * (k / K) ** 2
            layers += 1
    return layers / t_block


def cpu_baseline_reference(model, n_seq, seq):
    cores = os.cpu_count() or 1
    K, nb = min(MODELS[model][0], 4096), 4
    script = os.path.join(ROOT, 'oracle', 'ref_baseline.py')
    if not os.path.isdir(os.path.join(ROOT, 'oracle', '_ref', 'llmc')):
        raise RuntimeError('oracle/_ref missing (built by __graft_entry__.build() where /root/reference exists)')
    r = subprocess.run([sys.executable, script, '--K', str(K), '--seq', str(seq), '--batches', str(nb)],
                       capture_output=True, text=True, timeout=240)
    line = [l for l in r.stdout.splitlines() if l.startswith('{')]
    if r.returncode != 0 or not line:
        raise RuntimeError('reference baseline failed: ' + (r.stderr or r.stdout)[-400:])
    t = json.loads(line[-1])
    return {
        'value': _block_model(model, n_seq, t['t_hessian_per_seq'], t['t_factor'], t['t_loop'], K),
        'unit': 'layers/s', 'cores': t['threads'], 'kind': 'reference',
        'host_cores': t.get('host_cores', cores), 'hessian_thread_sweep_s': t.get('hessian_thread_sweep_s'),
        'hessian_gflops': t.get('hessian_gflops'),
        'sample': (f"llmc's own GPTQ (oracle/_ref = /root/reference after its ci_check/change_files.py CPU rewrite), torch CPU, "
                   f"{t['threads']} threads for the Hessian GEMM (the best of a sweep over 8..{t.get('host_cores', cores)} threads: "
                   f"{t.get('hessian_gflops', 0.0):.0f} GFLOP/s) and {t.get('threads_small_ops', t['threads'])} for the factorisations / "
                   f"column loop (more threads make those slower): add_batch on {nb} of {n_seq} sequences of one {K}-channel input "
                   f"({t['t_hessian_per_seq']:.3f} s/seq), process_hessian_and_weights ({t['t_factor']:.2f} s) and weight_transform "
                   f"({t['t_loop']:.2f} s) of one {K}x{K} layer in full; other shapes scaled by flop count; 7 Hessians per block as the "
                   'reference executes them'),
    }


def cpu_baseline_port(model, n_seq, seq, cfg):
    """Fallback: the oracle (a CPU port of the reference path) timed on the host cores on a bounded sample."""
    import numpy as np

    from oracle import gptq_ref as G
    from oracle import quant_ref as Q
    cores = os.cpu_count() or 1
    K, nb = min(MODELS[model][0], 4096), 4
    rng = np.random.RandomState(0)
    H, n = np.zeros((K, K), dtype=np.float32), 0
    t0 = time.time()
    for _ in range(nb):
        H, n = G.add_batch(H, n, rng.standard_normal((seq, K)).astype(np.float32))
    t_h_seq = (time.time() - t0) / nb
    H += np.eye(K, dtype=np.float32) * 0.1
    W = (rng.standard_normal((K, K)) * 0.02).astype(np.float32)
    t0 = time.time()
    Wp, U = G.process_hessian_and_weights(W, H, G.hessian_sorting(H), cfg.percdamp)
    t_chol = time.time() - t0
    qmin, qmax = Q.int_range(cfg.bit, cfg.symmetric)
    t0 = time.time()
    G.weight_transform(Wp, U, cfg.symmetric, qmin, qmax, cfg.group_size)
    t_loop = time.time() - t0
    return {'value': _block_model(model, n_seq, t_h_seq, t_chol, t_loop, K), 'unit': 'layers/s', 'cores': cores, 'kind': 'port',
            'sample': (f'numpy/C oracle on {cores} host threads: Hessian = {nb} of {n_seq} sequences of one {K}-channel input '
                       f'({t_h_seq:.3f} s/seq), factorisation ({t_chol:.2f} s) and column loop ({t_loop:.2f} s) of one {K}x{K} layer '
                       'in full; other shapes scaled by flop count')}


def cpu_baseline(model, n_seq, seq, cfg):
    try:
        return cpu_baseline_reference(model, n_seq, seq)
    except Exception as e:
        out = cpu_baseline_port(model, n_seq, seq, cfg)
        out['sample'] += f' [reference baseline unavailable: {type(e).__name__}: {str(e)[:120]}]'
        return out


# ---- compute back ends: the HIP path (product) and CPU stand-ins for --dry ---------------------------------------------------
class HipOps:
    """The product path: llmc_amd classes over libllmc_hip.so. Raises without an MI355X."""

    def __init__(self, dev, cfg, variant, exact_diag=True):
        import torch

        from llmc_amd.compression.quantization import IntegerQuantizer, pack_lsb
        from llmc_amd.compression.quantization import gptq_pipeline as P
        from llmc_amd.compression.quantization.hessian import HessianAccumulator
        self.torch, self.dev, self.cfg, self.variant, self.exact_diag = torch, dev, cfg, variant, bool(exact_diag)
        self.P, self.Acc, self.pack_lsb = P, HessianAccumulator, pack_lsb
        self.wq = IntegerQuantizer(cfg.bit, cfg.symmetric, 'per_group', group_size=cfg.group_size)
        self.accs, self.hwork, self.streams = {}, {}, {}
        self.timing = None

    def acc(self, name, K):
        if name not in self.accs:
            self.accs[name] = self.Acc(K, self.dev, exact_diag=self.exact_diag)
            self.hwork[name] = self.torch.empty((K, K), dtype=self.torch.float32, device=self.dev)
        a = self.accs[name]
        a.timing = self.timing
        return a

    def feed(self, name, K, x, calib_bs):
        """add_batch calls of one input (deferred: nothing is launched yet)"""
        a = self.acc(name, K)
        a.reset()
        if isinstance(x, (list, tuple)):          # per-call tensors, each its own allocation (hook calls)
            for xi in x:
                a.add(xi)
        else:
            for i in range(0, x.shape[0], calib_bs):
                a.add(x[i:i + calib_bs])
        return a

    def hessian(self, name, K, x, calib_bs):
        return self.feed(name, K, x, calib_bs).H

    def hessians(self, items, calib_bs, merge=True):
        """items: [(name, K, x)] -> {name: H}. merge: inputs of one width share ONE launch (HessianAccumulator.flush_many)."""
        accs = [self.feed(n, K, x, calib_bs) for n, K, x in items]
        if merge:
            self.Acc.flush_many(accs, mix_widths=merge > 1)
        return {n: a.H for (n, _, _), a in zip(items, accs)}

    def static_qparams(self, weights):
        if not self.cfg.static_groups:
            return None
        out = []   # collect_block_qparams (base_blockwise_quantization.py:338-365): RTN qparams of the original weights
        for w in weights:
            _, s, z, _, _ = self.wq.get_tensor_qparams(w)
            out.append((s, None if self.cfg.symmetric else z))
        return out

    def quantize(self, name, weights, H, rows=None):
        static = self.static_qparams(weights)
        res = self.P.quantize_stacked(weights, H, self.cfg, static_qparams=static, h_work=self.hwork.get(name), rows=rows)
        outs = [{'weight': r.weight, 'scales': r.scales, 'zeros': r.zeros, 'perm': r.perm, 'loss': r.loss, 'info': r.info} for r in res]
        if self.variant == 'vllm' and rows is None:
            for r, (s, _) in zip(res, static):
                a = {'scales': s, 'zeros': self.torch.tensor(0.0), 'qmax': self.wq.qmax, 'qmin': self.wq.qmin}
                codes, _, _ = self.wq.real_quant_weight_static(r.weight, a)     # GPTQ.w_q (gptq.py:412-422)
                outs.append(self.pack_lsb(codes, self.cfg.bit))
        return outs

    def block_output(self, x, w, dtype):
        """x [n_seq, seq, K] (or the list of per-call tensors) times the quantized weight w [R, K] (fp32 after GPTQ):
        the layer's output over the calibration set, as FakeQuantLinear.forward computes it (HIP GEMM)."""
        from llmc_amd.compression.quantization import awq_ops
        if isinstance(x, (list, tuple)):
            x = self.torch.cat(list(x), 0)
        return awq_ops.linear_auto(x.reshape(-1, x.shape[-1]), w.to(dtype), None)

    def stream(self, i):
        if i not in self.streams:
            self.streams[i] = self.torch.cuda.Stream(device=self.dev)
        return self.streams[i]

    def helper_streams(self, enable):
        from llmc_amd import _ffi
        return _ffi.helper_streams(enable)

    def sync(self):
        self.torch.cuda.synchronize()


class DryOps:
    """CPU stand-ins with the same call shapes (plumbing check only: no kernel, no parity claim)."""

    def __init__(self, cfg):
        import torch
        self.torch, self.cfg, self.timing = torch, cfg, None

    def hessian(self, name, K, x, calib_bs):
        if isinstance(x, (list, tuple)):
            x = self.torch.cat(list(x), 0)
        xf = x.reshape(-1, K).float()
        return (xf.T @ xf) * (2.0 / x.shape[0])

    def hessians(self, items, calib_bs, merge=True):
        return {n: self.hessian(n, K, x, calib_bs) for n, K, x in items}

    def quantize(self, name, weights, H, rows=None):
        w = self.torch.cat([t.float() for t in weights], 0)
        if rows is not None:
            w = w[rows[0]:rows[1]]
        s = w.abs().amax(1, keepdim=True).clamp(min=1e-5) / 7
        return [{'weight': (w / s).round().clamp(-8, 7) * s + 0 * H.diagonal().mean()}]

    def block_output(self, x, w, dtype):
        if isinstance(x, (list, tuple)):
            x = self.torch.cat(list(x), 0)
        return x.reshape(-1, x.shape[-1]).float() @ w.float().T

    def stream(self, i):
        return None

    def sync(self):
        pass


def run_extras(args):
    """The secondary workloads, a few steps each, as child runs (same code path as their own bench lines)."""
    extras = os.path.join(ROOT, 'tools', 'bench_extras.py')
    me = os.path.abspath(__file__)
    runs = {
        'awq_llama3_8b': (me, ['--workload', 'awq', '--steps', '2', '--warmup', '1']),
        'gptq_vllm_variant_packed': (me, ['--variant', 'vllm', '--steps', '3', '--warmup', '1']),
        'gptq_llama3_70b_shapes': (me, ['--model', 'llama3-70b', '--steps', '2', '--warmup', '1']),
        'fp8_mixtral_8x7b_shapes': (me, ['--workload', 'fp8', '--steps', '3', '--warmup', '1']),
    }
    out = {}
    assert os.path.exists(extras)
    for key, (script, flags) in runs.items():
        cmd = [sys.executable, script, '--gpus', '1', '--no-cpu-baseline', '--no-extras', '--dtype', args.dtype] + flags
        try:
            r = subprocess.run(cmd, capture_output=True, text=True, timeout=300)
            line = [l for l in r.stdout.splitlines() if l.startswith('{')]
            if r.returncode != 0 or not line:
                out[key] = {'error': (r.stderr or r.stdout)[-300:]}
                continue
            j = json.loads(line[-1])
            out[key] = {k: j[k] for k in ('metric', 'value', 'value_with_auto_clip', 'unit', 'steps', 'warmup', 'ms_per_step', 'config', 'roofline') if k in j}
        except Exception as e:
            out[key] = {'error': f'{type(e).__name__}: {str(e)[:200]}'}
    return out


def main():
    args = parse_args()
    rc = maybe_spawn(args)
    if rc is not None:
        sys.exit(rc)
    if args.workload in ('awq', 'fp8'):
        sys.path.insert(0, os.path.join(ROOT, 'tools'))
        import bench_extras
        return bench_extras.run_awq(args) if args.workload == 'awq' else bench_extras.run_fp8(args)

    import torch
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if args.gpus != world and 'WORLD_SIZE' in os.environ and rank == 0:
        print(f'bench.py: --gpus {args.gpus} but launcher world size is {world}; using {world}', file=sys.stderr)
    if args.dry:
        dev = torch.device('cpu')
        args.model, args.n_seq, args.seq_len = 'tiny', max(world, 4), 64
    else:
        if not torch.cuda.is_available():
            raise SystemExit('bench.py needs an MI355X (no CPU fallback in llmc_amd); --dry checks the plumbing only')
        if local_rank >= torch.cuda.device_count():
            raise SystemExit(f'rank {rank}: local rank {local_rank} but only {torch.cuda.device_count()} GPU(s) visible')
        torch.cuda.set_device(local_rank)
        dev = torch.device('cuda', local_rank)
    auto = args.mode is None and world > 1          # north_star's partition if its pre-flight step works, else block-sharded without traffic
    if args.mode is None:
        args.mode = 'handoff' if auto else 'independent'
    ctl = None          # control-plane group (flags, max-over-ranks times): Gloo, CPU tensors — never the data path's RCCL
    if world > 1:
        import datetime
        import torch.distributed as dist
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        # a bounded collective timeout, and on expiry the watchdog aborts the COMMUNICATOR, not the process: the blocked call
        # raises, the measurement is recorded as failed, the line is still printed
        os.environ.setdefault('TORCH_NCCL_ASYNC_ERROR_HANDLING', '2')
        if args.dry:
            dist.init_process_group('gloo', timeout=datetime.timedelta(seconds=180))
            ctl = dist.group.WORLD
        else:
            dist.init_process_group('nccl', device_id=dev, timeout=datetime.timedelta(seconds=120))
            ctl = dist.new_group(backend='gloo', timeout=datetime.timedelta(seconds=300))

    def ctl_reduce(value, op):
        """max / min over ranks of a Python number through the control group"""
        if world == 1:
            return value
        t = torch.tensor([float(value)], dtype=torch.float64)
        torch.distributed.all_reduce(t, op=op, group=ctl)
        return float(t.item())

    from llmc_amd.compression.quantization.gptq_pipeline import GptqConfig
    from llmc_amd.dist import layer_shard as LS

    dtype = torch.float32 if args.dry else (torch.bfloat16 if args.dtype == 'bf16' else torch.float16)
    cfg = GptqConfig(bit=4, symmetric=args.variant != 'w_only', group_size=128, actorder=True, static_groups=args.variant != 'w_only')
    groups = block_groups(args.model)
    ops = DryOps(cfg) if args.dry else HipOps(dev, cfg, args.variant, exact_diag=bool(args.exact_diag))
    n_layers_block = sum(len(ls) for _, _, ls in groups)
    timing = []
    # payload each mode moves between ranks per step and rank (a ring all-reduce moves 2 (N - 1) / N of its payload): handoff =
    # the block's first input; cooperative = per subset the broadcast Hessian (stacked subsets), the broadcast activations
    # (single-layer subsets) or the all-reduced Hessian (sample-sharded wide subset)
    esz = 4 if args.dry else 2
    interrank_bytes = {
        'independent': 0,
        'handoff': args.n_seq * args.seq_len * groups[0][1] * esz,
        'cooperative': sum((K * K * 4) if (K > 8192 or len(ls) > 1) else args.n_seq * args.seq_len * K * esz for _, K, ls in groups),
    }

    def prepare(mode):
        """Resident synthetic data and the step function of one mode. independent / handoff: every rank owns different blocks
        (different seeds). cooperative: one block; rank 0 holds the full activations of the broadcast subsets, every rank holds
        its own sequences (rank::world) of the sample-sharded ones; weights are the same on every rank."""
        coop = mode == 'cooperative' and world > 1
        handoff = mode == 'handoff' and world > 1
        acts, weights, plan = {}, {}, {}
        for gi, (name, K, layers) in enumerate(groups):
            seed_r = 0 if coop else rank
            plan[name] = ('sample' if K > 8192 else 'broadcast') if coop else 'local'
            if plan[name] == 'sample':
                acts[name] = synth_acts(len(range(rank, args.n_seq, world)), args.seq_len, K, 64 * rank + gi, dev, dtype)
            elif plan[name] == 'broadcast':
                acts[name] = synth_acts(args.n_seq, args.seq_len, K, gi, dev, dtype) if rank == 0 else None
            else:
                acts[name] = synth_acts(args.n_seq, args.seq_len, K, seed_r * 64 + gi, dev, dtype)
                # the block's first input arrives as ONE tensor from the previous owner under the hand-off
                if not (handoff and gi == 0) and args.calib_bs < args.n_seq and not args.dry:
                    # the hook calls' tensors: one allocation per call (calib.bs sequences each), nothing contiguous across calls
                    x = acts[name]
                    acts[name] = [x[i:i + args.calib_bs].clone() for i in range(0, args.n_seq, args.calib_bs)]
                    del x
            weights[name] = [synth_weight(R, K, seed_r * 64 + gi * 8 + li, dev, dtype) for li, (_, R) in enumerate(layers)]

        def step_independent(record):
            """K1 first — the MFMA kernel owns every CU, nothing overlaps with it: the Hessians of one width in one launch — then
            the subsets' factorisations and column loops, independent chains, on --overlap streams, widest first, the internal
            helper streams off (measured schedules: profiles/NOTES.md, r05_schedule_experiments.txt, r06_stream_map.txt)."""
            ops.timing = timing if record else None
            Hs = ops.hessians([(name, K, acts[name]) for name, K, _ in groups], args.calib_bs, merge=args.merge_k1)
            if args.overlap <= 1 or args.dry:
                return [ops.quantize(name, weights[name], Hs[name]) for name, _, _ in groups]
            cur = torch.cuda.current_stream()
            order = sorted(range(len(groups)), key=lambda i: -groups[i][1] * sum(r for _, r in groups[i][2]))
            slot, used = {}, []
            for si, gi in enumerate(order):
                name = groups[gi][0]
                st = ops.stream(min(si, args.overlap - 1))
                st.wait_stream(cur)
                with torch.cuda.stream(st), ops.helper_streams(False):
                    slot[gi] = ops.quantize(name, weights[name], Hs[name])
                used.append(st)
            for st in set(used):
                cur.wait_stream(st)
            return [slot[gi] for gi in range(len(groups))]      # in block order (o_proj's result is outs[1])

        def step_cooperative(record):
            ops.timing = timing if record else None
            outs = []
            for name, K, layers in groups:
                shape = (args.n_seq, args.seq_len, K)
                if plan[name] == 'sample':
                    outs.append(LS.run_subset_sample_sharded(
                        acts[name], weights[name],
                        hessian_fn=lambda x, n=name, k=K: ops.hessian(n, k, x, args.calib_bs),
                        quantize_rows_fn=lambda ws, H, rows, n=name: ops.quantize(n, ws, H, rows=rows)))
                else:
                    share = 'hessian' if len(layers) > 1 else 'activations'
                    outs.append(LS.run_block_cooperative(
                        list(range(len(layers))), acts[name], 0,
                        lambda li, shared, n=name, k=K, sh=share: ops.quantize(
                            n + str(li), [weights[n][li]],
                            shared if sh == 'hessian' else ops.hessian(n, k, shared, args.calib_bs)),
                        (shape, dtype, dev), share=share,
                        hessian_fn=lambda x, n=name, k=K: ops.hessian(n, k, x, args.calib_bs), gather_to=None, to_cpu=False))   # results stay with their owners
            return outs

        # ---- handoff: block-sharded ownership with a block's OUTPUT handed to the owner of the next block. In a model run the
        # inputs of block b are the outputs of block b - 1 (base_blockwise_quantization.py:367-402): whoever owns block b - 1
        # produces them and sends them on. Here every rank sends a tensor it has just computed from this step's result — o_proj's
        # quantized weights applied to o_proj's calibration input, [n_seq, seq, hidden] in the model dtype (ops.block_output: the
        # HIP GEMM of FakeQuantLinear.forward) — to its ring successor and receives what its step after next consumes from its
        # predecessor: the transfer of step t's output runs on its own stream under step t + 1's kernels (double-buffered). The
        # payload is produced by the step, so it cannot be elided, and it changes every step.
        hand = {}
        if handoff:
            name0 = groups[0][0]
            hand.update(cur=acts[name0], nxt=torch.empty_like(acts[name0]), out=acts[name0],
                        bytes=acts[name0].numel() * acts[name0].element_size(),
                        stream=None if args.dry else torch.cuda.Stream(device=dev))

        def handoff_start():
            import torch.distributed as dist
            p2p = [dist.P2POp(dist.isend, hand['out'], (rank + 1) % world), dist.P2POp(dist.irecv, hand['nxt'], (rank - 1) % world)]
            if hand['stream'] is None:
                hand['reqs'] = dist.batch_isend_irecv(p2p)
                return
            hand['stream'].wait_stream(torch.cuda.current_stream())      # the buffers are ready
            with torch.cuda.stream(hand['stream']):
                hand['reqs'] = dist.batch_isend_irecv(p2p)

        def handoff_finish(new_out):
            if hand['stream'] is None:
                for r in hand['reqs']:
                    r.wait()
            else:
                with torch.cuda.stream(hand['stream']):
                    for r in hand['reqs']:
                        r.wait()
                torch.cuda.current_stream().wait_stream(hand['stream'])   # the next step reads what arrived
            hand['cur'], hand['nxt'] = hand['nxt'], hand['cur']
            hand['out'] = new_out
            acts[groups[0][0]] = hand['cur']

        def step_handoff(record):
            if args.dry and os.environ.get('LLMC_BENCH_DRY_FAIL_HANDOFF') == '1':
                raise RuntimeError('injected hand-off failure (dry-run test of the fallback)')
            handoff_start()
            out = step_independent(record)
            y = ops.block_output(acts[groups[1][0]], out[1][0]['weight'], dtype)   # this step's block output: sent during the next step
            handoff_finish(y.reshape(hand['cur'].shape))
            return out

        step = step_cooperative if coop else (step_handoff if handoff else step_independent)
        return step, {'coop': coop, 'handoff': handoff, 'hand': hand, 'acts': acts, 'weights': weights, 'plan': plan}

    def barrier():
        if world > 1:
            torch.distributed.barrier(group=ctl)
        ops.sync()

    def release(ctx):
        for k in ('acts', 'weights', 'hand', 'plan'):
            ctx[k].clear()
        for k in ('accs', 'hwork'):
            if hasattr(ops, k):
                getattr(ops, k).clear()
        import gc
        gc.collect()
        if not args.dry:
            torch.cuda.empty_cache()

    def try_steps(mode, n_warm, n_timed):
        """A few steps of `mode` behind try / except: (layers/s | None, error | None). The ranks meet ONLY in control-group
        reductions that every rank reaches whether or not its phase raised (no barrier inside a try: a rank that failed never
        leaves its peers waiting in one — ADVICE r03), and skip the timed phase together if the warm-up failed anywhere."""
        ok, val, err, ctx2 = 1, 0.0, None, None

        def phase(n):
            nonlocal ok, err
            try:
                t1 = time.perf_counter()
                for _ in range(n):
                    step2(False)
                ops.sync()
                return time.perf_counter() - t1
            except Exception as e:      # noqa: BLE001
                ok, err = 0, err or f'{type(e).__name__}: {str(e)[:200]}'
                return 0.0
        try:
            step2, ctx2 = prepare(mode)
        except Exception as e:          # noqa: BLE001
            ok, err = 0, f'{type(e).__name__}: {str(e)[:200]}'
        if ctl_reduce(ok, torch.distributed.ReduceOp.MIN):
            phase(n_warm)
        if ctl_reduce(ok, torch.distributed.ReduceOp.MIN):
            val = phase(n_timed)
        ok = ctl_reduce(ok, torch.distributed.ReduceOp.MIN)
        try:
            if ctx2 is not None:
                release(ctx2)
        except Exception:           # noqa: BLE001
            pass
        if not ok:
            return None, err or 'failed on another rank'
        t = ctl_reduce(val, torch.distributed.ReduceOp.MAX)
        return n_layers_block * (1 if mode == 'cooperative' else world) * n_timed / t, None

    handoff_error = None
    if auto:
        # pre-flight: ONE hand-off step on every rank. Only if all of them succeed does the timed region exercise north_star's
        # partition; otherwise it runs the same ownership without data-path traffic and the line records why.
        v, handoff_error = try_steps('handoff', 1, 1)
        if v is None:
            args.mode = 'independent'
    step, ctx = prepare(args.mode)
    coop, handoff, hand = ctx['coop'], ctx['handoff'], ctx['hand']
    last = None
    if handoff and not auto:
        # an explicit --mode handoff keeps its safety net: if the transfer raises, all ranks agree to drop it
        ok = 1
        try:
            step(False)
            ops.sync()
        except Exception as e:      # noqa: BLE001
            ok, handoff_error = 0, f'{type(e).__name__}: {str(e)[:200]}'
        if ctl_reduce(ok, torch.distributed.ReduceOp.MIN) == 0:
            handoff_error = handoff_error or 'the hand-off failed on another rank'
            release(ctx)
            step, ctx = prepare('independent')
            coop, handoff, hand = ctx['coop'], ctx['handoff'], ctx['hand']
    for _ in range(args.warmup):
        step(False)
    barrier()
    # per-step end marks on the launch stream (every chain stream is joined into it at the end of a step): the median step
    # SURVEY 8d asks for, beside the contract's mean over exactly K steps
    marks = []
    if not args.dry:
        marks.append(torch.cuda.Event(enable_timing=True))
        marks[0].record()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        last = step(True)
        if not args.dry:
            marks.append(torch.cuda.Event(enable_timing=True))
            marks[-1].record()
    barrier()
    dt = time.perf_counter() - t0
    step_ms = sorted(marks[i].elapsed_time(marks[i + 1]) for i in range(len(marks) - 1))
    med_ms = (step_ms[len(step_ms) // 2] if len(step_ms) % 2 else 0.5 * (step_ms[len(step_ms) // 2 - 1] + step_ms[len(step_ms) // 2])) if step_ms else None
    barrier_timeouts = None       # round barriers of the last step's SYRK launches that gave up (never silent)
    if not args.dry and getattr(ops, 'accs', None):
        try:
            barrier_timeouts = sum(a.barrier_timeouts() for a in ops.accs.values())
        except Exception:        # noqa: BLE001
            barrier_timeouts = None

    def _infos(o):    # deferred positive-definiteness check of the factorisations (the classes check once per subset; here after timing)
        if isinstance(o, dict):
            if torch.is_tensor(o.get('info')):
                yield o['info']
            for v in o.values():
                yield from _infos(v)
        elif isinstance(o, (list, tuple)):
            for v in o:
                yield from _infos(v)
    bad = [int(t.item()) for t in _infos(last) if int(t.item()) != 0]
    if bad:
        raise SystemExit(f'bench.py: a Hessian was not positive definite (leading minors {bad}): results invalid')
    # what THIS run's last step computed, against fp64 on the same resident samples: diag(H) is the only part of the chain
    # whose last bits are used before the factorisation (actorder's sort key, the damping mean) and is not the reference's
    # arithmetic bit for bit; everything downstream is bit-exact given H (tests). One narrow and the wide input.
    parity_live = None
    if not args.dry and not coop and getattr(ops, 'accs', None):
        try:
            parity_live = {}
            for name in (groups[1][0], groups[-1][0]):
                x = ctx['acts'][name]
                xs = list(x) if isinstance(x, (list, tuple)) else [x]
                d = sum((xi.reshape(-1, xi.shape[-1])[i:i + 16384].double() ** 2).sum(0) for xi in xs
                        for i in range(0, xi.reshape(-1, xi.shape[-1]).shape[0], 16384)) * (2.0 / args.n_seq)
                h = torch.diagonal(ops.accs[name].H).double()
                parity_live[name] = {'K': int(h.numel()), 'H_diag_rel_max_vs_fp64': float(((h - d).abs() / d).max())}
            parity_live['note'] = ('measured by this run after the timed region: diag(H) of the last step against the fp64 sum over '
                                   'the same samples (an fp32 rounding of the exact value is <= 6e-8; the reference\'s own sgemm leaves '
                                   '1.2-1.7e-6 at K = 14336, profiles/r05_parity_envelope_full_down.txt)')
        except Exception as e:       # noqa: BLE001
            parity_live = {'error': f'{type(e).__name__}: {str(e)[:160]}'}
    dt = ctl_reduce(dt, torch.distributed.ReduceOp.MAX) if world > 1 else dt
    value_mode = 'cooperative' if coop else ('handoff_rccl' if handoff else 'independent_no_comm')
    if args.dry and handoff:
        value_mode = 'handoff_gloo_dry'
    parallelism = 'single GPU'
    if world > 1:
        parallelism = (
            f'cooperative x{world}: Hessian/activation broadcast + sample-sharded all_reduce, row-sharded column loop' if coop else
            (f'block-sharded x{world}, every block\'s output ({hand["bytes"] / 2**30:.2f} GiB per step and rank: o_proj\'s quantized weights '
             f'applied to its calibration input) handed owner-to-owner over RCCL send/recv (xGMI ring), overlapped with the next step'
             if handoff else f'layer-sharded x{world}, no data-path traffic'))

    # ---- the other modes, a few steps each, outside the timed region (N > 1 default only)
    secondary = {}
    if auto:
        last = None
        release(ctx)
        k2 = max(2, min(4, args.steps))
        for m in (['independent'] if handoff else []) + ['cooperative']:
            v, err = try_steps(m, 1, k2)
            secondary[m + ('_value' if v is not None else '_error')] = v if v is not None else err

    # ---- roofline of the dominant kernel (k_syrk4), HIP events on the launch stream, this rank
    fl = sum(f for (_, _, _, f, _) in timing)
    ms = sum(e0.elapsed_time(e1) for (e0, e1, _, _, _) in timing)
    ms_fix = sum(e1.elapsed_time(e2) for (_, e1, e2, _, _) in timing)
    n_launch = len(timing)
    achieved = fl / (ms * 1e-3) / 1e12 if ms > 0 else 0.0
    achieved_fix = fl / ((ms + ms_fix) * 1e-3) / 1e12 if ms > 0 else 0.0
    # HBM-side bytes of the dominant kernel come from PMC passes that cannot run inside the timed process (rocprofv3 --pmc
    # FETCH_SIZE / WRITE_SIZE in separate passes, tools/gpu_job.sh pmc); the committed summary is reported with its source
    traffic, traffic_src = None, None
    for tname in TRAFFIC_FILES:
        tpath = os.path.join(ROOT, 'profiles', tname)
        if args.model == 'llama3-8b' and args.n_seq == 128 and args.seq_len == 2048 and os.path.exists(tpath):
            try:
                tj = json.load(open(tpath))['k_syrk']
                traffic, traffic_src = tj['hbm_bytes_per_launch'], f'profiles/{tname} (' + tj['note'] + ')'
                break
            except Exception:
                traffic = None

    if rank == 0:
        layers_step = n_layers_block * (1 if coop else world)
        out = {
            'metric': 'layers/sec (GPTQ W4A16, %s Linear shapes, %dx%d calib)' % (
                {'llama3-8b': 'Llama-3-8B', 'llama3-70b': 'Llama-3-70B'}.get(args.model, args.model), args.n_seq, args.seq_len),
            'value': layers_step * args.steps / dt, 'unit': 'layers/s', 'n_gpus': world, 'steps': args.steps,
            'warmup': args.warmup, 'ms_per_step': dt / args.steps * 1e3,
            'ms_per_step_median': med_ms, 'ms_per_step_min_max': [step_ms[0], step_ms[-1]] if step_ms else None,
            'value_at_median_step': (layers_step / (med_ms * 1e-3)) if med_ms else None,      # this rank's steps; `value` is the contract's mean
            'higher_is_better': True,
            'scaling': 'strong' if coop else 'weak', 'vs_baseline': None, 'dtype': args.dtype,
            'data': 'dry-run (CPU stand-ins, no kernels): plumbing only' if args.dry else 'synthetic',
            'config': {
                'workload': (f'GPTQ W4A16 g128 ({args.variant}) on {args.model}-shaped random-init Linear layers, '
                             + ('1 transformer block (7 Linear, 4 distinct inputs) per step shared by all GPUs'
                                if coop else '1 transformer block (7 Linear, 4 distinct inputs) per step per GPU')),
                'n_seq': args.n_seq, 'seq_len': args.seq_len, 'calib_bs': args.calib_bs,
                'hessian_feed': (f'{-(-args.n_seq // args.calib_bs)} add_batch calls per input (calib.bs = {args.calib_bs}, one allocation '
                                 'per call), the inputs of one width walked by ONE launch through the sample table, no staging copy'
                                 if args.calib_bs < args.n_seq else 'one add_batch call per input on one resident tensor'),
                # w_only: a layer ends at compensated weights + scales / zeros (the reference cannot export actorder + dynamic
                # groups either, gptq.py:455-457); vllm: + INT4 codes packed (SURVEY 8d's full definition) = `value_packed`
                'packs_codes': args.variant == 'vllm',
                'symmetric': cfg.symmetric, 'actorder': cfg.actorder, 'static_groups': cfg.static_groups,
                'subset_overlap_streams': 0 if (args.dry or coop) else args.overlap,
                'hessian_exact_diag': bool(args.exact_diag), 'hessians_of_one_width_in_one_launch': bool(args.merge_k1),
                'parallelism': parallelism,
                'parity_envelope': parity_envelope_summary(args),
            },
            'parity_live': parity_live,
            'roofline': {
                'bound': 'mfma', 'achieved': achieved, 'peak': PEAK_MFMA_16BIT / 1e12, 'unit': 'TFLOP/s',
                'frac': achieved * 1e12 / PEAK_MFMA_16BIT, 'traffic': traffic, 'traffic_source': traffic_src,
                'kernel': 'k_syrk4 (llmc_hessian_accum_multi_partials)', 'launches': n_launch,
                'algorithmic_flops_per_launch': fl / max(1, n_launch),
                'avg_launch_ms': ms / max(1, n_launch),
                'achieved_incl_fixup': achieved_fix, 'avg_fixup_ms': ms_fix / max(1, n_launch),
                'round_barrier_timeouts_last_step': barrier_timeouts,
            },
        }
        if world > 1:
            out['value_mode'] = value_mode       # what the timed value exercised
            out['rccl_world'] = (torch.distributed.get_world_size() if not args.dry else None)        # ranks of the nccl (= RCCL) group
            out['data_backend'] = 'gloo (dry run)' if args.dry else torch.distributed.get_backend()
            out['interrank_bytes_per_step_per_rank'] = interrank_bytes
            out.update(secondary)    # independent_value / cooperative_value (layers/s, a few steps each, outside the timed region) or *_error
        if handoff_error is not None:
            out['handoff_error'] = handoff_error              # the run fell back to the ownership without the hand-off
        if world == 1 and not args.no_extras and not args.dry and args.model == 'llama3-8b' and args.variant == 'w_only':
            last = None
            release(ctx)            # the secondary workloads are child processes on the same GPU
            out['extra'] = run_extras(args)
            pk = out['extra'].get('gptq_vllm_variant_packed', {})
            # SURVEY 8d's full layer definition (... -> INT4 codes packed), first-class beside `value` (a few steps of the same script)
            out['value_packed'], out['ms_per_step_packed'] = pk.get('value'), pk.get('ms_per_step')
        if world == 1 and not args.no_cpu_baseline and not args.dry:
            try:
                out['cpu_baseline'] = cpu_baseline(args.model, args.n_seq, args.seq_len, cfg)
            except Exception as e:  # the baseline must never take the GPU number down with it
                out['cpu_baseline'] = {'value': None, 'unit': 'layers/s', 'cores': os.cpu_count(), 'kind': 'port',
                                       'sample': f'failed: {type(e).__name__}: {e}'}
        print(json.dumps(out), flush=True)
    if world > 1:
        try:
            torch.distributed.destroy_process_group()
        except Exception:      # noqa: BLE001 (a communicator aborted by a failed measurement)
            pass


if __name__ == '__main__':
    main()
