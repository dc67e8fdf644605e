"""Secondary workloads of bench.py (`--workload awq | fp8`): BASELINE.json configs[2] (AWQ W4A16 g128 scale search + fake-quant
evaluation on Llama-3-8B shapes) and configs[4] (FP8 e4m3 per-tensor quantization on Mixtral-8x7B block shapes). Same JSON contract
as bench.py; bench.py runs them as child processes and reports them under "extra"."""
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

PEAK_MFMA_16BIT = 2.5e15  # dense bf16/f16 MFMA, /opt/skills/guides/MI355X_MICROARCH.md


def _shared():
    import bench
    return bench.block_groups, bench.synth_acts, bench.synth_weight


def cpu_baseline_reference_awq(groups, N):
    """The reference's own AWQ grid step (oracle/ref_baseline.py --workload awq: Awq.get_scales / fake_quantize_weight /
    inspect_module_forward / calculate_loss of oracle/_ref) on two token counts of one 4096 x 4096 layer: t = a + b * tokens
    separates the per-weight work from the per-token work; a block = 4 searches of 21 evaluations, scaled by R * K."""
    script = os.path.join(ROOT, 'oracle', 'ref_baseline.py')
    if not os.path.isdir(os.path.join(ROOT, 'oracle', '_ref', 'llmc')):
        raise RuntimeError('oracle/_ref missing (built by __graft_entry__.build() where /root/reference exists)')
    K0, ts = 4096, {}
    thr = min(os.cpu_count() or 1, 32)      # 256 threads: 3.4 s per step whatever the token count (thread overhead); 16-32 suit these ops
    for tok in (2048, 4096):
        r = subprocess.run([sys.executable, script, '--workload', 'awq', '--K', str(K0), '--tokens', str(tok), '--threads', str(thr)],
                           capture_output=True, text=True, timeout=200)
        line = [l for l in r.stdout.splitlines() if l.startswith('{')]
        if r.returncode != 0 or not line:
            raise RuntimeError('reference AWQ baseline failed: ' + (r.stderr or r.stdout)[-300:])
        ts[tok] = json.loads(line[-1])
    b = max(0.0, (ts[4096]['t_grid_step'] - ts[2048]['t_grid_step']) / 2048)
    a = max(0.0, ts[2048]['t_grid_step'] - b * 2048)
    t_block, layers = 0.0, 0
    for _, K, ls in groups:
        rk = sum(r for _, r in ls) * K / (K0 * K0)
        t_block += 21 * (a + b * N) * rk
        layers += len(ls)
    return {'value': layers / t_block, 'unit': 'layers/s', 'cores': ts[2048]['threads'], 'kind': 'reference',
            'sample': (f"llmc's own Awq methods (oracle/_ref), torch CPU, {ts[2048]['threads']} threads: 3 grid steps (get_scales, x / s, "
                       f"fake_quantize_weight, F.linear through inspect_module_forward, calculate_loss, state-dict restore) of one "
                       f"{K0}x{K0} layer on 2048 and 4096 tokens ({ts[2048]['t_grid_step']:.2f} s and {ts[4096]['t_grid_step']:.2f} s per "
                       f'step); linear in tokens and in R*K to {N} tokens and the 4 stacked subsets, 21 evaluations per search')}



# ---------------------------------------------------------------------------------------------------------------
# AWQ workload (BASELINE.json configs[2]): per block, the four subsets' 20-point scale searches (awq.py:179-253) with
# inspect = the subset's Linear layers, N = 128 x 512 tokens in one batch, W4 symmetric g128, trans v2
# ---------------------------------------------------------------------------------------------------------------
def run_awq(args):
    import torch
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs an MI355X (no CPU fallback in llmc_amd)')
    torch.cuda.set_device(local_rank)
    dev = torch.device('cuda', local_rank)
    if world > 1:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        torch.distributed.init_process_group('nccl', device_id=dev)
    from llmc_amd.compression.quantization import IntegerQuantizer, awq_ops
    from llmc_amd.compression.quantization.awq_pipeline import search_scale_stacked
    dtype = torch.bfloat16 if args.dtype == 'bf16' else torch.float16
    n_seq, seq = 128, 512                                  # configs/quantization/methods/Awq/awq_w_only.yml:12-14
    N = n_seq * seq
    wq = IntegerQuantizer(4, True, 'per_group', group_size=128)
    block_groups, synth_acts, synth_weight = _shared()
    groups = block_groups(args.model)
    acts = {name: synth_acts(n_seq, seq, K, rank * 64 + gi, dev, dtype).reshape(N, K) for gi, (name, K, _) in enumerate(groups)}
    weights = {name: [synth_weight(R, K, rank * 64 + gi * 8 + li, dev, dtype) for li, (_, R) in enumerate(layers)]
               for gi, (name, K, layers) in enumerate(groups)}
    gemm_ev = []

    def step(record):
        out = []
        for name, K, layers in groups:
            out.append(search_scale_stacked(weights[name], acts[name], wq, 'v2', timing=gemm_ev if record else None))
        return out

    def barrier():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()
    for _ in range(args.warmup):
        step(False)
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step(True)
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        dt = float(t.item())
    # weight_clip: True in awq_w_only.yml runs AutoClipper after the scale search (auto_clip.py:37-77: every Linear except q / k,
    # n_sample_token = calib seq_len = 512). Not part of configs[2]'s metric (scale search + fake-quant evaluation); timed once,
    # outside the timed region, and reported beside it.
    clip_ms = None
    try:
        step_tok = max(1, N // seq)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for name, K, layers in groups:
            xs = acts[name][0::step_tok]
            for (lname, R), w in zip(layers, weights[name]):
                if lname in ('q_proj', 'k_proj'):
                    continue
                awq_ops.clip_search(w, xs, wq, True)
        e1.record()
        torch.cuda.synchronize()
        clip_ms = e0.elapsed_time(e1)
    except Exception:
        clip_ms = None
    n_layers = sum(len(ls) for _, _, ls in groups)
    fl_eval = sum(2.0 * N * sum(r for _, r in ls) * K for _, K, ls in groups)        # one evaluation of every subset
    fl = sum(f for _, _, f in gemm_ev)
    ms = sum(e0.elapsed_time(e1) for e0, e1, _ in gemm_ev)
    if rank == 0:
        ach = fl / (ms * 1e-3) / 1e12 if ms > 0 else 0.0
        awq_traffic = awq_traffic_src = None
        tpath = os.path.join(ROOT, 'profiles', 'r02_pmc_traffic_awq.json')
        if args.model == 'llama3-8b' and os.path.exists(tpath):
            try:
                tj = json.load(open(tpath))['k_linear_eval4']
                awq_traffic, awq_traffic_src = tj['hbm_bytes_per_launch'], 'profiles/r02_pmc_traffic_awq.json (' + tj['note'] + ')'
            except Exception:
                pass
        cpu = None
        if world == 1 and not args.no_cpu_baseline:
            try:
                cpu = cpu_baseline_reference_awq(groups, N)
            except Exception as e:
                cpu = {'value': None, 'unit': 'layers/s', 'cores': os.cpu_count(), 'kind': 'reference',
                       'sample': f'failed: {type(e).__name__}: {str(e)[:160]}'}
        print(json.dumps({
            'metric': 'layers/sec (AWQ W4A16 g128 scale search + fake-quant eval, %s Linear shapes, 128x512 calib)' % args.model,
            'value': n_layers * args.steps * world / dt, 'unit': 'layers/s', 'n_gpus': world, 'steps': args.steps,
            'warmup': args.warmup, 'ms_per_step': dt / args.steps * 1e3,
            # the shipped awq_w_only.yml also enables weight_clip: the same block with AutoClipper's search after the scale search
            'value_with_auto_clip': (n_layers * world / (dt / args.steps + clip_ms * 1e-3)) if clip_ms else None,
            'higher_is_better': True, 'scaling': 'weak',
            'vs_baseline': None, 'dtype': args.dtype, 'data': 'synthetic',
            'config': {'workload': f'AWQ W4A16 g128 sym, trans v2, 20-point scale search with inspect = the Linear layers, '
                                   f'{args.model}-shaped random-init layers, 1 block (7 Linear, 4 subsets) per step per GPU',
                       'n_seq': n_seq, 'seq_len': seq, 'parallelism': f'layer-sharded x{world}' if world > 1 else 'single GPU',
                       'auto_clip_ms_per_block_not_in_value': clip_ms},
            'roofline': {'bound': 'mfma', 'achieved': ach, 'peak': PEAK_MFMA_16BIT / 1e12, 'unit': 'TFLOP/s',
                         'frac': ach * 1e12 / PEAK_MFMA_16BIT, 'traffic': awq_traffic, 'traffic_source': awq_traffic_src,
                         'kernel': 'k_linear_eval4 (llmc_linear_eval_kt, the 21 products of a search; k_linear_eval when K % 128 != 0)', 'launches': len(gemm_ev),
                         'algorithmic_flops_per_launch': fl / max(1, len(gemm_ev)), 'avg_launch_ms': ms / max(1, len(gemm_ev)),
                         'whole_search_tflops': 21 * fl_eval * args.steps / dt / 1e12},
            'cpu_baseline': cpu,
        }), flush=True)
    if world > 1:
        torch.distributed.destroy_process_group()


PEAK_HBM = 8.0e12   # spec, /opt/skills/guides/MI355X_MICROARCH.md (about 6.3e12 measured)


def run_fp8(args):
    """BASELINE configs[4]: FP8 (e4m3) per-tensor weight + activation quantization on Mixtral-8x7B expert Linear shapes
    (configs/quantization/backend/vllm/fp8/*.yml with per_tensor granularity). Quantization time, per block: every Linear's
    weight -> absmax -> scale -> e4m3 codes (FloatQuantizer.real_quant_weight_dynamic: llmc_minmax_qparams + llmc_fp8_quant),
    and the static per-tensor range of every Linear input over the calibration tokens (mean of per-sample min / max,
    base_blockwise_quantization.py:253-263: llmc_minmax_samples, one launch pair per input)."""
    import torch
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs an MI355X (no CPU fallback in llmc_amd)')
    torch.cuda.set_device(local_rank)
    dev = torch.device('cuda', local_rank)
    if world > 1:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        torch.distributed.init_process_group('nccl', device_id=dev)
    _, synth_acts, synth_weight = _shared()
    from llmc_amd.compression.quantization import FloatQuantizer
    dtype = torch.bfloat16 if args.dtype == 'bf16' else torch.float16
    h, kv, ffn, n_exp = 4096, 1024, 14336, 8              # Mixtral-8x7B: llmc/models/mixtral.py:43-86
    layers = [('q_proj', h, h), ('k_proj', kv, h), ('v_proj', kv, h), ('o_proj', h, h)]
    for e in range(n_exp):
        layers += [(f'experts.{e}.w1', ffn, h), (f'experts.{e}.w3', ffn, h), (f'experts.{e}.w2', h, ffn)]
    n_seq, seq = 128, 512
    wq = FloatQuantizer('e4m3', True, 'per_tensor', use_qtorch=True)
    weights = [synth_weight(R, K, rank * 64 + i, dev, dtype) for i, (_, R, K) in enumerate(layers)]
    # inputs: attention in, o_proj in (all tokens); an expert sees top-2 of 8 = a quarter of the tokens on average
    acts = {'attn_in': synth_acts(n_seq, seq, h, rank * 64 + 1, dev, dtype), 'o_in': synth_acts(n_seq, seq, h, rank * 64 + 2, dev, dtype)}
    for e in range(n_exp):
        acts[f'e{e}_in'] = synth_acts(n_seq // 4, seq, h, rank * 64 + 8 + e, dev, dtype)
        acts[f'e{e}_mid'] = synth_acts(n_seq // 4, seq, ffn, rank * 64 + 24 + e, dev, dtype)

    from llmc_amd.compression.quantization.hist_range import sample_minmax
    samples = {k: [x[i] for i in range(x.shape[0])] for k, x in acts.items()}      # what the hooks deliver: one tensor per sample

    def step():
        out = [wq.real_quant_weight_dynamic(w) for w in weights]
        for k in acts:                # static_minmax: mean over samples of the per-sample range (register_act_qparams)
            mn, mx = sample_minmax(samples[k])
            out.append(torch.max(mx.mean().abs(), mn.mean().abs()).clamp(min=1e-5) / 448.0)
        return out

    def barrier():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()
    for _ in range(args.warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        dt = float(t.item())
    # the dominant kernel alone: the cast of one 14336 x 4096 weight, HIP events on the current stream
    w = weights[4]
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 10
    wq.real_quant_weight_dynamic(w)
    e0.record()
    for _ in range(reps):
        wq.real_quant_weight_dynamic(w)
    e1.record()
    torch.cuda.synchronize()
    t_w = e0.elapsed_time(e1) * 1e-3 / reps
    wbytes = 5.0 * w.numel()          # 2 B min/max pass + 2 B cast pass + 1 B codes
    if rank == 0:
        elems = sum(R * K for _, R, K in layers)
        abytes = sum(2.0 * x.numel() for x in acts.values())
        print(json.dumps({
            'metric': 'layers/sec (FP8 e4m3 per-tensor weight quantization + static activation ranges, Mixtral-8x7B block shapes)',
            'value': len(layers) * args.steps * world / dt, 'unit': 'layers/s', 'n_gpus': world, 'steps': args.steps,
            'warmup': args.warmup, 'ms_per_step': dt / args.steps * 1e3,
            'higher_is_better': True, 'scaling': 'weak',
            'vs_baseline': None, 'dtype': 'f8e4m3 codes from ' + args.dtype, 'data': 'synthetic',
            'config': {'workload': 'FP8 e4m3 symmetric per-tensor RTN: 28 Linear weights of one Mixtral-8x7B block (4 attention + 8 experts '
                                   'x 3) -> absmax, scale, codes; static per-tensor ranges of their 18 inputs (128 x 512 calibration '
                                   'tokens, experts a quarter each), 1 block per step per GPU',
                       'n_seq': n_seq, 'seq_len': seq, 'parallelism': f'block-sharded x{world}' if world > 1 else 'single GPU'},
            'roofline': {'bound': 'hbm', 'achieved': wbytes / t_w / 1e9, 'peak': PEAK_HBM / 1e9, 'unit': 'GB/s',
                         'frac': wbytes / t_w / PEAK_HBM, 'traffic': None,
                         'kernel': 'k_minmax_partial + k_fp8_cast on a 14336 x 4096 weight (5 B per element: two 16-bit reads, one code '
                                   'written); the Python call, events on the launch stream',
                         'launches': reps, 'algorithmic_bytes_per_launch': wbytes, 'avg_launch_ms': t_w * 1e3,
                         'whole_step_gbps': (5.0 * elems + abytes) * args.steps / dt / 1e9},
            'cpu_baseline': None,
        }), flush=True)
    if world > 1:
        torch.distributed.destroy_process_group()


