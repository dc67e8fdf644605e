"""§8(f) rank 2 — on-disk formats: deploy -> save_model -> exporter metadata, then read the checkpoint back like a
runtime would (safetensors + config.json) and decode it with an independent restatement of each format:
  compressed-tensors pack-quantized (vLLM): int32 words, 8 LSB-first nibbles of (code + 8), fp16 scales [R, K/g];
  AutoAWQ GEMM: qweight [K, R/8] with nibble order 0,2,4,6,1,3,5,7, qzeros [K/g, R/8], scales [K/g, R] fp16.
Decoded weights must equal the fake-quantized weights the same algorithm deploys (bit-exact in fp16 arithmetic)."""
import copy
import json
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


class Cfg(dict):
    __getattr__ = dict.get


def toy():
    from toy_model import ToyModel, calib_input
    model = ToyModel()
    model.skip_layer_name = lambda: ['lm_head']
    return model, calib_input(model)


def unpack_ct_int4(words, K):
    """compressed-tensors `unpack_from_int32`: nibble i of word j is column 8 j + i, stored offset by 8."""
    w = words.astype(np.uint32)
    cols = [((w >> (4 * i)) & 0xF).astype(np.int32) - 8 for i in range(8)]
    return np.stack(cols, axis=-1).reshape(words.shape[0], -1)[:, :K]


def unpack_awq(qweight, order=(0, 2, 4, 6, 1, 3, 5, 7)):
    """AutoAWQ GEMM: word c of a row holds output columns 8c + order[i] in nibble i."""
    q = qweight.astype(np.uint32)
    out = np.zeros((q.shape[0], q.shape[1] * 8), dtype=np.int32)
    for i, o in enumerate(order):
        out[:, o::8] = (q >> (4 * i)) & 0xF
    return out


def test_vllm_pack_quantized_checkpoint_round_trip(tmp_path):
    import llmc_amd.compression.quantization as Q
    from llmc_amd.utils import update_vllm_quant_config
    from safetensors.torch import load_file
    model, inp = toy()
    wcfg = Cfg(bit=4, symmetric=True, granularity='per_group', group_size=128, need_pack=True)
    config = Cfg(calib=Cfg(seq_len=64), model=Cfg(type='Toy'), quant=Cfg(weight=wcfg), save=Cfg(save_vllm=True))
    algo = Q.RTN(model, Cfg(weight=wcfg), inp, None, config)
    algo.run_block_loop()
    ref = copy.deepcopy(model)
    algo.deploy('vllm_quant')
    algo.save_model(str(tmp_path))
    cfg = update_vllm_quant_config(model, config, str(tmp_path))
    on_disk = json.load(open(tmp_path / 'config.json'))
    cc = on_disk['compression_config']
    assert cc == cfg['compression_config'] and 'quantization_config' not in on_disk
    assert cc['format'] == 'pack-quantized' and cc['quant_method'] == 'compressed-tensors' and cc['ignore'] == ['lm_head']
    g0 = cc['config_groups']['group_0']
    assert g0['targets'] == ['Linear'] and g0['input_activations'] is None
    assert g0['weights'] == {'num_bits': 4, 'type': 'int', 'symmetric': True, 'observer': 'minmax', 'observer_kwargs': {},
                             'dynamic': False, 'group_size': 128, 'strategy': 'group'}
    sd = load_file(str(tmp_path / 'model.safetensors'))
    q = Q.IntegerQuantizer(4, True, 'per_group', group_size=128)
    for bi, blk in enumerate(ref.get_blocks()):
        for name, lin in ref.get_block_linears(blk).items():
            p = f'blocks.{bi}.{name}.'
            packed, scale = sd[p + 'weight_packed'], sd[p + 'weight_scale']
            R, K = lin.weight.shape
            assert packed.dtype == torch.int32 and packed.shape == (R, K // 8)
            assert scale.dtype == torch.float16 and scale.shape == (R, K // 128)
            assert p + 'weight' not in sd
            codes = unpack_ct_int4(packed.numpy(), K)
            assert codes.min() >= -8 and codes.max() <= 7
            # what the runtime computes: code * scale per group; equals the deployed fake quantization (scales in fp16)
            deq = codes.reshape(R, K // 128, 128).astype(np.float32) * scale.float().numpy()[:, :, None]
            codes_ref, s_ref, _ = q.real_quant_weight_dynamic(lin.weight.data.cuda())
            np.testing.assert_array_equal(codes, codes_ref.cpu().numpy())
            np.testing.assert_array_equal(scale.numpy(), s_ref.to(torch.float16).cpu().numpy())
            fq = q.fake_quant_weight_dynamic(lin.weight.data.cuda()).float().cpu().numpy()
            assert np.abs(deq.reshape(R, K) - fq).max() <= 2 ** -7 * np.abs(fq).max()      # bf16 model-dtype rounding only


def test_autoawq_checkpoint_round_trip(tmp_path):
    import llmc_amd.compression.quantization as Q
    from llmc_amd.utils import update_autoawq_quant_config
    from safetensors.torch import load_file
    from toy_model import ToyModel, calib_input
    model = ToyModel(dtype=torch.float16)
    inp = calib_input(model)
    wcfg = Cfg(bit=4, symmetric=False, granularity='per_group', group_size=128, pack_version='gemm_pack')
    config = Cfg(calib=Cfg(seq_len=64), model=Cfg(type='Toy'), quant=Cfg(weight=wcfg), save=Cfg(save_autoawq=True))
    algo = Q.RTN(model, Cfg(weight=wcfg), inp, None, config)
    algo.run_block_loop()
    ref = copy.deepcopy(model)
    algo.deploy('autoawq_quant')
    algo.save_model(str(tmp_path))
    update_autoawq_quant_config(config, str(tmp_path))
    qc = json.load(open(tmp_path / 'config.json'))['quantization_config']
    assert qc == {'bits': 4, 'group_size': 128, 'modules_to_not_convert': None, 'quant_method': 'awq', 'version': 'gemm',
                  'zero_point': True}
    sd = load_file(str(tmp_path / 'model.safetensors'))
    q = Q.IntegerQuantizer(4, False, 'per_group', group_size=128)
    blk = ref.get_blocks()[1]
    for name, lin in ref.get_block_linears(blk).items():
        p = f'blocks.1.{name}.'
        R, K = lin.weight.shape
        qw, qz, sc = sd[p + 'qweight'], sd[p + 'qzeros'], sd[p + 'scales']
        assert qw.shape == (K, R // 8) and qz.shape == (K // 128, R // 8) and sc.shape == (K // 128, R)
        assert qw.dtype == qz.dtype == torch.int32 and sc.dtype == torch.float16
        codes = unpack_awq(qw.numpy())           # [K, R]
        zeros = unpack_awq(qz.numpy())           # [K/g, R]
        # AutoAWQ's dequantisation: (code - zero) * scale per (group of 128 input channels, output column)
        deq = (codes.reshape(K // 128, 128, R) - zeros[:, None, :]).astype(np.float32) * sc.float().numpy()[:, None, :]
        fq = q.fake_quant_weight_dynamic(lin.weight.data.cuda()).float().cpu().numpy()     # [R, K]
        got = deq.reshape(K, R).T
        # gemm_pack re-derives the codes as round((w + z*s)/s) in fp16 WITHOUT a clamp (module_utils.py:1018-1030, as
        # AutoAWQ's own packer does): fp16 rounding of w + z*s moves ~0.3 % of the codes by one step against
        # clamp(round(w/s) + z), and a group maximum can land on 16 and spill into the neighbouring nibble. So: the
        # stored codes equal an independent fp16 restatement of that formula wherever it stays inside [0, 15] ...
        _, s16, z16 = q.real_quant_weight_dynamic(lin.weight.data.cuda())
        s16, z16 = s16.to(torch.float16).cpu(), z16.cpu()
        w16 = lin.weight.data.cpu().to(torch.float16)
        sz = (s16 * z16.to(torch.float16))                                           # scale_zeros, fp16
        want = torch.round((w16.reshape(R, K // 128, 128) + sz[:, :, None]) / s16[:, :, None]).reshape(R, K).int()
        inside = (want >= 0) & (want <= 15)
        got_codes = torch.from_numpy(codes.T.copy())                                   # [R, K]
        spill = (~inside).T.reshape(K, R // 8, 8).any(-1).repeat_interleave(8, 1).T   # words (8 rows of R) hit by an overflow
        assert torch.equal(got_codes[~spill], want[~spill]) and float(spill.float().mean()) < 0.02
        np.testing.assert_array_equal(zeros, z16.numpy().astype(np.int32).T)
        np.testing.assert_array_equal(sc.numpy(), s16.numpy().T)
        # ... and the decoded weight is the deployed fake quantisation up to those one-step differences
        ok = np.abs(got - fq) <= 2 ** -10 * np.abs(fq).max() + 1e-7
        assert ok.mean() > 0.99, ok.mean()
