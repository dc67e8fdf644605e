"""The exporters' JSON against the REFERENCE's own exporters (llmc/utils/export_vllm.py, export_autoawq.py), run on the
same configs when /root/reference is present (build container); schema-only checks otherwise. CPU."""
import importlib.util
import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = '/root/reference'


class Cfg(dict):
    """attribute + .get access like llmc's EasyDict config"""
    __getattr__ = dict.get

    def __contains__(self, k):
        return dict.__contains__(self, k)


def cfg(d):
    return Cfg({k: cfg(v) if isinstance(v, dict) else v for k, v in d.items()})


class Model:
    def skip_layer_name(self):
        return ['lm_head']


CASES = {
    'w4a16_pack': {'weight': {'bit': 4, 'symmetric': True, 'granularity': 'per_group', 'group_size': 128, 'need_pack': True}},
    'w8a8_static_tensor': {'weight': {'bit': 8, 'symmetric': True, 'granularity': 'per_channel'},
                           'act': {'bit': 8, 'symmetric': True, 'granularity': 'per_tensor', 'static': True}},
    'w8a8_dynamic_token': {'weight': {'bit': 8, 'symmetric': True, 'granularity': 'per_channel'},
                           'act': {'bit': 8, 'symmetric': True, 'granularity': 'per_token'}},
    'fp8_w_only': {'weight': {'bit': 'e4m3', 'symmetric': True, 'granularity': 'per_channel', 'quant_type': 'float-quant'}},
    'fp8_static': {'weight': {'bit': 'e4m3', 'symmetric': True, 'granularity': 'per_tensor', 'quant_type': 'float-quant'},
                   'act': {'bit': 'e4m3', 'symmetric': True, 'granularity': 'per_tensor', 'quant_type': 'float-quant',
                           'static': True}},
    'fp8_block_dynamic': {'weight': {'bit': 'e4m3', 'symmetric': True, 'granularity': 'per_block', 'block_size': 128,
                                     'quant_type': 'float-quant'},
                          'act': {'bit': 'e4m3', 'symmetric': True, 'granularity': 'per_group', 'group_size': 128,
                                  'quant_type': 'float-quant'}},
}


def _load_ref(name):
    spec = importlib.util.spec_from_file_location('ref_' + name, os.path.join(REF, 'llmc', 'utils', name + '.py'))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def _run(fn, tmp, *args):
    os.makedirs(tmp, exist_ok=True)
    with open(os.path.join(tmp, 'config.json'), 'w') as f:
        json.dump({'architectures': ['LlamaForCausalLM'], 'quantization_config': {'stale': True}}, f)
    fn(*args)
    return json.load(open(os.path.join(tmp, 'config.json')))


@pytest.mark.parametrize('case', sorted(CASES))
def test_vllm_config_matches_reference(case, tmp_path):
    sys.path.insert(0, ROOT)
    from llmc_amd.utils import update_vllm_quant_config
    config = cfg({'quant': CASES[case]})
    ours = _run(update_vllm_quant_config, str(tmp_path / 'a'), Model(), config, str(tmp_path / 'a'))
    assert ours['architectures'] == ['LlamaForCausalLM']
    if case.startswith('fp8_static') or case == 'fp8_block_dynamic':
        assert ours['quantization_config']['quant_method'] == 'fp8'
    else:
        assert ours['compression_config']['quant_method'] == 'compressed-tensors'
    if not os.path.isdir(REF):
        pytest.skip('reference tree not present: schema-only checks done')
    ref = _run(_load_ref('export_vllm').update_vllm_quant_config, str(tmp_path / 'b'), Model(), config, str(tmp_path / 'b'))
    assert ours == ref


def test_autoawq_config_matches_reference(tmp_path):
    sys.path.insert(0, ROOT)
    from llmc_amd.utils import update_autoawq_quant_config
    for w in ({'bit': 4, 'symmetric': False, 'granularity': 'per_group', 'group_size': 128, 'pack_version': 'gemm_pack'},
              {'bit': 4, 'symmetric': False, 'granularity': 'per_channel', 'pack_version': 'gemv_pack'}):
        config = cfg({'quant': {'weight': w}})
        ours = _run(update_autoawq_quant_config, str(tmp_path / 'a'), config, str(tmp_path / 'a'))
        assert ours['quantization_config']['quant_method'] == 'awq' and 'stale' not in ours['quantization_config']
        if os.path.isdir(REF):
            ref = _run(_load_ref('export_autoawq').update_autoawq_quant_config, str(tmp_path / 'b'), config, str(tmp_path / 'b'))
            assert ours == ref
