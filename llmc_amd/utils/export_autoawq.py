"""`quantization_config` of an AutoAWQ (GEMM layout) checkpoint (llmc/utils/export_autoawq.py:4-30; called from
llmc/__main__.py:146-148 after `deploy('autoawq_quant')` + `save_model`). Tensors: qweight int32 [K, R/8],
qzeros int32 [K/g, R/8], scales f16 [K/g, R] (AutoawqRealQuantLinear, nibble order 0,2,4,6,1,3,5,7)."""
from .export_vllm import _get, _rewrite_config


def update_autoawq_quant_config(config, save_quant_path):
    w = _get(_get(config, 'quant'), 'weight')
    block = {
        'bits': _get(w, 'bit'),
        'group_size': _get(w, 'group_size') if _get(w, 'granularity') == 'per_group' else -1,
        'modules_to_not_convert': None,
        'quant_method': 'awq',
        'version': str(_get(w, 'pack_version')).split('_')[0],      # 'gemm_pack' -> 'gemm'
        'zero_point': not _get(w, 'symmetric'),
    }

    def mutate(c):
        c.pop('quantization_config', None)
        c['quantization_config'] = block
    return _rewrite_config(save_quant_path, mutate)
