"""The reference-named method surface of the mirrored classes (SURVEY.md §8b "signatures that must not change").

CPU: every public method the reference's class defines (read from oracle/_ref with `ast`, never imported) exists on
ours under the same name with the same positional argument names; what is deliberately absent is listed with the reason.
The arithmetic behind those names is checked on the GPU in tests/test_reference_methods_gpu.py."""
import ast
import inspect
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.path.join(ROOT, 'oracle', '_ref', 'llmc', 'compression', 'quantization')

# names the reference defines that are NOT on the path, with the reason
OUT_OF_SCOPE = {
    'get_hqq_qparams': 'hqq range search', 'optimize_weights_proximal': 'hqq range search',
    'get_float_qparams': 'FloatQuantizer without use_qtorch (simulated float path)',
    '_upscale_histogram': 'histogram observer internals live in hist_range.HistRange',
    '_combine_histograms': 'histogram observer internals live in hist_range.HistRange',
    'get_hist_threshold': 'histogram observer internals live in hist_range.HistRange',
    'get_norm': 'histogram observer internals live in hist_range.HistRange',
    'get_quantization_error': 'histogram observer internals live in hist_range.HistRange',
}
# base-class members that belong to subsystems §8 marks out of scope (rotations, shifts, KV cache, non-linear quant, tokenizer copy)
BASE_OUT_OF_SCOPE = {'apply_shift', 'bake_mean_into_fc', 'collect_layers_weights', 'contiguous_params', 'copy_tokenizer',
                     'fuse_ln_fcs', 'register_kv_cache', 'register_non_linear_qparams', 'remove_mean_from_embed',
                     'replace_act_fn', 'replace_attention', 'replace_rotate_linears', 'rotate_embeddings', 'rotate_head',
                     'rotate_post_layers', 'rotate_pre_layers', 'set_non_linear_mode', 'shift_fc_fc', 'shift_ln_fcs'}
# positional-argument names that differ on purpose: (class, method) -> reason
SIG_EXCEPTIONS = {
    ('GPTQ', '__init__'): 'ours takes the optional `modality` of newer llmc versions too',
    ('SpQR', '__init__'): 'same',
    ('SpQR', 'subset_transform'): "the reference's SpQR predates the base class's (subset, input_feat, subset_kwargs) protocol; ours follows the base class so the shared block loop drives it",
    ('SpQR', 'block_transform'): 'same (reference: *block_kwargs)',
    ('SpQR', 'w_q'): 'the reference spells the unused stub (weight, qargs); ours keeps the base protocol (module, wquantizer)',
    ('AutoClipper', '__init__'): 'keyword tail differs; the leading arguments are checked in test_host_classes.py',
}


def _ref_class_methods(fname, cls):
    tree = ast.parse(open(os.path.join(REF, fname)).read())
    classes = {n.name: n for n in tree.body if isinstance(n, ast.ClassDef)}
    out = {}

    def walk(c):
        node = classes.get(c)
        if node is None:
            return
        for b in node.bases:                       # base first, so that the subclass's definition wins
            if isinstance(b, ast.Name):
                walk(b.id)
        for m in node.body:
            if isinstance(m, ast.FunctionDef):
                out[m.name] = [a.arg for a in m.args.posonlyargs + m.args.args]
    walk(cls)
    return out


CASES = [('gptq.py', 'GPTQ'), ('spqr.py', 'SpQR'), ('awq.py', 'Awq'), ('auto_clip.py', 'AutoClipper'),
         ('quant.py', 'IntegerQuantizer'), ('quant.py', 'FloatQuantizer'), ('module_utils.py', 'FakeQuantLinear'),
         ('module_utils.py', 'EffcientFakeQuantLinear'), ('module_utils.py', 'VllmRealQuantLinear'),
         ('module_utils.py', 'AutoawqRealQuantLinear'), ('rtn.py', 'RTN')]


@pytest.mark.skipif(not os.path.isdir(REF), reason='oracle/_ref is built by __graft_entry__.build() where /root/reference exists')
@pytest.mark.parametrize('fname,cls', CASES)
def test_every_reference_method_exists_with_the_same_argument_names(fname, cls):
    import llmc_amd.compression.quantization as Q
    from llmc_amd.compression.quantization import auto_clip
    ours = getattr(Q, cls, None) or getattr(auto_clip, cls)
    ref = _ref_class_methods(fname, cls)
    missing, sig_diff = [], []
    for name, args in ref.items():
        if name in OUT_OF_SCOPE:
            continue
        if name.startswith('__') and name not in ('__init__',):
            continue
        f = inspect.getattr_static(ours, name, None)
        if f is None:
            missing.append(name)
            continue
        if (cls, name) in SIG_EXCEPTIONS:
            continue
        f = getattr(ours, name)
        try:
            mine = [p.name for p in inspect.signature(f).parameters.values()
                    if p.kind in (p.POSITIONAL_ONLY, p.POSITIONAL_OR_KEYWORD)]
        except (TypeError, ValueError):
            continue
        want = list(args)
        if want and want[0] in ('self', 'cls') and (not mine or mine[0] not in ('self', 'cls')):
            want = want[1:]                       # bound classmethod / staticmethod
        if mine[:len(want)] != want:
            sig_diff.append((name, want, mine))
    assert not missing, f'{cls}: reference methods without a counterpart: {missing}'
    assert not sig_diff, f'{cls}: positional arguments differ from the reference: {sig_diff}'


@pytest.mark.skipif(not os.path.isdir(REF), reason='oracle/_ref is built by __graft_entry__.build() where /root/reference exists')
def test_base_class_surface_minus_the_out_of_scope_subsystems():
    import llmc_amd.compression.quantization as Q
    ref = _ref_class_methods('base_blockwise_quantization.py', 'BaseBlockwiseQuantization')
    missing = [n for n in ref if not n.startswith('__') and n not in BASE_OUT_OF_SCOPE
               and inspect.getattr_static(Q.BaseBlockwiseQuantization, n, None) is None]
    assert not missing, missing


def test_a_subclass_overriding_a_reference_hook_is_routed_through_the_per_layer_flow():
    """GPTQ.subset_transform runs a whole subset through one stacked column loop — unless a class outside the package
    overrides one of the reference-named per-layer methods: then the reference's per-layer flow runs and honours it."""
    import llmc_amd.compression.quantization as Q

    class Mine(Q.GPTQ):
        def weight_transform(self, W, Hinv, Losses, tmp):      # noqa: D401
            return super().weight_transform(W, Hinv, Losses, tmp)

    a = object.__new__(Q.GPTQ)
    b = object.__new__(Mine)
    assert not a._overrides_reference_hooks()
    assert b._overrides_reference_hooks()
    s = object.__new__(Q.SpQR)
    assert not s._overrides_reference_hooks()
