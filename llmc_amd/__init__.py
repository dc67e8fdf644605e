"""llmc_amd — MI355X (gfx950) implementation of llmc's per-Linear weight-quantization hot path.

Python host classes mirror llmc's operator surface (GPTQ / Awq / RTN / IntegerQuantizer / ...);
all arithmetic runs in hand-written HIP kernels behind the C ABI declared in include/llmc_hip.h
(libllmc_hip.so, loaded with ctypes). There is no CPU fallback: without the library, or on
non-GPU tensors, the product path raises.
"""
__version__ = '0.1.0'


def register_into(registry, names=('GPTQ', 'Awq', 'RTN', 'SpQR')):
    """Bind this package's algorithm classes under llmc's keys in `registry` (llmc.utils.registry_factory.ALGO_REGISTRY,
    llmc/utils/registry_factory.py:9-23). Uses item assignment, which llmc's Register supports and which REPLACES an
    existing key, so it works before or after llmc has registered its own classes, in any import order; returns the
    bound classes. `llmc/__main__.py:43,62` then picks ours up through `ALGO_REGISTRY[config.quant.method]`."""
    from .compression import quantization as Q
    bound = {}
    for n in names:
        cls = getattr(Q, n)
        registry[n] = cls
        bound[n] = cls
    _announce_wrappers()
    return bound


def _announce_wrappers(ref_module_utils=None):
    """llmc's model adapters find "the Linear layers of a block" by type: `BaseModel.get_block_linears` /
    `replace_module_subset` test `isinstance(m, tuple(_LLMC_LINEAR_TYPES_ + _TRANSFORMERS_LINEAR_TYPES_))`
    (llmc/models/base_model.py:341-353, 424-431) against the lists of llmc/compression/quantization/module_utils.py:1118-1129.
    Our Linear wrappers (FakeQuantLinear, EffcientFakeQuantLinear, the real-quant ones) are other classes with the same
    names, so once a block holds them (`quant_out`, `true_sequential`, `deploy`) the reference's adapter would no longer see
    those layers. When llmc is imported in this process, append our wrapper classes to ITS lists (they are plain mutable
    lists that every importer shares). Executed by tests/test_ref_pipeline_gpu.py through llmc/__main__.py's own main()."""
    import sys
    mu = ref_module_utils or sys.modules.get('llmc.compression.quantization.module_utils')
    if mu is None:
        return False
    from .compression.quantization import module_utils as ours
    for lst in ('_LLMC_LINEAR_TYPES_', '_LLMC_LN_TYPES_'):
        theirs = getattr(mu, lst, None)
        if isinstance(theirs, list):
            for cls in getattr(ours, lst):
                if cls not in theirs:
                    theirs.append(cls)
    return True
