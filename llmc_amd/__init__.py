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
    return bound
