"""llmc_amd — MI355X (gfx950) implementation of llmc's per-Linear weight-quantization hot path.

Python host classes mirror llmc's operator surface (GPTQ / Awq / RTN / IntegerQuantizer / ...);
all arithmetic runs in hand-written HIP kernels behind the C ABI declared in include/llmc_hip.h
(libllmc_hip.so, loaded with ctypes). There is no CPU fallback: without the library, or on
non-GPU tensors, the product path raises.
"""
__version__ = '0.1.0'
