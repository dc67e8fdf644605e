from .algorithms_rtn import RTN  # noqa: F401  (module path kept for `from ...quantization.rtn import RTN`)
