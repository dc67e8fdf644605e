from .quant import BaseQuantizer, IntegerQuantizer, pack_lsb  # noqa: F401
