from .quant import (BaseQuantizer, FloatQuantizer, IntegerQuantizer,  # noqa: F401
                    pack_awq_gemm, pack_lsb)
