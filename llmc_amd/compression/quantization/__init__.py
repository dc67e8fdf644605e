from .quant import (BaseQuantizer, FloatQuantizer, IntegerQuantizer,  # noqa: F401
                    pack_awq_gemm, pack_lsb)
from .module_utils import (AutoawqRealQuantLinear, EffcientFakeQuantLinear,  # noqa: F401
                           FakeQuantLinear, LlmcFp8Linear, OriginFloatLinear, VllmRealQuantLinear)
from .base_blockwise_quantization import BaseBlockwiseQuantization  # noqa: F401
from .auto_clip import AutoClipper  # noqa: F401
from .rtn import RTN  # noqa: F401
from .gptq import GPTQ  # noqa: F401
from .awq import Awq  # noqa: F401
from .spqr import SpQR  # noqa: F401
