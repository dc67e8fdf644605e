from .export_autoawq import update_autoawq_quant_config  # noqa: F401,E402
from .export_vllm import update_vllm_quant_config  # noqa: F401,E402
