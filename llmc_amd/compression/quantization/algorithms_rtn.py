"""Round-to-nearest: the algorithm whose block loop does nothing.

llmc's RTN (llmc/compression/quantization/rtn.py:9-28) keeps the calibration loop only to collect static
activation ranges; the weight arithmetic happens when `deploy()` swaps the Linear layers for their fake- or
real-quant wrappers, which call `w_qdq` / `w_q` of the base class and therefore the quantizer kernels
(llmc_quant_dynamic, llmc_pack_lsb, llmc_fp8_quant)."""
import torch

from llmc_amd.utils.registry_factory import ALGO_REGISTRY

from .base_blockwise_quantization import BaseBlockwiseQuantization


class RTN(BaseBlockwiseQuantization):
    needs_calibration_pass = False

    def __init__(self, model, quant_config, input, padding_mask, config):
        BaseBlockwiseQuantization.__init__(self, model, quant_config, input, padding_mask, config)

    @torch.no_grad()
    def block_opt(self, block, *opt_kwargs):
        # static activation quantization is the one case that needs the hooks and a forward pass
        if not self.act_static:
            return None
        return BaseBlockwiseQuantization.block_opt(self, block, *opt_kwargs)

    @torch.no_grad()
    def subset_transform(self, subset, input_feat, subset_kwargs):
        return None


RTN = ALGO_REGISTRY(RTN)   # decorator protocol only: llmc's own Register has no other registration method
