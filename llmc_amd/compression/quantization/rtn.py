"""RTN (llmc/compression/quantization/rtn.py:9-28): nothing happens in the block loop unless static
activation quantization needs calibration; all weight arithmetic runs at deploy() through the quantizer."""
import torch

from llmc_amd.utils.registry_factory import ALGO_REGISTRY

from .base_blockwise_quantization import BaseBlockwiseQuantization


@ALGO_REGISTRY
class RTN(BaseBlockwiseQuantization):
    def __init__(self, model, quant_config, input, padding_mask, config):
        super().__init__(model, quant_config, input, padding_mask, config)

    @torch.no_grad()
    def block_opt(self, block, *opt_kwargs):
        if self.act_static:
            super().block_opt(block, *opt_kwargs)

    @torch.no_grad()
    def subset_transform(self, subset, input_feat, subset_kwargs):
        pass
