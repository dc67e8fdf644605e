"""Block-loop base class with the surface of llmc's BlockwiseOpt (llmc/compression/blockwise_optimization.py).

A transformer is compressed one block at a time: `run_block_loop()` visits the blocks in order and hands each
to `block_opt()`; subclasses capture the inputs of the block's Linear layers through forward hooks
(`cache_input_hook`) and rewrite the weights. Captured activations stay on the GPU here — 288 GB of HBM hold a
block's calibration activations many times over — where the reference parks every hooked input on the host and
brings it back for each use."""
import os
from abc import ABC, abstractmethod

import torch


def _strip_cache_kwargs(kwargs_list):
    """the first-block kwargs captured by the model adapter may carry a KV cache; it must not be reused"""
    for kw in kwargs_list:
        kw.pop('use_cache', None)
        for name in ('past_key_value', 'past_key_values'):      # the keyword was renamed between transformers releases
            if name in kw:
                kw[name] = None


class BlockwiseOpt(ABC):
    def __init__(self, model, compress_config, input, padding_mask, config):
        self.model, self.config = model, config
        self.quant_config = self.sparsity_config = compress_config
        self.input, self.padding_mask = input, padding_mask
        self.blocks = model.get_blocks()
        self.num_blocks = len(self.blocks)
        self.block_idx = None
        self.data_free = not self.input
        if not self.data_free:
            _strip_cache_kwargs(input['kwargs'])
            self.n_samples = int(sum(batch.shape[0] for batch in input['data']))

    # ---- driver -----------------------------------------------------------------------------------
    def run_block_loop(self):
        for idx, block in enumerate(self.blocks):
            self.block_idx = idx
            self.block_opt(block)
        self._dump_side_products()

    def _dump_side_products(self):
        """AWQ's searched scales / clip bounds, for llmc's two-stage pipelines (scales.pth, clips.pth)."""
        if getattr(self, 'save_scale', False):
            os.makedirs(self.scale_path, exist_ok=True)
            torch.save(self.act_scales, os.path.join(self.scale_path, 'scales.pth'))
        if getattr(self, 'save_clip', False):
            os.makedirs(self.clip_path, exist_ok=True)
            torch.save(self.auto_clipper.weight_clips, os.path.join(self.clip_path, 'clips.pth'))

    # ---- hooks --------------------------------------------------------------------------------------
    def cache_input_hook(self, m, x, y, name, feat_dict):
        captured = tuple(t.detach() for t in x)
        if len(captured) != 1:
            feat_dict[name].append(captured)
            return
        (inp,) = captured
        feat_dict[name].append(inp.unsqueeze(0) if inp.dim() == 2 else inp)

    @abstractmethod
    def block_opt(self, block):
        ...

    def layer_init(self, layer):
        return None

    def subset_init(self, subset):
        return None

    def block_init(self, block):
        return None
