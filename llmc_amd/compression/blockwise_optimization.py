"""Block loop base with llmc's BlockwiseOpt surface (llmc/compression/blockwise_optimization.py:8-114).

Difference in mechanics: captured Linear inputs stay on the GPU (the reference moves every hooked input to the
CPU, blockwise_optimization.py:53-61, and back for each use) — 288 GB of HBM hold a block's calibration
activations many times over."""
import os
from abc import ABCMeta, abstractmethod

import torch


class BlockwiseOpt(metaclass=ABCMeta):
    def __init__(self, model, compress_config, input, padding_mask, config):
        self.model = model
        self.blocks = model.get_blocks()
        self.quant_config = compress_config
        self.sparsity_config = compress_config
        self.input = input
        self.padding_mask = padding_mask
        self.data_free = False if self.input else True
        self.config = config
        self.block_idx = None
        self.num_blocks = len(self.blocks)
        if self.input:
            for kw in input['kwargs']:
                kw.pop('use_cache', None)
                if 'past_key_value' in kw:
                    kw['past_key_value'] = None
            self.n_samples = sum(d.shape[0] for d in input['data'])

    def run_block_loop(self):
        for i in range(len(self.blocks)):
            self.block_idx = i
            self.block_opt(self.blocks[i])
        if getattr(self, 'save_scale', False):
            os.makedirs(self.scale_path, exist_ok=True)
            torch.save(self.act_scales, os.path.join(self.scale_path, 'scales.pth'))
        if getattr(self, 'save_clip', False):
            os.makedirs(self.clip_path, exist_ok=True)
            torch.save(self.auto_clipper.weight_clips, os.path.join(self.clip_path, 'clips.pth'))

    def cache_input_hook(self, m, x, y, name, feat_dict):
        inputs = [i.detach() for i in x]
        if len(inputs) == 1:
            inp = inputs[0]
            if inp.dim() == 2:
                inp = inp.unsqueeze(0)
            feat_dict[name].append(inp)
        else:
            feat_dict[name].append(tuple(inputs))

    @abstractmethod
    def block_opt(self, block):
        pass

    def layer_init(self, layer):
        pass

    def subset_init(self, subset):
        pass

    def block_init(self, block):
        pass
