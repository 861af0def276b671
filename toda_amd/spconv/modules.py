"""SparseModule / SparseSequential (spconv.SparseSequential as used by post_act_block,
reference spconv_backbone.py:8-27): dense nn.Modules inside the sequence act on .features."""
from collections import OrderedDict

import torch.nn as nn

from .. import ops
from .core import SparseConvTensor


class SparseModule(nn.Module):
    """Marker base class: modules that consume and return a SparseConvTensor."""


def is_spconv_module(module):
    return isinstance(module, SparseModule)


def _sparse_convolution():
    from .conv import SparseConvolution
    return SparseConvolution


class SparseSequential(SparseModule):
    def __init__(self, *args, **kwargs):
        super().__init__()
        if len(args) == 1 and isinstance(args[0], OrderedDict):
            for name, module in args[0].items():
                self.add_module(name, module)
        else:
            for i, module in enumerate(args):
                self.add_module(str(i), module)
        for name, module in kwargs.items():
            if name in self._modules:
                raise ValueError(f"duplicate module name {name}")
            self.add_module(name, module)

    def __getitem__(self, i):
        if not -len(self) <= i < len(self):
            raise IndexError(i)
        return list(self._modules.values())[i % len(self)]

    def __len__(self):
        return len(self._modules)

    def add(self, module, name=None):
        self.add_module(name if name is not None else str(len(self._modules)), module)

    def forward(self, x):
        mods = list(self._modules.values())
        i = 0
        while i < len(mods):
            module = mods[i]
            if is_spconv_module(module):
                nxt = mods[i + 1] if i + 1 < len(mods) else None
                fuse = (isinstance(module, _sparse_convolution()) and type(nxt) is nn.BatchNorm1d and nxt.training and isinstance(x, SparseConvTensor)
                        and x.features.is_cuda)
                x = module(x, want_bn_stats=True) if fuse else module(x)
            elif isinstance(x, SparseConvTensor):
                if x.features.shape[0] != 0:
                    # BatchNorm1d (+ ReLU) over sparse rows: one fused HIP path instead of 2 modules
                    if isinstance(module, nn.BatchNorm1d) and ops.bn_rows_supported(x.features, module):
                        relu = i + 1 < len(mods) and type(mods[i + 1]) is nn.ReLU
                        prev = mods[i - 1] if i > 0 else None
                        x = x.replace_feature(ops.bn_rows(x.features, module, relu, sums=getattr(x, "bn_sums", None),
                                                          colsum=getattr(prev, "bias", None) is not None and is_spconv_module(prev)))
                        i += 2 if relu else 1
                        continue
                    x = x.replace_feature(module(x.features))
            else:
                x = module(x)
            i += 1
        return x
