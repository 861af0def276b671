"""SparseConvolution family: SubMConv3d / SparseConv3d with spconv-2 parameter names and weight
layout [Cout, kz, ky, kx, Cin], so reference checkpoints load key-for-key
(reference detector3d_template.py:330-359).  Arithmetic: toda_amd.ops -> libtoda_hip.so."""
import math

import torch
import torch.nn as nn

from .. import ops
from .core import SparseConvTensor
from .modules import SparseModule


def _triple(v):
    return [int(x) for x in v] if isinstance(v, (list, tuple)) else [int(v)] * 3


class SparseConvolution(SparseModule):
    def __init__(self, ndim, in_channels, out_channels, kernel_size=3, stride=1, padding=0, dilation=1, groups=1,
                 bias=True, subm=False, output_padding=0, transposed=False, inverse=False, indice_key=None,
                 algo=None, fp32_accum=None, name=None):
        super().__init__()
        if ndim != 3:
            raise NotImplementedError("only 3-D sparse convolution is on this path")
        if groups != 1 or transposed:
            raise NotImplementedError("groups != 1 / transposed sparse conv are not on this path")
        self.ndim = ndim
        self.in_channels, self.out_channels = in_channels, out_channels
        self.kernel_size = _triple(kernel_size)
        self.stride, self.padding, self.dilation = _triple(stride), _triple(padding), _triple(dilation)
        self.subm, self.inverse, self.transposed = subm, inverse, transposed
        self.indice_key = indice_key
        self.conv1x1 = all(k == 1 for k in self.kernel_size)
        self.weight = nn.Parameter(torch.empty(out_channels, *self.kernel_size, in_channels))
        if bias:
            self.bias = nn.Parameter(torch.empty(out_channels))
        else:
            self.register_parameter("bias", None)
        self.reset_parameters()
        self._packed = None  # ((version, data_ptr), packed forward weights)
        self._packed_dgrad = None  # ((version, data_ptr), packed data-gradient operand) when prepack() made it

    def reset_parameters(self):
        # same recipe as torch's _ConvNd: kaiming_uniform(a=sqrt(5)) over fan_in = K * Cin
        fan_in = self.in_channels * self.kernel_size[0] * self.kernel_size[1] * self.kernel_size[2]
        bound = math.sqrt(6.0 / ((1 + 5.0) * fan_in))
        with torch.no_grad():
            self.weight.uniform_(-bound, bound)
            if self.bias is not None:
                b = 1.0 / math.sqrt(fan_in)
                self.bias.uniform_(-b, b)

    def extra_repr(self):
        s = f"{self.in_channels}, {self.out_channels}, kernel_size={self.kernel_size}, stride={self.stride}"
        s += f", padding={self.padding}, subm={self.subm}, indice_key={self.indice_key}"
        return s + (", bias=False" if self.bias is None else "")

    def _packed_forward_weight(self):
        w = self.weight
        tag = (w._version, w.data_ptr(), ops.matrix_path())
        if self._packed is None or self._packed[0] != tag:
            with torch.no_grad():
                self._packed = (tag, ops.pack_weight(w.detach(), False, False))
        return self._packed[1]

    def _dgrad_operand(self):
        """The dgrad operand packed together with the forward one by prepack(), if it matches the current weights."""
        w = self.weight
        if self._packed_dgrad is not None and self._packed_dgrad[0] == (w._version, w.data_ptr(), ops.matrix_path()):
            return self._packed_dgrad[1]
        return None

    def _rulebook(self, x):
        """Find or build the rulebook; returns (rulebook, out_indices, out_shape, out_grid_index)."""
        cached = x.find_indice_pair(self.indice_key)
        if self.subm:
            if cached is not None and cached["n_in"] == x.indices.shape[0] and cached["kind"] == "subm" \
                    and cached["rb"].ksize == self.kernel_size:
                return cached["rb"], x.indices, x.spatial_shape, x.grid_index
            rb, gi = ops.build_subm_rulebook(x.indices, x.batch_size, x.spatial_shape, self.kernel_size,
                                             self.dilation, grid_index=x.grid_index)
            x.grid_index = gi
            if self.indice_key is not None:
                x.indice_dict[self.indice_key] = {"kind": "subm", "rb": rb, "n_in": x.indices.shape[0]}
            return rb, x.indices, x.spatial_shape, gi
        if cached is not None and cached["kind"] == "conv" and cached["n_in"] == x.indices.shape[0]:
            return cached["rb"], cached["out_indices"], cached["out_shape"], cached["gi"]
        if any(d != 1 for d in self.dilation):
            raise NotImplementedError("dilated strided sparse conv is not on this path")
        out_idx, out_shape, rb, gi = ops.build_conv_rulebook(x.indices, x.batch_size, x.spatial_shape,
                                                            self.kernel_size, self.stride, self.padding)
        if self.indice_key is not None:
            x.indice_dict[self.indice_key] = {"kind": "conv", "rb": rb, "n_in": x.indices.shape[0],
                                              "out_indices": out_idx, "out_shape": out_shape, "gi": gi}
        return rb, out_idx, out_shape, gi

    def _inverse_rulebook(self, x):
        """SparseInverseConv3d: the pairs of the strided convolution that wrote `indice_key`, read from its output set back to its input
        set (Rulebook.inverse); the output tensor sits on that convolution's input sites."""
        cached = x.find_indice_pair(self.indice_key)
        if cached is None or cached["kind"] != "conv":
            raise ValueError(f"SparseInverseConv3d: no strided convolution has written indice_key {self.indice_key!r}")
        rb = cached["rb"]
        if x.indices.shape[0] != rb.n_out:
            raise ValueError(f"SparseInverseConv3d: the input has {x.indices.shape[0]} sites, the convolution of {self.indice_key!r} produced {rb.n_out}")
        if rb.ksize != self.kernel_size:
            raise ValueError(f"SparseInverseConv3d: kernel size {self.kernel_size} differs from the convolution's {rb.ksize}")
        return rb.inverse(), rb.in_indices, rb.geom["in_shape"], None

    def forward(self, x, want_bn_stats=False):
        """want_bn_stats (used by SparseSequential / SparseBasicBlock when a training-mode BatchNorm1d follows): the output
        tensor carries `.bn_sums`, the moments of its features taken in the convolution's epilogue (None when unavailable)."""
        if not isinstance(x, SparseConvTensor):
            raise TypeError("sparse convolution expects a SparseConvTensor")
        if x.features.shape[1] != self.in_channels:
            raise ValueError(f"expected {self.in_channels} input channels, got {x.features.shape[1]}")
        if self.inverse:
            rb, out_idx, out_shape, gi = self._inverse_rulebook(x)
        else:
            rb, out_idx, out_shape, gi = self._rulebook(x)
        packed = self._packed_forward_weight()
        sums = None
        if want_bn_stats:
            dg = self._dgrad_operand()
            extra = {} if dg is None else {"packed_dgrad": dg}
            feats, sums = ops.sparse_conv(x.features, self.weight, self.bias, rb, packed, want_stats=True, **extra)
        else:
            dg = self._dgrad_operand()
            extra = {} if dg is None else {"packed_dgrad": dg}
            feats = ops.sparse_conv(x.features, self.weight, self.bias, rb, packed, **extra)
        out = SparseConvTensor(feats, out_idx, out_shape, x.batch_size, indice_dict=x.indice_dict)
        out.grid_index = gi
        out.bn_sums = sums
        return out


class SubMConv3d(SparseConvolution):
    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1, groups=1, bias=True,
                 indice_key=None, algo=None, fp32_accum=None, name=None):
        # stride/padding are accepted and ignored like spconv does for submanifold convs
        super().__init__(3, in_channels, out_channels, kernel_size, 1, padding, dilation, groups, bias, subm=True,
                         indice_key=indice_key)


class SparseConv3d(SparseConvolution):
    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1, groups=1, bias=True,
                 indice_key=None, algo=None, fp32_accum=None, name=None):
        super().__init__(3, in_channels, out_channels, kernel_size, stride, padding, dilation, groups, bias,
                         subm=False, indice_key=indice_key)


class SparseInverseConv3d(SparseConvolution):
    def __init__(self, in_channels, out_channels, kernel_size, indice_key=None, bias=True, **kw):
        super().__init__(3, in_channels, out_channels, kernel_size, bias=bias, inverse=True, indice_key=indice_key)


def prepack(module):
    """Pack the forward and (when gradients will be wanted) the data-gradient operands of EVERY sparse convolution under
    `module` whose weights changed since the last pack in one kernel launch.  Called once per forward by the backbones; a
    convolution that is not covered (first use, weights touched later) packs for itself as before."""
    convs = module.__dict__.get("_sparse_convs")
    if convs is None:
        convs = [m for m in module.modules() if isinstance(m, SparseConvolution)]
        module.__dict__["_sparse_convs"] = convs
    if not convs or not convs[0].weight.is_cuda:
        return
    want_dgrad = torch.is_grad_enabled()
    stale = []
    mm = ops.matrix_path()
    for c in convs:
        tag = (c.weight._version, c.weight.data_ptr(), mm)
        if c._packed is None or c._packed[0] != tag or (want_dgrad and c.weight.requires_grad and (c._packed_dgrad is None or c._packed_dgrad[0] != tag)):
            stale.append((c, tag))
    if len(stale) < 2:
        return
    with torch.no_grad():
        items = [(c.weight.detach(), False, False) for c, _ in stale]
        back = [(c, tag) for c, tag in stale if want_dgrad and c.weight.requires_grad]
        items += [(c.weight.detach(), True, c.subm) for c, _ in back]       # SubM dgrad reads the forward table with reversed offsets
        packed = ops.pack_weights_batched(items)
    for (c, tag), wp in zip(stale, packed[:len(stale)]):
        c._packed = (tag, wp)
    for (c, tag), wp in zip(back, packed[len(stale):]):
        c._packed_dgrad = (tag, wp)
