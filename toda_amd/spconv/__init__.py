"""spconv-compatible operator namespace backed by libtoda_hip.so.

Exactly the names the reference uses through `from pcdet.utils.spconv_utils import spconv`
(reference pcdet/utils/spconv_utils.py:3-6; spconv_backbone.py; height_compression.py:21):
SparseConvTensor, SparseSequential, SparseModule, SubMConv3d, SparseConv3d, SparseInverseConv3d,
conv.SparseConvolution, utils.Point2VoxelCPU3d / VoxelGenerator.  `toda_amd.spconv.pytorch` is the
same namespace (spconv 2.x import path)."""
from . import conv, utils  # noqa: F401
from .conv import SparseConv3d, SparseConvolution, SparseInverseConv3d, SubMConv3d  # noqa: F401
from .core import SparseConvTensor  # noqa: F401
from .modules import SparseModule, SparseSequential  # noqa: F401
from .plan import plan_indices, plan_input  # noqa: F401
from .conv import prepack  # noqa: F401

__version__ = "2.1.0+toda_amd"
