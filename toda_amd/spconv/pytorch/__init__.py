"""`import spconv.pytorch as spconv` compatibility: re-exports toda_amd.spconv."""
from .. import conv, utils  # noqa: F401
from ..conv import SparseConv3d, SparseConvolution, SparseInverseConv3d, SubMConv3d  # noqa: F401
from ..core import SparseConvTensor  # noqa: F401
from ..modules import SparseModule, SparseSequential  # noqa: F401
