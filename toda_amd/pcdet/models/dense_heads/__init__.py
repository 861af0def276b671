from .anchor_head_single import AnchorHeadSingle
from .anchor_head_template import AnchorHeadTemplate
from .center_head import CenterHead

__all__ = {
    "AnchorHeadTemplate": AnchorHeadTemplate,
    "AnchorHeadSingle": AnchorHeadSingle,
    "CenterHead": CenterHead,
}
