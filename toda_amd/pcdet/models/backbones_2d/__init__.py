from .base_bev_backbone import BaseBEVBackbone

__all__ = {
    "BaseBEVBackbone": BaseBEVBackbone,
}
