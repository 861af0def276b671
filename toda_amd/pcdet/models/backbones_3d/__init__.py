from .spconv_backbone import VoxelBackBone8x, VoxelResBackBone8x

__all__ = {
    "VoxelBackBone8x": VoxelBackBone8x,
    "VoxelResBackBone8x": VoxelResBackBone8x,
}
