"""spconv.utils voxel generators (reference data_processor.py:15-60).  Same constructor and call
signatures as spconv's CPU generators, but the work runs on the MI355X through libtoda_hip.so;
inputs may be numpy arrays (returned as numpy) or CUDA tensors (returned as tensors)."""
import numpy as np
import torch

from .. import ops


class _Out:
    """Mimics the cumm tensorview handles of spconv 2.x (.numpy())."""

    def __init__(self, t):
        self._t = t

    def numpy(self):
        return self._t.cpu().numpy()

    def numpy_view(self):
        return self.numpy()


class Point2VoxelCPU3d:
    def __init__(self, vsize_xyz, coors_range_xyz, num_point_features, max_num_points_per_voxel, max_num_voxels):
        self.vsize = [float(v) for v in vsize_xyz]
        self.range = [float(v) for v in coors_range_xyz]
        self.num_point_features = int(num_point_features)
        self.max_pts, self.max_voxels = int(max_num_points_per_voxel), int(max_num_voxels)
        self.grid_size = [int(v) for v in ops.grid_size_xyz(self.range, self.vsize)]

    def _run(self, pts):
        if isinstance(pts, np.ndarray):
            pts = torch.from_numpy(np.ascontiguousarray(pts, dtype=np.float32))
        elif hasattr(pts, "numpy") and not isinstance(pts, torch.Tensor):
            pts = torch.from_numpy(np.ascontiguousarray(pts.numpy(), dtype=np.float32))
        return ops.voxelize(pts.cuda().float(), self.range, self.vsize, self.max_pts, self.max_voxels)

    def point_to_voxel(self, pts):
        v, c, n = self._run(pts)
        return _Out(v), _Out(c), _Out(n)


Point2VoxelGPU3d = Point2VoxelCPU3d


class VoxelGenerator:
    """spconv 1.x style generator: generate(points) -> (voxels, coordinates, num_points_per_voxel)."""

    def __init__(self, voxel_size, point_cloud_range, max_num_points, max_voxels=20000, **kw):
        self._g = Point2VoxelCPU3d(voxel_size, point_cloud_range, 0, max_num_points, max_voxels)
        self.voxel_size, self.point_cloud_range = self._g.vsize, self._g.range
        self.grid_size = self._g.grid_size

    def generate(self, points, max_voxels=None):
        v, c, n = self._g._run(points)
        if isinstance(points, np.ndarray):
            return v.cpu().numpy(), c.cpu().numpy(), n.cpu().numpy()
        return v, c, n


VoxelGeneratorV2 = VoxelGenerator
