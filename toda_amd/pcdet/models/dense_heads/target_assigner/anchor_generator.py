"""AnchorGenerator (reference target_assigner/anchor_generator.py:4-59): per class a dense grid of
anchors [nz, ny, nx, n_size, n_rot, 7] on the head's feature map; z is shifted to the box centre."""
import torch


class AnchorGenerator:
    def __init__(self, anchor_range, anchor_generator_config):
        self.anchor_generator_cfg = anchor_generator_config
        self.anchor_range = [float(v) for v in anchor_range]
        self.anchor_sizes = [c["anchor_sizes"] for c in anchor_generator_config]
        self.anchor_rotations = [c["anchor_rotations"] for c in anchor_generator_config]
        self.anchor_heights = [c["anchor_bottom_heights"] for c in anchor_generator_config]
        self.align_center = [c.get("align_center", False) for c in anchor_generator_config]
        assert len(self.anchor_sizes) == len(self.anchor_rotations) == len(self.anchor_heights)
        self.num_of_anchor_sets = len(self.anchor_sizes)

    def generate_anchors(self, grid_sizes, device=None):
        assert len(grid_sizes) == self.num_of_anchor_sets
        r = self.anchor_range
        all_anchors, per_location = [], []
        for grid, sizes, rots, heights, centred in zip(grid_sizes, self.anchor_sizes, self.anchor_rotations,
                                                       self.anchor_heights, self.align_center):
            per_location.append(len(rots) * len(sizes) * len(heights))
            if centred:
                sx, sy = (r[3] - r[0]) / grid[0], (r[4] - r[1]) / grid[1]
                ox, oy = sx / 2, sy / 2
            else:
                sx, sy = (r[3] - r[0]) / (grid[0] - 1), (r[4] - r[1]) / (grid[1] - 1)
                ox, oy = 0, 0
            xs = torch.arange(r[0] + ox, r[3] + 1e-5, step=sx, dtype=torch.float32, device=device)
            ys = torch.arange(r[1] + oy, r[4] + 1e-5, step=sy, dtype=torch.float32, device=device)
            zs = xs.new_tensor(heights)
            size_t, rot_t = xs.new_tensor(sizes), xs.new_tensor(rots)
            nx, ny, nz, ns, nr = len(xs), len(ys), len(zs), size_t.shape[0], rot_t.shape[0]
            gx, gy, gz = torch.meshgrid([xs, ys, zs], indexing="ij")            # [x, y, z]
            a = torch.zeros((nx, ny, nz, ns, nr, 7), dtype=torch.float32, device=device)
            a[..., 0] = gx[..., None, None]
            a[..., 1] = gy[..., None, None]
            a[..., 2] = gz[..., None, None]
            a[..., 3:6] = size_t.view(1, 1, 1, ns, 1, 3)
            a[..., 6] = rot_t.view(1, 1, 1, 1, nr)
            a = a.permute(2, 1, 0, 3, 4, 5).contiguous()                          # [z, y, x, size, rot, 7]
            a[..., 2] += a[..., 5] / 2
            all_anchors.append(a)
        return all_anchors, per_location
