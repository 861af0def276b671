"""Shared test helpers: random sparse sets and the masked dense conv3d oracle."""
import numpy as np
import torch
import torch.nn.functional as F


def random_sparse(batch, shape, n_per_batch, c, seed=0, sort=False):
    """Unique random (b,z,y,x) int32 rows in arbitrary order + fp32 features."""
    rng = np.random.default_rng(seed)
    rows = []
    vol = int(np.prod(shape))
    for b in range(batch):
        lin = rng.choice(vol, size=min(n_per_batch, vol), replace=False)
        z, rem = np.divmod(lin, shape[1] * shape[2])
        y, x = np.divmod(rem, shape[2])
        rows.append(np.stack([np.full_like(z, b), z, y, x], 1))
    idx = np.concatenate(rows).astype(np.int32)
    if sort:
        key = ((idx[:, 0].astype(np.int64) * shape[0] + idx[:, 1]) * shape[1] + idx[:, 2]) * shape[2] + idx[:, 3]
        idx = idx[np.argsort(key)]
    else:
        idx = idx[rng.permutation(len(idx))]
    feat = rng.standard_normal((len(idx), c)).astype(np.float32)
    return idx, feat


def clustered_sparse(batch, shape, n_per_batch, c, seed=0):
    """LiDAR-ish occupancy: a noisy ground sheet plus a few blobs, so neighbours exist."""
    rng = np.random.default_rng(seed)
    D, H, W = shape
    rows = []
    for b in range(batch):
        y = rng.integers(0, H, n_per_batch)
        x = rng.integers(0, W, n_per_batch)
        z = np.clip((D // 3 + rng.normal(0, 0.7, n_per_batch)).round().astype(np.int64), 0, D - 1)
        pts = np.stack([np.full_like(z, b), z, y, x], 1)
        rows.append(np.unique(pts, axis=0))
    idx = np.concatenate(rows).astype(np.int32)
    idx = idx[rng.permutation(len(idx))]
    feat = rng.standard_normal((len(idx), c)).astype(np.float32)
    return idx, feat


def densify(idx, feat, batch, shape):
    c = feat.shape[1]
    dense = torch.zeros((batch, c) + tuple(shape), dtype=torch.float64)
    ii = torch.as_tensor(idx, dtype=torch.long)
    dense[ii[:, 0], :, ii[:, 1], ii[:, 2], ii[:, 3]] = torch.as_tensor(feat, dtype=torch.float64)
    return dense


def dense_weight(w):
    """[Cout,kz,ky,kx,Cin] -> conv3d layout [Cout,Cin,kz,ky,kx]."""
    return torch.as_tensor(w, dtype=torch.float64).permute(0, 4, 1, 2, 3).contiguous()


def active_mask(idx, batch, shape):
    m = torch.zeros((batch, 1) + tuple(shape), dtype=torch.float64)
    ii = torch.as_tensor(idx, dtype=torch.long)
    m[ii[:, 0], 0, ii[:, 1], ii[:, 2], ii[:, 3]] = 1
    return m


def dense_out_sites(idx, batch, shape, ksize, stride, pad):
    """Canonical (ascending) output sites of a strided sparse conv via the dense mask."""
    m = active_mask(idx, batch, shape)
    k = torch.ones((1, 1) + tuple(ksize), dtype=torch.float64)
    om = F.conv3d(m, k, stride=tuple(stride), padding=tuple(pad)) > 0
    sites = om[:, 0].nonzero()  # already lexicographic in (b,z,y,x)
    return sites.numpy().astype(np.int32), list(om.shape[2:])


class abi_calls:
    """Counts calls of C-ABI entry points while a test runs: `with abi_calls("toda_conv3x3_fwd", ...) as n: ...; n["toda_conv3x3_fwd"]`.
    The golden twins use it to prove that a fixture was served by the hand-written kernels and not by a library fallback."""

    def __init__(self, *names):
        self.names = names
        self.count = {n: 0 for n in names}
        self._orig = {}

    def __enter__(self):
        from toda_amd import lib as L

        self._lib = L.load()
        for n in self.names:
            orig = getattr(self._lib, n)
            self._orig[n] = orig

            def wrapper(*args, _orig=orig, _n=n):
                self.count[_n] += 1
                return _orig(*args)

            setattr(self._lib, n, wrapper)
        return self.count

    def __exit__(self, *exc):
        for n, orig in self._orig.items():
            setattr(self._lib, n, orig)
        return False
