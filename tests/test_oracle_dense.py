"""The CPU oracle against masked dense conv3d + autograd: the pin for the sparse arithmetic."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import oracle as O
from tests import helpers as H

CASES = [
    # (ksize, stride, pad, cin, cout)
    ((3, 3, 3), (2, 2, 2), (1, 1, 1), 4, 6),
    ((3, 3, 3), (2, 2, 2), (0, 1, 1), 5, 3),
    ((3, 1, 1), (2, 1, 1), (0, 0, 0), 3, 7),
    ((3, 3, 3), (2, 2, 2), (1, 1, 1), 32, 64),      # conv3.0's channel pair
    ((3, 3, 3), (2, 2, 2), (0, 1, 1), 64, 128),     # the Res backbone's conv4.0
    ((3, 1, 1), (2, 1, 1), (0, 0, 0), 64, 128),     # conv_out
]


# (cin, cout, lattice, sites per sample): the wide cases run the oracle's 64- and 128-channel code paths (the channel counts
# of conv3 / conv4 and of the Res backbone's last stage) on a lattice large enough for several OpenMP chunks
SUBM_CASES = [(5, 16, [7, 10, 9], 120), (16, 16, [7, 10, 9], 120), (4, 3, [7, 10, 9], 120),
              (64, 64, [9, 24, 24], 1500), (128, 128, [5, 20, 20], 700), (32, 64, [9, 16, 16], 600)]


@pytest.mark.parametrize("cin,cout,shape,n_sites", SUBM_CASES)
def test_subm_fwd_bwd_vs_dense(cin, cout, shape, n_sites):
    batch = 2
    idx, feat = H.clustered_sparse(batch, shape, n_sites, cin, seed=1)
    rng = np.random.default_rng(2)
    w = (rng.standard_normal((cout, 3, 3, 3, cin)) * (0.3 if cin <= 16 else 0.3 * (16.0 / cin) ** 0.5)).astype(np.float32)
    bias = rng.standard_normal(cout).astype(np.float32)
    nbr, cnt = O.rulebook_subm(idx, batch, shape)
    out = O.spconv_fwd(feat, w, nbr, bias)

    x = H.densify(idx, feat, batch, shape).requires_grad_(True)
    wd = H.dense_weight(w).requires_grad_(True)
    y = F.conv3d(x, wd, torch.as_tensor(bias, dtype=torch.float64), padding=1)
    ii = torch.as_tensor(idx, dtype=torch.long)
    y_rows = y[ii[:, 0], :, ii[:, 1], ii[:, 2], ii[:, 3]]
    tol = 2e-5 if cin <= 16 else 1e-4      # fp32 accumulation over 27 * cin terms against the fp64 dense result
    assert np.allclose(out, y_rows.detach().numpy(), atol=tol)
    # pair count == number of ordered active neighbour pairs
    m = H.active_mask(idx, batch, shape)
    nb = F.conv3d(m, torch.ones(1, 1, 3, 3, 3, dtype=torch.float64), padding=1) * m
    assert int(cnt.sum()) == int(nb.sum().item())

    g = rng.standard_normal(out.shape).astype(np.float32)
    (y_rows * torch.as_tensor(g, dtype=torch.float64)).sum().backward()
    din = O.spconv_dgrad(g, w, nbr, flip_k=True)
    dx_rows = x.grad[ii[:, 0], :, ii[:, 1], ii[:, 2], ii[:, 3]].numpy()
    assert np.allclose(din, dx_rows, atol=tol)
    dw = O.spconv_wgrad(feat, g, nbr, w.shape)
    ref_dw = wd.grad.permute(0, 2, 3, 4, 1).numpy()
    assert np.allclose(dw, ref_dw, atol=1e-4 * max(1.0, float(np.abs(ref_dw).max())))


@pytest.mark.parametrize("ks,st,pd,cin,cout", CASES)
def test_strided_fwd_bwd_vs_dense(ks, st, pd, cin, cout):
    shape, batch = [9, 12, 11], 2
    idx, feat = H.clustered_sparse(batch, shape, 90, cin, seed=3)
    rng = np.random.default_rng(4)
    w = (rng.standard_normal((cout,) + ks + (cin,)) * (0.3 if cin <= 16 else 0.3 * (16.0 / cin) ** 0.5)).astype(np.float32)
    tol = 2e-5 if cin <= 16 else 1e-4
    out_idx, sho, o2i, i2o, cnt = O.rulebook_conv(idx, batch, shape, ks, st, pd)
    sites, dsho = H.dense_out_sites(idx, batch, shape, ks, st, pd)
    assert sho == dsho
    assert np.array_equal(out_idx, sites)  # canonical ascending order, exactly the dense active set
    # o2i and i2o describe the same pair set
    K = o2i.shape[0]
    for k in range(K):
        o = np.nonzero(o2i[k] >= 0)[0]
        assert np.array_equal(i2o[k][o2i[k][o]], o)
        assert (i2o[k] >= 0).sum() == len(o) == cnt[k]

    out = O.spconv_fwd(feat, w, o2i)
    x = H.densify(idx, feat, batch, shape).requires_grad_(True)
    wd = H.dense_weight(w).requires_grad_(True)
    y = F.conv3d(x, wd, stride=st, padding=pd)
    oo = torch.as_tensor(out_idx, dtype=torch.long)
    y_rows = y[oo[:, 0], :, oo[:, 1], oo[:, 2], oo[:, 3]]
    assert np.allclose(out, y_rows.detach().numpy(), atol=tol)

    g = rng.standard_normal(out.shape).astype(np.float32)
    (y_rows * torch.as_tensor(g, dtype=torch.float64)).sum().backward()
    ii = torch.as_tensor(idx, dtype=torch.long)
    din = O.spconv_dgrad(g, w, i2o, flip_k=False)
    assert np.allclose(din, x.grad[ii[:, 0], :, ii[:, 1], ii[:, 2], ii[:, 3]].numpy(), atol=tol)
    dw = O.spconv_wgrad(feat, g, o2i, w.shape)
    assert np.allclose(dw, wd.grad.permute(0, 2, 3, 4, 1).numpy(), atol=1e-4)


def test_dense_roundtrip():
    shape, batch = [2, 6, 5], 3
    idx, feat = H.random_sparse(batch, shape, 20, 8, seed=5)
    d = O.sparse_to_dense_fwd(feat, idx, batch, shape)
    assert np.array_equal(d, H.densify(idx, feat, batch, shape).float().numpy())
    assert np.array_equal(O.sparse_to_dense_bwd(d, idx, shape), feat)
