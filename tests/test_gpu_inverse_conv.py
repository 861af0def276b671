"""spconv.SparseInverseConv3d (reference call site pcdet/models/backbones_3d/spconv_backbone.py:16-17,22-23: the "inverseconv" block of
the UNet-style backbones): the pairs of the strided SparseConv3d that wrote the indice_key, read from its output sites back to its
input sites.  Checked against the oracle's per-offset gather / GEMM / scatter over the swapped tables."""
import numpy as np
import pytest
import torch

from oracle import oracle as O
from tests import helpers as H

pytestmark = pytest.mark.gpu


def dev(a):
    return torch.as_tensor(np.ascontiguousarray(a)).cuda()


@pytest.mark.parametrize("cin,cout", [(64, 32), (32, 16), (128, 64)])
def test_inverse_conv_forward_and_gradients_against_the_oracle(cin, cout):
    from toda_amd import spconv

    shape, batch = [9, 40, 44], 2
    idx, feat_lo = H.clustered_sparse(batch, shape, 1500, cout, seed=cin + cout)
    torch.manual_seed(cin)
    down = spconv.SparseConv3d(cout, cin, 3, stride=2, padding=1, bias=False, indice_key="spconv2").cuda()
    up = spconv.SparseInverseConv3d(cin, cout, 3, indice_key="spconv2", bias=False).cuda()
    x = spconv.SparseConvTensor(dev(feat_lo), dev(idx), shape, batch)
    mid = down(x)
    hi = mid.features.detach().clone().requires_grad_(True)
    out = up(mid.replace_feature(hi))
    assert out.spatial_shape == shape and torch.equal(out.indices, x.indices) and out.features.shape == (len(idx), cout)

    idx_out, _, o2i, i2o, _ = O.rulebook_conv(idx, batch, shape, 3, 2, 1)
    assert np.array_equal(idx_out, mid.indices.cpu().numpy())
    w = up.weight.detach().cpu().numpy()
    f = hi.detach().cpu().numpy()
    ref = O.spconv_fwd(f, w, i2o)                                  # out[i] = sum_k W_k f[i2o[k][i]]
    np.testing.assert_allclose(out.features.detach().cpu().numpy(), ref, rtol=1e-4, atol=1e-4)

    g = np.random.default_rng(1).standard_normal(ref.shape).astype(np.float32)
    out.features.backward(dev(g))
    np.testing.assert_allclose(hi.grad.cpu().numpy(), O.spconv_dgrad(g, w, o2i, flip_k=False), rtol=1e-4, atol=1e-4)
    dw = O.spconv_wgrad(f, g, i2o, w.shape)
    np.testing.assert_allclose(up.weight.grad.cpu().numpy(), dw, rtol=1e-4, atol=1e-4 * max(1.0, float(np.abs(dw).max())))


def test_inverse_conv_needs_the_strided_convolution_of_its_key():
    from toda_amd import spconv

    idx, feat = H.clustered_sparse(1, [5, 24, 24], 300, 16, seed=3)
    x = spconv.SparseConvTensor(dev(feat), dev(idx), [5, 24, 24], 1)
    up = spconv.SparseInverseConv3d(16, 16, 3, indice_key="nowhere", bias=False).cuda()
    with pytest.raises(ValueError, match="indice_key"):
        up(x)
