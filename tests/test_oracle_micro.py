"""Hand-checkable micro cases for the CPU oracle."""
import numpy as np

from oracle import oracle as O


def test_voxelize_order_and_caps():
    rng = [0, 0, 0, 4, 4, 2]
    vs = [1, 1, 1]
    pts = np.array([
        [2.5, 0.5, 0.5, 10],   # voxel A (x=2,y=0,z=0)
        [0.5, 0.5, 0.5, 11],   # voxel B
        [2.6, 0.4, 0.1, 12],   # A, 2nd point
        [2.7, 0.3, 0.2, 13],   # A, 3rd point -> dropped at P=2
        [3.5, 3.5, 1.5, 14],   # voxel C
        [-0.1, 0.0, 0.0, 15],  # outside (x<0)
        [4.0, 1.0, 1.0, 16],   # on the upper bound -> floor == grid -> dropped
        [1.5, 1.5, 1.5, 17],   # voxel D -> beyond max_voxels=3 -> skipped
        [0.6, 0.6, 0.6, 18],   # B again: still accepted after the cap was hit (continue semantics)
        [np.nan, 0, 0, 19],    # NaN dropped
    ], np.float32)
    v, c, n = O.voxelize_hard(pts, rng, vs, max_pts=2, max_voxels=3)
    assert c.tolist() == [[0, 0, 2], [0, 0, 0], [1, 3, 3]]  # (z,y,x), first-appearance order
    assert n.tolist() == [2, 2, 1]
    assert v[0, :, 3].tolist() == [10, 12]
    assert v[1, :, 3].tolist() == [11, 18]
    assert v[2, :, 3].tolist() == [14, 0]
    assert (v[2, 1] == 0).all()


def test_voxelize_empty():
    v, c, n = O.voxelize_hard(np.zeros((0, 4), np.float32), [0, 0, 0, 1, 1, 1], [1, 1, 1], 3, 5)
    assert v.shape == (0, 3, 4) and c.shape == (0, 3) and n.shape == (0,)


def test_mean_vfe():
    vox = np.zeros((2, 3, 2), np.float32)
    vox[0, :2] = [[1, 2], [3, 6]]
    out = O.mean_vfe_fwd(vox, np.array([2, 0], np.float32))
    assert np.allclose(out, [[2, 4], [0, 0]])
    g = O.mean_vfe_bwd(np.array([[2, 4], [1, 1]], np.float32), np.array([2, 0], np.float32), 3)
    assert np.allclose(g[0], [[1, 2]] * 3) and np.allclose(g[1], [[1, 1]] * 3)


def test_subm_single_voxel_centre_tap_only():
    idx = np.array([[0, 2, 2, 2]], np.int32)
    nbr, cnt = O.rulebook_subm(idx, 1, [5, 5, 5])
    assert cnt.tolist() == [0] * 13 + [1] + [0] * 13
    w = np.random.default_rng(0).standard_normal((3, 3, 3, 3, 2)).astype(np.float32)
    x = np.array([[1.0, -2.0]], np.float32)
    out = O.spconv_fwd(x, w, nbr)
    assert np.allclose(out[0], w[:, 1, 1, 1, :] @ x[0], atol=1e-6)


def test_subm_two_face_adjacent_and_border():
    idx = np.array([[0, 0, 0, 0], [0, 0, 0, 1]], np.int32)  # on the grid corner, x-adjacent
    nbr, cnt = O.rulebook_subm(idx, 1, [4, 4, 4])
    # offset (0,0,+1) -> k = 13+1; offset (0,0,-1) -> k = 12
    assert nbr[13].tolist() == [0, 1]
    assert nbr[14].tolist() == [1, -1]
    assert nbr[12].tolist() == [-1, 0]
    assert cnt.sum() == 4


def test_stride2_parity_case():
    # inputs at z = 0 and z = 1, k=3 s=2 p=1: z=0 -> out 0 (kz=1); z=1 -> out 0 (kz=2) and out 1 (kz=0)
    idx = np.array([[0, 0, 0, 0], [0, 1, 0, 0]], np.int32)
    out_idx, sho, o2i, i2o, cnt = O.rulebook_conv(idx, 1, [4, 1, 1], [3, 1, 1], [2, 1, 1], [1, 0, 0])
    assert sho == [2, 1, 1]
    assert out_idx.tolist() == [[0, 0, 0, 0], [0, 1, 0, 0]]
    assert o2i.tolist() == [[-1, 1], [0, -1], [1, -1]]
    assert i2o.tolist() == [[-1, 1], [0, -1], [-1, 0]]
    assert cnt.tolist() == [1, 1, 1]


def test_conv_out_k311_s211_on_d5():
    # D=5 -> 2; out 0 <- z 0,1,2; out 1 <- z 2,3,4
    idx = np.array([[0, z, 0, 0] for z in range(5)], np.int32)
    out_idx, sho, o2i, i2o, cnt = O.rulebook_conv(idx, 1, [5, 1, 1], [3, 1, 1], [2, 1, 1], [0, 0, 0])
    assert sho == [2, 1, 1]
    assert o2i.tolist() == [[0, 2], [1, 3], [2, 4]]
    assert cnt.tolist() == [2, 2, 2]


def test_sparse_to_dense_channel_fold():
    idx = np.array([[1, 1, 0, 2]], np.int32)
    feat = np.array([[5.0, 7.0]], np.float32)
    d = O.sparse_to_dense_fwd(feat, idx, 2, [2, 1, 3])
    assert d.shape == (2, 2, 2, 1, 3)
    bev = d.reshape(2, 4, 1, 3)  # channel = c*D + d (height_compression.py:22-23)
    assert bev[1, 0 * 2 + 1, 0, 2] == 5 and bev[1, 1 * 2 + 1, 0, 2] == 7
    assert d.sum() == 12
    g = O.sparse_to_dense_bwd(d, idx, [2, 1, 3])
    assert g.tolist() == [[5.0, 7.0]]


def test_rotated_iou_hand_cases():
    a = np.array([[0, 0, 0, 4, 2, 1, 0], [0, 0, 0, 2, 2, 1, 0]], np.float32)
    b = np.array([[0, 0, 0, 4, 2, 1, 0],            # identical to a0 -> 1
                  [2, 0, 0, 4, 2, 1, 0],            # shifted by half its length: 4 / (8 + 8 - 4)
                  [0, 0, 0, 2, 2, 1, np.pi / 4],    # square vs itself turned 45 deg: octagon 8(sqrt2-1)
                  [0, 0, 0, 2, 4, 1, np.pi / 2],    # 2x4 turned 90 deg == 4x2
                  [9, 9, 0, 1, 1, 1, 0.3]], np.float32)
    iou = O.boxes_iou_bev(a, b)
    oct_area = 8 * (np.sqrt(2) - 1)
    assert np.allclose(iou[0], [1.0, 1 / 3, iou[0, 2], 1.0, 0.0], atol=1e-5)
    assert np.allclose(iou[1, 2], oct_area / (8 - oct_area), atol=1e-4)
    assert np.allclose(iou[1, 0], 0.5, atol=1e-5) and iou[1, 4] == 0.0


def test_nms_greedy_order():
    boxes = np.array([[0, 0, 0, 4, 2, 1, 0],        # kept
                      [0.2, 0, 0, 4, 2, 1, 0.05],   # suppressed by 0
                      [10, 0, 0, 4, 2, 1, 0],       # kept
                      [10, 0.1, 0, 4, 2, 1, 0],     # suppressed by 2
                      [0, 5, 0, 4, 2, 1, 1.0]], np.float32)
    assert O.nms_rotated(boxes, 0.5).tolist() == [0, 2, 4]
    assert O.nms_rotated(boxes, 0.99).tolist() == [0, 1, 2, 3, 4]
    assert O.nms_rotated(boxes[:0], 0.5).tolist() == []
