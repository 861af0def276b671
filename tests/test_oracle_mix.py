"""The oracle's restatement of the TODA mixing processors (oracle/mix.py) against golden vectors captured from the
reference's own Python processors (tests/golden/capture_mix.py), plus hand-checkable cases for the two compiled
helpers the reference could not provide here (points-in-box, BEV overlap)."""
import os

import numpy as np
import pytest

from oracle import mix as OM
from oracle import oracle as O

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
PC_RANGE = np.array([-54.0, -54.0, -5.0, 54.0, 54.0, 4.8], np.float32)


def load(name):
    z = np.load(os.path.join(GOLD, f"mix_{name}.npz"), allow_pickle=False)
    src = {"points": z["src_points"], "gt_boxes": z["src_boxes"]}
    tgt = {"points": z["tgt_points"], "gt_boxes": z["tgt_boxes"]}
    return z, src, tgt


def run_case(name, engine):
    """engine: module-like with cutmix / polarmix / polar_swap_with_range / lasermix_cyc / lasermix_sph / pseudobbox /
    pseudobackground / mixup taking numpy dicts."""
    z, src, tgt = load(name)
    if name.startswith("pseudo"):
        return z, getattr(engine, name)(src, tgt)
    rng = np.random.RandomState(int(z["seed"]))
    if name.startswith("polar_range"):
        pts, boxes = engine.polar_swap_with_range(src["points"], tgt["points"], float(z["lo"]), float(z["hi"]), src["gt_boxes"],
                                                  tgt["gt_boxes"], PC_RANGE, rng=rng)
        return z, {"points": pts, "gt_boxes": boxes}
    if name.startswith("lasermix_sph"):
        order = int(z["order"]) if "order" in z.files else "center"        # the entry point hands inc_method to `order`
        return z, engine.lasermix_sph(src, tgt, [int(v) for v in z["pitch"]], [int(v) for v in z["num_areas"]], order, rng=rng)
    if name == "cutmix":
        out = engine.cutmix(src, tgt, PC_RANGE, rng=rng)
    elif name.startswith("polarmix"):
        deg = float(z["degree"][0]) if bool(z["degree_is_float"]) else [float(v) for v in z["degree"]]
        out = engine.polarmix(src, tgt, int(z["rc"]), deg, float(z["pct"]), [str(m) for m in z["methods"]], str(z["inc"]), rng=rng,
                              use_pitch=name.startswith("polarmix_pitch"))
    elif name.startswith("lasermix"):
        out = engine.lasermix_cyc(src, tgt, int(z["num_areas"]), int(z["num_angles"]), PC_RANGE, str(z["inc"]), rng=rng)
    else:
        out = engine.mixup(src, tgt, float(z["alpha"]), collision=name.endswith("_cd"), rng=rng)
    return z, out


CASES = ["cutmix", "polarmix_center", "polarmix_corner", "polarmix_corner_del", "lasermix_center", "lasermix_corner_del",
         "mixup", "mixup_cd",
         # round 4: use_pitch, swap_with_range, spherical LaserMix, the pseudo mixes
         "polarmix_pitch_center", "polarmix_pitch_corner_del", "polar_range_near", "polar_range_far",
         "lasermix_sph_entry", "lasermix_sph_order0", "lasermix_sph_order1", "pseudobbox", "pseudobackground"]


@pytest.mark.parametrize("name", CASES)
def test_oracle_mixers_match_reference_outputs(name):
    z, out = run_case(name, OM)
    assert out["gt_boxes"].shape == z["out_boxes"].shape
    np.testing.assert_array_equal(out["gt_boxes"], z["out_boxes"])
    assert out["points"].shape == z["out_points"].shape
    np.testing.assert_array_equal(out["points"], z["out_points"])      # same rows, same order, same bits


def test_points_in_boxes_hand_cases():
    # axis-aligned 4 x 2 x 2 box at the origin: margin 1e-2 in xy (strict), none in z (inclusive)
    box = np.array([[0, 0, 0, 4, 2, 2, 0]], np.float32)
    pts = np.array([[0, 0, 0], [2.005, 0, 0], [2.02, 0, 0], [0, 1.005, 0], [0, 0, 1.0], [0, 0, 1.001], [-2.0, -1.0, -1.0]], np.float32)
    assert O.points_in_boxes(pts, box, 0)[0].tolist() == [1, 1, 0, 1, 1, 0, 1]
    # mode 1 (get_points_in_box): margin 0.1, inclusive
    pts1 = np.array([[2.1, 0, 0], [2.11, 0, 0], [0, 1.1, 1.0], [0, 1.1, 1.01]], np.float32)
    assert O.points_in_boxes(pts1, box, 1)[0].tolist() == [1, 0, 1, 0]
    # heading pi/2 swaps the roles of dx and dy
    boxr = np.array([[0, 0, 0, 4, 2, 2, np.pi / 2]], np.float32)
    ptsr = np.array([[0, 1.9, 0], [1.9, 0, 0], [0.9, 0, 0]], np.float32)
    assert O.points_in_boxes(ptsr, boxr, 0)[0].tolist() == [1, 0, 1]
    # empty inputs
    assert O.points_in_boxes(np.zeros((0, 3), np.float32), box, 0).shape == (1, 0)
    assert OM.points_in_any_box(pts, np.zeros((0, 7), np.float32)).sum() == 0


def test_bev_overlap_zero_and_positive():
    a = np.array([[0, 0, 0, 4, 2, 2, 0.3]], np.float32)
    b = np.array([[10, 0, 0, 4, 2, 2, 1.0], [1, 0.5, 0, 4, 2, 2, -0.4], [0, 0, 5, 4, 2, 2, 0.3]], np.float32)
    iou = OM.bev_overlap(a, b)
    assert iou[0, 0] == 0 and iou[0, 1] > 0.2 and abs(iou[0, 2] - 1.0) < 1e-5      # BEV ignores z


def test_corners_order_and_rotation():
    c = OM.boxes_to_corners(np.array([[1, 2, 3, 4, 2, 6, 0]], np.float32))[0]
    np.testing.assert_allclose(c[0], [3, 3, 0], atol=1e-6)
    np.testing.assert_allclose(c[6], [-1, 1, 6], atol=1e-6)
    c90 = OM.boxes_to_corners(np.array([[0, 0, 0, 4, 2, 2, np.pi / 2]], np.float32))[0]
    np.testing.assert_allclose(c90[0], [-1, 2, -1], atol=1e-5)


def test_polarmix_sector_draws_wrap_and_do_not_overlap():
    rng = np.random.RandomState(5)
    sectors = OM.polarmix_sectors(1.570796, 0.0, ["FIX", "FIX", "FIX"], rng)
    assert all(-np.pi <= lo <= hi <= np.pi for lo, hi in sectors)
    width = sum(hi - lo for lo, hi in sectors)
    assert abs(width - 3 * 1.570796) < 1e-9 or len(sectors) < 3


def test_mixup_edge_cases():
    rng = np.random.RandomState(0)
    d1 = {"points": np.zeros((0, 4), np.float32), "gt_boxes": np.zeros((0, 8), np.float32)}
    d2 = {"points": np.ones((10, 4), np.float32), "gt_boxes": np.array([[1, 1, 1, 2, 2, 2, 0, 1]], np.float32)}
    out = OM.mixup(d1, d2, 2.0, collision=True, rng=rng)      # no boxes in cloud 1: nothing collides, nothing is removed
    assert out["gt_boxes"].shape == (1, 8) and out["points"].shape[0] <= 10
