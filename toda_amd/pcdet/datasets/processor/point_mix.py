"""TODA's mixing processors with the point work on the MI355X.

The reference mixes two scenes with numpy and single-thread C++ inside DataLoader workers
(pcdet/datasets/processor/inter_domain_point_{cutmix,polarmix,lasermix,pseudomix}.py, intra_domain_point_mixup.py).
Here both raw clouds live in HBM; every per-point decision is a streaming HIP kernel
(csrc/points.hip: azimuth sector with range cut / elevation test, crop rectangle, LaserMix cylinder cell and elevation band,
point-in-box) and every
`points[mask]` / `np.delete` / `np.concatenate` is a stable compaction appended at a device-side cursor
(`ops.RowBuffer`), so a mixed scene is assembled without leaving the device and handed to the voxeliser.
The box bookkeeping (a few dozen rows) stays on the host, written with the reference's fp32/fp64
expressions; random numbers come from `rng` (default numpy's global state) in the reference's order,
so a seeded run makes the reference's decisions.

`points` may be a numpy array (uploaded; the result is returned as numpy, which makes the functions literal
drop-ins) or a CUDA tensor (the result stays on the device).
"""
import numpy as np
import torch

from .... import ops
from ...utils import box_utils

F32 = np.float32


# ------------------------------------------------------------------------------------ plumbing
class _Cloud:
    """Device point table + optional device-side row count."""

    def __init__(self, data, n_dev=None):
        self.data, self.n_dev = data, n_dev

    @property
    def cap(self):
        return self.data.shape[0]


def _upload(points):
    if isinstance(points, np.ndarray):
        return torch.from_numpy(np.ascontiguousarray(points, dtype=F32)).cuda(), True
    if not points.is_cuda:
        return points.float().cuda(), False
    return points.float().contiguous(), False


def _deliver(points, as_numpy):
    return points.cpu().numpy() if as_numpy else points


def _buffer_like(cloud_cap, c, device):
    return ops.RowBuffer(cloud_cap, c, device)


def _as_cloud(buf):
    return _Cloud(buf.data, buf.cursor)


def _boxes_dev(boxes, device):
    return torch.from_numpy(np.ascontiguousarray(boxes[:, :7], dtype=F32)).to(device)


def _drop_in_boxes(cloud, boxes, mode=0):
    """cloud without the points inside any of `boxes` (box_utils.remove_points_in_boxes3d, :75-89)."""
    if len(boxes) == 0:
        return cloud
    flags = ops.points_in_boxes(cloud.data, _boxes_dev(boxes, cloud.data.device), mode, cloud.n_dev)
    out = _buffer_like(cloud.cap, cloud.data.shape[1], cloud.data.device)
    out.append(cloud.data, flags, 1, invert=True, n_dev=cloud.n_dev)
    return _as_cloud(out)


def _yaw32(x, y):
    """-atan2 as the correctly rounded fp32 value (the device computes the same)."""
    return -(np.arctan2(np.asarray(y, np.float64), np.asarray(x, np.float64)).astype(F32))


def _bev_overlap(boxes_a, boxes_b, device):
    if len(boxes_a) == 0 or len(boxes_b) == 0:
        return np.zeros((len(boxes_a), len(boxes_b)), F32)
    return ops.boxes_iou_bev(_boxes_dev(boxes_a, device), _boxes_dev(boxes_b, device)).cpu().numpy()


# -------------------------------------------------------------------------------------- CutMix
def cutmix(source, target, pc_range, rng=np.random):
    """Reference inter_domain_point_cutmix.py:10-90: a random crop (side fractions ~ U(0.5, 1), aspect >= 0.75)
    centred on a random source point is filled with the target's points; boxes follow by corners."""
    pc_range = np.asarray(pc_range, F32)
    sp, as_numpy = _upload(source["points"])
    tp, _ = _upload(target["points"])
    span_xy = pc_range[3:5] - pc_range[0:2]
    frac = 0.5 + rng.rand(2) * 0.5
    redraws = 0
    while frac.min() / frac.max() < 0.75:
        redraws += 1
        frac = 0.5 + rng.rand(2) * 0.5
        if redraws > 100:
            break
    while True:
        half = span_xy * frac / 2.0                                   # fp64, as numpy promotes it
        centre = sp[int(rng.choice(sp.shape[0])), 0:2].cpu().numpy()
        hi, lo = centre + half, centre - half
        in_s = ops.points_rect(sp, lo, hi, closed=False)
        in_t = ops.points_rect(tp, lo, hi, closed=False)
        if int(in_t.sum().item()) > 10000:
            break
    out = ops.RowBuffer(sp.shape[0] + tp.shape[0], sp.shape[1], sp.device)
    out.append(tp, in_t, 1).append(sp, in_s, 1, invert=True)
    region = [lo[0], lo[1], pc_range[2], hi[0], hi[1], pc_range[5]]
    ms = box_utils.mask_boxes_outside_range_numpy(source["gt_boxes"], region, 1)
    mt = box_utils.mask_boxes_outside_range_numpy(target["gt_boxes"], region, 1)
    boxes = np.concatenate([source["gt_boxes"][~ms], target["gt_boxes"][mt]], 0)
    return {"points": _deliver(out.finish(), as_numpy), "gt_boxes": boxes}


# ------------------------------------------------------------------------------------ PolarMix
def _in_sector(yaw, lo, hi):
    return (yaw > F32(lo)) & (yaw < F32(hi))


def _swap_sector(cloud1, cloud2, lo, hi, box1, box2, inc_method, use_pitch=False):
    """One sector of PolarMix's scene-level swap (inter_domain_point_polarmix.py:44-99).  use_pitch (:81-93): cloud 2 also gives
    its points OUTSIDE the sector whose elevation lies outside cloud 1's elevation span (both judged beyond 1 m of range);
    they are placed before the sector's points.  The span is reduced on the device and read by the next kernel there."""
    if inc_method == "center":
        out1 = _in_sector(_yaw32(box1[:, 0], box1[:, 1]), lo, hi)
        in2 = _in_sector(_yaw32(box2[:, 0], box2[:, 1]), lo, hi)
    elif inc_method in ("corner", "corner_del"):
        c1 = box_utils.boxes_to_corners_3d(box1)[:, :, :2] if len(box1) else np.zeros((0, 8, 2), F32)
        c2 = box_utils.boxes_to_corners_3d(box2)[:, :, :2] if len(box2) else np.zeros((0, 8, 2), F32)
        s1 = _in_sector(_yaw32(c1[:, :, 0], c1[:, :, 1]), lo, hi)
        s2 = _in_sector(_yaw32(c2[:, :, 0], c2[:, :, 1]), lo, hi)
        out1, in2 = s1.any(1), s2.all(1)
        if inc_method == "corner_del":                                # boxes cut by the sector edge lose their points
            cloud1 = _drop_in_boxes(cloud1, box1[out1 != s1.all(1)])
            cloud2 = _drop_in_boxes(cloud2, box2[in2 != s2.any(1)])
    else:
        raise NotImplementedError(inc_method)
    boxes = np.concatenate([box1[~out1], box2[in2]], 0)
    f1 = ops.points_sector(cloud1.data, F32(lo), F32(hi), cloud1.n_dev)
    f2 = ops.points_sector(cloud2.data, F32(lo), F32(hi), cloud2.n_dev)
    out = ops.RowBuffer(cloud1.cap + cloud2.cap, cloud1.data.shape[1], cloud1.data.device)
    out.append(cloud1.data, f1, 1, invert=True, n_dev=cloud1.n_dev)
    if use_pitch:
        span = ops.points_pitch_range(cloud1.data, cloud1.n_dev)
        beyond = ops.points_polar_select(cloud2.data, F32(lo), F32(hi), outside=True, pitch_range=span, n_dev=cloud2.n_dev)
        out.append(cloud2.data, beyond, 1, n_dev=cloud2.n_dev)
    out.append(cloud2.data, f2, 1, n_dev=cloud2.n_dev)
    return _as_cloud(out), boxes


def _swap_sector_range(cloud1, cloud2, lo, hi, box1, box2, pc_range, rng):
    """swap_with_range (inter_domain_point_polarmix.py:101-151): the sector is cut at a random range; the near part (threshold
    beyond 40 % of the x range) or the far part changes hands, boxes follow by centre."""
    r_max = F32(np.asarray(pc_range, F32)[3])
    dis_th = F32(rng.random()) * r_max                                # python float * np.float32: fp32
    mode = 1 if F32(dis_th / r_max) > F32(0.4) else 2

    def chosen(x, y):
        x, y = np.asarray(x, F32), np.asarray(y, F32)
        d = np.sqrt(x * x + y * y)
        return _in_sector(_yaw32(x, y), lo, hi) & ((d < dis_th) if mode == 1 else (d > dis_th))

    b1, b2 = chosen(box1[:, 0], box1[:, 1]), chosen(box2[:, 0], box2[:, 1])
    f1 = ops.points_polar_select(cloud1.data, F32(lo), F32(hi), dis_mode=mode, dis_th=dis_th, n_dev=cloud1.n_dev)
    f2 = ops.points_polar_select(cloud2.data, F32(lo), F32(hi), dis_mode=mode, dis_th=dis_th, n_dev=cloud2.n_dev)
    out = ops.RowBuffer(cloud1.cap + cloud2.cap, cloud1.data.shape[1], cloud1.data.device)
    out.append(cloud1.data, f1, 1, invert=True, n_dev=cloud1.n_dev).append(cloud2.data, f2, 1, n_dev=cloud2.n_dev)
    return _as_cloud(out), np.concatenate([box1[~b1], box2[b2]], 0)


def polar_swap_with_range(pt1, pt2, lo, hi, box1, box2, pc_range, rng=np.random):
    """swap_with_range as a function of its own (numpy in -> numpy out, CUDA tensors stay on the device)."""
    p1, as_numpy = _upload(pt1)
    p2, _ = _upload(pt2)
    cloud, boxes = _swap_sector_range(_Cloud(p1), _Cloud(p2), lo, hi, box1, box2, pc_range, rng)
    return _deliver(cloud.data[:int(cloud.n_dev.item())], as_numpy), boxes


def polarmix_sectors(degree, train_percent, update_methods, rng):
    """Azimuth sectors [start, start + width] (inter_domain_point_polarmix.py:248-286): one per update method, start
    ~ U(-pi, pi) re-drawn (<= 100 times) until it clears the earlier ones; sectors crossing +pi are split."""
    if isinstance(degree, float):
        bounds = (degree, degree)
    else:
        bounds = (degree[0], degree[0]) if len(degree) == 1 else (degree[0], degree[1])
    grow = bounds[1] - bounds[0]
    sectors = []
    for method in update_methods:
        if method == "FIX":
            width = bounds[0]
        elif method == "RAND":
            width = rng.uniform(bounds[0], bounds[1])
        elif method == "ASC":
            width = bounds[0] + grow * train_percent
        elif method == "ASC_SIG":
            width = bounds[0] + grow * (1 / (1 + np.exp(-6 * (train_percent * 2 - 1))))
        elif method == "DESC":
            width = bounds[1] - grow * train_percent
        else:
            raise NotImplementedError(method)
        earlier = [tuple(sorted(s)) for s in sectors]
        for _ in range(100):
            start = (rng.random() * 2 - 1) * np.pi
            lo_, hi_ = sorted((start, start + width))
            touching = False
            for p, q in earlier:                                     # the reference stops at the first overlap
                touching = not (q < lo_ or hi_ < p)
                if touching:
                    break
            if not touching:
                sectors.append([start, start + width])
                break
        for i in range(len(sectors)):
            if sectors[i][1] > np.pi:
                sectors.append([-np.pi, sectors[i][1] - (np.pi * 2)])
                sectors[i][1] = np.pi
    return sectors


def _rotate_paste(cloud2, boxes2, omegas, placed_boxes):
    """Instance-level rotate-paste (inter_domain_point_polarmix.py:153-191): for every angle, the target's boxes
    (and the points inside them) rotated about z; copies that touch anything already placed are dropped."""
    device, c = cloud2.data.device, cloud2.data.shape[1]
    pasted = ops.RowBuffer(cloud2.cap * max(len(omegas), 1), c, device)
    new_boxes, placed = [], [placed_boxes]
    for om in omegas:
        cs, sn = np.cos(om), np.sin(om)
        rot = np.array([[cs, sn, 0], [-sn, cs, 0], [0, 0, 1]])
        moved = boxes2.copy()
        moved[:, :3] = np.dot(boxes2[:, :3], rot)
        moved[:, 6] += om
        free = _bev_overlap(np.concatenate(placed, 0), moved, device).sum(0) == 0
        moved = moved[free]
        new_boxes.append(moved)
        placed.append(moved)
        flags = ops.points_in_boxes(cloud2.data, _boxes_dev(boxes2[free], device), 0, cloud2.n_dev)
        inst = ops.RowBuffer(cloud2.cap, c, device).append(cloud2.data, flags, 1, n_dev=cloud2.n_dev)
        pasted.append(ops.points_rotate_z(inst.data, cs, sn, inst.cursor), n_dev=inst.cursor)
    return _as_cloud(pasted), np.concatenate(new_boxes, 0)


def polarmix(source, target, rot_copy_num, degree, train_percent, update_methods, inc_method="center", rng=np.random,
             polar_dis="FULL", use_pitch=False, pc_range=None):
    """Reference inter_domain_point_polarmix.py:193-300.  POLARMIX_DIS = RAND: the reference's own call of swap_with_range
    carries a keyword that function does not take (:215-220, a TypeError as shipped); this is that call without it."""
    if polar_dis not in ("FULL", "RAND"):
        raise NotImplementedError(polar_dis)
    if polar_dis == "RAND" and pc_range is None:
        raise ValueError("POLARMIX_DIS = RAND needs the point cloud range")
    sp, as_numpy = _upload(source["points"])
    tp, _ = _upload(target["points"])
    sectors = polarmix_sectors(degree, train_percent, update_methods, rng)
    omegas = [0, rng.random() * np.pi * 2 / 3, (rng.random() + 1) * np.pi * 2 / 3][:rot_copy_num]
    cloud, boxes, tcloud = _Cloud(sp), source["gt_boxes"], _Cloud(tp)
    rng.random()                              # the reference draws (and ignores) one number per stage: `random() < 1.0`
    for lo, hi in sectors:
        if polar_dis == "FULL":
            cloud, boxes = _swap_sector(cloud, tcloud, lo, hi, boxes, target["gt_boxes"], inc_method, use_pitch)
        else:
            cloud, boxes = _swap_sector_range(cloud, tcloud, lo, hi, boxes, target["gt_boxes"], pc_range, rng)
    rng.random()
    if len(omegas) == 0:
        raise ValueError("POLARMIX_RC_NUM must be >= 1 (the reference concatenates an empty list otherwise)")
    pasted, new_boxes = _rotate_paste(tcloud, target["gt_boxes"], omegas, boxes)
    cloud = _drop_in_boxes(cloud, new_boxes)
    out = ops.RowBuffer(cloud.cap + pasted.cap, sp.shape[1], sp.device)
    out.append(cloud.data, n_dev=cloud.n_dev).append(pasted.data, n_dev=pasted.n_dev)
    return {"points": _deliver(out.finish(), as_numpy), "gt_boxes": np.concatenate([boxes, new_boxes], 0)}


# ------------------------------------------------------------------------------------ LaserMix
def _wrap_phase(yaw, phase):
    y = (yaw + F32(phase)).astype(F32)
    y[y > F32(3.141592)] -= F32(6.283184)
    y[y < F32(-3.141592)] += F32(6.283184)
    return y


def lasermix_cyc(source, target, num_areas, num_angles, pc_range, inc_method="center", rng=np.random):
    """Reference laser_mix_transform_cyc (inter_domain_point_lasermix.py:88-173): a num_angles x num_areas polar grid,
    rotated by a random phase, whose cells alternate between the two scenes."""
    pc_range = np.asarray(pc_range, F32)
    sp, as_numpy = _upload(source["points"])
    tp, _ = _upload(target["points"])
    phase = rng.uniform(-3.141592, 3.141952)
    dis_edges = np.linspace(0, pc_range[3], num_areas + 1)
    yaw_edges = np.linspace(-np.pi, np.pi, num_angles + 1)
    r_lo, r_hi = F32(1e-05), F32(pc_range[3]) - F32(1e-05)

    def host_cells(x, y):
        yaw = _wrap_phase(_yaw32(x, y), phase)
        dis = np.clip(np.sqrt(x ** 2 + y ** 2), r_lo, r_hi)
        return yaw, dis

    scenes = []
    for pts, d in ((sp, source), (tp, target)):
        b = d["gt_boxes"]
        cor = box_utils.boxes_to_corners_3d(b)[:, :, :2] if len(b) else np.zeros((0, 8, 2), F32)
        scenes.append(dict(pts=pts, box=b, cell=ops.points_polar_cell(pts, F32(phase), yaw_edges, dis_edges, r_lo, r_hi),
                           centre=host_cells(b[:, 0], b[:, 1]), corner=host_cells(cor[:, :, 0], cor[:, :, 1])))
    first = rng.choice([0, 1])
    out = ops.RowBuffer(sp.shape[0] + tp.shape[0], sp.shape[1], sp.device)
    out_boxes = []
    for i in range(num_angles):
        turn = i % 2 + first
        for j in range(num_areas):
            s = scenes[turn % 2]
            ylo, yhi, dlo, dhi = yaw_edges[i], yaw_edges[i + 1], dis_edges[j], dis_edges[j + 1]
            key = i * num_areas + j
            if inc_method == "center":
                yb, db = s["centre"]
                out_boxes.append(s["box"][(yb > ylo) & (yb <= yhi) & (db > dlo) & (db <= dhi)])
                out.append(s["pts"], s["cell"], key)
            elif inc_method == "corner_del":
                yc, dc = s["corner"]
                ycell, dcell = (yc > ylo) & (yc <= yhi), (dc > dlo) & (dc <= dhi)
                cut = (ycell.any(1) != ycell.all(1)) | (dcell.any(1) != dcell.all(1))
                out_boxes.append(s["box"][ycell.all(1) & dcell.all(1)])
                cell_pts = ops.RowBuffer(s["pts"].shape[0], sp.shape[1], sp.device).append(s["pts"], s["cell"], key)
                kept = _drop_in_boxes(_as_cloud(cell_pts), s["box"][cut])
                out.append(kept.data, n_dev=kept.n_dev)
            else:
                raise NotImplementedError(inc_method)
            turn += 1
    return {"points": _deliver(out.finish(), as_numpy), "gt_boxes": np.concatenate(out_boxes, 0)}


def lasermix_sph(source, target, pitch_angles, num_areas, order=0, rng=np.random):
    """Reference laser_mix_transform_sph (inter_domain_point_lasermix.py:22-85): elevation bands (degrees, top to bottom)
    alternate between the scenes, band i from `source` when i % 2 == order.  Elevation = arctan2(z - 1.8, range) in radians,
    clipped against the DEGREE bounds +- 1e-5 exactly as the reference does, band edges compared in fp64.  The entry point
    hands inc_method to `order` (:186-192); a string never equals i % 2, so every band then comes from the target."""
    sp, as_numpy = _upload(source["points"])
    tp, _ = _upload(target["points"])
    lo, hi = F32(pitch_angles[0] + 1e-5), F32(pitch_angles[1] - 1e-5)
    n_bands = rng.choice(num_areas, size=1)[0]
    edges = np.linspace(pitch_angles[1], pitch_angles[0], n_bands + 1) / 180 * np.pi

    def box_elevation(b):
        x, y = np.asarray(b[:, 0], F32), np.asarray(b[:, 1], F32)
        rho = np.sqrt(x * x + y * y)
        e = np.arctan2((F32(-1.8) + np.asarray(b[:, 2], F32)).astype(np.float64), rho.astype(np.float64)).astype(F32)
        return np.clip(e, lo, hi)

    scenes = [dict(pts=p, box=d["gt_boxes"], band=ops.points_pitch_band(p, F32(-1.8), lo, hi, edges), ev=box_elevation(d["gt_boxes"]))
              for p, d in ((sp, source), (tp, target))]
    out = ops.RowBuffer(sp.shape[0] + tp.shape[0], sp.shape[1], sp.device)
    out_boxes = []
    for i in range(n_bands):
        s = scenes[0] if i % 2 == order else scenes[1]
        out.append(s["pts"], s["band"], i)
        out_boxes.append(s["box"][(s["ev"] > edges[i + 1]) & (s["ev"] <= edges[i])])
    return {"points": _deliver(out.finish(), as_numpy), "gt_boxes": np.concatenate(out_boxes, 0)}


# --------------------------------------------------------------------------------- pseudo mixes
def pseudobbox(source, target):
    """Reference inter_domain_point_pseudobbox (inter_domain_point_pseudomix.py:19-47): the target's boxes that touch no source
    box in BEV are pasted, with the points inside them, into the source scene, whose own points inside those boxes go."""
    sp, as_numpy = _upload(source["points"])
    tp, _ = _upload(target["points"])
    sb, tb = source["gt_boxes"], target["gt_boxes"]
    paste = tb[_bev_overlap(sb, tb, sp.device).sum(0) == 0]
    pb = _boxes_dev(paste, sp.device)
    out = ops.RowBuffer(sp.shape[0] + tp.shape[0], sp.shape[1], sp.device)
    out.append(sp, ops.points_in_boxes(sp, pb, 0), 1, invert=True).append(tp, ops.points_in_boxes(tp, pb, 0), 1)
    return {"points": _deliver(out.finish(), as_numpy), "gt_boxes": np.concatenate([sb, paste], 0)}


def pseudobackground(source, target):
    """Reference inter_domain_point_pseudobackground (:49-68): the source's objects (points inside its boxes) on the target's
    background (points outside the target's boxes); the boxes are the source's."""
    sp, as_numpy = _upload(source["points"])
    tp, _ = _upload(target["points"])
    sb, tb = source["gt_boxes"], target["gt_boxes"]
    out = ops.RowBuffer(sp.shape[0] + tp.shape[0], sp.shape[1], sp.device)
    out.append(sp, ops.points_in_boxes(sp, _boxes_dev(sb, sp.device), 0), 1)
    out.append(tp, ops.points_in_boxes(tp, _boxes_dev(tb, sp.device), 0), 1, invert=True)
    return {"points": _deliver(out.finish(), as_numpy), "gt_boxes": sb}


# --------------------------------------------------------------------------------------- MixUp
def mixup(d1, d2, alpha, collision=False, rng=np.random):
    """Reference intra_domain_point_mixup[_cd] (intra_domain_point_mixup.py:15-72): lambda ~ Beta(alpha, alpha); the
    first floor(lambda N1) / floor((1 - lambda) N2) points of the two shuffled clouds; with `collision`, boxes of
    cloud 2 that overlap a box of cloud 1 in BEV are dropped together with their points."""
    p1, as_numpy = _upload(d1["points"])
    p2, _ = _upload(d2["points"])
    lam = rng.beta(alpha, alpha)
    keep2 = d2["gt_boxes"]
    if collision and len(d1["gt_boxes"]) > 0:                         # no boxes in cloud 1: the reference's try-block raises and is skipped
        worst = _bev_overlap(d1["gt_boxes"], d2["gt_boxes"], p1.device).max(axis=0) if len(d2["gt_boxes"]) else np.zeros((0,), F32)
        keep2, gone = d2["gt_boxes"][worst == 0], d2["gt_boxes"][worst > 0]
        if len(gone):
            p2 = _drop_in_boxes(_Cloud(p2), gone, mode=1)
            p2 = p2.data[:int(p2.n_dev.item())]
    o1 = torch.from_numpy(rng.permutation(p1.shape[0])[:int(p1.shape[0] * lam)]).to(p1.device)
    o2 = torch.from_numpy(rng.permutation(p2.shape[0])[:int(p2.shape[0] * (1 - lam))]).to(p1.device)
    pts = torch.cat([p1.index_select(0, o1), p2.index_select(0, o2)], 0)
    return {"points": _deliver(pts, as_numpy), "gt_boxes": np.concatenate([d1["gt_boxes"], keep2], 0)}
