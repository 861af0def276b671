"""CPU restatement (numpy) of the TODA mixing processors — TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

What it restates (reference file:line):
  cutmix            pcdet/datasets/processor/inter_domain_point_cutmix.py:10-90
  polarmix          pcdet/datasets/processor/inter_domain_point_polarmix.py:44-99 (swap, with and without use_pitch),
                    :101-151 (swap_with_range), :153-191 (rotate_copy), :193-245 (polarmix), :247-300 (sector / Omega draws)
  lasermix (cyc)    pcdet/datasets/processor/inter_domain_point_lasermix.py:88-173
  lasermix (sph)    pcdet/datasets/processor/inter_domain_point_lasermix.py:22-85, entry point :176-192
  pseudo mixes      pcdet/datasets/processor/inter_domain_point_pseudomix.py:19-47 (pseudobbox), :49-68 (pseudobackground)
  mixup / mixup_cd  pcdet/datasets/processor/intra_domain_point_mixup.py:15-72
  primitives        pcdet/utils/box_utils.py:28-89, pcdet/utils/common_utils.py:34-63,
                    pcdet/ops/roiaware_pool3d/src/roiaware_pool3d.cpp:121-168 (oracle_points_in_boxes),
                    pcdet/ops/iou3d_nms/src/iou3d_cpu.cpp:232-252 (oracle_boxes_iou_bev)

Random numbers are drawn from `rng` (default: the global numpy state, like the reference) in the
reference's order, so a seeded run reproduces the reference's decisions.

Pinned by tests/golden/mix_*.npz: outputs of the reference's own Python mixers run in the build
container on seeded inputs (capture script tests/golden/capture_mix.py).  The reference's compiled
helpers (points_in_boxes_cpu, boxes_iou_bev_cpu) cannot be built here (CUDA headers / launchers), so
during capture those two symbols are served by this oracle's C restatements; the fixtures therefore
pin the mixing logic (draw order, masks, ordering of the output rows), the two helpers are pinned by
hand-checkable cases.

One deliberate difference: the azimuth is the correctly rounded fp32 value of atan2 (computed in
fp64, then rounded).  numpy's own fp32 arctan2 is a SIMD approximation whose last bit depends on the
host CPU (38 % of random inputs differ from the correctly rounded value on this machine), so the
reference itself is only defined up to that bit; a point flips sides only when a sector edge falls
inside that 1-ulp window (about 1e-8 per point and edge).
"""
import numpy as np
import torch

from . import oracle as O

F32 = np.float32


def yaw32(x, y):
    return -(np.arctan2(np.asarray(y, np.float64), np.asarray(x, np.float64)).astype(F32))


def _w(v):
    """A python-float threshold as numpy compares it against an fp32 array (weak scalar -> fp32)."""
    return F32(v)


# ---------------------------------------------------------------------------------- primitives
def boxes_to_corners(boxes):
    """[N, >=7] -> [N, 8, 3] fp32 with the reference's corner order and fp32 torch arithmetic
    (box_utils.py:28-54, common_utils.py:34-57)."""
    b = torch.from_numpy(np.ascontiguousarray(boxes, dtype=F32))
    template = b.new_tensor(([1, 1, -1], [1, -1, -1], [-1, -1, -1], [-1, 1, -1],
                             [1, 1, 1], [1, -1, 1], [-1, -1, 1], [-1, 1, 1])) / 2
    corners = b[:, None, 3:6].repeat(1, 8, 1) * template[None, :, :]
    ang = b[:, 6]
    cosa, sina = torch.cos(ang), torch.sin(ang)
    zeros, ones = ang.new_zeros(b.shape[0]), ang.new_ones(b.shape[0])
    rot = torch.stack((cosa, sina, zeros, -sina, cosa, zeros, zeros, zeros, ones), dim=1).view(-1, 3, 3).float()
    corners = torch.matmul(corners.view(-1, 8, 3), rot).view(-1, 8, 3)
    corners += b[:, None, 0:3]
    return corners.numpy()


def boxes_with_corners_in_range(boxes, limit, min_corners=1):
    """mask_boxes_outside_range_numpy (box_utils.py:57-72)."""
    if len(boxes) == 0:
        return np.zeros((0,), bool)
    corners = boxes_to_corners(boxes[:, 0:7])
    lim = np.asarray(limit, np.float64)
    inside = ((corners >= lim[0:3]) & (corners <= lim[3:6])).all(axis=2)
    return inside.sum(axis=1) >= min_corners


def points_in_any_box(points, boxes, mode=0):
    if len(boxes) == 0 or len(points) == 0:
        return np.zeros((len(points),), bool)
    return O.points_in_boxes(points[:, 0:3], boxes[:, 0:7], mode).sum(0) != 0


def drop_points_in_boxes(points, boxes):
    """box_utils.remove_points_in_boxes3d (box_utils.py:75-89)."""
    return points[~points_in_any_box(points, boxes, 0)]


def bev_overlap(boxes_a, boxes_b):
    if len(boxes_a) == 0 or len(boxes_b) == 0:
        return np.zeros((len(boxes_a), len(boxes_b)), F32)
    return O.boxes_iou_bev(boxes_a[:, :7], boxes_b[:, :7])


# -------------------------------------------------------------------------------------- CutMix
def cutmix(source, target, pc_range, rng=np.random):
    pc_range = np.asarray(pc_range, F32)
    span_xy = pc_range[3:5] - pc_range[0:2]
    frac = 0.5 + rng.rand(2) * 0.5
    tries = 0
    while frac.min() / frac.max() < 0.75:
        tries += 1
        frac = 0.5 + rng.rand(2) * 0.5
        if tries > 100:
            break
    sp, tp = source["points"], target["points"]
    while True:
        half = span_xy * frac / 2.0
        centre = sp[rng.choice(len(sp)), 0:3]
        hi, lo = centre[:2] + half, centre[:2] - half
        in_s = ((sp[:, :2] < hi).sum(1) == 2) & ((sp[:, :2] > lo).sum(1) == 2)
        in_t = ((tp[:, :2] < hi).sum(1) == 2) & ((tp[:, :2] > lo).sum(1) == 2)
        if in_t.sum() > 10000:
            break
    points = np.concatenate([tp[in_t], sp[~in_s]], 0)
    region = [lo[0], lo[1], pc_range[2], hi[0], hi[1], pc_range[5]]
    ms = boxes_with_corners_in_range(source["gt_boxes"], region, 1)
    mt = boxes_with_corners_in_range(target["gt_boxes"], region, 1)
    boxes = np.concatenate([source["gt_boxes"][~ms], target["gt_boxes"][mt]], 0)
    return {"points": points, "gt_boxes": boxes}


# ------------------------------------------------------------------------------------ PolarMix
def _sector(yaw, lo, hi):
    return (yaw > _w(lo)) & (yaw < _w(hi))


def pitch32(z, dis, sign=-1.0):
    """sign * arctan2(z, dis) as the correctly rounded fp32 value (see the module note on numpy's fp32 arctan2)."""
    v = np.arctan2(np.asarray(z, np.float64), np.asarray(dis, np.float64)).astype(F32)
    return -v if sign < 0 else v


def _range32(x, y):
    """np.sqrt(x ** 2 + y ** 2) of fp32 columns: every step rounded to fp32."""
    x, y = np.asarray(x, F32), np.asarray(y, F32)
    return np.sqrt(x * x + y * y)


def polar_swap(pt1, pt2, lo, hi, box1, box2, inc_method="center", use_pitch=False):
    if inc_method == "center":
        take1 = _sector(yaw32(box1[:, 0], box1[:, 1]), lo, hi)
        take2 = _sector(yaw32(box2[:, 0], box2[:, 1]), lo, hi)
    elif inc_method in ("corner", "corner_del"):
        c1, c2 = boxes_to_corners(box1)[:, :, :2], boxes_to_corners(box2)[:, :, :2]
        s1 = _sector(yaw32(c1[:, :, 0], c1[:, :, 1]), lo, hi)
        s2 = _sector(yaw32(c2[:, :, 0], c2[:, :, 1]), lo, hi)
        take1, take2 = s1.any(1), s2.all(1)
        if inc_method == "corner_del":
            pt1 = drop_points_in_boxes(pt1, box1[take1 != s1.all(1)][:, :7])
            pt2 = drop_points_in_boxes(pt2, box2[take2 != s2.any(1)][:, :7])
    else:
        raise NotImplementedError(inc_method)
    boxes = np.concatenate([box1[~take1], box2[take2]], 0)
    yaw1, yaw2 = yaw32(pt1[:, 0], pt1[:, 1]), yaw32(pt2[:, 0], pt2[:, 1])
    in1, in2 = _sector(yaw1, lo, hi), _sector(yaw2, lo, hi)
    if not use_pitch:
        return np.concatenate([pt1[~in1], pt2[in2]], 0), boxes
    # use_pitch (:81-93): cloud 2 also gives the points OUTSIDE the sector whose elevation lies outside cloud 1's
    # elevation span (both judged beyond 1 m of range); they come before the sector's points
    dis1, dis2 = _range32(pt1[:, 0], pt1[:, 1]), _range32(pt2[:, 0], pt2[:, 1])
    far1, far2 = dis1 > F32(1), dis2 > F32(1)
    pitch1, pitch2 = pitch32(pt1[:, 2], dis1), pitch32(pt2[:, 2], dis2)
    p_min, p_max = pitch1[far1].min(), pitch1[far1].max()
    beyond = ((yaw2 < _w(lo)) | (yaw2 > _w(hi))) & ((pitch2 < p_min) | (pitch2 > p_max)) & far2
    return np.concatenate([pt1[~in1], pt2[beyond], pt2[in2]], 0), boxes


def polar_swap_with_range(pt1, pt2, lo, hi, box1, box2, pc_range, rng=np.random):
    """swap_with_range (:101-151): the sector is cut at a random range - the near part (threshold beyond 40 % of the
    x range) or the far part is exchanged; boxes follow by centre."""
    r_max = F32(np.asarray(pc_range, F32)[3])
    dis_th = F32(rng.random()) * r_max                 # python float * np.float32 -> fp32
    near = F32(dis_th / r_max) > F32(0.4)

    def chosen(x, y):
        inside = _sector(yaw32(x, y), lo, hi)
        d = _range32(x, y)
        return inside & ((d < dis_th) if near else (d > dis_th))

    in1, in2 = chosen(pt1[:, 0], pt1[:, 1]), chosen(pt2[:, 0], pt2[:, 1])
    b1, b2 = chosen(box1[:, 0], box1[:, 1]), chosen(box2[:, 0], box2[:, 1])
    return np.concatenate([pt1[~in1], pt2[in2]], 0), np.concatenate([box1[~b1], box2[b2]], 0)


def rotate_paste_candidates(pts, boxes, omegas, existing):
    """rotate_copy: rotated copies of all `boxes` (+ their points) that do not touch anything placed so far."""
    out_pts, out_boxes, placed = [], [], [existing]
    for om in omegas:
        rot = np.array([[np.cos(om), np.sin(om), 0], [-np.sin(om), np.cos(om), 0], [0, 0, 1]])
        moved = boxes.copy()
        moved[:, :3] = np.dot(boxes[:, :3], rot)
        moved[:, 6] += om
        free = bev_overlap(np.concatenate(placed, 0), moved).sum(0) == 0
        moved = moved[free]
        out_boxes.append(moved)
        placed.append(moved)
        inst = pts[points_in_any_box(pts, boxes[free], 0)]
        new = np.zeros_like(inst)
        new[:, :3] = np.dot(inst[:, :3], rot)
        new[:, 3] = inst[:, 3]
        out_pts.append(new)
    return np.concatenate(out_pts, 0), np.concatenate(out_boxes, 0)


def polarmix_sectors(degree, train_percent, update_methods, rng):
    """Sector list of inter_domain_point_polarmix (:248-286) - same draws, same wrap handling."""
    if isinstance(degree, float):
        deg = [degree, degree]
    else:
        deg = [degree[0], degree[0]] if len(degree) == 1 else [degree[0], degree[1]]
    sectors = []
    for method in update_methods:
        if method == "FIX":
            width = deg[0]
        elif method == "RAND":
            width = rng.uniform(deg[0], deg[1])
        elif method == "ASC":
            width = deg[0] + (deg[1] - deg[0]) * train_percent
        elif method == "ASC_SIG":
            width = deg[0] + (deg[1] - deg[0]) * (1 / (1 + np.exp(-6 * (train_percent * 2 - 1))))
        elif method == "DESC":
            width = deg[1] - (deg[1] - deg[0]) * train_percent
        have = len(sectors)
        for _ in range(100):
            st = (rng.random() * 2 - 1) * np.pi
            a, b = st, st + width
            clash = False
            for i in range(have):
                p, q = sorted(sectors[i])
                lo_, hi_ = (a, b) if a <= b else (b, a)
                clash = not (q < lo_ or hi_ < p)
                if clash:
                    break
            if not clash:
                sectors.append([a, b])
                break
        for i in range(len(sectors)):
            if sectors[i][1] > np.pi:
                sectors.append([-np.pi, sectors[i][1] - (np.pi * 2)])
                sectors[i][1] = np.pi
    return sectors


def polarmix(source, target, rot_copy_num, degree, train_percent, update_methods, inc_method="center", rng=np.random,
             polar_dis="FULL", use_pitch=False, pc_range=None):
    """polar_dis = "RAND": the reference's own call of swap_with_range carries a keyword that function does not take
    (:215-220, a TypeError as shipped); this is that call without the stray keyword."""
    sectors = polarmix_sectors(degree, train_percent, update_methods, rng)
    omegas = [0, rng.random() * np.pi * 2 / 3, (rng.random() + 1) * np.pi * 2 / 3][:rot_copy_num]
    pts, boxes = source["points"], source["gt_boxes"]
    rng.random()                                    # the reference's `if np.random.random() < 1.0` (swap branch)
    for lo, hi in sectors:
        if polar_dis == "FULL":
            pts, boxes = polar_swap(pts, target["points"], lo, hi, boxes, target["gt_boxes"], inc_method, use_pitch)
        elif polar_dis == "RAND":
            pts, boxes = polar_swap_with_range(pts, target["points"], lo, hi, boxes, target["gt_boxes"], pc_range, rng)
        else:
            raise NotImplementedError(polar_dis)
    rng.random()                                    # ... and the rotate-paste branch
    new_pts, new_boxes = rotate_paste_candidates(target["points"], target["gt_boxes"], omegas, boxes)
    pts = drop_points_in_boxes(pts, new_boxes[:, :7])
    return {"points": np.concatenate([pts, new_pts], 0), "gt_boxes": np.concatenate([boxes, new_boxes], 0)}


# ------------------------------------------------------------------------------------ LaserMix
def _wrap(yaw, phase):
    y = (yaw + F32(phase)).astype(F32)
    y[y > F32(3.141592)] -= F32(6.283184)
    y[y < F32(-3.141592)] += F32(6.283184)
    return y


def _clip_range(x, y, r_max):
    hi = F32(r_max) - F32(1e-05)
    return np.clip(np.sqrt(x ** 2 + y ** 2), F32(1e-05), hi)


def lasermix_cyc(source, target, num_areas, num_angles, pc_range, inc_method="center", rng=np.random):
    pc_range = np.asarray(pc_range, F32)
    phase = rng.uniform(-3.141592, 3.141952)
    dis_edges = np.linspace(0, pc_range[3], num_areas + 1)
    yaw_edges = np.linspace(-np.pi, np.pi, num_angles + 1)
    dom = []
    for d in (source, target):
        p, b = d["points"], d["gt_boxes"]
        cor = boxes_to_corners(b)[:, :, :2] if len(b) else np.zeros((0, 8, 2), F32)
        dom.append(dict(
            pts=p, box=b,
            yaw_p=_wrap(yaw32(p[:, 0], p[:, 1]), phase), dis_p=_clip_range(p[:, 0], p[:, 1], pc_range[3]),
            yaw_b=_wrap(yaw32(b[:, 0], b[:, 1]), phase), dis_b=_clip_range(b[:, 0], b[:, 1], pc_range[3]),
            yaw_c=_wrap(yaw32(cor[:, :, 0], cor[:, :, 1]), phase), dis_c=_clip_range(cor[:, :, 0], cor[:, :, 1], pc_range[3])))
    start = rng.choice([0, 1])
    out_pts, out_box = [], []
    for i in range(num_angles):
        pick = i % 2 + start
        for j in range(num_areas):
            d = dom[pick % 2]
            ylo, yhi, dlo, dhi = yaw_edges[i], yaw_edges[i + 1], dis_edges[j], dis_edges[j + 1]
            in_p = (d["yaw_p"] > ylo) & (d["yaw_p"] <= yhi) & (d["dis_p"] > dlo) & (d["dis_p"] <= dhi)
            if inc_method == "center":
                in_b = (d["yaw_b"] > ylo) & (d["yaw_b"] <= yhi) & (d["dis_b"] > dlo) & (d["dis_b"] <= dhi)
                out_pts.append(d["pts"][in_p])
                out_box.append(d["box"][in_b])
            elif inc_method == "corner_del":
                ycell = (d["yaw_c"] > ylo) & (d["yaw_c"] <= yhi)
                dcell = (d["dis_c"] > dlo) & (d["dis_c"] <= dhi)
                partial = (ycell.any(1) != ycell.all(1)) | (dcell.any(1) != dcell.all(1))
                out_box.append(d["box"][ycell.all(1) & dcell.all(1)])
                out_pts.append(drop_points_in_boxes(d["pts"][in_p], d["box"][partial][:, :7]))
            else:
                raise NotImplementedError(inc_method)
            pick += 1
    return {"points": np.concatenate(out_pts, 0), "gt_boxes": np.concatenate(out_box, 0)}


def lasermix_sph(source, target, pitch_angles, num_areas, order=0, rng=np.random):
    """laser_mix_transform_sph (:22-85): elevation bands (degrees, top to bottom) alternate between the two scenes, band i
    from `source` when i % 2 == order.  Elevation = arctan2(z - 1.8, range) in radians, clipped against the DEGREE bounds
    +- 1e-5 (as the reference does: only the upper bound ever acts), band edges compared in fp64 (numpy promotes an fp32
    array against np.float64 scalars).  The entry point passes inc_method where `order` is expected (:186-192): a string
    never equals i % 2, so every band comes from the target - restated as is."""
    lo, hi = F32(pitch_angles[0] + 1e-5), F32(pitch_angles[1] - 1e-5)

    def elevation(x, y, z):
        return np.clip(pitch32(F32(-1.8) + np.asarray(z, F32), _range32(x, y), sign=+1.0), lo, hi)

    sp, sb, tp, tb = source["points"], source["gt_boxes"], target["points"], target["gt_boxes"]
    ev = {"sp": elevation(sp[:, 0], sp[:, 1], sp[:, 2]), "sb": elevation(sb[:, 0], sb[:, 1], sb[:, 2]),
          "tp": elevation(tp[:, 0], tp[:, 1], tp[:, 2]), "tb": elevation(tb[:, 0], tb[:, 1], tb[:, 2])}
    n_bands = rng.choice(num_areas, size=1)[0]
    edges = np.linspace(pitch_angles[1], pitch_angles[0], n_bands + 1)
    out_pts, out_box = [], []
    for i in range(n_bands):
        a, b = edges[i + 1] / 180 * np.pi, edges[i] / 180 * np.pi
        if i % 2 == order:
            out_pts.append(sp[(ev["sp"] > a) & (ev["sp"] <= b)])
            out_box.append(sb[(ev["sb"] > a) & (ev["sb"] <= b)])
        else:
            out_pts.append(tp[(ev["tp"] > a) & (ev["tp"] <= b)])
            out_box.append(tb[(ev["tb"] > a) & (ev["tb"] <= b)])
    return {"points": np.concatenate(out_pts, 0), "gt_boxes": np.concatenate(out_box, 0)}


# --------------------------------------------------------------------------------- pseudo mixes
def pseudobbox(source, target):
    """inter_domain_point_pseudobbox (inter_domain_point_pseudomix.py:19-47): the target's boxes that touch no source box in
    BEV are pasted, with their points, into the source scene (whose points inside those boxes go)."""
    sp, sb, tp, tb = source["points"], source["gt_boxes"], target["points"], target["gt_boxes"]
    free = bev_overlap(sb, tb).sum(0) == 0
    paste = tb[free]
    pts = np.concatenate([sp[~points_in_any_box(sp, paste, 0)], tp[points_in_any_box(tp, paste, 0)]], 0)
    return {"points": pts, "gt_boxes": np.concatenate([sb, paste], 0)}


def pseudobackground(source, target):
    """inter_domain_point_pseudobackground (:49-68): the source's objects (points inside its boxes) on the target's
    background (points outside the target's boxes); boxes = the source's."""
    sp, sb, tp, tb = source["points"], source["gt_boxes"], target["points"], target["gt_boxes"]
    pts = np.concatenate([sp[points_in_any_box(sp, sb, 0)], tp[~points_in_any_box(tp, tb, 0)]], 0)
    return {"points": pts, "gt_boxes": sb}


# --------------------------------------------------------------------------------------- MixUp
def mixup(d1, d2, alpha, collision=False, rng=np.random):
    lam = rng.beta(alpha, alpha)
    p2, keep2 = d2["points"], d2["gt_boxes"]
    if collision:
        try:
            iou = bev_overlap(d1["gt_boxes"], d2["gt_boxes"])
            worst = iou.max(axis=0)
            keep2 = d2["gt_boxes"][worst == 0]
            gone = d2["gt_boxes"][worst > 0]
            if len(gone):
                p2 = p2[~points_in_any_box(p2, gone, 1)]
        except ValueError:
            keep2 = d2["gt_boxes"]
    p1 = d1["points"][rng.permutation(d1["points"].shape[0])]
    p2 = p2[rng.permutation(p2.shape[0])]
    pts = np.concatenate([p1[:int(p1.shape[0] * lam)], p2[:int(p2.shape[0] * (1 - lam))]], 0)
    return {"points": pts, "gt_boxes": np.concatenate([d1["gt_boxes"], keep2], 0)}
