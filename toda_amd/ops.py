"""Torch-facing operators over libtoda_hip.so: tensors in, tensors out, autograd where the
reference differentiates (spconv convs, .dense(), MeanVFE).  Everything here runs on the GPU
through the C ABI; host tensors raise (toda_amd.lib.ptr).
"""
import numpy as np
import torch

from . import arena as _arena
from . import lib as L


def _triple(v):
    if isinstance(v, (list, tuple)):
        assert len(v) == 3
        return [int(x) for x in v]
    return [int(v)] * 3


def grid_size_xyz(pc_range, voxel_size):
    """reference data_processor.py:117-118: round((hi - lo) / voxel) in float64, xyz order."""
    r = np.asarray(pc_range, dtype=np.float64)
    return np.round((r[3:6] - r[0:3]) / np.asarray(voxel_size, dtype=np.float64)).astype(np.int64)


# ----------------------------------------------------------------------------- voxelisation
import os as _os
import os as _os_early
import time as _time

_COUNT_PINNED = {}
_LAST_WAIT_S = 0.0


def read_counts(counts_dev, while_waiting=None):
    """A small int32 device tensor -> python list: the host sync of the index building.  The copy goes to pinned memory and the
    host waits for the COPY's event, not for the stream (work queued by `while_waiting` in between keeps the GPU busy meanwhile).
    On a side stream (the input pipeline) the host sleeps on the event instead of spinning: the wait is off the training stream
    there, and a thread that spins for milliseconds every step eats into the process's CPU quota."""
    dev = counts_dev.device
    cur = torch.cuda.current_stream(dev)
    side = cur != torch.cuda.default_stream(dev)
    # one pinned buffer + event per STREAM (and size): two input pipelines alive together - train and eval prefetchers, or a
    # while_waiting callback that comes back here on another stream - must not share a host buffer
    key = (dev.index, counts_dev.numel(), int(cur.cuda_stream))
    ent = _COUNT_PINNED.get(key)
    if ent is None:
        ent = _COUNT_PINNED[key] = (torch.empty((counts_dev.numel(),), dtype=torch.int32).pin_memory(), torch.cuda.Event(blocking=side))
    host, ev = ent
    host.copy_(counts_dev, non_blocking=True)
    ev.record()
    if while_waiting is not None:
        while_waiting()
    if side:
        # hipEventSynchronize spins even on a "blocking" event here (the thread's CPU time did not move): poll and sleep instead.
        # The wait sits under a whole backward pass, 50 us of granularity cost nothing.
        global _LAST_WAIT_S
        t_w = _time.perf_counter()
        while not ev.query():
            _time.sleep(5e-5)
        _LAST_WAIT_S = _time.perf_counter() - t_w
    else:
        ev.synchronize()
    vals = host.tolist()
    # the step's one natural look at the device fault word (host-mapped, no sync): a bounded inter-workgroup wait that gave up
    L.check(L.load().toda_device_fault(), "toda_device_fault")
    return vals


def _alloc(shape, dtype, device):
    """Output of an index builder: from the current thread's arena slot when the input pipeline set one (toda_amd.arena),
    else from torch's allocator."""
    slot = _arena.current_slot()
    if slot is not None:
        return slot.take(shape, dtype)
    return torch.empty(shape, dtype=dtype, device=device)


def _persistent(name, meta, nbytes, device, init=None):
    """(buffer, fresh): a buffer whose CONTENTS carry over between uses when an arena slot is current (fresh = just created
    and `init`-ialised), else a new one every time (fresh = True, `init` run)."""
    slot = _arena.current_slot()
    if slot is not None:
        return slot.persistent(name, meta, nbytes, init)
    buf = torch.empty((max(int(nbytes), 256),), dtype=torch.uint8, device=device)
    if init is not None:
        init(buf)
    return buf, True


def _cloud_list(points_list):
    """[(tensor, first-column offset in floats, n, c, row stride in floats)] of per-sample [n, c] clouds."""
    out = []
    for p in points_list:
        assert p.dtype == torch.float32 and p.dim() == 2
        p = p.contiguous()
        out.append((p, 0, p.shape[0], p.shape[1], p.shape[1]))
    return out


def voxelize_enqueue(clouds, pc_range, voxel_size, max_pts, max_voxels, counts=None, tag=0):
    """Voxelise + collate a whole batch in three launches, NO host sync (toda_voxelize_batch).  clouds: see _cloud_list (or
    rows of one [sum N, 1 + C] batch tensor read in place).  Returns cap-sized (voxels, coords_bzyx, num_points) and the device
    counts [B + 1] = M_0 .. M_{B-1}, sum M (written into `counts[:B + 1]` when given)."""
    lib = L.load()
    bsz = len(clouds)
    dev = clouds[0][0].device
    c, stride = clouds[0][3], clouds[0][4]
    assert all(cl[3] == c and cl[4] == stride for cl in clouds)
    ns = [int(cl[2]) for cl in clouds]
    slot = _arena.current_slot()
    # per-sample capacity of the workspace layout: sticky per slot, so that batches of different sizes (C5: 180 k / 35 k
    # clouds) keep ONE layout and the hash table stays clean between calls
    n_cap = max(ns + [1])
    if slot is not None:
        n_cap = max(n_cap, slot.state.get(("vox_ncap", tag), 0))
        n_cap = (n_cap + 16383) // 16384 * 16384
        slot.state[("vox_ncap", tag)] = n_cap
    ws_bytes = lib.toda_voxelize_batch_workspace_bytes(bsz, n_cap)
    ws, fresh = _persistent(("vox_ws", tag), (bsz, n_cap), ws_bytes, dev)
    rows = sum(min(int(max_voxels), n) for n in ns)
    voxels = _alloc((max(rows, 1), int(max_pts), c), torch.float32, dev)
    coords = _alloc((max(rows, 1), 4), torch.int32, dev)
    num = _alloc((max(rows, 1),), torch.int32, dev)
    if counts is None:
        counts = _alloc((bsz + 1,), torch.int32, dev)
    rng, vs = L.host_f32(pc_range), L.host_f32(voxel_size)
    grid = L.host_i32(grid_size_xyz(pc_range, voxel_size))
    addrs = L.host_addrs([cl[0].data_ptr() + 4 * cl[1] for cl in clouds])
    rc = lib.toda_voxelize_batch(addrs, L.hptr(L.host_i32(ns)), bsz, n_cap, c, stride, L.hptr(rng), L.hptr(vs), L.hptr(grid),
                                 int(max_pts), int(max_voxels), L.ptr(voxels), L.ptr(coords), L.ptr(num), L.ptr(counts),
                                 L.ptr(ws), ws.numel(), 0 if fresh else 1, L.stream())
    L.check(rc, "toda_voxelize_batch")
    return voxels, coords, num, counts


def voxelize_raw(points, pc_range, voxel_size, max_pts, max_voxels):
    """One sample through the single-sample entry point (spconv's own layout: coords (z, y, x)), no host sync.  Returns
    buffers sized for the cap and the device-side count."""
    lib = L.load()
    assert points.dtype == torch.float32 and points.dim() == 2
    n, c = points.shape
    dev = points.device
    cap = int(min(max_voxels, max(n, 1)))
    voxels = torch.empty((cap, max_pts, c), dtype=torch.float32, device=dev)
    coords = torch.empty((cap, 3), dtype=torch.int32, device=dev)
    num = torch.empty((cap,), dtype=torch.int32, device=dev)
    m_dev = torch.zeros((1,), dtype=torch.int32, device=dev)
    ws_bytes = lib.toda_voxelize_workspace_bytes(n, cap)
    ws = torch.empty((ws_bytes,), dtype=torch.uint8, device=dev)
    rng, vs = L.host_f32(pc_range), L.host_f32(voxel_size)
    grid = L.host_i32(grid_size_xyz(pc_range, voxel_size))
    rc = lib.toda_voxelize_hard(L.ptr(points.contiguous()), n, c, L.hptr(rng), L.hptr(vs), L.hptr(grid), int(max_pts),
                                cap, L.ptr(voxels), L.ptr(coords), L.ptr(num), L.ptr(m_dev), L.ptr(ws), ws_bytes,
                                L.stream())
    L.check(rc, "toda_voxelize_hard")
    return voxels, coords, num, m_dev


def voxelize(points, pc_range, voxel_size, max_pts, max_voxels):
    """One sample -> (voxels [M,P,C], coords_zyx [M,3] int32, num_points [M] int32)."""
    voxels, coords, num, m_dev = voxelize_raw(points, pc_range, voxel_size, max_pts, max_voxels)
    m = int(m_dev.item())
    return voxels[:m], coords[:m], num[:m]


def voxelize_batch(points_list, pc_range, voxel_size, max_pts, max_voxels):
    """All samples of a batch in three launches and ONE host sync.  Mirrors voxelise + collate_batch
    (reference dataset.py:161-178): coords gain the batch column -> (b, z, y, x)."""
    if len(points_list) > 32:      # the kernel's sample table; larger batches in groups
        parts = [voxelize_batch(points_list[i:i + 32], pc_range, voxel_size, max_pts, max_voxels) for i in range(0, len(points_list), 32)]
        for g, (_, cb, _) in enumerate(parts):
            cb[:, 0] += 32 * g
        return tuple(torch.cat([p[k] for p in parts]) for k in range(3))
    voxels, coords, num, counts = voxelize_enqueue(_cloud_list(points_list), pc_range, voxel_size, max_pts, max_voxels)
    m = read_counts(counts)[-1]      # the one sync
    return voxels[:m], coords[:m], num[:m]


class _MeanVFE(torch.autograd.Function):
    @staticmethod
    def forward(ctx, voxels, num_points):
        lib = L.load()
        m, p, c = voxels.shape
        voxels = voxels.contiguous()
        num_f = num_points.to(torch.float32).contiguous()
        out = torch.empty((m, c), dtype=torch.float32, device=voxels.device)
        L.check(lib.toda_mean_vfe_fwd(L.ptr(voxels), L.ptr(num_f), m, p, c, L.ptr(out), L.stream()), "toda_mean_vfe_fwd")
        ctx.save_for_backward(num_f)
        ctx.shape = (m, p, c)
        return out

    @staticmethod
    def backward(ctx, gout):
        (num_f,) = ctx.saved_tensors
        m, p, c = ctx.shape
        g = torch.empty((m, p, c), dtype=torch.float32, device=gout.device)
        L.check(L.load().toda_mean_vfe_bwd(L.ptr(gout.contiguous()), L.ptr(num_f), m, p, c, L.ptr(g), L.stream()),
                "toda_mean_vfe_bwd")
        return g, None


def mean_vfe(voxels, num_points):
    return _MeanVFE.apply(voxels, num_points)


# ------------------------------------------------------------------------------- rulebooks
class GridIndex:
    """Bitmap + rank dictionary of one sparse level (device workspace)."""

    def __init__(self, batch, shape, device, buf=None):
        self.batch, self.shape = int(batch), [int(s) for s in shape]
        self._shape_c = L.host_i32(self.shape)
        nbytes = L.load().toda_gridindex_bytes(self.batch, L.hptr(self._shape_c))
        self.nbytes = nbytes
        self.buf = buf if buf is not None else torch.empty((nbytes,), dtype=torch.uint8, device=device)
        self.rowof = None  # None = rows already in canonical order

    @classmethod
    def from_coords(cls, indices, batch, shape):
        """Canonical ranks (popcount scan over the lattice): for coordinate lists whose index is also DECODED or ranked in
        canonical order (lazy single-layer builds)."""
        gi = cls(batch, shape, indices.device)
        n = indices.shape[0]
        gi.rowof = torch.empty((max(n, 1),), dtype=torch.int32, device=indices.device)
        rc = L.load().toda_gridindex_from_coords(L.ptr(indices), n, None, gi.batch, L.hptr(gi._shape_c), L.ptr(gi.buf),
                                                 L.ptr(gi.rowof), L.stream())
        L.check(rc, "toda_gridindex_from_coords")
        return gi

    @classmethod
    def unordered_begin(cls, batch, shape, device, name="gi0"):
        """First half of from_coords_unordered: fetch the bitmap - inside an arena slot a persistent buffer that the slot keeps
        all-zero between uses (the previous use's marks are taken off when the slot is re-acquired)."""
        lib = L.load()
        shape_c = L.host_i32([int(v) for v in shape])
        nbytes = lib.toda_gridindex_bytes(int(batch), L.hptr(shape_c))
        slot = _arena.current_slot()
        buf, fresh = _persistent(name, (int(batch), tuple(int(v) for v in shape)), nbytes, device)
        gi = cls(batch, shape, device, buf=buf)
        gi._name, gi._clean = name, 0
        if slot is not None and not fresh and name in slot.clean:       # ArenaSlot.reset ran the previous use's toda_gridindex_clear
            slot.clean.discard(name)
            gi._clean = 1
        return gi

    def unordered_build(self, indices, n_dev=None):
        n = indices.shape[0]
        self.rowof = _alloc((max(n, 1),), torch.int32, indices.device)
        rc = L.load().toda_gridindex_from_coords_unordered(L.ptr(indices), n, L.ptr(n_dev), self.batch, L.hptr(self._shape_c), L.ptr(self.buf),
                                                           L.ptr(self.rowof), self._clean, L.stream())
        L.check(rc, "toda_gridindex_from_coords_unordered")
        slot = _arena.current_slot()
        if slot is not None:
            # how to clean up: the listed words back to zero.  `indices` / `n_dev` are views into this slot's chunks, intact until
            # the next use carves them again - ArenaSlot.reset enqueues this before that.
            lib, buf, bt, shp = L.load(), self.buf, self.batch, self._shape_c

            def clear():
                L.check(lib.toda_gridindex_clear(L.ptr(indices), n, L.ptr(n_dev), bt, L.hptr(shp), L.ptr(buf), L.stream()), "toda_gridindex_clear")

            slot.marks[self._name] = clear
        return self

    @classmethod
    def from_coords_unordered(cls, indices, batch, shape, n_dev=None, name="gi0"):
        """The voxel level in O(sites) (toda_gridindex_from_coords_unordered): no memset, no scan over the lattice.  Inside an
        arena slot the bitmap is a persistent buffer kept all-zero between uses: the words the PREVIOUS build of this slot
        marked are cleared first, from the coordinate list that build left behind."""
        return cls.unordered_begin(batch, shape, indices.device, name).unordered_build(indices, n_dev)


class Rulebook:
    """Neighbour tables of one indice_key (what spconv keeps in SparseConvTensor.indice_dict)."""

    def __init__(self, kind, ksize, n_in, n_out, nbr_fwd, nbr_bwd, flip_bwd, pair_cnt, **geom):
        self.kind = kind  # 'subm' | 'conv'
        self.ksize = ksize
        self.k_vol = int(ksize[0] * ksize[1] * ksize[2])
        self.n_in, self.n_out = n_in, n_out
        self.nbr_fwd = nbr_fwd    # [K, n_out]: input row per output row and offset
        self.nbr_bwd = nbr_bwd    # [K, n_in]: output row per input row and offset (dgrad)
        self.flip_bwd = flip_bwd  # SubM: dgrad reuses nbr_fwd with the offsets reversed
        self.pair_cnt = pair_cnt  # [K] int32 on device
        # the tables carry their counters (python attribute, no device work): what sees only a table - bench.py's launch records -
        # can still count its pairs without reading the table itself, which may sit in a recycled arena slot by then
        nbr_fwd._toda_pair_cnt = pair_cnt
        if nbr_bwd is not nbr_fwd:
            nbr_bwd._toda_pair_cnt = pair_cnt
        self.geom = geom
        self.in_indices = None    # strided conv: coordinates of its input sites (rows of the data-gradient table)
        self._inverse = None
        self._class_order = None

    def class_order(self):
        """(order, cls_sorted) of the data-gradient table of a strided convolution: its rows regrouped by the residue class of
        (coordinate + padding) mod stride, which fixes the 1..8 kernel offsets a row can have (toda_rulebook_class_order).
        None when not applicable."""
        if (not CLASS_DGRAD or self.kind != "conv" or self.in_indices is None or self.k_vol > 27 or self.k_vol < 2
                or self.n_in < CLASS_DGRAD_MIN_ROWS or any(int(v) not in (1, 2) for v in self.geom["stride"])):
            return None
        if self._class_order is None:
            lib = L.load()
            dev = self.in_indices.device
            order = _alloc((self.n_in,), torch.int32, dev)
            cls = _alloc((self.n_in,), torch.uint8, dev)
            st, pd = L.host_i32(self.geom["stride"]), L.host_i32(self.geom["padding"])
            L.check(lib.toda_rulebook_class_order(L.ptr(self.in_indices), self.n_in, L.hptr(st), L.hptr(pd), L.ptr(order), L.ptr(cls), L.stream()),
                    "toda_rulebook_class_order")
            self._class_order = (order, cls)
        return self._class_order

    def num_pairs(self):
        return int(self.pair_cnt.sum().item())

    def inverse(self):
        """The same pairs read the other way round (spconv's SparseInverseConv3d, reference spconv_backbone.py:16-17,22-23): rows of the
        strided convolution's OUTPUT set are gathered, rows of its INPUT set are produced, offset k keeps its weight slice.  The
        i2o table is the forward table of that layer, o2i the table of its data gradient; no new index work."""
        if self.kind != "conv":
            raise ValueError("only the tables of a strided SparseConv3d have an inverse")
        if self._inverse is None:
            inv = Rulebook("inverse", self.ksize, self.n_out, self.n_in, self.nbr_bwd, self.nbr_fwd, False, self.pair_cnt, **self.geom)
            self._inverse = inv
        return self._inverse


def conv_out_shape(shape, ksize, stride, padding):
    return [(int(s) + 2 * p - k) // t + 1 for s, k, t, p in zip(shape, ksize, stride, padding)]


def build_subm_rulebook(indices, batch, shape, ksize=3, dilation=1, grid_index=None, pair_cnt=None):
    """pair_cnt: a zeroed [K] int32 view to count into (the plan zeroes every table's counters with one fill)."""
    lib = L.load()
    ks, dl = _triple(ksize), _triple(dilation)
    indices = indices.contiguous()
    assert indices.dtype == torch.int32 and indices.shape[1] == 4
    n = indices.shape[0]
    if grid_index is None:
        grid_index = GridIndex.from_coords_unordered(indices, batch, shape, name=("gi_lazy", tuple(int(v) for v in shape)))
    K = ks[0] * ks[1] * ks[2]
    nbr = _alloc((K, n), torch.int32, indices.device)
    zeroed = pair_cnt is not None
    cnt = pair_cnt if zeroed else torch.empty((K,), dtype=torch.int32, device=indices.device)
    ks_c, dl_c = L.host_i32(ks), L.host_i32(dl)
    rc = lib.toda_rulebook_subm(L.ptr(indices), n, int(batch), L.hptr(grid_index._shape_c), L.hptr(ks_c), L.hptr(dl_c),
                                L.ptr(grid_index.buf), L.ptr(grid_index.rowof), L.ptr(nbr), L.ptr(cnt), int(zeroed), L.stream())
    L.check(rc, "toda_rulebook_subm")
    return Rulebook("subm", ks, n, n, nbr, nbr, True, cnt, dilation=dl), grid_index


def _conv_tables(indices, n_in, idx_out, n_out, batch, shape, out_shape, ks, st, pd, gi_out, gi_in=None, pair_cnt=None):
    """Both neighbour tables of a strided convolution whose output set is known (toda_rulebook_conv)."""
    lib = L.load()
    dev = indices.device
    K = ks[0] * ks[1] * ks[2]
    o2i = _alloc((K, n_out), torch.int32, dev)
    i2o = _alloc((K, n_in), torch.int32, dev)
    zeroed = pair_cnt is not None
    cnt = pair_cnt if zeroed else torch.empty((K,), dtype=torch.int32, device=dev)
    hs = [L.host_i32(v) for v in (shape, ks, st, pd, out_shape)]
    rc = lib.toda_rulebook_conv(L.ptr(indices), n_in, int(batch), L.hptr(hs[0]), L.hptr(hs[1]), L.hptr(hs[2]), L.hptr(hs[3]),
                                L.hptr(hs[4]), L.ptr(gi_out.buf), n_out, L.ptr(o2i), L.ptr(i2o), L.ptr(cnt),
                                L.ptr(idx_out) if gi_in is not None else None, L.ptr(gi_in.buf) if gi_in is not None else None,
                                L.ptr(gi_in.rowof) if gi_in is not None else None, int(zeroed), L.stream())
    L.check(rc, "toda_rulebook_conv")
    rb = Rulebook("conv", ks, n_in, n_out, o2i, i2o, False, cnt, stride=st, padding=pd, in_shape=[int(v) for v in shape])
    rb.in_indices = indices
    return rb


def _conv_cap(n_upper, batch, out_shape, ks, st):
    """Upper bound of a strided convolution's output rows: every input reaches at most prod(ceil(k/s)) outputs; also the lattice."""
    per_in = 1
    for k, s_ in zip(ks, st):
        per_in *= -(-k // s_)
    return int(min(n_upper * per_in, batch * out_shape[0] * out_shape[1] * out_shape[2]))


def build_conv_rulebook(indices, batch, shape, ksize, stride, padding):
    """Strided sparse conv: returns (out_indices, out_shape, rulebook, grid index of the output set)."""
    lib = L.load()
    ks, st, pd = _triple(ksize), _triple(stride), _triple(padding)
    indices = indices.contiguous()
    assert indices.dtype == torch.int32 and indices.shape[1] == 4
    n_in = indices.shape[0]
    shape = [int(s) for s in shape]
    out_shape = conv_out_shape(shape, ks, st, pd)
    dev = indices.device
    gi_out = GridIndex(batch, out_shape, dev)
    cap = _conv_cap(n_in, batch, out_shape, ks, st)
    idx_out = torch.empty((max(cap, 1), 4), dtype=torch.int32, device=dev)
    n_out_dev = torch.zeros((1,), dtype=torch.int32, device=dev)
    shi, sho = L.host_i32(shape), L.host_i32(out_shape)
    ks_c, st_c, pd_c = L.host_i32(ks), L.host_i32(st), L.host_i32(pd)
    rc = lib.toda_gridindex_from_conv(L.ptr(indices), n_in, None, int(batch), L.hptr(shi), L.hptr(ks_c), L.hptr(st_c),
                                      L.hptr(pd_c), L.hptr(sho), L.ptr(gi_out.buf), L.ptr(idx_out), L.ptr(n_out_dev),
                                      cap, L.stream())
    L.check(rc, "toda_gridindex_from_conv")
    n_out = int(n_out_dev.item())  # host sync: table sizes depend on it
    if n_out > cap:
        raise RuntimeError(f"strided rulebook: {n_out} outputs exceed the bound {cap}")
    idx_out = idx_out[:n_out]
    rb = _conv_tables(indices, n_in, idx_out, n_out, batch, shape, out_shape, ks, st, pd, gi_out)
    return idx_out, out_shape, rb, gi_out


def _plan_enqueue(level0, batch, shape, steps, counts, c0, gi0=None):
    """Phase A of an index plan, no host sync: the voxel level's index (O(sites)) and the output index set of every strided
    convolution, each derived from the bitmap of the level before it (toda_gridindex_from_bitmap: output stationary, no
    atomics, the input row count never enters).  level0 = (indices [n_upper, 4], n_upper, n_dev or None).  counts: int32 device
    vector, slot c0 + l receives the row count of the l-th strided level.  Returns the level list."""
    lib = L.load()
    idx0, n_upper, n_dev = level0
    dev = idx0.device
    slot = _arena.current_slot()
    tag = slot.plans if slot is not None else 0
    if gi0 is None:
        gi0 = GridIndex.unordered_begin(batch, shape, dev, name=("gi0", tag))
    gi0.unordered_build(idx0, n_dev)
    levels = [{"idx": idx0, "n_upper": n_upper, "shape": [int(v) for v in shape], "gi": gi0}]
    li = 0
    for st in steps:
        if st["kind"] != "conv":
            continue
        cur = levels[-1]
        ks, sd, pd = _triple(st["ksize"]), _triple(st["stride"]), _triple(st["padding"])
        out_shape = conv_out_shape(cur["shape"], ks, sd, pd)
        cap = _conv_cap(cur["n_upper"], batch, out_shape, ks, sd)
        li += 1
        shape_c = L.host_i32(out_shape)
        buf, _ = _persistent(("gi", tag, li), (int(batch), tuple(out_shape)), lib.toda_gridindex_bytes(int(batch), L.hptr(shape_c)), dev)
        gi_out = GridIndex(batch, out_shape, dev, buf=buf)
        idx_out = _alloc((max(cap, 1), 4), torch.int32, dev)
        hs = [L.host_i32(v) for v in (cur["shape"], ks, sd, pd, out_shape)]
        rc = lib.toda_gridindex_from_bitmap(L.ptr(cur["gi"].buf), int(batch), L.hptr(hs[0]), L.hptr(hs[1]), L.hptr(hs[2]), L.hptr(hs[3]),
                                            L.hptr(hs[4]), L.ptr(gi_out.buf), L.ptr(idx_out), counts.data_ptr() + 4 * (c0 + li - 1),
                                            cap, 1 if li == 1 else 0, L.stream())      # the voxel level's index carries row marks
        L.check(rc, "toda_gridindex_from_bitmap")
        levels.append({"idx": idx_out, "n_upper": cap, "shape": out_shape, "gi": gi_out, "cap": cap, "step": st})
    if slot is not None:
        slot.plans += 1
    return levels


def _plan_tables(levels, level_counts, batch, steps, training):
    """Phase B: every neighbour table at its exact size.  level_counts[l] = rows of level l (l = 0: the voxel level)."""
    dev = levels[0]["idx"].device
    for lv, n in zip(levels, level_counts):
        if lv.get("cap") is not None and n > lv["cap"]:
            raise RuntimeError(f"strided rulebook {lv['step']['key']}: {n} outputs exceed the bound {lv['cap']}")
        lv["n"] = int(n)
        lv["idx"] = lv["idx"][:int(n)]
    keys = []
    for st in steps:
        if st["key"] not in keys:
            keys.append(st["key"])
    # the pair counters of every table: one fill.  NOT from the arena slot: they outlive the batch (bench.py's per-launch records sum
    # them after the timed region, when the slot's tables have long been overwritten by later batches) - 2 KB from torch's allocator
    cnts = torch.zeros((len(keys), 64), dtype=torch.int32, device=dev)
    out = {}
    li = 0
    for st in steps:
        cur = levels[li]
        if st["kind"] == "subm":
            if st["key"] in out:
                continue
            ks = _triple(st["ksize"])
            rb, gi = build_subm_rulebook(cur["idx"], batch, cur["shape"], st["ksize"], st.get("dilation", 1), grid_index=cur["gi"],
                                         pair_cnt=cnts[keys.index(st["key"])][:ks[0] * ks[1] * ks[2]])
            out[st["key"]] = {"kind": "subm", "rb": rb, "n_in": cur["n"]}
        else:
            nxt = levels[li + 1]
            ks, sd, pd = _triple(st["ksize"]), _triple(st["stride"]), _triple(st["padding"])
            rb = _conv_tables(cur["idx"], cur["n"], nxt["idx"], nxt["n"], batch, cur["shape"], nxt["shape"], ks, sd, pd, nxt["gi"],
                              gi_in=cur["gi"], pair_cnt=cnts[keys.index(st["key"])][:ks[0] * ks[1] * ks[2]])
            if training:
                rb.class_order()        # the data gradient's row order, here instead of on the training stream
            out[st["key"]] = {"kind": "conv", "rb": rb, "n_in": cur["n"], "out_indices": nxt["idx"],
                              "out_shape": nxt["shape"], "gi": nxt["gi"]}
            li += 1
    return out


def build_index_plan(indices, batch, shape, steps, while_waiting=None, training=None):
    """All rulebooks of a sequential sparse backbone with ONE host sync.

    while_waiting: optional callable run between the request for the level counts and the wait for them - work that does not
    depend on the plan (operand packing) is queued there, so the GPU has something to run while the host reads the counts and
    gets its launch lead back (the wait is for the copy's event, not for the stream).

    steps (forward order): {'kind': 'subm', 'key', 'ksize', 'dilation'} or
    {'kind': 'conv', 'key', 'ksize', 'stride', 'padding'}.  The output index sets of the strided
    convolutions are built back-to-back on the device (each from the bitmap of the level before it; buffers are sized by an
    upper bound), then the counts are read once and the neighbour tables are filled at their exact sizes.
    Returns {key: indice_dict entry}."""
    indices = indices.contiguous()
    n_conv = sum(1 for st in steps if st["kind"] == "conv")
    counts = _alloc((max(n_conv, 1),), torch.int32, indices.device)
    levels = _plan_enqueue((indices, indices.shape[0], None), batch, shape, steps, counts, 0)
    got = read_counts(counts, while_waiting) if n_conv else []
    if training is None:
        training = torch.is_grad_enabled()
    return _plan_tables(levels, [indices.shape[0]] + list(got[:n_conv]), batch, steps, training)


def build_input_plan(clouds, voxel_cfg, batch, shape, steps, training=True, while_waiting=None):
    """Voxelisation + collation + every rulebook of a sequential sparse backbone with ONE host sync (round 3: two, and two
    concatenations): the voxeliser's device-side counts feed the voxel level's index, each strided level follows from the
    bitmap before it, and one read-back returns [M_0 .. M_{B-1}, sum M, n_1 .. n_L].
    Returns (voxels [M, P, C], coords [M, 4] int32 (b, z, y, x), num_points [M] int32, plan)."""
    n_conv = sum(1 for st in steps if st["kind"] == "conv")
    bsz = len(clouds)
    dev = clouds[0][0].device
    slot = _arena.current_slot()
    gi0 = GridIndex.unordered_begin(batch, shape, dev, name=("gi0", slot.plans if slot is not None else 0))
    counts = _alloc((bsz + 1 + n_conv,), torch.int32, dev)
    voxels, coords, num, _ = voxelize_enqueue(clouds, voxel_cfg["point_cloud_range"], voxel_cfg["voxel_size"],
                                              voxel_cfg["max_points_per_voxel"], voxel_cfg["max_num_voxels"], counts=counts,
                                              tag=slot.plans if slot is not None else 0)
    levels = _plan_enqueue((coords, coords.shape[0], counts[bsz:bsz + 1]), batch, shape, steps, counts, bsz + 1, gi0=gi0)
    got = read_counts(counts, while_waiting)      # the one sync
    m = got[bsz]
    plan = _plan_tables(levels, [m] + list(got[bsz + 1:bsz + 1 + n_conv]), batch, steps, training)
    return voxels[:m], levels[0]["idx"], num[:m], plan


# --------------------------------------------------------------------------- sparse conv
MATRIX_PATHS = ("native", "split")
_MM = None          # the library's current path, read once (the library is only switched through set_matrix_path)


def matrix_path():
    """'native' (fp32 MFMA) or 'split' (exact bf16 hi/mid/lo split, 6 terms, fp32 accumulate) - toda_matrix_path; initial value TODA_MM."""
    global _MM
    if _MM is None:
        _MM = MATRIX_PATHS[L.load().toda_matrix_path()]
    return _MM


def set_matrix_path(name):
    """Switch the matrix path of the sparse gather-GEMMs.  Packed operands remember the path they were packed under (`_toda_mm`);
    the module caches (spconv/conv.py) are keyed on it, so the next forward re-packs."""
    global _MM
    L.check(L.load().toda_set_matrix_path(MATRIX_PATHS.index(name)), "toda_set_matrix_path")
    _MM = name


def _tag_mm(wp, cg, cp):
    wp._toda_mm = matrix_path() if L.load().toda_spconv_split_supported(int(cg), int(cp)) else None
    return wp


def _check_mm(wp):
    mm = getattr(wp, "_toda_mm", None)
    if mm is not None and mm != matrix_path():
        raise RuntimeError(f"packed sparse-conv operand was written under matrix path {mm!r}, the current path is {matrix_path()!r}: re-pack it")


def pack_weight(weight, transpose, flip_k):
    """weight [Cout, kz, ky, kx, Cin] -> MFMA fragment order (see csrc/spconv.hip)."""
    lib = L.load()
    cout, cin = weight.shape[0], weight.shape[-1]
    K = weight.numel() // (cout * cin)
    cg, cp = (cout, cin) if transpose else (cin, cout)
    n = lib.toda_spconv_packed_weight_floats(K, cg, cp)
    wp = torch.empty((n,), dtype=torch.float32, device=weight.device)
    rc = lib.toda_spconv_pack_weight(L.ptr(weight.contiguous()), cout, K, cin, int(transpose), int(flip_k), L.ptr(wp),
                                     L.stream())
    L.check(rc, "toda_spconv_pack_weight")
    return _tag_mm(wp, cg, cp)


def pack_weights_batched(items):
    """items: [(weight [Cout, kz, ky, kx, Cin], transpose, flip_k)] -> the packed operands, all written by ONE launch into
    one allocation (toda_spconv_pack_weights).  Used once per step for every sparse convolution of a backbone."""
    lib = L.load()
    if not items:
        return []
    dev = items[0][0].device
    sizes, couts, ks, cins, pairs = [], [], [], [], []
    for w, tr, _ in items:
        cout, cin = w.shape[0], w.shape[-1]
        K = w.numel() // (cout * cin)
        cg, cp = (cout, cin) if tr else (cin, cout)
        sizes.append(lib.toda_spconv_packed_weight_floats(K, cg, cp))
        couts.append(cout), ks.append(K), cins.append(cin), pairs.append((cg, cp))
    flat = torch.empty((sum(sizes),), dtype=torch.float32, device=dev)
    outs, off = [], 0
    for n, (cg, cp) in zip(sizes, pairs):
        outs.append(_tag_mm(flat[off:off + n], cg, cp))
        off += n
    ws = [w.contiguous() for w, _, _ in items]
    rc = lib.toda_spconv_pack_weights(len(items), L.host_ptrs(ws), L.hptr(L.host_i32(couts)), L.hptr(L.host_i32(ks)), L.hptr(L.host_i32(cins)),
                                      L.hptr(L.host_i32([int(bool(t)) for _, t, _ in items])), L.hptr(L.host_i32([int(bool(f)) for _, _, f in items])),
                                      L.host_ptrs(outs), L.stream())
    L.check(rc, "toda_spconv_pack_weights")
    return outs


def gather_gemm_stats_supported(c_gather, c_produce):
    return bool(L.load().toda_spconv_gather_gemm_stats_supported(int(c_gather), int(c_produce)))


def gather_gemm_with_stats(feat, wp, nbr, c_produce, bias=None, partials=False):
    """Forward gather-GEMM that also returns the BatchNorm moments of its output (sum, sum of squares per channel in
    sums[:2c], fp64) from the kernel's epilogue - the layout toda_bn_finalize reads.  partials=True: the per-workgroup partial
    sums are left unfolded and (out, sums, blocks) is returned for toda_bn_finalize_partials (one launch less per layer)."""
    lib = L.load()
    K, n_out = nbr.shape
    out = torch.empty((n_out, c_produce), dtype=torch.float32, device=feat.device)
    nd = lib.toda_spconv_gather_gemm_stats_doubles(n_out, c_produce)
    sums = torch.empty((nd,), dtype=torch.float64, device=feat.device)
    if partials:
        import ctypes
        blocks = ctypes.c_int(0)
        rc = lib.toda_spconv_gather_gemm_stats_partials(L.ptr(feat), feat.shape[0], feat.shape[1], L.ptr(wp), L.ptr(nbr), n_out, K, c_produce,
                                                        L.ptr(bias), L.ptr(out), L.ptr(sums), nd, ctypes.addressof(blocks), L.stream())
        L.check(rc, "toda_spconv_gather_gemm_stats_partials")
        return out, sums, int(blocks.value)
    rc = lib.toda_spconv_gather_gemm_stats(L.ptr(feat), feat.shape[0], feat.shape[1], L.ptr(wp), L.ptr(nbr), n_out, K, c_produce,
                                           L.ptr(bias), L.ptr(out), L.ptr(sums), nd, L.stream())
    L.check(rc, "toda_spconv_gather_gemm_stats")
    return out, sums


def gather_gemm(feat, wp, nbr, c_produce, bias=None, order=None):
    lib = L.load()
    K, n_out = nbr.shape
    out = torch.empty((n_out, c_produce), dtype=torch.float32, device=feat.device)
    rc = lib.toda_spconv_gather_gemm_ordered(L.ptr(feat), feat.shape[0], feat.shape[1], L.ptr(wp), L.ptr(nbr), n_out, K,
                                             c_produce, L.ptr(bias), L.ptr(out), L.ptr(order), L.stream())
    L.check(rc, "toda_spconv_gather_gemm_ordered")
    return out


def _refuse_retired_knobs():
    """Environment knobs of kernel families that were measured, lost and are no longer in the library (profiles/EXPERIMENTS.md), and of
    the measurement-only ablation builds: asking for one of them is an error, not a silent no-op."""
    retired = [k for k in ("TODA_GG_LINE", "TODA_HALO", "TODA_WG_TILE", "TODA_ROW_ORDER", "TODA_GG_WS", "TODA_GG_STAGE", "TODA_GG_WRES", "TODA_GG_LDS_PF",
                           "TODA_GG_BLK512", "TODA_GG_RT", "TODA_SPLIT_BLK", "TODA_SPLIT_KCS", "TODA_WINO_WGRAD") if _os.environ.get(k, "0") not in ("0", "")]
    if retired:
        raise RuntimeError(f"{', '.join(retired)}: these kernel variants were retired (profiles/EXPERIMENTS.md); unset the variable(s)")
    ablate = [k for k in ("TODA_GG_ABLATE", "TODA_WINO_ABLATE", "TODA_WINO_WG_ABLATE") if _os.environ.get(k, "0") not in ("0", "")]
    if ablate and not _os.environ.get("TODA_HIP_LIB"):
        raise RuntimeError(f"{', '.join(ablate)} only act in a measurement build (-DTODA_ABLATE=1, loaded through TODA_HIP_LIB): the library "
                           "that ships ignores them, and a run that believes it ablated something would report wrong conclusions")


_refuse_retired_knobs()


# narrow K = 27 layers (conv_input, the 16-channel SubM level, the strided 16 -> 32 and its data gradient) by per-offset compaction:
# measured on the C3 levels 5 -> 16 80 -> 42 us, 16 -> 16 70 -> 38, 16 -> 32 @ 682 k rows 141 -> 122, 32 -> 16 (dgrad) 69 -> 61
COMPACT = _os.environ.get("TODA_GG_COMPACT", "1") == "1"


def _compact_route(c_gather, c_produce, nbr):
    return (COMPACT and nbr.shape[0] == 27 and nbr.shape[1] > 0 and c_produce in (16, 32) and c_gather <= 32
            and (c_gather <= 16 or c_produce == 16))


def gather_gemm_compact_supported(c_gather, c_produce, k_vol):
    return bool(L.load().toda_spconv_gather_gemm_compact_supported(int(c_gather), int(c_produce), int(k_vol)))


def gather_gemm_compact(feat, weight, nbr, c_produce, bias=None, transpose=False, flip_k=False, stats=False):
    """Narrow K = 27 layers by per-offset compaction (toda_spconv_gather_gemm_compact); weight is the PLAIN [cout][3][3][3][cin] tensor.
    stats=True: also the BatchNorm moments of the output from the epilogue, unfolded: (out, sums, blocks) for toda_bn_finalize_partials."""
    lib = L.load()
    K, n_out = nbr.shape
    out = torch.empty((n_out, c_produce), dtype=torch.float32, device=feat.device)
    if stats:
        import ctypes
        nd = lib.toda_spconv_gather_gemm_compact_stats_doubles(n_out, c_produce)
        sums = torch.empty((nd,), dtype=torch.float64, device=feat.device)
        blocks = ctypes.c_int(0)
        rc = lib.toda_spconv_gather_gemm_compact_stats(L.ptr(feat), feat.shape[0], feat.shape[1], L.ptr(weight), weight.shape[0], weight.shape[-1],
                                                       int(bool(transpose)), int(bool(flip_k)), L.ptr(nbr), n_out, K, c_produce, L.ptr(bias), L.ptr(out),
                                                       L.ptr(sums), nd, ctypes.addressof(blocks), L.stream())
        L.check(rc, "toda_spconv_gather_gemm_compact_stats")
        return out, sums, int(blocks.value)
    rc = lib.toda_spconv_gather_gemm_compact(L.ptr(feat), feat.shape[0], feat.shape[1], L.ptr(weight), weight.shape[0], weight.shape[-1],
                                             int(bool(transpose)), int(bool(flip_k)), L.ptr(nbr), n_out, K, c_produce, L.ptr(bias), L.ptr(out), L.stream())
    L.check(rc, "toda_spconv_gather_gemm_compact")
    return out


def gather_gemm_classed(feat, wp, nbr, c_produce, order, cls_sorted, ksize, stride, padding):
    """Data gradient of a strided convolution over its class-sorted rows (Rulebook.class_order)."""
    lib = L.load()
    K, n_out = nbr.shape
    out = torch.empty((n_out, c_produce), dtype=torch.float32, device=feat.device)
    ks, st, pd = L.host_i32(ksize), L.host_i32(stride), L.host_i32(padding)
    rc = lib.toda_spconv_gather_gemm_classed(L.ptr(feat), feat.shape[0], feat.shape[1], L.ptr(wp), L.ptr(nbr), n_out, K, c_produce, None,
                                             L.ptr(out), L.ptr(order), L.ptr(cls_sorted), L.hptr(ks), L.hptr(st), L.hptr(pd), L.stream())
    L.check(rc, "toda_spconv_gather_gemm_classed")
    return out


def wgrad(feat, dout, nbr, wshape):
    lib = L.load()
    K, n_out = nbr.shape
    cout, cin = wshape[0], wshape[-1]
    dw = torch.empty(wshape, dtype=torch.float32, device=feat.device)
    ws_bytes = lib.toda_spconv_wgrad_workspace_bytes(n_out, K, cin, cout)
    ws = torch.empty((max(ws_bytes, 16),), dtype=torch.uint8, device=feat.device)
    rc = lib.toda_spconv_wgrad(L.ptr(feat), feat.shape[0], L.ptr(dout), L.ptr(nbr), n_out, K, cin, cout, L.ptr(dw),
                               L.ptr(ws), ws_bytes, L.stream())
    L.check(rc, "toda_spconv_wgrad")
    return dw


# data gradient of strided convolutions over residue-class-sorted rows (1..8 candidate offsets per row instead of 27)
CLASS_DGRAD = _os.environ.get("TODA_CLASS_DGRAD", "1") == "1"
CLASS_DGRAD_MIN_ROWS = int(_os.environ.get("TODA_CLASS_DGRAD_MIN_ROWS", "4096"))
WGRAD_ON_SIDE_STREAM = _os.environ.get("TODA_WGRAD_STREAM", "0") == "1"  # measured: no gain (each kernel already fills the chip)
_SIDE_STREAMS = {}


def _side_stream(device):
    key = (device.type, device.index)
    if key not in _SIDE_STREAMS:
        _SIDE_STREAMS[key] = torch.cuda.Stream(device=device)
    return _SIDE_STREAMS[key]


class _SparseConv(torch.autograd.Function):
    """out = conv(features; weight, bias, rulebook).  Backward = dgrad (same gather-GEMM kernel on
    the transposed table and transposed packed weights) + wgrad, as autograd does through spconv
    (reference tools/train_utils/train_utils.py:55)."""

    @staticmethod
    def forward(ctx, features, weight, bias, rb, wp_fwd, want_stats=False, wp_bwd=None):
        features = features.contiguous()
        compact = _compact_route(features.shape[1], weight.shape[0], rb.nbr_fwd)
        if wp_fwd is None and not compact:
            wp_fwd = pack_weight(weight, False, False)
        if wp_fwd is not None:
            _check_mm(wp_fwd)
        sums, blocks = None, 0
        if compact and want_stats:
            out, sums, blocks = gather_gemm_compact(features, weight.contiguous(), rb.nbr_fwd, weight.shape[0], bias, stats=True)
        elif want_stats and FOLD_IN_FINALIZE:
            out, sums, blocks = gather_gemm_with_stats(features, wp_fwd, rb.nbr_fwd, weight.shape[0], bias, partials=True)
        elif want_stats:
            out, sums = gather_gemm_with_stats(features, wp_fwd, rb.nbr_fwd, weight.shape[0], bias)
        elif compact:
            out = gather_gemm_compact(features, weight.contiguous(), rb.nbr_fwd, weight.shape[0], bias)
        else:
            out = gather_gemm(features, wp_fwd, rb.nbr_fwd, weight.shape[0], bias)
        ctx.set_materialize_grads(False)     # the moments output takes no gradient: no zero-filled double tensor per layer and step
        ctx.save_for_backward(features, weight)
        ctx.rb = rb
        ctx.wp_bwd = wp_bwd          # the dgrad operand when the module packed it with the rest of the backbone
        ctx.has_bias = bias is not None
        if want_stats:
            ctx.mark_non_differentiable(sums)
            return out, sums, blocks       # blocks > 0: sums holds unfolded per-workgroup partials (toda_bn_finalize_partials)
        return out

    @staticmethod
    def backward(ctx, gout, *unused):
        features, weight = ctx.saved_tensors
        rb = ctx.rb
        if gout is None:
            return None, None, None, None, None, None, None
        gout = gout.contiguous()
        gfeat = gw = gb = None
        need_d, need_w = ctx.needs_input_grad[0], ctx.needs_input_grad[1]
        side = _side_stream(gout.device) if (need_d and need_w and WGRAD_ON_SIDE_STREAM) else None
        if side is not None:
            # dgrad and wgrad only share their inputs: run wgrad on a second HIP stream so that the two
            # MFMA-bound kernels (each ~60 % matrix-pipe utilisation on its own) fill each other's gaps
            main = torch.cuda.current_stream(gout.device)
            side.wait_stream(main)
            with torch.cuda.stream(side):
                gw = wgrad(features, gout, rb.nbr_fwd, tuple(weight.shape))
            for t in (features, gout, rb.nbr_fwd):
                t.record_stream(side)
            need_w = False
        if need_d and _compact_route(gout.shape[1], weight.shape[-1], rb.nbr_bwd):
            gfeat = gather_gemm_compact(gout, weight.contiguous(), rb.nbr_bwd, weight.shape[-1], None, True, rb.flip_bwd)
        elif need_d:
            wp_t = ctx.wp_bwd if ctx.wp_bwd is not None else pack_weight(weight, True, rb.flip_bwd)
            _check_mm(wp_t)
            co = rb.class_order()
            if co is not None:
                gfeat = gather_gemm_classed(gout, wp_t, rb.nbr_bwd, weight.shape[-1], co[0], co[1], rb.ksize, rb.geom["stride"], rb.geom["padding"])
            else:
                gfeat = gather_gemm(gout, wp_t, rb.nbr_bwd, weight.shape[-1], None)
        if need_w:
            gw = wgrad(features, gout, rb.nbr_fwd, tuple(weight.shape))
        if ctx.has_bias and ctx.needs_input_grad[2]:
            gb = _take_colsum(gout)          # BatchNorm's backward has already summed its dx over the rows (toda_rows_bn_bwd_res_colsum)
            if gb is None:
                gb = gout.sum(0)
        if side is not None:
            torch.cuda.current_stream(gout.device).wait_stream(side)
            gw.record_stream(torch.cuda.current_stream(gout.device))
        return gfeat, gw, gb, None, None, None, None


FUSE_BN_STATS = _os.environ.get("TODA_FUSE_BN_STATS", "1") == "1"
FOLD_IN_FINALIZE = _os.environ.get("TODA_BN_FOLD_FINALIZE", "1") == "1"     # the fold of the epilogue's partial sums inside the finalise launch


def sparse_conv(features, weight, bias, rulebook, packed_weight=None, want_stats=False, packed_dgrad=None):
    """want_stats: also return the BatchNorm moments of the output when the kernel's epilogue can take them
    (returns (out, sums) with sums None when it cannot: empty tables, narrow channel pairs, mask-sorted row order)."""
    if not want_stats:
        return _SparseConv.apply(features, weight, bias, rulebook, packed_weight, False, packed_dgrad)
    narrow = FOLD_IN_FINALIZE and _compact_route(features.shape[1], weight.shape[0], rulebook.nbr_fwd)
    ok = (FUSE_BN_STATS and (narrow or gather_gemm_stats_supported(weight.shape[-1], weight.shape[0])) and rulebook.nbr_fwd.shape[1] > 1
          and features.shape[0] > 0)
    if not ok:
        return _SparseConv.apply(features, weight, bias, rulebook, packed_weight, False, packed_dgrad), None
    out, sums, blocks = _SparseConv.apply(features, weight, bias, rulebook, packed_weight, True, packed_dgrad)
    return out, ((sums, blocks) if blocks else sums)


# ------------------------------------------------------------------------ sparse <-> dense
class _ToDense(torch.autograd.Function):
    @staticmethod
    def forward(ctx, features, indices, batch, shape):
        lib = L.load()
        n, c = features.shape
        shape = [int(s) for s in shape]
        dense = torch.empty((batch, c, *shape), dtype=torch.float32, device=features.device)
        sh = L.host_i32(shape)
        rc = lib.toda_sparse_to_dense_fwd(L.ptr(features.contiguous()), L.ptr(indices), n, c, batch, L.hptr(sh),
                                          L.ptr(dense), L.stream())
        L.check(rc, "toda_sparse_to_dense_fwd")
        ctx.save_for_backward(indices)
        ctx.meta = (n, c, batch, shape)
        return dense

    @staticmethod
    def backward(ctx, gdense):
        (indices,) = ctx.saved_tensors
        n, c, batch, shape = ctx.meta
        g = torch.empty((n, c), dtype=torch.float32, device=gdense.device)
        sh = L.host_i32(shape)
        rc = L.load().toda_sparse_to_dense_bwd(L.ptr(gdense.contiguous()), L.ptr(indices), n, c, batch, L.hptr(sh),
                                               L.ptr(g), L.stream())
        L.check(rc, "toda_sparse_to_dense_bwd")
        return g, None, None, None


def sparse_to_dense(features, indices, batch, shape):
    return _ToDense.apply(features, indices.contiguous(), int(batch), shape)


class _PillarScatter(torch.autograd.Function):
    @staticmethod
    def forward(ctx, features, indices, batch, ny, nx):
        n, c = features.shape
        canvas = torch.empty((batch, c, ny, nx), dtype=torch.float32, device=features.device)
        rc = L.load().toda_pillar_scatter_fwd(L.ptr(features.contiguous()), L.ptr(indices), n, c, batch, ny, nx,
                                              L.ptr(canvas), L.stream())
        L.check(rc, "toda_pillar_scatter_fwd")
        ctx.save_for_backward(indices)
        ctx.meta = (n, c, batch, ny, nx)
        return canvas

    @staticmethod
    def backward(ctx, gcanvas):
        (indices,) = ctx.saved_tensors
        n, c, batch, ny, nx = ctx.meta
        g = torch.empty((n, c), dtype=torch.float32, device=gcanvas.device)
        rc = L.load().toda_pillar_scatter_bwd(L.ptr(gcanvas.contiguous()), L.ptr(indices), n, c, batch, ny, nx, L.ptr(g),
                                              L.stream())
        L.check(rc, "toda_pillar_scatter_bwd")
        return g, None, None, None, None


def pillar_scatter(features, indices, batch, ny, nx):
    return _PillarScatter.apply(features, indices.contiguous(), int(batch), int(ny), int(nx))


# ------------------------------------------------------------------- BN1d building blocks
def rows_moments(x):
    n, c = x.shape
    lib = L.load()
    sums = torch.empty((lib.toda_rows_reduce_doubles(n, c),), dtype=torch.float64, device=x.device)
    L.check(lib.toda_rows_moments(L.ptr(x.contiguous()), n, c, L.ptr(sums), L.stream()), "toda_rows_moments")
    return sums[:2 * c]


def rows_affine_act(x, scale, shift, residual=None, relu=True):
    n, c = x.shape
    y = torch.empty_like(x)
    rc = L.load().toda_rows_affine_act(L.ptr(x.contiguous()), L.ptr(scale.contiguous()), L.ptr(shift.contiguous()),
                                       L.ptr(residual), n, c, int(bool(relu)), L.ptr(y), L.stream())
    L.check(rc, "toda_rows_affine_act")
    return y


# Column sums of a BatchNorm backward's dx, handed to the backward of the convolution that produced x (its bias gradient): keyed by
# the gradient tensor, which autograd passes on untouched when the convolution's output has that one consumer.  The entry holds
# the tensor itself, so its address cannot be re-used while the entry is alive; whatever is left over is dropped by the next forward.
_DX_COLSUM = {}
BN_BWD_COLSUM = _os.environ.get("TODA_BN_BWD_COLSUM", "1") == "1"


def _take_colsum(gout):
    ent = _DX_COLSUM.pop(gout.data_ptr(), None) if _DX_COLSUM else None
    if ent is None or ent[0].shape != gout.shape or ent[0]._version != ent[2] or gout.stride() != ent[0].stride():
        return None
    return ent[1]


class _BNRows(torch.autograd.Function):
    """nn.BatchNorm1d(+ReLU) over the rows of a sparse level in 3 passes forward / 5 backward
    (torch: 5 / 7): toda_rows_moments -> toda_bn_finalize -> toda_rows_affine_act, toda_rows_bn_bwd."""

    @staticmethod
    def forward(ctx, x, weight, bias, running_mean, running_var, training, momentum, eps, relu, residual=None, sums=None, colsum=False):
        lib = L.load()
        if _DX_COLSUM:
            _DX_COLSUM.clear()
        ctx.colsum = bool(colsum) and BN_BWD_COLSUM
        x = x.contiguous()
        residual = residual.contiguous() if residual is not None else None
        n, c = x.shape
        dev = x.device
        stats = torch.empty((4, c), dtype=torch.float32, device=dev)  # mean, invstd, scale, shift
        if isinstance(sums, tuple) and training:      # unfolded partials from the convolution's epilogue: fold + finalise in one launch
            part, blocks = sums
            rc = lib.toda_bn_finalize_partials(L.ptr(part), int(blocks), n, c, L.ptr(weight), L.ptr(bias), L.ptr(running_mean), L.ptr(running_var),
                                               float(momentum), float(eps), L.ptr(stats[0]), L.ptr(stats[1]), L.ptr(stats[2]), L.ptr(stats[3]), L.stream())
            L.check(rc, "toda_bn_finalize_partials")
            sums = part
        else:
            if isinstance(sums, tuple):
                sums = None
            sums = _BNRows._finalize(lib, x, n, c, dev, sums, training, weight, bias, running_mean, running_var, momentum, eps, stats)
        y = torch.empty_like(x)
        rc = lib.toda_rows_affine_act(L.ptr(x), L.ptr(stats[2]), L.ptr(stats[3]), L.ptr(residual), n, c, int(bool(relu)), L.ptr(y),
                                      L.stream())
        L.check(rc, "toda_rows_affine_act")
        ctx.save_for_backward(x, stats, weight, residual)
        ctx.meta = (n, c, bool(relu), bool(training))
        return y

    @staticmethod
    def _finalize(lib, x, n, c, dev, sums, training, weight, bias, running_mean, running_var, momentum, eps, stats):
        if sums is None or not training:     # moments not delivered by the producing convolution: one pass over x
            sums = torch.empty((lib.toda_rows_reduce_doubles(n, c),), dtype=torch.float64, device=dev)   # [0:2c] result + per-block scratch
            if training:
                L.check(lib.toda_rows_moments(L.ptr(x), n, c, L.ptr(sums), L.stream()), "toda_rows_moments")
        rc = lib.toda_bn_finalize(L.ptr(sums), n, c, L.ptr(weight), L.ptr(bias), L.ptr(running_mean), L.ptr(running_var),
                                  float(momentum), float(eps), int(bool(training)), L.ptr(stats[0]), L.ptr(stats[1]),
                                  L.ptr(stats[2]), L.ptr(stats[3]), L.stream())
        L.check(rc, "toda_bn_finalize")
        return sums

    @staticmethod
    def backward(ctx, gy):
        x, stats, weight, residual = ctx.saved_tensors
        n, c, relu, training = ctx.meta
        gy = gy.contiguous()
        want_res = residual is not None and ctx.needs_input_grad[9]
        if not training:  # eval: plain affine map
            pre = x * stats[2] + stats[3]
            if residual is not None:
                pre = pre + residual
            dz = gy * (pre > 0) if relu else gy
            gx = dz * stats[2]
            xhat = (x - stats[0]) * stats[1]
            return gx, (dz * xhat).sum(0), dz.sum(0), None, None, None, None, None, None, (dz if want_res else None), None, None
        sums = torch.empty((L.load().toda_rows_reduce_doubles(n, c),), dtype=torch.float64, device=x.device)
        gx = torch.empty_like(x)
        gamma = weight if weight is not None else torch.ones(c, device=x.device)
        gres = torch.empty_like(x) if want_res else None
        if ctx.colsum:
            lib = L.load()
            cs_ws = torch.empty((lib.toda_rows_bn_bwd_colsum_doubles(n, c),), dtype=torch.float64, device=x.device)
            cs = torch.empty((c,), dtype=torch.float32, device=x.device)
            rc = lib.toda_rows_bn_bwd_res_colsum(L.ptr(gy), L.ptr(x), L.ptr(residual), L.ptr(stats), L.ptr(gamma), n, c, int(relu),
                                                 L.ptr(sums), L.ptr(gx), L.ptr(gres), L.ptr(cs_ws), L.ptr(cs), L.stream())
            L.check(rc, "toda_rows_bn_bwd_res_colsum")
            _DX_COLSUM[gx.data_ptr()] = (gx, cs, gx._version)
        else:
            rc = L.load().toda_rows_bn_bwd_res(L.ptr(gy), L.ptr(x), L.ptr(residual), L.ptr(stats), L.ptr(gamma), n, c, int(relu),
                                               L.ptr(sums), L.ptr(gx), L.ptr(gres), L.stream())
            L.check(rc, "toda_rows_bn_bwd_res")
        gs = sums[2 * c:3 * c].view(torch.float32)    # the kernel leaves (float)sums[0:2c] behind the double results
        return gx, gs[c:], gs[:c], None, None, None, None, None, None, gres, None, None


def bn_rows(x, bn, relu, residual=None, sums=None, colsum=False):
    """Apply an nn.BatchNorm1d module (its parameters, buffers and train/eval state) to rows [N, C], optionally
    fused with a shortcut addition (y = bn(x) + residual) and the ReLU that follows.  sums: the moments of x when the
    convolution that produced x has already taken them (ops.sparse_conv(..., want_stats=True)).  colsum: x comes straight out of a
    convolution with a bias - the backward also sums dx over the rows and leaves the result for that convolution's backward."""
    training = bn.training or not bn.track_running_stats
    bump_bn_counter(bn)
    momentum = 0.0 if bn.momentum is None else bn.momentum
    return _BNRows.apply(x, bn.weight, bn.bias, bn.running_mean, bn.running_var, training, momentum, bn.eps, relu, residual, sums, colsum)


def bn_rows_supported(x, bn):
    """The fused row passes implement nn.BatchNorm1d with per-rank statistics and nothing else: a SyncBatchNorm
    (tools/train.py --sync_bn, reference tools/train.py:117-118) or any other norm class takes its own module path."""
    c = x.shape[1]
    return (type(bn) is torch.nn.BatchNorm1d and x.is_cuda and x.dtype == torch.float32 and 4 <= c <= 128 and 256 % c == 0 and bn.affine and bn.momentum is not None
            and x.shape[0] > 1)


_BN2D_SYNC = {}      # (device, stream) -> [workspace, epoch]


def _bn2d_sync(device):
    """The partner workgroups' exchange area of toda_bn2d_*: zeroed once per (device, stream); every call takes the next epoch."""
    key = (device.index, L.stream())
    ent = _BN2D_SYNC.get(key)
    if ent is None:
        ent = _BN2D_SYNC[key] = [torch.zeros((L.load().toda_bn2d_sync_bytes(),), dtype=torch.uint8, device=device), 0]
    ent[1] = ent[1] % 0xFFFFFFF0 + 1
    return ent[0], ent[1]


class _BN2d(torch.autograd.Function):
    """Training-mode nn.BatchNorm2d (+ nn.ReLU) in one pass per direction (toda_bn2d_fwd / _bwd): a channel's values stay in
    registers between the statistics and the normalisation, so forward reads x once and writes y once; the running statistics
    are updated by the kernel like nn.BatchNorm2d does; backward recomputes the ReLU mask from x (reference
    base_bev_backbone.py:37-58, center_head.py:20-28)."""

    @staticmethod
    def forward(ctx, x, weight, bias, running_mean, running_var, momentum, eps, relu):
        lib = L.load()
        x = x.contiguous()
        b, c, h, w = x.shape
        y = torch.empty_like(x)
        save = torch.empty((2, c), dtype=torch.float32, device=x.device)
        ws, epoch = _bn2d_sync(x.device)
        rc = lib.toda_bn2d_fwd(L.ptr(x), b, c, h * w, L.ptr(weight), L.ptr(bias), L.ptr(running_mean), L.ptr(running_var), float(momentum),
                               float(eps), int(bool(relu)), L.ptr(y), L.ptr(save), L.ptr(ws), epoch, L.stream())
        L.check(rc, "toda_bn2d_fwd")
        ctx.save_for_backward(x, weight, bias, save)
        ctx.relu = bool(relu)
        return y

    @staticmethod
    def backward(ctx, gy):
        x, weight, bias, save = ctx.saved_tensors
        lib = L.load()
        b, c, h, w = x.shape
        gy = gy.contiguous()
        gx = torch.empty_like(x)
        gwb = torch.empty((2, c), dtype=torch.float32, device=x.device)
        ws, epoch = _bn2d_sync(x.device)
        rc = lib.toda_bn2d_bwd(L.ptr(x), L.ptr(gy), b, c, h * w, L.ptr(weight), L.ptr(bias), L.ptr(save), int(ctx.relu), L.ptr(gx),
                               L.ptr(gwb[0]), L.ptr(gwb[1]), L.ptr(ws), epoch, L.stream())
        L.check(rc, "toda_bn2d_bwd")
        return gx, gwb[0], gwb[1], None, None, None, None, None


class _BN2dCat(torch.autograd.Function):
    """torch.cat([relu(bn_i(x_i))], dim=1) for training-mode BatchNorm2d modules (the tails of the BEV neck's deblocks, reference
    base_bev_backbone.py:95-107): every toda_bn2d_fwd_into writes its channels straight into the concatenated map and the backward
    reads the map's gradient slice in place - no cat kernel forward (64 us on the C3 map), no slice copies backward (2 x 27 us).
    Arguments: relu, n, then per item x, weight, bias, running_mean, running_var, momentum, eps."""

    @staticmethod
    def forward(ctx, relu, n, *flat):
        lib = L.load()
        items = [flat[7 * i:7 * i + 7] for i in range(n)]
        xs = [it[0].contiguous() for it in items]
        b, _, h, w = xs[0].shape
        chans = [x.shape[1] for x in xs]
        out = torch.empty((b, sum(chans), h, w), dtype=torch.float32, device=xs[0].device)
        saves, c0 = [], 0
        for x, it in zip(xs, items):
            _, weight, bias, rm, rv, momentum, eps = it
            c = x.shape[1]
            save = torch.empty((2, c), dtype=torch.float32, device=x.device)
            ws, epoch = _bn2d_sync(x.device)
            rc = lib.toda_bn2d_fwd_into(L.ptr(x), b, c, h * w, L.ptr(weight), L.ptr(bias), L.ptr(rm), L.ptr(rv), float(momentum), float(eps),
                                        int(bool(relu)), L.ptr(out), out.shape[1], c0, L.ptr(save), L.ptr(ws), epoch, L.stream())
            L.check(rc, "toda_bn2d_fwd_into")
            saves.append(save)
            c0 += c
        ctx.save_for_backward(*xs, *[it[1] for it in items], *[it[2] for it in items], *saves)
        ctx.meta = (bool(relu), n, chans)
        return out

    @staticmethod
    def backward(ctx, g):
        relu, n, chans = ctx.meta
        t = ctx.saved_tensors
        xs, weights, biases, saves = t[:n], t[n:2 * n], t[2 * n:3 * n], t[3 * n:]
        lib = L.load()
        g = g.contiguous()
        b, ctot, h, w = g.shape
        grads, c0 = [], 0
        for x, weight, bias, save, c in zip(xs, weights, biases, saves, chans):
            gx = torch.empty_like(x)
            gwb = torch.empty((2, c), dtype=torch.float32, device=x.device)
            ws, epoch = _bn2d_sync(x.device)
            rc = lib.toda_bn2d_bwd_from(L.ptr(x), L.ptr(g), ctot, c0, b, c, h * w, L.ptr(weight), L.ptr(bias), L.ptr(save), int(relu), L.ptr(gx),
                                        L.ptr(gwb[0]), L.ptr(gwb[1]), L.ptr(ws), epoch, L.stream())
            L.check(rc, "toda_bn2d_bwd_from")
            grads += [gx, gwb[0], gwb[1], None, None, None, None]
            c0 += c
        return (None, None, *grads)


FUSED_BN2D = _os.environ.get("TODA_BN2D", "1") == "1"
BN2D_CAT = _os.environ.get("TODA_BN2D_CAT", "1") == "1"


def bn2d_supported(x, bn):
    """Training-mode nn.BatchNorm2d modules (exactly that class: SyncBatchNorm keeps its own path) on contiguous fp32 NCHW GPU
    tensors whose channel fits the kernel's register image."""
    if not (FUSED_BN2D and type(bn) is torch.nn.BatchNorm2d and bn.training and bn.affine and bn.track_running_stats and bn.momentum is not None):
        return False
    if not (x.is_cuda and x.dtype == torch.float32 and x.dim() == 4 and x.is_contiguous()):
        return False
    return bool(L.load().toda_bn2d_supported(x.shape[0], x.shape[1], x.shape[2] * x.shape[3]))


def bn2d(x, bn, relu):
    """Apply a training-mode nn.BatchNorm2d module (parameters, running statistics, step counter) fused with a following ReLU."""
    bump_bn_counter(bn)
    return _BN2d.apply(x, bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.momentum, bn.eps, relu)


def bn2d_cat(pairs, relu=True):
    """torch.cat([relu(bn(x)) for x, bn in pairs], dim=1) with every BatchNorm2d writing its slice of the result (see _BN2dCat)."""
    flat = []
    for x, bn in pairs:
        bump_bn_counter(bn)
        flat += [x, bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.momentum, bn.eps]
    return _BN2dCat.apply(relu, len(pairs), *flat)


def bn_relu_tail(seq):
    """(modules before the tail, the BatchNorm2d) when an nn.Sequential ends in BatchNorm2d + ReLU, else None."""
    mods = list(seq)
    if len(mods) >= 2 and type(mods[-1]) is torch.nn.ReLU and type(mods[-2]) is torch.nn.BatchNorm2d:
        return mods[:-2], mods[-2]
    return None


def run_dense_sequential(seq, x):
    """nn.Sequential forward that sends the 3x3 convolutions to the Winograd kernels and every training-mode BatchNorm2d (+ the ReLU
    after it) to the single-pass kernel (bn2d) on the GPU; any other module, and everything on the CPU, runs as is.  Module tree
    and state_dict are untouched."""
    mods = list(seq)
    i = 0
    while i < len(mods):
        m = mods[i]
        if isinstance(m, torch.nn.Sequential):
            x = run_dense_sequential(m, x)
        elif (type(m) is torch.nn.ZeroPad2d and tuple(m.padding) == (1, 1, 1, 1) and i + 1 < len(mods) and type(mods[i + 1]) is torch.nn.Conv2d
              and mods[i + 1].padding == (0, 0) and conv3x3_supported(x, mods[i + 1])):
            x = conv3x3(x, mods[i + 1].weight, mods[i + 1].bias)      # ZeroPad2d(1) + Conv2d(3x3, padding 0) = one padded convolution
            i += 2
            continue
        elif (type(m) is torch.nn.ZeroPad2d and tuple(m.padding) == (1, 1, 1, 1) and i + 1 < len(mods) and conv3x3s2_supported(x, mods[i + 1])):
            x = conv3x3s2(x, mods[i + 1].weight, mods[i + 1].bias)    # the stride-2 head of a block
            i += 2
            continue
        elif deconv_supported(x, m):
            x = deconv(x, m.weight, m.stride[0], m.bias)              # up-sampling deblock (kernel = stride)
        elif type(m) is torch.nn.Conv2d and m.padding == (1, 1) and conv3x3_supported(x, m):
            x = conv3x3(x, m.weight, m.bias)
        elif type(m) is torch.nn.BatchNorm2d and bn2d_supported(x, m):
            relu = i + 1 < len(mods) and type(mods[i + 1]) is torch.nn.ReLU
            x = bn2d(x, m, relu)
            i += 2 if relu else 1
            continue
        elif type(m) is torch.nn.BatchNorm2d and x.is_cuda and m.track_running_stats and m.momentum is not None:
            bump_bn_counter(m)        # nn.BatchNorm2d.forward minus its own counter launch
            x = torch.nn.functional.batch_norm(x, m.running_mean, m.running_var, m.weight, m.bias, m.training, m.momentum, m.eps)
        else:
            x = m(x)
        i += 1
    return x


# ------------------------------------------------------------------ dense 3x3 convolution (BEV neck, heads)
# TODA_DENSE_CONV=miopen sends every Conv2d back to torch / MIOpen (A/B knob and fallback for unsupported shapes)
DENSE_CONV = _os.environ.get("TODA_DENSE_CONV", "winograd")
_WINO_CACHE = {}


def conv3x3_supported(x, conv):
    """nn.Conv2d modules the Winograd F(4x4,3x3) kernels take: 3x3, stride 1, dilation 1, groups 1, padding 1 (or padding 0
    behind an explicit ZeroPad2d(1), which the caller folds in), fp32 NCHW on the GPU, channels multiples of 32, even W."""
    if DENSE_CONV != "winograd" or type(conv) is not torch.nn.Conv2d or not x.is_cuda or x.dtype != torch.float32 or x.dim() != 4:
        return False
    if conv.kernel_size != (3, 3) or conv.stride != (1, 1) or conv.dilation != (1, 1) or conv.groups != 1:
        return False
    if conv.padding_mode != "zeros" or conv.in_channels != x.shape[1]:
        return False
    b, c, h, w = x.shape
    return bool(L.load().toda_conv3x3_supported(b, conv.in_channels, conv.out_channels, h, w))


def conv3x3_transform_weight(weight, mode):
    """[Cout, Cin, 3, 3] -> G w G^T in the kernel's operand order (mode 0 forward, 1 data gradient, 2 both: [2, n])."""
    lib = L.load()
    cout, cin = weight.shape[0], weight.shape[1]
    n = lib.toda_conv3x3_weight_floats(cout, cin)
    u = torch.empty((2, n) if mode == 2 else (n,), dtype=torch.float32, device=weight.device)
    L.check(lib.toda_conv3x3_transform_weight(L.ptr(weight.contiguous()), cout, cin, int(mode), L.ptr(u), L.stream()),
            "toda_conv3x3_transform_weight")
    return u


def _conv3x3_workspace(device):
    """Per (device, stream) scratch of the stream-K work split: zeroed once, the kernel leaves its flags zero."""
    key = (device.index, L.stream())
    ws = _WINO_CACHE.get(key)
    if ws is None:
        ws = _WINO_CACHE[key] = torch.zeros((L.load().toda_conv3x3_workspace_bytes(),), dtype=torch.uint8, device=device)
    return ws


def conv3x3_run(x, u, bias, cout):
    lib = L.load()
    b, cin, h, w = x.shape
    y = torch.empty((b, cout, h, w), dtype=torch.float32, device=x.device)
    ws = _conv3x3_workspace(x.device)
    L.check(lib.toda_conv3x3_fwd(L.ptr(x), L.ptr(u), L.ptr(bias), b, cin, cout, h, w, L.ptr(y), L.ptr(ws), ws.numel(), L.stream()),
            "toda_conv3x3_fwd")
    return y


class _Conv3x3(torch.autograd.Function):
    """y = conv2d(x, weight, bias, stride 1, padding 1).  Backward: dX = the same kernel on the rotated / transposed filters,
    dW = the Winograd-domain wgrad kernel, dB = sum of dY."""

    @staticmethod
    def forward(ctx, x, weight, bias):
        x = x.contiguous()
        need_dx = ctx.needs_input_grad[0]
        u = conv3x3_transform_weight(weight, 2 if need_dx else 0)      # both operands in one launch when dX will be wanted
        y = conv3x3_run(x, u[0] if need_dx else u, bias, weight.shape[0])
        ctx.save_for_backward(x, weight, u[1] if need_dx else None)
        ctx.has_bias = bias is not None
        return y

    @staticmethod
    def backward(ctx, gy):
        x, weight, u_dgrad = ctx.saved_tensors
        gy = gy.contiguous()
        gx = gw = gb = None
        if ctx.needs_input_grad[0]:
            gx = conv3x3_run(gy, u_dgrad if u_dgrad is not None else conv3x3_transform_weight(weight, 1), None, weight.shape[1])
        if ctx.needs_input_grad[1]:
            gw = conv3x3_wgrad(x, gy, weight.shape)
        if ctx.has_bias and ctx.needs_input_grad[2]:
            gb = gy.sum((0, 2, 3))
        return gx, gw, gb


def conv3x3_wgrad(x, gy, wshape):
    lib = L.load()
    b, cin, h, w = x.shape
    cout = gy.shape[1]
    dw = torch.empty(tuple(wshape), dtype=torch.float32, device=x.device)
    ws_bytes = lib.toda_conv3x3_wgrad_workspace_bytes(b, cin, cout, h, w)
    ws = torch.empty((max(ws_bytes, 16),), dtype=torch.uint8, device=x.device)
    L.check(lib.toda_conv3x3_wgrad(L.ptr(x), L.ptr(gy), b, cin, cout, h, w, L.ptr(dw), L.ptr(ws), ws_bytes, L.stream()),
            "toda_conv3x3_wgrad")
    return dw


def conv3x3(x, weight, bias=None):
    return _Conv3x3.apply(x, weight, bias)


# ------------------------------------------- stride-2 3x3 convolution and the transposed-convolution deblocks of the BEV neck
class _Conv3x3S2(torch.autograd.Function):
    """ZeroPad2d(1) + Conv2d(3, stride 2, padding 0) = conv2d(x, weight, stride 2, padding 1) on even maps
    (reference base_bev_backbone.py:32-36), forward / data gradient / weight gradient through toda_conv3x3s2_*."""

    @staticmethod
    def forward(ctx, x, weight):
        lib = L.load()
        x, weight = x.contiguous(), weight.contiguous()
        b, cin, h, w = x.shape
        cout = weight.shape[0]
        y = torch.empty((b, cout, h // 2, w // 2), dtype=torch.float32, device=x.device)
        L.check(lib.toda_conv3x3s2_fwd(L.ptr(x), L.ptr(weight), b, cin, cout, h, w, L.ptr(y), L.stream()), "toda_conv3x3s2_fwd")
        ctx.save_for_backward(x, weight)
        return y

    @staticmethod
    def backward(ctx, gy):
        x, weight = ctx.saved_tensors
        lib = L.load()
        gy = gy.contiguous()
        b, cin, h, w = x.shape
        cout = weight.shape[0]
        gx = gw = None
        if ctx.needs_input_grad[0]:
            gx = torch.empty_like(x)
            L.check(lib.toda_conv3x3s2_dgrad(L.ptr(gy), L.ptr(weight), b, cin, cout, h, w, L.ptr(gx), L.stream()), "toda_conv3x3s2_dgrad")
        if ctx.needs_input_grad[1]:
            gw = torch.empty_like(weight)
            nb = lib.toda_conv3x3s2_wgrad_workspace_bytes(b, cin, cout, h, w)
            ws = torch.empty((max(nb, 16),), dtype=torch.uint8, device=x.device)
            L.check(lib.toda_conv3x3s2_wgrad(L.ptr(x), L.ptr(gy), b, cin, cout, h, w, L.ptr(gw), L.ptr(ws), nb, L.stream()), "toda_conv3x3s2_wgrad")
        return gx, gw


DENSE_GEMM = _os.environ.get("TODA_DENSE_GEMM", "1") == "1"      # 0: the stride-2 conv and the deblocks back on torch (MIOpen / rocBLAS) - A/B knob


def conv3x3s2_supported(x, conv):
    """nn.Conv2d(3x3, stride 2, padding 0) behind a ZeroPad2d(1) (the caller checks the pad), fp32 NCHW on the GPU, even H and W."""
    if not DENSE_GEMM or DENSE_CONV != "winograd" or type(conv) is not torch.nn.Conv2d or not x.is_cuda or x.dtype != torch.float32 or x.dim() != 4:
        return False
    if conv.kernel_size != (3, 3) or conv.stride != (2, 2) or conv.dilation != (1, 1) or conv.groups != 1 or conv.padding != (0, 0):
        return False
    if conv.padding_mode != "zeros" or conv.in_channels != x.shape[1]:
        return False
    b, c, h, w = x.shape
    return bool(L.load().toda_conv3x3s2_supported(b, conv.in_channels, conv.out_channels, h, w))


def conv3x3s2(x, weight, bias=None):
    y = _Conv3x3S2.apply(x, weight)
    return y if bias is None else y + bias.view(1, -1, 1, 1)


class _Deconv(torch.autograd.Function):
    """ConvTranspose2d(Cin, Cout, kernel = stride = s), s in {1, 2} (reference base_bev_backbone.py:47-66): one GEMM per direction;
    the pixel shuffle of s = 2 is the store of the forward GEMM and the gather of the backward ones (toda_deconv_*)."""

    @staticmethod
    def forward(ctx, x, weight, s):
        lib = L.load()
        x, weight = x.contiguous(), weight.contiguous()
        b, cin, h, w = x.shape
        cout = weight.shape[1]
        y = torch.empty((b, cout, h * s, w * s), dtype=torch.float32, device=x.device)
        L.check(lib.toda_deconv_fwd(L.ptr(x), L.ptr(weight), b, cin, cout, h, w, s, L.ptr(y), L.stream()), "toda_deconv_fwd")
        ctx.save_for_backward(x, weight)
        ctx.s = s
        return y

    @staticmethod
    def backward(ctx, gy):
        x, weight = ctx.saved_tensors
        lib = L.load()
        gy = gy.contiguous()
        b, cin, h, w = x.shape
        cout, s = weight.shape[1], ctx.s
        gx = gw = None
        if ctx.needs_input_grad[0]:
            gx = torch.empty_like(x)
            L.check(lib.toda_deconv_dgrad(L.ptr(gy), L.ptr(weight), b, cin, cout, h, w, s, L.ptr(gx), L.stream()), "toda_deconv_dgrad")
        if ctx.needs_input_grad[1]:
            gw = torch.empty_like(weight)
            nb = lib.toda_deconv_wgrad_workspace_bytes(b, cin, cout, h, w, s)
            ws = torch.empty((max(nb, 16),), dtype=torch.uint8, device=x.device)
            L.check(lib.toda_deconv_wgrad(L.ptr(x), L.ptr(gy), b, cin, cout, h, w, s, L.ptr(gw), L.ptr(ws), nb, L.stream()), "toda_deconv_wgrad")
        return gx, gw, None


def deconv_supported(x, m):
    """nn.ConvTranspose2d with kernel = stride in {1, 2}, no padding / output padding / dilation / groups, fp32 NCHW on the GPU."""
    if not DENSE_GEMM or DENSE_CONV != "winograd" or type(m) is not torch.nn.ConvTranspose2d or not x.is_cuda or x.dtype != torch.float32 or x.dim() != 4:
        return False
    s = m.stride[0]
    if m.kernel_size != (s, s) or m.stride != (s, s) or s not in (1, 2) or m.padding != (0, 0) or m.output_padding != (0, 0):
        return False
    if m.dilation != (1, 1) or m.groups != 1 or m.in_channels != x.shape[1]:
        return False
    return True


def deconv(x, weight, s, bias=None):
    y = _Deconv.apply(x, weight, int(s))
    return y if bias is None else y + bias.view(1, -1, 1, 1)


# ---------------------------------------------------- narrow-output 3x3 convolutions (last layer of the head branches)
def conv3x3_narrow_supported(x, conv):
    if DENSE_CONV != "winograd" or type(conv) is not torch.nn.Conv2d or not x.is_cuda or x.dtype != torch.float32 or x.dim() != 4:
        return False
    if conv.kernel_size != (3, 3) or conv.stride != (1, 1) or conv.dilation != (1, 1) or conv.groups != 1 or conv.padding != (1, 1):
        return False
    if conv.padding_mode != "zeros" or conv.in_channels != x.shape[1]:
        return False
    b, c, h, w = x.shape
    return bool(L.load().toda_conv3x3_narrow_supported(b, c, conv.out_channels, h, w))


class _Conv3x3NarrowGroup(torch.autograd.Function):
    """y_i = conv2d(x_i, w_i, b_i, padding 1) for the n branches of a head in one launch per direction.
    apply(n, fused, x_0..x_{m-1}, w_0.., b_0..) -> (y_0, .., y_{n-1}); biases may be None.  fused = False: m = n separate
    inputs [B, C, H, W]; fused = True: m = 1, branch i reads channels [i C, (i + 1) C) of one [B, n C, H, W] tensor (and the
    input gradient is produced as one tensor of that shape)."""

    @staticmethod
    def forward(ctx, n, fused, *args):
        lib = L.load()
        m = 1 if fused else n
        xs = [a.contiguous() for a in args[:m]]
        ws = [a.contiguous() for a in args[m:m + n]]
        bs = list(args[m + n:m + 2 * n])
        b, ctot, h, w = xs[0].shape
        cin = ctot // n if fused else ctot
        stride = ctot * h * w if fused else 0
        x_addr = [xs[0].data_ptr() + 4 * i * cin * h * w for i in range(n)] if fused else [L.ptr(x) for x in xs]
        couts = [int(wt.shape[0]) for wt in ws]
        ys = [torch.empty((b, co, h, w), dtype=torch.float32, device=xs[0].device) for co in couts]
        co_host = L.host_i32(couts)
        has_bias = any(t is not None for t in bs)
        rc = lib.toda_conv3x3_narrow_fwd(n, L.host_addrs(x_addr), L.host_ptrs(ws), L.host_ptrs(bs) if has_bias else None, L.hptr(co_host),
                                         b, cin, h, w, stride, L.host_ptrs(ys), L.stream())
        L.check(rc, "toda_conv3x3_narrow_fwd")
        ctx.n, ctx.m, ctx.couts, ctx.geom = n, m, couts, (b, cin, h, w, stride)
        ctx.has_bias = [t is not None for t in bs]
        ctx.save_for_backward(*xs, *ws)
        return tuple(ys)

    @staticmethod
    def backward(ctx, *gys):
        lib = L.load()
        n, m, couts = ctx.n, ctx.m, ctx.couts
        b, cin, h, w, stride = ctx.geom
        xs, ws = ctx.saved_tensors[:m], ctx.saved_tensors[m:]
        gys = [g.contiguous() for g in gys]
        dev = gys[0].device
        co_host = L.host_i32(couts)
        gxs = [None] * m
        if any(ctx.needs_input_grad[2:2 + m]):
            if stride:
                gxs = [torch.empty((b, n * cin, h, w), dtype=torch.float32, device=dev)]
                gx_addr = [gxs[0].data_ptr() + 4 * i * cin * h * w for i in range(n)]
            else:
                gxs = [torch.empty((b, cin, h, w), dtype=torch.float32, device=dev) for _ in range(n)]
                gx_addr = [L.ptr(g) for g in gxs]
            rc = lib.toda_conv3x3_narrow_dgrad(n, L.host_ptrs(gys), L.host_ptrs(ws), L.hptr(co_host), b, cin, h, w, stride, L.host_addrs(gx_addr),
                                               L.stream())
            L.check(rc, "toda_conv3x3_narrow_dgrad")
        x_addr = [xs[0].data_ptr() + 4 * i * cin * h * w for i in range(n)] if stride else [L.ptr(x) for x in xs]
        total = sum(co * (cin * 9 + 1) for co in couts)
        out = torch.empty((total,), dtype=torch.float32, device=dev)
        ws_bytes = lib.toda_conv3x3_narrow_wgrad_workspace_bytes(n, b, cin, h)
        wsp = torch.empty((ws_bytes,), dtype=torch.uint8, device=dev)
        rc = lib.toda_conv3x3_narrow_wgrad(n, L.host_addrs(x_addr), L.host_ptrs(gys), L.hptr(co_host), b, cin, h, w, stride, L.ptr(out), L.ptr(wsp),
                                           ws_bytes, L.stream())
        L.check(rc, "toda_conv3x3_narrow_wgrad")
        gws, gbs, off = [], [], 0
        for co, hb in zip(couts, ctx.has_bias):
            gws.append(out[off:off + co * cin * 9].view(co, cin, 3, 3))
            off += co * cin * 9
            gbs.append(out[off:off + co] if hb else None)
            off += co
        return (None, None, *gxs, *gws, *gbs)


def conv3x3_narrow_group(xs, convs):
    """The final nn.Conv2d of several head branches (same input geometry) in one launch; returns the list of outputs."""
    n = len(xs)
    return list(_Conv3x3NarrowGroup.apply(n, False, *xs, *[c.weight for c in convs], *[c.bias for c in convs]))


def conv3x3_narrow_group_fused(x, convs):
    """Same, the branch inputs being the len(convs) equal channel slices of x [B, n C, H, W] (no slice copies)."""
    n = len(convs)
    return list(_Conv3x3NarrowGroup.apply(n, True, x, *[c.weight for c in convs], *[c.bias for c in convs]))


# ------------------------------------------------- BatchNorm2d modules of parallel branches as one normalisation
_BN_COUNTERS = []
_BN_DEFER = [0]


class bn_counter_scope:
    """Inside the scope (a detector's forward) the `num_batches_tracked += 1` of every BatchNorm the hand-written paths
    apply is collected and issued as ONE foreach launch on exit instead of one 4-us kernel per layer (53 per C3 step)."""

    def __enter__(self):
        _BN_DEFER[0] += 1
        return self

    def __exit__(self, *exc):
        _BN_DEFER[0] -= 1
        if _BN_DEFER[0] == 0 and _BN_COUNTERS:
            pending = list(_BN_COUNTERS)
            _BN_COUNTERS.clear()
            torch._foreach_add_(pending, 1)
        return False


def bump_bn_counter(bn):
    if bn.training and bn.track_running_stats and bn.num_batches_tracked is not None:
        if _BN_DEFER[0] and bn.num_batches_tracked.is_cuda:
            _BN_COUNTERS.append(bn.num_batches_tracked)
        else:
            bn.num_batches_tracked.add_(1)


def _aliased_running_stats(owner, bns):
    """running_mean / running_var of the n BatchNorm2d modules as consecutive slices of two [n C] tensors, so that one
    batch_norm call updates all of them in place.  The modules keep their buffers (same names, same values, state_dict
    unchanged); the aliasing is re-established whenever something (module.to(), a fresh load) has replaced a buffer."""
    c = bns[0].num_features
    cache = getattr(owner, "_fused_bn_stats", None)
    ok = cache is not None and cache[0].device == bns[0].running_mean.device
    if ok:
        for i, bn in enumerate(bns):
            if bn.running_mean.data_ptr() != cache[0].data_ptr() + 4 * c * i or bn.running_var.data_ptr() != cache[1].data_ptr() + 4 * c * i:
                ok = False
                break
    if not ok:
        with torch.no_grad():
            rm = torch.cat([bn.running_mean for bn in bns])
            rv = torch.cat([bn.running_var for bn in bns])
            for i, bn in enumerate(bns):
                bn.running_mean = rm[i * c:(i + 1) * c]
                bn.running_var = rv[i * c:(i + 1) * c]
        cache = (rm, rv)
        object.__setattr__(owner, "_fused_bn_stats", cache)
    return cache


def fused_branch_hidden_supported(x, blocks):
    """blocks: the first layers Sequential(Conv2d(C, C, 3, padding 1), BatchNorm2d(C), ReLU) of n parallel branches on x."""
    if len(blocks) < 2 or not x.is_cuda:
        return False
    c = x.shape[1]
    ref_bn = blocks[0][1] if len(blocks[0]) == 3 else None
    for blk in blocks:
        if type(blk) is not torch.nn.Sequential or len(blk) != 3:
            return False
        conv, bn, act = blk[0], blk[1], blk[2]
        if type(conv) is not torch.nn.Conv2d or type(bn) is not torch.nn.BatchNorm2d or type(act) is not torch.nn.ReLU:
            return False
        if conv.out_channels != c or conv.padding != (1, 1) or (conv.bias is None) != (blocks[0][0].bias is None) or not conv3x3_supported(x, conv):
            return False
        if (not bn.affine or not bn.track_running_stats or bn.momentum is None or bn.momentum != ref_bn.momentum or bn.eps != ref_bn.eps
                or bn.training != ref_bn.training or bn.num_features != c):
            return False
    return bool(L.load().toda_conv3x3_supported(x.shape[0], c, c * len(blocks), x.shape[2], x.shape[3]))


def fused_branch_hidden(owner, x, blocks):
    """ReLU(BN_i(conv_i(x))) of n parallel branches as ONE C -> n C convolution, ONE batch norm over the n C channels and
    one ReLU (reference center_head.py:20-26 runs them branch by branch).  Per-channel batch statistics make the wide norm
    equal to the n narrow ones; the backward of the wide convolution sums the branches' input gradients inside its GEMM.
    Returns [B, n C, H, W]; branch i owns channels [i C, (i + 1) C)."""
    convs, bns = [blk[0] for blk in blocks], [blk[1] for blk in blocks]
    w = torch.cat([cv.weight for cv in convs], 0)
    bias = torch.cat([cv.bias for cv in convs]) if convs[0].bias is not None else None
    y = conv3x3(x, w, bias)
    rm, rv = _aliased_running_stats(owner, bns)
    for bn in bns:
        bump_bn_counter(bn)
    gamma, beta = torch.cat([bn.weight for bn in bns]), torch.cat([bn.bias for bn in bns])
    if (FUSED_BN2D and bns[0].training and all(type(bn) is torch.nn.BatchNorm2d for bn in bns) and y.is_contiguous()
            and L.load().toda_bn2d_supported(y.shape[0], y.shape[1], y.shape[2] * y.shape[3])):
        return _BN2d.apply(y, gamma, beta, rm, rv, bns[0].momentum, bns[0].eps, True)
    y = torch.nn.functional.batch_norm(y, rm, rv, gamma, beta, bns[0].training, bns[0].momentum, bns[0].eps)
    return torch.relu_(y)


# --------------------------------------------------------------- CenterHead target assign
def center_assign(gt_boxes, num_classes, fm_w, fm_h, pc_range, voxel_size, fm_stride, max_objs=500, overlap=0.1,
                  min_radius=2):
    """gt_boxes [B, G, code] with the class column already remapped to 1..num_classes of this head
    (0 = not in this head / padding).  Returns heatmap, ret_boxes, inds (int64), mask (int64)."""
    lib = L.load()
    gt = gt_boxes.to(torch.float32).contiguous()
    batch, n_gt, code = gt.shape
    dev = gt.device
    hm = torch.empty((batch, num_classes, fm_h, fm_w), dtype=torch.float32, device=dev)
    rb = torch.empty((batch, max_objs, code), dtype=torch.float32, device=dev)
    inds = torch.empty((batch, max_objs), dtype=torch.int64, device=dev)
    mask = torch.empty((batch, max_objs), dtype=torch.int64, device=dev)
    rng, vs = L.host_f32(pc_range), L.host_f32(voxel_size)
    rc = lib.toda_center_assign(L.ptr(gt), batch, n_gt, code, int(num_classes), int(fm_w), int(fm_h), L.hptr(rng),
                                L.hptr(vs), int(fm_stride), int(max_objs), float(overlap), int(min_radius), L.ptr(hm),
                                L.ptr(rb), L.ptr(inds), L.ptr(mask), L.stream())
    L.check(rc, "toda_center_assign")
    return hm, rb, inds, mask


class _CenterLoss(torch.autograd.Function):
    """CenterHead.get_loss of one head group in three launches forward / one backward (toda_center_loss_*).
    apply(n, hm_logits, reg_0..reg_{n-1}, heatmap, inds, mask, target_boxes, code_weights, cls_weight, loc_weight)
    -> (hm_loss, loc_loss, clamped sigmoid); the first two are 0-d tensors carrying the graph."""

    @staticmethod
    def forward(ctx, n, hm, *args):
        lib = L.load()
        regs = [a.contiguous() for a in args[:n]]
        heatmap, inds, mask, target, code_w, cls_w, loc_w = args[n:]
        hm = hm.contiguous()
        b, c, h, w = hm.shape
        k, d = target.shape[1], target.shape[2]
        chans = [int(r.shape[1]) for r in regs]
        dev = hm.device
        ws_bytes = lib.toda_center_loss_workspace_bytes(b, c, h, w, k, d)
        ws = torch.empty((ws_bytes,), dtype=torch.uint8, device=dev)
        prob = torch.empty_like(hm)
        out4 = torch.empty((4,), dtype=torch.float32, device=dev)
        inds, mask = inds.contiguous(), mask.contiguous()
        cw = L.host_f32(code_w)
        rc = lib.toda_center_loss_fwd(L.ptr(hm), L.ptr(heatmap.contiguous()), b, c, h, w, n, L.host_ptrs(regs), L.hptr(L.host_i32(chans)),
                                      L.ptr(inds), L.ptr(mask), L.ptr(target.contiguous().float()), k, d, L.hptr(cw), float(cls_w), float(loc_w),
                                      L.ptr(prob), L.ptr(out4), L.ptr(ws), ws_bytes, L.stream())
        L.check(rc, "toda_center_loss_fwd")
        ctx.save_for_backward(out4, ws, inds, mask)
        ctx.meta = (n, b, c, h, w, k, d, chans, [float(v) for v in code_w], float(cls_w), float(loc_w))
        ctx.mark_non_differentiable(prob)
        return out4[0], out4[1], prob

    @staticmethod
    def backward(ctx, g_hm, g_loc, _g_prob):
        lib = L.load()
        out4, ws, inds, mask = ctx.saved_tensors
        n, b, c, h, w, k, d, chans, code_w, cls_w, loc_w = ctx.meta
        dev = out4.device
        g_hm = g_hm.contiguous() if g_hm is not None else torch.zeros((), device=dev)
        g_loc = g_loc.contiguous() if g_loc is not None else torch.zeros((), device=dev)
        dz = torch.empty((b, c, h, w), dtype=torch.float32, device=dev)
        flat = torch.empty((b * d * h * w,), dtype=torch.float32, device=dev)      # the branch gradients, back to back: one zero fill
        grads, off = [], 0
        for ch in chans:
            grads.append(flat[off:off + b * ch * h * w].view(b, ch, h, w))
            off += b * ch * h * w
        cw = L.host_f32(code_w)
        rc = lib.toda_center_loss_bwd(L.ptr(out4), L.ptr(g_hm), L.ptr(g_loc), b, c, h, w, n, L.host_ptrs(grads), L.hptr(L.host_i32(chans)),
                                      L.ptr(inds), L.ptr(mask), k, d, L.hptr(cw), cls_w, loc_w, L.ptr(dz), L.ptr(ws), ws.numel(), L.stream())
        L.check(rc, "toda_center_loss_bwd")
        return (None, dz, *grads, None, None, None, None, None, None, None)


def center_loss(hm_logits, regs, heatmap, inds, mask, target_boxes, code_weights, cls_weight, loc_weight):
    """(hm_loss * cls_weight, loc_loss * loc_weight, clamp(sigmoid(hm_logits), 1e-4, 1 - 1e-4)) of one CenterHead group;
    regs: the regression maps in HEAD_ORDER."""
    return _CenterLoss.apply(len(regs), hm_logits, *regs, heatmap, inds, mask, target_boxes, list(code_weights), cls_weight, loc_weight)


def center_loss_supported(hm_logits, regs, target_boxes):
    return (hm_logits.is_cuda and hm_logits.dtype == torch.float32 and 1 <= len(regs) <= 8 and target_boxes.shape[2] <= 16
            and all(r.dtype == torch.float32 and r.shape[0] == hm_logits.shape[0] and r.shape[2:] == hm_logits.shape[2:] for r in regs)
            and sum(int(r.shape[1]) for r in regs) == target_boxes.shape[2] and _os.environ.get("TODA_FUSED_LOSS", "1") == "1")


# ------------------------------------------------------------------ rotated IoU / NMS (eval path)
def boxes_iou_bev(boxes_a, boxes_b):
    a, b = boxes_a[:, :7].contiguous().float(), boxes_b[:, :7].contiguous().float()
    iou = torch.zeros((a.shape[0], b.shape[0]), dtype=torch.float32, device=a.device)
    L.check(L.load().toda_boxes_iou_bev(L.ptr(a), a.shape[0], L.ptr(b), b.shape[0], L.ptr(iou), L.stream()),
            "toda_boxes_iou_bev")
    return iou


def boxes_overlap_bev(boxes_a, boxes_b):
    """Intersection area of rotated BEV rectangles, [N, 7] x [M, 7] -> [N, M]."""
    lib = L.load()
    a, b = boxes_a[:, :7].contiguous().float(), boxes_b[:, :7].contiguous().float()
    out = torch.zeros((a.shape[0], b.shape[0]), dtype=torch.float32, device=a.device)
    rc = lib.toda_boxes_overlap_bev(L.ptr(a), a.shape[0], L.ptr(b), b.shape[0], L.ptr(out), L.stream())
    L.check(rc, "toda_boxes_overlap_bev")
    return out


def nms_rotated(boxes_sorted, thresh):
    """Greedy rotated NMS over boxes sorted by descending score.  Returns (keep [n] int64 padded,
    n_keep [1] int32), both on the device - no host sync."""
    lib = L.load()
    b = boxes_sorted[:, :7].contiguous().float()
    n = b.shape[0]
    keep = torch.empty((max(n, 1),), dtype=torch.int64, device=b.device)
    n_keep = torch.zeros((1,), dtype=torch.int32, device=b.device)
    ws_bytes = lib.toda_nms_workspace_bytes(n)
    ws = torch.empty((ws_bytes,), dtype=torch.uint8, device=b.device)
    rc = lib.toda_nms_rotated(L.ptr(b), n, float(thresh), L.ptr(keep), L.ptr(n_keep), L.ptr(ws), ws_bytes, L.stream())
    L.check(rc, "toda_nms_rotated")
    return keep, n_keep


# --------------------------------------------------------------------------- point tables (mix processors, range mask)
def _rows(points, n_dev):
    if points.dtype != torch.float32 or points.dim() != 2:
        raise RuntimeError("point tables are [n, c] float32")
    return points.shape[0], points.shape[1], (L.ptr(n_dev) if n_dev is not None else None)


def points_in_boxes(points, boxes, mode=0, n_dev=None):
    """flags[j] = 1 iff some box holds point j (mode 0: roiaware points_in_boxes_cpu test, mode 1: get_points_in_box)."""
    lib = L.load()
    n, c, nd = _rows(points, n_dev)
    flags = torch.full((n,), -1 if mode == 2 else 0, dtype=torch.int32, device=points.device)   # mode 2: first box index
    k = int(boxes.shape[0]) if boxes is not None and boxes.dim() == 2 else 0
    if n == 0 or k == 0:
        return flags
    boxes = boxes.contiguous().float()
    rc = lib.toda_points_in_boxes(L.ptr(points), n, nd, c, L.ptr(boxes), k, boxes.shape[1], int(mode), L.ptr(flags), L.stream())
    L.check(rc, "toda_points_in_boxes")
    return flags


def points_sector(points, lo, hi, n_dev=None):
    lib = L.load()
    n, c, nd = _rows(points, n_dev)
    flags = torch.zeros((n,), dtype=torch.int32, device=points.device)
    rc = lib.toda_points_sector(L.ptr(points), n, nd, c, float(lo), float(hi), L.ptr(flags), L.stream())
    L.check(rc, "toda_points_sector")
    return flags


def points_rect(points, lo_xy, hi_xy, closed=False, n_dev=None):
    lib = L.load()
    n, c, nd = _rows(points, n_dev)
    flags = torch.zeros((n,), dtype=torch.int32, device=points.device)
    lo, hi = L.host_f64(lo_xy), L.host_f64(hi_xy)
    rc = lib.toda_points_rect(L.ptr(points), n, nd, c, L.hptr(lo), L.hptr(hi), int(bool(closed)), L.ptr(flags), L.stream())
    L.check(rc, "toda_points_rect")
    return flags


def points_polar_cell(points, phase, yaw_edges, dis_edges, dis_lo, dis_hi, n_dev=None):
    lib = L.load()
    n, c, nd = _rows(points, n_dev)
    cell = torch.full((n,), -1, dtype=torch.int32, device=points.device)
    ye, de = L.host_f64(yaw_edges), L.host_f64(dis_edges)
    rc = lib.toda_points_polar_cell(L.ptr(points), n, nd, c, float(phase), L.hptr(ye), len(yaw_edges) - 1, L.hptr(de),
                                    len(dis_edges) - 1, float(dis_lo), float(dis_hi), L.ptr(cell), L.stream())
    L.check(rc, "toda_points_polar_cell")
    return cell


def points_polar_select(points, lo, hi, outside=False, dis_mode=0, dis_th=0.0, pitch_range=None, n_dev=None):
    """flags of PolarMix's richer sector tests (C ABI: toda_points_polar_select): yaw inside (lo, hi) or - `outside` - beyond it,
    optionally cut at a range (dis_mode 1: nearer than dis_th, 2: farther) and / or kept only where the elevation lies outside
    `pitch_range` (device float[2] from points_pitch_range)."""
    lib = L.load()
    n, c, nd = _rows(points, n_dev)
    flags = torch.zeros((n,), dtype=torch.int32, device=points.device)
    rc = lib.toda_points_polar_select(L.ptr(points), n, nd, c, float(lo), float(hi), 2 if outside else 1, int(dis_mode), float(dis_th),
                                      L.ptr(pitch_range) if pitch_range is not None else None, L.ptr(flags), L.stream())
    L.check(rc, "toda_points_polar_select")
    return flags


def points_pitch_range(points, n_dev=None):
    """Device float[2]: min and max of -atan2(z, range) over the rows beyond 1 m of range (no host sync)."""
    lib = L.load()
    n, c, nd = _rows(points, n_dev)
    out = torch.empty((2,), dtype=torch.float32, device=points.device)
    ws_bytes = lib.toda_points_pitch_range_workspace_bytes()
    ws = torch.empty((ws_bytes,), dtype=torch.uint8, device=points.device)
    rc = lib.toda_points_pitch_range(L.ptr(points), n, nd, c, L.ptr(out), L.ptr(ws), ws_bytes, L.stream())
    L.check(rc, "toda_points_pitch_range")
    return out


def points_pitch_band(points, z_offset, clip_lo, clip_hi, edges, n_dev=None):
    """Elevation band of every row (spherical LaserMix): `edges` descending, radians; -1 outside every band."""
    lib = L.load()
    n, c, nd = _rows(points, n_dev)
    band = torch.full((n,), -1, dtype=torch.int32, device=points.device)
    ed = L.host_f64(edges)
    rc = lib.toda_points_pitch_band(L.ptr(points), n, nd, c, float(z_offset), float(clip_lo), float(clip_hi), L.hptr(ed), len(edges) - 1,
                                    L.ptr(band), L.stream())
    L.check(rc, "toda_points_pitch_band")
    return band


def points_rotate_z(points, cosv, sinv, n_dev=None):
    lib = L.load()
    n, c, nd = _rows(points, n_dev)
    out = torch.empty_like(points)
    rc = lib.toda_points_rotate_z(L.ptr(points), n, nd, c, float(cosv), float(sinv), L.ptr(out), L.stream())
    L.check(rc, "toda_points_rotate_z")
    return out


def points_world_transform(points, flip_x=False, flip_y=False, rot=None, scale=None, n_dev=None, out=None):
    """Flip / rotate about z (rot = (cos, sin) as fp32 values) / scale in one pass; returns a new table unless out is given."""
    lib = L.load()
    n, c, nd = _rows(points, n_dev)
    out = torch.empty_like(points) if out is None else out
    cs, sn = (float(rot[0]), float(rot[1])) if rot is not None else (1.0, 0.0)
    rc = lib.toda_points_world_transform(L.ptr(points), n, nd, c, int(bool(flip_x)), int(bool(flip_y)), int(rot is not None), cs, sn,
                                         int(scale is not None), float(scale if scale is not None else 1.0), L.ptr(out), L.stream())
    L.check(rc, "toda_points_world_transform")
    return out


class RowBuffer:
    """A point table under construction on the device: rows are appended by stable selection at a device-side
    cursor, so a whole mix runs without a host round trip; `finish()` reads the row count (one sync)."""

    def __init__(self, cap_rows, c, device):
        self.cap = int(cap_rows)
        self.data = torch.empty((max(self.cap, 1), c), dtype=torch.float32, device=device)
        self.cursor = torch.zeros((1,), dtype=torch.int32, device=device)

    def append(self, src, keys=None, match=1, invert=False, n_dev=None):
        lib = L.load()
        n, c, nd = _rows(src, n_dev)
        if c != self.data.shape[1]:
            raise RuntimeError("RowBuffer.append: column count differs")
        if n == 0:
            return self
        ws_bytes = lib.toda_rows_select_workspace_bytes(n)
        ws = torch.empty((ws_bytes,), dtype=torch.uint8, device=src.device)
        rc = lib.toda_rows_select_append(L.ptr(src), n, nd, c, L.ptr(keys), int(match), int(bool(invert)), L.ptr(self.data),
                                         self.cap, L.ptr(self.cursor), L.ptr(ws), ws_bytes, L.stream())
        L.check(rc, "toda_rows_select_append")
        return self

    def finish(self):
        n = int(self.cursor.item())
        if n > self.cap:
            raise RuntimeError(f"RowBuffer overflow: {n} rows for a capacity of {self.cap}")
        return self.data[:n]
