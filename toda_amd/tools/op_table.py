"""Per-kernel roofline table of the hand-written path (bench.py --layers, SURVEY.md §8 d): every entry point of
libtoda_hip.so that a training step calls is bracketed by HIP events on the current stream and priced with its ALGORITHMIC
bytes / FLOPs (formulas of SURVEY.md §8 d and DESIGN.md §3), so each row carries GB/s against the 8 TB/s HBM roof and / or
TFLOP/s against the 157.3 TFLOP/s fp32 matrix roof and the fraction of the binding one.

Event brackets include the launch overhead of the call (a few microseconds) and whatever idle time the host left between
the first event and the launch, so `ms` is the MEDIAN over the calls of a shape and rows of kernels shorter than ~20 us read
low; the rocprofv3 kernel trace in profiles/ has the dispatch-stamped durations of the same kernels."""
import collections

import torch

from .. import lib as L
from .. import ops

HBM_GBS = 8000.0
MFMA_TF = 157.3


class OpTable:
    def __init__(self):
        self.lib = L.load()
        self.enabled = False
        self.events = collections.defaultdict(list)      # key -> [(e0, e1)]
        self.cost = {}                                    # key -> (bytes, flops, note)
        self._pairs = {}
        self._orig = {}
        for name in ("toda_voxelize_hard", "toda_voxelize_batch", "toda_gridindex_from_coords_unordered", "toda_gridindex_from_bitmap", "toda_gridindex_clear",
                     "toda_rulebook_class_order", "toda_mean_vfe_fwd", "toda_mean_vfe_bwd", "toda_gridindex_from_coords", "toda_gridindex_from_conv",
                     "toda_rulebook_subm", "toda_rulebook_conv", "toda_sparse_to_dense_fwd", "toda_sparse_to_dense_bwd", "toda_rows_moments",
                     "toda_rows_affine_act", "toda_rows_bn_bwd_res", "toda_bn2d_fwd", "toda_bn2d_bwd", "toda_spconv_pack_weight", "toda_conv3x3_transform_weight",
                     "toda_center_assign", "toda_bn2d_fwd_into", "toda_bn2d_bwd_from", "toda_clip_adam_step", "toda_conv3x3s2_fwd", "toda_conv3x3s2_dgrad",
                     "toda_conv3x3s2_wgrad", "toda_deconv_fwd", "toda_deconv_dgrad", "toda_deconv_wgrad", "toda_conv3x3_narrow_fwd", "toda_conv3x3_narrow_dgrad",
                     "toda_conv3x3_narrow_wgrad", "toda_center_loss_fwd", "toda_center_loss_bwd"):
            if hasattr(self, "_c_" + name):
                self._wrap_c(name)
        self._wrap_py("gather_gemm", self._cost_gather_gemm)
        self._wrap_py("gather_gemm_with_stats", self._cost_gather_gemm)
        self._wrap_py("gather_gemm_classed", self._cost_gather_gemm_classed)
        self._wrap_py("gather_gemm_compact", self._cost_gather_gemm_compact)
        self._wrap_py("wgrad", self._cost_wgrad)
        self._wrap_py("conv3x3_run", self._cost_conv3x3)
        self._wrap_py("conv3x3_wgrad", self._cost_conv3x3_wgrad)

    # ------------------------------------------------------------------ wrappers
    def _bracket(self, key, cost, fn, *args):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        out = fn(*args)
        e1.record()
        self.events[key].append((e0, e1))
        self.cost.setdefault(key, cost)
        return out

    def _wrap_c(self, name):
        fn = getattr(self.lib, name)
        self._orig[("c", name)] = fn

        def timed(*a):
            if not self.enabled:
                return fn(*a)
            key, cost = getattr(self, "_c_" + name)(a)
            return self._bracket((name,) + key, cost, fn, *a)

        setattr(self.lib, name, timed)

    def _wrap_py(self, name, cost_fn):
        fn = getattr(ops, name)
        self._orig[("py", name)] = fn

        def timed(*a, **kw):
            if not self.enabled:
                return fn(*a, **kw)
            key, cost = cost_fn(*a, **kw)
            return self._bracket((name,) + key, cost, lambda: fn(*a, **kw))

        setattr(ops, name, timed)

    def restore(self):
        for (kind, name), fn in self._orig.items():
            setattr(self.lib if kind == "c" else ops, name, fn)

    # ------------------------------------------------------------------ algorithmic costs (bytes, flops, what was counted)
    def _count_pairs(self, nbr):
        key = (nbr.data_ptr(), tuple(nbr.shape))
        if key not in self._pairs:
            self._pairs[key] = nbr       # counted after the clock stops (rows())
        return key

    def _cost_gather_gemm(self, feat, wp, nbr, c_produce, bias=None, order=None, partials=False):
        K, n_out = nbr.shape
        n_src, cg = feat.shape
        pk = self._count_pairs(nbr)
        return (n_out, K, cg, c_produce), ("pairs", pk, lambda p: (4.0 * (n_src * cg + n_out * c_produce + K * cg * c_produce) + 8.0 * p,
                                                                   2.0 * p * cg * c_produce))

    def _cost_gather_gemm_classed(self, feat, wp, nbr, c_produce, order, cls_sorted, ksize, stride, padding):
        return self._cost_gather_gemm(feat, wp, nbr, c_produce)

    def _cost_gather_gemm_compact(self, feat, weight, nbr, c_produce, bias=None, transpose=False, flip_k=False, stats=False):
        return self._cost_gather_gemm(feat, None, nbr, c_produce)

    def _cost_wgrad(self, feat, dout, nbr, wshape):
        K, n_out = nbr.shape
        cout, cin = wshape[0], wshape[-1]
        pk = self._count_pairs(nbr)
        return (n_out, K, cin, cout), ("pairs", pk, lambda p: (4.0 * (feat.shape[0] * cin + n_out * cout + K * cin * cout) + 8.0 * p,
                                                               2.0 * p * cin * cout))

    @staticmethod
    def _cost_conv3x3(x, u, bias, cout):
        b, cin, h, w = x.shape
        flops = 2.0 * 9 * cin * cout * b * h * w
        return (b, cin, cout, h, w), ("fixed", 4.0 * b * h * w * (cin + cout) + 4.0 * 9 * cin * cout, flops,
                                      "direct-convolution FLOPs; the Winograd F(4x4,3x3) kernel issues 1/4 of them on the matrix cores")

    @staticmethod
    def _cost_conv3x3_wgrad(x, gy, wshape):
        b, cin, h, w = x.shape
        cout = gy.shape[1]
        return (b, cin, cout, h, w), ("fixed", 4.0 * b * h * w * (cin + cout) + 4.0 * 9 * cin * cout, 2.0 * 9 * cin * cout * b * h * w,
                                      "direct-convolution FLOPs (Winograd-domain contraction: 1/4 of them)")

    # C entry points: argument tuples as toda_amd/lib.py SIGNATURES
    @staticmethod
    def _c_toda_voxelize_hard(a):
        n, c, max_pts, cap = a[1], a[2], a[6], a[7]
        m = min(cap, n)
        return (n, c, max_pts), ("fixed", 4.0 * n * c + 2 * 4.0 * m * max_pts * c + 20.0 * m, 0.0, "M bounded by min(cap, points)")

    @staticmethod
    def _host_i32(arg, n):
        import ctypes
        addr = arg.value if hasattr(arg, "value") else int(arg)
        return list(ctypes.cast(addr, ctypes.POINTER(ctypes.c_int32))[0:n])

    @classmethod
    def _c_toda_voxelize_batch(cls, a):
        batch, c, max_pts, cap = a[2], a[4], a[9], a[10]
        ns = cls._host_i32(a[1], batch)
        byts = sum(4.0 * n * c + 2 * 4.0 * min(cap, n) * max_pts * c + 20.0 * min(cap, n) for n in ns)
        return (batch, sum(ns), c, max_pts), ("fixed", byts, 0.0, "whole batch (3 launches): per sample 4NC + 8MPC + 20M, M bounded by min(cap, points)")

    def _c_toda_gridindex_from_coords_unordered(self, a):
        n = a[1]
        return (n,), ("fixed", 16.0 * n + 8.0 * n, 0.0, "16 B coordinate row + one 8-byte cell per site; n = upper bound (cap) of the rows")

    _c_toda_gridindex_clear = _c_toda_gridindex_from_coords_unordered

    @classmethod
    def _c_toda_gridindex_from_bitmap(cls, a):
        batch = a[1]
        di, do = cls._host_i32(a[2], 3), cls._host_i32(a[6], 3)      # (a[11]: the input rows are marked)
        ci, co = batch * di[0] * di[1] * di[2] / 32.0, batch * do[0] * do[1] * do[2] / 32.0
        return (batch, *do), ("fixed", 8.0 * ci + 8.0 * co, 0.0, "one pass over the input bitmap's and the output bitmap's 8-byte words (output rows not counted)")

    @staticmethod
    def _c_toda_rulebook_class_order(a):
        n = a[1]
        return (n,), ("fixed", 16.0 * n + 5.0 * n, 0.0, "16 B coordinate row in, order + class out")

    @staticmethod
    def _c_toda_mean_vfe_fwd(a):
        m, p, c = a[2], a[3], a[4]
        return (m, p, c), ("fixed", 4.0 * m * (p * c + c + 1), 0.0, "")

    _c_toda_mean_vfe_bwd = _c_toda_mean_vfe_fwd

    def _c_toda_gridindex_from_coords(self, a):
        n = a[1]
        return (n,), ("fixed", 16.0 * n + 8.0 * n, 0.0, "16 B coordinate row + one 8-byte cell per site (bitmap clear / scan not counted)")

    @staticmethod
    def _c_toda_gridindex_from_conv(a):
        n = a[1]
        return (n,), ("fixed", 16.0 * n + 16.0 * n, 0.0, "16 B in + 16 B out per site (upper bound on N_out = N_in)")

    @staticmethod
    def _c_toda_rulebook_subm(a):
        n = a[1]
        return (n,), ("fixed", 16.0 * n + 16.0 * n + 4.0 * 27 * n, 0.0, "16 N_in + 16 N_out + the 27 x N table it writes")

    @staticmethod
    def _c_toda_rulebook_conv(a):
        n_in, n_out = a[1], a[9]
        return (n_in, n_out), ("fixed", 16.0 * n_in + 16.0 * n_out + 4.0 * 27 * (n_in + n_out), 0.0, "both tables (o2i, i2o)")

    @staticmethod
    def _c_toda_sparse_to_dense_fwd(a):
        import ctypes
        n, c, batch = a[2], a[3], a[4]
        addr = a[5].value if hasattr(a[5], "value") else int(a[5])
        d, h, w = ctypes.cast(addr, ctypes.POINTER(ctypes.c_int32))[0:3]
        return (n, c, batch, d, h, w), ("fixed", 4.0 * n * c + 4.0 * batch * c * d * h * w, 0.0, "sparse rows + the dense map")

    _c_toda_sparse_to_dense_bwd = _c_toda_sparse_to_dense_fwd

    @staticmethod
    def _c_toda_rows_moments(a):
        n, c = a[1], a[2]
        return (n, c), ("fixed", 4.0 * n * c, 0.0, "")

    @staticmethod
    def _c_toda_rows_affine_act(a):
        n, c = a[4], a[5]
        return (n, c), ("fixed", (12.0 if a[3] else 8.0) * n * c, 0.0, "")

    @staticmethod
    def _c_toda_rows_bn_bwd_res(a):
        n, c = a[5], a[6]
        return (n, c), ("fixed", 20.0 * n * c, 0.0, "")

    @staticmethod
    def _c_toda_bn2d_fwd(a):
        b, c, hw = a[1], a[2], a[3]
        return (b, c, hw), ("fixed", 8.0 * b * c * hw, 0.0, "x read once, y written once")

    @staticmethod
    def _c_toda_bn2d_bwd(a):
        b, c, hw = a[2], a[3], a[4]
        return (b, c, hw), ("fixed", 12.0 * b * c * hw, 0.0, "x once, dy once, dx once")

    @staticmethod
    def _c_toda_bn2d_fwd_into(a):
        b, c, hw = a[1], a[2], a[3]
        return (b, c, hw, a[12]), ("fixed", 8.0 * b * c * hw, 0.0, "x read once, y written once into its channel slice of the concatenated map")

    @staticmethod
    def _c_toda_bn2d_bwd_from(a):
        b, c, hw = a[4], a[5], a[6]
        return (b, c, hw, a[2]), ("fixed", 12.0 * b * c * hw, 0.0, "x once, dy (a channel slice) once, dx once")

    @staticmethod
    def _c_toda_clip_adam_step(a):
        n_chunks = a[7]
        p = 8192.0 * n_chunks
        return (n_chunks,), ("fixed", 4.0 * 9 * p, 0.0, "g twice, p / m / v read and written, g written: 36 bytes per parameter (upper bound: whole chunks)")

    @staticmethod
    def _pix_gemm(a, s, what):
        b, cin, cout, h, w = a[2], a[3], a[4], a[5], a[6]
        if what == "conv":       # 3x3 / stride 2: out = (h/2, w/2), 9 taps
            flops, byts = 2.0 * 9 * cin * cout * b * (h // 2) * (w // 2), 4.0 * b * (cin * h * w + cout * (h // 2) * (w // 2)) + 4.0 * 9 * cin * cout
        else:                    # transposed conv, kernel = stride s: out = (s h, s w)
            flops, byts = 2.0 * cin * cout * s * s * b * h * w, 4.0 * b * h * w * (cin + cout * s * s) + 4.0 * cin * cout * s * s
        return (b, cin, cout, h, w, s), ("fixed", byts, flops, "")

    def _c_toda_conv3x3s2_fwd(self, a):
        return self._pix_gemm(a, 2, "conv")

    _c_toda_conv3x3s2_dgrad = _c_toda_conv3x3s2_fwd
    _c_toda_conv3x3s2_wgrad = _c_toda_conv3x3s2_fwd

    def _c_toda_deconv_fwd(self, a):
        return self._pix_gemm(a, a[7], "deconv")

    _c_toda_deconv_dgrad = _c_toda_deconv_fwd
    _c_toda_deconv_wgrad = _c_toda_deconv_fwd

    @staticmethod
    def _c_toda_spconv_pack_weight(a):
        return (a[1], a[2], a[3]), ("fixed", 8.0 * a[1] * a[2] * a[3], 0.0, "")

    @staticmethod
    def _c_toda_conv3x3_transform_weight(a):
        return (a[1], a[2], a[3]), ("fixed", 4.0 * a[1] * a[2] * (9 + 36 * (2 if a[3] == 2 else 1)), 0.0, "")

    @staticmethod
    def _c_toda_center_assign(a):
        return (), ("fixed", 0.0, 0.0, "latency bound")

    # ------------------------------------------------------------------ report
    def rows(self, steps):
        torch.cuda.synchronize()
        pair_counts = {k: int((t >= 0).sum().item()) for k, t in self._pairs.items()}
        out = []
        for key, evs in self.events.items():
            ms = [a.elapsed_time(b) for a, b in evs]
            mean = sorted(ms)[len(ms) // 2]      # median: a bracket also holds any idle time the host left in front of the launch
            cost = self.cost[key]
            note = ""
            if cost[0] == "pairs":
                byts, flops = cost[2](pair_counts[cost[1]])
                note = f"pairs={pair_counts[cost[1]]}"
            else:
                byts, flops, note = cost[1], cost[2], cost[3]
            t_hbm, t_mfma = byts / (HBM_GBS * 1e9), flops / (MFMA_TF * 1e12)
            sec = mean * 1e-3
            row = {"op": key[0], "shape": list(key[1:]), "calls_per_step": round(len(ms) / steps, 2), "ms": round(mean, 4),
                   "ms_per_step": round(sum(ms) / steps, 4), "alg_bytes": byts, "alg_flops": flops,
                   "GB_per_s": round(byts / sec / 1e9, 1), "TFLOP_per_s": round(flops / sec / 1e12, 2),
                   "bound": "hbm" if t_hbm >= t_mfma else "mfma", "frac_of_roof": round(max(t_hbm, t_mfma) / sec, 4) if sec > 0 else None}
            if note:
                row["note"] = note
            out.append(row)
        out.sort(key=lambda r: -r["ms_per_step"])
        return out
