// Point-cloud side of the TODA mixing processors (PolarMix / CutMix / LaserMix / MixUp) and of the
// range mask of the data processor, for gfx950.  Every kernel is a streaming pass over the
// [n, c] fp32 point table (HBM-bound: 4*c bytes in per point, 4 bytes of flag out, then one
// stable compaction), so a mixed scene never leaves the device between the two raw clouds and
// the voxeliser.
//
// Decisions are made with exactly the roundings of the reference's numpy / C++ code so that
// the same random draws give the same point sets in the same order:
//   * azimuth  yaw = -arctan2(y, x) as an fp32 value        (inter_domain_point_polarmix.py:78-79)
//   * comparisons against thresholds in fp64 (thresholds that numpy treats as fp32 are rounded
//     to fp32 by the caller)
//   * point-in-box tests of roiaware_pool3d.cpp:121-141 and augmentor_utils.py:474-491.
// All kernels accept an optional device-side row count (rows = min(n, *n_dev)) so that a chain of
// select -> flag -> select runs without a host round trip.
#include "common.h"
#include "scan.cuh"

namespace toda {

constexpr int PT_BLOCK = 256;
constexpr int PT_MAX_EDGES = 33;

__device__ __forceinline__ float yaw_f32(float x, float y) { return -(float)atan2((double)y, (double)x); }

struct BoxPre {
    float cx, cy, cz, dx, dy, dz, cosa, sina;
};

// mode 0: roiaware_pool3d.cpp:121-141 (|z-cz| > dz/2 rejects, |local| < d/2 + 1e-2, fp64 compare)
// mode 1: augmentor_utils.py:474-491   (|z-cz| <= dz/2, |local| <= fp32(d/2 + 0.1))
// mode 2: roiaware_pool3d_kernel.cu:23-36 (as mode 0 with margin 1e-5 and fp32 cos / sin) - points_in_boxes_gpu
template <int MODE>
__device__ __forceinline__ bool point_in_box(float x, float y, float z, const BoxPre& b) {
    const float sz = z - b.cz;
    if (MODE == 0 || MODE == 2) {
        if ((double)fabsf(sz) > (double)b.dz / 2.0) return false;
    } else {
        if (!(fabsf(sz) <= b.dz / 2.0f)) return false;
    }
    const float sx = x - b.cx, sy = y - b.cy;
    const float lx = sx * b.cosa + sy * (-b.sina);
    const float ly = sx * b.sina + sy * b.cosa;
    if (MODE == 0 || MODE == 2) {
        const double m = MODE == 0 ? (double)1e-2f : (double)1e-5f;
        return fabs((double)lx) < (double)b.dx / 2.0 + m && fabs((double)ly) < (double)b.dy / 2.0 + m;
    }
    const float mx = b.dx / 2.0f + 0.1f, my = b.dy / 2.0f + 0.1f;
    return fabsf(lx) <= mx && fabsf(ly) <= my;
}

template <int MODE>
__global__ void __launch_bounds__(PT_BLOCK)
points_in_boxes_kernel(const float* __restrict__ pts, int n, const int32_t* __restrict__ n_dev, int c,
                       const float* __restrict__ boxes, int k, int box_stride, int32_t* __restrict__ flags) {
    extern __shared__ BoxPre sbox[];
    for (int i = threadIdx.x; i < k; i += PT_BLOCK) {
        const float* b = boxes + (size_t)i * box_stride;
        BoxPre p;
        p.cx = b[0], p.cy = b[1], p.cz = b[2], p.dx = b[3], p.dy = b[4], p.dz = b[5];
        if (MODE == 2) {
            p.cosa = cosf(-b[6]);
            p.sina = sinf(-b[6]);
        } else {
            const double a = (double)(-b[6]);
            p.cosa = (float)cos(a);
            p.sina = (float)sin(a);
        }
        sbox[i] = p;
    }
    __syncthreads();
    const int rows = eff_n(n, n_dev);
    const int j = blockIdx.x * PT_BLOCK + threadIdx.x;
    if (j >= rows) return;
    const float x = pts[(size_t)j * c], y = pts[(size_t)j * c + 1], z = pts[(size_t)j * c + 2];
    if (MODE == 2) {  // index of the first box that holds the point, -1 for none (points_in_boxes_kernel, :313-336)
        int first = -1;
        for (int i = k - 1; i >= 0; --i)
            if (point_in_box<MODE>(x, y, z, sbox[i])) first = i;
        flags[j] = first;
        return;
    }
    int hit = 0;
    for (int i = 0; i < k; ++i) hit |= point_in_box<MODE>(x, y, z, sbox[i]) ? 1 : 0;
    flags[j] = hit;
}

__global__ void __launch_bounds__(PT_BLOCK)
points_sector_kernel(const float* __restrict__ pts, int n, const int32_t* __restrict__ n_dev, int c, double lo, double hi,
                     int32_t* __restrict__ flags) {
    const int rows = eff_n(n, n_dev);
    const int j = blockIdx.x * PT_BLOCK + threadIdx.x;
    if (j >= rows) return;
    const double yaw = (double)yaw_f32(pts[(size_t)j * c], pts[(size_t)j * c + 1]);
    flags[j] = (yaw > lo) & (yaw < hi);
}

// closed = 0: lo < v < hi on x and y (CutMix crop, inter_domain_point_cutmix.py:44-54)
// closed = 1: lo <= v <= hi (mask_points_by_range, common_utils.py:60-63)
__global__ void __launch_bounds__(PT_BLOCK)
points_rect_kernel(const float* __restrict__ pts, int n, const int32_t* __restrict__ n_dev, int c, double lox, double loy,
                   double hix, double hiy, int closed, int32_t* __restrict__ flags) {
    const int rows = eff_n(n, n_dev);
    const int j = blockIdx.x * PT_BLOCK + threadIdx.x;
    if (j >= rows) return;
    const double x = (double)pts[(size_t)j * c], y = (double)pts[(size_t)j * c + 1];
    flags[j] = closed ? ((x >= lox) & (x <= hix) & (y >= loy) & (y <= hiy)) : ((x > lox) & (x < hix) & (y > loy) & (y < hiy));
}

struct PolarGrid {
    double yaw_edges[PT_MAX_EDGES], dis_edges[PT_MAX_EDGES];
    int n_yaw, n_dis;
    float phase, dis_lo, dis_hi;
};

// LaserMix cylinder cell of every point (inter_domain_point_lasermix.py:89-128): azimuth shifted by
// the phase and wrapped with the reference's own constants, range clipped, cell = i * n_dis + j
// with edges (lo, hi]; -1 when no cell matches (e.g. yaw == -pi exactly).
__global__ void __launch_bounds__(PT_BLOCK)
points_polar_cell_kernel(const float* __restrict__ pts, int n, const int32_t* __restrict__ n_dev, int c, PolarGrid g,
                         int32_t* __restrict__ cell) {
    const int rows = eff_n(n, n_dev);
    const int j = blockIdx.x * PT_BLOCK + threadIdx.x;
    if (j >= rows) return;
    const float x = pts[(size_t)j * c], y = pts[(size_t)j * c + 1];
    float yaw = yaw_f32(x, y) + g.phase;
    if (yaw > 3.141592f) yaw -= 6.283184f;
    if (yaw < -3.141592f) yaw += 6.283184f;
    float dis = sqrtf(x * x + y * y);
    dis = fminf(fmaxf(dis, g.dis_lo), g.dis_hi);
    int ci = -1, cj = -1;
    for (int i = 0; i < g.n_yaw; ++i)
        if ((double)yaw > g.yaw_edges[i] && (double)yaw <= g.yaw_edges[i + 1]) ci = i;
    for (int i = 0; i < g.n_dis; ++i)
        if ((double)dis > g.dis_edges[i] && (double)dis <= g.dis_edges[i + 1]) cj = i;
    cell[j] = (ci >= 0 && cj >= 0) ? ci * g.n_dis + cj : -1;
}

// ---- PolarMix with a range cut (swap_with_range) or an elevation test (swap, use_pitch) ---------------------------------
// range = sqrt(x^2 + y^2) with every step rounded to fp32 (np.sqrt(x ** 2 + y ** 2) on fp32 columns; -ffp-contract=off)
__device__ __forceinline__ float range_f32(float x, float y) { return sqrtf(x * x + y * y); }
// sign * arctan2(z, range) as an fp32 value (same convention as yaw_f32: fp64 evaluation, rounded once)
__device__ __forceinline__ float pitch_f32(float z, float range) { return (float)atan2((double)z, (double)range); }

// flags[j] = yaw test & range test & elevation test
//   yaw_mode 1: lo < yaw < hi, 2: yaw < lo | yaw > hi
//   dis_mode 0: none, 1: range < dis_th, 2: range > dis_th      (swap_with_range, inter_domain_point_polarmix.py:101-123)
//   pitch_range != NULL: range > 1 and -arctan2(z, range) outside [pitch_range[0], pitch_range[1]]   (swap, :81-93)
__global__ void __launch_bounds__(PT_BLOCK)
points_polar_select_kernel(const float* __restrict__ pts, int n, const int32_t* __restrict__ n_dev, int c, double lo, double hi,
                           int yaw_mode, int dis_mode, double dis_th, const float* __restrict__ pitch_range,
                           int32_t* __restrict__ flags) {
    const int rows = eff_n(n, n_dev);
    const int j = blockIdx.x * PT_BLOCK + threadIdx.x;
    if (j >= rows) return;
    const float x = pts[(size_t)j * c], y = pts[(size_t)j * c + 1];
    const double yaw = (double)yaw_f32(x, y);
    bool sel = yaw_mode == 1 ? ((yaw > lo) & (yaw < hi)) : ((yaw < lo) | (yaw > hi));
    const float dis = range_f32(x, y);
    if (dis_mode == 1) sel = sel && (double)dis < dis_th;
    if (dis_mode == 2) sel = sel && (double)dis > dis_th;
    if (pitch_range) {
        const float p = -pitch_f32(pts[(size_t)j * c + 2], dis);
        sel = sel && dis > 1.0f && (p < pitch_range[0] || p > pitch_range[1]);
    }
    flags[j] = sel ? 1 : 0;
}

// min / max of -arctan2(z, range) over the rows with range > 1 (pitch1[mask1].min() / .max(), :88): per-workgroup partials,
// then one workgroup folds them - no atomics, nothing to initialise.  No such row: (+inf, -inf).
__global__ void __launch_bounds__(PT_BLOCK)
points_pitch_range_kernel(const float* __restrict__ pts, int n, const int32_t* __restrict__ n_dev, int c,
                          const float* __restrict__ partial_in, int n_partial, float* __restrict__ out) {
    __shared__ float smin[PT_BLOCK / 64], smax[PT_BLOCK / 64];
    float lo = INFINITY, hi = -INFINITY;
    if (partial_in) {                                       // second pass: fold the partials
        for (int i = threadIdx.x; i < n_partial; i += PT_BLOCK) {
            lo = fminf(lo, partial_in[2 * i]);
            hi = fmaxf(hi, partial_in[2 * i + 1]);
        }
    } else {
        const int rows = eff_n(n, n_dev);
        for (int j = blockIdx.x * PT_BLOCK + threadIdx.x; j < rows; j += gridDim.x * PT_BLOCK) {
            const float x = pts[(size_t)j * c], y = pts[(size_t)j * c + 1];
            const float dis = range_f32(x, y);
            if (dis > 1.0f) {
                const float p = -pitch_f32(pts[(size_t)j * c + 2], dis);
                lo = fminf(lo, p);
                hi = fmaxf(hi, p);
            }
        }
    }
    for (int d = 32; d >= 1; d >>= 1) {
        lo = fminf(lo, __shfl_xor(lo, d, 64));
        hi = fmaxf(hi, __shfl_xor(hi, d, 64));
    }
    if ((threadIdx.x & 63) == 0) smin[threadIdx.x >> 6] = lo, smax[threadIdx.x >> 6] = hi;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < PT_BLOCK / 64; ++w) lo = fminf(lo, smin[w]), hi = fmaxf(hi, smax[w]);
        out[2 * blockIdx.x] = lo;
        out[2 * blockIdx.x + 1] = hi;
    }
}

struct PitchBands {
    double edges[PT_MAX_EDGES];      // descending, radians
    int n;
    float z_offset, clip_lo, clip_hi;
};

// Spherical LaserMix (laser_mix_transform_sph, inter_domain_point_lasermix.py:40-47,62-80): elevation = arctan2(z_offset + z,
// range) clipped to [clip_lo, clip_hi] in fp32, band i holds edges[i + 1] < elevation <= edges[i] (fp64 compare); -1: no band.
__global__ void __launch_bounds__(PT_BLOCK)
points_pitch_band_kernel(const float* __restrict__ pts, int n, const int32_t* __restrict__ n_dev, int c, PitchBands g,
                         int32_t* __restrict__ band) {
    const int rows = eff_n(n, n_dev);
    const int j = blockIdx.x * PT_BLOCK + threadIdx.x;
    if (j >= rows) return;
    const float x = pts[(size_t)j * c], y = pts[(size_t)j * c + 1];
    float p = pitch_f32(g.z_offset + pts[(size_t)j * c + 2], range_f32(x, y));
    p = fminf(fmaxf(p, g.clip_lo), g.clip_hi);
    int b = -1;
    for (int i = 0; i < g.n; ++i)
        if ((double)p > g.edges[i + 1] && (double)p <= g.edges[i]) b = i;
    band[j] = b;
}

// ---- stable compaction appended at a device-side cursor ------------------------------------
__global__ void __launch_bounds__(PT_BLOCK)
select_mark_kernel(const int32_t* __restrict__ keys, int n, const int32_t* __restrict__ n_dev, int match, int invert,
                   int32_t* __restrict__ rank) {
    const int rows = eff_n(n, n_dev);
    const int j = blockIdx.x * PT_BLOCK + threadIdx.x;
    if (j >= n) return;
    int sel = 0;
    if (j < rows) sel = keys ? ((keys[j] == match) != (invert != 0)) : 1;
    rank[j] = sel;
}

// rank[] holds the exclusive prefix; a row is selected iff the prefix steps at it.
__global__ void __launch_bounds__(PT_BLOCK)
select_copy_kernel(const float* __restrict__ src, int n, int c, const int32_t* __restrict__ rank,
                   const int32_t* __restrict__ total, float* __restrict__ dst, int cap_rows,
                   const int32_t* __restrict__ cursor) {
    const long long e = (long long)blockIdx.x * PT_BLOCK + threadIdx.x;
    const int j = (int)(e / c), ch = (int)(e % c);
    if (j >= n) return;
    const int r = rank[j];
    const int nxt = j + 1 < n ? rank[j + 1] : *total;
    if (nxt == r) return;
    const long long row = (long long)*cursor + r;
    if (row < cap_rows) dst[row * c + ch] = src[(size_t)j * c + ch];
}

__global__ void cursor_add_kernel(int32_t* cursor, const int32_t* total) { *cursor += *total; }

// new[:, 0:2] = fp32(fp64 rotation of x, y), z and column 3 copied, further columns zero
// (rotate_copy, inter_domain_point_polarmix.py:160-188: np.zeros_like + [:, :3] + [:, 3])
__global__ void __launch_bounds__(PT_BLOCK)
points_rotate_z_kernel(const float* __restrict__ src, int n, const int32_t* __restrict__ n_dev, int c, double cosv, double sinv,
                       float* __restrict__ dst) {
    const int rows = eff_n(n, n_dev);
    const int j = blockIdx.x * PT_BLOCK + threadIdx.x;
    if (j >= rows) return;
    const float* p = src + (size_t)j * c;
    const double x = (double)p[0], y = (double)p[1];
    const float z = p[2], f = c > 3 ? p[3] : 0.f;
    float* q = dst + (size_t)j * c;
    q[0] = (float)(x * cosv + y * (-sinv));
    q[1] = (float)(x * sinv + y * cosv);
    q[2] = z;
    if (c > 3) q[3] = f;
    for (int ch = 4; ch < c; ++ch) q[ch] = 0.f;
}

// Global augmentations of a cloud in one pass, applied in the reference's order with fp32 arithmetic
// (augmentor_utils.py:8-81): flip along x (y -> -y), flip along y (x -> -x), rotation about z
// (row vector times [[c, s], [-s, c]], c / s are fp32 values), uniform scaling of x, y, z.
__global__ void __launch_bounds__(PT_BLOCK)
points_world_transform_kernel(const float* __restrict__ src, int n, const int32_t* __restrict__ n_dev, int c, int flip_x,
                              int flip_y, int rotate, float cosv, float sinv, int rescale, float scale, float* __restrict__ dst) {
    const int rows = eff_n(n, n_dev);
    const int j = blockIdx.x * PT_BLOCK + threadIdx.x;
    if (j >= rows) return;
    const float* p = src + (size_t)j * c;
    float x = p[0], y = p[1], z = p[2];
    if (flip_x) y = -y;
    if (flip_y) x = -x;
    if (rotate) {
        const float nx = x * cosv + y * (-sinv);
        const float ny = x * sinv + y * cosv;
        x = nx, y = ny;
    }
    if (rescale) x *= scale, y *= scale, z *= scale;
    float* q = dst + (size_t)j * c;
    q[0] = x, q[1] = y, q[2] = z;
    for (int ch = 3; ch < c; ++ch) q[ch] = p[ch];
}

}  // namespace toda

using namespace toda;

#define PT_COMMON_CHECK(name)                                                                  \
    TODA_CHECK_ARG(n >= 0 && c >= 3, name ": need n >= 0 and at least 3 columns (x, y, z)");    \
    hipStream_t s = (hipStream_t)stream;                                                       \
    if (n == 0) return TODA_OK

extern "C" int toda_points_in_boxes(const float* points, int n, const int32_t* n_dev, int c, const float* boxes, int k,
                                    int box_stride, int mode, int32_t* flags, void* stream) {
    PT_COMMON_CHECK("points_in_boxes");
    TODA_CHECK_ARG(k >= 0 && k <= 4096 && box_stride >= 7 && mode >= 0 && mode <= 2, "points_in_boxes: k in [0,4096], stride >= 7, mode 0|1|2");
    const size_t lds = (size_t)k * sizeof(BoxPre);
    if (mode == 0)
        hipLaunchKernelGGL(points_in_boxes_kernel<0>, dim3(cdiv(n, PT_BLOCK)), dim3(PT_BLOCK), lds, s, points, n, n_dev, c, boxes, k,
                           box_stride, flags);
    else if (mode == 1)
        hipLaunchKernelGGL(points_in_boxes_kernel<1>, dim3(cdiv(n, PT_BLOCK)), dim3(PT_BLOCK), lds, s, points, n, n_dev, c, boxes, k,
                           box_stride, flags);
    else
        hipLaunchKernelGGL(points_in_boxes_kernel<2>, dim3(cdiv(n, PT_BLOCK)), dim3(PT_BLOCK), lds, s, points, n, n_dev, c, boxes, k,
                           box_stride, flags);
    TODA_LAUNCH_CHECK();
    return TODA_OK;
}

extern "C" int toda_points_sector(const float* points, int n, const int32_t* n_dev, int c, double lo, double hi,
                                  int32_t* flags, void* stream) {
    PT_COMMON_CHECK("points_sector");
    hipLaunchKernelGGL(points_sector_kernel, dim3(cdiv(n, PT_BLOCK)), dim3(PT_BLOCK), 0, s, points, n, n_dev, c, lo, hi, flags);
    TODA_LAUNCH_CHECK();
    return TODA_OK;
}

extern "C" int toda_points_rect(const float* points, int n, const int32_t* n_dev, int c, const double* lo_xy_host,
                                const double* hi_xy_host, int closed, int32_t* flags, void* stream) {
    PT_COMMON_CHECK("points_rect");
    hipLaunchKernelGGL(points_rect_kernel, dim3(cdiv(n, PT_BLOCK)), dim3(PT_BLOCK), 0, s, points, n, n_dev, c, lo_xy_host[0],
                       lo_xy_host[1], hi_xy_host[0], hi_xy_host[1], closed, flags);
    TODA_LAUNCH_CHECK();
    return TODA_OK;
}

extern "C" int toda_points_polar_cell(const float* points, int n, const int32_t* n_dev, int c, float phase,
                                      const double* yaw_edges_host, int n_yaw, const double* dis_edges_host, int n_dis,
                                      float dis_lo, float dis_hi, int32_t* cell, void* stream) {
    PT_COMMON_CHECK("points_polar_cell");
    TODA_CHECK_ARG(n_yaw >= 1 && n_yaw < PT_MAX_EDGES && n_dis >= 1 && n_dis < PT_MAX_EDGES, "points_polar_cell: 1..32 bins per axis");
    PolarGrid g;
    for (int i = 0; i <= n_yaw; ++i) g.yaw_edges[i] = yaw_edges_host[i];
    for (int i = 0; i <= n_dis; ++i) g.dis_edges[i] = dis_edges_host[i];
    g.n_yaw = n_yaw, g.n_dis = n_dis, g.phase = phase, g.dis_lo = dis_lo, g.dis_hi = dis_hi;
    hipLaunchKernelGGL(points_polar_cell_kernel, dim3(cdiv(n, PT_BLOCK)), dim3(PT_BLOCK), 0, s, points, n, n_dev, c, g, cell);
    TODA_LAUNCH_CHECK();
    return TODA_OK;
}

extern "C" int toda_points_polar_select(const float* points, int n, const int32_t* n_dev, int c, double lo, double hi, int yaw_mode,
                                        int dis_mode, double dis_th, const float* pitch_range_dev, int32_t* flags, void* stream) {
    PT_COMMON_CHECK("points_polar_select");
    TODA_CHECK_ARG((yaw_mode == 1 || yaw_mode == 2) && dis_mode >= 0 && dis_mode <= 2, "points_polar_select: yaw_mode 1|2, dis_mode 0|1|2");
    hipLaunchKernelGGL(points_polar_select_kernel, dim3(cdiv(n, PT_BLOCK)), dim3(PT_BLOCK), 0, s, points, n, n_dev, c, lo, hi, yaw_mode,
                       dis_mode, dis_th, pitch_range_dev, flags);
    TODA_LAUNCH_CHECK();
    return TODA_OK;
}

constexpr int PT_RANGE_BLOCKS = 256;

extern "C" size_t toda_points_pitch_range_workspace_bytes(void) { return (size_t)PT_RANGE_BLOCKS * 2 * sizeof(float); }

extern "C" int toda_points_pitch_range(const float* points, int n, const int32_t* n_dev, int c, float* range_dev, void* ws,
                                       size_t ws_bytes, void* stream) {
    TODA_CHECK_ARG(n >= 0 && c >= 3, "points_pitch_range: need n >= 0 and at least 3 columns (x, y, z)");
    hipStream_t s = (hipStream_t)stream;
    if (ws_bytes < toda_points_pitch_range_workspace_bytes()) {
        set_error("points_pitch_range: workspace %zu < required %zu", ws_bytes, toda_points_pitch_range_workspace_bytes());
        return TODA_EWORKSPACE;
    }
    const int blocks = n > 0 ? (cdiv(n, PT_BLOCK) < PT_RANGE_BLOCKS ? cdiv(n, PT_BLOCK) : PT_RANGE_BLOCKS) : 1;
    float* partial = (float*)ws;
    hipLaunchKernelGGL(points_pitch_range_kernel, dim3(blocks), dim3(PT_BLOCK), 0, s, points, n, n_dev, c, (const float*)nullptr, 0, partial);
    hipLaunchKernelGGL(points_pitch_range_kernel, dim3(1), dim3(PT_BLOCK), 0, s, points, n, n_dev, c, (const float*)partial, blocks, range_dev);
    TODA_LAUNCH_CHECK();
    return TODA_OK;
}

extern "C" int toda_points_pitch_band(const float* points, int n, const int32_t* n_dev, int c, float z_offset, float clip_lo,
                                      float clip_hi, const double* edges_host, int n_bands, int32_t* band, void* stream) {
    PT_COMMON_CHECK("points_pitch_band");
    TODA_CHECK_ARG(n_bands >= 1 && n_bands < PT_MAX_EDGES, "points_pitch_band: 1..32 bands");
    PitchBands g;
    for (int i = 0; i <= n_bands; ++i) g.edges[i] = edges_host[i];
    g.n = n_bands, g.z_offset = z_offset, g.clip_lo = clip_lo, g.clip_hi = clip_hi;
    hipLaunchKernelGGL(points_pitch_band_kernel, dim3(cdiv(n, PT_BLOCK)), dim3(PT_BLOCK), 0, s, points, n, n_dev, c, g, band);
    TODA_LAUNCH_CHECK();
    return TODA_OK;
}

extern "C" size_t toda_rows_select_workspace_bytes(int n) {
    return align_up((size_t)(n > 0 ? n : 1) * sizeof(int32_t), 256) + scan_partials_bytes(n > 0 ? n : 1) + 256;
}

extern "C" int toda_rows_select_append(const float* src, int n, const int32_t* n_dev, int c, const int32_t* keys, int match,
                                       int invert, float* dst, int cap_rows, int32_t* cursor_dev, void* ws, size_t ws_bytes,
                                       void* stream) {
    TODA_CHECK_ARG(n >= 0 && c >= 1 && cap_rows >= 0, "rows_select_append: bad sizes");
    hipStream_t s = (hipStream_t)stream;
    if (n == 0) return TODA_OK;
    if (ws_bytes < toda_rows_select_workspace_bytes(n)) {
        set_error("rows_select_append: workspace %zu < required %zu", ws_bytes, toda_rows_select_workspace_bytes(n));
        return TODA_EWORKSPACE;
    }
    char* base = (char*)ws;
    int32_t* rank = (int32_t*)base;
    base += align_up((size_t)n * sizeof(int32_t), 256);
    int32_t* partials = (int32_t*)base;
    base += scan_partials_bytes(n);
    int32_t* total = (int32_t*)base;
    hipLaunchKernelGGL(select_mark_kernel, dim3(cdiv(n, PT_BLOCK)), dim3(PT_BLOCK), 0, s, keys, n, n_dev, match, invert, rank);
    int rc = exclusive_scan(PlainAccess{rank}, n, partials, total, s);
    if (rc != TODA_OK) return rc;
    hipLaunchKernelGGL(select_copy_kernel, dim3(cdiv((long long)n * c, PT_BLOCK)), dim3(PT_BLOCK), 0, s, src, n, c, rank, total, dst,
                       cap_rows, cursor_dev);
    hipLaunchKernelGGL(cursor_add_kernel, dim3(1), dim3(1), 0, s, cursor_dev, total);
    TODA_LAUNCH_CHECK();
    return TODA_OK;
}

extern "C" int toda_points_rotate_z(const float* src, int n, const int32_t* n_dev, int c, double cosv, double sinv, float* dst,
                                    void* stream) {
    PT_COMMON_CHECK("points_rotate_z");
    hipLaunchKernelGGL(points_rotate_z_kernel, dim3(cdiv(n, PT_BLOCK)), dim3(PT_BLOCK), 0, s, src, n, n_dev, c, cosv, sinv, dst);
    TODA_LAUNCH_CHECK();
    return TODA_OK;
}

extern "C" int toda_points_world_transform(const float* src, int n, const int32_t* n_dev, int c, int flip_x, int flip_y,
                                           int rotate, float cosv, float sinv, int rescale, float scale, float* dst, void* stream) {
    PT_COMMON_CHECK("points_world_transform");
    hipLaunchKernelGGL(points_world_transform_kernel, dim3(cdiv(n, PT_BLOCK)), dim3(PT_BLOCK), 0, s, src, n, n_dev, c, flip_x, flip_y,
                       rotate, cosv, sinv, rescale, scale, dst);
    TODA_LAUNCH_CHECK();
    return TODA_OK;
}
