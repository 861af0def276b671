// Hard voxelisation + MeanVFE for gfx950.
//
// The reference voxelises sequentially on the CPU (spconv Point2VoxelCPU3d, called from
// pcdet/datasets/processor/data_processor.py:44-60).  The sequential result (voxel ids in
// first-appearance order, first P points kept, voxels past the cap dropped) is reproduced
// bit-exactly by an order-free parallel formulation:
//   1. hash-insert every point's cell key; per slot keep min(point index) and a count
//   2. a point is a "founder" iff it equals its slot's min index; an exclusive scan of the
//      founder flags over the points IS the first-appearance voxel id (no sort needed)
//   3. voxels with id >= cap are dropped (equivalent to spconv's `continue` at the cap)
//   4. points are binned per voxel (CSR via a scan of the counts); each voxel picks its P
//      smallest point indices in ascending order == the first P points of the sequential pass
// All traffic is a few MB of 4-byte index work: HBM/L2-latency bound, no FLOPs.
#include "scan.cuh"

namespace toda {

constexpr int VOX_BLOCK = 256;

struct VoxGeom {
    float r0[3];   // range min xyz
    float vs[3];   // voxel size xyz
    int grid[3];   // cells xyz
};

__device__ __forceinline__ unsigned hash_u32(unsigned k) {
    k ^= k >> 16;
    k *= 0x7feb352dU;
    k ^= k >> 15;
    k *= 0x846ca68bU;
    k ^= k >> 16;
    return k;
}

// 1. hash insert.  keys: -1 = empty.  first: init 0x7f7f7f7f.  cnt: init 0.
__global__ void __launch_bounds__(VOX_BLOCK)
vox_insert_kernel(const float* __restrict__ pts, int n, int c, VoxGeom g, int* __restrict__ keys,
                  int* __restrict__ first, int* __restrict__ cnt, unsigned mask, int* __restrict__ pt_slot) {
    const int i = blockIdx.x * VOX_BLOCK + threadIdx.x;
    if (i >= n) return;
    const float* p = pts + (size_t)i * c;
    int cc[3];
    bool ok = true;
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        // same fp32 expression as the sequential voxeliser: floor((p - lo) / size)
        float f = floorf((p[j] - g.r0[j]) / g.vs[j]);
        ok = ok && (f >= 0.0f) && (f < (float)g.grid[j]);  // NaN fails both
        cc[j] = (int)f;
    }
    if (!ok) {
        pt_slot[i] = -1;
        return;
    }
    const int key = (cc[2] * g.grid[1] + cc[1]) * g.grid[0] + cc[0];
    unsigned s = hash_u32((unsigned)key) & mask;
    while (true) {
        int prev = atomicCAS(&keys[s], -1, key);
        if (prev == -1 || prev == key) break;
        s = (s + 1) & mask;
    }
    atomicMin(&first[s], i);
    atomicAdd(&cnt[s], 1);
    pt_slot[i] = (int)s;
}

// 2. founder flags
__global__ void __launch_bounds__(VOX_BLOCK)
vox_founder_kernel(const int* __restrict__ pt_slot, const int* __restrict__ first, int n, int* __restrict__ flag) {
    const int i = blockIdx.x * VOX_BLOCK + threadIdx.x;
    if (i >= n) return;
    const int s = pt_slot[i];
    flag[i] = (s >= 0 && first[s] == i) ? 1 : 0;
}

// 3. founders publish their voxel: id per slot, count per id, coordinates, clamped count
__global__ void __launch_bounds__(VOX_BLOCK)
vox_publish_kernel(const int* __restrict__ pt_slot, const int* __restrict__ first, const int* __restrict__ rank,
                   const int* __restrict__ keys, const int* __restrict__ cnt, int n, VoxGeom g, int max_pts,
                   int max_voxels, const int* __restrict__ total_dev, int* __restrict__ vid_of_slot,
                   int* __restrict__ cnt_by_vid, int* __restrict__ coords_zyx, int* __restrict__ num_pts,
                   int* __restrict__ m_dev) {
    const int i = blockIdx.x * VOX_BLOCK + threadIdx.x;
    if (i == 0) {
        int t = *total_dev;
        *m_dev = t < max_voxels ? t : max_voxels;
    }
    if (i >= n) return;
    const int s = pt_slot[i];
    if (s < 0 || first[s] != i) return;
    const int vid = rank[i];
    vid_of_slot[s] = vid;
    const int cn = cnt[s];
    cnt_by_vid[vid] = vid < max_voxels ? cn : 0;
    if (vid < max_voxels) {
        int key = keys[s];
        int x = key % g.grid[0];
        key /= g.grid[0];
        int y = key % g.grid[1];
        int z = key / g.grid[1];
        coords_zyx[3 * vid + 0] = z;
        coords_zyx[3 * vid + 1] = y;
        coords_zyx[3 * vid + 2] = x;
        num_pts[vid] = cn < max_pts ? cn : max_pts;
    }
}

// 4a. bin point indices per voxel (order inside a bin is arbitrary; 4b restores it)
__global__ void __launch_bounds__(VOX_BLOCK)
vox_bin_kernel(const int* __restrict__ pt_slot, const int* __restrict__ vid_of_slot, const int* __restrict__ off,
               int n, int max_voxels, int* __restrict__ fill, int* __restrict__ list) {
    const int i = blockIdx.x * VOX_BLOCK + threadIdx.x;
    if (i >= n) return;
    const int s = pt_slot[i];
    if (s < 0) return;
    const int vid = vid_of_slot[s];
    if (vid >= max_voxels) return;
    const int pos = atomicAdd(&fill[vid], 1);
    list[off[vid] + pos] = i;
}

// 4b. one lane group of LPV lanes per voxel: slot r takes the (r+1)-th smallest point index of
// the bin; each lane then copies one point row.  LPV = power of two >= max_pts (<= 64).
template <int LPV>
__global__ void __launch_bounds__(VOX_BLOCK)
vox_emit_kernel(const float* __restrict__ pts, int c, const int* __restrict__ off, const int* __restrict__ cnt_by_vid,
                const int* __restrict__ list, const int* __restrict__ m_dev, int max_pts, float* __restrict__ voxels) {
    const int m = *m_dev;
    const int gid = (blockIdx.x * VOX_BLOCK + threadIdx.x) / LPV;
    const int r = threadIdx.x % LPV;
    if (gid >= m || r >= max_pts) return;
    const int o = off[gid], len = cnt_by_vid[gid];
    float* dst = voxels + ((size_t)gid * max_pts + r) * c;
    if (r >= len) {
        for (int j = 0; j < c; ++j) dst[j] = 0.0f;
        return;
    }
    // rank selection: the element with exactly r smaller elements (indices are unique)
    int pick = -1;
    for (int a = 0; a < len && pick < 0; ++a) {
        const int va = list[o + a];
        int smaller = 0;
        for (int b = 0; b < len; ++b) smaller += list[o + b] < va;
        if (smaller == r) pick = va;
    }
    const float* src = pts + (size_t)pick * c;
    for (int j = 0; j < c; ++j) dst[j] = src[j];
}

// MeanVFE forward: one thread per (voxel, channel)
__global__ void __launch_bounds__(VOX_BLOCK)
mean_vfe_fwd_kernel(const float* __restrict__ voxels, const float* __restrict__ num_pts, int m, int p, int c,
                    float* __restrict__ out) {
    const long long t = (long long)blockIdx.x * VOX_BLOCK + threadIdx.x;
    if (t >= (long long)m * c) return;
    const int v = (int)(t / c), j = (int)(t % c);
    float s = 0.0f;
    for (int q = 0; q < p; ++q) s += voxels[((size_t)v * p + q) * c + j];
    const float nrm = fmaxf(num_pts[v], 1.0f);
    out[t] = s / nrm;
}

__global__ void __launch_bounds__(VOX_BLOCK)
mean_vfe_bwd_kernel(const float* __restrict__ gout, const float* __restrict__ num_pts, int m, int p, int c,
                    float* __restrict__ gvox) {
    const long long t = (long long)blockIdx.x * VOX_BLOCK + threadIdx.x;
    if (t >= (long long)m * p * c) return;
    const int j = (int)(t % c);
    const int v = (int)(t / ((long long)p * c));
    gvox[t] = gout[(size_t)v * c + j] / fmaxf(num_pts[v], 1.0f);
}

struct VoxWs {
    size_t cap;  // hash capacity (power of two)
    size_t o_keys, o_first, o_cnt, o_fill, o_vid, o_slot, o_flag, o_cbv, o_list, o_part, o_total, bytes;
};

static VoxWs vox_layout(int n, int /*max_voxels*/) {
    VoxWs w;
    size_t cap = 1024;
    while (cap < (size_t)2 * (size_t)(n > 0 ? n : 1)) cap <<= 1;
    w.cap = cap;
    size_t o = 0;
    auto take = [&](size_t bytes) {
        size_t at = o;
        o += align_up(bytes, 256);
        return at;
    };
    const size_t nn = (size_t)(n > 0 ? n : 1);
    w.o_keys = take(cap * 4);
    w.o_first = take(cap * 4);
    w.o_cnt = take(cap * 4);   // cnt and fill are zeroed together
    w.o_fill = take(nn * 4);
    w.o_vid = take(cap * 4);
    w.o_slot = take(nn * 4);
    w.o_flag = take(nn * 4);
    w.o_cbv = take((nn + 1) * 4);
    w.o_list = take(nn * 4);
    w.o_part = take(scan_partials_bytes((long long)nn + 1));
    w.o_total = take(256);
    w.bytes = o;
    return w;
}

}  // namespace toda

using namespace toda;

extern "C" size_t toda_voxelize_workspace_bytes(int n_points, int max_voxels) {
    return vox_layout(n_points, max_voxels).bytes;
}

extern "C" int toda_voxelize_hard(const float* points, int n, int c, const float* range_host, const float* vsize_host,
                                  const int32_t* grid_host, int max_pts, int max_voxels, float* voxels,
                                  int32_t* coords_zyx, int32_t* num_pts, int32_t* m_dev, void* ws, size_t ws_bytes,
                                  void* stream) {
    hipStream_t s = (hipStream_t)stream;
    TODA_CHECK_ARG(n >= 0 && c >= 3, "voxelize: need n >= 0 and c >= 3 (got n=%d c=%d)", n, c);
    TODA_CHECK_ARG(max_pts >= 1 && max_pts <= 64, "voxelize: max_pts must be in [1,64] (got %d)", max_pts);
    TODA_CHECK_ARG(max_voxels >= 1, "voxelize: max_voxels must be >= 1");
    TODA_CHECK_ARG((long long)grid_host[0] * grid_host[1] * grid_host[2] < (1LL << 31),
                   "voxelize: grid of one sample must have < 2^31 cells");
    const VoxWs w = vox_layout(n, max_voxels);
    if (ws_bytes < w.bytes) {
        set_error("voxelize: workspace %zu < required %zu", ws_bytes, w.bytes);
        return TODA_EWORKSPACE;
    }
    if (n == 0) {
        TODA_HIP(hipMemsetAsync(m_dev, 0, sizeof(int32_t), s));
        return TODA_OK;
    }
    char* b = (char*)ws;
    int* keys = (int*)(b + w.o_keys);
    int* first = (int*)(b + w.o_first);
    int* cnt = (int*)(b + w.o_cnt);
    int* fill = (int*)(b + w.o_fill);
    int* vid = (int*)(b + w.o_vid);
    int* slot = (int*)(b + w.o_slot);
    int* flag = (int*)(b + w.o_flag);
    int* cbv = (int*)(b + w.o_cbv);
    int* list = (int*)(b + w.o_list);
    int* part = (int*)(b + w.o_part);
    int* total = (int*)(b + w.o_total);

    VoxGeom g;
    for (int j = 0; j < 3; ++j) {
        g.r0[j] = range_host[j];
        g.vs[j] = vsize_host[j];
        g.grid[j] = grid_host[j];
    }
    TODA_HIP(hipMemsetAsync(keys, 0xFF, w.cap * 4, s));
    TODA_HIP(hipMemsetAsync(first, 0x7F, w.cap * 4, s));
    TODA_HIP(hipMemsetAsync(cnt, 0, (w.o_vid - w.o_cnt), s));  // cnt + fill
    TODA_HIP(hipMemsetAsync(cbv, 0, ((size_t)n + 1) * 4, s));

    const int nb = cdiv(n, VOX_BLOCK);
    hipLaunchKernelGGL(vox_insert_kernel, dim3(nb), dim3(VOX_BLOCK), 0, s, points, n, c, g, keys, first, cnt,
                       (unsigned)(w.cap - 1), slot);
    hipLaunchKernelGGL(vox_founder_kernel, dim3(nb), dim3(VOX_BLOCK), 0, s, slot, first, n, flag);
    int rc = exclusive_scan(PlainAccess{flag}, n, part, total, s);
    if (rc) return rc;
    hipLaunchKernelGGL(vox_publish_kernel, dim3(nb), dim3(VOX_BLOCK), 0, s, slot, first, flag, keys, cnt, n, g, max_pts,
                       max_voxels, total, vid, cbv, coords_zyx, num_pts, m_dev);
    // CSR offsets over voxel ids (at most n voxels); cbv keeps the counts, off goes to `flag`
    TODA_HIP(hipMemcpyAsync(flag, cbv, (size_t)n * 4, hipMemcpyDeviceToDevice, s));
    rc = exclusive_scan(PlainAccess{flag}, n, part, nullptr, s);
    if (rc) return rc;
    hipLaunchKernelGGL(vox_bin_kernel, dim3(nb), dim3(VOX_BLOCK), 0, s, slot, vid, flag, n, max_voxels, fill, list);
    int lpv = 1;
    while (lpv < max_pts) lpv <<= 1;
    const int mcap = n < max_voxels ? n : max_voxels;
    const int eb = cdiv((long long)mcap * lpv, VOX_BLOCK);
#define EMIT(L)                                                                                                     \
    hipLaunchKernelGGL(HIP_KERNEL_NAME(vox_emit_kernel<L>), dim3(eb), dim3(VOX_BLOCK), 0, s, points, c, flag, cbv, \
                       list, m_dev, max_pts, voxels)
    switch (lpv) {
        case 1: EMIT(1); break;
        case 2: EMIT(2); break;
        case 4: EMIT(4); break;
        case 8: EMIT(8); break;
        case 16: EMIT(16); break;
        case 32: EMIT(32); break;
        default: EMIT(64); break;
    }
#undef EMIT
    TODA_LAUNCH_CHECK();
    return TODA_OK;
}

extern "C" int toda_mean_vfe_fwd(const float* voxels, const float* num_pts, int m, int p, int c, float* out,
                                 void* stream) {
    TODA_CHECK_ARG(m >= 0 && p >= 1 && c >= 1, "mean_vfe_fwd: bad shape m=%d p=%d c=%d", m, p, c);
    if (m == 0) return TODA_OK;
    hipLaunchKernelGGL(mean_vfe_fwd_kernel, dim3(cdiv((long long)m * c, VOX_BLOCK)), dim3(VOX_BLOCK), 0,
                       (hipStream_t)stream, voxels, num_pts, m, p, c, out);
    TODA_LAUNCH_CHECK();
    return TODA_OK;
}

extern "C" int toda_mean_vfe_bwd(const float* grad_out, const float* num_pts, int m, int p, int c, float* grad_voxels,
                                 void* stream) {
    TODA_CHECK_ARG(m >= 0 && p >= 1 && c >= 1, "mean_vfe_bwd: bad shape m=%d p=%d c=%d", m, p, c);
    if (m == 0) return TODA_OK;
    hipLaunchKernelGGL(mean_vfe_bwd_kernel, dim3(cdiv((long long)m * p * c, VOX_BLOCK)), dim3(VOX_BLOCK), 0,
                       (hipStream_t)stream, grad_out, num_pts, m, p, c, grad_voxels);
    TODA_LAUNCH_CHECK();
    return TODA_OK;
}
