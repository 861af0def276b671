// Rotated BEV IoU and greedy NMS for gfx950 (needed by CenterHead decoding at
// eval time and by pseudo-label generation; reference pcdet/ops/iou3d_nms/src/iou3d_nms_kernel.cu
// + the CPU sweep of iou3d_nms.cpp:100-135).
//
// One wave = 64 lanes = one 64-bit suppression word, so the pairwise pass maps 1:1 onto wave64:
// workgroup (cb, rb) of 64 threads compares 64 row boxes against the 64 column boxes staged in LDS
// and writes one u64 per row.  The greedy sweep, which the reference runs on the HOST after a
// blocking copy, stays on the device: a single wave walks the score-ordered boxes, keeps the
// "removed" bitset in LDS and ORs in the mask row of every survivor (coalesced 8-byte loads); the
// keep list and its length never leave the GPU, so decoding needs no synchronisation.
#include "common.h"

namespace toda {

struct P2 {
    float x, y;
};

__device__ __forceinline__ float cross3(P2 p1, P2 p2, P2 p0) {
    return (p1.x - p0.x) * (p2.y - p0.y) - (p2.x - p0.x) * (p1.y - p0.y);
}

// proper crossing of segments p0p1 and q0q1 (touching end points do not count)
__device__ __forceinline__ bool seg_cross(P2 p1, P2 p0, P2 q1, P2 q0, P2* hit) {
    const bool boxes_meet = fminf(p0.x, p1.x) <= fmaxf(q0.x, q1.x) && fminf(q0.x, q1.x) <= fmaxf(p0.x, p1.x) &&
                            fminf(p0.y, p1.y) <= fmaxf(q0.y, q1.y) && fminf(q0.y, q1.y) <= fmaxf(p0.y, p1.y);
    if (!boxes_meet) return false;
    const float s1 = cross3(q0, p1, p0), s2 = cross3(p1, q1, p0), s3 = cross3(p0, q1, q0), s4 = cross3(q1, p1, q0);
    if (!(s1 * s2 > 0.f && s3 * s4 > 0.f)) return false;
    const float s5 = cross3(q1, p1, p0);
    if (fabsf(s5 - s1) > 1e-8f) {
        hit->x = (s5 * q0.x - s1 * q1.x) / (s5 - s1);
        hit->y = (s5 * q0.y - s1 * q1.y) / (s5 - s1);
    } else {
        const float a0 = p0.y - p1.y, b0 = p1.x - p0.x, c0 = p0.x * p1.y - p1.x * p0.y;
        const float a1 = q0.y - q1.y, b1 = q1.x - q0.x, c1 = q0.x * q1.y - q1.x * q0.y;
        const float D = a0 * b1 - a1 * b0;
        hit->x = (b0 * c1 - b1 * c0) / D;
        hit->y = (a1 * c0 - a0 * c1) / D;
    }
    return true;
}

__device__ __forceinline__ bool inside(const float* box, P2 p) {  // 1e-2 margin as in the reference
    const float c = cosf(-box[6]), s = sinf(-box[6]);
    const float rx = (p.x - box[0]) * c + (p.y - box[1]) * (-s);
    const float ry = (p.x - box[0]) * s + (p.y - box[1]) * c;
    return fabsf(rx) < box[3] / 2 + 1e-2f && fabsf(ry) < box[4] / 2 + 1e-2f;
}

__device__ __forceinline__ void corners(const float* b, P2* c) {
    const float hx = b[3] / 2, hy = b[4] / 2, cs = cosf(b[6]), sn = sinf(b[6]);
    const float lx[4] = {-hx, hx, hx, -hx}, ly[4] = {-hy, -hy, hy, hy};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const float px = b[0] + lx[k], py = b[1] + ly[k];
        c[k].x = (px - b[0]) * cs + (py - b[1]) * (-sn) + b[0];
        c[k].y = (px - b[0]) * sn + (py - b[1]) * cs + b[1];
    }
    c[4] = c[0];
}

// area of the intersection of two rotated rectangles (x, y, z, dx, dy, dz, heading)
__device__ float overlap_area(const float* a, const float* b) {
    P2 ca[5], cb[5], pts[16], ctr = {0.f, 0.f};
    int cnt = 0;
    corners(a, ca);
    corners(b, cb);
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j)
            if (seg_cross(ca[i + 1], ca[i], cb[j + 1], cb[j], &pts[cnt])) {
                ctr.x += pts[cnt].x;
                ctr.y += pts[cnt].y;
                ++cnt;
            }
    for (int k = 0; k < 4; ++k) {
        if (inside(a, cb[k])) {
            ctr.x += cb[k].x;
            ctr.y += cb[k].y;
            pts[cnt++] = cb[k];
        }
        if (inside(b, ca[k])) {
            ctr.x += ca[k].x;
            ctr.y += ca[k].y;
            pts[cnt++] = ca[k];
        }
    }
    if (cnt == 0) return 0.f;
    ctr.x /= cnt;
    ctr.y /= cnt;
    float ang[16];
    for (int i = 0; i < cnt; ++i) ang[i] = atan2f(pts[i].y - ctr.y, pts[i].x - ctr.x);
    for (int j = 0; j < cnt - 1; ++j)  // <= 16 points: exchange sort by polar angle
        for (int i = 0; i < cnt - j - 1; ++i)
            if (ang[i] > ang[i + 1]) {
                const P2 t = pts[i];
                pts[i] = pts[i + 1];
                pts[i + 1] = t;
                const float ta = ang[i];
                ang[i] = ang[i + 1];
                ang[i + 1] = ta;
            }
    float area = 0.f;
    for (int k = 0; k < cnt - 1; ++k)
        area += (pts[k].x - pts[0].x) * (pts[k + 1].y - pts[0].y) - (pts[k].y - pts[0].y) * (pts[k + 1].x - pts[0].x);
    return fabsf(area) / 2.0f;
}

__device__ __forceinline__ float iou_bev(const float* a, const float* b) {
    const float sa = a[3] * a[4], sb = b[3] * b[4], so = overlap_area(a, b);
    return so / fmaxf(sa + sb - so, 1e-8f);
}

__global__ void __launch_bounds__(256)
iou_matrix_kernel(const float* __restrict__ a, int na, const float* __restrict__ b, int nb, float* __restrict__ iou, int area_only) {
    const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
    if (t >= (long long)na * nb) return;
    const int i = (int)(t / nb), j = (int)(t % nb);
    float ba[7], bb[7];
#pragma unroll
    for (int k = 0; k < 7; ++k) {
        ba[k] = a[(size_t)i * 7 + k];
        bb[k] = b[(size_t)j * 7 + k];
    }
    iou[t] = area_only ? overlap_area(ba, bb) : iou_bev(ba, bb);
}

// mask[row * cb + col_block] bit j = IoU(row, col_block*64 + j) > thresh, for j after `row` only
__global__ void __launch_bounds__(64)
nms_mask_kernel(const float* __restrict__ boxes, int n, float thresh, unsigned long long* __restrict__ mask) {
    __shared__ float cols[64 * 7];
    const int cblk = blockIdx.x, rblk = blockIdx.y;
    const int ncol = min(64, n - cblk * 64);
    if (threadIdx.x < ncol)
        for (int k = 0; k < 7; ++k) cols[threadIdx.x * 7 + k] = boxes[(size_t)(cblk * 64 + threadIdx.x) * 7 + k];
    __syncthreads();
    const int row = rblk * 64 + threadIdx.x;
    if (row >= n) return;
    const int cb = (n + 63) / 64;
    unsigned long long word = 0;
    if (cblk >= rblk) {  // only later boxes can be suppressed by `row`
        float me[7];
        for (int k = 0; k < 7; ++k) me[k] = boxes[(size_t)row * 7 + k];
        const int first = cblk == rblk ? threadIdx.x + 1 : 0;
        for (int j = first; j < ncol; ++j)
            if (iou_bev(me, cols + j * 7) > thresh) word |= 1ull << j;
    }
    mask[(size_t)row * cb + cblk] = word;
}

// single wave: greedy sweep in score order; removed-set in LDS
__global__ void __launch_bounds__(64)
nms_sweep_kernel(const unsigned long long* __restrict__ mask, int n, long long* __restrict__ keep,
                 int* __restrict__ n_keep) {
    extern __shared__ unsigned long long removed[];
    const int cb = (n + 63) / 64, lane = threadIdx.x;
    for (int w = lane; w < cb; w += 64) removed[w] = 0ull;
    __builtin_amdgcn_wave_barrier();
    int kept = 0;
    for (int i = 0; i < n; ++i) {
        const unsigned long long word = removed[i >> 6];  // broadcast read
        if (word & (1ull << (i & 63))) continue;          // wave-uniform
        if (lane == 0) keep[kept] = i;
        ++kept;
        for (int w = (i >> 6) + lane; w < cb; w += 64) removed[w] |= mask[(size_t)i * cb + w];
        __builtin_amdgcn_wave_barrier();
    }
    if (lane == 0) *n_keep = kept;
}

}  // namespace toda

using namespace toda;

extern "C" int toda_boxes_iou_bev(const float* boxes_a, int na, const float* boxes_b, int nb, float* iou, void* stream) {
    TODA_CHECK_ARG(na >= 0 && nb >= 0, "boxes_iou_bev: negative size");
    if (na == 0 || nb == 0) return TODA_OK;
    hipLaunchKernelGGL(iou_matrix_kernel, dim3(cdiv((long long)na * nb, 256)), dim3(256), 0, (hipStream_t)stream, boxes_a, na,
                       boxes_b, nb, iou, 0);
    TODA_LAUNCH_CHECK();
    return TODA_OK;
}

extern "C" int toda_boxes_overlap_bev(const float* boxes_a, int na, const float* boxes_b, int nb, float* overlap, void* stream) {
    TODA_CHECK_ARG(na >= 0 && nb >= 0, "boxes_overlap_bev: negative size");
    if (na == 0 || nb == 0) return TODA_OK;
    hipLaunchKernelGGL(iou_matrix_kernel, dim3(cdiv((long long)na * nb, 256)), dim3(256), 0, (hipStream_t)stream, boxes_a, na,
                       boxes_b, nb, overlap, 1);
    TODA_LAUNCH_CHECK();
    return TODA_OK;
}

extern "C" size_t toda_nms_workspace_bytes(int n) {
    const size_t cb = (size_t)(n + 63) / 64;
    return align_up((size_t)(n > 0 ? n : 1) * (cb > 0 ? cb : 1) * sizeof(unsigned long long), 256);
}

extern "C" int toda_nms_rotated(const float* boxes_sorted, int n, float thresh, int64_t* keep, int32_t* n_keep_dev,
                                void* ws, size_t ws_bytes, void* stream) {
    hipStream_t s = (hipStream_t)stream;
    TODA_CHECK_ARG(n >= 0, "nms_rotated: negative size");
    if (n == 0) {
        TODA_HIP(hipMemsetAsync(n_keep_dev, 0, sizeof(int32_t), s));
        return TODA_OK;
    }
    const int cb = (n + 63) / 64;
    TODA_CHECK_ARG((size_t)cb * 8 <= 64 * 1024, "nms_rotated: at most 524288 boxes");
    if (ws_bytes < toda_nms_workspace_bytes(n)) {
        set_error("nms_rotated: workspace %zu < required %zu", ws_bytes, toda_nms_workspace_bytes(n));
        return TODA_EWORKSPACE;
    }
    unsigned long long* mask = (unsigned long long*)ws;
    hipLaunchKernelGGL(nms_mask_kernel, dim3(cb, cb), dim3(64), 0, s, boxes_sorted, n, thresh, mask);
    hipLaunchKernelGGL(nms_sweep_kernel, dim3(1), dim3(64), cb * sizeof(unsigned long long), s, mask, n, (long long*)keep,
                       n_keep_dev);
    TODA_LAUNCH_CHECK();
    return TODA_OK;
}
