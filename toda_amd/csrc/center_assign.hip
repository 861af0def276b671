// CenterHead target assignment on the GPU (reference: a Python loop on the CPU,
// pcdet/models/dense_heads/center_head.py:103-219 + model_utils/centernet_utils.py:9-69).
// One workgroup per sample: boxes are compacted per head in gt order, each box's gaussian window
// is max-blended into the heat-map with integer atomicMax (values are >= 0, so the fp32 bit
// pattern orders like an int and the blend is order independent -> deterministic).
#include <float.h>

#include "common.h"

namespace toda {

constexpr int CA_BLOCK = 256;

struct AssignGeom {
    float x0, y0, vx, vy;
    int fm_w, fm_h, stride, max_objs, min_radius, num_classes;
    double overlap;
};

// centernet_utils.py:9-35 in fp32, same operation order as torch evaluates it.
// __f*_rn intrinsics forbid fma contraction, which would change the rounding.
__device__ float gaussian_radius_f32(float height, float width, double min_overlap) {
    const float hw = __fadd_rn(height, width);
    const float b1 = hw;
    const float c1 = __fdiv_rn(__fmul_rn(__fmul_rn(width, height), (float)(1 - min_overlap)), (float)(1 + min_overlap));
    const float sq1 = __fsqrt_rn(__fsub_rn(__fmul_rn(b1, b1), __fmul_rn(4.0f, c1)));
    const float r1 = __fdiv_rn(__fadd_rn(b1, sq1), 2.0f);
    const float b2 = __fmul_rn(2.0f, hw);
    const float c2 = __fmul_rn(__fmul_rn((float)(1 - min_overlap), width), height);
    const float sq2 = __fsqrt_rn(__fsub_rn(__fmul_rn(b2, b2), __fmul_rn(16.0f, c2)));
    const float r2 = __fdiv_rn(__fadd_rn(b2, sq2), 2.0f);
    const double a3 = 4 * min_overlap;
    const float b3 = __fmul_rn((float)(-2 * min_overlap), hw);
    const float c3 = __fmul_rn(__fmul_rn((float)(min_overlap - 1), width), height);
    const float sq3 = __fsqrt_rn(__fsub_rn(__fmul_rn(b3, b3), __fmul_rn((float)(4 * a3), c3)));
    const float r3 = __fdiv_rn(__fadd_rn(b3, sq3), 2.0f);
    return fminf(fminf(r1, r2), r3);
}

__global__ void __launch_bounds__(CA_BLOCK)
center_assign_kernel(const float* __restrict__ gt, int n_gt, int code, AssignGeom g, float* __restrict__ heatmap,
                     float* __restrict__ ret_boxes, long long* __restrict__ inds, long long* __restrict__ mask) {
    __shared__ int s_pos[CA_BLOCK];
    __shared__ int s_carry;
    const int b = blockIdx.x;
    const float* boxes = gt + (size_t)b * n_gt * code;
    if (threadIdx.x == 0) s_carry = 0;
    __syncthreads();
    for (int start = 0; start < n_gt; start += CA_BLOCK) {
        // --- per-head compaction index of box `gi` (position among boxes of this head, gt order)
        const int gi = start + threadIdx.x;
        int cls = 0;
        if (gi < n_gt) cls = (int)boxes[(size_t)gi * code + code - 1];
        const int in_head = (gi < n_gt && cls >= 1 && cls <= g.num_classes) ? 1 : 0;
        s_pos[threadIdx.x] = in_head;
        __syncthreads();
        if (threadIdx.x == 0) {  // <= 256 entries, serial prefix is plenty
            int run = s_carry;
            for (int t = 0; t < CA_BLOCK; ++t) {
                const int f = s_pos[t];
                s_pos[t] = run;
                run += f;
            }
            s_carry = run;
        }
        __syncthreads();
        const int chunk = min(CA_BLOCK, n_gt - start);
        // --- every box of the chunk is drawn by the whole workgroup
        for (int t = 0; t < chunk; ++t) {
            const float* box = boxes + (size_t)(start + t) * code;
            const int bc = (int)box[code - 1];
            if (bc < 1 || bc > g.num_classes) continue;  // uniform across the block
            const int kk = s_pos[t];
            if (kk >= g.max_objs) continue;
            float cx = __fdiv_rn(__fdiv_rn(__fsub_rn(box[0], g.x0), g.vx), (float)g.stride);
            float cy = __fdiv_rn(__fdiv_rn(__fsub_rn(box[1], g.y0), g.vy), (float)g.stride);
            const float mx = (float)((double)g.fm_w - 0.5), my = (float)((double)g.fm_h - 0.5);
            cx = fminf(fmaxf(cx, 0.0f), mx);
            cy = fminf(fmaxf(cy, 0.0f), my);
            const int ix = (int)cx, iy = (int)cy;
            const float dx = __fdiv_rn(__fdiv_rn(box[3], g.vx), (float)g.stride);
            const float dy = __fdiv_rn(__fdiv_rn(box[4], g.vy), (float)g.stride);
            if (dx <= 0.0f || dy <= 0.0f) continue;
            if (!(0 <= ix && ix <= g.fm_w && 0 <= iy && iy <= g.fm_h)) continue;
            int radius = (int)gaussian_radius_f32(dx, dy, g.overlap);
            if (radius < g.min_radius) radius = g.min_radius;
            const int diameter = 2 * radius + 1;
            const double sigma = (double)diameter / 6.0;
            const int left = min(ix, radius), right = min(g.fm_w - ix, radius + 1);
            const int top = min(iy, radius), bottom = min(g.fm_h - iy, radius + 1);
            const int ww = left + right, hh = top + bottom;
            int* hm = reinterpret_cast<int*>(heatmap + ((size_t)b * g.num_classes + (bc - 1)) * g.fm_h * g.fm_w);
            for (int e = threadIdx.x; e < ww * hh; e += CA_BLOCK) {
                const int xx = e % ww - left, yy = e / ww - top;
                double h = exp(-(double)(xx * xx + yy * yy) / (2.0 * sigma * sigma));
                if (h < DBL_EPSILON) h = 0.0;
                const float hv = (float)h;
                atomicMax(&hm[(size_t)(iy + yy) * g.fm_w + (ix + xx)], __float_as_int(hv));
            }
            if (threadIdx.x == 0) {
                inds[(size_t)b * g.max_objs + kk] = (long long)iy * g.fm_w + ix;
                mask[(size_t)b * g.max_objs + kk] = 1;
                float* rb = ret_boxes + ((size_t)b * g.max_objs + kk) * code;
                rb[0] = __fsub_rn(cx, (float)ix);
                rb[1] = __fsub_rn(cy, (float)iy);
                rb[2] = box[2];
                rb[3] = logf(box[3]);
                rb[4] = logf(box[4]);
                rb[5] = logf(box[5]);
                rb[6] = cosf(box[6]);
                rb[7] = sinf(box[6]);
                for (int e = 8; e < code; ++e) rb[e] = box[e - 1];
            }
        }
        __syncthreads();
    }
}

}  // namespace toda

using namespace toda;

extern "C" int toda_center_assign(const float* gt_boxes, int batch, int n_gt, int code_size, int num_classes, int fm_w,
                                  int fm_h, const float* range_host, const float* vsize_host, int fm_stride,
                                  int max_objs, double gaussian_overlap, int min_radius, float* heatmap,
                                  float* ret_boxes, int64_t* inds, int64_t* mask, void* stream) {
    hipStream_t s = (hipStream_t)stream;
    TODA_CHECK_ARG(batch >= 1 && n_gt >= 0 && code_size >= 8, "center_assign: need batch>=1, code_size>=8");
    TODA_CHECK_ARG(num_classes >= 1 && fm_w >= 1 && fm_h >= 1 && fm_stride >= 1 && max_objs >= 1,
                   "center_assign: bad head geometry");
    TODA_HIP(hipMemsetAsync(heatmap, 0, (size_t)batch * num_classes * fm_h * fm_w * sizeof(float), s));
    TODA_HIP(hipMemsetAsync(ret_boxes, 0, (size_t)batch * max_objs * code_size * sizeof(float), s));
    TODA_HIP(hipMemsetAsync(inds, 0, (size_t)batch * max_objs * sizeof(int64_t), s));
    TODA_HIP(hipMemsetAsync(mask, 0, (size_t)batch * max_objs * sizeof(int64_t), s));
    if (n_gt == 0) return TODA_OK;
    AssignGeom g;
    g.x0 = range_host[0];
    g.y0 = range_host[1];
    g.vx = vsize_host[0];
    g.vy = vsize_host[1];
    g.fm_w = fm_w;
    g.fm_h = fm_h;
    g.stride = fm_stride;
    g.max_objs = max_objs;
    g.min_radius = min_radius;
    g.num_classes = num_classes;
    g.overlap = gaussian_overlap;
    hipLaunchKernelGGL(center_assign_kernel, dim3(batch), dim3(CA_BLOCK), 0, s, gt_boxes, n_gt, code_size, g, heatmap,
                       ret_boxes, (long long*)inds, (long long*)mask);
    TODA_LAUNCH_CHECK();
    return TODA_OK;
}
