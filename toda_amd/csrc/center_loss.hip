// CenterHead.get_loss of one head group (reference pcdet/models/dense_heads/center_head.py:229-262 with
// pcdet/utils/loss_utils.py:264-385): clamp(sigmoid(hm)), the penalty-reduced focal loss over the heat-map and the masked L1
// loss of the regression maps gathered at the object centres - value AND gradient.  As torch operators this is ~110 launches
// of 2-10 us (and as many host dispatches: 2 ms of a 21 ms step during which the device mostly idles); here it is
//   forward   center_loss_hm_kernel     one pass over the heat-map: clamped sigmoid p, per-block fp64 partial sums of the positive /
//                                       negative terms and of the positive count, and dT/dz of every element (unnormalised)
//             center_loss_reg_kernel    one block per sample: the K object slots gather their D regression values from the branch maps,
//                                       masked |pred - target| summed per code dimension (fixed-order tree), sign * mask kept for backward
//             center_loss_final_kernel  folds the partials in block order -> hm_loss, loc_loss, 1 / max(num_pos, 1), 1 / max(num_obj, 1)
//   backward  center_loss_bwd_kernel    dz = dT/dz * (-cls_weight / num_pos * upstream); the regression gradient is scattered back into the
//                                       (zero-filled) branch maps - slots that share a cell are summed in slot order by the first of them
// No float atomics: results are run-to-run identical.
#include "common.h"

namespace toda {

constexpr int CL_MAX_BRANCH = 8;
constexpr int CL_MAX_DIM = 16;
constexpr int CL_BLOCK = 256;
constexpr int CL_ITEMS = 4;

struct CenterLossMaps {
    float* chan[CL_MAX_DIM];       // per code dimension d: channel plane of sample 0 in its branch map [B][c_j][H][W]
                                   // (forward: the regression maps, read; backward: their gradients, written)
    int sample_stride[CL_MAX_DIM]; // c_j * H * W of that branch
};

__device__ __forceinline__ double cl_block_sum(double v, double* red) {
    // fixed butterfly inside the wave, then the waves in order
    for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    __syncthreads();
    if (lane == 0) red[wave] = v;
    __syncthreads();
    double t = 0.0;
    for (int w = 0; w < (int)(blockDim.x >> 6); ++w) t += red[w];
    return t;
}

__global__ void __launch_bounds__(CL_BLOCK)
center_loss_hm_kernel(const float* __restrict__ z, const float* __restrict__ gt, long long n, float* __restrict__ p_out,
                      float* __restrict__ g_out, double* __restrict__ partials) {
    __shared__ double red[CL_BLOCK / 64];
    double pos = 0.0, neg = 0.0, cnt = 0.0;
    const long long base = (long long)blockIdx.x * (CL_BLOCK * CL_ITEMS) + threadIdx.x;
#pragma unroll
    for (int it = 0; it < CL_ITEMS; ++it) {
        const long long i = base + (long long)it * CL_BLOCK;
        if (i >= n) break;
        const float s = 1.0f / (1.0f + expf(-z[i]));
        const bool inside = s >= 1e-4f && s <= 1.0f - 1e-4f;          // torch.clamp passes the gradient on [min, max]
        const float p = fminf(fmaxf(s, 1e-4f), 1.0f - 1e-4f);
        const float t = gt[i];
        const float q = 1.0f - p;
        float dT;
        if (t == 1.0f) {
            const float lp = logf(p);
            pos += (double)(lp * (q * q));
            cnt += 1.0;
            dT = (q * q) / p - 2.0f * q * lp;
        } else if (t < 1.0f) {
            const float w1 = 1.0f - t, w4 = (w1 * w1) * (w1 * w1), lq = logf(q);
            neg += (double)(lq * (p * p) * w4);
            dT = w4 * (2.0f * p * lq - (p * p) / q);
        } else {
            dT = 0.0f;
        }
        p_out[i] = p;
        g_out[i] = inside ? dT * (s * (1.0f - s)) : 0.0f;
    }
    const double a = cl_block_sum(pos, red), b = cl_block_sum(neg, red), c = cl_block_sum(cnt, red);
    if (threadIdx.x == 0) {
        partials[3 * blockIdx.x + 0] = a;
        partials[3 * blockIdx.x + 1] = b;
        partials[3 * blockIdx.x + 2] = c;
    }
}

// grid = B, block = CL_BLOCK.  reg_part[b][0..D-1] = sum over slots of |pred m - target m|, reg_part[b][D] = sum of mask
__global__ void __launch_bounds__(CL_BLOCK)
center_loss_reg_kernel(const CenterLossMaps maps, const long long* __restrict__ inds, const long long* __restrict__ mask,
                       const float* __restrict__ target, int K, int D, int hw, float* __restrict__ sgn, double* __restrict__ reg_part) {
    __shared__ double red[CL_BLOCK / 64];
    const int b = blockIdx.x;
    double acc[CL_MAX_DIM + 1];
#pragma unroll
    for (int d = 0; d <= CL_MAX_DIM; ++d) acc[d] = 0.0;
    for (int k = threadIdx.x; k < K; k += CL_BLOCK) {
        const long long slot = (long long)b * K + k;
        const long long cell = inds[slot];
        const float m = (float)mask[slot];
        acc[CL_MAX_DIM] += (double)m;
#pragma unroll
        for (int d = 0; d < CL_MAX_DIM; ++d) {
            if (d >= D) break;
            const float t = target[slot * D + d];
            const float mm = m * (t != t ? 0.0f : 1.0f);               // mask * (~isnan(target))
            const float diff = maps.chan[d][(size_t)b * maps.sample_stride[d] + cell] * mm - t * mm;
            acc[d] += (double)fabsf(diff);
            sgn[slot * D + d] = (diff > 0.0f ? 1.0f : diff < 0.0f ? -1.0f : diff) * mm;   // sign(NaN) stays NaN, as torch's
        }
    }
#pragma unroll
    for (int d = 0; d <= CL_MAX_DIM; ++d) {
        if (d < D || d == CL_MAX_DIM) {
            const double s = cl_block_sum(acc[d], red);
            if (threadIdx.x == 0) reg_part[(size_t)b * (CL_MAX_DIM + 1) + d] = s;
        }
    }
}

struct CenterLossWeights {
    float code[CL_MAX_DIM];
    float cls_weight, loc_weight;
};

// out[0] = hm_loss, out[1] = loc_loss, out[2] = 1 / max(num_pos, 1), out[3] = 1 / max(num_obj, 1).  One wave: lane l adds
// partials l, l + 64, ... in that order, then a fixed butterfly.
__device__ __forceinline__ double cl_wave_sum(double v) {
    for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
    return v;
}

__global__ void __launch_bounds__(64)
center_loss_final_kernel(const double* __restrict__ partials, int n_blocks, const double* __restrict__ reg_part, int B, int D,
                         const CenterLossWeights w, float* __restrict__ out) {
    const int lane = threadIdx.x;
    double pos = 0.0, neg = 0.0, cnt = 0.0;
    for (int i = lane; i < n_blocks; i += 64) {
        pos += partials[3 * i + 0];
        neg += partials[3 * i + 1];
        cnt += partials[3 * i + 2];
    }
    pos = cl_wave_sum(pos), neg = cl_wave_sum(neg), cnt = cl_wave_sum(cnt);
    // lane d < D: code dimension d; lane 63: the object count
    double mine = 0.0;
    const int col = lane == 63 ? CL_MAX_DIM : lane;
    if (lane < D || lane == 63)
        for (int b = 0; b < B; ++b) mine += reg_part[(size_t)b * (CL_MAX_DIM + 1) + col];
    const double num = __shfl(mine, 63, 64);
    const double nn = num > 1.0 ? num : 1.0;
    double term = lane < D ? (double)(float)(mine / nn) * (double)w.code[lane < CL_MAX_DIM ? lane : 0] : 0.0;
    double loc = 0.0;
    for (int d = 0; d < D; ++d) loc += __shfl(term, d, 64);       // in code order
    if (lane == 0) {
        const double np = cnt > 1.0 ? cnt : 1.0;
        out[0] = (float)(-(pos + neg) / np * (double)w.cls_weight);
        out[1] = (float)(loc * (double)w.loc_weight);
        out[2] = (float)(1.0 / np);
        out[3] = (float)(1.0 / nn);
    }
}

// blocks [0, hm_blocks): dz = g * (-cls_weight * out[2] * up[0]).  blocks [hm_blocks, hm_blocks + B): the slots of one sample.
__global__ void __launch_bounds__(CL_BLOCK)
center_loss_bwd_kernel(const float* __restrict__ g, long long n, int hm_blocks, const float* __restrict__ out, const float* __restrict__ up_hm,
                       const float* __restrict__ up_loc, const CenterLossWeights w, const CenterLossMaps grads, const long long* __restrict__ inds,
                       const long long* __restrict__ mask, const float* __restrict__ sgn, int B, int K, int D, int hw, float* __restrict__ dz) {
    if ((int)blockIdx.x < hm_blocks) {
        const float scale = -w.cls_weight * out[2] * up_hm[0];
        const long long base = (long long)blockIdx.x * (CL_BLOCK * CL_ITEMS) + threadIdx.x;
#pragma unroll
        for (int it = 0; it < CL_ITEMS; ++it) {
            const long long i = base + (long long)it * CL_BLOCK;
            if (i < n) dz[i] = g[i] * scale;
        }
        return;
    }
    // one block per sample: the K cell indices in LDS, then every slot looks for an earlier slot on the same cell (owner = first)
    extern __shared__ int cells[];
    const int b = blockIdx.x - hm_blocks;
    const long long* row = inds + (long long)b * K;
    for (int k = threadIdx.x; k < K; k += CL_BLOCK) cells[k] = mask[(long long)b * K + k] != 0 ? (int)row[k] : -1;   // empty slots: zero gradient
    __syncthreads();
    const float scale = w.loc_weight * out[3] * up_loc[0];
    for (int k = threadIdx.x; k < K; k += CL_BLOCK) {
        const int cell = cells[k];
        if (cell < 0) continue;
        bool owner = true;
        for (int k2 = 0; k2 < k; ++k2) owner = owner && cells[k2] != cell;
        if (!owner) continue;
        float acc[CL_MAX_DIM];
#pragma unroll
        for (int d = 0; d < CL_MAX_DIM; ++d) acc[d] = 0.0f;
        for (int k2 = k; k2 < K; ++k2) {
            if (cells[k2] != cell) continue;
            const float* sg = sgn + ((long long)b * K + k2) * D;
#pragma unroll
            for (int d = 0; d < CL_MAX_DIM; ++d)
                if (d < D) acc[d] += sg[d] * (w.code[d] * scale);
        }
#pragma unroll
        for (int d = 0; d < CL_MAX_DIM; ++d)
            if (d < D) grads.chan[d][(size_t)b * grads.sample_stride[d] + cell] = acc[d];
    }
}

static int cl_check(const char* who, int n_branch, const int32_t* ch, int batch, int classes, int H, int W, int K, int D) {
    TODA_CHECK_ARG(n_branch >= 1 && n_branch <= CL_MAX_BRANCH, "%s: 1..%d regression branches (got %d)", who, CL_MAX_BRANCH, n_branch);
    int sum = 0;
    for (int j = 0; j < n_branch; ++j) {
        TODA_CHECK_ARG(ch[j] >= 1, "%s: branch %d has %d channels", who, j, ch[j]);
        sum += ch[j];
    }
    TODA_CHECK_ARG(sum == D && D <= CL_MAX_DIM, "%s: branch channels sum to %d, code size is %d (at most %d)", who, sum, D, CL_MAX_DIM);
    TODA_CHECK_ARG(batch >= 1 && classes >= 1 && H >= 1 && W >= 1 && K >= 1 && K <= 8192, "%s: empty geometry or more than 8192 object slots", who);
    TODA_CHECK_ARG((long long)H * W < (1LL << 31), "%s: feature map too large", who);
    return TODA_OK;
}

}  // namespace toda

using namespace toda;

extern "C" size_t toda_center_loss_workspace_bytes(int batch, int classes, int H, int W, int max_objs, int code_size) {
    const long long n = (long long)batch * classes * H * W;
    const size_t blocks = (size_t)cdiv(n, CL_BLOCK * CL_ITEMS);
    // dT/dz of the heat-map | sign * mask of the regression slots | hm partials | reg partials
    return align_up((size_t)n * 4, 256) + align_up((size_t)batch * max_objs * code_size * 4, 256) + align_up(blocks * 3 * 8, 256) +
           align_up((size_t)batch * (CL_MAX_DIM + 1) * 8, 256);
}

struct ClLayout {
    float* g;
    float* sgn;
    double* partials;
    double* reg_part;
    int blocks;
};

static ClLayout cl_layout(void* ws, int batch, int classes, int H, int W, int K, int D) {
    const long long n = (long long)batch * classes * H * W;
    ClLayout l;
    char* p = (char*)ws;
    l.blocks = cdiv(n, CL_BLOCK * CL_ITEMS);
    l.g = (float*)p;
    p += align_up((size_t)n * 4, 256);
    l.sgn = (float*)p;
    p += align_up((size_t)batch * K * D * 4, 256);
    l.partials = (double*)p;
    p += align_up((size_t)l.blocks * 3 * 8, 256);
    l.reg_part = (double*)p;
    return l;
}

extern "C" int toda_center_loss_fwd(const float* hm_logits, const float* heatmap, int batch, int classes, int H, int W, int n_branch,
                                    const float* const* reg_maps, const int32_t* reg_channels, const int64_t* inds, const int64_t* mask,
                                    const float* target_boxes, int max_objs, int code_size, const float* code_weights, float cls_weight,
                                    float loc_weight, float* hm_prob, float* out4, void* ws, size_t ws_bytes, void* stream) {
    int rc = cl_check("center_loss_fwd", n_branch, reg_channels, batch, classes, H, W, max_objs, code_size);
    if (rc) return rc;
    TODA_CHECK_ARG(hm_logits && heatmap && reg_maps && inds && mask && target_boxes && code_weights && hm_prob && out4, "center_loss_fwd: null pointer");
    if (!ws || ws_bytes < toda_center_loss_workspace_bytes(batch, classes, H, W, max_objs, code_size)) {
        toda::set_error("center_loss_fwd: workspace too small");
        return TODA_EWORKSPACE;
    }
    const ClLayout l = cl_layout(ws, batch, classes, H, W, max_objs, code_size);
    CenterLossMaps maps = {};
    for (int j = 0, d = 0; j < n_branch; ++j) {
        TODA_CHECK_ARG(reg_maps[j] != nullptr, "center_loss_fwd: null regression map");
        for (int c = 0; c < reg_channels[j]; ++c, ++d) {
            maps.chan[d] = const_cast<float*>(reg_maps[j]) + (size_t)c * H * W;
            maps.sample_stride[d] = reg_channels[j] * H * W;
        }
    }
    CenterLossWeights w = {};
    for (int d = 0; d < code_size; ++d) w.code[d] = code_weights[d];
    w.cls_weight = cls_weight, w.loc_weight = loc_weight;
    hipStream_t s = (hipStream_t)stream;
    const long long n = (long long)batch * classes * H * W;
    hipLaunchKernelGGL(center_loss_hm_kernel, dim3(l.blocks), dim3(CL_BLOCK), 0, s, hm_logits, heatmap, n, hm_prob, l.g, l.partials);
    hipLaunchKernelGGL(center_loss_reg_kernel, dim3(batch), dim3(CL_BLOCK), 0, s, maps, (const long long*)inds, (const long long*)mask, target_boxes,
                       max_objs, code_size, H * W, l.sgn, l.reg_part);
    hipLaunchKernelGGL(center_loss_final_kernel, dim3(1), dim3(64), 0, s, (const double*)l.partials, l.blocks, (const double*)l.reg_part, batch,
                       code_size, w, out4);
    TODA_LAUNCH_CHECK();
    return TODA_OK;
}

extern "C" int toda_center_loss_bwd(const float* out4, const float* up_hm, const float* up_loc, int batch, int classes, int H, int W,
                                    int n_branch, float* const* reg_grads, const int32_t* reg_channels, const int64_t* inds, const int64_t* mask,
                                    int max_objs, int code_size, const float* code_weights, float cls_weight, float loc_weight, float* hm_grad, const void* ws,
                                    size_t ws_bytes, void* stream) {
    int rc = cl_check("center_loss_bwd", n_branch, reg_channels, batch, classes, H, W, max_objs, code_size);
    if (rc) return rc;
    TODA_CHECK_ARG(out4 && up_hm && up_loc && reg_grads && inds && mask && code_weights && hm_grad, "center_loss_bwd: null pointer");
    if (!ws || ws_bytes < toda_center_loss_workspace_bytes(batch, classes, H, W, max_objs, code_size)) {
        toda::set_error("center_loss_bwd: workspace too small");
        return TODA_EWORKSPACE;
    }
    const ClLayout l = cl_layout(const_cast<void*>(ws), batch, classes, H, W, max_objs, code_size);
    CenterLossMaps grads = {};
    hipStream_t s = (hipStream_t)stream;
    for (int j = 0, d = 0; j < n_branch; ++j) {
        TODA_CHECK_ARG(reg_grads[j] != nullptr, "center_loss_bwd: null gradient map");
        for (int c = 0; c < reg_channels[j]; ++c, ++d) {
            grads.chan[d] = reg_grads[j] + (size_t)c * H * W;
            grads.sample_stride[d] = reg_channels[j] * H * W;
        }
    }
    // zero fill: one memset per run of maps that follow each other in memory (the caller carves them from one allocation)
    for (int j = 0; j < n_branch;) {
        size_t bytes = (size_t)batch * reg_channels[j] * H * W * sizeof(float);
        int e = j + 1;
        while (e < n_branch && (char*)reg_grads[e] == (char*)reg_grads[j] + bytes) bytes += (size_t)batch * reg_channels[e++] * H * W * sizeof(float);
        TODA_HIP(hipMemsetAsync(reg_grads[j], 0, bytes, s));
        j = e;
    }
    CenterLossWeights w = {};
    for (int d = 0; d < code_size; ++d) w.code[d] = code_weights[d];
    w.cls_weight = cls_weight, w.loc_weight = loc_weight;
    const long long n = (long long)batch * classes * H * W;
    hipLaunchKernelGGL(center_loss_bwd_kernel, dim3(l.blocks + batch), dim3(CL_BLOCK), (size_t)max_objs * sizeof(int), s, (const float*)l.g, n, l.blocks, out4, up_hm, up_loc,
                       w, grads, (const long long*)inds, (const long long*)mask, (const float*)l.sgn, batch, max_objs, code_size, H * W, hm_grad);
    TODA_LAUNCH_CHECK();
    return TODA_OK;
}
