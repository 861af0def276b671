// The optimizer step of the reference trainers as two launches: clip_grad_norm_ (tools/train_utils/train_utils.py:57) + the
// OptimWrapper's decoupled weight decay + Adam (tools/train_utils/optimization/fastai_optim.py:104-236 over torch.optim.Adam)
// over ALL parameter tensors of the model.  Through torch the same step is 22 launches (foreach norms, stack, norm, reciprocal,
// clamp, foreach scale, foreach decay per group, fused Adam per group and dtype bucket, step counters) and reads the gradients
// three times; here:
//   pass 1  clip_adam_norm_kernel    sum of squares of every 8192-element chunk of every gradient -> one partial per block
//   pass 2  clip_adam_update_kernel  every block folds the partials in index order (fixed order: deterministic) -> total norm,
//                                    coefficient min(1, max_norm / (norm + 1e-6)); then g *= coef (written back, as the reference
//                                    leaves it), p *= 1 - lr * wd, m = m + (g - m)(1 - b1), v = b2 v + (1 - b2) g^2,
//                                    p -= (lr / (1 - b1^t)) * m / (sqrt(v) / sqrt(1 - b2^t) + eps)      (torch's fused kernel's form)
// The tensors are described by a pointer table the host rebuilds every step (gradient storage is reallocated by zero_grad).
#include "common.h"

namespace toda {

constexpr int OPT_BLOCK = 256;
constexpr int OPT_CHUNK = 8192;          // elements per block: 32 per thread

typedef float f32x4 __attribute__((ext_vector_type(4)));

struct OptTable {
    const unsigned long long* param;     // [n] device addresses
    const unsigned long long* grad;
    const unsigned long long* exp_avg;
    const unsigned long long* exp_avg_sq;
    const long long* numel;              // [n]
    const int* chunk_tensor;             // [n_chunks]
    const long long* chunk_off;          // [n_chunks] first element of the chunk inside its tensor
};

__device__ __forceinline__ double block_sum(double v, double* sh) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) sh[wave] = v;
    __syncthreads();
    double s = 0.0;
    if (threadIdx.x == 0) {
#pragma unroll
        for (int w = 0; w < OPT_BLOCK / 64; ++w) s += sh[w];
    }
    return s;            // valid in thread 0
}

__global__ void __launch_bounds__(OPT_BLOCK)
clip_adam_norm_kernel(const OptTable t, double* __restrict__ partial) {
    __shared__ double sh[OPT_BLOCK / 64];
    const int b = blockIdx.x;
    const int ti = t.chunk_tensor[b];
    const long long off = t.chunk_off[b];
    const long long left = t.numel[ti] - off;
    const int len = left < OPT_CHUNK ? (int)left : OPT_CHUNK;
    const float* g = reinterpret_cast<const float*>(t.grad[ti]) + off;
    float s0 = 0.f, s1 = 0.f;
    if ((reinterpret_cast<unsigned long long>(g) & 15ull) == 0) {
        const f32x4* g4 = reinterpret_cast<const f32x4*>(g);
        const int n4 = len >> 2;
        for (int i = threadIdx.x; i < n4; i += 2 * OPT_BLOCK) {
            const f32x4 a = g4[i];
            s0 += a[0] * a[0] + a[1] * a[1] + a[2] * a[2] + a[3] * a[3];
            if (i + OPT_BLOCK < n4) {
                const f32x4 c = g4[i + OPT_BLOCK];
                s1 += c[0] * c[0] + c[1] * c[1] + c[2] * c[2] + c[3] * c[3];
            }
        }
        for (int i = (n4 << 2) + threadIdx.x; i < len; i += OPT_BLOCK) s0 += g[i] * g[i];
    } else {
        for (int i = threadIdx.x; i < len; i += OPT_BLOCK) s0 += g[i] * g[i];
    }
    const double s = block_sum((double)s0 + (double)s1, sh);
    if (threadIdx.x == 0) partial[b] = s;
}

struct OptHyper {
    float max_norm;          // <= 0: no clipping
    float lr, beta1, beta2, eps;
    float decay;             // p *= decay before the Adam update (1 - lr * wd)
    float bias1, bias2_sqrt; // 1 - beta1^t, sqrt(1 - beta2^t)
};

__global__ void __launch_bounds__(OPT_BLOCK)
clip_adam_update_kernel(const OptTable t, const double* __restrict__ partial, int n_chunks, const OptHyper h, float* __restrict__ norm_out) {
    __shared__ double sh[OPT_BLOCK / 64];
    __shared__ float s_coef;
    {
        // every block folds all partials in the same order: thread i takes i, i + 256, ...; the block tree is fixed
        double v = 0.0;
        for (int i = threadIdx.x; i < n_chunks; i += OPT_BLOCK) v += partial[i];
        const double tot = block_sum(v, sh);
        if (threadIdx.x == 0) {
            const float norm = (float)sqrt(tot);
            float coef = 1.0f;
            if (h.max_norm > 0.f) {
                coef = h.max_norm / (norm + 1e-6f);
                if (coef > 1.0f) coef = 1.0f;
            }
            s_coef = coef;
            if (blockIdx.x == 0) norm_out[0] = norm;
        }
        __syncthreads();
    }
    const float coef = s_coef;
    const int b = blockIdx.x;
    const int ti = t.chunk_tensor[b];
    const long long off = t.chunk_off[b];
    const long long left = t.numel[ti] - off;
    const int len = left < OPT_CHUNK ? (int)left : OPT_CHUNK;
    float* p = reinterpret_cast<float*>(t.param[ti]) + off;
    float* g = reinterpret_cast<float*>(t.grad[ti]) + off;
    float* m = reinterpret_cast<float*>(t.exp_avg[ti]) + off;
    float* v = reinterpret_cast<float*>(t.exp_avg_sq[ti]) + off;
    const float step_size = h.lr / h.bias1;
    auto one = [&](float& pp, float& gg, float& mm, float& vv) {
        gg *= coef;
        pp *= h.decay;
        mm = mm + (gg - mm) * (1.0f - h.beta1);
        vv = vv * h.beta2 + (1.0f - h.beta2) * gg * gg;
        const float denom = sqrtf(vv) / h.bias2_sqrt + h.eps;
        pp -= step_size * (mm / denom);
    };
    const bool vec = ((reinterpret_cast<unsigned long long>(p) | reinterpret_cast<unsigned long long>(g) | reinterpret_cast<unsigned long long>(m) |
                       reinterpret_cast<unsigned long long>(v)) & 15ull) == 0;
    int done = 0;
    if (vec) {
        const int n4 = len >> 2;
        f32x4 *p4 = reinterpret_cast<f32x4*>(p), *g4 = reinterpret_cast<f32x4*>(g), *m4 = reinterpret_cast<f32x4*>(m), *v4 = reinterpret_cast<f32x4*>(v);
        for (int i = threadIdx.x; i < n4; i += OPT_BLOCK) {
            f32x4 pp = p4[i], gg = g4[i], mm = m4[i], vv = v4[i];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float a = pp[j], b2 = gg[j], c = mm[j], d = vv[j];
                one(a, b2, c, d);
                pp[j] = a, gg[j] = b2, mm[j] = c, vv[j] = d;
            }
            p4[i] = pp;
            g4[i] = gg;
            m4[i] = mm;
            v4[i] = vv;
        }
        done = n4 << 2;
    }
    for (int i = done + threadIdx.x; i < len; i += OPT_BLOCK) one(p[i], g[i], m[i], v[i]);
}

}  // namespace toda

using namespace toda;

extern "C" int toda_clip_adam_chunk(void) { return OPT_CHUNK; }

// table: 7 device arrays (see OptTable); partial: n_chunks doubles of scratch; norm_out: 1 float (the total gradient norm before
// clipping, what clip_grad_norm_ returns).  step >= 1 is the Adam step count AFTER this update (torch increments first).
extern "C" int toda_clip_adam_step(const unsigned long long* param, const unsigned long long* grad, const unsigned long long* exp_avg,
                                   const unsigned long long* exp_avg_sq, const long long* numel, const int* chunk_tensor,
                                   const long long* chunk_off, int n_chunks, double* partial, float* norm_out, float max_norm, float lr,
                                   float beta1, float beta2, float eps, float weight_decay, int step, void* stream) {
    TODA_CHECK_ARG(param && grad && exp_avg && exp_avg_sq && numel && chunk_tensor && chunk_off && partial && norm_out, "clip_adam_step: null argument");
    TODA_CHECK_ARG(n_chunks >= 0 && step >= 1, "clip_adam_step: n_chunks %d, step %d", n_chunks, step);
    if (n_chunks == 0) return TODA_OK;
    const OptTable t{param, grad, exp_avg, exp_avg_sq, numel, chunk_tensor, chunk_off};
    OptHyper h;
    h.max_norm = max_norm;
    h.lr = lr, h.beta1 = beta1, h.beta2 = beta2, h.eps = eps;
    h.decay = 1.0f - lr * weight_decay;
    h.bias1 = (float)(1.0 - pow((double)beta1, (double)step));
    h.bias2_sqrt = (float)sqrt(1.0 - pow((double)beta2, (double)step));
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(clip_adam_norm_kernel, dim3(n_chunks), dim3(OPT_BLOCK), 0, s, t, partial);
    hipLaunchKernelGGL(clip_adam_update_kernel, dim3(n_chunks), dim3(OPT_BLOCK), 0, s, t, partial, n_chunks, h, norm_out);
    TODA_LAUNCH_CHECK();
    return TODA_OK;
}
