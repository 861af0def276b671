// Single-pass BatchNorm2d(+ReLU) for the dense half (BEV neck, CenterHead): training-mode nn.BatchNorm2d on an NCHW fp32
// tensor followed by nn.ReLU (reference pcdet/models/backbones_2d/base_bev_backbone.py:37-58, dense_heads/center_head.py:20-28,73-80).
//
// One 1024-thread workgroup owns ONE channel: its B x H x W values (70 k floats at 2 x 188 x 188) are loaded once into
// registers (<= 18 float4 per thread), the mean and then the centred second moment are reduced from there (fp32 inside a
// thread, fp64 across the workgroup - the two-pass form costs nothing when the data sits in registers), and the normalised,
// rectified result is written straight from the registers: one read and one write of the tensor, no partial-sum buffers,
// no second launch.  Against the MIOpen BatchNorm + clamp pair that is half the traffic forward (72 instead of 144 MB at
// 2 x 128 x 188 x 188).  Backward keeps the masked dy in registers, reduces (sum g, sum g * xhat) while x streams past, then
// streams x a second time (it comes from the L2 / infinity cache: a channel is 283 KB) to write dx: x twice, dy once, dx once
// instead of MIOpen's BatchNorm backward + threshold_backward (216 MB -> 144 MB).  The ReLU mask is recomputed from x with
// the forward's own expression (x * scale + shift, same operation order, -ffp-contract=off), so y is never read back.
// Deterministic: fixed reduction tree, no atomics.
#include <type_traits>

#include "common.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace toda {
constexpr int BN2_BLOCK = 1024;

// sum of v over the workgroup in fp64, broadcast to every thread (two barriers)
__device__ __forceinline__ double bn2_block_sum(double v, double* sh) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    __syncthreads();                    // sh may still be read from the previous call
    if (lane == 0) sh[wave] = v;
    __syncthreads();
    double t = 0.0;
#pragma unroll
    for (int w = 0; w < BN2_BLOCK / 64; ++w) t += sh[w];      // same order in every thread
    return t;
}

// A plane (one channel of one sample) as a bounds-checked buffer: lanes past its end read zeros and their stores are dropped,
// so neither kernel has a branch (or an address select) around a memory instruction.  The whole offset goes into the VECTOR
// offset: the range check of a raw buffer covers vector + immediate offset only, a scalar offset would walk into the next plane.
__device__ __forceinline__ __amdgpu_buffer_rsrc_t bn2_plane(const float* base, size_t plane, int hw4) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(base + plane * (size_t)hw4 * 4), 0, hw4 * 16, 0x00020000);
}
template <int AUX>
__device__ __forceinline__ f32x4 bn2_load(__amdgpu_buffer_rsrc_t rs, unsigned voff, int k) {
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    return __builtin_bit_cast(f32x4, (u32x4)__builtin_amdgcn_raw_buffer_load_b128(rs, voff + (unsigned)k * (BN2_BLOCK * 16u), 0, AUX));
}
__device__ __forceinline__ void bn2_store(__amdgpu_buffer_rsrc_t rs, unsigned voff, int k, f32x4 v) {
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), rs, voff + (unsigned)k * (BN2_BLOCK * 16u), 0, 0);
}
constexpr int BN2_NT = 2;      // aux bit 1 (slc / nt): streamed once

template <int BB, int K, bool RELU>
__global__ void __launch_bounds__(BN2_BLOCK)
bn2d_fwd_kernel(const float* __restrict__ x, int C, int hw4, const float* __restrict__ gamma, const float* __restrict__ beta,
                float* __restrict__ running_mean, float* __restrict__ running_var, float momentum, float eps,
                float* __restrict__ y, float* __restrict__ save) {
    __shared__ double sh[BN2_BLOCK / 64];
    const int c = blockIdx.x, t = threadIdx.x;
    const unsigned voff = (unsigned)t * 16u;
    f32x4 r[BB][K];
#pragma unroll
    for (int b = 0; b < BB; ++b) {
        const __amdgpu_buffer_rsrc_t rs = bn2_plane(x, (size_t)b * C + c, hw4);
#pragma unroll
        for (int k = 0; k < K; ++k) r[b][k] = bn2_load<BN2_NT>(rs, voff, k);
    }
    const double n = (double)BB * hw4 * 4.0;
    float s = 0.f;
#pragma unroll
    for (int b = 0; b < BB; ++b)
#pragma unroll
        for (int k = 0; k < K; ++k) s += (r[b][k][0] + r[b][k][1]) + (r[b][k][2] + r[b][k][3]);
    const double mean_d = bn2_block_sum((double)s, sh) / n;
    const float mean = (float)mean_d;
    float q = 0.f;
#pragma unroll
    for (int b = 0; b < BB; ++b)
#pragma unroll
        for (int k = 0; k < K; ++k) {
            const bool ok = t + k * BN2_BLOCK < hw4;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float d = ok ? r[b][k][j] - mean : 0.f;
                q += d * d;
            }
        }
    // sum (x - mean_f)^2 = sum (x - mean_d)^2 + n (mean_d - mean_f)^2: take the rounding of the fp32 mean out again
    const double dm = mean_d - (double)mean;
    const double var_d = bn2_block_sum((double)q, sh) / n - dm * dm;
    const float var = (float)(var_d > 0.0 ? var_d : 0.0);
    const float invstd = 1.0f / sqrtf(var + eps);
    const float scale = gamma[c] * invstd;
    const float shift = beta[c] - mean * scale;
#pragma unroll
    for (int b = 0; b < BB; ++b) {
        const __amdgpu_buffer_rsrc_t rs = bn2_plane(y, (size_t)b * C + c, hw4);
#pragma unroll
        for (int k = 0; k < K; ++k) {
            f32x4 o;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float z = r[b][k][j] * scale + shift;
                o[j] = RELU ? (z > 0.f ? z : 0.f) : z;
            }
            bn2_store(rs, voff, k, o);
        }
    }
    if (t == 0) {
        save[c] = mean;
        save[C + c] = invstd;
        if (running_mean) {       // as nn.BatchNorm2d: biased variance normalises, the unbiased one goes into the running estimate
            const double unbiased = n > 1.0 ? (var_d > 0.0 ? var_d : 0.0) * n / (n - 1.0) : 0.0;
            running_mean[c] = (1.0f - momentum) * running_mean[c] + momentum * mean;
            running_var[c] = (1.0f - momentum) * running_var[c] + momentum * (float)unbiased;
        }
    }
}

template <int BB, int K, bool RELU>
__global__ void __launch_bounds__(BN2_BLOCK)
bn2d_bwd_kernel(const float* __restrict__ x, const float* __restrict__ dy, int C, int hw4, const float* __restrict__ gamma,
                const float* __restrict__ beta, const float* __restrict__ save, float* __restrict__ dx,
                float* __restrict__ dgamma, float* __restrict__ dbeta) {
    __shared__ double sh[BN2_BLOCK / 64];
    const int c = blockIdx.x, t = threadIdx.x;
    const unsigned voff = (unsigned)t * 16u;
    const float mean = save[c], invstd = save[C + c];
    const float scale = gamma[c] * invstd;
    const float shift = beta[c] - mean * scale;
    f32x4 g[BB][K];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int b = 0; b < BB; ++b) {
        const __amdgpu_buffer_rsrc_t rx = bn2_plane(x, (size_t)b * C + c, hw4), rg = bn2_plane(dy, (size_t)b * C + c, hw4);
#pragma unroll
        for (int k = 0; k < K; ++k) {
            const f32x4 xv = bn2_load<0>(rx, voff, k);                  // kept in the caches for the second sweep
            f32x4 gv = bn2_load<BN2_NT>(rg, voff, k);                   // past the end of the plane: zeros
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if (RELU) gv[j] = xv[j] * scale + shift > 0.f ? gv[j] : 0.f;      // the forward's expression, bit for bit
                s1 += gv[j];
                s2 += gv[j] * ((xv[j] - mean) * invstd);
            }
            g[b][k] = gv;
            // at most three (x, dy) pairs in flight on top of the g image (128 registers at 16 waves per workgroup)
            if ((b * K + k) % 3 == 2) __builtin_amdgcn_sched_barrier(0);
        }
    }
    const double n = (double)BB * hw4 * 4.0;
    const double sum_g = bn2_block_sum((double)s1, sh);
    const double sum_gx = bn2_block_sum((double)s2, sh);
    const float m1 = (float)(sum_g / n), m2 = (float)(sum_gx / n);
#pragma unroll
    for (int b = 0; b < BB; ++b) {
        const __amdgpu_buffer_rsrc_t rx = bn2_plane(x, (size_t)b * C + c, hw4), rd = bn2_plane(dx, (size_t)b * C + c, hw4);
#pragma unroll
        for (int k = 0; k < K; ++k) {
            const f32x4 xv = bn2_load<BN2_NT>(rx, voff, k);
            f32x4 o;
#pragma unroll
            for (int j = 0; j < 4; ++j) o[j] = scale * (g[b][k][j] - m1 - (xv[j] - mean) * invstd * m2);
            bn2_store(rd, voff, k, o);
            if ((b * K + k) % 3 == 2) __builtin_amdgcn_sched_barrier(0);
        }
    }
    if (t == 0) {
        dgamma[c] = (float)sum_gx;
        dbeta[c] = (float)sum_g;
    }
}

// smallest instantiated K >= float4 per plane and thread
static int bn2_pick_k(int hw4) {
    const int need = cdiv(hw4, BN2_BLOCK);
    for (int k : {1, 2, 3, 4, 8, 9})
        if (k >= need) return k;
    return 0;
}
}  // namespace toda

using namespace toda;

extern "C" int toda_bn2d_supported(int batch, int c, int hw) {
    if (batch < 1 || c < 1 || hw < 4 || (hw & 3)) return 0;
    const int k = bn2_pick_k(hw / 4);
    if (!k) return 0;
    if (batch == 1 || batch == 2) return 1;
    return batch == 4 && k <= 4 ? 1 : 0;
}

#define BN2_DISPATCH(KERNEL, ...)                                                                                          \
    do {                                                                                                                   \
        const int k = bn2_pick_k(hw / 4);                                                                                  \
        const dim3 grid(c), block(BN2_BLOCK);                                                                              \
        hipStream_t s = (hipStream_t)stream;                                                                               \
        bool done = true;                                                                                                  \
        auto go = [&](auto bb, auto kk) {                                                                                  \
            if (relu)                                                                                                      \
                hipLaunchKernelGGL(HIP_KERNEL_NAME(KERNEL<decltype(bb)::value, decltype(kk)::value, true>), grid, block, 0, s, __VA_ARGS__);  \
            else                                                                                                           \
                hipLaunchKernelGGL(HIP_KERNEL_NAME(KERNEL<decltype(bb)::value, decltype(kk)::value, false>), grid, block, 0, s, __VA_ARGS__); \
        };                                                                                                                 \
        auto by_k = [&](auto bb) {                                                                                         \
            switch (k) {                                                                                                   \
                case 1: go(bb, std::integral_constant<int, 1>{}); break;                                                  \
                case 2: go(bb, std::integral_constant<int, 2>{}); break;                                                  \
                case 3: go(bb, std::integral_constant<int, 3>{}); break;                                                  \
                case 4: go(bb, std::integral_constant<int, 4>{}); break;                                                  \
                case 8: if constexpr (decltype(bb)::value <= 2) go(bb, std::integral_constant<int, 8>{}); else done = false; break; \
                case 9: if constexpr (decltype(bb)::value <= 2) go(bb, std::integral_constant<int, 9>{}); else done = false; break; \
                default: done = false;                                                                                     \
            }                                                                                                              \
        };                                                                                                                 \
        switch (batch) {                                                                                                   \
            case 1: by_k(std::integral_constant<int, 1>{}); break;                                                         \
            case 2: by_k(std::integral_constant<int, 2>{}); break;                                                         \
            case 4: by_k(std::integral_constant<int, 4>{}); break;                                                         \
            default: done = false;                                                                                         \
        }                                                                                                                  \
        TODA_CHECK_ARG(done, "bn2d: no instantiation for batch %d, hw %d", batch, hw);                                     \
    } while (0)

extern "C" int toda_bn2d_fwd(const float* x, int batch, int c, int hw, const float* gamma, const float* beta, float* running_mean,
                             float* running_var, float momentum, float eps, int relu, float* y, float* save, void* stream) {
    TODA_CHECK_ARG(x && gamma && beta && y && save, "bn2d_fwd: null argument");
    TODA_CHECK_ARG((running_mean == nullptr) == (running_var == nullptr), "bn2d_fwd: running_mean and running_var go together");
    TODA_CHECK_ARG(toda_bn2d_supported(batch, c, hw), "bn2d_fwd: unsupported shape (batch %d, channels %d, hw %d)", batch, c, hw);
    BN2_DISPATCH(bn2d_fwd_kernel, x, c, hw / 4, gamma, beta, running_mean, running_var, momentum, eps, y, save);
    TODA_LAUNCH_CHECK();
    return TODA_OK;
}

extern "C" int toda_bn2d_bwd(const float* x, const float* dy, int batch, int c, int hw, const float* gamma, const float* beta,
                             const float* save, int relu, float* dx, float* dgamma, float* dbeta, void* stream) {
    TODA_CHECK_ARG(x && dy && gamma && beta && save && dx && dgamma && dbeta, "bn2d_bwd: null argument");
    TODA_CHECK_ARG(toda_bn2d_supported(batch, c, hw), "bn2d_bwd: unsupported shape (batch %d, channels %d, hw %d)", batch, c, hw);
    BN2_DISPATCH(bn2d_bwd_kernel, x, dy, c, hw / 4, gamma, beta, save, dx, dgamma, dbeta);
    TODA_LAUNCH_CHECK();
    return TODA_OK;
}
