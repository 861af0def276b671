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

#include <stdlib.h>

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
template <int V>
__device__ __forceinline__ __amdgpu_buffer_rsrc_t bn2_plane(const float* base, size_t plane, int hwv) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(base + plane * (size_t)hwv * V), 0, hwv * V * 4, 0x00020000);
}
// V = 4: planes whose size is a multiple of 4 floats (16-byte loads); V = 1: any size, dword loads (a plane then starts at any
// 4-byte boundary).  Same code for both: a "vector" of V floats per load.
template <int V> struct bn2_vec { typedef f32x4 type; };
template <> struct bn2_vec<1> { typedef float type; };
__device__ __forceinline__ float bn2_get(const f32x4& v, int j) { return v[j]; }
__device__ __forceinline__ float bn2_get(const float& v, int) { return v; }
__device__ __forceinline__ void bn2_set(f32x4& v, int j, float a) { v[j] = a; }
__device__ __forceinline__ void bn2_set(float& v, int, float a) { v = a; }
template <int V, int AUX>
__device__ __forceinline__ typename bn2_vec<V>::type bn2_load(__amdgpu_buffer_rsrc_t rs, unsigned voff, int k) {
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    if constexpr (V == 4)
        return __builtin_bit_cast(f32x4, (u32x4)__builtin_amdgcn_raw_buffer_load_b128(rs, voff + (unsigned)k * (BN2_BLOCK * 16u), 0, AUX));
    else
        return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, voff + (unsigned)k * (BN2_BLOCK * 4u), 0, AUX));
}
template <int V>
__device__ __forceinline__ void bn2_store(__amdgpu_buffer_rsrc_t rs, unsigned voff, int k, typename bn2_vec<V>::type v) {
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    if constexpr (V == 4)
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), rs, voff + (unsigned)k * (BN2_BLOCK * 16u), 0, 0);
    else
        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), rs, voff + (unsigned)k * (BN2_BLOCK * 4u), 0, 0);
}
constexpr int BN2_NT = 2;      // aux bit 1 (slc / nt): streamed once

template <int V, int BB, int K, bool RELU>
__global__ void __launch_bounds__(BN2_BLOCK)
bn2d_fwd_kernel(const float* __restrict__ x, int C, int hwv, const float* __restrict__ gamma, const float* __restrict__ beta,
                float* __restrict__ running_mean, float* __restrict__ running_var, float momentum, float eps,
                float* __restrict__ y, int yC, float* __restrict__ save) {
    // y may be a channel slice of a wider tensor (the concatenated BEV map): y points at the slice's first plane, yC is the
    // channel count of the tensor it lives in (= C for a tensor of its own)
    __shared__ double sh[BN2_BLOCK / 64];
    const int c = blockIdx.x, t = threadIdx.x;
    const unsigned voff = (unsigned)t * (V * 4u);
    typedef typename bn2_vec<V>::type vec_t;
    vec_t r[BB][K];
#pragma unroll
    for (int b = 0; b < BB; ++b) {
        const __amdgpu_buffer_rsrc_t rs = bn2_plane<V>(x, (size_t)b * C + c, hwv);
#pragma unroll
        for (int k = 0; k < K; ++k) r[b][k] = bn2_load<V, BN2_NT>(rs, voff, k);
    }
    const double n = (double)BB * hwv * V;
    float s = 0.f;
#pragma unroll
    for (int b = 0; b < BB; ++b)
#pragma unroll
        for (int k = 0; k < K; ++k) {
            if constexpr (V == 4) s += (r[b][k][0] + r[b][k][1]) + (r[b][k][2] + r[b][k][3]);
            else s += r[b][k];
        }
    const double mean_d = bn2_block_sum((double)s, sh) / n;
    const float mean = (float)mean_d;
    float q = 0.f;
#pragma unroll
    for (int b = 0; b < BB; ++b)
#pragma unroll
        for (int k = 0; k < K; ++k) {
            const bool ok = t + k * BN2_BLOCK < hwv;
#pragma unroll
            for (int j = 0; j < V; ++j) {
                const float d = ok ? bn2_get(r[b][k], j) - mean : 0.f;
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
        const __amdgpu_buffer_rsrc_t rs = bn2_plane<V>(y, (size_t)b * yC + c, hwv);
#pragma unroll
        for (int k = 0; k < K; ++k) {
            vec_t o;
#pragma unroll
            for (int j = 0; j < V; ++j) {
                const float z = bn2_get(r[b][k], j) * scale + shift;
                bn2_set(o, j, RELU ? (z > 0.f ? z : 0.f) : z);
            }
            bn2_store<V>(rs, voff, k, o);
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

template <int V, int BB, int K, bool RELU>
__global__ void __launch_bounds__(BN2_BLOCK)
bn2d_bwd_kernel(const float* __restrict__ x, const float* __restrict__ dy, int gC, int C, int hwv, const float* __restrict__ gamma,
                const float* __restrict__ beta, const float* __restrict__ save, float* __restrict__ dx,
                float* __restrict__ dgamma, float* __restrict__ dbeta) {
    __shared__ double sh[BN2_BLOCK / 64];
    const int c = blockIdx.x, t = threadIdx.x;
    const unsigned voff = (unsigned)t * (V * 4u);
    const float mean = save[c], invstd = save[C + c];
    const float scale = gamma[c] * invstd;
    const float shift = beta[c] - mean * scale;
    typedef typename bn2_vec<V>::type vec_t;
    vec_t g[BB][K];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int b = 0; b < BB; ++b) {
        const __amdgpu_buffer_rsrc_t rx = bn2_plane<V>(x, (size_t)b * C + c, hwv), rg = bn2_plane<V>(dy, (size_t)b * gC + c, hwv);
#pragma unroll
        for (int k = 0; k < K; ++k) {
            vec_t xv = bn2_load<V, 0>(rx, voff, k);                     // kept in the caches for the second sweep
            vec_t gv = bn2_load<V, BN2_NT>(rg, voff, k);                // past the end of the plane: zeros
#pragma unroll
            for (int j = 0; j < V; ++j) {
                const float xj = bn2_get(xv, j);
                const float gj = (!RELU || xj * scale + shift > 0.f) ? bn2_get(gv, j) : 0.f;      // the forward's expression, bit for bit
                bn2_set(gv, j, gj);
                s1 += gj;
                s2 += gj * ((xj - mean) * invstd);
            }
            g[b][k] = gv;
            // at most three (x, dy) pairs in flight on top of the g image (128 registers at 16 waves per workgroup)
            if ((b * K + k) % 3 == 2) __builtin_amdgcn_sched_barrier(0);
        }
    }
    const double n = (double)BB * hwv * V;
    const double sum_g = bn2_block_sum((double)s1, sh);
    const double sum_gx = bn2_block_sum((double)s2, sh);
    const float m1 = (float)(sum_g / n), m2 = (float)(sum_gx / n);
#pragma unroll
    for (int b = 0; b < BB; ++b) {
        const __amdgpu_buffer_rsrc_t rx = bn2_plane<V>(x, (size_t)b * C + c, hwv), rd = bn2_plane<V>(dx, (size_t)b * C + c, hwv);
#pragma unroll
        for (int k = 0; k < K; ++k) {
            vec_t xv = bn2_load<V, BN2_NT>(rx, voff, k);
            vec_t o;
#pragma unroll
            for (int j = 0; j < V; ++j) bn2_set(o, j, scale * (bn2_get(g[b][k], j) - m1 - (bn2_get(xv, j) - mean) * invstd * m2));
            bn2_store<V>(rd, voff, k, o);
            if ((b * K + k) % 3 == 2) __builtin_amdgcn_sched_barrier(0);
        }
    }
    if (t == 0) {
        dgamma[c] = (float)sum_gx;
        dbeta[c] = (float)sum_g;
    }
}

// ---- one workgroup per (channel, sample) plane, partners exchange their partial sums ------------------------------------
// With one workgroup per channel a 128-channel layer occupies half of the 256 CUs and a thread carries batch x K vectors (the
// backward kernel spills at 2 x 188 x 188).  Here the P = batch workgroups of a channel each hold ONE plane, publish two fp64
// partial results to a small workspace (agent-scope atomics: partners may sit on different XCDs whose L2s are not coherent
// for plain accesses) and wait for each other on per-workgroup epoch flags - `epoch` is a number the caller never repeats on a
// workspace, so nothing has to be reset between launches and launches with different channel counts can share it.  Partners
// get neighbouring workgroup ids (same XCD when the channel count is a multiple of 8): dispatch is in id order, so a waiting
// workgroup's partners are resident or next in line - no deadlock however full the GPU is.  Every partner adds the P
// partials in the same order: identical statistics in all of them, run-to-run deterministic.
constexpr int BN2_MAX_SYNC_C = 4096, BN2_MAX_P = 4;
struct Bn2Sync {
    unsigned long long val[BN2_MAX_SYNC_C * BN2_MAX_P][2];
    unsigned flag[BN2_MAX_SYNC_C * BN2_MAX_P];
};

__device__ __forceinline__ void bn2_plane_of_block(int C, int P, int* c, int* b) {
    const int id = blockIdx.x;
    if ((C & 7) == 0) {
        const int slot = id >> 3;
        *b = slot % P;
        *c = (slot / P) * 8 + (id & 7);
    } else {
        *c = id / P;
        *b = id - *c * P;
    }
}

// thread 0 of the block publishes (v0, v1) for (c, b), waits for the P - 1 partners and returns all P pairs through `sh` (LDS, 2 P doubles)
__device__ __forceinline__ void bn2_exchange(Bn2Sync* sy, int c, int b, int P, unsigned epoch, double v0, double v1, double* sh, unsigned* fault) {
    if (threadIdx.x == 0) {
        const int me = c * P + b;
        // relaxed agent-scope atomics only (they go through to memory past the XCD-private caches): a release / acquire pair
        // here means writing back and invalidating the whole L2 around the exchange - measured 8-10 us per kernel
        __hip_atomic_store(&sy->val[me][0], (unsigned long long)__double_as_longlong(v0), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(&sy->val[me][1], (unsigned long long)__double_as_longlong(v1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // the two values have left before the flag goes up
        __hip_atomic_store(&sy->flag[me], epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        for (int p = 0; p < P; ++p) {
            const int o = c * P + p;
            if (p != b) {
                // bounded: a partner that never becomes resident (fewer CUs than the ordering argument above assumes) must not
                // hang the GPU - after ~2 s the fault word goes up and the launch finishes with invalid statistics
                unsigned polls = 0;
                while (__hip_atomic_load(&sy->flag[o], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != epoch) {
                    if (++polls > FAULT_SPIN_LIMIT) {
                        fault_raise(fault, TODA_FAULT_BN2D);
                        break;
                    }
                    __builtin_amdgcn_s_sleep(2);
                }
            }
            asm volatile("" ::: "memory");
            sh[2 * p] = __longlong_as_double((long long)__hip_atomic_load(&sy->val[o][0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
            sh[2 * p + 1] = __longlong_as_double((long long)__hip_atomic_load(&sy->val[o][1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
        }
    }
    __syncthreads();
}

template <int V, int K, bool RELU>
__global__ void __launch_bounds__(BN2_BLOCK)
bn2d_fwd_split_kernel(const float* __restrict__ x, int C, int P, int hwv, const float* __restrict__ gamma, const float* __restrict__ beta,
                      float* __restrict__ running_mean, float* __restrict__ running_var, float momentum, float eps,
                      float* __restrict__ y, int yC, float* __restrict__ save, Bn2Sync* __restrict__ sy, unsigned epoch, unsigned* __restrict__ fault) {
    __shared__ double sh[BN2_BLOCK / 64];
    __shared__ double part[2 * BN2_MAX_P];
    int c, b;
    bn2_plane_of_block(C, P, &c, &b);
    const int t = threadIdx.x;
    const unsigned voff = (unsigned)t * (V * 4u);
    typedef typename bn2_vec<V>::type vec_t;
    vec_t r[K];
    const __amdgpu_buffer_rsrc_t rs = bn2_plane<V>(x, (size_t)b * C + c, hwv);
#pragma unroll
    for (int k = 0; k < K; ++k) r[k] = bn2_load<V, BN2_NT>(rs, voff, k);
    const double n_b = (double)hwv * V;
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < K; ++k) {
        if constexpr (V == 4) s += (r[k][0] + r[k][1]) + (r[k][2] + r[k][3]);
        else s += r[k];
    }
    const double mean_b_d = bn2_block_sum((double)s, sh) / n_b;
    const float mean_b = (float)mean_b_d;
    float q = 0.f;
#pragma unroll
    for (int k = 0; k < K; ++k) {
        const bool ok = t + k * BN2_BLOCK < hwv;
#pragma unroll
        for (int j = 0; j < V; ++j) {
            const float d = ok ? bn2_get(r[k], j) - mean_b : 0.f;
            q += d * d;
        }
    }
    const double dmb = mean_b_d - (double)mean_b;
    const double m2_b = bn2_block_sum((double)q, sh) - n_b * dmb * dmb;        // sum (x - mean_b)^2 of this plane
    // planes -> channel (Chan et al.): mean = sum n_b mean_b / n, M2 = sum M2_b + sum n_b (mean_b - mean)^2; all n_b are equal
    bn2_exchange(sy, c, b, P, epoch, mean_b_d, m2_b, part, fault);
    double mean_d = 0.0;
    for (int p = 0; p < P; ++p) mean_d += part[2 * p];
    mean_d /= P;
    double m2 = 0.0;
    for (int p = 0; p < P; ++p) m2 += part[2 * p + 1] + n_b * (part[2 * p] - mean_d) * (part[2 * p] - mean_d);
    const double n = n_b * P;
    const double var_d = m2 > 0.0 ? m2 / n : 0.0;
    const float mean = (float)mean_d;
    const float invstd = 1.0f / sqrtf((float)var_d + eps);
    const float scale = gamma[c] * invstd;
    const float shift = beta[c] - mean * scale;
    const __amdgpu_buffer_rsrc_t ry = bn2_plane<V>(y, (size_t)b * yC + c, hwv);
#pragma unroll
    for (int k = 0; k < K; ++k) {
        vec_t o;
#pragma unroll
        for (int j = 0; j < V; ++j) {
            const float z = bn2_get(r[k], j) * scale + shift;
            bn2_set(o, j, RELU ? (z > 0.f ? z : 0.f) : z);
        }
        bn2_store<V>(ry, voff, k, o);
    }
    if (t == 0 && b == 0) {
        save[c] = mean;
        save[C + c] = invstd;
        if (running_mean) {
            const double unbiased = n > 1.0 ? var_d * n / (n - 1.0) : 0.0;
            running_mean[c] = (1.0f - momentum) * running_mean[c] + momentum * mean;
            running_var[c] = (1.0f - momentum) * running_var[c] + momentum * (float)unbiased;
        }
    }
}

template <int V, int K, bool RELU>
__global__ void __launch_bounds__(BN2_BLOCK)
bn2d_bwd_split_kernel(const float* __restrict__ x, const float* __restrict__ dy, int gC, int C, int P, int hwv, const float* __restrict__ gamma,
                      const float* __restrict__ beta, const float* __restrict__ save, float* __restrict__ dx,
                      float* __restrict__ dgamma, float* __restrict__ dbeta, Bn2Sync* __restrict__ sy, unsigned epoch, unsigned* __restrict__ fault) {
    __shared__ double sh[BN2_BLOCK / 64];
    __shared__ double part[2 * BN2_MAX_P];
    int c, b;
    bn2_plane_of_block(C, P, &c, &b);
    const int t = threadIdx.x;
    const unsigned voff = (unsigned)t * (V * 4u);
    const float mean = save[c], invstd = save[C + c];
    const float scale = gamma[c] * invstd;
    const float shift = beta[c] - mean * scale;
    typedef typename bn2_vec<V>::type vec_t;
    vec_t g[K], xh[K];         // masked dy and the normalised x of this plane: nothing is read twice
    float s1 = 0.f, s2 = 0.f;
    const __amdgpu_buffer_rsrc_t rx = bn2_plane<V>(x, (size_t)b * C + c, hwv), rg = bn2_plane<V>(dy, (size_t)b * gC + c, hwv);
#pragma unroll
    for (int k = 0; k < K; ++k) {
        vec_t xv = bn2_load<V, BN2_NT>(rx, voff, k);
        vec_t gv = bn2_load<V, BN2_NT>(rg, voff, k);                // past the end of the plane: zeros
#pragma unroll
        for (int j = 0; j < V; ++j) {
            const float xj = bn2_get(xv, j);
            const float gj = (!RELU || xj * scale + shift > 0.f) ? bn2_get(gv, j) : 0.f;      // the forward's expression, bit for bit
            const float hj = (xj - mean) * invstd;
            bn2_set(gv, j, gj);
            bn2_set(xv, j, hj);
            s1 += gj;
            s2 += gj * hj;
        }
        g[k] = gv;
        xh[k] = xv;
    }
    const double p1 = bn2_block_sum((double)s1, sh);
    const double p2 = bn2_block_sum((double)s2, sh);
    bn2_exchange(sy, c, b, P, epoch, p1, p2, part, fault);
    double sum_g = 0.0, sum_gx = 0.0;
    for (int p = 0; p < P; ++p) sum_g += part[2 * p], sum_gx += part[2 * p + 1];
    const double n = (double)hwv * V * P;
    const float m1 = (float)(sum_g / n), m2 = (float)(sum_gx / n);
    const __amdgpu_buffer_rsrc_t rd = bn2_plane<V>(dx, (size_t)b * C + c, hwv);
#pragma unroll
    for (int k = 0; k < K; ++k) {
        vec_t o;
#pragma unroll
        for (int j = 0; j < V; ++j) bn2_set(o, j, scale * (bn2_get(g[k], j) - m1 - bn2_get(xh[k], j) * m2));
        bn2_store<V>(rd, voff, k, o);
    }
    if (t == 0 && b == 0) {
        dgamma[c] = (float)sum_gx;
        dbeta[c] = (float)sum_g;
    }
}

// smallest instantiated K >= vectors per plane and thread (0: none).  V = 4: K in {1, 2, 3, 4, 8, 9}; V = 1: {4, 9, 18, 36}
static int bn2_pick_k(int hwv, int v) {
    const int need = cdiv(hwv, BN2_BLOCK);
    if (v == 4) {
        for (int k : {1, 2, 3, 4, 8, 9})
            if (k >= need) return k;
    } else {
        for (int k : {4, 9, 18, 36})
            if (k >= need) return k;
    }
    return 0;
}
static int bn2_floats_per_thread(int batch, int hw) {      // one workgroup per channel; 0: unsupported
    if (batch < 1 || hw < 1) return 0;
    const int v = (hw & 3) ? 1 : 4;
    const int k = bn2_pick_k(hw / v, v);
    if (!k || !(batch == 1 || batch == 2 || batch == 4)) return 0;
    return batch * k * v <= 72 ? batch * k * v : 0;            // the register image: 72 floats per thread
}
static bool bn2_split_ok(int batch, int c, int hw) {          // one workgroup per plane: 36 floats of x (and of dy) per thread
    if (!(batch == 2 || batch == 4) || c < 1 || c > BN2_MAX_SYNC_C || hw < 1) return false;
    const int v = (hw & 3) ? 1 : 4;
    const int k = bn2_pick_k(hw / v, v);
    return k > 0 && k * v <= 36;
}
}  // namespace toda

using namespace toda;

extern "C" size_t toda_bn2d_sync_bytes(void) { return sizeof(Bn2Sync); }

extern "C" int toda_bn2d_supported(int batch, int c, int hw) {
    return c >= 1 && (bn2_floats_per_thread(batch, hw) > 0 || bn2_split_ok(batch, c, hw)) ? 1 : 0;
}

template <int V, int BB, int K>
static void bn2_launch_fwd(bool relu, dim3 grid, hipStream_t s, const float* x, int c, int hwv, const float* gamma, const float* beta, float* rm,
                           float* rv, float momentum, float eps, float* y, int yC, float* save) {
    if constexpr (BB * K * V <= 72) {
        if (relu) hipLaunchKernelGGL(HIP_KERNEL_NAME(bn2d_fwd_kernel<V, BB, K, true>), grid, dim3(BN2_BLOCK), 0, s, x, c, hwv, gamma, beta, rm, rv, momentum, eps, y, yC, save);
        else hipLaunchKernelGGL(HIP_KERNEL_NAME(bn2d_fwd_kernel<V, BB, K, false>), grid, dim3(BN2_BLOCK), 0, s, x, c, hwv, gamma, beta, rm, rv, momentum, eps, y, yC, save);
    }
}
template <int V, int BB, int K>
static void bn2_launch_bwd(bool relu, dim3 grid, hipStream_t s, const float* x, const float* dy, int gC, int c, int hwv, const float* gamma,
                           const float* beta, const float* save, float* dx, float* dgamma, float* dbeta) {
    if constexpr (BB * K * V <= 72) {
        if (relu) hipLaunchKernelGGL(HIP_KERNEL_NAME(bn2d_bwd_kernel<V, BB, K, true>), grid, dim3(BN2_BLOCK), 0, s, x, dy, gC, c, hwv, gamma, beta, save, dx, dgamma, dbeta);
        else hipLaunchKernelGGL(HIP_KERNEL_NAME(bn2d_bwd_kernel<V, BB, K, false>), grid, dim3(BN2_BLOCK), 0, s, x, dy, gC, c, hwv, gamma, beta, save, dx, dgamma, dbeta);
    }
}
// split kernels: BB is the dummy 0 so that the same BN2_BY_K switch serves them; the trailing arguments carry P, sync, epoch
template <int V, int BB, int K>
static void bn2_launch_fwd_split(bool relu, dim3 grid, hipStream_t s, const float* x, int c, int P, int hwv, const float* gamma, const float* beta,
                                 float* rm, float* rv, float momentum, float eps, float* y, int yC, float* save, Bn2Sync* sy, unsigned epoch, unsigned* fault) {
    if constexpr (K * V <= 36) {
        if (relu) hipLaunchKernelGGL(HIP_KERNEL_NAME(bn2d_fwd_split_kernel<V, K, true>), grid, dim3(BN2_BLOCK), 0, s, x, c, P, hwv, gamma, beta, rm, rv, momentum, eps, y, yC, save, sy, epoch, fault);
        else hipLaunchKernelGGL(HIP_KERNEL_NAME(bn2d_fwd_split_kernel<V, K, false>), grid, dim3(BN2_BLOCK), 0, s, x, c, P, hwv, gamma, beta, rm, rv, momentum, eps, y, yC, save, sy, epoch, fault);
    }
}
template <int V, int BB, int K>
static void bn2_launch_bwd_split(bool relu, dim3 grid, hipStream_t s, const float* x, const float* dy, int gC, int c, int P, int hwv, const float* gamma,
                                 const float* beta, const float* save, float* dx, float* dgamma, float* dbeta, Bn2Sync* sy, unsigned epoch, unsigned* fault) {
    if constexpr (K * V <= 36) {
        if (relu) hipLaunchKernelGGL(HIP_KERNEL_NAME(bn2d_bwd_split_kernel<V, K, true>), grid, dim3(BN2_BLOCK), 0, s, x, dy, gC, c, P, hwv, gamma, beta, save, dx, dgamma, dbeta, sy, epoch, fault);
        else hipLaunchKernelGGL(HIP_KERNEL_NAME(bn2d_bwd_split_kernel<V, K, false>), grid, dim3(BN2_BLOCK), 0, s, x, dy, gC, c, P, hwv, gamma, beta, save, dx, dgamma, dbeta, sy, epoch, fault);
    }
}

// (V, batch, K) -> instantiation; LAUNCH is one of the four launchers above
#define BN2_BY_K(LAUNCH, VV, BBV, ...)                                      \
    switch (k) {                                                            \
        case 1: LAUNCH<VV, BBV, 1>(__VA_ARGS__); break;                     \
        case 2: LAUNCH<VV, BBV, 2>(__VA_ARGS__); break;                     \
        case 3: LAUNCH<VV, BBV, 3>(__VA_ARGS__); break;                     \
        case 4: LAUNCH<VV, BBV, 4>(__VA_ARGS__); break;                     \
        case 8: LAUNCH<VV, BBV, 8>(__VA_ARGS__); break;                     \
        case 9: LAUNCH<VV, BBV, 9>(__VA_ARGS__); break;                     \
        case 18: LAUNCH<VV, BBV, 18>(__VA_ARGS__); break;                   \
        default: LAUNCH<VV, BBV, 36>(__VA_ARGS__); break;                   \
    }
#define BN2_DISPATCH(LAUNCH, ...)                                           \
    do {                                                                    \
        const dim3 grid(c);                                                 \
        if (v == 4) {                                                       \
            if (batch == 1) { BN2_BY_K(LAUNCH, 4, 1, relu != 0, grid, s, __VA_ARGS__) }       \
            else if (batch == 2) { BN2_BY_K(LAUNCH, 4, 2, relu != 0, grid, s, __VA_ARGS__) }  \
            else { BN2_BY_K(LAUNCH, 4, 4, relu != 0, grid, s, __VA_ARGS__) }                  \
        } else {                                                            \
            if (batch == 1) { BN2_BY_K(LAUNCH, 1, 1, relu != 0, grid, s, __VA_ARGS__) }       \
            else if (batch == 2) { BN2_BY_K(LAUNCH, 1, 2, relu != 0, grid, s, __VA_ARGS__) }  \
            else { BN2_BY_K(LAUNCH, 1, 4, relu != 0, grid, s, __VA_ARGS__) }                  \
        }                                                                   \
    } while (0)
#define BN2_DISPATCH_SPLIT(LAUNCH, ...)                                     \
    do {                                                                    \
        const dim3 grid(c * batch);                                         \
        if (v == 4) { BN2_BY_K(LAUNCH, 4, 0, relu != 0, grid, s, __VA_ARGS__) }               \
        else { BN2_BY_K(LAUNCH, 1, 0, relu != 0, grid, s, __VA_ARGS__) }                      \
    } while (0)

// a fault raised by an earlier launch (bounded spin gave up) is reported by the next call; the message names the kernels
static int bn2_fault_poll(const char* who) {
    const unsigned v = fault_take();
    if (!v) return TODA_OK;
    toda::set_error("%s: device fault word 0x%x raised by an earlier launch (bounded inter-workgroup wait gave up: %s%s) - its results are invalid",
                    who, v, (v & TODA_FAULT_BN2D) ? "bn2d split kernel " : "", (v & TODA_FAULT_WINO) ? "wino_fwd_ws_kernel" : "");
    return TODA_EFAULT;
}

// y = channels [y_channel0, y_channel0 + c) of a [batch][y_channels][hw] tensor (the deblocks of the BEV neck write straight into the
// concatenated map: reference base_bev_backbone.py:104-107 torch.cat(ups, dim=1))
extern "C" int toda_bn2d_fwd_into(const float* x, int batch, int c, int hw, const float* gamma, const float* beta, float* running_mean,
                                  float* running_var, float momentum, float eps, int relu, float* y, int y_channels, int y_channel0,
                                  float* save, void* sync, unsigned epoch, void* stream) {
    TODA_CHECK_ARG(x && gamma && beta && y && save, "bn2d_fwd: null argument");
    TODA_CHECK_ARG(y_channel0 >= 0 && y_channel0 + c <= y_channels, "bn2d_fwd: channels [%d, %d) outside the %d channels of y", y_channel0,
                   y_channel0 + c, y_channels);
    float* const ys = y + (size_t)y_channel0 * hw;
    const int yC = y_channels;
    TODA_CHECK_ARG((running_mean == nullptr) == (running_var == nullptr), "bn2d_fwd: running_mean and running_var go together");
    // forward: the exchange costs 2-3 us, the single workgroup per channel only half-fills the GPU on <= 128 channels - measured
    // 19.1 vs 21.3 us at 2 x 128 x 188 x 188, 9.9 vs 12.6 at 2 x 256 x 94 x 94: one workgroup per channel wherever it fits
    if (int rc = bn2_fault_poll("bn2d_fwd")) return rc;
    const bool split = sync != nullptr && epoch != 0 && bn2_split_ok(batch, c, hw) && bn2_floats_per_thread(batch, hw) == 0;
    TODA_CHECK_ARG(split || bn2_floats_per_thread(batch, hw) > 0, "bn2d_fwd: unsupported shape (batch %d, channels %d, hw %d)%s", batch, c, hw,
                   sync ? "" : " without a sync workspace");
    const int v = (hw & 3) ? 1 : 4, hwv = hw / v;
    const int k = bn2_pick_k(hwv, v);
    hipStream_t s = (hipStream_t)stream;
    if (split) BN2_DISPATCH_SPLIT(bn2_launch_fwd_split, x, c, batch, hwv, gamma, beta, running_mean, running_var, momentum, eps, ys, yC, save, (Bn2Sync*)sync, epoch, fault_word_dev());
    else BN2_DISPATCH(bn2_launch_fwd, x, c, hwv, gamma, beta, running_mean, running_var, momentum, eps, ys, yC, save);
    TODA_LAUNCH_CHECK();
    return TODA_OK;
}

extern "C" int toda_bn2d_fwd(const float* x, int batch, int c, int hw, const float* gamma, const float* beta, float* running_mean,
                             float* running_var, float momentum, float eps, int relu, float* y, float* save, void* sync, unsigned epoch,
                             void* stream) {
    return toda_bn2d_fwd_into(x, batch, c, hw, gamma, beta, running_mean, running_var, momentum, eps, relu, y, c, 0, save, sync, epoch, stream);
}

// dy = channels [dy_channel0, dy_channel0 + c) of a [batch][dy_channels][hw] gradient (a slice of the concatenated map's gradient)
extern "C" int toda_bn2d_bwd_from(const float* x, const float* dy, int dy_channels, int dy_channel0, int batch, int c, int hw, const float* gamma,
                                  const float* beta, const float* save, int relu, float* dx, float* dgamma, float* dbeta, void* sync,
                                  unsigned epoch, void* stream) {
    TODA_CHECK_ARG(x && dy && gamma && beta && save && dx && dgamma && dbeta, "bn2d_bwd: null argument");
    TODA_CHECK_ARG(dy_channel0 >= 0 && dy_channel0 + c <= dy_channels, "bn2d_bwd: channels [%d, %d) outside the %d channels of dy", dy_channel0,
                   dy_channel0 + c, dy_channels);
    const float* const gs = dy + (size_t)dy_channel0 * hw;
    const int gC = dy_channels;
    // backward: per plane when a channel's dy no longer leaves room for anything else in the registers (the channel kernel then
    // spills and reads x twice: 51.6 vs 28.5 us at 2 x 128 x 188 x 188; at 2 x 256 x 94 x 94 it is 12.8 vs 17.7 the other way)
    const int per_channel = bn2_floats_per_thread(batch, hw);
    if (int rc = bn2_fault_poll("bn2d_bwd")) return rc;
    // TODA_BN2D_SPLIT=0: never the partner-exchange kernel where the per-channel one exists (it spills there, it cannot wait)
    static const int env_split = getenv("TODA_BN2D_SPLIT") ? atoi(getenv("TODA_BN2D_SPLIT")) : 1;
    const bool split = sync != nullptr && epoch != 0 && bn2_split_ok(batch, c, hw) && (per_channel == 0 || (per_channel > 36 && env_split));
    TODA_CHECK_ARG(split || per_channel > 0, "bn2d_bwd: unsupported shape (batch %d, channels %d, hw %d)%s", batch, c, hw,
                   sync ? "" : " without a sync workspace");
    const int v = (hw & 3) ? 1 : 4, hwv = hw / v;
    const int k = bn2_pick_k(hwv, v);
    hipStream_t s = (hipStream_t)stream;
    if (split) BN2_DISPATCH_SPLIT(bn2_launch_bwd_split, x, gs, gC, c, batch, hwv, gamma, beta, save, dx, dgamma, dbeta, (Bn2Sync*)sync, epoch, fault_word_dev());
    else BN2_DISPATCH(bn2_launch_bwd, x, gs, gC, c, hwv, gamma, beta, save, dx, dgamma, dbeta);
    TODA_LAUNCH_CHECK();
    return TODA_OK;
}

extern "C" int toda_bn2d_bwd(const float* x, const float* dy, int batch, int c, int hw, const float* gamma, const float* beta,
                             const float* save, int relu, float* dx, float* dgamma, float* dbeta, void* sync, unsigned epoch, void* stream) {
    return toda_bn2d_bwd_from(x, dy, c, 0, batch, c, hw, gamma, beta, save, relu, dx, dgamma, dbeta, sync, epoch, stream);
}
