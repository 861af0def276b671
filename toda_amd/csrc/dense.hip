// Sparse -> dense scatter (SparseConvTensor.dense() / PointPillarScatter) and the per-channel
// row statistics / affine+ReLU passes of the BatchNorm1d+ReLU pair, for gfx950.
// All HBM-bandwidth work.  The scatter moves a [64 rows x 32 channels] tile through LDS so that
// feature reads are 128-byte row segments and dense writes run along x (consecutive canonical
// rows are x-neighbours), instead of 4-byte accesses strided by a whole channel plane.
#include "common.h"

namespace toda {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int DN_BLOCK = 256;
constexpr int DN_ROWS = 64;
constexpr int DN_CH = 32;

// spatial offset of row r inside one channel plane; plane(b, c) = (b*C + c) * vol
struct DenseGeom {
    int D, H, W;  // pillar scatter: D = 1, H = ny, W = nx
    int pillar;   // index = z + y*W + x (pointpillar_scatter.py:27)
};

__device__ __forceinline__ long long spatial_offset(const int4 q, const DenseGeom& g) {
    if (g.pillar) return (long long)q.y + (long long)q.z * g.W + q.w;
    return ((long long)q.y * g.H + q.z) * g.W + q.w;
}

template <bool FWD>
__global__ void __launch_bounds__(DN_BLOCK)
dense_tile_kernel(float* __restrict__ feat, const int4* __restrict__ idx, int n, int c, DenseGeom g,
                  float* __restrict__ dense) {
    __shared__ float tile[DN_CH][DN_ROWS + 1];
    __shared__ long long base[DN_ROWS];  // plane-0 offset of each row: b*C*vol + spatial
    const int row0 = blockIdx.x * DN_ROWS, c0 = blockIdx.y * DN_CH;
    const long long vol = (long long)g.D * g.H * g.W;
    if (threadIdx.x < DN_ROWS) {
        const int r = row0 + threadIdx.x;
        long long b = -1;
        if (r < n) {
            const int4 q = idx[r];
            b = (long long)q.x * c * vol + spatial_offset(q, g);
        }
        base[threadIdx.x] = b;
    }
    const int fr = threadIdx.x >> 2, fc = (threadIdx.x & 3) * 8;  // feature side: 64 rows x (4 x 8 channels)
    const int dr = threadIdx.x & 63, dc = threadIdx.x >> 6;       // dense side: row fastest
    if (FWD) {
        if (row0 + fr < n) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int ch = c0 + fc + j;
                tile[fc + j][fr] = ch < c ? feat[(size_t)(row0 + fr) * c + ch] : 0.f;
            }
        }
        __syncthreads();
        const long long b = base[dr];
        if (b >= 0) {
#pragma unroll
            for (int j = 0; j < DN_CH / 4; ++j) {
                const int ch = dc + 4 * j;
                if (c0 + ch < c) dense[b + (long long)(c0 + ch) * vol] = tile[ch][dr];
            }
        }
    } else {
        __syncthreads();
        const long long b = base[dr];
#pragma unroll
        for (int j = 0; j < DN_CH / 4; ++j) {
            const int ch = dc + 4 * j;
            tile[ch][dr] = (b >= 0 && c0 + ch < c) ? dense[b + (long long)(c0 + ch) * vol] : 0.f;
        }
        __syncthreads();
        if (row0 + fr < n) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int ch = c0 + fc + j;
                if (ch < c) feat[(size_t)(row0 + fr) * c + ch] = tile[fc + j][fr];
            }
        }
    }
}

// Per-channel reductions over the rows of [n, c] (c a multiple of 4 that divides 256).  Thread t
// owns 4 consecutive channels (one 16-byte load per row) for a stripe of rows; two independent
// partial sums per quantity give the loads ILP; fp32 inside a thread (<= MOM_ROWS/rows_par terms),
// fp64 across threads and blocks: LDS fold, one plain store of the block's 2c partial sums, then a second
// kernel adds the per-block partials in a fixed order - no atomics (one double atomic per channel and block
// was the bottleneck: 2.1-3.2 TB/s at 512 rows per block and worse with more blocks), bit-reproducible.
#ifndef TODA_MOM_ROWS
#define TODA_MOM_ROWS 256
#endif
constexpr int MOM_ROWS = TODA_MOM_ROWS;      // rows per block while that gives <= MOM_MAX_BLOCKS blocks
constexpr int MOM_MAX_BLOCKS = 2048;

template <class Load>
__device__ __forceinline__ void rows_reduce2(int n, int c, int rows_per_block, Load load, double* __restrict__ sums) {
    __shared__ float sh[2][DN_BLOCK][4];
    const int lanes = c >> 2;                 // threads per row
    const int rows_par = DN_BLOCK / lanes;    // rows per block iteration
    const int cl = threadIdx.x % lanes, rsub = threadIdx.x / lanes;
    const int row_begin = blockIdx.x * rows_per_block;
    const int row_end = min(n, row_begin + rows_per_block);
    f32x4 s0 = f32x4{0.f, 0.f, 0.f, 0.f}, t0 = s0, s1 = s0, t1 = s0;
    int r = row_begin + rsub;
    for (; r + rows_par < row_end; r += 2 * rows_par) {
        f32x4 a, b, a2, b2;
        load(r, cl, a, b);
        load(r + rows_par, cl, a2, b2);
        s0 += a;
        t0 += b;
        s1 += a2;
        t1 += b2;
    }
    if (r < row_end) {
        f32x4 a, b;
        load(r, cl, a, b);
        s0 += a;
        t0 += b;
    }
    s0 += s1;
    t0 += t1;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        sh[0][threadIdx.x][j] = s0[j];
        sh[1][threadIdx.x][j] = t0[j];
    }
    __syncthreads();
    if (threadIdx.x < 2 * c) {  // thread -> (quantity, channel)
        const int q = threadIdx.x / c, ch = threadIdx.x % c;
        double acc = 0.0;
        for (int j = 0; j < rows_par; ++j) acc += sh[q][j * lanes + (ch >> 2)][ch & 3];
        sums[2 * c + (size_t)(q * c + ch) * gridDim.x + blockIdx.x] = acc;   // scratch [2c][blocks] behind the 2c results
    }
}

// sums[col] = sum_g scratch[col][g]: one block per column, strided partial sums then an LDS tree - fixed order
__global__ void __launch_bounds__(DN_BLOCK)
fold_partials_kernel(double* __restrict__ sums, int blocks, int cols) {
    __shared__ double part[DN_BLOCK];
    const double* src = sums + cols + (size_t)blockIdx.x * blocks;
    double acc = 0.0;
    for (int g = threadIdx.x; g < blocks; g += DN_BLOCK) acc += src[g];
    part[threadIdx.x] = acc;
    __syncthreads();
    for (int w = DN_BLOCK / 2; w > 0; w >>= 1) {
        if (threadIdx.x < w) part[threadIdx.x] += part[threadIdx.x + w];
        __syncthreads();
    }
    if (threadIdx.x == 0) sums[blockIdx.x] = part[0];
}

static inline void reduce_plan(int n, int* blocks, int* rows_per_block) {
    const int g = cdiv(n > 0 ? n : 1, MOM_ROWS);
    *blocks = g < MOM_MAX_BLOCKS ? g : MOM_MAX_BLOCKS;
    *rows_per_block = cdiv(n > 0 ? n : 1, *blocks);
}

__global__ void __launch_bounds__(DN_BLOCK)
rows_moments_kernel(const float* __restrict__ x, int n, int c, int rows_per_block, double* __restrict__ sums) {
    const f32x4* x4 = reinterpret_cast<const f32x4*>(x);
    const int lanes = c >> 2;
    rows_reduce2(n, c, rows_per_block, [&](int r, int cl, f32x4& a, f32x4& b) {
        a = x4[(size_t)r * lanes + cl];
        b = a * a;
    }, sums);
}

// Elementwise passes over [n, c] rows: the grid stride (gridDim * 256 float4) is a multiple of c/4,
// so a thread meets the same 4 channels in every iteration and keeps their coefficients in
// registers instead of re-loading per-channel vectors for every element.
constexpr int EW_BLOCKS = 2048;

__global__ void __launch_bounds__(DN_BLOCK)
rows_affine_act_kernel(const f32x4* __restrict__ x, const float* __restrict__ scale, const float* __restrict__ shift,
                       const f32x4* __restrict__ res, long long n4, int c, int relu, f32x4* __restrict__ y) {
    const long long t0 = (long long)blockIdx.x * DN_BLOCK + threadIdx.x;
    const long long stride = (long long)gridDim.x * DN_BLOCK;
    const int ch = (int)((t0 * 4) % c);
    const f32x4 sc = *reinterpret_cast<const f32x4*>(scale + ch), sf = *reinterpret_cast<const f32x4*>(shift + ch);
    for (long long t = t0; t < n4; t += stride) {
        f32x4 v = x[t] * sc + sf;
        if (res) v += res[t];
        if (relu) {
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = fmaxf(v[j], 0.f);
        }
        y[t] = v;
    }
}

// BatchNorm1d(+ReLU) backward over sparse rows, pass 1: per-channel sums of dz and dz*xhat where
// dz = dy * (x*scale+shift > 0) when the block ends in a ReLU (the mask is recomputed from x, the
// forward output is not read), xhat = (x - mean) * invstd.
__global__ void __launch_bounds__(DN_BLOCK)
rows_bn_bwd_reduce_kernel(const float* __restrict__ dy, const float* __restrict__ x, const float* __restrict__ res,
                          const float* __restrict__ stats, int n, int c, int relu, int rows_per_block,
                          double* __restrict__ sums) {
    const f32x4* dy4 = reinterpret_cast<const f32x4*>(dy);
    const f32x4* x4 = reinterpret_cast<const f32x4*>(x);
    const f32x4* r4 = reinterpret_cast<const f32x4*>(res);
    const int lanes = c >> 2;
    const int cl0 = threadIdx.x % lanes;
    const f32x4 mu = reinterpret_cast<const f32x4*>(stats)[cl0];
    const f32x4 is = reinterpret_cast<const f32x4*>(stats + c)[cl0];
    const f32x4 sc = reinterpret_cast<const f32x4*>(stats + 2 * c)[cl0];
    const f32x4 sf = reinterpret_cast<const f32x4*>(stats + 3 * c)[cl0];
    rows_reduce2(n, c, rows_per_block, [&](int r, int cl, f32x4& a, f32x4& b) {
        const f32x4 g = dy4[(size_t)r * lanes + cl], xv = x4[(size_t)r * lanes + cl];
        const f32x4 rv = r4 ? r4[(size_t)r * lanes + cl] : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float pre = xv[j] * sc[j] + sf[j];
            if (r4) pre += rv[j];   // same two roundings as the forward pass (affine, then + residual)
            const float dz = (relu && !(pre > 0.f)) ? 0.f : g[j];
            a[j] = dz;
            b[j] = dz * ((xv[j] - mu[j]) * is[j]);
        }
    }, sums);
}

// pass 2: dx = gamma * invstd * (dz - mean(dz) - xhat * mean(dz * xhat)) = k1 * (dz - m1 - (x - mu) * k2)
__global__ void __launch_bounds__(DN_BLOCK)
rows_bn_bwd_apply_kernel(const f32x4* __restrict__ dy, const f32x4* __restrict__ x, const f32x4* __restrict__ res,
                         const float* __restrict__ stats, const float* __restrict__ gamma, double* __restrict__ sums,
                         long long n4, int c, int n, int relu, f32x4* __restrict__ dx, f32x4* __restrict__ dres,
                         double* __restrict__ colsum_part) {
    // the parameter gradients as float32 for the caller (sum dz = d beta, sum dz * xhat = d gamma): written over the per-block
    // scratch behind the 2c double results, which the fold has finished with
    if (blockIdx.x == 0 && threadIdx.x < 2 * c) reinterpret_cast<float*>(sums + 2 * c)[threadIdx.x] = (float)sums[threadIdx.x];
    const long long t0 = (long long)blockIdx.x * DN_BLOCK + threadIdx.x;
    const long long stride = (long long)gridDim.x * DN_BLOCK;
    const int ch = (int)((t0 * 4) % c);
    const float inv_n = 1.0f / (float)n;
    const f32x4 mu = *reinterpret_cast<const f32x4*>(stats + ch);
    const f32x4 is = *reinterpret_cast<const f32x4*>(stats + c + ch);
    const f32x4 sc = *reinterpret_cast<const f32x4*>(stats + 2 * c + ch);
    const f32x4 sf = *reinterpret_cast<const f32x4*>(stats + 3 * c + ch);
    const f32x4 ga = *reinterpret_cast<const f32x4*>(gamma + ch);
    f32x4 m1, m2;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        m1[j] = (float)(sums[ch + j] * (double)inv_n);
        m2[j] = (float)(sums[c + ch + j] * (double)inv_n);
    }
    const f32x4 k1 = ga * is, k2 = is * m2;  // xhat * m2 = (x - mu) * (is * m2)
    f32x4 csum = {0.f, 0.f, 0.f, 0.f};
    for (long long t = t0; t < n4; t += stride) {
        const f32x4 g = dy[t], xv = x[t];
        const f32x4 rv = res ? res[t] : f32x4{0.f, 0.f, 0.f, 0.f};
        f32x4 out, dzv;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float pre = xv[j] * sc[j] + sf[j];
            if (res) pre += rv[j];
            const float dz = (relu && !(pre > 0.f)) ? 0.f : g[j];
            dzv[j] = dz;
            out[j] = k1[j] * (dz - m1[j] - (xv[j] - mu[j]) * k2[j]);
        }
        dx[t] = out;
        csum += out;
        if (dres) dres[t] = dzv;   // gradient of the shortcut branch
    }
    // Column sums of dx = the bias gradient of the convolution that produced x (SparseBasicBlock's convolutions carry a bias,
    // reference spconv_backbone.py:37-40; in exact arithmetic it is zero - BatchNorm's input gradient sums to zero per channel -
    // and the reference's autograd returns the rounding noise of that sum; so does this): fp32 over a thread's rows, fp64 across
    // the workgroup, per-workgroup partials [c][gridDim.x] folded in fixed order by colsum_fold_kernel.
    if (colsum_part) {
        __shared__ float cs[DN_BLOCK][4];
#pragma unroll
        for (int j = 0; j < 4; ++j) cs[threadIdx.x][j] = csum[j];
        __syncthreads();
        if (threadIdx.x < c) {
            const int lanes = c >> 2, cl = threadIdx.x >> 2, j = threadIdx.x & 3;
            double acc = 0.0;
            for (int t = cl; t < DN_BLOCK; t += lanes) acc += (double)cs[t][j];   // DN_BLOCK % lanes == 0: thread t holds channels 4 (t % lanes) ..
            colsum_part[(size_t)threadIdx.x * gridDim.x + blockIdx.x] = acc;
        }
    }
}

__global__ void __launch_bounds__(DN_BLOCK)
colsum_fold_kernel(const double* __restrict__ part, int blocks, float* __restrict__ out) {
    __shared__ double sh[DN_BLOCK];
    const double* src = part + (size_t)blockIdx.x * blocks;
    double acc = 0.0;
    for (int g = threadIdx.x; g < blocks; g += DN_BLOCK) acc += src[g];
    sh[threadIdx.x] = acc;
    __syncthreads();
    for (int w = DN_BLOCK / 2; w > 0; w >>= 1) {
        if (threadIdx.x < w) sh[threadIdx.x] += sh[threadIdx.x + w];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[blockIdx.x] = (float)sh[0];
}

// one block: batch statistics -> (mean, invstd, scale, shift) + running-stat update, exactly
// nn.BatchNorm1d's training-mode bookkeeping (biased variance for normalisation, unbiased for
// running_var, running = (1-m)*running + m*batch).  training == 0: use the running statistics.
__global__ void __launch_bounds__(DN_BLOCK)
bn_finalize_kernel(const double* __restrict__ sums, int n, int c, const float* __restrict__ gamma,
                   const float* __restrict__ beta, float* __restrict__ running_mean, float* __restrict__ running_var,
                   float momentum, float eps, int training, float* __restrict__ mean_out, float* __restrict__ invstd_out,
                   float* __restrict__ scale_out, float* __restrict__ shift_out) {
    const int ch = threadIdx.x;
    if (ch >= c) return;
    float mean, var;
    if (training) {
        const double m = sums[ch] / (double)n;
        double v = sums[c + ch] / (double)n - m * m;
        if (v < 0.0) v = 0.0;
        mean = (float)m;
        var = (float)v;
        if (running_mean) {
            const double unbiased = n > 1 ? v * (double)n / (double)(n - 1) : v;
            running_mean[ch] = (1.0f - momentum) * running_mean[ch] + momentum * mean;
            running_var[ch] = (1.0f - momentum) * running_var[ch] + momentum * (float)unbiased;
        }
    } else {
        mean = running_mean[ch];
        var = running_var[ch];
    }
    const float invstd = 1.0f / sqrtf(var + eps);
    const float g = gamma ? gamma[ch] : 1.0f, b = beta ? beta[ch] : 0.0f;
    mean_out[ch] = mean;
    invstd_out[ch] = invstd;
    scale_out[ch] = g * invstd;
    shift_out[ch] = b - mean * g * invstd;
}

// fold_partials_kernel + bn_finalize_kernel in one launch (training mode): block ch folds its two columns of per-workgroup
// partials (sum and sum of squares of channel ch) with EXACTLY fold_partials_kernel's order - thread-strided partial sums, then the
// LDS tree - leaves the totals in sums[ch] / sums[c + ch] and finalises the channel.  Saves a launch per BatchNorm1d.
__global__ void __launch_bounds__(DN_BLOCK)
bn_fold_finalize_kernel(double* __restrict__ sums, int blocks, int n, int c, const float* __restrict__ gamma,
                        const float* __restrict__ beta, float* __restrict__ running_mean, float* __restrict__ running_var,
                        float momentum, float eps, float* __restrict__ mean_out, float* __restrict__ invstd_out,
                        float* __restrict__ scale_out, float* __restrict__ shift_out) {
    __shared__ double part[DN_BLOCK];
    const int ch = blockIdx.x;
    double tot[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const double* src = sums + 2 * c + (size_t)(q * c + ch) * blocks;
        double acc = 0.0;
        for (int g = threadIdx.x; g < blocks; g += DN_BLOCK) acc += src[g];
        part[threadIdx.x] = acc;
        __syncthreads();
        for (int w = DN_BLOCK / 2; w > 0; w >>= 1) {
            if (threadIdx.x < w) part[threadIdx.x] += part[threadIdx.x + w];
            __syncthreads();
        }
        tot[q] = part[0];
        __syncthreads();
    }
    if (threadIdx.x != 0) return;
    sums[ch] = tot[0];
    sums[c + ch] = tot[1];
    const double m = tot[0] / (double)n;
    double v = tot[1] / (double)n - m * m;
    if (v < 0.0) v = 0.0;
    const float mean = (float)m, var = (float)v;
    if (running_mean) {
        const double unbiased = n > 1 ? v * (double)n / (double)(n - 1) : v;
        running_mean[ch] = (1.0f - momentum) * running_mean[ch] + momentum * mean;
        running_var[ch] = (1.0f - momentum) * running_var[ch] + momentum * (float)unbiased;
    }
    const float invstd = 1.0f / sqrtf(var + eps);
    const float g = gamma ? gamma[ch] : 1.0f, b = beta ? beta[ch] : 0.0f;
    mean_out[ch] = mean;
    invstd_out[ch] = invstd;
    scale_out[ch] = g * invstd;
    shift_out[ch] = b - mean * g * invstd;
}

// grid for the grid-stride elementwise kernels: <= EW_BLOCKS blocks and (blocks * 256 * 4) % c == 0
static int ew_grid(long long n4, int c) {
    long long blocks = (n4 + DN_BLOCK - 1) / DN_BLOCK;
    if (blocks > EW_BLOCKS) blocks = EW_BLOCKS;
    // 256 * 4 = 1024 floats per block; c divides 1024 for c in {4,...,256 powers of two}; otherwise round up
    while ((blocks * DN_BLOCK * 4) % c != 0) ++blocks;
    return (int)blocks;
}

static int scatter_common(bool fwd, float* feat, const int32_t* idx, int n, int c, int batch, DenseGeom g,
                          float* dense, hipStream_t s) {
    TODA_CHECK_ARG(n >= 0 && c >= 1 && batch >= 1, "sparse<->dense: bad sizes n=%d c=%d batch=%d", n, c, batch);
    const size_t total = (size_t)batch * c * g.D * g.H * g.W;
    if (fwd) TODA_HIP(hipMemsetAsync(dense, 0, total * sizeof(float), s));
    if (n == 0) return TODA_OK;
    const dim3 grid(cdiv(n, DN_ROWS), cdiv(c, DN_CH));
    if (fwd)
        hipLaunchKernelGGL(HIP_KERNEL_NAME(dense_tile_kernel<true>), grid, dim3(DN_BLOCK), 0, s, feat, (const int4*)idx, n,
                           c, g, dense);
    else
        hipLaunchKernelGGL(HIP_KERNEL_NAME(dense_tile_kernel<false>), grid, dim3(DN_BLOCK), 0, s, feat, (const int4*)idx, n,
                           c, g, dense);
    TODA_LAUNCH_CHECK();
    return TODA_OK;
}

}  // namespace toda

using namespace toda;

extern "C" int toda_sparse_to_dense_fwd(const float* feat, const int32_t* idx, int n, int c, int batch,
                                        const int32_t* shape_host, float* dense, void* stream) {
    const DenseGeom g{shape_host[0], shape_host[1], shape_host[2], 0};
    return scatter_common(true, const_cast<float*>(feat), idx, n, c, batch, g, dense, (hipStream_t)stream);
}
extern "C" int toda_sparse_to_dense_bwd(const float* grad_dense, const int32_t* idx, int n, int c, int batch,
                                        const int32_t* shape_host, float* grad_feat, void* stream) {
    const DenseGeom g{shape_host[0], shape_host[1], shape_host[2], 0};
    return scatter_common(false, grad_feat, idx, n, c, batch, g, const_cast<float*>(grad_dense), (hipStream_t)stream);
}
extern "C" int toda_pillar_scatter_fwd(const float* feat, const int32_t* idx, int n, int c, int batch, int ny, int nx,
                                       float* canvas, void* stream) {
    const DenseGeom g{1, ny, nx, 1};
    return scatter_common(true, const_cast<float*>(feat), idx, n, c, batch, g, canvas, (hipStream_t)stream);
}
extern "C" int toda_pillar_scatter_bwd(const float* grad_canvas, const int32_t* idx, int n, int c, int batch, int ny,
                                       int nx, float* grad_feat, void* stream) {
    const DenseGeom g{1, ny, nx, 1};
    return scatter_common(false, grad_feat, idx, n, c, batch, g, const_cast<float*>(grad_canvas), (hipStream_t)stream);
}

extern "C" size_t toda_rows_reduce_doubles(int n, int c) {
    int blocks, rpb;
    reduce_plan(n, &blocks, &rpb);
    return (size_t)2 * c * (1 + (size_t)blocks);
}

extern "C" int toda_rows_moments(const float* x, int n, int c, double* sums, void* stream) {
    hipStream_t s = (hipStream_t)stream;
    TODA_CHECK_ARG(c >= 4 && c % 4 == 0 && c <= DN_BLOCK / 2 && DN_BLOCK % c == 0,
                   "rows_moments: channels must be a multiple of 4 dividing 256, <= 128 (got %d)", c);
    if (n <= 0) {
        TODA_HIP(hipMemsetAsync(sums, 0, 2 * c * sizeof(double), s));
        return TODA_OK;
    }
    int blocks, rpb;
    reduce_plan(n, &blocks, &rpb);
    hipLaunchKernelGGL(rows_moments_kernel, dim3(blocks), dim3(DN_BLOCK), 0, s, x, n, c, rpb, sums);
    hipLaunchKernelGGL(fold_partials_kernel, dim3(2 * c), dim3(DN_BLOCK), 0, s, sums, blocks, 2 * c);
    TODA_LAUNCH_CHECK();
    return TODA_OK;
}

extern "C" int toda_rows_affine_act(const float* x, const float* scale, const float* shift, const float* residual,
                                    int n, int c, int relu, float* y, void* stream) {
    TODA_CHECK_ARG(c >= 4 && c % 4 == 0 && c <= 1024, "rows_affine_act: channels must be a multiple of 4, <= 1024 (got %d)", c);
    if (n <= 0) return TODA_OK;
    const long long n4 = (long long)n * c / 4;
    hipLaunchKernelGGL(rows_affine_act_kernel, dim3(ew_grid(n4, c)), dim3(DN_BLOCK), 0, (hipStream_t)stream,
                       (const f32x4*)x, scale, shift, (const f32x4*)residual, n4, c, relu, (f32x4*)y);
    TODA_LAUNCH_CHECK();
    return TODA_OK;
}

extern "C" size_t toda_rows_bn_bwd_colsum_doubles(int n, int c) {
    if (n <= 0 || c < 4) return 1;
    return (size_t)c * ew_grid((long long)n * c / 4, c);
}

extern "C" int toda_rows_bn_bwd_res_colsum(const float* dy, const float* x, const float* residual, const float* stats,
                                           const float* gamma, int n, int c, int relu, double* sums, float* dx, float* dres,
                                           double* colsum_ws, float* dx_colsum, void* stream) {
    hipStream_t s = (hipStream_t)stream;
    TODA_CHECK_ARG((colsum_ws == nullptr) == (dx_colsum == nullptr), "rows_bn_bwd: colsum workspace and result go together");
    TODA_CHECK_ARG(c >= 4 && c % 4 == 0 && c <= DN_BLOCK / 2 && DN_BLOCK % c == 0,
                   "rows_bn_bwd: channels must be a multiple of 4 dividing 256, <= 128 (got %d)", c);
    if (n <= 0) {
        TODA_HIP(hipMemsetAsync(sums, 0, 3 * c * sizeof(double), s));      // results + their float32 copy
        if (dx_colsum) TODA_HIP(hipMemsetAsync(dx_colsum, 0, c * sizeof(float), s));
        return TODA_OK;
    }
    int blocks, rpb;
    reduce_plan(n, &blocks, &rpb);
    hipLaunchKernelGGL(rows_bn_bwd_reduce_kernel, dim3(blocks), dim3(DN_BLOCK), 0, s, dy, x, residual, stats, n, c, relu, rpb, sums);
    hipLaunchKernelGGL(fold_partials_kernel, dim3(2 * c), dim3(DN_BLOCK), 0, s, sums, blocks, 2 * c);
    const long long n4 = (long long)n * c / 4;
    const int grid = ew_grid(n4, c);
    hipLaunchKernelGGL(rows_bn_bwd_apply_kernel, dim3(grid), dim3(DN_BLOCK), 0, s, (const f32x4*)dy,
                       (const f32x4*)x, (const f32x4*)residual, stats, gamma, sums, n4, c, n, relu, (f32x4*)dx, (f32x4*)dres, colsum_ws);
    if (dx_colsum) hipLaunchKernelGGL(colsum_fold_kernel, dim3(c), dim3(DN_BLOCK), 0, s, (const double*)colsum_ws, grid, dx_colsum);
    TODA_LAUNCH_CHECK();
    return TODA_OK;
}

extern "C" int toda_rows_bn_bwd_res(const float* dy, const float* x, const float* residual, const float* stats,
                                    const float* gamma, int n, int c, int relu, double* sums, float* dx, float* dres,
                                    void* stream) {
    return toda_rows_bn_bwd_res_colsum(dy, x, residual, stats, gamma, n, c, relu, sums, dx, dres, nullptr, nullptr, stream);
}

extern "C" int toda_rows_bn_bwd(const float* dy, const float* x, const float* stats, const float* gamma, int n, int c,
                                int relu, double* sums, float* dx, void* stream) {
    return toda_rows_bn_bwd_res(dy, x, nullptr, stats, gamma, n, c, relu, sums, dx, nullptr, stream);
}

// Training-mode toda_bn_finalize over UNFOLDED partials (toda_spconv_gather_gemm_stats_partials): `blocks` partial sums per column
// behind the 2 c result slots; the totals are left in sums[0:2c] as toda_bn_finalize would have found them.
extern "C" int toda_bn_finalize_partials(double* sums, int blocks, int n, int c, const float* gamma, const float* beta,
                                         float* running_mean, float* running_var, float momentum, float eps, float* mean,
                                         float* invstd, float* scale, float* shift, void* stream) {
    TODA_CHECK_ARG(sums && mean && invstd && scale && shift, "bn_finalize_partials: null argument");
    TODA_CHECK_ARG(c >= 1 && c <= DN_BLOCK && blocks >= 1 && n >= 1, "bn_finalize_partials: channels must be <= 256, blocks and rows >= 1 (got %d, %d, %d)", c,
                   blocks, n);
    TODA_CHECK_ARG((running_mean == nullptr) == (running_var == nullptr), "bn_finalize_partials: running_mean and running_var go together");
    hipLaunchKernelGGL(bn_fold_finalize_kernel, dim3(c), dim3(DN_BLOCK), 0, (hipStream_t)stream, sums, blocks, n, c, gamma, beta, running_mean,
                       running_var, momentum, eps, mean, invstd, scale, shift);
    TODA_LAUNCH_CHECK();
    return TODA_OK;
}

extern "C" int toda_bn_finalize(const double* sums, int n, int c, const float* gamma, const float* beta,
                                float* running_mean, float* running_var, float momentum, float eps, int training,
                                float* mean, float* invstd, float* scale, float* shift, void* stream) {
    TODA_CHECK_ARG(c >= 1 && c <= DN_BLOCK, "bn_finalize: channels must be <= 256 (got %d)", c);
    TODA_CHECK_ARG(training || (running_mean && running_var), "bn_finalize: eval mode needs running statistics");
    hipLaunchKernelGGL(bn_finalize_kernel, dim3(1), dim3(DN_BLOCK), 0, (hipStream_t)stream, sums, n, c, gamma, beta,
                       running_mean, running_var, momentum, eps, training, mean, invstd, scale, shift);
    TODA_LAUNCH_CHECK();
    return TODA_OK;
}
