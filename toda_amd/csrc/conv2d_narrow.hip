// 3x3 / stride 1 / pad 1 fp32 convolutions with a handful of OUTPUT channels: the last layer of every CenterHead branch
// (64 -> 2 center, 1 center_z, 3 dim, 2 rot, n_cls hm; reference pcdet/models/dense_heads/center_head.py:20-28).  With 1..4
// output channels there is no matrix shape to speak of (the Winograd kernels of conv2d.hip want multiples of 32): the work is
// one pass over the 64-channel input (forward, wgrad) or output (dgrad) map, i.e. HBM-bound, and the library path spends 9-10
// launches of ~30 us on the five branches plus their NHWC transposes.  Here ALL branches of a head go through ONE launch per
// direction (blockIdx.z = branch):
//   forward  one wave per image row, a lane owns 4 consecutive pixels: per input channel three 16-byte row loads, the halo
//            columns by wave-wide DPP shifts (lane 0 / the lane past the row end read the zero padding), 36 FMAs per output
//            channel with the 9 filter taps in scalar registers
//   dgrad    same geometry, the (<= 4 channel) output gradient neighbourhood is held in registers and the 64 input channels
//            are produced one after the other
//   wgrad    a block = (input channel, 32-row band): per row the 9 x Cout products of the lane's 4 pixels, summed over the band
//            in registers, wave-reduced once, slabs per band folded in fixed order (deterministic); the bias gradient rides
//            along in the blocks of input channel 0
#include "common.h"

namespace toda {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int NW_MAX_BRANCH = 8;   // branches per launch
constexpr int NW_CO = 4;           // output channels per branch handled by one instantiation
constexpr int NW_BAND = 8;         // rows per wgrad block
constexpr unsigned NW_OOB = 0xFFFFFFF0u;

struct NarrowArgs {
    const float* x[NW_MAX_BRANCH];     // [B][Cin][H][W] input of the branch (forward, wgrad) / nullptr
    const float* w[NW_MAX_BRANCH];     // [cout][Cin][3][3]
    const float* b[NW_MAX_BRANCH];     // [cout] or nullptr
    float* y[NW_MAX_BRANCH];           // forward: [B][cout][H][W]; dgrad: dx [B][Cin][H][W]
    const float* dy[NW_MAX_BRANCH];    // [B][cout][H][W] (dgrad, wgrad)
    float* dw[NW_MAX_BRANCH];          // wgrad: slab base of the branch
    float* db[NW_MAX_BRANCH];          // wgrad: bias-gradient slab base or nullptr
    int cout[NW_MAX_BRANCH];
    int n, B, Cin, H, W;
    int xbs;                           // floats between two images of x (forward, wgrad) / dx (dgrad): Cin * H * W, or more when the
                                       // branch inputs are channel slices of one wider tensor
};

__device__ __forceinline__ __amdgpu_buffer_rsrc_t nw_rsrc(const float* base, size_t bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(base), 0, (unsigned)bytes, 0x00020000);
}
__device__ __forceinline__ float nw_from_prev_lane(float v) {   // lane l <- lane l-1 (lane 0 <- 0)
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x138, 0xF, 0xF, true));
}
__device__ __forceinline__ float nw_from_next_lane(float v) {   // lane l <- lane l+1 (lane 63 <- 0)
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x130, 0xF, 0xF, true));
}

// one image row segment of a lane: pixels x0-1 .. x0+4 (x0 = 4 * lane) of row yy, zeros outside the image
struct Row6 {
    float v[6];
};
__device__ __forceinline__ Row6 nw_load_row(__amdgpu_buffer_rsrc_t r, int plane_off, int yy, int H, int W, int lane) {
    const bool ok = yy >= 0 && yy < H && 4 * lane < W;
    const f32x4 c = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, ok ? (unsigned)(plane_off + yy * W + 4 * lane) * 4u : NW_OOB, 0, 0));
    Row6 o;
    o.v[1] = c[0], o.v[2] = c[1], o.v[3] = c[2], o.v[4] = c[3];
    o.v[0] = nw_from_prev_lane(c[3]);
    o.v[5] = nw_from_next_lane(c[0]);
    return o;
}

// grid (B * H, 1, branches), 256 threads: a block = one image row, a lane = 4 pixels, the four waves split the input channels
// (16 each at Cin = 64; enough waves per CU to cover the load latency of the short channel loop) and meet in LDS.
__global__ void __launch_bounds__(256)
narrow_fwd_kernel(const NarrowArgs a) {
    __shared__ f32x4 part[3][NW_CO][64];
    const int br = blockIdx.z, lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int row = blockIdx.x;
    const int b = row / a.H, y = row - b * a.H;
    const int cout = a.cout[br];
    const float* __restrict__ w = a.w[br];
    const __amdgpu_buffer_rsrc_t xr = nw_rsrc(a.x[br], ((size_t)(a.B - 1) * a.xbs + (size_t)a.Cin * a.H * a.W) * 4u);
    f32x4 acc[NW_CO];
#pragma unroll
    for (int co = 0; co < NW_CO; ++co) acc[co] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int per = (a.Cin + 3) >> 2;
    const int ci_end = min(a.Cin, (wv + 1) * per);
    for (int ci = wv * per; ci < ci_end; ++ci) {
        const int plane = b * a.xbs + ci * a.H * a.W;
        Row6 r[3];
#pragma unroll
        for (int dy = 0; dy < 3; ++dy) r[dy] = nw_load_row(xr, plane, y + dy - 1, a.H, a.W, lane);
#pragma unroll
        for (int co = 0; co < NW_CO; ++co) {
            if (co >= cout) break;      // wave-uniform
            const float* wk = w + ((size_t)co * a.Cin + ci) * 9;
#pragma unroll
            for (int ta = 0; ta < 3; ++ta)
#pragma unroll
                for (int tb = 0; tb < 3; ++tb) {
                    const float wv2 = wk[ta * 3 + tb];
#pragma unroll
                    for (int p = 0; p < 4; ++p) acc[co][p] = __builtin_fmaf(wv2, r[ta].v[p + tb], acc[co][p]);
                }
        }
    }
    if (wv > 0) {
#pragma unroll
        for (int co = 0; co < NW_CO; ++co) part[wv - 1][co][lane] = acc[co];
    }
    __syncthreads();
    if (wv == 0 && 4 * lane < a.W) {
#pragma unroll
        for (int co = 0; co < NW_CO; ++co) {
            if (co >= cout) break;
            const float bv = a.b[br] ? a.b[br][co] : 0.0f;
            const f32x4 v = ((acc[co] + part[0][co][lane]) + part[1][co][lane]) + part[2][co][lane] + f32x4{bv, bv, bv, bv};
            *reinterpret_cast<f32x4*>(a.y[br] + (((size_t)b * cout + co) * a.H + y) * a.W + 4 * lane) = v;
        }
    }
}

// dx[b][ci][y][x] = sum_co sum_ab w[co][ci][a][b] dy[b][co][y - a + 1][x - b + 1]; block = one row, the waves split ci
__global__ void __launch_bounds__(256)
narrow_dgrad_kernel(const NarrowArgs a) {
    const int br = blockIdx.z, lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int row = blockIdx.x;
    const int b = row / a.H, y = row - b * a.H;
    const int cout = a.cout[br];
    const float* __restrict__ w = a.w[br];
    const __amdgpu_buffer_rsrc_t gr = nw_rsrc(a.dy[br], (size_t)a.B * cout * a.H * a.W * 4u);
    Row6 g[NW_CO][3];       // g[co][d] = row y - 1 + d of the output gradient
#pragma unroll
    for (int co = 0; co < NW_CO; ++co)
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            if (co < cout) g[co][d] = nw_load_row(gr, (b * cout + co) * a.H * a.W, y + d - 1, a.H, a.W, lane);
            else
#pragma unroll
                for (int i = 0; i < 6; ++i) g[co][d].v[i] = 0.0f;
        }
    float* const dx = a.y[br];
    const int per = (a.Cin + 3) >> 2;
    const int ci_end = min(a.Cin, (wv + 1) * per);
    for (int ci = wv * per; ci < ci_end; ++ci) {
        f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int co = 0; co < NW_CO; ++co) {
            if (co >= cout) break;
            const float* wk = w + ((size_t)co * a.Cin + ci) * 9;
#pragma unroll
            for (int ta = 0; ta < 3; ++ta)
#pragma unroll
                for (int tb = 0; tb < 3; ++tb) {
                    const float wv2 = wk[ta * 3 + tb];
                    // dy row y - ta + 1 = g[co][2 - ta]; column x - tb + 1 = v[p + 2 - tb]
#pragma unroll
                    for (int p = 0; p < 4; ++p) acc[p] = __builtin_fmaf(wv2, g[co][2 - ta].v[p + 2 - tb], acc[p]);
                }
        }
        if (4 * lane < a.W) *reinterpret_cast<f32x4*>(dx + (size_t)b * a.xbs + ((size_t)ci * a.H + y) * a.W + 4 * lane) = acc;
    }
}

// grid (ceil(Cin / 4), bands = ceil(B * H / NW_BAND), branches): wave = one input channel over the band's rows; partial
// dw[co][ci][3][3] -> slab[band][branch offset + ...]; the wave of ci == 0 also writes the band's bias gradient behind
// the branch's dw
__global__ void __launch_bounds__(256)
narrow_wgrad_kernel(const NarrowArgs a, const int band_stride) {
    const int br = blockIdx.z, band = blockIdx.y;
    const int lane = threadIdx.x & 63, ci = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (ci >= a.Cin) return;
    const int cout = a.cout[br];
    const __amdgpu_buffer_rsrc_t xr = nw_rsrc(a.x[br], ((size_t)(a.B - 1) * a.xbs + (size_t)a.Cin * a.H * a.W) * 4u);
    const __amdgpu_buffer_rsrc_t gr = nw_rsrc(a.dy[br], (size_t)a.B * cout * a.H * a.W * 4u);
    float acc[NW_CO][9], bsum[NW_CO];
#pragma unroll
    for (int co = 0; co < NW_CO; ++co) {
        bsum[co] = 0.0f;
#pragma unroll
        for (int t = 0; t < 9; ++t) acc[co][t] = 0.0f;
    }
    const int rows_total = a.B * a.H;
    const int r_end = min(rows_total, (band + 1) * NW_BAND);
    for (int rr = band * NW_BAND; rr < r_end; ++rr) {
        const int b = rr / a.H, y = rr - b * a.H;
        Row6 r[3];
#pragma unroll
        for (int d = 0; d < 3; ++d) r[d] = nw_load_row(xr, b * a.xbs + ci * a.H * a.W, y + d - 1, a.H, a.W, lane);
#pragma unroll
        for (int co = 0; co < NW_CO; ++co) {
            if (co >= cout) break;
            const bool ok = 4 * lane < a.W;
            const f32x4 gy = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(gr, ok ? (unsigned)(((b * cout + co) * a.H + y) * a.W + 4 * lane) * 4u : NW_OOB, 0, 0));
#pragma unroll
            for (int ta = 0; ta < 3; ++ta)
#pragma unroll
                for (int tb = 0; tb < 3; ++tb)
#pragma unroll
                    for (int p = 0; p < 4; ++p) acc[co][ta * 3 + tb] = __builtin_fmaf(gy[p], r[ta].v[p + tb], acc[co][ta * 3 + tb]);
            bsum[co] += (gy[0] + gy[1]) + (gy[2] + gy[3]);
        }
    }
    // wave reduction (fixed butterfly)
#pragma unroll
    for (int co = 0; co < NW_CO; ++co) {
        if (co >= cout) break;
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            float v = acc[co][t];
#pragma unroll
            for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
            if (lane == 0) a.dw[br][(size_t)band * band_stride + ((size_t)co * a.Cin + ci) * 9 + t] = v;
        }
        if (ci == 0) {
            float v = bsum[co];
#pragma unroll
            for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
            if (lane == 0) a.db[br][(size_t)band * band_stride + co] = v;
        }
    }
}

// out[e] = sum over bands of slab[band][e], in band order
__global__ void __launch_bounds__(256)
narrow_fold_kernel(const float* __restrict__ slab, int bands, int elems, float* __restrict__ out) {
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= elems) return;
    float acc = 0.0f;
    for (int bd = 0; bd < bands; ++bd) acc += slab[(size_t)bd * elems + e];
    out[e] = acc;
}

static int narrow_check(const char* who, int n, int batch, int cin, int H, int W, const int* cout, long long xbs) {
    TODA_CHECK_ARG(xbs == 0 || (xbs >= (long long)cin * H * W && xbs % 4 == 0), "%s: image stride %lld below one image (%d x %d x %d) or not a multiple of 4", who, xbs, cin, H, W);
    TODA_CHECK_ARG(4LL * batch * (xbs ? xbs : (long long)cin * H * W) < (1LL << 32) - 65536, "%s: tensor above 4 GiB", who);
    TODA_CHECK_ARG(n >= 1 && n <= NW_MAX_BRANCH, "%s: 1..%d branches per call (got %d)", who, NW_MAX_BRANCH, n);
    TODA_CHECK_ARG(batch >= 1 && cin >= 1 && H >= 1 && W >= 4 && W % 4 == 0 && W <= 256, "%s: needs W %% 4 == 0 and W <= 256 (got %d x %d)", who, H, W);
    for (int i = 0; i < n; ++i) TODA_CHECK_ARG(cout[i] >= 1 && cout[i] <= NW_CO, "%s: 1..%d output channels per branch (got %d)", who, NW_CO, cout[i]);
    return TODA_OK;
}

}  // namespace toda

using namespace toda;

extern "C" int toda_conv3x3_narrow_supported(int batch, int cin, int cout, int H, int W) {
    return batch >= 1 && cin >= 1 && cout >= 1 && cout <= NW_CO && H >= 1 && W >= 4 && W % 4 == 0 && W <= 256 &&
           4LL * batch * cin * H * W < (1LL << 32) - 65536;
}

extern "C" int toda_conv3x3_narrow_fwd(int n, const float* const* x, const float* const* w, const float* const* bias, const int32_t* cout,
                                       int batch, int cin, int H, int W, long long x_image_stride, float* const* y, void* stream) {
    int rc = narrow_check("conv3x3_narrow_fwd", n, batch, cin, H, W, cout, x_image_stride);
    if (rc) return rc;
    NarrowArgs a = {};
    for (int i = 0; i < n; ++i) {
        TODA_CHECK_ARG(x[i] && w[i] && y[i], "conv3x3_narrow_fwd: null pointer");
        a.x[i] = x[i], a.w[i] = w[i], a.b[i] = bias ? bias[i] : nullptr, a.y[i] = y[i], a.cout[i] = cout[i];
    }
    a.n = n, a.B = batch, a.Cin = cin, a.H = H, a.W = W;
    a.xbs = x_image_stride ? (int)x_image_stride : cin * H * W;
    hipLaunchKernelGGL(narrow_fwd_kernel, dim3(batch * H, 1, n), dim3(256), 0, (hipStream_t)stream, a);
    TODA_LAUNCH_CHECK();
    return TODA_OK;
}

extern "C" int toda_conv3x3_narrow_dgrad(int n, const float* const* dy, const float* const* w, const int32_t* cout, int batch, int cin, int H,
                                         int W, long long dx_image_stride, float* const* dx, void* stream) {
    int rc = narrow_check("conv3x3_narrow_dgrad", n, batch, cin, H, W, cout, dx_image_stride);
    if (rc) return rc;
    NarrowArgs a = {};
    for (int i = 0; i < n; ++i) {
        TODA_CHECK_ARG(dy[i] && w[i] && dx[i], "conv3x3_narrow_dgrad: null pointer");
        a.dy[i] = dy[i], a.w[i] = w[i], a.y[i] = dx[i], a.cout[i] = cout[i];
    }
    a.n = n, a.B = batch, a.Cin = cin, a.H = H, a.W = W;
    a.xbs = dx_image_stride ? (int)dx_image_stride : cin * H * W;
    hipLaunchKernelGGL(narrow_dgrad_kernel, dim3(batch * H, 1, n), dim3(256), 0, (hipStream_t)stream, a);
    TODA_LAUNCH_CHECK();
    return TODA_OK;
}

extern "C" size_t toda_conv3x3_narrow_wgrad_workspace_bytes(int n, int batch, int cin, int H) {
    const size_t bands = (size_t)cdiv((long long)batch * H, NW_BAND);
    return (size_t)n * bands * NW_CO * ((size_t)cin * 9 + 1) * sizeof(float);
}

// out: for branch i in order, dw_i [cout_i][cin][3][3] followed by db_i [cout_i] (sum over i of cout_i * (9 cin + 1) floats)
extern "C" int toda_conv3x3_narrow_wgrad(int n, const float* const* x, const float* const* dy, const int32_t* cout, int batch, int cin, int H,
                                         int W, long long x_image_stride, float* out, void* ws, size_t ws_bytes, void* stream) {
    int rc = narrow_check("conv3x3_narrow_wgrad", n, batch, cin, H, W, cout, x_image_stride);
    if (rc) return rc;
    TODA_CHECK_ARG(out != nullptr, "conv3x3_narrow_wgrad: null output");
    if (!ws || ws_bytes < toda_conv3x3_narrow_wgrad_workspace_bytes(n, batch, cin, H)) {
        toda::set_error("conv3x3_narrow_wgrad: workspace too small");
        return TODA_EWORKSPACE;
    }
    const int bands = cdiv((long long)batch * H, NW_BAND);
    int total = 0;
    for (int i = 0; i < n; ++i) total += cout[i] * (cin * 9 + 1);
    NarrowArgs a = {};
    int off = 0;
    for (int i = 0; i < n; ++i) {      // slab [band][total]: branch i at offset off, its bias gradient behind its dw
        TODA_CHECK_ARG(x[i] && dy[i], "conv3x3_narrow_wgrad: null pointer");
        a.x[i] = x[i], a.dy[i] = dy[i], a.cout[i] = cout[i];
        a.dw[i] = (float*)ws + off;
        a.db[i] = (float*)ws + off + cout[i] * cin * 9;
        off += cout[i] * (cin * 9 + 1);
    }
    a.n = n, a.B = batch, a.Cin = cin, a.H = H, a.W = W;
    a.xbs = x_image_stride ? (int)x_image_stride : cin * H * W;
    a.y[0] = nullptr;
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(narrow_wgrad_kernel, dim3(cdiv(cin, 4), bands, n), dim3(256), 0, s, a, total);
    hipLaunchKernelGGL(narrow_fold_kernel, dim3(cdiv(total, 256)), dim3(256), 0, s, (const float*)ws, bands, total, out);
    TODA_LAUNCH_CHECK();
    return TODA_OK;
}
