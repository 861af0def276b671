// Dense 3x3 / stride 1 / pad 1 fp32 convolution for the BEV neck and the dense heads on gfx950:
// fused Winograd F(4x4, 3x3) on the fp32 matrix cores (v_mfma_f32_16x16x4_f32).
//
//   Y = A^T [ sum_cin (G g G^T) .* (B^T d B) ] A        d: 6x6 input tile, g: 3x3 filter, Y: 4x4 outputs
//
// 36 multiplies per 16 outputs and input channel instead of 144 (4x fewer matrix FLOPs than a direct / implicit-GEMM
// convolution; MIOpen's fp32 path here is the VALU Winograd F(2x2,3x3) at ~98 TFLOP/s direct-equivalent).  The 36
// "frequencies" are 36 independent GEMMs [tiles x Cin] x [Cin x Cout]; one workgroup owns 32 tiles x 32 output channels
// for ALL 36 frequencies, so the output transform of a (tile, channel) pair is pure per-lane register arithmetic on the
// 36 accumulators - nothing of the Winograd domain ever touches HBM:
//
//   per 8-channel chunk:  thread (tile, cin): 6x6 input patch (bounds-checked buffer loads: padding = hardware zero)
//                         -> B^T d B in registers -> 36 ds_write_b32 into the A image [freq][group][lane][2]
//                         transformed filters (pre-arranged in exactly the LDS image order by wino_weight_kernel)
//                         -> global_load_lds 16 B / lane straight into the B image, no registers
//                         barrier; wave (wt, wc): 36 x {ds_read_b64 A, ds_read_b64 B, 2 MFMA}; barrier
//   epilogue:             per lane 4 tiles x 1 channel: A^T m A on the accumulators, bias, 8-byte stores (NCHW)
//
// Interpolation points {0, -1, 1, 1/2, -2, inf} (Barabasz et al.: ~30 % lower fp32 error than {0, +-1, +-2, inf}; measured
// rms error 8.8e-7 against 1.6e-7 for a direct fp32 convolution at Cin = 128).
// dgrad is the same kernel on the 180-degree-rotated, channel-transposed filters (wino_weight_kernel mode 1).
// wgrad contracts over TILES in the Winograd domain (dU[f] = dM[f]^T V[f], dM = A dY A^T, V = B^T d B) and maps back with
// dg = G^T dU G (wino_wgrad_kernel + wino_wgrad_finish_kernel; slabs + fixed-order fold: deterministic, no float atomics).
//
// Replaces torch.nn.Conv2d -> MIOpen for: pcdet/models/backbones_2d/base_bev_backbone.py:37-58,81-112 and
// pcdet/models/dense_heads/center_head.py:20-28,73-80 (reference paths).
#include <hip/hip_ext.h>

#include "common.h"

namespace toda {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

constexpr int WN_BLOCK = 256;     // 4 waves: (wt, wc) = 2 tile groups x 2 channel groups
constexpr int WN_TILES = 32;      // tiles per workgroup
constexpr int WN_COUT = 32;       // output channels per workgroup
constexpr int WN_KC = 8;          // input channels per chunk
constexpr int WN_FREQ = 36;
constexpr int WN_IMG = 256;       // floats per frequency in an LDS image: 2 groups x 64 lanes x 2
constexpr unsigned WN_OOB = 0xFFFFFFF0u;

struct WinoGeom {
    int B, Cin, Cout, H, W;
    int tiles_x, tiles_y, tiles_img, n_tiles;
    int n_tile_blocks, n_cout_blocks, n_chunks;
};

// G for the points {0, -1, 1, 1/2, -2, inf} (6 x 3)
__device__ __constant__ double WN_G[6][3] = {
    {1.0, 0.0, 0.0},
    {-1.0 / 3, 1.0 / 3, -1.0 / 3},
    {1.0 / 3, 1.0 / 3, 1.0 / 3},
    {-16.0 / 15, -8.0 / 15, -4.0 / 15},
    {1.0 / 15, -2.0 / 15, 4.0 / 15},
    {0.0, 0.0, 1.0},
};

// B^T (6 x 6) applied to (d0..d5):
//   [1 -1.5 -2 1.5 1 0; 0 1 -2.5 .5 1 0; 0 -1 .5 2.5 1 0; 0 -2 -1 2 1 0; 0 .5 -1 -.5 1 0; 0 1 -1.5 -2 1.5 1]
__device__ __forceinline__ void wn_bt(float d0, float d1, float d2, float d3, float d4, float d5, float& v0, float& v1,
                                      float& v2, float& v3, float& v4, float& v5) {
    const float a = d3 - d1, b = d4 - d2;
    v0 = __builtin_fmaf(1.5f, a, __builtin_fmaf(-2.0f, d2, d0 + d4));
    v1 = __builtin_fmaf(-2.5f, d2, __builtin_fmaf(0.5f, d3, d1 + d4));
    v2 = __builtin_fmaf(2.5f, d3, __builtin_fmaf(0.5f, d2, d4 - d1));
    v3 = __builtin_fmaf(2.0f, a, b);
    v4 = __builtin_fmaf(-0.5f, a, b);
    v5 = __builtin_fmaf(1.5f, b, __builtin_fmaf(-2.0f, d3, d1 + d5));
}

// A^T (4 x 6) applied to (m0..m5): [1 1 1 1 1 0; 0 -1 1 .5 -2 0; 0 1 1 .25 4 0; 0 -1 1 .125 -8 1]
__device__ __forceinline__ void wn_at(float m0, float m1, float m2, float m3, float m4, float m5, float& y0, float& y1,
                                      float& y2, float& y3) {
    const float s = m1 + m2, d = m2 - m1;
    y0 = (m0 + s) + (m3 + m4);
    y1 = __builtin_fmaf(-2.0f, m4, __builtin_fmaf(0.5f, m3, d));
    y2 = __builtin_fmaf(4.0f, m4, __builtin_fmaf(0.25f, m3, s));
    y3 = __builtin_fmaf(-8.0f, m4, __builtin_fmaf(0.125f, m3, d)) + m5;
}

// A (6 x 4) applied to (y0..y3): the transpose map, used by wgrad on the output gradient
__device__ __forceinline__ void wn_a(float y0, float y1, float y2, float y3, float& m0, float& m1, float& m2, float& m3,
                                     float& m4, float& m5) {
    const float e = y0 + y2, o = y1 + y3;
    m0 = y0;
    m1 = e - o;
    m2 = e + o;
    m3 = __builtin_fmaf(0.125f, y3, __builtin_fmaf(0.25f, y2, __builtin_fmaf(0.5f, y1, y0)));
    m4 = __builtin_fmaf(-8.0f, y3, __builtin_fmaf(4.0f, y2, __builtin_fmaf(-2.0f, y1, y0)));
    m5 = y3;
}

// ------------------------------------------------------------------------------------------------------------------
// U = G g G^T for every (cout, cin), written in the order the forward kernel's B image is read:
//   u[((cb * n_chunks + chunk) * 36 + f) * 256 + wc * 128 + (j * 16 + n) * 2 + s]
//   cout = cb * 32 + wc * 16 + n,  cin = chunk * 8 + 2 j + s
// mode 0: g = w[cout][cin] (forward);  mode 1: g = rot180(w[cin][cout]) with the channel roles swapped (dgrad).
// w is torch's [Cout][Cin][3][3].  CO / CI below are the channel counts of the convolution that will RUN.
// ------------------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(WN_BLOCK)
wino_weight_kernel(const float* __restrict__ w, int cout, int cin, int mode, float* __restrict__ u) {
    const int CO = mode ? cin : cout, CI = mode ? cout : cin;
    const int e = blockIdx.x * WN_BLOCK + threadIdx.x;
    if (e >= CO * CI) return;
    const int co = e / CI, ci = e % CI;
    double g[3][3];
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int b = 0; b < 3; ++b)
            g[a][b] = mode ? (double)w[(((size_t)ci * cin + co) * 3 + (2 - a)) * 3 + (2 - b)]
                           : (double)w[(((size_t)co * cin + ci) * 3 + a) * 3 + b];
    double t[6][3];
#pragma unroll
    for (int i = 0; i < 6; ++i)
#pragma unroll
        for (int b = 0; b < 3; ++b) t[i][b] = WN_G[i][0] * g[0][b] + WN_G[i][1] * g[1][b] + WN_G[i][2] * g[2][b];
    const int cb = co >> 5, wc = (co >> 4) & 1, n = co & 15;
    const int chunk = ci >> 3, j = (ci & 7) >> 1, s = ci & 1;
    const int n_chunks = CI >> 3;
    float* dst = u + ((size_t)(cb * n_chunks + chunk) * WN_FREQ) * WN_IMG + wc * 128 + (j * 16 + n) * 2 + s;
#pragma unroll
    for (int i = 0; i < 6; ++i)
#pragma unroll
        for (int jj = 0; jj < 6; ++jj)
            dst[(size_t)(i * 6 + jj) * WN_IMG] = (float)(t[i][0] * WN_G[jj][0] + t[i][1] * WN_G[jj][1] + t[i][2] * WN_G[jj][2]);
}

__device__ __forceinline__ __amdgpu_buffer_rsrc_t wn_rsrc(const float* base, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(base), 0, bytes, 0x00020000);
}

// XCD-aware work order (speed only): workgroups are dealt round-robin over the 8 XCDs, so hand each XCD one contiguous
// range of work ids; the channel blocks of one tile block are consecutive ids -> they share the input patch in one L2.
__device__ __forceinline__ int wn_work_id() {
    const int nwg = gridDim.x, xcd = blockIdx.x & 7, q8 = nwg >> 3, r8 = nwg & 7;
    return (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (blockIdx.x >> 3);
}

// 6x6 input patch of (tile, channel): rows y0-1 .. y0+4, columns x0-1 .. x0+4; everything outside the image is the
// zero padding and comes back as a hardware zero from an out-of-range buffer offset (no branch around a load).
struct PatchAddr {
    int off0;            // float index of (row y0-1, col x0-1) in this thread's channel plane of chunk 0
    unsigned rowmask;    // bit r: input row y0-1+r inside the image (and the tile exists)
    unsigned colmask;    // bit 0: col x0-1; bit 1: cols x0, x0+1; bit 2: cols x0+2, x0+3; bit 3: col x0+4
};

__device__ __forceinline__ PatchAddr wn_patch_addr(int tile, int chan, int C, const WinoGeom& g) {
    PatchAddr p;
    const bool exists = tile < g.n_tiles;
    const int tl = exists ? tile : 0;
    const int b = tl / g.tiles_img, rem = tl - b * g.tiles_img;
    const int ty = rem / g.tiles_x, tx = rem - ty * g.tiles_x;
    const int y0 = 4 * ty, x0 = 4 * tx;
    p.off0 = ((b * C + chan) * g.H + (y0 - 1)) * g.W + (x0 - 1);
    p.rowmask = 0u;
#pragma unroll
    for (int r = 0; r < 6; ++r) {
        const int y = y0 - 1 + r;
        if (exists && y >= 0 && y < g.H) p.rowmask |= 1u << r;
    }
    p.colmask = (x0 > 0 ? 1u : 0u) | 2u | (x0 + 3 < g.W ? 4u : 0u) | (x0 + 4 < g.W ? 8u : 0u);
    return p;
}

__device__ __forceinline__ void wn_load_patch(__amdgpu_buffer_rsrc_t rsrc, const PatchAddr& p, int W, unsigned soff,
                                              float (&d)[6][6]) {
#pragma unroll
    for (int r = 0; r < 6; ++r) {
        const bool rv = (p.rowmask >> r) & 1u;
        const unsigned o = (unsigned)(p.off0 + r * W) * 4u;
        d[r][0] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc, (rv && (p.colmask & 1u)) ? o : WN_OOB, soff, 0));
        const f32x2 m = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(rsrc, rv ? o + 4u : WN_OOB, soff, 0));
        const f32x2 n = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(rsrc, (rv && (p.colmask & 4u)) ? o + 12u : WN_OOB, soff, 0));
        d[r][5] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc, (rv && (p.colmask & 8u)) ? o + 20u : WN_OOB, soff, 0));
        d[r][1] = m[0];
        d[r][2] = m[1];
        d[r][3] = n[0];
        d[r][4] = n[1];
    }
}

// d (6x6) -> B^T d B, written to the A image of the thread's (group, tile t, channel c)
__device__ __forceinline__ void wn_input_transform_store(const float (&d)[6][6], float* __restrict__ dst) {
    float t[6][6];
#pragma unroll
    for (int x = 0; x < 6; ++x) wn_bt(d[0][x], d[1][x], d[2][x], d[3][x], d[4][x], d[5][x], t[0][x], t[1][x], t[2][x], t[3][x], t[4][x], t[5][x]);
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        float v0, v1, v2, v3, v4, v5;
        wn_bt(t[i][0], t[i][1], t[i][2], t[i][3], t[i][4], t[i][5], v0, v1, v2, v3, v4, v5);
        dst[(i * 6 + 0) * WN_IMG] = v0;
        dst[(i * 6 + 1) * WN_IMG] = v1;
        dst[(i * 6 + 2) * WN_IMG] = v2;
        dst[(i * 6 + 3) * WN_IMG] = v3;
        dst[(i * 6 + 4) * WN_IMG] = v4;
        dst[(i * 6 + 5) * WN_IMG] = v5;
    }
}

// ------------------------------------------------------------------------------------------------------------------
// forward / dgrad.  x [B][Cin][H][W], u from wino_weight_kernel, y [B][Cout][H][W].
// ------------------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(WN_BLOCK, 2)
wino_fwd_kernel(const float* __restrict__ x, const float* __restrict__ u, const float* __restrict__ bias,
                float* __restrict__ y, const WinoGeom g) {
    __shared__ float lds[2 * WN_FREQ * WN_IMG];   // A image (transformed input) | B image (transformed filters)
    float* const ldsA = lds;
    float* const ldsB = lds + WN_FREQ * WN_IMG;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int work = wn_work_id();
    const int tb = work / g.n_cout_blocks, cb = work - tb * g.n_cout_blocks;
    const int tile0 = tb * WN_TILES;

    // ---- transform role: thread = (tile group, tile t, channel c of the chunk)
    const int t_t = tid & 15, t_c = (tid >> 4) & 7, t_grp = tid >> 7;
    const PatchAddr pa = wn_patch_addr(tile0 + t_grp * 16 + t_t, t_c, g.Cin, g);
    const __amdgpu_buffer_rsrc_t xr = wn_rsrc(x, (unsigned)((size_t)g.B * g.Cin * g.H * g.W * 4u));
    float* const a_dst = ldsA + t_grp * 128 + ((t_c >> 1) * 16 + t_t) * 2 + (t_c & 1);
    const unsigned chunk_bytes = (unsigned)(WN_KC * g.H * g.W) * 4u;

    // ---- matrix role: wave = (wt, wc)
    const int wt = wave >> 1, wc = wave & 1;
    const f32x2* const a_frag = reinterpret_cast<const f32x2*>(ldsA) + wt * 64 + lane;
    const f32x2* const b_frag = reinterpret_cast<const f32x2*>(ldsB) + wc * 64 + lane;
    const float* const u_blk = u + (size_t)cb * g.n_chunks * (WN_FREQ * WN_IMG) + tid * 4;

    f32x4 acc[WN_FREQ];
#pragma unroll
    for (int f = 0; f < WN_FREQ; ++f) acc[f] = f32x4{0.f, 0.f, 0.f, 0.f};

    float d[6][6];
    wn_load_patch(xr, pa, g.W, 0u, d);
    for (int chunk = 0; chunk < g.n_chunks; ++chunk) {
        // transformed filters of this chunk: 36 KiB straight into the B image (lane-linear, 16 B per lane)
        const float* usrc = u_blk + (size_t)chunk * (WN_FREQ * WN_IMG);
#pragma unroll
        for (int it = 0; it < (WN_FREQ * WN_IMG) / (WN_BLOCK * 4); ++it)
            __builtin_amdgcn_global_load_lds(usrc + it * (WN_BLOCK * 4), ldsB + it * (WN_BLOCK * 4) + wave * 256, 16, 0, 0);
        wn_input_transform_store(d, a_dst);
        __syncthreads();
        if (chunk + 1 < g.n_chunks) wn_load_patch(xr, pa, g.W, (unsigned)(chunk + 1) * chunk_bytes, d);   // in flight under the MFMAs
#pragma unroll
        for (int f = 0; f < WN_FREQ; f += 2) {
            const f32x2 a0 = a_frag[f * 128], b0 = b_frag[f * 128];
            const f32x2 a1 = a_frag[(f + 1) * 128], b1 = b_frag[(f + 1) * 128];
            acc[f] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[0], b0[0], acc[f], 0, 0, 0);
            acc[f + 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[0], b1[0], acc[f + 1], 0, 0, 0);
            acc[f] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[1], b0[1], acc[f], 0, 0, 0);
            acc[f + 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[1], b1[1], acc[f + 1], 0, 0, 0);
        }
        __syncthreads();
    }

    // ---- epilogue: lane = (channel n = lane & 15, tile quad q = lane >> 4); register r = tile 4 q + r of the wave's 16
    const int co = cb * WN_COUT + wc * 16 + (lane & 15);
    const float bv = bias ? bias[co] : 0.0f;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int tile = tile0 + wt * 16 + (lane >> 4) * 4 + r;
        if (tile >= g.n_tiles) continue;
        const int b = tile / g.tiles_img, rem = tile - b * g.tiles_img;
        const int ty = rem / g.tiles_x, tx = rem - ty * g.tiles_x;
        float tmp[4][6];
#pragma unroll
        for (int j = 0; j < 6; ++j)
            wn_at(acc[0 * 6 + j][r], acc[1 * 6 + j][r], acc[2 * 6 + j][r], acc[3 * 6 + j][r], acc[4 * 6 + j][r], acc[5 * 6 + j][r],
                  tmp[0][j], tmp[1][j], tmp[2][j], tmp[3][j]);
        float* const row0 = y + (((size_t)b * g.Cout + co) * g.H + 4 * ty) * g.W + 4 * tx;
        const bool right = 4 * tx + 3 < g.W;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            float o0, o1, o2, o3;
            wn_at(tmp[i][0], tmp[i][1], tmp[i][2], tmp[i][3], tmp[i][4], tmp[i][5], o0, o1, o2, o3);
            if (4 * ty + i < g.H) {
                float* p = row0 + (size_t)i * g.W;
                *reinterpret_cast<f32x2*>(p) = f32x2{o0 + bv, o1 + bv};
                if (right) *reinterpret_cast<f32x2*>(p + 2) = f32x2{o2 + bv, o3 + bv};
            }
        }
    }
}

static int wino_geom(const char* who, int batch, int cin, int cout, int H, int W, WinoGeom* g) {
    TODA_CHECK_ARG(batch >= 1 && H >= 1 && W >= 2 && W % 2 == 0, "%s: needs batch >= 1, H >= 1 and an even W (got %d x %d x %d)", who, batch, H, W);
    TODA_CHECK_ARG(cin >= WN_KC && cin % WN_KC == 0, "%s: input channels must be a multiple of %d (got %d)", who, WN_KC, cin);
    TODA_CHECK_ARG(cout >= WN_COUT && cout % WN_COUT == 0, "%s: output channels must be a multiple of %d (got %d)", who, WN_COUT, cout);
    const long long in_bytes = 4LL * batch * cin * H * W, out_bytes = 4LL * batch * cout * H * W;
    TODA_CHECK_ARG(in_bytes < (1LL << 32) - 65536 && out_bytes < (1LL << 32) - 65536, "%s: tensor above 4 GiB", who);
    g->B = batch, g->Cin = cin, g->Cout = cout, g->H = H, g->W = W;
    g->tiles_x = (W + 3) / 4, g->tiles_y = (H + 3) / 4;
    g->tiles_img = g->tiles_x * g->tiles_y;
    g->n_tiles = batch * g->tiles_img;
    g->n_tile_blocks = cdiv(g->n_tiles, WN_TILES);
    g->n_cout_blocks = cout / WN_COUT;
    g->n_chunks = cin / WN_KC;
    return TODA_OK;
}

}  // namespace toda

using namespace toda;

extern "C" size_t toda_conv3x3_weight_floats(int cout, int cin) {
    return (size_t)WN_FREQ * cout * cin;
}

extern "C" int toda_conv3x3_supported(int batch, int cin, int cout, int H, int W) {
    if (batch < 1 || H < 1 || W < 2 || (W & 1)) return 0;
    if (cin < 32 || cin % 32 || cout < 32 || cout % 32) return 0;   // both directions (dgrad swaps the roles) and wgrad
    if (4LL * batch * (cin > cout ? cin : cout) * H * W >= (1LL << 32) - 65536) return 0;
    return 1;
}

extern "C" int toda_conv3x3_transform_weight(const float* w, int cout, int cin, int mode, float* u, void* stream) {
    TODA_CHECK_ARG(w && u && (mode == 0 || mode == 1), "conv3x3_transform_weight: null pointer or bad mode");
    const int CO = mode ? cin : cout, CI = mode ? cout : cin;
    TODA_CHECK_ARG(CO % WN_COUT == 0 && CI % WN_KC == 0 && CO > 0 && CI > 0,
                   "conv3x3_transform_weight: produced channels %% 32 and contracted channels %% 8 must be 0 (got %d, %d)", CO, CI);
    hipLaunchKernelGGL(wino_weight_kernel, dim3(cdiv((long long)CO * CI, WN_BLOCK)), dim3(WN_BLOCK), 0, (hipStream_t)stream, w, cout,
                       cin, mode, u);
    TODA_LAUNCH_CHECK();
    return TODA_OK;
}

extern "C" int toda_conv3x3_fwd(const float* x, const float* u, const float* bias, int batch, int cin, int cout, int H, int W,
                                float* y, void* stream) {
    TODA_CHECK_ARG(x && u && y, "conv3x3_fwd: null pointer");
    WinoGeom g;
    int rc = wino_geom("conv3x3_fwd", batch, cin, cout, H, W, &g);
    if (rc) return rc;
    hipLaunchKernelGGL(wino_fwd_kernel, dim3(g.n_tile_blocks * g.n_cout_blocks), dim3(WN_BLOCK), 0, (hipStream_t)stream, x, u,
                       bias, y, g);
    TODA_LAUNCH_CHECK();
    return TODA_OK;
}
