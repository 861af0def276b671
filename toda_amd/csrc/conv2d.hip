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
#include <stdlib.h>

#include "common.h"

#ifndef TODA_ABLATE
#define TODA_ABLATE 0      // -DTODA_ABLATE=1: measurement builds only (WRONG numbers) - the TODA_WINO_ABLATE / TODA_WINO_WG_ABLATE knobs switch parts of
#endif                     // the Winograd kernels off; in the library that ships the ablation branches fold away and the knobs are not read

namespace toda {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4w __attribute__((ext_vector_type(4)));

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
    // one thread per (cb, chunk, frequency row fi, position in the 256-float image row): the six frequencies (fi, 0..5) are
    // six stores with the whole wave on consecutive floats; mode 2 writes the forward operand followed by the data-gradient
    // operand (36 * cout * cin floats each)
    const unsigned per = (unsigned)6 * cout * cin;          // threads per operand
    unsigned e = blockIdx.x * WN_BLOCK + threadIdx.x;
    if (e >= (mode == 2 ? 2 * per : per)) return;
    const int md = mode == 2 ? (e >= per ? 1 : 0) : mode;
    float* dst = u;
    if (e >= per) {
        e -= per;
        dst += (size_t)WN_FREQ * cout * cin;
    }
    const int CI = md ? cout : cin;
    const unsigned n_chunks = (unsigned)CI >> 3;
    // e = ((cb * n_chunks + chunk) * 6 + fi) * 256 + wc * 128 + (j * 16 + n) * 2 + s
    const int s_ = (int)(e & 1), n = (int)((e >> 1) & 15), j = (int)((e >> 5) & 3), wc = (int)((e >> 7) & 1);
    unsigned t = e >> 8;
    const int fi = (int)(t % 6u);
    t /= 6u;
    const int chunk = (int)(t % n_chunks), cb = (int)(t / n_chunks);
    const int co = cb * 32 + wc * 16 + n, ci = chunk * 8 + 2 * j + s_;
    const float* const gp = w + (md ? ((size_t)ci * cin + co) : ((size_t)co * cin + ci)) * 9;
    double tb[3];       // (G g)[fi][b]
#pragma unroll
    for (int b2 = 0; b2 < 3; ++b2) {
        double acc = 0.0;
#pragma unroll
        for (int a = 0; a < 3; ++a) acc += WN_G[fi][a] * (double)(md ? gp[(2 - a) * 3 + (2 - b2)] : gp[a * 3 + b2]);
        tb[b2] = acc;
    }
    dst += ((size_t)(cb * n_chunks + chunk) * WN_FREQ + fi * 6) * WN_IMG + (e & 255u);
#pragma unroll
    for (int fj = 0; fj < 6; ++fj) dst[(size_t)fj * WN_IMG] = (float)(tb[0] * WN_G[fj][0] + tb[1] * WN_G[fj][1] + tb[2] * WN_G[fj][2]);
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

// 6x6 input patch of (tile, channel): rows y0-1 .. y0+4, columns x0-1 .. x0+4.  The vector-memory path charges per LANE
// ADDRESS (~16 cycles per wave-instruction whatever its width: 24 narrow loads per patch made the loads, not the matrix
// pipe, the pace-setter), so a lane fetches only the four columns it owns, x0 .. x0+3, as ONE 16-byte load per row; the
// halo columns x0-1 and x0+4 are the neighbouring tiles' columns 3 and 0 and arrive by DPP from lanes t-1 / t+1 of the
// 16-tile group (tiles of a group are consecutive in x; where the sequence wraps to the next image row the halo is the
// zero padding anyway).  Only lanes 0 and 15 of a group fetch their outer halo element themselves (one more load with 2
// useful lanes per group).  Rows outside the image / tiles past the end come back as hardware zeros from an out-of-range
// buffer offset; the channel chunk advances through the scalar offset of the load.
struct PatchOff {
    unsigned offc[6];    // byte offset of (row, x0) or out of range
    unsigned offe[6];    // lane 0 of a group: (row, x0-1); lane 15: (row, x0+4); other lanes / outside the image: out of range
    bool left, right;    // halo column x0-1 / x0+4 inside the image (or fetched by this lane itself)
    bool tail;           // W % 4 == 2 and last tile of an image row: columns x0+2, x0+3 are padding
};

__device__ __forceinline__ PatchOff wn_patch_off(int tile, int t, int chan, int C, const WinoGeom& g) {
    PatchOff p;
    const bool exists = tile < g.n_tiles;
    const int tl = exists ? tile : 0;
    const int b = tl / g.tiles_img, rem = tl - b * g.tiles_img;
    const int ty = rem / g.tiles_x, tx = rem - ty * g.tiles_x;
    const int y0 = 4 * ty, x0 = 4 * tx;
    const int base = ((b * C + chan) * g.H + (y0 - 1)) * g.W + x0;
    const bool has_l = x0 > 0, has_r = x0 + 4 < g.W;
    p.left = has_l || t == 0;
    p.right = has_r || t == 15;
    p.tail = x0 + 3 >= g.W;
#pragma unroll
    for (int r = 0; r < 6; ++r) {
        const int yy = y0 - 1 + r;
        const bool rv = exists && yy >= 0 && yy < g.H;
        const int row = base + r * g.W;
        p.offc[r] = rv ? (unsigned)row * 4u : WN_OOB;
        p.offe[r] = (rv && t == 0 && has_l) ? (unsigned)(row - 1) * 4u : (rv && t == 15 && has_r) ? (unsigned)(row + 4) * 4u : WN_OOB;
    }
    return p;
}

// Patch as the transform wants it: column pairs (1,2), (3,4) and (0,5) as 2-wide vectors (v_pk_* math in the column pass)
struct Patch {
    f32x2 p[6], q[6], e[6];     // (d[r][1], d[r][2]), (d[r][3], d[r][4]), (d[r][0], d[r][5])
};

__device__ __forceinline__ float wn_dpp_from_prev(float own, float v) {   // lane t <- lane t-1 of its 16-lane row; lane 0 keeps own
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, own), __builtin_bit_cast(int, v), 0x111, 0xF, 0xF, false));
}
__device__ __forceinline__ float wn_dpp_from_next(float own, float v) {   // lane t <- lane t+1; lane 15 keeps own
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, own), __builtin_bit_cast(int, v), 0x101, 0xF, 0xF, false));
}

__device__ __forceinline__ void wn_load_patch(__amdgpu_buffer_rsrc_t rsrc, const PatchOff& o, unsigned soff, Patch& d) {
#pragma unroll
    for (int r = 0; r < 6; ++r) {
        const f32x4 c = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, o.offc[r], soff, 0));
        const float e = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc, o.offe[r], soff, 0));
        const float l = wn_dpp_from_prev(e, c[3]), rr = wn_dpp_from_next(e, c[0]);
        d.p[r] = f32x2{c[0], c[1]};
        d.q[r] = f32x2{o.tail ? 0.0f : c[2], o.tail ? 0.0f : c[3]};
        d.e[r] = f32x2{o.left ? l : 0.0f, o.right ? rr : 0.0f};
    }
}

// The same patch as it leaves the loads: nothing here touches the loaded registers, so the compiler has no reason to wait for
// them at the point of issue (wn_load_patch's DPP / select post-processing forces an s_waitcnt right behind the loads: zero
// look-ahead).  wn_finish_patch does that post-processing when the patch is transformed, two chunks later.
struct PatchRaw {
    f32x4 c[6];
    float e[6];
    bool left, right, tail;
};

__device__ __forceinline__ void wn_load_raw(__amdgpu_buffer_rsrc_t rsrc, const PatchOff& o, unsigned soff, PatchRaw& d) {
#pragma unroll
    for (int r = 0; r < 6; ++r) {
        d.c[r] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, o.offc[r], soff, 0));
        d.e[r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc, o.offe[r], soff, 0));
    }
    d.left = o.left, d.right = o.right, d.tail = o.tail;
}

__device__ __forceinline__ void wn_finish_patch(const PatchRaw& s, Patch& d) {
#pragma unroll
    for (int r = 0; r < 6; ++r) {
        const float l = wn_dpp_from_prev(s.e[r], s.c[r][3]), rr = wn_dpp_from_next(s.e[r], s.c[r][0]);
        d.p[r] = f32x2{s.c[r][0], s.c[r][1]};
        d.q[r] = f32x2{s.tail ? 0.0f : s.c[r][2], s.tail ? 0.0f : s.c[r][3]};
        d.e[r] = f32x2{s.left ? l : 0.0f, s.right ? rr : 0.0f};
    }
}

__device__ __forceinline__ f32x2 wn_fma2(float c, f32x2 a, f32x2 b) {
    return __builtin_elementwise_fma(f32x2{c, c}, a, b);
}

// B^T on two independent columns at once
__device__ __forceinline__ void wn_bt2(const f32x2 (&d)[6], f32x2 (&v)[6]) {
    const f32x2 a = d[3] - d[1], b = d[4] - d[2];
    v[0] = wn_fma2(1.5f, a, wn_fma2(-2.0f, d[2], d[0] + d[4]));
    v[1] = wn_fma2(-2.5f, d[2], wn_fma2(0.5f, d[3], d[1] + d[4]));
    v[2] = wn_fma2(2.5f, d[3], wn_fma2(0.5f, d[2], d[4] - d[1]));
    v[3] = wn_fma2(2.0f, a, b);
    v[4] = wn_fma2(-0.5f, a, b);
    v[5] = wn_fma2(1.5f, b, wn_fma2(-2.0f, d[3], d[1] + d[5]));
}

// fp32 MFMAs and fp32 vector instructions run on the SAME lanes of a SIMD (matrix fp32 peak = packed vector fp32 peak; the ablation
// of either kernel shows their times ADD): the transforms are not hidden behind the matrix work, every vector instruction counts.  So
// the transforms are written on register PAIRS (v_pk_*_f32: two lanes' worth per issue slot): the column passes on two columns at a
// time, the row passes inside one 6-vector with broadcast / negated halves (op_sel, neg_lo / neg_hi).
__device__ __forceinline__ f32x2 w2_lo(f32x2 a) { return __builtin_shufflevector(a, a, 0, 0); }
__device__ __forceinline__ f32x2 w2_hi(f32x2 a) { return __builtin_shufflevector(a, a, 1, 1); }
__device__ __forceinline__ f32x2 w2_fma(f32x2 a, f32x2 b, f32x2 c) { return __builtin_elementwise_fma(a, b, c); }
// a pair of fp32 constants as the 64-bit scalar operand of a packed instruction
#define W2_K(a_, b_) ((unsigned long long)__builtin_bit_cast(unsigned, (float)(a_)) | ((unsigned long long)__builtin_bit_cast(unsigned, (float)(b_)) << 32))
#define W2_KK(a_) W2_K(a_, a_)
// The packed instructions themselves: left alone the compiler splits a packed operation with a constant or a half-broadcast operand into
// two single ones (an inline constant is free there), which is the wrong trade when every vector issue slot is taken from the MFMAs.
__device__ __forceinline__ f32x2 w2_add(f32x2 a, f32x2 b) {
    f32x2 d;
    asm("v_pk_add_f32 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b));
    return d;
}
__device__ __forceinline__ f32x2 w2_sub(f32x2 a, f32x2 b) {
    f32x2 d;
    asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(d) : "v"(a), "v"(b));
    return d;
}
__device__ __forceinline__ f32x2 w2_mul(f32x2 a, f32x2 b) {
    f32x2 d;
    asm("v_pk_mul_f32 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b));
    return d;
}
__device__ __forceinline__ f32x2 w2_fmak(unsigned long long k, f32x2 a, f32x2 c) {       // k * a + c
    f32x2 d;
    asm("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(d) : "s"(k), "v"(a), "v"(c));
    return d;
}
// B^T on two independent columns at once (wn_bt2 in packed instructions)
__device__ __forceinline__ void w2_bt2(const f32x2 (&d)[6], f32x2 (&v)[6]) {
    const f32x2 a = w2_sub(d[3], d[1]), b = w2_sub(d[4], d[2]);
    v[0] = w2_fmak(W2_KK(1.5f), a, w2_fmak(W2_KK(-2.0f), d[2], w2_add(d[0], d[4])));
    v[1] = w2_fmak(W2_KK(-2.5f), d[2], w2_fmak(W2_KK(0.5f), d[3], w2_add(d[1], d[4])));
    v[2] = w2_fmak(W2_KK(2.5f), d[3], w2_fmak(W2_KK(0.5f), d[2], w2_sub(d[4], d[1])));
    v[3] = w2_fmak(W2_KK(2.0f), a, b);
    v[4] = w2_fmak(W2_KK(-0.5f), a, b);
    v[5] = w2_fmak(W2_KK(1.5f), b, w2_fmak(W2_KK(-2.0f), d[3], w2_add(d[1], d[5])));
}

// Row pass of B^T d B on ONE row of the column-passed patch, p = (d1, d2), q = (d3, d4), e = (d0, d5):
//   (a, b) = q - p;  (v3, v4) = (2, -.5) a + b;  (v1, v2) = (d4 + d1, d4 - d1) + (.5, 2.5) d3 + (-2.5, .5) d2;
//   (v0, v5) = 1.5 (a, b) + e + (d4, d1) - 2 (d2, d3)          (the last two terms cross the pairs: four single instructions)
// 6 packed + 4 single instructions instead of 16 single ones.
__device__ __forceinline__ void wn_bt_row(f32x2 p, f32x2 q, f32x2 e, float (&v)[6]) {
    f32x2 ab, v34, x, z, v12, w;
    asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(ab) : "v"(q), "v"(p));
    asm("v_pk_fma_f32 %0, %1, %2, %2 op_sel:[0,0,1] op_sel_hi:[1,0,1]" : "=v"(v34) : "s"(W2_K(2.0f, -0.5f)), "v"(ab));
    asm("v_pk_add_f32 %0, %1, %2 op_sel:[1,0] op_sel_hi:[1,0] neg_hi:[0,1]" : "=v"(x) : "v"(q), "v"(p));
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,0,0] op_sel_hi:[0,1,1]" : "=v"(z) : "v"(q), "s"(W2_K(0.5f, 2.5f)), "v"(x));
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[1,1,1]" : "=v"(v12) : "v"(p), "s"(W2_K(-2.5f, 0.5f)), "v"(z));
    asm("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(w) : "s"(W2_K(1.5f, 1.5f)), "v"(ab), "v"(e));
    v[0] = __builtin_fmaf(-2.0f, p[1], w[0] + q[1]);
    v[1] = v12[0];
    v[2] = v12[1];
    v[3] = v34[0];
    v[4] = v34[1];
    v[5] = __builtin_fmaf(-2.0f, q[0], w[1] + p[0]);
}

// patch -> B^T d B, written to the A image slot of the thread's (group, tile row m, channel c)
__device__ __forceinline__ void wn_input_transform_store(const Patch& d, float* __restrict__ dst) {
    f32x2 tp[6], tq[6], te[6];      // column pass: t[i][1..2], t[i][3..4], t[i][0 | 5]
    w2_bt2(d.p, tp);
    w2_bt2(d.q, tq);
    w2_bt2(d.e, te);
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        float v[6];
        wn_bt_row(tp[i], tq[i], te[i], v);
#pragma unroll
        for (int j = 0; j < 6; ++j) dst[(i * 6 + j) * WN_IMG] = v[j];
    }
}

// MFMA row m of a 16-tile group holds tile perm(m) = 4 (m & 3) + (m >> 2) of the group (an involution).  In the
// accumulator lane quad q, register r is row 4 q + r = tile 4 r + q: for a fixed register the four lane quads hold four
// CONSECUTIVE tiles, so one store instruction writes 64 contiguous bytes per channel and image row.
__device__ __forceinline__ int wn_row_of_tile(int t) { return 4 * (t & 3) + (t >> 2); }

// Output transform + store of one wave's 16 tiles x 16 channels: lane = (channel n = lane & 15, quad q = lane >> 4),
// accumulator register r = tile 4 r + q of the group (wn_row_of_tile).
__device__ __forceinline__ void wn_epilogue(const f32x4 (&acc)[WN_FREQ], float* __restrict__ y, const float* __restrict__ bias,
                                            int tile_base, int co, const WinoGeom& g, int lane) {
    const float bv = bias ? bias[co] : 0.0f;
    const bool vec4 = (g.W & 3) == 0;      // image rows 16-byte aligned: one 16-byte store per tile row
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int tile = tile_base + 4 * r + (lane >> 4);
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
                if (vec4) {
                    *reinterpret_cast<f32x4*>(p) = f32x4{o0 + bv, o1 + bv, o2 + bv, o3 + bv};
                } else {
                    *reinterpret_cast<f32x2*>(p) = f32x2{o0 + bv, o1 + bv};
                    if (right) *reinterpret_cast<f32x2*>(p + 2) = f32x2{o2 + bv, o3 + bv};
                }
            }
        }
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
    const PatchOff pa = wn_patch_off(tile0 + t_grp * 16 + t_t, t_t, t_c, g.Cin, g);
    const __amdgpu_buffer_rsrc_t xr = wn_rsrc(x, (unsigned)((size_t)g.B * g.Cin * g.H * g.W * 4u));
    float* const a_dst = ldsA + t_grp * 128 + ((t_c >> 1) * 16 + wn_row_of_tile(t_t)) * 2 + (t_c & 1);
    const unsigned chunk_bytes = (unsigned)(WN_KC * g.H * g.W) * 4u;

    // ---- matrix role: wave = (wt, wc)
    const int wt = wave >> 1, wc = wave & 1;
    const f32x2* const a_frag = reinterpret_cast<const f32x2*>(ldsA) + wt * 64 + lane;
    const f32x2* const b_frag = reinterpret_cast<const f32x2*>(ldsB) + wc * 64 + lane;
    const float* const u_blk = u + (size_t)cb * g.n_chunks * (WN_FREQ * WN_IMG) + tid * 4;

    f32x4 acc[WN_FREQ];
#pragma unroll
    for (int f = 0; f < WN_FREQ; ++f) acc[f] = f32x4{0.f, 0.f, 0.f, 0.f};

    Patch d;
    wn_load_patch(xr, pa, 0u, d);
    for (int chunk = 0; chunk < g.n_chunks; ++chunk) {
        // transformed filters of this chunk: 36 KiB straight into the B image (lane-linear, 16 B per lane)
        const float* usrc = u_blk + (size_t)chunk * (WN_FREQ * WN_IMG);
#pragma unroll
        for (int it = 0; it < (WN_FREQ * WN_IMG) / (WN_BLOCK * 4); ++it)
            __builtin_amdgcn_global_load_lds(usrc + it * (WN_BLOCK * 4), ldsB + it * (WN_BLOCK * 4) + wave * 256, 16, 0, 0);
        wn_input_transform_store(d, a_dst);
        __syncthreads();
        if (chunk + 1 < g.n_chunks) wn_load_patch(xr, pa, (unsigned)(chunk + 1) * chunk_bytes, d);   // in flight under the MFMAs
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

    wn_epilogue(acc, y, bias, tile0 + wt * 16, cb * WN_COUT + wc * 16 + (lane & 15), g, lane);
}

// ------------------------------------------------------------------------------------------------------------------
// Wave-specialised persistent variant (the default): 512-thread workgroups, one per CU.
//   waves 0-3 (one per SIMD)  CONSUMERS: own the 36 x f32x4 accumulators of (16 tiles x 16 channels), read the A / B
//                             fragments of the current chunk from LDS a few frequencies ahead of the MFMAs, run the output
//                             transform at the end of a unit.  They issue no global loads and almost no VALU work.
//   waves 4-7 (one per SIMD)  PRODUCERS: fetch the 6x6 patches two chunks ahead (registers), transform chunk q+1 and write
//                             its A image, and start the 36 KiB LDS-DMA of its transformed filters, all while the consumers
//                             multiply chunk q.  Double-buffered A and B images (144 KiB), ONE barrier per chunk.
// With every wave doing both jobs (wino_fwd_kernel) the 144 accumulator registers, the prefetched patch and the transform
// temporaries do not fit 256 VGPRs: no room to read fragments ahead, each MFMA group waits for its own LDS round trip.
//
// Work split ("stream-K"): the job is the sequence of (unit, chunk) steps, unit = (tile block, channel block) in that
// order; workgroup w of G takes steps [w S / G, (w + 1) S / G) - every CU multiplies the same number of chunks (+-1) although
// the unit count (556 or 288 for the neck layers) is no multiple of the 256 CUs.  A unit cut by a range boundary is finished
// by the workgroup that holds its chunk 0 (the OWNER): the other workgroups on it (their ranges START inside the unit, so it is
// the first thing they do) write their output-transformed partial sums to a per-workgroup slab and raise a per-wave flag
// (agent-scope release); the owner, which reaches that unit at the END of its range, acquires the flags and adds the slabs in
// workgroup order before its single store of y.  Fixed summation order: deterministic.  The slab bytes are stored
// write-through (sc1) and each writing wave drains them before its flag store (cdna_hip_programming.md Guideline 16); a flag is
// lowered again by the one wave that waited for it, so the caller zeroes the workspace once, not before every launch.
// ------------------------------------------------------------------------------------------------------------------
constexpr int WS_BLOCK = 512;
constexpr size_t WS_FLAG_BYTES = 4096;   // WS_MAX_GRID x 4 ints, at the start of the workspace
constexpr int WS_MAX_GRID = 256;    // workgroups (= slabs) at most: one per CU of an MI355X

struct WinoCursor {
    int unit, chunk;
};
constexpr int WS_SLAB_FLOATS = WN_COUT * WN_TILES * 16;      // one unit's outputs: [wave][channel n][row i][quad tile][4]

__device__ __forceinline__ long long ws_range_lo(int w, int G, long long S) { return (long long)w * S / G; }

// Output transform of one wave's 16 tiles x 16 channels with the three endings of a stream-K segment:
//   slab_out != null            contributor: partial sums -> slab (no bias), then publish
//   else                        add the n_in contributor slabs slab_in + k * stride (k = 0 .. n_in - 1, in order), bias, store y
// Slab element of (lane n, quad q, register r, row i): ((n * 4 + i) * 4 + r) * 16 + q * 4 .. +3 inside the wave's quarter:
// the four quads of a store instruction write 64 contiguous bytes.
__device__ __forceinline__ void ws_epilogue(const f32x4 (&acc)[WN_FREQ], float* __restrict__ y, const float* __restrict__ bias,
                                            int tile_base, int co, const WinoGeom& g, int lane, float* __restrict__ slab_out,
                                            const float* __restrict__ slab_in, int n_in, size_t slab_stride) {
    const float bv = (bias && !slab_out) ? bias[co] : 0.0f;
    const bool vec4 = (g.W & 3) == 0;
    const int n = lane & 15, q = lane >> 4;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int tile = tile_base + 4 * r + q;
        float tmp[4][6];
#pragma unroll
        for (int j = 0; j < 6; ++j)
            wn_at(acc[0 * 6 + j][r], acc[1 * 6 + j][r], acc[2 * 6 + j][r], acc[3 * 6 + j][r], acc[4 * 6 + j][r], acc[5 * 6 + j][r],
                  tmp[0][j], tmp[1][j], tmp[2][j], tmp[3][j]);
        f32x4 o[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            float o0, o1, o2, o3;
            wn_at(tmp[i][0], tmp[i][1], tmp[i][2], tmp[i][3], tmp[i][4], tmp[i][5], o0, o1, o2, o3);
            o[i] = f32x4{o0, o1, o2, o3};
        }
        if (slab_out) {
            // write-through (sc1) stores: the bytes leave this XCD's L2 without an L2-wide release fence (a buffer_wbl2 per
            // publishing wave wrote back every dirty line of the XCD - the other workgroups' output rows included)
            const __amdgpu_buffer_rsrc_t sr = wn_rsrc(slab_out, (unsigned)(WS_SLAB_FLOATS / 4) * 4u);
#pragma unroll
            for (int i = 0; i < 4; ++i)
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4w, o[i]), sr, (unsigned)(((n * 4 + i) * 4 + r) * 16 + q * 4) * 4u, 0, 16);
            continue;
        }
        for (int k = 0; k < n_in; ++k) {
            const float* sl = slab_in + (size_t)k * slab_stride;
#pragma unroll
            for (int i = 0; i < 4; ++i) o[i] += *reinterpret_cast<const f32x4*>(sl + ((n * 4 + i) * 4 + r) * 16 + q * 4);
        }
        if (tile >= g.n_tiles) continue;
        const int b = tile / g.tiles_img, rem = tile - b * g.tiles_img;
        const int ty = rem / g.tiles_x, tx = rem - ty * g.tiles_x;
        float* const row0 = y + (((size_t)b * g.Cout + co) * g.H + 4 * ty) * g.W + 4 * tx;
        const bool right = 4 * tx + 3 < g.W;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            if (4 * ty + i < g.H) {
                float* p = row0 + (size_t)i * g.W;
                const f32x4 v = o[i] + f32x4{bv, bv, bv, bv};
                if (vec4) {
                    *reinterpret_cast<f32x4*>(p) = v;
                } else {
                    *reinterpret_cast<f32x2*>(p) = f32x2{v[0], v[1]};
                    if (right) *reinterpret_cast<f32x2*>(p + 2) = f32x2{v[2], v[3]};
                }
            }
        }
    }
}

__global__ void __launch_bounds__(WS_BLOCK, 2)
wino_fwd_ws_kernel(const float* __restrict__ x, const float* __restrict__ u, const float* __restrict__ bias,
                   float* __restrict__ y, const WinoGeom g, const int n_units, float* __restrict__ slabs,
                   int* __restrict__ flags, const int gang, const int ablate_arg, unsigned* __restrict__ fault) {
    const int ablate = TODA_ABLATE ? ablate_arg : 0;
    __shared__ float lds[4 * WN_FREQ * WN_IMG];   // A0 | A1 | B0 | B1
    constexpr int IMG = WN_FREQ * WN_IMG;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int w = wn_work_id();
    // Gangs: the n_cout_blocks channel blocks of a tile block read the SAME input patches.  When the grid is a multiple of
    // that count, consecutive workgroups (one XCD under the work-id remap) form a gang that walks the tile blocks in step,
    // member m always on channel block m: the four (eight) readers of a patch run at the same time and share it in L2
    // instead of re-fetching it 16 chunks later, when the XCD's 4 MiB have long been replaced (PMC: 204 MB fetched per
    // launch against 36 MB of input).  The sequence a gang splits by stream-K is then (tile block, chunk); the partner of a
    // cut unit is the same member of the next gang.  Otherwise (gsz = 1) the sequence is (tile block, channel block, chunk).
    // Round 4: a gang may hold FEWER members than there are channel blocks (gang = its size, a divisor of the count).  The sequence is
    // then (channel-block group, tile block, chunk): an eighth of it - one XCD - stays inside one group, whose transformed filters
    // (members x 36 x Cin x 32 floats, sized by the host to fit the XCD's L2 beside the streaming patches) are then read from HBM once per
    // XCD instead of once per TILE BLOCK (PMC r04: 350 MB per launch, of which the 9.4 MB of filters of a 256 -> 256 layer 36 times over).
    const int gsz = gang ? gang : 1;
    const int G = gridDim.x / gsz;                  // ranges of the step sequence
    const int rng = w / gsz, member = w - rng * gsz;
    const int n_groups = gang ? g.n_cout_blocks / gsz : 1;
    const long long S = (long long)(gang ? n_groups * g.n_tile_blocks : n_units) * g.n_chunks;
    const long long lo = ws_range_lo(rng, G, S), hi = ws_range_lo(rng + 1, G, S);
    // (64-bit division runs on the vector ALU: hand the wave-uniform results back to scalar registers, or every use as a
    // scalar operand - the loads' soffset - becomes a waterfall loop)
    const int total = __builtin_amdgcn_readfirstlane((int)(hi - lo));   // chunks this workgroup multiplies = barriers every wave passes
    if (total == 0) return;
    auto tb_of = [&](int useq) { return gang ? useq % g.n_tile_blocks : useq / g.n_cout_blocks; };
    auto cb_of = [&](int useq) { return gang ? (useq / g.n_tile_blocks) * gsz + member : useq % g.n_cout_blocks; };

    if (wave < 4) {
        // ------------------------------------------------------------------ consumers
        const int wt = wave >> 1, wc = wave & 1;
        const f32x2* const a_frag = reinterpret_cast<const f32x2*>(lds) + wt * 64 + lane;
        const f32x2* const b_frag = reinterpret_cast<const f32x2*>(lds + 2 * IMG) + wc * 64 + lane;
        int q = 0;
        long long s = lo;
        while (s < hi) {
            const int unit = (int)(s / g.n_chunks);
            const int c_begin = (int)(s - (long long)unit * g.n_chunks);
            const int c_end = (hi - s < g.n_chunks - c_begin) ? c_begin + (int)(hi - s) : g.n_chunks;
            const int tb = tb_of(unit), cb = cb_of(unit);
            f32x4 acc[WN_FREQ];
#pragma unroll
            for (int f = 0; f < WN_FREQ; ++f) acc[f] = f32x4{0.f, 0.f, 0.f, 0.f};
            for (int chunk = c_begin; chunk < c_end; ++chunk, ++q) {
                __syncthreads();                                  // images of chunk q complete, images of q - 1 free
                // Fragment reads as inline-asm ds_read_b64 with a counted lgkmcnt: left to itself hipcc fuses the 8-byte reads of
                // two frequencies into ds_read2st64_b64, which moves the same bytes at HALF the LDS rate (128 instead of 256
                // B/clk, MI355X_MICROARCH.md LDS table) - with 72 fragment reads per chunk and wave the LDS pipe, not the matrix
                // pipe, then sets the pace.  LDS returns in order: after the reads of frequencies f+4, f+5 are issued, 8 newer
                // reads than those of f, f+1 are outstanding.
                if (ablate & 8) continue;
                const unsigned a_addr = (unsigned)(size_t)(__attribute__((address_space(3))) const void*)(a_frag + (q & 1) * (IMG / 2));
                const unsigned b_addr = (unsigned)(size_t)(__attribute__((address_space(3))) const void*)(b_frag + (q & 1) * (IMG / 2));
                f32x2 fa[WN_FREQ], fb[WN_FREQ];
#define WN_RD(f_)                                                                                                      \
    asm volatile("ds_read_b64 %0, %2 offset:%4\n\tds_read_b64 %1, %3 offset:%4" : "=&v"(fa[f_]), "=&v"(fb[f_]) : "v"(a_addr), "v"(b_addr), "n"((f_) * 1024))
#define WN_WAIT(f_, n_)                                                                                                \
    asm volatile("s_waitcnt lgkmcnt(" #n_ ")" : "+v"(fa[f_]), "+v"(fb[f_]), "+v"(fa[(f_) + 1]), "+v"(fb[(f_) + 1]))
                WN_RD(0);
                WN_RD(1);
                WN_RD(2);
                WN_RD(3);
#pragma unroll
                for (int f = 0; f < WN_FREQ; f += 2) {
                    if (f + 4 < WN_FREQ) {
                        WN_RD(f + 4);
                        WN_RD(f + 5);
                        WN_WAIT(f, 8);
                    } else if (f + 2 < WN_FREQ) {
                        WN_WAIT(f, 4);
                    } else {
                        WN_WAIT(f, 0);
                    }
                    if (ablate & 8) continue;
                    acc[f] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[f][0], fb[f][0], acc[f], 0, 0, 0);
                    acc[f + 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[f + 1][0], fb[f + 1][0], acc[f + 1], 0, 0, 0);
                    acc[f] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[f][1], fb[f][1], acc[f], 0, 0, 0);
                    acc[f + 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[f + 1][1], fb[f + 1][1], acc[f + 1], 0, 0, 0);
                }
#undef WN_RD
#undef WN_WAIT
            }
            s += c_end - c_begin;
            const int tile_base = tb * WN_TILES + wt * 16, co = cb * WN_COUT + wc * 16 + (lane & 15);
            if (ablate & 16) {
                if (acc[0][0] + acc[35][3] + acc[17][1] == 12345.f) y[0] = 1.f;
            } else if (c_begin > 0) {
                // contributor: this range starts inside the unit - partial sums to this workgroup's slab, then publish
                float* const mine = slabs + (size_t)w * WS_SLAB_FLOATS + wave * (WS_SLAB_FLOATS / 4);
                ws_epilogue(acc, y, bias, tile_base, co, g, lane, mine, nullptr, 0, 0);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // this wave's write-through stores have left
                if (lane == 0) __hip_atomic_store(flags + w * 4 + wave, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            } else {
                // owner (or sole worker) of the unit: wait for the workgroups w+1.. whose ranges start inside it
                int n_in = 0;
                if (c_end < g.n_chunks) {
                    const long long unit_end = (long long)(unit + 1) * g.n_chunks;
                    for (int r2 = rng + 1; r2 < G && ws_range_lo(r2, G, S) < unit_end; ++r2) {
                        int* const fl = flags + (r2 * gsz + member) * 4 + wave;       // same member of the following ranges
                        // bounded: the contributor sits in a later workgroup, which is resident or next in line
                        // only while dispatch is in id order and the grid fits the CUs it was sized for; if that ever fails the
                        // wait gives up after ~2 s, raises the fault word and the launch ends with a wrong unit, not a hung GPU
                        unsigned polls = 0;
                        while (__hip_atomic_load(fl, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0) {
                            if (++polls > FAULT_SPIN_LIMIT) {
                                if (lane == 0) fault_raise(fault, TODA_FAULT_WINO);
                                break;
                            }
                            __builtin_amdgcn_s_sleep(8);
                        }
                        ++n_in;
                    }
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                }
                ws_epilogue(acc, y, bias, tile_base, co, g, lane, nullptr,
                            slabs + (size_t)(w + gsz) * WS_SLAB_FLOATS + wave * (WS_SLAB_FLOATS / 4), n_in, (size_t)gsz * WS_SLAB_FLOATS);
                if (n_in) {      // every flag is read by exactly one wave: that wave lowers it again once the slab is in its registers,
                                 // so the workspace is all-zero between launches and no memset node is needed in front of each one
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    if (lane == 0)
                        for (int k2 = 1; k2 <= n_in; ++k2) __hip_atomic_store(flags + (w + k2 * gsz) * 4 + wave, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }

            }
        }
    } else {
        // ------------------------------------------------------------------ producers
        const int pt = tid - 256;
        const int t_t = pt & 15, t_c = (pt >> 4) & 7, t_grp = pt >> 7;
        const int a_off = t_grp * 128 + ((t_c >> 1) * 16 + wn_row_of_tile(t_t)) * 2 + (t_c & 1);
        const unsigned chunk_bytes = (unsigned)(WN_KC * g.H * g.W) * 4u;

        auto patch_of = [&](int unit) {
            const int tb = tb_of(unit);
            return wn_patch_off(tb * WN_TILES + t_grp * 16 + t_t, t_t, t_c, g.Cin, g);
        };
        // Everything a chunk needs from global memory (its 6x6 patch and the thread's 9 x 16 bytes of the transformed filters)
        // is LOADED TWO CHUNKS AHEAD into registers and only touched when the chunk is produced; the barrier of a producer is
        // "my LDS writes are done" (lgkmcnt) - it never waits for loads in flight.  Measured (128 -> 128 @ 188^2, 105 us): this
        // removes the forced waits the ISA showed (a patch post-processed at issue, a vmcnt(0) in front of every barrier) but not
        // the time - the loads cost ~10 us each (patches, filters) whether they get one period of cover or two, i.e. memory-path
        // throughput, not latency; the filters by LDS-DMA issued first in the period measure the same.  What does not overlap
        // at all is the transform's vector-ALU work: fp32 MFMA and packed fp32 VALU share one datapath on this part (equal
        // vendor peaks, 157.3 TFLOP/s), so the ~180 VALU instructions per chunk and producer wave ADD ~0.3 us to every chunk's 1.0
        // us of MFMAs (ablation: arithmetic without stores +10.7 us, stores without arithmetic +2.7 us).
        constexpr int B_IT = IMG / 1024;
        struct Stage {
            PatchRaw d;
            f32x4 b[B_IT];
        };
        auto produce = [&](const Stage& st, int buf) {
            float* const bdst = lds + (2 + buf) * IMG + pt * 4;
#pragma unroll
            for (int it = 0; it < B_IT; ++it) *reinterpret_cast<f32x4*>(bdst + it * 1024) = st.b[it];
            Patch d;
            wn_finish_patch(st.d, d);
            if (ablate & 64) {             // 36 LDS stores, no transform arithmetic
                float* const dst = lds + buf * IMG + a_off;
#pragma unroll
                for (int i = 0; i < 6; ++i) {
                    dst[(i * 6 + 0) * WN_IMG] = d.e[i][0];
                    dst[(i * 6 + 1) * WN_IMG] = d.p[i][0];
                    dst[(i * 6 + 2) * WN_IMG] = d.p[i][1];
                    dst[(i * 6 + 3) * WN_IMG] = d.q[i][0];
                    dst[(i * 6 + 4) * WN_IMG] = d.q[i][1];
                    dst[(i * 6 + 5) * WN_IMG] = d.e[i][1];
                }
            } else if (!(ablate & 4)) wn_input_transform_store(d, lds + buf * IMG + a_off);
            else lds[buf * IMG + a_off] = d.p[0][0] + d.e[5][1] + d.q[2][0];
        };

        // chunk stream of this workgroup: step lo + q -> (unit, chunk); units are consecutive
        const int unit0 = __builtin_amdgcn_readfirstlane((int)(lo / g.n_chunks));
        const int chunk0 = __builtin_amdgcn_readfirstlane((int)(lo - (long long)unit0 * g.n_chunks));
        WinoCursor ld{unit0, chunk0};       // next chunk to fetch
        PatchOff pa = patch_of(unit0);
        const unsigned x_bytes = (unsigned)((size_t)g.B * g.Cin * g.H * g.W * 4u), u_bytes = (unsigned)((size_t)WN_FREQ * g.Cin * g.Cout * 4u);
        const unsigned u_voff = (unsigned)pt * 16u;
        auto fetch = [&](Stage& st, int q_fetch) {
            // ALWAYS the same 21 loads, past the end of the range against an empty descriptor (hardware zeros, no memory access):
            // the compiler's vmcnt bookkeeping is then the same on every path, and a wait for the previous period's registers
            // never has to cover "or the fetch was skipped" by draining the loads just issued.  No vector-ALU work between the
            // loads: per-thread offsets are fixed registers, the chunk moves the scalar offset.
            const bool live = q_fetch < total;
            const __amdgpu_buffer_rsrc_t xr_q = wn_rsrc(x, live && !(ablate & 1) ? x_bytes : 0u);
            const __amdgpu_buffer_rsrc_t ur_q = wn_rsrc(u, live && !(ablate & 2) ? u_bytes : 0u);
            wn_load_raw(xr_q, pa, (unsigned)ld.chunk * chunk_bytes, st.d);
            const unsigned u_soff = (unsigned)((cb_of(ld.unit) * g.n_chunks + ld.chunk) * IMG) * 4u;
#pragma unroll
            for (int it = 0; it < B_IT; ++it)
                st.b[it] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(ur_q, u_voff, u_soff + (unsigned)it * 4096u, 0));
            if (live) {
                if (++ld.chunk == g.n_chunks) {
                    ld.chunk = 0;
                    ld.unit += 1;
                    if (q_fetch + 1 < total && (gang || ld.unit % g.n_cout_blocks == 0)) pa = patch_of(ld.unit);   // next tile block
                }
            }
        };
        auto barrier = [&]() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); };
        Stage s0, s1;
        fetch(s0, 0);
        fetch(s1, 1);
        produce(s0, 0);                 // chunk 0 -> images 0
        // loop body unrolled by two so that the stage registers alternate without copies
        int q = 0;
        while (true) {
            barrier();                  // barrier q
            if (q + 1 >= total) break;
            fetch(s0, q + 2);
            produce(s1, (q + 1) & 1);
            ++q;
            barrier();                  // barrier q
            if (q + 1 >= total) break;
            fetch(s1, q + 2);
            produce(s0, (q + 1) & 1);
            ++q;
        }
    }
}

// ------------------------------------------------------------------------------------------------------------------
// Weight gradient in the Winograd domain: dU[f][ci][co] = sum over tiles V[f][tile][ci] * dM[f][tile][co],
// V = B^T x B (the forward's input transform), dM = A dY A^T (the transpose of the output transform), then
// dw[co][ci] = G^T dU G.  Same machine as the forward kernel - the contraction index is the TILE instead of the input
// channel: a chunk is 8 consecutive tiles, the A image holds V (rows = 32 input channels), the B image dM (columns = 32 output
// channels), 36 frequencies x 2 MFMA k-steps per chunk and consumer wave.  Producers: waves 4-5 transform x (one thread =
// one channel x two neighbouring tiles: both patches of an image row from two 16-byte loads and two halo loads), waves 6-7
// transform dY; a thread's two tiles are the two k-slots of one 8-byte LDS store.
// Stream-K over (unit = (ci block, co block), chunk): every segment's 36 x 32 x 32 partial sums go to slab (w + unit);
// wino_wgrad_finish_kernel adds the slabs of a unit in workgroup order and applies G^T . G.  No float atomics.
// ------------------------------------------------------------------------------------------------------------------
constexpr int WG_KT = 8;                                  // tiles per chunk
constexpr int WG_SLAB_FLOATS = WN_FREQ * 32 * 32;         // [f][co 32][ci 32]

// Producer thread of the wgrad kernel: ONE tile k of the chunk and one channel slot n of each operand (input channel
// 32 cib + slot, output channel 32 cob + slot).  Lanes of a quad are four consecutive tiles: the inner halo columns of the x
// patch arrive by DPP inside the quad, lanes 0 / 3 of the quad fetch their outer halo element themselves.  A half-wave covers
// 4 tiles x 8 channels, so the 4-byte LDS stores of a frequency hit each bank at most twice (free).
struct TilePos {
    int b, ty, tx;
    bool exists;
};

__device__ __forceinline__ TilePos wn_tile_pos(int tile, const WinoGeom& g) {
    TilePos t;
    t.exists = tile < g.n_tiles;
    const int tl = t.exists ? tile : 0;
    t.b = tl / g.tiles_img;
    const int rem = tl - t.b * g.tiles_img;
    t.ty = rem / g.tiles_x;
    t.tx = rem - t.ty * g.tiles_x;
    return t;
}

__device__ __forceinline__ void wn_tile_advance(TilePos& t, int step, int tile_after, const WinoGeom& g) {
    t.tx += step;
    while (t.tx >= g.tiles_x) {
        t.tx -= g.tiles_x;
        if (++t.ty == g.tiles_y) {
            t.ty = 0;
            ++t.b;
        }
    }
    t.exists = tile_after < g.n_tiles;
}

__device__ __forceinline__ float wn_quad_from_prev(float own, float v) {   // lane t <- lane t-1 of its quad (quad_perm [0,0,1,2]); lane 0: own
    const float s = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x90, 0xF, 0xF, false));
    return (threadIdx.x & 3) == 0 ? own : s;
}
__device__ __forceinline__ float wn_quad_from_next(float own, float v) {   // lane t <- lane t+1 of its quad (quad_perm [1,2,3,3]); lane 3: own
    const float s = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xF9, 0xF, 0xF, false));
    return (threadIdx.x & 3) == 3 ? own : s;
}

struct XRaw {
    f32x4 c[6];
    float e[6];
    bool left, right, tail;
};

__device__ __forceinline__ void wn_load_x(__amdgpu_buffer_rsrc_t rsrc, const TilePos& t, int chan, const WinoGeom& g, XRaw& d) {
    const int y0 = 4 * t.ty, x0 = 4 * t.tx, tq = threadIdx.x & 3;
    const int base = ((t.b * g.Cin + chan) * g.H + (y0 - 1)) * g.W + x0;
    const bool has_l = x0 > 0, has_r = x0 + 4 < g.W;
    d.left = has_l || tq == 0;
    d.right = has_r || tq == 3;
    d.tail = x0 + 3 >= g.W;
#pragma unroll
    for (int r = 0; r < 6; ++r) {
        const int yy = y0 - 1 + r;
        const bool rv = t.exists && yy >= 0 && yy < g.H;
        const int row = base + r * g.W;
        d.c[r] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, rv ? (unsigned)row * 4u : WN_OOB, 0, 0));
        const unsigned oe = (rv && tq == 0 && has_l) ? (unsigned)(row - 1) * 4u : (rv && tq == 3 && has_r) ? (unsigned)(row + 4) * 4u : WN_OOB;
        d.e[r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc, oe, 0, 0));
    }
}

__device__ __forceinline__ void wn_x_transform_store(const XRaw& d, float* __restrict__ dst) {
    Patch a;
#pragma unroll
    for (int r = 0; r < 6; ++r) {
        const f32x4 c = d.c[r];
        const float l = wn_quad_from_prev(d.e[r], c[3]), rr = wn_quad_from_next(d.e[r], c[0]);
        a.p[r] = f32x2{c[0], c[1]};
        a.q[r] = f32x2{d.tail ? 0.0f : c[2], d.tail ? 0.0f : c[3]};
        a.e[r] = f32x2{d.left ? l : 0.0f, d.right ? rr : 0.0f};
    }
    wn_input_transform_store(a, dst);
}

struct DyRaw {
    f32x4 c[4];
    bool tail;
};

__device__ __forceinline__ void wn_load_dy(__amdgpu_buffer_rsrc_t rsrc, const TilePos& t, int chan, const WinoGeom& g, DyRaw& d) {
    const int y0 = 4 * t.ty, x0 = 4 * t.tx;
    const int base = ((t.b * g.Cout + chan) * g.H + y0) * g.W + x0;
    d.tail = x0 + 3 >= g.W;
#pragma unroll
    for (int i = 0; i < 4; ++i)
        d.c[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, (t.exists && y0 + i < g.H) ? (unsigned)(base + i * g.W) * 4u : WN_OOB, 0, 0));
}

// dY tile (4 x 4) -> A dY A^T, on register pairs: the column pass on the column pairs (0, 1) and (2, 3), each row of the row pass in
// five packed instructions  (P = (y0, y1), Q = (y2, y3): (e, o) = P + Q; (m1, m2) = e -+ o; (m3, m4) = y0 + (.5, -2) y1 + (.25, 4) y2 +
// (.125, -8) y3; m0 = y0, m5 = y3)
__device__ __forceinline__ void wn_a_cols(const f32x2 (&y)[4], f32x2 (&m)[6]) {
    const f32x2 e = w2_add(y[0], y[2]), o = w2_add(y[1], y[3]);
    m[0] = y[0];
    m[1] = w2_sub(e, o);
    m[2] = w2_add(e, o);
    m[3] = w2_fmak(W2_KK(0.125f), y[3], w2_fmak(W2_KK(0.25f), y[2], w2_fmak(W2_KK(0.5f), y[1], y[0])));
    m[4] = w2_fmak(W2_KK(-8.0f), y[3], w2_fmak(W2_KK(4.0f), y[2], w2_fmak(W2_KK(-2.0f), y[1], y[0])));
    m[5] = y[3];
}
__device__ __forceinline__ void wn_dy_transform_store(const DyRaw& d, float* __restrict__ dst) {
    f32x2 a[4], b[4], ma[6], mb[6];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        a[i] = f32x2{d.c[i][0], d.c[i][1]};
        b[i] = f32x2{d.tail ? 0.0f : d.c[i][2], d.tail ? 0.0f : d.c[i][3]};
    }
    wn_a_cols(a, ma);
    wn_a_cols(b, mb);
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        const f32x2 P = ma[i], Q = mb[i];
        const f32x2 eo = w2_add(P, Q);
        f32x2 m12, u;
        asm("v_pk_add_f32 %0, %1, %1 op_sel:[0,1] op_sel_hi:[0,1] neg_lo:[0,1]" : "=v"(m12) : "v"(eo));
        asm("v_pk_fma_f32 %0, %1, %2, %1 op_sel:[1,0,0] op_sel_hi:[1,1,0]" : "=v"(u) : "v"(P), "s"(W2_K(0.5f, -2.0f)));
        asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,0,0] op_sel_hi:[0,1,1]" : "+v"(u) : "v"(Q), "s"(W2_K(0.25f, 4.0f)));
        asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,0,0] op_sel_hi:[1,1,1]" : "+v"(u) : "v"(Q), "s"(W2_K(0.125f, -8.0f)));
        dst[(i * 6 + 0) * WN_IMG] = P[0];
        dst[(i * 6 + 1) * WN_IMG] = m12[0];
        dst[(i * 6 + 2) * WN_IMG] = m12[1];
        dst[(i * 6 + 3) * WN_IMG] = u[0];
        dst[(i * 6 + 4) * WN_IMG] = u[1];
        dst[(i * 6 + 5) * WN_IMG] = Q[1];
    }
}

struct WgradGeom {
    WinoGeom g;
    int n_ci_blocks, n_co_blocks, n_units, steps_per_unit;
};

__global__ void __launch_bounds__(WS_BLOCK, 2)
wino_wgrad_kernel(const float* __restrict__ x, const float* __restrict__ dy, const WgradGeom wg, float* __restrict__ slabs, const int ablate_arg) {
    const int ablate = TODA_ABLATE ? ablate_arg : 0;
    __shared__ float lds[4 * WN_FREQ * WN_IMG];   // V0 | V1 | dM0 | dM1
    constexpr int IMG = WN_FREQ * WN_IMG;
    const WinoGeom& g = wg.g;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int G = gridDim.x;
    const int w = wn_work_id();
    const long long S = (long long)wg.n_units * wg.steps_per_unit;
    const long long lo = ws_range_lo(w, G, S), hi = ws_range_lo(w + 1, G, S);
    const int total = __builtin_amdgcn_readfirstlane((int)(hi - lo));
    if (total == 0) return;

    if (wave < 4) {
        // consumers: rows = input channels (V image, group wm), columns = output channels (dM image, group wn)
        const int wm = wave >> 1, wn = wave & 1;
        const f32x2* const a_frag = reinterpret_cast<const f32x2*>(lds) + wm * 64 + lane;
        const f32x2* const b_frag = reinterpret_cast<const f32x2*>(lds + 2 * IMG) + wn * 64 + lane;
        int q = 0;
        long long s = lo;
        while (s < hi) {
            const int unit = (int)(s / wg.steps_per_unit);
            const int c_begin = (int)(s - (long long)unit * wg.steps_per_unit);
            const int c_end = (hi - s < wg.steps_per_unit - c_begin) ? c_begin + (int)(hi - s) : wg.steps_per_unit;
            f32x4 acc[WN_FREQ];
#pragma unroll
            for (int f = 0; f < WN_FREQ; ++f) acc[f] = f32x4{0.f, 0.f, 0.f, 0.f};
            for (int chunk = c_begin; chunk < c_end; ++chunk, ++q) {
                __syncthreads();
                // fragment reads as single ds_read_b64 with a counted lgkmcnt, as in wino_fwd_ws_kernel (left alone hipcc pairs
                // them into ds_read2st64_b64: the same bytes at half the LDS rate)
                const unsigned a_addr = (unsigned)(size_t)(__attribute__((address_space(3))) const void*)(a_frag + (q & 1) * (IMG / 2));
                const unsigned b_addr = (unsigned)(size_t)(__attribute__((address_space(3))) const void*)(b_frag + (q & 1) * (IMG / 2));
                f32x2 fa[WN_FREQ], fb[WN_FREQ];
#define WN_RD(f_)                                                                                                      \
    asm volatile("ds_read_b64 %0, %2 offset:%4\n\tds_read_b64 %1, %3 offset:%4" : "=&v"(fa[f_]), "=&v"(fb[f_]) : "v"(a_addr), "v"(b_addr), "n"((f_) * 1024))
#define WN_WAIT(f_, n_)                                                                                                \
    asm volatile("s_waitcnt lgkmcnt(" #n_ ")" : "+v"(fa[f_]), "+v"(fb[f_]), "+v"(fa[(f_) + 1]), "+v"(fb[(f_) + 1]))
                WN_RD(0);
                WN_RD(1);
                WN_RD(2);
                WN_RD(3);
#pragma unroll
                for (int f = 0; f < WN_FREQ; f += 2) {
                    if (f + 4 < WN_FREQ) {
                        WN_RD(f + 4);
                        WN_RD(f + 5);
                        WN_WAIT(f, 8);
                    } else if (f + 2 < WN_FREQ) {
                        WN_WAIT(f, 4);
                    } else {
                        WN_WAIT(f, 0);
                    }
                    if (ablate & 8) continue;
                    acc[f] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[f][0], fb[f][0], acc[f], 0, 0, 0);
                    acc[f + 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[f + 1][0], fb[f + 1][0], acc[f + 1], 0, 0, 0);
                    acc[f] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[f][1], fb[f][1], acc[f], 0, 0, 0);
                    acc[f + 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[f + 1][1], fb[f + 1][1], acc[f + 1], 0, 0, 0);
                }
#undef WN_RD
#undef WN_WAIT
            }
            s += c_end - c_begin;
            // segment done: acc[f][r] = dU[f][ci = 16 wm + 4 quad + r][co = 16 wn + (lane & 15)] -> slab [f][co][ci]
            float* const sl = slabs + (size_t)(w + unit) * WG_SLAB_FLOATS + (wn * 16 + (lane & 15)) * 32 + wm * 16 + (lane >> 4) * 4;
#pragma unroll
            for (int f = 0; f < WN_FREQ; ++f) *reinterpret_cast<f32x4*>(sl + f * 1024) = acc[f];
        }
    } else {
        // producers: thread = (tile k of the chunk, channel slot); see TilePos
        const int pw = wave - 4;
        const int k = (lane & 3) + 4 * (pw & 1), slot = (lane >> 2) + 16 * (pw >> 1);
        const int img_off = (slot >> 4) * 128 + ((k >> 1) * 16 + (slot & 15)) * 2 + (k & 1);
        int unit = __builtin_amdgcn_readfirstlane((int)(lo / wg.steps_per_unit));
        int chunk = __builtin_amdgcn_readfirstlane((int)(lo - (long long)unit * wg.steps_per_unit));
        TilePos tp = wn_tile_pos(chunk * WG_KT + k, g);
        const unsigned x_bytes = (unsigned)((size_t)g.B * g.Cin * g.H * g.W * 4u), y_bytes = (unsigned)((size_t)g.B * g.Cout * g.H * g.W * 4u);

        // Two register stages, loaded two chunks ahead and untouched until their chunk is produced; always the same loads (an
        // empty descriptor past the end of the range: hardware zeros, no memory access) so that the compiler's vmcnt waits are the
        // same on every path; the producers' barrier waits for their LDS writes only (see wino_fwd_ws_kernel).
        struct WStage {
            XRaw x;
            DyRaw y;
        };
        auto fetch = [&](WStage& st, int q_fetch) {
            const bool live = q_fetch < total;
            const __amdgpu_buffer_rsrc_t xr = wn_rsrc(x, live && !(ablate & 1) ? x_bytes : 0u), yr = wn_rsrc(dy, live && !(ablate & 2) ? y_bytes : 0u);
            const int cib = unit / wg.n_co_blocks, cob = unit - cib * wg.n_co_blocks;
            wn_load_x(xr, tp, cib * 32 + slot, g, st.x);
            wn_load_dy(yr, tp, cob * 32 + slot, g, st.y);
            if (live) {
                if (++chunk == wg.steps_per_unit) {
                    chunk = 0;
                    ++unit;
                    tp = wn_tile_pos(k, g);
                } else {
                    wn_tile_advance(tp, WG_KT, chunk * WG_KT + k, g);
                }
            }
        };
        auto produce = [&](const WStage& st, int buf) {
            if (ablate & 4) {           // the stores without the transform arithmetic
                float* const d0 = lds + buf * IMG + img_off;
                float* const d1 = lds + (2 + buf) * IMG + img_off;
#pragma unroll
                for (int f = 0; f < WN_FREQ; ++f) {
                    d0[f * WN_IMG] = st.x.c[f % 6][f & 3];
                    d1[f * WN_IMG] = st.y.c[f & 3][f % 4];
                }
                return;
            }
            wn_x_transform_store(st.x, lds + buf * IMG + img_off);
            wn_dy_transform_store(st.y, lds + (2 + buf) * IMG + img_off);
        };
        auto barrier = [&]() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); };
        WStage s0, s1;
        fetch(s0, 0);
        fetch(s1, 1);
        produce(s0, 0);
        int q = 0;
        while (true) {
            barrier();                  // barrier q: images of chunk q published
            if (q + 1 >= total) break;
            fetch(s0, q + 2);
            produce(s1, (q + 1) & 1);
            ++q;
            barrier();
            if (q + 1 >= total) break;
            fetch(s1, q + 2);
            produce(s0, (q + 1) & 1);
            ++q;
        }
    }
}

// stage 1 of the fold: red[unit][f][co][ci] = sum of the unit's segment slabs in workgroup order (one thread per element:
// coalesced, and enough threads even when a 64-channel layer has 4 units cut into 64 segments each)
template <int SLAB>
__global__ void __launch_bounds__(WN_BLOCK)
wino_wgrad_reduce_kernel(const float* __restrict__ slabs, const int n_units, const int steps_per_unit, int G, float* __restrict__ red) {
    // SLAB is a multiple of the block size: a block lies inside one unit, whose segment list (which workgroups of the
    // stream-K split touched it) thread 0 works out once - the 64-bit divisions of ws_range_lo per thread and per segment cost more
    // than the fold itself; the loads of eight segments are then in flight together, the adds stay in workgroup order
    static_assert(SLAB % WN_BLOCK == 0, "a fold block must not straddle two units");
    __shared__ int seg[WS_MAX_GRID];
    __shared__ int n_seg;
    const long long e = (long long)blockIdx.x * WN_BLOCK + threadIdx.x;
    const int unit = (int)(((long long)blockIdx.x * WN_BLOCK) / SLAB);
    if (threadIdx.x == 0) {
        const long long S = (long long)n_units * steps_per_unit;
        const long long u_lo = (long long)unit * steps_per_unit, u_hi = u_lo + steps_per_unit;
        int w0 = (int)(u_lo * G / S);
        while (w0 > 0 && ws_range_lo(w0, G, S) > u_lo) --w0;
        while (w0 + 1 < G && ws_range_lo(w0 + 1, G, S) <= u_lo) ++w0;
        int n = 0;
        long long lo = ws_range_lo(w0, G, S);
        for (int w = w0; w < G && lo < u_hi; ++w) {
            const long long next = ws_range_lo(w + 1, G, S);
            if (next > u_lo && next != lo) seg[n++] = w;
            lo = next;
        }
        n_seg = n;
    }
    __syncthreads();
    if (unit >= n_units) return;
    const int off = (int)(e - (long long)unit * SLAB);
    const float* p = slabs + (size_t)unit * SLAB + off;
    const int n = n_seg;
    float acc = 0.0f;
    int i = 0;
    for (; i + 8 <= n; i += 8) {
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = p[(size_t)seg[i + j] * SLAB];
#pragma unroll
        for (int j = 0; j < 8; ++j) acc += v[j];
    }
    for (; i < n; ++i) acc += p[(size_t)seg[i] * SLAB];
    red[e] = acc;
}

// stage 2: dw[co][ci][3][3] = G^T dU G
__global__ void __launch_bounds__(WN_BLOCK)
wino_wgrad_finish_kernel(const float* __restrict__ red, const WgradGeom wg, float* __restrict__ dw) {
    const int e = blockIdx.x * WN_BLOCK + threadIdx.x;
    const int cin = wg.g.Cin, cout = wg.g.Cout;
    if (e >= cin * cout) return;
    const int co = e / cin, ci = e - co * cin;
    const int unit = (ci >> 5) * wg.n_co_blocks + (co >> 5);
    const float* sl = red + (size_t)unit * WG_SLAB_FLOATS + (co & 31) * 32 + (ci & 31);
    double du[WN_FREQ];
#pragma unroll
    for (int f = 0; f < WN_FREQ; ++f) du[f] = (double)sl[f * 1024];
    double t[3][6];     // G^T dU: t[a][j] = sum_i G[i][a] dU[i][j]
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int j = 0; j < 6; ++j) {
            double acc = 0.0;
#pragma unroll
            for (int i = 0; i < 6; ++i) acc += WN_G[i][a] * du[i * 6 + j];
            t[a][j] = acc;
        }
    float* out = dw + ((size_t)co * cin + ci) * 9;
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int b = 0; b < 3; ++b) {
            double acc = 0.0;
#pragma unroll
            for (int j = 0; j < 6; ++j) acc += t[a][j] * WN_G[j][b];
            out[a * 3 + b] = (float)acc;
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
    TODA_CHECK_ARG(w && u && mode >= 0 && mode <= 2, "conv3x3_transform_weight: null pointer or bad mode");
    const int CO = mode == 1 ? cin : cout, CI = mode == 1 ? cout : cin;
    TODA_CHECK_ARG(CO % WN_COUT == 0 && CI % WN_KC == 0 && CO > 0 && CI > 0 && (mode != 2 || (CI % WN_COUT == 0)),
                   "conv3x3_transform_weight: produced channels %% 32 and contracted channels %% 8 must be 0 (got %d, %d)", CO, CI);
    const long long threads = 6LL * cout * cin * (mode == 2 ? 2 : 1);
    TODA_CHECK_ARG(threads * 6 < (1LL << 31), "conv3x3_transform_weight: operand above 2^31 elements");
    hipLaunchKernelGGL(wino_weight_kernel, dim3(cdiv(threads, WN_BLOCK)), dim3(WN_BLOCK), 0, (hipStream_t)stream, w, cout, cin, mode, u);
    TODA_LAUNCH_CHECK();
    return TODA_OK;
}

extern "C" size_t toda_conv3x3_workspace_bytes(void) {
    // 4 flags per workgroup, then 256 per-workgroup slabs of one unit's partial outputs (64 KiB each); see wino_fwd_ws_kernel
    return WS_FLAG_BYTES + (size_t)WS_MAX_GRID * WS_SLAB_FLOATS * sizeof(float);
}

extern "C" int toda_conv3x3_fwd(const float* x, const float* u, const float* bias, int batch, int cin, int cout, int H, int W,
                                float* y, void* ws, size_t ws_bytes, void* stream) {
    TODA_CHECK_ARG(x && u && y, "conv3x3_fwd: null pointer");
    WinoGeom g;
    int rc = wino_geom("conv3x3_fwd", batch, cin, cout, H, W, &g);
    if (rc) return rc;
    static const int variant = getenv("TODA_WINO_VARIANT") ? atoi(getenv("TODA_WINO_VARIANT")) : 1;
    const int n_units = g.n_tile_blocks * g.n_cout_blocks;
    if (variant == 0) {
        hipLaunchKernelGGL(wino_fwd_kernel, dim3(n_units), dim3(WN_BLOCK), 0, (hipStream_t)stream, x, u, bias, y, g);
    } else {
        TODA_CHECK_ARG(ws != nullptr, "conv3x3_fwd: workspace missing");
        if (ws_bytes < toda_conv3x3_workspace_bytes()) {
            toda::set_error("conv3x3_fwd: workspace too small (%zu < %zu bytes)", ws_bytes, toda_conv3x3_workspace_bytes());
            return TODA_EWORKSPACE;
        }
        static int n_cu = 0;
        if (!n_cu) {
            int dev = 0;
            hipDeviceProp_t prop;
            TODA_HIP(hipGetDevice(&dev));
            TODA_HIP(hipGetDeviceProperties(&prop, dev));
            n_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
            if (n_cu > WS_MAX_GRID) n_cu = WS_MAX_GRID;
        }
        // one 144-KiB workgroup per CU; never more workgroups than chunk steps
        // gang mode: n_cout_blocks workgroups per range of the (tile block, chunk) sequence (see the kernel)
        // gang size: the largest power-of-two divisor of the channel-block count whose filters (size x 36 x Cin x 32 floats) fit
        // TODA_WINO_GANG_KB (2560 KiB of the XCD's 4 MiB L2); TODA_WINO_GANG = 0: no gangs, 1: this rule, -1: round 3's gang of all blocks
        static const int env_gang = getenv("TODA_WINO_GANG") ? atoi(getenv("TODA_WINO_GANG")) : 1;
        static const int gang_kb = getenv("TODA_WINO_GANG_KB") ? atoi(getenv("TODA_WINO_GANG_KB")) : 2560;
        const int ncb = g.n_cout_blocks;
        int gang = 0;
        if (env_gang && ncb > 1 && ncb <= 32 && (32 % ncb) == 0 && (n_cu % ncb) == 0) {
            gang = ncb;
            const long long slice = (long long)WN_FREQ * cin * WN_COUT * 4;
            while (env_gang > 0 && gang > 1 && gang * slice > (long long)gang_kb * 1024) gang >>= 1;
            if ((long long)(ncb / gang) * g.n_tile_blocks * g.n_chunks < n_cu / gang) gang = 0;
        }
        const long long steps = (long long)n_units * g.n_chunks;
        const int grid = gang ? n_cu : (steps < n_cu ? (int)steps : n_cu);
        int* flags = (int*)ws;                                  // 4 words per workgroup: zero on entry, zero again on exit
        float* slabs = (float*)((char*)ws + WS_FLAG_BYTES);
        static const int ablate = (TODA_ABLATE && getenv("TODA_WINO_ABLATE")) ? atoi(getenv("TODA_WINO_ABLATE")) : 0;
        if (const unsigned fv = fault_take()) {
            // a bounded wait of an earlier launch gave up: its flags may still be up - put the workspace back into its all-zero
            // state behind that launch and report
            (void)hipMemsetAsync(ws, 0, WS_FLAG_BYTES, (hipStream_t)stream);
            toda::set_error("conv3x3_fwd: device fault word 0x%x raised by an earlier launch (bounded inter-workgroup wait gave up: %s%s) - its "
                            "results are invalid; the stream-K flags have been re-zeroed (TODA_WINO_VARIANT=0 selects the kernel without hand-offs)",
                            fv, (fv & TODA_FAULT_BN2D) ? "bn2d split kernel " : "", (fv & TODA_FAULT_WINO) ? "wino_fwd_ws_kernel" : "");
            return TODA_EFAULT;
        }
        hipLaunchKernelGGL(wino_fwd_ws_kernel, dim3(grid), dim3(WS_BLOCK), 0, (hipStream_t)stream, x, u, bias, y, g, n_units, slabs, flags,
                           gang, ablate, fault_word_dev());
    }
    TODA_LAUNCH_CHECK();
    return TODA_OK;
}

static int wgrad_geom(const char* who, int batch, int cin, int cout, int H, int W, WgradGeom* wg) {
    int rc = wino_geom(who, batch, cin, cout, H, W, &wg->g);
    if (rc) return rc;
    TODA_CHECK_ARG(cin % 32 == 0, "%s: input channels must be a multiple of 32 (got %d)", who, cin);
    wg->n_ci_blocks = cin / 32;
    wg->n_co_blocks = cout / 32;
    wg->n_units = wg->n_ci_blocks * wg->n_co_blocks;
    wg->steps_per_unit = cdiv(wg->g.n_tiles, WG_KT);
    return TODA_OK;
}


extern "C" size_t toda_conv3x3_wgrad_workspace_bytes(int batch, int cin, int cout, int H, int W) {
    // one slab per stream-K segment (index w + unit) + the per-unit sums
    (void)batch, (void)H, (void)W;
    const size_t units = (size_t)(cin / 32) * (cout / 32);
    return (WS_MAX_GRID + 2 * units) * WG_SLAB_FLOATS * sizeof(float);
}

extern "C" int toda_conv3x3_wgrad(const float* x, const float* dy, int batch, int cin, int cout, int H, int W, float* dw, void* ws,
                                  size_t ws_bytes, void* stream) {
    TODA_CHECK_ARG(x && dy && dw, "conv3x3_wgrad: null pointer");
    WgradGeom wg;
    int rc = wgrad_geom("conv3x3_wgrad", batch, cin, cout, H, W, &wg);
    if (rc) return rc;
    const size_t need = toda_conv3x3_wgrad_workspace_bytes(batch, cin, cout, H, W);
    if (!ws || ws_bytes < need) {
        toda::set_error("conv3x3_wgrad: workspace too small (%zu < %zu bytes)", ws_bytes, need);
        return TODA_EWORKSPACE;
    }
    static int n_cu = 0;
    if (!n_cu) {
        int dev = 0;
        hipDeviceProp_t prop;
        TODA_HIP(hipGetDevice(&dev));
        TODA_HIP(hipGetDeviceProperties(&prop, dev));
        n_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
        if (n_cu > WS_MAX_GRID) n_cu = WS_MAX_GRID;
    }
    const long long steps = (long long)wg.n_units * wg.steps_per_unit;
    const int grid = steps < n_cu ? (int)steps : n_cu;
    static const int ablate = (TODA_ABLATE && getenv("TODA_WINO_WG_ABLATE")) ? atoi(getenv("TODA_WINO_WG_ABLATE")) : 0;   // measurement builds only
    hipLaunchKernelGGL(wino_wgrad_kernel, dim3(grid), dim3(WS_BLOCK), 0, (hipStream_t)stream, x, dy, wg, (float*)ws, ablate);
    TODA_LAUNCH_CHECK();
    float* red = (float*)ws + (size_t)(WS_MAX_GRID + wg.n_units) * WG_SLAB_FLOATS;
    hipLaunchKernelGGL(wino_wgrad_reduce_kernel<WG_SLAB_FLOATS>, dim3(cdiv((long long)wg.n_units * WG_SLAB_FLOATS, WN_BLOCK)), dim3(WN_BLOCK), 0,
                       (hipStream_t)stream, (const float*)ws, wg.n_units, wg.steps_per_unit, grid, red);
    TODA_LAUNCH_CHECK();
    hipLaunchKernelGGL(wino_wgrad_finish_kernel, dim3(cdiv((long long)cin * cout, WN_BLOCK)), dim3(WN_BLOCK), 0, (hipStream_t)stream,
                       (const float*)red, wg, dw);
    TODA_LAUNCH_CHECK();
    return TODA_OK;
}
