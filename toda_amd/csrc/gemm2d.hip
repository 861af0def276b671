// The three convolutions of the BEV neck that are not 3x3 / stride 1 (reference pcdet/models/backbones_2d/base_bev_backbone.py:32-36,
// 47-66): the stride-2 3x3 head of a block (ZeroPad2d(1) + Conv2d(3, stride 2)), and the up-sampling deblocks
// ConvTranspose2d(k = s, stride = s) with s = 1 (a 1x1 convolution) and s = 2.  Forward, data gradient and weight gradient, NCHW in
// and out, on the fp32 matrix cores - in round 2 these ran on MIOpen / rocBLAS (implicit-GEMM + transposes, GEMM + im2col / col2im:
// 1.2 ms of an 18 ms step, 23 launches).
//
// One kernel core: C[R1 x R2] += O1[R1 x c] . O2[R2 x c]^T over tiles of 128 x 128 x 16, 4 waves of 64 x 64
// (v_mfma_f32_16x16x4_f32, 16 accumulator tiles per wave), operands staged through LDS as [row][16 contraction values] so that a
// lane's fragment of FOUR MFMA steps is one ds_read_b128 (the k-permutation of spconv.hip: lane group g holds contraction indices
// 4g..4g+3, step j contracts over {4g'+j}), register-staged double buffering.  What differs between the nine GEMMs is only how an
// operand element is found in memory and where a result element goes - "accessors":
//   forward / data gradient: O1 = weights (rows = produced channels), O2 = pixels of the map (rows = pixels, contraction = gathered
//     channels x taps); the pixel index mapping carries the stride-2 gather, the pixel shuffle of the transposed convolution (as an
//     epilogue store: no separate shuffle pass) and the parity classes of the stride-2 data gradient (an input pixel reaches only the
//     taps of its parity: 1, 2, 2 or 4 of 9 - one launch covers the four classes, no multiplication by structural zeros);
//   weight gradient: both operands are maps, the contraction runs over pixels, split over workgroups into slabs that a second
//     kernel adds in fixed order (deterministic, no float atomics).
#include <stdlib.h>

#include "common.h"
#include "split_common.cuh"

extern "C" int toda_matrix_path(void);      // spconv.hip: 0 = native fp32 MFMA, 1 = the exact bf16 hi/mid/lo split

namespace toda {

typedef float pg4 __attribute__((ext_vector_type(4)));

constexpr int PG_K = 32;           // contraction values per stage
constexpr int PG_LD = PG_K + 4;    // LDS row stride in floats of a [row][contraction] image (16-byte aligned, spreads the b128 fragment reads)
constexpr int PG_BLOCK = 256;

// ---- operand accessors ----------------------------------------------------------------------------------------------------
// A thread fetches 8 values per stage: COL operands 8 consecutive ROWS at one contraction index (the rows - pixels - are the same
// for the whole contraction loop: init() decomposes them ONCE, load() only adds the contraction index), other operands 8
// consecutive CONTRACTION indices of one row (init() fixes the row, load() walks the contraction).  Index arithmetic is the cost
// of these kernels, not bytes: no division by a run-time value inside the loop.  Out-of-range elements read as 0.

struct NoCur {};      // (operands without a pixel cursor, see RowCur)

// dense matrix with strides: element (r, c) = p[r * sr + c * sc]
template <bool COLSHAPE>
struct MatOp {
    static constexpr bool COL = COLSHAPE;
    static constexpr bool CURSOR = false;
    typedef NoCur Cur;
    const float* p;
    int rows, cols;
    long long sr, sc;
    struct State {
        long long base;
        int r;
    };
    __device__ __forceinline__ void init(State& s, int r) const {
        s.r = r;
        s.base = (long long)r * sr;
    }
    __device__ __forceinline__ void load(const State& s, int c, float (&v)[8]) const {
        if (COL) {
            const bool cok = c < cols;
            const long long o = s.base + (long long)c * sc;
            if (sr == 1 && cok && s.r + 8 <= rows && (o & 3) == 0) {        // 8 consecutive rows = 8 consecutive floats
                const pg4 a = *reinterpret_cast<const pg4*>(p + o), b = *reinterpret_cast<const pg4*>(p + o + 4);
                v[0] = a[0], v[1] = a[1], v[2] = a[2], v[3] = a[3], v[4] = b[0], v[5] = b[1], v[6] = b[2], v[7] = b[3];
                return;
            }
#pragma unroll
            for (int i = 0; i < 8; ++i) v[i] = (cok && s.r + i < rows) ? p[o + (long long)i * sr] : 0.0f;
        } else {
            const bool rok = s.r < rows;
            if (sc == 1 && rok && c + 8 <= cols && ((s.base + c) & 3) == 0) {
                const pg4 a = *reinterpret_cast<const pg4*>(p + s.base + c), b = *reinterpret_cast<const pg4*>(p + s.base + c + 4);
                v[0] = a[0], v[1] = a[1], v[2] = a[2], v[3] = a[3], v[4] = b[0], v[5] = b[1], v[6] = b[2], v[7] = b[3];
                return;
            }
#pragma unroll
            for (int i = 0; i < 8; ++i) v[i] = (rok && c + i < cols) ? p[s.base + (long long)(c + i) * sc] : 0.0f;
        }
    }
};

// Pixels of an NCHW map as operand ROWS, channels as contraction: element (n = b * hw + pix, k) = x[(b * C + k) * hw + pix].
struct PixPlain {
    static constexpr bool COL = true;
    static constexpr bool CURSOR = false;
    typedef NoCur Cur;
    const float* x;
    int B, C, hw;
    struct State {
        long long off[8];      // element offset of channel 0 of each pixel, < 0 = outside
        bool vec;
    };
    __device__ __forceinline__ void init(State& s, int n) const {
        int b = n / hw, pix = n - b * hw;          // ONE division; the other seven pixels follow by carry
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            s.off[i] = b < B ? (long long)b * C * hw + pix : -1;
            if (++pix == hw) pix = 0, ++b;
        }
        // two aligned groups of four pixels inside one image each
        s.vec = (hw & 3) == 0 && s.off[0] >= 0 && s.off[7] >= 0 && s.off[3] == s.off[0] + 3 && s.off[7] == s.off[4] + 3;
    }
    __device__ __forceinline__ void load(const State& s, int k, float (&v)[8]) const {
        const bool kok = k < C;
        const long long d = (long long)k * hw;
        if (s.vec && kok) {
            const pg4 a = *reinterpret_cast<const pg4*>(x + s.off[0] + d), b = *reinterpret_cast<const pg4*>(x + s.off[4] + d);
            v[0] = a[0], v[1] = a[1], v[2] = a[2], v[3] = a[3], v[4] = b[0], v[5] = b[1], v[6] = b[2], v[7] = b[3];
            return;
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = (kok && s.off[i] >= 0) ? x[s.off[i] + d] : 0.0f;
    }
};

// walks n = b * hw + pix in steps without dividing: position of the thread's first pixel of the current stage
struct PixCursor {
    int b, pix;
};

// ---- contraction over PIXELS (the weight gradients) ------------------------------------------------------------------------
// The contraction index n = (b, y, x) of a stage starts at a WAVE-UNIFORM position and a thread's eight values sit a fixed 0 .. 24
// places behind it.  fp32 MFMAs and vector instructions share the lanes of a SIMD (the time of either adds to the other: measured
// on conv2d.hip's kernels), and a 32-bit multiply is quarter rate, a division a dozen instructions with four
// of those - the round-3 form of these accessors spent 7 vector instructions per MFMA (a fifth of them quarter-rate) on two
// divisions per call and 64-bit address products per element.  FAST form: the stage's position (b, y, x) is a cursor in scalar
// registers that is stepped, not divided; the (at most three) grid rows a stage can touch have their offsets and flags worked out
// there too; a thread only picks between them (compare + select) and adds, in 32 bits.  It needs grid rows of at least one stage
// (32 pixels) and maps below 2^31 elements (the host picks; the plain form stays for everything else).
struct RowCur {
    int b, y, x;           // position of the stage's first contraction index over [B][rows][width]
    int off[3];            // element offset of grid row j = 0, 1, 2 counted from (b, y): b_j * image + y_j * row
    int flags;             // bit j: b_j < B;  bit 4 + j: y_j == 0
};
__device__ __forceinline__ void rowcur_rows(RowCur& c, int B, int rows, int image, int row) {
    int b = c.b, y = c.y;
    c.flags = 0;
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        c.off[j] = b * image + y * row;
        c.flags |= ((b < B ? 1 : 0) | (y == 0 ? 16 : 0)) << j;
        if (++y == rows) y = 0, ++b;
    }
}
__device__ __forceinline__ void rowcur_set(RowCur& c, int n, int B, int rows, int width, int image, int row) {
    c.b = n / (rows * width);
    const int rem = n - c.b * rows * width;
    c.y = rem / width;
    c.x = rem - c.y * width;
    rowcur_rows(c, B, rows, image, row);
}
__device__ __forceinline__ void rowcur_step(RowCur& c, int step, int B, int rows, int width, int image, int row) {
    c.x += step;
    if (c.x >= width) {           // width >= step: one carry at most
        c.x -= width;
        if (++c.y == rows) c.y = 0, ++c.b;
    }
    rowcur_rows(c, B, rows, image, row);
}
// a thread's eight places behind the cursor: xs = column of the first one, jw = how many places its grid row still has (the rest of
// the eight lie in the next one), w = 1 when the first one is already past the cursor's row
struct RowPick {
    int xs, jw, w;
};
__device__ __forceinline__ RowPick rowcur_pick(const RowCur& c, int tn, int width) {
    RowPick p;
    p.xs = c.x + tn;
    p.w = p.xs >= width ? 1 : 0;
    p.xs -= p.w ? width : 0;
    p.jw = width - p.xs;
    return p;
}

// Channels of an NCHW map as operand rows, pixels as contraction (weight gradients): element (ch, n) = x[(b * C + ch) * hw + pix].
template <bool FAST>
struct ChanPlain {
    static constexpr bool COL = false;
    static constexpr bool CURSOR = FAST;
    const float* x;
    int B, C, hw;
    struct State {
        int ch;
    };
    typedef RowCur Cur;
    static bool fast_ok(int hw) { return (hw & 3) == 0 && hw >= PG_K; }          // an image = one grid row of hw pixels
    __device__ __forceinline__ void cur_set(Cur& c, int n) const { rowcur_set(c, n, B, 1, hw, C * hw, 0); }
    __device__ __forceinline__ void cur_step(Cur& c) const { rowcur_step(c, PG_K, B, 1, hw, C * hw, 0); }
    __device__ __forceinline__ void init(State& s, int ch) const { s.ch = ch; }
    __device__ __forceinline__ void load(const State& s, const Cur& c, int tn, float (&v)[8]) const {
        const bool ok = s.ch < C;
        const int rowbase = s.ch * hw;
#pragma unroll
        for (int h = 0; h < 2; ++h) {          // two aligned groups of four pixels, each inside one image
            const RowPick p = rowcur_pick(c, tn + 4 * h, hw);
            const bool in = ok && ((c.flags >> p.w) & 1);
            pg4 t = pg4{0.f, 0.f, 0.f, 0.f};
            if (in) t = *reinterpret_cast<const pg4*>(x + ((p.w ? c.off[1] : c.off[0]) + rowbase + p.xs));
#pragma unroll
            for (int i = 0; i < 4; ++i) v[4 * h + i] = t[i];
        }
    }
    __device__ __forceinline__ void load(const State& s, int n, float (&v)[8]) const {
        const bool ok = s.ch < C;
        const int b0 = n / hw, p0 = n - b0 * hw;          // (one division per stage and thread; the 8 pixels follow by carry)
        if ((hw & 3) == 0) {
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                int b = b0, pix = p0 + 4 * h;
                if (pix >= hw) pix -= hw, ++b;
                pg4 t = pg4{0.f, 0.f, 0.f, 0.f};
                if (ok && b < B) t = *reinterpret_cast<const pg4*>(x + ((long long)b * C + s.ch) * hw + pix);
#pragma unroll
                for (int i = 0; i < 4; ++i) v[4 * h + i] = t[i];
            }
        } else {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                int b = b0, pix = p0 + i;
                while (pix >= hw) pix -= hw, ++b;
                v[i] = (ok && b < B) ? x[((long long)b * C + s.ch) * hw + pix] : 0.0f;
            }
        }
    }
};

// stride-2 3x3 gather: x [B][C][H][W] (H, W even), output pixels (b, oy, ox) of the Ho x Wo map, taps k = ci * 9 + ky * 3 + kx
// reading x[b][ci][2 oy + ky - 1][2 ox + kx - 1] (ZeroPad2d(1) + Conv2d(3, stride 2, padding 0)).  Only the top row and the left
// column of taps can fall outside (2 oy + 1 <= H - 1 on an even map).
struct ConvS2Geom {
    const float* x;
    int B, C, H, W, Ho, Wo;
};
struct PixConvS2 {          // rows = output pixels, contraction = (ci, ky, kx)
    static constexpr bool COL = true;
    static constexpr bool CURSOR = false;
    typedef NoCur Cur;
    ConvS2Geom g;
    struct State {
        long long off[8];      // offset of x[b][0][2 oy - 1][2 ox - 1]; < 0 marks an invalid pixel through `ok`
        unsigned ok, top, left;   // bit i: pixel valid / oy == 0 / ox == 0
    };
    __device__ __forceinline__ void init(State& s, int n) const {
        const int howo = g.Ho * g.Wo;
        s.ok = s.top = s.left = 0u;
        int b = n / howo;
        const int rem = n - b * howo;
        int oy = rem / g.Wo, ox = rem - oy * g.Wo;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            s.off[i] = ((long long)b * g.C * g.H + (2 * oy - 1)) * g.W + 2 * ox - 1;
            s.ok |= (unsigned)(b < g.B) << i;
            s.top |= (unsigned)(oy == 0) << i;
            s.left |= (unsigned)(ox == 0) << i;
            if (++ox == g.Wo) {
                ox = 0;
                if (++oy == g.Ho) oy = 0, ++b;
            }
        }
    }
    __device__ __forceinline__ void load(const State& s, int k, float (&v)[8]) const {
        const int ci = k / 9, t = k - ci * 9;
        const int ky = t / 3, kx = t - ky * 3;
        const long long d = ((long long)ci * g.H + ky) * g.W + kx;
        unsigned m = ci < g.C ? s.ok : 0u;
        if (ky == 0) m &= ~s.top;
        if (kx == 0) m &= ~s.left;
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = ((m >> i) & 1u) ? g.x[s.off[i] + d] : 0.0f;
    }
};
template <bool FAST>
struct TapConvS2 {          // rows = (ci, ky, kx), contraction = output pixels (weight gradient)
    static constexpr bool COL = false;
    static constexpr bool CURSOR = FAST;
    ConvS2Geom g;
    int rows;
    struct State {
        long long d;
        int ky, kx;
        bool ok;
    };
    typedef RowCur Cur;
    // grid row = output row oy (Wo pixels), element offset of (b, oy, 0) = b * C H W + oy * 2 W
    static bool fast_ok(int Wo) { return Wo >= PG_K; }
    __device__ __forceinline__ void cur_set(Cur& c, int n) const { rowcur_set(c, n, g.B, g.Ho, g.Wo, g.C * g.H * g.W, 2 * g.W); }
    __device__ __forceinline__ void cur_step(Cur& c) const { rowcur_step(c, PG_K, g.B, g.Ho, g.Wo, g.C * g.H * g.W, 2 * g.W); }
    __device__ __forceinline__ void init(State& s, int k) const {
        const int ci = k / 9, t = k - ci * 9;
        s.ky = t / 3, s.kx = t - s.ky * 3;
        s.ok = k < rows;
        s.d = ((long long)ci * g.H + s.ky - 1) * g.W + s.kx - 1;
    }
    __device__ __forceinline__ void load(const State& s, const Cur& c, int tn, float (&v)[8]) const {
        const RowPick p = rowcur_pick(c, tn, g.Wo);
        const int fa = c.flags >> p.w, fb = fa >> 1;                  // flags of the thread's first row and of the one after it
        const bool row_a = s.ok && (fa & 1) && !(s.ky == 0 && (fa & 16)), row_b = s.ok && (fb & 1) && !(s.ky == 0 && (fb & 16));
        const unsigned low = p.jw >= 8 ? 0xFFu : (1u << p.jw) - 1u;
        unsigned m = (row_a ? low : 0u) | (row_b ? (0xFFu & ~low) : 0u);
        if (s.kx == 0) {              // the left tap of the first pixel of a row is padding
            if (p.xs == 0) m &= ~1u;
            if (p.jw < 8) m &= ~(1u << p.jw);
        }
        const int d = (int)s.d;
        const int ea = (p.w ? c.off[1] : c.off[0]) + 2 * p.xs + d, eb = (p.w ? c.off[2] : c.off[1]) + d - 2 * p.jw;
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = ((m >> i) & 1u) ? g.x[(i < p.jw ? ea : eb) + 2 * i] : 0.0f;
    }
    __device__ __forceinline__ void load(const State& s, int n, float (&v)[8]) const {
        const int howo = g.Ho * g.Wo;
        int b = n / howo;
        const int rem = n - b * howo;
        int oy = rem / g.Wo, ox = rem - oy * g.Wo;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const bool in = s.ok && b < g.B && !(s.ky == 0 && oy == 0) && !(s.kx == 0 && ox == 0);
            v[i] = in ? g.x[((long long)b * g.C * g.H + 2 * oy) * g.W + 2 * ox + s.d] : 0.0f;
            if (++ox == g.Wo) {
                ox = 0;
                if (++oy == g.Ho) oy = 0, ++b;
            }
        }
    }
};

// pixel un-shuffle of a 2x up-sampled map: dy [B][C][2h][2w]; element ((b, y, x), k = co * 4 + i * 2 + j) = dy[b][co][2y + i][2x + j]
struct UnshuffleGeom {
    const float* dy;
    int B, C, h, w;      // C = channels of dy, (h, w) = the LOW-resolution map
};
struct PixUnshuffle {       // rows = low-resolution pixels, contraction = (co, i, j)
    static constexpr bool COL = true;
    static constexpr bool CURSOR = false;
    typedef NoCur Cur;
    UnshuffleGeom g;
    struct State {
        long long off[8];
        unsigned ok;
    };
    __device__ __forceinline__ void init(State& s, int n) const {
        const int hw = g.h * g.w;
        s.ok = 0u;
        int b = n / hw;
        const int rem = n - b * hw;
        int y = rem / g.w, x = rem - y * g.w;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            s.off[i] = ((long long)b * g.C * (2 * g.h) + 2 * y) * (2 * g.w) + 2 * x;
            s.ok |= (unsigned)(b < g.B) << i;
            if (++x == g.w) {
                x = 0;
                if (++y == g.h) y = 0, ++b;
            }
        }
    }
    __device__ __forceinline__ void load(const State& s, int k, float (&v)[8]) const {
        const int co = k >> 2, i2 = (k >> 1) & 1, j2 = k & 1;
        const long long d = ((long long)co * (2 * g.h) + i2) * (2 * g.w) + j2;
        const unsigned m = co < g.C ? s.ok : 0u;
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = ((m >> i) & 1u) ? g.dy[s.off[i] + d] : 0.0f;
    }
};
template <bool FAST>
struct ChanUnshuffle {      // rows = (co, i, j), contraction = low-resolution pixels (weight gradient)
    static constexpr bool COL = false;
    static constexpr bool CURSOR = FAST;
    UnshuffleGeom g;
    int rows;
    struct State {
        long long d;
        bool ok;
    };
    typedef RowCur Cur;
    // grid row = low-resolution row y (w pixels), element offset of (b, y, 0) = b * C * 4 h w + y * 4 w
    static bool fast_ok(int w) { return w >= PG_K; }
    __device__ __forceinline__ void cur_set(Cur& c, int n) const { rowcur_set(c, n, g.B, g.h, g.w, g.C * 4 * g.h * g.w, 4 * g.w); }
    __device__ __forceinline__ void cur_step(Cur& c) const { rowcur_step(c, PG_K, g.B, g.h, g.w, g.C * 4 * g.h * g.w, 4 * g.w); }
    __device__ __forceinline__ void init(State& s, int k) const {
        const int co = k >> 2, i2 = (k >> 1) & 1, j2 = k & 1;
        s.ok = k < rows;
        s.d = ((long long)co * (2 * g.h) + i2) * (2 * g.w) + j2;
    }
    __device__ __forceinline__ void load(const State& s, const Cur& c, int tn, float (&v)[8]) const {
        const RowPick p = rowcur_pick(c, tn, g.w);
        const int fa = c.flags >> p.w, fb = fa >> 1;
        const unsigned low = p.jw >= 8 ? 0xFFu : (1u << p.jw) - 1u;
        const unsigned m = ((s.ok && (fa & 1)) ? low : 0u) | ((s.ok && (fb & 1)) ? (0xFFu & ~low) : 0u);
        const int d = (int)s.d;
        const int ea = (p.w ? c.off[1] : c.off[0]) + 2 * p.xs + d, eb = (p.w ? c.off[2] : c.off[1]) + d - 2 * p.jw;
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = ((m >> i) & 1u) ? g.dy[(i < p.jw ? ea : eb) + 2 * i] : 0.0f;
    }
    __device__ __forceinline__ void load(const State& s, int n, float (&v)[8]) const {
        const int hw = g.h * g.w;
        int b = n / hw;
        const int rem = n - b * hw;
        int y = rem / g.w, x = rem - y * g.w;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            v[i] = (s.ok && b < g.B) ? g.dy[((long long)b * g.C * (2 * g.h) + 2 * y) * (2 * g.w) + 2 * x + s.d] : 0.0f;
            if (++x == g.w) {
                x = 0;
                if (++y == g.h) y = 0, ++b;
            }
        }
    }
};

// Data gradient of the stride-2 3x3 convolution, one parity class (py, px) of the INPUT pixels per blockIdx.z: input pixel
// (2 y' + py, 2 x' + px) is reached through the taps ky = iy + 1 (mod 2): ky = 1 for even rows (output row y'), {0, 2} for odd
// rows (output rows y' + 1 and y'), the same in x.  Contraction index k = (co << lt) + tap, lt = log2(taps of the class).
struct S2Class {
    int py, px, lty, ltx;      // log2 of the taps per axis (0 or 1)
    __device__ __forceinline__ void set(int cls) {
        py = cls >> 1, px = cls & 1;
        lty = py, ltx = px;
    }
};
struct PixS2Dgrad {         // rows = input pixels of the class (b, y', x'), contraction = (co, tap of the class)
    static constexpr bool COL = true;
    static constexpr bool CURSOR = false;
    typedef NoCur Cur;
    const float* dy;        // [B][Cout][Ho][Wo]
    int B, Cout, H, W, Ho, Wo;
    S2Class c;
    struct State {
        long long off[8];      // offset of dy[b][0][y'][x']
        unsigned ok, bottom, right;
    };
    __device__ __forceinline__ void init(State& s, int n) const {
        const int hh = H / 2, ww = W / 2;
        s.ok = s.bottom = s.right = 0u;
        int b = n / (hh * ww);
        const int rem = n - b * (hh * ww);
        int yy = rem / ww, xx = rem - yy * ww;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            s.off[i] = ((long long)b * Cout * Ho + yy) * Wo + xx;
            s.ok |= (unsigned)(b < B) << i;
            s.bottom |= (unsigned)(yy + 1 >= Ho) << i;
            s.right |= (unsigned)(xx + 1 >= Wo) << i;
            if (++xx == ww) {
                xx = 0;
                if (++yy == hh) yy = 0, ++b;
            }
        }
    }
    __device__ __forceinline__ void load(const State& s, int k, float (&v)[8]) const {
        const int lt = c.lty + c.ltx;
        const int co = k >> lt, t = k & ((1 << lt) - 1);
        const int ty = c.lty ? (t >> c.ltx) : 0, tx = c.ltx ? (t & 1) : 0;
        // odd parity: tap index 0 = kernel tap 0 = output y' + 1, tap index 1 = kernel tap 2 = output y'
        const int dyy = (c.lty && ty == 0) ? 1 : 0, dxx = (c.ltx && tx == 0) ? 1 : 0;
        const long long d = ((long long)co * Ho + dyy) * Wo + dxx;
        unsigned m = co < Cout ? s.ok : 0u;
        if (dyy) m &= ~s.bottom;
        if (dxx) m &= ~s.right;
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = ((m >> i) & 1u) ? dy[s.off[i] + d] : 0.0f;
    }
};
struct WS2Dgrad {           // rows = ci, contraction = (co, tap of the class): w[co][ci][ky][kx]
    static constexpr bool COL = false;
    static constexpr bool CURSOR = false;
    typedef NoCur Cur;
    const float* w;
    int Cin, Cout;
    S2Class c;
    struct State {
        int ci;
    };
    __device__ __forceinline__ void init(State& s, int ci) const { s.ci = ci; }
    __device__ __forceinline__ void load(const State& s, int k, float (&v)[8]) const {
        const int lt = c.lty + c.ltx;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int kk = k + i;
            const int co = kk >> lt, t = kk & ((1 << lt) - 1);
            const int ty = c.lty ? (t >> c.ltx) : 0, tx = c.ltx ? (t & 1) : 0;
            const int ky = c.lty ? 2 * ty : 1, kx = c.ltx ? 2 * tx : 1;
            v[i] = (s.ci < Cin && co < Cout) ? w[(((long long)co * Cin + s.ci) * 3 + ky) * 3 + kx] : 0.0f;
        }
    }
};

// ---- result stores: column n of the tile product is decomposed once (prep), then its 16 rows are written (put) ------------------
struct StoreNCHW {          // r1 = channel m, r2 = pixel n = b * hw + pix
    static constexpr bool ROWS4 = true;      // four consecutive r2 of one r1 are four consecutive floats when hw % 4 == 0
    float* y;
    int M, N, hw;
    __device__ __forceinline__ bool vec_ok() const { return (hw & 3) == 0 && (N & 3) == 0; }
    __device__ __forceinline__ void put4(long long base, int m, pg4 v) const {
        if (base >= 0 && m < M) *reinterpret_cast<pg4*>(y + base + (long long)m * hw) = v;
    }
    __device__ __forceinline__ long long prep(int n) const {
        if (n >= N) return -1;
        const int b = n / hw, pix = n - b * hw;
        return (long long)b * M * hw + pix;
    }
    __device__ __forceinline__ void put(long long base, int m, float v) const {
        if (base >= 0 && m < M) y[base + (long long)m * hw] = v;
    }
};
struct StoreShuffle {       // r1 = (co, i, j), r2 = (b, y, x) of the low-resolution map: y_out[b][co][2y + i][2x + j]
    static constexpr bool ROWS4 = false;
    float* y;
    int M, N, C, h, w;
    __device__ __forceinline__ long long prep(int n) const {
        if (n >= N) return -1;
        const int hw = h * w;
        const int b = n / hw, rem = n - b * hw;
        const int yy = rem / w, xx = rem - yy * w;
        return ((long long)b * C * (2 * h) + 2 * yy) * (2 * w) + 2 * xx;
    }
    __device__ __forceinline__ void put(long long base, int m, float v) const {
        if (base >= 0 && m < M) {
            const int co = m >> 2, i = (m >> 1) & 1, j = m & 1;
            y[base + ((long long)co * (2 * h) + i) * (2 * w) + j] = v;
        }
    }
};
struct StoreS2Class {       // r1 = ci, r2 = (b, y', x') of the class: dx[b][ci][2y' + py][2x' + px]
    static constexpr bool ROWS4 = false;
    float* dx;
    int M, N, H, W, py, px;
    __device__ __forceinline__ long long prep(int n) const {
        if (n >= N) return -1;
        const int hh = H / 2, ww = W / 2;
        const int b = n / (hh * ww), rem = n - b * (hh * ww);
        const int yy = rem / ww, xx = rem - yy * ww;
        return ((long long)b * M * H + 2 * yy + py) * W + 2 * xx + px;
    }
    __device__ __forceinline__ void put(long long base, int m, float v) const {
        if (base >= 0 && m < M) dx[base + (long long)m * H * W] = v;
    }
};
struct StoreSlab {          // weight gradients: partial sums of contraction split z -> slab[z][r1][r2]
    static constexpr bool ROWS4 = true;
    float* slab;
    int M, N;
    __device__ __forceinline__ bool vec_ok() const { return (N & 3) == 0; }
    __device__ __forceinline__ void put4(long long base, int m, pg4 v) const {
        if (base >= 0 && m < M) *reinterpret_cast<pg4*>(slab + base + (long long)m * N) = v;
    }
    __device__ __forceinline__ long long prep(int n) const { return n < N ? (long long)blockIdx.z * M * N + n : -1; }
    __device__ __forceinline__ void put(long long base, int m, float v) const {
        if (base >= 0 && m < M) slab[base + (long long)m * N] = v;
    }
};

// ---- the tile product --------------------------------------------------------------------------------------------------------
// grid: x = tiles over R2, y = tiles over R1, z = contraction split (weight gradients) or parity class (stride-2 data gradient).
// LDS images: an operand whose CONTRACTION index is contiguous in memory is staged [row][PG_K] (16-byte stores, one ds_read_b128
// per fragment of four steps); a COL operand (rows contiguous in memory: pixels of an NCHW map) is staged as it is read,
// [contraction][row] (16-byte stores again - transposing it in the store cost 16-way bank conflicts), and its fragment is four
// ds_read_b32.  Both follow the same k-permutation: lane group g holds contraction indices 4g..4g+3 of each group of 16.
template <class Op, int T>
struct PgStage {
    static constexpr int FLOATS = Op::COL ? PG_K * (T + 4) : T * PG_LD;
};

// Results of a wave's quadrant.  D: column = lane & 15 (operand-2 row), row = 4 * (lane >> 4) + reg (operand-1 row).  Results whose
// operand-2 index is contiguous in memory go out as 16-byte stores: the quadrant is turned in `img` (a wave-private [T/2][T/2] float
// image in LDS) and a lane then owns 4 consecutive columns of a row.
template <int T, class Store>
__device__ __forceinline__ void pg_epilogue(const pg4 (&acc)[T / 32][T / 32], const Store& st, float* img, int r1_0, int r2_0) {
    constexpr int WT = T / 32;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int fi = lane & 15, fg = lane >> 4;
    if constexpr (Store::ROWS4) {
        if (st.vec_ok()) {
            constexpr int QW = T / 2;              // quadrant width
#pragma unroll
            for (int a = 0; a < WT; ++a)
#pragma unroll
                for (int b = 0; b < WT; ++b)
#pragma unroll
                    for (int reg = 0; reg < 4; ++reg) img[(a * 16 + 4 * fg + reg) * QW + b * 16 + fi] = acc[a][b][reg];
            __builtin_amdgcn_wave_barrier();
            constexpr int LPR = QW / 4;            // lanes per row
            const int c4 = (lane % LPR) * 4;
            const long long base = st.prep(r2_0 + wn * QW + c4);
#pragma unroll
            for (int it = 0; it < QW / (64 / LPR); ++it) {
                const int row = it * (64 / LPR) + lane / LPR;
                st.put4(base, r1_0 + wm * QW + row, *reinterpret_cast<const pg4*>(&img[row * QW + c4]));
            }
            return;
        }
    }
#pragma unroll
    for (int b = 0; b < WT; ++b) {
        const long long base = st.prep(r2_0 + wn * (T / 2) + b * 16 + fi);
#pragma unroll
        for (int a = 0; a < WT; ++a)
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) st.put(base, r1_0 + wm * (T / 2) + a * 16 + 4 * fg + reg, acc[a][b][reg]);
    }
}

// T = tile rows of both operands (128 or 64): 4 waves of (T / 2) x (T / 2).  The 64-row tile is for the GEMMs whose pixel count
// gives too few 128 x 128 tiles to fill 256 CUs (the 94 x 94 maps: 278 tiles).
template <int T, class Op1, class Op2, class Store>
__device__ __forceinline__ void pg_tile(const Op1& o1, const Op2& o2, const Store& st, int c_begin, int c_end) {
    constexpr int LC = T + 4;                 // row stride of a [contraction][row] image
    constexpr int WT = T / 32;                // 16 x 16 result tiles per wave and dimension
    constexpr int PER = T / 64;               // groups of 8 values a thread fetches per operand and stage
    __shared__ __attribute__((aligned(16))) float s1[2][PgStage<Op1, T>::FLOATS];
    __shared__ __attribute__((aligned(16))) float s2[2][PgStage<Op2, T>::FLOATS];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int r1_0 = blockIdx.y * T, r2_0 = blockIdx.x * T;
    const int wm = wave >> 1, wn = wave & 1;                 // quadrant of this wave
    const int fi = lane & 15, fg = lane >> 4;

    // a thread's share of a stage (PER groups of 8 values):
    //   COL: contraction index t >> 3 (0..31), rows (t & 7) * 8 PER .. ;   else: row t / (4 / PER) ..., 8 PER consecutive contraction values
    constexpr int ROW_DIV = 4 / PER;          // threads per row of a [row][contraction] image
    typename Op1::State q1[PER];
    typename Op2::State q2[PER];
#pragma unroll
    for (int h = 0; h < PER; ++h) {
        if (Op1::COL) o1.init(q1[h], r1_0 + (t & 7) * (8 * PER) + 8 * h);
        else if (h == 0) o1.init(q1[0], r1_0 + t / ROW_DIV);
        if (Op2::COL) o2.init(q2[h], r2_0 + (t & 7) * (8 * PER) + 8 * h);
        else if (h == 0) o2.init(q2[0], r2_0 + t / ROW_DIV);
    }
    float v1[PER][8], v2[PER][8];
    // operands whose contraction runs over pixels (FAST form): the stage's position as a wave-uniform cursor, stepped before every fetch
    typename Op1::Cur k1;
    typename Op2::Cur k2;
    if constexpr (Op1::CURSOR) o1.cur_set(k1, c_begin);
    if constexpr (Op2::CURSOR) o2.cur_set(k2, c_begin);
    auto fetch = [&](int c0) {
#pragma unroll
        for (int h = 0; h < PER; ++h) {
            const int tn = (t % ROW_DIV) * (8 * PER) + 8 * h;
            if constexpr (Op1::COL) o1.load(q1[h], c0 + (t >> 3), v1[h]);
            else if constexpr (Op1::CURSOR) o1.load(q1[0], k1, tn, v1[h]);
            else o1.load(q1[0], c0 + tn, v1[h]);
            if constexpr (Op2::COL) o2.load(q2[h], c0 + (t >> 3), v2[h]);
            else if constexpr (Op2::CURSOR) o2.load(q2[0], k2, tn, v2[h]);
            else o2.load(q2[0], c0 + tn, v2[h]);
        }
    };
    auto stash_one = [&](float* img, bool col, const float (&v)[PER][8]) {
        float* dst = col ? img + (t >> 3) * LC + (t & 7) * (8 * PER) : img + (t / ROW_DIV) * PG_LD + (t % ROW_DIV) * (8 * PER);
#pragma unroll
        for (int h = 0; h < PER; ++h) {
            *reinterpret_cast<pg4*>(dst + 8 * h) = pg4{v[h][0], v[h][1], v[h][2], v[h][3]};
            *reinterpret_cast<pg4*>(dst + 8 * h + 4) = pg4{v[h][4], v[h][5], v[h][6], v[h][7]};
        }
    };

    pg4 acc[WT][WT];
#pragma unroll
    for (int a = 0; a < WT; ++a)
#pragma unroll
        for (int b = 0; b < WT; ++b) acc[a][b] = pg4{0.f, 0.f, 0.f, 0.f};

    fetch(c_begin);
    stash_one(s1[0], Op1::COL, v1);
    stash_one(s2[0], Op2::COL, v2);
    __syncthreads();
    int buf = 0;
    for (int c0 = c_begin; c0 < c_end; c0 += PG_K) {
        const bool more = c0 + PG_K < c_end;
        if (more) {
            if constexpr (Op1::CURSOR) o1.cur_step(k1);
            if constexpr (Op2::CURSOR) o2.cur_step(k2);
            fetch(c0 + PG_K);
        }
#pragma unroll
        for (int kk = 0; kk < PG_K; kk += 16) {           // groups of 16 contraction values = 4 MFMA steps
            pg4 fa[WT], fb[WT];
#pragma unroll
            for (int a = 0; a < WT; ++a) {
                if (Op1::COL) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) fa[a][j] = s1[buf][(kk + 4 * fg + j) * LC + wm * (T / 2) + a * 16 + fi];
                } else {
                    fa[a] = *reinterpret_cast<const pg4*>(&s1[buf][(wm * (T / 2) + a * 16 + fi) * PG_LD + kk + 4 * fg]);
                }
            }
#pragma unroll
            for (int b = 0; b < WT; ++b) {
                if (Op2::COL) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) fb[b][j] = s2[buf][(kk + 4 * fg + j) * LC + wn * (T / 2) + b * 16 + fi];
                } else {
                    fb[b] = *reinterpret_cast<const pg4*>(&s2[buf][(wn * (T / 2) + b * 16 + fi) * PG_LD + kk + 4 * fg]);
                }
            }
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int a = 0; a < WT; ++a)
#pragma unroll
                    for (int b = 0; b < WT; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[a][j], fb[b][j], acc[a][b], 0, 0, 0);
        }
        if (more) {
            stash_one(s1[buf ^ 1], Op1::COL, v1);
            stash_one(s2[buf ^ 1], Op2::COL, v2);
        }
        __syncthreads();
        buf ^= 1;
    }
    static_assert(PgStage<Op1, T>::FLOATS * 2 >= 2 * (T / 2) * (T / 2) && PgStage<Op2, T>::FLOATS * 2 >= 2 * (T / 2) * (T / 2), "quadrant images must fit the operand buffers");
    pg_epilogue<T>(acc, st, (wave & 2 ? &s2[0][0] : &s1[0][0]) + (wave & 1) * ((T / 2) * (T / 2)), r1_0, r2_0);      // (the operand images are free after the loop's last barrier)
}

// ---- the same tile product on the bf16 matrix pipe (matrix path "split", spconv_split.cuh has the arithmetic) ----------------------
// Every fp32 operand value is taken apart into three bf16 planes WHEN IT IS STAGED (5.5 vector instructions per value once per
// workgroup, not once per use), six v_mfma_f32_16x16x32_bf16 per 16 x 16 x 32 block, the three 2^-16 terms into their own accumulator
// set (a contraction over thousands of pixels makes the accumulator large against one instruction's contribution and every matrix
// instruction rounds it; see wgrad_split_kernel).  A stage = 32 contraction values = ONE matrix-instruction depth.
// LDS images per plane:  [row][32 bf16] for an operand whose contraction index is contiguous in memory (row stride 80 bytes: the
// 16 rows of a fragment read hit 16 different 16-byte bank groups), one ds_read_b128 per fragment and plane;  [16 k-pairs][row] of
// packed (k, k + 1) dwords for a COL operand - a thread holds one contraction index of 8 rows, the thread 8 lanes on holds the next
// index of the same rows: one DPP row rotation hands each of them the other's half, both pack 4 rows x (k, k + 1) and store one
// 16-byte line; fragment = four conflict-free ds_read_b32 per plane.  Both layouts give lane group g the contraction indices
// 8 g .. 8 g + 7 in order.
// 128 x 128 tiles only, 512 threads: the three planes of both operands, double buffered, are 100-123 KiB of LDS = ONE workgroup per
// CU; with 256 threads that is one wave per SIMD and a wave's fetch -> split -> store -> barrier -> matrix work runs with nothing
// beside it (measured: no faster than the fp32 form).  Eight waves share the images (2 per SIMD): wave (wm, wn) owns 64 x 32 of the tile.
constexpr int PGS_BLOCK = 512;
template <class Op1, class Op2, class Store>
__device__ __forceinline__ void pg_tile_split(const Op1& o1, const Op2& o2, const Store& st, int c_begin, int c_end) {
    static_assert(PG_K == 32, "one bf16 matrix-instruction depth per stage");
    constexpr int T = 128, TA = 4, TB = 2;      // 16 x 16 result tiles per wave: 64 operand-1 rows x 32 operand-2 rows
    constexpr int RS = 20;                      // dwords per row of a [row][32 bf16] plane (16 + 4 of padding)
    constexpr int LC2 = T + 4;                  // dwords per k-pair line of a [16 k-pairs][row] plane
    constexpr int P1 = Op1::COL ? 16 * LC2 : T * RS, P2 = Op2::COL ? 16 * LC2 : T * RS;      // dwords per plane
    constexpr int IMG = 2 * 3 * (P1 + P2), EPI = 8 * 64 * 32;
    __shared__ __attribute__((aligned(16))) unsigned lds[IMG > EPI ? IMG : EPI];
    unsigned* const s1 = lds;                   // [buffer][plane][P1]
    unsigned* const s2 = lds + 2 * 3 * P1;
    typedef unsigned pu4 __attribute__((ext_vector_type(4)));
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int r1_0 = blockIdx.y * T, r2_0 = blockIdx.x * T;
    const int wm = wave >> 2, wn = wave & 3;
    const int fi = lane & 15, fg = lane >> 4;
    // a thread's 8 values of a stage.  contraction-contiguous operand: row t >> 2, contraction values 8 (t & 3) .. + 7.
    // COL operand: contraction index kc, rows 8 rg .. + 7, with kc ^ 1 on lane ^ 8 (same 16-lane DPP row).
    const int rg = (t & 7) + 8 * ((t >> 4) & 1), kc = 2 * (t >> 5) + ((t >> 3) & 1);
    const bool odd = (t >> 3) & 1;

    typename Op1::State q1;
    typename Op2::State q2;
    if (Op1::COL) o1.init(q1, r1_0 + 8 * rg);
    else o1.init(q1, r1_0 + (t >> 2));
    if (Op2::COL) o2.init(q2, r2_0 + 8 * rg);
    else o2.init(q2, r2_0 + (t >> 2));
    float v1[8], v2[8];
    typename Op1::Cur k1;
    typename Op2::Cur k2;
    if constexpr (Op1::CURSOR) o1.cur_set(k1, c_begin);
    if constexpr (Op2::CURSOR) o2.cur_set(k2, c_begin);
    auto fetch = [&](int c0) {
        const int tn = (t & 3) * 8;
        if constexpr (Op1::COL) o1.load(q1, c0 + kc, v1);
        else if constexpr (Op1::CURSOR) o1.load(q1, k1, tn, v1);
        else o1.load(q1, c0 + tn, v1);
        if constexpr (Op2::COL) o2.load(q2, c0 + kc, v2);
        else if constexpr (Op2::CURSOR) o2.load(q2, k2, tn, v2);
        else o2.load(q2, c0 + tn, v2);
    };
    // split + store of a thread's share of a stage; img = plane 0 of the buffer, PSZ = dwords per plane
    auto stash_one = [&](unsigned* img, const int PSZ, const bool col, const float (&v)[8]) {
        pu4 ph, pm, pl;
        unsigned* dst;
        if (!col) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                unsigned a, b, c;
                sp_split2(v[2 * q], v[2 * q + 1], a, b, c);
                ph[q] = a, pm[q] = b, pl[q] = c;
            }
            dst = img + (t >> 2) * RS + (t & 3) * 4;
        } else {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float send = odd ? v[r] : v[4 + r], keep = odd ? v[4 + r] : v[r];
                const float got = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, send), 0x128 /* row_ror:8 */, 0xf, 0xf, false));
                unsigned a, b, c;
                sp_split2(odd ? got : keep, odd ? keep : got, a, b, c);      // (even index, odd index) of row 8 rg + 4 odd + r
                ph[r] = a, pm[r] = b, pl[r] = c;
            }
            dst = img + (t >> 5) * LC2 + 8 * rg + 4 * (int)odd;
        }
        *reinterpret_cast<pu4*>(dst) = ph;
        *reinterpret_cast<pu4*>(dst + PSZ) = pm;
        *reinterpret_cast<pu4*>(dst + 2 * PSZ) = pl;
    };

    pg4 acc[TA][TB], acc_s[TA][TB];
#pragma unroll
    for (int a = 0; a < TA; ++a)
#pragma unroll
        for (int b = 0; b < TB; ++b) acc[a][b] = acc_s[a][b] = pg4{0.f, 0.f, 0.f, 0.f};

    fetch(c_begin);
    stash_one(s1, P1, Op1::COL, v1);
    stash_one(s2, P2, Op2::COL, v2);
    __syncthreads();
    int buf = 0;
    for (int c0 = c_begin; c0 < c_end; c0 += PG_K) {
        const bool more = c0 + PG_K < c_end;
        if (more) {
            if constexpr (Op1::CURSOR) o1.cur_step(k1);
            if constexpr (Op2::CURSOR) o2.cur_step(k2);
            fetch(c0 + PG_K);
        }
        const unsigned* b1 = s1 + buf * 3 * P1;
        const unsigned* b2 = s2 + buf * 3 * P2;
        pu4 fa[TA][3];
#pragma unroll
        for (int a = 0; a < TA; ++a)
#pragma unroll
            for (int p = 0; p < 3; ++p) {
                if (Op1::COL) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) fa[a][p][j] = b1[p * P1 + (4 * fg + j) * LC2 + wm * 64 + a * 16 + fi];
                } else {
                    fa[a][p] = *reinterpret_cast<const pu4*>(&b1[p * P1 + (wm * 64 + a * 16 + fi) * RS + 4 * fg]);
                }
            }
#pragma unroll
        for (int b = 0; b < TB; ++b) {
            pu4 fb[3];
#pragma unroll
            for (int p = 0; p < 3; ++p) {
                if (Op2::COL) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) fb[p][j] = b2[p * P2 + (4 * fg + j) * LC2 + wn * 32 + b * 16 + fi];
                } else {
                    fb[p] = *reinterpret_cast<const pu4*>(&b2[p * P2 + (wn * 32 + b * 16 + fi) * RS + 4 * fg]);
                }
            }
#define PG_TERM(ACC, PA, PB)                                                                                                        \
    _Pragma("unroll") for (int a = 0; a < TA; ++a) ACC[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(                            \
        __builtin_bit_cast(bf16x8, fa[a][PA]), __builtin_bit_cast(bf16x8, fb[PB]), ACC[a][b], 0, 0, 0)
            PG_TERM(acc_s, 2, 0);      // lo . hi
            PG_TERM(acc_s, 0, 2);      // hi . lo
            PG_TERM(acc_s, 1, 1);      // mid . mid
            PG_TERM(acc, 1, 0);        // mid . hi
            PG_TERM(acc, 0, 1);        // hi . mid
            PG_TERM(acc, 0, 0);        // hi . hi
#undef PG_TERM
        }
        if (more) {
            stash_one(s1 + (buf ^ 1) * 3 * P1, P1, Op1::COL, v1);
            stash_one(s2 + (buf ^ 1) * 3 * P2, P2, Op2::COL, v2);
        }
        __syncthreads();
        buf ^= 1;
    }
#pragma unroll
    for (int a = 0; a < TA; ++a)
#pragma unroll
        for (int b = 0; b < TB; ++b) acc[a][b] = acc_s[a][b] + acc[a][b];
    // D: column = lane & 15 (operand-2 row), row = 4 * (lane >> 4) + reg (operand-1 row); the wave's block is 64 x 32
    if constexpr (Store::ROWS4) {
        if (st.vec_ok()) {
            float* img = reinterpret_cast<float*>(lds) + wave * (64 * 32);      // (the images are free after the loop's last barrier)
#pragma unroll
            for (int a = 0; a < TA; ++a)
#pragma unroll
                for (int b = 0; b < TB; ++b)
#pragma unroll
                    for (int reg = 0; reg < 4; ++reg) img[(a * 16 + 4 * fg + reg) * 32 + b * 16 + fi] = acc[a][b][reg];
            __builtin_amdgcn_wave_barrier();
            const int c4 = (lane & 7) * 4;
            const long long base = st.prep(r2_0 + wn * 32 + c4);
#pragma unroll
            for (int it = 0; it < 8; ++it) {
                const int row = it * 8 + (lane >> 3);
                st.put4(base, r1_0 + wm * 64 + row, *reinterpret_cast<const pg4*>(&img[row * 32 + c4]));
            }
            return;
        }
    }
#pragma unroll
    for (int b = 0; b < TB; ++b) {
        const long long base = st.prep(r2_0 + wn * 32 + b * 16 + fi);
#pragma unroll
        for (int a = 0; a < TA; ++a)
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) st.put(base, r1_0 + wm * 64 + a * 16 + 4 * fg + reg, acc[a][b][reg]);
    }
}

template <class Op1, class Op2, class Store>
__global__ void __launch_bounds__(PGS_BLOCK)
pg_gemm_split_kernel(const Op1 o1, const Op2 o2, const Store st, int c_total, int c_per_split) {
    const int c_begin = blockIdx.z * c_per_split;
    const int c_end = min(c_total, c_begin + c_per_split);
    pg_tile_split(o1, o2, st, c_begin, c_end > c_begin ? c_end : c_begin);      // (an empty split still owns a slab: it stores zeros)
}

__global__ void __launch_bounds__(PGS_BLOCK)
pg_s2_dgrad_split_kernel(WS2Dgrad o1, PixS2Dgrad o2, StoreS2Class st) {
    S2Class c;
    c.set(blockIdx.z);
    o1.c = c;
    o2.c = c;
    st.py = c.py, st.px = c.px;
    pg_tile_split(o1, o2, st, 0, o2.Cout << (c.lty + c.ltx));
}

template <int T, class Op1, class Op2, class Store>
__global__ void __launch_bounds__(PG_BLOCK)
pg_gemm_kernel(const Op1 o1, const Op2 o2, const Store st, int c_total, int c_per_split) {
    const int c_begin = blockIdx.z * c_per_split;
    const int c_end = min(c_total, c_begin + c_per_split);
    pg_tile<T>(o1, o2, st, c_begin, c_end > c_begin ? c_end : c_begin);      // (an empty split still owns a slab: it stores zeros)
}

// stride-2 data gradient: blockIdx.z = parity class
template <int T>
__global__ void __launch_bounds__(PG_BLOCK)
pg_s2_dgrad_kernel(WS2Dgrad o1, PixS2Dgrad o2, StoreS2Class st) {
    S2Class c;
    c.set(blockIdx.z);
    o1.c = c;
    o2.c = c;
    st.py = c.py, st.px = c.px;
    pg_tile<T>(o1, o2, st, 0, o2.Cout << (c.lty + c.ltx));
}

__global__ void __launch_bounds__(PG_BLOCK)
pg_slab_reduce_kernel(const float* __restrict__ slab, int splits, long long elems, float* __restrict__ out) {
    const long long e = (long long)blockIdx.x * PG_BLOCK + threadIdx.x;
    if (e >= elems) return;
    float s = 0.f;
    int z = 0;
    for (; z + 8 <= splits; z += 8) {          // adds in split order; eight loads in flight
        float v[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = slab[(long long)(z + i) * elems + e];
#pragma unroll
        for (int i = 0; i < 8; ++i) s += v[i];
    }
    for (; z < splits; ++z) s += slab[(long long)z * elems + e];
    out[e] = s;
}

static int pg_splits(long long n_pixels, int tiles) {
    // weight gradients: the output is a few tiles, the parallelism comes from splitting the pixels - about 512 workgroups = the
    // 256 CUs x 2 resident slots filled once (equal-length chunks: 768 workgroups were two rounds for one and a half rounds of work -
    // stride-2 conv wgrad 182 -> 167 us, the deblocks' 82 / 156 -> 81 / 156; TODA_PG_WG_TARGET), contraction chunks of at least 8 stages
    static const int target = getenv("TODA_PG_WG_TARGET") ? atoi(getenv("TODA_PG_WG_TARGET")) : 512;
    int want = cdiv(target, tiles > 0 ? tiles : 1);
    long long max_by_len = n_pixels / (8 * PG_K);
    if (max_by_len < 1) max_by_len = 1;
    if (want > max_by_len) want = (int)max_by_len;
    if (want > 512) want = 512;
    if (want < 1) want = 1;
    return want;
}

// 128-row tiles when they fill the chip (or when contraction splits do: gz > 1, the weight gradients), 64-row tiles otherwise
static inline bool pg_small(long long r1, long long r2, int gz) { return gz == 1 && (long long)cdiv(r1, 128) * cdiv(r2, 128) < 768; }
// matrix path "split" (toda_set_matrix_path): the same accessors and stores around pg_tile_split - 128 x 128 tiles, one 512-thread
// workgroup per CU.  Also for the grids that fill the 256 CUs badly at that tile size (278 tiles: the stride-2 forward 149 us against 157
// on the fp32 form, the 2 x 2 deblock's data gradient 125 against 129); 128 x 64 tiles for those measured slower (156 / 132).
static inline bool pg_split_grid(long long r1, long long r2, int gz) {
    if (toda_matrix_path() != 1) return false;
    static const int env = getenv("TODA_PG_SPLIT") ? atoi(getenv("TODA_PG_SPLIT")) : 1;
    (void)r1, (void)r2, (void)gz;
    return env != 0;
}
#define PG_LAUNCH(O1, O2, ST, r1, r2, gz, ...)                                                                                             \
    do {                                                                                                                                   \
        if (pg_split_grid(r1, r2, gz))                                                                                                      \
            hipLaunchKernelGGL(HIP_KERNEL_NAME(pg_gemm_split_kernel<O1, O2, ST>), dim3(cdiv(r2, 128), cdiv(r1, 128), gz), dim3(PGS_BLOCK), 0,  \
                               (hipStream_t)stream, __VA_ARGS__);                                                                          \
        else if (pg_small(r1, r2, gz))                                                                                                       \
            hipLaunchKernelGGL(HIP_KERNEL_NAME(pg_gemm_kernel<64, O1, O2, ST>), dim3(cdiv(r2, 64), cdiv(r1, 64), gz), dim3(PG_BLOCK), 0,    \
                               (hipStream_t)stream, __VA_ARGS__);                                                                          \
        else                                                                                                                               \
            hipLaunchKernelGGL(HIP_KERNEL_NAME(pg_gemm_kernel<128, O1, O2, ST>), dim3(cdiv(r2, 128), cdiv(r1, 128), gz), dim3(PG_BLOCK), 0, \
                               (hipStream_t)stream, __VA_ARGS__);                                                                          \
    } while (0)

static int pg_check(const char* who, int B, int Cin, int Cout, int H, int W) {
    TODA_CHECK_ARG(B >= 1 && Cin >= 1 && Cout >= 1 && H >= 1 && W >= 1, "%s: bad shape", who);
    TODA_CHECK_ARG(4LL * B * (long long)(Cin > Cout ? Cin : Cout) * H * W * 4 < (1LL << 40), "%s: map too large", who);
    return TODA_OK;
}

}  // namespace toda

using namespace toda;

// ---------------------------------------------------------------- ZeroPad2d(1) + Conv2d(3, stride 2): base_bev_backbone.py:32-36
extern "C" int toda_conv3x3s2_supported(int batch, int cin, int cout, int H, int W) {
    return (batch >= 1 && cin >= 1 && cout >= 1 && H >= 2 && W >= 2 && (H % 2) == 0 && (W % 2) == 0) ? 1 : 0;
}

extern "C" int toda_conv3x3s2_fwd(const float* x, const float* w, int batch, int cin, int cout, int H, int W, float* y, void* stream) {
    TODA_CHECK_ARG(x && w && y && toda_conv3x3s2_supported(batch, cin, cout, H, W), "conv3x3s2_fwd: null pointer or odd map");
    if (int rc = pg_check("conv3x3s2_fwd", batch, cin, cout, H, W)) return rc;
    const int Ho = H / 2, Wo = W / 2, N = batch * Ho * Wo, K = cin * 9;
    MatOp<false> o1{w, cout, K, K, 1};
    PixConvS2 o2{ConvS2Geom{x, batch, cin, H, W, Ho, Wo}};
    StoreNCHW st{y, cout, N, Ho * Wo};
    PG_LAUNCH(MatOp<false>, PixConvS2, StoreNCHW, cout, N, 1, o1, o2, st, K, K);
    TODA_LAUNCH_CHECK();
    return TODA_OK;
}

extern "C" int toda_conv3x3s2_dgrad(const float* dy, const float* w, int batch, int cin, int cout, int H, int W, float* dx, void* stream) {
    TODA_CHECK_ARG(dy && w && dx && toda_conv3x3s2_supported(batch, cin, cout, H, W), "conv3x3s2_dgrad: null pointer or odd map");
    if (int rc = pg_check("conv3x3s2_dgrad", batch, cin, cout, H, W)) return rc;
    const int N = batch * (H / 2) * (W / 2);
    WS2Dgrad o1{w, cin, cout, S2Class{}};
    PixS2Dgrad o2{dy, batch, cout, H, W, H / 2, W / 2, S2Class{}};
    StoreS2Class st{dx, cin, N, H, W, 0, 0};
    if (pg_split_grid(cin, N, 4))
        hipLaunchKernelGGL(pg_s2_dgrad_split_kernel, dim3(cdiv(N, 128), cdiv(cin, 128), 4), dim3(PGS_BLOCK), 0, (hipStream_t)stream, o1, o2, st);
    else if ((long long)cdiv(cin, 128) * cdiv(N, 128) * 4 < 768)
        hipLaunchKernelGGL(pg_s2_dgrad_kernel<64>, dim3(cdiv(N, 64), cdiv(cin, 64), 4), dim3(PG_BLOCK), 0, (hipStream_t)stream, o1, o2, st);
    else
        hipLaunchKernelGGL(pg_s2_dgrad_kernel<128>, dim3(cdiv(N, 128), cdiv(cin, 128), 4), dim3(PG_BLOCK), 0, (hipStream_t)stream, o1, o2, st);
    TODA_LAUNCH_CHECK();
    return TODA_OK;
}

extern "C" size_t toda_conv3x3s2_wgrad_workspace_bytes(int batch, int cin, int cout, int H, int W) {
    const long long n = (long long)batch * (H / 2) * (W / 2);
    const int splits = pg_splits(n, cdiv(cin * 9, 128) * cdiv(cout, 128));
    return align_up((size_t)splits * cout * cin * 9 * sizeof(float), 256);
}

extern "C" int toda_conv3x3s2_wgrad(const float* x, const float* dy, int batch, int cin, int cout, int H, int W, float* dw, void* ws, size_t ws_bytes,
                                    void* stream) {
    TODA_CHECK_ARG(x && dy && dw && ws && toda_conv3x3s2_supported(batch, cin, cout, H, W), "conv3x3s2_wgrad: null pointer or odd map");
    if (int rc = pg_check("conv3x3s2_wgrad", batch, cin, cout, H, W)) return rc;
    const int Ho = H / 2, Wo = W / 2, N = batch * Ho * Wo, K = cin * 9;
    const int splits = pg_splits(N, cdiv(K, 128) * cdiv(cout, 128));
    if (ws_bytes < toda_conv3x3s2_wgrad_workspace_bytes(batch, cin, cout, H, W)) {
        set_error("conv3x3s2_wgrad: workspace too small");
        return TODA_EWORKSPACE;
    }
    int per = cdiv(cdiv(N, splits), PG_K) * PG_K;
    StoreSlab st{(float*)ws, cout, K};
    const bool small_map = (long long)batch * (cin > cout ? cin : cout) * H * W < (1LL << 31);      // 32-bit element offsets in the FAST accessors
    if (small_map && ChanPlain<true>::fast_ok(Ho * Wo) && TapConvS2<true>::fast_ok(Wo)) {
        ChanPlain<true> o1{dy, batch, cout, Ho * Wo};                       // rows co, contraction pixels
        TapConvS2<true> o2{ConvS2Geom{x, batch, cin, H, W, Ho, Wo}, K};   // rows (ci, ky, kx)
        PG_LAUNCH(ChanPlain<true>, TapConvS2<true>, StoreSlab, cout, K, splits, o1, o2, st, N, per);
    } else {
        ChanPlain<false> o1{dy, batch, cout, Ho * Wo};
        TapConvS2<false> o2{ConvS2Geom{x, batch, cin, H, W, Ho, Wo}, K};
        PG_LAUNCH(ChanPlain<false>, TapConvS2<false>, StoreSlab, cout, K, splits, o1, o2, st, N, per);
    }
    const long long elems = (long long)cout * K;
    hipLaunchKernelGGL(pg_slab_reduce_kernel, dim3(cdiv(elems, PG_BLOCK)), dim3(PG_BLOCK), 0, (hipStream_t)stream, (const float*)ws, splits, elems, dw);
    TODA_LAUNCH_CHECK();
    return TODA_OK;
}

// --------------------------------------------- ConvTranspose2d(Cin, Cout, k = s, stride = s), s in {1, 2}: base_bev_backbone.py:47-66
// weight [Cin][Cout][s][s]
extern "C" int toda_deconv_supported(int batch, int cin, int cout, int H, int W, int s) {
    return (batch >= 1 && cin >= 1 && cout >= 1 && H >= 1 && W >= 1 && (s == 1 || s == 2)) ? 1 : 0;
}

extern "C" int toda_deconv_fwd(const float* x, const float* w, int batch, int cin, int cout, int H, int W, int s, float* y, void* stream) {
    TODA_CHECK_ARG(x && w && y && toda_deconv_supported(batch, cin, cout, H, W, s), "deconv_fwd: null pointer or stride not in {1, 2}");
    if (int rc = pg_check("deconv_fwd", batch, cin, cout, H * s, W * s)) return rc;
    const int N = batch * H * W, M = cout * s * s;
    MatOp<true> o1{w, M, cin, 1, M};          // element (m, k) = w[k][m]: rows contiguous
    PixPlain o2{x, batch, cin, H * W};
    if (s == 1) {
        StoreNCHW st{y, cout, N, H * W};
        PG_LAUNCH(MatOp<true>, PixPlain, StoreNCHW, M, N, 1, o1, o2, st, cin, cin);
    } else {
        StoreShuffle st{y, M, N, cout, H, W};
        PG_LAUNCH(MatOp<true>, PixPlain, StoreShuffle, M, N, 1, o1, o2, st, cin, cin);
    }
    TODA_LAUNCH_CHECK();
    return TODA_OK;
}

extern "C" int toda_deconv_dgrad(const float* dy, const float* w, int batch, int cin, int cout, int H, int W, int s, float* dx, void* stream) {
    TODA_CHECK_ARG(dy && w && dx && toda_deconv_supported(batch, cin, cout, H, W, s), "deconv_dgrad: null pointer or stride not in {1, 2}");
    if (int rc = pg_check("deconv_dgrad", batch, cin, cout, H * s, W * s)) return rc;
    const int N = batch * H * W, K = cout * s * s;
    MatOp<false> o1{w, cin, K, K, 1};         // element (ci, k) = w[ci][k]
    StoreNCHW st{dx, cin, N, H * W};
    if (s == 1) {
        PixPlain o2{dy, batch, cout, H * W};
        PG_LAUNCH(MatOp<false>, PixPlain, StoreNCHW, cin, N, 1, o1, o2, st, K, K);
    } else {
        PixUnshuffle o2{UnshuffleGeom{dy, batch, cout, H, W}};
        PG_LAUNCH(MatOp<false>, PixUnshuffle, StoreNCHW, cin, N, 1, o1, o2, st, K, K);
    }
    TODA_LAUNCH_CHECK();
    return TODA_OK;
}

extern "C" size_t toda_deconv_wgrad_workspace_bytes(int batch, int cin, int cout, int H, int W, int s) {
    const long long n = (long long)batch * H * W;
    const int K = cout * s * s;
    const int splits = pg_splits(n, cdiv(K, 128) * cdiv(cin, 128));
    return align_up((size_t)splits * cin * K * sizeof(float), 256);
}

extern "C" int toda_deconv_wgrad(const float* x, const float* dy, int batch, int cin, int cout, int H, int W, int s, float* dw, void* ws, size_t ws_bytes,
                                 void* stream) {
    TODA_CHECK_ARG(x && dy && dw && ws && toda_deconv_supported(batch, cin, cout, H, W, s), "deconv_wgrad: null pointer or stride not in {1, 2}");
    if (int rc = pg_check("deconv_wgrad", batch, cin, cout, H * s, W * s)) return rc;
    const int N = batch * H * W, K = cout * s * s;
    const int splits = pg_splits(N, cdiv(K, 128) * cdiv(cin, 128));
    if (ws_bytes < toda_deconv_wgrad_workspace_bytes(batch, cin, cout, H, W, s)) {
        set_error("deconv_wgrad: workspace too small");
        return TODA_EWORKSPACE;
    }
    int per = cdiv(cdiv(N, splits), PG_K) * PG_K;
    StoreSlab st{(float*)ws, cin, K};
    const bool small_map = (long long)batch * (cin > cout ? cin : cout) * H * s * W * s < (1LL << 31);      // 32-bit element offsets in the FAST accessors
    const bool fast = small_map && ChanPlain<true>::fast_ok(H * W) && (s == 1 || ChanUnshuffle<true>::fast_ok(W));
    if (s == 1 && fast) {
        ChanPlain<true> o1{x, batch, cin, H * W}, o2{dy, batch, cout, H * W};                           // rows ci / co, contraction pixels
        PG_LAUNCH(ChanPlain<true>, ChanPlain<true>, StoreSlab, cin, K, splits, o1, o2, st, N, per);
    } else if (s == 1) {
        ChanPlain<false> o1{x, batch, cin, H * W}, o2{dy, batch, cout, H * W};
        PG_LAUNCH(ChanPlain<false>, ChanPlain<false>, StoreSlab, cin, K, splits, o1, o2, st, N, per);
    } else if (fast) {
        ChanPlain<true> o1{x, batch, cin, H * W};
        ChanUnshuffle<true> o2{UnshuffleGeom{dy, batch, cout, H, W}, K};
        PG_LAUNCH(ChanPlain<true>, ChanUnshuffle<true>, StoreSlab, cin, K, splits, o1, o2, st, N, per);
    } else {
        ChanPlain<false> o1{x, batch, cin, H * W};
        ChanUnshuffle<false> o2{UnshuffleGeom{dy, batch, cout, H, W}, K};
        PG_LAUNCH(ChanPlain<false>, ChanUnshuffle<false>, StoreSlab, cin, K, splits, o1, o2, st, N, per);
    }
    const long long elems = (long long)cin * K;
    hipLaunchKernelGGL(pg_slab_reduce_kernel, dim3(cdiv(elems, PG_BLOCK)), dim3(PG_BLOCK), 0, (hipStream_t)stream, (const float*)ws, splits, elems, dw);
    TODA_LAUNCH_CHECK();
    return TODA_OK;
}
