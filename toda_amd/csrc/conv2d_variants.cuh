// Opt-in: the REGISTER form of the Winograd weight gradient (round 4; measured, not taken - DESIGN.md section 7).  Included by conv2d.hip
// inside namespace toda when the library is built with `make VARIANTS=1`; selected at run time with TODA_WINO_WGRAD=2.
#pragma once

// ------------------------------------------------------------------------------------------------------------------
// Weight gradient, REGISTER form (round 4).  The 16 x 16 x 4 MFMA wants, in lane (i, k) = (lane & 15, lane >> 4), A[i][k] and
// B[k][i].  With i = a channel and k = one of four consecutive tiles that is exactly what ONE lane has after transforming ONE
// patch: lane (ci, tile) holds V[f][tile][ci] for all 36 frequencies, lane (co, tile) holds dM[f][tile][co].  So nothing is
// staged: no LDS image, no producer / consumer waves, no barrier.  A wave owns 16 input x 32 output channels (72 accumulators of
// 4 registers: the AGPR file and a little more), a workgroup of four waves 32 x 64 - the waves of a pair read the same patches
// (second reader: L1) -, one wave per SIMD.  Per step of four tiles a lane issues 12 + 8 loads (a 6 x 6 patch as 16 + 8 bytes
// per row from x0 - 1, two 4 x 4 tiles of dY), ~400 vector instructions of transform and 72 MFMAs.
// What the staged kernel above paid (ablation, profiles/r04_wino_wgrad_experiments.txt): with loads, transforms and MFMAs all
// removed it still took 58 % of its time - 72 ds_write_b32 per producer lane and 72 ds_read_b64 per consumer lane and chunk.
// Stream-K over (unit = (32-ci block, 64-co block), step); segments dump their accumulators as they stand (1 KiB per store
// instruction), wino_wgrad_reduce_kernel<W2_SLAB_FLOATS> folds them in workgroup order, wino_wgrad_reg_finish_kernel applies
// G^T . G.  No float atomics; same bits every run.
// ------------------------------------------------------------------------------------------------------------------
#ifndef W2_ALT
#define W2_ALT 0
#endif
#ifndef W2_HALO
#define W2_HALO 1      // 0: measurement only (the in-loop patches come without their halo columns: wrong numbers)
#endif
constexpr int W2_BLOCK = 256;
#ifndef W2_TSPLIT
#define W2_TSPLIT 1
#endif
#if W2_TSPLIT
// a unit = ONE wave tile (16 input x 32 output channels); the four waves of a workgroup take four quarters of the workgroup's run of
// steps and add their accumulators up through LDS before ONE of them dumps: a quarter of the slab bytes (written, folded) of the other
// arrangement, paid with patches that no longer have a second reader in the same L1
constexpr int W2_CI = 16, W2_CO = 32;
#else
constexpr int W2_CI = 32, W2_CO = 64;                      // a unit = 2 x 2 wave tiles over the same steps
#endif
constexpr int W2_SLAB_FLOATS = WN_FREQ * W2_CI * W2_CO;   // ([wave 4])[co half 2][f 36][lane 64][4]
constexpr unsigned W2_OOB = 0x80000000u;                  // operands stay below 2 GiB (toda_conv3x3_wgrad checks): offset + 16 cannot wrap

struct Wg2Geom {
    WinoGeom g;
    int n_cib, n_cob, n_units, steps_per_unit;
};

// one row of a patch: p = columns (x0, x0+1), q = (x0+2, x0+3), e = (x0-1, x0+4); a patch = six of them (by pointer: the pipeline
// keeps a second copy of the last W2_ALT rows and swaps the two from step to step)
struct X2Row {
    f32x2 p, q, e;
};
struct X2Patch {
    X2Row* r[6];
};
struct Y2Raw {
    f32x2 a[4], b[4];     // columns (x0, x0+1), (x0+2, x0+3) of the four rows
};

// branch-free step to the tile four places on (tiles_x >= 4)
__device__ __forceinline__ void w2_advance(TilePos& t, int tile_after, const WinoGeom& g) {
    t.tx += 4;
    const bool wx = t.tx >= g.tiles_x;
    t.tx -= wx ? g.tiles_x : 0;
    t.ty += wx ? 1 : 0;
    const bool wy = t.ty >= g.tiles_y;
    t.ty = wy ? 0 : t.ty;
    t.b += wy ? 1 : 0;
    t.exists = tile_after < g.n_tiles;
}

// Addresses of a patch: per-lane byte offsets of (row y0, column x0) for the row classes (above the image / rows y0, y0+1 / rows
// y0+2, y0+3 / row y0+4; H is even), the row itself in the wave-uniform scalar offset - one multiply chain per patch, not per row.
struct X2Off {
    unsigned vtop, vall, vmid, vbot;
};

__device__ __forceinline__ X2Off w2_x_off(const TilePos& t, int chan, const WinoGeom& g) {
    const int y0 = 4 * t.ty;
    const unsigned base = (unsigned)(((t.b * g.Cin + chan) * g.H + y0) * g.W + 4 * t.tx) * 4u;
    X2Off o;
    o.vtop = (t.exists && t.ty > 0) ? base - (unsigned)g.W * 4u : W2_OOB;
    o.vall = t.exists ? base : W2_OOB;
    o.vmid = (t.exists && y0 + 2 < g.H) ? base : W2_OOB;
    o.vbot = (t.exists && y0 + 4 < g.H) ? base : W2_OOB;
    return o;
}

// Row r of the patch: 16 bytes from x0 and the two halo columns.  The left one is read at offset - 4: an out-of-range offset stays out
// of range (0x7FFFFFFC), the offset of column 0 wraps to 0xFFFFFFFC (out of range as well; that column is padding).  Halo columns
// outside the image are multiplied away after the column pass (w2_x_mask).
__device__ __forceinline__ void w2_load_row(__amdgpu_buffer_rsrc_t rsrc, const X2Off& o, int w4, int r, const X2Patch& pt, bool halo = true) {
    const unsigned v = r == 0 ? o.vtop : r < 3 ? o.vall : r < 5 ? o.vmid : o.vbot;
    const int so = r == 0 ? 0 : (r - 1) * w4;
    const f32x4 c = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, v, so, 0));
    const float l = halo ? __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc, v - 4u, so, 0)) : 0.0f;
    const float rr = halo ? __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc, v + 16u, so, 0)) : 0.0f;
    X2Row& d = *pt.r[r];
    d.p = f32x2{c[0], c[1]};
    d.q = f32x2{c[2], c[3]};
    d.e = f32x2{l, rr};
}

// both dY tiles of a lane (output channels co and co + 16: the second one through the scalar offset)
__device__ __forceinline__ void w2_y_off(const TilePos& t, int chan, const WinoGeom& g, unsigned& va, unsigned& vb) {
    const int y0 = 4 * t.ty;
    const unsigned base = (unsigned)(((t.b * g.Cout + chan) * g.H + y0) * g.W + 4 * t.tx) * 4u;
    va = t.exists ? base : W2_OOB;
    vb = (t.exists && y0 + 2 < g.H) ? base : W2_OOB;
}

__device__ __forceinline__ void w2_load_y(__amdgpu_buffer_rsrc_t rsrc, unsigned va, unsigned vb, int w4, int chan_off, Y2Raw& d) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const f32x4 c = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, i < 2 ? va : vb, chan_off + i * w4, 0));
        d.a[i] = f32x2{c[0], c[1]};
        d.b[i] = f32x2{c[2], c[3]};
    }
}

// what of a tile's columns lies inside the image, as factors: W is even, so x0+2 and x0+3 exist together
struct X2Mask {
    f32x2 e;      // (x0 - 1 inside, x0 + 4 inside)
    f32x2 q;      // (x0 + 2 inside) twice
};
__device__ __forceinline__ X2Mask w2_mask(const TilePos& t, const WinoGeom& g) {
    const int nv = g.W - 4 * t.tx;
    X2Mask m;
    m.e = f32x2{t.tx > 0 ? 1.0f : 0.0f, nv > 4 ? 1.0f : 0.0f};
    const float q = nv > 2 ? 1.0f : 0.0f;
    m.q = f32x2{q, q};
    return m;
}

// One column pair of the column pass of B^T d B IN PLACE (which: 0 = p, 1 = q, 2 = e), masked on the way out
__device__ __forceinline__ void w2_x_cols(const X2Patch& pt, int which, const X2Mask& mk) {
    f32x2 d[6], t[6];
#pragma unroll
    for (int r = 0; r < 6; ++r) d[r] = which == 0 ? pt.r[r]->p : which == 1 ? pt.r[r]->q : pt.r[r]->e;
    w2_bt2(d, t);
#pragma unroll
    for (int r = 0; r < 6; ++r) {
        if (which == 0) pt.r[r]->p = t[r];
        if (which == 1) pt.r[r]->q = w2_mul(t[r], mk.q);
        if (which == 2) pt.r[r]->e = w2_mul(t[r], mk.e);
    }
}

// Row i of the row pass: with p = (d1, d2), q = (d3, d4), e = (d0, d5):
//   (a, b) = q - p;  (v3, v4) = (2, -.5) a + b;  (v1, v2) = (d4 + d1, d4 - d1) + (.5, 2.5) d3 + (-2.5, .5) d2;
//   (v0, v5) = 1.5 (a, b) + e + (d4, d1) - 2 (d2, d3)          (the last two terms cross the pairs: four single instructions)
// Written as the instructions themselves: the compiler splits a packed operation whose operands are half-broadcasts into two single ones.
__device__ __forceinline__ void w2_x_rowpass(const X2Patch& pt, int i, float* __restrict__ v) {
    const f32x2 p = pt.r[i]->p, q = pt.r[i]->q, e = pt.r[i]->e;
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

// dY tile -> dM = A dY A^T in two steps: the column pass m = A dY on column pairs (rows 0 and 5 of m ARE rows 0 and 3 of dY: the raw
// tile, masked in place, stays until the last row pass), then row i of dM = m[i] A^T (index 6 i + j)
struct Y2Mid {
    f32x2 a[4], b[4];     // rows 1 .. 4 of A dY
};
__device__ __forceinline__ void w2_y_cols(const f32x2 (&y)[4], f32x2 (&m)[4]) {
    const f32x2 e = w2_add(y[0], y[2]), o = w2_add(y[1], y[3]);
    m[0] = w2_sub(e, o);
    m[1] = w2_add(e, o);
    m[2] = w2_fmak(W2_KK(0.125f), y[3], w2_fmak(W2_KK(0.25f), y[2], w2_fmak(W2_KK(0.5f), y[1], y[0])));
    m[3] = w2_fmak(W2_KK(-8.0f), y[3], w2_fmak(W2_KK(4.0f), y[2], w2_fmak(W2_KK(-2.0f), y[1], y[0])));
}
__device__ __forceinline__ void w2_y_colpass(Y2Raw& d, const X2Mask& mk, Y2Mid& m) {
#pragma unroll
    for (int i = 0; i < 4; ++i) d.b[i] = w2_mul(d.b[i], mk.q);
    w2_y_cols(d.a, m.a);
    w2_y_cols(d.b, m.b);
}
// with P = (y0, y1), Q = (y2, y3): (e, o) = P + Q; (m1, m2) = e -+ o; (m3, m4) = y0 + (.5, -2) y1 + (.25, 4) y2 + (.125, -8) y3
__device__ __forceinline__ void w2_y_rowpass(const Y2Raw& d, const Y2Mid& m, int i, float* __restrict__ v) {
    const f32x2 P = i == 0 ? d.a[0] : i == 5 ? d.a[3] : m.a[i - 1], Q = i == 0 ? d.b[0] : i == 5 ? d.b[3] : m.b[i - 1];
    const f32x2 eo = w2_add(P, Q);
    f32x2 m12, u;
    asm("v_pk_add_f32 %0, %1, %1 op_sel:[0,1] op_sel_hi:[0,1] neg_lo:[0,1]" : "=v"(m12) : "v"(eo));
    asm("v_pk_fma_f32 %0, %1, %2, %1 op_sel:[1,0,0] op_sel_hi:[1,1,0]" : "=v"(u) : "v"(P), "s"(W2_K(0.5f, -2.0f)));
    asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,0,0] op_sel_hi:[0,1,1]" : "+v"(u) : "v"(Q), "s"(W2_K(0.25f, 4.0f)));
    asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,0,0] op_sel_hi:[1,1,1]" : "+v"(u) : "v"(Q), "s"(W2_K(0.125f, -8.0f)));
    v[0] = P[0];
    v[1] = m12[0];
    v[2] = m12[1];
    v[3] = u[0];
    v[4] = u[1];
    v[5] = Q[1];
}

// 72 accumulators of four registers: 64 fill the AGPR file, the last row of frequencies (f >= 32) lives in VGPRs - the compiler
// uses ONE register class for every MFMA builtin of a function (AGPR form here), so those eight are written as the instruction itself.
// Their results are read by the same instruction one step (72 MFMAs) later and by the dump, which waits (s_nop) first.
#define W2_MFMA(f_, a_, b_, acc_)                                                                                       \
    do {                                                                                                                \
        if ((f_) < 32) {                                                                                                \
            (acc_) = __builtin_amdgcn_mfma_f32_16x16x4f32((a_), (b_), (acc_), 0, 0, 0);                                 \
        } else {                                                                                                        \
            asm("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(acc_) : "v"(a_), "v"(b_));                               \
        }                                                                                                               \
    } while (0)

__global__ void __launch_bounds__(W2_BLOCK, 1)
wino_wgrad_reg_kernel(const float* __restrict__ x, const float* __restrict__ dy, const Wg2Geom wg, float* __restrict__ slabs, const int ablate) {
#if W2_TSPLIT
    __shared__ float red_lds[2 * W2_SLAB_FLOATS];          // 144 KiB: two waves' accumulators
#endif
    const WinoGeom& g = wg.g;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));     // (wave-uniform: it decides scalar things)
    const int wm = wave >> 1, wn = wave & 1, li = lane & 15, k = lane >> 4;
    (void)wm, (void)wn;
    const int G = gridDim.x;
    const int w = wn_work_id();
    const long long S = (long long)wg.n_units * wg.steps_per_unit;
    const long long lo = ws_range_lo(w, G, S), hi = ws_range_lo(w + 1, G, S);
    if (hi == lo) return;
    // The four waves of the workgroup run the same instruction stream on the four SIMDs and meet at the CU's one address unit with
    // every load (16 cycles per wave-instruction there): start them a fraction of a step apart and they stay apart.
    for (int i = 0, n = wave * (ablate >> 8); i < n; ++i) __builtin_amdgcn_s_sleep(1);
    const unsigned x_bytes = (ablate & 1) ? 0u : (unsigned)((size_t)g.B * g.Cin * g.H * g.W * 4u), y_bytes = (ablate & 2) ? 0u : (unsigned)((size_t)g.B * g.Cout * g.H * g.W * 4u);
    const __amdgpu_buffer_rsrc_t xr = wn_rsrc(x, x_bytes), yr = wn_rsrc(dy, y_bytes);

    long long s = lo;
    while (s < hi) {
        const int unit = (int)(s / wg.steps_per_unit);
        const int seg_begin = (int)(s - (long long)unit * wg.steps_per_unit);
        const int seg_end = (hi - s < wg.steps_per_unit - seg_begin) ? seg_begin + (int)(hi - s) : wg.steps_per_unit;
        const int cib = unit / wg.n_cob, cob = unit - cib * wg.n_cob;
#if W2_TSPLIT
        const int ci = cib * W2_CI + li, co = cob * W2_CO + li;
        const int seg_n = seg_end - seg_begin;
        const int c_begin = seg_begin + (wave * seg_n) / 4, c_end = seg_begin + ((wave + 1) * seg_n) / 4;      // this wave's quarter (may be empty)
#else
        const int ci = cib * W2_CI + wm * 16 + li, co = cob * W2_CO + wn * 32 + li;
        const int c_begin = seg_begin, c_end = seg_end;
#endif
        f32x4 acc0[WN_FREQ], acc1[WN_FREQ];
#pragma unroll
        for (int f = 0; f < WN_FREQ; ++f) acc0[f] = acc1[f] = f32x4{0.f, 0.f, 0.f, 0.f};

        // Rolling pipeline: ONE register block per operand, rewritten row by row (six frequencies) behind the MFMAs that read it.
        // At the top of step c: va = V(c), vb = dM0(c); ry1 = raw dY1(c); (xl, hc) = raw x(c+1) and ry0 = raw dY0(c+1), in flight.
        //   first half   rows of acc0 += va . vb, each followed by  vb row <- dM1(c) row;  at the end the column pass of x(c+1) and ry1 <- dY1(c+1)
        //   second half  rows of acc1 += va . vb, each followed by  vb row <- dM0(c+1) row, va row <- V(c+1) row;  x(c+2): rows 3-5 into
        //                the OTHER set of rows 3-5 at the start of the half, rows 0-2 as the row pass frees them;  ry0 <- dY0(c+2) at the end
        // Twelve scheduling regions of one frequency row each (sched_barrier) keep the compiler from computing a new row ahead of the
        // MFMAs that read the old one (it would need a second register block) and the loads where they are written.  The loop runs an
        // even number of steps: behind the end of the segment the loads see an empty descriptor (hardware zeros, no access) and the
        // phantom step adds zeros.
        if (c_begin < c_end) {          // (an empty quarter: the accumulators stay zero)
        float va[WN_FREQ], vb[WN_FREQ];
        Y2Mid m;
        X2Row xr0, xr1, xr2, xr3, xr4, xr5, xs3, xs4, xs5;          // rows 0 .. 5 and the second copy of rows 3 .. 5 (the last W2_ALT of them used)
        const X2Patch pa = {{&xr0, &xr1, &xr2, &xr3, &xr4, &xr5}};
        const X2Patch pb = {{&xr0, &xr1, &xr2, W2_ALT >= 3 ? &xs3 : &xr3, W2_ALT >= 2 ? &xs4 : &xr4, W2_ALT >= 1 ? &xs5 : &xr5}};
        Y2Raw ry0, ry1;
        const int w4 = g.W * 4, co16 = 16 * g.H * g.W * 4;
        TilePos tn = wn_tile_pos(c_begin * 4 + k, g);
        X2Mask mk0 = w2_mask(tn, g), mk1;
        unsigned ya, yb;
        X2Off xo = w2_x_off(tn, ci, g);
        {
#pragma unroll
            for (int r = 0; r < 6; ++r) w2_load_row(xr, xo, w4, r, pa);
            w2_y_off(tn, co, g, ya, yb);
            w2_load_y(yr, ya, yb, w4, 0, ry0);
            w2_load_y(yr, ya, yb, w4, co16, ry1);
#pragma unroll
            for (int j = 0; j < 3; ++j) w2_x_cols(pa, j, mk0);
#pragma unroll
            for (int i = 0; i < 6; ++i) w2_x_rowpass(pa, i, va + 6 * i);
            w2_y_colpass(ry0, mk0, m);
#pragma unroll
            for (int i = 0; i < 6; ++i) w2_y_rowpass(ry0, m, i, vb + 6 * i);
        }
        w2_advance(tn, (c_begin + 1) * 4 + k, g);          // step c + 1
        {
            xo = w2_x_off(tn, ci, g);
            mk1 = w2_mask(tn, g);
            const bool more = c_begin + 1 < c_end;
            const __amdgpu_buffer_rsrc_t xq = wn_rsrc(x, more ? x_bytes : 0u), yq = wn_rsrc(dy, more ? y_bytes : 0u);
#pragma unroll
            for (int r = 0; r < 6; ++r) w2_load_row(xq, xo, w4, r, pa);
            w2_y_off(tn, co, g, ya, yb);                     // (ya, yb): step c + 1 until the end of step c replaces them
            w2_load_y(yq, ya, yb, w4, 0, ry0);
        }
        w2_advance(tn, (c_begin + 2) * 4 + k, g);          // step c + 2

#define W2_REGION() __builtin_amdgcn_sched_barrier(0)
#define W2_ROW(i_, acc_)                                                                                                \
    _Pragma("unroll") for (int j_ = 0; j_ < 6; ++j_) W2_MFMA(6 * (i_) + j_, va[6 * (i_) + j_], vb[6 * (i_) + j_], acc_[6 * (i_) + j_])

        // one step; hc = the rows 3-5 that hold x(c+1), hn = the set that receives x(c+2)
        auto step = [&](int c, const X2Patch& hc, const X2Patch& hn) {
            const __amdgpu_buffer_rsrc_t xq = wn_rsrc(x, c + 2 < c_end ? x_bytes : 0u), yq0 = wn_rsrc(dy, c + 2 < c_end ? y_bytes : 0u),
                                         yq1 = wn_rsrc(dy, c + 1 < c_end ? y_bytes : 0u);
            // ---- first half
            w2_y_colpass(ry1, mk0, m);
            W2_ROW(0, acc0);
            W2_REGION();
            W2_ROW(1, acc0);
            w2_y_rowpass(ry1, m, 0, vb);
            w2_y_rowpass(ry1, m, 1, vb + 6);
            W2_REGION();
            W2_ROW(2, acc0);
            w2_y_rowpass(ry1, m, 2, vb + 12);
            W2_REGION();
            W2_ROW(3, acc0);
            w2_y_rowpass(ry1, m, 3, vb + 18);
            W2_REGION();
            W2_ROW(4, acc0);
            w2_y_rowpass(ry1, m, 4, vb + 24);
            W2_REGION();
            W2_ROW(5, acc0);
            w2_y_rowpass(ry1, m, 5, vb + 30);
#pragma unroll
            for (int j = 0; j < 3; ++j) w2_x_cols(hc, j, mk1);      // as late as the row pass allows: the last row of x(c+1) was requested five regions ago
            w2_load_y(yq1, ya, yb, w4, co16, ry1);          // dY1(c+1): read at the top of the next step
            xo = w2_x_off(tn, ci, g);                       // addresses of x(c+2)
            W2_REGION();
            // ---- second half
            w2_y_colpass(ry0, mk1, m);
#pragma unroll
            for (int r = 6 - W2_ALT; r < 6; ++r) w2_load_row(xq, xo, w4, r, hn);
            mk0 = mk1;
            mk1 = w2_mask(tn, g);
            W2_ROW(0, acc1);
            W2_REGION();
            W2_ROW(1, acc1);
            w2_y_rowpass(ry0, m, 0, vb);
            w2_x_rowpass(hc, 0, va);
            w2_load_row(xq, xo, w4, 0, hn, W2_HALO);
            W2_REGION();
            W2_ROW(2, acc1);
            w2_y_rowpass(ry0, m, 1, vb + 6);
            w2_x_rowpass(hc, 1, va + 6);
            w2_load_row(xq, xo, w4, 1, hn, W2_HALO);
            W2_REGION();
            W2_ROW(3, acc1);
            w2_y_rowpass(ry0, m, 2, vb + 12);
            w2_x_rowpass(hc, 2, va + 12);
            w2_load_row(xq, xo, w4, 2, hn, W2_HALO);
            W2_REGION();
            W2_ROW(4, acc1);
            w2_y_rowpass(ry0, m, 3, vb + 18);
            w2_x_rowpass(hc, 3, va + 18);
            w2_y_rowpass(ry0, m, 4, vb + 24);
            w2_x_rowpass(hc, 4, va + 24);
            if (W2_ALT < 3) w2_load_row(xq, xo, w4, 3, hn, W2_HALO);
            if (W2_ALT < 2) w2_load_row(xq, xo, w4, 4, hn, W2_HALO);
            W2_REGION();
            W2_ROW(5, acc1);
            w2_y_rowpass(ry0, m, 5, vb + 30);
            w2_x_rowpass(hc, 5, va + 30);
            if (W2_ALT < 1) w2_load_row(xq, xo, w4, 5, hn, W2_HALO);
            w2_y_off(tn, co, g, ya, yb);
            w2_load_y(yq0, ya, yb, w4, 0, ry0);             // dY0(c+2): read at the top of the next step's second half
            w2_advance(tn, (c + 3) * 4 + k, g);
            W2_REGION();
        };
#if W2_ALT
        for (int c = c_begin; c < c_end; c += 2) {
            step(c, pa, pb);
            step(c + 1, pb, pa);
        }
#else
        for (int c = c_begin; c < c_end; ++c) step(c, pa, pa);
#endif
#undef W2_REGION
#undef W2_ROW
        }
        s += seg_end - seg_begin;
        asm volatile("s_nop 15" ::: "memory");             // the last MFMAs that write VGPR accumulators have left the pipe
#if W2_TSPLIT
        // acc*[f][r] = this quarter's part of dU[f][ci = 4 k + r][co = (16) + li].  (wave 0 + wave 2) + (wave 1 + wave 3) through LDS, in
        // that order whatever the timing; wave 0 dumps the sum, one contiguous KiB per store instruction.
        {
            float* const mine = red_lds + (size_t)(wave & 1) * W2_SLAB_FLOATS + lane * 4;
            if (wave >= 2) {
#pragma unroll
                for (int f = 0; f < WN_FREQ; ++f) {
                    *reinterpret_cast<f32x4*>(mine + f * 256) = acc0[f];
                    *reinterpret_cast<f32x4*>(mine + (WN_FREQ + f) * 256) = acc1[f];
                }
            }
            __syncthreads();
            if (wave < 2) {
#pragma unroll
                for (int f = 0; f < WN_FREQ; ++f) {
                    acc0[f] += *reinterpret_cast<const f32x4*>(mine + f * 256);
                    acc1[f] += *reinterpret_cast<const f32x4*>(mine + (WN_FREQ + f) * 256);
                }
            }
            __syncthreads();
            if (wave == 1) {
#pragma unroll
                for (int f = 0; f < WN_FREQ; ++f) {
                    *reinterpret_cast<f32x4*>(red_lds + lane * 4 + f * 256) = acc0[f];
                    *reinterpret_cast<f32x4*>(red_lds + lane * 4 + (WN_FREQ + f) * 256) = acc1[f];
                }
            }
            __syncthreads();
            if (wave == 0) {
                float* const sl = slabs + (size_t)(w + unit) * W2_SLAB_FLOATS + lane * 4;
#pragma unroll
                for (int f = 0; f < WN_FREQ; ++f) {
                    *reinterpret_cast<f32x4*>(sl + f * 256) = acc0[f] + *reinterpret_cast<const f32x4*>(red_lds + lane * 4 + f * 256);
                    *reinterpret_cast<f32x4*>(sl + (WN_FREQ + f) * 256) = acc1[f] + *reinterpret_cast<const f32x4*>(red_lds + lane * 4 + (WN_FREQ + f) * 256);
                }
            }
            __syncthreads();                               // the next segment's partial sums go into the same LDS
        }
#else
        // acc*[f][r] = dU[f][ci = 16 wm + 4 k + r][co = 32 wn + (16) + li]: dumped as it stands, one contiguous KiB per instruction
        float* const sl = slabs + (size_t)(w + unit) * W2_SLAB_FLOATS + (size_t)(wave * 2) * (WN_FREQ * 256) + lane * 4;
#pragma unroll
        for (int f = 0; f < WN_FREQ; ++f) {
            *reinterpret_cast<f32x4*>(sl + f * 256) = acc0[f];
            *reinterpret_cast<f32x4*>(sl + (WN_FREQ + f) * 256) = acc1[f];
        }
#endif
    }
}
#undef W2_MFMA

// dw[co][ci][3][3] = G^T dU G from the folded slabs of wino_wgrad_reg_kernel.  One thread per (unit, wave, half, lane, register) in
// slab order: the 36 reads of a wave are 36 contiguous 256-byte runs.
__global__ void __launch_bounds__(WN_BLOCK)
wino_wgrad_reg_finish_kernel(const float* __restrict__ red, const Wg2Geom wg, float* __restrict__ dw) {
    const int e = blockIdx.x * WN_BLOCK + threadIdx.x;
    const int cin = wg.g.Cin, cout = wg.g.Cout;
    if (e >= cin * cout) return;
#if W2_TSPLIT
    const int r = e & 3, lane = (e >> 2) & 63, half = (e >> 8) & 1, wave = 0, unit = e >> 9;                  // 512 (ci, co) pairs per unit
#else
    const int r = e & 3, lane = (e >> 2) & 63, half = (e >> 8) & 1, wave = (e >> 9) & 3, unit = e >> 11;      // 2048 (ci, co) pairs per unit
#endif
    const int cib = unit / wg.n_cob, cob = unit - cib * wg.n_cob;
    const int ci = cib * W2_CI + (wave >> 1) * 16 + (lane >> 4) * 4 + r, co = cob * W2_CO + (wave & 1) * 32 + half * 16 + (lane & 15);
    const float* sl = red + (size_t)unit * W2_SLAB_FLOATS + (size_t)((wave * 2 + half) * WN_FREQ) * 256 + lane * 4 + r;
    double du[WN_FREQ];
#pragma unroll
    for (int f = 0; f < WN_FREQ; ++f) du[f] = (double)sl[f * 256];
    double t[3][6];
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

