// Exact fp32 gather-GEMM on the BF16 matrix pipe (the "split" matrix path; included by spconv.hip).
//
// gfx950 multiplies fp32 operands at 64 FLOP/clk/SIMD (v_mfma_f32_16x16x4_f32) and bf16 operands at 1024
// (v_mfma_f32_16x16x32_bf16): 16 x.  An fp32 value is EXACTLY the sum of three bf16 values,
//     x = hi + mid + lo,   hi = x with its low 16 bits cleared, mid = (x - hi) likewise, lo = x - hi - mid
// (24 significant bits = 8 + 8 + 8; both differences are exact in fp32, and lo has at most 8 significant bits, so it is a bf16
// number), and a product of two bf16 values is exact in fp32.  So
//     a . b = ah.bh + (ah.bm + am.bh) + (ah.bl + am.bm + al.bh) + [am.bl + al.bm + al.bl]
// where the bracket is dropped: with truncated planes |mid| < 2^-7 |x| and |lo| < 2^-15 |x|, so each of am.bl and al.bm is below 2^-22
// of |a . b| in the worst case and 2^-25 on average (uniform low bits) - the size of the rounding of an fp32 product (2^-24); the gate
// is measured, not argued: rms and max error against fp64 <= 1.5 x the fp32 instructions' (tests/test_gpu_split.py).  Six bf16
// matrix instructions with fp32 accumulation stand for sixteen fp32 ones (K = 32 against K = 4 per instruction at half the
// cycles: 6 / 16 of the matrix cycles).  Every (tile, 32-channel chunk) adds its six terms smallest first.
//
// Structure = the per-offset output-stationary kernel of spconv.hip (gather_gemm_lds_kernel) with these differences:
//   * weights are split ONCE per step by the pack kernel into three bf16 planes in fragment order (6 bytes per weight; the slice of an
//     offset goes global -> LDS by LDS-DMA, no staging registers, double buffered);
//   * gathered rows arrive as fp32 (same 16-byte bounds-checked buffer loads, "no neighbour" = out-of-range offset = zeros) and are
//     split in registers (5.5 vector instructions per value) - the vector pipe issues beside the matrix pipe of the other waves of the SIMD;
//   * the rows of offset k + 1 are requested before the matrix work of offset k (one offset of look-ahead per wave) and land under it.
// MFMA operand maps (cdna_hip_programming.md section 3): A[i = lane & 15][k = 8 (lane >> 4) + j], B[k = 8 (lane >> 4) + j][lane & 15],
// D: col = lane & 15, row = 4 (lane >> 4) + reg (the map of the fp32 instruction: the epilogue is the one of gather_gemm_lds_kernel).
// k-permutation: lane (r, g) contracts element j of chunk kc over the gathered channel 32 kc + 16 (j >> 2) + 4 g + (j & 3): its two
// 16-byte loads per chunk are the pieces 4 g .. 4 g + 3 of the 64-byte segments 2 kc and 2 kc + 1 of its row, so ONE load instruction
// of the wave reads 16 rows x one whole 64-byte segment (a lane reading 32 KC contiguous bytes makes every instruction touch all
// segments of all 16 rows for a quarter of their bytes: 4 x the cache-line accesses).  The packed planes follow the same map.
// Produced channel of column c of tile n: NT c + n (vector stores).
#pragma once
#include "split_common.cuh"

namespace toda {

#ifndef WGS_ACCS
#define WGS_ACCS 2       // accumulator sets of the split weight gradient (see wgrad_split_kernel; 1 and 3 are measurement builds)
#endif
#ifndef SP_STAMPS
#define SP_STAMPS 0      // diagnostic build only: per-wave cycle sums of the loop's segments through the statistics pointer (no statistics)
#endif
#if SP_STAMPS
#define SP_STAMP(i)                                                                               \
    do {                                                                                          \
        unsigned long long t_;                                                                    \
        __builtin_amdgcn_sched_barrier(0);                                                        \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");                \
        __builtin_amdgcn_sched_barrier(0);                                                        \
        seg[i] += t_ - t_prev;                                                                    \
        t_prev = t_;                                                                              \
    } while (0)
#else
#define SP_STAMP(i)
#endif
#ifndef SP_ABLATE
#define SP_ABLATE 0      // measurement builds only (wrong numbers): 1 = no operand split, 2 = no matrix instructions
#endif
// 8 consecutive fp32 values -> the three bf16 planes of an A fragment
__device__ __forceinline__ void sp_split8(const f32x4& v0, const f32x4& v1, u32x4& h, u32x4& m, u32x4& l) {
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const float x0 = p < 2 ? v0[2 * p] : v1[2 * p - 4], x1 = p < 2 ? v0[2 * p + 1] : v1[2 * p - 3];
        h[p] = sp_pack_hi(x0, x1);
#if SP_ABLATE & 1
        m[p] = h[p] ^ 0x00010001u, l[p] = h[p] ^ 0x00020002u;
        continue;
#endif
        const float r0 = x0 - sp_trunc(x0), r1 = x1 - sp_trunc(x1);
        m[p] = sp_pack_hi(r0, r1);
        const float l0 = r0 - sp_trunc(r0), l1 = r1 - sp_trunc(r1);
        l[p] = sp_pack_hi(l0, l1);
    }
}
__device__ __forceinline__ unsigned short sp_plane_of(float v, int plane) {
    const float h = sp_trunc(v), r = v - h, m = sp_trunc(r), l = r - m;
    const float pick = plane == 0 ? h : (plane == 1 ? m : l);
    return (unsigned short)(__builtin_bit_cast(unsigned, pick) >> 16);
}

// ---- packed operand: u32x4 units  wps[((((k KC + kc) NT + n) 3 + plane) 64 + lane)] = 8 bf16, j = 0 .. 7 ---------------------------
constexpr int SPLIT_PACK_MAX_SEG = 48;
struct SplitPackBatch {
    const float* w[SPLIT_PACK_MAX_SEG];
    u32x4* wps[SPLIT_PACK_MAX_SEG];
    int cout[SPLIT_PACK_MAX_SEG], K[SPLIT_PACK_MAX_SEG], cin[SPLIT_PACK_MAX_SEG];
    unsigned char transpose[SPLIT_PACK_MAX_SEG], flip[SPLIT_PACK_MAX_SEG], KC[SPLIT_PACK_MAX_SEG], NT[SPLIT_PACK_MAX_SEG];
};
// one thread = one lane's 16 bytes of one plane; blockIdx.y = segment (all split operands of a step in one launch)
__global__ void __launch_bounds__(SC_BLOCK)
split_pack_batch_kernel(const SplitPackBatch b) {
    const int sg = blockIdx.y;
    const int K = b.K[sg], KC = b.KC[sg], NT = b.NT[sg], cin = b.cin[sg], cout = b.cout[sg];
    const long long e = (long long)blockIdx.x * SC_BLOCK + threadIdx.x;
    if (e >= (long long)K * KC * NT * 192) return;
    const int lane = (int)(e & 63);
    long long t = e >> 6;
    const int plane = (int)(t % 3);
    t /= 3;
    const int n = (int)(t % NT);
    t /= NT;
    const int kc = (int)(t % KC);
    const int k = (int)(t / KC);
    const int c = lane & 15, g = lane >> 4;
    const int pch = NT * c + n;
    const int kk = b.flip[sg] ? K - 1 - k : k;
    const float* __restrict__ w = b.w[sg];
    unsigned short v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int gch = 32 * kc + 16 * (j >> 2) + 4 * g + (j & 3);
        float x = 0.0f;
        if (!b.transpose[sg]) {
            if (gch < cin && pch < cout) x = w[((size_t)pch * K + kk) * cin + gch];
        } else {
            if (gch < cout && pch < cin) x = w[((size_t)gch * K + kk) * cin + pch];
        }
        v[j] = sp_plane_of(x, plane);
    }
    b.wps[sg][e] = u32x4{(unsigned)v[0] | ((unsigned)v[1] << 16), (unsigned)v[2] | ((unsigned)v[3] << 16), (unsigned)v[4] | ((unsigned)v[5] << 16),
                         (unsigned)v[6] | ((unsigned)v[7] << 16)};
}

// ---- the kernel -------------------------------------------------------------------------------------------------------------------
// A STAGE is KCS of the KC 32-channel chunks of one offset: its weight slice (KCS NT 3 KiB) is what the workgroup shares through LDS
// (double buffered, LDS-DMA), its rows' 128 KCS bytes per row are what a lane gathers one stage ahead.  64 -> 64: one stage per offset
// (24 KiB); 128 -> 128: four stages per offset (24 KiB each: the whole 96 KiB slice of an offset would not fit twice).
// BLK threads = BLK / 64 waves share the slice: every workgroup re-reads the whole packed operand, so with 256-thread workgroups the
// slices are HALF of the kernel's vector-memory traffic at 64 -> 64 (2.0 GB beside 1.9 GB of gathered rows at 389 k rows).
// Measured (C3 / C5 levels, ms per launch): ONE chunk per stage wins wherever it was tried - 64 -> 64 @ 389 k rows 0.346 (256 threads,
// KCS 1: 122 registers, 4 waves per SIMD, 24 KiB of LDS) against 0.354 (768 threads, KCS 2) and 0.367 (256, KCS 2: 168 registers, 3 waves);
// @ 117 k rows 0.124 / 0.179 / 0.135; strided 64 -> 64 forward 0.104 / 0.154 / 0.122: more resident waves hide more of the gathers.
template <int KC, int KCS, int NT, int RT, int BLK, int WAVES, bool IL = false>
__global__ void __launch_bounds__(BLK, WAVES)
gg_split_kernel(const float* __restrict__ in, int n_in, const u32x4* __restrict__ wps, const int* __restrict__ nbr, int n_out, int K, int cp,
                const float* __restrict__ bias, float* __restrict__ out, const int* __restrict__ order, double* __restrict__ stats,
                const unsigned char* __restrict__ cls_sorted, const GatherClasses classes) {
    static_assert(KC % KCS == 0, "whole stages per offset");
    constexpr int CG = 32 * KC;
    constexpr int SPO = KC / KCS;            // stages per offset
    constexpr int UNITS = KCS * NT * 3;      // 1 KiB wave-instruction images per stage
    constexpr int SLICE = UNITS * 64;        // u32x4 per stage
    static_assert(2 * SLICE * 16 >= (BLK / 64) * 2 * 16 * NT * 4, "the statistics scratch aliases the weight buffers");
    __shared__ u32x4 wl[2 * SLICE];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int blk = xcd_chunked_block(blockIdx.x, gridDim.x);
    const int wave = blk * (BLK / 64) + wv;
    const int r = lane & 15, g = lane >> 4;
    const int row0 = wave * (16 * RT);

    f32x4 acc[RT][NT];
#pragma unroll
    for (int n = 0; n < NT; ++n) {
        float b = 0.0f;
        if (bias && NT * r + n < cp) b = bias[NT * r + n];
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) acc[rt][n] = f32x4{b, b, b, b};
    }
    const __amdgpu_buffer_rsrc_t in_rsrc = table_rsrc(in, (unsigned)n_in * (unsigned)CG * 4u);
    int rows[RT];
    bool live[RT];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
        live[rt] = row0 + rt * 16 + r < n_out;       // rows past the end read the ids of the last row and are never stored
        rows[rt] = live[rt] ? (order ? order[row0 + rt * 16 + r] : row0 + rt * 16 + r) : n_out - 1;
    }
    // Data gradient of a strided convolution over class-sorted rows (GatherClasses, spconv.hip): when all rows of the WORKGROUP share one
    // residue class it walks only that class's offset list (<= 8 of 27, ascending: the same sums in the same order); the barriers
    // are the workgroup's, so one wave of another class sends the whole workgroup over all K offsets.
    int KE = K;                                  // offsets this workgroup walks
    bool listed = false;
    unsigned long long kpack = 0;                // their indices, one byte each (a class has <= 8), in a scalar register pair
    if (cls_sorted) {
        const int p = row0 + (lane & (16 * RT - 1));
        const int c = cls_sorted[p < n_out ? p : n_out - 1], c0 = __builtin_amdgcn_readfirstlane(c);
        const bool same = __all(c == c0 || p >= n_out);
        int* cw = reinterpret_cast<int*>(wl);
        if (lane == 0) cw[wv] = same ? c0 : -1;
        __syncthreads();
        int cb = cw[0];
#pragma unroll
        for (int w = 1; w < BLK / 64; ++w) cb = (cw[w] == cb) ? cb : -1;
        __syncthreads();                          // before the first slice lands in wl
        if (cb >= 0 && classes.count[cb & 7] > 0 && classes.count[cb & 7] <= 8) {
            listed = true;
            KE = classes.count[cb & 7];
#pragma unroll
            for (int i = 0; i < 8; ++i) kpack |= (unsigned long long)classes.k[cb & 7][i] << (8 * i);
            kpack = ((unsigned long long)__builtin_amdgcn_readfirstlane((int)(kpack >> 32)) << 32) | (unsigned)__builtin_amdgcn_readfirstlane((int)kpack);
        }
    }
    auto kof = [&](int j) { return listed ? (int)((kpack >> (8 * j)) & 0xFF) : j; };
    auto load_ids = [&](int j, int (&dst)[RT]) {
        const int kk = kof(j < KE ? j : KE - 1);
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
            dst[rt] = __builtin_nontemporal_load(nbr + (size_t)kk * n_out + rows[rt]);      // past the last offset: its ids again, never used
        }
    };
    // rows of stage `part` of an offset: the lane's 16 bytes of the 64-byte segments 2 KCS part .. 2 KCS (part + 1) - 1
    auto gather = [&](const int (&src)[RT], int part, bool valid, f32x4 (&raw)[RT][2 * KCS]) {      // valid (wave-uniform): the offset exists
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
            const bool ok = src[rt] >= 0 && valid;
            const unsigned base = (unsigned)src[rt] * (unsigned)(CG * 4) + (unsigned)(16 * g) + (unsigned)(128 * KCS) * (unsigned)part;
#pragma unroll
            for (int i = 0; i < 2 * KCS; ++i)
                raw[rt][i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(in_rsrc, ok ? base + 64u * i : OOB, 0, 0));
        }
    };
    auto gather_rt = [&](const int (&src)[RT], int part, bool valid, f32x4 (&raw)[RT][2 * KCS], int rt) {
        const bool ok = src[rt] >= 0 && valid;
        const unsigned base = (unsigned)src[rt] * (unsigned)(CG * 4) + (unsigned)(16 * g) + (unsigned)(128 * KCS) * (unsigned)part;
#pragma unroll
        for (int i = 0; i < 2 * KCS; ++i)
            raw[rt][i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(in_rsrc, ok ? base + 64u * i : OOB, 0, 0));
    };
    auto stage = [&](int st, int buf) {       // the slice of stage st = j SPO + part: global -> LDS, 1 KiB per wave-instruction, no registers
        const int sc = st < KE * SPO ? st : KE * SPO - 1;
        const int ss = kof(sc / SPO) * SPO + sc % SPO;
#pragma unroll
        for (int t = 0; t < (UNITS + BLK / 64 - 1) / (BLK / 64); ++t) {
            const int u = t * (BLK / 64) + wv;
            if (UNITS % (BLK / 64) == 0 || u < UNITS)
                __builtin_amdgcn_global_load_lds(reinterpret_cast<const float*>(wps + (size_t)ss * SLICE + u * 64 + lane),
                                                 reinterpret_cast<float*>(&wl[buf * SLICE + u * 64]), 16, 0, 0);
        }
    };

    // ids live in three register sets that rotate by NAME (the loop body is written three times): a copy of a set whose load is
    // still in flight, or the "row exists" select applied at load time, would be a wait for that load at the end of every offset
    int idA[RT], idB[RT], idC[RT];
    f32x4 raw[RT][2 * KCS];
    load_ids(0, idA);
    load_ids(1, idB);
    stage(0, 0);
    gather(idA, 0, true, raw);
    __syncthreads();

#if SP_STAMPS
    unsigned long long seg[6] = {0, 0, 0, 0, 0, 0}, t_prev;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_prev)::"memory");
#endif
    auto body = [&](int k, const int (&ids_cur)[RT], const int (&ids_nxt)[RT], int (&ids_new)[RT]) {
        SP_STAMP(5);
        bool any = false;
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) any = any || __any(ids_cur[rt] >= 0);
#pragma unroll
        for (int part = 0; part < SPO; ++part) {
            const int cur = (k * SPO + part) & 1;
            // one straight-line body per (wave, offset) that has a neighbour in any of its rows: on the submanifold tables 0.83 of the
            // executed tile rows are pairs this way against 0.85 with a test per 16-row tile (C3, 389 k rows) - not worth three bodies
            // (hipcc joins them with 32 accumulator copies per offset)
            u32x4 ah[RT][KCS], am[RT][KCS], al[RT][KCS];
            f32x4 (&rw)[RT][2 * KCS] = raw;      // this stage's rows
            if (any) {
#pragma unroll
                for (int rt = 0; rt < RT; ++rt)
#pragma unroll
                    for (int kc = 0; kc < KCS; ++kc) sp_split8(rw[rt][2 * kc], rw[rt][2 * kc + 1], ah[rt][kc], am[rt][kc], al[rt][kc]);
            }
            SP_STAMP(0);
            if constexpr (IL) {
                // IL: the stage's vector-memory instructions are issued BETWEEN the groups of matrix instructions instead of in front of
                // them (in front, a wave sits in the issue of its ~9 loads for 0.25-0.45 of its time - in-kernel stamps - with its matrix
                // instructions waiting behind them in program order).  Same relative order (slice, rows, ids): the counted waits below
                // hold.  Measured (C3 levels, ms per launch): 32 -> 32 @ 682 k 0.230 -> 0.222, 64 -> 32 class-sorted dgrad 0.180 -> 0.166 (plain
                // 0.351 -> 0.283), but 64 -> 64 0.347 -> 0.347 and the strided 64 -> 64 forward 0.107 -> 0.110: the launches with 32 produced
                // channels (few matrix instructions per gathered byte) take it, the others do not.  What the ablation builds say about
                // 64 -> 64 (SP_ABLATE): without matrix instructions 0.232 ms, without row gathers 0.245, without slice DMA 0.307, without the
                // operand split 0.361 of 0.365 - the L2 -> CU path alone needs two thirds of the kernel's time (4 GB per launch: 1.9 GB of
                // 64-byte row segments + 2.0 GB of weight slices), the vector pipe is hidden, and what is left is imperfect overlap.
                constexpr int ITER = KCS * NT, PIECES = RT + 2;      // slice | rows of tile 0 .. RT-1 | ids
                auto piece = [&](int p) {
                    if (p == 0) {
#if !(SP_ABLATE & 4)
                        stage(k * SPO + part + 1, cur ^ 1);
#endif
                    } else if (p <= RT) {
                        if (part + 1 < SPO) gather_rt(ids_cur, part + 1, true, raw, p - 1);
                        else gather_rt(ids_nxt, 0, k + 1 < KE, raw, p - 1);
                    } else if (part == SPO - 1) load_ids(k + 2, ids_new);
                    asm volatile("" ::: "memory");
                };
                if (any) {
                    const u32x4* __restrict__ wb = wl + cur * SLICE + lane;
#pragma unroll
                    for (int st = 0; st < ITER; ++st) {
                        const int kc = st / NT, n = st % NT;
                        const bf16x8 bh = __builtin_bit_cast(bf16x8, wb[(st * 3 + 0) * 64]), bm = __builtin_bit_cast(bf16x8, wb[(st * 3 + 1) * 64]),
                                     bl = __builtin_bit_cast(bf16x8, wb[(st * 3 + 2) * 64]);
#pragma unroll
                        for (int p = 0; p < PIECES; ++p)
                            if (p * ITER / PIECES == st) piece(p);
#define SP_TERM(AA, BB)                                                                                                              \
    _Pragma("unroll") for (int rt = 0; rt < RT; ++rt)                                                                                \
        acc[rt][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, AA[rt][kc]), BB, acc[rt][n], 0, 0, 0)
                        SP_TERM(al, bh);
                        SP_TERM(ah, bl);
                        SP_TERM(am, bm);
                        SP_TERM(am, bh);
                        SP_TERM(ah, bm);
                        SP_TERM(ah, bh);
#undef SP_TERM
                    }
                } else {
#pragma unroll
                    for (int p = 0; p < PIECES; ++p) piece(p);
                }
            } else {
            // issue order = the order the waits below and at the top of the next stage retire them in: the slice of the next stage (needed
            // by every wave behind the barrier), the rows of the next stage, the ids of offset k + 2
#if !(SP_ABLATE & 4)
            stage(k * SPO + part + 1, cur ^ 1);
#endif
            asm volatile("" ::: "memory");
#if SP_ABLATE & 8
            if (k < 0)
#endif
            if (part + 1 < SPO) gather(ids_cur, part + 1, true, raw);
            else gather(ids_nxt, 0, k + 1 < KE, raw);          // in flight under the matrix work below
            asm volatile("" ::: "memory");
#if !(SP_ABLATE & 16)
            if (part == SPO - 1) load_ids(k + 2, ids_new);
#endif
            asm volatile("" ::: "memory");
            SP_STAMP(1);
            if (any) {
                const u32x4* __restrict__ wb = wl + cur * SLICE + lane;
                // (reading the three planes of a step one step ahead of its matrix instructions - 12 more registers, the order pinned
                // with sched_group_barrier - measured nothing at three waves per SIMD and spilled in the widest variants)
#pragma unroll
                for (int st = 0; st < KCS * NT; ++st) {
                    const int kc = st / NT, n = st % NT;
                    const bf16x8 bh = __builtin_bit_cast(bf16x8, wb[(st * 3 + 0) * 64]), bm = __builtin_bit_cast(bf16x8, wb[(st * 3 + 1) * 64]),
                                 bl = __builtin_bit_cast(bf16x8, wb[(st * 3 + 2) * 64]);
#if SP_ABLATE & 2
#define SP_TERM(AA, BB)                                                                                                              \
    _Pragma("unroll") for (int rt = 0; rt < RT; ++rt)                                                                                \
        acc[rt][n][0] += __builtin_bit_cast(float, AA[rt][kc][0] ^ __builtin_bit_cast(u32x4, BB)[0])
#else
#define SP_TERM(AA, BB)                                                                                                              \
    _Pragma("unroll") for (int rt = 0; rt < RT; ++rt)                                                                                \
        acc[rt][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, AA[rt][kc]), BB, acc[rt][n], 0, 0, 0)
#endif
                    SP_TERM(al, bh);
                    SP_TERM(ah, bl);
                    SP_TERM(am, bm);
                    SP_TERM(am, bh);
                    SP_TERM(ah, bm);
                    SP_TERM(ah, bh);
#undef SP_TERM
                }
            }
            }
            // The barrier only hands over the LDS slice: wait for this wave's share of it (the oldest of the loads above) and leave the
            // rows and ids in flight across the barrier - __syncthreads() would drain them all (vmcnt(0): an LDS-DMA is a pending LDS write)
            SP_STAMP(2);
#if SP_ABLATE & (4 | 8 | 16)
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#else
            if (part == SPO - 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(RT * 2 * KCS + RT) : "memory");
            else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(RT * 2 * KCS) : "memory");
#endif
            SP_STAMP(3);
            __builtin_amdgcn_s_barrier();
            SP_STAMP(4);
        }
    };
    for (int k = 0;;) {          // k counts walked offsets (positions of the class list when there is one)
        body(k, idA, idB, idC);
        if (++k >= KE) break;
        body(k, idB, idC, idA);
        if (++k >= KE) break;
        body(k, idC, idA, idB);
        if (++k >= KE) break;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (row0 >= n_out && !stats) return;

#if SP_STAMPS
    if (stats) {
        if (lane == 0)
            for (int i = 0; i < 6; ++i) reinterpret_cast<unsigned long long*>(stats)[(size_t)wave * 6 + i] = seg[i];
        stats = nullptr;
    }
#endif
    // BatchNorm statistics of the layer's output from the accumulators - as gather_gemm_lds_kernel (same scratch layout and fold)
    if (stats) {
        float (*st_sh)[2][16 * NT] = reinterpret_cast<float (*)[2][16 * NT]>(wl);      // every wave is past its last slice read (barrier above)
        float sm[NT], sq[NT];
#pragma unroll
        for (int n = 0; n < NT; ++n) {
            sm[n] = sq[n] = 0.0f;
#pragma unroll
            for (int rt = 0; rt < RT; ++rt)
#pragma unroll
                for (int reg = 0; reg < 4; ++reg)
                    if (row0 + rt * 16 + 4 * g + reg < n_out) {
                        const float v = acc[rt][n][reg];
                        sm[n] += v;
                        sq[n] += v * v;
                    }
            sm[n] += __shfl_xor(sm[n], 16, 64);
            sq[n] += __shfl_xor(sq[n], 16, 64);
            sm[n] += __shfl_xor(sm[n], 32, 64);
            sq[n] += __shfl_xor(sq[n], 32, 64);
            if (g == 0) {
                st_sh[wv][0][NT * r + n] = sm[n];
                st_sh[wv][1][NT * r + n] = sq[n];
            }
        }
        __syncthreads();
        if ((int)threadIdx.x < 2 * cp) {
            const int qq = threadIdx.x / cp, ch = threadIdx.x - qq * cp;
            double a = 0.0;
#pragma unroll
            for (int w = 0; w < BLK / 64; ++w) a += (double)st_sh[w][qq][ch];
            stats[2 * cp + (size_t)(qq * cp + ch) * gridDim.x + blk] = a;
        }
    }

    const bool full = cp == 16 * NT;
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
            const int pos = row0 + rt * 16 + 4 * g + reg;
            const int row = __shfl(rows[rt], 4 * g + reg, 64);
            if (pos >= n_out) continue;
            float* dst = out + (size_t)row * cp + NT * r;
            if (full) {
                if constexpr (NT == 2) {
                    *reinterpret_cast<float2*>(dst) = make_float2(acc[rt][0][reg], acc[rt][1][reg]);
                } else {
#pragma unroll
                    for (int n = 0; n < NT; n += 4)
                        *reinterpret_cast<f32x4*>(dst + n) = f32x4{acc[rt][n][reg], acc[rt][n + 1][reg], acc[rt][n + 2][reg], acc[rt][n + 3][reg]};
                }
            } else {
#pragma unroll
                for (int n = 0; n < NT; ++n)
                    if (NT * r + n < cp) dst[n] = acc[rt][n][reg];
            }
        }
    }
}

// ---- weight gradient on the split path -----------------------------------------------------------------------------------------------
// dW[co][k][ci] = sum over the pairs (i, o) of offset k of in[i][ci] dout[o][co]: the structure of wgrad_kernel (spconv.hip: one workgroup
// per (row chunk, offset), every wave compacts the valid pairs of its rows into an LDS queue and contracts over PAIRS, per-chunk slabs
// folded in fixed order) with 32 pairs per matrix instruction instead of 4: lane (ii, g) holds the pairs 8 g .. 8 g + 7 of a round for the
// channels MT ii .. MT ii + MT - 1 of both sides - one 16-byte (MT = 4) or 8-byte (MT = 2) load per pair and side, the 16 lanes of a
// group read one whole row -, splits the 8 MT values of a side into three bf16 planes whose registers already ARE the fragments
// (element j of tile m = pair 8 g + j, channel MT ii + m), and the six cross terms go to v_mfma_f32_16x16x32_bf16.  Both operands are
// activations here, so both are split in registers: 11 vector instructions per 2 values and side.
// TWO accumulator sets: the three terms of order 2^-16 (al.bh, ah.bl, am.bm) go to their own registers and are added once at the end.
// A contraction over hundreds of pairs makes the accumulator large against one instruction's contribution, and every matrix
// instruction rounds its accumulator: six roundings per 32 pairs into ONE set measured 1.88 x the native kernel's rms error against fp64
// (64 x 64 @ 389 k rows: 7.8e-7 against 4.2e-7 of the result's rms), the small terms apart 0.91 x (3.8e-7), three sets (hh alone) 0.55 x
// but 192 accumulator registers spill; rounded-to-nearest planes instead of truncated ones change nothing (the dropped cross terms are
// not what is measured).  Same time per launch as one set.
// QM x QN quarters (128-channel sides, as wgrad_kernel's cooperative form): the weight block is QM x QN quarters of (16 MTB) x (16 NTB); the four
// waves are 4 / (QM QN) TEAMS per quarter.  The waves of a team's quarter walk the SAME pairs (lane channels (MTB QM) ii + MTB q .. of
// a side: the same 16-byte loads at a wider lane stride), the teams of a quarter take alternate 64-row batches and are folded in
// fixed order.  1 x 1: four teams, one block (32 / 64-channel sides); 2 x 2: 128 x 128, no fold; 1 x 2 / 2 x 1: 64 x 128 / 128 x 64, two teams.
template <int MTB, int NTB, int QM = 1, int QN = 1>
// 32 x 64 wants 175 registers: two waves per SIMD without spills (0.129 ms at spconv3 of C3) beat three with 13 spilled dwords (0.162)
__global__ void __launch_bounds__(SC_BLOCK, (MTB * NTB <= 4) ? 4 : ((MTB == 4 && NTB == 2) ? 3 : 2))
wgrad_split_kernel(const float* __restrict__ in, int n_in, const float* __restrict__ dout, const int* __restrict__ nbr, int n_out, int K,
                   int rows_per_chunk, float* __restrict__ slab, int xcd_chunks) {
    static_assert((MTB == 2 || MTB == 4) && (NTB == 2 || NTB == 4), "32 or 64 channels a side");
    static_assert(QM * QN == 1 || (MTB == 4 && NTB == 4), "quarters are 64 x 64");
    static_assert(QM * QN == 1 || QM * QN == 2 || QM * QN == 4, "1, 2 or 4 quarters");
    constexpr int NQ = QM * QN, TEAMS = (SC_BLOCK / 64) / NQ;
    constexpr int MT = MTB * QM, NT = NTB * QN;      // tiles of the whole sides
    constexpr int cin = 16 * MT, cout = 16 * NT;
    constexpr int QCAP = 64 + 32;
    __shared__ float red[TEAMS == 1 ? 1 : NQ * MTB * NTB * 4 * 64];
    __shared__ int q_in[SC_BLOCK / 64][QCAP], q_out[SC_BLOCK / 64][QCAP];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int ii = lane & 15, g = lane >> 4;
    const int n_chunks = xcd_chunks < 0 ? -xcd_chunks : xcd_chunks;
    int chunk, k;
    {
        const int b = blockIdx.x;
        if (xcd_chunks > 0) {      // the K offset-blocks of a chunk side by side on ONE XCD (see wgrad_kernel)
            const int xcd = b & 7, t = (b >> 3) / K;
            k = (b >> 3) - t * K;
            chunk = t * 8 + xcd;
        } else {
            chunk = b % n_chunks;
            k = b / n_chunks;
        }
    }
    const int row_begin = chunk * rows_per_chunk;
    if (row_begin >= n_out) return;      // a padding chunk of the XCD-ordered grid (whole block, before any barrier)
    const int row_end = min(n_out, row_begin + rows_per_chunk);
    int* qi = q_in[wv];
    int* qo = q_out[wv];

    f32x4 acc[MTB][NTB];
#if WGS_ACCS >= 2
    f32x4 acc_s[MTB][NTB];
#endif
#if WGS_ACCS >= 3
    f32x4 acc_m[MTB][NTB];
#endif
#pragma unroll
    for (int m = 0; m < MTB; ++m)
#pragma unroll
        for (int n = 0; n < NTB; ++n) {
            acc[m][n] = f32x4{0.f, 0.f, 0.f, 0.f};
#if WGS_ACCS >= 2
            acc_s[m][n] = f32x4{0.f, 0.f, 0.f, 0.f};
#endif
#if WGS_ACCS >= 3
            acc_m[m][n] = f32x4{0.f, 0.f, 0.f, 0.f};
#endif
        }
    const __amdgpu_buffer_rsrc_t in_rsrc = table_rsrc(in, (unsigned)n_in * (unsigned)cin * 4u);
    const __amdgpu_buffer_rsrc_t dout_rsrc = table_rsrc(dout, (unsigned)n_out * (unsigned)cout * 4u);
    const int qd = wv % NQ, team = wv / NQ;
    const int m0 = (qd / QN) * MTB, n0 = (qd % QN) * NTB;
    const unsigned lane_a = (unsigned)(MT * ii + m0) * 4u, lane_b = (unsigned)(NT * ii + n0) * 4u;
    constexpr unsigned row_a = (unsigned)cin * 4u, row_b = (unsigned)cout * 4u;

    // one round: queue entries [d, d + 32); entries >= limit contribute zeros (out-of-range buffer offsets)
    auto round32 = [&](int d, int limit, auto full_tag) {
        constexpr bool FULL = decltype(full_tag)::value;
        float a[8][MTB], b[8][NTB];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int p = d + 8 * g + j;
            const bool ok = FULL || p < limit;
            const int pc = ok ? p : d;
            const unsigned ia = (unsigned)qi[pc] + lane_a, ib = (unsigned)qo[pc] + lane_b;
            if constexpr (MTB == 4) {
                const f32x4 v = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(in_rsrc, ok ? ia : OOB, 0, 0));
#pragma unroll
                for (int m = 0; m < 4; ++m) a[j][m] = v[m];
            } else {
                const f32x2w v = __builtin_bit_cast(f32x2w, __builtin_amdgcn_raw_buffer_load_b64(in_rsrc, ok ? ia : OOB, 0, 0));
                a[j][0] = v[0], a[j][1] = v[1];
            }
            if constexpr (NTB == 4) {
                const f32x4 v = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(dout_rsrc, ok ? ib : OOB, 0, 0));
#pragma unroll
                for (int n = 0; n < 4; ++n) b[j][n] = v[n];
            } else {
                const f32x2w v = __builtin_bit_cast(f32x2w, __builtin_amdgcn_raw_buffer_load_b64(dout_rsrc, ok ? ib : OOB, 0, 0));
                b[j][0] = v[0], b[j][1] = v[1];
            }
        }
        // planes: element j of tile m = a[j][m]
        u32x4 ah[MTB], am[MTB], al[MTB];
#pragma unroll
        for (int m = 0; m < MTB; ++m)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                unsigned h_, m_, l_;
                sp_split2(a[2 * q][m], a[2 * q + 1][m], h_, m_, l_);
                ah[m][q] = h_, am[m][q] = m_, al[m][q] = l_;
            }
#pragma unroll
        for (int n = 0; n < NTB; ++n) {
            u32x4 bh, bm, bl;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                unsigned h_, m_, l_;
                sp_split2(b[2 * q][n], b[2 * q + 1][n], h_, m_, l_);
                bh[q] = h_, bm[q] = m_, bl[q] = l_;
            }
#define WS_TERM(ACC, AA, BB)                                                                                             \
    _Pragma("unroll") for (int m = 0; m < MTB; ++m)                                                                       \
        ACC[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, AA[m]), __builtin_bit_cast(bf16x8, BB), ACC[m][n], 0, 0, 0)
#if WGS_ACCS == 1
            WS_TERM(acc, al, bh);
            WS_TERM(acc, ah, bl);
            WS_TERM(acc, am, bm);
            WS_TERM(acc, am, bh);
            WS_TERM(acc, ah, bm);
            WS_TERM(acc, ah, bh);
#elif WGS_ACCS == 2
            WS_TERM(acc_s, al, bh);
            WS_TERM(acc_s, ah, bl);
            WS_TERM(acc_s, am, bm);
            WS_TERM(acc, am, bh);
            WS_TERM(acc, ah, bm);
            WS_TERM(acc, ah, bh);
#else
            WS_TERM(acc_s, al, bh);
            WS_TERM(acc_s, ah, bl);
            WS_TERM(acc_s, am, bm);
            WS_TERM(acc_m, am, bh);
            WS_TERM(acc_m, ah, bm);
            WS_TERM(acc, ah, bh);
#endif
#undef WS_TERM
        }
    };

    int qn = 0;  // wave-uniform queue length (< 32 between batches)
    constexpr int BSTEP = TEAMS * 64;
    const int base0 = row_begin + team * 64;
    int iv_next = nbr[(size_t)k * n_out + min(base0 + lane, n_out - 1)];  // clamped, unconditional: the ids of the NEXT 64 rows are in flight
    for (int base = base0; base < row_end; base += BSTEP) {               // under this batch's rounds
        const int o = base + lane;
        const int iv = iv_next;
        iv_next = nbr[(size_t)k * n_out + min(o + BSTEP, n_out - 1)];
        const int i = o < row_end ? iv : -1;
        const unsigned long long vote = __ballot(i >= 0);
        if (vote == 0) continue;
        if (i >= 0) {
            const int pos = qn + __popcll(vote & ((1ull << lane) - 1));
            qi[pos] = (int)((unsigned)i * row_a);      // byte offsets of the rows
            qo[pos] = (int)((unsigned)o * row_b);
        }
        qn += __popcll(vote);
        __builtin_amdgcn_wave_barrier();
        int done = 0;
        while (qn - done >= 32) {
            round32(done, qn, std::true_type{});
            done += 32;
        }
        const int left = qn - done;
        if (done > 0 && left > 0) {  // move the tail (< 32 entries) to the front of the queue
            int ti = 0, to = 0;
            if (lane < left) {
                ti = qi[done + lane];
                to = qo[done + lane];
            }
            __builtin_amdgcn_wave_barrier();
            if (lane < left) {
                qi[lane] = ti;
                qo[lane] = to;
            }
        }
        qn = left;
        __builtin_amdgcn_wave_barrier();
    }
    if (qn > 0) round32(0, qn, std::false_type{});
#if WGS_ACCS >= 2
#pragma unroll
    for (int m = 0; m < MTB; ++m)
#pragma unroll
        for (int n = 0; n < NTB; ++n) {
#if WGS_ACCS >= 3
            acc[m][n] = (acc_s[m][n] + acc_m[m][n]) + acc[m][n];
#else
            acc[m][n] = acc_s[m][n] + acc[m][n];
#endif
        }
#endif

    // fold the 4 waves of the block in fixed order 0+1+2+3, then the chunk's slab (as wgrad_kernel)
    for (int src = 1; src < TEAMS; ++src) {
        float* const rq = red + qd * (MTB * NTB * 4 * 64);
        if (team == src) {
#pragma unroll
            for (int m = 0; m < MTB; ++m)
#pragma unroll
                for (int n = 0; n < NTB; ++n)
#pragma unroll
                    for (int reg = 0; reg < 4; ++reg) rq[((m * NTB + n) * 4 + reg) * 64 + lane] = acc[m][n][reg];
        }
        __syncthreads();
        if (team == 0) {
#pragma unroll
            for (int m = 0; m < MTB; ++m)
#pragma unroll
                for (int n = 0; n < NTB; ++n)
#pragma unroll
                    for (int reg = 0; reg < 4; ++reg) acc[m][n][reg] += rq[((m * NTB + n) * 4 + reg) * 64 + lane];
        }
        __syncthreads();
    }
    if (team != 0) return;
    // D: col = lane & 15 -> produced-channel tile column, row = 4 (lane >> 4) + reg -> gathered-channel tile row
    float* dst = slab + (size_t)chunk * cout * K * cin;
#pragma unroll
    for (int m = 0; m < MTB; ++m)
#pragma unroll
        for (int n = 0; n < NTB; ++n)
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) {
                const int ci = MT * (4 * g + reg) + m0 + m;
                const int co = NT * ii + n0 + n;
                dst[((size_t)co * K + k) * cin + ci] = acc[m][n][reg];
            }
}

static inline bool wgrad_split_shape_ok(int cin, int cout) {
    return ((cin == 32 || cin == 64) && (cout == 32 || cout == 64)) || ((cin == 64 || cin == 128) && (cout == 64 || cout == 128));
}

// ---- host side ----------------------------------------------------------------------------------------------------------------------
static inline bool split_shape_ok(int c_gather, int c_produce) {
    return (c_gather == 32 || c_gather == 64 || c_gather == 128) && (c_produce == 32 || c_produce == 64 || c_produce == 128);
}
static inline size_t split_packed_bytes(int k_vol, int c_gather, int c_produce) {
    return (size_t)k_vol * (c_gather / 32) * (c_produce / 16) * 3 * 1024;
}

}  // namespace toda
