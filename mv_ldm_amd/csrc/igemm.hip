// Implicit-GEMM convolution / Linear on CDNA4 matrix cores (see include/mvldm.h: mvldm_igemm_fwd).
//
//   out[m][n] = epi( sum_k A[m][k] * W[n][k] + bias[n] + row_bias[img(m)][n] ) * scale + residual[m][n]
//
// m = (image, oy, ox) output pixel, k = (tap, channel).  A is never materialised: every 16-byte
// chunk of a K-tile row is fetched straight from the NHWC activation(s) -- zero for padding taps,
// from the second source for the skip-concat channels, from (iy>>1, ix>>1) for the fused nearest
// upsample.  W is the pre-packed K-major weight.  Both tiles are staged through LDS (double
// buffered, register prefetch of the next K-tile while MFMAs run on the current one) and consumed by
// v_mfma_f32_32x32x16_{bf16,f16} / v_mfma_f32_32x32x2_f32 with fp32 accumulation.
//
// LDS layout (16-bit types): 128-byte rows of 8 x 16-byte chunks, chunk index XOR-swizzled with
// (row>>1)&7 so that the 16-lane groups of a ds_read_b128 fragment read hit 16 distinct 16-byte
// slots of the 256-byte bank row (conflict-free); fp32: rows of 32 floats at pitch 33.
// Workgroup -> tile mapping is XCD-aware: each of the 8 XCDs owns a contiguous range of
// (split, n-tile, m-tile) ids with m fastest, so one weight tile is streamed from HBM by one XCD only.
#include <algorithm>
#include <cstdio>
#include <cstdlib>

#include "common.h"

namespace mvldm {

struct IgemmParams {
    const void* src0; const void* src1; const void* weight;
    const float* bias; const float* row_bias; const void* residual; void* dst; float* ws;
    int c0, c1, ctot;
    int n_img, h_in, w_in, h_out, w_out, hw_out;
    int ksize, stride, pad, upsample;
    int M, n_out, n_pad, n_dst, k_pad, taps;
    int row_bias_ld, epilogue, dst_f32, dst_ld;
    float out_scale;
    int splitk, k_tiles, k_tiles_per_split;
    int tiles_m, tiles_n;
    int korder;                 // 0: k = (tap, channel)   1: k = (channel block of BK, tap, channel in block)
    int px, sub_m, sub_n, m_fast;  // XCD-aware 2-D tile partition
    int grp_m, grp_n;              // > 1: inside an XCD's partition consecutive workgroups form grp_m x grp_n blocks of tiles (map_block)
    int rb_vec;                    // row_bias rows are 16-byte addressable (base and leading dimension)
    int bias_vec;                  // bias is 16-byte aligned
    int ty0, tx0, cy, cx;          // tap origin relative to (oy*stride, ox*stride), and the always-inside reference tap
    int scatter, ph_y, ph_x;       // sub-pixel phase of a decomposed nearest-2x upsampling conv: output row m -> (2i+py, 2j+px)
    int fake;                      // EXPERIMENT knob (MVLDM_IGEMM_FAKE): bit 2 = no global stores / residual loads, bit 3 = no epilogue
    unsigned src0_bytes, src1_bytes, w_bytes;   // buffer-descriptor extents (lean 16-bit loop)
    int use_bl, stage_epi;
    int nt_store;                  // staged epilogue: streaming (non-temporal) output stores
};

// ---- per-dtype MFMA + LDS policy -------------------------------------------------------------------
template <typename T> struct Mma;

template <typename T16, typename FragT> struct Mma16 {
    static constexpr int KI = 16;        // K per MFMA
    static constexpr int BK = 64;        // K per LDS tile
    static constexpr int PITCH = 128;    // bytes per LDS row
    using Frag = FragT;
    static __device__ __forceinline__ void store(char* tile, int r, int kc, u32x4 v) {
        *reinterpret_cast<u32x4*>(tile + r * PITCH + ((kc ^ ((r >> 1) & 7)) << 4)) = v;
    }
    static __device__ __forceinline__ Frag load(const char* tile, int r, int kk, int hi) {
        const int kc = kk * 2 + hi;
        return *reinterpret_cast<const Frag*>(tile + r * PITCH + ((kc ^ ((r >> 1) & 7)) << 4));
    }
};
template <> struct Mma<bf16_t> : Mma16<bf16_t, bf16x8> {
    static __device__ __forceinline__ f32x16 mma(Frag a, Frag b, f32x16 c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
    }
};
template <> struct Mma<f16_t> : Mma16<f16_t, f16x8> {
    static __device__ __forceinline__ f32x16 mma(Frag a, Frag b, f32x16 c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
    }
};
template <> struct Mma<float> {
    static constexpr int KI = 2;
    static constexpr int BK = 32;
    static constexpr int PITCH = 33 * 4;
    using Frag = float;
    static __device__ __forceinline__ void store(char* tile, int r, int kc, u32x4 v) {
        uint32_t* p = reinterpret_cast<uint32_t*>(tile + r * PITCH + kc * 16);
        p[0] = v[0]; p[1] = v[1]; p[2] = v[2]; p[3] = v[3];
    }
    static __device__ __forceinline__ Frag load(const char* tile, int r, int kk, int hi) {
        return *reinterpret_cast<const float*>(tile + r * PITCH + (kk * 2 + hi) * 4);
    }
    static __device__ __forceinline__ f32x16 mma(Frag a, Frag b, f32x16 c) {
        return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
    }
};

// packed weight row -> original output column (GEGLU rows alternate [value|gate] in blocks of 32)
__device__ __forceinline__ int orig_col(int n_packed, int n_out, bool geglu) {
    if (!geglu) return n_packed;
    const int blk = n_packed >> 5, w = n_packed & 31;
    return (blk & 1) ? (n_out >> 1) + (blk >> 1) * 32 + w : (blk >> 1) * 32 + w;
}

// SCATTER false: the per-element fallback epilogue inside the GEMM kernels -- phase convs never take it (fill_params), and with
// the two divisions in its 64-fold unrolled body hipcc stops unrolling and demotes the accumulators to scratch
template <typename T, bool SCATTER = false>
__device__ __forceinline__ void epilogue_store(const IgemmParams& p, int m, int n_dst_col, float v) {
    // v already includes bias/row_bias/activation
    v *= p.out_scale;
    if (p.residual) v += to_f32<T>(reinterpret_cast<const T*>(p.residual)[(size_t)m * p.n_dst + n_dst_col]);
    size_t drow = (size_t)m;
    if (SCATTER && p.scatter) {   // sub-pixel phase of a decomposed nearest-2x upsampling conv (the split-K reduce of a phase lands here)
        const int img = m / p.hw_out, rem = m - img * p.hw_out;
        const int i = rem / p.w_out, j = rem - i * p.w_out;
        drow = ((size_t)img * (2 * p.h_out) + 2 * i + p.ph_y) * (size_t)(2 * p.w_out) + 2 * j + p.ph_x;
    }
    const size_t o = drow * p.dst_ld + n_dst_col;
    if (p.dst_f32) reinterpret_cast<float*>(p.dst)[o] = v;
    else reinterpret_cast<T*>(p.dst)[o] = from_f32<T>(v);
}

// Workgroup -> (split, m-tile, n-tile).  The hardware places workgroup b on XCD b % 8 (observed; used for
// speed only): the 8 XCDs form a px x py grid over the tile space so that each XCD's private 4 MB L2
// sees one slice of A and one slice of W -- the host picks (px, py) minimising py*bytes(A) + px*bytes(W),
// the traffic that crosses the fabric.  Inside an XCD, tiles that share the larger operand are adjacent.
__device__ __forceinline__ bool map_block(const IgemmParams& p, int& split, int& tm, int& tn) {
    const int b = blockIdx.x, xcd = b & 7, idx = b >> 3;
    const int xm = xcd % p.px, xn = xcd / p.px;
    const int per = p.sub_m * p.sub_n;
    split = idx / per;
    const int r = idx - split * per;
    int tml, tnl;
    if (p.grp_m * p.grp_n > 1) {
        // The ~32 workgroups an XCD runs at the same time stream their operands in step: a tile row of A is fetched once for the grp_n
        // column tiles that share it, a W panel once for the grp_m row tiles -- fabric traffic of the XCD's partition ~
        // A * (sub_n / grp_n) + W * (sub_m / grp_m).  One row (or column) of 32 tiles re-reads the other operand once per tile:
        // measured 4.8x / 7.7x the algorithmic bytes on the level-1 / level-2 GEGLU projections (4.5 GB at 3.9 TB/s: bandwidth-bound).
        // Order: super-rows of grp_m row tiles; inside, chunks of grp_n columns; inside a chunk the row index runs fastest.
        const int gm = p.grp_m, gn = p.grp_n;
        const int nfull = p.sub_m / gm, per_super = gm * p.sub_n;
        int mg, gm_eff, rr;
        if (r < nfull * per_super) { mg = r / per_super; rr = r - mg * per_super; gm_eff = gm; }
        else { mg = nfull; rr = r - nfull * per_super; gm_eff = p.sub_m - nfull * gm; }
        const int ng = rr / (gm_eff * gn), r2 = rr - ng * gm_eff * gn;
        tnl = ng * gn + r2 / gm_eff;
        tml = mg * gm + r2 % gm_eff;
    } else if (p.m_fast) { tnl = r / p.sub_m; tml = r - tnl * p.sub_m; }
    else { tml = r / p.sub_n; tnl = r - tml * p.sub_n; }
    tm = xm * p.sub_m + tml;
    tn = xn * p.sub_n + tnl;
    return tm < p.tiles_m && tn < p.tiles_n;
}

// ---- epilogue shared by both main-loop variants ---------------------------------------------------
template <typename T, int BM, int BN, int WM, int WN>
__device__ __forceinline__ void igemm_epilogue(const IgemmParams& p, f32x16 (&acc)[BM / WM / 32][BN / WN / 32], int tm, int tn,
                                               int split, int wm, int wn, int hi, int l31) {
    constexpr int TM = BM / WM / 32, TN = BN / WN / 32;
    const bool geglu = p.epilogue == MVLDM_EPI_GEGLU;
    if (p.splitk > 1) {
        float* ws = p.ws + (size_t)split * p.M * p.n_pad;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int n = tn * BN + wn * (BN / WN) + j * 32 + l31;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int m = tm * BM + wm * (BM / WM) + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
                    if (m < p.M && n < p.n_pad) ws[(size_t)m * p.n_pad + n] = acc[i][j][r];
                }
            }
        return;
    }
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        if (geglu) {
            if constexpr (TN % 2 == 0) {
#pragma unroll
                for (int j = 0; j < TN; j += 2) {
                    const int nb = tn * BN + wn * (BN / WN) + j * 32;  // packed col of the value block
                    const int col = (nb >> 6) * 32 + l31;             // output column
                    if (col >= p.n_dst) continue;
                    const float bv = p.bias ? p.bias[col] : 0.f;
                    const float bg = p.bias ? p.bias[p.n_dst + col] : 0.f;
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int m = tm * BM + wm * (BM / WM) + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
                        if (m >= p.M) continue;
                        epilogue_store<T>(p, m, col, (acc[i][j][r] + bv) * gelu_erf_fast(acc[i][j + 1][r] + bg));
                    }
                }
            }
        } else {
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int n = tn * BN + wn * (BN / WN) + j * 32 + l31;
                if (n >= p.n_out) continue;
                const float bv = p.bias ? p.bias[n] : 0.f;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int m = tm * BM + wm * (BM / WM) + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
                    if (m >= p.M) continue;
                    float v = acc[i][j][r] + bv;
                    if (p.row_bias) v += p.row_bias[(size_t)(m / p.hw_out) * p.row_bias_ld + n];
                    if (p.epilogue == MVLDM_EPI_SILU) v = silu_f(v);
                    else if (p.epilogue == MVLDM_EPI_GELU) v = gelu_erf_fast(v);
                    epilogue_store<T>(p, m, n, v);
                }
            }
        }
    }
}

// ---- LDS-staged epilogue (16-bit loops) --------------------------------------------------------------
// The MFMA accumulator layout gives a lane ONE column and 16 scattered rows: storing from it means 2-byte
// stores in 64-byte runs (and the residual is read the same way).  Here every wave parks a finished 32-row
// block of its tile in LDS as RAW fp32 accumulators (16 ds_write_b32 per block, nothing else), then re-reads it
// row-major: a lane owns 8 consecutive output columns of one row, so bias, time-embedding row and residual
// arrive as 16/32-byte loads, the activation / GEGLU product runs on 8 values at a time, and one 16-byte store
// leaves; a store instruction covers 8 full 128-byte lines.  The epilogue mode is decided once per group,
// outside the element loops (the first version branched and waited on a bias load per accumulator block: PMC /
// `MVLDM_IGEMM_FAKE=8` showed the epilogue costing as much as the whole main loop at K = 320).  Split-K partial
// slabs take the same route with 16-byte fp32 stores.
// A wave tile wider than 4 column blocks is parked in groups of <= 4 blocks (the 8 park buffers must fit the ring).
constexpr int park_blocks(int tn) { return tn <= 4 ? tn : 4; }

enum { EPI_PLAIN = 0, EPI_ACT_SILU = 1, EPI_PAIR_GEGLU = 2, EPI_PARTIAL = 3, EPI_ACT_GELU = 4 };

// one parked group: JN column blocks of one 32-row block.  m0: global row of block row 0; pcol0: first packed
// column of the group.
template <typename T, int JN, int MODE, int PITCH>
__device__ __forceinline__ void epi_rows(const IgemmParams& p, const float* st, int m0, int pcol0, int split, int lane) {
    constexpr bool PAIR = MODE == EPI_PAIR_GEGLU;
    constexpr int WC = PAIR ? JN * 16 : JN * 32;      // output columns of the group
    constexpr int CPR = WC / 8;                       // 8-column chunks per row
    static_assert(64 % CPR == 0 && (32 * CPR) % 64 == 0, "row-major mapping");
    constexpr int RSTEP = 64 / CPR, ITERS = 32 / RSTEP;
    const int ch = lane % CPR, row0 = lane / CPR;     // a lane keeps its columns over the rows it visits
    const int ncol0 = PAIR ? (pcol0 >> 1) : pcol0;
    const int n0 = ncol0 + ch * 8;
    const int n_lim = MODE == EPI_PARTIAL ? p.n_pad : p.n_dst;
    if (n0 >= n_lim) return;
    // value (and, for GEGLU, gate) position of this lane's chunk inside a parked row
    const int voff = PAIR ? (2 * (ch >> 2)) * 32 + (ch & 3) * 8 : ch * 8;
    float bv[8], bg[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) bv[e] = bg[e] = 0.f;
    if (MODE != EPI_PARTIAL && p.bias) {
        if (p.bias_vec) {
            const f32x4 b0 = *reinterpret_cast<const f32x4*>(p.bias + n0), b1 = *reinterpret_cast<const f32x4*>(p.bias + n0 + 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) { bv[e] = b0[e]; bv[4 + e] = b1[e]; }
            if constexpr (PAIR) {
                const f32x4 g0 = *reinterpret_cast<const f32x4*>(p.bias + p.n_dst + n0);
                const f32x4 g1 = *reinterpret_cast<const f32x4*>(p.bias + p.n_dst + n0 + 4);
#pragma unroll
                for (int e = 0; e < 4; ++e) { bg[e] = g0[e]; bg[4 + e] = g1[e]; }
            }
        } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                bv[e] = p.bias[n0 + e];
                if constexpr (PAIR) bg[e] = p.bias[p.n_dst + n0 + e];
            }
        }
    }
    const bool rb_on = MODE != EPI_PARTIAL && MODE != EPI_PAIR_GEGLU && p.row_bias != nullptr;
    // The residual rows of ALL the lane's iterations are requested here, before the first store: inside the loop below every
    // load sits behind the previous iteration's store to `dst` (which the compiler must assume may alias it), i.e. one full
    // memory latency PER ITERATION -- measured on the level-0 output projection (K = 320, 256 x 320 tile): 25 us of epilogue
    // per tile against 13 us of main loop.  <= 8 chunks = 32 registers (the accumulators of the later row blocks are still live).
    Chunk<T> rpre[ITERS];
    if constexpr (MODE != EPI_PARTIAL) {
        if (p.residual && !(p.fake & 4)) {
#pragma unroll
            for (int it = 0; it < ITERS; ++it) {
                const int m = m0 + row0 + it * RSTEP;
                if (m < p.M) rpre[it] = load_chunk<T>(reinterpret_cast<const T*>(p.residual) + (size_t)m * p.n_dst + n0);
            }
        }
    }
    // ... and the time-embedding row when the whole 32-row block lies in one image (always, unless an image ends inside it)
    float rbv[8];
    bool rb_pre = false;
    if (rb_on) {
        const int img0 = m0 / p.hw_out;
        rb_pre = p.rb_vec && (min(m0 + 31, p.M - 1) / p.hw_out) == img0;
        if (rb_pre) {
            const float* rb = p.row_bias + (size_t)img0 * p.row_bias_ld + n0;
            const f32x4 r0 = *reinterpret_cast<const f32x4*>(rb), r1 = *reinterpret_cast<const f32x4*>(rb + 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) { rbv[e] = r0[e]; rbv[4 + e] = r1[e]; }
        }
    }
#pragma unroll
    for (int it = 0; it < ITERS; ++it) {
        const int row = row0 + it * RSTEP;
        const int m = m0 + row;
        if (m >= p.M) continue;
        const f32x4 a = *reinterpret_cast<const f32x4*>(st + row * PITCH + voff);
        const f32x4 b = *reinterpret_cast<const f32x4*>(st + row * PITCH + voff + 4);
        if constexpr (MODE == EPI_PARTIAL) {
            float* o = p.ws + (size_t)split * p.M * p.n_pad + (size_t)m * p.n_pad + n0;
            *reinterpret_cast<f32x4*>(o) = a;
            *reinterpret_cast<f32x4*>(o + 4) = b;
        } else {
            float v[8] = {a[0] + bv[0], a[1] + bv[1], a[2] + bv[2], a[3] + bv[3], b[0] + bv[4], b[1] + bv[5], b[2] + bv[6], b[3] + bv[7]};
            if constexpr (PAIR) {
                const f32x4 ga = *reinterpret_cast<const f32x4*>(st + row * PITCH + voff + 32);
                const f32x4 gb = *reinterpret_cast<const f32x4*>(st + row * PITCH + voff + 36);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    v[e] *= gelu_erf_16(ga[e] + bg[e]);
                    v[4 + e] *= gelu_erf_16(gb[e] + bg[4 + e]);
                }
            } else {
                if (rb_on && rb_pre) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] += rbv[e];
                } else if (rb_on) {   // per-image row (time embedding): one division per 8 outputs
                    const float* rb = p.row_bias + (size_t)(m / p.hw_out) * p.row_bias_ld + n0;
                    if (p.rb_vec) {
                        const f32x4 r0 = *reinterpret_cast<const f32x4*>(rb), r1 = *reinterpret_cast<const f32x4*>(rb + 4);
#pragma unroll
                        for (int e = 0; e < 4; ++e) { v[e] += r0[e]; v[4 + e] += r1[e]; }
                    } else {
#pragma unroll
                        for (int e = 0; e < 8; ++e) v[e] += rb[e];
                    }
                }
                if constexpr (MODE == EPI_ACT_SILU) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] = silu_f(v[e]);
                }
                if constexpr (MODE == EPI_ACT_GELU) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] = gelu_erf_fast(v[e]);
                }
            }
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] *= p.out_scale;
            if (p.fake & 4) { if (v[0] == 1.2345e33f) p.ws[0] = v[1]; continue; }
            if (p.residual) {
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] += rpre[it].get(e);
            }
            Chunk<T> oc;
#pragma unroll
            for (int e = 0; e < 8; ++e) oc.set(e, v[e]);
            size_t drow = (size_t)m;
            if (p.scatter) {   // sub-pixel phase: low-resolution pixel (i, j) of image `img` -> (2i+py, 2j+px) of the 2x output
                const int img = m / p.hw_out, rem = m - img * p.hw_out;
                const int i = rem / p.w_out, j = rem - i * p.w_out;
                drow = ((size_t)img * (2 * p.h_out) + 2 * i + p.ph_y) * (size_t)(2 * p.w_out) + 2 * j + p.ph_x;
            }
            T* const dptr = reinterpret_cast<T*>(p.dst) + drow * p.dst_ld + n0;
            if (p.nt_store) __builtin_nontemporal_store(oc.raw, reinterpret_cast<u32x4*>(dptr));
            else store_chunk<T>(dptr, oc);
        }
    }
}

// park JN column blocks (from J0) of row block I of the wave's accumulators: raw fp32, [row][col].  The accumulators are named by
// template indices, never through a reference to a sub-array: with five epilogue modes behind it hipcc otherwise stops promoting
// the 64 x 64 wave tile's `acc` to registers (320 bytes of scratch per lane, written and re-read once per tile: 3-4x slower)
template <int TM, int TN, int I, int J0, int JN, int PITCH>
__device__ __forceinline__ void epi_park(const f32x16 (&acc)[TM][TN], float* st, int lane) {
    const int hi = lane >> 5, l31 = lane & 31;
#pragma unroll
    for (int j = 0; j < JN; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) st[((r & 3) + 8 * (r >> 2) + 4 * hi) * PITCH + j * 32 + l31] = acc[I][J0 + j][r];
    // (same wave wrote and reads: LDS serves a wave's requests in order; the compiler's own lgkmcnt wait covers
    //  the data dependence through `st`)
}

template <typename T, int JN, int PITCH>
__device__ __forceinline__ void epi_group_rows(const IgemmParams& p, const float* st, int m0, int pcol0, int split, int mode, int lane) {
    if (mode == EPI_PARTIAL) epi_rows<T, JN, EPI_PARTIAL, PITCH>(p, st, m0, pcol0, split, lane);
    else if (mode == EPI_PAIR_GEGLU) {
        if constexpr (JN % 2 == 0) epi_rows<T, JN, EPI_PAIR_GEGLU, PITCH>(p, st, m0, pcol0, split, lane);
    } else if (mode == EPI_ACT_SILU) epi_rows<T, JN, EPI_ACT_SILU, PITCH>(p, st, m0, pcol0, split, lane);
    else if (mode == EPI_ACT_GELU) epi_rows<T, JN, EPI_ACT_GELU, PITCH>(p, st, m0, pcol0, split, lane);
    else epi_rows<T, JN, EPI_PLAIN, PITCH>(p, st, m0, pcol0, split, lane);
}

template <typename T, int BM, int BN, int WM, int WN, int I>
__device__ __forceinline__ void igemm_epilogue_rowblock(const IgemmParams& p, const f32x16 (&acc)[BM / WM / 32][BN / WN / 32], float* st, int tm, int tn,
                                                        int split, int wm, int wn, int mode, int lane) {
    constexpr int TM = BM / WM / 32, TN = BN / WN / 32;
    constexpr int WCOLS = BN / WN;          // packed columns of a wave tile
    constexpr int JG = park_blocks(TN);
    constexpr int PITCH = JG * 32 + 4;      // floats
    const int m0 = tm * BM + wm * (BM / WM) + I * 32;
    const int pcol0 = tn * BN + wn * WCOLS;
    epi_park<TM, TN, I, 0, JG, PITCH>(acc, st, lane);
    epi_group_rows<T, JG, PITCH>(p, st, m0, pcol0, split, mode, lane);
    if constexpr (TN > JG) {
        epi_park<TM, TN, I, JG, TN - JG, PITCH>(acc, st, lane);
        epi_group_rows<T, TN - JG, PITCH>(p, st, m0, pcol0 + JG * 32, split, mode, lane);
    }
    if constexpr (I + 1 < TM) igemm_epilogue_rowblock<T, BM, BN, WM, WN, I + 1>(p, acc, st, tm, tn, split, wm, wn, mode, lane);
}

template <typename T, int BM, int BN, int WM, int WN>
__device__ __forceinline__ void igemm_epilogue_staged(const IgemmParams& p, f32x16 (&acc)[BM / WM / 32][BN / WN / 32], int tm,
                                                      int tn, int split, int wm, int wn, int wave, int lane, char* smem) {
    constexpr int TN = BN / WN / 32;
    constexpr int JG = park_blocks(TN);
    constexpr int PITCH = JG * 32 + 4;      // floats
    static_assert(TN <= 2 * JG, "at most two park groups");
    if (p.fake & 8) return;
    float* st = reinterpret_cast<float*>(smem) + wave * (32 * PITCH);
    const int mode = p.splitk > 1 ? EPI_PARTIAL
                                  : (p.epilogue == MVLDM_EPI_GEGLU ? EPI_PAIR_GEGLU
                                     : (p.epilogue == MVLDM_EPI_SILU ? EPI_ACT_SILU : (p.epilogue == MVLDM_EPI_GELU ? EPI_ACT_GELU : EPI_PLAIN)));
    __syncthreads();   // every wave is done with the operand ring
    igemm_epilogue_rowblock<T, BM, BN, WM, WN, 0>(p, acc, st, tm, tn, split, wm, wn, mode, lane);
}

template <typename T, int BM, int BN, int WM, int WN>
__global__ __launch_bounds__(WM* WN * 64) void igemm_kernel(const IgemmParams p) {
    using M_ = Mma<T>;
    constexpr int NT = WM * WN * 64;
    constexpr int EPC = Elt<T>::EPC;
    constexpr int BK = M_::BK;
    constexpr int TM = BM / WM / 32, TN = BN / WN / 32;
    constexpr int A_BYTES = BM * M_::PITCH, B_BYTES = BN * M_::PITCH;
    constexpr int A_IT = BM * 8 / NT, B_IT = BN * 8 / NT;
    static_assert(A_IT >= 1 && B_IT >= 1 && TM >= 1 && TN >= 1, "bad tile");
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int hi = lane >> 5, l31 = lane & 31;

    int split, tm, tn;
    if (!map_block(p, split, tm, tn)) return;   // uniform per workgroup, before any barrier
    const int kt0 = split * p.k_tiles_per_split;
    const int kt1 = min(kt0 + p.k_tiles_per_split, p.k_tiles);

    // ---- loader coordinates (fixed per thread across the K loop) ----
    const int kc = tid & 7, r0 = tid >> 3;
    int a_img[A_IT], a_y[A_IT], a_x[A_IT];
#pragma unroll
    for (int it = 0; it < A_IT; ++it) {
        const int m = tm * BM + r0 + it * (NT / 8);
        if (m < p.M) {
            const int img = m / p.hw_out, rem = m - img * p.hw_out;
            const int oy = rem / p.w_out;
            a_img[it] = img;
            a_y[it] = oy * p.stride - p.pad;
            a_x[it] = (rem - oy * p.w_out) * p.stride - p.pad;
        } else {
            a_img[it] = -1; a_y[it] = 0; a_x[it] = 0;
        }
    }
    const T* wbase = reinterpret_cast<const T*>(p.weight) + (size_t)(tn * BN + r0) * p.k_pad + kc * EPC;
    const int hs = p.upsample ? 2 * p.h_in : p.h_in, wsz = p.upsample ? 2 * p.w_in : p.w_in;

    u32x4 areg[A_IT], breg[B_IT];
    auto load_tile = [&](int kt) {
        int tap, c;
        if (p.korder) {
            const int cb = kt / p.taps;
            tap = kt - cb * p.taps;
            c = cb * BK + kc * EPC;
        } else {
            const int ke = kt * BK + kc * EPC;
            tap = ke / p.ctot;
            c = ke - tap * p.ctot;
        }
        const bool tap_ok = tap < p.taps;
        const int ky = tap / p.ksize, kx = tap - ky * p.ksize;
        const bool from0 = c < p.c0;
        const T* sbase = from0 ? reinterpret_cast<const T*>(p.src0) + c
                               : reinterpret_cast<const T*>(p.src1) + (c - p.c0);
        const int cs = from0 ? p.c0 : p.c1;
#pragma unroll
        for (int it = 0; it < A_IT; ++it) {
            int iy = a_y[it] + ky, ix = a_x[it] + kx;
            const bool ok = tap_ok && a_img[it] >= 0 && iy >= 0 && iy < hs && ix >= 0 && ix < wsz;
            if (p.upsample) { iy >>= 1; ix >>= 1; }
            if (ok) {
                const size_t off = ((size_t)(a_img[it] * p.h_in + iy) * p.w_in + ix) * cs;
                areg[it] = *reinterpret_cast<const u32x4*>(sbase + off);
            } else {
                areg[it] = u32x4{0u, 0u, 0u, 0u};
            }
        }
#pragma unroll
        for (int it = 0; it < B_IT; ++it) {
            const int n = tn * BN + r0 + it * (NT / 8);
            if (n < p.n_pad)
                breg[it] = *reinterpret_cast<const u32x4*>(wbase + (size_t)it * (NT / 8) * p.k_pad + (size_t)kt * BK);
            else
                breg[it] = u32x4{0u, 0u, 0u, 0u};
        }
    };
    auto store_tile = [&](int stage) {
        char* at = smem + stage * (A_BYTES + B_BYTES);
        char* bt = at + A_BYTES;
#pragma unroll
        for (int it = 0; it < A_IT; ++it) M_::store(at, r0 + it * (NT / 8), kc, areg[it]);
#pragma unroll
        for (int it = 0; it < B_IT; ++it) M_::store(bt, r0 + it * (NT / 8), kc, breg[it]);
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    if (kt0 < kt1) {
        load_tile(kt0);
        store_tile(0);
    }
    __syncthreads();
    int cur = 0;
    for (int kt = kt0; kt < kt1; ++kt) {
        const bool more = kt + 1 < kt1;
        if (more) load_tile(kt + 1);
        const char* at = smem + cur * (A_BYTES + B_BYTES);
        const char* bt = at + A_BYTES;
#pragma unroll
        for (int kk = 0; kk < BK / M_::KI; ++kk) {
            typename M_::Frag a[TM], b[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) a[i] = M_::load(at, wm * (BM / WM) + i * 32 + l31, kk, hi);
#pragma unroll
            for (int j = 0; j < TN; ++j) b[j] = M_::load(bt, wn * (BN / WN) + j * 32 + l31, kk, hi);
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) acc[i][j] = M_::mma(a[i], b[j], acc[i][j]);
        }
        if (more) store_tile(cur ^ 1);
        __syncthreads();
        cur ^= 1;
    }

    // 16-bit problems that cannot take the lean loop (conv_in of the UNet and of the VAE encoder: 3 / 11 input channels) still write
    // 16-byte rows through the LDS-staged epilogue: the per-element form stores 2-byte values 64 B per row and instruction -- the VAE's
    // conv_in at 32 images of 256 x 256 (537 MB of output, 14 GFLOP) took 2.3 ms of a 68 ms training step, 0.23 TB/s
    if constexpr (sizeof(T) == 2) {
        if (p.stage_epi) {
            igemm_epilogue_staged<T, BM, BN, WM, WN>(p, acc, tm, tn, split, wm, wn, wave, lane, smem);
            return;
        }
    }
    igemm_epilogue<T, BM, BN, WM, WN>(p, acc, tm, tn, split, wm, wn, hi, l31);
}

// ---- 16-bit main loop, lean form: buffer-load LDS-DMA, unrolled taps -----------------------------------
// For the block-major K order every K-tile is (64-channel block cb, tap): the tap loop is unrolled, so each
// lane's pixel offset for each tap is a REGISTER computed once per workgroup (out-of-image taps and rows
// beyond M hold an out-of-range offset: the buffer descriptor's bounds check returns zeros, which the DMA
// writes to LDS -- no zero page, no select).  Inside the loop a tile costs per wave: A_IT + B_IT
// `buffer_load_dwordx4 ... lds` with a scalar soffset (channel block / K position), the M0 updates, the
// fragment ds_reads and the MFMAs -- no vector address arithmetic at all (PMC of the previous loop: 11
// VALU + 16 SALU instructions per MFMA).
constexpr unsigned kOob = 0xFFFFFFF0u;

// Per-lane source addressing of the A pieces.  Without upsampling every tap of a pixel is the centre tap's
// byte offset plus a displacement that is the same for all lanes, so a lane keeps ONE offset per piece and
// a 9-bit validity mask; the displacement rides in the scalar offset of the buffer load (the descriptor's
// base is moved back by one row + one pixel so that it is never negative).  Nearest-2x upsampling makes the
// displacement depend on the parity of the lane's pixel: those (few) launches keep a per-tap table.
template <int TAPS, int A_IT, bool DUAL, bool UPS> struct BlAddr {
    unsigned a0[UPS ? TAPS : 1][A_IT];
    unsigned a1[DUAL ? (UPS ? TAPS : 1) : 1][DUAL ? A_IT : 1];
    unsigned mask[UPS ? 1 : A_IT];
};

// issue K-tile (channel block cb, tap t) into the ring slot at `stage_base`
// (STAGES is carried only to give every kernel instantiation its own copy: sharing one specialization
//  between two kernels trips the host pass of hipcc 7.2)
// (LO, HI: the pieces [LO, HI) of the tile's A_IT + B_IT, activation pieces first -- the spread issue of the main loops)
template <typename T, int BM, int BN, int NW, int KS, bool DUAL, int A_IT, int B_IT, int STAGES, bool UPS, int t, int LO = 0, int HI = 1 << 20>
__device__ __forceinline__ void bl_issue(const IgemmParams& p, char* stage_base, int wave, int cb,
                                         const BlAddr<KS * KS, A_IT, DUAL, UPS>& ad, const unsigned (&vb)[B_IT]) {
    constexpr int BK = 64, TAPS = KS * KS;
    const int lead = (!UPS && KS > 1) ? p.w_in + 1 : 0;                                         // pixels
    const int disp = UPS ? 0 : (t / KS - p.cy) * p.w_in + (t % KS - p.cx) + lead;               // >= 0
    const unsigned lead0 = (unsigned)lead * (unsigned)p.c0 * 2u, lead1 = (unsigned)lead * (unsigned)p.c1 * 2u;
    const __amdgpu_buffer_rsrc_t r0 = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<char*>(reinterpret_cast<const char*>(p.src0)) - lead0, 0, p.src0_bytes + lead0, 0x00020000);
    const __amdgpu_buffer_rsrc_t r1 = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<char*>(reinterpret_cast<const char*>(DUAL ? p.src1 : p.src0)) - (DUAL ? lead1 : lead0), 0,
        DUAL ? p.src1_bytes + lead1 : p.src0_bytes + lead0, 0x00020000);
    const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.weight), 0, p.w_bytes, 0x00020000);
    char* at = stage_base;
    char* bt = at + BM * 128;
    const int c = cb * BK;
    const bool from0 = !DUAL || c < p.c0;
    const int soff = ((from0 ? c : c - p.c0) + disp * (from0 ? p.c0 : p.c1)) * 2;
#pragma unroll
    for (int it = (LO > 0 ? LO : 0); it < (HI < A_IT ? HI : A_IT); ++it) {
        __attribute__((address_space(3))) void* dst = (__attribute__((address_space(3))) void*)(at + (wave + NW * it) * 1024);
        unsigned v0, v1;
        if constexpr (UPS) {
            v0 = ad.a0[t][it];
            v1 = ad.a1[DUAL ? t : 0][DUAL ? it : 0];
        } else {
            const bool ok = (ad.mask[it] >> t) & 1u;
            v0 = ok ? ad.a0[0][it] : kOob;
            v1 = ok ? ad.a1[0][DUAL ? it : 0] : kOob;
        }
        if (from0) __builtin_amdgcn_raw_ptr_buffer_load_lds(r0, dst, 16, v0, soff, 0, 0);
        else __builtin_amdgcn_raw_ptr_buffer_load_lds(r1, dst, 16, v1, soff, 0, 0);
    }
    const int koff = (cb * TAPS + t) * (BK * 2);
#pragma unroll
    for (int it = (LO > A_IT ? LO - A_IT : 0); it < (HI - A_IT < B_IT ? HI - A_IT : B_IT); ++it)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (__attribute__((address_space(3))) void*)(bt + (wave + NW * it) * 1024), 16,
                                                 vb[it], koff, 0, 0);
}

// One k-sub-step (16 of the tile's 64 K values) of operand fragments, and the MFMAs that consume them.  The
// main loop keeps TWO of these live and always has the next one's ds_reads in flight while the current
// one's MFMAs run -- including across the ring barrier (the first fragments of tile t+1 are fetched under
// the last MFMAs of tile t), so one wave alone covers the LDS latency instead of leaning on occupancy.
template <typename T, int TM, int TN> struct BlFrags {
    typename Mma<T>::Frag a[TM], b[TN];
};

template <typename T, int BM, int BN, int WM, int WN>
__device__ __forceinline__ void bl_load(const char* stage_base, BlFrags<T, BM / WM / 32, BN / WN / 32>& f, int kk, int wm, int wn,
                                        int hi, int l31) {
    using M_ = Mma<T>;
    constexpr int TM = BM / WM / 32, TN = BN / WN / 32;
    const char* at = stage_base;
    const char* bt = at + BM * 128;
#pragma unroll
    for (int i = 0; i < TM; ++i) f.a[i] = M_::load(at, wm * (BM / WM) + i * 32 + l31, kk, hi);
#pragma unroll
    for (int j = 0; j < TN; ++j) f.b[j] = M_::load(bt, wn * (BN / WN) + j * 32 + l31, kk, hi);
}

template <typename T, int TM, int TN>
__device__ __forceinline__ void bl_mma(const BlFrags<T, TM, TN>& f, f32x16 (&acc)[TM][TN]) {
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = Mma<T>::mma(f.a[i], f.b[j], acc[i][j]);
}

// whole K-tile, fragments fetched right before use: the 4-wave tiles run 2-4 workgroups per CU and hide the LDS
// latency with occupancy (the pipelined form above costs them registers and measured 10-20 % slower)
template <typename T, int BM, int BN, int WM, int WN>
__device__ __forceinline__ void bl_compute(const char* stage_base, f32x16 (&acc)[BM / WM / 32][BN / WN / 32], int wm, int wn,
                                           int hi, int l31) {
    constexpr int TM = BM / WM / 32, TN = BN / WN / 32;
#pragma unroll
    for (int kk = 0; kk < 64 / Mma<T>::KI; ++kk) {
        BlFrags<T, TM, TN> f;
        bl_load<T, BM, BN, WM, WN>(stage_base, f, kk, wm, wn, hi, l31);
        bl_mma<T, TM, TN>(f, acc);
    }
}

// s_waitcnt vmcnt(min(young, MAXY) * LPT) lgkmcnt(0): `young` tiles of LPT loads per wave may stay in flight behind the awaited one
template <int LPT, int MAXY>
__device__ __forceinline__ void bl_wait_young(int young) {
    if constexpr (MAXY >= 1) {
        if (young >= MAXY) {
            asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(MAXY * LPT) : "memory");
            return;
        }
        bl_wait_young<LPT, MAXY - 1>(young);
    } else {
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    }
}

template <typename T, int BM, int BN, int WM, int WN, int KS, bool DUAL, int STAGES, bool UPS>
__global__ __launch_bounds__(WM* WN * 64) void igemm_bl_kernel(const IgemmParams p) {
    using M_ = Mma<T>;
    static_assert(sizeof(T) == 2, "16-bit activation types only");
    constexpr int NW = WM * WN, TAPS = KS * KS;
    constexpr int BK = 64, EPC = 8;
    constexpr int TM = BM / WM / 32, TN = BN / WN / 32;
    constexpr int A_BYTES = BM * 128, B_BYTES = BN * 128, STAGE_BYTES = A_BYTES + B_BYTES;
    constexpr int A_IT = BM / 8 / NW, B_IT = BN / 8 / NW, LPT = A_IT + B_IT;
    static_assert(A_IT >= 1 && B_IT >= 1 && (BM / 8) % NW == 0 && (BN / 8) % NW == 0, "bad tile");
    static_assert(STAGES >= 2 && STAGES <= 8, "ring depth");
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int hi = lane >> 5, l31 = lane & 31;
    int split, tm, tn;
    if (!map_block(p, split, tm, tn)) return;
    // split-K partitions channel blocks (k_tiles_per_split is a multiple of TAPS for this kernel)
    const int cb0 = split * (p.k_tiles_per_split / TAPS);
    const int cb1 = min(cb0 + p.k_tiles_per_split / TAPS, p.k_tiles / TAPS);

    const int slot = lane & 7, rsub = lane >> 3;
    BlAddr<TAPS, A_IT, DUAL, UPS> ad;
#pragma unroll
    for (int it = 0; it < A_IT; ++it) {
        const int row = (wave + NW * it) * 8 + rsub;
        const int m = tm * BM + row;
        const unsigned chunk = (unsigned)((slot ^ ((row >> 1) & 7)) * EPC);
        const bool live = m < p.M;
        const int img = live ? m / p.hw_out : 0, rem = live ? m - img * p.hw_out : 0;
        const int oy = rem / p.w_out, ox = rem - oy * p.w_out;
        if constexpr (UPS) {
            const unsigned hs = 2 * p.h_in, wsz = 2 * p.w_in;
            const int y0 = live ? oy * p.stride - p.pad : -(1 << 20), x0 = ox * p.stride - p.pad;
#pragma unroll
            for (int t = 0; t < TAPS; ++t) {
                const int iy = y0 + t / KS, ix = x0 + t % KS;
                const bool ok = (unsigned)iy < hs && (unsigned)ix < wsz;
                const unsigned pix = (unsigned)(img * p.h_in + (iy >> 1)) * (unsigned)p.w_in + (unsigned)(ix >> 1);
                ad.a0[t][it] = ok ? (pix * (unsigned)p.c0 + chunk) * 2u : kOob;
                if constexpr (DUAL) ad.a1[t][it] = ok ? (pix * (unsigned)p.c1 + chunk) * 2u : kOob;
            }
        } else {
            // centre tap (inside the image for every live row: checked on the host)
            const int yc = oy * p.stride + p.ty0 + p.cy, xc = ox * p.stride + p.tx0 + p.cx;
            const unsigned pix = (unsigned)(img * p.h_in + yc) * (unsigned)p.w_in + (unsigned)xc;
            unsigned msk = 0;
#pragma unroll
            for (int t = 0; t < TAPS; ++t) {
                const int iy = yc + t / KS - p.cy, ix = xc + t % KS - p.cx;
                msk |= (live && (unsigned)iy < (unsigned)p.h_in && (unsigned)ix < (unsigned)p.w_in) ? (1u << t) : 0u;
            }
            ad.a0[0][it] = (pix * (unsigned)p.c0 + chunk) * 2u;
            if constexpr (DUAL) ad.a1[0][it] = (pix * (unsigned)p.c1 + chunk) * 2u;
            ad.mask[it] = msk;
        }
    }
    unsigned vb[B_IT];
#pragma unroll
    for (int it = 0; it < B_IT; ++it) {
        const int row = (wave + NW * it) * 8 + rsub;
        const int n = tn * BN + row;
        const unsigned chunk = (unsigned)((slot ^ ((row >> 1) & 7)) * EPC);
        vb[it] = n < p.n_pad ? ((unsigned)n * (unsigned)p.k_pad + chunk) * 2u : kOob;
    }

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // (no lambdas around the buffer builtins: an opaque __amdgpu_buffer_rsrc_t inside a lambda makes the
    //  host pass drop the kernel's stub -- free function templates instead)
#define MVLDM_BL_ISSUE(stage_, cb_, t_) \
    bl_issue<T, BM, BN, NW, KS, DUAL, A_IT, B_IT, STAGES, UPS, t_>(p, smem + (stage_) * STAGE_BYTES, wave, cb_, ad, vb)
#define MVLDM_BL_ISSUE_R(stage_, cb_, t_, lo_, hi_) \
    bl_issue<T, BM, BN, NW, KS, DUAL, A_IT, B_IT, STAGES, UPS, t_, lo_, hi_>(p, smem + (stage_) * STAGE_BYTES, wave, cb_, ad, vb)
#define MVLDM_BL_NEXT(t_, d_) (((t_) + (d_)) % TAPS)
#define MVLDM_BL_LOAD(f_, slot_, kk_) bl_load<T, BM, BN, WM, WN>(smem + (slot_) * STAGE_BYTES, f_, kk_, wm, wn, hi, l31)
#define MVLDM_BL_MMA(f_)                  \
    __builtin_amdgcn_sched_barrier(0);    \
    bl_mma<T, TM, TN>(f_, acc);           \
    __builtin_amdgcn_sched_barrier(0);
    // One K-tile.  On entry f0 holds (in flight) the kk=0 fragments of the tile in slot_c and tiles
    // T+1 .. T+STAGES-1 are in the ring.  After the last fragments of tile T are read, every wave waits for
    // its pieces of tile T+1, the barrier publishes them and retires slot_c, which is refilled with tile
    // T+STAGES at once; the kk=0 fragments of tile T+1 are then fetched under tile T's last MFMAs.
    // (round 6: the spread issue of MVLDM_BL_STEP_SIMPLE below was built here too -- 4/9 of the pieces behind the barrier, the rest in front of sub-steps
    //  1 and 2 of the next step -- and is neutral in the step's op table: up1 / down2 +1 ... +1.5 %, up2 / down1 -1 ... -1.5 %; not kept)
#define MVLDM_BL_STEP(t_)                                                                                         \
    {                                                                                                             \
        static_assert(64 / M_::KI == 4, "four k-sub-steps per K-tile");                                          \
        MVLDM_BL_LOAD(f1, slot_c, 1);                                                                             \
        MVLDM_BL_MMA(f0)                                                                                          \
        MVLDM_BL_LOAD(f0, slot_c, 2);                                                                             \
        MVLDM_BL_MMA(f1)                                                                                          \
        MVLDM_BL_LOAD(f1, slot_c, 3);                                                                             \
        MVLDM_BL_MMA(f0)                                                                                          \
        /* tile T+1 must have landed: behind it only tile T+2 can be in flight (3-deep ring, and only if it */    \
        /* exists -- nothing is issued past the end of K, so the tail drains with vmcnt(0)) */                     \
        if (STAGES == 3 && cb + ((t_) + 2) / TAPS < cb1) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(LPT) : "memory"); \
        else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");                                          \
        __builtin_amdgcn_s_barrier();                                                                             \
        {                                                                                                         \
            /* (no constexpr locals as template arguments: the host pass rejects them inside a kernel) */       \
            const int cbn_ = cb + ((t_) + STAGES) / TAPS;                                                         \
            if (cbn_ < cb1) { MVLDM_BL_ISSUE(slot_c, cbn_, MVLDM_BL_NEXT(t_, STAGES)); }                          \
        }                                                                                                         \
        slot_c = slot_c + 1 == STAGES ? 0 : slot_c + 1;                                                           \
        MVLDM_BL_LOAD(f0, slot_c, 0);                                                                             \
        MVLDM_BL_MMA(f1)                                                                                          \
    }
    // 4-wave tiles: 2-slot ring, one barrier per tile, fragments fetched right before use
    // (round 6, SPREAD ISSUE: the next tile's LPT pieces used to go out in one burst behind the barrier -- every wave of the CU in the address path at
    //  once, ~ 70 cycles per piece with the matrix pipe idle (the same finding as tile 13's).  Now 4/9 of the pieces go out behind the
    //  barrier and the rest in front of sub-steps 1 and 2 (in front of their fragment reads: with the fragments live next to the piece offsets the
    //  256 x 320 3x3 kernel spilled); the last piece still has two sub-steps of MFMAs in front of the wait that needs it.  8-wave tiles only (tile 10: the 256 x 320
    //  convs of the 32 x 32 level, -3 % in the step's op table).  -DMVLDM_BL_BURST: the old order, A/B.)
#ifdef MVLDM_BL_BURST
    constexpr bool SPREAD_S = false;
#elif defined(MVLDM_BL_SPREAD_ALL)
    constexpr bool SPREAD_S = true;         // (experiment: the 4-wave tiles too)
#else
    constexpr bool SPREAD_S = NW == 8;      // (the 4-wave tiles run 2 - 4 workgroups per CU whose bursts already interleave: with it, -DMVLDM_BL_SPREAD_ALL, the 1 / 4-scene steps are 0.6 - 1 % slower)
#endif
#define MVLDM_BL_STEP_SIMPLE(t_)                                                                                  \
    {                                                                                                             \
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");                                               \
        __builtin_amdgcn_s_barrier();                                                                             \
        if constexpr (!SPREAD_S) {                                                                                \
            const int cbn_ = cb + ((t_) + 1) / TAPS;                                                              \
            if (cbn_ < cb1) { MVLDM_BL_ISSUE(slot_c ^ 1, cbn_, MVLDM_BL_NEXT(t_, 1)); }                           \
            bl_compute<T, BM, BN, WM, WN>(smem + slot_c * STAGE_BYTES, acc, wm, wn, hi, l31);                     \
        } else {                                                                                                  \
            constexpr int Q0_ = (4 * LPT + 8) / 9, Q1_ = Q0_ + (LPT - Q0_ + 1) / 2;                              \
            const int cbn_ = cb + ((t_) + 1) / TAPS;                                                              \
            const bool more_ = cbn_ < cb1;                                                                        \
            if (more_) { MVLDM_BL_ISSUE_R(slot_c ^ 1, cbn_, MVLDM_BL_NEXT(t_, 1), 0, Q0_); }                      \
            {                                                                                                     \
                BlFrags<T, TM, TN> fs_;                                                                           \
                MVLDM_BL_LOAD(fs_, slot_c, 0);                                                                    \
                bl_mma<T, TM, TN>(fs_, acc);                                                                      \
            }                                                                                                     \
            if (more_) { MVLDM_BL_ISSUE_R(slot_c ^ 1, cbn_, MVLDM_BL_NEXT(t_, 1), Q0_, Q1_); }                    \
            {                                                                                                     \
                BlFrags<T, TM, TN> fs_;                                                                           \
                MVLDM_BL_LOAD(fs_, slot_c, 1);                                                                    \
                bl_mma<T, TM, TN>(fs_, acc);                                                                      \
            }                                                                                                     \
            if (more_) { MVLDM_BL_ISSUE_R(slot_c ^ 1, cbn_, MVLDM_BL_NEXT(t_, 1), Q1_, LPT); }                    \
            {                                                                                                     \
                BlFrags<T, TM, TN> fs_;                                                                           \
                MVLDM_BL_LOAD(fs_, slot_c, 2);                                                                    \
                bl_mma<T, TM, TN>(fs_, acc);                                                                      \
                MVLDM_BL_LOAD(fs_, slot_c, 3);                                                                    \
                bl_mma<T, TM, TN>(fs_, acc);                                                                      \
            }                                                                                                     \
        }                                                                                                         \
        slot_c ^= 1;                                                                                              \
    }
    // Deep ring (STAGES >= 4, the small-launch tiles 16 - 18): tiles t+1 .. t+STAGES-1 are in flight while tile t is consumed.  With a
    // few hundred output rows a K-tile is a handful of MFMAs, so a step of the 2-slot loop costs one exposed L2 / HBM round trip
    // (the 4x4-level convs of one scene: 18 steps x ~0.8 us for 30 MB of weights); here the round trip is shared by STAGES - 1 steps.
#define MVLDM_BL_STEP_DEEP(t_)                                                                                    \
    {                                                                                                             \
        bl_wait_young<LPT, STAGES - 2>((cb1 - cb) * TAPS - (t_) - 1);      /* tile t has landed (younger ones stay in flight) */ \
        __builtin_amdgcn_s_barrier();                                      /* ... for every wave, and tile t-1's slot is free */ \
        {                                                                                                         \
            const int cbn_ = cb + ((t_) + STAGES - 1) / TAPS;                                                     \
            if (cbn_ < cb1) { MVLDM_BL_ISSUE(slot_c == 0 ? STAGES - 1 : slot_c - 1, cbn_, MVLDM_BL_NEXT(t_, STAGES - 1)); } \
        }                                                                                                         \
        bl_compute<T, BM, BN, WM, WN>(smem + slot_c * STAGE_BYTES, acc, wm, wn, hi, l31);                         \
        slot_c = slot_c + 1 == STAGES ? 0 : slot_c + 1;                                                           \
    }
#define MVLDM_BL_PRO(j_)                                                                                          \
    if constexpr (STAGES - 1 > (j_)) {                                                                            \
        if (cb0 + (j_) / TAPS < cb1) { MVLDM_BL_ISSUE((j_), cb0 + (j_) / TAPS, ((j_) % TAPS)); }                  \
    }
    // (the pipelined form needs 2 x (TM + TN) fragments next to the accumulators: not with 10 accumulator blocks)
    // (round 5 re-tried it for tile 10 with the per-tap offsets kept out of registers: the 1x1 form fits -- and measures +-0 on every Linear --,
    //  the 3x3 form still spills 52 B per lane into the loop: 870 -> 993 us)
    constexpr bool PIPE = NW == 8 && TM * TN <= 8 && STAGES <= 3;
    if constexpr (STAGES > 3) {
        static_assert((STAGES - 2) * LPT <= 63, "vmcnt is a 6-bit counter");
        if (cb0 < cb1) {
            MVLDM_BL_ISSUE(0, cb0, 0);
            MVLDM_BL_PRO(1) MVLDM_BL_PRO(2) MVLDM_BL_PRO(3) MVLDM_BL_PRO(4) MVLDM_BL_PRO(5) MVLDM_BL_PRO(6)
            int slot_c = 0;
            for (int cb = cb0; cb < cb1; ++cb) {
                MVLDM_BL_STEP_DEEP(0)
                if constexpr (TAPS == 4) { MVLDM_BL_STEP_DEEP(1) MVLDM_BL_STEP_DEEP(2) MVLDM_BL_STEP_DEEP(3) }
                if constexpr (TAPS == 9) {
                    MVLDM_BL_STEP_DEEP(1) MVLDM_BL_STEP_DEEP(2) MVLDM_BL_STEP_DEEP(3) MVLDM_BL_STEP_DEEP(4)
                    MVLDM_BL_STEP_DEEP(5) MVLDM_BL_STEP_DEEP(6) MVLDM_BL_STEP_DEEP(7) MVLDM_BL_STEP_DEEP(8)
                }
            }
        }
    } else if constexpr (!PIPE) {
        static_assert(STAGES == 2, "the plain loop uses the 2-slot ring");
        if (cb0 < cb1) {
            MVLDM_BL_ISSUE(0, cb0, 0);
            int slot_c = 0;
            for (int cb = cb0; cb < cb1; ++cb) {
                MVLDM_BL_STEP_SIMPLE(0)
                if constexpr (TAPS == 4) { MVLDM_BL_STEP_SIMPLE(1) MVLDM_BL_STEP_SIMPLE(2) MVLDM_BL_STEP_SIMPLE(3) }
                if constexpr (TAPS == 9) {
                    MVLDM_BL_STEP_SIMPLE(1) MVLDM_BL_STEP_SIMPLE(2) MVLDM_BL_STEP_SIMPLE(3) MVLDM_BL_STEP_SIMPLE(4)
                    MVLDM_BL_STEP_SIMPLE(5) MVLDM_BL_STEP_SIMPLE(6) MVLDM_BL_STEP_SIMPLE(7) MVLDM_BL_STEP_SIMPLE(8)
                }
            }
        }
    } else if (cb0 < cb1) {
        // prologue: fill the whole ring (up to STAGES tiles in flight), wait for the first
        MVLDM_BL_ISSUE(0, cb0, 0);
        const bool has1 = cb0 + 1 / TAPS < cb1, has2 = STAGES == 3 && cb0 + 2 / TAPS < cb1;
        if (has1) { MVLDM_BL_ISSUE(1, cb0 + 1 / TAPS, (1 % TAPS)); }
        if constexpr (STAGES == 3) {
            if (has2) { MVLDM_BL_ISSUE(2, cb0 + 2 / TAPS, (2 % TAPS)); }
        }
        if (has2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * LPT) : "memory");
        else if (has1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(LPT) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        int slot_c = 0;
        BlFrags<T, TM, TN> f0, f1;
        MVLDM_BL_LOAD(f0, 0, 0);
        for (int cb = cb0; cb < cb1; ++cb) {
            MVLDM_BL_STEP(0)
            if constexpr (TAPS == 4) { MVLDM_BL_STEP(1) MVLDM_BL_STEP(2) MVLDM_BL_STEP(3) }
            if constexpr (TAPS == 9) {
                MVLDM_BL_STEP(1) MVLDM_BL_STEP(2) MVLDM_BL_STEP(3) MVLDM_BL_STEP(4)
                MVLDM_BL_STEP(5) MVLDM_BL_STEP(6) MVLDM_BL_STEP(7) MVLDM_BL_STEP(8)
            }
        }
    }
#undef MVLDM_BL_STEP_SIMPLE
#undef MVLDM_BL_STEP_DEEP
#undef MVLDM_BL_PRO
#undef MVLDM_BL_LOAD
#undef MVLDM_BL_MMA
#undef MVLDM_BL_NEXT
#undef MVLDM_BL_STEP
#undef MVLDM_BL_ISSUE
#undef MVLDM_BL_ISSUE_R
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if constexpr (TM * TN > 4) {
        // (the per-element fallback does not unroll at 8 accumulator blocks and would push them to scratch:
        //  the host only picks such a tile when the staged epilogue applies)
        igemm_epilogue_staged<T, BM, BN, WM, WN>(p, acc, tm, tn, split, wm, wn, wave, lane, smem);
    } else {
        if (p.stage_epi) igemm_epilogue_staged<T, BM, BN, WM, WN>(p, acc, tm, tn, split, wm, wn, wave, lane, smem);
        else igemm_epilogue<T, BM, BN, WM, WN>(p, acc, tm, tn, split, wm, wn, hi, l31);
    }
}

// (buffer descriptors live in free functions, never in a kernel body: see the note at igemm_bl_kernel)
template <bool DUAL>
__device__ __forceinline__ void halo_issue_a(const IgemmParams& p, char* dst, unsigned v0, unsigned v1, int cb) {
    const __amdgpu_buffer_rsrc_t r0 = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.src0), 0, p.src0_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t r1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(DUAL ? p.src1 : p.src0), 0,
                                                                         DUAL ? p.src1_bytes : p.src0_bytes, 0x00020000);
    const int c = cb * 64;
    const bool from0 = !DUAL || c < p.c0;
    const int soff = (from0 ? c : c - p.c0) * 2;
    if (from0) __builtin_amdgcn_raw_ptr_buffer_load_lds(r0, (__attribute__((address_space(3))) void*)dst, 16, v0, soff, 0, 0);
    else __builtin_amdgcn_raw_ptr_buffer_load_lds(r1, (__attribute__((address_space(3))) void*)dst, 16, v1, soff, 0, 0);
}
__device__ __forceinline__ void halo_issue_w(const IgemmParams& p, char* dst, unsigned v, int koff) {
    const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.weight), 0, p.w_bytes, 0x00020000);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (__attribute__((address_space(3))) void*)dst, 16, v, koff, 0, 0);
}

// ---- 3x3 stride-1 convolution with an LDS-resident pixel halo ------------------------------------------------
// The 9 taps of a 3x3 conv read the same 64-channel slice of the same pixels, shifted by dy*W + dx rows of the
// NHWC pixel array.  Instead of fetching a shifted 256-row A tile per tap (9 x 32 KB per channel block through the
// L2 -> LDS path, which bounds the loop above), this kernel fetches ONE contiguous range of
// 256 + 2*(W+1) pixel rows per channel block (the tile's pixels plus W+1 rows of halo on either side) and serves all
// 9 taps from it: tap (dy,dx) of tile row r is halo row r + (W+1) + dy*W + dx -- a lane-uniform displacement.
// Image borders are per-lane 9-bit masks; a masked lane reads a 128-byte row of zeros.  Only the W tiles (16 KB per
// tap) still stream per K-tile, through a 3-slot ring; the halo of the next channel block arrives piecewise under
// the 9 taps of the current one (2 slots).  L2 -> LDS bytes per channel block: 41 + 9*16 = 185 KB instead of 432 KB.
template <typename T, bool DUAL>
__global__ __launch_bounds__(512) void igemm_halo_kernel(const IgemmParams p, int halo_rows) {
    using M_ = Mma<T>;
    constexpr int BM = 256, BN = 128, WM = 4, WN = 2, NW = 8, TM = 2, TN = 2, B_IT = 2, KA = 6, EPC = 8;
    constexpr int W_BYTES = BN * 128;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int a_bytes = halo_rows * 128;
    char* const wring = smem + 2 * a_bytes;
    char* const zrow = wring + 3 * W_BYTES;
    char* const dummy = zrow + 128;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int hi = lane >> 5, l31 = lane & 31;
    int split, tm, tn;
    if (!map_block(p, split, tm, tn)) return;
    const int cb1 = p.k_tiles / 9;
    const int lead = p.w_in + 1;
    const int m0 = tm * BM;
    const int np = halo_rows / 8;                       // 1 KiB pieces of a halo tile
    const int slot = lane & 7, rsub = lane >> 3;
    const int m_tot = p.n_img * p.h_in * p.w_in;

    if (tid < 8) *reinterpret_cast<u32x4*>(zrow + tid * 16) = u32x4{0u, 0u, 0u, 0u};

    // halo pieces of this wave: q = wave + 8k
    unsigned off0[KA], off1[DUAL ? KA : 1];
#pragma unroll
    for (int k = 0; k < KA; ++k) {
        const int q = wave + NW * k;
        const int hr = q * 8 + rsub;
        const int pm = m0 - lead + hr;
        const bool ok = q < np && pm >= 0 && pm < m_tot;
        const unsigned chunk = (unsigned)((slot ^ ((hr >> 1) & 7)) * EPC);
        off0[k] = ok ? ((unsigned)pm * (unsigned)p.c0 + chunk) * 2u : kOob;
        if constexpr (DUAL) off1[k] = ok ? ((unsigned)pm * (unsigned)p.c1 + chunk) * 2u : kOob;
    }
    unsigned vb[B_IT];
#pragma unroll
    for (int it = 0; it < B_IT; ++it) {
        const int row = (wave + NW * it) * 8 + rsub;
        const int n = tn * BN + row;
        const unsigned chunk = (unsigned)((slot ^ ((row >> 1) & 7)) * EPC);
        vb[it] = n < p.n_pad ? ((unsigned)n * (unsigned)p.k_pad + chunk) * 2u : kOob;
    }
    // per-row tap validity
    unsigned mask[TM];
    int rloc[TM];
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        rloc[i] = wm * (BM / WM) + i * 32 + l31;
        const int m = m0 + rloc[i];
        unsigned msk = 0;
        if (m < p.M) {
            const int rem = m % p.hw_out;
            const int y = rem / p.w_out, x = rem - y * p.w_out;
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                const int iy = y + t / 3 - 1, ix = x + t % 3 - 1;
                msk |= ((unsigned)iy < (unsigned)p.h_in && (unsigned)ix < (unsigned)p.w_in) ? (1u << t) : 0u;
            }
        }
        mask[i] = msk;
    }

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // one halo piece (index k of this wave) of channel block cb into halo slot cb & 1; out-of-range: zeros into `dummy`
#define MVLDM_HALO_A(k_, cb_)                                                                                          \
    {                                                                                                                  \
        const int q_ = wave + NW * (k_);                                                                               \
        const bool real_ = (cb_) < cb1 && q_ < np;                                                                     \
        halo_issue_a<DUAL>(p, real_ ? smem + ((cb_) & 1) * a_bytes + q_ * 1024 : dummy, real_ ? off0[k_] : kOob,       \
                           real_ ? off1[DUAL ? (k_) : 0] : kOob, (cb_) < cb1 ? (cb_) : 0);                             \
    }
    // W tile of K-tile index kt_ (= cb*9 + tap) into ring slot ws_; past the end of K: zeros
#define MVLDM_HALO_W(kt_, ws_)                                                                                         \
    {                                                                                                                  \
        const bool real_ = (kt_) < cb1 * 9;                                                                            \
        _Pragma("unroll") for (int it = 0; it < B_IT; ++it)                                                            \
            halo_issue_w(p, wring + (ws_) * W_BYTES + (wave + NW * it) * 1024, real_ ? vb[it] : kOob, real_ ? (kt_) * 128 : 0); \
    }

    // prologue: the whole halo of block 0, W tiles 0..2
#pragma unroll
    for (int k = 0; k < KA; ++k) MVLDM_HALO_A(k, 0)
    MVLDM_HALO_W(0, 0)
    MVLDM_HALO_W(1, 1)
    MVLDM_HALO_W(2, 2)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    // fragment addressing of one tap: masked lanes read the zero row
    const char* abase[TM];
    int arow[TM];
    const char* bt;
#define MVLDM_HALO_ADDR(cb_, t_, ws_)                                                                                  \
    {                                                                                                                  \
        const int disp_ = lead + ((t_) / 3 - 1) * p.w_in + ((t_) % 3 - 1);   /* lane-uniform row displacement */        \
        const char* as_ = smem + ((cb_) & 1) * a_bytes;                                                                \
        _Pragma("unroll") for (int i = 0; i < TM; ++i) {                                                               \
            const bool ok_ = (mask[i] >> (t_)) & 1u;                                                                   \
            abase[i] = ok_ ? as_ : zrow;                                                                               \
            arow[i] = ok_ ? rloc[i] + disp_ : 0;                                                                       \
        }                                                                                                              \
        bt = wring + (ws_) * W_BYTES;                                                                                  \
    }
#define MVLDM_HALO_LOAD(f_, kk_)                                                                                       \
    {                                                                                                                  \
        _Pragma("unroll") for (int i = 0; i < TM; ++i) f_.a[i] = M_::load(abase[i], arow[i], kk_, hi);                 \
        _Pragma("unroll") for (int j = 0; j < TN; ++j) f_.b[j] = M_::load(bt, wn * (BN / WN) + j * 32 + l31, kk_, hi); \
    }
#define MVLDM_HALO_MMA(f_)                \
    __builtin_amdgcn_sched_barrier(0);    \
    bl_mma<T, TM, TN>(f_, acc);           \
    __builtin_amdgcn_sched_barrier(0);

    int wslot = 0, kt = 0;
    BlFrags<T, TM, TN> f0, f1;
    MVLDM_HALO_ADDR(0, 0, 0)
    MVLDM_HALO_LOAD(f0, 0)
    for (int cb = 0; cb < cb1; ++cb) {
#pragma unroll
        for (int t = 0; t < 9; ++t, ++kt) {
            MVLDM_HALO_LOAD(f1, 1)
            MVLDM_HALO_MMA(f0)
            MVLDM_HALO_LOAD(f0, 2)
            MVLDM_HALO_MMA(f1)
            MVLDM_HALO_LOAD(f1, 3)
            MVLDM_HALO_MMA(f0)
            // W tile kt+1 must have landed; behind it at most {halo piece, W tile kt+2, halo piece} = 4 loads are in flight
            asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            MVLDM_HALO_W(kt + 3, wslot)
            if (t < KA) { MVLDM_HALO_A(t, cb + 1) }
            else { MVLDM_HALO_A(0, cb1) }                                     // (keeps the per-step load count uniform)
            wslot = wslot == 2 ? 0 : wslot + 1;
            // first fragments of the next K-tile under the last MFMAs of this one (past the end: harmless reads)
            if (t < 8) { MVLDM_HALO_ADDR(cb, t + 1, wslot) }
            else { MVLDM_HALO_ADDR(cb + 1, 0, wslot) }
            MVLDM_HALO_LOAD(f0, 0)
            MVLDM_HALO_MMA(f1)
        }
    }
#undef MVLDM_HALO_ADDR
#undef MVLDM_HALO_LOAD
#undef MVLDM_HALO_MMA
#undef MVLDM_HALO_A
#undef MVLDM_HALO_W
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    igemm_epilogue_staged<T, BM, BN, WM, WN>(p, acc, tm, tn, split, wm, wn, wave, lane, smem);
}

// ---- tile 17 (round 5): the pixel halo under a 256 x 320 tile, for maps up to 24 pixels wide ------------------------------------
// Tile 10 (256 x 320, the headline's 3x3 convs) idles its matrix pipes 31 % of the time with both waves of a SIMD parked on the next
// K-tile's round trip: 72 KB per CU per step through the 2-slot ring.  With the halo resident (one fill of 256 + 2 (W + 1) pixel
// rows per channel block serves the 9 taps) a step moves 40 KB of weights + 1/9 of the 37 KB halo = 44 KB.  LDS: two halo buffers +
// a 2-slot weight ring = 2 x 37 + 2 x 40 KB at W = 16 (156 KB); a 32-wide map needs 164 KB -- level 0 stays on tile 10.
// 8 waves of 64 x 160 like tile 10 (10 accumulator blocks: fragments are fetched right before use, the two waves of a SIMD cover each
// other's LDS latency); per step a wave issues 5 weight pieces + 1 halo piece behind the barrier, and the counted wait in front of
// the next barrier (vmcnt(1): only the halo piece may still be in flight) is for the weights issued one step earlier.
template <typename T>
__global__ __launch_bounds__(512) void igemm_halow_kernel(const IgemmParams p, int halo_rows) {
    using M_ = Mma<T>;
    constexpr int BM = 256, BN = 320, WM = 4, WN = 2, NW = 8, TM = 2, TN = 5, B_IT = 5, KA = 6, EPC = 8;
    constexpr int W_BYTES = BN * 128;
    static_assert(64 / M_::KI == 4, "four k-sub-steps per K-tile");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int a_bytes = halo_rows * 128;
    char* const wring = smem + 2 * a_bytes;
    char* const zrow = wring + 2 * W_BYTES;
    char* const dummy = zrow + 128;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int hi = lane >> 5, l31 = lane & 31;
    int split, tm, tn;
    if (!map_block(p, split, tm, tn)) return;
    const int cb1 = p.k_tiles / 9;
    const int lead = p.w_in + 1;
    const int m0 = tm * BM;
    const int np = halo_rows / 8;                       // 1 KiB pieces of a halo tile
    const int slot = lane & 7, rsub = lane >> 3;
    const int m_tot = p.n_img * p.h_in * p.w_in;

    if (tid < 8) *reinterpret_cast<u32x4*>(zrow + tid * 16) = u32x4{0u, 0u, 0u, 0u};

    unsigned off0[KA];
#pragma unroll
    for (int k = 0; k < KA; ++k) {
        const int q = wave + NW * k;
        const int hr = q * 8 + rsub;
        const int pm = m0 - lead + hr;
        const bool ok = q < np && pm >= 0 && pm < m_tot;
        const unsigned chunk = (unsigned)((slot ^ ((hr >> 1) & 7)) * EPC);
        off0[k] = ok ? ((unsigned)pm * (unsigned)p.c0 + chunk) * 2u : kOob;
    }
    unsigned vb[B_IT];
#pragma unroll
    for (int it = 0; it < B_IT; ++it) {
        const int row = (wave + NW * it) * 8 + rsub;
        const int n = tn * BN + row;
        const unsigned chunk = (unsigned)((slot ^ ((row >> 1) & 7)) * EPC);
        vb[it] = n < p.n_pad ? ((unsigned)n * (unsigned)p.k_pad + chunk) * 2u : kOob;
    }
    unsigned mask[TM];
    int rloc[TM];
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        rloc[i] = wm * (BM / WM) + i * 32 + l31;
        const int m = m0 + rloc[i];
        unsigned msk = 0;
        if (m < p.M) {
            const int rem = m % p.hw_out;
            const int y = rem / p.w_out, x = rem - y * p.w_out;
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                const int iy = y + t / 3 - 1, ix = x + t % 3 - 1;
                msk |= ((unsigned)iy < (unsigned)p.h_in && (unsigned)ix < (unsigned)p.w_in) ? (1u << t) : 0u;
            }
        }
        mask[i] = msk;
    }

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

#define MVLDM_HW_A(k_, cb_)                                                                                            \
    {                                                                                                                  \
        const int q_ = wave + NW * (k_);                                                                               \
        const bool real_ = (cb_) < cb1 && q_ < np;                                                                     \
        halo_issue_a<false>(p, real_ ? smem + ((cb_) & 1) * a_bytes + q_ * 1024 : dummy, real_ ? off0[k_] : kOob, kOob, \
                            (cb_) < cb1 ? (cb_) : 0);                                                                  \
    }
#define MVLDM_HW_W(kt_, ws_)                                                                                           \
    {                                                                                                                  \
        const bool real_ = (kt_) < cb1 * 9;                                                                            \
        _Pragma("unroll") for (int it = 0; it < B_IT; ++it)                                                            \
            halo_issue_w(p, wring + (ws_) * W_BYTES + (wave + NW * it) * 1024, real_ ? vb[it] : kOob, real_ ? (kt_) * 128 : 0); \
    }

    // prologue: the whole halo of block 0, W tiles 0 and 1
#pragma unroll
    for (int k = 0; k < KA; ++k) MVLDM_HW_A(k, 0)
    MVLDM_HW_W(0, 0)
    MVLDM_HW_W(1, 1)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    int kt = 0;
    for (int cb = 0; cb < cb1; ++cb) {
        const char* as = smem + (cb & 1) * a_bytes;
        // (the per-tap row offsets are loop-invariant: hipcc hoists all 9 x TM of them and spills 46 registers to scratch, whose reloads
        //  count in vmcnt like the DMA pieces.  Opaque per-iteration values keep the two selects per tap inside the loop.)
        asm volatile("" : "+v"(mask[0]), "+v"(mask[1]), "+v"(rloc[0]), "+v"(rloc[1]));
#pragma unroll
        for (int t = 0; t < 9; ++t, ++kt) {
            const int ws = kt & 1;
            const int disp = lead + (t / 3 - 1) * p.w_in + (t % 3 - 1);      // lane-uniform row displacement of the tap
            const char* abase[TM];
            int arow[TM];
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const bool ok = (mask[i] >> t) & 1u;
                abase[i] = ok ? as : zrow;
                arow[i] = ok ? rloc[i] + disp : 0;
            }
            const char* bt = wring + ws * W_BYTES;
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                BlFrags<T, TM, TN> f;
#pragma unroll
                for (int i = 0; i < TM; ++i) f.a[i] = M_::load(abase[i], arow[i], kk, hi);
#pragma unroll
                for (int j = 0; j < TN; ++j) f.b[j] = M_::load(bt, wn * (BN / WN) + j * 32 + l31, kk, hi);
                bl_mma<T, TM, TN>(f, acc);
            }
            // W tile kt+1 has landed (behind it only the halo piece issued with it may still be in flight); every wave is done with
            // W slot `ws` -- and, after tap 8, with this block's halo buffer
            asm volatile("s_waitcnt vmcnt(1) lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            MVLDM_HW_W(kt + 2, ws)
            if (t < KA) { MVLDM_HW_A(t, cb + 1) }
            else { MVLDM_HW_A(0, cb1) }                                       // (keeps the per-step load count uniform)
        }
    }
#undef MVLDM_HW_A
#undef MVLDM_HW_W
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    igemm_epilogue_staged<T, BM, BN, WM, WN>(p, acc, tm, tn, split, wm, wn, wave, lane, smem);
}

// split-K: sum the fp32 partial slabs and run the same epilogue (deterministic, no atomics)
template <typename T> __global__ __launch_bounds__(256) void igemm_splitk_reduce(const IgemmParams p) {
    const bool geglu = p.epilogue == MVLDM_EPI_GEGLU;
    const size_t total = (size_t)p.M * p.n_dst;
    const size_t slab = (size_t)p.M * p.n_pad;
    for (size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (size_t)gridDim.x * 256) {
        const int m = (int)(idx / p.n_dst), col = (int)(idx - (size_t)m * p.n_dst);
        float v;
        if (geglu) {
            const int nv = (col >> 5) * 64 + (col & 31), ng = nv + 32;
            float a = 0.f, g = 0.f;
            for (int s = 0; s < p.splitk; ++s) {
                a += p.ws[s * slab + (size_t)m * p.n_pad + nv];
                g += p.ws[s * slab + (size_t)m * p.n_pad + ng];
            }
            if (p.bias) { a += p.bias[col]; g += p.bias[p.n_dst + col]; }
            v = a * gelu_erf_fast(g);
        } else {
            float a = 0.f;
            for (int s = 0; s < p.splitk; ++s) a += p.ws[s * slab + (size_t)m * p.n_pad + col];
            if (p.bias) a += p.bias[col];
            if (p.row_bias) a += p.row_bias[(size_t)(m / p.hw_out) * p.row_bias_ld + col];
            if (p.epilogue == MVLDM_EPI_SILU) a = silu_f(a);
            else if (p.epilogue == MVLDM_EPI_GELU) a = gelu_erf_fast(a);
            v = a;
        }
        epilogue_store<T, true>(p, m, col, v);
    }
}

// the same reduction, 8 output columns per thread (16-bit output, 8-column-aligned rows): 16-byte slab / bias / residual loads
// and one 16-byte store instead of eight scalar round trips and eight index divisions.  At one scene a DDIM step runs ~125
// split-K launches and the scalar form (7.8 us per launch on average) was 11 % of the GPU time.
template <typename T> __global__ __launch_bounds__(256) void igemm_splitk_reduce_vec(const IgemmParams p) {
    const bool geglu = p.epilogue == MVLDM_EPI_GEGLU;
    const int cpr = p.n_dst / 8;                                  // 8-column chunks per output row
    const size_t total = (size_t)p.M * cpr;
    const size_t slab = (size_t)p.M * p.n_pad;
    for (size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (size_t)gridDim.x * 256) {
        const int m = (int)(idx / cpr), col = (int)(idx - (size_t)m * cpr) * 8;
        const int nv = geglu ? (col >> 5) * 64 + (col & 31) : col;
        const float* w0 = p.ws + (size_t)m * p.n_pad + nv;
        float a[8], g[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) a[e] = g[e] = 0.f;
        // the slabs are summed in split order (fixed: same bits on every run), FOUR slabs' loads in flight at a time: with one slab per
        // iteration every add waited for its own L2 round trip -- 10 splits = 10 dependent latencies = 5-6 us for a 7 MB read
        // (rocprofv3, the 4x4-level convs of one scene), as long as a third of the GEMM it finishes
        int s = 0;
        for (; s + 4 <= p.splitk; s += 4) {
            f32x4 lo[4], hi[4], gl[4], gh[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                lo[u] = *reinterpret_cast<const f32x4*>(w0 + (s + u) * slab);
                hi[u] = *reinterpret_cast<const f32x4*>(w0 + (s + u) * slab + 4);
                if (geglu) {
                    gl[u] = *reinterpret_cast<const f32x4*>(w0 + (s + u) * slab + 32);
                    gh[u] = *reinterpret_cast<const f32x4*>(w0 + (s + u) * slab + 36);
                }
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
#pragma unroll
                for (int e = 0; e < 4; ++e) { a[e] += lo[u][e]; a[4 + e] += hi[u][e]; }
                if (geglu) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) { g[e] += gl[u][e]; g[4 + e] += gh[u][e]; }
                }
            }
        }
        for (; s < p.splitk; ++s) {
            const f32x4 lo = *reinterpret_cast<const f32x4*>(w0 + s * slab), hi = *reinterpret_cast<const f32x4*>(w0 + s * slab + 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) { a[e] += lo[e]; a[4 + e] += hi[e]; }
            if (geglu) {
                const f32x4 gl = *reinterpret_cast<const f32x4*>(w0 + s * slab + 32), gh = *reinterpret_cast<const f32x4*>(w0 + s * slab + 36);
#pragma unroll
                for (int e = 0; e < 4; ++e) { g[e] += gl[e]; g[4 + e] += gh[e]; }
            }
        }
        if (p.bias) {
#pragma unroll
            for (int e = 0; e < 8; ++e) a[e] += p.bias[col + e];
        }
        if (geglu) {
#pragma unroll
            for (int e = 0; e < 8; ++e) a[e] *= gelu_erf_fast(g[e] + (p.bias ? p.bias[p.n_dst + col + e] : 0.f));
        } else {
            if (p.row_bias) {
                const float* rb = p.row_bias + (size_t)(m / p.hw_out) * p.row_bias_ld + col;
#pragma unroll
                for (int e = 0; e < 8; ++e) a[e] += rb[e];
            }
            if (p.epilogue == MVLDM_EPI_SILU) {
#pragma unroll
                for (int e = 0; e < 8; ++e) a[e] = silu_f(a[e]);
            } else if (p.epilogue == MVLDM_EPI_GELU) {
#pragma unroll
                for (int e = 0; e < 8; ++e) a[e] = gelu_erf_fast(a[e]);
            }
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) a[e] *= p.out_scale;
        if (p.residual) {
            const Chunk<T> rc = load_chunk<T>(reinterpret_cast<const T*>(p.residual) + (size_t)m * p.n_dst + col);
#pragma unroll
            for (int e = 0; e < 8; ++e) a[e] += rc.get(e);
        }
        size_t drow = (size_t)m;
        if (p.scatter) {
            const int img = m / p.hw_out, rem = m - img * p.hw_out;
            const int i = rem / p.w_out, j = rem - i * p.w_out;
            drow = ((size_t)img * (2 * p.h_out) + 2 * i + p.ph_y) * (size_t)(2 * p.w_out) + 2 * j + p.ph_x;
        }
        Chunk<T> oc;
#pragma unroll
        for (int e = 0; e < 8; ++e) oc.set(e, a[e]);
        store_chunk<T>(reinterpret_cast<T*>(p.dst) + drow * p.dst_ld + col, oc);
    }
}

// ---- weight packing -------------------------------------------------------------------------------
// One kernel for every pack: a workgroup executes the body its job's `kind` names on its block index inside the job.  A single
// pack (mvldm_pack_weight) passes its job by value; the batched launch (mvldm_pack_weight_batch) finds the job of a workgroup in
// the device job list by its `block0`.  Same bodies, hence the same bytes out either way.
enum { PACK_GENERIC = 0, PACK_FWD_K1 = 1, PACK_FWD_K3 = 2, PACK_T_K3 = 3, PACK_T_K1 = 4 };

// generic: one thread per packed element, any dtype / K order (grid-stride over the job's `nb` workgroups)
template <typename T>
__device__ __forceinline__ void pack_generic_body(const mvldm_pack_job& j, int bx, int nb) {
    T* __restrict__ dst = reinterpret_cast<T*>(j.dst);
    const float* __restrict__ src = j.src;
    const int bk = sizeof(T) == 4 ? 32 : 64;
    const size_t total = (size_t)j.n_pad * j.k_pad;
    const int taps = j.ksize * j.ksize;
    for (size_t idx = (size_t)bx * 256 + threadIdx.x; idx < total; idx += (size_t)nb * 256) {
        const int np = (int)(idx / j.k_pad), k = (int)(idx - (size_t)np * j.k_pad);
        int tap, c;
        if (j.k_order) {
            const int kt = k / bk, cb = kt / taps;
            tap = kt - cb * taps;
            c = cb * bk + (k - kt * bk);
        } else {
            tap = k / j.c_pad;
            c = k - tap * j.c_pad;
        }
        float v = 0.f;
        if (j.transpose) {
            // data-gradient weight: row np = input channel c_off + np, K channel `c` = OUTPUT channel, taps flipped
            if (np < j.n_rows && tap < taps && c < j.n_out)
                v = src[((size_t)c * j.c_in + (j.c_off + np)) * taps + (taps - 1 - tap)];
        } else {
            const int n = orig_col(np, j.n_out, j.geglu != 0);
            if (n < j.n_out && tap < taps && c < j.c_in)
                v = src[((size_t)n * j.c_in + c) * taps + tap];
        }
        dst[idx] = from_f32<T>(v);
    }
}

// ---- fast packers for the block-major 16-bit layout (k_order 1, 64-channel blocks) ----------------------------------------
// The generic body above is one thread per packed element with three index divisions, 2-byte stores, and -- for 3x3 and for
// the transposed (data-gradient) packs -- source reads 36 bytes to kilobytes apart: re-packing the 926 M trained parameters
// after every optimizer step took 8.4 ms.  These go through an LDS tile so that both sides are contiguous runs.
// forward, 1x1 / Linear: a row copy with conversion, 4 columns per thread (GEGLU only permutes rows)
template <typename T>
__device__ __forceinline__ void pack_fwd_k1_body(const mvldm_pack_job& j, int bx, int nb) {
    T* __restrict__ dst = reinterpret_cast<T*>(j.dst);
    const float* __restrict__ src = j.src;
    const int kq = j.k_pad / 4;
    const size_t total = (size_t)j.n_pad * kq;
    for (size_t idx = (size_t)bx * 256 + threadIdx.x; idx < total; idx += (size_t)nb * 256) {
        const int np = (int)(idx / kq), k = (int)(idx - (size_t)np * kq) * 4;
        const int n = orig_col(np, j.n_out, j.geglu != 0);
        float v[4] = {0.f, 0.f, 0.f, 0.f};
        if (n < j.n_out) {
            const float* sp = src + (size_t)n * j.c_in + k;
            if (k + 3 < j.c_in && (((size_t)n * j.c_in + k) & 3) == 0 && (reinterpret_cast<uintptr_t>(src) & 15) == 0) {
                const f32x4 q = *reinterpret_cast<const f32x4*>(sp);
                v[0] = q[0]; v[1] = q[1]; v[2] = q[2]; v[3] = q[3];
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) if (k + e < j.c_in) v[e] = sp[e];
            }
        }
        T o[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = from_f32<T>(v[e]);
        *reinterpret_cast<u32x2*>(dst + (size_t)np * j.k_pad + k) = *reinterpret_cast<const u32x2*>(o);
    }
}
// forward, 3x3: 4 output rows x one 64-channel block per workgroup; [c][tap] runs of 576 floats in, [tap][c] runs out
template <typename T>
__device__ __forceinline__ void pack_fwd_k3_body(const mvldm_pack_job& j, int bx, float* lds) {
    T* __restrict__ dst = reinterpret_cast<T*>(j.dst);
    const float* __restrict__ src = j.src;
    const int ncb = j.c_pad / 64;
    const int cb = bx % ncb, np0 = (bx / ncb) * 4;
    if (cb * 64 + 64 <= j.c_in && (j.c_in & 3) == 0 && (reinterpret_cast<uintptr_t>(src) & 15) == 0) {
        // whole 64-channel block inside the weight: 16-byte loads of the [c][tap] run (576 floats per output row, 16-byte aligned)
        for (int i = threadIdx.x; i < 4 * 144; i += 256) {
            const int r = i / 144, e = (i - r * 144) * 4;
            const int n = np0 + r;
            f32x4 q = {0.f, 0.f, 0.f, 0.f};
            if (n < j.n_out) q = *reinterpret_cast<const f32x4*>(src + ((size_t)n * j.c_in + cb * 64) * 9 + e);
#pragma unroll
            for (int u = 0; u < 4; ++u) lds[r * 577 + e + u] = q[u];
        }
    } else {
        for (int i = threadIdx.x; i < 4 * 576; i += 256) {
            const int r = i / 576, e = i - r * 576;
            const int n = np0 + r, c = cb * 64 + e / 9;
            lds[r * 577 + e] = (n < j.n_out && c < j.c_in) ? src[((size_t)n * j.c_in + cb * 64) * 9 + e] : 0.f;
        }
    }
    __syncthreads();
    // two consecutive output elements (channels cw, cw + 1 of one tap) per thread: 4-byte stores, 256 bytes per wave instruction
    for (int i = threadIdx.x; i < 4 * 288; i += 256) {
        const int r = i / 288, o = (i - r * 288) * 2;
        const int tap = o >> 6, cw = o & 63;
        if (np0 + r < j.n_pad) {
            T pr[2] = {from_f32<T>(lds[r * 577 + cw * 9 + tap]), from_f32<T>(lds[r * 577 + (cw + 1) * 9 + tap])};
            *reinterpret_cast<uint32_t*>(dst + (size_t)(np0 + r) * j.k_pad + cb * 576 + o) = *reinterpret_cast<const uint32_t*>(pr);
        }
    }
}
// transposed (data-gradient) pack: rows = input channels, K = (64-output-channel block, flipped tap, output channel)
template <typename T, int KS>
__device__ __forceinline__ void pack_t_body(const mvldm_pack_job& j, int bx, float* lds) {
    constexpr int TAPS = KS * KS, RB = KS == 3 ? 8 : 64, RUN = RB * TAPS, PITCH = RUN + 1;
    T* __restrict__ dst = reinterpret_cast<T*>(j.dst);
    const float* __restrict__ src = j.src;
    const int nnb = j.c_pad / 64;
    const int nb = bx % nnb, r0 = (bx / nnb) * RB;
    if (r0 + RB <= j.n_rows && (j.c_in & 3) == 0 && (j.c_off & 3) == 0 && (reinterpret_cast<uintptr_t>(src) & 15) == 0) {
        // the whole run of RB rows x TAPS lies inside the weight: 16-byte loads (RUN floats per output channel, 16-byte aligned)
        for (int i = threadIdx.x; i < 64 * (RUN / 4); i += 256) {
            const int nw = i / (RUN / 4), e = (i - nw * (RUN / 4)) * 4;
            const int n = nb * 64 + nw;
            f32x4 q = {0.f, 0.f, 0.f, 0.f};
            if (n < j.n_out) q = *reinterpret_cast<const f32x4*>(src + ((size_t)n * j.c_in + j.c_off + r0) * TAPS + e);
#pragma unroll
            for (int u = 0; u < 4; ++u) lds[nw * PITCH + e + u] = q[u];
        }
    } else {
        for (int i = threadIdx.x; i < 64 * RUN; i += 256) {
            const int nw = i / RUN, e = i - nw * RUN;
            const int n = nb * 64 + nw, r = r0 + e / TAPS;
            lds[nw * PITCH + e] = (n < j.n_out && r < j.n_rows) ? src[((size_t)n * j.c_in + j.c_off + r0) * TAPS + e] : 0.f;
        }
    }
    __syncthreads();
    // two consecutive output channels per thread: 4-byte stores (the odd LDS pitch keeps the two reads on different banks)
    for (int i = threadIdx.x; i < 32 * RUN; i += 256) {
        const int nw = (i & 31) * 2, q = i >> 5;
        const int rl = q / TAPS, tp = q - rl * TAPS;
        if (r0 + rl < j.n_pad) {
            const int e = rl * TAPS + (TAPS - 1 - tp);
            T pr[2] = {from_f32<T>(lds[nw * PITCH + e]), from_f32<T>(lds[(nw + 1) * PITCH + e])};
            *reinterpret_cast<uint32_t*>(dst + (size_t)(r0 + rl) * j.k_pad + (nb * TAPS + tp) * 64 + nw) = *reinterpret_cast<const uint32_t*>(pr);
        }
    }
}

constexpr int kPackLds = 64 * 73;      // floats: the transposed 3x3 tile (64 output channels x (8 rows x 9 taps + 1))
template <typename T>
__global__ __launch_bounds__(256) void pack_job_kernel(const mvldm_pack_job* __restrict__ jobs, int n_jobs, const int32_t* __restrict__ block_job,
                                                       const mvldm_pack_job single) {
    __shared__ float lds[kPackLds];
    mvldm_pack_job j = single;
    int bx = blockIdx.x;
    if (jobs) {      // batched launch: the caller's workgroup -> job table, or the last job whose first workgroup is <= this one
        int lo = 0, hi = n_jobs - 1;      // (block0 ascending; everything here is workgroup-uniform: scalar loads)
        if (block_job) {
            lo = block_job[bx];
        } else {
            while (lo < hi) {
                const int mid = (lo + hi + 1) >> 1;
                if (jobs[mid].block0 <= bx) lo = mid; else hi = mid - 1;
            }
        }
        j = jobs[lo];
        bx -= j.block0;
    }
    if constexpr (sizeof(T) == 2) {
        switch (j.kind) {
            case PACK_FWD_K1: pack_fwd_k1_body<T>(j, bx, j.blocks); return;
            case PACK_FWD_K3: pack_fwd_k3_body<T>(j, bx, lds); return;
            case PACK_T_K3: pack_t_body<T, 3>(j, bx, lds); return;
            case PACK_T_K1: pack_t_body<T, 1>(j, bx, lds); return;
            default: break;
        }
    }
    pack_generic_body<T>(j, bx, j.blocks);
}

// ---- host side ------------------------------------------------------------------------------------
static inline int cdiv(int a, int b) { return (a + b - 1) / b; }
struct TileCfg { int bm, bn, threads; };
static const TileCfg kTiles[] = {{0, 0, 0}, {128, 128, 256}, {128, 64, 256}, {64, 128, 256}, {64, 64, 128}, {32, 64, 64},
                                 {256, 64, 256},     // tile 6: 64x64 wave tile, lean 16-bit loop only
                                 {256, 128, 512},    // tile 7: 8 waves of 64x64 -- half the L2->LDS bytes per flop of tile 2
                                 {128, 256, 512},    // tile 8
                                 {256, 256, 512},    // tile 9: 8 waves of 64x128, 2-deep ring (128 KB): 128 flop per L2->LDS byte
                                 {256, 320, 512},    // tile 10: 8 waves of 64x160 -- every channel count of this UNet is a
                                                     // multiple of 320 (no N padding); 142 flop per L2->LDS byte
                                 {256, 128, 512},    // tile 11: 256x128 with the LDS-resident pixel halo (3x3 stride-1 convs)
                                 {0, 0, 0}, {0, 0, 0}, {0, 0, 0}, {0, 0, 0},      // 12 - 14: linear_pp / linear_pw / linear_ws (own files); 15 unused
                                 {0, 0, 0},          // 16: unused (deep-ring form of tile 2, measured slower, not built)
                                 {256, 320, 512},    // tile 17: 256x320 with the LDS-resident pixel halo (3x3 stride-1 convs on maps <= 24 wide)
                                 // deep-ring tile for launches of a few hundred rows (weight-bound: levels 2 - 4 at a few scenes):
                                 {192, 128, 512}};   // tile 18: 8 waves of 96x32, 4 slots (160 KB): <= 192 rows read every weight byte once
constexpr int kNumTiles = 11;
static inline bool deep_tile(int tile) { return tile == 18; }

// tuning knobs (read once): MVLDM_IGEMM_STAGES (0 = heuristic), MVLDM_IGEMM_TARGET (split-K workgroup
// target), MVLDM_IGEMM_SYNC=1 (force the register-prefetch main loop for 16-bit types: A/B testing)
static int env_int(const char* name, int dflt) { return knob_int(name, dflt); }      // (experiment builds only: common.h)
static const int kEnvTarget = env_int("MVLDM_IGEMM_TARGET", 0);
static const int kEnvSync = env_int("MVLDM_IGEMM_SYNC", 0);
static const int kEnvPx = env_int("MVLDM_IGEMM_PX", 0);
static const int kEnvNoStage = env_int("MVLDM_IGEMM_NOSTAGE", 0);
// MVLDM_IGEMM_FAKE makes every conv/linear return WRONG results (operands read as zeros, epilogue skipped): it exists
// for the roofline experiments of tools/fake_probe.sh only and is compiled in only with -DMVLDM_EXPERIMENTS
// (MVLDM_EXPERIMENTS=1 python -m mv_ldm_amd._build --force); the product library ignores the variable.
#ifdef MVLDM_EXPERIMENTS
static const int kEnvFake = env_int("MVLDM_IGEMM_FAKE", 0);
#else
static constexpr int kEnvFake = 0;
#endif

template <typename KernT> static int launch_kernel(KernT kern, std::atomic<uint64_t>& attr_done, int smem, int blocks, int threads,
                                                   const IgemmParams& p, hipStream_t s) {
    if (int rc0 = ensure_dyn_smem(reinterpret_cast<const void*>(kern), smem, attr_done)) return rc0;
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(threads), smem, s, p);
    return check_launch();
}

template <typename T, int BM, int BN, int WM, int WN> static int launch_sync(const IgemmParams& p, hipStream_t s) {
    static std::atomic<uint64_t> done{0};
    constexpr int ring = 2 * (BM + BN) * Mma<T>::PITCH, park = WM * WN * 32 * (park_blocks(BN / WN / 32) * 32 + 4) * 4;
    return launch_kernel(igemm_kernel<T, BM, BN, WM, WN>, done, (sizeof(T) == 2 && park > ring) ? park : ring,
                         8 * p.sub_m * p.sub_n * p.splitk, WM * WN * 64, p, s);
}

// per-call tuning overrides ride in the upper bits of desc.tile: bits 8-11 px (XCD grid), bit 12 register-prefetch
// loop instead of the LDS-DMA one (A/B testing)
static thread_local int t_force_sync = 0;

template <typename T, int BM, int BN, int WM, int WN, int KS, bool DUAL, int STAGES, bool UPS>
static int launch_bl_s(const IgemmParams& p, hipStream_t s) {
    static std::atomic<uint64_t> done{0};
    // the epilogue parks one 32-row fp32 block per wave in the (then idle) ring
    constexpr int ring = STAGES * (BM + BN) * 128, park = WM * WN * 32 * (park_blocks(BN / WN / 32) * 32 + 4) * 4;
    return launch_kernel(igemm_bl_kernel<T, BM, BN, WM, WN, KS, DUAL, STAGES, UPS>, done, ring > park ? ring : park,
                         8 * p.sub_m * p.sub_n * p.splitk, WM * WN * 64, p, s);
}
template <typename T, int BM, int BN, int WM, int WN, int KS, bool DUAL, int DEPTH = 0>
static int launch_bl(const IgemmParams& p, hipStream_t s) {
    // ring depth is fixed per tile (sweeps in profiles/r01_igemm_sweep*.json): the 4-wave tiles run 2-3
    // workgroups per CU and lose more to a third slot than they gain; the 8-wave 256x128 / 128x256 tiles own
    // the CU and take 3 slots; 256x256 only has room for 2.  DEPTH > 0: the deep-ring tiles (16 - 18) name theirs.
    constexpr int STAGES = DEPTH ? DEPTH : ((WM * WN == 8 && 3 * (BM + BN) * 128 <= 160 * 1024) ? 3 : 2);
    if (p.upsample) {
        // per-tap address tables: only built for the two tiles the host maps upsampling convs to
        if constexpr (KS == 3 && !DUAL && ((BM == 128 && BN == 64) || (BM == 256 && BN == 128)))
            return launch_bl_s<T, BM, BN, WM, WN, KS, DUAL, STAGES, true>(p, s);
        else
            return set_error(MVLDM_ERR_ARG, "igemm: upsampling 3x3 conv needs tile 2 or 7 on the 16-bit path");
    }
    return launch_bl_s<T, BM, BN, WM, WN, KS, DUAL, STAGES, false>(p, s);
}
// the deep-ring tiles: 1x1 / 3x3, one or two sources, no upsampling forms (fill_params maps those to tile 2)
template <typename T, int BM, int BN, int WM, int WN, int DEPTH> static int launch_bl_deep(const IgemmParams& p, hipStream_t s) {
    const bool dual = p.c1 > 0;
    if (p.ksize == 3) return dual ? launch_bl<T, BM, BN, WM, WN, 3, true, DEPTH>(p, s) : launch_bl<T, BM, BN, WM, WN, 3, false, DEPTH>(p, s);
    if (p.ksize == 1) return dual ? launch_bl<T, BM, BN, WM, WN, 1, true, DEPTH>(p, s) : launch_bl<T, BM, BN, WM, WN, 1, false, DEPTH>(p, s);
    return set_error(MVLDM_ERR_ARG, "igemm: the deep-ring tiles take 1x1 and 3x3 convs");
}

template <typename T, int BM, int BN, int WM, int WN> static int launch_bl_any(const IgemmParams& p, hipStream_t s) {
    const bool dual = p.c1 > 0;
    if (p.ksize == 2) {   // the four 2x2 phases of a decomposed nearest-2x upsampling conv: tiles 2, 7 and 10 only
        if constexpr ((BM == 128 && BN == 64) || (BM == 256 && BN == 128) || (BM == 256 && BN == 320))
            return launch_bl<T, BM, BN, WM, WN, 2, false>(p, s);
        else
            return set_error(MVLDM_ERR_ARG, "igemm: 2x2 phase conv needs tile 2, 7 or 10");
    }
    if (p.ksize == 3) {
        if (!dual) return launch_bl<T, BM, BN, WM, WN, 3, false>(p, s);
        // two-source 3x3 convs: the 256x256 / 256x320 tiles do not fit the register file with a second set of per-piece offsets
        // (hipcc: 31-33 spilled registers, 128-136 B of scratch per lane) and are not instantiated; the rules never pick them
        // (choose_config) and an explicit request is refused.  (The UNet has no such conv: GroupNorm materialises the skip
        // concat before conv1; only the 1x1 shortcuts read two sources.)
        if constexpr (BM == 256 && BN >= 256)
            return set_error(MVLDM_ERR_UNSUPPORTED, "igemm: tiles 9 / 10 do not take a two-source 3x3 conv (use tile 7 or 11)");
        else
            return launch_bl<T, BM, BN, WM, WN, 3, true>(p, s);
    }
    return dual ? launch_bl<T, BM, BN, WM, WN, 1, true>(p, s) : launch_bl<T, BM, BN, WM, WN, 1, false>(p, s);
}

template <typename T, int BM, int BN, int WM, int WN> static int launch_tile(const IgemmParams& p, hipStream_t s) {
    if constexpr (sizeof(T) == 2) {
        if (p.use_bl) return launch_bl_any<T, BM, BN, WM, WN>(p, s);
    }
    // f32, and the few 16-bit problems whose channel counts are not multiples of 64 (conv_in, VAE conv_in/out,
    // quant convs): register-prefetch loop
    return launch_sync<T, BM, BN, WM, WN>(p, s);
}

static inline int halo_rows_for(int w_in) { return (256 + 2 * (w_in + 1) + 7) / 8 * 8; }
template <typename T> static int launch_halo(const IgemmParams& p, hipStream_t s) {
    const int hr = halo_rows_for(p.w_in);
    const int smem = 2 * hr * 128 + 3 * 128 * 128 + 128 + 1024;
    const int blocks = 8 * p.sub_m * p.sub_n;
    static std::atomic<uint64_t> done0{0}, done1{0};
    if (p.c1 > 0) {
        if (int rc0 = ensure_dyn_smem(reinterpret_cast<const void*>(igemm_halo_kernel<T, true>), 160 * 1024, done1)) return rc0;
        hipLaunchKernelGGL((igemm_halo_kernel<T, true>), dim3(blocks), dim3(512), smem, s, p, hr);
    } else {
        if (int rc0 = ensure_dyn_smem(reinterpret_cast<const void*>(igemm_halo_kernel<T, false>), 160 * 1024, done0)) return rc0;
        hipLaunchKernelGGL((igemm_halo_kernel<T, false>), dim3(blocks), dim3(512), smem, s, p, hr);
    }
    return check_launch();
}

static inline int halow_smem(int w_in) { return 2 * halo_rows_for(w_in) * 128 + 2 * 320 * 128 + 128 + 1024; }
template <typename T> static int launch_halow(const IgemmParams& p, hipStream_t s) {
    static std::atomic<uint64_t> done{0};
    if (int rc0 = ensure_dyn_smem(reinterpret_cast<const void*>(igemm_halow_kernel<T>), 160 * 1024, done)) return rc0;
    hipLaunchKernelGGL((igemm_halow_kernel<T>), dim3(8 * p.sub_m * p.sub_n), dim3(512), halow_smem(p.w_in), s, p, halo_rows_for(p.w_in));
    return check_launch();
}

template <typename T> static int launch_igemm(IgemmParams& p, int tile, hipStream_t s) {
    switch (tile) {
#ifndef MVLDM_TILE_SUBSET      // (resource-usage / quick-compile experiments build the tiles under study only; never the library)
        case 1: return launch_tile<T, 128, 128, 2, 2>(p, s);
        case 2: return launch_tile<T, 128, 64, 4, 1>(p, s);
        case 3: return launch_tile<T, 64, 128, 2, 2>(p, s);
        case 4: return launch_tile<T, 64, 64, 2, 1>(p, s);
        case 5: return launch_tile<T, 32, 64, 1, 1>(p, s);
        case 6:
            if constexpr (sizeof(T) == 2) {
                if (p.use_bl) return launch_bl_any<T, 256, 64, 4, 1>(p, s);
            }
            return set_error(MVLDM_ERR_ARG, "igemm: tile 6 needs the 16-bit block-major path");
        case 7:
            if constexpr (sizeof(T) == 2) {
                if (p.use_bl) return launch_bl_any<T, 256, 128, 4, 2>(p, s);
            }
            return set_error(MVLDM_ERR_ARG, "igemm: tile 7 needs the 16-bit block-major path");
        case 8:
            if constexpr (sizeof(T) == 2) {
                if (p.use_bl) return launch_bl_any<T, 128, 256, 2, 4>(p, s);
            }
            return set_error(MVLDM_ERR_ARG, "igemm: tile 8 needs the 16-bit block-major path");
        case 9:
            if constexpr (sizeof(T) == 2) {
                if (p.use_bl) return launch_bl_any<T, 256, 256, 4, 2>(p, s);
            }
            return set_error(MVLDM_ERR_ARG, "igemm: tile 9 needs the 16-bit block-major path");
        case 11:
            if constexpr (sizeof(T) == 2) {
                if (p.use_bl) return launch_halo<T>(p, s);
            }
            return set_error(MVLDM_ERR_ARG, "igemm: tile 11 needs the 16-bit block-major path");
#endif
        case 10:
            if constexpr (sizeof(T) == 2) {
                if (p.use_bl) return launch_bl_any<T, 256, 320, 4, 2>(p, s);
            }
            return set_error(MVLDM_ERR_ARG, "igemm: tile 10 needs the 16-bit block-major path");
        case 17:
            if constexpr (sizeof(T) == 2) {
                if (p.use_bl) return launch_halow<T>(p, s);
            }
            return set_error(MVLDM_ERR_ARG, "igemm: tile 17 needs the 16-bit block-major path");
        case 18:
            // (deep-ring forms of tiles 2 / 4 -- 128x64 with 5 slots, 64x64 with 6 -- were built and measured SLOWER than their 2-slot forms on every
            //  one-scene shape (tools/skinny_probe.py: 25.2 / 30.2 us against 22.0 / 21.8 on the 4x4-level conv): several 2-slot
            //  workgroups per CU already overlap each other's round trips; not instantiated)
            // (GEGLU pairs a value block with the gate block 32 columns on INSIDE a wave's tile: a 32-column wave tile cannot -- refused,
            //  never remapped.  The first build let it through; the epilogue then read the neighbouring wave's parked block, which is the
            //  right one whenever that wave had already parked it: correct in most runs, different between eager and graph replay.)
            if (p.epilogue == MVLDM_EPI_GEGLU) return set_error(MVLDM_ERR_UNSUPPORTED, "igemm: tile 18 does not take the GEGLU epilogue");
            if constexpr (sizeof(T) == 2) {
                if (p.use_bl && !p.upsample) return launch_bl_deep<T, 192, 128, 2, 4, 4>(p, s);
            }
            return set_error(MVLDM_ERR_ARG, "igemm: tile 18 needs the 16-bit block-major path (no upsampling forms)");
        default: return set_error(MVLDM_ERR_ARG, "igemm: bad tile %d", tile);
    }
}


// Pick tile + split-K.  Target: >= ~2 workgroups per CU (256 CUs) without shredding K below 4 tiles.
static void choose_config(const mvldm_igemm_desc& d, int M, int k_tiles, int& tile, int& splitk, size_t ws_bytes) {
    const int target = kEnvTarget ? kEnvTarget : 512;
    if (tile == 0) {
        // measured on MI355X (tools/igemm_sweep.py, profiles/r01_igemm_sweep_*.json): the 32x64 wave tile
        // wins everywhere at these sizes (3 workgroups per CU with a 2-deep ring); 128x64 for 3x3 convs
        // and narrow N, 64x128 for wide Linear layers; tiny M gets the small tiles.
        // Large problems are bound by the L2 -> LDS fill rate (PMC: ~18 TB/s at 43 flop/byte with 128x64
        // tiles): the 8-wave 256x128 / 128x256 tiles halve the bytes per flop and take a 3-deep ring.  They
        // need >= ~300 workgroups to keep 256 CUs busy (profiles/r01_igemm_sweep2_*.json).
        // 256x320: no N padding at this UNet's channel counts (320/640/960/1280) and the best bytes-per-flop, but
        // only ~1 workgroup per CU: needs whole rounds of 256 workgroups (sweep5: +14..33 % on the 32x32-level
        // convs / Linears at 32 scenes, a loss below ~2 rounds)
        const bool can10 = d.act_dtype != MVLDM_F32 && d.k_order == 1 && d.dst_dtype != MVLDM_F32 && d.n_pad % 320 == 0 &&
                           d.epilogue != MVLDM_EPI_GEGLU && d.upsample != 1 && d.n_out % 8 == 0 && (d.dst_ld <= 0 || d.dst_ld % 8 == 0) &&
                           !(d.ksize == 3 && d.src1);
        const int wgs10 = cdiv(M, 256) * (d.n_pad / 320);
        const double eff10 = (double)wgs10 / (256.0 * cdiv(wgs10, 256));
        // (Linears keep winning down to 2 ragged rounds: 908 vs 850 TFLOP/s at 576 workgroups; 3x3 convs do not)
        if (can10 && (wgs10 >= 1024 || (wgs10 >= 512 && (eff10 >= 0.9 || d.ksize == 1)))) tile = 10;
        else if (M <= 32) tile = 5;
        else if (M <= 64) tile = 4;
        else if (d.ksize >= 2) tile = cdiv(M, 256) * cdiv(d.n_pad, 128) >= 300 ? 7 : 2;
        else if (d.n_pad >= 768 && d.act_dtype != MVLDM_F32 && d.k_order == 1 && d.dst_dtype != MVLDM_F32 &&
                 cdiv(M, 256) * cdiv(d.n_pad, 256) >= 1024)
            tile = 9;   // wide Linear (QKV, GEGLU) with >= 4 rounds of workgroups: 256x256 measured +10-15 % over
                        // 128x256 / 64x128 (sweep4); at 1-2 rounds its quantisation loses 2x (N=1280 at 8x8)
        else if (d.k_pad >= 1024 && d.n_pad >= 1024 && cdiv(M, 128) * cdiv(d.n_pad, 256) >= 300) tile = 8;
        else tile = d.n_pad < 640 ? 2 : 3;
    }
    const int tm = cdiv(M, kTiles[tile].bm), tn = cdiv(d.n_pad, kTiles[tile].bn);
    if (splitk == 0) {
        splitk = 1;
        const int wgs = tm * tn;
        if (wgs < target) splitk = std::min(std::max(k_tiles / 4, 1), cdiv(target, wgs));
        if (d.workspace == nullptr) splitk = 1;
        while (splitk > 1 && (size_t)splitk * M * d.n_pad * sizeof(float) > ws_bytes) --splitk;
    }
    splitk = std::max(1, std::min(splitk, k_tiles));
}

static int fill_params(const mvldm_igemm_desc& d, IgemmParams& p, int& tile) {
    const int epc = d.act_dtype == MVLDM_F32 ? 4 : 8;
    const int bk = d.act_dtype == MVLDM_F32 ? 32 : 64;
    MVLDM_REQUIRE(d.src0 && d.weight && d.dst, "igemm: null pointer");
    // upsample: 0 none, 1 nearest-2x before a 3x3 conv (gather form), 2..5 = sub-pixel phase (py, px) = ((u-2)>>1, (u-2)&1) of the
    // same conv decomposed into four 2x2 convs on the low-resolution image (weights pre-summed by the caller)
    const bool phase = d.upsample >= 2;
    MVLDM_REQUIRE(d.upsample >= 0 && d.upsample <= 5, "igemm: upsample %d", d.upsample);
    MVLDM_REQUIRE(phase ? (d.ksize == 2 && d.stride == 1 && d.h_out == d.h_in && d.w_out == d.w_in && !d.src1 && !d.residual &&
                           !d.row_bias && d.act_dtype != MVLDM_F32 && d.dst_dtype == d.act_dtype && d.k_order == 1)
                        : (d.ksize == 1 || d.ksize == 3),
                  "igemm: ksize %d / upsample %d", d.ksize, d.upsample);
    MVLDM_REQUIRE(d.stride == 1 || d.stride == 2, "igemm: stride %d", d.stride);
    MVLDM_REQUIRE(d.c0 % epc == 0 && d.c1 % epc == 0 && d.c0 > 0, "igemm: channels (%d,%d) must be multiples of %d", d.c0, d.c1, epc);
    MVLDM_REQUIRE((d.c1 == 0) == (d.src1 == nullptr), "igemm: src1/c1 mismatch");
    MVLDM_REQUIRE(d.k_pad % bk == 0 && d.k_pad >= d.ksize * d.ksize * (d.c0 + d.c1), "igemm: k_pad %d", d.k_pad);
    MVLDM_REQUIRE(d.n_pad % 64 == 0 && d.n_pad >= d.n_out, "igemm: n_pad %d (n_out %d)", d.n_pad, d.n_out);
    MVLDM_REQUIRE(d.dst_dtype == d.act_dtype || d.dst_dtype == MVLDM_F32, "igemm: dst dtype");
    if (d.epilogue == MVLDM_EPI_GEGLU)
        MVLDM_REQUIRE(d.n_out % 64 == 0 && !d.row_bias, "igemm: GEGLU needs n_out %% 64 == 0");
    p.src0 = d.src0; p.src1 = d.src1; p.weight = d.weight; p.bias = d.bias; p.row_bias = d.row_bias;
    p.residual = d.residual; p.dst = d.dst; p.ws = d.workspace;
    p.c0 = d.c0; p.c1 = d.c1; p.ctot = d.c0 + d.c1;
    p.n_img = d.n_img; p.h_in = d.h_in; p.w_in = d.w_in; p.h_out = d.h_out; p.w_out = d.w_out;
    p.hw_out = d.h_out * d.w_out;
    p.ksize = d.ksize; p.stride = d.stride; p.pad = d.pad; p.upsample = phase ? 0 : d.upsample; p.taps = d.ksize * d.ksize;
    p.scatter = phase; p.ph_y = phase ? (d.upsample - 2) >> 1 : 0; p.ph_x = phase ? (d.upsample - 2) & 1 : 0;
    // tap t reads input pixel (oy*stride + ty0 + t/ks, ox*stride + tx0 + t%ks); (cy, cx) is a tap offset that is inside the image
    // for every output pixel: the centre of a padded 3x3 / the pixel itself for a 2x2 phase (whose taps start at py-1, px-1)
    p.ty0 = phase ? p.ph_y - 1 : -d.pad; p.tx0 = phase ? p.ph_x - 1 : -d.pad;
    p.cy = phase ? 1 - p.ph_y : d.ksize / 2; p.cx = phase ? 1 - p.ph_x : d.ksize / 2;
    p.M = d.n_img * p.hw_out; p.n_out = d.n_out; p.n_pad = d.n_pad; p.k_pad = d.k_pad;
    p.n_dst = d.epilogue == MVLDM_EPI_GEGLU ? d.n_out / 2 : d.n_out;
    p.rb_vec = d.row_bias && ((uintptr_t)d.row_bias % 16 == 0) && d.row_bias_ld % 4 == 0;
    p.bias_vec = d.bias && ((uintptr_t)d.bias % 16 == 0);
    p.row_bias_ld = d.row_bias_ld; p.epilogue = d.epilogue; p.dst_f32 = d.dst_dtype == MVLDM_F32;
    p.out_scale = d.out_scale;
    p.dst_ld = d.dst_ld > 0 ? d.dst_ld : p.n_dst;
    MVLDM_REQUIRE(p.dst_ld >= p.n_dst, "igemm: dst_ld %d < n_dst %d", p.dst_ld, p.n_dst);
    p.k_tiles = d.k_pad / bk;
    tile = d.tile & 63;
    t_force_sync = (d.tile >> 12) & 1;
    const int force_px = (d.tile >> 8) & 15;
    int splitk = d.splitk;
    MVLDM_REQUIRE(((tile >= 0 && tile <= kNumTiles) || tile == 17 || deep_tile(tile)) && splitk >= 0, "igemm: tile/splitk");
    choose_config(d, p.M, p.k_tiles, tile, splitk, d.workspace_bytes);
    // (a phase conv may split K like any other: its partial slabs are indexed by the LOW-resolution row and the reduce kernel scatters)
    if (splitk > 1)
        MVLDM_REQUIRE(d.workspace && (size_t)splitk * p.M * d.n_pad * sizeof(float) <= d.workspace_bytes,
                      "igemm: split-K workspace too small");
    // lean 16-bit loop: block-major K, extents addressable by a 32-bit buffer offset
    const double es = 2.0;
    const double b0 = (double)d.n_img * d.h_in * d.w_in * d.c0 * es, b1 = (double)d.n_img * d.h_in * d.w_in * d.c1 * es;
    const double bw = (double)d.n_pad * d.k_pad * es;
    p.use_bl = d.act_dtype != MVLDM_F32 && d.k_order == 1 && !t_force_sync && !kEnvSync &&
               b0 < 4.0e9 && b1 < 4.0e9 && bw < 4.0e9;
    if (d.act_dtype != MVLDM_F32 && d.k_order == 1 && !t_force_sync && !kEnvSync && !p.use_bl) {
        // a source or the weight is beyond the 32-bit buffer-offset range of the LDS-DMA loop: the launch is still correct
        // on the 64-bit-pointer register-prefetch loop, but 3-5x slower -- say so once (callers chunk the batch: vae._chunks)
        static bool warned = false;
        if (!warned) {
            warned = true;
            fprintf(stderr, "[mvldm] igemm: operand of %.2f GB exceeds the 4 GB range of the fast 16-bit loop; using the slower "
                            "64-bit-address loop (split the batch to avoid this)\n", std::max(std::max(b0, b1), bw) / 1e9);
        }
    }
    p.src0_bytes = (unsigned)b0; p.src1_bytes = (unsigned)b1; p.w_bytes = (unsigned)bw;
    p.fake = kEnvFake;
    // An output of half the 256 MB Infinity Cache or more is gone from every cache before its consumer starts: written with
    // streaming stores it does not evict the operand tiles the other workgroups are re-reading (-0.4 % per DDIM step at 64 scenes,
    // same-box A/B); small outputs (a few scenes) stay cacheable for the next op.  MVLDM_STREAM_STORES=0 / 1 forces it.
    p.nt_store = stream_stores((size_t)p.M * (size_t)p.n_dst * (p.dst_f32 || d.act_dtype == MVLDM_F32 ? 4 : 2));
    if (kEnvFake & 1) p.src0_bytes = p.src1_bytes = 0;   // EXPERIMENT ONLY: every A piece fails the range check (zeros, no L2 traffic)
    if (kEnvFake & 2) p.w_bytes = 0;                     // same for W
    if (tile >= 6 && !p.use_bl) tile = 2;
    if (deep_tile(tile) && d.upsample) tile = 2;      // (no per-tap tables / phase scatter in the deep-ring instantiations)
    p.splitk = splitk;
    if (p.use_bl) {   // splits own whole channel blocks (all taps of a block stay together)
        const int cbs = p.k_tiles / p.taps;
        const int per = cdiv(cbs, std::min(splitk, cbs));
        p.k_tiles_per_split = per * p.taps;
        p.splitk = cdiv(cbs, per);
    } else {
        p.k_tiles_per_split = cdiv(p.k_tiles, splitk);
        p.splitk = cdiv(p.k_tiles, p.k_tiles_per_split);  // drop empty splits
    }
    // LDS-staged epilogue: 16-byte rows need 8-column alignment of the 16-bit output (or a split-K slab)
    p.stage_epi = d.act_dtype != MVLDM_F32 && !kEnvNoStage && (p.splitk > 1 || (!p.dst_f32 && p.n_dst % 8 == 0 && p.dst_ld % 8 == 0));
    if (tile >= 9 && tile <= 10 && (!p.stage_epi || (tile == 10 && d.epilogue == MVLDM_EPI_GEGLU))) tile = 7;   // no per-element epilogue there; odd TN cannot pair GEGLU columns
    if (tile == 11 && !(p.use_bl && p.stage_epi && p.splitk == 1 && d.ksize == 3 && d.stride == 1 && d.pad == 1 && !d.upsample &&
                        d.h_out == d.h_in && d.w_out == d.w_in && halo_rows_for(d.w_in) <= 384 &&
                        2 * halo_rows_for(d.w_in) * 128 + 3 * 128 * 128 + 1152 <= 160 * 1024))
        tile = 7;   // the halo kernel only does 3x3 / stride 1 / pad 1 on images up to 63 pixels wide, one K pass
    if (tile == 17 && !(p.use_bl && p.stage_epi && p.splitk == 1 && d.ksize == 3 && d.stride == 1 && d.pad == 1 && !d.upsample && d.c1 == 0 &&
                        d.h_out == d.h_in && d.w_out == d.w_in && d.epilogue != MVLDM_EPI_GEGLU && halow_smem(d.w_in) <= 160 * 1024))
        tile = 7;   // the wide halo kernel: one-source 3x3 / stride 1 / pad 1 on maps up to 24 pixels wide, one K pass
    if (phase) {
        MVLDM_REQUIRE(p.use_bl && p.stage_epi, "igemm: 2x2 phase conv needs the lean 16-bit loop and an 8-aligned 16-bit output");
        if (tile != 7 && tile != 10) tile = 2;
    }
    if (p.use_bl && d.upsample == 1 && tile != 7) tile = 2;
    if (p.use_bl && d.upsample != 1) {
        // the lean loop addresses every tap relative to the centre tap: it must lie inside the image
        const int hc = (d.h_out - 1) * d.stride + p.ty0 + p.cy, wc = (d.w_out - 1) * d.stride + p.tx0 + p.cx;
        MVLDM_REQUIRE(p.ty0 + p.cy >= 0 && p.tx0 + p.cx >= 0 && hc < d.h_in && wc < d.w_in,
                      "igemm: conv geometry (pad %d, stride %d) not supported", d.pad, d.stride);
    }
    p.tiles_m = cdiv(p.M, kTiles[tile].bm);
    p.tiles_n = cdiv(d.n_pad, kTiles[tile].bn);
    p.korder = d.k_order;
    if (d.k_order)
        MVLDM_REQUIRE(d.k_order == 1 && p.ctot % bk == 0 && d.c0 % bk == 0 && d.k_pad == p.taps * p.ctot,
                      "igemm: k_order 1 needs channel counts (%d,%d) multiples of %d", d.c0, d.c1, bk);
    // XCD grid px x py (px * py = 8): fabric traffic ~ py * bytes(A) + px * bytes(W), idle tiles penalised
    {
        const double a_bytes = (double)d.n_img * d.h_in * d.w_in * p.ctot, w_bytes = (double)d.n_pad * d.k_pad;
        double best = 1e300;
        for (int px = 1; px <= 8; px *= 2) {
            const int py = 8 / px;
            const int sm = cdiv(p.tiles_m, px), sn = cdiv(p.tiles_n, py);
            const double waste = (double)(8 * sm * sn) / ((double)p.tiles_m * p.tiles_n);
            const double cost = (py * a_bytes + px * w_bytes) * waste * waste;
            if (cost < best) { best = cost; p.px = px; p.sub_m = sm; p.sub_n = sn; }
        }
        p.m_fast = w_bytes >= a_bytes;
        const int fpx = force_px ? force_px : kEnvPx;
        if (fpx > 0 && fpx <= 8 && (8 % fpx) == 0) {
            p.px = fpx; p.sub_m = cdiv(p.tiles_m, p.px); p.sub_n = cdiv(p.tiles_n, 8 / p.px);
        }
        // block of tiles the XCD's concurrently running workgroups cover (8-wave tiles: one workgroup per CU, 32 CUs): the shape that
        // minimises the partition's fabric traffic A_x * ceil(sub_n / gn) + W_x * ceil(sub_m / gm) with gm * gn = 32.
        // MVLDM_IGEMM_GROUP=0 keeps the one-row / one-column order (A/B knob), "gm" forces the row count.
        p.grp_m = p.grp_n = 1;
        static const int kGroup = knob_int("MVLDM_IGEMM_GROUP", -1);
        if (kGroup != 0 && ((tile >= 7 && tile <= 10) || tile == 17) && p.sub_m * p.sub_n > 32) {
            const double ax = a_bytes / p.px, wx = w_bytes / (8 / p.px);
            double bestc = 1e300;
            for (int gm = 1; gm <= 32; gm *= 2) {
                if (kGroup > 0 && gm != kGroup) continue;
                const int gme = std::min(gm, p.sub_m), gne = std::min(32 / gm, p.sub_n);
                const double c = ax * cdiv(p.sub_n, gne) + wx * cdiv(p.sub_m, gme);
                if (c < bestc) { bestc = c; p.grp_m = gme; p.grp_n = gne; }
            }
        }
    }
    return MVLDM_OK;
}

// tile 12: the persistent, epilogue-pipelined Linear kernel of linear_pp.hip; tile 13: the persistent wide Linear of linear_pw.hip
int linear_pp_run(const mvldm_igemm_desc& d, hipStream_t s);
int linear_pw_run(const mvldm_igemm_desc& d, hipStream_t s);
int linear_ws_run(const mvldm_igemm_desc& d, hipStream_t s);      // tile 14: weight-stationary Linear for K = 320 (linear_ws.hip)
int skinny_run(const mvldm_igemm_desc& d, hipStream_t s);         // tile 15: skinny-M weight-streaming GEMM on the fragment-order pack (skinny.hip)
int linear_rs_run(const mvldm_igemm_desc& d, hipStream_t s);      // tile 19: register-staged persistent Linear, 4 waves of 128 x 128 (linear_rs.hip)

int igemm_run(const mvldm_igemm_desc& d, hipStream_t s) {
    IgemmParams p;
    int tile = 0;
    if (d.n_img == 0 || d.h_out == 0 || d.w_out == 0) return MVLDM_OK;   // empty batch: nothing to do (its buffers may be null)
    if ((d.tile & 63) == 12) {
        MVLDM_REQUIRE(d.src0 && d.weight && d.dst, "igemm: null pointer");
        return linear_pp_run(d, s);
    }
    if ((d.tile & 63) == 13) {
        MVLDM_REQUIRE(d.src0 && d.weight && d.dst, "igemm: null pointer");
        return linear_pw_run(d, s);
    }
    if ((d.tile & 63) == 14) {
        MVLDM_REQUIRE(d.src0 && d.weight && d.dst, "igemm: null pointer");
        return linear_ws_run(d, s);
    }
    if ((d.tile & 63) == 15) return skinny_run(d, s);
    if ((d.tile & 63) == 19) {
        MVLDM_REQUIRE(d.src0 && d.weight && d.dst, "igemm: null pointer");
        return linear_rs_run(d, s);
    }
    MVLDM_REQUIRE(d.k_order != 2, "igemm: the fragment-order pack (k_order 2) is read by tile 15 only");
    int rc = fill_params(d, p, tile);
    if (rc) return rc;
    if (p.M == 0) return MVLDM_OK;
    rc = dispatch_dtype(d.act_dtype, [&](auto t) {
        using T = decltype(t);
        int r = launch_igemm<T>(p, tile, s);
        if (r) return r;
        if (p.splitk > 1) {
            if constexpr (sizeof(T) == 2) {
                if (!p.dst_f32 && p.n_dst % 8 == 0 && p.dst_ld % 8 == 0 && ((uintptr_t)p.ws % 16) == 0 &&
                    (!p.residual || ((uintptr_t)p.residual % 16) == 0) && ((uintptr_t)p.dst % 16) == 0) {
                    const size_t chunks = (size_t)p.M * (p.n_dst / 8);
                    hipLaunchKernelGGL(igemm_splitk_reduce_vec<T>, dim3((unsigned)std::min<size_t>((chunks + 255) / 256, 4096)), dim3(256), 0, s, p);
                    return check_launch();
                }
            }
            const size_t total = (size_t)p.M * p.n_dst;
            const int blocks = (int)std::min<size_t>((total + 255) / 256, 4096);
            hipLaunchKernelGGL(igemm_splitk_reduce<T>, dim3(blocks), dim3(256), 0, s, p);
            return check_launch();
        }
        return (int)MVLDM_OK;
    });
    return rc;
}

}  // namespace mvldm

using namespace mvldm;

extern "C" int mvldm_igemm_fwd(const mvldm_igemm_desc* d, mvldm_stream_t stream) {
    MVLDM_REQUIRE(d != nullptr, "igemm: null desc");
    return igemm_run(*d, (hipStream_t)stream);
}

extern "C" size_t mvldm_igemm_workspace_bytes(const mvldm_igemm_desc* d) {
    if (!d) return 0;
    const size_t M = (size_t)d->n_img * d->h_out * d->w_out;
    // the heuristic never splits deeper than ~512 workgroups' worth; 16 slabs is a safe ceiling
    return (size_t)16 * M * d->n_pad * sizeof(float);
}

static int pack_job_prepare(mvldm_pack_job& j, int dst_dtype) {
    const int bk = dst_dtype == MVLDM_F32 ? 32 : 64;
    MVLDM_REQUIRE(dst_dtype == MVLDM_F32 || dst_dtype == MVLDM_BF16 || dst_dtype == MVLDM_F16, "pack_weight: bad dtype %d", dst_dtype);
    MVLDM_REQUIRE(j.ksize >= 1 && j.n_out > 0 && j.c_in > 0, "pack_weight: bad dims");
    MVLDM_REQUIRE(j.k_order == 0 || (j.k_order == 1 && j.c_pad % bk == 0), "pack_weight: k_order 1 needs c_pad %% %d == 0", bk);
    if (j.transpose)
        MVLDM_REQUIRE(j.src && j.dst && !j.geglu && j.c_pad >= j.n_out && j.c_off >= 0 && j.n_rows > 0 && j.c_off + j.n_rows <= j.c_in &&
                      j.n_pad >= j.n_rows && j.k_pad >= j.ksize * j.ksize * j.c_pad, "pack_weight (transpose): bad dims");
    else
        MVLDM_REQUIRE(j.src && j.dst && j.c_pad >= j.c_in && j.n_pad >= j.n_out && j.k_pad >= j.ksize * j.ksize * j.c_pad, "pack_weight: bad dims");
    MVLDM_REQUIRE(!j.geglu || j.n_out % 64 == 0, "pack_weight: GEGLU needs n_out %% 64 == 0");
    const size_t total = (size_t)j.n_pad * j.k_pad;
    j.kind = PACK_GENERIC;
    j.blocks = (int)std::min<size_t>((total + 255) / 256, 65535);
    const int taps = j.ksize * j.ksize;
    // block-major 16-bit layout: the LDS-tiled packers (same bytes out as the generic body)
    if (dst_dtype != MVLDM_F32 && j.k_order == 1 && j.c_pad % 64 == 0 && j.k_pad == taps * j.c_pad && (j.ksize == 1 || j.ksize == 3)) {
        if (j.transpose) {
            const int rb = j.ksize == 3 ? 8 : 64;
            j.kind = j.ksize == 3 ? PACK_T_K3 : PACK_T_K1;
            j.blocks = (j.c_pad / 64) * ((j.n_pad + rb - 1) / rb);
        } else if (j.ksize == 3) {
            j.kind = PACK_FWD_K3;
            j.blocks = (j.c_pad / 64) * ((j.n_pad + 3) / 4);
        } else {
            const size_t quads = (size_t)j.n_pad * (j.k_pad / 4);
            j.kind = PACK_FWD_K1;
            j.blocks = (int)std::min<size_t>((quads + 255) / 256, 65535);
        }
    }
    return MVLDM_OK;
}

extern "C" int mvldm_pack_job_prepare(mvldm_pack_job* job, int dst_dtype) {
    MVLDM_REQUIRE(job != nullptr, "pack_job_prepare: null job");
    return pack_job_prepare(*job, dst_dtype);
}

extern "C" int mvldm_pack_weight_batch(const mvldm_pack_job* jobs, int n_jobs, const int32_t* block_job, int total_blocks, int dst_dtype,
                                       mvldm_stream_t stream) {
    MVLDM_REQUIRE(n_jobs >= 0 && total_blocks >= 0, "pack_weight_batch: negative count");
    if (n_jobs == 0 || total_blocks == 0) return MVLDM_OK;
    MVLDM_REQUIRE(jobs != nullptr, "pack_weight_batch: null job list");
    return dispatch_dtype(dst_dtype, [&](auto t) {
        using T = decltype(t);
        hipLaunchKernelGGL(pack_job_kernel<T>, dim3((unsigned)total_blocks), dim3(256), 0, (hipStream_t)stream, jobs, n_jobs, block_job, mvldm_pack_job{});
        return check_launch();
    });
}

extern "C" int mvldm_pack_weight(const float* src, void* dst, int n_out, int c_in, int ksize, int c_pad, int n_pad,
                                 int k_pad, int geglu, int k_order, int dst_dtype, int transpose, int c_off, int n_rows,
                                 mvldm_stream_t stream) {
    mvldm_pack_job j{src, dst, n_out, c_in, ksize, c_pad, n_pad, k_pad, geglu, k_order, transpose, c_off, n_rows, 0, 0, 0};
    const int rc = pack_job_prepare(j, dst_dtype);
    if (rc) return rc;
    return dispatch_dtype(dst_dtype, [&](auto t) {
        using T = decltype(t);
        hipLaunchKernelGGL(pack_job_kernel<T>, dim3((unsigned)j.blocks), dim3(256), 0, (hipStream_t)stream, (const mvldm_pack_job*)nullptr, 0, (const int32_t*)nullptr, j);
        return check_launch();
    });
}
