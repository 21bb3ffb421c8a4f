// Weight gradient of the implicit-GEMM convolution / Linear (include/mvldm.h: mvldm_igemm_wgrad).
//
//   dW[n][tap][c] = sum_m dY[m][n] * A[m][(tap, c)]          A = the same on-the-fly gather as the forward pass
//
// i.e. a GEMM whose reduction index is the PIXEL index m: both operands are needed "transposed" (pixel-major in HBM,
// reduction-major in the MFMA fragments).  A workgroup owns a [128 n] x [64 c] output tile of ONE tap and walks the
// pixels 64 at a time: the dY tile [64 px][128 n] and the gathered X tile [64 px][64 c] are staged row-major in LDS
// (coalesced 16-byte global loads, zeros for padding taps / rows past M / columns past n_out) and consumed through the
// LDS transpose read ds_read_b64_tr_b16, which hands a lane 4 consecutive PIXELS of its own column -- exactly the
// K-major fragment of v_mfma_f32_32x32x16_{bf16,f16} (both operands see the same pixel permutation, so the
// contraction is exact).  fp32: v_mfma_f32_32x32x2_f32 reads its scalars straight from the row-major tiles.
// Register prefetch of the next pixel block overlaps the MFMAs of the current one.
//
// The pixel range is split over gridDim.z (split-K) so that small layers still fill 256 CUs; every split writes its
// fp32 partial tile to a workspace slab and `wgrad_reduce_kernel` folds the slabs in a fixed order (deterministic, no
// atomics) into the PyTorch-layout fp32 gradient ([n][c][ky][kx] / [n][c]), accumulating when asked to (micro-batches).
#include <algorithm>
#include <cstdlib>

#include "common.h"

namespace mvldm {

struct WgradParams {
    const void* src0; const void* src1; const void* dy; float* ws;
    int c0, c1, ctot;
    int n_img, h_in, w_in, h_out, w_out, hw_out;
    int ksize, stride, ty0, tx0, upsample, taps;
    int M, n_out, dy_ld;
    int tiles_n, tiles_c;           // grid.x = tiles_n * taps * tiles_c
    int rows_per_split;             // pixels per split (multiple of 64)
    int n_tiles, n_splits;          // 1-D grid of 8 * ceil(n_splits / 8) * n_tiles workgroups (wg_map)
    int xcd_map;                    // 0: tile fastest, split slowest (hardware round-robin over the XCDs)
    int kdim;                       // taps * ctot (row length of a slab)
};

constexpr int WG_BN = 128, WG_BC = 64, WG_BP = 64;

// workgroup -> (tile, split).  Default: tile fastest -- the tiles of a split (which walk the SAME pixel rows: dY rows for every
// tap and channel tile, X rows shifted by the tap) are dealt round-robin over the 8 XCDs and run at the same time, so a row
// comes out of HBM once and is handed to the eight L2s by the memory-side cache.  MVLDM_WGRAD_XCD=1 keeps a split on ONE XCD
// instead (split = 8 * group + b mod 8; fewer than 8 splits: split s owns the XCDs {x : x mod n_splits == s}): 4-7 % faster when the
// operands are already cache-resident (tools/wgrad_bench.py repeats one launch), 3-7 % SLOWER inside the training plan, where
// they come from HBM (wgrad total 17.6 -> 18.2 ms) -- measured, left off.
__device__ __forceinline__ bool wg_map(const WgradParams& p, int& tile, int& split) {
    const int b = blockIdx.x, xcd = b & 7, i = b >> 3;
    if (!p.xcd_map) {
        tile = b % p.n_tiles;
        split = b / p.n_tiles;
        return split < p.n_splits;
    }
    if (p.n_splits >= 8) {
        tile = i % p.n_tiles;
        split = (i / p.n_tiles) * 8 + xcd;
        return split < p.n_splits;
    }
    split = xcd % p.n_splits;
    const int rank = xcd / p.n_splits, cnt = (8 - split + p.n_splits - 1) / p.n_splits;      // this XCD's rank among the split's XCDs
    tile = i * cnt + rank;
    return tile < p.n_tiles;
}
static const int kWgXcd = knob_int("MVLDM_WGRAD_XCD", 0);
static inline int wg_grid(int tiles, int splits) {
    if (!kWgXcd) return tiles * splits;
    return splits >= 8 ? 8 * ((splits + 7) / 8) * tiles : 8 * ((tiles + (8 / splits) - 1) / (8 / splits));
}

template <typename T> struct WgMma;
template <> struct WgMma<bf16_t> {
    typedef __attribute__((ext_vector_type(4))) __bf16 Half;
    using Frag = bf16x8;
    static __device__ __forceinline__ Half tr(const bf16_t* p) { return __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) Half*)(p)); }
    static __device__ __forceinline__ f32x16 mma(Frag a, Frag b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0); }
};
template <> struct WgMma<f16_t> {
    typedef __attribute__((ext_vector_type(4))) _Float16 Half;
    using Frag = f16x8;
    static __device__ __forceinline__ Half tr(const f16_t* p) {
        typedef __attribute__((ext_vector_type(4))) __fp16 fp16x4_b;
        const fp16x4_b v = __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) fp16x4_b*)(p));
        return __builtin_bit_cast(Half, v);
    }
    static __device__ __forceinline__ f32x16 mma(Frag a, Frag b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0); }
};

constexpr int wg_tr_pitch(int cols) {        // elements (16-bit)
    int bytes = 2 * cols;
    while (bytes % 256 != 64 && bytes % 256 != 192) bytes += 32;
    return bytes / 2;
}

template <typename T>
__global__ __launch_bounds__(256) void wgrad_kernel(const WgradParams p) {
    constexpr bool F32 = std::is_same<T, float>::value;
    constexpr int EPC = Elt<T>::EPC;
    // row pitches (elements): a transpose-read half-wave touches 4 pixel rows x two 32-byte column pieces, conflict-free when the
    // pitch in bytes is 64 or 192 (mod 256) -- the rule attention.hip's v_pitch() measured.  (Rounds 2 - 3 used "+32 bytes" here:
    // 288 / 160 bytes, SQ_LDS_BANK_CONFLICT = 33 % of the LDS cycles of this kernel, tools/pmc_kernels.sh over tools/wgrad_bench.py.)
    // fp32 rows are read as scalars (lanes = consecutive columns).
    constexpr int PD = F32 ? WG_BN + 4 : wg_tr_pitch(WG_BN);
    constexpr int PX = F32 ? WG_BC + 4 : wg_tr_pitch(WG_BC);
    constexpr int D_CH = WG_BN / EPC, X_CH = WG_BC / EPC;          // 16-byte chunks per tile row
    constexpr int D_IT = WG_BP * D_CH / 256, X_IT = WG_BP * X_CH / 256;
    __shared__ __attribute__((aligned(16))) T s_d[WG_BP * PD];
    __shared__ __attribute__((aligned(16))) T s_x[WG_BP * PX];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int hi = lane >> 5, l31 = lane & 31;
    int b, split;
    if (!wg_map(p, b, split)) return;              // uniform per workgroup, before any barrier
    const int tc = b % p.tiles_c; b /= p.tiles_c;
    const int tap = b % p.taps;
    const int tn = b / p.taps;
    const int m_begin = split * p.rows_per_split, m_end = min(p.M, m_begin + p.rows_per_split);
    const int ky = tap / p.ksize, kx = tap - ky * p.ksize;
    const int hs = p.upsample ? 2 * p.h_in : p.h_in, wsz = p.upsample ? 2 * p.w_in : p.w_in;

    // loader coordinates (fixed per thread)
    int d_row[D_IT], d_col[D_IT], x_row[X_IT], x_col[X_IT];
#pragma unroll
    for (int j = 0; j < D_IT; ++j) { const int idx = tid + 256 * j; d_row[j] = idx / D_CH; d_col[j] = (idx - d_row[j] * D_CH) * EPC; }
#pragma unroll
    for (int j = 0; j < X_IT; ++j) { const int idx = tid + 256 * j; x_row[j] = idx / X_CH; x_col[j] = (idx - x_row[j] * X_CH) * EPC; }

    Chunk<T> rd[D_IT], rx[X_IT];
    auto load = [&](int m0) {
#pragma unroll
        for (int j = 0; j < D_IT; ++j) {
            const int m = m0 + d_row[j], n = tn * WG_BN + d_col[j];
            if (m < m_end && n < p.n_out) rd[j] = load_chunk<T>(reinterpret_cast<const T*>(p.dy) + (size_t)m * p.dy_ld + n);
            else rd[j].zero();
        }
#pragma unroll
        for (int j = 0; j < X_IT; ++j) {
            const int m = m0 + x_row[j], ch = tc * WG_BC + x_col[j];
            bool ok = m < m_end && ch < p.ctot;
            size_t pix = 0;
            if (ok) {
                if (p.ksize == 1 && p.stride == 1 && !p.upsample) {
                    pix = (size_t)m;
                } else {
                    const int img = m / p.hw_out, rem = m - img * p.hw_out;
                    const int oy = rem / p.w_out, ox = rem - oy * p.w_out;
                    int iy = oy * p.stride + p.ty0 + ky, ix = ox * p.stride + p.tx0 + kx;
                    ok = (unsigned)iy < (unsigned)hs && (unsigned)ix < (unsigned)wsz;
                    if (p.upsample) { iy >>= 1; ix >>= 1; }
                    pix = ((size_t)img * p.h_in + iy) * p.w_in + ix;
                }
            }
            if (ok) {
                const bool from0 = ch < p.c0;
                const T* sp = from0 ? reinterpret_cast<const T*>(p.src0) + pix * p.c0 + ch
                                    : reinterpret_cast<const T*>(p.src1) + pix * p.c1 + (ch - p.c0);
                rx[j] = load_chunk<T>(sp);
            } else {
                rx[j].zero();
            }
        }
    };
    auto store = [&]() {
#pragma unroll
        for (int j = 0; j < D_IT; ++j) *reinterpret_cast<u32x4*>(s_d + d_row[j] * PD + d_col[j]) = rd[j].raw;
#pragma unroll
        for (int j = 0; j < X_IT; ++j) *reinterpret_cast<u32x4*>(s_x + x_row[j] * PX + x_col[j]) = rx[j].raw;
    };

    // wave tile: 32 n (block `wave`) x 64 c (two 32-column blocks)
    f32x16 acc[2];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;

    if (m_begin < m_end) load(m_begin);
    for (int m0 = m_begin; m0 < m_end; m0 += WG_BP) {
        __syncthreads();          // the previous block's fragments have been read
        store();
        __syncthreads();
        if (m0 + WG_BP < m_end) load(m0 + WG_BP);
        if constexpr (F32) {
#pragma unroll 8
            for (int kk = 0; kk < WG_BP / 2; ++kk) {
                const int px = 2 * kk + hi;
                const float a = s_d[px * PD + wave * 32 + l31];
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, s_x[px * PX + j * 32 + l31], acc[j], 0, 0, 0);
            }
        } else {
            // transpose-read source of this lane: pixel row 4*(gi>>1) + (sl>>2) (+8 for the upper half), columns
            // (gi&1)*16 + 4*(sl&3) .. +3 of a 32-column block (gi = lane>>4, sl = lane&15) -- see attention.hip
            const int gi = lane >> 4, sl = lane & 15;
            const int prow = 4 * (gi >> 1) + (sl >> 2), pcol = (gi & 1) * 16 + 4 * (sl & 3);
            const T* da = s_d + prow * PD + wave * 32 + pcol;
            const T* xa = s_x + prow * PX + pcol;
#pragma unroll
            for (int ks = 0; ks < WG_BP / 16; ++ks) {
                const auto alo = WgMma<T>::tr(da + (ks * 16) * PD), aup = WgMma<T>::tr(da + (ks * 16 + 8) * PD);
                const typename WgMma<T>::Frag a = __builtin_shufflevector(alo, aup, 0, 1, 2, 3, 4, 5, 6, 7);
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const auto blo = WgMma<T>::tr(xa + (ks * 16) * PX + j * 32), bup = WgMma<T>::tr(xa + (ks * 16 + 8) * PX + j * 32);
                    const typename WgMma<T>::Frag bf = __builtin_shufflevector(blo, bup, 0, 1, 2, 3, 4, 5, 6, 7);
                    acc[j] = WgMma<T>::mma(a, bf, acc[j]);
                }
            }
        }
    }

    // partial tile -> slab [split][n][tap * ctot + c]
    float* slab = p.ws + (size_t)split * p.n_out * p.kdim;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int c = tc * WG_BC + j * 32 + l31;
        if (c >= p.ctot) continue;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int n = tn * WG_BN + wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
            if (n < p.n_out) slab[(size_t)n * p.kdim + tap * p.ctot + c] = acc[j][r];
        }
    }
}

// ---- 16-bit wide-tile form: LDS-DMA staging, [320 n] x [128 c] per tap and workgroup ----------------------------------
// The kernel above stages through registers (global -> VGPR -> ds_write_b128, 13 LDS cycles per 1 KB wave store) and owns a
// [128 n] x [64 c] tile: at 5 workgroups per CU the LDS pipe is busier with the tile WRITES than the matrix cores are with
// the MFMAs (op table of the training plan: 291 TFLOP/s), and n_out = 320 wastes 17 % of a 3 x 128 tile row.  Here:
//   * tile [320 n] x [128 c] of one tap, 8 waves, wave tile [160 n] x [32 c] (5 accumulator blocks): every channel count of
//     this UNet is a multiple of 320 on the n side; 93 flop per staged byte (was 43); 12 transpose reads per 5 MFMAs (was 3 per 1);
//   * both operand tiles go global -> LDS by `buffer_load_dwordx4 ... lds` (no staging registers, no ds_write) into a 2-slot
//     ring, one barrier per 64-pixel step; rows past the split's range, columns past n_out / c_tot and out-of-image taps use
//     an out-of-range buffer offset, which the descriptor's range check turns into zeros;
//   * row pitches 704 B (dY: 640 + 64) and 320 B (X: 256 + 64): the four pixel rows of a transpose-read half-wave land on
//     the four different 64-byte bank groups; the pad chunks are DMA lanes parked out of range;
//   * the pixel -> (y, x) split of the 3x3 gather is shifts and masks (the host takes this path for power-of-two maps only).
// Same slabs / same reduce kernel / same fragment permutation as above: bit-identical sums per split, different split count.
constexpr int WD_BN = 320, WD_BC = 128;
constexpr int WD_DP = 704, WD_XP = 320;                         // row pitches (bytes): one pixel row of both tiles = 1 KB = one DMA piece
constexpr int WD_RING = 128 * 1024;                             // LDS ring: BP pixels per slot, 128 / BP slots
static_assert(WD_DP + WD_XP == 1024, "a pixel row of the stage is one 1 KB piece");
constexpr unsigned kWdOob = 0xFFFFFFF0u;

struct WgradDmaParams {
    WgradParams w;
    unsigned x_bytes;               // extent of the activation tensor
    int lw, lh;                     // log2 of w_out / h_out (3x3 gather)
};

// issue the BP-pixel tile starting at pixel m0 into the ring slot at `stage`: BP pieces of 1 KB, BP / 8 per wave.  Piece index
// < BP * 704 / 1024 belongs to the dY region ([BP][704 B]), the rest to the X region ([BP][320 B]) behind it.
template <typename T, int BP>
__device__ __forceinline__ void wd_issue(const WgradDmaParams& q, char* stage, int wave, int m0, int m_end, int dy_disp,
                                         const unsigned (&dv)[BP / 8], const int (&xrow)[BP / 8], const unsigned (&xcol)[BP / 8], int ky, int kx) {
    constexpr int DQ = BP * WD_DP / 1024;
    const WgradParams& p = q.w;
    // dY: the descriptor starts at the tile's first pixel row and ends with the split's range
    const char* dbase = reinterpret_cast<const char*>(p.dy) + (size_t)m0 * (size_t)p.dy_ld * 2u;
    const size_t dleft = (size_t)(m_end - m0) * (size_t)p.dy_ld * 2u;
    const __amdgpu_buffer_rsrc_t rd = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(dbase), 0, dleft > 0xFFFFFFFFull ? 0xFFFFFFFFu : (unsigned)dleft, 0x00020000);
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.src0), 0, q.x_bytes, 0x00020000);
    const int wmask = p.w_out - 1, hmask = p.h_out - 1;
#pragma unroll
    for (int j = 0; j < BP / 8; ++j) {
        const int piece = wave + 8 * j;                            // wave-uniform
        __attribute__((address_space(3))) void* dst = (__attribute__((address_space(3))) void*)(stage + piece * 1024);
        if (piece < DQ) {
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rd, dst, 16, dv[j], 0, 0, 0);
        } else {
            const int m = m0 + xrow[j];
            bool ok = m < m_end;
            if (p.ksize == 3) {
                const int ox = m & wmask, oy = (m >> q.lw) & hmask;
                ok = ok && (unsigned)(oy + ky - 1) < (unsigned)p.h_out && (unsigned)(ox + kx - 1) < (unsigned)p.w_out;
            }
            const unsigned v = (ok && xcol[j] != kWdOob) ? (unsigned)(m + dy_disp) * (unsigned)p.ctot * 2u + xcol[j] : kWdOob;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, dst, 16, v, 0, 0, 0);
        }
    }
}

template <typename T, int BP>
__global__ __launch_bounds__(512) void wgrad_dma_kernel(const WgradDmaParams q) {
    static_assert(sizeof(T) == 2, "wide-tile weight gradient: 16-bit activations");
    static_assert(BP == 64 || BP == 48 || BP == 32, "64-pixel slots x 2, 48 x 3 or 32 x 4");
    constexpr int STAGE = BP * 1024, NSLOT = BP == 48 ? 3 : WD_RING / STAGE, PW = BP / 8;        // pieces per wave and tile
    constexpr int DQ = BP * WD_DP / 1024;
    const WgradParams& p = q.w;
    extern __shared__ __attribute__((aligned(16))) char wd_smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int hi = lane >> 5, l31 = lane & 31;
    int b, split;
    if (!wg_map(p, b, split)) return;              // uniform per workgroup, before any barrier
    const int tc = b % p.tiles_c; b /= p.tiles_c;
    const int tap = b % p.taps;
    const int tn = b / p.taps;
    const int m_begin = split * p.rows_per_split, m_end = min(p.M, m_begin + p.rows_per_split);
    const int ky = tap / p.ksize, kx = tap - ky * p.ksize;
    const int disp = p.ksize == 3 ? (ky - 1) * p.w_out + (kx - 1) : 0;

    // per-lane slots of this wave's DMA pieces (fixed over the pixel loop)
    unsigned dv[PW], xcol[PW];
    int xrow[PW];
#pragma unroll
    for (int j = 0; j < PW; ++j) {
        const int piece = wave + 8 * j;
        dv[j] = kWdOob; xcol[j] = kWdOob; xrow[j] = 0;
        if (piece < DQ) {
            const int s = piece * 64 + lane, row = s / (WD_DP / 16), ch = s - row * (WD_DP / 16);
            const int n = tn * WD_BN + ch * 8;
            if (ch < WD_BN / 8 && n < p.n_out) dv[j] = ((unsigned)row * (unsigned)p.dy_ld + (unsigned)n) * 2u;
        } else {
            const int s = (piece - DQ) * 64 + lane, row = s / (WD_XP / 16), ch = s - row * (WD_XP / 16);
            const int c = tc * WD_BC + ch * 8;
            xrow[j] = row;
            if (ch < WD_BC / 8 && c < p.ctot) xcol[j] = (unsigned)c * 2u;
        }
    }

    const int wn = wave >> 2, wc = wave & 3;                       // wave tile: n rows [wn*160, +160), c columns [wc*32, +32)
    f32x16 acc[5];
#pragma unroll
    for (int i = 0; i < 5; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    // transpose-read source of this lane (see wgrad_kernel): pixel row 4*(gi>>1) + (sl>>2) (+8), columns (gi&1)*16 + 4*(sl&3) .. +3
    const int gi = lane >> 4, sl = lane & 15;
    const int prow = 4 * (gi >> 1) + (sl >> 2), pcol = (gi & 1) * 16 + 4 * (sl & 3);
    const int d_off = prow * WD_DP + (wn * 160 + pcol) * 2, x_off = BP * WD_DP + prow * WD_XP + (wc * 32 + pcol) * 2;

    // ring: tiles t+1 .. t+NSLOT-1 are in flight while tile t is consumed.  Fragments are double-buffered (fa / fb): the last
    // 16-pixel group of tile t is multiplied AFTER the barrier that publishes tile t+1, under the DMA issue of tile t+NSLOT and
    // the LDS latency of tile t+1's first fragments (the igemm loop's arrangement: one wave covers the latency by itself).
    using Frag = typename WgMma<T>::Frag;
    struct Frags { Frag a[5], b; };
    auto load_frags = [&](Frags& f, const char* base, int ks) __attribute__((always_inline)) {
        const char* xa = base + x_off + ks * 16 * WD_XP;
        const auto blo = WgMma<T>::tr(reinterpret_cast<const T*>(xa)), bup = WgMma<T>::tr(reinterpret_cast<const T*>(xa + 8 * WD_XP));
        f.b = __builtin_shufflevector(blo, bup, 0, 1, 2, 3, 4, 5, 6, 7);
#pragma unroll
        for (int i = 0; i < 5; ++i) {
            const char* da = base + d_off + ks * 16 * WD_DP + i * 64;
            const auto alo = WgMma<T>::tr(reinterpret_cast<const T*>(da)), aup = WgMma<T>::tr(reinterpret_cast<const T*>(da + 8 * WD_DP));
            f.a[i] = __builtin_shufflevector(alo, aup, 0, 1, 2, 3, 4, 5, 6, 7);
        }
    };
    auto mma = [&](const Frags& f) __attribute__((always_inline)) {
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < 5; ++i) acc[i] = WgMma<T>::mma(f.a[i], f.b, acc[i]);
        __builtin_amdgcn_sched_barrier(0);
    };
    constexpr int NKS = BP / 16;
    const int n_tile = (m_end - m_begin + BP - 1) / BP;
#pragma unroll
    for (int k = 0; k < NSLOT - 1; ++k)
        if (k < n_tile) wd_issue<T, BP>(q, wd_smem + k * STAGE, wave, m_begin + k * BP, m_end, disp, dv, xrow, xcol, ky, kx);
    Frags fa, fb;
    if (n_tile > 0) {
        // tile 0 has landed (the NSLOT - 2 younger ones may still be in flight)
        if (NSLOT == 4 && n_tile > 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * PW) : "memory");
        else if (NSLOT >= 3 && n_tile > 1 && !(NSLOT == 4 && n_tile > 2)) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PW) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (NSLOT - 1 < n_tile) wd_issue<T, BP>(q, wd_smem + (NSLOT - 1) * STAGE, wave, m_begin + (NSLOT - 1) * BP, m_end, disp, dv, xrow, xcol, ky, kx);
        load_frags(fa, wd_smem, 0);
    }
    int slot = 0;
    for (int t = 0; t < n_tile; ++t) {
        const char* base = wd_smem + slot * STAGE;
        // groups 0 .. NKS-2 of tile t, each under the fragment reads of the next group (NKS is even or odd: ping-pong by parity)
#pragma unroll
        for (int ks = 0; ks + 1 < NKS; ++ks) {
            if (ks & 1) { load_frags(fa, base, ks + 1); mma(fb); }
            else { load_frags(fb, base, ks + 1); mma(fa); }
        }
        Frags& last = ((NKS - 1) & 1) ? fb : fa;            // holds group NKS-1 of tile t
        Frags& next = ((NKS - 1) & 1) ? fa : fb;
        const int nslot = slot + 1 == NSLOT ? 0 : slot + 1;
        if (t + 1 < n_tile) {
            // tile t+1 must have landed (younger: t+2 .. t+NSLOT-1 if they exist); every wave's reads of tile t are done (lgkmcnt)
            const int younger = min(n_tile - 2 - t, NSLOT - 2);
            if (NSLOT == 4 && younger == 2) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(2 * PW) : "memory");
            else if (NSLOT >= 3 && younger == 1) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(PW) : "memory");
            else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            if (t + NSLOT < n_tile)        // tile t's slot is free now
                wd_issue<T, BP>(q, wd_smem + slot * STAGE, wave, m_begin + (t + NSLOT) * BP, m_end, disp, dv, xrow, xcol, ky, kx);
            load_frags(next, wd_smem + nslot * STAGE, 0);
        }
        mma(last);
        if constexpr ((NKS & 1) != 0) {       // odd group count: tile t+1's group 0 sits in the other set -- swap roles by copying
            if (t + 1 < n_tile) fa = next;
        }
        slot = nslot;
    }

    // partial tile -> slab [split][n][tap * ctot + c]
    float* slab = p.ws + (size_t)split * p.n_out * p.kdim;
    const int c = tc * WD_BC + wc * 32 + l31;
    if (c < p.ctot) {
#pragma unroll
        for (int i = 0; i < 5; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int n = tn * WD_BN + wn * 160 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
                if (n < p.n_out) slab[(size_t)n * p.kdim + tap * p.ctot + c] = acc[i][r];
            }
    }
}

// grad[n][c][tap] (PyTorch [n_out][c_in][k][k]; c < c_in: padding channels are dropped) = / += sum_splits slab[s][n][tap*ctot + c]
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* __restrict__ ws, float* __restrict__ grad, int n_out, int c_in, int ctot,
                                                           int taps, int splits, int accumulate) {
    const size_t total = (size_t)n_out * c_in * taps;
    const size_t kdim = (size_t)taps * ctot, slab = (size_t)n_out * kdim;
    for (size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (size_t)gridDim.x * 256) {
        const int tap = (int)(idx % taps);
        const size_t nc = idx / taps;
        const int c = (int)(nc % c_in), n = (int)(nc / c_in);
        const float* s = ws + (size_t)n * kdim + (size_t)tap * ctot + c;
        float t = 0.f;
        for (int k = 0; k < splits; ++k) t += s[(size_t)k * slab];
        grad[idx] = accumulate ? grad[idx] + t : t;
    }
}

static inline int cdiv_(int a, int b) { return (a + b - 1) / b; }

int wgrad_plan(const mvldm_wgrad_desc& d, WgradParams& p, int& splits, int tcode = 0) {
    const int epc = d.act_dtype == MVLDM_F32 ? 4 : 8;
    MVLDM_REQUIRE(d.src0 && d.dy && d.grad, "wgrad: null pointer");
    MVLDM_REQUIRE(d.ksize == 1 || d.ksize == 3, "wgrad: ksize %d", d.ksize);
    MVLDM_REQUIRE(d.stride == 1 || d.stride == 2, "wgrad: stride %d", d.stride);
    MVLDM_REQUIRE(d.upsample == 0 || d.upsample == 1, "wgrad: upsample %d (phase convs take the gather form here)", d.upsample);
    MVLDM_REQUIRE(d.c0 > 0 && d.c0 % epc == 0 && d.c1 % epc == 0 && (d.c1 == 0) == (d.src1 == nullptr), "wgrad: channels (%d,%d)", d.c0, d.c1);
    MVLDM_REQUIRE(d.c_in > 0 && d.c_in <= d.c0 + d.c1, "wgrad: c_in %d", d.c_in);
    MVLDM_REQUIRE(d.n_out > 0 && d.dy_ld >= d.n_out && d.dy_ld % epc == 0, "wgrad: n_out %d / dy_ld %d", d.n_out, d.dy_ld);
    p.src0 = d.src0; p.src1 = d.src1; p.dy = d.dy; p.ws = d.workspace;
    p.c0 = d.c0; p.c1 = d.c1; p.ctot = d.c0 + d.c1;
    p.n_img = d.n_img; p.h_in = d.h_in; p.w_in = d.w_in; p.h_out = d.h_out; p.w_out = d.w_out; p.hw_out = d.h_out * d.w_out;
    p.ksize = d.ksize; p.stride = d.stride; p.ty0 = -d.pad; p.tx0 = -d.pad; p.upsample = d.upsample; p.taps = d.ksize * d.ksize;
    p.M = d.n_img * p.hw_out; p.n_out = d.n_out; p.dy_ld = d.dy_ld;
    p.tiles_n = cdiv_(d.n_out, WG_BN); p.tiles_c = cdiv_(p.ctot, WG_BC);
    p.kdim = p.taps * p.ctot;
    const int tiles = p.tiles_n * p.taps * p.tiles_c;
    const int blocks = cdiv_(p.M, WG_BP);
    static const int kTarget = knob_int("MVLDM_WGRAD_TARGET", 512);   // workgroups aimed at (512 / 1024 / 2048 / 4096: 303 / 300 / 300 / 292 training views/s)
    // (bits 10-12 of the caller's `accumulate` -- the host's per-problem choice, plan.autotune_wgrad -- name the target: 64 << code)
    const int target = tcode ? (64 << tcode) : kTarget;
    splits = std::max(1, std::min(cdiv_(target, tiles), std::max(1, blocks / 4)));
    const size_t slab = (size_t)d.n_out * p.kdim * sizeof(float);
    while (splits > 1 && (size_t)splits * slab > d.workspace_bytes) --splits;
    MVLDM_REQUIRE(d.workspace && (size_t)splits * slab <= d.workspace_bytes, "wgrad: workspace of %zu bytes too small (need >= %zu)",
                  d.workspace_bytes, slab);
    p.rows_per_split = cdiv_(blocks, splits) * WG_BP;
    splits = cdiv_(p.M, p.rows_per_split);
    return MVLDM_OK;
}

// One split of a Linear whose gradient is to be WRITTEN (accumulate 0: the first write of a store-first window plan, train.py): the slab
// [n][tap * ctot + c] of the only split IS the PyTorch layout [n_out][c_in] -- the kernel writes the gradient itself, no slab, no reduce
// launch (the big 1280-wide FF / QKV projections: 52 MB slabs that were written, re-read and written again).  MVLDM_WGRAD_DIRECT=0: A/B knob.
static inline bool wgrad_direct(const mvldm_wgrad_desc& d, const WgradParams& p, int splits) {
    static const int on = knob_int("MVLDM_WGRAD_DIRECT", 1);
    return on && splits == 1 && p.taps == 1 && d.c_in == p.ctot && (d.accumulate & 1) == 0;
}

static inline int ilog2_exact(int v) {       // log2 of a power of two, -1 otherwise
    if (v <= 0 || (v & (v - 1))) return -1;
    int l = 0;
    while ((1 << l) < v) ++l;
    return l;
}

// the wide-tile LDS-DMA form: 16-bit, one source, stride 1, no upsampling, 1x1 or a 3x3 "same" conv over a power-of-two map,
// enough rows / columns to fill the tile.  Everything else (f32, conv_in, conv_out, downsamplers, skip-concat shortcuts) keeps
// the register-staged kernel.  MVLDM_WGRAD_WIDE=0 forces the old form (A/B knob).
static bool wgrad_wide_ok(const mvldm_wgrad_desc& d) {
    if (d.act_dtype == MVLDM_F32 || d.src1 || d.c1 || d.stride != 1 || d.upsample) return false;
    if (d.n_out < 160 || d.c0 < 64 || d.c0 % 8 || d.dy_ld % 8) return false;
    if (d.ksize == 3 && (d.pad != 1 || d.h_in != d.h_out || d.w_in != d.w_out || ilog2_exact(d.w_out) < 0 || ilog2_exact(d.h_out) < 0)) return false;
    if (d.ksize == 1 && (d.pad != 0 || d.h_in != d.h_out || d.w_in != d.w_out)) return false;
    const size_t xb = (size_t)d.n_img * d.h_in * d.w_in * d.c0 * 2u;
    return xb < 0xFFFFFFF0ull;
}

static int wgrad_run_wide(const mvldm_wgrad_desc& d, hipStream_t s, int tcode = 0) {
    WgradDmaParams q;
    WgradParams& p = q.w;
    int splits = 1;
    int rc = wgrad_plan(d, p, splits);          // argument checks + the common fields
    if (rc) return rc;
    p.tiles_n = cdiv_(d.n_out, WD_BN);
    p.tiles_c = cdiv_(p.ctot, WD_BC);
    q.x_bytes = (unsigned)((size_t)d.n_img * d.h_in * d.w_in * d.c0 * 2u);
    q.lw = d.ksize == 3 ? ilog2_exact(d.w_out) : 0;
    q.lh = d.ksize == 3 ? ilog2_exact(d.h_out) : 0;
    // one workgroup per CU: whole rounds of 256, at least 4 pixel steps per split
    static const int kBp = knob_int("MVLDM_WGRAD_WIDE_BP", 64);      // 64: 2-slot ring (default), 32: 4-slot (measured 18 % slower)
    const int tiles = p.tiles_n * p.taps * p.tiles_c, blocks = cdiv_(p.M, 64);
    static const int kTarget = knob_int("MVLDM_WGRAD_WIDE_TARGET", 256);      // one round of workgroups: half the slab traffic of two (512: 2324 us over tools/wgrad_bench.py, 256: 2190)
    const int target = tcode ? (64 << tcode) : kTarget;
    splits = std::max(1, std::min(target / std::max(tiles, 1), std::max(1, blocks / 4)));
    const size_t slab = (size_t)d.n_out * p.kdim * sizeof(float);
    while (splits > 1 && (size_t)splits * slab > d.workspace_bytes) --splits;
    MVLDM_REQUIRE((size_t)splits * slab <= d.workspace_bytes, "wgrad: workspace of %zu bytes too small (need >= %zu)", d.workspace_bytes, slab);
    p.rows_per_split = cdiv_(blocks, splits) * 64;
    splits = cdiv_(p.M, p.rows_per_split);
    p.n_tiles = tiles;
    p.n_splits = splits;
    p.xcd_map = kWgXcd;
    const bool direct = wgrad_direct(d, p, splits);
    if (direct) p.ws = d.grad;
    const dim3 grid(wg_grid(tiles, splits));
    static std::atomic<uint64_t> done_b{0}, done_h{0}, done_b32{0}, done_h32{0}, done_b48{0}, done_h48{0};
    rc = dispatch_dtype(d.act_dtype, [&](auto t) {
        using T = decltype(t);
        if constexpr (sizeof(T) == 2) {
            constexpr bool B16 = std::is_same<T, bf16_t>::value;
            if (kBp == 64) {
                auto* kern = wgrad_dma_kernel<T, 64>;
                if (int rc0 = ensure_dyn_smem(reinterpret_cast<const void*>(kern), WD_RING, B16 ? done_b : done_h)) return rc0;
                hipLaunchKernelGGL(kern, grid, dim3(512), WD_RING, s, q);
            } else if (kBp == 48) {
                auto* kern = wgrad_dma_kernel<T, 48>;
                if (int rc0 = ensure_dyn_smem(reinterpret_cast<const void*>(kern), 3 * 48 * 1024, B16 ? done_b48 : done_h48)) return rc0;
                hipLaunchKernelGGL(kern, grid, dim3(512), 3 * 48 * 1024, s, q);
            } else {
                auto* kern = wgrad_dma_kernel<T, 32>;
                if (int rc0 = ensure_dyn_smem(reinterpret_cast<const void*>(kern), WD_RING, B16 ? done_b32 : done_h32)) return rc0;
                hipLaunchKernelGGL(kern, grid, dim3(512), WD_RING, s, q);
            }
            return check_launch();
        } else {
            return set_error(MVLDM_ERR_ARG, "wgrad: wide form is 16-bit only");
        }
    });
    if (rc) return rc;
    if (direct) return MVLDM_OK;
    const size_t total = (size_t)d.n_out * d.c_in * p.taps;
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)std::min<size_t>((total + 255) / 256, 16384)), dim3(256), 0, s, d.workspace, d.grad, d.n_out,
                       d.c_in, p.ctot, p.taps, splits, d.accumulate);
    return check_launch();
}

// which form?  `accumulate` bits 8-9 carry the caller's choice (1 = register-staged, 2 = wide; the host's plan-time selection
// times both, plan.autotune_wgrad); 0 = the rule below, from tools/wgrad_bench.py at 32 images (profiles/r03_wgrad_forms.txt):
// the wide form wins on the 3x3 convs with long pixel ranges (level 0: 1.7x) and with short ones (<= 4096 pixels, many tiles),
// and on Linears with K >= 1024 and few output columns; the 16x16 level and the wide-N Linears stay with the small tile.
static bool wgrad_pick_wide(const mvldm_wgrad_desc& d) {
    static const int force = knob_int("MVLDM_WGRAD_WIDE", -1);      // A/B knob: 0 never, 1 wherever it applies
    const int form = (d.accumulate >> 8) & 3;
    if (!wgrad_wide_ok(d)) return false;
    if (form) return form == 2;
    if (force >= 0) return force != 0;
    const long m = (long)d.n_img * d.h_out * d.w_out;
    if (d.ksize == 3) return m >= 16384 || m <= 4096;
    return d.c0 >= 1024 && d.n_out <= 640;
}

int wgrad_run(const mvldm_wgrad_desc& d0, hipStream_t s) {
    if (d0.n_img == 0 || d0.h_out == 0 || d0.w_out == 0 || d0.n_out == 0) return MVLDM_OK;
    const int form = (d0.accumulate >> 8) & 3;
    MVLDM_REQUIRE(form != 2 || wgrad_wide_ok(d0), "wgrad: the wide form does not take this problem");
    const bool wide = wgrad_pick_wide(d0);
    const int tcode = (d0.accumulate >> 10) & 7;      // split target chosen by the host (0: the library's)
    mvldm_wgrad_desc d = d0;
    d.accumulate &= 1;
    if (wide) return wgrad_run_wide(d, s, tcode);
    WgradParams p;
    int splits = 1;
    int rc = wgrad_plan(d, p, splits, tcode);
    if (rc) return rc;
    p.n_tiles = p.tiles_n * p.taps * p.tiles_c;
    p.n_splits = splits;
    p.xcd_map = kWgXcd;
    const bool direct = wgrad_direct(d, p, splits);
    if (direct) p.ws = d.grad;
    const dim3 grid(wg_grid(p.n_tiles, splits));
    rc = dispatch_dtype(d.act_dtype, [&](auto t) {
        using T = decltype(t);
        hipLaunchKernelGGL(wgrad_kernel<T>, grid, dim3(256), 0, s, p);
        return check_launch();
    });
    if (rc) return rc;
    if (direct) return MVLDM_OK;
    const size_t total = (size_t)d.n_out * d.c_in * p.taps;
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)std::min<size_t>((total + 255) / 256, 16384)), dim3(256), 0, s, d.workspace, d.grad, d.n_out,
                       d.c_in, p.ctot, p.taps, splits, d.accumulate);
    return check_launch();
}

}  // namespace mvldm

extern "C" int mvldm_igemm_wgrad(const mvldm_wgrad_desc* d, mvldm_stream_t stream) {
    MVLDM_REQUIRE(d != nullptr, "wgrad: null desc");
    return mvldm::wgrad_run(*d, (hipStream_t)stream);
}
