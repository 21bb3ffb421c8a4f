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
    int kdim;                       // taps * ctot (row length of a slab)
};

constexpr int WG_BN = 128, WG_BC = 64, WG_BP = 64;

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

template <typename T>
__global__ __launch_bounds__(256) void wgrad_kernel(const WgradParams p) {
    constexpr bool F32 = std::is_same<T, float>::value;
    constexpr int EPC = Elt<T>::EPC;
    // row pitches (elements): 16-bit tiles keep the 4 pixel rows of a transpose-read group on disjoint banks (+16 elements
    // = 32 bytes, the pitch rule of the attention V tile); fp32 rows are read as scalars (lanes = consecutive columns)
    constexpr int PD = F32 ? WG_BN + 4 : WG_BN + 16;
    constexpr int PX = F32 ? WG_BC + 4 : WG_BC + 16;
    constexpr int D_CH = WG_BN / EPC, X_CH = WG_BC / EPC;          // 16-byte chunks per tile row
    constexpr int D_IT = WG_BP * D_CH / 256, X_IT = WG_BP * X_CH / 256;
    __shared__ __attribute__((aligned(16))) T s_d[WG_BP * PD];
    __shared__ __attribute__((aligned(16))) T s_x[WG_BP * PX];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int hi = lane >> 5, l31 = lane & 31;
    int b = blockIdx.x;
    const int tc = b % p.tiles_c; b /= p.tiles_c;
    const int tap = b % p.taps;
    const int tn = b / p.taps;
    const int split = blockIdx.z;
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

int wgrad_plan(const mvldm_wgrad_desc& d, WgradParams& p, int& splits) {
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
    static const int kTarget = getenv("MVLDM_WGRAD_TARGET") ? atoi(getenv("MVLDM_WGRAD_TARGET")) : 512;   // workgroups aimed at (512 / 1024 / 2048 / 4096: 303 / 300 / 300 / 292 training views/s)
    splits = std::max(1, std::min(cdiv_(kTarget, tiles), std::max(1, blocks / 4)));
    const size_t slab = (size_t)d.n_out * p.kdim * sizeof(float);
    while (splits > 1 && (size_t)splits * slab > d.workspace_bytes) --splits;
    MVLDM_REQUIRE(d.workspace && (size_t)splits * slab <= d.workspace_bytes, "wgrad: workspace of %zu bytes too small (need >= %zu)",
                  d.workspace_bytes, slab);
    p.rows_per_split = cdiv_(blocks, splits) * WG_BP;
    splits = cdiv_(p.M, p.rows_per_split);
    return MVLDM_OK;
}

int wgrad_run(const mvldm_wgrad_desc& d, hipStream_t s) {
    if (d.n_img == 0 || d.h_out == 0 || d.w_out == 0 || d.n_out == 0) return MVLDM_OK;
    WgradParams p;
    int splits = 1;
    int rc = wgrad_plan(d, p, splits);
    if (rc) return rc;
    const dim3 grid(p.tiles_n * p.taps * p.tiles_c, 1, splits);
    rc = dispatch_dtype(d.act_dtype, [&](auto t) {
        using T = decltype(t);
        hipLaunchKernelGGL(wgrad_kernel<T>, grid, dim3(256), 0, s, p);
        return check_launch();
    });
    if (rc) return rc;
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
