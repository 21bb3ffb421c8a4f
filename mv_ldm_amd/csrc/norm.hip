// GroupNorm(+SiLU) and LayerNorm for NHWC / token-major activations (see include/mvldm.h).
//
// GroupNorm: ONE launch (`gn_fused_kernel`, below) when an (image, channel-span) slab fits a workgroup's registers --
// every case of the UNet; otherwise (VAE at 256x256) two launches.  (1) `gn_stats`: each workgroup streams a contiguous slab of rows of one
// image with 16-byte loads (fully coalesced), every thread owning one fixed 16-byte channel chunk
// column so its per-channel sum / sum-of-squares stay in registers (fp64: the variance is exact to
// double round-off, no E[x^2]-E[x]^2 cancellation issue), then folds channels into the 32 groups
// through LDS and writes one partial per (image, slab, group).  (2) `gn_apply`: re-reads the slab
// (L2/MALL resident at these sizes), turns the partials into per-channel scale/shift in LDS and
// writes normalised (+SiLU) activations.  HBM-bound: algorithmic traffic = 2 reads + 1 write.
//
// LayerNorm: one wave per token row, row held in registers, two-pass mean / centred variance with
// wave64 butterfly reductions, fp32.
#include "common.h"
#include <algorithm>
#include <cstdlib>

namespace mvldm {

// ---------------------------------------------------------------------------------------------- GN
template <typename T>
__global__ __launch_bounds__(256) void gn_stats_kernel(const T* __restrict__ x0, const T* __restrict__ x1, int c0,
                                                       double* __restrict__ partial, int hw, int c, int groups,
                                                       int rows_per_chunk, int nchunk) {
    constexpr int EPC = Elt<T>::EPC;
    __shared__ double s_sum[64], s_sq[64];  // groups <= 64
    const int img = blockIdx.x / nchunk, chunk = blockIdx.x % nchunk;
    const int ncc = c / EPC;                 // 16-byte chunk columns per row
    const int cpg = c / groups;
    const int tid = threadIdx.x;
    if (tid < 64) { s_sum[tid] = 0.0; s_sq[tid] = 0.0; }
    __syncthreads();
    const int r_begin = chunk * rows_per_chunk, r_end = min(hw, r_begin + rows_per_chunk);
    // thread -> (row lane, chunk column); the block covers R rows per sweep.  If ncc > blockDim the
    // columns themselves are swept as well.
    const int nthr = blockDim.x;
    for (int cc0 = 0; cc0 < ncc; cc0 += nthr) {
        const int span = min(ncc - cc0, nthr);   // columns handled in this sweep
        const int R = max(1, nthr / span);
        const int cc = cc0 + tid % span, rl = tid / span;
        double s[EPC], q[EPC];
#pragma unroll
        for (int i = 0; i < EPC; ++i) { s[i] = 0.0; q[i] = 0.0; }
        if (rl < R) {
            const int ch0 = cc * EPC;
            const bool first = ch0 < c0;
            const int cs = first ? c0 : c - c0;
            const T* base = (first ? x0 + ch0 : x1 + (ch0 - c0)) + ((size_t)img * hw) * cs;
            for (int r = r_begin + rl; r < r_end; r += R) {
                const Chunk<T> v = load_chunk<T>(base + (size_t)r * cs);
#pragma unroll
                for (int i = 0; i < EPC; ++i) {
                    const double f = (double)v.get(i);
                    s[i] += f;
                    q[i] += f * f;
                }
            }
#pragma unroll
            for (int i = 0; i < EPC; ++i) {
                const int g = (cc * EPC + i) / cpg;
                atomicAdd(&s_sum[g], s[i]);
                atomicAdd(&s_sq[g], q[i]);
            }
        }
    }
    __syncthreads();
    if (tid < groups) {
        double* o = partial + (((size_t)img * nchunk + chunk) * groups + tid) * 2;
        o[0] = s_sum[tid];
        o[1] = s_sq[tid];
    }
}

template <typename T>
__global__ __launch_bounds__(256) void gn_apply_kernel(const T* __restrict__ x0, const T* __restrict__ x1, int c0,
                                                       T* __restrict__ y,
                                                       const float* __restrict__ gamma, const float* __restrict__ beta,
                                                       const double* __restrict__ partial, int hw, int c, int groups,
                                                       int rows_per_chunk, int nchunk, float eps, int silu, float* __restrict__ stats_out) {
    constexpr int EPC = Elt<T>::EPC;
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    float* s_scale = reinterpret_cast<float*>(smem_raw);  // [c]
    float* s_shift = s_scale + c;                          // [c]
    float* s_mean = s_shift + c;                           // [groups]
    float* s_rstd = s_mean + groups;                       // [groups]
    const int img = blockIdx.x / nchunk, chunk = blockIdx.x % nchunk;
    const int tid = threadIdx.x;
    const int cpg = c / groups;
    if (tid < groups) {
        double s = 0.0, q = 0.0;
        for (int k = 0; k < nchunk; ++k) {
            const double* pp = partial + (((size_t)img * nchunk + k) * groups + tid) * 2;
            s += pp[0];
            q += pp[1];
        }
        const double n = (double)hw * cpg;
        const double mean = s / n;
        double var = q / n - mean * mean;
        var = var > 0.0 ? var : 0.0;
        s_mean[tid] = (float)mean;
        s_rstd[tid] = (float)(1.0 / sqrt(var + (double)eps));
        if (stats_out && chunk == 0) {       // (mean, rstd) per (image, group): what the backward pass needs
            stats_out[((size_t)img * groups + tid) * 2] = s_mean[tid];
            stats_out[((size_t)img * groups + tid) * 2 + 1] = s_rstd[tid];
        }
    }
    __syncthreads();
    for (int ch = tid; ch < c; ch += blockDim.x) {
        const int g = ch / cpg;
        const float sc = s_rstd[g] * gamma[ch];
        s_scale[ch] = sc;
        s_shift[ch] = beta[ch] - s_mean[g] * sc;
    }
    __syncthreads();
    const int ncc = c / EPC;
    const int r_begin = chunk * rows_per_chunk, r_end = min(hw, r_begin + rows_per_chunk);
    const size_t total = (size_t)(r_end - r_begin) * ncc;
    const size_t row0 = (size_t)img * hw + r_begin;
    T* yb = y + row0 * c;
    const int c1 = c - c0;
    (void)total;
    // thread -> one fixed 16-byte channel chunk column (its 8 scale/shift pairs live in registers), rows
    // swept with a constant stride: no division in the streaming loop
    const int nthr = blockDim.x;
    for (int cc0 = 0; cc0 < ncc; cc0 += nthr) {
        const int span = min(ncc - cc0, nthr);
        const int R = max(1, nthr / span);
        const int cc = cc0 + tid % span, rl = tid / span;
        if (rl >= R) continue;
        const int ch0 = cc * EPC;
        float sc[EPC], sh[EPC];
#pragma unroll
        for (int i = 0; i < EPC; ++i) { sc[i] = s_scale[ch0 + i]; sh[i] = s_shift[ch0 + i]; }
        const bool first = ch0 < c0;
        const int cs = first ? c0 : c1;
        const T* src = (first ? x0 + ch0 : x1 + (ch0 - c0)) + row0 * cs;
        T* dst = yb + ch0;
        const int nrows = r_end - r_begin;
        for (int r = rl; r < nrows; r += R) {
            const Chunk<T> v = load_chunk<T>(src + (size_t)r * cs);
            Chunk<T> o;
#pragma unroll
            for (int i = 0; i < EPC; ++i) {
                float f = fmaf(v.get(i), sc[i], sh[i]);
                if (silu) f = f * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * f));
                o.set(i, f);
            }
            store_chunk<T>(dst + (size_t)r * c, o);
        }
    }
}

// Single-launch GroupNorm for images whose (image, channel-span) slab fits the registers of one workgroup:
// every thread keeps <= KT 16-byte chunks of ONE chunk column (its 8 channels belong to at most two groups), the
// statistics are reduced through LDS (fp64), and the normalised (+SiLU) values are written from the registers --
// the tensor is read once instead of twice (algorithmic traffic 1 read + 1 write) and one launch disappears.
// A span is a whole number of groups and of 16-byte chunks; the host picks the widest one with <= 16 chunks per
// thread.  Workgroups of one image are adjacent after the XCD remap (their 128-byte lines overlap).
template <typename T, int KT>
__global__ __launch_bounds__(1024) void gn_fused_kernel(const T* __restrict__ x0, const T* __restrict__ x1, int c0,
                                                        T* __restrict__ y, const float* __restrict__ gamma,
                                                        const float* __restrict__ beta, int hw, int c, int groups, int span,
                                                        float eps, int silu, float* __restrict__ stats_out) {
    constexpr int EPC = Elt<T>::EPC;
    __shared__ double s_sum[64], s_sq[64];
    __shared__ float s_mean[64], s_rstd[64];
    const int bid = xcd_remap(blockIdx.x, gridDim.x);
    const int nspan = c / span, img = bid / nspan, sp = bid - img * nspan;
    const int cps = span / EPC, cpg = c / groups;
    const int tid = threadIdx.x, nthr = blockDim.x;
    const int ch = tid % cps, row0 = tid / cps, rstep = nthr / cps;
    const int col = sp * span + ch * EPC;
    const bool first = col < c0;
    const int cs = first ? c0 : c - c0;
    const T* src = (first ? x0 + col : x1 + (col - c0)) + (size_t)img * hw * cs;
    const int gbase = (sp * span) / cpg, ng = span / cpg;
    const int g0 = col / cpg - gbase;
    const int split = min(EPC, (gbase + g0 + 1) * cpg - col);     // elements [0, split) belong to group g0, the rest to g0 + 1
    if (tid < 64) { s_sum[tid] = 0.0; s_sq[tid] = 0.0; }
    __syncthreads();
    Chunk<T> v[KT];
    float s[EPC], q[EPC];
#pragma unroll
    for (int i = 0; i < EPC; ++i) { s[i] = 0.f; q[i] = 0.f; }
#pragma unroll
    for (int k = 0; k < KT; ++k) {
        const int row = row0 + k * rstep;
        if (row < hw) {
            v[k] = load_chunk<T>(src + (size_t)row * cs);
#pragma unroll
            for (int i = 0; i < EPC; ++i) {
                const float f = v[k].get(i);
                s[i] += f;
                q[i] = fmaf(f, f, q[i]);
            }
        }
    }
    double sa = 0.0, qa = 0.0, sb = 0.0, qb = 0.0;
#pragma unroll
    for (int i = 0; i < EPC; ++i) {
        if (i < split) { sa += (double)s[i]; qa += (double)q[i]; }
        else { sb += (double)s[i]; qb += (double)q[i]; }
    }
    atomicAdd(&s_sum[g0], sa);
    atomicAdd(&s_sq[g0], qa);
    if (split < EPC) {
        atomicAdd(&s_sum[g0 + 1], sb);
        atomicAdd(&s_sq[g0 + 1], qb);
    }
    __syncthreads();
    if (tid < ng) {
        const double n = (double)hw * cpg;
        const double mean = s_sum[tid] / n;
        double var = s_sq[tid] / n - mean * mean;
        var = var > 0.0 ? var : 0.0;
        s_mean[tid] = (float)mean;
        s_rstd[tid] = (float)(1.0 / sqrt(var + (double)eps));
        if (stats_out) {
            stats_out[((size_t)img * groups + gbase + tid) * 2] = s_mean[tid];
            stats_out[((size_t)img * groups + gbase + tid) * 2 + 1] = s_rstd[tid];
        }
    }
    __syncthreads();
    float sc[EPC], sh[EPC];
#pragma unroll
    for (int i = 0; i < EPC; ++i) {
        const int g = i < split ? g0 : g0 + 1;
        sc[i] = s_rstd[g] * gamma[col + i];
        sh[i] = beta[col + i] - s_mean[g] * sc[i];
    }
    T* dst = y + (size_t)img * hw * c + col;
#pragma unroll
    for (int k = 0; k < KT; ++k) {
        const int row = row0 + k * rstep;
        if (row < hw) {
            Chunk<T> o;
#pragma unroll
            for (int i = 0; i < EPC; ++i) {
                float f = fmaf(v[k].get(i), sc[i], sh[i]);
                if (silu) f = f * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * f));
                o.set(i, f);
            }
            store_chunk<T>(dst + (size_t)row * c, o);
        }
    }
}

// channel span one workgroup normalises (<= 16 chunks per thread); 0: use the two-launch path.  The widest span that still gives the
// chip >= 2 workgroups per CU (large batches: fewer, fatter workgroups); small batches take the NARROWEST span instead -- at one
// scene (9 images) the widest one left 36 workgroups of 960 threads on a 256-CU chip, 11 serial chunks each (16.7 us per launch,
// 61 launches = 14 % of a DDIM step).  (MVLDM_GN_WIDE=1: always the widest, the round-2 rule -- A/B knob.)
static int gn_fused_plan(int hw, int c, int groups, int epc, int n_img, int& nthr, int& kt) {
    static const int force_wide = knob_int("MVLDM_GN_WIDE", 0);
    const int cpg = c / groups;
    if (!((cpg >= epc && true) || (epc % cpg == 0 && epc / cpg == 2))) return 0;
    auto gcd = [](int a, int b) { while (b) { int t = a % b; a = b; b = t; } return a; };
    const int base = cpg / gcd(cpg, epc) * epc;       // lcm: whole groups and whole chunks
    if (base > c || c % base) return 0;
    int best = 0;
    static const int x_span = knob_int("MVLDM_GN_SPAN", 0), x_nthr = knob_int("MVLDM_GN_NTHR", 1024);      // profiles/r05_gn_sweep.txt
    for (int m = 1; base * m <= c; ++m) {
        const int span = base * m;
        if (c % span || span / cpg > 64) continue;
        if (x_span && x_span <= c && span != x_span) continue;
        const int cps = span / epc;
        const int l = cps / gcd(cps, 64) * 64;
        if (l > 1024) continue;
        // threads: whole waves, a whole number of rows per sweep, and no more than the slab has chunks (small images)
        int n = std::max(l, std::min(1024, x_nthr) / l * l);
        const long long chunks = (long long)hw * cps;
        while (n > l && (long long)(n - l) >= chunks) n -= l;
        const int k = (int)((chunks + n - 1) / n);
        if (k > 16) continue;
        const bool enough = (long long)n_img * (c / span) >= 512;
        if (best == 0 || force_wide || enough) { best = span; nthr = n; kt = k; }
        if (!force_wide && !enough) break;      // narrower spans already cannot fill the chip: keep the narrowest valid one
    }
    return best;
}

// ---------------------------------------------------------------------------------------------- LN
template <typename T, int MAXCH>
__global__ __launch_bounds__(256) void layernorm_kernel(const T* __restrict__ x, T* __restrict__ y,
                                                        const float* __restrict__ gamma, const float* __restrict__ beta,
                                                        int rows, int c, float eps) {
    constexpr int EPC = Elt<T>::EPC;
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int ncc = c / EPC;
    const T* xr = x + (size_t)row * c;
    Chunk<T> v[MAXCH];
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < MAXCH; ++k) {
        const int cc = lane + k * 64;
        if (cc < ncc) {
            v[k] = load_chunk<T>(xr + cc * EPC);
#pragma unroll
            for (int i = 0; i < EPC; ++i) s += v[k].get(i);
        }
    }
    const float mean = wave_sum(s) / (float)c;
    float q = 0.f;
#pragma unroll
    for (int k = 0; k < MAXCH; ++k) {
        const int cc = lane + k * 64;
        if (cc < ncc) {
#pragma unroll
            for (int i = 0; i < EPC; ++i) {
                const float d = v[k].get(i) - mean;
                q += d * d;
            }
        }
    }
    const float rstd = rsqrtf(wave_sum(q) / (float)c + eps);
    T* yr = y + (size_t)row * c;
#pragma unroll
    for (int k = 0; k < MAXCH; ++k) {
        const int cc = lane + k * 64;
        if (cc < ncc) {
            Chunk<T> o;
#pragma unroll
            for (int i = 0; i < EPC; ++i) {
                const int ch = cc * EPC + i;
                o.set(i, (v[k].get(i) - mean) * rstd * gamma[ch] + beta[ch]);
            }
            store_chunk<T>(yr + cc * EPC, o);
        }
    }
}

// Narrow rows (the UNet's 320 / 640 / 1280 channels = 40 / 80 / 160 16-byte chunks): one wave per row leaves 24 of 64 lanes idle
// at c = 320 and keeps a single 16-byte load per lane in flight (3.7 TB/s).  Here a wave takes 64 / LPR rows at once, LPR lanes
// per row with CPL chunks each (chunk index = lane-in-row + k * LPR: every load instruction covers whole 128-byte lines), all
// CPL loads of a lane in flight together; gamma / beta stay in registers over the rows a wave walks.  Same two-pass statistics.
template <typename T, int LPR, int CPL>
__global__ __launch_bounds__(256) void layernorm_rows_kernel(const T* __restrict__ x, T* __restrict__ y, const float* __restrict__ gamma,
                                                             const float* __restrict__ beta, int rows, int c, float eps) {
    constexpr int EPC = Elt<T>::EPC, RPW = 64 / LPR;
    const int lane = threadIdx.x & 63, sub = lane & (LPR - 1), rsel = lane / LPR;
    float g[CPL][EPC], b[CPL][EPC];
#pragma unroll
    for (int k = 0; k < CPL; ++k)
#pragma unroll
        for (int i = 0; i < EPC; ++i) {
            g[k][i] = gamma[(sub + k * LPR) * EPC + i];
            b[k][i] = beta[(sub + k * LPR) * EPC + i];
        }
    const float inv_c = 1.0f / (float)c;
    const int stride = gridDim.x * 4 * RPW;
    for (int row0 = (blockIdx.x * 4 + (threadIdx.x >> 6)) * RPW; row0 < rows; row0 += stride) {
        const int row = row0 + rsel;
        const bool ok = row < rows;
        const T* xr = x + (size_t)(ok ? row : 0) * c;
        Chunk<T> v[CPL];
#pragma unroll
        for (int k = 0; k < CPL; ++k) v[k] = load_chunk<T>(xr + (sub + k * LPR) * EPC);
        float s = 0.f;
#pragma unroll
        for (int k = 0; k < CPL; ++k)
#pragma unroll
            for (int i = 0; i < EPC; ++i) s += v[k].get(i);
#pragma unroll
        for (int o = LPR / 2; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
        const float mean = s * inv_c;
        float q = 0.f;
#pragma unroll
        for (int k = 0; k < CPL; ++k)
#pragma unroll
            for (int i = 0; i < EPC; ++i) {
                const float d = v[k].get(i) - mean;
                q += d * d;
            }
#pragma unroll
        for (int o = LPR / 2; o > 0; o >>= 1) q += __shfl_xor(q, o, 64);
        const float rstd = rsqrtf(q * inv_c + eps);
        if (!ok) continue;
        T* yr = y + (size_t)row * c;
#pragma unroll
        for (int k = 0; k < CPL; ++k) {
            Chunk<T> o;
#pragma unroll
            for (int i = 0; i < EPC; ++i) o.set(i, (v[k].get(i) - mean) * rstd * g[k][i] + b[k][i]);
            store_chunk<T>(yr + (sub + k * LPR) * EPC, o);
        }
    }
}

template <typename T, int CPL>
static bool layernorm_rows_launch(const T* x, T* y, const float* gamma, const float* beta, int rows, int c, float eps, int lpr, hipStream_t s) {
    const int rpw = 64 / lpr;
    const int blocks = std::min((rows + 4 * rpw - 1) / (4 * rpw), 256 * 16);
    if (lpr == 8) hipLaunchKernelGGL((layernorm_rows_kernel<T, 8, CPL>), dim3(blocks), dim3(256), 0, s, x, y, gamma, beta, rows, c, eps);
    else if (lpr == 16) hipLaunchKernelGGL((layernorm_rows_kernel<T, 16, CPL>), dim3(blocks), dim3(256), 0, s, x, y, gamma, beta, rows, c, eps);
    else if (lpr == 32) hipLaunchKernelGGL((layernorm_rows_kernel<T, 32, CPL>), dim3(blocks), dim3(256), 0, s, x, y, gamma, beta, rows, c, eps);
    else return false;
    return true;
}

int groupnorm_run(const void* x, const void* x1, void* y, const float* gamma, const float* beta, int n_img, int hw, int c0,
                  int c1, int groups, float eps, int silu, int dtype, void* stats_ws, float* stats_out, hipStream_t s) {
    MVLDM_REQUIRE(groups > 0 && groups <= 64 && (c0 + c1) % groups == 0, "groupnorm: c=%d groups=%d", c0 + c1, groups);
    if (n_img == 0 || hw == 0) return MVLDM_OK;   // empty: buffers may be null
    MVLDM_REQUIRE(x && y && gamma && beta && stats_ws, "groupnorm: null pointer");
    MVLDM_REQUIRE((c1 == 0) == (x1 == nullptr), "groupnorm: x1/c1 mismatch");
    const int c = c0 + c1;
    MVLDM_REQUIRE(groups > 0 && groups <= 64 && c % groups == 0, "groupnorm: c=%d groups=%d", c, groups);
    const int epc = dtype == MVLDM_F32 ? 4 : 8;
    MVLDM_REQUIRE(c0 % epc == 0 && c1 % epc == 0, "groupnorm: channels (%d,%d) must be multiples of %d", c0, c1, epc);
    if (n_img == 0 || hw == 0) return MVLDM_OK;
    static const int no_fused = knob_int("MVLDM_GN_TWOPASS", 0);
    int f_nthr = 0, f_kt = 0;
    const int f_span = no_fused ? 0 : gn_fused_plan(hw, c, groups, epc, n_img, f_nthr, f_kt);
    if (f_span) {
        return dispatch_dtype(dtype, [&](auto t) {
            using T = decltype(t);
            const dim3 grid(n_img * (c / f_span)), block(f_nthr);
#define MVLDM_GN_FUSED(KT_)                                                                                             \
    hipLaunchKernelGGL((gn_fused_kernel<T, KT_>), grid, block, 0, s, reinterpret_cast<const T*>(x), reinterpret_cast<const T*>(x1), \
                       c0, reinterpret_cast<T*>(y), gamma, beta, hw, c, groups, f_span, eps, silu, stats_out)
            if (f_kt <= 2) MVLDM_GN_FUSED(2);
            else if (f_kt <= 4) MVLDM_GN_FUSED(4);
            else if (f_kt <= 8) MVLDM_GN_FUSED(8);
            else if (f_kt <= 12) MVLDM_GN_FUSED(12);
            else MVLDM_GN_FUSED(16);
#undef MVLDM_GN_FUSED
            return check_launch();
        });
    }
    // slabs: enough workgroups to cover the chip, at least 8 rows each
    int nchunk = std::min(MVLDM_GN_MAX_CHUNKS, std::max(1, std::min(hw / 8, (1024 + n_img - 1) / n_img)));
    const int rows_per_chunk = (hw + nchunk - 1) / nchunk;
    nchunk = (hw + rows_per_chunk - 1) / rows_per_chunk;
    const size_t smem = (size_t)(2 * c + 2 * groups) * sizeof(float);
    return dispatch_dtype(dtype, [&](auto t) {
        using T = decltype(t);
        hipLaunchKernelGGL(gn_stats_kernel<T>, dim3(n_img * nchunk), dim3(256), 0, s, reinterpret_cast<const T*>(x),
                           reinterpret_cast<const T*>(x1), c0, reinterpret_cast<double*>(stats_ws), hw, c, groups,
                           rows_per_chunk, nchunk);
        int rc = check_launch();
        if (rc) return rc;
        hipLaunchKernelGGL(gn_apply_kernel<T>, dim3(n_img * nchunk), dim3(256), smem, s, reinterpret_cast<const T*>(x),
                           reinterpret_cast<const T*>(x1), c0, reinterpret_cast<T*>(y), gamma, beta, reinterpret_cast<const double*>(stats_ws), hw, c, groups,
                           rows_per_chunk, nchunk, eps, silu, stats_out);
        return check_launch();
    });
}

int layernorm_run(const void* x, void* y, const float* gamma, const float* beta, int rows, int c, float eps, int dtype,
                  hipStream_t s) {
    if (rows == 0) return MVLDM_OK;   // empty: buffers may be null
    MVLDM_REQUIRE(x && y && gamma && beta, "layernorm: null pointer");
    const int epc = dtype == MVLDM_F32 ? 4 : 8;
    MVLDM_REQUIRE(c % epc == 0, "layernorm: c=%d must be a multiple of %d", c, epc);
    const int ncc = c / epc;
    MVLDM_REQUIRE(ncc <= 64 * 16, "layernorm: c=%d too wide", c);
    if (rows == 0) return MVLDM_OK;
    const int blocks = (rows + 3) / 4;
    return dispatch_dtype(dtype, [&](auto t) {
        using T = decltype(t);
        const T* xp = reinterpret_cast<const T*>(x);
        T* yp = reinterpret_cast<T*>(y);
        if constexpr (sizeof(T) == 2) {
            // chunks = LPR * CPL with LPR in {8, 16, 32} lanes per row and CPL in {3, 4, 5} chunks per lane, many rows
            if (rows >= 1024) {
                for (int cpl = 5; cpl >= 3; --cpl) {
                    if (ncc % cpl) continue;
                    const int lpr = ncc / cpl;
                    if (lpr != 8 && lpr != 16 && lpr != 32) continue;
                    const bool ok = cpl == 5 ? layernorm_rows_launch<T, 5>(xp, yp, gamma, beta, rows, c, eps, lpr, s)
                                  : cpl == 4 ? layernorm_rows_launch<T, 4>(xp, yp, gamma, beta, rows, c, eps, lpr, s)
                                             : layernorm_rows_launch<T, 3>(xp, yp, gamma, beta, rows, c, eps, lpr, s);
                    if (ok) return check_launch();
                }
            }
        }
        if (ncc <= 64) hipLaunchKernelGGL((layernorm_kernel<T, 1>), dim3(blocks), dim3(256), 0, s, xp, yp, gamma, beta, rows, c, eps);
        else if (ncc <= 128) hipLaunchKernelGGL((layernorm_kernel<T, 2>), dim3(blocks), dim3(256), 0, s, xp, yp, gamma, beta, rows, c, eps);
        else if (ncc <= 256) hipLaunchKernelGGL((layernorm_kernel<T, 4>), dim3(blocks), dim3(256), 0, s, xp, yp, gamma, beta, rows, c, eps);
        else if (ncc <= 512) hipLaunchKernelGGL((layernorm_kernel<T, 8>), dim3(blocks), dim3(256), 0, s, xp, yp, gamma, beta, rows, c, eps);
        else hipLaunchKernelGGL((layernorm_kernel<T, 16>), dim3(blocks), dim3(256), 0, s, xp, yp, gamma, beta, rows, c, eps);
        return check_launch();
    });
}

}  // namespace mvldm

extern "C" int mvldm_groupnorm_passes(int n_img, int hw, int c, int groups, int dtype) {
    if (n_img <= 0 || hw <= 0 || groups <= 0 || c <= 0 || c % groups) return 3;
    static const int no_fused = mvldm::knob_int("MVLDM_GN_TWOPASS", 0);
    int nthr = 0, kt = 0;
    return (!no_fused && mvldm::gn_fused_plan(hw, c, groups, dtype == MVLDM_F32 ? 4 : 8, n_img, nthr, kt)) ? 2 : 3;
}

extern "C" int mvldm_groupnorm_fwd(const void* x0, const void* x1, void* y, const float* gamma, const float* beta,
                                   int n_img, int hw, int c0, int c1, int groups, float eps, int silu, int dtype,
                                   void* stats_ws, float* stats_out, mvldm_stream_t stream) {
    return mvldm::groupnorm_run(x0, x1, y, gamma, beta, n_img, hw, c0, c1, groups, eps, silu, dtype, stats_ws, stats_out,
                                (hipStream_t)stream);
}

extern "C" int mvldm_layernorm_fwd(const void* x, void* y, const float* gamma, const float* beta, int rows, int c,
                                   float eps, int dtype, mvldm_stream_t stream) {
    return mvldm::layernorm_run(x, y, gamma, beta, rows, c, eps, dtype, (hipStream_t)stream);
}
