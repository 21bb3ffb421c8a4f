// Flash-attention backward for CDNA4 (include/mvldm.h: mvldm_attention_bwd): gradients of
//     out = softmax(q k^T * scale) v        per (segment, head)
// with the probabilities recomputed from the forward pass's log-sum-exp (never materialised in HBM; the reference's
// autograd keeps the [heads, L, L] fp32 `sim` and its softmax alive, mvdream/attention.py:188-199).
//
//   P = exp2(s*c - lse),  dP = dO V^T,  dS = P o (dP - delta),  delta[q] = sum_d dO[q,d] O[q,d]
//   dV = P^T dO,   dK = scale * dS^T Q,   dQ = scale * dS K
//
// Two launches of ONE kernel template, each a mirror image of the forward kernel (attention.hip) with the roles of the
// "resident" operand (fragments held in registers, one row per lane) and the "streamed" operand (64-row tiles through
// LDS) swapped, so that no gradient needs atomics and every per-row quantity is lane-local or a broadcast LDS read:
//   MODE 0 (dQ):    resident = 128 queries (Q, dO fragments); streamed = K, V tiles.
//                   S^T[key][q] = K Q^T,  dP^T = V dO^T  (A from LDS rows, B from registers);  dQ^T[d][q] += K^T dS^T
//   MODE 1 (dK,dV): resident = 128 keys (K, V fragments);     streamed = Q, dO tiles (+ their lse / delta).
//                   S[q][key] = Q K^T,    dP = dO V^T;   dV^T[d][key] += dO^T P,   dK^T[d][key] += Q^T dS
// In both, the third product takes its A operand with the LDS transpose read (ds_read_b64_tr_b16) from the row-major
// tile and its B operand straight from the score accumulators, exactly like the forward kernel's P V product.
// 16-bit activations only; fp32 (the parity mode) and heads wider than 160 take the VALU reference kernels below.
#include <algorithm>

#include "common.h"

namespace mvldm {

struct AttnBwdParams {
    const void* q; const void* k; const void* v; const void* dout;
    void* dq; void* dk; void* dv;
    const float* lse; const float* delta;
    const int32_t* seg;
    int ld_q, ld_k, ld_v, ld_do, ld_dq, ld_dk, ld_dv, heads, d, stat_ld;
    float scale, scale_log2e;
    int ntile;        // resident tiles per (head, segment)
};

template <typename T> struct BwdMma;
template <> struct BwdMma<bf16_t> {
    typedef __attribute__((ext_vector_type(4))) __bf16 Half;
    using Frag = bf16x8;
    static __device__ __forceinline__ Half tr(const bf16_t* p) { return __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) Half*)(p)); }
    static __device__ __forceinline__ f32x16 mma(Frag a, Frag b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0); }
};
template <> struct BwdMma<f16_t> {
    typedef __attribute__((ext_vector_type(4))) _Float16 Half;
    using Frag = f16x8;
    static __device__ __forceinline__ Half tr(const f16_t* p) {
        typedef __attribute__((ext_vector_type(4))) __fp16 fp16x4_b;
        const fp16x4_b v = __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) fp16x4_b*)(p));
        return __builtin_bit_cast(Half, v);
    }
    static __device__ __forceinline__ f32x16 mma(Frag a, Frag b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0); }
};

constexpr int BR = 128, BS = 64;     // resident rows per workgroup, streamed rows per tile
// transpose-read tile pitch: bytes = 64 or 192 (mod 256) keeps the 8 row pieces of a half-wave on distinct bank octets (attention.hip)
constexpr int bwd_v_pitch(int dv) {
    int bytes = 2 * dv;
    while (bytes % 256 != 64 && bytes % 256 != 192) bytes += 32;
    return bytes / 2;
}

// register budget: the d = 40 heads (DP = 48: 75 % of the attention time of the multi-view blocks) in MODE 1 came out at 198 VGPRs + 64
// AGPRs = 6 over the 256 that let a second wave share the SIMD -- and a lone wave issues its VALU at half rate (attention.hip); asking for
// two waves makes hipcc fit them (kernel_resources.json: no scratch)
template <typename T, int DP, int MODE>
__global__ __launch_bounds__(256, ((DP <= 64 || (MODE == 0 && DP <= 96)) ? 2 : 1)) void attention_bwd_kernel(const AttnBwdParams p) {
    constexpr int EPC = 8;
    constexpr int DV = (DP + 31) / 32 * 32, NDB = DV / 32;
    constexpr int KP = DP + 8, VP = bwd_v_pitch(DV);     // row pitches (elements): row-read layout / transpose-read layout
    constexpr int NCH = DP / EPC, NQ = DP / 16;
    constexpr int NST = (BS * NCH + 255) / 256;
    using Frag = typename BwdMma<T>::Frag;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    T* s1r = reinterpret_cast<T*>(smem);      // streamed operand 1 (K | Q), row-read layout
    T* s2r = s1r + BS * KP;                    // streamed operand 2 (V | dO), row-read layout
    T* s1t = s2r + BS * KP;                    // operand 1, transpose-read layout (zero padded to DV columns)
    T* s2t = s1t + BS * VP;                    // operand 2, transpose-read layout (MODE 1 only)
    float* s_lse = reinterpret_cast<float*>(s2t + BS * VP);   // MODE 1: lse / delta of the streamed queries
    float* s_del = s_lse + BS;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int hi = lane >> 5, l31 = lane & 31;
    const int rt = blockIdx.x % p.ntile, hs_ = blockIdx.x / p.ntile;
    const int head = hs_ % p.heads;
    const int4 sg = reinterpret_cast<const int4*>(p.seg)[hs_ / p.heads];
    const int q_row0 = sg.x, q_len = sg.y, kv_row0 = sg.z, kv_len = sg.w;
    const int res_len = MODE == 0 ? q_len : kv_len, str_len = MODE == 0 ? kv_len : q_len;
    const int res_row0 = MODE == 0 ? q_row0 : kv_row0, str_row0 = MODE == 0 ? kv_row0 : q_row0;
    if (rt * BR >= res_len) return;
    const int d = p.d;

    // ---- resident fragments (one row per lane) ----
    const int r_local = rt * BR + wave * 32 + l31;
    const bool r_ok = r_local < res_len;
    const T* r1p = reinterpret_cast<const T*>(MODE == 0 ? p.q : p.k) + (size_t)(res_row0 + r_local) * (MODE == 0 ? p.ld_q : p.ld_k) + head * d;
    const T* r2p = reinterpret_cast<const T*>(MODE == 0 ? p.dout : p.v) + (size_t)(res_row0 + r_local) * (MODE == 0 ? p.ld_do : p.ld_v) + head * d;
    Frag r1f[NQ], r2f[NQ];
#pragma unroll
    for (int kk = 0; kk < NQ; ++kk) {
        const int dk = kk * 16 + hi * 8;
        u32x4 a = u32x4{0u, 0u, 0u, 0u}, b = a;
        if (r_ok && dk < d) { a = *reinterpret_cast<const u32x4*>(r1p + dk); b = *reinterpret_cast<const u32x4*>(r2p + dk); }
        r1f[kk] = __builtin_bit_cast(Frag, a);
        r2f[kk] = __builtin_bit_cast(Frag, b);
    }
    float r_lse = 0.f, r_del = 0.f;       // MODE 0: per-lane query statistics
    if (MODE == 0 && r_ok) {
        r_lse = p.lse[(size_t)head * p.stat_ld + q_row0 + r_local];
        r_del = p.delta[(size_t)head * p.stat_ld + q_row0 + r_local];
    }

    f32x16 acc1[NDB], acc2[MODE == 1 ? NDB : 1];      // MODE 0: dQ^T;  MODE 1: dK^T, dV^T
#pragma unroll
    for (int db = 0; db < NDB; ++db)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            acc1[db][r] = 0.f;
            if constexpr (MODE == 1) acc2[db][r] = 0.f;
        }

    const T* s1base = reinterpret_cast<const T*>(MODE == 0 ? p.k : p.q) + (size_t)str_row0 * (MODE == 0 ? p.ld_k : p.ld_q) + head * d;
    const T* s2base = reinterpret_cast<const T*>(MODE == 0 ? p.v : p.dout) + (size_t)str_row0 * (MODE == 0 ? p.ld_v : p.ld_do) + head * d;
    const int ld1 = MODE == 0 ? p.ld_k : p.ld_q, ld2 = MODE == 0 ? p.ld_v : p.ld_do;
    const int ntile = (str_len + BS - 1) / BS;

    // zero the padding columns of the transpose-read tiles once (columns [DP, DV) are never written by the staging below)
    if constexpr (DV > DP) {
        for (int i = tid; i < BS * (DV - DP); i += 256) {
            const int row = i / (DV - DP), col = DP + i % (DV - DP);
            s1t[row * VP + col] = from_f32<T>(0.f);
            s2t[row * VP + col] = from_f32<T>(0.f);
        }
    }

    Chunk<T> g1[NST], g2[NST];
    auto load_tile = [&](int t) {
#pragma unroll
        for (int j = 0; j < NST; ++j) {
            const int idx = tid + 256 * j;
            const int row = idx / NCH, ch = idx - row * NCH;
            const bool ok = idx < BS * NCH && ch * EPC < d && t * BS + row < str_len;
            if (ok) {
                g1[j] = load_chunk<T>(s1base + (size_t)(t * BS + row) * ld1 + ch * EPC);
                g2[j] = load_chunk<T>(s2base + (size_t)(t * BS + row) * ld2 + ch * EPC);
            } else {
                g1[j].zero();
                g2[j].zero();
            }
        }
    };
    auto store_tile = [&](int t) {
#pragma unroll
        for (int j = 0; j < NST; ++j) {
            const int idx = tid + 256 * j;
            if (idx < BS * NCH) {
                const int row = idx / NCH, ch = idx - row * NCH;
                *reinterpret_cast<u32x4*>(s1r + row * KP + ch * EPC) = g1[j].raw;
                *reinterpret_cast<u32x4*>(s2r + row * KP + ch * EPC) = g2[j].raw;
                *reinterpret_cast<u32x4*>(s1t + row * VP + ch * EPC) = g1[j].raw;
                if (MODE == 1) *reinterpret_cast<u32x4*>(s2t + row * VP + ch * EPC) = g2[j].raw;
            }
        }
        if (MODE == 1 && tid < BS) {
            const int qr = t * BS + tid;
            const bool ok = qr < str_len;
            s_lse[tid] = ok ? p.lse[(size_t)head * p.stat_ld + q_row0 + qr] : INFINITY;     // exp2(x - inf) = 0: rows past the end vanish
            s_del[tid] = ok ? p.delta[(size_t)head * p.stat_ld + q_row0 + qr] : 0.f;
        }
    };
    if (ntile > 0) load_tile(0);

    const float c = p.scale_log2e;
    for (int t = 0; t < ntile; ++t) {
        store_tile(t);
        __syncthreads();
        if (t + 1 < ntile) load_tile(t + 1);

        // ---- X = S1 R1^T (scores),  Y = S2 R2^T (dP), both [streamed row][resident lane] ----
        f32x16 x[2], y[2];
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
            x[kb] = f32x16{0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            y[kb] = x[kb];
            const T* a1 = s1r + (kb * 32 + l31) * KP;
            const T* a2 = s2r + (kb * 32 + l31) * KP;
#pragma unroll
            for (int kk = 0; kk < NQ; ++kk) {
                x[kb] = BwdMma<T>::mma(*reinterpret_cast<const Frag*>(a1 + kk * 16 + hi * 8), r1f[kk], x[kb]);
                y[kb] = BwdMma<T>::mma(*reinterpret_cast<const Frag*>(a2 + kk * 16 + hi * 8), r2f[kk], y[kb]);
            }
        }
        // ---- P = exp2(X c - lse),  dS = P (Y - delta) ----
        const bool ragged = t == ntile - 1 && (str_len & (BS - 1)) != 0;
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int srow = kb * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
                float lse, del;
                if (MODE == 0) { lse = r_lse; del = r_del; }
                else { lse = s_lse[srow]; del = s_del[srow]; }
                float pv = __builtin_amdgcn_exp2f(fmaf(x[kb][r], c, -lse));
                if (MODE == 0 && ragged && t * BS + srow >= str_len) pv = 0.f;      // keys past the end (MODE 1: lse = inf does it)
                x[kb][r] = pv;
                y[kb][r] = pv * (y[kb][r] - del);
            }
        // ---- third products: A by transpose read of the row-major tiles, B = P / dS from the accumulators ----
        Frag pf[4], sf[4];
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                pf[ks][j] = from_f32<T>(x[ks >> 1][(ks & 1) * 8 + j]);
                sf[ks][j] = from_f32<T>(y[ks >> 1][(ks & 1) * 8 + j]);
            }
        const int gi = lane >> 4, sl = lane & 15;
        const int toff = (4 * (gi >> 1) + (sl >> 2)) * VP + (gi & 1) * 16 + 4 * (sl & 3);
#pragma unroll
        for (int db = 0; db < NDB; ++db) {
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const T* pa = s1t + toff + (ks * 16) * VP + db * 32;
                const Frag a = __builtin_shufflevector(BwdMma<T>::tr(pa), BwdMma<T>::tr(pa + 8 * VP), 0, 1, 2, 3, 4, 5, 6, 7);
                acc1[db] = BwdMma<T>::mma(a, sf[ks], acc1[db]);          // dQ^T += K^T dS^T   |   dK^T += Q^T dS
                if constexpr (MODE == 1) {
                    const T* pb = s2t + toff + (ks * 16) * VP + db * 32;
                    const Frag b = __builtin_shufflevector(BwdMma<T>::tr(pb), BwdMma<T>::tr(pb + 8 * VP), 0, 1, 2, 3, 4, 5, 6, 7);
                    acc2[db] = BwdMma<T>::mma(b, pf[ks], acc2[db]);      // dV^T += dO^T P
                }
            }
        }
        __syncthreads();
    }

    if (!r_ok) return;
    auto write = [&](void* base, int ld, const f32x16 (&acc)[NDB], float mul) {
        T* op = reinterpret_cast<T*>(base) + (size_t)(res_row0 + r_local) * ld + head * d;
#pragma unroll
        for (int db = 0; db < NDB; ++db)
#pragma unroll
            for (int r4 = 0; r4 < 4; ++r4) {
                const int dd = db * 32 + 8 * r4 + 4 * hi;
                if (dd < d) {
                    union { T e[4]; u32x2 raw; } w;
#pragma unroll
                    for (int e = 0; e < 4; ++e) w.e[e] = from_f32<T>(acc[db][r4 * 4 + e] * mul);
                    *reinterpret_cast<u32x2*>(op + dd) = w.raw;
                }
            }
    };
    if constexpr (MODE == 0) {
        write(p.dq, p.ld_dq, acc1, p.scale);
    } else {
        write(p.dk, p.ld_dk, acc1, p.scale);
        write(p.dv, p.ld_dv, reinterpret_cast<const f32x16(&)[NDB]>(acc2), 1.0f);
    }
}

// ---- delta[head][q] = sum_d dO[q, head, d] * O[q, head, d] ------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void attn_delta_kernel(const T* __restrict__ o, const T* __restrict__ dout, float* __restrict__ delta, int rows,
                                                         int heads, int d, int ld_o, int ld_do, int stat_ld) {
    constexpr int EPC = Elt<T>::EPC;
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= rows * heads) return;
    const int row = idx / heads, head = idx - row * heads;
    const T* a = o + (size_t)row * ld_o + head * d;
    const T* b = dout + (size_t)row * ld_do + head * d;
    float s = 0.f;
    for (int i = 0; i < d; i += EPC) {
        const Chunk<T> u = load_chunk<T>(a + i), v = load_chunk<T>(b + i);
#pragma unroll
        for (int e = 0; e < EPC; ++e) s += u.get(e) * v.get(e);
    }
    delta[(size_t)head * stat_ld + row] = s;
}

// ---- VALU reference kernels: fp32 (the parity mode) and head dims the MFMA kernels do not cover --------------------
// One wave per resident row, the head dimension spread over the lanes (16-byte chunks), streamed rows one at a time.
template <typename T, int NCHK, int MODE>
__global__ __launch_bounds__(256) void attention_bwd_ref_kernel(const AttnBwdParams p) {
    constexpr int EPC = Elt<T>::EPC;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int head = blockIdx.y;
    const int4 sg = reinterpret_cast<const int4*>(p.seg)[blockIdx.z];
    const int q_row0 = sg.x, q_len = sg.y, kv_row0 = sg.z, kv_len = sg.w;
    const int res_len = MODE == 0 ? q_len : kv_len, str_len = MODE == 0 ? kv_len : q_len;
    const int r_local = blockIdx.x * 4 + wave;
    if (r_local >= res_len) return;
    const int d = p.d, nch = d / EPC;
    const int res_row = (MODE == 0 ? q_row0 : kv_row0) + r_local;
    const T* r1p = reinterpret_cast<const T*>(MODE == 0 ? p.q : p.k) + (size_t)res_row * (MODE == 0 ? p.ld_q : p.ld_k) + head * d;
    const T* r2p = reinterpret_cast<const T*>(MODE == 0 ? p.dout : p.v) + (size_t)res_row * (MODE == 0 ? p.ld_do : p.ld_v) + head * d;
    float r1[NCHK][EPC], r2[NCHK][EPC], a1[NCHK][EPC], a2[NCHK][EPC];
#pragma unroll
    for (int j = 0; j < NCHK; ++j) {
        const int ch = lane + 64 * j;
        Chunk<T> u, v;
        if (ch < nch) { u = load_chunk<T>(r1p + ch * EPC); v = load_chunk<T>(r2p + ch * EPC); } else { u.zero(); v.zero(); }
#pragma unroll
        for (int i = 0; i < EPC; ++i) { r1[j][i] = u.get(i); r2[j][i] = v.get(i); a1[j][i] = 0.f; a2[j][i] = 0.f; }
    }
    const T* s1b = reinterpret_cast<const T*>(MODE == 0 ? p.k : p.q) + (size_t)(MODE == 0 ? kv_row0 : q_row0) * (MODE == 0 ? p.ld_k : p.ld_q) + head * d;
    const T* s2b = reinterpret_cast<const T*>(MODE == 0 ? p.v : p.dout) + (size_t)(MODE == 0 ? kv_row0 : q_row0) * (MODE == 0 ? p.ld_v : p.ld_do) + head * d;
    const int ld1 = MODE == 0 ? p.ld_k : p.ld_q, ld2 = MODE == 0 ? p.ld_v : p.ld_do;
    const float* lse = p.lse + (size_t)head * p.stat_ld + q_row0;
    const float* del = p.delta + (size_t)head * p.stat_ld + q_row0;
    for (int srow = 0; srow < str_len; ++srow) {
        Chunk<T> u[NCHK], v[NCHK];
        float px = 0.f, py = 0.f;
#pragma unroll
        for (int j = 0; j < NCHK; ++j) {
            const int ch = lane + 64 * j;
            if (ch < nch) { u[j] = load_chunk<T>(s1b + (size_t)srow * ld1 + ch * EPC); v[j] = load_chunk<T>(s2b + (size_t)srow * ld2 + ch * EPC); }
            else { u[j].zero(); v[j].zero(); }
#pragma unroll
            for (int i = 0; i < EPC; ++i) { px += u[j].get(i) * r1[j][i]; py += v[j].get(i) * r2[j][i]; }
        }
        const float x = wave_sum(px), y = wave_sum(py);
        const int qi = MODE == 0 ? r_local : srow;
        const float pv = exp2f(x * p.scale_log2e - lse[qi]);
        const float ds = pv * (y - del[qi]);
#pragma unroll
        for (int j = 0; j < NCHK; ++j)
#pragma unroll
            for (int i = 0; i < EPC; ++i) {
                a1[j][i] += ds * u[j].get(i);                       // dQ += dS K   |   dK += dS Q
                if (MODE == 1) a2[j][i] += pv * v[j].get(i);        // dV += P dO
            }
    }
    T* o1 = reinterpret_cast<T*>(MODE == 0 ? p.dq : p.dk) + (size_t)res_row * (MODE == 0 ? p.ld_dq : p.ld_dk) + head * d;
    T* o2 = reinterpret_cast<T*>(p.dv) + (size_t)res_row * p.ld_dv + head * d;
#pragma unroll
    for (int j = 0; j < NCHK; ++j) {
        const int ch = lane + 64 * j;
        if (ch < nch) {
            Chunk<T> w;
#pragma unroll
            for (int i = 0; i < EPC; ++i) w.set(i, a1[j][i] * p.scale);
            store_chunk<T>(o1 + ch * EPC, w);
            if (MODE == 1) {
#pragma unroll
                for (int i = 0; i < EPC; ++i) w.set(i, a2[j][i]);
                store_chunk<T>(o2 + ch * EPC, w);
            }
        }
    }
}

template <typename T, int DP> static int launch_bwd_mfma(AttnBwdParams p, int n_seg, int max_q_len, int max_kv_len, hipStream_t s) {
    constexpr int DV = (DP + 31) / 32 * 32;
    constexpr int smem = (2 * BS * (DP + 8) + 2 * BS * bwd_v_pitch(DV)) * 2 + 2 * BS * 4;
    static std::atomic<uint64_t> done0{0}, done1{0};
    if (int rc0 = ensure_dyn_smem(reinterpret_cast<const void*>(attention_bwd_kernel<T, DP, 0>), smem, done0)) return rc0;
    if (int rc1 = ensure_dyn_smem(reinterpret_cast<const void*>(attention_bwd_kernel<T, DP, 1>), smem, done1)) return rc1;
    p.ntile = (max_q_len + BR - 1) / BR;
    hipLaunchKernelGGL((attention_bwd_kernel<T, DP, 0>), dim3(p.ntile * p.heads * n_seg), dim3(256), smem, s, p);
    int rc = check_launch();
    if (rc) return rc;
    p.ntile = (max_kv_len + BR - 1) / BR;
    hipLaunchKernelGGL((attention_bwd_kernel<T, DP, 1>), dim3(p.ntile * p.heads * n_seg), dim3(256), smem, s, p);
    return check_launch();
}

template <typename T, int NCHK> static int launch_bwd_ref(const AttnBwdParams& p, int n_seg, int max_q_len, int max_kv_len, hipStream_t s) {
    hipLaunchKernelGGL((attention_bwd_ref_kernel<T, NCHK, 0>), dim3((max_q_len + 3) / 4, p.heads, n_seg), dim3(256), 0, s, p);
    int rc = check_launch();
    if (rc) return rc;
    hipLaunchKernelGGL((attention_bwd_ref_kernel<T, NCHK, 1>), dim3((max_kv_len + 3) / 4, p.heads, n_seg), dim3(256), 0, s, p);
    return check_launch();
}

int attention_bwd_run(const mvldm_attn_bwd_desc& a, hipStream_t s) {
    const int epc = a.dtype == MVLDM_F32 ? 4 : 8;
    MVLDM_REQUIRE(a.head_dim > 0 && a.head_dim % epc == 0, "attention_bwd: head_dim %d must be a multiple of %d", a.head_dim, epc);
    if (a.n_seg == 0 || a.max_q_len == 0 || a.max_kv_len == 0) return MVLDM_OK;
    MVLDM_REQUIRE(a.q && a.k && a.v && a.out && a.dout && a.dq && a.dk && a.dv && a.lse && a.delta && a.seg, "attention_bwd: null pointer");
    MVLDM_REQUIRE(a.ld_q % epc == 0 && a.ld_k % epc == 0 && a.ld_v % epc == 0 && a.ld_o % epc == 0 && a.ld_do % epc == 0 && a.ld_dq % 4 == 0 &&
                  a.ld_dk % 4 == 0 && a.ld_dv % 4 == 0, "attention_bwd: row strides must keep 16-byte alignment");
    AttnBwdParams p{a.q, a.k, a.v, a.dout, a.dq, a.dk, a.dv, a.lse, a.delta, a.seg, a.ld_q, a.ld_k, a.ld_v, a.ld_do, a.ld_dq, a.ld_dk, a.ld_dv,
                    a.heads, a.head_dim, a.stat_ld, a.scale, a.scale * 1.4426950408889634f, 0};
    const int dp = (a.head_dim + 15) / 16 * 16;
    return dispatch_dtype(a.dtype, [&](auto t) -> int {
        using T = decltype(t);
        const int rows = a.total_q_rows;
        hipLaunchKernelGGL(attn_delta_kernel<T>, dim3((rows * a.heads + 255) / 256), dim3(256), 0, s, reinterpret_cast<const T*>(a.out),
                           reinterpret_cast<const T*>(a.dout), a.delta, rows, a.heads, a.head_dim, a.ld_o, a.ld_do, a.stat_ld);
        int rc = check_launch();
        if (rc) return rc;
        if constexpr (sizeof(T) == 2) {
            switch (a.head_dim <= 160 ? dp : 0) {
                case 16: return launch_bwd_mfma<T, 16>(p, a.n_seg, a.max_q_len, a.max_kv_len, s);
                case 32: return launch_bwd_mfma<T, 32>(p, a.n_seg, a.max_q_len, a.max_kv_len, s);
                case 48: return launch_bwd_mfma<T, 48>(p, a.n_seg, a.max_q_len, a.max_kv_len, s);
                case 64: return launch_bwd_mfma<T, 64>(p, a.n_seg, a.max_q_len, a.max_kv_len, s);
                case 80: return launch_bwd_mfma<T, 80>(p, a.n_seg, a.max_q_len, a.max_kv_len, s);
                case 96: return launch_bwd_mfma<T, 96>(p, a.n_seg, a.max_q_len, a.max_kv_len, s);
                case 112: return launch_bwd_mfma<T, 112>(p, a.n_seg, a.max_q_len, a.max_kv_len, s);
                case 128: return launch_bwd_mfma<T, 128>(p, a.n_seg, a.max_q_len, a.max_kv_len, s);
                case 144: return launch_bwd_mfma<T, 144>(p, a.n_seg, a.max_q_len, a.max_kv_len, s);
                case 160: return launch_bwd_mfma<T, 160>(p, a.n_seg, a.max_q_len, a.max_kv_len, s);
                default: break;
            }
        }
        const int nch = a.head_dim / epc;
        if (nch <= 64) return launch_bwd_ref<T, 1>(p, a.n_seg, a.max_q_len, a.max_kv_len, s);
        if (nch <= 128) return launch_bwd_ref<T, 2>(p, a.n_seg, a.max_q_len, a.max_kv_len, s);
        return set_error(MVLDM_ERR_UNSUPPORTED, "attention_bwd: head_dim %d too wide", a.head_dim);
    });
}

}  // namespace mvldm

extern "C" int mvldm_attention_bwd(const mvldm_attn_bwd_desc* d, mvldm_stream_t stream) {
    MVLDM_REQUIRE(d != nullptr, "attention_bwd: null desc");
    return mvldm::attention_bwd_run(*d, (hipStream_t)stream);
}
