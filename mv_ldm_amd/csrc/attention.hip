// Flash-style attention forward for CDNA4 (see include/mvldm.h: mvldm_attention_fwd).
//
// One workgroup = 4 waves = 128 queries of one (segment, head); K/V are streamed in 64-key tiles
// through LDS and the score matrix never leaves the chip (the reference materialises a
// [heads, L, L] fp32 `sim`, 839 MB at L = 5120: mvdream/attention.py:188-199).
//
// Both products run "transposed" so that every per-query quantity is lane-local:
//     S^T[key][q] = K Q^T        (A = K tile from LDS, B = Q fragments held in registers)
//     O^T[d][q]   = V^T P^T      (A = V^T tile from LDS, B = P straight from the S^T accumulators)
// With the 32x32 MFMA accumulator layout (col = lane&31, row = (r&3)+8*(r>>2)+4*(lane>>5)) a lane
// owns ONE query column: the online-softmax max / sum / rescale need no cross-lane traffic except
// one lane^32 exchange, and P feeds the second MFMA without leaving registers -- the key order seen
// by that MFMA is the accumulator's row order, and the V^T tile is written to LDS in the same
// permuted key order, so no shuffle is needed.  Softmax statistics and both accumulations are fp32;
// P is rounded to the activation dtype for the PV product (fp32 path: everything fp32 on
// v_mfma_f32_32x32x2_f32).
//
// LDS: K tile [64][DP+8] (pitch = odd multiple of 16 B -> conflict-free ds_read_b128 fragment
// reads); V tile row-major [64][DV+16] read back transposed by ds_read_b64_tr_b16 (f32: V^T [DV][65]).  Head dims are zero-padded to DP = roundup(d,16) / DV = roundup(d,32)
// inside LDS only -- never in HBM.
#include <algorithm>
#include <cstdlib>

#include "common.h"

namespace mvldm {

struct AttnParams {
    const void* q; const void* k; const void* v; void* out;
    const int32_t* seg;
    int ld_q, ld_k, ld_v, ld_o, heads, d;
    float scale_log2e;
    int nqt;   // query tiles per (head, segment) in the 1-D grid of attention_kernel (set at launch)
    int remap; // 1: XCD-contiguous workgroup order (set at launch)
    float* lse; // optional [heads][total q rows]: log2-domain log-sum-exp of the scaled scores (what the backward pass needs)
    int lse_ld;
};

template <typename T> struct AttnMma;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4_t;
typedef __attribute__((ext_vector_type(4))) _Float16 f16x4_t;
template <> struct AttnMma<bf16_t> {
    using Frag = bf16x8;
    using Half = bf16x4_t;
    static __device__ __forceinline__ Half tr_read(const bf16_t* p) {
        return __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) Half*)(p));
    }
    static __device__ __forceinline__ f32x16 mma(Frag a, Frag b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0); }
};
template <> struct AttnMma<f16_t> {
    using Frag = f16x8;
    using Half = f16x4_t;
    static __device__ __forceinline__ Half tr_read(const f16_t* p) {
        typedef __attribute__((ext_vector_type(4))) __fp16 fp16x4_b;
        const fp16x4_b v = __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) fp16x4_b*)(p));
        return __builtin_bit_cast(Half, v);
    }
    static __device__ __forceinline__ f32x16 mma(Frag a, Frag b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0); }
};

constexpr int BQ = 128, BKV = 64;

// Row pitch (elements) of the row-major 16-bit V tile read with ds_read_b64_tr_b16.  One half-wave of that read touches
// 4 key rows x two 32-byte column pieces (16-lane groups gi = 0, 1): conflict-free needs the 8 pieces on 8 different bank
// octets of the 64-bank array, i.e. the pitch in bytes = 64 or 192 (mod 256) -- rows at banks {0,16,32,48} (some order), the
// second group 8 banks further.  (The first layout, 2*DV + 32 bytes, put row 3 of group 1 on the banks of row 0 of group 0:
// SQ_LDS_BANK_CONFLICT = 40 % of the LDS-active cycles of the d = 40 kernel, profiles/r02_attention_pmc.json.)
constexpr int v_pitch(int dv) {
    int bytes = 2 * dv;
    while (bytes % 256 != 64 && bytes % 256 != 192) bytes += 32;
    return bytes / 2;
}

// position of key `t` (0..63) inside the V^T tile for 16-bit types: the 8 consecutive slots
// [ks*16 + hi*8, +8) must hold the keys that S^T registers (ks&1)*8 .. +8 of block ks>>1 belong to.
__device__ __forceinline__ int vt_pos16(int t) {
    const int kb = t >> 5, u32_ = t & 31;
    const int g16 = u32_ >> 4, u = u32_ & 15;
    const int hi = (u >> 2) & 1, j = (u & 3) + 4 * (u >> 3);
    return kb * 32 + g16 * 16 + hi * 8 + j;
}

// 16-byte chunk through a buffer descriptor: a lane whose offset is out of range (keys past the segment) reads zeros -- no
// branch, no select, 32-bit offsets (descriptors only in free functions: hipcc's host pass, see igemm.hip)
__device__ __forceinline__ u32x4 attn_buf_load(const void* base, unsigned bytes, unsigned voff) {
    const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, bytes, 0x00020000);
    return __builtin_amdgcn_raw_buffer_load_b128(r, voff, 0, 0);
}

// register budget: occupancy (waves per SIMD) is what overlaps one wave's softmax with another's MFMAs; the
// 16-bit kernels are pinned to 4 waves/SIMD (<= 128 registers) for head dims <= 64 and 3 up to 96
// ONES (16-bit, head_dim < DP): the first zero-padding chunk of every V row holds 1.0 instead, so the PV MFMA
// accumulates the softmax denominator sum_k p[k] in the otherwise idle output column `head_dim` -- the 32 v_add_f32
// per tile and lane of the row sum disappear from the VALU, which is what bounds this kernel (PMC: 79 % VALU-busy,
// 45 % MFMA-busy).  The denominator then sums the SAME rounded probabilities the numerator multiplies.
// QB (query blocks per wave, 1 | 2): with QB = 2 a wave owns TWO 32-query blocks.  Their chains are independent, so
// inside ONE wave the softmax VALU work of block A (33 v_exp_f32 + ~80 other VALU per tile) issues under the QK^T MFMAs
// of block B, and block B's under the PV MFMAs of block A -- the VALU and the matrix pipe were adding up per wave
// (~480 + ~450 cycles per tile at d = 40) and only overlapped across waves.  K fragments / V transpose reads are shared
// by the two blocks (half the LDS reads per flop), a workgroup covers 256 queries (half the K/V tile fills per flop).
// Costs registers (two S / P / O / Q sets): 2 waves per SIMD, head dims <= 64 only.
template <typename T, int DP, bool ONES, int QB>
__global__ __launch_bounds__(256, (sizeof(T) == 2 ? (QB == 2 ? 2 : (DP <= 64 ? 4 : (DP <= 96 ? 3 : 2))) : 1))
void attention_kernel(const AttnParams p) {
    static_assert(!ONES || sizeof(T) == 2, "ones column: 16-bit kernels only");
    static_assert(QB == 1 || sizeof(T) == 2, "two query blocks per wave: 16-bit kernels only");
    constexpr bool F32 = std::is_same<T, float>::value;
    constexpr int EPC = Elt<T>::EPC;
    constexpr int DV = (DP + 31) / 32 * 32;
    constexpr int NDB = DV / 32;
    constexpr int KP = F32 ? (DP + 1) : (DP + 8);     // K tile pitch (elements)
    // f32: V^T tile [DV][BKV+1].  16-bit: V stays ROW-major [BKV][DV+16] (coalesced 16-byte writes, like K) and
    // the PV fragments are fetched with ds_read_b64_tr_b16, the LDS transpose read: inside a 16-lane group,
    // lane i / element j receives element (i&3) of the 8-byte row piece addressed by lane 4j + (i>>2)
    // (measured on gfx950, tools/probe/tr_probe.hip), i.e. a [4 keys][16 d] block comes back as lane = d,
    // element = key.  Pitch = 2*DV + 32 bytes puts the 4 key rows of a group on disjoint banks.
    constexpr int VP = F32 ? (BKV + 1) : v_pitch(DV);
    constexpr int NCH = DP / EPC;                      // 16-byte chunks per K/V row
    constexpr int NQ = F32 ? DP / 2 : DP / 16;         // Q fragments per lane
    constexpr int WQ = BQ * QB;                        // queries per workgroup
    extern __shared__ __attribute__((aligned(16))) char smem[];
    T* Ks = reinterpret_cast<T*>(smem);
    T* Vt = Ks + BKV * KP;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int hi = lane >> 5, l31 = lane & 31;
    // 1-D grid, query tile fastest.  For short sequences (per-view and SD self-attention, <= 2048 keys) the XCD
    // remap puts all query tiles of one (head, segment) -- which share its K/V -- on one XCD, whose L2 then
    // fetches K/V once (+3..9 %).  The 3-D attention (V*HW = 5120 keys at 32x32) is faster WITHOUT it (-17 %
    // with): every CU of the XCD then streams the same 1.3 MB of K/V through the same L2 channels at once,
    // while the round-robin order spreads 8 different (head, segment) streams over each XCD.
    const int lid = p.remap ? xcd_remap(blockIdx.x, gridDim.x) : blockIdx.x;
    const int qt = lid % p.nqt, hs_ = lid / p.nqt;
    const int head = hs_ % p.heads;
    const int4 sg = reinterpret_cast<const int4*>(p.seg)[hs_ / p.heads];
    const int q_row0 = sg.x, q_len = sg.y, kv_row0 = sg.z, kv_len = sg.w;
    if (qt * WQ >= q_len) return;  // uniform per workgroup
    const int d = p.d;

    // ---- Q fragments -> registers ----
    int q_local[QB];
    bool q_ok[QB];
    typename std::conditional<F32, float, typename AttnMma<typename std::conditional<F32, bf16_t, T>::type>::Frag>::type qf[QB][NQ];
#pragma unroll
    for (int qb = 0; qb < QB; ++qb) {
        q_local[qb] = qt * WQ + (wave * QB + qb) * 32 + l31;
        q_ok[qb] = q_local[qb] < q_len;
        const T* qp = reinterpret_cast<const T*>(p.q) + (size_t)(q_row0 + q_local[qb]) * p.ld_q + head * d;
        if constexpr (F32) {
#pragma unroll
            for (int kk = 0; kk < NQ; ++kk) {
                const int dk = 2 * kk + hi;
                qf[qb][kk] = (q_ok[qb] && dk < d) ? qp[dk] : 0.f;
            }
        } else {
#pragma unroll
            for (int kk = 0; kk < NQ; ++kk) {
                const int dk = kk * 16 + hi * 8;
                u32x4 raw = u32x4{0u, 0u, 0u, 0u};
                if (q_ok[qb] && dk < d) raw = *reinterpret_cast<const u32x4*>(qp + dk);
                qf[qb][kk] = __builtin_bit_cast(typename AttnMma<T>::Frag, raw);
            }
        }
    }

    f32x16 o[QB][NDB];
    float m_run[QB], l_run[QB];
#pragma unroll
    for (int qb = 0; qb < QB; ++qb) {
        m_run[qb] = -INFINITY;
        l_run[qb] = 0.f;
#pragma unroll
        for (int db = 0; db < NDB; ++db)
#pragma unroll
            for (int r = 0; r < 16; ++r) o[qb][db][r] = 0.f;
    }

    const T* kbase = reinterpret_cast<const T*>(p.k) + (size_t)kv_row0 * p.ld_k + head * d;
    const T* vbase = reinterpret_cast<const T*>(p.v) + (size_t)kv_row0 * p.ld_v + head * d;
    const int ntile = (kv_len + BKV - 1) / BKV;

    // staging registers: chunk idx = tid + 256*j of the [64 keys][NCH chunks] tile.  Loads go through a buffer descriptor over
    // this (segment, head) slice whose base advances one tile per iteration (scalar arithmetic): a lane keeps ONE 32-bit
    // offset per chunk and matrix, keys past kv_len are out of range and read as zeros -- no 64-bit address arithmetic, no
    // compare, no branch in the loop (the pointer form spilled 2..8 registers at the 128-register budget of DP <= 64).
    // Chunks that never change -- the zero padding from d to DP and, with ONES, the chunk of 1.0 right after the head in V --
    // are written to LDS once, before the loop; their lanes then neither load nor store.
    constexpr int NST = (BKV * NCH + 255) / 256;
    Chunk<T> rk[NST], rv[NST];
    unsigned st_ko[NST], st_vo[NST], live_bits = 0;
#pragma unroll
    for (int j = 0; j < NST; ++j) {
        const int idx = tid + 256 * j;
        const int key = idx / NCH, ch = idx - key * NCH;
        const bool in_tile = idx < BKV * NCH;
        const bool live = in_tile && ch * EPC < d;
        if (live) live_bits |= 1u << j;
        st_ko[j] = live ? ((unsigned)key * (unsigned)p.ld_k + (unsigned)(ch * EPC)) * (unsigned)sizeof(T) : 0xFFFFFFF0u;
        st_vo[j] = live ? ((unsigned)key * (unsigned)p.ld_v + (unsigned)(ch * EPC)) * (unsigned)sizeof(T) : 0xFFFFFFF0u;
        if (in_tile && !live) {
            Chunk<T> zk, zv;
            zk.zero();
            zv.zero();
            if constexpr (ONES) {
                if (ch * EPC == d) {
                    const uint32_t f = std::is_same<T, bf16_t>::value ? 0x3F803F80u : 0x3C003C00u;
                    zv.raw = u32x4{f, f, f, f};
                }
            }
            if constexpr (F32) {
#pragma unroll
                for (int i = 0; i < EPC; ++i) {
                    Ks[key * KP + ch * EPC + i] = 0.f;
                    Vt[(ch * EPC + i) * VP + key] = 0.f;
                }
            } else {
                *reinterpret_cast<u32x4*>(Ks + key * KP + ch * EPC) = zk.raw;
                *reinterpret_cast<u32x4*>(Vt + key * VP + ch * EPC) = zv.raw;
            }
        }
    }
    // extents of the slice (a segment's K / V slice stays below 4 GB)
    const unsigned kbytes = kv_len > 0 ? ((unsigned)(kv_len - 1) * (unsigned)p.ld_k + (unsigned)d) * (unsigned)sizeof(T) : 0u;
    const unsigned vbytes = kv_len > 0 ? ((unsigned)(kv_len - 1) * (unsigned)p.ld_v + (unsigned)d) * (unsigned)sizeof(T) : 0u;
    const unsigned ktile_bytes = (unsigned)BKV * (unsigned)p.ld_k * (unsigned)sizeof(T);
    const unsigned vtile_bytes = (unsigned)BKV * (unsigned)p.ld_v * (unsigned)sizeof(T);
    auto load_tile = [&](int kt) {
        const unsigned ko = (unsigned)kt * ktile_bytes, vo = (unsigned)kt * vtile_bytes;      // < kbytes / vbytes: kt < ntile
        const char* kt_base = reinterpret_cast<const char*>(kbase) + ko;
        const char* vt_base = reinterpret_cast<const char*>(vbase) + vo;
#pragma unroll
        for (int j = 0; j < NST; ++j) {
            rk[j].raw = attn_buf_load(kt_base, kbytes - ko, st_ko[j]);
            rv[j].raw = attn_buf_load(vt_base, vbytes - vo, st_vo[j]);
        }
    };
    auto store_tile = [&]() {
#pragma unroll
        for (int j = 0; j < NST; ++j) {
            const int idx = tid + 256 * j;
            if ((live_bits >> j) & 1u) {
                const int key = idx / NCH, ch = idx - key * NCH;   // (loop-invariant: hoisted by the compiler)
                if constexpr (F32) {
#pragma unroll
                    for (int i = 0; i < EPC; ++i) {
                        Ks[key * KP + ch * EPC + i] = rk[j].e[i];
                        Vt[(ch * EPC + i) * VP + key] = rv[j].e[i];
                    }
                } else {
                    *reinterpret_cast<u32x4*>(Ks + key * KP + ch * EPC) = rk[j].raw;
                    *reinterpret_cast<u32x4*>(Vt + key * VP + ch * EPC) = rv[j].raw;
                }
            }
        }
    };
    if (ntile > 0) load_tile(0);
    const float c = p.scale_log2e;

    for (int kt = 0; kt < ntile; ++kt) {
        // ---- K / V^T tile kt is in registers (loaded during the previous tile's math): park it in LDS ----
        store_tile();
        __syncthreads();
        if (kt + 1 < ntile) load_tile(kt + 1);   // next tile's global loads fly under this tile's MFMAs

        // ---- S^T = K Q^T for every query block of the wave (the K fragment is read once per block pair) ----
        f32x16 s[QB][2];
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
            for (int qb = 0; qb < QB; ++qb)
                s[qb][kb] = f32x16{0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};   // inline-constant C of the first MFMA
            const T* krow = Ks + (kb * 32 + l31) * KP;
            if constexpr (F32) {
#pragma unroll
                for (int kk = 0; kk < NQ; ++kk)
                    s[0][kb] = __builtin_amdgcn_mfma_f32_32x32x2f32(krow[2 * kk + hi], qf[0][kk], s[0][kb], 0, 0, 0);
            } else {
#pragma unroll
                for (int kk = 0; kk < NQ; ++kk) {
                    const auto a = *reinterpret_cast<const typename AttnMma<T>::Frag*>(krow + kk * 16 + hi * 8);
#pragma unroll
                    for (int qb = 0; qb < QB; ++qb) s[qb][kb] = AttnMma<T>::mma(a, qf[qb][kk], s[qb][kb]);
                }
            }
        }

        const bool ragged = kt == ntile - 1 && (kv_len & (BKV - 1)) != 0;
        typename std::conditional<F32, int, typename AttnMma<typename std::conditional<F32, bf16_t, T>::type>::Frag>::type pf[QB][4];
        // online softmax, per query = per lane column, in two parts.  Raw scores stay unscaled: max is taken on them
        // (scale > 0) and the scale rides in the exp2 argument, p = exp2(s*c - m*c): one fma + one exp per score.
        // part 1 (`sm_max`): mask (last tile only), running max, rescale of O when the max moved (wave-uniform branch);
        // part 2 (`sm_exp`): the 32 exponentials + conversion of P to the MFMA operand type -- branch-free, so that with
        // QB = 2 it can be scheduled INTO the PV MFMAs of the other query block.
        float mc[QB];
        auto sm_max = [&](const int qb) {
            if (ragged) {
#pragma unroll
                for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int key = kt * BKV + kb * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
                        s[qb][kb][r] = key < kv_len ? s[qb][kb][r] : -INFINITY;
                    }
            }
            float mx = s[qb][0][0];
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int r = 0; r < 16; ++r) mx = fmaxf(mx, s[qb][kb][r]);
            mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
            const float m_new = fmaxf(m_run[qb], mx);
            const float alpha = __builtin_amdgcn_exp2f((m_run[qb] - m_new) * c);   // raw v_exp_f32: arguments are <= 0, no denormal-range fix-up needed
            m_run[qb] = m_new;
            mc[qb] = -m_new * c;
            if constexpr (!ONES) l_run[qb] *= alpha;
            if (!__all(alpha == 1.0f)) {   // once the running max has settled the O rescale is skipped (wave-uniform)
#pragma unroll
                for (int db = 0; db < NDB; ++db)
#pragma unroll
                    for (int r = 0; r < 16; ++r) o[qb][db][r] *= alpha;
            }
        };
        auto sm_exp = [&](const int qb) {
            float rs = 0.f;
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float pv = __builtin_amdgcn_exp2f(fmaf(s[qb][kb][r], c, mc[qb]));
                    s[qb][kb][r] = pv;
                    if constexpr (!ONES) rs += pv;
                }
            if constexpr (!ONES) l_run[qb] += rs;
            if constexpr (!F32) {
#pragma unroll
                for (int ks = 0; ks < 4; ++ks)
#pragma unroll
                    for (int j = 0; j < 8; ++j) pf[qb][ks][j] = from_f32<T>(s[qb][ks >> 1][(ks & 1) * 8 + j]);
            }
        };

        if constexpr (F32) {
            sm_max(0);
            sm_exp(0);
            // ---- O^T += V^T P^T ----
#pragma unroll
            for (int db = 0; db < NDB; ++db) {
                const T* vrow = Vt + (db * 32 + l31) * VP;
#pragma unroll
                for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int key = kb * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
                        o[0][db] = __builtin_amdgcn_mfma_f32_32x32x2f32(vrow[key], s[0][kb][r], o[0][db], 0, 0, 0);
                    }
            }
        } else {
            // ---- O^T += V^T P^T: source piece of this lane for the transpose read: key row 4*hi + (s>>2) (+8 for the upper
            // half of the fragment), d columns (gi&1)*16 + 4*(s&3) .. +3, with gi = lane>>4, s = lane&15
            const int gi = lane >> 4, sl = lane & 15;
            const T* vsrc = Vt + (4 * (gi >> 1) + (sl >> 2)) * VP + (gi & 1) * 16 + 4 * (sl & 3);
            auto pv = [&](const int qb) {
#pragma unroll
                for (int db = 0; db < NDB; ++db) {
#pragma unroll
                    for (int ks = 0; ks < 4; ++ks) {
                        const T* pa = vsrc + (ks * 16) * VP + db * 32;
                        const auto lo = AttnMma<T>::tr_read(pa);
                        const auto up = AttnMma<T>::tr_read(pa + 8 * VP);
                        const typename AttnMma<T>::Frag a = __builtin_shufflevector(lo, up, 0, 1, 2, 3, 4, 5, 6, 7);
                        o[qb][db] = AttnMma<T>::mma(a, pf[qb][ks], o[qb][db]);
                    }
                }
            };
            if constexpr (QB == 1) {
                sm_max(0);
                sm_exp(0);
                pv(0);
            } else {
                // block A's softmax runs while block B's QK^T MFMAs are still in the matrix pipe; block B's exponentials are
                // then scheduled between the PV MFMAs of block A (one MFMA, then a slice of the ~100 VALU instructions)
                sm_max(0);
                sm_exp(0);
                sm_max(1);
                __builtin_amdgcn_sched_barrier(0);
                pv(0);
                sm_exp(1);
#pragma unroll
                for (int g = 0; g < NDB * 4; ++g) {
                    __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);                        // the fragment's two transpose reads
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                        // one MFMA
                    __builtin_amdgcn_sched_group_barrier(0x002, (96 + NDB * 4 - 1) / (NDB * 4), 0);   // a slice of the VALU work
                }
                __builtin_amdgcn_sched_barrier(0);
                pv(1);
            }
        }
        __syncthreads();
    }

    // ---- normalise and write O[q][d] ----
#pragma unroll
    for (int qb = 0; qb < QB; ++qb) {
        float l_tot;
        if constexpr (ONES) {
            // output column d (hi = 0 lanes) / d + 4 (hi = 1 lanes) of O^T: both inside the ones chunk, both the full sum
            l_tot = 0.f;
#pragma unroll
            for (int db = 0; db < NDB; ++db)
#pragma unroll
                for (int q8 = 0; q8 < 4; ++q8)
                    if (db * 32 + q8 * 8 == d) l_tot = o[qb][db][4 * q8];
        } else {
            l_tot = l_run[qb] + __shfl_xor(l_run[qb], 32, 64);
        }
        const float inv = l_tot > 0.f ? 1.0f / l_tot : 0.f;
        if (!q_ok[qb]) continue;
        if (p.lse && hi == 0) p.lse[(size_t)head * p.lse_ld + q_row0 + q_local[qb]] = m_run[qb] * p.scale_log2e + __log2f(l_tot);   // P = exp2(s*c - lse)
        T* op = reinterpret_cast<T*>(p.out) + (size_t)(q_row0 + q_local[qb]) * p.ld_o + head * d;
#pragma unroll
        for (int db = 0; db < NDB; ++db)
#pragma unroll
            for (int r4 = 0; r4 < 4; ++r4) {
                const int dd = db * 32 + 8 * r4 + 4 * hi;
                if (dd < d) {
                    if constexpr (F32) {
                        f32x4 w = {o[qb][db][r4 * 4 + 0] * inv, o[qb][db][r4 * 4 + 1] * inv, o[qb][db][r4 * 4 + 2] * inv, o[qb][db][r4 * 4 + 3] * inv};
                        *reinterpret_cast<f32x4*>(op + dd) = w;
                    } else {
                        union { T e[4]; u32x2 raw; } w;
#pragma unroll
                        for (int e = 0; e < 4; ++e) w.e[e] = from_f32<T>(o[qb][db][r4 * 4 + e] * inv);
                        *reinterpret_cast<u32x2*>(op + dd) = w.raw;
                    }
                }
            }
    }
}


// ---- wide-head fallback (head_dim > 160: the VAE mid-block's single 512-wide head) ----------------
// One wave per query row; the head dimension is spread over the 64 lanes (16-byte chunks), keys are
// streamed from L2 two at a time, dot products are wave-reduced, online softmax in fp32.  VALU only:
// this path carries < 0.1 % of the FLOPs of a sample and is kept simple.
template <typename T, int NCH>
__global__ __launch_bounds__(256) void attention_wide_kernel(const AttnParams p) {
    constexpr int EPC = Elt<T>::EPC;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int head = blockIdx.y;
    const int4 sg = reinterpret_cast<const int4*>(p.seg)[blockIdx.z];
    const int q_row0 = sg.x, q_len = sg.y, kv_row0 = sg.z, kv_len = sg.w;
    const int q_local = blockIdx.x * 4 + wave;
    if (q_local >= q_len) return;
    const int d = p.d, nch = d / EPC;
    const T* qp = reinterpret_cast<const T*>(p.q) + (size_t)(q_row0 + q_local) * p.ld_q + head * d;
    float qv[NCH][EPC], ov[NCH][EPC];
#pragma unroll
    for (int j = 0; j < NCH; ++j) {
        const int ch = lane + 64 * j;
        Chunk<T> c;
        if (ch < nch) c = load_chunk<T>(qp + ch * EPC); else c.zero();
#pragma unroll
        for (int i = 0; i < EPC; ++i) { qv[j][i] = c.get(i) * p.scale_log2e; ov[j][i] = 0.f; }
    }
    const T* kb = reinterpret_cast<const T*>(p.k) + (size_t)kv_row0 * p.ld_k + head * d;
    const T* vb = reinterpret_cast<const T*>(p.v) + (size_t)kv_row0 * p.ld_v + head * d;
    float m_run = -INFINITY, l_run = 0.f;
    for (int key = 0; key < kv_len; ++key) {
        float part = 0.f;
        Chunk<T> vc[NCH];
#pragma unroll
        for (int j = 0; j < NCH; ++j) {
            const int ch = lane + 64 * j;
            Chunk<T> kc;
            if (ch < nch) {
                kc = load_chunk<T>(kb + (size_t)key * p.ld_k + ch * EPC);
                vc[j] = load_chunk<T>(vb + (size_t)key * p.ld_v + ch * EPC);
            } else {
                kc.zero();
                vc[j].zero();
            }
#pragma unroll
            for (int i = 0; i < EPC; ++i) part += qv[j][i] * kc.get(i);
        }
        const float sv = wave_sum(part);
        const float m_new = fmaxf(m_run, sv);
        const float alpha = exp2f(m_run - m_new), pv = exp2f(sv - m_new);
        m_run = m_new;
        l_run = l_run * alpha + pv;
#pragma unroll
        for (int j = 0; j < NCH; ++j)
#pragma unroll
            for (int i = 0; i < EPC; ++i) ov[j][i] = ov[j][i] * alpha + pv * vc[j].get(i);
    }
    const float inv = l_run > 0.f ? 1.0f / l_run : 0.f;
    T* op = reinterpret_cast<T*>(p.out) + (size_t)(q_row0 + q_local) * p.ld_o + head * d;
#pragma unroll
    for (int j = 0; j < NCH; ++j) {
        const int ch = lane + 64 * j;
        if (ch < nch) {
            Chunk<T> c;
#pragma unroll
            for (int i = 0; i < EPC; ++i) c.set(i, ov[j][i] * inv);
            store_chunk<T>(op + ch * EPC, c);
        }
    }
}

// ---- wide heads on the matrix cores: the head dimension split over the four waves of a workgroup ----------------
// The VAE mid-block attention is ONE head of 512 columns over 1024 tokens per image.  The flash kernel
// above keeps O^T[d][q] and the Q fragments of a 32-query block in one wave's registers: at d = 512 that is 256 + 128 registers.
// Here the four waves of a workgroup share the SAME 32 queries and split d into four slices of DSL = d / 4 columns:
//     wave w:  S_w^T[key][q] = K[:, slice w] Q[:, slice w]^T      (a partial sum over its slice: DSL / 16 MFMAs per 32 keys)
//     all:     S^T = S_0^T + S_1^T + S_2^T + S_3^T                 (exchanged through LDS; every wave then holds the full scores
//                                                                    of its lanes' queries and runs the same online softmax)
//     wave w:  O^T[slice w][q] += V[:, slice w]^T P^T              (its DSL output columns: DSL / 32 x 2 MFMAs per 32 keys)
// K / V tiles of 32 keys x d are staged through LDS by all 256 threads (register prefetch of the next tile under the math, as
// above); same transposed products, accumulator-row key order and ds_read_b64_tr_b16 V fragments as attention_kernel.
// 16-bit types, head_dim = 4 * DSL with DSL a multiple of 32 (instantiated: DSL = 128, the VAE's 512).
constexpr int BKW = 32;      // keys per tile
template <typename T, int DSL>
__global__ __launch_bounds__(256, 2)
void attention_dsplit_kernel(const AttnParams p) {
    static_assert(sizeof(T) == 2 && DSL % 32 == 0 && DSL <= 128, "d-split attention: 16-bit, slices of 32..128 columns");
    using Frag = typename AttnMma<T>::Frag;
    constexpr int EPC = 8, D = 4 * DSL;
    constexpr int NDB = DSL / 32, NQ = DSL / 16;
    constexpr int KP = D + 8, VP = v_pitch(D);
    constexpr int NCH = D / EPC;                         // 16-byte chunks per K / V row
    constexpr int NST = BKW * NCH / 256;                 // chunks per thread and matrix
    static_assert(BKW * NCH % 256 == 0, "whole chunks per thread");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    T* Ks = reinterpret_cast<T*>(smem);
    T* Vt = Ks + BKW * KP;
    float* Xs = reinterpret_cast<float*>(Vt + BKW * VP);     // partial scores: [wave][lane][16]

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int hi = lane >> 5, l31 = lane & 31;
    const int lid = p.remap ? xcd_remap(blockIdx.x, gridDim.x) : blockIdx.x;
    const int qt = lid % p.nqt, hs_ = lid / p.nqt;
    const int head = hs_ % p.heads;
    const int4 sg = reinterpret_cast<const int4*>(p.seg)[hs_ / p.heads];
    const int q_row0 = sg.x, q_len = sg.y, kv_row0 = sg.z, kv_len = sg.w;
    if (qt * 32 >= q_len) return;                        // uniform per workgroup
    const int dsl0 = wave * DSL;                          // this wave's columns of the head

    const int q_local = qt * 32 + l31;
    const bool q_ok = q_local < q_len;
    Frag qf[NQ];
    {
        const T* qp = reinterpret_cast<const T*>(p.q) + (size_t)(q_row0 + q_local) * p.ld_q + head * D + dsl0;
#pragma unroll
        for (int kk = 0; kk < NQ; ++kk) {
            u32x4 raw = u32x4{0u, 0u, 0u, 0u};
            if (q_ok) raw = *reinterpret_cast<const u32x4*>(qp + kk * 16 + hi * 8);
            qf[kk] = __builtin_bit_cast(Frag, raw);
        }
    }
    f32x16 o[NDB];
#pragma unroll
    for (int db = 0; db < NDB; ++db)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[db][r] = 0.f;
    float m_run = -INFINITY, l_run = 0.f;

    const void* const kseg = reinterpret_cast<const T*>(p.k) + (size_t)kv_row0 * p.ld_k + head * D;
    const void* const vseg = reinterpret_cast<const T*>(p.v) + (size_t)kv_row0 * p.ld_v + head * D;
    // extents of this (segment, head) slice: keys past kv_len are out of range and read as zeros (a segment's K / V slice stays
    // below 4 GB: 32-bit offsets)
    const unsigned kbytes = kv_len > 0 ? ((unsigned)(kv_len - 1) * (unsigned)p.ld_k + (unsigned)D) * 2u : 0u;
    const unsigned vbytes = kv_len > 0 ? ((unsigned)(kv_len - 1) * (unsigned)p.ld_v + (unsigned)D) * 2u : 0u;
    const int ntile = (kv_len + BKW - 1) / BKW;
    // thread t holds chunk (t & (NCH-1)) of keys (t / NCH) + (256 / NCH) j of every tile
    static_assert((NCH & (NCH - 1)) == 0 && 256 % NCH == 0, "chunks per row: a power of two");
    constexpr int KSTEP = 256 / NCH;
    const int st_key = tid / NCH, st_ch = tid & (NCH - 1);
    const unsigned st_k0 = ((unsigned)st_key * (unsigned)p.ld_k + (unsigned)st_ch * EPC) * 2u;
    const unsigned st_v0 = ((unsigned)st_key * (unsigned)p.ld_v + (unsigned)st_ch * EPC) * 2u;
    u32x4 rk[NST], rv[NST];
    auto load_tile = [&](int kt) {
#pragma unroll
        for (int j = 0; j < NST; ++j) {
            const unsigned row = (unsigned)(kt * BKW + KSTEP * j);
            rk[j] = attn_buf_load(kseg, kbytes, st_k0 + row * (unsigned)p.ld_k * 2u);
            rv[j] = attn_buf_load(vseg, vbytes, st_v0 + row * (unsigned)p.ld_v * 2u);
        }
    };
    auto store_tile = [&]() {
#pragma unroll
        for (int j = 0; j < NST; ++j) {
            *reinterpret_cast<u32x4*>(Ks + (st_key + KSTEP * j) * KP + st_ch * EPC) = rk[j];
            *reinterpret_cast<u32x4*>(Vt + (st_key + KSTEP * j) * VP + st_ch * EPC) = rv[j];
        }
    };
    if (ntile > 0) load_tile(0);
    const float c = p.scale_log2e;
    const int gi = lane >> 4, sl = lane & 15;
    const T* vsrc = Vt + (4 * (gi >> 1) + (sl >> 2)) * VP + (gi & 1) * 16 + 4 * (sl & 3) + dsl0;

    for (int kt = 0; kt < ntile; ++kt) {
        store_tile();
        __syncthreads();
        if (kt + 1 < ntile) load_tile(kt + 1);
        // ---- partial S^T over this wave's slice of d ----
        f32x16 s = f32x16{0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        {
            const T* krow = Ks + l31 * KP + dsl0;
#pragma unroll
            for (int kk = 0; kk < NQ; ++kk) {
                const auto a = *reinterpret_cast<const Frag*>(krow + kk * 16 + hi * 8);
                s = AttnMma<T>::mma(a, qf[kk], s);
            }
        }
        // ---- exchange: every wave adds the other three partial tiles (same lane = same query column / key rows) ----
        {
            float* mine = Xs + (wave * 64 + lane) * 16;
#pragma unroll
            for (int q4 = 0; q4 < 4; ++q4) *reinterpret_cast<f32x4*>(mine + 4 * q4) = f32x4{s[4 * q4], s[4 * q4 + 1], s[4 * q4 + 2], s[4 * q4 + 3]};
            __syncthreads();
#pragma unroll
            for (int w = 1; w < 4; ++w) {
                const float* other = Xs + (((wave + w) & 3) * 64 + lane) * 16;
#pragma unroll
                for (int q4 = 0; q4 < 4; ++q4) {
                    const f32x4 t = *reinterpret_cast<const f32x4*>(other + 4 * q4);
#pragma unroll
                    for (int e = 0; e < 4; ++e) s[4 * q4 + e] += t[e];
                }
            }
        }
        // (the four partial sums are added in a wave-dependent order: each wave's softmax state may differ from its neighbours'
        //  in the last bit -- every wave normalises ITS output columns with ITS OWN denominator, so rows stay consistent)
        if (kt == ntile - 1 && (kv_len & (BKW - 1)) != 0) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int key = kt * BKW + (r & 3) + 8 * (r >> 2) + 4 * hi;
                s[r] = key < kv_len ? s[r] : -INFINITY;
            }
        }
        float mx = s[0];
#pragma unroll
        for (int r = 1; r < 16; ++r) mx = fmaxf(mx, s[r]);
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        const float m_new = fmaxf(m_run, mx);
        const float alpha = __builtin_amdgcn_exp2f((m_run - m_new) * c);
        m_run = m_new;
        const float mc = -m_new * c;
        l_run *= alpha;
        if (!__all(alpha == 1.0f)) {
#pragma unroll
            for (int db = 0; db < NDB; ++db)
#pragma unroll
                for (int r = 0; r < 16; ++r) o[db][r] *= alpha;
        }
        float rs = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            s[r] = __builtin_amdgcn_exp2f(fmaf(s[r], c, mc));
            rs += s[r];
        }
        l_run += rs;
        Frag pf[2];
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int j = 0; j < 8; ++j) pf[ks][j] = from_f32<T>(s[ks * 8 + j]);
        // ---- O^T[slice][q] += V[:, slice]^T P^T ----
#pragma unroll
        for (int db = 0; db < NDB; ++db)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                const T* pa = vsrc + (ks * 16) * VP + db * 32;
                const auto lo = AttnMma<T>::tr_read(pa);
                const auto up = AttnMma<T>::tr_read(pa + 8 * VP);
                const Frag a = __builtin_shufflevector(lo, up, 0, 1, 2, 3, 4, 5, 6, 7);
                o[db] = AttnMma<T>::mma(a, pf[ks], o[db]);
            }
        __syncthreads();
    }
    const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
    const float inv = l_tot > 0.f ? 1.0f / l_tot : 0.f;
    if (!q_ok) return;
    T* op = reinterpret_cast<T*>(p.out) + (size_t)(q_row0 + q_local) * p.ld_o + head * D + dsl0;
#pragma unroll
    for (int db = 0; db < NDB; ++db)
#pragma unroll
        for (int r4 = 0; r4 < 4; ++r4) {
            union { T e[4]; u32x2 raw; } w;
#pragma unroll
            for (int e = 0; e < 4; ++e) w.e[e] = from_f32<T>(o[db][r4 * 4 + e] * inv);
            *reinterpret_cast<u32x2*>(op + db * 32 + 8 * r4 + 4 * hi) = w.raw;
        }
}

template <typename T, int DSL> static int launch_attn_dsplit(AttnParams p, int n_seg, int max_q_len, hipStream_t s) {
    constexpr int D = 4 * DSL;
    constexpr int smem = (BKW * (D + 8) + BKW * v_pitch(D)) * 2 + 4 * 64 * 16 * 4;
    auto kern = attention_dsplit_kernel<T, DSL>;
    static std::atomic<uint64_t> attr_done{0};
    if (int rc0 = ensure_dyn_smem(reinterpret_cast<const void*>(kern), smem, attr_done)) return rc0;
    p.nqt = (max_q_len + 31) / 32;
    p.remap = 1;                 // all query tiles of a (head, segment) on one XCD: its K / V stream once through that L2
    hipLaunchKernelGGL(kern, dim3(p.nqt * p.heads * n_seg), dim3(256), smem, s, p);
    return check_launch();
}

template <typename T> static int launch_attn_wide(const AttnParams& p, int n_seg, int max_q_len, hipStream_t s) {
    const int nch = p.d / Elt<T>::EPC;
    dim3 grid((max_q_len + 3) / 4, p.heads, n_seg);
    if (nch <= 64) hipLaunchKernelGGL((attention_wide_kernel<T, 1>), grid, dim3(256), 0, s, p);
    else if (nch <= 128) hipLaunchKernelGGL((attention_wide_kernel<T, 2>), grid, dim3(256), 0, s, p);
    else if (nch <= 256) hipLaunchKernelGGL((attention_wide_kernel<T, 4>), grid, dim3(256), 0, s, p);
    else return set_error(MVLDM_ERR_UNSUPPORTED, "attention: head_dim %d too wide", p.d);
    return check_launch();
}

// MVLDM_ATTN_QB=2 selects the two-query-blocks-per-wave form where it applies (A/B knob).  Measured on MI355X at 64 scenes
// (tools/attn_bench.py, profiles/r02_attention_pmc.json): 3-D attention 8 x 40 over 5120 keys 6.15 ms vs 6.34 ms with one
// block (+3 %), per-view / SD self-attention -2...-6 %, the block-B-exponentials-between-block-A-MFMAs schedule 6.41 ms:
// the kernel is bound by VALU issue (70 % VALU-active at 42 % MFMA-busy, 11 VALU instructions per MFMA at d = 40), which
// neither sharing fragments nor re-ordering reduces.  Default: one block per wave.
static const int kEnvAttnQB = knob_int("MVLDM_ATTN_QB", 1);

template <typename T, int DP, bool ONES, int QB> static int launch_attn_k(AttnParams p, int n_seg, int max_q_len, hipStream_t s) {
    constexpr bool F32 = std::is_same<T, float>::value;
    constexpr int DV = (DP + 31) / 32 * 32;
    constexpr int KP = F32 ? (DP + 1) : (DP + 8);
    constexpr int VP = F32 ? (BKV + 1) : v_pitch(DV);
    constexpr int smem = (BKV * KP + (F32 ? DV * VP : BKV * VP)) * (int)sizeof(T);
    auto kern = attention_kernel<T, DP, ONES, QB>;
    static std::atomic<uint64_t> attr_done{0};
    if (int rc0 = ensure_dyn_smem(reinterpret_cast<const void*>(kern), smem, attr_done)) return rc0;
    p.nqt = (max_q_len + BQ * QB - 1) / (BQ * QB);
    p.remap = max_q_len <= 2048;
    dim3 grid(p.nqt * p.heads * n_seg);
    hipLaunchKernelGGL(kern, grid, dim3(256), smem, s, p);
    return check_launch();
}

template <typename T, int DP, bool ONES> static int launch_attn_q(AttnParams p, int n_seg, int max_q_len, hipStream_t s) {
    // two query blocks per wave (256 queries per workgroup) for the narrow heads of the long sequences: needs enough query
    // tiles to fill the chip with half as many workgroups
    if constexpr (sizeof(T) == 2 && DP <= 64) {
        const long wgs2 = (long)((max_q_len + 2 * BQ - 1) / (2 * BQ)) * p.heads * n_seg;
        if (kEnvAttnQB == 2 && max_q_len >= 2 * BQ && wgs2 >= 512) return launch_attn_k<T, DP, ONES, 2>(p, n_seg, max_q_len, s);
    }
    return launch_attn_k<T, DP, ONES, 1>(p, n_seg, max_q_len, s);
}

template <typename T, int DP> static int launch_attn(AttnParams p, int n_seg, int max_q_len, hipStream_t s) {
    if constexpr (sizeof(T) == 2) {
        if (p.d % 8 == 0 && p.d < DP) return launch_attn_q<T, DP, true>(p, n_seg, max_q_len, s);
    }
    return launch_attn_q<T, DP, false>(p, n_seg, max_q_len, s);
}

int attention_run(const void* q, const void* k, const void* v, void* out, int ld_q, int ld_k, int ld_v, int ld_o,
                  int heads, int head_dim, const int32_t* seg, int n_seg, int max_q_len, float scale, int dtype,
                  float* lse, int lse_ld, hipStream_t s) {
    const int epc = dtype == MVLDM_F32 ? 4 : 8;
    MVLDM_REQUIRE(head_dim > 0 && head_dim % epc == 0, "attention: head_dim %d must be a multiple of %d", head_dim, epc);
    if (n_seg == 0 || max_q_len == 0) return MVLDM_OK;   // empty: buffers may be null
    MVLDM_REQUIRE(q && k && v && out && seg, "attention: null pointer");
    MVLDM_REQUIRE(ld_q % epc == 0 && ld_k % epc == 0 && ld_v % epc == 0 && ld_o % 4 == 0, "attention: row strides must keep 16-byte alignment");
    if (n_seg == 0 || max_q_len == 0) return MVLDM_OK;
    MVLDM_REQUIRE(!lse || head_dim <= 160, "attention: log-sum-exp output only on the MFMA kernels (head_dim <= 160)");
    AttnParams p{q, k, v, out, seg, ld_q, ld_k, ld_v, ld_o, heads, head_dim, scale * 1.4426950408889634f, 0, 0, lse, lse_ld};
    const int dp = (head_dim + 15) / 16 * 16;
    return dispatch_dtype(dtype, [&](auto t) {
        using T = decltype(t);
        if (head_dim > 160) {
            if constexpr (sizeof(T) == 2) {       // the VAE's single 512-wide head: d split over the four waves, MFMA (MVLDM_ATTN_WIDE_VALU=1: the VALU form)
                static const bool valu = knob_int("MVLDM_ATTN_WIDE_VALU", 0) != 0;
                if (head_dim == 512 && !valu && !p.lse) return launch_attn_dsplit<T, 128>(p, n_seg, max_q_len, s);
            }
            return launch_attn_wide<T>(p, n_seg, max_q_len, s);
        }
        switch (dp) {
            case 16: return launch_attn<T, 16>(p, n_seg, max_q_len, s);
            case 32: return launch_attn<T, 32>(p, n_seg, max_q_len, s);
            case 48: return launch_attn<T, 48>(p, n_seg, max_q_len, s);
            case 64: return launch_attn<T, 64>(p, n_seg, max_q_len, s);
            case 80: return launch_attn<T, 80>(p, n_seg, max_q_len, s);
            case 96: return launch_attn<T, 96>(p, n_seg, max_q_len, s);
            case 112: return launch_attn<T, 112>(p, n_seg, max_q_len, s);
            case 128: return launch_attn<T, 128>(p, n_seg, max_q_len, s);
            case 144: return launch_attn<T, 144>(p, n_seg, max_q_len, s);
            case 160: return launch_attn<T, 160>(p, n_seg, max_q_len, s);
            default: return set_error(MVLDM_ERR_UNSUPPORTED, "attention: head_dim %d", head_dim);
        }
    });
}

// ---- combine of two attention results over disjoint key sets (see mvldm_attention_merge) -------------------------
template <typename T>
__global__ __launch_bounds__(256) void attn_merge_kernel(const T* __restrict__ oa, const float* __restrict__ lse_a, const T* __restrict__ ob,
                                                         const float* __restrict__ lse_b, T* __restrict__ out, const int32_t* __restrict__ a_img,
                                                         const int32_t* __restrict__ b_img, const int32_t* __restrict__ out_img, int n_img, int tokens,
                                                         int heads, int d, int ld_a, int ld_b, int ld_o, int lla, int llb) {
    constexpr int EPC = Elt<T>::EPC;
    const int cpr = heads * d / EPC;                     // chunks per row
    const size_t total = (size_t)n_img * tokens * cpr;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int c = (int)(i % cpr);
        const size_t rt = i / cpr;
        const int t = (int)(rt % tokens), k = (int)(rt / tokens);
        const size_t ra = (size_t)a_img[k] * tokens + t, rb = (size_t)b_img[k] * tokens + t, ro = (size_t)out_img[k] * tokens + t;
        const int head = c * EPC / d;
        const float la = lse_a[(size_t)head * lla + ra], lb = lse_b[(size_t)head * llb + rb];
        const float mx = fmaxf(la, lb);
        const float ea = __builtin_amdgcn_exp2f(la - mx), eb = __builtin_amdgcn_exp2f(lb - mx);
        const float inv = 1.0f / (ea + eb);
        const float wa = ea * inv, wb = eb * inv;
        const Chunk<T> va = load_chunk<T>(oa + ra * ld_a + c * EPC), vb = load_chunk<T>(ob + rb * ld_b + c * EPC);
        Chunk<T> o;
#pragma unroll
        for (int e = 0; e < EPC; ++e) o.set(e, va.get(e) * wa + vb.get(e) * wb);
        store_chunk<T>(out + ro * ld_o + c * EPC, o);
    }
}

int attention_merge_run(const void* oa, const float* lse_a, const void* ob, const float* lse_b, void* out, const int32_t* a_img,
                        const int32_t* b_img, const int32_t* out_img, int n_img, int tokens, int heads, int head_dim, int ld_a, int ld_b,
                        int ld_o, int lse_ld_a, int lse_ld_b, int dtype, hipStream_t s) {
    const int epc = dtype == MVLDM_F32 ? 4 : 8;
    if (n_img == 0 || tokens == 0) return MVLDM_OK;
    MVLDM_REQUIRE(oa && ob && out && lse_a && lse_b && a_img && b_img && out_img, "attention_merge: null pointer");
    MVLDM_REQUIRE(head_dim % epc == 0 && ld_a % epc == 0 && ld_b % epc == 0 && ld_o % epc == 0, "attention_merge: 16-byte alignment");
    const size_t total = (size_t)n_img * tokens * (heads * head_dim / epc);
    return dispatch_dtype(dtype, [&](auto t) {
        using T = decltype(t);
        hipLaunchKernelGGL(attn_merge_kernel<T>, dim3((unsigned)std::min<size_t>((total + 255) / 256, 16384)), dim3(256), 0, s, reinterpret_cast<const T*>(oa),
                           lse_a, reinterpret_cast<const T*>(ob), lse_b, reinterpret_cast<T*>(out), a_img, b_img, out_img, n_img, tokens, heads, head_dim,
                           ld_a, ld_b, ld_o, lse_ld_a, lse_ld_b);
        return check_launch();
    });
}

}  // namespace mvldm

extern "C" int mvldm_attention_fwd(const void* q, const void* k, const void* v, void* out, int ld_q, int ld_k, int ld_v,
                                   int ld_o, int heads, int head_dim, const int32_t* seg, int n_seg, int max_q_len,
                                   float scale, int dtype, float* lse, int lse_ld, mvldm_stream_t stream) {
    return mvldm::attention_run(q, k, v, out, ld_q, ld_k, ld_v, ld_o, heads, head_dim, seg, n_seg, max_q_len, scale,
                                dtype, lse, lse_ld, (hipStream_t)stream);
}
extern "C" int mvldm_attention_merge(const void* oa, const float* lse_a, const void* ob, const float* lse_b, void* out, const int32_t* a_img,
                                     const int32_t* b_img, const int32_t* out_img, int n_img, int tokens, int heads, int head_dim, int ld_a,
                                     int ld_b, int ld_o, int lse_ld_a, int lse_ld_b, int dtype, mvldm_stream_t stream) {
    return mvldm::attention_merge_run(oa, lse_a, ob, lse_b, out, a_img, b_img, out_img, n_img, tokens, heads, head_dim, ld_a, ld_b, ld_o, lse_ld_a,
                               lse_ld_b, dtype, (hipStream_t)stream);
}
