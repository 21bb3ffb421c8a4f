// Weight-stationary Linear for K = 320: tile 14 of the implicit-GEMM family (include/mvldm.h: mvldm_igemm_fwd; 16-bit activations,
// one source, K == 320, packed width a multiple of 320).  Round 4.
//
// The level-0 Linears of the UNet (QKV, to_out, proj_in / proj_out, GEGLU: 11 ms of a DDIM step at 64 scenes) have K = 320 and
// half a million rows: five K-tiles per output tile.  tools/pw_probe.py on tile 13 shows what a tiled kernel cannot avoid there: per
// tile the store burst (160 KB per CU, every CU at once) and its issue time, the exposed GELU (GEGLU: 250 us of 1350), one DMA
// latency.  This kernel has no output tiles:
//   * a WAVE keeps the weights of its 32 output columns -- 32 x 320, 80 VGPRs -- in registers for the life of the kernel (W is read
//     from L2 once per workgroup); ten waves = 320 columns per workgroup;
//   * the workgroup streams 64-row slots of the activation matrix through a 3-slot LDS-DMA ring (40 KB per slot, full 128-byte lines,
//     the XOR-swizzled [k-block][row][128 B] layout of igemm.hip); 320 flop per L2->LDS byte;
//   * per 32-row block a wave issues 20 MFMAs (transposed product, permuted weight rows: a lane ends up with 16 consecutive output
//     columns of ONE row, like tile 13) into one of two 16-register accumulators; the finished block's epilogue -- residual add,
//     GEGLU product with the exact-erf GELU, conversion, two 16-byte stores -- is interleaved with the NEXT block's MFMAs (separate
//     pipes), so the matrix pipe sees one uninterrupted stream and the stores leave at a steady two per 20 MFMAs and wave.
// An eleventh wave is the LOADER: it issues every LDS-DMA piece and does every counted wait for them, so the compute waves' stores are
// never held back by a load wait (VMEM returns in order per wave: see the loader's comment in the kernel).
// GEGLU: a wave's 32 weight rows are 16 value + 16 gate columns (the packed weight alternates [32 value | 32 gate]); both halves use
// the same permutation, so a lane holds value and gate of its 8 columns: one 16-byte store per lane and block.
#include <algorithm>
#include <cstdlib>

#include "common.h"

namespace mvldm {

struct LinWSParams {
    const void* a; const void* w; const float* bias; const void* residual; void* dst;
    int M, n_out, n_pad, n_dst, dst_ld;
    int n_slices, groups, n_slots;     // column slices of 320 packed columns; row groups per XCD; 64-row slots in all
    int nt_store;
    float out_scale;
    unsigned a_bytes, w_bytes, bias_bytes, res_bytes, dst_bytes;
};

#ifdef MVLDM_EXPERIMENTS
static const int kWsFake = knob_int("MVLDM_WS_FAKE", 0);   // 1: no A traffic, 4: no stores
#else
static constexpr int kWsFake = 0;
#endif
#ifdef MVLDM_EXPERIMENTS_NOGELU
#define WS_GELU(x) (x)
#else
#define WS_GELU(x) gelu_erf_16(x)
#endif

constexpr unsigned kWsOob = 0xFFFFFFF0u;
constexpr int WS_K = 320, WS_KS = WS_K / 16, WS_NW = 10, WS_BN = 320, WS_ROWS = 64;
constexpr int WS_SLOT = WS_ROWS * WS_K * 2;             // 40 KB: [5 k-blocks][64 rows][128 B]
constexpr int WS_RING = 3 * WS_SLOT;
constexpr int WS_SLAB = WS_RING;                        // bias of the workgroup's 320 packed columns (fp32)
constexpr int WS_SMEM = WS_SLAB + 2048;

constexpr int ws_wait(int n) { return (n & 15) | ((n >> 4) << 14) | 0x0F70; }   // s_waitcnt vmcnt(n) only (gfx9 encoding)

template <typename T> struct WsMma;
template <> struct WsMma<bf16_t> {
    using Frag = bf16x8;
    static __device__ __forceinline__ f32x16 mma(Frag a, Frag b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0); }
};
template <> struct WsMma<f16_t> {
    using Frag = f16x8;
    static __device__ __forceinline__ f32x16 mma(Frag a, Frag b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0); }
};

// MFMA M index mu -> column of the wave's 32-column block it is fed from: registers 0..7 / 8..15 of a lane = columns 8h .. 8h+7 /
// 16 + 8h .. of its row (linear_pw.hip).  GEGLU: mu < 16 -> value column, mu >= 16 -> gate column of the same 16-column group
__device__ __forceinline__ int ws_perm(int mu) {
    const int a = mu >> 3, h = (mu >> 2) & 1, e = mu & 3;
    return 16 * (a >> 1) + 8 * h + 4 * (a & 1) + e;
}

// (buffer descriptors only in free functions: an opaque __amdgpu_buffer_rsrc_t inside a lambda trips hipcc's host pass)
// the LOADER wave (below) issues all 40 pieces of a slot: piece q = k-block q / 8, rows (q % 8) * 8 .. + 7; a lane fetches the 16-byte
// chunk that belongs at its (linear) LDS position under the XOR swizzle
__device__ __forceinline__ void ws_issue(const LinWSParams& p, char* slot_base, int row0, int lane, bool live) {
    const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.a), 0, p.a_bytes, 0x00020000);
    const int soff = row0 * (WS_K * 2);
#pragma unroll
    for (int q = 0; q < WS_SLOT / 1024; ++q) {
        const int lrow = (q & 7) * 8 + (lane >> 3);
        const unsigned chunk = (unsigned)((lane & 7) ^ ((lrow >> 1) & 7));
        const unsigned aoff = (unsigned)lrow * (WS_K * 2) + (unsigned)(q >> 3) * 128u + chunk * 16u;
        const unsigned off = (live && row0 + lrow < p.M) ? aoff : kWsOob;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, (__attribute__((address_space(3))) void*)(slot_base + q * 1024), 16, off, soff, 0, 0);
    }
}

template <typename T> __device__ __forceinline__ typename WsMma<T>::Frag ws_frag(const char* p) {
    return *reinterpret_cast<const typename WsMma<T>::Frag*>(p);
}

// every asm VMEM statement opens with `s_nop 4` (an SGPR of its descriptor may have just come back from a spill lane: linear_pw.hip) and
// a store closes with `s_nop 2` (its data registers must not be rewritten while it reads them)
template <bool NT> __device__ __forceinline__ void ws_store(const u32x4& rdst, const u32x4& d, unsigned o) {
    if constexpr (NT) asm volatile("s_nop 4\n\tbuffer_store_dwordx4 %0, %1, %2, 0 offen nt\n\ts_nop 2" ::"v"(d), "v"(o), "s"(rdst) : "memory");
    else asm volatile("s_nop 4\n\tbuffer_store_dwordx4 %0, %1, %2, 0 offen\n\ts_nop 2" ::"v"(d), "v"(o), "s"(rdst) : "memory");
}

template <typename T, int EPI, bool RES, bool NT>
__global__ __launch_bounds__(704) void linear_ws_kernel(const LinWSParams p) {
    using Frag = typename WsMma<T>::Frag;
    constexpr bool GEGLU = EPI == MVLDM_EPI_GEGLU;
    static_assert(!(GEGLU && RES), "no caller");
    constexpr int NST = GEGLU ? 1 : 2;             // 16-byte stores per lane and block
    constexpr int PF = 3;                          // activation fragments in flight ahead of the MFMA that uses them (register budget: 168)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int hi = lane >> 5, l31 = lane & 31;

    // workgroup -> (column slice, row group): the slices of one row group sit on ONE XCD (workgroup b runs on XCD b % 8) and walk the
    // same slots at the same time, so an activation line crosses the fabric once
    const int xcd = blockIdx.x & 7, lid = blockIdx.x >> 3;
    const int slice = lid % p.n_slices, rg = lid / p.n_slices;
    if (rg >= p.groups) return;
    const int g = xcd * p.groups + rg, G = 8 * p.groups;
    const int my_slots = g < p.n_slots ? (p.n_slots - g + G - 1) / G : 0;       // slots g, g + G, ...
    if (my_slots == 0) return;

    // ---- wave 10: the LOADER.  It owns the ring: all 40 LDS-DMA pieces of a slot and the counted waits for them.  VMEM returns in
    // order per wave, so a wave that both loads and stores can only keep as many stores in flight as its load prefetch distance allows
    // (first version: two slots = 80 KB per CU; at ~6 us from store to acknowledgement that is 3.3 TB/s, which is what it ran at while
    // moving the minimum number of bytes -- profiles/r04_pmc_linear.txt).  With the loads in a wave of their own the ten compute
    // waves never wait for a load, and their stores queue as deep as the hardware lets them.
    if (wave == WS_NW) {
        ws_issue(p, smem, g * WS_ROWS, lane, true);
        ws_issue(p, smem + WS_SLOT, (g + G) * WS_ROWS, lane, my_slots > 1);
        __builtin_amdgcn_s_waitcnt(ws_wait(0));
        __builtin_amdgcn_s_barrier();                                  // (1) slots 0 and 1 are in LDS (and the bias slab, written by the others)
        int fill = 2;                                                  // ring slot of t + 2
        for (int t = 0; t < my_slots; ++t) {
            if (t > 0) __builtin_amdgcn_s_barrier();                   // top of slot t: slot t - 1 has been read by every compute wave
            ws_issue(p, smem + fill * WS_SLOT, (g + (t + 2) * G) * WS_ROWS, lane, t + 2 < my_slots);
            fill = fill == 2 ? 0 : fill + 1;
            __builtin_amdgcn_s_waitcnt(ws_wait(WS_SLOT / 1024));       // slot t + 1 has landed: everything but the 40 newest pieces
        }
        return;
    }

    u32x4 rdst, rres;
    rdst[0] = (unsigned)(uintptr_t)p.dst; rdst[1] = (unsigned)((uintptr_t)p.dst >> 32) & 0xFFFFu; rdst[2] = p.dst_bytes; rdst[3] = 0x00020000u;
    rres[0] = (unsigned)(uintptr_t)p.residual; rres[1] = (unsigned)((uintptr_t)p.residual >> 32) & 0xFFFFu; rres[2] = p.res_bytes; rres[3] = 0x00020000u;

    // ---- the wave's weights: 32 packed columns x 320 -> 20 MFMA A-operand fragments (row = permuted column, 16-byte k-chunk 2 kk + hi)
    Frag wf[WS_KS];
    {
        int prow;                                              // packed weight row this lane feeds to M index l31
        if constexpr (GEGLU) {
            // the wave's 16 output columns: packed rows [value 16 | ... ] live in blocks of 32 value / 32 gate columns
            const int oc = wave * 16 + (ws_perm(l31) & 15);    // output column within the slice's 160
            const int gate = ws_perm(l31) >> 4;
            prow = slice * WS_BN + (oc >> 5) * 64 + gate * 32 + (oc & 31);
        } else {
            prow = slice * WS_BN + wave * 32 + ws_perm(l31);
        }
        const char* wp = reinterpret_cast<const char*>(p.w) + (size_t)prow * (WS_K * 2) + hi * 16;
        const bool ok = prow < p.n_pad;
#pragma unroll
        for (int kk = 0; kk < WS_KS; ++kk) {
            u32x4 v = u32x4{0u, 0u, 0u, 0u};
            if (ok) v = *reinterpret_cast<const u32x4*>(wp + kk * 32);
            wf[kk] = *reinterpret_cast<Frag*>(&v);
        }
    }
    // ---- bias of the slice's 320 packed columns -> LDS (torch-layout bias: GEGLU value columns first, gate columns at n_dst)
    if (tid < WS_BN / 4) {
        f32x4 b = f32x4{0.f, 0.f, 0.f, 0.f};
        const int pc = slice * WS_BN + 4 * tid;
        if (p.bias && pc < p.n_out) {
            int oc = pc;
            if constexpr (GEGLU) { const int blk = pc >> 5, w_ = pc & 31; oc = ((blk & 1) ? p.n_dst : 0) + (blk >> 1) * 32 + w_; }
            b = *reinterpret_cast<const f32x4*>(p.bias + oc);
        }
        *reinterpret_cast<f32x4*>(smem + WS_SLAB + tid * 16) = b;
    }

    // ---- per-lane addressing
    // fragment reads: row block i, k-step kk = 4 kb + q: kb * 8192 + (32 i + l31) * 128 + ((2 q + hi) ^ ((l31 >> 1) & 7)) * 16
    const int a_off = l31 * 128 + ((hi ^ ((l31 >> 1) & 7)) << 4);
    // epilogue: this lane's row inside a block = l31; first output column of its 8 (GEGLU) / 8 + 8 columns
    const int col0 = GEGLU ? slice * (WS_BN / 2) + wave * 16 + 8 * hi : slice * WS_BN + wave * 32 + 8 * hi;
    const float* slab = reinterpret_cast<const float*>(smem + WS_SLAB);

    // ---- (1): slots 0 and 1 and the bias slab are in LDS (the weight loads above are waited for by their first use)
    __builtin_amdgcn_s_waitcnt(0xC07F);               // lgkmcnt(0): the slab write
    __builtin_amdgcn_s_barrier();

    f32x16 acc0, acc1;
    u32x4 res[2];                                     // residual chunks of the block whose epilogue comes next
    res[0] = res[1] = u32x4{0u, 0u, 0u, 0u};
    unsigned rowoff_prev = kWsOob;                    // dst byte offset of this lane's row in the previous block

// bias -> accumulator (registers 4a + e of half-wave h = packed column 16 (a >> 1) + 8 h + 4 (a & 1) + e of the wave's 32)
#define WS_INIT(acc_)                                                                                              \
    _Pragma("unroll") for (int a = 0; a < 4; ++a) {                                                                \
        const int pcw = 16 * (a >> 1) + 8 * hi + 4 * (a & 1);                                                      \
        int sidx;                                                                                                  \
        if constexpr (GEGLU) { const int oc = wave * 16 + (pcw & 15); sidx = (oc >> 5) * 64 + (pcw >> 4) * 32 + (oc & 31); }   \
        else sidx = wave * 32 + pcw;                                                                               \
        const f32x4 b = *reinterpret_cast<const f32x4*>(slab + sidx);                                              \
        _Pragma("unroll") for (int e = 0; e < 4; ++e) acc_[4 * a + e] = b[e];                                      \
    }
// micro-step m = 0..7 of the previous block's epilogue, one per MFMA of the current block (as one lump every wave of the SIMD did it
// at the same moment -- the waves run in lockstep behind the slot barrier -- and the matrix pipe idled: the GELU of the FF projection cost
// 300 us of 1350 although it runs on another pipe).  GEGLU: output m = value m x GELU(gate m); otherwise outputs 2m, 2m+1 (+ residual)
#define WS_MICRO(acc_, m_)                                                                                         \
    {                                                                                                              \
        if constexpr (GEGLU) {                                                                                     \
            oc0_.set(m_, acc_[m_] * WS_GELU(acc_[8 + (m_)]) * p.out_scale);                                        \
        } else {                                                                                                   \
            _Pragma("unroll") for (int e = 2 * (m_); e < 2 * (m_) + 2; ++e) {                                      \
                float v_ = acc_[e] * p.out_scale;                                                                  \
                if constexpr (RES) { Chunk<T> rc_; rc_.raw = res[e >> 3]; v_ += rc_.get(e & 7); }                  \
                if (e < 8) oc0_.set(e, v_); else oc1_.set(e - 8, v_);                                              \
            }                                                                                                      \
        }                                                                                                          \
    }
// One 32-row block: 20 MFMAs on acc_cur_ (fragments fetched PF steps ahead: with one, every MFMA of a wave waited for an LDS round trip -- 2.3x the matrix-pipe time per slot),
// and -- between them -- the epilogue of the PREVIOUS block (acc_prev_) and, with a residual, the residual request of THIS one (one
// register set: block b's request goes out after block b-1's has been consumed).  VMEM issue order of a compute wave:
//     ... L(b-1) 2 | S(b-2) NST | L(b) 2 | S(b-1) NST ...       (L only with a residual; the ring refills are the loader wave's)
// the only counted wait: L(b-1) in block b leaves S(b-2), the NST operations younger than it, in flight; without a residual a compute
// wave never waits for memory at all
#define WS_BLOCK(acc_cur_, acc_prev_, blk_base_, rowoff_new_, rowres_new_)                                         \
    {                                                                                                              \
        Frag fq_[PF + 1];                                                                                          \
        _Pragma("unroll") for (int q = 0; q < PF; ++q) fq_[q] = ws_frag<T>(blk_base_ + (q >> 2) * 8192 + (a_off ^ ((q & 3) << 5)));   \
        Chunk<T> oc0_, oc1_;                                                                                       \
        oc0_.zero(); oc1_.zero();                                                                                  \
        _Pragma("unroll") for (int kk = 0; kk < WS_KS; ++kk) {                                                     \
            if (kk + PF < WS_KS) fq_[(kk + PF) % (PF + 1)] = ws_frag<T>(blk_base_ + ((kk + PF) >> 2) * 8192 + (a_off ^ (((kk + PF) & 3) << 5)));   \
            __builtin_amdgcn_sched_barrier(0);                                                                     \
            acc_cur_ = WsMma<T>::mma(wf[kk], fq_[kk % (PF + 1)], acc_cur_);                                        \
            if (kk == 4) {                                                                                         \
                if constexpr (RES) asm volatile("s_waitcnt vmcnt(%2)" : "+v"(res[0]), "+v"(res[1]) : "n"(NST));   \
            }                                                                                                      \
            if (kk >= 4 && kk < 12) WS_MICRO(acc_prev_, kk - 4)                                                    \
            if (RES && kk == 12) {                                                                                 \
                const unsigned o0_ = rowres_new_ != kWsOob ? rowres_new_ + (unsigned)col0 * 2u : kWsOob;           \
                asm volatile("s_nop 4\n\tbuffer_load_dwordx4 %0, %2, %4, 0 offen\n\tbuffer_load_dwordx4 %1, %3, %4, 0 offen"                 \
                             : "=&v"(res[0]), "=&v"(res[1]) : "v"(o0_), "v"(o0_ == kWsOob ? kWsOob : o0_ + 32u), "s"(rres));                   \
            }                                                                                                      \
            if (kk == 13) {                                                                                        \
                ws_store<NT>(rdst, oc0_.raw, rowoff_prev == kWsOob ? kWsOob : rowoff_prev + (unsigned)col0 * 2u);  \
                if constexpr (NST == 2) ws_store<NT>(rdst, oc1_.raw, rowoff_prev == kWsOob ? kWsOob : rowoff_prev + (unsigned)col0 * 2u + 32u);   \
            }                                                                                                      \
        }                                                                                                          \
        __builtin_amdgcn_sched_barrier(0);                                                                         \
        rowoff_prev = rowoff_new_;                                                                                 \
    }

    int rsl = 0;                                      // ring slot of the current slot
    for (int t = 0; t < my_slots; ++t) {
        const int row0 = (g + t * G) * WS_ROWS;
        if (t > 0) {
            __builtin_amdgcn_s_waitcnt(0xC07F);                        // lgkmcnt(0): this wave's reads of the slot about to be refilled
            __builtin_amdgcn_s_barrier();                              // the loader arrives here once slot t has landed
        }
        const char* sb = smem + rsl * WS_SLOT;
        {
            const int m = row0 + l31;
            const unsigned ro = m < p.M ? (unsigned)m * (unsigned)p.dst_ld * 2u : kWsOob;
            const unsigned rr = m < p.M ? (unsigned)m * (unsigned)p.n_dst * 2u : kWsOob;
            WS_INIT(acc0)
            WS_BLOCK(acc0, acc1, sb, ro, rr)
        }
        {
            const int m = row0 + 32 + l31;
            const unsigned ro = m < p.M ? (unsigned)m * (unsigned)p.dst_ld * 2u : kWsOob;
            const unsigned rr = m < p.M ? (unsigned)m * (unsigned)p.n_dst * 2u : kWsOob;
            WS_INIT(acc1)
            WS_BLOCK(acc1, acc0, sb + 4096, ro, rr)
        }
        rsl = rsl == 2 ? 0 : rsl + 1;
    }
    // ---- drain: the last block's epilogue
    {
        Chunk<T> oc0_, oc1_;
        oc0_.zero(); oc1_.zero();
        if constexpr (RES) asm volatile("s_waitcnt vmcnt(0)" : "+v"(res[0]), "+v"(res[1]));
#pragma unroll
        for (int m = 0; m < 8; ++m) WS_MICRO(acc1, m)
        ws_store<NT>(rdst, oc0_.raw, rowoff_prev == kWsOob ? kWsOob : rowoff_prev + (unsigned)col0 * 2u);
        if constexpr (NST == 2) ws_store<NT>(rdst, oc1_.raw, rowoff_prev == kWsOob ? kWsOob : rowoff_prev + (unsigned)col0 * 2u + 32u);
    }
#undef WS_INIT
#undef WS_MICRO
#undef WS_BLOCK
}

bool linear_ws_applicable(const mvldm_igemm_desc& d) {
    if (d.act_dtype == MVLDM_F32 || d.dst_dtype != d.act_dtype) return false;
    if (d.ksize != 1 || d.stride != 1 || d.upsample != 0 || d.row_bias || d.k_order != 1 || d.splitk > 1) return false;
    if (d.h_in != d.h_out || d.w_in != d.w_out || d.pad != 0) return false;
    if (d.epilogue != MVLDM_EPI_NONE && d.epilogue != MVLDM_EPI_GEGLU) return false;
    if (d.epilogue != MVLDM_EPI_NONE && d.residual) return false;
    if (d.src1 || d.c1 || d.c0 != WS_K || d.k_pad != WS_K || d.n_pad % WS_BN || d.n_out != d.n_pad) return false;
    const int n_dst = d.epilogue == MVLDM_EPI_GEGLU ? d.n_out / 2 : d.n_out;
    const int dst_ld = d.dst_ld > 0 ? d.dst_ld : n_dst;
    if (n_dst % 8 || dst_ld % 8 || dst_ld < n_dst) return false;
    if (d.bias && ((uintptr_t)d.bias % 16)) return false;
    if (((uintptr_t)d.dst % 16) || ((uintptr_t)d.weight % 16) || (d.residual && ((uintptr_t)d.residual % 16))) return false;
    const double m = (double)d.n_img * d.h_out * d.w_out;
    return m * WS_K * 2.0 < 4.0e9 && m * dst_ld * 2.0 < 4.0e9 && m * n_dst * 2.0 < 4.0e9;
}

template <typename T, int EPI, bool RES, bool NT> static int linear_ws_launch1(const LinWSParams& p, int grid, hipStream_t s) {
    static std::atomic<uint64_t> done{0};
    if (int rc0 = ensure_dyn_smem(reinterpret_cast<const void*>(linear_ws_kernel<T, EPI, RES, NT>), WS_SMEM, done)) return rc0;
    hipLaunchKernelGGL((linear_ws_kernel<T, EPI, RES, NT>), dim3(grid), dim3(704), WS_SMEM, s, p);
    return check_launch();
}
template <typename T, int EPI, bool RES> static int linear_ws_launch(const LinWSParams& p, int grid, hipStream_t s) {
    return p.nt_store ? linear_ws_launch1<T, EPI, RES, true>(p, grid, s) : linear_ws_launch1<T, EPI, RES, false>(p, grid, s);
}

int linear_ws_run(const mvldm_igemm_desc& d, hipStream_t s) {
    MVLDM_REQUIRE(linear_ws_applicable(d), "igemm: tile 14 (weight-stationary Linear, K = 320) does not apply to this problem");
    LinWSParams p;
    p.a = d.src0; p.w = d.weight; p.bias = d.bias; p.residual = d.residual; p.dst = d.dst;
    p.M = d.n_img * d.h_out * d.w_out; p.n_out = d.n_out; p.n_pad = d.n_pad;
    p.n_dst = d.epilogue == MVLDM_EPI_GEGLU ? d.n_out / 2 : d.n_out;
    p.dst_ld = d.dst_ld > 0 ? d.dst_ld : p.n_dst;
    p.out_scale = d.out_scale;
    p.n_slices = d.n_pad / WS_BN;
    p.n_slots = (p.M + WS_ROWS - 1) / WS_ROWS;
    static int n_cu = 0;
    if (!n_cu) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0)
            n_cu = prop.multiProcessorCount;
        else
            n_cu = 256;
    }
    const int cu_x = std::max(1, n_cu / 8);
    MVLDM_REQUIRE(p.n_slices <= cu_x, "igemm: tile 14: %d column slices exceed the %d CUs of an XCD", p.n_slices, cu_x);
    p.groups = std::max(1, std::min(cu_x / p.n_slices, (p.n_slots + 7) / 8));
    p.a_bytes = (unsigned)((double)p.M * WS_K * 2.0); p.w_bytes = (unsigned)((double)d.n_pad * WS_K * 2.0);
    p.bias_bytes = d.bias ? (unsigned)d.n_out * 4u : 0u;
    p.res_bytes = d.residual ? (unsigned)((double)p.M * p.n_dst * 2.0) : 0u;
    p.dst_bytes = (unsigned)((double)p.M * p.dst_ld * 2.0);
    static const int kNt = knob_int("MVLDM_STREAM_STORES", 0);
    p.nt_store = kNt == 1;
    if (kWsFake & 1) p.a_bytes = 0;
    if (kWsFake & 4) p.dst_bytes = 0;
    const int grid = 8 * p.groups * p.n_slices;
    const bool res = d.residual != nullptr;
    return dispatch_dtype(d.act_dtype, [&](auto t) -> int {
        using T = decltype(t);
        if constexpr (sizeof(T) == 2) {
            if (d.epilogue == MVLDM_EPI_GEGLU) return linear_ws_launch<T, MVLDM_EPI_GEGLU, false>(p, grid, s);
            return res ? linear_ws_launch<T, MVLDM_EPI_NONE, true>(p, grid, s) : linear_ws_launch<T, MVLDM_EPI_NONE, false>(p, grid, s);
        } else {
            return set_error(MVLDM_ERR_ARG, "igemm: tile 14 needs a 16-bit activation type");
        }
    });
}

}  // namespace mvldm
