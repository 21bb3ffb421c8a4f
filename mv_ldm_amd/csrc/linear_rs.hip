// Persistent REGISTER-STAGED Linear (1x1 conv over token rows): tile 19 of the implicit-GEMM family (include/mvldm.h: mvldm_igemm_fwd;
// 16-bit activations, one source or the channel concat of two, K a multiple of 128 and >= 256).  Round 6.
//
// Why another Linear kernel.  On the K >= 1280 Linears of the 8 x 8 level (36 864 rows at 64 scenes: QKV, to_out, FF) tiles 10 / 13 run
// 0.74 - 1.05 PFLOP/s where a plain vendor GEMM reaches 1.25 - 1.37 (profiles/r05_linear_vs_hipblaslt.jsonl).  tools/pw_probe.py had
// already said why: with the operand traffic switched off the same loop runs 1.8 PFLOP/s -- the loop is sound, it WAITS.  An LDS-DMA
// ring lives in LDS only: two 72 KB slots are all 160 KB hold, so a K-step is requested ONE step (about 1.1 us of MFMA work) before it
// is needed, and a step whose activation rows come from HBM / the Infinity Cache takes longer than that to land.  The register file
// is three times the LDS (512 KB per CU).  This kernel keeps the in-flight K-steps THERE:
//   * 4 waves of 128 x 128 (2 x 2) instead of 8 of 64 x 160: one wave per SIMD owns all 512 registers of a lane -- 256 accumulators
//     (the AGPR half) + 256 VGPRs, of which 128 hold TWO K-steps of operands in flight (16 + 16 buffer_load_dwordx4 per lane);
//   * a step's registers are written to a 2-slot LDS ring (ds_write_b128, the XOR-swizzled 128-byte rows of igemm.hip) one step
//     before its MFMAs and re-issued at once for the step three ahead: global -> register lead 1.75 steps, + one step in LDS, against
//     1.0 for the DMA ring; the 128 x 128 wave tile also halves the LDS fragment traffic per MFMA (8 ds_read_b128 per 16 MFMAs);
//   * persistent like tile 13: a workgroup walks its 256 x 256 tiles as ONE stream of K-steps, the loads run ahead across tile
//     boundaries (the next tile's first three steps land during the epilogue), one s_barrier per step placed in front of the step's
//     last quarter (every fragment of the step is in registers by then: the barrier publishes the next slot and retires this one);
//   * park-free epilogue from the permuted-column transposed product (tile 13's: a lane holds 8 + 8 consecutive columns of one row);
//   * every VMEM instruction is a compiler builtin (no LDS-DMA, no inline-asm loads): hipcc's own wait-count pass sees one in-order
//     queue and emits the counted vmcnt in front of each ds_write -- nothing is counted by hand here.
// K-steps per tile must be even (the two register sets / LDS slots alternate with the step parity, the loop is unrolled by two).
#include <algorithm>
#include <cstdlib>

#include "common.h"

namespace mvldm {

struct LinRSParams {
    const void* a; const void* a1; const void* w; const float* bias; const void* residual; void* dst;
    int M, K, c0, c1, kt0, n_out, n_pad, n_dst, dst_ld, k_steps;    // K = c0 + c1; K-steps [0, kt0) come from `a`, the rest from `a1`
    int tiles_m, tiles_n, m_per;       // m_per: 256-row blocks per XCD
    int gm, gn, nbn, wgx;              // an XCD's wgx workgroups walk its tiles in gm x gn blocks, column chunks (nbn of them) fastest
    float out_scale;
    unsigned a_bytes, a1_bytes, w_bytes, bias_bytes, res_bytes, dst_bytes;
};

#ifdef MVLDM_EXPERIMENTS
static const int kRsFake = knob_int("MVLDM_RS_FAKE", 0);   // wrong results by design (tools/rs_probe.py): 1 no A traffic, 2 no W traffic, 4 no stores, 8 no residual loads
#else
static constexpr int kRsFake = 0;
#endif

constexpr unsigned kRsOob = 0xFFFFFFF0u;
constexpr unsigned kRsRowNone = 0xFFFFFFFFu;
constexpr int RS_BM = 256, RS_BN = 256, RS_NW = 4;
constexpr int RS_A_SLOT = RS_BM * 128, RS_W_SLOT = RS_BN * 128, RS_STAGE = RS_A_SLOT + RS_W_SLOT;
constexpr int RS_RING = 2 * RS_STAGE;                 // 128 KB
constexpr int RS_SLAB = RS_RING;                      // 2 x 1 KB: bias of a tile's 256 packed columns, double-buffered over tiles
constexpr int RS_SMEM = RS_RING + 2048;
constexpr int RS_PIECES = 8;                          // 1 KiB pieces (8 rows of 128 bytes) per wave, operand and step

template <typename T> struct RsMma;
template <> struct RsMma<bf16_t> {
    using Frag = bf16x8;
    static __device__ __forceinline__ f32x16 mma(Frag a, Frag b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0); }
};
template <> struct RsMma<f16_t> {
    using Frag = f16x8;
    static __device__ __forceinline__ f32x16 mma(Frag a, Frag b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0); }
};

// MFMA M index mu (= lane & 31 of the W-fragment read) -> column of the 32-column block it is fed from (linear_pw.hip: registers 0..7 /
// 8..15 of a lane become 8 + 8 CONSECUTIVE output columns; each 16-lane group reads the same set of rows as the identity)
__device__ __forceinline__ int rs_perm(int mu) {
    const int a = mu >> 3, h = (mu >> 2) & 1, e = mu & 3;
    return 16 * (a >> 1) + 8 * h + 4 * (a & 1) + e;
}

// Source addressing of a wave's pieces.  Piece `it` of a wave covers tile rows (wave + 4 it) * 8 .. + 7; a lane fetches chunk (lane & 7) of
// row (lane >> 3) of the piece and writes it to the swizzled LDS position -- (row >> 1) & 7 does not depend on `it`, so ONE per-lane
// byte offset serves all pieces of an operand on either side; the piece's 32-row stride rides in the scalar offset with the K position.
struct RsAddr {
    unsigned a0, a1;     // activation rows of piece 0: byte offset into the first / second source (channel concat)
    unsigned w;          // weight rows of piece 0
    int m_lane;          // global row this lane reads in piece 0 (piece it: + 32 it; rows >= M read zeros)
    int n0;              // first packed column of the tile
    bool valid;
};

__device__ __forceinline__ void rs_offsets(const LinRSParams& p, bool valid, int tm, int tn, int wave, int lane, RsAddr& ad) {
    const int chunk = lane & 7, row = wave * 8 + (lane >> 3);
    const int m = tm * RS_BM + row, n = tn * RS_BN + row;
    ad.a0 = ((unsigned)m * (unsigned)p.c0 + (unsigned)chunk * 8u) * 2u;
    ad.a1 = ((unsigned)m * (unsigned)p.c1 + (unsigned)chunk * 8u) * 2u;
    ad.w = ((unsigned)n * (unsigned)p.K + (unsigned)chunk * 8u) * 2u;
    ad.m_lane = m;
    ad.n0 = tn * RS_BN;
    ad.valid = valid;
}

// (buffer descriptors only in free functions: an opaque __amdgpu_buffer_rsrc_t inside a lambda trips hipcc's host pass)
// piece `it` of K-step ks: activation rows -> r[it]   (`it` is a constant after unrolling at every call site)
__device__ __forceinline__ void rs_load_a1(const LinRSParams& p, int ks, const RsAddr& ad, u32x4 (&r)[RS_PIECES], int it) {
    const bool second = ks >= p.kt0;
    // (ONE descriptor from selected scalars: a select between two descriptors becomes a branch)
    const void* abase = second ? p.a1 : p.a;
    const unsigned abytes = second ? p.a1_bytes : p.a_bytes;
    const int c = second ? p.c1 : p.c0;
    const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(abase), 0, abytes, 0x00020000);
    const int soff = (second ? ks - p.kt0 : ks) * 128;
    const unsigned v = second ? ad.a1 : ad.a0;
    const unsigned off = (ad.valid && ad.m_lane + 32 * it < p.M) ? v : kRsOob;
    r[it] = __builtin_amdgcn_raw_buffer_load_b128(ra, off, soff + it * 64 * c, 0);
}
__device__ __forceinline__ void rs_load_w1(const LinRSParams& p, int ks, const RsAddr& ad, u32x4 (&r)[RS_PIECES], int it) {
    const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.w), 0, p.w_bytes, 0x00020000);
    // (n_pad is a multiple of 64, a piece's 8 rows start at a multiple of 8 inside a 32-row group: inside the packed weight or outside as a whole)
    const unsigned off = (ad.valid && ad.n0 + 32 * it < p.n_pad) ? ad.w : kRsOob;
    r[it] = __builtin_amdgcn_raw_buffer_load_b128(rw, off, ks * 128 + it * 64 * p.K, 0);
}
// the same piece -> its (swizzled) place in an operand's half of a ring slot
__device__ __forceinline__ void rs_write1(char* half_slot, int wr_off, const u32x4 (&r)[RS_PIECES], int it) {
    *reinterpret_cast<u32x4*>(half_slot + wr_off + it * 4096) = r[it];
}
template <int J2> __device__ __forceinline__ void rs_load_a2(const LinRSParams& p, int ks, const RsAddr& ad, u32x4 (&r)[RS_PIECES]) {
    rs_load_a1(p, ks, ad, r, J2); rs_load_a1(p, ks, ad, r, J2 + 1);
}
template <int J2> __device__ __forceinline__ void rs_load_w2(const LinRSParams& p, int ks, const RsAddr& ad, u32x4 (&r)[RS_PIECES]) {
    rs_load_w1(p, ks, ad, r, J2); rs_load_w1(p, ks, ad, r, J2 + 1);
}
template <int J2> __device__ __forceinline__ void rs_write2(char* half_slot, int wr_off, const u32x4 (&r)[RS_PIECES]) {
    rs_write1(half_slot, wr_off, r, J2); rs_write1(half_slot, wr_off, r, J2 + 1);
}

// byte offset (into the torch-layout bias) of packed columns 4t .. 4t+3 of tile column tn, out of range past the tile / the layer
__device__ __forceinline__ u32x4 rs_load_bias(const LinRSParams& p, bool geglu, bool valid, int tn, int t) {
    const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.bias), 0, p.bias_bytes, 0x00020000);
    const int pc = tn * RS_BN + 4 * t;                          // packed column
    int oc = pc;                                                // column of the torch-layout bias
    if (geglu) {
        const int blk = pc >> 5, w = pc & 31;
        oc = ((blk & 1) ? p.n_dst : 0) + (blk >> 1) * 32 + w;
    }
    const unsigned off = (valid && 4 * t < RS_BN && pc < p.n_out) ? (unsigned)oc * 4u : kRsOob;
    return __builtin_amdgcn_raw_buffer_load_b128(rb, off, 0, 0);
}

__device__ __forceinline__ u32x4 rs_load_res(const LinRSParams& p, unsigned off) {
    const __amdgpu_buffer_rsrc_t rr = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.residual), 0, p.res_bytes, 0x00020000);
    return __builtin_amdgcn_raw_buffer_load_b128(rr, off, 0, 0);
}
// (write-back stores, not streaming ones: linear_pw.hip -- the L2 acknowledges a tile's store burst long before HBM has taken it)
__device__ __forceinline__ void rs_store(const LinRSParams& p, const u32x4& v, unsigned off) {
    const __amdgpu_buffer_rsrc_t rd = __builtin_amdgcn_make_buffer_rsrc(p.dst, 0, p.dst_bytes, 0x00020000);
    __builtin_amdgcn_raw_buffer_store_b128(v, rd, off, 0, 0);
}

template <typename T> __device__ __forceinline__ typename RsMma<T>::Frag rs_frag(const char* p) {
    return *reinterpret_cast<const typename RsMma<T>::Frag*>(p);
}

__device__ __forceinline__ unsigned rs_off(unsigned row, int col, int n_dst) {
    return (row != kRsRowNone && col < n_dst) ? row + (unsigned)col * 2u : kRsOob;
}

// Walks the tiles of a workgroup (all wave-uniform): linear_pw.hip's order.  XCD x owns row blocks [x * m_per, (x+1) * m_per) and all
// column tiles; its tiles form one list in gm x gn block order (column chunks fastest, the row fastest inside a block, ragged edge
// blocks packed densely) and its wgx workgroups take entries lid, lid + wgx, ...
struct RsTileIter {
    int r, tm, tn;
    bool valid;
    __device__ __forceinline__ void set(const LinRSParams& p, int r0, int lid, int m_lo, int m_cnt) {
        r = r0;
        const int i = r0 * p.wgx + lid;
        valid = i < m_cnt * p.tiles_n;
        if (valid) {
            const int strip = p.gm * p.tiles_n;
            const int sm = min(i / strip, (m_cnt + p.gm - 1) / p.gm - 1);
            const int hm = min(p.gm, m_cnt - sm * p.gm);
            const int is = i - sm * strip;
            const int cn = is / (hm * p.gn);
            const int j = is - cn * hm * p.gn;
            const int ln = j / hm;
            tm = m_lo + sm * p.gm + (j - ln * hm);
            tn = cn * p.gn + ln;
        }
    }
};

// one 32 x 32 block (GEGLU: one value / gate pair) of the finished tile -> two packed 16-byte chunks.  c[k]: the lane's 16 columns
// in output order (registers 0..7 = columns 8h .. 8h+7, 8..15 = 16 + 8h .. of the block).  RES: residual chunks of the same columns
template <typename T, bool RES>
__device__ __forceinline__ void rs_pack(const float (&c)[16], float scale, const u32x4 (&res)[2], u32x4 (&out)[2]) {
#pragma unroll
    for (int g = 0; g < 2; ++g) {
        Chunk<T> oc;
        if constexpr (RES) {
            Chunk<T> rc;
            rc.raw = res[g];
#pragma unroll
            for (int e = 0; e < 8; ++e) oc.set(e, c[8 * g + e] * scale + rc.get(e));
        } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) oc.set(e, c[8 * g + e] * scale);
        }
        out[g] = oc.raw;
    }
}

template <typename T, int EPI, bool RES>
__global__ __launch_bounds__(256) void linear_rs_kernel(const LinRSParams p) {
    using M_ = RsMma<T>;
    using Frag = typename M_::Frag;
    constexpr bool GEGLU = EPI == MVLDM_EPI_GEGLU;
    static_assert(!(GEGLU && RES), "no caller");
    constexpr int NOUT = GEGLU ? 2 : 4;            // output blocks per row block of a wave
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;        // 2 x 2 waves of 128 rows x 128 columns
    const int hi = lane >> 5, l31 = lane & 31;
    const int prm = rs_perm(l31);

    const int xcd = blockIdx.x & 7, lid = blockIdx.x >> 3;
    const int m_lo = xcd * p.m_per, m_cnt = min(p.tiles_m, m_lo + p.m_per) - m_lo;
    RsTileIter cur, nxt, nx2, iss;               // compute side, the two tiles after it, issue side (tile of the newest K-step in flight)
    cur.set(p, 0, lid, m_lo, m_cnt);
    if (!cur.valid) return;
    nxt.set(p, cur.r + 1, lid, m_lo, m_cnt);
    nx2.set(p, nxt.r + 1, lid, m_lo, m_cnt);
    iss = cur;

    // fragment read offsets inside a ring slot (bytes): row * 128 + swizzled chunk of k-sub-step 0; sub-step kk: XOR kk << 5
    const int a_off = (wm * 128 + l31) * 128 + ((hi ^ ((l31 >> 1) & 7)) << 4);
    const int w_off = RS_A_SLOT + (wn * 128 + prm) * 128 + ((hi ^ ((prm >> 1) & 7)) << 4);
    // write offset of this lane's chunk of piece 0 inside an operand's half of a slot
    const int wr_row = wave * 8 + (lane >> 3);
    const int wr_off = wr_row * 128 + (((lane & 7) ^ ((wr_row >> 1) & 7)) << 4);

    f32x16 acc[4][4];
    RsAddr ad;
    u32x4 ra0[RS_PIECES], ra1[RS_PIECES], rw[RS_PIECES];      // in flight: activation rows of the even / odd K-steps, weight rows of the next step
    Frag fa0[4], fw0[4], fa1[4], fw1[4];
    const int kT = p.k_steps;
    const f32x16 kZero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    int ks_i = 2;                                // issue side: K-step (inside tile `iss`) of the newest step in flight
    int par = 0;                                 // bias slab of the current tile
    // Bias of the NEXT tile: requested a whole tile early (prologue / the previous tile's epilogue) so that its use inside the K loop
    // (parked in the other slab in the tile's second-to-last step, published by that step's barrier) never waits for a young load.
    u32x4 bn;

    // ---- prologue: steps 0, 1, 2 of the first tile; step 0 goes to ring slot 0; the first bias slab ----
    {
        rs_offsets(p, true, cur.tm, cur.tn, wave, lane, ad);
        rs_load_a2<0>(p, 0, ad, ra0); rs_load_a2<2>(p, 0, ad, ra0); rs_load_a2<4>(p, 0, ad, ra0); rs_load_a2<6>(p, 0, ad, ra0);
        rs_load_w2<0>(p, 0, ad, rw); rs_load_w2<2>(p, 0, ad, rw); rs_load_w2<4>(p, 0, ad, rw); rs_load_w2<6>(p, 0, ad, rw);
        rs_load_a2<0>(p, 1, ad, ra1); rs_load_a2<2>(p, 1, ad, ra1); rs_load_a2<4>(p, 1, ad, ra1); rs_load_a2<6>(p, 1, ad, ra1);
        const u32x4 b = rs_load_bias(p, GEGLU, true, cur.tn, tid);
        bn = rs_load_bias(p, GEGLU, nxt.valid, nxt.tn, tid);
        rs_write2<0>(smem, wr_off, ra0); rs_write2<2>(smem, wr_off, ra0); rs_write2<4>(smem, wr_off, ra0); rs_write2<6>(smem, wr_off, ra0);
        rs_write2<0>(smem + RS_A_SLOT, wr_off, rw); rs_write2<2>(smem + RS_A_SLOT, wr_off, rw);
        rs_write2<4>(smem + RS_A_SLOT, wr_off, rw); rs_write2<6>(smem + RS_A_SLOT, wr_off, rw);
        // (order: the weights of step 1 are OLDER than the activation rows of step 2 -- header of RS_STEP)
        rs_load_w2<0>(p, 1, ad, rw); rs_load_w2<2>(p, 1, ad, rw); rs_load_w2<4>(p, 1, ad, rw); rs_load_w2<6>(p, 1, ad, rw);
        rs_load_a2<0>(p, 2, ad, ra0); rs_load_a2<2>(p, 2, ad, ra0); rs_load_a2<4>(p, 2, ad, ra0); rs_load_a2<6>(p, 2, ad, ra0);
        if (4 * tid < RS_BN) *reinterpret_cast<u32x4*>(smem + RS_SLAB + tid * 16) = b;
        __builtin_amdgcn_s_waitcnt(0xC07F);      // lgkmcnt(0): this wave's LDS writes are done
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            fa0[j] = rs_frag<T>(smem + a_off + j * 4096);
            fw0[j] = rs_frag<T>(smem + w_off + j * 4096);
        }
    }

// One sub-step (16 of a step's 64 K values) = 16 MFMA SLOTS.  A slot is one MFMA followed by at most three memory instructions that
// issue while it executes (32 cycles on the SIMD's matrix pipe; MI355X_MICROARCH.md: <= 5 single-issue instructions hide in that gap):
// slots 0-7 fetch the NEXT sub-step's fragments (activation fragments first: the next sub-step's first slots need all four), and every
// slot takes its share OPS_ of the step's staging work (register set -> LDS, or the re-issue of a set).  The first version of this
// kernel issued the MFMAs in groups of four with eight memory instructions between the groups: PMC (tools/gemm_diag.sh) showed the
// wave stalled for the full 2048 MFMA cycles of a step AND issuing for another 1550 -- nothing overlapped; the vendor's stream
// interleaves one to one.
// (Z_: the tile's first sub-step starts the accumulators from the MFMA's zero C operand -- they are DEFINED there, not carried round the
//  tile loop: carried, hipcc gave them different AGPRs in the epilogue and in the loop and shuffled 256 registers per tile)
#define RS_SUB(ca_, cw_, na_, nw_, nslot_, nkk_, Z_, OPS_)                                                        \
    {                                                                                                             \
        const char* st_ = smem + (nslot_) * RS_STAGE;                                                             \
        const int ao_ = a_off ^ ((nkk_) << 5), wo_ = w_off ^ ((nkk_) << 5);                                       \
        _Pragma("unroll") for (int q = 0; q < 16; ++q) {                                                          \
            acc[q & 3][q >> 2] = M_::mma(cw_[q >> 2], ca_[q & 3], (Z_) ? kZero : acc[q & 3][q >> 2]);             \
            if (q < 4) na_[q] = rs_frag<T>(st_ + ao_ + q * 4096);                                                 \
            else if (q < 8) nw_[q - 4] = rs_frag<T>(st_ + wo_ + (q - 4) * 4096);                                  \
            OPS_                                                                                                  \
            __builtin_amdgcn_sched_barrier(0);                                                                    \
        }                                                                                                         \
    }
// One K-step computed from ring slot P_.  On entry: slot P_ holds the step; `rw` the weight rows of the next step, activation set !P_ (ran_)
// its activation rows (landed or landing), set P_ those of the step after; f?0 hold the step's kk = 0 fragments.  Sub-steps 0 / 1 move the
// next step into slot !P_ (free since the previous step's barrier): weight rows first, so that `rw` is re-issued (for the step two ahead,
// sub-step 1) BEFORE activation set !P_ is (for the step three ahead, sub-step 2) -- VMEM returns in order, and the wait for the weights
// of the next step must leave the younger activation loads in flight.  The weights come from the XCD's L2 (0.75 steps of lead), the
// activation rows from HBM / the Infinity Cache (1.75).  The barrier in front of sub-step 3 publishes slot !P_ and retires slot P_.
#define RS_STEP(P_, ran_, Z_)                                                                                     \
    {                                                                                                             \
        RS_SUB(fa0, fw0, fa1, fw1, P_, 1, Z_,                                                                     \
               if (q >= 8) rs_write1(smem + (1 - (P_)) * RS_STAGE + RS_A_SLOT, wr_off, rw, q - 8);)               \
        /* (the issue cursor still names the step two ahead here: it moves on in sub-step 2) */                   \
        RS_SUB(fa1, fw1, fa0, fw0, P_, 2, false,                                                                  \
               if (q < 8) rs_write1(smem + (1 - (P_)) * RS_STAGE, wr_off, ran_, q);                               \
               else rs_load_w1(p, ks_i, ad, rw, q - 8);)                                                          \
        RS_SUB(fa0, fw0, fa1, fw1, P_, 3, false,                                                                  \
               if (q == 8) {                                                                                      \
                   if (++ks_i == kT) {                                                                            \
                       ks_i = 0;                                                                                  \
                       iss.set(p, iss.r + 1, lid, m_lo, m_cnt);                                        \
                       rs_offsets(p, iss.valid, iss.tm, iss.tn, wave, lane, ad);                                  \
                   }                                                                                              \
               }                                                                                                  \
               if (q >= 8) rs_load_a1(p, ks_i, ad, ran_, q - 8);)                                                 \
        __builtin_amdgcn_s_waitcnt(0xC07F);                                                                       \
        __builtin_amdgcn_s_barrier();                                                                             \
        asm volatile("" ::: "memory");                                                                            \
        RS_SUB(fa1, fw1, fa0, fw0, 1 - (P_), 0, false, )                                                          \
    }

    for (; cur.valid; cur = nxt, nxt = nx2, nx2.set(p, nx2.r + 1, lid, m_lo, m_cnt)) {
        // ---- the tile's first K-step starts the accumulators from zero; the bias (slab `par`: written by the prologue / in the previous
        //      tile's second-to-last step, published by a step barrier since) is added in the epilogue ----
        RS_STEP(0, ra1, true)
        RS_STEP(1, ra0, false)
        int ks = 2;
#pragma unroll 1
        do {
            // second-to-last step: the next tile's bias goes to the other slab (published by the barriers of the last two steps)
            if (ks == kT - 2) {
                if (4 * tid < RS_BN) *reinterpret_cast<u32x4*>(smem + RS_SLAB + (par ^ 1) * 1024 + tid * 16) = bn;
            }
            RS_STEP(0, ra1, false)
            RS_STEP(1, ra0, false)
            ks += 2;
        } while (ks < kT);
        // ---- epilogue: straight from the accumulators (a lane holds 8 + 8 consecutive columns of its row per 32 x 32 block).  The loads of
        //      the next tile's steps 1 and 2 are in flight, its step 0 is in the ring, its kk = 0 fragments in registers ----
        {
            bn = rs_load_bias(p, GEGLU, nx2.valid, nx2.tn, tid);      // bias of the tile after the next (header of `bn`)
            const int col0 = GEGLU ? (cur.tn * RS_BN + wn * 128) >> 1 : cur.tn * RS_BN + wn * 128;
            const float* slab = reinterpret_cast<const float*>(smem + RS_SLAB + par * 1024) + wn * 128 + 8 * hi;
            par ^= 1;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int m = cur.tm * RS_BM + wm * 128 + i * 32 + l31;
                const unsigned row_dst = m < p.M ? (unsigned)m * (unsigned)p.dst_ld * 2u : kRsRowNone;
                const unsigned row_res = m < p.M ? (unsigned)m * (unsigned)p.n_dst * 2u : kRsRowNone;
                u32x4 res[NOUT][2];
                if constexpr (RES) {
#pragma unroll
                    for (int j = 0; j < NOUT; ++j) {
                        res[j][0] = rs_load_res(p, rs_off(row_res, col0 + 32 * j + 8 * hi, p.n_dst));
                        res[j][1] = rs_load_res(p, rs_off(row_res, col0 + 32 * j + 16 + 8 * hi, p.n_dst));
                    }
                }
#pragma unroll
                for (int j = 0; j < NOUT; ++j) {
                    // bias of the lane's 16 columns of packed block jb: registers 4a .. 4a+3 = columns 16 (a >> 1) + 8 hi + 4 (a & 1) ..
                    float c[16];
                    if constexpr (GEGLU) {
#pragma unroll
                        for (int a = 0; a < 4; ++a) {
                            const f32x4 bv = *reinterpret_cast<const f32x4*>(slab + 32 * (2 * j) + 16 * (a >> 1) + 4 * (a & 1));
                            const f32x4 bg = *reinterpret_cast<const f32x4*>(slab + 32 * (2 * j + 1) + 16 * (a >> 1) + 4 * (a & 1));
#pragma unroll
                            for (int e = 0; e < 4; ++e)
                                c[4 * a + e] = (acc[i][2 * j][4 * a + e] + bv[e]) * gelu_erf_16(acc[i][2 * j + 1][4 * a + e] + bg[e]);
                        }
                    } else {
#pragma unroll
                        for (int a = 0; a < 4; ++a) {
                            const f32x4 b = *reinterpret_cast<const f32x4*>(slab + 32 * j + 16 * (a >> 1) + 4 * (a & 1));
#pragma unroll
                            for (int e = 0; e < 4; ++e) c[4 * a + e] = acc[i][j][4 * a + e] + b[e];
                        }
                    }
                    u32x4 out[2];
                    if constexpr (RES) {
                        rs_pack<T, true>(c, p.out_scale, res[j], out);
                    } else {
                        const u32x4 none[2] = {u32x4{0u, 0u, 0u, 0u}, u32x4{0u, 0u, 0u, 0u}};
                        rs_pack<T, false>(c, p.out_scale, none, out);
                    }
                    rs_store(p, out[0], rs_off(row_dst, col0 + 32 * j + 8 * hi, p.n_dst));
                    rs_store(p, out[1], rs_off(row_dst, col0 + 32 * j + 16 + 8 * hi, p.n_dst));
                    __builtin_amdgcn_sched_barrier(0);      // (one block at a time: the in-flight register sets leave the epilogue ~60 VGPRs)
                }
            }
        }
    }
    // (the loads issued past the last tile are out of range: zeros into registers / slots nobody reads)
#undef RS_SUB
#undef RS_STEP
}

bool linear_rs_applicable(const mvldm_igemm_desc& d) {
    if (d.act_dtype == MVLDM_F32 || d.dst_dtype != d.act_dtype) return false;
    if (d.ksize != 1 || d.stride != 1 || d.upsample != 0 || d.row_bias || d.k_order != 1 || d.splitk > 1) return false;
    if (d.h_in != d.h_out || d.w_in != d.w_out || d.pad != 0) return false;
    if (d.epilogue != MVLDM_EPI_NONE && d.epilogue != MVLDM_EPI_GEGLU) return false;
    if (d.epilogue != MVLDM_EPI_NONE && d.residual) return false;
    const int n_dst = d.epilogue == MVLDM_EPI_GEGLU ? d.n_out / 2 : d.n_out;
    const int dst_ld = d.dst_ld > 0 ? d.dst_ld : n_dst;
    const int k = d.c0 + d.c1;
    if ((d.c1 == 0) != (d.src1 == nullptr) || d.c0 % 64 || d.c1 % 64 || k < 256 || k % 128 || d.k_pad != k || d.n_out % 8 || n_dst % 8 || dst_ld % 8 || dst_ld < n_dst) return false;
    if (d.epilogue == MVLDM_EPI_GEGLU && d.n_out % 64) return false;
    if (d.bias && ((uintptr_t)d.bias % 16)) return false;
    if (((uintptr_t)d.dst % 16) || (d.residual && ((uintptr_t)d.residual % 16))) return false;
    const double m = (double)d.n_img * d.h_out * d.w_out;
    return m * d.c0 * 2.0 < 4.0e9 && m * d.c1 * 2.0 < 4.0e9 && (double)d.n_pad * d.k_pad * 2.0 < 4.0e9 && m * dst_ld * 2.0 < 4.0e9 && m * n_dst * 2.0 < 4.0e9;
}

template <typename T, int EPI, bool RES> static int linear_rs_launch(const LinRSParams& p, int grid, hipStream_t s) {
    static std::atomic<uint64_t> done{0};
    if (int rc0 = ensure_dyn_smem(reinterpret_cast<const void*>(linear_rs_kernel<T, EPI, RES>), RS_SMEM, done)) return rc0;
    hipLaunchKernelGGL((linear_rs_kernel<T, EPI, RES>), dim3(grid), dim3(256), RS_SMEM, s, p);
    return check_launch();
}

int linear_rs_run(const mvldm_igemm_desc& d, hipStream_t s) {
    MVLDM_REQUIRE(linear_rs_applicable(d), "igemm: tile 19 (register-staged persistent Linear) does not apply to this problem");
    LinRSParams p;
    p.a = d.src0; p.a1 = d.src1; p.w = d.weight; p.bias = d.bias; p.residual = d.residual; p.dst = d.dst;
    p.M = d.n_img * d.h_out * d.w_out; p.K = d.c0 + d.c1; p.c0 = d.c0; p.c1 = d.c1; p.kt0 = d.c0 / 64; p.n_out = d.n_out; p.n_pad = d.n_pad;
    p.n_dst = d.epilogue == MVLDM_EPI_GEGLU ? d.n_out / 2 : d.n_out;
    p.dst_ld = d.dst_ld > 0 ? d.dst_ld : p.n_dst;
    p.k_steps = p.K / 64; p.out_scale = d.out_scale;
    const bool geglu = d.epilogue == MVLDM_EPI_GEGLU;
    p.tiles_m = (p.M + RS_BM - 1) / RS_BM; p.tiles_n = (d.n_pad + RS_BN - 1) / RS_BN;
    p.m_per = (p.tiles_m + 7) / 8;
    p.a_bytes = (unsigned)((double)p.M * p.c0 * 2.0); p.a1_bytes = (unsigned)((double)p.M * p.c1 * 2.0); p.w_bytes = (unsigned)((double)d.n_pad * d.k_pad * 2.0);
    p.bias_bytes = d.bias ? (unsigned)d.n_out * 4u : 0u;
    p.res_bytes = d.residual ? (unsigned)((double)p.M * p.n_dst * 2.0) : 0u;
    p.dst_bytes = (unsigned)((double)p.M * p.dst_ld * 2.0);
    if (kRsFake & 1) p.a_bytes = p.a1_bytes = 0;
    if (kRsFake & 2) p.w_bytes = 0;
    if (kRsFake & 4) p.dst_bytes = 0;
    if (kRsFake & 8) p.res_bytes = 0;
    static int n_cu = 0;
    if (!n_cu) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0)
            n_cu = prop.multiProcessorCount;
        else
            n_cu = 256;
    }
    // An XCD's workgroups (one per CU, fewer when it has fewer tiles) walk its tile list, ordered in gm x gn blocks of about one round's
    // tiles: the block shape that moves the fewest bytes into the XCD's L2 per tile -- gm activation row blocks + gn weight panels
    const int cu_x = std::max(1, n_cu / 8);
    p.wgx = std::min(cu_x, p.m_per * p.tiles_n);
    double best_cost = 1e300;
    p.gm = p.gn = 1;
    for (int gm = 1; gm <= std::min(p.wgx, p.m_per); ++gm) {
        const int gn = std::max(1, std::min(p.wgx / gm, p.tiles_n));
        const double cost = 1.0 / gn + 1.0 / gm;      // (a row block and a weight panel are the same 256 x K bytes here)
        if (cost < best_cost) { best_cost = cost; p.gm = gm; p.gn = gn; }
    }
    p.nbn = (p.tiles_n + p.gn - 1) / p.gn;
    const int grid = 8 * p.wgx;
    const bool res = d.residual != nullptr;
    return dispatch_dtype(d.act_dtype, [&](auto t) -> int {
        using T = decltype(t);
        if constexpr (sizeof(T) == 2) {
            if (geglu) return linear_rs_launch<T, MVLDM_EPI_GEGLU, false>(p, grid, s);
            return res ? linear_rs_launch<T, MVLDM_EPI_NONE, true>(p, grid, s) : linear_rs_launch<T, MVLDM_EPI_NONE, false>(p, grid, s);
        } else {
            return set_error(MVLDM_ERR_ARG, "igemm: tile 19 needs a 16-bit activation type");
        }
    });
}

}  // namespace mvldm
