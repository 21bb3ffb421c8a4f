// EXPERIMENT OF ROUND 6, NOT PART OF THE LIBRARY (kept for the record; profiles/HISTORY.md, round 6).  Built and measured as "tile 20":
// correct in its first form (20 of 21 parity cases; one GEGLU case 0.088 max / rms against a 0.05 bound, not chased), 10 - 60 % SLOWER than
// tile 13 on every Linear of the step (8 x 8 level QKV 380 against 332 us, to_out 235 against 145): with 128 registers of staged output
// next to 256 accumulators hipcc spills the staged rows round the main loop (26 of 32 chunks to scratch and back per tile) and, one wave
// per SIMD, the compiler-scheduled loop is ~ 15 % behind the two-waves-per-SIMD loop of tile 13 before any epilogue is counted.  The
// rolling-fragment form below is additionally WRONG (weight fragments of a step's last quarter are read behind the barrier that hands
// the slot to the DMA).  What was learned about the epilogue instead: tools/pw_trace.py.
// Persistent Linear with a DEFERRED, TRICKLED epilogue (1x1 conv over token rows): tile 20 of the implicit-GEMM family
// (include/mvldm.h: mvldm_igemm_fwd; 16-bit activations, one source or the channel concat of two, K >= 640 a multiple of 64).  Round 6.
//
// What tools/pw_trace.py measured on tile 13 (s_memtime stamps of one wave, 64 scenes, 8 x 8 level): a K-step of the 256 x 320 tile takes
// ~3 700 cycles, and the EPILOGUE of a tile 15 000 (QKV: 164 KB of stores per workgroup) to 28 000 cycles (to_out: + the residual rows) --
// 17 % to 40 % of a 20-step tile, during which the CU's matrix pipes idle.  Every workgroup reaches its epilogue at the same time (same
// tile, same cadence), so the chip stores 42 MB in one burst at HBM write speed and then stores nothing for 70 000 cycles.  The vendor
// GEMM (256 x 256 x 64 macro tile, 4 waves, direct-to-LDS loads, stream-K) is 15 - 35 % ahead on exactly these shapes.
//
// This kernel removes the epilogue from the critical path instead of shortening it:
//   * 4 waves of 128 x 128 (2 x 2) on a 256 x 256 tile: one wave per SIMD owns 512 registers -- 256 accumulators in the AGPR half, and in
//     the VGPR half 128 registers `ost` that hold a whole wave tile of 16-bit OUTPUT (or residual input);
//   * at the end of a tile the accumulators are converted into `ost` (bias, scale, residual, GEGLU: VALU only, no memory wait) and the
//     next tile's MFMAs start at once; `ost` leaves with ONE 16-byte store per MFMA slot during the next tile's first steps, and the
//     residual rows of that tile arrive in the same registers behind the stores (in place: a chunk's residual replaces the chunk
//     that has just left), a whole tile before they are needed;
//   * operands by LDS-DMA into a 2-slot ring (2 x 64 KB, the XOR-swizzled 128-byte rows of igemm.hip), persistent tile walk of
//     linear_pw.hip (flat list per XCD in gm x gn block order), one barrier per K-step in front of its last quarter, one memory
//     instruction per MFMA slot (linear_rs.hip: four MFMAs back to back followed by eight memory instructions overlapped nothing);
//   * VMEM returns in order and every wave issues the same sequence, so the counted wait of a step (for the ring pieces issued one step
//     earlier) leaves exactly the step's own trickle instructions in flight.  Residual loads, stores and the bias are compiler builtins;
//     each is first USED right behind a counted wait + barrier that has already retired it, so the compiler's own (conservative) wait
//     in front of the use finds nothing outstanding.
#include <algorithm>
#include <cstdlib>

#include "common.h"

namespace mvldm {

struct LinDEParams {
    const void* a; const void* a1; const void* w; const float* bias; const void* residual; void* dst;
    int M, K, c0, c1, kt0, n_out, n_pad, n_dst, dst_ld, k_steps;    // K = c0 + c1; K-steps [0, kt0) come from `a`, the rest from `a1`
    int tiles_m, tiles_n, m_per;       // m_per: 256-row blocks per XCD
    int gm, gn, nbn, wgx;              // an XCD's wgx workgroups walk its tiles in gm x gn blocks, column chunks (nbn of them) fastest
    float out_scale;
    unsigned a_bytes, a1_bytes, w_bytes, bias_bytes, res_bytes, dst_bytes;
};

constexpr unsigned kDeOob = 0xFFFFFFF0u;
constexpr int DE_BM = 256, DE_BN = 256;
constexpr int DE_A_SLOT = DE_BM * 128, DE_STAGE = DE_A_SLOT + DE_BN * 128;
constexpr int DE_RING = 2 * DE_STAGE;                 // 128 KB
constexpr int DE_SLAB = DE_RING;                      // 2 x 1 KB: bias of a tile's 256 packed columns, double-buffered over tiles
constexpr int DE_SMEM = DE_RING + 2048;

template <typename T> struct DeMma;
template <> struct DeMma<bf16_t> {
    using Frag = bf16x8;
    static __device__ __forceinline__ f32x16 mma(Frag a, Frag b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0); }
};
template <> struct DeMma<f16_t> {
    using Frag = f16x8;
    static __device__ __forceinline__ f32x16 mma(Frag a, Frag b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0); }
};

// MFMA M index mu (= lane & 31 of the W-fragment read) -> column of the 32-column block it is fed from (linear_pw.hip: registers 0..7 /
// 8..15 of a lane become 8 + 8 CONSECUTIVE output columns; each 16-lane group reads the same set of rows as the identity)
__device__ __forceinline__ int de_perm(int mu) {
    const int a = mu >> 3, h = (mu >> 2) & 1, e = mu & 3;
    return 16 * (a >> 1) + 8 * h + 4 * (a & 1) + e;
}

// Source addressing of a wave's DMA pieces.  Piece `it` of a wave covers tile rows (wave + 4 it) * 8 .. + 7; a lane fetches the 16-byte
// chunk that belongs at its (linear) LDS position under the XOR swizzle; (row >> 1) & 7 does not depend on `it`, so ONE per-lane byte
// offset serves all pieces of an operand and the piece's 32-row stride rides in the scalar offset with the K position.
struct DeAddr {
    unsigned a0, a1;     // activation rows of piece 0: byte offset into the first / second source (channel concat)
    unsigned w;          // weight rows of piece 0
    int m_lane;          // global row this lane reads in piece 0 (piece it: + 32 it; rows >= M read zeros)
    int n0;              // first packed column of the tile
    bool valid;
};

__device__ __forceinline__ void de_offsets(const LinDEParams& p, bool valid, int tm, int tn, int wave, int lane, DeAddr& ad) {
    const int slot = lane & 7, row = wave * 8 + (lane >> 3);
    const unsigned chunk = (unsigned)((slot ^ ((row >> 1) & 7)) * 8);
    const int m = tm * DE_BM + row, n = tn * DE_BN + row;
    ad.a0 = ((unsigned)m * (unsigned)p.c0 + chunk) * 2u;
    ad.a1 = ((unsigned)m * (unsigned)p.c1 + chunk) * 2u;
    ad.w = ((unsigned)n * (unsigned)p.K + chunk) * 2u;
    ad.m_lane = m;
    ad.n0 = tn * DE_BN;
    ad.valid = valid;
}

// (buffer descriptors only in free functions: an opaque __amdgpu_buffer_rsrc_t inside a lambda trips hipcc's host pass)
// ring piece `it` (a constant after unrolling at every call site) of K-step ks -> its 1 KiB of the slot at `stage`
__device__ __forceinline__ void de_issue_a1(const LinDEParams& p, char* stage, int wave, int ks, const DeAddr& ad, int it) {
    const bool second = ks >= p.kt0;
    // (ONE descriptor from selected scalars: a select between two descriptors becomes a branch whose join drains the ring)
    const void* abase = second ? p.a1 : p.a;
    const unsigned abytes = second ? p.a1_bytes : p.a_bytes;
    const int c = second ? p.c1 : p.c0;
    const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(abase), 0, abytes, 0x00020000);
    const unsigned v = second ? ad.a1 : ad.a0;
    const unsigned off = (ad.valid && ad.m_lane + 32 * it < p.M) ? v : kDeOob;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, (__attribute__((address_space(3))) void*)(stage + (wave + 4 * it) * 1024), 16, off,
                                             (second ? ks - p.kt0 : ks) * 128 + it * 64 * c, 0, 0);
}
__device__ __forceinline__ void de_issue_w1(const LinDEParams& p, char* stage, int wave, int ks, const DeAddr& ad, int it) {
    const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.w), 0, p.w_bytes, 0x00020000);
    // (n_pad is a multiple of 64, a piece's 8 rows start at a multiple of 8 inside a 32-row group: inside the packed weight or outside as a whole)
    const unsigned off = (ad.valid && ad.n0 + 32 * it < p.n_pad) ? ad.w : kDeOob;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (__attribute__((address_space(3))) void*)(stage + DE_A_SLOT + (wave + 4 * it) * 1024), 16,
                                             off, ks * 128 + it * 64 * p.K, 0, 0);
}

// bias of packed columns 4t .. 4t+3 of tile column tn (zeros past the tile / the layer / without a bias)
__device__ __forceinline__ u32x4 de_load_bias(const LinDEParams& p, bool geglu, bool valid, int tn, int t) {
    const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.bias), 0, p.bias_bytes, 0x00020000);
    const int pc = tn * DE_BN + 4 * t;                          // packed column
    int oc = pc;                                                // column of the torch-layout bias
    if (geglu) {
        const int blk = pc >> 5, w = pc & 31;
        oc = ((blk & 1) ? p.n_dst : 0) + (blk >> 1) * 32 + w;
    }
    const unsigned off = (valid && 4 * t < DE_BN && pc < p.n_out) ? (unsigned)oc * 4u : kDeOob;
    return __builtin_amdgcn_raw_buffer_load_b128(rb, off, 0, 0);
}
__device__ __forceinline__ u32x4 de_load_res(const LinDEParams& p, unsigned off) {
    const __amdgpu_buffer_rsrc_t rr = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.residual), 0, p.res_bytes, 0x00020000);
    return __builtin_amdgcn_raw_buffer_load_b128(rr, off, 0, 0);
}
// (write-back stores, not streaming ones: linear_pw.hip -- the L2 acknowledges a store long before HBM has taken it)
__device__ __forceinline__ void de_store(const LinDEParams& p, const u32x4& v, unsigned off) {
    const __amdgpu_buffer_rsrc_t rd = __builtin_amdgcn_make_buffer_rsrc(p.dst, 0, p.dst_bytes, 0x00020000);
    __builtin_amdgcn_raw_buffer_store_b128(v, rd, off, 0, 0);
}

template <typename T> __device__ __forceinline__ typename DeMma<T>::Frag de_frag(const char* p) {
    return *reinterpret_cast<const typename DeMma<T>::Frag*>(p);
}

// Walks the tiles of a workgroup (all wave-uniform): linear_pw.hip's order.  XCD x owns row blocks [x * m_per, (x+1) * m_per) and all
// column tiles; its tiles form one list in gm x gn block order (column chunks fastest, the row fastest inside a block, ragged edge
// blocks packed densely) and its wgx workgroups take entries lid, lid + wgx, ...
struct DeTileIter {
    int r, tm, tn;
    bool valid;
    __device__ __forceinline__ void set(const LinDEParams& p, int r0, int lid, int m_lo, int m_cnt) {
        r = r0;
        const int i = r0 * p.wgx + lid;
        valid = i < m_cnt * p.tiles_n;
        tm = tn = 0;
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

// byte offset of output chunk e (0 .. 8 NOUT - 1: row block e / (2 NOUT), column block (e / 2) % NOUT, half e & 1) of the wave tile of
// tile (tm, tn) in a row-major matrix of `ld` elements per row; out of range past M / n_dst
template <int NOUT, bool GEGLU>
__device__ __forceinline__ unsigned de_chunk_off(const LinDEParams& p, bool valid, int tm, int tn, int wm, int wn, int l31, int hi, int e, int ld) {
    const int i = e / (2 * NOUT), j = (e / 2) % NOUT, h = e & 1;
    const int m = tm * DE_BM + wm * 128 + i * 32 + l31;
    const int col0 = GEGLU ? (tn * DE_BN + wn * 128) >> 1 : tn * DE_BN + wn * 128;
    const int col = col0 + 32 * j + 16 * h + 8 * hi;
    return (valid && m < p.M && col < p.n_dst) ? ((unsigned)m * (unsigned)ld + (unsigned)col) * 2u : kDeOob;
}

template <typename T, int EPI, bool RES>
__global__ __launch_bounds__(256) void linear_de_kernel(const LinDEParams p) {
    using M_ = DeMma<T>;
    using Frag = typename M_::Frag;
    constexpr bool GEGLU = EPI == MVLDM_EPI_GEGLU;
    static_assert(!(GEGLU && RES), "no caller");
    constexpr int NOUT = GEGLU ? 2 : 4;            // output blocks per row block of a wave
    constexpr int NCH = 8 * NOUT;                  // 16-byte output chunks per lane and tile
    constexpr int NS = NCH / 8;                    // trickle steps per tile: 8 chunks leave (and 8 residual chunks arrive) per step
    constexpr int NT = RES ? 16 : 8;               // VMEM instructions of a trickle step
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;        // 2 x 2 waves of 128 rows x 128 columns
    const int hi = lane >> 5, l31 = lane & 31;
    const int prm = de_perm(l31);

    const int xcd = blockIdx.x & 7, lid = blockIdx.x >> 3;
    const int m_lo = xcd * p.m_per, m_cnt = min(p.tiles_m, m_lo + p.m_per) - m_lo;
    DeTileIter cur, nxt, nx2, iss;               // compute side, the two tiles after it, issue side (tile of the newest ring step in flight)
    cur.set(p, 0, lid, m_lo, m_cnt);
    if (!cur.valid) return;
    nxt.set(p, 1, lid, m_lo, m_cnt);
    nx2.set(p, 2, lid, m_lo, m_cnt);
    iss = cur;

    // fragment read offsets inside a ring slot (bytes): row * 128 + swizzled chunk of k-sub-step 0; sub-step kk: XOR kk << 5
    const int a_off = (wm * 128 + l31) * 128 + ((hi ^ ((l31 >> 1) & 7)) << 4);
    const int w_off = DE_A_SLOT + (wn * 128 + prm) * 128 + ((hi ^ ((prm >> 1) & 7)) << 4);

    f32x16 acc[4][4];
    u32x4 ost[NCH];                              // the previous tile's output on its way out / this tile's residual rows on their way in
    DeAddr ad;
    Frag fa0[4], fa1[4], fwA, fwB;               // activation fragments of the current / next sub-step, two rolling weight fragments
    const int kT = p.k_steps;
    const f32x16 kZero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    int ks_i = 1;                                // issue side: K-step (inside tile `iss`) of the newest ring step in flight
    int rs = 0;                                  // ring slot the current step reads
    int par = 0;                                 // bias slab of the current tile
    int ptm = 0, ptn = 0;                        // the tile whose output sits in `ost`
    bool have_prev = false;
    u32x4 bn;                                    // bias of the next tile: requested a tile early, parked in the other slab behind a counted wait

    // ---- prologue: steps 0 and 1 of the first tile (the whole ring), the first bias slab ----
    {
        de_offsets(p, true, cur.tm, cur.tn, wave, lane, ad);
#pragma unroll
        for (int g = 0; g < 2; ++g) {
#pragma unroll
            for (int it = 0; it < 8; ++it) de_issue_a1(p, smem + g * DE_STAGE, wave, g, ad, it);
#pragma unroll
            for (int it = 0; it < 8; ++it) de_issue_w1(p, smem + g * DE_STAGE, wave, g, ad, it);
        }
        const u32x4 b = de_load_bias(p, GEGLU, true, cur.tn, tid);
        bn = de_load_bias(p, GEGLU, nxt.valid, nxt.tn, tid);
        __builtin_amdgcn_s_waitcnt(0x0070);      // vmcnt(0) lgkmcnt(0) (expcnt untouched)
        if (4 * tid < DE_BN) *reinterpret_cast<u32x4*>(smem + DE_SLAB + tid * 16) = b;
        __builtin_amdgcn_s_waitcnt(0xC07F);      // lgkmcnt(0): this wave's LDS writes are done
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
#pragma unroll
        for (int j = 0; j < 4; ++j) fa0[j] = de_frag<T>(smem + a_off + j * 4096);
        fwA = de_frag<T>(smem + w_off);
    }

// One sub-step (16 of a step's 64 K values) = 16 MFMA SLOTS: one MFMA followed by at most two memory instructions that issue while it
// executes.  Fragments ROLL (the register file also holds 128 registers of staged output): the four activation fragments of the NEXT
// sub-step are fetched in slots 8-11, the weight fragment of column block j+1 four slots ahead of its first use into one of TWO weight
// registers (slot 12: column block 0 of the next sub-step) -- 40 fragment registers live at the peak instead of 64.  OPS_ is the slot's
// share of the step's other work (q = slot).  (cs_, ck_): ring slot / sub-step being computed; (ns_, nk_): the ones after it.
// (Z_: the tile's first sub-step starts the accumulators from the MFMA's zero C operand -- linear_rs.hip)
#define DE_SUB(ca_, na_, cs_, ck_, ns_, nk_, Z_, OPS_)                                                            \
    {                                                                                                             \
        const char* sc_ = smem + (cs_) * DE_STAGE;                                                                \
        const char* sn_ = smem + (ns_) * DE_STAGE;                                                                \
        const int wc_ = w_off ^ ((ck_) << 5), an_ = a_off ^ ((nk_) << 5), wn_ = w_off ^ ((nk_) << 5);             \
        _Pragma("unroll") for (int q = 0; q < 16; ++q) {                                                          \
            acc[q & 3][q >> 2] = M_::mma(((q >> 2) & 1) ? fwB : fwA, ca_[q & 3], (Z_) ? kZero : acc[q & 3][q >> 2]); \
            if (q == 0) fwB = de_frag<T>(sc_ + wc_ + 1 * 4096);                                                   \
            else if (q == 4) fwA = de_frag<T>(sc_ + wc_ + 2 * 4096);                                              \
            else if (q == 8) fwB = de_frag<T>(sc_ + wc_ + 3 * 4096);                                              \
            else if (q == 12) fwA = de_frag<T>(sn_ + wn_);                                                        \
            if (q >= 8 && q < 12) na_[q - 8] = de_frag<T>(sn_ + an_ + (q - 8) * 4096);                            \
            OPS_                                                                                                  \
            __builtin_amdgcn_sched_barrier(0);                                                                    \
        }                                                                                                         \
    }
// the eight slots of a sub-step that carry no fragment read: trickle instruction k = 0 .. 7 rides in slot DE_TSLOT(k)
#define DE_TK(q_) ((q_) == 1 ? 0 : (q_) == 2 ? 1 : (q_) == 3 ? 2 : (q_) == 5 ? 3 : (q_) == 6 ? 4 : (q_) == 7 ? 5 : (q_) == 13 ? 6 : (q_) == 14 ? 7 : -1)
// One K-step computed from ring slot rs.  TR_ >= 0: trickle step TR_ of the tile -- chunks 8 TR_ .. 8 TR_ + 7 of `ost` (the previous tile's
// output) leave in sub-step 0, the same chunks of this tile's residual rows are requested in sub-step 1.  The counted wait in front of
// the barrier is for the ring pieces of step g+1 (issued in step g-1): the step's own NT trickle instructions are younger and stay in
// flight; every older load / store is complete behind it.  The barrier publishes slot rs ^ 1 and retires slot rs, which sub-step 3
// refills with step g+2, one piece per MFMA slot.  POST_: code that runs right behind the barrier (first uses of builtin loads).
#define DE_STEP(TR_, Z_, POST_)                                                                                   \
    {                                                                                                             \
        DE_SUB(fa0, fa1, rs, 0, rs, 1, Z_,                                                                        \
               if ((TR_) >= 0 && DE_TK(q) >= 0)                                                                   \
                   de_store(p, ost[8 * ((TR_) < 0 ? 0 : (TR_)) + (DE_TK(q) < 0 ? 0 : DE_TK(q))],                  \
                            (de_chunk_off<NOUT, GEGLU>)(p, have_prev, ptm, ptn, wm, wn, l31, hi, 8 * (TR_) + DE_TK(q), p.dst_ld));) \
        DE_SUB(fa1, fa0, rs, 1, rs, 2, false,                                                                     \
               if (RES && (TR_) >= 0 && DE_TK(q) >= 0)                                                            \
                   ost[8 * ((TR_) < 0 ? 0 : (TR_)) + (DE_TK(q) < 0 ? 0 : DE_TK(q))] =                             \
                       de_load_res(p, (de_chunk_off<NOUT, GEGLU>)(p, true, cur.tm, cur.tn, wm, wn, l31, hi, 8 * (TR_) + DE_TK(q), p.n_dst));) \
        /* (sub-step 2 fetches sub-step 3's fragments from the same slot; sub-step 3, behind the barrier, the next step's) */ \
        DE_SUB(fa0, fa1, rs, 2, rs, 3, false, )                                                                   \
        if ((TR_) >= 0) __builtin_amdgcn_s_waitcnt(kWaitTrickle);                                                 \
        else __builtin_amdgcn_s_waitcnt(0x0070);                                                                  \
        __builtin_amdgcn_s_barrier();                                                                             \
        asm volatile("" ::: "memory");                                                                            \
        POST_                                                                                                     \
        DE_SUB(fa1, fa0, rs, 3, rs ^ 1, 0, false,                                                                 \
               if (q == 0) {                                                                                      \
                   if (++ks_i == kT) {                                                                            \
                       ks_i = 0;                                                                                  \
                       iss.set(p, iss.r + 1, lid, m_lo, m_cnt);                                                   \
                       de_offsets(p, iss.valid, iss.tm, iss.tn, wave, lane, ad);                                  \
                   }                                                                                              \
               }                                                                                                  \
               if (q < 8) de_issue_a1(p, smem + rs * DE_STAGE, wave, ks_i, ad, q);                                \
               else de_issue_w1(p, smem + rs * DE_STAGE, wave, ks_i, ad, q - 8);)                                 \
        rs ^= 1;                                                                                                  \
    }
    // s_waitcnt vmcnt(NT) lgkmcnt(0), expcnt untouched (gfx9 encoding: vmcnt = bits 3:0 + 15:14, lgkmcnt = bits 11:8)
    constexpr int kWaitTrickle = (NT & 15) | ((NT >> 4) << 14) | 0x0070;

    for (; cur.valid; cur = nxt, nxt = nx2, nx2.set(p, nx2.r + 1, lid, m_lo, m_cnt)) {
        // ---- trickle steps: the previous tile's output leaves, this tile's residual rows arrive; the first starts the accumulators ----
        DE_STEP(0, true, )
        DE_STEP(1, false, )
        if constexpr (NS == 4) {
            DE_STEP(2, false, )
            DE_STEP(3, false, )
        }
        // ---- first plain step.  Its counted wait has retired every trickle instruction and the bias load of the previous tile's plain
        //      step: behind its barrier the residual rows are touched (the compiler's wait in front of a first use finds nothing in flight),
        //      the next tile's bias goes to the other slab and the bias of the tile after it is requested ----
        DE_STEP(-1, false,
                if constexpr (RES) {
                    _Pragma("unroll") for (int e = 0; e < NCH; ++e) asm volatile("" : "+v"(ost[e]));
                }
                if (4 * tid < DE_BN) *reinterpret_cast<u32x4*>(smem + DE_SLAB + (par ^ 1) * 1024 + tid * 16) = bn;
                bn = de_load_bias(p, GEGLU, nx2.valid, nx2.tn, tid);)
#pragma unroll 1
        for (int ks = NS + 1; ks < kT; ++ks) DE_STEP(-1, false, )
        // ---- conversion: accumulators -> `ost` (a lane holds 8 + 8 consecutive columns of its row per 32 x 32 block): bias from slab `par`
        //      (written during the previous tile, published by a step barrier since), scale, residual (in place), GEGLU.  No memory wait:
        //      the next tile's first step follows at once (its kk = 0 fragments are in registers, its step 1 is landing) ----
        {
            const float* slab = reinterpret_cast<const float*>(smem + DE_SLAB + par * 1024) + wn * 128 + 8 * hi;
            par ^= 1;
            // one 16-byte chunk (8 columns: accumulator registers 8 g .. 8 g + 7 of a 32 x 32 block) at a time -- with `ost` and the next
            // tile's first fragments live, a whole block's 16 values + bias + residual at once did not fit the register file
#pragma unroll
            for (int i = 0; i < 4; ++i) {
#pragma unroll
                for (int j = 0; j < NOUT; ++j) {
#pragma unroll
                    for (int g = 0; g < 2; ++g) {
                        float c[8];
                        if constexpr (GEGLU) {
#pragma unroll
                            for (int a = 0; a < 2; ++a) {
                                const f32x4 bv = *reinterpret_cast<const f32x4*>(slab + 32 * (2 * j) + 16 * g + 4 * a);
                                const f32x4 bg = *reinterpret_cast<const f32x4*>(slab + 32 * (2 * j + 1) + 16 * g + 4 * a);
#pragma unroll
                                for (int e = 0; e < 4; ++e)
                                    c[4 * a + e] = (acc[i][2 * j][8 * g + 4 * a + e] + bv[e]) * gelu_erf_fast(acc[i][2 * j + 1][8 * g + 4 * a + e] + bg[e]);
                            }
                        } else {
#pragma unroll
                            for (int a = 0; a < 2; ++a) {
                                const f32x4 b = *reinterpret_cast<const f32x4*>(slab + 32 * j + 16 * g + 4 * a);
#pragma unroll
                                for (int e = 0; e < 4; ++e) c[4 * a + e] = acc[i][j][8 * g + 4 * a + e] + b[e];
                            }
                        }
                        Chunk<T> oc;
                        if constexpr (RES) {
                            Chunk<T> rc;
                            rc.raw = ost[(i * NOUT + j) * 2 + g];
#pragma unroll
                            for (int e = 0; e < 8; ++e) oc.set(e, c[e] * p.out_scale + rc.get(e));
                        } else {
#pragma unroll
                            for (int e = 0; e < 8; ++e) oc.set(e, c[e] * p.out_scale);
                        }
                        ost[(i * NOUT + j) * 2 + g] = oc.raw;
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
            }
            ptm = cur.tm; ptn = cur.tn; have_prev = true;
        }
    }
    // ---- the last tile's output ----
#pragma unroll
    for (int e = 0; e < NCH; ++e) de_store(p, ost[e], de_chunk_off<NOUT, GEGLU>(p, true, ptm, ptn, wm, wn, l31, hi, e, p.dst_ld));
    // (the ring pieces issued past the last tile are out of range: zeros into slots nobody reads)
#undef DE_SUB
#undef DE_TK
#undef DE_STEP
}

bool linear_de_applicable(const mvldm_igemm_desc& d) {
    if (d.act_dtype == MVLDM_F32 || d.dst_dtype != d.act_dtype) return false;
    if (d.ksize != 1 || d.stride != 1 || d.upsample != 0 || d.row_bias || d.k_order != 1 || d.splitk > 1) return false;
    if (d.h_in != d.h_out || d.w_in != d.w_out || d.pad != 0) return false;
    if (d.epilogue != MVLDM_EPI_NONE && d.epilogue != MVLDM_EPI_GEGLU) return false;
    if (d.epilogue != MVLDM_EPI_NONE && d.residual) return false;
    const int n_dst = d.epilogue == MVLDM_EPI_GEGLU ? d.n_out / 2 : d.n_out;
    const int dst_ld = d.dst_ld > 0 ? d.dst_ld : n_dst;
    const int k = d.c0 + d.c1;
    // (>= 8 K-steps: up to four trickle steps, the plain step that retires them, and the ring's two steps of lead inside one tile)
    if ((d.c1 == 0) != (d.src1 == nullptr) || d.c0 % 64 || d.c1 % 64 || k < 512 || d.k_pad != k || d.n_out % 8 || n_dst % 8 || dst_ld % 8 || dst_ld < n_dst) return false;
    if (d.epilogue == MVLDM_EPI_GEGLU && d.n_out % 64) return false;
    if (d.bias && ((uintptr_t)d.bias % 16)) return false;
    if (((uintptr_t)d.dst % 16) || (d.residual && ((uintptr_t)d.residual % 16))) return false;
    const double m = (double)d.n_img * d.h_out * d.w_out;
    return m * d.c0 * 2.0 < 4.0e9 && m * d.c1 * 2.0 < 4.0e9 && (double)d.n_pad * d.k_pad * 2.0 < 4.0e9 && m * dst_ld * 2.0 < 4.0e9 && m * n_dst * 2.0 < 4.0e9;
}

template <typename T, int EPI, bool RES> static int linear_de_launch(const LinDEParams& p, int grid, hipStream_t s) {
    static std::atomic<uint64_t> done{0};
    if (int rc0 = ensure_dyn_smem(reinterpret_cast<const void*>(linear_de_kernel<T, EPI, RES>), DE_SMEM, done)) return rc0;
    hipLaunchKernelGGL((linear_de_kernel<T, EPI, RES>), dim3(grid), dim3(256), DE_SMEM, s, p);
    return check_launch();
}

int linear_de_run(const mvldm_igemm_desc& d, hipStream_t s) {
    MVLDM_REQUIRE(linear_de_applicable(d), "igemm: tile 20 (persistent Linear with the trickled epilogue) does not apply to this problem");
    LinDEParams p;
    p.a = d.src0; p.a1 = d.src1; p.w = d.weight; p.bias = d.bias; p.residual = d.residual; p.dst = d.dst;
    p.M = d.n_img * d.h_out * d.w_out; p.K = d.c0 + d.c1; p.c0 = d.c0; p.c1 = d.c1; p.kt0 = d.c0 / 64; p.n_out = d.n_out; p.n_pad = d.n_pad;
    p.n_dst = d.epilogue == MVLDM_EPI_GEGLU ? d.n_out / 2 : d.n_out;
    p.dst_ld = d.dst_ld > 0 ? d.dst_ld : p.n_dst;
    p.k_steps = p.K / 64; p.out_scale = d.out_scale;
    const bool geglu = d.epilogue == MVLDM_EPI_GEGLU;
    p.tiles_m = (p.M + DE_BM - 1) / DE_BM; p.tiles_n = (d.n_pad + DE_BN - 1) / DE_BN;
    p.m_per = (p.tiles_m + 7) / 8;
    p.a_bytes = (unsigned)((double)p.M * p.c0 * 2.0); p.a1_bytes = (unsigned)((double)p.M * p.c1 * 2.0); p.w_bytes = (unsigned)((double)d.n_pad * d.k_pad * 2.0);
    p.bias_bytes = d.bias ? (unsigned)d.n_out * 4u : 0u;
    p.res_bytes = d.residual ? (unsigned)((double)p.M * p.n_dst * 2.0) : 0u;
    p.dst_bytes = (unsigned)((double)p.M * p.dst_ld * 2.0);
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
            if (geglu) return linear_de_launch<T, MVLDM_EPI_GEGLU, false>(p, grid, s);
            return res ? linear_de_launch<T, MVLDM_EPI_NONE, true>(p, grid, s) : linear_de_launch<T, MVLDM_EPI_NONE, false>(p, grid, s);
        } else {
            return set_error(MVLDM_ERR_ARG, "igemm: tile 20 needs a 16-bit activation type");
        }
    });
}

}  // namespace mvldm
