// Persistent WIDE Linear (1x1 conv over token rows): tile 13 of the implicit-GEMM family (include/mvldm.h: mvldm_igemm_fwd;
// 16-bit activations, one source or the channel concat of two, K a multiple of 64 and >= 320).  Round 4.
//
// What the op tables of rounds 2-3 said about the K = 320 ... 1280 Linears (40 of the 105 ms of a DDIM step at 64 scenes): in the
// one-tile-per-workgroup kernels (igemm.hip) ring fill, main loop and epilogue of a tile ADD UP on a CU that holds one workgroup;
// the LDS park of the epilogue alone (160 ds_write_b32 + 80 ds_read_b128 per wave and tile) is as long as the main loop at
// K = 320.  Tile 12 (linear_pp.hip) hides the epilogue under the next tile's MFMAs but pays for its two accumulator sets with a
// 128-column tile: 85 flop per L2->LDS byte.  This kernel keeps the WIDE tile (256 x 320, or 256 x 256 for GEGLU: 142 / 128 flop
// per byte) and removes the other two costs:
//   * PERSISTENT: a workgroup walks its output tiles as one stream of K-steps of 64; the 2-slot LDS-DMA ring (2 x 72 KB at 256 x 320)
//     never drains -- a slot is retired as soon as its last fragments are in registers (in front of the step's last quarter) and
//     refilled at once with step g+2, across tile boundaries, so the next tile's first two steps land while the epilogue runs;
//   * PARK-FREE EPILOGUE: the product is computed transposed (W fragment = MFMA A operand), so a lane holds ONE output row; the
//     W rows a wave feeds to the MFMA's M index are PERMUTED (mu = 8a + 4h + e  <-  column 16(a>>1) + 8h + 4(a&1) + e of the
//     32-column block), which costs nothing (it is the lane's LDS read address) and leaves accumulator registers 0..7 / 8..15 of
//     a lane = 8 + 8 CONSECUTIVE output columns: two 16-byte stores per 32 x 32 block straight from registers -- no LDS park, no
//     v_permlane swaps; a store instruction covers 32 rows x 32 contiguous bytes.  GEGLU: value and gate blocks use the same
//     permutation, so a lane holds a column's value AND gate.
// Counted waits: every wave issues the same VMEM sequence.  The loads of the epilogue (bias slab; residual: inline asm, rolling
// window) are consumed behind explicit waits; the output stores are inline asm (hipcc treats loads and stores in flight as unordered
// and would fall back to vmcnt(0) around them) and are issued at the END of the epilogue, after the last load has been consumed.
// VMEM returns in order: the wait in a tile's step 0 (for step 1's pieces, issued by the previous tile's last step) leaves exactly the
// previous epilogue's stores outstanding; the first wait that has to count them is the one in step 1.  The epilogue itself does not
// drain the ring: the next tile's bias is requested before the last step's pieces and waited for with those in flight.
// History of the round (profiles/r04_pw_probe_*.txt, r04_dma_rate_probe.txt): the first version used K-steps of 32 in a 4-slot
// ring (64-byte LDS rows).  tools/pw_probe.py showed its K loop sound (L0 QKV at 64 scenes 371 us with the stores off, 664 with)
// but tools/probe/dma_rate.hip showed what 64-byte row pieces cost: an LDS-DMA stream of half cache lines runs at 16 TB/s from a hot
// L2 against 28 TB/s with full 128-byte lines, and at 3.6 against 6.2 TB/s from HBM (every line fetched twice).  Hence full lines
// (BK = 64, the XOR-swizzled 128-byte LDS rows of igemm.hip) and the deepest ring 160 KB allow.  Write-back (not streaming) stores:
// the L2 acknowledges a tile's store burst (160 KB per CU, every CU at once) long before HBM has taken it, which is what the next
// tile's first counted wait needs (QKV 712 -> 552 us).
#include <algorithm>
#include <cstdlib>

#include "common.h"

namespace mvldm {

struct LinPWParams {
    const void* a; const void* a1; const void* w; const float* bias; const void* residual; void* dst;
    int M, K, c0, c1, kt0, n_out, n_pad, n_dst, dst_ld, k_steps;    // K = c0 + c1; K-steps [0, kt0) come from `a`, the rest from `a1`
    int tiles_m, tiles_n, m_per;       // m_per: 256-row blocks per XCD
    int gm, gn, nbn, wgx;              // an XCD's wgx workgroups walk its tiles in gm x gn blocks, column chunks (nbn of them) fastest
    int nt_store, touch;               // touch: L2 prefetch of the activation rows (pw_touch_a)
    int spread;                        // the pieces of a middle step are issued in three groups over three sub-steps (PW_STEP)
    float out_scale;
    unsigned a_bytes, a1_bytes, w_bytes, bias_bytes, res_bytes, dst_bytes;
    unsigned* trace; int trace_blk, trace_wave, trace_stagger;     // EXPERIMENT (-DMVLDM_PW_TRACE, tools/pw_trace.py): s_memtime stamps of one wave
};

#ifdef MVLDM_PW_TRACE
// stamp k of the traced wave: low 32 bits of the shader clock into the spare LDS behind the bias slab (192 stamps), dumped at the end
#define PW_STAMP() if (tr_on && tr_n < 190) { reinterpret_cast<unsigned*>(smem + G::SLAB + 1280)[tr_n++] = (unsigned)__builtin_readcyclecounter(); }
#else
#define PW_STAMP()
#endif

#ifdef MVLDM_EXPERIMENTS
static const int kPwFake = knob_int("MVLDM_PW_FAKE", 0);   // 1: no A traffic, 2: no W traffic, 4: no stores
#else
static constexpr int kPwFake = 0;
#endif
#ifdef MVLDM_EXPERIMENTS_NOGELU
#define PW_GELU(x) (x)
#else
#define PW_GELU(x) gelu_erf_16(x)
#endif

constexpr unsigned kPwOob = 0xFFFFFFF0u;
constexpr unsigned kPwRowNone = 0xFFFFFFFFu;
constexpr int PW_BM = 256, PW_NW = 8;
#ifndef PW_SP3
#define PW_SP3 4      // placement of a middle step's pieces (PwGeo::Q*; experiment builds override it: tools/pw_spread.sh, profiles/r06_pw_spread_variants.txt)
#define PW_SP0 3
#define PW_SP1 2
#endif

template <int TN> struct PwGeo {
    static constexpr int BN = 64 * TN;                  // 2 column waves of TN 32-column blocks
    static constexpr int A_SLOT = PW_BM * 128, W_SLOT = BN * 128, STAGE = A_SLOT + W_SLOT;
    static constexpr int A_IT = PW_BM / 8 / PW_NW;      // 1 KiB DMA pieces (8 rows of 128 bytes) per wave: activation rows
    static constexpr int W_IT = BN / 8 / PW_NW;         // ... weight rows (= TN)
    static constexpr int P = A_IT + W_IT;               // DMA instructions per wave and step
    // `spread`: pieces [0, Q0) go out behind the barrier (sub-step 3), [Q0, Q1) / [Q1, Q2) / [Q2, P) in sub-steps 0 / 1 / 2 of the next step
    static constexpr int Q0 = PW_SP3 < P ? PW_SP3 : P, Q1 = Q0 + PW_SP0 < P ? Q0 + PW_SP0 : P, Q2 = Q1 + PW_SP1 < P ? Q1 + PW_SP1 : P;
    static constexpr int RING = 2 * STAGE;
    static constexpr int SLAB = RING;                   // bias of the tile's BN packed columns
    static constexpr int SMEM = SLAB + 2048;
};

constexpr int pw_wait(int n) { return (n & 15) | ((n >> 4) << 14) | 0x0F70; }   // s_waitcnt vmcnt(n), expcnt / lgkmcnt untouched (gfx9 encoding)

template <typename T> struct PwMma;
template <> struct PwMma<bf16_t> {
    using Frag = bf16x8;
    static __device__ __forceinline__ f32x16 mma(Frag a, Frag b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0); }
};
template <> struct PwMma<f16_t> {
    using Frag = f16x8;
    static __device__ __forceinline__ f32x16 mma(Frag a, Frag b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0); }
};

// MFMA M index mu (= lane & 31 of the W-fragment read) -> column of the 32-column block it is fed from (header).  The permutation
// maps each 16-lane group of a ds_read_b128 onto the same SET of rows as the identity, so the swizzle stays conflict-free.
__device__ __forceinline__ int pw_perm(int mu) {
    const int a = mu >> 3, h = (mu >> 2) & 1, e = mu & 3;
    return 16 * (a >> 1) + 8 * h + 4 * (a & 1) + e;
}

// Source addressing of a wave's DMA pieces.  Piece `it` of a wave covers tile rows (wave + 8 it) * 8 .. + 7, a lane fetches the
// 16-byte chunk that belongs at its (linear) LDS position under the XOR swizzle; (row >> 1) & 7 does not depend on `it`, so ONE
// per-lane byte offset serves all pieces of an operand and the piece's 64-row stride rides in the scalar offset with the K position.
struct PwAddr {
    unsigned a0, a1;     // activation rows of piece 0: byte offset into the first / second source (channel concat)
    unsigned w;          // weight rows of piece 0
    int m0, n0;          // first row of the tile this lane's pieces belong to (validity: m0 + row < M; W pieces: wave-uniform)
    bool valid;
};

// (buffer descriptors only in free functions: an opaque __amdgpu_buffer_rsrc_t inside a lambda trips hipcc's host pass)
// The DMA pieces of a step are issued in two groups (activation rows, weight rows) so that the main loop can place each between
// MFMAs: issued back to back at the top of a step they held BOTH waves of a SIMD in the address path while its matrix pipe idled.
template <int TN, int LO = 0, int HI = 1 << 20>
__device__ __forceinline__ void pw_issue_a(const LinPWParams& p, char* smem, int slot, int wave, int lane, int ks, const PwAddr& ad) {
    using G = PwGeo<TN>;
    const bool second = ks >= p.kt0;
    // (ONE descriptor from selected scalars: a select between two descriptors becomes a branch whose join drains the ring)
    const void* abase = second ? p.a1 : p.a;
    const unsigned abytes = second ? p.a1_bytes : p.a_bytes;
    const int c = second ? p.c1 : p.c0;
    const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(abase), 0, abytes, 0x00020000);
    const int soff = (second ? ks - p.kt0 : ks) * 128;
    char* stage = smem + slot * G::STAGE;
    const unsigned v = second ? ad.a1 : ad.a0;
    const int row = wave * 8 + (lane >> 3);
#pragma unroll
    for (int it = (LO > 0 ? LO : 0); it < (HI < G::A_IT ? HI : G::A_IT); ++it) {
        const unsigned off = (ad.valid && ad.m0 + row + 64 * it < p.M) ? v : kPwOob;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, (__attribute__((address_space(3))) void*)(stage + (wave + PW_NW * it) * 1024), 16, off,
                                                 soff + it * 128 * c, 0, 0);
    }
}
template <int TN, int LO = 0, int HI = 1 << 20>
__device__ __forceinline__ void pw_issue_w(const LinPWParams& p, char* smem, int slot, int wave, int ks, const PwAddr& ad) {
    using G = PwGeo<TN>;
    const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.w), 0, p.w_bytes, 0x00020000);
    const int soff = ks * 128;
    char* stage = smem + slot * G::STAGE + G::A_SLOT;
#pragma unroll
    for (int it = (LO > 0 ? LO : 0); it < (HI < G::W_IT ? HI : G::W_IT); ++it) {
        // (n_pad is a multiple of 64 = the row stride of the pieces: a piece is inside the packed weight or outside as a whole)
        const unsigned off = (ad.valid && ad.n0 + 64 * it < p.n_pad) ? ad.w : kPwOob;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (__attribute__((address_space(3))) void*)(stage + (wave + PW_NW * it) * 1024), 16, off,
                                                 soff + it * 128 * p.K, 0, 0);
    }
}

// pieces [LO, HI) of a step's P = A_IT + W_IT (activation pieces first)
template <int TN, int LO, int HI>
__device__ __forceinline__ void pw_issue_group(const LinPWParams& p, char* smem, int slot, int wave, int lane, int ks, const PwAddr& ad) {
    using G = PwGeo<TN>;
    if constexpr (LO < HI && LO < G::A_IT) pw_issue_a<TN, LO, HI>(p, smem, slot, wave, lane, ks, ad);
    if constexpr (LO < HI && HI > G::A_IT) pw_issue_w<TN, LO - G::A_IT, HI - G::A_IT>(p, smem, slot, wave, ks, ad);
}

// L2 PREFETCH of the activation rows of a K-step a few steps ahead of the ring (round 6).  The ring lives in LDS: a step's pieces are
// requested ONE step before they are needed, and activation rows that come from HBM / the Infinity Cache take longer than that under
// load (tools/gemm_diag.sh: fabric read latency ~ 800 clocks on average, 27 % of the L2 requests miss).  A "touch" is a 4-byte LDS-DMA
// load per 64-byte half line into a dummy LDS word: it costs one VMEM instruction per wave and step, no register, and pulls the line
// into the XCD's L2, where the ring's request two steps later finds it.  VMEM returns in order, so the touch is issued at the TOP of a
// step: the step's counted wait (for the pieces issued one step earlier, which are OLDER) leaves it in flight -- vmcnt(1) -- and it has
// 1.75 steps to return before the next wait needs it gone.
template <int TN>
__device__ __forceinline__ void pw_touch_a(const LinPWParams& p, char* smem, int wave, int lane, int tm, int ks, bool valid) {
    using G = PwGeo<TN>;
    const bool second = ks >= p.kt0;
    const void* abase = second ? p.a1 : p.a;
    const unsigned abytes = second ? p.a1_bytes : p.a_bytes;
    const int c = second ? p.c1 : p.c0;
    const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(abase), 0, abytes, 0x00020000);
    const int m = tm * PW_BM + wave * 32 + (lane >> 1);
    const unsigned off = (valid && m < p.M) ? ((unsigned)m * (unsigned)c * 2u + (unsigned)(lane & 1) * 64u) : kPwOob;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, (__attribute__((address_space(3))) void*)(smem + G::SLAB + 1792), 4, off,
                                             (second ? ks - p.kt0 : ks) * 128, 0, 0);
}

template <int TN>
__device__ __forceinline__ void pw_offsets(const LinPWParams& p, bool valid, int tm, int tn, int wave, int lane, PwAddr& ad) {
    using G = PwGeo<TN>;
    const int slot = lane & 7, rsub = lane >> 3;
    const int row = wave * 8 + rsub;
    const unsigned chunk = (unsigned)((slot ^ ((row >> 1) & 7)) * 8);
    const int m = tm * PW_BM + row, n = tn * G::BN + row;
    ad.a0 = ((unsigned)m * (unsigned)p.c0 + chunk) * 2u;
    ad.a1 = ((unsigned)m * (unsigned)p.c1 + chunk) * 2u;
    ad.w = ((unsigned)n * (unsigned)p.K + chunk) * 2u;
    ad.m0 = tm * PW_BM;
    ad.n0 = tn * G::BN;
    ad.valid = valid;
}

// byte offset (into the torch-layout bias) of packed columns 4t .. 4t+3 of tile column tn, out of range past the tile / the layer
template <int TN>
__device__ __forceinline__ unsigned pw_bias_off(const LinPWParams& p, bool geglu, bool valid, int tn, int t) {
    using G = PwGeo<TN>;
    const int pc = tn * G::BN + 4 * t;
    int oc = pc;
    if (geglu) {
        const int blk = pc >> 5, w = pc & 31;
        oc = ((blk & 1) ? p.n_dst : 0) + (blk >> 1) * 32 + w;
    }
    return (valid && 4 * t < G::BN && pc < p.n_out) ? (unsigned)oc * 4u : kPwOob;
}

// bias of the BN packed columns of tile column tn: thread t fetches packed columns 4t .. 4t+3 (zeros past the tile / without a bias)
template <int TN>
__device__ __forceinline__ u32x4 pw_load_bias(const LinPWParams& p, bool geglu, bool valid, int tn, int t) {
    using G = PwGeo<TN>;
    const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.bias), 0, p.bias_bytes, 0x00020000);
    const int pc = tn * G::BN + 4 * t;                          // packed column
    int oc = pc;                                                // column of the torch-layout bias
    if (geglu) {
        const int blk = pc >> 5, w = pc & 31;
        oc = ((blk & 1) ? p.n_dst : 0) + (blk >> 1) * 32 + w;
    }
    const unsigned off = (valid && 4 * t < G::BN && pc < p.n_out) ? (unsigned)oc * 4u : kPwOob;
    return __builtin_amdgcn_raw_buffer_load_b128(rb, off, 0, 0);
}

// two 16-byte stores the compiler's wait-count bookkeeping does not see (header).  Every asm VMEM statement opens with `s_nop 4`: its
// descriptor lives in SGPRs that hipcc may have just reloaded from a spill lane (v_readlane = a VALU write of an SGPR), and a VMEM
// instruction reading such an SGPR within 5 wait states is a hazard nobody pads inside an asm string (found as wrong tiles in the
// kernels with SGPR spills).  Store data is read at issue on gfx9, but over several cycles: a VALU write of the data registers within 2
// wait states of a > 8-byte store corrupts it -- the closing nop.
template <bool NT>
__device__ __forceinline__ void pw_store2(const u32x4& rdst, const u32x4& d0, unsigned o0, const u32x4& d1, unsigned o1) {
    if constexpr (NT)
        asm volatile("s_nop 4\n\tbuffer_store_dwordx4 %0, %1, %4, 0 offen nt\n\tbuffer_store_dwordx4 %2, %3, %4, 0 offen nt\n\ts_nop 2"
                     ::"v"(d0), "v"(o0), "v"(d1), "v"(o1), "s"(rdst) : "memory");
    else
        asm volatile("s_nop 4\n\tbuffer_store_dwordx4 %0, %1, %4, 0 offen\n\tbuffer_store_dwordx4 %2, %3, %4, 0 offen\n\ts_nop 2"
                     ::"v"(d0), "v"(o0), "v"(d1), "v"(o1), "s"(rdst) : "memory");
}

template <typename T> __device__ __forceinline__ typename PwMma<T>::Frag pw_frag(const char* p) {
    return *reinterpret_cast<const typename PwMma<T>::Frag*>(p);
}

// ---- whole-line epilogue traffic (round 6) ---------------------------------------------------------------------------------------------
// The transposed product leaves a lane with ONE row and, per 32-column block, two 16-byte chunks of it; the two lanes of a row (hi = 0 / 1)
// interleave, so a store instruction writes 32 rows x 32 contiguous bytes -- four requests per 128-byte line, and tools/pw_trace.py showed
// the XCD's L2 taking a tile's 164 KB of stores (and as many residual bytes) at REQUEST rate: 15 - 28 k cycles per tile next to 3.7 k per
// K-step.  For a PAIR of blocks (j even, j + 1) a lane holds chunks C0..C3 = chunk indices hi, 2 + hi, 4 + hi, 6 + hi of its row's 128-byte
// segment.  The four lanes of a quad (four consecutive rows, same hi) TRANSPOSE the 4 x 4 (chunk, lane) matrix -- lane k ends with chunk
// k of rows 0..3, D_t(k) = C_k(lane t) -- so that store instruction t writes, per quad, chunks 2k + hi of row t: with both halves 8
// consecutive chunks = one whole line per row, 8 lines per instruction.  Two butterfly stages by DPP quad permutes (lane ^ 1, lane ^ 2), one
// select per register each.  The transpose is its own inverse: residual chunks loaded with the same line-shaped addressing go through it
// once to reach the accumulator layout.
template <int CTRL> __device__ __forceinline__ u32x4 pw_qperm(const u32x4& v) {
    u32x4 r;
#pragma unroll
    for (int c = 0; c < 4; ++c) r[c] = (unsigned)__builtin_amdgcn_update_dpp(0, (int)v[c], CTRL, 0xF, 0xF, true);
    return r;
}
__device__ __forceinline__ u32x4 pw_sel(bool take_a, const u32x4& a, const u32x4& b) {
    u32x4 r;
#pragma unroll
    for (int c = 0; c < 4; ++c) r[c] = take_a ? a[c] : b[c];
    return r;
}
__device__ __forceinline__ void pw_quad_transpose(u32x4& c0, u32x4& c1, u32x4& c2, u32x4& c3, bool odd, bool upper) {
    constexpr int X1 = 0xB1, X2 = 0x4E;        // quad_perm [1,0,3,2] / [2,3,0,1]
    const u32x4 x0 = pw_sel(odd, pw_qperm<X1>(c1), c0), x1 = pw_sel(odd, c1, pw_qperm<X1>(c0));
    const u32x4 x2 = pw_sel(odd, pw_qperm<X1>(c3), c2), x3 = pw_sel(odd, c3, pw_qperm<X1>(c2));
    c0 = pw_sel(upper, pw_qperm<X2>(x2), x0);
    c2 = pw_sel(upper, x2, pw_qperm<X2>(x0));
    c1 = pw_sel(upper, pw_qperm<X2>(x3), x1);
    c3 = pw_sel(upper, x3, pw_qperm<X2>(x1));
}
// byte offset (row-major, `ld` elements per row) at which lane (l31, hi) stores / loads transposed chunk t = 2 (j & 1) + half of the block
// pair (j & ~1, j | 1) of row block i: row (quad's first row) + t, chunk 2 (l31 & 3) + hi of the pair's 128-byte segment
__device__ __forceinline__ unsigned pw_line_off(int M, int tm, int wm, int i, int j, int half, int l31, int hi, int col0, int ld, int n_dst) {
    const int t = 2 * (j & 1) + half, k = l31 & 3;
    const int m = tm * PW_BM + wm * 64 + i * 32 + (l31 & ~3) + t;
    const int col = col0 + 32 * (j & ~1) + 8 * (2 * k + hi);
    return (m < M && col < n_dst) ? ((unsigned)m * (unsigned)ld + (unsigned)col) * 2u : kPwOob;
}
__device__ __forceinline__ unsigned pw_off(unsigned row, int col, int n_dst) {
    return (row != kPwRowNone && col < n_dst) ? row + (unsigned)col * 2u : kPwOob;
}

// Walks the tiles of a workgroup (all wave-uniform).  XCD x owns row blocks [x * m_per, (x+1) * m_per) and all column tiles.  Its tiles
// form ONE list in block order -- gm x gn blocks of tiles, column chunks (nbn of them) fastest, inside a block the row fastest; ragged
// blocks at the edges are packed densely -- and its wgx workgroups take list entries lid, lid + wgx, lid + 2 wgx ...: the workgroups an XCD
// runs at the same time share about gm activation row blocks and gn weight panels in its L2 (with one row of 32 column tiles in flight the
// 6.5 MB weight of the level-1 GEGLU projection streamed through the 4 MB L2 once per row block), and a round leaves no CU idle unless
// the list ends.  (Rounds 4-5 walked whole blocks, one per round: a block shape that did not divide the XCD's tile grid idled workgroups
// in EVERY round -- 27 of 32 on the 8 x 8 level QKV, 8 rounds for 6.75 rounds of work.)
struct PwTileIter {
    int r, tm, tn;
    bool valid;
    __device__ __forceinline__ void set(const LinPWParams& p, int r0, int lid, int m_lo, int m_cnt) {
        r = r0;
        const int i = r0 * p.wgx + lid;
        valid = i < m_cnt * p.tiles_n;
        if (valid) {
            const int strip = p.gm * p.tiles_n;                          // tiles of a full strip of gm row blocks
            const int sm = min(i / strip, (m_cnt + p.gm - 1) / p.gm - 1);
            const int hm = min(p.gm, m_cnt - sm * p.gm);                 // rows of this strip (the last one may be lower)
            const int is = i - sm * strip;
            const int cn = is / (hm * p.gn);                             // column chunk (the last one may be narrower)
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
__device__ __forceinline__ void pw_pack(const float (&c)[16], float scale, const u32x4 (&res)[2], u32x4 (&out)[2]) {
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

template <typename T, int TN, int EPI, bool RES, bool NT>
__global__ __launch_bounds__(512) void linear_pw_kernel(const LinPWParams p) {
    using G = PwGeo<TN>;
    using Frag = typename PwMma<T>::Frag;
    constexpr bool GEGLU = EPI == MVLDM_EPI_GEGLU;
    static_assert(!GEGLU || TN % 2 == 0, "GEGLU pairs value / gate blocks inside a wave tile");
    static_assert(!(GEGLU && RES), "no caller");
    constexpr int NOUT = GEGLU ? TN / 2 : TN;      // output blocks per row block of a wave
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;        // 4 x 2 waves of 64 rows x 32*TN columns
    const int hi = lane >> 5, l31 = lane & 31;
    const int prm = pw_perm(l31);

    const int xcd = blockIdx.x & 7, lid = blockIdx.x >> 3;
    const int m_lo = xcd * p.m_per, m_cnt = min(p.tiles_m, m_lo + p.m_per) - m_lo;
    PwTileIter cur, nxt, iss;                    // compute side, the tile after it, issue side (newest ring step in flight)
    cur.set(p, 0, lid, m_lo, m_cnt);
    if (!cur.valid) return;
    nxt.set(p, cur.r + 1, lid, m_lo, m_cnt);
    iss = cur;

    u32x4 rdst;
    rdst[0] = (unsigned)(uintptr_t)p.dst; rdst[1] = (unsigned)((uintptr_t)p.dst >> 32) & 0xFFFFu; rdst[2] = p.dst_bytes; rdst[3] = 0x00020000u;

    // fragment read offsets inside a ring slot (bytes): row * 128 + swizzled chunk of k-sub-step 0; sub-step kk: XOR kk << 5
    const int a_off = (wm * 64 + l31) * 128 + ((hi ^ ((l31 >> 1) & 7)) << 4);
    const int w_off = G::A_SLOT + (wn * 32 * TN + prm) * 128 + ((hi ^ ((prm >> 1) & 7)) << 4);

    f32x16 acc[2][TN];
    PwAddr ad;
    int ks_i = 1;                                // issue side: K-step of the newest ring step in flight
    const int kT = p.k_steps;

#ifdef MVLDM_PW_TRACE
    // (trace builds only, MVLDM_PW_STAGGER = n: workgroup lid starts (lid & 3) * n * 512 cycles late -- does a tile's epilogue get shorter
    //  when the workgroups of an XCD do not reach it together?)
    for (int i = 0; i < (lid & 3) * p.trace_stagger; ++i) __builtin_amdgcn_s_sleep(8);
#endif
    // ---- prologue: steps 0 and 1 of the first tile (the whole ring), its bias slab ----
    {
        pw_offsets<TN>(p, true, cur.tm, cur.tn, wave, lane, ad);
#pragma unroll
        for (int g = 0; g < 2; ++g) {
            pw_issue_a<TN>(p, smem, g, wave, lane, g, ad);
            pw_issue_w<TN>(p, smem, g, wave, g, ad);
        }
        const u32x4 b = pw_load_bias<TN>(p, GEGLU, true, cur.tn, tid);
        __builtin_amdgcn_s_waitcnt(pw_wait(0));
        if (4 * tid < G::BN) *reinterpret_cast<u32x4*>(smem + G::SLAB + tid * 16) = b;
    }
    int rs = 0;                                  // ring slot the current step reads
#ifdef MVLDM_PW_TRACE
    const bool tr_on = p.trace && (int)blockIdx.x == p.trace_blk && wave == p.trace_wave;
    int tr_n = 0;
#endif

// One K-step (64 of K) = four sub-steps of 16.  A sub-step runs the 2 * TN MFMAs of its fragments column by column (two row blocks per
// W fragment) and, between the columns, fetches the NEXT sub-step's fragments one column ahead -- a W fragment's registers are free as
// soon as its two MFMAs are issued, so about TN + 1 W fragments are live instead of two full sets (two sets beside 160 accumulators
// spilled).  In front of sub-step 3 every fragment of step g is in registers: the wave waits for step g+1's pieces (WAIT_) and its own
// LDS reads, the barrier publishes step g+1 and RETIRES step g's slot, and sub-step 3 refills it at once with the pieces of step g+2
// (activation rows after column 0, weight rows after column 2) while it fetches the kk = 0 fragments of step g+1 under its MFMAs.
#define PW_SUB(ca_, cw_, na_, nw_, NEXT_, nslot_, nkk_, MODE_, SPN_)                                              \
    {                                                                                                             \
        const char* st_ = smem + (nslot_) * G::STAGE;                                                             \
        const int ao_ = a_off ^ ((nkk_) << 5), wo_ = w_off ^ ((nkk_) << 5);                                       \
        if (NEXT_) {                                                                                              \
            na_[0] = pw_frag<T>(st_ + ao_);                                                                       \
            na_[1] = pw_frag<T>(st_ + ao_ + 4096);                                                                \
            nw_[0] = pw_frag<T>(st_ + wo_);                                                                       \
        }                                                                                                         \
        _Pragma("unroll") for (int j = 0; j < TN; ++j) {                                                          \
            __builtin_amdgcn_sched_barrier(0);                                                                    \
            acc[0][j] = PwMma<T>::mma(cw_[j], ca_[0], acc[0][j]);                                                 \
            acc[1][j] = PwMma<T>::mma(cw_[j], ca_[1], acc[1][j]);                                                 \
            if (NEXT_ && j + 1 < TN) nw_[j + 1] = pw_frag<T>(st_ + wo_ + (j + 1) * 4096);                         \
            if ((MODE_) == 1 && j == 0) {                                                                         \
                if (++ks_i == kT) {                                                                               \
                    ks_i = 0;                                                                                     \
                    iss.set(p, iss.r + 1, lid, m_lo, m_cnt);                                                  \
                    pw_offsets<TN>(p, iss.valid, iss.tm, iss.tn, wave, lane, ad);                                 \
                }                                                                                                 \
                if (SPN_) pw_issue_group<TN, 0, G::Q0>(p, smem, rs, wave, lane, ks_i, ad);                        \
                else pw_issue_a<TN>(p, smem, rs, wave, lane, ks_i, ad);                                           \
            }                                                                                                     \
            if ((MODE_) == 1 && j == 2 && !(SPN_)) pw_issue_w<TN>(p, smem, rs, wave, ks_i, ad);                   \
            if ((MODE_) == 2 && j == 1 && pend) pw_issue_group<TN, G::Q0, G::Q1>(p, smem, rs ^ 1, wave, lane, ks_i, ad); \
            if ((MODE_) == 3 && j == 1 && pend) pw_issue_group<TN, G::Q1, G::Q2>(p, smem, rs ^ 1, wave, lane, ks_i, ad); \
            if ((MODE_) == 4 && j == 1 && pend) pw_issue_group<TN, G::Q2, G::P>(p, smem, rs ^ 1, wave, lane, ks_i, ad);  \
        }                                                                                                         \
        __builtin_amdgcn_sched_barrier(0);                                                                        \
    }
// SPREAD ISSUE (round 6; tile bit 14 switches it off for A/B).  The P = 8 / 9 pieces a wave requests per step (step g+2's, into the slot the
// barrier of step g has just retired) used to go out in one burst behind that barrier: 8 waves x P LDS-DMA instructions at once, each holding
// its wave in the address path for 70 - 75 cycles while the matrix pipe waited (the vendor GEMM pays ~ 50 per piece).  Now 4 go
// out behind the barrier and 3 + 2 in sub-steps 0 and 1 of the NEXT step (`pend`); the wait in front of that step's barrier is still vmcnt(0) and
// still covers them, the results are bit-identical.  -4 ... -10 % on the residual Linears, -1 ... -6 % on the QKV projections, GEGLU unchanged
// (profiles/r06_pw_spread_variants.txt: placements 3-3-3 ... 6-1-2 and 0-3-3, which is slower than the burst).  Only between two steps that both
// have those sub-steps free: the last two steps of a tile (next tile's steps 0 / 1, waited for round the epilogue's own loads and stores) keep the
// burst, and so does the touch variant (its vmcnt(1) needs the touch to be the step's youngest VMEM instruction).
#define PW_STEP(LAST_, WAIT_, TOUCH_, SPN_)                                                                       \
    {                                                                                                             \
        PW_STAMP()                                                                                                \
        if ((TOUCH_) && p.touch) {                                                                                \
            /* (ONE call with scalar-selected arguments: two call sites merged into a waterfall loop over the descriptor) */ \
            const int kt_ = ks_t + PW_TOUCH_AHEAD;                                                                \
            const bool own_ = kt_ < kT;                                                                           \
            pw_touch_a<TN>(p, smem, wave, lane, __builtin_amdgcn_readfirstlane(own_ ? cur.tm : nxt.tm),           \
                           __builtin_amdgcn_readfirstlane(own_ ? kt_ : kt_ - kT), own_ || nxt.valid);             \
        }                                                                                                         \
        PW_SUB(fa0, fw0, fa1, fw1, true, rs, 1, 2, false)                                                         \
        PW_SUB(fa1, fw1, fa0, fw0, true, rs, 2, 3, false)                                                         \
        PW_SUB(fa0, fw0, fa1, fw1, true, rs, 3, 4, false)                                                         \
        PW_STAMP()                                                                                                \
        if ((TOUCH_) && p.touch) __builtin_amdgcn_s_waitcnt(kWaitStep1);                                          \
        else __builtin_amdgcn_s_waitcnt(WAIT_);                                                                   \
        PW_STAMP()                                                                                                \
        __builtin_amdgcn_s_barrier();                                                                             \
        PW_STAMP()                                                                                                \
        {                                                                                                         \
            const bool spn_ = (SPN_);                                                                             \
            PW_SUB(fa1, fw1, fa0, fw0, !(LAST_), rs ^ 1, 0, 1, spn_)                                              \
            pend = spn_;                                                                                          \
        }                                                                                                         \
        rs ^= 1;                                                                                                  \
    }
    bool pend = false;                                           // groups 2 and 3 of the newest step's pieces are still to be issued
    const bool sp = p.spread && !p.touch && kT >= 4;
    Frag fa0[2], fw0[TN], fa1[2], fw1[TN];
    constexpr int kWaitLds = 0xC07F;                             // lgkmcnt(0) only
    constexpr int kWaitFirst = pw_wait(4 * NOUT) & ~0x0F00;      // step 1 of the tile has landed: everything but the previous epilogue's 4 * NOUT stores
    constexpr int kWaitStepT = pw_wait(0) & ~0x0F00;             // vmcnt(0) lgkmcnt(0): step g+1 has landed (and every older store)
    constexpr int kWaitStep1 = pw_wait(1) & ~0x0F00;             // ... vmcnt(1): the step's own touch (younger) may be in flight
    constexpr int PW_TOUCH_AHEAD = 4;                            // the ring requests step g+2 in step g: the touch runs two steps ahead of it

    for (; cur.valid; cur = nxt, nxt.set(p, nxt.r + 1, lid, m_lo, m_cnt)) {
        // ---- the tile starts from its bias (slab written in the prologue / the previous epilogue, published by this barrier).  Steps 0
        //      and 1 found their pieces retired by the epilogue's (the prologue's) load wait ----
        __builtin_amdgcn_s_barrier();
        {
            const float* slab = reinterpret_cast<const float*>(smem + G::SLAB) + wn * 32 * TN + 8 * hi;
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int a = 0; a < 4; ++a) {
                    const f32x4 b = *reinterpret_cast<const f32x4*>(slab + 32 * j + 16 * (a >> 1) + 4 * (a & 1));
#pragma unroll
                    for (int e = 0; e < 4; ++e) { acc[0][j][4 * a + e] = b[e]; acc[1][j][4 * a + e] = b[e]; }
                }
        }
        {
            const char* st0 = smem + rs * G::STAGE;
#pragma unroll
            for (int i = 0; i < 2; ++i) fa0[i] = pw_frag<T>(st0 + a_off + i * 4096);
#pragma unroll
            for (int j = 0; j < TN; ++j) fw0[j] = pw_frag<T>(st0 + w_off + j * 4096);
        }
        { const int ks_t = 0; PW_STEP(false, kWaitFirst, false, sp) }
        // (the middle steps carry one touch each: issued at the top of the step, i.e. younger than the pieces the step waits for)
#pragma unroll 1
        for (int ks_t = 1; ks_t < kT - 1; ++ks_t) PW_STEP(false, kWaitStepT, true, sp && ks_t + 2 < kT)
        // Last step: nothing of the next tile is read before the epilogue.  The next tile's bias is requested here (inline asm: consumed
        // in the epilogue behind a counted wait that leaves the ring pieces this step issues in flight -- the epilogue does not drain the ring)
        u32x4 bnext;
        {
            u32x4 rbias;
            rbias[0] = (unsigned)(uintptr_t)p.bias; rbias[1] = (unsigned)((uintptr_t)p.bias >> 32) & 0xFFFFu; rbias[2] = p.bias_bytes; rbias[3] = 0x00020000u;
            const unsigned boff = pw_bias_off<TN>(p, GEGLU, nxt.valid, nxt.tn, tid);
            asm volatile("s_nop 4\n\tbuffer_load_dwordx4 %0, %1, %2, 0 offen" : "=v"(bnext) : "v"(boff), "s"(rbias));
        }
        { const int ks_t = 0; PW_STEP(true, kWaitLds, false, false) }
        // ---- epilogue: straight from the accumulators (header) ----
        PW_STAMP()
        {
            const int col0 = GEGLU ? (cur.tn * G::BN + wn * 32 * TN) >> 1 : cur.tn * G::BN + wn * 32 * TN;
            const bool q_odd = (l31 & 1) != 0, q_upper = (l31 & 2) != 0;      // this lane's place in its quad (pw_quad_transpose)
            unsigned row_dst[2], row_res[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int m = cur.tm * PW_BM + wm * 64 + i * 32 + l31;
                row_dst[i] = m < p.M ? (unsigned)m * (unsigned)p.dst_ld * 2u : kPwRowNone;
                row_res[i] = m < p.M ? (unsigned)m * (unsigned)p.n_dst * 2u : kPwRowNone;
            }
            u32x4 out[2][NOUT][2];
            if constexpr (RES) {
                // Residual rows: a ROLLING window of PW_D blocks of inline-asm loads (two 16-byte chunks per 32 x 32 block), so that
                // the 160 accumulators + the window fit the register file (all 20 chunks of a 256 x 320 tile up front spilled loop
                // invariants into the K loop).  Loads only are in flight here (stores last), VMEM returns in order: the wait in
                // front of block b leaves the younger blocks of the window outstanding.  The next tile's bias rides in front.
                constexpr int NB = 2 * TN;
                u32x4 rres;
                rres[0] = (unsigned)(uintptr_t)p.residual; rres[1] = (unsigned)((uintptr_t)p.residual >> 32) & 0xFFFFu; rres[2] = p.res_bytes; rres[3] = 0x00020000u;
                u32x4 r[NB][2];
// (a block of a PAIR is loaded line-shaped -- pw_line_off -- and transposed into the accumulator layout when its partner has landed; the odd
//  last block of a TN = 5 wave tile keeps the 32-byte form)
#define PW_RES_OFFS(b_)                                                                                                         \
        const bool pair_ = (((b_) % TN) | 1) < TN;                                                                              \
        const unsigned o0_ = pair_ ? pw_line_off(p.M, cur.tm, wm, (b_) / TN, (b_) % TN, 0, l31, hi, col0, p.n_dst, p.n_dst)     \
                                   : pw_off(row_res[(b_) / TN], col0 + 32 * ((b_) % TN) + 8 * hi, p.n_dst);                     \
        const unsigned o1_ = pair_ ? pw_line_off(p.M, cur.tm, wm, (b_) / TN, (b_) % TN, 1, l31, hi, col0, p.n_dst, p.n_dst)     \
                                   : pw_off(row_res[(b_) / TN], col0 + 32 * ((b_) % TN) + 16 + 8 * hi, p.n_dst);
#define PW_RES_ISSUE(b_)                                                                                                        \
    if constexpr ((b_) < NB) {                                                                                                  \
        PW_RES_OFFS(b_)                                                                                                         \
        asm volatile("s_nop 4\n\tbuffer_load_dwordx4 %0, %2, %4, 0 offen\n\tbuffer_load_dwordx4 %1, %3, %4, 0 offen"             \
                     : "=&v"(r[(b_) < NB ? (b_) : 0][0]), "=&v"(r[(b_) < NB ? (b_) : 0][1]) : "v"(o0_), "v"(o1_), "s"(rres));   \
    }
#define PW_RES_BLOCK(b_)                                                                                                        \
    if constexpr ((b_) < NB) {                                                                                                  \
        constexpr int i_ = (b_) / TN, j_ = (b_) % TN;                                                                           \
        constexpr int left_ = NB - 1 - (b_) < PW_D - 1 ? NB - 1 - (b_) : PW_D - 1;                                              \
        if constexpr ((b_) == 0) {                                                                                              \
            asm volatile("s_waitcnt vmcnt(%3)" : "+v"(r[0][0]), "+v"(r[0][1]), "+v"(bnext) : "n"(2 * left_));                  \
            if (4 * tid < G::BN) *reinterpret_cast<u32x4*>(smem + G::SLAB + tid * 16) = bnext;                                  \
        } else {                                                                                                                \
            asm volatile("s_waitcnt vmcnt(%2)" : "+v"(r[(b_) < NB ? (b_) : 0][0]), "+v"(r[(b_) < NB ? (b_) : 0][1]) : "n"(2 * left_)); \
        }                                                                                                                       \
        constexpr bool paired_ = (j_ | 1) < TN;                                                                                 \
        if constexpr (!paired_) {                                                                                               \
            float c_[16];                                                                                                       \
            _Pragma("unroll") for (int k = 0; k < 16; ++k) c_[k] = acc[i_][j_][k];                                              \
            pw_pack<T, true>(c_, p.out_scale, r[(b_) < NB ? (b_) : 0], out[i_][j_]);                                            \
        } else if constexpr ((j_ & 1) == 1) {                                                                                   \
            /* the pair (b - 1, b) has landed: residual -> accumulator layout, both blocks packed, outputs -> line layout */    \
            constexpr int bp_ = (b_) > 0 ? (b_) - 1 : 0;                                                                        \
            pw_quad_transpose(r[bp_][0], r[bp_][1], r[(b_) < NB ? (b_) : 0][0], r[(b_) < NB ? (b_) : 0][1], q_odd, q_upper);    \
            float c_[16];                                                                                                       \
            _Pragma("unroll") for (int k = 0; k < 16; ++k) c_[k] = acc[i_][j_ - 1][k];                                          \
            pw_pack<T, true>(c_, p.out_scale, r[bp_], out[i_][j_ - 1]);                                                         \
            _Pragma("unroll") for (int k = 0; k < 16; ++k) c_[k] = acc[i_][j_][k];                                              \
            pw_pack<T, true>(c_, p.out_scale, r[(b_) < NB ? (b_) : 0], out[i_][j_]);                                            \
            pw_quad_transpose(out[i_][j_ - 1][0], out[i_][j_ - 1][1], out[i_][j_][0], out[i_][j_][1], q_odd, q_upper);          \
        }                                                                                                                       \
        PW_RES_ISSUE((b_) + PW_D)                                                                                               \
    }
                constexpr int PW_D = 4;
                PW_RES_ISSUE(0) PW_RES_ISSUE(1) PW_RES_ISSUE(2) PW_RES_ISSUE(3)
                PW_RES_BLOCK(0) PW_RES_BLOCK(1) PW_RES_BLOCK(2) PW_RES_BLOCK(3) PW_RES_BLOCK(4)
                PW_RES_BLOCK(5) PW_RES_BLOCK(6) PW_RES_BLOCK(7) PW_RES_BLOCK(8) PW_RES_BLOCK(9)
                static_assert(NB <= 10, "blocks listed above");
#undef PW_RES_ISSUE
#undef PW_RES_BLOCK
            } else {
                // (the bias is older than the P ring pieces the last step issued: they stay in flight)
                asm volatile("s_waitcnt vmcnt(%1)" : "+v"(bnext) : "n"(G::P));
                // (every wave has read the current slab long ago: at its step 0, barriers since)
                if (4 * tid < G::BN) *reinterpret_cast<u32x4*>(smem + G::SLAB + tid * 16) = bnext;
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < NOUT; ++j) {
                        float c[16];
                        if constexpr (GEGLU) {
#pragma unroll
                            for (int k = 0; k < 16; ++k) c[k] = acc[i][2 * j][k] * PW_GELU(acc[i][2 * j + 1][k]);
                        } else {
#pragma unroll
                            for (int k = 0; k < 16; ++k) c[k] = acc[i][j][k];
                        }
                        const u32x4 none[2] = {u32x4{0u, 0u, 0u, 0u}, u32x4{0u, 0u, 0u, 0u}};
                        pw_pack<T, false>(c, p.out_scale, none, out[i][j]);
                        if ((j & 1) == 1) pw_quad_transpose(out[i][j - 1][0], out[i][j - 1][1], out[i][j][0], out[i][j][1], q_odd, q_upper);
                    }
            }
            // stores last: no load is waited for while they are in flight (header)
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < NOUT; ++j)
                    if ((j | 1) < NOUT)      // a pair: whole 128-byte row segments (the chunks were transposed inside the quad)
                        pw_store2<NT>(rdst, out[i][j][0], pw_line_off(p.M, cur.tm, wm, i, j, 0, l31, hi, col0, p.dst_ld, p.n_dst), out[i][j][1],
                                      pw_line_off(p.M, cur.tm, wm, i, j, 1, l31, hi, col0, p.dst_ld, p.n_dst));
                    else
                        pw_store2<NT>(rdst, out[i][j][0], pw_off(row_dst[i], col0 + 32 * j + 8 * hi, p.n_dst), out[i][j][1],
                                      pw_off(row_dst[i], col0 + 32 * j + 16 + 8 * hi, p.n_dst));
        }
    }
    // (the ring pieces issued past the last tile are out of range: zeros into slots nobody reads; nothing to drain but the stores,
    //  which the end of the program waits for)
#ifdef MVLDM_PW_TRACE
    if (tr_on) {
        PW_STAMP()
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        for (int i = lane; i < 192; i += 64) p.trace[i] = i < tr_n ? reinterpret_cast<const unsigned*>(smem + G::SLAB + 1280)[i] : 0u;
    }
#endif
#undef PW_SUB
#undef PW_STEP
}

bool linear_pw_applicable(const mvldm_igemm_desc& d) {
    if (d.act_dtype == MVLDM_F32 || d.dst_dtype != d.act_dtype) return false;
    if (d.ksize != 1 || d.stride != 1 || d.upsample != 0 || d.row_bias || d.k_order != 1 || d.splitk > 1) return false;
    if (d.h_in != d.h_out || d.w_in != d.w_out || d.pad != 0) return false;
    if (d.epilogue != MVLDM_EPI_NONE && d.epilogue != MVLDM_EPI_GEGLU) return false;
    if (d.epilogue != MVLDM_EPI_NONE && d.residual) return false;
    const int n_dst = d.epilogue == MVLDM_EPI_GEGLU ? d.n_out / 2 : d.n_out;
    const int dst_ld = d.dst_ld > 0 ? d.dst_ld : n_dst;
    if ((d.c1 == 0) != (d.src1 == nullptr) || d.c0 % 64 || d.c1 % 64 || d.c0 + d.c1 < 320 || d.k_pad != d.c0 + d.c1 || d.n_out % 8 || n_dst % 8 || dst_ld % 8 || dst_ld < n_dst) return false;
    if (d.epilogue == MVLDM_EPI_GEGLU && d.n_out % 64) return false;
    if (d.bias && ((uintptr_t)d.bias % 16)) return false;
    if (((uintptr_t)d.dst % 16) || (d.residual && ((uintptr_t)d.residual % 16))) return false;
    const double m = (double)d.n_img * d.h_out * d.w_out;
    return m * d.c0 * 2.0 < 4.0e9 && m * d.c1 * 2.0 < 4.0e9 && (double)d.n_pad * d.k_pad * 2.0 < 4.0e9 && m * dst_ld * 2.0 < 4.0e9 && m * n_dst * 2.0 < 4.0e9;
}

template <typename T, int TN, int EPI, bool RES, bool NT> static int linear_pw_launch1(const LinPWParams& p, int grid, hipStream_t s) {
    static std::atomic<uint64_t> done{0};
    if (int rc0 = ensure_dyn_smem(reinterpret_cast<const void*>(linear_pw_kernel<T, TN, EPI, RES, NT>), PwGeo<TN>::SMEM, done)) return rc0;
    hipLaunchKernelGGL((linear_pw_kernel<T, TN, EPI, RES, NT>), dim3(grid), dim3(512), PwGeo<TN>::SMEM, s, p);
    return check_launch();
}
template <typename T, int TN, int EPI, bool RES> static int linear_pw_launch(const LinPWParams& p, int grid, hipStream_t s) {
    return p.nt_store ? linear_pw_launch1<T, TN, EPI, RES, true>(p, grid, s) : linear_pw_launch1<T, TN, EPI, RES, false>(p, grid, s);
}

int linear_pw_run(const mvldm_igemm_desc& d, hipStream_t s) {
    MVLDM_REQUIRE(linear_pw_applicable(d), "igemm: tile 13 (persistent wide Linear) does not apply to this problem");
    LinPWParams p;
    p.a = d.src0; p.a1 = d.src1; p.w = d.weight; p.bias = d.bias; p.residual = d.residual; p.dst = d.dst;
    p.M = d.n_img * d.h_out * d.w_out; p.K = d.c0 + d.c1; p.c0 = d.c0; p.c1 = d.c1; p.kt0 = d.c0 / 64; p.n_out = d.n_out; p.n_pad = d.n_pad;
    p.n_dst = d.epilogue == MVLDM_EPI_GEGLU ? d.n_out / 2 : d.n_out;
    p.dst_ld = d.dst_ld > 0 ? d.dst_ld : p.n_dst;
    p.k_steps = p.K / 64; p.out_scale = d.out_scale;
    // 256 x 320 when the packed width is a multiple of 320 (every channel count of this UNet), else 256 x 256; GEGLU pairs need an even
    // number of column blocks per wave.  Round 6: ... unless 256 x 256 tiles need clearly fewer MFMA cycles per CU -- a launch lasts
    // rounds x tile area, rounds = ceil(tiles of the XCD / its 32 CUs): the 8 x 8 level's N = 1280 Linears (36 864 rows: 72 tiles of
    // 256 x 320 per XCD = 2.25 rounds, run as 3) take 3 rounds of the SMALLER tile instead (90 tiles = 2.8 rounds), -20 %.
    const bool geglu = d.epilogue == MVLDM_EPI_GEGLU;
    static const int kForceTn = knob_int("MVLDM_PW_TN", 0);
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
    p.tiles_m = (p.M + PW_BM - 1) / PW_BM;
    p.m_per = (p.tiles_m + 7) / 8;
    auto launch_cost = [&](int tnb) {      // MFMA time of the busiest CU, in 256 x 64-column units
        const int tiles = p.m_per * ((d.n_pad + 64 * tnb - 1) / (64 * tnb));
        const int wg = std::min(cu_x, tiles);
        return (double)((tiles + wg - 1) / wg) * tnb;
    };
    int tn_blocks = (!geglu && d.n_pad % 320 == 0) ? 5 : 4;
    if (tn_blocks == 5 && launch_cost(4) < 0.95 * launch_cost(5)) tn_blocks = 4;
    if (kForceTn == 4 || (kForceTn == 5 && !geglu)) tn_blocks = kForceTn;
    const int bn = 64 * tn_blocks;
    p.tiles_n = (d.n_pad + bn - 1) / bn;
    p.a_bytes = (unsigned)((double)p.M * p.c0 * 2.0); p.a1_bytes = (unsigned)((double)p.M * p.c1 * 2.0); p.w_bytes = (unsigned)((double)d.n_pad * d.k_pad * 2.0);
    p.bias_bytes = d.bias ? (unsigned)d.n_out * 4u : 0u;
    p.res_bytes = d.residual ? (unsigned)((double)p.M * p.n_dst * 2.0) : 0u;
    p.dst_bytes = (unsigned)((double)p.M * p.dst_ld * 2.0);
    // write-back stores unless forced (header): MVLDM_STREAM_STORES=1 is the A/B knob
    static const int kNt = knob_int("MVLDM_STREAM_STORES", 0);
    p.nt_store = kNt == 1;
    p.spread = !((d.tile >> 14) & 1);             // bit 14 of `tile` = A/B: issue every step's pieces in one burst behind its barrier (the round-4/5 form)
    p.touch = (d.tile >> 13) & 1;                 // bit 13 of `tile`: the L2 prefetch (a tuner candidate: +4 % on some shapes, -9 % on others)
    p.trace = nullptr; p.trace_blk = 0; p.trace_wave = 0; p.trace_stagger = 0;
#ifdef MVLDM_PW_TRACE
    if (const char* tp = getenv("MVLDM_PW_TRACE_PTR")) {
        p.trace = reinterpret_cast<unsigned*>(strtoull(tp, nullptr, 16));
        p.trace_blk = getenv("MVLDM_PW_TRACE_BLK") ? atoi(getenv("MVLDM_PW_TRACE_BLK")) : 8;
        p.trace_wave = getenv("MVLDM_PW_TRACE_WAVE") ? atoi(getenv("MVLDM_PW_TRACE_WAVE")) : 0;
        p.trace_stagger = getenv("MVLDM_PW_STAGGER") ? atoi(getenv("MVLDM_PW_STAGGER")) : 0;
    }
#endif
    if (kPwFake & 1) p.a_bytes = p.a1_bytes = 0;
    if (kPwFake & 2) p.w_bytes = 0;
    if (kPwFake & 4) p.dst_bytes = 0;
    // An XCD's workgroups (one per CU, fewer when it has fewer tiles) walk its tile list, which is ordered in gm x gn blocks (PwTileIter):
    // the block shape of about one round's tiles that moves the fewest bytes into the XCD's L2 per round -- gm activation row blocks + gn
    // weight panels
    const double a_t = 256.0 * p.K * 2.0, w_t = (double)bn * p.K * 2.0;
    p.wgx = std::min(cu_x, p.m_per * p.tiles_n);
    double best_cost = 1e300;
    p.gm = p.gn = 1;
    for (int gm = 1; gm <= std::min(p.wgx, p.m_per); ++gm) {
        const int gn = std::max(1, std::min(p.wgx / gm, p.tiles_n));
        // (cost per tile of the block: a block smaller than a round shares less)
        const double cost = (gm * a_t + gn * w_t) / (gm * gn);
        if (cost < best_cost) { best_cost = cost; p.gm = gm; p.gn = gn; }
    }
    static const int kForceGm = knob_int("MVLDM_PW_GM", 0);   // tuning: force the block shape
    if (kForceGm > 0) { p.gm = std::min(std::min(kForceGm, p.wgx), p.m_per); p.gn = std::max(1, std::min(p.wgx / p.gm, p.tiles_n)); }
    p.nbn = (p.tiles_n + p.gn - 1) / p.gn;
    const int grid = 8 * p.wgx;
    const bool res = d.residual != nullptr;
    return dispatch_dtype(d.act_dtype, [&](auto t) -> int {
        using T = decltype(t);
        if constexpr (sizeof(T) == 2) {
            if (geglu) return linear_pw_launch<T, 4, MVLDM_EPI_GEGLU, false>(p, grid, s);
            if (tn_blocks == 5) return res ? linear_pw_launch<T, 5, MVLDM_EPI_NONE, true>(p, grid, s) : linear_pw_launch<T, 5, MVLDM_EPI_NONE, false>(p, grid, s);
            return res ? linear_pw_launch<T, 4, MVLDM_EPI_NONE, true>(p, grid, s) : linear_pw_launch<T, 4, MVLDM_EPI_NONE, false>(p, grid, s);
        } else {
            return set_error(MVLDM_ERR_ARG, "igemm: tile 13 needs a 16-bit activation type");
        }
    });
}

}  // namespace mvldm
