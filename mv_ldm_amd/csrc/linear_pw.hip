// Persistent WIDE Linear (1x1 conv over token rows): tile 13 of the implicit-GEMM family (include/mvldm.h: mvldm_igemm_fwd;
// 16-bit activations, one source or the channel concat of two, K a multiple of 64 and >= 320).  Round 4.
//
// What the op tables of rounds 2-3 said about the K = 320 ... 1280 Linears (40 of the 105 ms of a DDIM step at 64 scenes): in the
// one-tile-per-workgroup kernels (igemm.hip) ring fill, main loop and epilogue of a tile ADD UP on a CU that holds one workgroup;
// the LDS park of the epilogue alone (160 ds_write_b32 + 80 ds_read_b128 per wave and tile) is as long as the main loop at
// K = 320.  Tile 12 (linear_pp.hip) hides the epilogue under the next tile's MFMAs but pays for its two accumulator sets with a
// 128-column tile: 85 flop per L2->LDS byte, and these shapes are bound by that fill.  This kernel keeps the WIDE tile
// (256 x 320, or 256 x 256 for GEGLU: 142 / 128 flop per byte) and removes the other two costs:
//   * PERSISTENT: a workgroup walks its output tiles as one stream of K-steps of 32 (BK = 32: 36 KB per step at 256 x 320); the
//     4-slot LDS-DMA ring never drains -- step g+3 is issued at the top of step g, across tile boundaries, so three steps
//     (110 KB) are in flight per CU at any time and the next tile's first three steps land while the epilogue runs;
//   * PARK-FREE EPILOGUE: the product is computed transposed (W fragment = MFMA A operand), so a lane holds ONE output row; the
//     W rows a wave feeds to the MFMA's M index are PERMUTED (mu = 8a + 4h + e  <-  column 16(a>>1) + 8h + 4(a&1) + e of the
//     32-column block), which costs nothing (it is the lane's LDS read address) and leaves accumulator registers 0..7 / 8..15 of
//     a lane = 8 + 8 CONSECUTIVE output columns: two 16-byte stores per 32 x 32 block straight from registers -- no LDS park, no
//     v_permlane swaps; a store instruction covers 32 rows x 32 contiguous bytes.  GEGLU: value and gate blocks use the same
//     permutation, so a lane holds a column's value AND gate.
// Counted waits: every wave issues the same VMEM sequence.  All loads (bias slab, residual) are hipcc-visible builtins consumed
// behind explicit waits; the output stores are inline asm (hipcc treats loads and stores in flight as unordered and would fall back
// to vmcnt(0) around them) and are issued at the END of the epilogue, after the last load has been consumed, so the compiler's
// model and the hardware counter agree wherever a load is waited for.  Steps 0..2 of a tile need no wait (the epilogue's load
// wait retired every older ring piece: VMEM returns in order); from step 3 on vmcnt(2P) leaves two steps in flight.
// LDS rows are 64 bytes (4 x 16-byte chunks), chunk index XOR (row >> 2) & 3: the 16-lane groups of a ds_read_b128 fragment read
// hit 16 distinct 16-byte slots (checked for the natural AND the permuted row order, which keeps the lane groups' row sets).
#include <algorithm>
#include <cstdlib>

#include "common.h"

namespace mvldm {

struct LinPWParams {
    const void* a; const void* a1; const void* w; const float* bias; const void* residual; void* dst;
    int M, K, c0, c1, kt0, n_out, n_pad, n_dst, dst_ld, k_steps;    // K = c0 + c1; K-steps [0, kt0) come from `a`, the rest from `a1`
    int tiles_m, tiles_n, m_per;       // m_per: 256-row blocks per XCD
    int cpt, nch;                      // a unit = up to `cpt` consecutive column tiles of one row block; nch units per row block
    int nt_store;
    float out_scale;
    unsigned a_bytes, a1_bytes, w_bytes, bias_bytes, res_bytes, dst_bytes;
};

static const int kPwCpt = getenv("MVLDM_PW_CPT") ? atoi(getenv("MVLDM_PW_CPT")) : 0;   // tuning: force the unit length
#ifdef MVLDM_EXPERIMENTS
static const int kPwFake = getenv("MVLDM_PW_FAKE") ? atoi(getenv("MVLDM_PW_FAKE")) : 0;   // 1: no A traffic, 2: no W traffic, 4: no stores
#else
static constexpr int kPwFake = 0;
#endif

constexpr unsigned kPwOob = 0xFFFFFFF0u;
constexpr unsigned kPwRowNone = 0xFFFFFFFFu;
constexpr int PW_BM = 256, PW_NW = 8;

template <int TN> struct PwGeo {
    static constexpr int BN = 64 * TN;                  // 2 column waves of TN 32-column blocks
    static constexpr int A_SLOT = PW_BM * 64, W_SLOT = BN * 64, STAGE = A_SLOT + W_SLOT;
    static constexpr int W_PIECES = BN / 16;            // 1 KiB DMA pieces (16 rows of 64 bytes) of a W step
    static constexpr int W_IT = (W_PIECES + PW_NW - 1) / PW_NW;
    static constexpr int P = 2 + W_IT;                  // DMA instructions per wave and step (A: 2)
    static constexpr int RING = 4 * STAGE;
    static constexpr int DUMMY = RING;                  // target of the padding pieces (TN = 5: waves 4..7 issue a third W piece nobody reads)
    static constexpr int SLAB = RING + 1024;            // bias of the tile's BN packed columns
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

// MFMA M index mu (= lane & 31 of the W-fragment read) -> column of the 32-column block it is fed from (header)
__device__ __forceinline__ int pw_perm(int mu) {
    const int a = mu >> 3, h = (mu >> 2) & 1, e = mu & 3;
    return 16 * (a >> 1) + 8 * h + 4 * (a & 1) + e;
}

// (buffer descriptors only in free functions: an opaque __amdgpu_buffer_rsrc_t inside a lambda trips hipcc's host pass)
template <int TN>
__device__ __forceinline__ void pw_issue(const LinPWParams& p, char* smem, int slot, int wave, int ks, const unsigned (&ao)[2][2],
                                         const unsigned (&bo)[3]) {
    using G = PwGeo<TN>;
    const bool second = ks >= p.kt0;
    // (ONE descriptor from selected scalars: a select between two descriptors becomes a branch whose join drains the ring)
    const void* abase = second ? p.a1 : p.a;
    const unsigned abytes = second ? p.a1_bytes : p.a_bytes;
    const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(abase), 0, abytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.w), 0, p.w_bytes, 0x00020000);
    const int soff_a = (second ? ks - p.kt0 : ks) * 64, soff_w = ks * 64;
    char* stage = smem + slot * G::STAGE;
#pragma unroll
    for (int it = 0; it < 2; ++it)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, (__attribute__((address_space(3))) void*)(stage + (wave + PW_NW * it) * 1024), 16,
                                                 second ? ao[1][it] : ao[0][it], soff_a, 0, 0);
#pragma unroll
    for (int it = 0; it < G::W_IT; ++it) {
        const int q = wave + PW_NW * it;
        char* dst = q < G::W_PIECES ? stage + G::A_SLOT + q * 1024 : smem + G::DUMMY;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (__attribute__((address_space(3))) void*)dst, 16, bo[it], soff_w, 0, 0);
    }
}

// per-lane source offsets of this wave's DMA pieces for output tile (tm, tn): piece q covers tile rows 16q .. 16q+15 (64 bytes of K
// each); a lane fetches the 16-byte chunk that belongs at its (linear) LDS position under the XOR swizzle.  valid == false: every
// piece out of range (the ring keeps its cadence past the last tile: zeros into slots nobody reads)
template <int TN>
__device__ __forceinline__ void pw_offsets(const LinPWParams& p, bool valid, int tm, int tn, int wave, int lane, unsigned (&ao)[2][2],
                                           unsigned (&bo)[3]) {
    using G = PwGeo<TN>;
    const int cp = lane & 3, rsub = lane >> 2;
#pragma unroll
    for (int it = 0; it < 2; ++it) {
        const int row = (wave + PW_NW * it) * 16 + rsub;
        const int m = tm * PW_BM + row;
        const unsigned chunk = (unsigned)((cp ^ ((row >> 2) & 3)) * 8);
        ao[0][it] = (valid && m < p.M) ? ((unsigned)m * (unsigned)p.c0 + chunk) * 2u : kPwOob;
        ao[1][it] = (valid && m < p.M) ? ((unsigned)m * (unsigned)p.c1 + chunk) * 2u : kPwOob;
    }
#pragma unroll
    for (int it = 0; it < G::W_IT; ++it) {
        const int q = wave + PW_NW * it;
        const int row = q * 16 + rsub;
        const int n = tn * G::BN + row;
        const unsigned chunk = (unsigned)((cp ^ ((row >> 2) & 3)) * 8);
        bo[it] = (valid && q < G::W_PIECES && n < p.n_pad) ? ((unsigned)n * (unsigned)p.K + chunk) * 2u : kPwOob;
    }
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

__device__ __forceinline__ u32x4 pw_load_res(const LinPWParams& p, unsigned off) {
    const __amdgpu_buffer_rsrc_t rr = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.residual), 0, p.res_bytes, 0x00020000);
    return __builtin_amdgcn_raw_buffer_load_b128(rr, off, 0, 0);
}

// two 16-byte stores the compiler's wait-count bookkeeping does not see (header).  Store data is read at issue on gfx9, but over
// several cycles: a VALU write of the data registers within 2 wait states of a > 8-byte store corrupts it -- the nop is inside
template <bool NT>
__device__ __forceinline__ void pw_store2(const u32x4& rdst, const u32x4& d0, unsigned o0, const u32x4& d1, unsigned o1) {
    if constexpr (NT)
        asm volatile("buffer_store_dwordx4 %0, %1, %4, 0 offen nt\n\tbuffer_store_dwordx4 %2, %3, %4, 0 offen nt\n\ts_nop 2"
                     ::"v"(d0), "v"(o0), "v"(d1), "v"(o1), "s"(rdst) : "memory");
    else
        asm volatile("buffer_store_dwordx4 %0, %1, %4, 0 offen\n\tbuffer_store_dwordx4 %2, %3, %4, 0 offen\n\ts_nop 2"
                     ::"v"(d0), "v"(o0), "v"(d1), "v"(o1), "s"(rdst) : "memory");
}

template <typename T> __device__ __forceinline__ typename PwMma<T>::Frag pw_frag(const char* p) {
    return *reinterpret_cast<const typename PwMma<T>::Frag*>(p);
}

__device__ __forceinline__ unsigned pw_off(unsigned row, int col, int n_dst) {
    return (row != kPwRowNone && col < n_dst) ? row + (unsigned)col * 2u : kPwOob;
}

// walks the tiles of a workgroup (all wave-uniform)
struct PwTileIter {
    int q, tm, tn, left;       // unit, tile coordinates, tiles left in the unit after this one
    bool valid;
    __device__ __forceinline__ void set(const LinPWParams& p, int unit, int m_lo, int n_units) {
        q = unit;
        valid = unit < n_units;
        const int rb = unit / p.nch, ch = unit - rb * p.nch;
        tm = m_lo + rb;
        tn = ch * p.cpt;
        left = min(p.cpt, p.tiles_n - tn) - 1;
    }
    __device__ __forceinline__ void advance(const LinPWParams& p, int wpx, int m_lo, int n_units) {
        if (left > 0) { --left; ++tn; }
        else set(p, q + wpx, m_lo, n_units);
    }
};

// one 32 x 32 block (GEGLU: one value / gate pair) of the finished tile -> two packed 16-byte chunks.  c[k]: the lane's 16 columns
// in output order (registers 0..7 = columns 8h .. 8h+7, 8..15 = 16 + 8h .. of the block).  RESV: residual chunks of the same columns
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

    // tiles of this workgroup (the walk of linear_pp.hip): XCD x owns row blocks [x * m_per, (x+1) * m_per), cut into units of up to
    // `cpt` consecutive column tiles; the XCD's workgroups take its units round-robin, row block major
    const int xcd = blockIdx.x & 7, lid = blockIdx.x >> 3, wpx = gridDim.x >> 3;
    const int m_lo = xcd * p.m_per, m_cnt = min(p.tiles_m, m_lo + p.m_per) - m_lo;
    const int n_units = m_cnt > 0 ? m_cnt * p.nch : 0;
    if (lid >= n_units) return;
    PwTileIter cur, nxt, iss;                    // compute side, the tile after it, issue side (newest ring step in flight)
    cur.set(p, lid, m_lo, n_units);
    nxt = cur; nxt.advance(p, wpx, m_lo, n_units);
    iss = cur;

    u32x4 rdst;
    rdst[0] = (unsigned)(uintptr_t)p.dst; rdst[1] = (unsigned)((uintptr_t)p.dst >> 32) & 0xFFFFu; rdst[2] = p.dst_bytes; rdst[3] = 0x00020000u;

    // fragment read offsets inside a ring slot (bytes): row * 64 + swizzled chunk; k-sub-step kk = 0, 1 of the step's 32 K values
    int a_off[2], w_off[2];
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
        a_off[kk] = (wm * 64 + l31) * 64 + (((kk * 2 + hi) ^ ((l31 >> 2) & 3)) << 4);
        w_off[kk] = G::A_SLOT + (wn * 32 * TN + prm) * 64 + (((kk * 2 + hi) ^ ((prm >> 2) & 3)) << 4);
    }

    f32x16 acc[2][TN];
    unsigned ao[2][2], bo[3];   // (fixed bound: a dependent one in the helpers' signatures fails substitution in hipcc's host pass)
    int ks_i = 2;                                // issue side: K-step of the newest ring step in flight
    const int kT = p.k_steps;

    // ---- prologue: steps 0..2 of the first tile, its bias slab ----
    {
        pw_offsets<TN>(p, true, cur.tm, cur.tn, wave, lane, ao, bo);
        pw_issue<TN>(p, smem, 0, wave, 0, ao, bo);
        pw_issue<TN>(p, smem, 1, wave, 1, ao, bo);
        pw_issue<TN>(p, smem, 2, wave, 2, ao, bo);
        const u32x4 b = pw_load_bias<TN>(p, GEGLU, true, cur.tn, tid);
        __builtin_amdgcn_s_waitcnt(pw_wait(0));
        if (4 * tid < G::BN) *reinterpret_cast<u32x4*>(smem + G::SLAB + tid * 16) = b;
    }
    int rs = 0;                                  // ring slot the current step reads

// top of a step: (from step 3 on) the step's operands have landed -- everything but the two newest steps' pieces --, every wave is
// done with the slot the next pieces go to
#define PW_ISSUE_NEXT()                                                                   \
    {                                                                                     \
        if (++ks_i == kT) {                                                               \
            ks_i = 0;                                                                     \
            iss.advance(p, wpx, m_lo, n_units);                                           \
            pw_offsets<TN>(p, iss.valid, iss.tm, iss.tn, wave, lane, ao, bo);             \
        }                                                                                 \
        pw_issue<TN>(p, smem, (rs + 3) & 3, wave, ks_i, ao, bo);                          \
    }
// the 4 * TN MFMAs of a step (transposed product: the W fragment is the A operand)
#define PW_COMPUTE()                                                                      \
    {                                                                                     \
        const char* st = smem + rs * G::STAGE;                                            \
        _Pragma("unroll") for (int kk = 0; kk < 2; ++kk) {                                \
            Frag fa[2], fw[TN];                                                           \
            _Pragma("unroll") for (int i = 0; i < 2; ++i) fa[i] = pw_frag<T>(st + a_off[kk] + i * 2048);   \
            _Pragma("unroll") for (int j = 0; j < TN; ++j) fw[j] = pw_frag<T>(st + w_off[kk] + j * 2048);  \
            _Pragma("unroll") for (int i = 0; i < 2; ++i) _Pragma("unroll") for (int j = 0; j < TN; ++j)   \
                acc[i][j] = PwMma<T>::mma(fw[j], fa[i], acc[i][j]);                       \
        }                                                                                 \
        rs = (rs + 1) & 3;                                                                \
    }

    for (; cur.valid; cur = nxt, nxt.advance(p, wpx, m_lo, n_units)) {
        // ---- step 0: the tile starts from its bias (slab written in the prologue / the previous epilogue, published by this barrier) ----
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
        PW_ISSUE_NEXT()
        PW_COMPUTE()
        // ---- steps 1, 2: their pieces were retired by the previous epilogue's (the prologue's) load wait ----
        __builtin_amdgcn_s_barrier();
        PW_ISSUE_NEXT()
        PW_COMPUTE()
        __builtin_amdgcn_s_barrier();
        PW_ISSUE_NEXT()
        PW_COMPUTE()
        // ---- steps 3 .. kT-1 ----
#pragma unroll 1
        for (int ks = 3; ks < kT; ++ks) {
            __builtin_amdgcn_s_waitcnt(pw_wait(2 * G::P));
            __builtin_amdgcn_s_barrier();
            PW_ISSUE_NEXT()
            PW_COMPUTE()
        }
        // ---- epilogue: straight from the accumulators (header) ----
        {
            const int col0 = GEGLU ? (cur.tn * G::BN + wn * 32 * TN) >> 1 : cur.tn * G::BN + wn * 32 * TN;
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
                u32x4 rres, rbias;
                rres[0] = (unsigned)(uintptr_t)p.residual; rres[1] = (unsigned)((uintptr_t)p.residual >> 32) & 0xFFFFu; rres[2] = p.res_bytes; rres[3] = 0x00020000u;
                rbias[0] = (unsigned)(uintptr_t)p.bias; rbias[1] = (unsigned)((uintptr_t)p.bias >> 32) & 0xFFFFu; rbias[2] = p.bias_bytes; rbias[3] = 0x00020000u;
                u32x4 bnext, r[NB][2];
                {
                    const int pc = nxt.tn * G::BN + 4 * tid;
                    const unsigned boff = (nxt.valid && 4 * tid < G::BN && pc < p.n_out) ? (unsigned)pc * 4u : kPwOob;
                    asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen" : "=v"(bnext) : "v"(boff), "s"(rbias));
                }
#define PW_RES_ISSUE(b_)                                                                                                        \
    if constexpr ((b_) < NB) {                                                                                                  \
        const unsigned o0_ = pw_off(row_res[(b_) / TN], col0 + 32 * ((b_) % TN) + 8 * hi, p.n_dst);                             \
        const unsigned o1_ = pw_off(row_res[(b_) / TN], col0 + 32 * ((b_) % TN) + 16 + 8 * hi, p.n_dst);                        \
        asm volatile("buffer_load_dwordx4 %0, %2, %4, 0 offen\n\tbuffer_load_dwordx4 %1, %3, %4, 0 offen"                       \
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
        float c_[16];                                                                                                           \
        _Pragma("unroll") for (int k = 0; k < 16; ++k) c_[k] = acc[i_][j_][k];                                                  \
        pw_pack<T, true>(c_, p.out_scale, r[(b_) < NB ? (b_) : 0], out[i_][j_]);                                                \
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
                const u32x4 bnext = pw_load_bias<TN>(p, GEGLU, nxt.valid, nxt.tn, tid);
                __builtin_amdgcn_s_waitcnt(pw_wait(0));
                // (every wave has read the current slab long ago: at its step 0, barriers since)
                if (4 * tid < G::BN) *reinterpret_cast<u32x4*>(smem + G::SLAB + tid * 16) = bnext;
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < NOUT; ++j) {
                        float c[16];
                        if constexpr (GEGLU) {
#pragma unroll
                            for (int k = 0; k < 16; ++k) c[k] = acc[i][2 * j][k] * gelu_erf_fast(acc[i][2 * j + 1][k]);
                        } else {
#pragma unroll
                            for (int k = 0; k < 16; ++k) c[k] = acc[i][j][k];
                        }
                        const u32x4 none[2] = {u32x4{0u, 0u, 0u, 0u}, u32x4{0u, 0u, 0u, 0u}};
                        pw_pack<T, false>(c, p.out_scale, none, out[i][j]);
                    }
            }
            // stores last: no load is waited for while they are in flight (header)
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < NOUT; ++j)
                    pw_store2<NT>(rdst, out[i][j][0], pw_off(row_dst[i], col0 + 32 * j + 8 * hi, p.n_dst), out[i][j][1],
                                  pw_off(row_dst[i], col0 + 32 * j + 16 + 8 * hi, p.n_dst));
        }
    }
    // (the ring pieces issued past the last tile are out of range: zeros into slots nobody reads; nothing to drain but the stores,
    //  which the end of the program waits for)
#undef PW_ISSUE_NEXT
#undef PW_COMPUTE
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
    p.M = d.n_img * d.h_out * d.w_out; p.K = d.c0 + d.c1; p.c0 = d.c0; p.c1 = d.c1; p.kt0 = d.c0 / 32; p.n_out = d.n_out; p.n_pad = d.n_pad;
    p.n_dst = d.epilogue == MVLDM_EPI_GEGLU ? d.n_out / 2 : d.n_out;
    p.dst_ld = d.dst_ld > 0 ? d.dst_ld : p.n_dst;
    p.k_steps = p.K / 32; p.out_scale = d.out_scale;
    // 256 x 320 when the packed width is a multiple of 320 (every channel count of this UNet), else 256 x 256; GEGLU pairs need an even
    // number of column blocks per wave
    const bool geglu = d.epilogue == MVLDM_EPI_GEGLU;
    static const int kForceTn = getenv("MVLDM_PW_TN") ? atoi(getenv("MVLDM_PW_TN")) : 0;
    int tn_blocks = (!geglu && d.n_pad % 320 == 0) ? 5 : 4;
    if (kForceTn == 4 || (kForceTn == 5 && !geglu)) tn_blocks = kForceTn;
    const int bn = 64 * tn_blocks;
    p.tiles_m = (p.M + PW_BM - 1) / PW_BM; p.tiles_n = (d.n_pad + bn - 1) / bn;
    p.m_per = (p.tiles_m + 7) / 8;
    p.a_bytes = (unsigned)((double)p.M * p.c0 * 2.0); p.a1_bytes = (unsigned)((double)p.M * p.c1 * 2.0); p.w_bytes = (unsigned)((double)d.n_pad * d.k_pad * 2.0);
    p.bias_bytes = d.bias ? (unsigned)d.n_out * 4u : 0u;
    p.res_bytes = d.residual ? (unsigned)((double)p.M * p.n_dst * 2.0) : 0u;
    p.dst_bytes = (unsigned)((double)p.M * p.dst_ld * 2.0);
    p.nt_store = stream_stores((size_t)p.M * (size_t)p.n_dst * 2);
    if (kPwFake & 1) p.a_bytes = p.a1_bytes = 0;
    if (kPwFake & 2) p.w_bytes = 0;
    if (kPwFake & 4) p.dst_bytes = 0;
    static int n_cu = 0;
    if (!n_cu) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0)
            n_cu = prop.multiProcessorCount;
        else
            n_cu = 256;
    }
    // workgroups per XCD: one per CU, fewer when the busiest XCD has fewer tiles.  Unit length: the most consecutive column
    // tiles (= L2 hits on the activation rows) for which the row blocks in flight on an XCD -- ceil(wpx / units per row block)
    // + 1 of 256 x K x 2 bytes -- stay within about half of its 4 MB L2
    const int wpx = std::max(1, std::min(n_cu / 8, p.m_per * p.tiles_n));
    const double rb_bytes = 256.0 * p.K * 2.0;
    p.cpt = 1;
    for (int c = p.tiles_n; c >= 1; --c) {
        const int nch = (p.tiles_n + c - 1) / c;
        if (((wpx + nch - 1) / nch + 1) * rb_bytes <= 2.0e6) { p.cpt = c; break; }
    }
    if (kPwCpt > 0) p.cpt = std::min(kPwCpt, p.tiles_n);
    p.nch = (p.tiles_n + p.cpt - 1) / p.cpt;
    p.cpt = (p.tiles_n + p.nch - 1) / p.nch;                    // even units
    const int grid = 8 * wpx;
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
