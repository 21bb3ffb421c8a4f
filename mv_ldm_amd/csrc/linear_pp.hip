// Persistent, epilogue-pipelined Linear (1x1 conv over token rows) for the transformer blocks: tile 12 of the implicit-GEMM
// family (include/mvldm.h: mvldm_igemm_fwd; 16-bit activations, one source or the channel concat of two, K a multiple of 64
// and >= 320).
//
// Why: the FF / QKV / projection GEMMs of the UNet have K = 320 ... 1280, i.e. 5-20 K-tiles per output tile.  In the general
// kernel (igemm.hip) one workgroup owns the CU, so per output tile the ring fill latency, the main loop and the epilogue --
// LDS park, bias / exact-erf GELU / residual arithmetic (VALU), 16-byte stores -- ADD UP, and at K = 320 the epilogue costs as
// much as the main loop (595 TFLOP/s on the level-0 GEGLU projections, profiles/r02_optable_b64.json).
// Here a workgroup is PERSISTENT: it walks a sequence of 256 x 128 output tiles as ONE flat stream of K-tile steps
//   * the 3-slot operand ring never drains: K-tile s+2 is issued (LDS-DMA) at the top of step s, across tile boundaries, and the
//     top of a step waits with a COUNTED vmcnt(6) (everything but the newest step's six DMA pieces), so two steps of HBM / L2
//     latency are covered;
//   * the epilogue of tile t runs in four slices INSIDE the main loop of tile t+1: the finished accumulators stay in registers
//     (two accumulator sets per wave: 2 x 64 VGPRs); steps 1..4 of the next tile each finish one 32 x 32 block between their
//     MFMAs -- MFMA and VALU are separate pipes.  No LDS park: the product is computed TRANSPOSED (W fragment as the MFMA A
//     operand), so a lane holds one output row and 4-column groups of it; v_permlane32_swap pairs the groups of the two
//     half-waves into 8 consecutive columns = one 16-byte store;
//   * nothing in the loop may make the compiler drain the DMA ring: the bias is folded into the accumulator initialisation
//     (a 512-byte slab per tile, fetched by LDS-DMA one tile ahead), residual rows are fetched one step before their use and
//     are older than the ring pieces the counted wait leaves in flight, and the stores are issued (inline asm: hipcc's
//     wait-count pass treats loads and stores in flight together as unordered and would fall back to vmcnt(0)) at the top of
//     the following step, in front of the ring pieces.
// Tile walk: XCD x (= workgroup id % 8) owns a contiguous range of 256-row blocks and walks their column tiles, so every
// activation row block crosses the fabric into ONE L2.
// Same XOR-swizzled 128-byte LDS rows and fragment reads as igemm.hip.  GEGLU: value / gate column blocks are adjacent (the
// packed weight interleaves them by 32): step 1/3 evaluate GELU(gate) of row block 0/1 into registers, step 2/4 the product.
#include <algorithm>
#include <cstdlib>

#include "common.h"

namespace mvldm {

struct LinPPParams {
    const void* a; const void* a1; const void* w; const float* bias; const void* residual; void* dst;   // a1: second source of a channel concat (or NULL)
    int M, K, c0, c1, kt0, n_out, n_pad, n_dst, dst_ld, k_tiles;    // K = c0 + c1; K-tiles [0, kt0) come from `a`, the rest from `a1`
    int tiles_m, tiles_n, m_per;       // m_per: 256-row blocks per XCD
    int cpt, nch;                      // a unit = up to `cpt` consecutive column tiles of one row block; nch units per row block
    float out_scale;
    unsigned a_bytes, a1_bytes, w_bytes, bias_bytes, res_bytes, dst_bytes;
};

// MVLDM_LPP_FAKE (roofline experiments of tools/fake_probe.sh only, compiled in only with -DMVLDM_EXPERIMENTS; results are
// WRONG): 1 = activation pieces read as zeros without memory traffic, 2 = same for the weight, 4 = no stores
static const int kLppCpt = knob_int("MVLDM_LPP_CPT", 0);   // tuning: force the unit length
#ifdef MVLDM_EXPERIMENTS
static const int kLppFake = knob_int("MVLDM_LPP_FAKE", 0);
#else
static constexpr int kLppFake = 0;
#endif

constexpr unsigned kLinOob = 0xFFFFFFF0u;
constexpr unsigned kRowNone = 0xFFFFFFFFu;
constexpr int LP_BM = 256, LP_BN = 128, LP_NW = 8, LP_STAGE = (LP_BM + LP_BN) * 128;
constexpr int LP_SLAB = 3 * LP_STAGE;           // bias slab of the next tile: 128 floats (+ 512 bytes the DMA instruction also writes)
constexpr int LP_SMEM = LP_SLAB + 1024;
constexpr int kWaitAllButRing = 0x0F76;         // s_waitcnt vmcnt(6), expcnt / lgkmcnt untouched (gfx9 encoding)
constexpr int kWaitVm0 = 0x0F70;

template <typename T> struct LpMma;
template <> struct LpMma<bf16_t> {
    using Frag = bf16x8;
    static __device__ __forceinline__ f32x16 mma(Frag a, Frag b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0); }
};
template <> struct LpMma<f16_t> {
    using Frag = f16x8;
    static __device__ __forceinline__ f32x16 mma(Frag a, Frag b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0); }
};

// (buffer descriptors only in free functions: an opaque __amdgpu_buffer_rsrc_t inside a lambda trips hipcc's host pass)
__device__ __forceinline__ void lp_issue(const LinPPParams& p, char* stage, int wave, int kt, const unsigned (&ao)[2][4], const unsigned (&bo)[2]) {
    // (skip-concat 1x1 convs: the K-tiles past kt0 read the second tensor -- its own descriptor, row pitch and K-tile origin)
    const bool second = kt >= p.kt0;
    // (ONE descriptor from selected scalars: a select between two descriptors becomes a branch, and hipcc then drains the ring
    //  with vmcnt(0) at the join)
    const void* abase = second ? p.a1 : p.a;
    const unsigned abytes = second ? p.a1_bytes : p.a_bytes;
    const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(abase), 0, abytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.w), 0, p.w_bytes, 0x00020000);
    const int soff_a = (second ? kt - p.kt0 : kt) * 128, soff_w = kt * 128;
#pragma unroll
    for (int it = 0; it < 4; ++it)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, (__attribute__((address_space(3))) void*)(stage + (wave + LP_NW * it) * 1024), 16,
                                                 second ? ao[1][it] : ao[0][it], soff_a, 0, 0);
#pragma unroll
    for (int it = 0; it < 2; ++it)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (__attribute__((address_space(3))) void*)(stage + LP_BM * 128 + (wave + LP_NW * it) * 1024), 16,
                                                 bo[it], soff_w, 0, 0);
}

// per-lane source offsets of this wave's DMA pieces for output tile (tm, tn): piece q covers tile rows 8q .. 8q+7, a lane
// fetches the 16-byte chunk that belongs at its (linear) LDS position under the XOR swizzle.  valid == false: every piece
// out of range (the ring keeps its cadence past the last tile: zeros into a slot nobody reads)
__device__ __forceinline__ void lp_offsets(const LinPPParams& p, bool valid, int tm, int tn, int wave, int lane, unsigned (&ao)[2][4], unsigned (&bo)[2]) {
    const int slot = lane & 7, rsub = lane >> 3;
#pragma unroll
    for (int it = 0; it < 4; ++it) {
        const int row = (wave + LP_NW * it) * 8 + rsub;
        const int m = tm * LP_BM + row;
        const unsigned chunk = (unsigned)((slot ^ ((row >> 1) & 7)) * 8);
        ao[0][it] = (valid && m < p.M) ? ((unsigned)m * (unsigned)p.c0 + chunk) * 2u : kLinOob;
        ao[1][it] = (valid && m < p.M) ? ((unsigned)m * (unsigned)p.c1 + chunk) * 2u : kLinOob;
    }
#pragma unroll
    for (int it = 0; it < 2; ++it) {
        const int row = (wave + LP_NW * it) * 8 + rsub;
        const int n = tn * LP_BN + row;
        const unsigned chunk = (unsigned)((slot ^ ((row >> 1) & 7)) * 8);
        bo[it] = (valid && n < p.n_pad) ? ((unsigned)n * (unsigned)p.K + chunk) * 2u : kLinOob;
    }
}

// bias of the 128 packed columns of tile column tn -> LDS slab (one DMA instruction of wave 0: lanes 0..31 fetch 4 floats each)
__device__ __forceinline__ void lp_issue_bias(const LinPPParams& p, char* smem, bool geglu, bool valid, int tn, int lane) {
    const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.bias), 0, p.bias_bytes, 0x00020000);
    const int pc = tn * LP_BN + 4 * lane;                       // packed column
    int oc = pc;                                                // column of the torch-layout bias
    if (geglu) {
        const int blk = pc >> 5, w = pc & 31;
        oc = ((blk & 1) ? p.n_dst : 0) + (blk >> 1) * 32 + w;
    }
    const unsigned off = (valid && lane < 32 && pc < p.n_out) ? (unsigned)oc * 4u : kLinOob;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rb, (__attribute__((address_space(3))) void*)(smem + LP_SLAB), 16, off, 0, 0, 0);
}

__device__ __forceinline__ u32x4 lp_load_res(const LinPPParams& p, unsigned off) {
    const __amdgpu_buffer_rsrc_t rr = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.residual), 0, p.res_bytes, 0x00020000);
    return __builtin_amdgcn_raw_buffer_load_b128(rr, off, 0, 0);
}

// two 16-byte stores the compiler's wait-count bookkeeping does not see (header).  Store data is read at issue on gfx9, but over
// several cycles: a VALU write of the data registers within 2 wait states of a > 8-byte store corrupts it (LLVM inserts the nop
// for its own stores; it cannot see into the asm string, and the registers are dead -- i.e. reusable -- right after it)
__device__ __forceinline__ void lp_store2(const u32x4& rdst, const u32x4& d0, unsigned o0, const u32x4& d1, unsigned o1) {
    asm volatile("buffer_store_dwordx4 %0, %1, %4, 0 offen\n\tbuffer_store_dwordx4 %2, %3, %4, 0 offen\n\ts_nop 2"
                 ::"v"(d0), "v"(o0), "v"(d1), "v"(o1), "s"(rdst) : "memory");
}

template <typename T> __device__ __forceinline__ typename LpMma<T>::Frag lp_frag(const char* tile, int r, int kc) {
    return *reinterpret_cast<const typename LpMma<T>::Frag*>(tile + r * 128 + ((kc ^ ((r >> 1) & 7)) << 4));
}

#ifdef MVLDM_EXPERIMENTS_NOGELU
#define LP_GELU(x) (x)
#else
#define LP_GELU(x) gelu_erf_16(x)
#endif

// epilogue-side coordinates of the finished tile, per lane (lane & 31 = row inside a 32-row block)
struct LpEpi {
    unsigned row_dst[2], row_res[2];   // byte offset of this lane's row in dst / residual for row block i, kRowNone past M
    int col0;                          // first output column of the wave (GEGLU: of the value / product columns)
};

__device__ __forceinline__ unsigned lp_off(unsigned row, int col, int n_dst) {
    return (row != kRowNone && col < n_dst) ? row + (unsigned)col * 2u : kLinOob;
}

// epilogue slot SLOT of the finished tile.  plain / SiLU / GELU: SLOT = 2i + j = accumulator block (i, j).  GEGLU: SLOT = 2i + ph:
// ph 0 = GELU of the gate block (i, 1) into `gl`, ph 1 = value block (i, 0) times `gl`.
// In: accP (bias already inside, see the handover), res = the two residual chunks of this lane.  Out: two packed 16-byte chunks
// and their dst offsets (kLinOob where nothing is to be stored).
template <typename T, int EPI, bool RES, int SLOT>
__device__ __forceinline__ void lp_epi_compute(const LinPPParams& p, const f32x16 (&accP)[2][2], const LpEpi& ep, const u32x4 (&res)[2], u32x4 (&out)[2],
                                               unsigned (&out_off)[2], float (&gl)[16], int hi) {
    constexpr bool GEGLU = EPI == MVLDM_EPI_GEGLU;
    constexpr int i = SLOT >> 1;
    constexpr int j = GEGLU ? 1 - (SLOT & 1) : (SLOT & 1);
    constexpr bool gate = GEGLU && (SLOT & 1) == 0;
    float c[16];
    // transposed product: register 4q + e of this lane = column 8q + 4hi + e of row (lane & 31).  Swapping the upper half-wave's
    // group 2g with the lower half-wave's group 2g+1 leaves registers 8g .. 8g+7 = columns 16g + 8hi + (0..7)
#pragma unroll
    for (int g = 0; g < 2; ++g)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(accP[i][j][8 * g + e]), __float_as_uint(accP[i][j][8 * g + 4 + e]), false, false);
            c[8 * g + e] = __uint_as_float(r[0]);
            c[8 * g + 4 + e] = __uint_as_float(r[1]);
        }
    if constexpr (gate) {
#pragma unroll
        for (int k = 0; k < 16; ++k) gl[k] = LP_GELU(c[k]);
        return;
    }
    if constexpr (GEGLU) {
#pragma unroll
        for (int k = 0; k < 16; ++k) c[k] *= gl[k];
    } else if constexpr (EPI == MVLDM_EPI_SILU) {
#pragma unroll
        for (int k = 0; k < 16; ++k) c[k] = silu_f(c[k]);
    } else if constexpr (EPI == MVLDM_EPI_GELU) {
#pragma unroll
        for (int k = 0; k < 16; ++k) c[k] = gelu_erf_16(c[k]);
    }
#pragma unroll
    for (int g = 0; g < 2; ++g) {
        const int col = ep.col0 + (GEGLU ? 0 : 32 * j) + 16 * g + 8 * hi;
        Chunk<T> oc;
        if constexpr (RES) {
            Chunk<T> rc;
            rc.raw = res[g];
#pragma unroll
            for (int e = 0; e < 8; ++e) oc.set(e, c[8 * g + e] * p.out_scale + rc.get(e));
        } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) oc.set(e, c[8 * g + e] * p.out_scale);
        }
        out[g] = oc.raw;
        out_off[g] = lp_off(ep.row_dst[i], col, p.n_dst);
    }
}

// residual chunks of slot SLOT (plain epilogues only)
template <int SLOT> __device__ __forceinline__ void lp_res_fetch(const LinPPParams& p, const LpEpi& ep, u32x4 (&res)[2], int hi) {
    constexpr int i = SLOT >> 1, j = SLOT & 1;
#pragma unroll
    for (int g = 0; g < 2; ++g) res[g] = lp_load_res(p, lp_off(ep.row_res[i], ep.col0 + 32 * j + 16 * g + 8 * hi, p.n_dst));
}

// walks the tiles of a workgroup (all wave-uniform)
struct LpTileIter {
    int q, tm, tn, left;       // unit, tile coordinates, tiles left in the unit after this one
    bool valid;
    __device__ __forceinline__ void set(const LinPPParams& p, int unit, int m_lo, int n_units) {
        q = unit;
        valid = unit < n_units;
        const int rb = unit / p.nch, ch = unit - rb * p.nch;
        tm = m_lo + rb;
        tn = ch * p.cpt;
        left = min(p.cpt, p.tiles_n - tn) - 1;
    }
    __device__ __forceinline__ void advance(const LinPPParams& p, int wpx, int m_lo, int n_units) {
        if (left > 0) { --left; ++tn; }
        else set(p, q + wpx, m_lo, n_units);
    }
};

template <typename T, int EPI, bool RES>
__global__ __launch_bounds__(512) void linear_pp_kernel(const LinPPParams p) {
    constexpr bool GEGLU = EPI == MVLDM_EPI_GEGLU;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;        // 4 x 2 waves of 64 x 64
    const int hi = lane >> 5, l31 = lane & 31;

    // tiles of this workgroup.  XCD x owns row blocks [x * m_per, (x+1) * m_per), cut into units of up to `cpt` consecutive
    // column tiles; the XCD's workgroups take its units round-robin, row block major.  The first tile of a unit fetches the
    // activation rows from HBM (the workgroups holding the other units of the row block do so at the same time: one fill),
    // the rest of the unit finds them in L2; the host picks cpt so that the row blocks in flight on an XCD stay there
    const int xcd = blockIdx.x & 7, lid = blockIdx.x >> 3, wpx = gridDim.x >> 3;
    const int m_lo = xcd * p.m_per, m_cnt = min(p.tiles_m, m_lo + p.m_per) - m_lo;
    const int n_units = m_cnt > 0 ? m_cnt * p.nch : 0;
    if (lid >= n_units) return;
    LpTileIter cur, nxt, iss;                    // compute side, the tile after it, issue side (newest ring piece in flight)
    cur.set(p, lid, m_lo, n_units);
    nxt = cur; nxt.advance(p, wpx, m_lo, n_units);
    iss = cur;

    u32x4 rdst;
    rdst[0] = (unsigned)(uintptr_t)p.dst; rdst[1] = (unsigned)((uintptr_t)p.dst >> 32) & 0xFFFFu; rdst[2] = p.dst_bytes; rdst[3] = 0x00020000u;

    f32x16 accC[2][2], accP[2][2];
    float gl[16];
    u32x4 res[2][2], out[2];
    unsigned out_off[2] = {kLinOob, kLinOob};
    LpEpi ep;
    ep.row_dst[0] = ep.row_dst[1] = ep.row_res[0] = ep.row_res[1] = kRowNone;
    ep.col0 = 0;
#pragma unroll
    for (int k = 0; k < 16; ++k) gl[k] = 0.f;
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int g = 0; g < 2; ++g) res[a][g] = u32x4{0u, 0u, 0u, 0u};
    out[0] = out[1] = u32x4{0u, 0u, 0u, 0u};
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) accP[i][j][r] = 0.f;

    // ---- prologue: bias slab of the first tile, K-tiles 0 and 1 ----
    unsigned ao[2][4], bo[2];
    int kt_i = 1;                                // issue side: K-tile of the newest piece in flight
    {
        if (wave == 0) lp_issue_bias(p, smem, GEGLU, true, cur.tn, lane);
        lp_offsets(p, true, cur.tm, cur.tn, wave, lane, ao, bo);
        lp_issue(p, smem, wave, 0, ao, bo);
        lp_issue(p, smem + LP_STAGE, wave, 1, ao, bo);
    }
    __builtin_amdgcn_s_waitcnt(kWaitVm0);
    __builtin_amdgcn_s_barrier();
    const float* slab = reinterpret_cast<const float*>(smem + LP_SLAB) + wn * 64 + 4 * hi;
#define LP_INIT_ACC()                                                                         \
    _Pragma("unroll") for (int j = 0; j < 2; ++j) _Pragma("unroll") for (int q = 0; q < 4; ++q) { \
        const f32x4 b = *reinterpret_cast<const f32x4*>(slab + 32 * j + 8 * q);               \
        _Pragma("unroll") for (int e = 0; e < 4; ++e) { accC[0][j][4 * q + e] = b[e]; accC[1][j][4 * q + e] = b[e]; } \
    }
    LP_INIT_ACC()

    int rs = 0;                                  // ring slot the current step reads

// top of a step: the step's operands have landed (everything but the newest six ring pieces), every wave is done with the
// slot the next pieces go to; then, in this order: stores of the slot finished last step, residual rows of the next slot,
// bias slab (step 1), ring pieces of step s+2
#define LP_TOP_BEGIN()                              \
    __builtin_amdgcn_s_waitcnt(kWaitAllButRing);    \
    __builtin_amdgcn_s_barrier();
#define LP_STORE_OUT()                                                        \
    lp_store2(rdst, out[0], out_off[0], out[1], out_off[1]);                    \
    out_off[0] = out_off[1] = kLinOob;
#define LP_RING_ISSUE()                                                                                   \
    {                                                                                                     \
        if (++kt_i == p.k_tiles) {                                                                        \
            kt_i = 0;                                                                                     \
            iss.advance(p, wpx, m_lo, n_units);                                                           \
            lp_offsets(p, iss.valid, iss.tm, iss.tn, wave, lane, ao, bo);                                 \
        }                                                                                                 \
        const int is = rs == 0 ? 2 : rs - 1;                                                              \
        lp_issue(p, smem + is * LP_STAGE, wave, kt_i, ao, bo);                                            \
    }
// the 16 MFMAs of a step (transposed product: W fragment is the A operand)
#define LP_MFMA_HALF(KK0)                                                                                 \
    _Pragma("unroll") for (int kk = (KK0); kk < (KK0) + 2; ++kk) {                                        \
        typename LpMma<T>::Frag fa[2], fb[2];                                                             \
        _Pragma("unroll") for (int i = 0; i < 2; ++i) fa[i] = lp_frag<T>(at, wm * 64 + i * 32 + l31, kk * 2 + hi); \
        _Pragma("unroll") for (int j = 0; j < 2; ++j) fb[j] = lp_frag<T>(bt, wn * 64 + j * 32 + l31, kk * 2 + hi); \
        _Pragma("unroll") for (int i = 0; i < 2; ++i) _Pragma("unroll") for (int j = 0; j < 2; ++j)       \
            accC[i][j] = LpMma<T>::mma(fb[j], fa[i], accC[i][j]);                                         \
    }
#define LP_STEP_END() rs = rs == 2 ? 0 : rs + 1;

    for (; cur.valid; cur = nxt, nxt.advance(p, wpx, m_lo, n_units)) {
        // ---- step 0: stores of the older tile's last slot; residual rows of slot 0 ----
        {
            LP_TOP_BEGIN()
            LP_STORE_OUT()
            if constexpr (RES) lp_res_fetch<0>(p, ep, res[0], hi);
            LP_RING_ISSUE()
            const char* at = smem + rs * LP_STAGE; const char* bt = at + LP_BM * 128;
            LP_MFMA_HALF(0) LP_MFMA_HALF(2)
            LP_STEP_END()
        }
        // ---- step 1: slot 0; the next tile's bias slab ----
        {
            LP_TOP_BEGIN()
            if constexpr (RES) lp_res_fetch<1>(p, ep, res[1], hi);
            if (wave == 0) {
                lp_issue_bias(p, smem, GEGLU, nxt.valid, nxt.tn, lane);
            }
            LP_RING_ISSUE()
            const char* at = smem + rs * LP_STAGE; const char* bt = at + LP_BM * 128;
            LP_MFMA_HALF(0)
            lp_epi_compute<T, EPI, RES, 0>(p, accP, ep, res[0], out, out_off, gl, hi);
            LP_MFMA_HALF(2)
            LP_STEP_END()
        }
        // ---- step 2: slot 1 ----
        {
            LP_TOP_BEGIN()
            if constexpr (!GEGLU) { LP_STORE_OUT() }
            if constexpr (RES) lp_res_fetch<2>(p, ep, res[0], hi);
            LP_RING_ISSUE()
            const char* at = smem + rs * LP_STAGE; const char* bt = at + LP_BM * 128;
            LP_MFMA_HALF(0)
            lp_epi_compute<T, EPI, RES, 1>(p, accP, ep, res[1], out, out_off, gl, hi);
            LP_MFMA_HALF(2)
            LP_STEP_END()
        }
        // ---- step 3: slot 2 ----
        {
            LP_TOP_BEGIN()
            LP_STORE_OUT()
            if constexpr (RES) lp_res_fetch<3>(p, ep, res[1], hi);
            LP_RING_ISSUE()
            const char* at = smem + rs * LP_STAGE; const char* bt = at + LP_BM * 128;
            LP_MFMA_HALF(0)
            lp_epi_compute<T, EPI, RES, 2>(p, accP, ep, res[0], out, out_off, gl, hi);
            LP_MFMA_HALF(2)
            LP_STEP_END()
        }
        // ---- step 4: slot 3 ----
        {
            LP_TOP_BEGIN()
            if constexpr (!GEGLU) { LP_STORE_OUT() }
            LP_RING_ISSUE()
            const char* at = smem + rs * LP_STAGE; const char* bt = at + LP_BM * 128;
            LP_MFMA_HALF(0)
            lp_epi_compute<T, EPI, RES, 3>(p, accP, ep, res[1], out, out_off, gl, hi);
            LP_MFMA_HALF(2)
            LP_STEP_END()
        }
        // ---- steps 5 .. k_tiles-1: main loop only ----
#pragma unroll 1
        for (int kt = 5; kt < p.k_tiles; ++kt) {
            LP_TOP_BEGIN()
            LP_STORE_OUT()
            LP_RING_ISSUE()
            const char* at = smem + rs * LP_STAGE; const char* bt = at + LP_BM * 128;
            LP_MFMA_HALF(0) LP_MFMA_HALF(2)
            LP_STEP_END()
        }
        // ---- handover: the finished accumulators go to the epilogue side, the next tile starts from its bias ----
        {
            const int tm = cur.tm, tn = cur.tn;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int m = tm * LP_BM + wm * 64 + i * 32 + l31;
                ep.row_dst[i] = m < p.M ? (unsigned)m * (unsigned)p.dst_ld * 2u : kRowNone;
                ep.row_res[i] = m < p.M ? (unsigned)m * (unsigned)p.n_dst * 2u : kRowNone;
            }
            ep.col0 = GEGLU ? (tn * LP_BN + wn * 64) >> 1 : tn * LP_BN + wn * 64;
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) accP[i][j] = accC[i][j];
            LP_INIT_ACC()
        }
    }
    // ---- drain: the last tile's epilogue ----
    __builtin_amdgcn_s_waitcnt(kWaitVm0);
    LP_STORE_OUT()
#define LP_DRAIN(S)                                                               \
    if constexpr (RES) lp_res_fetch<S>(p, ep, res[0], hi);                        \
    lp_epi_compute<T, EPI, RES, S>(p, accP, ep, res[0], out, out_off, gl, hi);    \
    LP_STORE_OUT()
    LP_DRAIN(0) LP_DRAIN(1) LP_DRAIN(2) LP_DRAIN(3)
#undef LP_DRAIN
#undef LP_INIT_ACC
#undef LP_TOP_BEGIN
#undef LP_STORE_OUT
#undef LP_RING_ISSUE
#undef LP_MFMA_HALF
#undef LP_STEP_END
}

bool linear_pp_applicable(const mvldm_igemm_desc& d) {
    if (d.act_dtype == MVLDM_F32 || d.dst_dtype != d.act_dtype) return false;
    if (d.ksize != 1 || d.stride != 1 || d.upsample != 0 || d.row_bias || d.k_order != 1 || d.splitk > 1) return false;
    if (d.h_in != d.h_out || d.w_in != d.w_out || d.pad != 0) return false;
    if (d.epilogue != MVLDM_EPI_NONE && d.residual) return false;      // activation + residual: no caller, not instantiated
    const int n_dst = d.epilogue == MVLDM_EPI_GEGLU ? d.n_out / 2 : d.n_out;
    const int dst_ld = d.dst_ld > 0 ? d.dst_ld : n_dst;
    if ((d.c1 == 0) != (d.src1 == nullptr) || d.c0 % 64 || d.c1 % 64 || d.c0 + d.c1 < 320 || d.k_pad != d.c0 + d.c1 || d.n_out % 8 || n_dst % 8 || dst_ld % 8 || dst_ld < n_dst) return false;
    if (d.epilogue == MVLDM_EPI_GEGLU && d.n_out % 64) return false;
    if (d.bias && ((uintptr_t)d.bias % 16)) return false;
    const double m = (double)d.n_img * d.h_out * d.w_out;
    return m * d.c0 * 2.0 < 4.0e9 && m * d.c1 * 2.0 < 4.0e9 && (double)d.n_pad * d.k_pad * 2.0 < 4.0e9 && m * dst_ld * 2.0 < 4.0e9 && m * n_dst * 2.0 < 4.0e9;
}

template <typename T, int EPI, bool RES> static int linear_pp_launch(const LinPPParams& p, int grid, hipStream_t s) {
    static std::atomic<uint64_t> done{0};
    if (int rc0 = ensure_dyn_smem(reinterpret_cast<const void*>(linear_pp_kernel<T, EPI, RES>), LP_SMEM, done)) return rc0;
    hipLaunchKernelGGL((linear_pp_kernel<T, EPI, RES>), dim3(grid), dim3(512), LP_SMEM, s, p);
    return check_launch();
}

int linear_pp_run(const mvldm_igemm_desc& d, hipStream_t s) {
    MVLDM_REQUIRE(linear_pp_applicable(d), "igemm: tile 12 (persistent pipelined Linear) does not apply to this problem");
    LinPPParams p;
    p.a = d.src0; p.a1 = d.src1; p.w = d.weight; p.bias = d.bias; p.residual = d.residual; p.dst = d.dst;
    p.M = d.n_img * d.h_out * d.w_out; p.K = d.c0 + d.c1; p.c0 = d.c0; p.c1 = d.c1; p.kt0 = d.c0 / 64; p.n_out = d.n_out; p.n_pad = d.n_pad;
    p.n_dst = d.epilogue == MVLDM_EPI_GEGLU ? d.n_out / 2 : d.n_out;
    p.dst_ld = d.dst_ld > 0 ? d.dst_ld : p.n_dst;
    p.k_tiles = p.K / 64; p.out_scale = d.out_scale;
    p.tiles_m = (p.M + LP_BM - 1) / LP_BM; p.tiles_n = (d.n_pad + LP_BN - 1) / LP_BN;
    p.m_per = (p.tiles_m + 7) / 8;
    p.a_bytes = (unsigned)((double)p.M * p.c0 * 2.0); p.a1_bytes = (unsigned)((double)p.M * p.c1 * 2.0); p.w_bytes = (unsigned)((double)d.n_pad * d.k_pad * 2.0);
    p.bias_bytes = d.bias ? (unsigned)d.n_out * 4u : 0u;
    p.res_bytes = d.residual ? (unsigned)((double)p.M * p.n_dst * 2.0) : 0u;
    p.dst_bytes = (unsigned)((double)p.M * p.dst_ld * 2.0);
    if (kLppFake & 1) p.a_bytes = p.a1_bytes = 0;
    if (kLppFake & 2) p.w_bytes = 0;
    if (kLppFake & 4) p.dst_bytes = 0;
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
    if (kLppCpt > 0) p.cpt = std::min(kLppCpt, p.tiles_n);
    p.nch = (p.tiles_n + p.cpt - 1) / p.cpt;
    p.cpt = (p.tiles_n + p.nch - 1) / p.nch;                    // even units
    const int grid = 8 * wpx;
    const bool res = d.residual != nullptr;
    return dispatch_dtype(d.act_dtype, [&](auto t) -> int {
        using T = decltype(t);
        if constexpr (sizeof(T) == 2) {
            switch (d.epilogue) {
                case MVLDM_EPI_NONE: return res ? linear_pp_launch<T, MVLDM_EPI_NONE, true>(p, grid, s) : linear_pp_launch<T, MVLDM_EPI_NONE, false>(p, grid, s);
                case MVLDM_EPI_SILU: return linear_pp_launch<T, MVLDM_EPI_SILU, false>(p, grid, s);
                case MVLDM_EPI_GELU: return linear_pp_launch<T, MVLDM_EPI_GELU, false>(p, grid, s);
                case MVLDM_EPI_GEGLU: return linear_pp_launch<T, MVLDM_EPI_GEGLU, false>(p, grid, s);
                default: return set_error(MVLDM_ERR_ARG, "igemm: tile 12: epilogue %d", d.epilogue);
            }
        } else {
            return set_error(MVLDM_ERR_ARG, "igemm: tile 12 needs a 16-bit activation type");
        }
    });
}

}  // namespace mvldm
