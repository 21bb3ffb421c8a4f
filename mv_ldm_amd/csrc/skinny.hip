// Skinny-M weight-streaming implicit GEMM: tile 15 of the implicit-GEMM family (include/mvldm.h: mvldm_igemm_fwd; 16-bit activations).
// Round 5.
//
// The one-scene regime (a `sample()` of generate_mvldm.py: 9 images per UNet pass) runs the 8x8 / 4x4 levels of the UNet
// (mvunet.py:150-166 mid, :169-200 up0 / up1, the deep down blocks) at M = 144 ... 576 output rows: every one of those ~100 launches is a
// WEIGHT STREAM (3 ... 60 MB read once) with almost no arithmetic, and the tiled kernels of igemm.hip reach 0.7 TB/s on them -- their
// weights go global -> LDS (33 B/clk/CU of LDS-DMA fill), output tiles are 64 columns wide so K has to be split over workgroups, and the
// fp32 slabs need a second launch to fold.  Here
//   * the weight is packed in MFMA-FRAGMENT order (mvldm_pack_skinny: [16-column tile][32-wide k-step][lane][16 B]) and goes
//     global -> VGPR: one buffer_load_dwordx4 per fragment = 1 KB contiguous, DS stages of them in flight per wave, never through LDS;
//   * a workgroup owns NT 16-column tiles over the FULL K for MT 16-row tiles (whole images), so there is no slab and no reduce launch:
//     K is split over the workgroup's WAVES -- by tap for a 3x3 conv (wave = tap, the 9 waves share one LDS image of the 64-channel
//     block and read it at their tap's shift: the shift is a per-lane LDS address, zero padding is a row of zeros), by k-step for a
//     Linear -- and the waves' partial accumulators are folded through LDS in a fixed order (deterministic);
//   * the small activation block (MT*16 source rows x 64 channels per stage) is staged by LDS-DMA into a ring as deep as the weight
//     ring, full 128-byte lines, XOR-swizzled like igemm.hip's tiles;
//   * v_mfma_f32_16x16x32 with the WEIGHTS as the A operand: a lane ends up with 4 consecutive output channels of one pixel.
// Workgroups that share a weight panel (the image groups of one column group) sit on one XCD and run at the same time, so a weight
// byte crosses the fabric once.
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "common.h"

namespace mvldm {

struct SkParams {
    const void* src0; const void* src1; const void* w;
    const float* bias; const float* row_bias; const void* residual; void* dst;
    int c0, c1;                       // source channels (multiples of 64)
    int n_img, h_in, w_in, hw_in, h_out, w_out, hw_out;
    int stride, ty0, tx0, ksize;      // tap t reads input pixel (oy*stride + ty0 + t/ksize, ox*stride + tx0 + t%ksize)
    int M, n_out, n_dst, dst_ld, row_bias_ld, rb_hw;   // rb_hw: output rows per row of row_bias (the caller's image size)
    int epilogue, dst_f32;
    float out_scale;
    int n_tiles;                      // 16-column tiles of the packed weight (n_pad / 16)
    int n_groups, m_groups, G;        // column groups of NT tiles; image groups of G images
    int n_cbs, cb0;                   // 64-channel blocks in all / in src0
    int n_stages;                     // ceil(n_cbs / SC)
    int ksteps;                       // k-steps (32 wide) per tile = n_cbs * taps * 2
    int scatter, ph_y, ph_x;          // sub-pixel phase of a decomposed nearest-2x upsampling conv: output row -> (2i+py, 2j+px)
    int hw_sh, w_sh, rb_sh;           // log2 of hw_out / w_out / rb_hw when they are powers of two (every UNet shape), else -1
    unsigned src0_bytes, src1_bytes, w_bytes;
    unsigned long long* trace;        // EXPERIMENT (MVLDM_SK_TRACE): per-stage s_memtime stamps of workgroup 0
    int fake;                         // EXPERIMENT knob (MVLDM_SK_FAKE, -DMVLDM_EXPERIMENTS builds only: tools/sk_probe.sh)
};

#ifdef MVLDM_EXPERIMENTS
static const int kSkFake = knob_int("MVLDM_SK_FAKE", 0);   // 1 no W traffic, 2 no A traffic, 4 no stages, 8 empty kernel, 16 no fold / epilogue
#else
static constexpr int kSkFake = 0;
#endif

constexpr unsigned kSkOob = 0xFFFFFFF0u;

template <typename T> struct SkMma;
template <> struct SkMma<bf16_t> {
    using Frag = bf16x8;
    static __device__ __forceinline__ f32x4 mma(Frag a, Frag b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0); }
};
template <> struct SkMma<f16_t> {
    using Frag = f16x8;
    static __device__ __forceinline__ f32x4 mma(Frag a, Frag b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0); }
};

constexpr int sk_vm(int n) { return (n & 15) | ((n >> 4) << 14) | 0x0F70; }   // s_waitcnt vmcnt(n) only (gfx9 encoding)

// compile-time geometry of one instantiation.  (A "balanced" 3x3 form -- 8 waves, 4 channel blocks per stage, wave w owns tap w of all 8
// (channel block, k-step) pairs plus tap 8 of pair w: 27 MFMAs per barrier instead of 6 -- was built and measured EQUAL to the
// tap-per-wave form, 14.8 against 14.7 us on the 4x4-level conv: the cost per stage is its lockstep phases, see the loop; removed.)
template <int TAPS_, int NW_, int SC_, int MT_, int NT_, int DS_, int SM_, bool DUAL_ = false> struct SkCfg {
    static constexpr int TAPS = TAPS_, NW = NW_, SC = SC_, MT = MT_, NT = NT_, DS = DS_, SM = SM_;
    static constexpr bool DUAL = DUAL_;                      // two sources concatenated along the channels (1x1 shortcut convs)
    static constexpr int SRP = MT * 16 * SM;                 // source pixel rows of the workgroup's images (SM = 4: stride-2 conv)
    static constexpr int CBS = (SRP + 8) * 128;              // one 64-channel block of a slot: the rows + 8 rows of zeros
    static constexpr int SLOT = SC * CBS;
    static constexpr int RING = DS * SLOT;                   // as many activation slots as weight-ring stages (see the loop)
    static constexpr int PPC = SRP / 8;                      // 1 KB LDS-DMA pieces per channel block
    static constexpr int PIECES = SC * PPC;                  // ... per stage
    static constexpr int PA = (PIECES + NW - 1) / NW;        // ... per wave
    static constexpr int ITEMS = SC * TAPS * 2 / NW;         // (channel block, tap, k-step) items per wave and stage
    static constexpr int PARK = NW * MT * NT * 1024;
    static constexpr int DUMP = (RING > PARK ? RING : PARK); // 1 KB: where the surplus DMA pieces of the last wave land
    static constexpr int SMEM = DUMP + 1024;
    static constexpr int WAIT = (DS - 2) * (PA + ITEMS * NT) + ITEMS * NT;   // what may be in flight when A(s) must have landed (see the loop)
    static constexpr int WAIT_B = (DS - 2) * (PA + ITEMS * NT);               // ... for the waves that issue before they compute
    static constexpr bool STAGGER = NW >= 8;                                  // two wave groups with swapped phases (the loop's comment)
    static_assert((SC * TAPS * 2) % NW == 0 && NW % TAPS == 0, "items must divide over the waves with one tap per wave");
    static_assert(DS >= 3 && WAIT <= 63, "ring depth / wait count");
    static_assert(SMEM <= 160 * 1024, "LDS");
};

// item ii of wave `wave` in a stage -> (tap, channel block inside the stage, k-step)
struct SkItem { int tap, cbl, ks; };
template <typename C> __device__ __forceinline__ SkItem sk_item(int wave, int ii) {
    const int it = wave + C::NW * ii, kk = it / C::TAPS;
    return {it % C::TAPS, kk >> 1, kk & 1};
}

// ---- the epilogue both kernels share: the waves' partial tiles are folded through LDS in wave order (deterministic), then bias /
// time-embedding row / activation / residual and the store.  Item k of a thread: e = tid + k * threads -> (tile t = e >> 6, lane l = e & 63)
// = output pixel row 16 i + (l & 15), packed columns 16 (tile0 + j) + 4 (l >> 4) .. + 3 (the MFMA takes the WEIGHTS as its A operand: a
// lane holds 4 consecutive output channels of one pixel).  `fetch` requests the operands -- at the top of the kernel when the register
// budget allows (EPRE): they are then the oldest entries of the wave's memory queue, cost the loop nothing, and the fold does not wait for
// a round trip.  (The residual may alias dst -- x += f(x) -- every element is read by the thread that writes it.)
template <typename T, int NW, int MT, int NT> struct SkEpi {
    static constexpr int EIT = (MT * NT + NW - 1) / NW;
    static constexpr bool EPRE = MT * NT <= 9;
    f32x4 bias[EIT], gate[EIT], rb[EIT];
    u32x2 res[EIT];
    int col[EIT], m[EIT];
    __device__ __forceinline__ void fetch(const SkParams& p, int k, int tid, int tile0, int row0, int rows) {
        const bool geglu = p.epilogue == MVLDM_EPI_GEGLU;
        const int nto = geglu ? NT / 2 : NT;
        const int e = tid + k * NW * 64;
        const int t = e >> 6, l = e & 63;
        const int i = t / nto, jo = t - i * nto;
        const int rl = i * 16 + (l & 15);
        const int c = geglu ? ((tile0 + 2 * jo) >> 1) * 16 + 4 * (l >> 4) : (tile0 + jo) * 16 + 4 * (l >> 4);
        const bool ok = e < MT * nto * 64 && rl < rows && c < p.n_dst;
        col[k] = ok ? c : -1;
        m[k] = row0 + rl;
        bias[k] = gate[k] = rb[k] = f32x4{0.f, 0.f, 0.f, 0.f};
        res[k] = u32x2{0u, 0u};
        if (ok) {
            if (p.bias) {
                bias[k] = *reinterpret_cast<const f32x4*>(p.bias + c);
                if (geglu) gate[k] = *reinterpret_cast<const f32x4*>(p.bias + p.n_dst + c);
            }
            if (p.row_bias) {
                const int img = p.rb_sh >= 0 ? m[k] >> p.rb_sh : m[k] / p.rb_hw;
                rb[k] = *reinterpret_cast<const f32x4*>(p.row_bias + (size_t)img * p.row_bias_ld + c);
            }
            if (p.residual) res[k] = *reinterpret_cast<const u32x2*>(reinterpret_cast<const T*>(p.residual) + (size_t)m[k] * p.n_dst + c);
        }
    }
    // `smem`: the parked partial tiles [wave][row tile][column tile][lane][4 floats] (after the barrier that publishes them)
    __device__ __forceinline__ void finish(const SkParams& p, const char* smem, int tid, int tile0, int row0, int rows) {
        const bool geglu = p.epilogue == MVLDM_EPI_GEGLU;
        const int nto = geglu ? NT / 2 : NT;
#pragma unroll
        for (int k = 0; k < EIT; ++k) {
            if constexpr (!EPRE) fetch(p, k, tid, tile0, row0, rows);
            const int c = col[k], mm = m[k];
            if (c < 0) continue;
            const int e = tid + k * NW * 64;
            const int t = e >> 6, l = e & 63;
            const int i = t / nto, jo = t - i * nto;
            const int j = geglu ? 2 * jo : jo;
            f32x4 v = f32x4{0.f, 0.f, 0.f, 0.f}, g = v;
            for (int w = 0; w < NW; ++w) {
                const f32x4 a = *reinterpret_cast<const f32x4*>(smem + ((w * MT + i) * NT + j) * 1024 + l * 16);
                v[0] += a[0]; v[1] += a[1]; v[2] += a[2]; v[3] += a[3];
            }
            if (geglu) {
                for (int w = 0; w < NW; ++w) {
                    const f32x4 a = *reinterpret_cast<const f32x4*>(smem + ((w * MT + i) * NT + j + 1) * 1024 + l * 16);
                    g[0] += a[0]; g[1] += a[1]; g[2] += a[2]; g[3] += a[3];
                }
#pragma unroll
                for (int q = 0; q < 4; ++q) v[q] = (v[q] + bias[k][q]) * gelu_erf_fast(g[q] + gate[k][q]);
            } else {
#pragma unroll
                for (int q = 0; q < 4; ++q) v[q] += bias[k][q] + rb[k][q];
                if (p.epilogue == MVLDM_EPI_SILU) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) v[q] = silu_f(v[q]);
                } else if (p.epilogue == MVLDM_EPI_GELU) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) v[q] = gelu_erf_fast(v[q]);
                }
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) v[q] *= p.out_scale;
            if (p.residual) {
                const T* re = reinterpret_cast<const T*>(&res[k]);
#pragma unroll
                for (int q = 0; q < 4; ++q) v[q] += to_f32<T>(re[q]);
            }
            size_t drow = (size_t)mm;
            if (p.scatter) {
                const int img = p.hw_sh >= 0 ? mm >> p.hw_sh : mm / p.hw_out, rem = mm - img * p.hw_out;
                const int oi = p.w_sh >= 0 ? rem >> p.w_sh : rem / p.w_out, oj = rem - oi * p.w_out;
                drow = ((size_t)img * (2 * p.h_out) + 2 * oi + p.ph_y) * (size_t)(2 * p.w_out) + 2 * oj + p.ph_x;
            }
            const size_t o = drow * p.dst_ld + c;
            if (p.dst_f32) {
                *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(p.dst) + o) = v;
            } else {
                u32x2 out;
                T* oe = reinterpret_cast<T*>(&out);
#pragma unroll
                for (int q = 0; q < 4; ++q) oe[q] = from_f32<T>(v[q]);
                *reinterpret_cast<u32x2*>(reinterpret_cast<T*>(p.dst) + o) = out;
            }
        }
    }
};

// ---- the two streams of a wave.  Everything that does not change from stage to stage is computed once (per-lane source offsets of the
// LDS-DMA pieces, LDS destinations, per-item fragment offsets, one buffer descriptor per column tile whose extent IS the tile's K range,
// so fragments past the end of K read zeros by the range check): a stage issues its loads with one scalar add and no selects.  (First
// version: ~65 scalar and ~20 vector instructions per stage -- 64-bit multiply-adds, descriptor selects, kernel-argument reloads
// behind lgkmcnt(0) -- for 6 MFMAs: PMC showed a wave 26 % issuing, 25 % stalled on issue, 49 % waiting.)
template <typename C> struct SkStreams {
    unsigned a_voff[C::PA];           // per lane: byte offset of its chunk in source 0 (row, swizzled chunk; the channel block rides in the scalar offset) or out of range
    unsigned a_voff1[C::PA];          // ... in source 1 (the second half of a skip concat; 1x1 convs only)
    int a_dst[C::PA];                 // LDS byte offset of the piece inside a slot (-1: surplus piece -> the dump KB)
    int a_cbl[C::PA];                 // its channel block inside the stage
    unsigned w_voff[C::ITEMS];        // per lane: byte offset of its 16 bytes in the tile's stream for the NEXT stage to issue
};

// (buffer descriptors only in plain free functions, one per form: an opaque __amdgpu_buffer_rsrc_t in an `if constexpr` branch or a lambda trips
// hipcc's host pass)
template <typename C>
__device__ __forceinline__ void sk_issue_a1(const SkParams& p, const SkStreams<C>& st, char* smem, char* slot_base, int s) {
    const int cb_first = s * C::SC;
    const __amdgpu_buffer_rsrc_t r0 = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.src0), 0, p.src0_bytes, 0x00020000);
#pragma unroll
    for (int pp = 0; pp < C::PA; ++pp) {
        const int cb = cb_first + st.a_cbl[pp];
        const bool in_k = cb < p.n_cbs;               // channel blocks past K (the stages past the end, a ragged last stage) read zeros
        char* dst = st.a_dst[pp] >= 0 ? slot_base + st.a_dst[pp] : smem + C::DUMP;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(r0, (__attribute__((address_space(3))) void*)dst, 16, in_k ? st.a_voff[pp] : kSkOob, cb * 128, 0, 0);
    }
}
// two sources concatenated along the channels: a piece reads the source its channel block lies in
template <typename C>
__device__ __forceinline__ void sk_issue_a2(const SkParams& p, const SkStreams<C>& st, char* smem, char* slot_base, int s) {
    const int cb_first = s * C::SC;
#pragma unroll
    for (int pp = 0; pp < C::PA; ++pp) {
        const int cb = cb_first + st.a_cbl[pp];
        const bool in_k = cb < p.n_cbs;
        char* dst = st.a_dst[pp] >= 0 ? slot_base + st.a_dst[pp] : smem + C::DUMP;
        // one descriptor from selected scalars (a select between two descriptors becomes a branch whose join drains the queue)
        const bool from0 = cb < p.cb0;
        const void* base = from0 ? p.src0 : p.src1;
        const unsigned bytes = from0 ? p.src0_bytes : p.src1_bytes;
        const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, bytes, 0x00020000);
        const unsigned v = from0 ? st.a_voff[pp] : st.a_voff1[pp];
        __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)dst, 16, in_k ? v : kSkOob, (from0 ? cb : cb - p.cb0) * 128, 0, 0);
    }
}
template <typename C>
__device__ __forceinline__ void sk_issue_a(const SkParams& p, const SkStreams<C>& st, char* smem, char* slot_base, int s) {
    if constexpr (C::DUAL) sk_issue_a2<C>(p, st, smem, slot_base, s);
    else sk_issue_a1<C>(p, st, smem, slot_base, s);
}

// the weight fragments of one stage: item ii -> (tap, channel block, k-step) by sk_item; the fragment of tile nt is 1 KB at
// (nt * ksteps + kg) * 1024, lane l reads its 16 bytes = W[16 nt + (l & 15)][32 kg + 8 (l >> 4) .. + 8].  `w_voff` walks the stream.
template <typename C>
__device__ __forceinline__ void sk_issue_w(const SkParams& p, SkStreams<C>& st, u32x4 (&wf)[C::ITEMS][C::NT], int tile0) {
#pragma unroll
    for (int j = 0; j < C::NT; ++j) {
        const int nt = tile0 + j;
        const char* base = reinterpret_cast<const char*>(p.w) + (size_t)nt * (size_t)p.ksteps * 1024u;
        const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(base), 0, nt < p.n_tiles ? (unsigned)p.ksteps * 1024u : 0u, 0x00020000);
#pragma unroll
        for (int ii = 0; ii < C::ITEMS; ++ii) wf[ii][j] = __builtin_amdgcn_raw_buffer_load_b128(rw, st.w_voff[ii], 0, 0);
    }
#pragma unroll
    for (int ii = 0; ii < C::ITEMS; ++ii) st.w_voff[ii] += (unsigned)(C::SC * C::TAPS * 2) * 1024u;
}

template <typename T, typename C>
__global__ __launch_bounds__(C::NW * 64) void skinny_kernel(const SkParams p) {
    using Frag = typename SkMma<T>::Frag;
    constexpr int MT = C::MT, NT = C::NT, NW = C::NW, DS = C::DS, ITEMS = C::ITEMS;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

    // workgroup -> (column group, image group): the image groups of one column group stream the same weights; they sit on ONE XCD
    // (workgroup b runs on XCD b % 8) next to each other in time, so a weight line crosses the fabric once
    const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
    const int mg = idx % p.m_groups, ng = (idx / p.m_groups) * 8 + xcd;
    if (ng >= p.n_groups) return;
#ifdef MVLDM_EXPERIMENTS
    if (p.fake & 8) return;
#endif
    const int tile0 = ng * NT;
    const int img0 = mg * p.G, imgs = min(p.G, p.n_img - img0);
    const int src_row0 = img0 * p.hw_in, src_rows = imgs * p.hw_in;
    const int row0 = img0 * p.hw_out, rows = imgs * p.hw_out;

    // ---- the rows of zeros behind every channel block of the ring (never overwritten by the DMA)
    for (int z = tid; z < DS * C::SC * 64; z += NW * 64) {
        const int blk = z >> 6;
        *reinterpret_cast<u32x4*>(smem + (blk / C::SC) * C::SLOT + (blk % C::SC) * C::CBS + C::SRP * 128 + (z & 63) * 16) = u32x4{0u, 0u, 0u, 0u};
    }

    // ---- per-lane fragment addresses: row tile i of this wave's tap -> LDS byte offset of (shifted source pixel, 16-byte chunk lane >> 4) of
    // k-step 0 inside a channel block; k-step 1 = the same ^ 64 (the swizzle is an XOR of the chunk index); padding taps -> the rows of zeros
    int aoff[MT];
    {
        const int tap = wave % C::TAPS;
        const int ty = tap / p.ksize, tx = tap - ty * p.ksize;
#pragma unroll
        for (int i = 0; i < MT; ++i) {
            const int r = i * 16 + (lane & 15);
            const int il = p.hw_sh >= 0 ? r >> p.hw_sh : r / p.hw_out, rem = r - il * p.hw_out;
            const int oy = p.w_sh >= 0 ? rem >> p.w_sh : rem / p.w_out, ox = rem - oy * p.w_out;
            const int iy = oy * p.stride + p.ty0 + ty, ix = ox * p.stride + p.tx0 + tx;
            const bool ok = r < rows && iy >= 0 && iy < p.h_in && ix >= 0 && ix < p.w_in;
            const int srow = ok ? il * p.hw_in + iy * p.w_in + ix : C::SRP;
            aoff[i] = srow * 128 + ((((lane >> 4) ^ (srow >> 1)) & 7) << 4);
        }
    }
    constexpr int NFO = MT <= 6 ? ITEMS : 1;   // (many row tiles: the offsets are formed at the read, one v_xor each, instead of kept)
    int foff[NFO][MT];                // LDS byte offset (inside a slot) of the activation fragment of (item, row tile)
    int it_off[ITEMS], it_x[ITEMS];
    SkStreams<C> st;
#pragma unroll
    for (int ii = 0; ii < ITEMS; ++ii) {
        const SkItem it = sk_item<C>(wave, ii);
        it_off[ii] = it.cbl * C::CBS;
        it_x[ii] = it.ks * 64;
        if (ii < NFO) {
#pragma unroll
            for (int i = 0; i < MT; ++i) foff[ii][i] = it.cbl * C::CBS + (aoff[i] ^ (it.ks * 64));
        }
        st.w_voff[ii] = (unsigned)lane * 16u + (unsigned)((it.cbl * C::TAPS + it.tap) * 2 + it.ks) * 1024u;
    }
#pragma unroll
    for (int pp = 0; pp < C::PA; ++pp) {
        const int pc = wave + NW * pp;
        const bool live = pc < C::PIECES;
        const int cbl = pc / C::PPC, pr = pc - cbl * C::PPC;
        const int row = pr * 8 + (lane >> 3);
        const unsigned chunk = (unsigned)((lane & 7) ^ ((row >> 1) & 7));
        const bool ok = live && row < src_rows;
        st.a_voff[pp] = ok ? (unsigned)(src_row0 + row) * (unsigned)(p.c0 * 2) + chunk * 16u : kSkOob;
        st.a_voff1[pp] = (C::DUAL && ok) ? (unsigned)(src_row0 + row) * (unsigned)(p.c1 * 2) + chunk * 16u : kSkOob;
        st.a_dst[pp] = live ? cbl * C::CBS + pr * 1024 : -1;
        st.a_cbl[pp] = cbl;
    }

    SkEpi<T, NW, MT, NT> epi;         // (its operands are requested now when the register budget allows)
    if constexpr (SkEpi<T, NW, MT, NT>::EPRE) {
#pragma unroll
        for (int k = 0; k < SkEpi<T, NW, MT, NT>::EIT; ++k) epi.fetch(p, k, tid, tile0, row0, rows);
    }

    f32x4 acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    u32x4 wr[DS][ITEMS][NT];          // the weight ring: stage s lives in wr[s % DS]

    __builtin_amdgcn_s_waitcnt(0xC07F);               // lgkmcnt(0): the zero rows (the first barrier below publishes them)

    // ---- the stage loop.  VMEM returns IN ORDER per wave, so a wait for the activation pieces of stage s also waits for every weight
    // fragment issued before them: the activation ring is as deep as the weight ring (DS slots) and both run DS stages ahead; stage s
    // refills the slot of stage s-1, which every wave has left at the stage-s barrier.
    // The barrier puts the waves of a workgroup in lockstep, and in lockstep their phases ADD UP instead of overlapping: per stage all
    // waves burst their ds_reads (54 x 5 cycles of LDS), then their MFMAs, then their 27 VMEM instructions through the CU's one address
    // path (16 cycles each): s_memtime stamps showed 350-570 cycles of "compute" and up to 470 of "issue" in a 1330-cycle stage.  So the
    // waves run in TWO GROUPS with the phases swapped: group A (waves 0-3, 8) computes stage s and then issues its refills, group B (waves
    // 4-7: the SIMD partners of 0-3) issues its refills first and computes second -- while one group is on the LDS / matrix pipes the other
    // is on the memory path.  Group B's refills are the ones group A would have issued at the end of the previous stage:
    //   A: W(0) | A(0) W(1) | ... | A(DS-2) W(DS-1) | [stage s: compute(s), A(s+DS-1), W(s+DS)]      wait at the top: (DS-2)(PA+I NT) + I NT
    //   B: A(0) W(0) | ... | A(DS-2) W(DS-2)         | [stage s: A(s+DS-1), W(s+DS-1), compute(s)]    wait at the top: (DS-2)(PA+I NT)
    // (what may still be in flight when A(s) and W(s) must have landed).
    // The loop runs in whole rounds of DS stages (static ring indices, ONE back edge, no exit inside a round: with a per-stage exit hipcc
    // merges the exits with the back edge and the merged scoreboard drains the ring at the top of every round).  Stages past the end of
    // K read zeros on both sides (out-of-range offsets) and add nothing; DS is chosen to divide the usual stage counts.
#ifdef MVLDM_SK_TRACE
#define SK_STAMP(k_) if (p.trace && blockIdx.x == 8 && lane == 0 && s < 40) p.trace[(wave * 40 + s) * 8 + (k_)] = __builtin_readcyclecounter();
#else
#define SK_STAMP(k_)
#endif
    // one stage's MFMAs from ring slot j.  Units of <= 6 row tiles of one item, software-pipelined one unit ahead: the fragment reads of
    // unit u+1 are ISSUED before the MFMAs of unit u (sched_barrier pins it: left alone, hipcc sinks every ds_read next to its MFMA)
#define SK_COMPUTE(j_)                                                                                                      \
    {                                                                                                                       \
        const char* sb = smem + (j_) * C::SLOT;                                                                             \
        constexpr int NH = (MT + 5) / 6, RC = (MT + NH - 1) / NH, NU = ITEMS * NH;                                          \
        Frag bq[2][RC];                                                                                                     \
        _Pragma("unroll") for (int r = 0; r < RC; ++r) bq[0][r] = *reinterpret_cast<const Frag*>(sb + frag_off(0, r));      \
        _Pragma("unroll") for (int u = 0; u < NU; ++u) {                                                                    \
            if (u + 1 < NU) {                                                                                               \
                const int ii1 = (u + 1) / NH, h1 = (u + 1) % NH;                                                            \
                _Pragma("unroll") for (int r = 0; r < RC; ++r)                                                              \
                    if (h1 * RC + r < MT) bq[(u + 1) & 1][r] = *reinterpret_cast<const Frag*>(sb + frag_off(ii1, h1 * RC + r)); \
            }                                                                                                               \
            __builtin_amdgcn_sched_barrier(0);                                                                              \
            const int ii = u / NH, h = u % NH;                                                                              \
            _Pragma("unroll") for (int r = 0; r < RC; ++r)                                                                  \
                if (h * RC + r < MT) {                                                                                      \
                    _Pragma("unroll") for (int jn = 0; jn < NT; ++jn)                                                       \
                        acc[h * RC + r][jn] = SkMma<T>::mma(*reinterpret_cast<Frag*>(&wr[j_][ii][jn]), bq[u & 1][r], acc[h * RC + r][jn]); \
                }                                                                                                           \
            __builtin_amdgcn_sched_barrier(0);                                                                              \
        }                                                                                                                   \
    }
    auto frag_off = [&](int ii, int i) -> int {
        if constexpr (MT <= 6) return foff[ii][i];
        else return it_off[ii] + (aoff[i] ^ it_x[ii]);
    };
    const int n_rounds = (p.n_stages + DS - 1) / DS;
    const bool grp_b = C::STAGGER && wave >= 4 && wave < 8;
    if (!grp_b) {
        sk_issue_w<C>(p, st, wr[0], tile0);
#pragma unroll
        for (int v = 1; v < DS; ++v) {
            sk_issue_a<C>(p, st, smem, smem + (v - 1) * C::SLOT, v - 1);
            sk_issue_w<C>(p, st, wr[v], tile0);
        }
        for (int rnd = 0; rnd < n_rounds; ++rnd) {
#pragma unroll
            for (int j = 0; j < DS; ++j) {
                const int s = rnd * DS + j;
                SK_STAMP(0)
                __builtin_amdgcn_s_waitcnt(sk_vm(C::WAIT));
                SK_STAMP(1)
                __builtin_amdgcn_s_barrier();         // A(s) of every wave is in LDS; the slot of stage s-1 has been read by every wave
                SK_STAMP(2)
                SK_COMPUTE(j)
                SK_STAMP(3)
                sk_issue_a<C>(p, st, smem, smem + ((j + DS - 1) % DS) * C::SLOT, s + DS - 1);
                sk_issue_w<C>(p, st, wr[j], tile0);
                SK_STAMP(4)
            }
        }
    } else {
#pragma unroll
        for (int v = 0; v < DS - 1; ++v) {
            sk_issue_a<C>(p, st, smem, smem + v * C::SLOT, v);
            sk_issue_w<C>(p, st, wr[v], tile0);
        }
        for (int rnd = 0; rnd < n_rounds; ++rnd) {
#pragma unroll
            for (int j = 0; j < DS; ++j) {
                const int s = rnd * DS + j;
                SK_STAMP(0)
                __builtin_amdgcn_s_waitcnt(sk_vm(C::WAIT_B));
                SK_STAMP(1)
                __builtin_amdgcn_s_barrier();
                SK_STAMP(2)
                sk_issue_a<C>(p, st, smem, smem + ((j + DS - 1) % DS) * C::SLOT, s + DS - 1);
                sk_issue_w<C>(p, st, wr[(j + DS - 1) % DS], tile0);
                SK_STAMP(3)
                SK_COMPUTE(j)
                SK_STAMP(4)
            }
        }
    }
#undef SK_COMPUTE
#ifdef MVLDM_EXPERIMENTS
    if (p.fake & 16) return;
#endif
    // ---- fold the waves' partial tiles through LDS (the ring is dead) in wave order, then bias / time-embedding row / activation /
    // residual (all requested at the top of the kernel) and the store
    __builtin_amdgcn_s_waitcnt(sk_vm(0));             // the out-of-range refills of the last stages (they write zeros into the ring)
    __syncthreads();
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) *reinterpret_cast<f32x4*>(smem + ((wave * MT + i) * NT + j) * 1024 + lane * 16) = acc[i][j];
    __syncthreads();

    epi.finish(p, smem, tid, tile0, row0, rows);
}

// =====================================================================================================================================
// Independent wave streams (the second form of tile 15).  The kernel above shares one activation block between the waves of a workgroup,
// which costs a barrier per stage -- and behind a barrier the waves run in lockstep: their LDS reads, their MFMAs and their VMEM issue
// each arrive as one burst and ADD UP (s_memtime stamps: a 6-MFMA stage takes ~1300 cycles; the youngest wave, which loses every
// arbitration, is the critical path).  Here a wave never meets another one until the final fold:
//   * K is dealt to the waves by 64-channel block (wave w owns blocks w, w + NW, ...); for its block a wave computes ALL taps: the
//     per-lane tap addresses are a register table (TAPS x MT), the block's weights are one contiguous 2 TAPS KB run per column tile;
//   * each wave stages ITS blocks of the activations into ITS OWN LDS ring by LDS-DMA and reads them back after its own counted
//     s_waitcnt vmcnt -- no barrier, nothing shared, so the waves drift apart and their LDS / matrix / memory phases overlap by themselves;
//   * the weight ring is refilled fragment by fragment: the load of (step q + DS, item it) is issued right behind the MFMAs of
//     (step q, item it), which spreads the VMEM issue between the MFMAs instead of bursting it.
// VMEM issue order of a wave: W(0,*) A(0) | W(1,*) A(1) | ... | W(DS-1,*) A(DS-1) | [step 0] W(DS,0) .. W(DS,I-1) A(DS) | [step 1] ...
// At the top of step q the pieces A(q) (and the older W(q,*)) must have landed: (DS-1) (I NT + PPC) younger operations may be in flight.
template <int TAPS_, int NW_, int MT_, int NT_, int DS_, int SM_, bool DUAL_ = false> struct IwCfg {
    static constexpr int TAPS = TAPS_, NW = NW_, MT = MT_, NT = NT_, DS = DS_, SM = SM_, SC = 1;
    static constexpr bool DUAL = DUAL_;
    static constexpr int SRP = MT * 16 * SM;
    static constexpr int SLOT = (SRP + 8) * 128;             // one channel block of the wave's images + 8 rows of zeros
    static constexpr int WLDS = DS * SLOT;                   // a wave's ring
    static constexpr int RING = NW * WLDS;
    static constexpr int PPC = SRP / 8;                      // 1 KB LDS-DMA pieces per channel block (all issued by the owning wave)
    static constexpr int ITEMS = 2 * TAPS;                   // (tap, k-step) items of a step
    static constexpr int PARK = NW * MT * NT * 1024;
    static constexpr int SMEM = RING > PARK ? RING : PARK;
    static constexpr int WAIT = (DS - 1) * (ITEMS * NT + PPC);
    static_assert(DS >= 2 && WAIT <= 63, "ring depth / wait count");
    static_assert(SMEM <= 160 * 1024, "LDS");
};

template <typename C>
__device__ __forceinline__ void iw_issue_a(const SkParams& p, const unsigned (&a_voff)[C::PPC], const unsigned (&a_voff1)[C::PPC], char* slot, int cb) {
    const bool in_k = cb < p.n_cbs;                   // steps past the end of K stage zeros
    const bool from0 = !C::DUAL || cb < p.cb0;
    // one descriptor from selected scalars (a select between two descriptors becomes a branch whose join drains the queue)
    const void* base = from0 ? p.src0 : p.src1;
    const unsigned bytes = from0 ? p.src0_bytes : p.src1_bytes;
    const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, bytes, 0x00020000);
    const int soff = (from0 ? cb : cb - p.cb0) * 128;
#pragma unroll
    for (int pc = 0; pc < C::PPC; ++pc) {
        const unsigned v = from0 ? a_voff[pc] : a_voff1[pc];
        __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)(slot + pc * 1024), 16, in_k ? v : kSkOob, soff, 0, 0);
    }
}

// one fragment per column tile: (step's channel block cb, item it) = 1 KB at ((nt * ksteps + cb * ITEMS + it) * 1024)
template <typename C>
__device__ __forceinline__ void iw_issue_w(const SkParams& p, u32x4 (&wf)[C::NT], int cb, int it, int lane, int tile0) {
    const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.w), 0, p.w_bytes, 0x00020000);
    const bool in_k = cb < p.n_cbs;
#pragma unroll
    for (int j = 0; j < C::NT; ++j) {
        const int nt = tile0 + j;
        const unsigned soff = ((unsigned)nt * (unsigned)p.ksteps + (unsigned)(cb * C::ITEMS + it)) * 1024u;
        wf[j] = __builtin_amdgcn_raw_buffer_load_b128(rw, (in_k && nt < p.n_tiles) ? (unsigned)lane * 16u : kSkOob, soff, 0);
    }
}

template <typename T, typename C>
__global__ __launch_bounds__(C::NW * 64) void skinny_iws_kernel(const SkParams p) {
    using Frag = typename SkMma<T>::Frag;
    constexpr int MT = C::MT, NT = C::NT, NW = C::NW, DS = C::DS, ITEMS = C::ITEMS, TAPS = C::TAPS;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

    const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
    const int mg = idx % p.m_groups, ng = (idx / p.m_groups) * 8 + xcd;
    if (ng >= p.n_groups) return;
#ifdef MVLDM_EXPERIMENTS
    if (p.fake & 8) return;
#endif
    const int tile0 = ng * NT;
    const int img0 = mg * p.G, imgs = min(p.G, p.n_img - img0);
    const int src_row0 = img0 * p.hw_in, src_rows = imgs * p.hw_in;
    const int row0 = img0 * p.hw_out, rows = imgs * p.hw_out;
    char* wb = smem + wave * C::WLDS;                 // this wave's ring

    // the rows of zeros behind every slot of the wave's ring (never overwritten by the DMA)
#pragma unroll
    for (int d = 0; d < DS; ++d) *reinterpret_cast<u32x4*>(wb + d * C::SLOT + C::SRP * 128 + lane * 16) = u32x4{0u, 0u, 0u, 0u};

    // per-lane fragment addresses: (tap, row tile) -> LDS byte offset of (shifted source pixel, 16-byte chunk lane >> 4) of k-step 0
    // inside a slot; k-step 1 = the same ^ 64 (the swizzle is an XOR of the chunk index); padding taps point at the rows of zeros
    int aoff[TAPS][MT];
    {
        int p_il[MT], p_oy[MT], p_ox[MT];             // (image, y, x) of the lane's output pixel in each row tile: once, not per tap
#pragma unroll
        for (int i = 0; i < MT; ++i) {
            const int r = i * 16 + (lane & 15);
            const int il = p.hw_sh >= 0 ? r >> p.hw_sh : r / p.hw_out, rem = r - il * p.hw_out;
            const int oy = p.w_sh >= 0 ? rem >> p.w_sh : rem / p.w_out;
            p_il[i] = r < rows ? il : -1; p_oy[i] = oy * p.stride + p.ty0; p_ox[i] = (rem - oy * p.w_out) * p.stride + p.tx0;
        }
#pragma unroll
        for (int tap = 0; tap < TAPS; ++tap) {
            const int ty = tap / p.ksize, tx = tap - ty * p.ksize;
#pragma unroll
            for (int i = 0; i < MT; ++i) {
                const int iy = p_oy[i] + ty, ix = p_ox[i] + tx;
                const bool ok = p_il[i] >= 0 && iy >= 0 && iy < p.h_in && ix >= 0 && ix < p.w_in;
                const int srow = ok ? p_il[i] * p.hw_in + iy * p.w_in + ix : C::SRP;
                aoff[tap][i] = srow * 128 + ((((lane >> 4) ^ (srow >> 1)) & 7) << 4);
            }
        }
    }
    // per-lane source offsets of the LDS-DMA pieces (row, swizzled chunk; the channel block rides in the scalar offset)
    unsigned a_voff[C::PPC], a_voff1[C::PPC];
#pragma unroll
    for (int pc = 0; pc < C::PPC; ++pc) {
        const int row = pc * 8 + (lane >> 3);
        const unsigned chunk = (unsigned)((lane & 7) ^ ((row >> 1) & 7));
        const bool ok = row < src_rows;
        a_voff[pc] = ok ? (unsigned)(src_row0 + row) * (unsigned)(p.c0 * 2) + chunk * 16u : kSkOob;
        a_voff1[pc] = (C::DUAL && ok) ? (unsigned)(src_row0 + row) * (unsigned)(p.c1 * 2) + chunk * 16u : kSkOob;
    }

    SkEpi<T, NW, MT, NT> epi;
    if constexpr (SkEpi<T, NW, MT, NT>::EPRE) {
#pragma unroll
        for (int k = 0; k < SkEpi<T, NW, MT, NT>::EIT; ++k) epi.fetch(p, k, tid, tile0, row0, rows);
    }

    f32x4 acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    u32x4 wr[DS][ITEMS][NT];          // the weight ring: step q lives in wr[q % DS]

    __builtin_amdgcn_s_waitcnt(0xC07F);               // lgkmcnt(0): the zero rows (same wave: program order is enough after this)

    // prologue: DS steps in flight
#pragma unroll
    for (int v = 0; v < DS; ++v) {
        const int cb = wave + v * NW;
#pragma unroll
        for (int it = 0; it < ITEMS; ++it) iw_issue_w<C>(p, wr[v][it], cb, it, lane, tile0);
        iw_issue_a<C>(p, a_voff, a_voff1, wb + v * C::SLOT, cb);
    }
    // whole rounds of DS steps (static ring indices, one back edge); steps past the wave's last channel block read zeros on both sides
    const int n_steps = (p.n_cbs + NW - 1) / NW;
    const int n_rounds = (n_steps + DS - 1) / DS;
    for (int rnd = 0; rnd < n_rounds; ++rnd) {
#pragma unroll
        for (int j = 0; j < DS; ++j) {
            const int q = rnd * DS + j;
            const int cb_next = wave + (q + DS) * NW;
            __builtin_amdgcn_s_waitcnt(sk_vm(C::WAIT));       // A(q) -- and the older W(q, *) -- have landed
            const char* sb = wb + j * C::SLOT;
            // items software-pipelined one ahead: the fragment reads of item it+1 are issued before the MFMAs of item it (sched_barrier
            // pins it: left alone hipcc sinks every ds_read next to its MFMA, which then waits a full LDS round trip)
            Frag bq[2][MT];
#pragma unroll
            for (int i = 0; i < MT; ++i) bq[0][i] = *reinterpret_cast<const Frag*>(sb + aoff[0][i]);
#pragma unroll
            for (int it = 0; it < ITEMS; ++it) {
                if (it + 1 < ITEMS) {
                    const int tap1 = (it + 1) >> 1, x1 = ((it + 1) & 1) * 64;
#pragma unroll
                    for (int i = 0; i < MT; ++i) bq[(it + 1) & 1][i] = *reinterpret_cast<const Frag*>(sb + (aoff[tap1 < TAPS ? tap1 : 0][i] ^ x1));
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int i = 0; i < MT; ++i)
#pragma unroll
                    for (int jn = 0; jn < NT; ++jn) acc[i][jn] = SkMma<T>::mma(*reinterpret_cast<Frag*>(&wr[j][it][jn]), bq[it & 1][i], acc[i][jn]);
                __builtin_amdgcn_sched_barrier(0);
                iw_issue_w<C>(p, wr[j][it], cb_next, it, lane, tile0);
            }
            // the slot is free: every read of it has been consumed by an MFMA above
            iw_issue_a<C>(p, a_voff, a_voff1, wb + j * C::SLOT, cb_next);
        }
    }
#ifdef MVLDM_EXPERIMENTS
    if (p.fake & 16) return;
#endif
    __builtin_amdgcn_s_waitcnt(sk_vm(0));             // the out-of-range refills of the last steps (they write zeros into the ring)
    __syncthreads();                                  // the rings are dead: the partial tiles are parked over them
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) *reinterpret_cast<f32x4*>(smem + ((wave * MT + i) * NT + j) * 1024 + lane * 16) = acc[i][j];
    __syncthreads();
    epi.finish(p, smem, tid, tile0, row0, rows);
}

// ---- the fragment-order pack: dst 16-byte unit ((nt * ksteps + kg) * 64 + lane) = src unit [row(nt, lane & 15)][4 kg + (lane >> 4)] of the
// K-major [n_pad][k_pad] pack (k_order 1).  GEGLU: tiles (2q, 2q + 1) = value / gate columns of output columns 16 q .. 16 q + 15 (the
// K-major pack alternates blocks of 32 value / 32 gate rows)
__global__ __launch_bounds__(256) void pack_skinny_kernel(const u32x4* __restrict__ src, u32x4* __restrict__ dst, int n_tiles, int ksteps, int geglu) {
    const size_t total = (size_t)n_tiles * ksteps * 64;
    for (size_t u = (size_t)blockIdx.x * 256 + threadIdx.x; u < total; u += (size_t)gridDim.x * 256) {
        const int lane = (int)(u & 63);
        const size_t f = u >> 6;
        const int kg = (int)(f % ksteps), nt = (int)(f / ksteps);
        int row;
        if (geglu) { const int q = nt >> 1, gate = nt & 1; row = (q >> 1) * 64 + gate * 32 + (q & 1) * 16 + (lane & 15); }
        else row = nt * 16 + (lane & 15);
        dst[u] = src[(size_t)row * (ksteps * 4) + 4 * kg + (lane >> 4)];
    }
}

// ---- host side ------------------------------------------------------------------------------------
// configurations (desc.tile bits 8-13; 0 = the rule below).  TAPS / waves / channel blocks per stage / row tiles / column tiles / ring
// depth / source-row multiple
struct SkinnyCfgInfo { int taps, nw, sc, mt, nt, ds, sm, bal; };     // (sc = 0: the independent-wave-streams form)
static const SkinnyCfgInfo kSkCfgs[] = {
    {0, 0, 0, 0, 0, 0, 0, 0},
    {9, 9, 1, 3, 1, 10, 1, 0},   // 1: 3x3, 48 rows (three 4x4 images / ...)
    {9, 9, 1, 4, 1, 10, 1, 0},   // 2: 3x3, 64 rows (one 8x8 image)
    {0, 0, 0, 0, 0, 0, 0, 0}, {0, 0, 0, 0, 0, 0, 0, 0},
    {9, 9, 1, 3, 1, 5, 4, 0},    // 5: 3x3 stride 2, 48 output rows from 192 source rows
    {4, 8, 2, 3, 1, 5, 1, 0},    // 6: 2x2 phase conv, 48 rows
    {4, 8, 2, 4, 1, 5, 1, 0},    // 7: 2x2 phase conv, 64 rows
    {1, 8, 4, 3, 4, 5, 1, 0},    // 8: Linear / 1x1, 48 rows x 64 columns
    {1, 8, 4, 3, 2, 5, 1, 0},    // 9: Linear / 1x1, 48 rows x 32 columns
    {1, 8, 4, 1, 4, 5, 1, 0},    // 10: Linear / 1x1, 16 rows x 64 columns
    {0, 0, 0, 0, 0, 0, 0, 0},
    {1, 8, 4, 3, 1, 5, 1, 0},    // 12: Linear / 1x1, 48 rows x 16 columns
    {1, 4, 2, 6, 2, 5, 1, 0},    // 13: Linear / 1x1, 96 rows x 32 columns (4 waves)
    {0, 0, 0, 0, 0, 0, 0, 0}, {0, 0, 0, 0, 0, 0, 0, 0},
    {0, 0, 0, 0, 0, 0, 0, 0},
    {0, 0, 0, 0, 0, 0, 0, 0},
    {9, 9, 1, 12, 1, 3, 1, 0},   // 18: 3x3, 192 rows (three 8x8 images)
    {4, 8, 2, 12, 1, 3, 1, 0},   // 19: 2x2 phase conv, 192 rows
    {1, 8, 4, 1, 2, 5, 1, 0},    // 20: Linear / 1x1, 16 rows x 32 columns (the time embedding: 9 rows)
    {0, 0, 0, 0, 0, 0, 0, 0}, {0, 0, 0, 0, 0, 0, 0, 0}, {0, 0, 0, 0, 0, 0, 0, 0}, {0, 0, 0, 0, 0, 0, 0, 0}, {0, 0, 0, 0, 0, 0, 0, 0}, {0, 0, 0, 0, 0, 0, 0, 0},
    {0, 0, 0, 0, 0, 0, 0, 0}, {0, 0, 0, 0, 0, 0, 0, 0}, {0, 0, 0, 0, 0, 0, 0, 0}, {0, 0, 0, 0, 0, 0, 0, 0}, {0, 0, 0, 0, 0, 0, 0, 0},     // 21 - 31: unused
    // independent wave streams (no barrier in the loop; sc = 0): taps / waves / - / row tiles / column tiles / ring depth / source-row multiple.
    // (3x3 and phase forms of it were built and lose to the shared-block kernel: 16.3 / 17.6 us with 4 / 8 waves against 14.7 on the
    //  4x4-level conv -- with one tap table per wave a wave's own LDS latency is exposed -- so only the 1x1 / Linear forms are kept)
    {0, 0, 0, 0, 0, 0, 0, 0}, {0, 0, 0, 0, 0, 0, 0, 0}, {0, 0, 0, 0, 0, 0, 0, 0}, {0, 0, 0, 0, 0, 0, 0, 0}, {0, 0, 0, 0, 0, 0, 0, 0}, {0, 0, 0, 0, 0, 0, 0, 0},     // 32 - 37
    {1, 4, 0, 3, 1, 5, 1, 0},    // 38: Linear / 1x1, 48 rows x 16 columns, 4 waves
    {1, 4, 0, 3, 2, 5, 1, 0},    // 39: Linear / 1x1, 48 rows x 32 columns
    {1, 4, 0, 3, 4, 5, 1, 0},    // 40: Linear / 1x1, 48 rows x 64 columns
    {1, 4, 0, 1, 2, 5, 1, 0},    // 41: Linear / 1x1, 16 rows x 32 columns
    {1, 4, 0, 1, 4, 5, 1, 0},    // 42: Linear / 1x1, 16 rows x 64 columns
    {1, 8, 0, 3, 1, 2, 1, 0},    // 43: Linear / 1x1, 48 rows x 16 columns, 8 waves
    {1, 4, 0, 6, 2, 2, 1, 0},    // 44: Linear / 1x1, 96 rows x 32 columns
    {0, 0, 0, 0, 0, 0, 0, 0},
    {1, 8, 0, 3, 2, 2, 1, 0},    // 46: Linear / 1x1, 48 rows x 32 columns, 8 waves
};
constexpr int kNumSkCfgs = sizeof(kSkCfgs) / sizeof(kSkCfgs[0]);

template <typename T, typename C> static int sk_launch(const SkParams& p, int grid, hipStream_t s) {
    static std::atomic<uint64_t> done{0};
    if (int rc0 = ensure_dyn_smem(reinterpret_cast<const void*>(skinny_kernel<T, C>), C::SMEM, done)) return rc0;
    hipLaunchKernelGGL((skinny_kernel<T, C>), dim3(grid), dim3(C::NW * 64), C::SMEM, s, p);
    return check_launch();
}

template <typename T, typename C> static int iw_launch(const SkParams& p, int grid, hipStream_t s) {
    static std::atomic<uint64_t> done{0};
    if (int rc0 = ensure_dyn_smem(reinterpret_cast<const void*>(skinny_iws_kernel<T, C>), C::SMEM, done)) return rc0;
    hipLaunchKernelGGL((skinny_iws_kernel<T, C>), dim3(grid), dim3(C::NW * 64), C::SMEM, s, p);
    return check_launch();
}

template <typename T> static int sk_dispatch(int cfg, const SkParams& p, int grid, hipStream_t s) {
    if (p.c1 > 0) {       // two sources: the 1x1 / Linear configurations only
        switch (cfg) {
            case 38: return iw_launch<T, IwCfg<1, 4, 3, 1, 5, 1, true>>(p, grid, s);
            case 39: return iw_launch<T, IwCfg<1, 4, 3, 2, 5, 1, true>>(p, grid, s);
            case 40: return iw_launch<T, IwCfg<1, 4, 3, 4, 5, 1, true>>(p, grid, s);
            case 44: return iw_launch<T, IwCfg<1, 4, 6, 2, 2, 1, true>>(p, grid, s);
            case 8: return sk_launch<T, SkCfg<1, 8, 4, 3, 4, 5, 1, true>>(p, grid, s);
            case 9: return sk_launch<T, SkCfg<1, 8, 4, 3, 2, 5, 1, true>>(p, grid, s);
            case 10: return sk_launch<T, SkCfg<1, 8, 4, 1, 4, 5, 1, true>>(p, grid, s);
            case 12: return sk_launch<T, SkCfg<1, 8, 4, 3, 1, 5, 1, true>>(p, grid, s);
            case 13: return sk_launch<T, SkCfg<1, 4, 2, 6, 2, 5, 1, true>>(p, grid, s);
            case 20: return sk_launch<T, SkCfg<1, 8, 4, 1, 2, 5, 1, true>>(p, grid, s);
            default: return set_error(MVLDM_ERR_UNSUPPORTED, "igemm: tile 15: configuration %d does not take two sources", cfg);
        }
    }
    switch (cfg) {
        case 1: return sk_launch<T, SkCfg<9, 9, 1, 3, 1, 10, 1>>(p, grid, s);
        case 2: return sk_launch<T, SkCfg<9, 9, 1, 4, 1, 10, 1>>(p, grid, s);
        case 5: return sk_launch<T, SkCfg<9, 9, 1, 3, 1, 5, 4>>(p, grid, s);
        case 6: return sk_launch<T, SkCfg<4, 8, 2, 3, 1, 5, 1>>(p, grid, s);
        case 7: return sk_launch<T, SkCfg<4, 8, 2, 4, 1, 5, 1>>(p, grid, s);
        case 8: return sk_launch<T, SkCfg<1, 8, 4, 3, 4, 5, 1>>(p, grid, s);
        case 9: return sk_launch<T, SkCfg<1, 8, 4, 3, 2, 5, 1>>(p, grid, s);
        case 10: return sk_launch<T, SkCfg<1, 8, 4, 1, 4, 5, 1>>(p, grid, s);
        case 12: return sk_launch<T, SkCfg<1, 8, 4, 3, 1, 5, 1>>(p, grid, s);
        case 13: return sk_launch<T, SkCfg<1, 4, 2, 6, 2, 5, 1>>(p, grid, s);
        case 18: return sk_launch<T, SkCfg<9, 9, 1, 12, 1, 3, 1>>(p, grid, s);
        case 19: return sk_launch<T, SkCfg<4, 8, 2, 12, 1, 3, 1>>(p, grid, s);
        case 20: return sk_launch<T, SkCfg<1, 8, 4, 1, 2, 5, 1>>(p, grid, s);
        case 38: return iw_launch<T, IwCfg<1, 4, 3, 1, 5, 1>>(p, grid, s);
        case 39: return iw_launch<T, IwCfg<1, 4, 3, 2, 5, 1>>(p, grid, s);
        case 40: return iw_launch<T, IwCfg<1, 4, 3, 4, 5, 1>>(p, grid, s);
        case 41: return iw_launch<T, IwCfg<1, 4, 1, 2, 5, 1>>(p, grid, s);
        case 42: return iw_launch<T, IwCfg<1, 4, 1, 4, 5, 1>>(p, grid, s);
        case 43: return iw_launch<T, IwCfg<1, 8, 3, 1, 2, 1>>(p, grid, s);
        case 44: return iw_launch<T, IwCfg<1, 4, 6, 2, 2, 1>>(p, grid, s);
        case 46: return iw_launch<T, IwCfg<1, 8, 3, 2, 2, 1>>(p, grid, s);
        default: return set_error(MVLDM_ERR_ARG, "igemm: tile 15: bad configuration %d", cfg);
    }
}

// a 1x1 / stride-1 conv works per pixel: its "images" are single rows (any 16 MT rows form a group, whatever the image size)
struct SkGeom { int n_img, h_in, w_in, h_out, w_out; };
static SkGeom sk_geom(const mvldm_igemm_desc& d) {
    if (d.ksize == 1 && d.stride == 1 && d.pad == 0 && d.h_in == d.h_out && d.w_in == d.w_out) return {d.n_img * d.h_out * d.w_out, 1, 1, 1, 1};
    return {d.n_img, d.h_in, d.w_in, d.h_out, d.w_out};
}

// does configuration `c` compute problem `d`?  (whole images per workgroup: the row tiles of a group hold G = 16 MT / hw_out images)
static bool sk_cfg_fits(const SkinnyCfgInfo& c, const mvldm_igemm_desc& d) {
    const SkGeom g = sk_geom(d);
    const int taps = d.ksize * d.ksize, hw_out = g.h_out * g.w_out, hw_in = g.h_in * g.w_in;
    if (c.taps != taps) return false;
    const int rows = c.mt * 16;
    if (hw_out > rows || rows % hw_out) return false;
    const int G = rows / hw_out;
    if (G * hw_in > rows * c.sm) return false;
    if (d.epilogue == MVLDM_EPI_GEGLU && (c.nt & 1)) return false;
    if (c.taps == 0) return false;
    if (d.c1 > 0) {           // two sources: configurations 8, 9, 10, 12, 13, 20 and 38, 39, 40, 44
        const bool sk2 = c.sc > 0 && c.taps == 1 && ((c.nw == 8 && c.sc == 4 && (c.mt == 3 || c.mt == 1)) || (c.nw == 4 && c.mt == 6));
        const bool iw2 = c.sc == 0 && c.taps == 1 && c.nw == 4 && (c.mt == 3 || c.mt == 6) && !(c.mt == 1);
        if (!sk2 && !iw2) return false;
    }
    return true;
}

bool skinny_applicable(const mvldm_igemm_desc& d) {
    if (d.act_dtype == MVLDM_F32 || d.k_order != 2) return false;
    if (!(d.ksize == 1 || d.ksize == 3 || (d.ksize == 2 && d.upsample >= 2)) || (d.upsample == 1)) return false;
    if (d.ksize != 2 && d.upsample != 0) return false;
    if (d.c0 % 64 || d.c1 % 64 || d.c0 <= 0 || (d.c1 == 0) != (d.src1 == nullptr)) return false;
    if (d.k_pad != d.ksize * d.ksize * (d.c0 + d.c1) || d.n_pad % 64 || d.n_out % 4 || d.n_out > d.n_pad) return false;
    if (d.dst_dtype != d.act_dtype && d.dst_dtype != MVLDM_F32) return false;
    if (d.epilogue == MVLDM_EPI_GEGLU && (d.n_out % 64 || d.row_bias || d.n_out != d.n_pad)) return false;
    const int n_dst = d.epilogue == MVLDM_EPI_GEGLU ? d.n_out / 2 : d.n_out;
    const int dst_ld = d.dst_ld > 0 ? d.dst_ld : n_dst;
    if (dst_ld % 4 || dst_ld < n_dst) return false;
    if (d.bias && ((uintptr_t)d.bias % 16)) return false;
    if (d.row_bias && (((uintptr_t)d.row_bias % 16) || d.row_bias_ld % 4)) return false;
    if (((uintptr_t)d.dst % 16) || ((uintptr_t)d.weight % 16) || (d.residual && ((uintptr_t)d.residual % 8))) return false;
    if (((uintptr_t)d.src0 % 16) || (d.src1 && ((uintptr_t)d.src1 % 16))) return false;
    const double px = (double)d.n_img * d.h_in * d.w_in;
    if (px * d.c0 * 2.0 >= 4.0e9 || px * d.c1 * 2.0 >= 4.0e9 || (double)d.n_pad * d.k_pad * 2.0 >= 4.0e9) return false;
    if (d.residual && d.upsample >= 2) return false;
    return true;
}

// the rule (no plan-time tuning): the configuration that won the shape class in tools/skinny_bench.py (profiles/r05_skinny_bench.json)
static int sk_choose(const mvldm_igemm_desc& d) {
    const SkGeom g = sk_geom(d);
    const int hw = g.h_out * g.w_out, taps = d.ksize * d.ksize, M = g.n_img * hw;
    int pref[6] = {0, 0, 0, 0, 0, 0};
    if (taps == 9) {
        if (d.stride == 2) { pref[0] = 5; }
        else if (hw > 48) { pref[0] = g.n_img >= 2 ? 18 : 2; pref[1] = 2; pref[2] = 18; }
        else { pref[0] = 1; pref[1] = 18; pref[2] = 2; }
    } else if (taps == 4) {
        if (hw > 48) { pref[0] = g.n_img >= 2 ? 19 : 7; pref[1] = 7; pref[2] = 19; }
        else { pref[0] = 6; pref[1] = 7; pref[2] = 19; }
    } else {
        if (M <= 16) { pref[0] = 20; pref[1] = 41; pref[2] = 10; }
        else if (M <= 160) {
            if (d.n_pad >= 2560) { pref[0] = 8; pref[1] = 40; }
            else { pref[0] = 38; pref[1] = 12; }
            pref[2] = 9;
        } else {
            if (d.n_pad <= 1280) { pref[0] = 44; pref[1] = 13; pref[2] = 9; }
            else { pref[0] = 8; pref[1] = 40; }
        }
        pref[4] = 12; pref[5] = 8;
    }
    for (int c : pref)
        if (c > 0 && c < kNumSkCfgs && sk_cfg_fits(kSkCfgs[c], d)) return c;
    for (int c = 1; c < kNumSkCfgs; ++c)
        if (sk_cfg_fits(kSkCfgs[c], d)) return c;
    return 0;
}

int skinny_run(const mvldm_igemm_desc& d, hipStream_t s) {
    MVLDM_REQUIRE(d.src0 && d.weight && d.dst, "igemm: null pointer");
    MVLDM_REQUIRE(skinny_applicable(d), "igemm: tile 15 (skinny weight-streaming GEMM) does not apply to this problem (needs the fragment-order "
                                        "pack, k_order 2, 16-bit activations, channels in multiples of 64, ksize 1 / 3 or a 2x2 phase conv)");
    int cfg = (d.tile >> 8) & 63;
    if (cfg == 0) cfg = sk_choose(d);
    MVLDM_REQUIRE(cfg > 0 && cfg < kNumSkCfgs && sk_cfg_fits(kSkCfgs[cfg], d), "igemm: tile 15: configuration %d does not compute this problem", cfg);
    const SkinnyCfgInfo& k = kSkCfgs[cfg];
    SkParams p;
    const bool phase = d.upsample >= 2;
    p.src0 = d.src0; p.src1 = d.src1; p.w = d.weight; p.bias = d.bias; p.row_bias = d.row_bias; p.residual = d.residual; p.dst = d.dst;
    p.c0 = d.c0; p.c1 = d.c1;
    const SkGeom g = sk_geom(d);
    p.n_img = g.n_img; p.h_in = g.h_in; p.w_in = g.w_in; p.hw_in = g.h_in * g.w_in; p.h_out = g.h_out; p.w_out = g.w_out; p.hw_out = g.h_out * g.w_out;
    p.rb_hw = d.h_out * d.w_out;
    auto lg2 = [](int v) { int sh = 0; while ((1 << sh) < v) ++sh; return (1 << sh) == v ? sh : -1; };
    p.hw_sh = lg2(p.hw_out); p.w_sh = lg2(p.w_out); p.rb_sh = lg2(p.rb_hw);
    p.stride = d.stride; p.ksize = d.ksize;
    p.scatter = phase; p.ph_y = phase ? (d.upsample - 2) >> 1 : 0; p.ph_x = phase ? (d.upsample - 2) & 1 : 0;
    p.ty0 = phase ? p.ph_y - 1 : -d.pad; p.tx0 = phase ? p.ph_x - 1 : -d.pad;
    p.M = p.n_img * p.hw_out; p.n_out = d.n_out;
    p.n_dst = d.epilogue == MVLDM_EPI_GEGLU ? d.n_out / 2 : d.n_out;
    p.dst_ld = d.dst_ld > 0 ? d.dst_ld : p.n_dst;
    p.row_bias_ld = d.row_bias_ld; p.epilogue = d.epilogue; p.dst_f32 = d.dst_dtype == MVLDM_F32; p.out_scale = d.out_scale;
    p.n_tiles = d.n_pad / 16;
    p.n_groups = (p.n_tiles + k.nt - 1) / k.nt;
    p.G = k.mt * 16 / p.hw_out;
    p.m_groups = (p.n_img + p.G - 1) / p.G;
    p.n_cbs = (d.c0 + d.c1) / 64; p.cb0 = d.c0 / 64;
    p.n_stages = k.sc ? (p.n_cbs + k.sc - 1) / k.sc : p.n_cbs;
    p.ksteps = p.n_cbs * k.taps * 2;
    const double px = (double)d.n_img * d.h_in * d.w_in;
    p.src0_bytes = (unsigned)(px * d.c0 * 2.0); p.src1_bytes = (unsigned)(px * d.c1 * 2.0);
    p.w_bytes = (unsigned)((double)d.n_pad * d.k_pad * 2.0);
    p.fake = kSkFake;
    p.trace = nullptr;
#ifdef MVLDM_SK_TRACE
    static unsigned long long* trace = nullptr;
    if (getenv("MVLDM_SK_TRACE")) {
        if (!trace) hipHostMalloc((void**)&trace, 16 * 40 * 8 * 8, hipHostMallocMapped);
        memset(trace, 0, 16 * 40 * 8 * 8);
        p.trace = trace;
    }
#endif
    if (kSkFake & 1) p.w_bytes = 0;
    if (kSkFake & 2) p.src0_bytes = p.src1_bytes = 0;
    if (kSkFake & 4) p.n_stages = 0;
    if (p.M == 0) return MVLDM_OK;
    const int grid = 8 * ((p.n_groups + 7) / 8) * p.m_groups;
    const int rc = dispatch_dtype(d.act_dtype, [&](auto t) -> int {
        using T = decltype(t);
        if constexpr (sizeof(T) == 2) return sk_dispatch<T>(cfg, p, grid, s);
        else return set_error(MVLDM_ERR_ARG, "igemm: tile 15 needs a 16-bit activation type");
    });
#ifdef MVLDM_SK_TRACE
    if (p.trace && getenv("MVLDM_SK_TRACE_DUMP")) {
        hipDeviceSynchronize();
        const unsigned long long t0 = p.trace[0];
        for (int w = 0; w < k.nw; ++w) {
            fprintf(stderr, "wave %d:", w);
            for (int st = 0; st < std::min(p.n_stages, 24); ++st) {
                const unsigned long long* e = p.trace + (w * 40 + st) * 8;
                fprintf(stderr, " [s%d +%llu w%llu b%llu c%llu i%llu]", st, e[0] - t0, e[1] - e[0], e[2] - e[1], e[3] - e[2], e[4] - e[3]);
            }
            fprintf(stderr, "\n");
        }
    }
#endif
    return rc;
}

}  // namespace mvldm

using namespace mvldm;

extern "C" int mvldm_igemm_skinny_config(const mvldm_igemm_desc* d) {
    if (!d) return 0;
    mvldm_igemm_desc t = *d;
    t.k_order = 2;                // the question is about the problem, not about which pack the caller holds at the moment
    if (!t.weight) t.weight = t.src0;
    if (!skinny_applicable(t)) return 0;
    return sk_choose(t);
}

extern "C" int mvldm_pack_skinny(const void* packed, void* dst, int n_pad, int k_pad, int geglu, int dtype, mvldm_stream_t stream) {
    MVLDM_REQUIRE(packed && dst, "pack_skinny: null pointer");
    MVLDM_REQUIRE(dtype == MVLDM_BF16 || dtype == MVLDM_F16, "pack_skinny: 16-bit types only");
    MVLDM_REQUIRE(n_pad > 0 && n_pad % 64 == 0 && k_pad > 0 && k_pad % 64 == 0, "pack_skinny: n_pad %d / k_pad %d must be multiples of 64", n_pad, k_pad);
    const int n_tiles = n_pad / 16, ksteps = k_pad / 32;
    const size_t total = (size_t)n_tiles * ksteps * 64;
    const int blocks = (int)std::min<size_t>((total + 255) / 256, 8192);
    hipLaunchKernelGGL(pack_skinny_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, reinterpret_cast<const u32x4*>(packed),
                       reinterpret_cast<u32x4*>(dst), n_tiles, ksteps, geglu);
    return check_launch();
}
