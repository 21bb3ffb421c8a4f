// Plan executor: a whole forward as a flat op list, run without touching Python, optionally captured
// into a hipGraph (include/mvldm.h, "Plans").
#include <vector>

#include "common.h"

namespace mvldm {
int igemm_run(const mvldm_igemm_desc& d, hipStream_t s);
int groupnorm_run(const void* x, const void* x1, void* y, const float* gamma, const float* beta, int n_img, int hw, int c0,
                  int c1, int groups, float eps, int silu, int dtype, void* stats_ws, float* stats_out, hipStream_t s);
int layernorm_run(const void* x, void* y, const float* gamma, const float* beta, int rows, int c, float eps, int dtype,
                  hipStream_t s);
int attention_run(const void* q, const void* k, const void* v, void* out, int ld_q, int ld_k, int ld_v, int ld_o,
                  int heads, int head_dim, const int32_t* seg, int n_seg, int max_q_len, float scale, int dtype,
                  float* lse, int lse_ld, hipStream_t s);
int temb_run(const int64_t* ts, const float* freqs, void* out, int n, int dim, int flip, int dst_dtype, hipStream_t s);
int eltwise_run(const void* x, void* y, size_t n, int op, int src_dtype, int dst_dtype, hipStream_t s);
int ddim_run(const float* eps, const float* x_t, float* x_next, const int32_t* cond_img, const int32_t* uncond_img,
             int n_tgt, int hw, int c, float cfg_scale, const float* coef, const int32_t* step_ptr, void* unet_in,
             int unet_in_c, int unet_in_dtype, int n_steps, float clip_range, hipStream_t s);
int advance_run(int32_t* step_ptr, const int64_t* t_table, int n_steps, int64_t* timesteps, const int32_t* tgt_rows,
                int n_rows, hipStream_t s);
int to_nhwc_run(const float* src, void* dst, int n_img, int c, int hw, int dst_c, int dst_c_off, int dst_dtype, float scale,
                float shift, const int32_t* img_map, hipStream_t s);
int ray_run(const float* extr, const float* intr, int n_cam, int h, int w, float* out_nchw, void* out_nhwc, int nhwc_c,
            int nhwc_c_off, int nhwc_dtype, const int32_t* img_map, int mode, int no, int nd, int plucker, hipStream_t s);
int posterior_run(const float* moments, const float* noise, float* out, int n, int c, int hw, float scale, hipStream_t s);
int wgrad_run(const mvldm_wgrad_desc& d, hipStream_t s);
int attention_bwd_run(const mvldm_attn_bwd_desc& a, hipStream_t s);
int groupnorm_bwd_run(const void* x0, const void* x1, const void* dy, void* dx0, void* dx1, const float* gamma, const float* beta,
                      const float* stats, float* dgamma, float* dbeta, int n_img, int hw, int c0, int c1, int groups, int silu, int dtype,
                      float* ws, size_t ws_bytes, hipStream_t s);
int layernorm_bwd_run(const void* x, const void* dy, void* dx, const float* gamma, float* dgamma, float* dbeta, int rows, int c, float eps, int dtype,
                      float* ws, size_t ws_bytes, hipStream_t s);
int colsum_run(const void* x, float* dst, float* ws, size_t ws_bytes, int n_seg, int rows_per_seg, int n, int ld, int ld_dst, int per_seg,
               int accumulate, int dtype, hipStream_t s);
int train_eltwise_run(int op, const void* a, const void* b, void* out, size_t rows, int d, int a_dtype, int dtype, hipStream_t s);
int pool2x2_run(const void* du, void* dx, int n_img, int h, int w, int c, int dtype, hipStream_t s);
int zero_insert_run(const void* x, void* out, int n_img, int h, int w, int c, int dtype, hipStream_t s);
int add_noise_run(const float* x0, const float* noise, const float* coef, void* dst, int n, int c, int hw, int dst_c, int dst_c_off, int dst_dtype,
                  const int32_t* img_map, hipStream_t s);
int mse_run(const float* pred, const float* noise, const int32_t* tgt_img, int n_tgt, int hw, int c, float* loss, int accumulate, float loss_scale,
            void* dpred, int dc, int dtype, float grad_scale, double* ws, hipStream_t s);
int to_nchw_run(const void* src, float* dst, int n_img, int c, int hw, int src_c, int src_c_off, int src_dtype, float scale,
                float shift, int clamp01, hipStream_t s);
int gather_rows_run(const void* src, void* dst, const int32_t* src_index, const int32_t* dst_index, int n_rows, size_t row_bytes, hipStream_t s);
int attention_merge_run(const void* oa, const float* lse_a, const void* ob, const float* lse_b, void* out, const int32_t* a_img,
                        const int32_t* b_img, const int32_t* out_img, int n_img, int tokens, int heads, int head_dim, int ld_a, int ld_b,
                        int ld_o, int lse_ld_a, int lse_ld_b, int dtype, hipStream_t s);

static int run_op(const mvldm_op& op, hipStream_t s) {
    switch (op.kind) {
        case MVLDM_OP_IGEMM: return igemm_run(op.u.igemm, s);
        case MVLDM_OP_GROUPNORM: {
            const auto& g = op.u.groupnorm;
            return groupnorm_run(g.x, g.x1, g.y, g.gamma, g.beta, g.n_img, g.hw, g.c0, g.c1, g.groups, g.eps, g.silu, g.dtype,
                                 g.stats_ws, g.stats_out, s);
        }
        case MVLDM_OP_LAYERNORM: {
            const auto& l = op.u.layernorm;
            return layernorm_run(l.x, l.y, l.gamma, l.beta, l.rows, l.c, l.eps, l.dtype, s);
        }
        case MVLDM_OP_ATTENTION: {
            const auto& a = op.u.attention;
            return attention_run(a.q, a.k, a.v, a.out, a.ld_q, a.ld_k, a.ld_v, a.ld_o, a.heads, a.head_dim, a.seg, a.n_seg,
                                 a.max_q_len, a.scale, a.dtype, a.lse, a.lse_ld, s);
        }
        case MVLDM_OP_TIMESTEP_EMBED: {
            const auto& t = op.u.temb;
            return temb_run(t.timesteps, t.freqs, t.out, t.n, t.dim, t.flip, t.dst_dtype, s);
        }
        case MVLDM_OP_ELTWISE: {
            const auto& e = op.u.eltwise;
            return eltwise_run(e.x, e.y, e.n, e.op, e.src_dtype, e.dst_dtype, s);
        }
        case MVLDM_OP_DDIM_STEP: {
            const auto& d = op.u.ddim;
            return ddim_run(d.eps, d.x_t, d.x_next, d.cond_img, d.uncond_img, d.n_tgt, d.hw, d.c, d.cfg_scale, d.coef,
                            d.step_ptr, d.unet_in, d.unet_in_c, d.unet_in_dtype, d.n_steps, d.clip_range, s);
        }
        case MVLDM_OP_DDIM_ADVANCE: {
            const auto& a = op.u.advance;
            return advance_run(a.step_ptr, a.t_table, a.n_steps, a.timesteps, a.tgt_rows, a.n_rows, s);
        }
        case MVLDM_OP_NCHW_TO_NHWC: {
            const auto& l = op.u.layout;
            return to_nhwc_run(reinterpret_cast<const float*>(l.src), l.dst, l.n_img, l.c, l.hw, l.other_c, l.other_c_off, l.dtype,
                               l.scale, l.shift, l.img_map, s);
        }
        case MVLDM_OP_NHWC_TO_NCHW: {
            const auto& l = op.u.layout;
            return to_nchw_run(l.src, reinterpret_cast<float*>(l.dst), l.n_img, l.c, l.hw, l.other_c, l.other_c_off, l.dtype,
                               l.scale, l.shift, l.clamp01, s);
        }
        case MVLDM_OP_RAY_ENCODE: {
            const auto& r = op.u.rays;
            return ray_run(r.extrinsics, r.intrinsics, r.n_cam, r.h, r.w, r.out_nchw, r.out_nhwc, r.nhwc_c, r.nhwc_c_off,
                           r.nhwc_dtype, r.img_map, r.mode, r.n_origin_octaves, r.n_dir_octaves, r.plucker, s);
        }
        case MVLDM_OP_POSTERIOR_SAMPLE: {
            const auto& q = op.u.posterior;
            return posterior_run(q.moments, q.noise, q.out, q.n, q.c, q.hw, q.scale, s);
        }
        case MVLDM_OP_WGRAD: return wgrad_run(op.u.wgrad, s);
        case MVLDM_OP_ATTENTION_BWD: return attention_bwd_run(op.u.attention_bwd, s);
        case MVLDM_OP_GROUPNORM_BWD: {
            const auto& g = op.u.groupnorm_bwd;
            return groupnorm_bwd_run(g.x0, g.x1, g.dy, g.dx0, g.dx1, g.gamma, g.beta, g.stats, g.dgamma, g.dbeta, g.n_img, g.hw, g.c0, g.c1,
                                     g.groups, g.silu, g.dtype, g.workspace, g.workspace_bytes, s);
        }
        case MVLDM_OP_LAYERNORM_BWD: {
            const auto& l = op.u.layernorm_bwd;
            return layernorm_bwd_run(l.x, l.dy, l.dx, l.gamma, l.dgamma, l.dbeta, l.rows, l.c, l.eps, l.dtype, l.workspace, l.workspace_bytes, s);
        }
        case MVLDM_OP_COLSUM: {
            const auto& c = op.u.colsum;
            return colsum_run(c.x, c.dst, c.workspace, c.workspace_bytes, c.n_seg, c.rows_per_seg, c.n, c.ld, c.ld_dst, c.per_seg, c.accumulate,
                              c.dtype, s);
        }
        case MVLDM_OP_TRAIN_ELTWISE: {
            const auto& e = op.u.train_eltwise;
            return train_eltwise_run(e.op, e.a, e.b, e.out, e.rows, e.d, e.a_dtype, e.dtype, s);
        }
        case MVLDM_OP_POOL2X2: {
            const auto& r = op.u.resample;
            return pool2x2_run(r.src, r.dst, r.n_img, r.h, r.w, r.c, r.dtype, s);
        }
        case MVLDM_OP_ZERO_INSERT: {
            const auto& r = op.u.resample;
            return zero_insert_run(r.src, r.dst, r.n_img, r.h, r.w, r.c, r.dtype, s);
        }
        case MVLDM_OP_ADD_NOISE: {
            const auto& a = op.u.add_noise;
            return add_noise_run(a.x0, a.noise, a.coef, a.dst, a.n, a.c, a.hw, a.dst_c, a.dst_c_off, a.dst_dtype, a.img_map, s);
        }
        case MVLDM_OP_MSE_LOSS: {
            const auto& m = op.u.mse;
            return mse_run(m.pred, m.noise, m.tgt_img, m.n_tgt, m.hw, m.c, m.loss, m.accumulate, m.loss_scale, m.dpred, m.dpred_c, m.dpred_dtype,
                           m.grad_scale, m.workspace, s);
        }
        case MVLDM_OP_FILL_ZERO: {
            const auto& f = op.u.fill;
            if (f.bytes == 0) return MVLDM_OK;
            MVLDM_CHECK_HIP(hipMemsetAsync(f.dst, 0, f.bytes, s));
            return MVLDM_OK;
        }
        case MVLDM_OP_MEMCPY: {
            const auto& m = op.u.memcpy_;
            if (m.bytes == 0) return MVLDM_OK;
            MVLDM_CHECK_HIP(hipMemcpyAsync(m.dst, m.src, m.bytes, hipMemcpyDeviceToDevice, s));
            return MVLDM_OK;
        }
        case MVLDM_OP_GATHER_ROWS: {
            const auto& g = op.u.gather;
            return gather_rows_run(g.src, g.dst, g.src_index, g.dst_index, g.n_rows, g.row_bytes, s);
        }
        case MVLDM_OP_ATTN_MERGE: {
            const auto& m = op.u.attn_merge;
            return attention_merge_run(m.oa, m.lse_a, m.ob, m.lse_b, m.out, m.a_img, m.b_img, m.out_img, m.n_img, m.tokens, m.heads, m.head_dim,
                                       m.ld_a, m.ld_b, m.ld_o, m.lse_ld_a, m.lse_ld_b, m.dtype, s);
        }
        case MVLDM_OP_PAR_BEGIN:
        case MVLDM_OP_PAR_NEXT:
        case MVLDM_OP_PAR_END: return MVLDM_OK;      // lane markers: serial execution is always valid
        default: return set_error(MVLDM_ERR_ARG, "plan: unknown op kind %d", op.kind);
    }
}
}  // namespace mvldm

struct mvldm_plan {
    std::vector<mvldm_op> ops;
    hipGraph_t graph = nullptr;
    hipGraphExec_t exec = nullptr;
    // parallel lanes (MVLDM_OP_PAR_*): side streams and the fork / join events, created on first use
    std::vector<hipStream_t> side;
    std::vector<hipEvent_t> join;
    hipEvent_t fork = nullptr;
};

static int ensure_lane(mvldm_plan* p, int lane) {      // lane >= 1 -> p->side[lane - 1]
    if (!p->fork) MVLDM_CHECK_HIP(hipEventCreateWithFlags(&p->fork, hipEventDisableTiming));
    while ((int)p->side.size() < lane) {
        hipStream_t st = nullptr;
        hipEvent_t ev = nullptr;
        MVLDM_CHECK_HIP(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
        MVLDM_CHECK_HIP(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
        p->side.push_back(st);
        p->join.push_back(ev);
    }
    return MVLDM_OK;
}

using namespace mvldm;

extern "C" int mvldm_op_run(const mvldm_op* op, mvldm_stream_t stream) {
    MVLDM_REQUIRE(op, "op_run: null op");
    return run_op(*op, (hipStream_t)stream);
}

extern "C" int mvldm_plan_create(const mvldm_op* ops, int n_ops, mvldm_plan** out) {
    MVLDM_REQUIRE(ops && out && n_ops >= 0, "plan_create: bad arguments");
    mvldm_plan* p = new mvldm_plan();
    p->ops.assign(ops, ops + n_ops);
    *out = p;
    return MVLDM_OK;
}

extern "C" int mvldm_plan_num_ops(const mvldm_plan* p) { return p ? (int)p->ops.size() : 0; }

extern "C" int mvldm_plan_run_range(mvldm_plan* p, int first, int last, mvldm_stream_t stream) {
    MVLDM_REQUIRE(p && first >= 0 && last <= (int)p->ops.size() && first <= last, "plan_run_range: bad range");
    hipStream_t s = (hipStream_t)stream, cur = s;
    static const bool serial = knob_int("MVLDM_PLAN_SERIAL", 0) != 0;      // A/B knob: ignore the lane markers
    bool in_group = false;
    int lane = 0;
    if (!serial) {
        // validate the lane markers of the range BEFORE anything is launched: a cut inside a parallel group used to be noticed only
        // after earlier ops (or the group's lanes) had been enqueued, leaving side-stream work unordered against `s`
        bool open = false;
        for (int i = first; i < last; ++i) {
            const int k = p->ops[i].kind;
            if (k == MVLDM_OP_PAR_BEGIN) {
                MVLDM_REQUIRE(!open, "plan: nested parallel group at op %d", i);
                open = true;
            } else if (k == MVLDM_OP_PAR_NEXT || k == MVLDM_OP_PAR_END) {
                MVLDM_REQUIRE(open, "plan: the range [%d, %d) starts inside a parallel group", first, last);
                if (k == MVLDM_OP_PAR_END) open = false;
            }
        }
        MVLDM_REQUIRE(!open, "plan: the range [%d, %d) ends inside a parallel group", first, last);
    }
    for (int i = first; i < last; ++i) {
        const mvldm_op& op = p->ops[i];
        if (!serial && op.kind == MVLDM_OP_PAR_BEGIN) {
            MVLDM_REQUIRE(!in_group, "plan: nested parallel group at op %d", i);
            if (int rc = ensure_lane(p, 0)) return rc;
            MVLDM_CHECK_HIP(hipEventRecord(p->fork, s));
            in_group = true;
            lane = 0;
            cur = s;
        } else if (!serial && op.kind == MVLDM_OP_PAR_NEXT) {
            MVLDM_REQUIRE(in_group, "plan: the range [%d, %d) starts inside a parallel group", first, last);
            ++lane;
            if (int rc = ensure_lane(p, lane)) return rc;
            cur = p->side[lane - 1];
            MVLDM_CHECK_HIP(hipStreamWaitEvent(cur, p->fork, 0));
        } else if (!serial && op.kind == MVLDM_OP_PAR_END) {
            MVLDM_REQUIRE(in_group, "plan: the range [%d, %d) starts inside a parallel group", first, last);
            for (int l = 1; l <= lane; ++l) {
                MVLDM_CHECK_HIP(hipEventRecord(p->join[l - 1], p->side[l - 1]));
                MVLDM_CHECK_HIP(hipStreamWaitEvent(s, p->join[l - 1], 0));
            }
            in_group = false;
            cur = s;
        } else {
            int rc = run_op(op, cur);
            if (rc) {
                if (in_group)      // leave no side stream dangling (a capture in progress must still be joinable)
                    for (int l = 1; l <= lane; ++l)
                        if (hipEventRecord(p->join[l - 1], p->side[l - 1]) == hipSuccess) (void)hipStreamWaitEvent(s, p->join[l - 1], 0);
                return rc;
            }
        }
    }
    MVLDM_REQUIRE(!in_group, "plan: the range [%d, %d) ends inside a parallel group", first, last);
    return MVLDM_OK;
}

extern "C" int mvldm_plan_run(mvldm_plan* p, mvldm_stream_t stream) {
    MVLDM_REQUIRE(p, "plan_run: null plan");
    return mvldm_plan_run_range(p, 0, (int)p->ops.size(), stream);
}

extern "C" int mvldm_plan_capture(mvldm_plan* p, mvldm_stream_t stream) {
    MVLDM_REQUIRE(p && stream, "plan_capture: needs a non-default stream");
    hipStream_t s = (hipStream_t)stream;
    if (p->exec) { hipGraphExecDestroy(p->exec); p->exec = nullptr; }
    if (p->graph) { hipGraphDestroy(p->graph); p->graph = nullptr; }
    // one eager pass first so that lazily-set function attributes are in place before capture
    int rc = mvldm_plan_run(p, stream);
    if (rc) return rc;
    MVLDM_CHECK_HIP(hipStreamSynchronize(s));
    MVLDM_CHECK_HIP(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
    rc = mvldm_plan_run(p, stream);
    hipGraph_t g = nullptr;
    hipError_t e = hipStreamEndCapture(s, &g);
    if (rc) { if (g) hipGraphDestroy(g); return rc; }
    if (e != hipSuccess) return set_error(MVLDM_ERR_HIP, "hipStreamEndCapture: %s", hipGetErrorString(e));
    p->graph = g;
    MVLDM_CHECK_HIP(hipGraphInstantiate(&p->exec, g, nullptr, nullptr, 0));
    return MVLDM_OK;
}

extern "C" int mvldm_plan_replay(mvldm_plan* p, mvldm_stream_t stream) {
    MVLDM_REQUIRE(p && p->exec, "plan_replay: plan was not captured");
    MVLDM_CHECK_HIP(hipGraphLaunch(p->exec, (hipStream_t)stream));
    return MVLDM_OK;
}

extern "C" int mvldm_plan_profile(mvldm_plan* p, mvldm_stream_t stream, int iters, float* per_op_ms) {
    MVLDM_REQUIRE(p && per_op_ms && iters > 0, "plan_profile: bad arguments");
    hipStream_t s = (hipStream_t)stream;
    const int n = (int)p->ops.size();
    std::vector<hipEvent_t> ev(n + 1);
    for (auto& e : ev) MVLDM_CHECK_HIP(hipEventCreate(&e));
    std::vector<double> acc(n, 0.0);
    int rc = MVLDM_OK;
    for (int it = 0; it < iters && !rc; ++it) {
        MVLDM_CHECK_HIP(hipEventRecord(ev[0], s));
        for (int i = 0; i < n && !rc; ++i) {
            rc = run_op(p->ops[i], s);
            hipEventRecord(ev[i + 1], s);
        }
        MVLDM_CHECK_HIP(hipStreamSynchronize(s));
        for (int i = 0; i < n && !rc; ++i) {
            float ms = 0.f;
            hipEventElapsedTime(&ms, ev[i], ev[i + 1]);
            acc[i] += ms;
        }
    }
    for (auto& e : ev) hipEventDestroy(e);
    for (int i = 0; i < n; ++i) per_op_ms[i] = (float)(acc[i] / iters);
    return rc;
}

extern "C" void mvldm_plan_destroy(mvldm_plan* p) {
    if (!p) return;
    if (p->exec) hipGraphExecDestroy(p->exec);
    if (p->graph) hipGraphDestroy(p->graph);
    for (hipStream_t st : p->side) hipStreamDestroy(st);
    for (hipEvent_t ev : p->join) hipEventDestroy(ev);
    if (p->fork) hipEventDestroy(p->fork);
    delete p;
}
