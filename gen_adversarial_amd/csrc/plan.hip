// Plan replay, timing helper and library metadata.
#include "ga_common.h"
#include <vector>

namespace ga { thread_local hipError_t g_last_err = hipSuccess; }

static int run_one(const ga_op& op, void* stream) {
    switch (op.kind) {
        case GA_OP_CONV:         return ga_conv2d(&op.u.conv, stream);
        case GA_OP_DWCONV5:      return ga_dwconv5(&op.u.dw, stream);
        case GA_OP_REDUCE:       return ga_rowchan_reduce(&op.u.red, stream);
        case GA_OP_SE_EXCITE:    return ga_se_excite(&op.u.se, stream);
        case GA_OP_SE_APPLY:     return ga_se_apply(&op.u.app, stream);
        case GA_OP_BILINEAR_BWD: return ga_bilinear_up2_bwd(&op.u.bil, stream);
        case GA_OP_SAMPLER:      return ga_sampler_mix(&op.u.smp, stream);
        case GA_OP_DML:          return ga_dml_mean(&op.u.dml, stream);
        case GA_OP_MAXPOOL:      return ga_maxpool2(&op.u.mp, stream);
        case GA_OP_IMAGE_IO:     return ga_image_io(&op.u.io, stream);
        case GA_OP_REP_SUM:      return ga_rep_sum(op.u.rs.x, op.u.rs.y, op.u.rs.rows, op.u.rs.inner, op.u.rs.rep, op.u.rs.accumulate, stream);
        case GA_OP_BLUR:         return ga_gauss_blur(&op.u.blur, stream);
        case GA_OP_INTERLEAVE2:  return ga_interleave2(&op.u.il, stream);
        case GA_OP_MAXPOOL3S2:   return ga_maxpool3s2(&op.u.mp3, stream);
        case GA_OP_AVGPOOL_ACT:  return ga_avgpool_act(&op.u.ap, stream);
        case GA_OP_GCONV:        return ga_gconv(&op.u.gc, stream);
        case GA_OP_PRELU:        return ga_prelu(&op.u.pr, stream);
        case GA_OP_UNARY:        return ga_unary(&op.u.un, stream);
        case GA_OP_MODOUT:       return ga_modout(&op.u.mo, stream);
        case GA_OP_UP2_BLUR:     return ga_up2_blur(&op.u.ub, stream);
        case GA_OP_PIXELNORM:    return ga_pixelnorm(op.u.pn.x, op.u.pn.y, op.u.pn.rows, op.u.pn.C, stream);
        case GA_OP_LATENT_MIX:   return ga_latent_mix(&op.u.lm, stream);
        case GA_OP_POOL_DENORM:  return ga_pool_denorm(&op.u.pd, stream);
        case GA_OP_ATTN:         return ga_attn(&op.u.at, stream);
        case GA_OP_LAYERNORM:    return ga_layernorm(&op.u.ln, stream);
        case GA_OP_RESIZE2_CROP: return ga_resize2_crop(&op.u.rc, stream);
        case GA_OP_DEC_CELL:     return ga_dec_cell(&op.u.dc, stream);
        case GA_OP_AVAE:         return ga_avae(&op.u.av, stream);
        case GA_OP_DEC_CELL_HALO: return ga_dec_cell_halo(&op.u.dh, stream);
        case GA_OP_AXPBY:        return ga_axpby(op.u.ax.x, op.u.ax.y, op.u.ax.n, op.u.ax.alpha, op.u.ax.beta, stream);
        default:                 return GA_E_UNSUPPORTED;
    }
}

extern "C" int ga_plan_run(const ga_op* ops, int n, void* stream, int* failed_index) {
    if (!ops || n < 0) return GA_E_BADARG;
    for (int i = 0; i < n; ++i) {
        const int rc = run_one(ops[i], stream);
        if (rc != GA_OK) { if (failed_index) *failed_index = i; return rc; }
    }
    return GA_OK;
}

// Times `iters` replays of the plan with HIP events recorded on the plan's own stream.  When conv_ms is requested
// every GA_OP_CONV launch of ONE extra replay is bracketed by its own event pair (per-launch device time).
extern "C" int ga_plan_time(const ga_op* ops, int n, void* stream_, int iters, float* total_ms, float* conv_ms,
                            long* conv_launches) {
    if (!ops || n <= 0 || iters <= 0 || !total_ms) return GA_E_BADARG;
    hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
    hipEvent_t e0, e1;
    if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) return GA_E_LAUNCH;
    int rc = GA_OK;
    hipEventRecord(e0, stream);
    for (int it = 0; it < iters && rc == GA_OK; ++it) rc = ga_plan_run(ops, n, stream_, nullptr);
    hipEventRecord(e1, stream);
    hipEventSynchronize(e1);
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    *total_ms = ms / (float)iters;
    hipEventDestroy(e0); hipEventDestroy(e1);
    if (rc != GA_OK) return rc;
    if (conv_ms) {
        std::vector<hipEvent_t> ev;
        long nconv = 0;
        for (int i = 0; i < n; ++i) if (ops[i].kind == GA_OP_CONV) ++nconv;
        ev.resize(2 * nconv);
        for (auto& e : ev) hipEventCreate(&e);
        long k = 0;
        for (int i = 0; i < n && rc == GA_OK; ++i) {
            if (ops[i].kind == GA_OP_CONV) {
                hipEventRecord(ev[2 * k], stream);
                rc = run_one(ops[i], stream_);
                hipEventRecord(ev[2 * k + 1], stream);
                ++k;
            } else {
                rc = run_one(ops[i], stream_);
            }
        }
        hipStreamSynchronize(stream);
        double sum = 0.0;
        for (long j = 0; j < k; ++j) { float t = 0.f; hipEventElapsedTime(&t, ev[2 * j], ev[2 * j + 1]); sum += t; }
        for (auto& e : ev) hipEventDestroy(e);
        *conv_ms = (float)sum;
        if (conv_launches) *conv_launches = nconv;
    }
    return rc;
}

// per-op device time of one replay (event pair around every op) — profiling aid for tools/ and bench.py
extern "C" int ga_plan_profile(const ga_op* ops, int n, void* stream_, float* per_op_ms) {
    if (!ops || n <= 0 || !per_op_ms) return GA_E_BADARG;
    hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
    std::vector<hipEvent_t> ev(n + 1);
    for (auto& e : ev) hipEventCreate(&e);
    int rc = GA_OK;
    hipEventRecord(ev[0], stream);
    for (int i = 0; i < n && rc == GA_OK; ++i) {
        rc = run_one(ops[i], stream_);
        hipEventRecord(ev[i + 1], stream);
    }
    hipStreamSynchronize(stream);
    if (rc == GA_OK)
        for (int i = 0; i < n; ++i) hipEventElapsedTime(&per_op_ms[i], ev[i], ev[i + 1]);
    for (auto& e : ev) hipEventDestroy(e);
    return rc;
}

// ---- HIP graphs: one replay of a plan captured into an executable graph (the plan only enqueues kernels on one stream,
//      all pointers are baked into the descriptors, so the capture is a pure kernel chain).  The stream must not be the
//      NULL stream (capture is illegal there) and the plan must have run eagerly once (first launches set function
//      attributes).  Launching the graph costs one host call instead of ~700.
struct ga_graph_t { hipGraph_t graph; hipGraphExec_t exec; };

extern "C" int ga_graph_capture(const ga_op* ops, int n, void* stream_, void** out) {
    if (!ops || n <= 0 || !out || !stream_) return GA_E_BADARG;
    hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
    ga::clear_stale_error();
    if (hipStreamBeginCapture(stream, hipStreamCaptureModeThreadLocal) != hipSuccess) { ga::g_last_err = hipGetLastError(); return GA_E_LAUNCH; }
    const int rc = ga_plan_run(ops, n, stream_, nullptr);
    hipGraph_t graph = nullptr;
    const hipError_t e = hipStreamEndCapture(stream, &graph);
    if (rc != GA_OK) { if (graph) hipGraphDestroy(graph); return rc; }
    if (e != hipSuccess || !graph) { ga::g_last_err = e; return GA_E_LAUNCH; }
    hipGraphExec_t exec = nullptr;
    const hipError_t e2 = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
    if (e2 != hipSuccess) { hipGraphDestroy(graph); ga::g_last_err = e2; return GA_E_LAUNCH; }
    ga_graph_t* g = new ga_graph_t{graph, exec};
    *out = g;
    return GA_OK;
}

extern "C" int ga_graph_launch(void* graph, void* stream_) {
    if (!graph) return GA_E_BADARG;
    ga_graph_t* g = reinterpret_cast<ga_graph_t*>(graph);
    const hipError_t e = hipGraphLaunch(g->exec, reinterpret_cast<hipStream_t>(stream_));
    if (e != hipSuccess) { ga::g_last_err = e; return GA_E_LAUNCH; }
    return GA_OK;
}

extern "C" int ga_graph_destroy(void* graph) {
    if (!graph) return GA_E_BADARG;
    ga_graph_t* g = reinterpret_cast<ga_graph_t*>(graph);
    hipGraphExecDestroy(g->exec);
    hipGraphDestroy(g->graph);
    delete g;
    return GA_OK;
}

extern "C" const char* ga_last_hip_error(void) { return hipGetErrorString(ga::g_last_err); }
extern "C" int ga_abi_version(void) { return GA_ABI_VERSION; }
extern "C" unsigned long ga_sizeof_op(void) { return sizeof(ga_op); }
