// Step executor: replays the kernel list of one denoise step for n steps with a device-resident step
// counter (every per-step quantity — time-embedding row, posterior coefficients, noise slice, last-step
// mode — is a table lookup by that counter, so the launch sequence is identical every step and can be
// captured once in a HIP graph).  Mirrors the loop of DenoisingModel.forward_denoising
// (/root/reference/ddpm/models/diffusion_denoising.py:189-212) with no host sync inside.
#include "ccdm_common.h"

#include <map>
#include <vector>

namespace ccdm {

__global__ void k_step_set(int32_t* p, int32_t v) { *p = v; }
__global__ void k_step_inc(int32_t* p) { *p += 1; }

struct Op {
    int kind;   // 0 conv, 1 attention core, 2 statistics fold, 3 GroupNorm + qkv + attention core, 4 resample (updown ResBlocks)
    ccdm_conv_args conv;
    const float* qkv; float* out; int N, T, C, heads, order;
    const double* fin; double* fout; int S_in, S_out;
    ccdm_attn_block_args ab;
    ccdm_resample_args rs;
};

static int launch_op(const Op& op, hipStream_t s) {
    switch (op.kind) {
        case 0: return launch_conv(op.conv, s);
        case 1: return launch_attention(op.qkv, op.out, op.N, op.T, op.T, op.C, op.heads, op.order, s);
        case 2: return launch_stats_fold(op.fin, op.N, op.S_in, op.C, op.S_out, op.fout, s);
        case 3: return launch_attn_block(op.ab, s);
        default: return launch_resample(op.rs, s);
    }
}

}  // namespace ccdm

struct ccdm_engine {
    int32_t* step = nullptr;
    std::vector<ccdm::Op> ops;
    ccdm_post_args post{};
    bool has_post = false;
    // graph of one step (ops + epilogue + counter increment)
    hipGraph_t graph = nullptr;
    hipGraphExec_t exec = nullptr;
    bool graph_valid = false;
    int graph_with_epilogue = -1;
    // timing taps: HIP events around every launch of the tapped ops (op index -> events, launches recorded)
    struct Tap { std::vector<hipEvent_t> ev; int n = 0; };
    std::map<int, Tap> taps;
};

using namespace ccdm;

static void drop_graph(ccdm_engine* e) {
    if (e->exec) { (void)hipGraphExecDestroy(e->exec); e->exec = nullptr; }
    if (e->graph) { (void)hipGraphDestroy(e->graph); e->graph = nullptr; }
    e->graph_valid = false;
}

static int launch_step(ccdm_engine* e, int with_epilogue, hipStream_t s, bool profile) {
    for (size_t i = 0; i < e->ops.size(); ++i) {
        const Op& op = e->ops[i];
        ccdm_engine::Tap* tp = nullptr;
        if (profile) {
            auto it = e->taps.find((int)i);
            if (it != e->taps.end() && (size_t)(2 * it->second.n + 1) < it->second.ev.size()) tp = &it->second;
        }
        if (tp) (void)hipEventRecord(tp->ev[2 * tp->n], s);
        int rc = launch_op(op, s);
        if (tp) { (void)hipEventRecord(tp->ev[2 * tp->n + 1], s); tp->n++; }
        if (rc) return rc;
    }
    if (with_epilogue && e->has_post) {
        int rc = launch_posterior(e->post, s);
        if (rc) return rc;
    }
    hipLaunchKernelGGL(k_step_inc, dim3(1), dim3(1), 0, s, e->step);
    CCDM_CHECK_LAUNCH("step_inc");
    return 0;
}

extern "C" ccdm_engine* ccdm_engine_create(int32_t* step_counter) {
    if (!step_counter) { fail("engine_create: null step counter"); return nullptr; }
    ccdm_engine* e = new ccdm_engine();
    e->step = step_counter;
    return e;
}

extern "C" void ccdm_engine_destroy(ccdm_engine* e) {
    if (!e) return;
    drop_graph(e);
    for (auto& kv : e->taps)
        for (hipEvent_t ev : kv.second.ev) (void)hipEventDestroy(ev);
    delete e;
}

extern "C" int ccdm_engine_add_conv(ccdm_engine* e, const ccdm_conv_args* a) {
    CCDM_REQUIRE(e && a, "engine_add_conv: null");
    Op op{};
    op.kind = 0;
    op.conv = *a;
    op.conv.step_ptr = e->step;
    e->ops.push_back(op);
    drop_graph(e);
    return (int)e->ops.size() - 1;
}

extern "C" int ccdm_engine_add_attention(ccdm_engine* e, const float* qkv, float* out, int N, int T, int C, int heads, int order) {
    CCDM_REQUIRE(e && qkv && out, "engine_add_attention: null");
    Op op{};
    op.kind = 1;
    op.qkv = qkv; op.out = out; op.N = N; op.T = T; op.C = C; op.heads = heads; op.order = order;
    e->ops.push_back(op);
    drop_graph(e);
    return (int)e->ops.size() - 1;
}

extern "C" int ccdm_engine_add_norm_qkv_attention(ccdm_engine* e, const ccdm_attn_block_args* a) {
    CCDM_REQUIRE(e && a, "engine_add_norm_qkv_attention: null");
    CCDM_REQUIRE(attn_block_supported(a->T, a->C, a->heads), "engine_add_norm_qkv_attention: (T=%d, C=%d, heads=%d) is not built", a->T, a->C, a->heads);
    Op op{};
    op.kind = 3;
    op.ab = *a;
    e->ops.push_back(op);
    drop_graph(e);
    return (int)e->ops.size() - 1;
}

extern "C" int ccdm_engine_add_stats_fold(ccdm_engine* e, const double* in, int N, int S_in, int C, int S_out, double* out) {
    CCDM_REQUIRE(e && in && out, "engine_add_stats_fold: null");
    Op op{};
    op.kind = 2;
    op.fin = in; op.fout = out; op.N = N; op.S_in = S_in; op.C = C; op.S_out = S_out;
    e->ops.push_back(op);
    drop_graph(e);
    return (int)e->ops.size() - 1;
}

extern "C" int ccdm_engine_add_resample(ccdm_engine* e, const ccdm_resample_args* a) {
    CCDM_REQUIRE(e && a, "engine_add_resample: null");
    Op op{};
    op.kind = 4;
    op.rs = *a;
    e->ops.push_back(op);
    drop_graph(e);
    return (int)e->ops.size() - 1;
}

extern "C" int ccdm_engine_set_epilogue(ccdm_engine* e, const ccdm_post_args* a) {
    CCDM_REQUIRE(e && a, "engine_set_epilogue: null");
    e->post = *a;
    e->post.step_ptr = e->step;
    e->has_post = true;
    drop_graph(e);
    return 0;
}

extern "C" int ccdm_engine_num_ops(const ccdm_engine* e) { return e ? (int)e->ops.size() : -1; }

extern "C" int ccdm_engine_set_run(ccdm_engine* e, const float* noise, int64_t noise_step_stride, int32_t noise_row0, uint64_t philox_seed,
                                   uint32_t sample_offset, float* out_probs, int64_t* out_onehot, float* posterior_out) {
    CCDM_REQUIRE(e && e->has_post, "engine_set_run: no epilogue set");
    ccdm_post_args& p = e->post;
    const bool same = p.noise == noise && p.noise_step_stride == noise_step_stride && p.noise_row0 == noise_row0 && p.philox_seed == philox_seed &&
                      p.sample_offset == sample_offset && p.out_probs == out_probs && p.out_onehot == out_onehot &&
                      p.posterior_out == posterior_out;
    if (!same) drop_graph(e);
    p.noise = noise; p.noise_step_stride = noise_step_stride; p.noise_row0 = noise_row0; p.philox_seed = philox_seed; p.sample_offset = sample_offset;
    p.out_probs = out_probs; p.out_onehot = out_onehot; p.posterior_out = posterior_out;
    return 0;
}

extern "C" int ccdm_engine_run(ccdm_engine* e, int first_row, int n_steps, int with_epilogue, int use_graph, void* stream) {
    CCDM_REQUIRE(e, "engine_run: null engine");
    CCDM_REQUIRE(n_steps >= 0, "engine_run: n_steps=%d", n_steps);
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(k_step_set, dim3(1), dim3(1), 0, s, e->step, (int32_t)first_row);
    CCDM_CHECK_LAUNCH("step_set");
    const bool profile = !e->taps.empty();
    if (profile) {
        use_graph = 0;
        if (first_row == 0) for (auto& kv : e->taps) kv.second.n = 0;     // a new sampling run starts a new series
    }
    if (use_graph) {
        if (!e->graph_valid || e->graph_with_epilogue != with_epilogue) {
            drop_graph(e);
            hipError_t err = hipStreamBeginCapture(s, hipStreamCaptureModeRelaxed);
            if (err != hipSuccess) return fail("engine_run: BeginCapture: %s", hipGetErrorString(err));
            int rc = launch_step(e, with_epilogue, s, false);
            err = hipStreamEndCapture(s, &e->graph);
            if (rc) { if (e->graph) { (void)hipGraphDestroy(e->graph); e->graph = nullptr; } return rc; }
            if (err != hipSuccess) return fail("engine_run: EndCapture: %s", hipGetErrorString(err));
            err = hipGraphInstantiate(&e->exec, e->graph, nullptr, nullptr, 0);
            if (err != hipSuccess) return fail("engine_run: GraphInstantiate: %s", hipGetErrorString(err));
            e->graph_valid = true;
            e->graph_with_epilogue = with_epilogue;
        }
        for (int i = 0; i < n_steps; ++i) {
            hipError_t err = hipGraphLaunch(e->exec, s);
            if (err != hipSuccess) return fail("engine_run: GraphLaunch: %s", hipGetErrorString(err));
        }
        return 0;
    }
    for (int i = 0; i < n_steps; ++i) {
        int rc = launch_step(e, with_epilogue, s, profile);
        if (rc) return rc;
    }
    return 0;
}

extern "C" int ccdm_engine_profile_op(ccdm_engine* e, int op_index, int capacity) {
    CCDM_REQUIRE(e, "engine_profile_op: null engine");
    if (op_index < 0) {                                   // drop every tap
        for (auto& kv : e->taps)
            for (hipEvent_t ev : kv.second.ev) (void)hipEventDestroy(ev);
        e->taps.clear();
        return 0;
    }
    CCDM_REQUIRE(op_index < (int)e->ops.size(), "engine_profile_op: op %d out of range", op_index);
    ccdm_engine::Tap& t = e->taps[op_index];
    t.n = 0;
    while ((int)t.ev.size() < 2 * capacity) {
        hipEvent_t ev;
        if (hipEventCreate(&ev) != hipSuccess) return fail("engine_profile_op: hipEventCreate failed");
        t.ev.push_back(ev);
    }
    return 0;
}

extern "C" int ccdm_engine_profile_read(ccdm_engine* e, int op_index, double* mean_ms, double* min_ms, double* max_ms) {
    CCDM_REQUIRE(e, "engine_profile_read: null engine");
    auto it = e->taps.find(op_index);
    CCDM_REQUIRE(it != e->taps.end(), "engine_profile_read: op %d is not tapped", op_index);
    ccdm_engine::Tap& t = it->second;
    double sum = 0, mn = 1e30, mx = 0;
    int n = 0;
    for (int i = 0; i < t.n; ++i) {
        float ms = 0.f;
        if (hipEventSynchronize(t.ev[2 * i + 1]) != hipSuccess) continue;
        if (hipEventElapsedTime(&ms, t.ev[2 * i], t.ev[2 * i + 1]) != hipSuccess) continue;
        sum += ms; if (ms < mn) mn = ms; if (ms > mx) mx = ms; ++n;
    }
    if (mean_ms) *mean_ms = n ? sum / n : 0.0;
    if (min_ms) *min_ms = n ? mn : 0.0;
    if (max_ms) *max_ms = n ? mx : 0.0;
    return n;
}

extern "C" int ccdm_engine_input_absmax(ccdm_engine* e, float* out, void* stream) {
    CCDM_REQUIRE(e && out, "engine_input_absmax: null");
    for (size_t i = 0; i < e->ops.size(); ++i) {
        if (e->ops[i].kind != 0) continue;
        const int rc = launch_conv_input_absmax(e->ops[i].conv, out + i, (hipStream_t)stream);
        if (rc) return rc;
    }
    return 0;
}

extern "C" int ccdm_engine_describe_op(const ccdm_engine* e, int i, char* buf, int buflen) {
    CCDM_REQUIRE(e && buf && buflen > 0, "engine_describe_op: bad args");
    CCDM_REQUIRE(i >= 0 && i < (int)e->ops.size(), "engine_describe_op: op %d out of range", i);
    const Op& op = e->ops[i];
    if (op.kind == 0) {
        const ccdm_conv_args& a = op.conv;
        snprintf(buf, buflen, "conv%dx%d %d%s->%d in%dx%d out%dx%d s%d%s%s%s%s%s%s", a.ksize, a.ksize, a.C0 + a.C1,
                 a.C1 ? "(cat)" : "", a.Cout, a.Hin, a.Win, a.Hout, a.Wout, a.stride, a.up ? " up2x" : "",
                 a.stats0 ? " gn" : "", a.act ? " silu" : "", a.emb_off >= 0 ? " +emb" : "", a.resid ? " +res" : "",
                 a.out_stats ? " stats" : "");
    } else if (op.kind == 1) {
        snprintf(buf, buflen, "attention T=%d C=%d heads=%d order=%d", op.T, op.C, op.heads, op.order);
    } else if (op.kind == 2) {
        snprintf(buf, buflen, "stats fold %d -> %d slices, C=%d", op.S_in, op.S_out, op.C);
    } else if (op.kind == 3) {
        snprintf(buf, buflen, "norm+qkv+attention T=%d C=%d heads=%d", op.ab.T, op.ab.C, op.ab.heads);
    } else {
        snprintf(buf, buflen, "resample %s C=%d in%dx%d%s%s%s%s", op.rs.mode == CCDM_RESAMPLE_AVGPOOL2 ? "avgpool2" : "nearest-up2", op.rs.C, op.rs.Hin,
                 op.rs.Win, op.rs.stats ? " gn" : "", op.rs.act ? " silu" : "", op.rs.out_act ? " ->act" : "", op.rs.out_raw ? " ->raw" : "");
    }
    return 0;
}
