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
// the per-run epilogue fields travel as a kernel argument into their device-resident block (stream-ordered: launches already in
// flight have read the old values, later ones see the new)
__global__ void k_run_set(ccdm_post_run* dst, const ccdm_post_run v) { *dst = v; }
// largest |x| of a flat fp32 buffer (Inf if a value is not finite): the attention core's qkv operand in the range diagnostics
__global__ __launch_bounds__(256) void k_absmax(const float* __restrict__ x, size_t n4, float* out) {
    float m = 0.f;
    bool bad = false;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
        const float4 v = reinterpret_cast<const float4*>(x)[i];
        bad |= !(fabsf(v.x) <= 3.0e38f) | !(fabsf(v.y) <= 3.0e38f) | !(fabsf(v.z) <= 3.0e38f) | !(fabsf(v.w) <= 3.0e38f);
        m = fmaxf(m, fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w))));
    }
    if (bad) m = __builtin_inff();
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off));
    if ((threadIdx.x & 63) == 0) atomicMax(reinterpret_cast<unsigned*>(out), __float_as_uint(m));
}

struct Op {
    int kind;   // 0 conv, 1 attention core, 2 statistics fold, 3 GroupNorm + qkv + attention core, 4 resample (updown ResBlocks), 5 stem conv, 6 head conv + epilogue
    ccdm_conv_args conv;
    const float* qkv; float* out; int N, T, C, heads, order;
    const double* fin; double* fout; int S_in, S_out;
    ccdm_attn_block_args ab;
    ccdm_resample_args rs;
    ccdm_stem_args st;
    ccdm_head_args hd;
};

static int launch_op(const Op& op, hipStream_t s) {
    switch (op.kind) {
        case 6: return fail("engine: the head + epilogue op is launched with the epilogue's arguments");
        case 0: return launch_conv(op.conv, s);
        case 1: return launch_attention(op.qkv, op.out, op.N, op.T, op.T, op.C, op.heads, op.order, s);
        case 2: return launch_stats_fold(op.fin, op.N, op.S_in, op.C, op.S_out, op.fout, s);
        case 3: return launch_attn_block(op.ab, s);
        case 4: return launch_resample(op.rs, s);
        default: return launch_stem(op.st, s);
    }
}

}  // namespace ccdm

struct ccdm_engine {
    int32_t* step = nullptr;
    std::vector<ccdm::Op> ops;
    ccdm_post_args post{};
    bool has_post = false;
    // per-run epilogue fields: host copy, and the device block the epilogue kernel reads them from (NULL: they are kernel arguments)
    ccdm_post_run run{};
    ccdm_post_run* run_dev = nullptr;
    bool run_dirty = false;
    int next_row = 0;            // where the device step counter stands after the last run (first_row + n_steps)
    // graph of one step (ops + epilogue + counter increment)
    hipGraph_t graph = nullptr;
    hipGraphExec_t exec = nullptr;
    bool graph_valid = false;
    int graph_with_epilogue = -1;
    int graph_steps = 1;         // denoise steps per captured graph
    int exp_id = 0;              // creation index (experiments builds: per-stream op skipping)
    int captures = 0;            // how often the step has been captured and instantiated (tests: a new Philox key must not re-capture)
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

#ifdef CCDM_EXPERIMENTS
// sensitivity probe (experiments builds only; results are garbage): CCDM_SKIP_OPS="a-b,c,d-e" leaves those ops of the step out, to see
// how much of the step time — in whatever launch mode — a stage is worth before anyone rewrites its kernels.  CCDM_SKIP_OPS_B, when
// set, is the list for every second engine created (the second of two sub-batch streams): one stream can run only the
// full-resolution ops and the other only the low-resolution ones.
static int g_exp_engines = 0;
static bool exp_skip_op(int engine_id, size_t i) {
    static std::vector<std::pair<int, int>> ranges[2];
    static bool parsed = false;
    if (!parsed) {
        parsed = true;
        const char* va = getenv("CCDM_SKIP_OPS");
        const char* vb = getenv("CCDM_SKIP_OPS_B");
        const char* vs[2] = {va, vb ? vb : va};
        for (int q = 0; q < 2; ++q) {
            const char* v = vs[q];
            while (v && *v) {
                char* end;
                const int a = (int)strtol(v, &end, 10);
                int b = a;
                if (*end == '-') b = (int)strtol(end + 1, &end, 10);
                ranges[q].push_back({a, b});
                if (end == v) break;
                v = *end == ',' ? end + 1 : end;
            }
        }
    }
    for (auto& r : ranges[engine_id & 1]) if ((int)i >= r.first && (int)i <= r.second) return true;
    return false;
}
#endif

static int launch_step(ccdm_engine* e, int with_epilogue, hipStream_t s, bool profile) {
    for (size_t i = 0; i < e->ops.size(); ++i) {
        const Op& op = e->ops[i];
#ifdef CCDM_EXPERIMENTS
        if (exp_skip_op(e->exp_id, i)) continue;
#endif
        ccdm_engine::Tap* tp = nullptr;
        if (profile) {
            auto it = e->taps.find((int)i);
            if (it != e->taps.end() && (size_t)(2 * it->second.n + 1) < it->second.ev.size()) tp = &it->second;
        }
        if (tp) (void)hipEventRecord(tp->ev[2 * tp->n], s);
        int rc;
        if (op.kind == 6) {
            // head conv + epilogue in one launch: the epilogue's current arguments travel with it
            rc = (with_epilogue && e->has_post) ? launch_head(op.hd, e->post, s) : fail("engine_run: this step ends in a fused head + epilogue op: with_epilogue must be 1 and an epilogue set");
        } else rc = launch_op(op, s);
        if (tp) { (void)hipEventRecord(tp->ev[2 * tp->n + 1], s); tp->n++; }
        if (rc) return rc;
    }
    if (with_epilogue && e->has_post && !(e->ops.size() && e->ops.back().kind == 6)) {
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
#ifdef CCDM_EXPERIMENTS
    e->exp_id = g_exp_engines++;
#endif
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

extern "C" int ccdm_engine_add_stem(ccdm_engine* e, const ccdm_stem_args* a) {
    CCDM_REQUIRE(e && a, "engine_add_stem: null");
    CCDM_REQUIRE(ccdm_stem_conv_supported(a->Cs, a->Cout, a->H, a->W, CCDM_PREC_F16X3), "engine_add_stem: Cs=%d Cout=%d %dx%d is not built", a->Cs, a->Cout, a->H, a->W);
    Op op{};
    op.kind = 5;
    op.st = *a;
    e->ops.push_back(op);
    drop_graph(e);
    return (int)e->ops.size() - 1;
}

extern "C" int ccdm_engine_add_head_posterior(ccdm_engine* e, const ccdm_head_args* a) {
    CCDM_REQUIRE(e && a, "engine_add_head_posterior: null");
    CCDM_REQUIRE(ccdm_head_posterior_supported(a->C, a->K, a->H, a->W, CCDM_PREC_F16X3), "engine_add_head_posterior: C=%d K=%d %dx%d is not built", a->C, a->K, a->H, a->W);
    Op op{};
    op.kind = 6;
    op.hd = *a;
    e->ops.push_back(op);
    drop_graph(e);
    return (int)e->ops.size() - 1;
}

extern "C" int ccdm_engine_set_epilogue(ccdm_engine* e, const ccdm_post_args* a) {
    CCDM_REQUIRE(e && a, "engine_set_epilogue: null");
    e->post = *a;
    e->post.step_ptr = e->step;
    e->has_post = true;
    if (e->run_dev) {
        // a run block is installed: the kernels read the per-run fields from it, whatever order the two calls came in — keep it
        // attached and reseed it from the new epilogue (a caller's NULL `run` must not leave set_run updating a block nothing reads)
        e->post.run = e->run_dev;
        const ccdm_post_args& p = e->post;
        e->run = ccdm_post_run{p.noise, p.noise_step_stride, p.philox_seed, p.sample_offset, p.noise_row0, p.out_probs, p.out_onehot, p.posterior_out};
        e->run_dirty = true;
    }
    drop_graph(e);
    return 0;
}

extern "C" int ccdm_engine_num_ops(const ccdm_engine* e) { return e ? (int)e->ops.size() : -1; }
extern "C" int ccdm_engine_num_captures(const ccdm_engine* e) { return e ? e->captures : -1; }

extern "C" int ccdm_engine_set_run_block(ccdm_engine* e, void* dev_block) {
    CCDM_REQUIRE(e && e->has_post, "engine_set_run_block: no epilogue set");
    CCDM_REQUIRE(dev_block && (reinterpret_cast<uintptr_t>(dev_block) & 7) == 0, "engine_set_run_block: null or misaligned block");
    e->run_dev = static_cast<ccdm_post_run*>(dev_block);
    e->post.run = e->run_dev;
    const ccdm_post_args& p = e->post;
    e->run = ccdm_post_run{p.noise, p.noise_step_stride, p.philox_seed, p.sample_offset, p.noise_row0, p.out_probs, p.out_onehot, p.posterior_out};
    e->run_dirty = true;
    drop_graph(e);
    return 0;
}

extern "C" int ccdm_engine_set_run(ccdm_engine* e, const float* noise, int64_t noise_step_stride, int32_t noise_row0, uint64_t philox_seed,
                                   uint32_t sample_offset, float* out_probs, int64_t* out_onehot, float* posterior_out) {
    CCDM_REQUIRE(e && e->has_post, "engine_set_run: no epilogue set");
    if (e->run_dev) {
        // device-resident fields: the captured graph does not depend on them; the block is rewritten at the head of the next run
        const ccdm_post_run r{noise, noise_step_stride, philox_seed, sample_offset, noise_row0, out_probs, out_onehot, posterior_out};
        const ccdm_post_run& o = e->run;
        const bool same = o.noise == r.noise && o.noise_step_stride == r.noise_step_stride && o.philox_seed == r.philox_seed &&
                          o.sample_offset == r.sample_offset && o.noise_row0 == r.noise_row0 && o.out_probs == r.out_probs &&
                          o.out_onehot == r.out_onehot && o.posterior_out == r.posterior_out;
        if (!same) { e->run = r; e->run_dirty = true; }
        return 0;
    }
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
    e->next_row = first_row + n_steps;
    if (e->run_dev && e->run_dirty) {
        hipLaunchKernelGGL(k_run_set, dim3(1), dim3(1), 0, s, e->run_dev, e->run);
        CCDM_CHECK_LAUNCH("run_set");
        e->run_dirty = false;
    }
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
            int rc = 0;
            e->graph_steps = 1;
#ifdef CCDM_EXPERIMENTS
            if (exp_env("CCDM_GRAPH_STEPS") > 1) e->graph_steps = exp_env("CCDM_GRAPH_STEPS");      // probe: several denoise steps per captured graph
#endif
            for (int g = 0; g < e->graph_steps && !rc; ++g) rc = launch_step(e, with_epilogue, s, false);
            err = hipStreamEndCapture(s, &e->graph);
            if (rc) { if (e->graph) { (void)hipGraphDestroy(e->graph); e->graph = nullptr; } return rc; }
            if (err != hipSuccess) return fail("engine_run: EndCapture: %s", hipGetErrorString(err));
            err = hipGraphInstantiate(&e->exec, e->graph, nullptr, nullptr, 0);
            if (err != hipSuccess) return fail("engine_run: GraphInstantiate: %s", hipGetErrorString(err));
            e->graph_valid = true;
            e->graph_with_epilogue = with_epilogue;
            e->captures++;
        }
        int i = 0;
        for (; i + e->graph_steps <= n_steps; i += e->graph_steps) {
            hipError_t err = hipGraphLaunch(e->exec, s);
            if (err != hipSuccess) return fail("engine_run: GraphLaunch: %s", hipGetErrorString(err));
        }
        for (; i < n_steps; ++i) {                  // (a remainder shorter than the captured graph: eager)
            int rc = launch_step(e, with_epilogue, s, false);
            if (rc) return rc;
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

extern "C" int ccdm_engine_input_absmax(ccdm_engine* e, float* out, int row, void* stream) {
    CCDM_REQUIRE(e && out, "engine_input_absmax: null");
    hipStream_t s = (hipStream_t)stream;
    // the activations were produced with table row `row` (default: the last one the last run executed); the counter stands one past it
    const int r = row >= 0 ? row : (e->next_row > 0 ? e->next_row - 1 : 0);
    hipLaunchKernelGGL(k_step_set, dim3(1), dim3(1), 0, s, e->step, (int32_t)r);
    CCDM_CHECK_LAUNCH("step_set");
    int rc = 0;
    for (size_t i = 0; i < e->ops.size() && !rc; ++i) {
        const Op& op = e->ops[i];
        if (op.kind == 0) rc = launch_conv_input_absmax(op.conv, out + i, s);
        else if (op.kind == 6) {
            ccdm_conv_args c{};                  // what the head stages: SiLU(GroupNorm(x))
            c.in0 = op.hd.x; c.C0 = op.hd.C; c.stats0 = op.hd.stats; c.slices0 = op.hd.slices; c.gamma = op.hd.gamma; c.beta = op.hd.beta;
            c.eps = op.hd.eps; c.act = CCDM_ACT_SILU; c.N = op.hd.N; c.Hin = c.Hout = op.hd.H; c.Win = c.Wout = op.hd.W; c.ksize = 3; c.stride = 1;
            c.emb_off = -1; c.prec = CCDM_PREC_F16X3;
            rc = launch_conv_input_absmax(c, out + i, s);
        }
        else if (op.kind == 5 || op.kind == 1) {
            // (stem: the image channels of xin — its one-hot channels hold 0 / 1 or nothing; attention core: q, k, v)
            const size_t n4 = op.kind == 5 ? (size_t)op.st.N * op.st.H * op.st.W * op.st.Cs / 4 : (size_t)op.N * op.T * 3 * op.C / 4;
            const float* src = op.kind == 5 ? op.st.xin : op.qkv;
            const unsigned bx = (unsigned)(n4 / 256 < 1 ? 1 : (n4 / 256 > 1024 ? 1024 : n4 / 256));
            hipLaunchKernelGGL(k_absmax, dim3(bx), dim3(256), 0, s, src, n4, out + i);
            if (hipGetLastError() != hipSuccess) rc = fail("engine_input_absmax: attention operand probe failed to launch");
        }
    }
    hipLaunchKernelGGL(k_step_set, dim3(1), dim3(1), 0, s, e->step, (int32_t)e->next_row);
    CCDM_CHECK_LAUNCH("step_set");
    return rc;
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
    } else if (op.kind == 6) {
        snprintf(buf, buflen, "head gn silu conv3x3 %d->%d @%dx%d + posterior / draw", op.hd.C, op.hd.K, op.hd.H, op.hd.W);
    } else if (op.kind == 5) {
        snprintf(buf, buflen, "stem conv3x3 onehot(%d)+image(%d)->%d @%dx%d stats", op.st.K, op.st.Cs - op.st.K, op.st.Cout, op.st.H, op.st.W);
    } else {
        snprintf(buf, buflen, "resample %s C=%d in%dx%d%s%s%s%s", op.rs.mode == CCDM_RESAMPLE_AVGPOOL2 ? "avgpool2" : "nearest-up2", op.rs.C, op.rs.Hin,
                 op.rs.Win, op.rs.stats ? " gn" : "", op.rs.act ? " silu" : "", op.rs.out_act ? " ->act" : "", op.rs.out_raw ? " ->raw" : "");
    }
    return 0;
}
