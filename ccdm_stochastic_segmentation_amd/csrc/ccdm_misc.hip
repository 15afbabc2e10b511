// Small kernels of the sampler: standalone GroupNorm statistics, streaming self-attention core,
// time-conditioning tables, boundary re-layout.  gfx950 only.
#include "ccdm_common.h"

#include <stdarg.h>

namespace ccdm {

static thread_local std::string g_err;
void set_error(const std::string& s) { g_err = s; }
int fail(const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
    return -1;
}

// ---------------------------------------------------------------------------------------------------
// Per-channel (sum, sum^2) partials of an NHWC tensor.  grid (slices, N); block = Q*R threads with
// Q = C/4 float4 columns, R pixel rows in flight: a thread's channel quad is fixed, so it accumulates in
// registers; the R partials are then added in ascending r.  Deterministic.
// ---------------------------------------------------------------------------------------------------
__global__ void k_gn_stats(const float* __restrict__ x, int HW, int C, int slices, double* __restrict__ stats) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    double* red = reinterpret_cast<double*>(smem);          // [R][Q][8]
    const int Q = C >> 2, R = blockDim.x / Q;
    const int q = threadIdx.x % Q, r = threadIdx.x / Q;
    const int s = blockIdx.x, n = blockIdx.y;
    const int p0 = (int)((long long)HW * s / slices), p1 = (int)((long long)HW * (s + 1) / slices);
    double a[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    const float* base = x + (size_t)n * HW * C + 4 * q;
    for (int p = p0 + r; p < p1; p += R) {
        const float4 v = *reinterpret_cast<const float4*>(base + (size_t)p * C);
        a[0] += v.x; a[1] += (double)v.x * v.x;
        a[2] += v.y; a[3] += (double)v.y * v.y;
        a[4] += v.z; a[5] += (double)v.z * v.z;
        a[6] += v.w; a[7] += (double)v.w * v.w;
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) red[(r * Q + q) * 8 + i] = a[i];
    __syncthreads();
    if (r == 0) {
        double t[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) t[i] = 0.0;
        for (int rr = 0; rr < R; ++rr)
#pragma unroll
            for (int i = 0; i < 8; ++i) t[i] += red[(rr * Q + q) * 8 + i];
        double* o = stats + (((size_t)n * slices + s) * C + 4 * q) * 2;
#pragma unroll
        for (int i = 0; i < 8; ++i) o[i] = t[i];
    }
}

// ---------------------------------------------------------------------------------------------------
// Self-attention core (A11).  One wave per 64 queries of one (sample, head); keys/values stream through
// LDS in tiles of 64; online softmax in registers, rescaled once per 16 keys.  d = 32 channels per head
// is a compile-time constant of every reference config (num_head_channels: 32); other head widths take
// the generic template.  The [T,T] score matrix never exists.
// ---------------------------------------------------------------------------------------------------
template <int D>
__global__ __launch_bounds__(64) void k_attention(const float* __restrict__ qkv, float* __restrict__ out,
                                                  int T, int Ta, int C, int heads, int order) {      // Ta: token rows allocated per sample (>= T)
    __shared__ __attribute__((aligned(16))) float kt[64 * D];
    __shared__ __attribute__((aligned(16))) float vt[64 * D];
    const int lane = threadIdx.x;
    const int h = blockIdx.y, n = blockIdx.z;
    const int t = blockIdx.x * 64 + lane;
    const int C3 = 3 * C;
    int qoff, koff, voff;
    if (order == 0) { qoff = h * 3 * D; koff = qoff + D; voff = qoff + 2 * D; }      // legacy: head-major
    else { qoff = h * D; koff = C + h * D; voff = 2 * C + h * D; }                  // new: qkv-major
    const float scale = (float)(1.0 / sqrt(sqrt((double)D)));                      // ch^-1/4 on q and on k (python double -> fp32)
    const float* base = qkv + (size_t)n * Ta * C3;

    float q[D], o[D];
    const int tq = t < T ? t : T - 1;
#pragma unroll
    for (int c = 0; c < D; c += 4) {
        const float4 v = *reinterpret_cast<const float4*>(base + (size_t)tq * C3 + qoff + c);
        q[c] = v.x * scale; q[c + 1] = v.y * scale; q[c + 2] = v.z * scale; q[c + 3] = v.w * scale;
        o[c] = o[c + 1] = o[c + 2] = o[c + 3] = 0.f;
    }
    float m = -INFINITY, l = 0.f;

    for (int j0 = 0; j0 < T; j0 += 64) {
        __syncthreads();
        for (int item = lane; item < 64 * (D / 4); item += 64) {
            const int row = item / (D / 4), c4 = item % (D / 4);
            float4 kv = make_float4(0, 0, 0, 0), vv = make_float4(0, 0, 0, 0);
            if (j0 + row < T) {
                const float* p = base + (size_t)(j0 + row) * C3;
                kv = *reinterpret_cast<const float4*>(p + koff + 4 * c4);
                vv = *reinterpret_cast<const float4*>(p + voff + 4 * c4);
            }
            kv.x *= scale; kv.y *= scale; kv.z *= scale; kv.w *= scale;
            *reinterpret_cast<float4*>(kt + row * D + 4 * c4) = kv;
            *reinterpret_cast<float4*>(vt + row * D + 4 * c4) = vv;
        }
        __syncthreads();
        const int jn = (T - j0) < 64 ? (T - j0) : 64;
        for (int jb = 0; jb < jn; jb += 16) {
            float s[16];
            float mx = m;
#pragma unroll
            for (int jj = 0; jj < 16; ++jj) {
                const float* kr = kt + (jb + jj) * D;
                float acc = 0.f;
#pragma unroll
                for (int c = 0; c < D; ++c) acc = fmaf(q[c], kr[c], acc);
                s[jj] = (jb + jj < jn) ? acc : -INFINITY;
                mx = fmaxf(mx, s[jj]);
            }
            const float corr = expf(m - mx);     // m = -inf on the first tile -> 0
            l *= corr;
#pragma unroll
            for (int c = 0; c < D; ++c) o[c] *= corr;
            m = mx;
#pragma unroll
            for (int jj = 0; jj < 16; ++jj) {
                const float p = expf(s[jj] - mx);   // masked keys: exp(-inf) = 0
                l += p;
                const float* vr = vt + (jb + jj) * D;
#pragma unroll
                for (int c = 0; c < D; ++c) o[c] = fmaf(p, vr[c], o[c]);
            }
        }
    }
    if (t < T) {
        const float inv = 1.0f / l;
        float* dst = out + ((size_t)n * Ta + t) * C + h * D;
#pragma unroll
        for (int c = 0; c < D; c += 4)
            *reinterpret_cast<float4*>(dst + c) = make_float4(o[c] * inv, o[c + 1] * inv, o[c + 2] * inv, o[c + 3] * inv);
    }
}

int launch_attention_mfma(const float* qkv, float* out, int N, int T, int Ta, int C, int heads, int order, hipStream_t s);
int attention_mfma_width(int D);

int launch_attention(const float* qkv, float* out, int N, int T, int Ta, int C, int heads, int order, hipStream_t s) {
    CCDM_REQUIRE(qkv && out, "attention: null pointer");
    CCDM_REQUIRE(heads > 0 && C % heads == 0, "attention: C=%d not divisible by heads=%d", C, heads);
    CCDM_REQUIRE(T > 0 && Ta >= T, "attention: T=%d, %d rows allocated per sample", T, Ta);
    const int D = C / heads;
    const bool force_valu = (order & CCDM_ATTENTION_FORCE_VALU) != 0;       // the exact-fp32 vector-pipe kernel (range fallback, tests)
    order &= 255;
    // Matrix-core kernel: every head width that is a multiple of 4 up to 128 (padded to 32 / 64 / 96 / 128 inside).  Head width 32 with
    // a token count that is not a multiple of 32 keeps the VALU kernel it has always run on (round-1 behaviour, bit for bit).
    const bool mfma_ok = attention_mfma_width(D) > 0 && !(D == 32 && (T % 32 != 0 || Ta != T));
    if (mfma_ok && !force_valu) return launch_attention_mfma(qkv, out, N, T, Ta, C, heads, order, s);
    dim3 grid(cdiv(T, 64), heads, N), block(64);
    switch (D) {      // VALU kernel: q and o of one query live in a lane's registers, so the widths are instantiated one by one
        case 4: hipLaunchKernelGGL(k_attention<4>, grid, block, 0, s, qkv, out, T, Ta, C, heads, order); break;
        case 8: hipLaunchKernelGGL(k_attention<8>, grid, block, 0, s, qkv, out, T, Ta, C, heads, order); break;
        case 12: hipLaunchKernelGGL(k_attention<12>, grid, block, 0, s, qkv, out, T, Ta, C, heads, order); break;
        case 16: hipLaunchKernelGGL(k_attention<16>, grid, block, 0, s, qkv, out, T, Ta, C, heads, order); break;
        case 24: hipLaunchKernelGGL(k_attention<24>, grid, block, 0, s, qkv, out, T, Ta, C, heads, order); break;
        case 32: hipLaunchKernelGGL(k_attention<32>, grid, block, 0, s, qkv, out, T, Ta, C, heads, order); break;
        case 48: hipLaunchKernelGGL(k_attention<48>, grid, block, 0, s, qkv, out, T, Ta, C, heads, order); break;
        case 64: hipLaunchKernelGGL(k_attention<64>, grid, block, 0, s, qkv, out, T, Ta, C, heads, order); break;
        default: return fail("attention: head width %d (C=%d / heads=%d) is built neither on the matrix cores (multiples of 4 up to 128) nor "
                             "on the vector path (4, 8, 12, 16, 24, 32, 48, 64): choose num_heads / num_head_channels accordingly", D, C, heads);
    }
    CCDM_CHECK_LAUNCH("attention");
    return 0;
}

// ---------------------------------------------------------------------------------------------------
// Time conditioning tables.  One block per step row.  fp32, sequential-k dot products.
// ---------------------------------------------------------------------------------------------------
__device__ __forceinline__ float silu_t(float x) { return x / (1.0f + expf(-x)); }

__global__ void k_time_table(const float* __restrict__ sinus, int mc,
                             const float* __restrict__ w0, const float* __restrict__ b0,
                             const float* __restrict__ w2, const float* __restrict__ b2,
                             const float* __restrict__ wcat, const float* __restrict__ bcat, int E,
                             float* __restrict__ emb_out, float* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int ted = 4 * mc;
    float* x = sm;             // [mc]
    float* h1 = sm + mc;       // [ted] SiLU(Linear0)
    float* e = h1 + ted;       // [ted] SiLU(emb)
    const int s = blockIdx.x;
    for (int i = threadIdx.x; i < mc; i += blockDim.x) x[i] = sinus[(size_t)s * mc + i];
    __syncthreads();
    for (int j = threadIdx.x; j < ted; j += blockDim.x) {
        float acc = 0.f;
        for (int i = 0; i < mc; ++i) acc = fmaf(w0[(size_t)j * mc + i], x[i], acc);
        h1[j] = silu_t(acc + b0[j]);
    }
    __syncthreads();
    for (int j = threadIdx.x; j < ted; j += blockDim.x) {
        float acc = 0.f;
        for (int i = 0; i < ted; ++i) acc = fmaf(w2[(size_t)j * ted + i], h1[i], acc);
        const float v = acc + b2[j];
        if (emb_out) emb_out[(size_t)s * ted + j] = v;
        e[j] = silu_t(v);
    }
    __syncthreads();
    for (int j = threadIdx.x; j < E; j += blockDim.x) {
        float acc = 0.f;
        for (int i = 0; i < ted; ++i) acc = fmaf(wcat[(size_t)j * ted + i], e[i], acc);
        out[(size_t)s * E + j] = acc + bcat[j];
    }
}

// ---------------------------------------------------------------------------------------------------
// Boundary re-layout
// ---------------------------------------------------------------------------------------------------
__global__ void k_nchw_to_nhwc(const float* __restrict__ src, float* __restrict__ dst, int C, int HW,
                               int dst_stride, int dst_off, size_t total) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int p = (int)(i % HW);
    const size_t nc = i / HW;
    const int c = (int)(nc % C);
    const size_t n = nc / C;
    dst[(n * HW + p) * dst_stride + dst_off + c] = src[i];
}

__global__ void k_onehot_to_xin(const uint8_t* __restrict__ idx, float* __restrict__ xin, int K, int stride, size_t npix) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= npix) return;
    const int k = idx[i];
    float* d = xin + i * stride;
    for (int c = 0; c < K; ++c) d[c] = (c == k) ? 1.0f : 0.0f;
}

}  // namespace ccdm

using namespace ccdm;

extern "C" int ccdm_version(void) { return CCDM_ABI_VERSION; }
extern "C" const char* ccdm_last_error_string(void) { return g_err.c_str(); }

// Fold S_in statistics slices into S_out <= CCDM_STATS_MAX_SLICES: out slice j = in slices [j*S_in/S_out, (j+1)*S_in/S_out)
// added in ascending order (fixed order: deterministic, independent of the batch size).
namespace ccdm {
__global__ void k_stats_fold(const double* __restrict__ in, int S_in, int C2, int S_out, double* __restrict__ out) {
    const int n = blockIdx.y;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;        // (out slice, channel*2 + {sum, sumsq})
    if (i >= S_out * C2) return;
    const int j = i / C2, c = i % C2;
    const int s0 = (int)((long long)j * S_in / S_out), s1 = (int)((long long)(j + 1) * S_in / S_out);
    const double* p = in + (size_t)n * S_in * C2 + c;
    // 16 independent loads per round (clamped; added in ascending order, masked): a 384 -> 16 fold is 2 memory round trips instead of
    // 24 dependent ones (17 -> ~4 us per launch; a 512x1024 step has 25 of them)
    double t = 0.0;
    for (int s = s0; s < s1; s += 16) {
        double v[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) v[u] = p[(size_t)min(s + u, s1 - 1) * C2];
#pragma unroll
        for (int u = 0; u < 16; ++u) t += s + u < s1 ? v[u] : 0.0;
    }
    out[(size_t)n * S_out * C2 + i] = t;
}
int launch_stats_fold(const double* in, int N, int S_in, int C, int S_out, double* out, hipStream_t s) {
    CCDM_REQUIRE(in && out && N > 0 && C > 0, "stats_fold: bad arguments");
    CCDM_REQUIRE(S_out >= 1 && S_out <= CCDM_STATS_MAX_SLICES && S_in >= S_out, "stats_fold: %d -> %d slices", S_in, S_out);
    dim3 grid((unsigned)cdiv(S_out * C * 2, 256), (unsigned)N);
    hipLaunchKernelGGL(k_stats_fold, grid, dim3(256), 0, s, in, S_in, 2 * C, S_out, out);
    CCDM_CHECK_LAUNCH("stats_fold");
    return 0;
}
}  // namespace ccdm

extern "C" int ccdm_stats_fold(const double* in, int N, int S_in, int C, int S_out, double* out, void* stream) {
    return ccdm::launch_stats_fold(in, N, S_in, C, S_out, out, (hipStream_t)stream);
}

extern "C" int ccdm_gn_stats(const float* x, int N, int HW, int C, int slices, double* stats, void* stream) {
    CCDM_REQUIRE(x && stats, "gn_stats: null pointer");
    CCDM_REQUIRE(C % 4 == 0 && C >= 4 && C <= 1024, "gn_stats: C=%d must be a multiple of 4 in [4,1024]", C);
    CCDM_REQUIRE(slices >= 1 && slices <= CCDM_STATS_MAX_SLICES, "gn_stats: slices=%d", slices);
    const int Q = C / 4;
    const int R = 256 / Q > 0 ? 256 / Q : 1;
    dim3 grid(slices, N), block(Q * R);
    const size_t lds = (size_t)Q * R * 8 * sizeof(double);
    hipLaunchKernelGGL(k_gn_stats, grid, block, lds, (hipStream_t)stream, x, HW, C, slices, stats);
    CCDM_CHECK_LAUNCH("gn_stats");
    return 0;
}

extern "C" int ccdm_attention(const float* qkv, float* out, int N, int T, int C, int heads, int order, void* stream) {
    return launch_attention(qkv, out, N, T, T, C, heads, order, (hipStream_t)stream);
}

extern "C" int ccdm_time_table(const float* sinus, int S, int mc, const float* w0, const float* b0,
                               const float* w2, const float* b2, const float* wcat, const float* bcat, int E,
                               float* emb_out, float* out, void* stream) {
    CCDM_REQUIRE(sinus && w0 && b0 && w2 && b2 && out, "time_table: null pointer");
    CCDM_REQUIRE(E == 0 || (wcat && bcat), "time_table: null wcat/bcat");
    CCDM_REQUIRE(mc > 0 && mc <= 1024, "time_table: model_channels=%d", mc);
    if (S <= 0) return 0;
    const size_t lds = (size_t)(mc + 8 * mc) * sizeof(float);
    hipLaunchKernelGGL(k_time_table, dim3(S), dim3(256), lds, (hipStream_t)stream, sinus, mc, w0, b0, w2, b2,
                       wcat, bcat, E, emb_out, out);
    CCDM_CHECK_LAUNCH("time_table");
    return 0;
}

extern "C" int ccdm_nchw_to_nhwc(const float* src, float* dst, int N, int C, int HW, int dst_stride, int dst_off, void* stream) {
    CCDM_REQUIRE(src && dst, "nchw_to_nhwc: null pointer");
    const size_t total = (size_t)N * C * HW;
    if (!total) return 0;
    hipLaunchKernelGGL(k_nchw_to_nhwc, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       src, dst, C, HW, dst_stride, dst_off, total);
    CCDM_CHECK_LAUNCH("nchw_to_nhwc");
    return 0;
}

extern "C" int ccdm_onehot_to_xin(const uint8_t* idx, float* xin, int N, int HW, int K, int xin_stride, void* stream) {
    CCDM_REQUIRE(idx && xin, "onehot_to_xin: null pointer");
    const size_t npix = (size_t)N * HW;
    if (!npix) return 0;
    hipLaunchKernelGGL(k_onehot_to_xin, dim3((unsigned)((npix + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       idx, xin, K, xin_stride, npix);
    CCDM_CHECK_LAUNCH("onehot_to_xin");
    return 0;
}


// ---------------------------------------------------------------------------------------------------
// DINO ViT-S/8 key-feature extractor: the pieces the conv / attention kernels do not cover.
// The network itself (facebookresearch/dino vision_transformer.py, loaded by the reference through torch.hub,
// ddpm/models/dino.py:58-82) is third-party code that is absent from /root/reference: restated from its published form.
// ---------------------------------------------------------------------------------------------------
namespace ccdm {

// nn.LayerNorm(C, eps) over the last axis: one wave per token row, the row in registers, two passes (mean, then centred
// variance) like ATen's; C <= 64 * LN_MAX_PER_LANE
constexpr int LN_MAX_PER_LANE = 24;
__global__ __launch_bounds__(256) void k_layernorm(const float* __restrict__ x, const float* __restrict__ gamma, const float* __restrict__ beta,
                                                   float eps, long rows, int C, float* __restrict__ out) {
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (row >= rows) return;
    const float* xr = x + row * C;
    float v[LN_MAX_PER_LANE];
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < LN_MAX_PER_LANE; ++i) {
        const int c = lane + 64 * i;
        v[i] = c < C ? xr[c] : 0.f;
        sum += v[i];
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) sum += __shfl_xor(sum, off);
    const float mean = sum / (float)C;
    float sq = 0.f;
#pragma unroll
    for (int i = 0; i < LN_MAX_PER_LANE; ++i) {
        const int c = lane + 64 * i;
        const float d = c < C ? v[i] - mean : 0.f;
        sq = fmaf(d, d, sq);
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) sq += __shfl_xor(sq, off);
    const float rstd = 1.0f / sqrtf(sq / (float)C + eps);
    float* orow = out + row * C;
#pragma unroll
    for (int i = 0; i < LN_MAX_PER_LANE; ++i) {
        const int c = lane + 64 * i;
        if (c < C) orow[c] = (v[i] - mean) * rstd * gamma[c] + beta[c];
    }
}

// nn.GELU() (exact, erf form)
__global__ __launch_bounds__(256) void k_gelu(const float* __restrict__ x, size_t n, float* __restrict__ out) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const float v = x[i];
        out[i] = 0.5f * v * (1.0f + erff(v * 0.70710678118654752440f));
    }
}

}  // namespace ccdm

extern "C" int ccdm_attention_ex(const float* qkv, float* out, int N, int T, int T_alloc, int C, int heads, int order, void* stream) {
    return ccdm::launch_attention(qkv, out, N, T, T_alloc, C, heads, order, (hipStream_t)stream);
}

extern "C" int ccdm_layernorm(const float* x, const float* gamma, const float* beta, float eps, long rows, int C, float* out, void* stream) {
    if (!x || !gamma || !beta || !out) return ccdm::fail("ccdm_layernorm: null pointer");
    if (rows <= 0 || C <= 0 || C > 64 * ccdm::LN_MAX_PER_LANE) return ccdm::fail("ccdm_layernorm: rows=%ld C=%d (C <= %d)", rows, C, 64 * ccdm::LN_MAX_PER_LANE);
    hipLaunchKernelGGL(ccdm::k_layernorm, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, (hipStream_t)stream, x, gamma, beta, eps, rows, C, out);
    CCDM_CHECK_LAUNCH("layernorm");
    return 0;
}

extern "C" int ccdm_gelu(const float* x, size_t n, float* out, void* stream) {
    if (!x || !out) return ccdm::fail("ccdm_gelu: null pointer");
    if (!n) return 0;
    const size_t blocks = (n + 255) / 256;
    hipLaunchKernelGGL(ccdm::k_gelu, dim3((unsigned)(blocks < 16384 ? blocks : 16384)), dim3(256), 0, (hipStream_t)stream, x, n, out);
    CCDM_CHECK_LAUNCH("gelu");
    return 0;
}
