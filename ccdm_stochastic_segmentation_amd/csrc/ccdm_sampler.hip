// Fused epilogue of one denoise step: softmax_K -> categorical posterior -> clamp -> normalise ->
// Exp(1)-race argmax (or the last-step outputs).  One thread per pixel, everything in registers.
// HBM-bound: reads K logits + 1 byte x_t (+ K noise floats), writes 1 byte + K floats.
//
// Arithmetic order (all fp32, IEEE division, no contraction surprises — fmaf is written where meant):
//   x0_k  = exp(l_k - max) / sum_j exp(l_j - max)                                   unet.py:706 (nn.Softmax(dim=1))
//   A_k   = a*[k == x_t] + (1-a)/K ; b = (1-c)/K ; S = sum_k A_k (k ascending)      diffusion_denoising.py:110-128,
//   r_d   = x0_d / (c*A_d + b*S) ; R = sum_d r_d ; P_k = A_k*(c*r_k + b*R)          O(K) closed form
//   P_k   = max(P_k, 1e-12)                                                         diffusion_denoising.py:204
//   Phat  = P_k / sum_K P   (cascade order: blocks of 16 sequential, tail first)    torch Categorical normalisation
//   idx   = argmax_k Phat_k / E_k, first maximum wins                               torch.multinomial(n=1)
#include "ccdm_common.h"
#include "ccdm_sampler_common.h"

#include <algorithm>

namespace ccdm {

template <int KP>
__global__ __launch_bounds__(256) void k_posterior(const ccdm_post_args a_in) {
    const ccdm_post_args a = post_resolve_run(a_in);
    const size_t npix = (size_t)a.N * a.HW;
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= npix) return;
    const int K = a.K;
    const int step = a.step_ptr ? *a.step_ptr : 0;
    float x0[KP];
    const float* hp = a.head + i * a.head_stride;
#pragma unroll
    for (int k = 0; k < KP; ++k) x0[k] = k < K ? hp[k] : -INFINITY;
    posterior_pixel<KP>(a, i, x0, step);
}

// Many classes (K > 4): a thread's K head values are K * 4 bytes apart from its neighbour's, and so are the one-hot channels it writes into
// the stem's input — as per-thread accesses that is K load and K store instructions per wave touching 64 different lines each (K = 20,
// 16 x 256x512: 289 us, a quarter of the HBM rate, half of it waiting on the memory front end).  Here the block's 256 pixels move as what
// they are, one contiguous run of 256 * K floats in and 256 * xin_stride floats out: 16-byte requests in lane order, exchanged through LDS
// (rows padded to an odd pitch: conflict-free per-thread reads).  The stem input's image channels (positions K.. of each pixel) are not
// touched: a 16-byte piece that lies wholly inside one-hot channels is one store, a piece that straddles image channels is written
// element by element.  Same arithmetic (posterior_pixel), same bits.
template <int KP>
__global__ __launch_bounds__(256) void k_posterior_staged(const ccdm_post_args a_in) {
    constexpr int PITCH = KP | 1;
    __shared__ float sx[256 * PITCH];
    __shared__ int sb[256];
    const ccdm_post_args a = post_resolve_run(a_in);
    const size_t npix = (size_t)a.N * a.HW;
    const size_t i0 = (size_t)blockIdx.x * 256;
    const int tid = threadIdx.x;
    const int nvalid = (int)std::min<size_t>(256, npix - i0);
    const int K = a.K;
    const int step = a.step_ptr ? *a.step_ptr : 0;
    // idx / d for idx < 256 * 64, d in [5, 64]: (idx * ceil(2^20 / d)) >> 20 (error < idx / 2^20 < 1 / d)
    const unsigned hs = (unsigned)a.head_stride;                       // K <= hs <= KP (launch_posterior)
    const unsigned MK = ((1u << 20) + hs - 1u) / hs;
    {
        const float* src = a.head + i0 * hs;
        const int total = nvalid * (int)hs;
        if ((total & 3) == 0 && (reinterpret_cast<uintptr_t>(src) & 15) == 0) {
            for (int q = tid; q < total / 4; q += 256) {
                const f32x4 v = reinterpret_cast<const f32x4*>(src)[q];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const unsigned idx = 4u * (unsigned)q + (unsigned)e, p = (idx * MK) >> 20, k = idx - p * hs;
                    sx[p * PITCH + k] = v[e];
                }
            }
        } else {
            for (int idx = tid; idx < total; idx += 256) {
                const unsigned p = ((unsigned)idx * MK) >> 20, k = (unsigned)idx - p * hs;
                sx[p * PITCH + k] = src[idx];
            }
        }
    }
    __syncthreads();
    int bi = 0;
    if (tid < nvalid) {
        float x0[KP];
#pragma unroll
        for (int k = 0; k < KP; ++k) x0[k] = k < K ? sx[tid * PITCH + k] : -INFINITY;
        posterior_pixel<KP>(a, i0 + tid, x0, step, &bi);
    }
    const int mode = (int)a.step_table[(size_t)step * 4 + 2];          // uniform
    if (mode != CCDM_STEP_SAMPLE || !a.xin) return;
    sb[tid] = bi;
    __syncthreads();
    {
        const unsigned stride = (unsigned)a.xin_stride;
        const unsigned MS = ((1u << 20) + stride - 1u) / stride;
        float* dst = a.xin + i0 * stride;
        const int total = nvalid * (int)stride;
        if ((total & 3) == 0 && (reinterpret_cast<uintptr_t>(dst) & 15) == 0) {
            for (int q = tid; q < total / 4; q += 256) {
                f32x4 v;
                bool all = true;
                bool oh[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const unsigned idx = 4u * (unsigned)q + (unsigned)e, p = (idx * MS) >> 20, c = idx - p * stride;
                    oh[e] = c < (unsigned)K;
                    all = all && oh[e];
                    v[e] = (int)c == sb[p] ? 1.0f : 0.0f;
                }
                if (all) reinterpret_cast<f32x4*>(dst)[q] = v;
                else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) if (oh[e]) dst[4 * q + e] = v[e];
                }
            }
        } else {
            for (int idx = tid; idx < total; idx += 256) {
                const unsigned p = ((unsigned)idx * MS) >> 20, c = (unsigned)idx - p * stride;
                if (c < (unsigned)K) dst[idx] = (int)c == sb[p] ? 1.0f : 0.0f;
            }
        }
    }
}

// More than 32 classes (round 5; no reference dataset has them — builder.py:36-45 accepts any label shape): a pixel's K values live in
// an LDS row of the block instead of registers and every pass over the classes is a run-time loop — the arithmetic, its order
// (k ascending; the normalising sum in 16-blocks, then the tail, then tail + blocks) and the Philox counters are posterior_pixel_core's,
// statement for statement (tested bit for bit against the register kernels at K <= 32, against the reference's own draws at K = 40).
// 128 pixels per block, row pitch K | 1 floats (conflict-free per-thread walks): K <= 255 (x_t is a uint8 class index).
__global__ __launch_bounds__(128) void k_posterior_many(const ccdm_post_args a_in) {
    extern __shared__ float smany[];
    const ccdm_post_args a = post_resolve_run(a_in);
    const int K = a.K, PITCH = K | 1;
    const size_t npix = (size_t)a.N * a.HW;
    const size_t i0 = (size_t)blockIdx.x * 128;
    const int tid = threadIdx.x;
    const int nvalid = (int)std::min<size_t>(128, npix - i0);
    const int step = a.step_ptr ? *a.step_ptr : 0;
    {   // the block's head values: one contiguous run when the head rows are dense, in lane order either way
        const unsigned hs = (unsigned)a.head_stride;
        const float* src = a.head + i0 * hs;
        const int total = nvalid * K;
        for (int idx = tid; idx < total; idx += 128) {
            const unsigned p = (unsigned)idx / (unsigned)K, k = (unsigned)idx - p * (unsigned)K;
            smany[p * PITCH + k] = src[(size_t)p * hs + k];
        }
    }
    __syncthreads();
    const float* trow = a.step_table + (size_t)step * 4;
    const float al = trow[0], cu = trow[1];
    const int mode = (int)trow[2];
    int bi = -1;
    if (tid < nvalid) {
        float* x = smany + tid * PITCH;
        const size_t i = i0 + tid;
        const int xt = mode == CCDM_STEP_SOFTMAX_ONLY ? 0 : (int)a.xt[i];
        if (a.range_flag) {
            float chk = 0.f;
            for (int k = 0; k < K; ++k) chk += fabsf(x[k]);
            if (!(chk <= 3.0e38f)) *a.range_flag = 1;
        }
        if (a.softmax & 1) {
            float mx = x[0];
            for (int k = 1; k < K; ++k) mx = fmaxf(mx, x[k]);
            float sum = 0.f;
            for (int k = 0; k < K; ++k) { x[k] = expf(x[k] - mx); sum += x[k]; }
            for (int k = 0; k < K; ++k) x[k] = x[k] / sum;
        }
        if (mode == CCDM_STEP_SOFTMAX_ONLY) {
            if (a.out_probs) for (int k = 0; k < K; ++k) a.out_probs[i * K + k] = x[k];
        } else {
            const float Kf = (float)K;
            const float u = (1.0f - al) / Kf, b = (1.0f - cu) / Kf;
            auto Ak = [&](const int k) { return (k == xt ? al : 0.0f) + u; };
            float S = Ak(0);
            for (int k = 1; k < K; ++k) S = S + Ak(k);
            const float bS = b * S;
            float R = 0.f;
            for (int k = 0; k < K; ++k) {
                const float rk = x[k] / (cu * Ak(k) + bS);
                x[k] = rk;
                R = k == 0 ? rk : R + rk;
            }
            const float bR = b * R;
            for (int k = 0; k < K; ++k) x[k] = fmaxf(Ak(k) * (cu * x[k] + bR), 1e-12f);
            float tot;
            {
                const int full = (K / 16) * 16;
                float hi = 0.f, tail = 0.f;
                for (int s0 = 0; s0 < full; s0 += 16) {
                    float blk = x[s0];
                    for (int k = 1; k < 16; ++k) blk = blk + x[s0 + k];
                    hi = s0 == 0 ? blk : hi + blk;
                }
                for (int k = full; k < K; ++k) tail = k == full ? x[k] : tail + x[k];
                tot = full < K ? (full > 0 ? tail + hi : tail) : hi;
            }
            for (int k = 0; k < K; ++k) x[k] = x[k] / tot;
            if (a.posterior_out) for (int k = 0; k < K; ++k) a.posterior_out[i * K + k] = x[k];
            if (mode == CCDM_STEP_SAMPLE) {
                float best = -INFINITY;
                bi = 0;
                if (a.noise) {
                    const float* e = a.noise + (size_t)(step - a.noise_row0) * a.noise_step_stride + i * K;
                    for (int k = 0; k < K; ++k) {
                        const float qv = x[k] / e[k];
                        if (qv > best) { best = qv; bi = k; }
                    }
                } else {
                    const uint32_t pix = (uint32_t)(i % a.HW), smp = (uint32_t)(i / a.HW) + a.sample_offset;
                    const uint32_t k0 = (uint32_t)a.philox_seed, k1 = (uint32_t)(a.philox_seed >> 32);
                    for (int kq = 0; kq * 4 < K; ++kq) {
                        uint32_t w[4];
                        Philox::run(pix, smp, (uint32_t)step, (uint32_t)kq, k0, k1, w);
                        for (int j = 0; j < 4; ++j) {
                            const int k = kq * 4 + j;
                            if (k < K) {
                                const float qv = x[k] / u32_to_exp1(w[j]);
                                if (qv > best) { best = qv; bi = k; }
                            }
                        }
                    }
                }
                a.xt_next[i] = (uint8_t)bi;
                if (a.xin) {
                    float* d = a.xin + i * a.xin_stride;
                    for (int k = 0; k < K; ++k) d[k] = (k == bi) ? 1.0f : 0.0f;
                }
            } else if (mode == CCDM_STEP_LAST_CONFIDENCE) {
                if (a.out_probs) for (int k = 0; k < K; ++k) a.out_probs[i * K + k] = x[k];
            } else if (mode == CCDM_STEP_LAST_MAJORITY) {
                float best = x[0];
                bi = 0;
                for (int k = 1; k < K; ++k) if (x[k] > best) { best = x[k]; bi = k; }
                if (a.out_onehot) for (int k = 0; k < K; ++k) a.out_onehot[i * K + k] = (k == bi) ? 1 : 0;
                a.xt_next[i] = (uint8_t)bi;
            }
        }
    }
}

// the staged form needs contiguous head rows and strides its index arithmetic covers
static int posterior_kp(int K) { return K <= 2 ? 2 : K <= 4 ? 4 : K <= 8 ? 8 : K <= 16 ? 16 : K <= 20 ? 20 : K <= 24 ? 24 : 32; }
static bool posterior_staged_ok(const ccdm_post_args& a) {
    return a.K > 4 && a.head_stride <= posterior_kp(a.K) && (!a.xin || (a.xin_stride >= a.K && a.xin_stride <= 64));
}

int launch_posterior(const ccdm_post_args& a, hipStream_t s) {
    CCDM_REQUIRE(a.head && a.xt && a.step_table && a.xt_next, "posterior: null pointer");
    CCDM_REQUIRE(a.K >= 2 && a.K <= CCDM_MAX_CLASSES, "posterior: K=%d outside [2,%d]", a.K, CCDM_MAX_CLASSES);
    CCDM_REQUIRE(a.head_stride >= a.K, "posterior: head_stride %d < K %d", a.head_stride, a.K);
    const size_t npix = (size_t)a.N * a.HW;
    if (!npix) return 0;
    if (a.K > 32 || (a.softmax & CCDM_POST_DIAG_MANY)) {       // LDS-resident classes, run-time loops (any K; the diagnostic bit: parity tests at K <= 32)
        const size_t lds = (size_t)128 * (a.K | 1) * sizeof(float);
        static size_t reserved = 64 * 1024;               // (raised once per size class, not on every enqueue)
        if (lds > reserved) {
            if (hipFuncSetAttribute(reinterpret_cast<const void*>(k_posterior_many), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)
                return fail("posterior: cannot reserve %zu bytes of LDS for K=%d", lds, a.K);
            reserved = 160 * 1024;
        }
        hipLaunchKernelGGL(k_posterior_many, dim3((unsigned)((npix + 127) / 128)), dim3(128), lds, s, a);
        CCDM_CHECK_LAUNCH("posterior(many classes)");
        return 0;
    }
    dim3 grid((unsigned)((npix + 255) / 256)), block(256);
    if (posterior_staged_ok(a)) {
        if (a.K <= 8) hipLaunchKernelGGL(k_posterior_staged<8>, grid, block, 0, s, a);
        else if (a.K <= 16) hipLaunchKernelGGL(k_posterior_staged<16>, grid, block, 0, s, a);
        else if (a.K <= 20) hipLaunchKernelGGL(k_posterior_staged<20>, grid, block, 0, s, a);
        else if (a.K <= 24) hipLaunchKernelGGL(k_posterior_staged<24>, grid, block, 0, s, a);
        else hipLaunchKernelGGL(k_posterior_staged<32>, grid, block, 0, s, a);
        CCDM_CHECK_LAUNCH("posterior");
        return 0;
    }
    if (a.K <= 2) hipLaunchKernelGGL(k_posterior<2>, grid, block, 0, s, a);
    else if (a.K <= 4) hipLaunchKernelGGL(k_posterior<4>, grid, block, 0, s, a);
    else if (a.K <= 8) hipLaunchKernelGGL(k_posterior<8>, grid, block, 0, s, a);
    else if (a.K <= 16) hipLaunchKernelGGL(k_posterior<16>, grid, block, 0, s, a);
    // (the per-class work — four IEEE divisions, an exponential, a quarter Philox block — is predicated, not skipped, beyond K: Cityscapes'
    //  K = 20 on the 32-wide instantiation did 60 % more arithmetic than it needed)
    else if (a.K <= 20) hipLaunchKernelGGL(k_posterior<20>, grid, block, 0, s, a);
    else if (a.K <= 24) hipLaunchKernelGGL(k_posterior<24>, grid, block, 0, s, a);
    else hipLaunchKernelGGL(k_posterior<32>, grid, block, 0, s, a);
    CCDM_CHECK_LAUNCH("posterior");
    return 0;
}

}  // namespace ccdm

// ---------------------------------------------------------------------------------------------------
// Training-time forward pieces — elementwise, BCHW fp32, one thread per pixel, K in registers.
// Each is one pass over its operands: the bound is HBM (read 2K + write K floats per pixel for theta_post).
// ---------------------------------------------------------------------------------------------------
namespace ccdm {

__global__ __launch_bounds__(256) void k_mix_uniform(const float* __restrict__ x, const float* __restrict__ s, int K, int HW,
                                                     float* __restrict__ out) {
    // grid.y = sample; a thread walks float4 quads of the sample's contiguous K*HW block
    const int n = blockIdx.y;
    const float sn = s[n], u = (1.0f - sn) / (float)K;
    const size_t per = (size_t)K * HW;
    const float* xs = x + n * per;
    float* os = out + n * per;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < per; i += (size_t)gridDim.x * blockDim.x) os[i] = sn * xs[i] + u;
}

template <int KP>
__global__ __launch_bounds__(256) void k_theta_post(const float* __restrict__ xt, const float* __restrict__ x0, const float* __restrict__ a,
                                                    const float* __restrict__ c, int K, int HW, int prob_mode, float* __restrict__ out) {
    const int n = blockIdx.y;
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= HW) return;
    const float an = a[n], cn = c[n];
    const float u = (1.0f - an) / (float)K, b = (1.0f - cn) / (float)K;
    const size_t base = (size_t)n * K * HW + p;
    float A[KP], X[KP];
#pragma unroll
    for (int k = 0; k < KP; ++k)
        if (k < K) { A[k] = an * xt[base + (size_t)k * HW] + u; X[k] = x0[base + (size_t)k * HW]; }
    if (!prob_mode) {
        // theta_k = A_k * (c*x0_k + b), normalised (sum in ascending k)
        float sum = 0.f;
#pragma unroll
        for (int k = 0; k < KP; ++k)
            if (k < K) { X[k] = A[k] * (cn * X[k] + b); sum += X[k]; }
#pragma unroll
        for (int k = 0; k < KP; ++k)
            if (k < K) out[base + (size_t)k * HW] = X[k] / sum;
    } else {
        // S = sum_k A_k ; r_d = theta_d / (c*A_d + b*S) ; R = sum_d r_d ; out_k = A_k * (c*r_k + b*R)
        float S = 0.f;
#pragma unroll
        for (int k = 0; k < KP; ++k)
            if (k < K) S += A[k];
        float R = 0.f;
#pragma unroll
        for (int k = 0; k < KP; ++k)
            if (k < K) { X[k] = X[k] / (cn * A[k] + b * S); R += X[k]; }
#pragma unroll
        for (int k = 0; k < KP; ++k)
            if (k < K) out[base + (size_t)k * HW] = A[k] * (cn * X[k] + b * R);
    }
}

__global__ __launch_bounds__(256) void k_kl_clamped(const float* __restrict__ p, const float* __restrict__ q, size_t n, float floor,
                                                    float* __restrict__ out) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const float pv = p[i];
        // torch's kl_div(input = log q, target = p): xlogy(p, p) - p * input, i.e. 0 contribution where p == 0
        out[i] = pv > 0.f ? pv * (logf(pv) - logf(fmaxf(q[i], floor))) : (pv == 0.f ? 0.f : NAN);
    }
}

}  // namespace ccdm

extern "C" int ccdm_mix_uniform(const float* x, const float* s, int N, int K, int HW, float* out, void* stream) {
    if (!x || !s || !out) return ccdm::fail("ccdm_mix_uniform: null pointer");
    if (N <= 0 || K < 2 || HW <= 0) return ccdm::fail("ccdm_mix_uniform: bad shape N=%d K=%d HW=%d", N, K, HW);
    const size_t per = (size_t)K * HW;
    dim3 grid((unsigned)std::min<size_t>((per + 255) / 256, 4096), (unsigned)N);
    hipLaunchKernelGGL(ccdm::k_mix_uniform, grid, dim3(256), 0, (hipStream_t)stream, x, s, K, HW, out);
    CCDM_CHECK_LAUNCH("mix_uniform");
    return 0;
}

extern "C" int ccdm_theta_post(const float* xt, const float* x0, const float* a, const float* c, int N, int K, int HW, int prob_mode,
                               float* out, void* stream) {
    if (!xt || !x0 || !a || !c || !out) return ccdm::fail("ccdm_theta_post: null pointer");
    if (N <= 0 || HW <= 0) return ccdm::fail("ccdm_theta_post: bad shape N=%d HW=%d", N, HW);
    if (K < 2 || K > 32) return ccdm::fail("ccdm_theta_post: K=%d outside [2,32]", K);
    dim3 grid((unsigned)((HW + 255) / 256), (unsigned)N), block(256);
    hipStream_t s = (hipStream_t)stream;
    if (K <= 2) hipLaunchKernelGGL(ccdm::k_theta_post<2>, grid, block, 0, s, xt, x0, a, c, K, HW, prob_mode, out);
    else if (K <= 4) hipLaunchKernelGGL(ccdm::k_theta_post<4>, grid, block, 0, s, xt, x0, a, c, K, HW, prob_mode, out);
    else if (K <= 8) hipLaunchKernelGGL(ccdm::k_theta_post<8>, grid, block, 0, s, xt, x0, a, c, K, HW, prob_mode, out);
    else if (K <= 16) hipLaunchKernelGGL(ccdm::k_theta_post<16>, grid, block, 0, s, xt, x0, a, c, K, HW, prob_mode, out);
    else hipLaunchKernelGGL(ccdm::k_theta_post<32>, grid, block, 0, s, xt, x0, a, c, K, HW, prob_mode, out);
    CCDM_CHECK_LAUNCH("theta_post");
    return 0;
}

extern "C" int ccdm_kl_clamped(const float* p, const float* q, size_t n, float floor, float* out, void* stream) {
    if (!p || !q || !out) return ccdm::fail("ccdm_kl_clamped: null pointer");
    if (!n) return 0;
    dim3 grid((unsigned)std::min<size_t>((n + 255) / 256, 8192));
    hipLaunchKernelGGL(ccdm::k_kl_clamped, grid, dim3(256), 0, (hipStream_t)stream, p, q, n, floor, out);
    CCDM_CHECK_LAUNCH("kl_clamped");
    return 0;
}

extern "C" int ccdm_posterior_sample(const ccdm_post_args* a, void* stream) {
    if (!a) return ccdm::fail("ccdm_posterior_sample: null args");
    return ccdm::launch_posterior(*a, (hipStream_t)stream);
}
