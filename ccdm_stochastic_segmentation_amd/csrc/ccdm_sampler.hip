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

int launch_posterior(const ccdm_post_args& a, hipStream_t s) {
    CCDM_REQUIRE(a.head && a.xt && a.step_table && a.xt_next, "posterior: null pointer");
    CCDM_REQUIRE(a.K >= 2 && a.K <= 32, "posterior: K=%d outside [2,32]", a.K);
    CCDM_REQUIRE(a.head_stride >= a.K, "posterior: head_stride %d < K %d", a.head_stride, a.K);
    const size_t npix = (size_t)a.N * a.HW;
    if (!npix) return 0;
    dim3 grid((unsigned)((npix + 255) / 256)), block(256);
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
