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

#include <algorithm>

namespace ccdm {

struct Philox {
    static constexpr uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u, W0 = 0x9E3779B9u, W1 = 0xBB67AE85u;
    __device__ static inline void run(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1, uint32_t out[4]) {
#pragma unroll
        for (int r = 0; r < 10; ++r) {
            const uint32_t hi0 = __umulhi(M0, c0), lo0 = M0 * c0;
            const uint32_t hi1 = __umulhi(M1, c2), lo1 = M1 * c2;
            const uint32_t n0 = hi1 ^ c1 ^ k0, n2 = hi0 ^ c3 ^ k1;
            c0 = n0; c1 = lo1; c2 = n2; c3 = lo0;
            k0 += W0; k1 += W1;
        }
        out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
    }
};

__device__ __forceinline__ float u32_to_exp1(uint32_t bits) {
    const float u = ((float)(bits >> 8) + 0.5f) * 5.9604644775390625e-08f;   // (0,1), 2^-24 grid
    return -logf(u);
}

template <int KP>
__global__ __launch_bounds__(256) void k_posterior(const ccdm_post_args a_in) {
    // per-run fields: from the device-resident block when there is one (uniform scalar loads), else the arguments themselves
    ccdm_post_args a = a_in;
    if (a_in.run) {
        const ccdm_post_run r = *a_in.run;
        a.noise = r.noise; a.noise_step_stride = r.noise_step_stride; a.philox_seed = r.philox_seed; a.sample_offset = r.sample_offset;
        a.noise_row0 = r.noise_row0; a.out_probs = r.out_probs; a.out_onehot = r.out_onehot; a.posterior_out = r.posterior_out;
    }
    const size_t npix = (size_t)a.N * a.HW;
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= npix) return;
    const int K = a.K;
    const int step = a.step_ptr ? *a.step_ptr : 0;
    const float* row = a.step_table + (size_t)step * 4;
    const float al = row[0], cu = row[1];
    const int mode = (int)row[2];

    float x0[KP];
    const float* hp = a.head + i * a.head_stride;
#pragma unroll
    for (int k = 0; k < KP; ++k) x0[k] = k < K ? hp[k] : -INFINITY;
    if (a.range_flag) {
        // a non-finite head value is how an F16X3 range overflow anywhere upstream surfaces (include/ccdm_hip.h): NaN/Inf
        // survive every conv, GroupNorm and attention on the way here.  (The clamp below would hide it: fmaxf(NaN, 1e-12) = 1e-12.)
        float chk = 0.f;
#pragma unroll
        for (int k = 0; k < KP; ++k) if (k < K) chk += fabsf(x0[k]);
        if (!(chk <= 3.0e38f)) *a.range_flag = 1;        // benign race: every writer stores the same value
    }
    if (a.softmax) {
        float mx = x0[0];
#pragma unroll
        for (int k = 1; k < KP; ++k) mx = fmaxf(mx, x0[k]);
        float sum = 0.f;
#pragma unroll
        for (int k = 0; k < KP; ++k) {
            x0[k] = k < K ? expf(x0[k] - mx) : 0.f;
            if (k < K) sum += x0[k];
        }
#pragma unroll
        for (int k = 0; k < KP; ++k) x0[k] = x0[k] / sum;
    }
    if (mode == CCDM_STEP_SOFTMAX_ONLY) {
        if (a.out_probs) {
#pragma unroll
            for (int k = 0; k < KP; ++k) if (k < K) a.out_probs[i * K + k] = x0[k];
        }
        return;
    }
    const int xt = a.xt[i];
    const float Kf = (float)K;
    const float u = (1.0f - al) / Kf, b = (1.0f - cu) / Kf;
    float A[KP];
    float S = 0.f;
#pragma unroll
    for (int k = 0; k < KP; ++k) {
        A[k] = (k == xt ? al : 0.0f) + u;          // a*1 + u  /  a*0 + u
        if (k == 0) S = A[0]; else if (k < K) S = S + A[k];
    }
    float r[KP];
    float R = 0.f;
    const float bS = b * S;
#pragma unroll
    for (int k = 0; k < KP; ++k) {
        r[k] = k < K ? x0[k] / (cu * A[k] + bS) : 0.f;
        if (k == 0) R = r[0]; else if (k < K) R = R + r[k];
    }
    const float bR = b * R;
    float P[KP];
#pragma unroll
    for (int k = 0; k < KP; ++k) {
        P[k] = A[k] * (cu * r[k] + bR);
        P[k] = fmaxf(P[k], 1e-12f);
    }
    // normalise, cascade order (== sequential for K <= 16)
    float tot;
    {
        float hi = 0.f, tail = 0.f;
        bool have_hi = false, have_tail = false;
        const int full = (K / 16) * 16;
#pragma unroll
        for (int s0 = 0; s0 < KP; s0 += 16) {
            if (s0 + 16 <= full) {
                float blk = P[s0];
#pragma unroll
                for (int k = 1; k < 16; ++k) if (s0 + k < KP) blk = blk + P[s0 + k];
                hi = have_hi ? hi + blk : blk;
                have_hi = true;
            }
        }
#pragma unroll
        for (int k = 0; k < KP; ++k) {
            if (k >= full && k < K) { tail = have_tail ? tail + P[k] : P[k]; have_tail = true; }
        }
        tot = have_tail ? (have_hi ? tail + hi : tail) : hi;
    }
#pragma unroll
    for (int k = 0; k < KP; ++k) P[k] = P[k] / tot;

    if (a.posterior_out) {
#pragma unroll
        for (int k = 0; k < KP; ++k) if (k < K) a.posterior_out[i * K + k] = P[k];
    }

    if (mode == CCDM_STEP_SAMPLE) {
        float best = -INFINITY;
        int bi = 0;
        if (a.noise) {
            const float* e = a.noise + (size_t)(step - a.noise_row0) * a.noise_step_stride + i * K;
#pragma unroll
            for (int k = 0; k < KP; ++k) {
                if (k < K) {
                    const float qv = P[k] / e[k];
                    if (qv > best) { best = qv; bi = k; }
                }
            }
        } else {
            const uint32_t pix = (uint32_t)(i % a.HW), smp = (uint32_t)(i / a.HW) + a.sample_offset;
            const uint32_t k0 = (uint32_t)a.philox_seed, k1 = (uint32_t)(a.philox_seed >> 32);
#pragma unroll
            for (int kq = 0; kq < (KP + 3) / 4; ++kq) {
                if (kq * 4 < K) {
                    uint32_t w[4];
                    Philox::run(pix, smp, (uint32_t)step, (uint32_t)kq, k0, k1, w);
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int k = kq * 4 + j;
                        if (k < K && k < KP) {
                            const float qv = P[k] / u32_to_exp1(w[j]);
                            if (qv > best) { best = qv; bi = k; }
                        }
                    }
                }
            }
        }
        a.xt_next[i] = (uint8_t)bi;
        if (a.xin) {
            float* d = a.xin + i * a.xin_stride;
#pragma unroll
            for (int k = 0; k < KP; ++k) if (k < K) d[k] = (k == bi) ? 1.0f : 0.0f;
        }
    } else if (mode == CCDM_STEP_LAST_CONFIDENCE) {
        if (a.out_probs) {
#pragma unroll
            for (int k = 0; k < KP; ++k) if (k < K) a.out_probs[i * K + k] = P[k];
        }
    } else if (mode == CCDM_STEP_LAST_MAJORITY) {
        float best = P[0];
        int bi = 0;
#pragma unroll
        for (int k = 1; k < KP; ++k) if (k < K && P[k] > best) { best = P[k]; bi = k; }
        if (a.out_onehot) {
#pragma unroll
            for (int k = 0; k < KP; ++k) if (k < K) a.out_onehot[i * K + k] = (k == bi) ? 1 : 0;
        }
        a.xt_next[i] = (uint8_t)bi;
    }
    // CCDM_STEP_LAST_KEEP: x_t is returned unchanged (step_T_sample neither "majority" nor "confidence")
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
