// The per-pixel arithmetic of the step epilogue — softmax_K -> categorical posterior -> clamp -> normalise -> Exp(1)-race argmax (or the
// last-step outputs) — shared by the stand-alone epilogue kernel (ccdm_sampler.hip) and the head kernel that ends in it
// (ccdm_head.hip): ONE definition, so both produce the same bits from the same logits.  See ccdm_sampler.hip for the arithmetic order.
#pragma once
#include "ccdm_common.h"

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

// per-run fields of the epilogue: from the device-resident block when there is one (uniform scalar loads), else the arguments themselves
__device__ __forceinline__ ccdm_post_args post_resolve_run(const ccdm_post_args& a_in) {
    ccdm_post_args a = a_in;
    if (a_in.run) {
        const ccdm_post_run r = *a_in.run;
        a.noise = r.noise; a.noise_step_stride = r.noise_step_stride; a.philox_seed = r.philox_seed; a.sample_offset = r.sample_offset;
        a.noise_row0 = r.noise_row0; a.out_probs = r.out_probs; a.out_onehot = r.out_onehot; a.posterior_out = r.posterior_out;
    }
    return a;
}

// pixel i (global index n * HW + pixel) with the head's K values x0[0..K) (logits, or probabilities if !a.softmax; x0[k >= K] = -inf);
// `a` already resolved by post_resolve_run, `step` = the table row of this denoise step
// (core: the step's coefficients and the pixel's x_t are handed in, so that a caller may fetch them ahead of time)
template <int KP>
__device__ __forceinline__ void posterior_pixel_core(const ccdm_post_args& a, const size_t i, float (&x0)[KP], const int step, const float al, const float cu,
                                                     const int mode, const int xt, int* const chosen = nullptr) {
    const int K = a.K;
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
        if (chosen) *chosen = bi;            // the caller writes the one-hot channels itself (k_posterior_staged)
        else if (a.xin) {
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

template <int KP>
__device__ __forceinline__ void posterior_pixel(const ccdm_post_args& a, const size_t i, float (&x0)[KP], const int step, int* const chosen = nullptr) {
    const float* row = a.step_table + (size_t)step * 4;
    const float al = row[0], cu = row[1];
    const int mode = (int)row[2];
    const int xt = mode == CCDM_STEP_SOFTMAX_ONLY ? 0 : (int)a.xt[i];
    posterior_pixel_core<KP>(a, i, x0, step, al, cu, mode, xt, chosen);
}

}  // namespace ccdm
