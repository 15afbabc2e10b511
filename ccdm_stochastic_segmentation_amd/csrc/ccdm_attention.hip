// Self-attention core on the matrix cores: head width 32 (every reference U-Net config: num_head_channels: 32) and 64 (the DINO
// ViT-S/8 feature encoder, whose token rows may be padded: Ta rows allocated per sample, T of them tokens).
//
//   softmax((q*s)(k*s)^T) v   per (sample, head),   s = 32^-1/4        unet.py:343-360 (legacy) / :376-395 (new order)
//
// Flash-style streaming over key tiles of 64; the [T,T] score matrix never exists.  Both products are computed
// TRANSPOSED so that the query index is the MFMA column = lane & 31:
//     S^T = K  * Q^T   (A = K tile from LDS,  B = Q fragment held in registers)   -> lane (q, half) holds 16 keys
//     O^T = V^T * P^T  (A = V^T tile from LDS, B = P straight from the S^T registers) -> lane (q, half) holds 16 d's
// so the row max / row sum are 15 in-lane ops + one lane^32 exchange, the online rescale is a per-lane scalar, and
// P never moves between lanes: the k-slot a lane's 8 operand elements occupy is arbitrary as long as A and B agree,
// so V^T is simply read with the key permutation the S^T accumulator layout already has.
// Precision: every product is the 3-term fp16 hi/lo split (lo*hi + hi*lo + hi*hi, fp32 accumulate, ~2^-22) used by
// the convs; P in [0,1] needs no scaling.  exp is v_exp_f32.
#include "ccdm_common.h"

namespace ccdm {

static constexpr int KT = 64;              // keys per tile
template <int D> struct KRow { static constexpr int B = 4 * D + 16; };   // bytes per K-tile row: D hi | D lo halfs | 16 pad (36 / 68 dwords: conflict-free b128)
static constexpr int VROW = 2 * KT * 2 + 8;  // bytes per V^T row: 64 hi | 64 lo halfs | 8 pad (66 dwords... 8-B aligned, b64 reads)

__device__ __forceinline__ void split4(const float4 v, f16x4& hi, f16x4& lo) {
    hi[0] = (_Float16)v.x; hi[1] = (_Float16)v.y; hi[2] = (_Float16)v.z; hi[3] = (_Float16)v.w;
    lo[0] = (_Float16)(v.x - (float)hi[0]); lo[1] = (_Float16)(v.y - (float)hi[1]);
    lo[2] = (_Float16)(v.z - (float)hi[2]); lo[3] = (_Float16)(v.w - (float)hi[3]);
}

template <int WAVES, int D>
__global__ __launch_bounds__(WAVES * 64) void k_attention_mfma(const float* __restrict__ qkv, float* __restrict__ out,
                                                              int T, int Ta, int C, int order) {
    constexpr int NT = WAVES * 64, KROW = KRow<D>::B, DS = D / 16 /* 16-wide k-steps over d */, DM = D / 32 /* 32-row tiles of d */;
    __shared__ __attribute__((aligned(16))) char kt[KT * KROW];
    __shared__ __attribute__((aligned(16))) char vt[D * VROW];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int h = blockIdx.y, n = blockIdx.z;
    const int q0 = (blockIdx.x * WAVES + wave) * 32;
    const int C3 = 3 * C;
    int qoff, koff, voff;
    if (order == 0) { qoff = h * 3 * D; koff = qoff + D; voff = qoff + 2 * D; }
    else { qoff = h * D; koff = C + h * D; voff = 2 * C + h * D; }
    const float scale = (float)(1.0 / sqrt(sqrt((double)D)));
    const float* base = qkv + (size_t)n * Ta * C3;
    const int qi = lane & 31, half = lane >> 5;

    // ---- Q^T fragment (B operand): column = query, k-slot (half, j) = d 16*s + 8*half + j ----
    f16x8 qh[DS], ql[DS];
    {
        const int tq = min(q0 + qi, T - 1);
#pragma unroll
        for (int s = 0; s < DS; ++s) {
            const float* p = base + (size_t)tq * C3 + qoff + 16 * s + 8 * half;
            float4 a = *reinterpret_cast<const float4*>(p), b = *reinterpret_cast<const float4*>(p + 4);
            a.x *= scale; a.y *= scale; a.z *= scale; a.w *= scale;
            b.x *= scale; b.y *= scale; b.z *= scale; b.w *= scale;
            f16x4 h0, l0, h1, l1;
            split4(a, h0, l0); split4(b, h1, l1);
#pragma unroll
            for (int j = 0; j < 4; ++j) { qh[s][j] = h0[j]; qh[s][4 + j] = h1[j]; ql[s][j] = l0[j]; ql[s][4 + j] = l1[j]; }
        }
    }
    f32x16 o[DM];
#pragma unroll
    for (int mt = 0; mt < DM; ++mt)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[mt][r] = 0.f;
    float m = -INFINITY, l = 0.f;

    // K/V staging items of a key tile: KT keys x D/4 float4 columns over NT threads.  The next tile's are requested (into registers,
    // rows clamped into the sample; rows beyond T are zeroed at staging) before the current tile's MFMAs, so their round trip to L2/HBM
    // runs under the matrix work instead of in front of the next staging pass.
    constexpr int NSTG = KT * (D / 4) / NT;
    static_assert(KT * (D / 4) % NT == 0, "staging items divide evenly over the block");
    f32x4 pk[NSTG], pv[NSTG];
    auto request = [&](const int j0) {
#pragma unroll
        for (int u = 0; u < NSTG; ++u) {
            const int item = tid + u * NT;
            const int key = min(j0 + item / (D / 4), T - 1), c4 = item % (D / 4);
            const float* p = base + (size_t)key * C3;
            pk[u] = *reinterpret_cast<const f32x4*>(p + koff + 4 * c4);
            pv[u] = *reinterpret_cast<const f32x4*>(p + voff + 4 * c4);
        }
    };
    request(0);
    for (int j0 = 0; j0 < T; j0 += KT) {
        __syncthreads();
        // ---- stage K (row-major, scaled) and V (transposed) tiles as fp16 hi/lo ----
#pragma unroll
        for (int u = 0; u < NSTG; ++u) {
            const int item = tid + u * NT;
            const int key = item / (D / 4), c4 = item % (D / 4);
            const bool in = j0 + key < T;
            float4 kv = make_float4(in ? pk[u][0] : 0.f, in ? pk[u][1] : 0.f, in ? pk[u][2] : 0.f, in ? pk[u][3] : 0.f);
            const float4 vv = make_float4(in ? pv[u][0] : 0.f, in ? pv[u][1] : 0.f, in ? pv[u][2] : 0.f, in ? pv[u][3] : 0.f);
            kv.x *= scale; kv.y *= scale; kv.z *= scale; kv.w *= scale;
            f16x4 hi, lo;
            split4(kv, hi, lo);
            *reinterpret_cast<f16x4*>(kt + key * KROW + 8 * c4) = hi;
            *reinterpret_cast<f16x4*>(kt + key * KROW + 2 * D + 8 * c4) = lo;
            split4(vv, hi, lo);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                *reinterpret_cast<_Float16*>(vt + (4 * c4 + e) * VROW + 2 * key) = hi[e];
                *reinterpret_cast<_Float16*>(vt + (4 * c4 + e) * VROW + 2 * KT + 2 * key) = lo[e];
            }
        }
        __syncthreads();
        if (j0 + KT < T) request(j0 + KT);
        const int nsub = (T - j0) >= KT ? 2 : ((T - j0) + 31) / 32;     // 32-key sub-tiles in this tile
        f32x16 sc[2];
        float mx = m;
#pragma unroll
        for (int st = 0; st < 2; ++st) {
            if (st < nsub) {
                f32x16 acc;
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
                for (int s = 0; s < DS; ++s) {     // A = K rows (key = 32*st + lane&31), k-slot (half, j) = d 16*s + 8*half + j
                    const char* p = kt + (32 * st + qi) * KROW + 32 * s + 16 * half;
                    const f16x8 kh = *reinterpret_cast<const f16x8*>(p);
                    const f16x8 kl = *reinterpret_cast<const f16x8*>(p + 2 * D);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(kl, qh[s], acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(kh, ql[s], acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(kh, qh[s], acc, 0, 0, 0);
                }
                // acc[r] = score(query = lane&31, key = 32*st + (r&3) + 8*(r>>2) + 4*half); mask keys beyond T
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int key = j0 + 32 * st + (r & 3) + 8 * (r >> 2) + 4 * half;
                    acc[r] = key < T ? acc[r] : -INFINITY;
                    mx = fmaxf(mx, acc[r]);
                }
                sc[st] = acc;
            }
        }
        mx = fmaxf(mx, __shfl_xor(mx, 32));
        const float corr = __expf(m - mx);            // first tile: exp(-inf) = 0
        m = mx;
        l *= corr;
#pragma unroll
        for (int mt = 0; mt < DM; ++mt)
#pragma unroll
            for (int r = 0; r < 16; ++r) o[mt][r] *= corr;
#pragma unroll
        for (int st = 0; st < 2; ++st) {
            if (st < nsub) {
                float p[16];
#pragma unroll
                for (int r = 0; r < 16; ++r) { p[r] = __expf(sc[st][r] - mx); l += p[r]; }
                // O^T += V^T * P^T.  k-step s covers this lane's registers r = 8s..8s+7, i.e. keys
                // 32*st + 16*s + {0..3, 8..11} + 4*half — the same keys are read from V^T for the A operand.
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    f16x8 ph, pl;
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const float pv = p[8 * s + j];
                        ph[j] = (_Float16)pv;
                        pl[j] = (_Float16)(pv - (float)ph[j]);
                    }
#pragma unroll
                    for (int mt = 0; mt < DM; ++mt) {
                        const char* vp = vt + (32 * mt + qi) * VROW + 2 * (32 * st + 16 * s + 4 * half);     // row d = 32*mt + lane&31
                        f16x8 vh, vl;
                        const f16x4 vh0 = *reinterpret_cast<const f16x4*>(vp), vh1 = *reinterpret_cast<const f16x4*>(vp + 16);
                        const f16x4 vl0 = *reinterpret_cast<const f16x4*>(vp + 2 * KT), vl1 = *reinterpret_cast<const f16x4*>(vp + 2 * KT + 16);
#pragma unroll
                        for (int j = 0; j < 4; ++j) { vh[j] = vh0[j]; vh[4 + j] = vh1[j]; vl[j] = vl0[j]; vl[4 + j] = vl1[j]; }
                        o[mt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vl, ph, o[mt], 0, 0, 0);
                        o[mt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vh, pl, o[mt], 0, 0, 0);
                        o[mt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vh, ph, o[mt], 0, 0, 0);
                    }
                }
            }
        }
    }
    l += __shfl_xor(l, 32);
    if (q0 + qi < T) {
        const float inv = 1.0f / l;
        // o[r] = O[query = lane&31][d = (r&3) + 8*(r>>2) + 4*half]: four float4 rows of 4 consecutive d each
        float* dst = out + ((size_t)n * Ta + q0 + qi) * C + h * D + 4 * half;
#pragma unroll
        for (int mt = 0; mt < DM; ++mt)
#pragma unroll
            for (int g = 0; g < 4; ++g)
                *reinterpret_cast<float4*>(dst + 32 * mt + 8 * g) =
                    make_float4(o[mt][4 * g] * inv, o[mt][4 * g + 1] * inv, o[mt][4 * g + 2] * inv, o[mt][4 * g + 3] * inv);
    }
}

int launch_attention_mfma(const float* qkv, float* out, int N, int T, int Ta, int C, int heads, int order, hipStream_t s) {
    const int D = C / heads;
    const int waves = T >= 128 ? 4 : (T >= 64 ? 2 : 1);
    dim3 grid(cdiv(T, 32 * waves), heads, N);
    if (D == 64) {
        if (waves == 4) hipLaunchKernelGGL((k_attention_mfma<4, 64>), grid, dim3(256), 0, s, qkv, out, T, Ta, C, order);
        else if (waves == 2) hipLaunchKernelGGL((k_attention_mfma<2, 64>), grid, dim3(128), 0, s, qkv, out, T, Ta, C, order);
        else hipLaunchKernelGGL((k_attention_mfma<1, 64>), grid, dim3(64), 0, s, qkv, out, T, Ta, C, order);
    } else {
        if (waves == 4) hipLaunchKernelGGL((k_attention_mfma<4, 32>), grid, dim3(256), 0, s, qkv, out, T, Ta, C, order);
        else if (waves == 2) hipLaunchKernelGGL((k_attention_mfma<2, 32>), grid, dim3(128), 0, s, qkv, out, T, Ta, C, order);
        else hipLaunchKernelGGL((k_attention_mfma<1, 32>), grid, dim3(64), 0, s, qkv, out, T, Ta, C, order);
    }
    CCDM_CHECK_LAUNCH("attention_mfma");
    return 0;
}

}  // namespace ccdm
