// EXPERIMENT (round 3, measured, NOT adopted: compiled only into CCDM_EXPERIMENTS builds, tools/bench_attention.py / tools/visit_attn*.sh
// time it; DESIGN.md section 9 has the numbers).  Self-attention core for LONG sequences at head width 32 (the Cityscapes-shaped
// configs: 2048 and 8192 tokens) with the fp16 hi/lo split of K and V taken out of the key loop, two query tiles per wave, and the
// softmax of one query tile issued between the MFMAs of the other.  Result on MI355X (T = 8192, N = 4, 4 heads): 504-529 us in every
// variant against 514-521 us of ccdm_attention.hip — the kernel's time is the SUM of its matrix and vector work whatever the issue
// order (phase ablations: MFMAs + K fragment reads alone 322 us, + tile copy / barrier / V reads 369, + P split 417, + softmax 514),
// and the matrix work alone already runs at the data-dependent (power-limited) MFMA rate: tools/ubench/mfma_chain.hip measures 38.7
// "cycles at 2.0 GHz" per v_mfma_f32_32x32x16_f16 with random operands against 27.1 with smooth ones, i.e. ~1.73 PFLOP/s, not 2.5.
//
//   softmax((q*s)(k*s)^T) v   per (sample, head),   s = 32^-1/4        unet.py:343-360 (legacy) / :376-395 (new order)
//
// ccdm_attention.hip splits every K/V tile to fp16 hi/lo while staging it — once per block, i.e. T/256 times per key — and runs one
// 32-query tile per wave; at T = 8192 its vector pipe (66 % active) is the bound, not the matrix pipe (44 % busy).  Here
//   1. k_kv_split32 writes, once per launch, the image the key loop wants: per (sample, head, 64-key tile) 16 KB =
//      K [64 keys][32 hi | 32 lo halfs] (scaled by s) and V [hi|lo][d/16][64 keys][16 halfs] — byte for byte what the key loop
//      keeps in LDS (keys beyond T are zeros).  It lives in a caller-provided workspace (ccdm_attention_workspace_bytes): the library
//      allocates nothing;
//   2. the key loop copies a tile with two 16-byte loads per thread (no arithmetic) and every wave runs TWO 32-query tiles against
//      it: the K and V^T operand fragments are read from LDS once per 2048 scores instead of once per 1024;
//   3. row maxima by v_max3, the exponent's subtraction and the row sums by packed fp32 adds.
// Same arithmetic as ccdm_attention.hip (3-term fp16 split products, fp32 accumulate, exp2 with log2(e) folded into q), same LDS
// layouts and operand orientation — see the comments there.
#include "ccdm_common.h"
#include "ccdm_conv_common.h"
#include <type_traits>

namespace ccdm {

static constexpr int SKT = 64;                      // keys per tile
static constexpr int SKROW = 4 * 32 + 16;           // bytes per K row in LDS: 32 hi | 32 lo halfs | 16 pad (36 dwords: conflict-free b128)
static constexpr int SVPLANE = SKT * 32 + 128;      // one V plane (16 d-columns of 64 keys) + 128 B so two planes sit on opposite bank halves
static constexpr int KV_TILE_BYTES = 16384;         // K image 8 KB (rows of 128 B, no pad) + V image 8 KB ([hi|lo][plane][key][16 halfs])

typedef unsigned s_u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ void split8_frag_s(const float* v, f16x8& hi, f16x8& lo) {
    s_u32x4 h, l;
    unsigned a, b;
    split2_f16(v[0], v[1], a, b); h[0] = a; l[0] = b;
    split2_f16(v[2], v[3], a, b); h[1] = a; l[1] = b;
    split2_f16(v[4], v[5], a, b); h[2] = a; l[2] = b;
    split2_f16(v[6], v[7], a, b); h[3] = a; l[3] = b;
    hi = __builtin_bit_cast(f16x8, h);
    lo = __builtin_bit_cast(f16x8, l);
}

// one thread = one (key, 8 d-columns) item of a tile: two float4 of K and of V in, 16 B hi + 16 B lo of each out
__global__ __launch_bounds__(256) void k_kv_split32(const float* __restrict__ qkv, char* __restrict__ img, int T, int Ta, int C, int order) {
    const int tile = blockIdx.x, h = blockIdx.y, n = blockIdx.z, tid = threadIdx.x;
    const int C3 = 3 * C;
    const int koff = order == 0 ? h * 96 + 32 : C + h * 32, voff = order == 0 ? h * 96 + 64 : 2 * C + h * 32;
    const float scale = (float)(1.0 / sqrt(sqrt(32.0)));
    const int key = tid >> 2, c8 = tid & 3;
    const int tk = tile * SKT + key;
    const bool in = tk < T;
    const float* p = qkv + ((size_t)n * Ta + min(tk, T - 1)) * C3;
    const float4 k0 = *reinterpret_cast<const float4*>(p + koff + 8 * c8), k1 = *reinterpret_cast<const float4*>(p + koff + 8 * c8 + 4);
    const float4 v0 = *reinterpret_cast<const float4*>(p + voff + 8 * c8), v1 = *reinterpret_cast<const float4*>(p + voff + 8 * c8 + 4);
    const float ks = in ? scale : 0.f, vs = in ? 1.f : 0.f;
    const float kk[8] = {k0.x * ks, k0.y * ks, k0.z * ks, k0.w * ks, k1.x * ks, k1.y * ks, k1.z * ks, k1.w * ks};
    const float vv[8] = {v0.x * vs, v0.y * vs, v0.z * vs, v0.w * vs, v1.x * vs, v1.y * vs, v1.z * vs, v1.w * vs};
    f16x8 kh, kl, vh, vl;
    split8_frag_s(kk, kh, kl);
    split8_frag_s(vv, vh, vl);
    char* dst = img + (((size_t)n * gridDim.y + h) * gridDim.x + tile) * KV_TILE_BYTES;
    *reinterpret_cast<f16x8*>(dst + key * 128 + 16 * c8) = kh;
    *reinterpret_cast<f16x8*>(dst + key * 128 + 64 + 16 * c8) = kl;
    char* vd = dst + 8192 + (c8 >> 1) * 2048 + key * 32 + (c8 & 1) * 16;      // plane = d / 16, then key-major rows of 16 halfs
    *reinterpret_cast<f16x8*>(vd) = vh;
    *reinterpret_cast<f16x8*>(vd + 4096) = vl;                                  // the lo image follows the two hi planes
}

template <int WAVES, int QT>
__global__ __launch_bounds__(WAVES * 64) void k_attention_split32(const float* __restrict__ qkv, const char* __restrict__ img, float* __restrict__ out,
                                                                 int T, int Ta, int C, int order, int ntile) {
    constexpr int NT = WAVES * 64, D = 32, DS = 2;
    constexpr int NCH = 512 / NT;                   // 16-byte chunks of each of the K and V images per thread
    static_assert(512 % NT == 0, "a tile's chunks divide evenly over the block");
    __shared__ __attribute__((aligned(16))) char kt[SKT * SKROW];
    __shared__ __attribute__((aligned(16))) char vt[4 * SVPLANE];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int h = blockIdx.y, n = blockIdx.z, heads = gridDim.y;
    const int q0 = (blockIdx.x * WAVES + wave) * 32 * QT;
    const int C3 = 3 * C;
    const int qoff = order == 0 ? h * 96 : h * 32;
    const float qscale = (float)(1.0 / sqrt(sqrt(32.0))) * 1.4426950408889634f;       // scores in units of log2(e): every exponential is one v_exp_f32
    const float* base = qkv + (size_t)n * Ta * C3;
    const int qi = lane & 31, half = lane >> 5;

    // ---- Q^T fragments (B operand): column = query, k-slot (half, j) = d 16*s + 8*half + j ----
    f16x8 qh[QT][DS], ql[QT][DS];
#pragma unroll
    for (int t = 0; t < QT; ++t) {
        const int tq = min(q0 + 32 * t + qi, T - 1);
#pragma unroll
        for (int s = 0; s < DS; ++s) {
            const float* p = base + (size_t)tq * C3 + qoff + 16 * s + 8 * half;
            const float4 a = *reinterpret_cast<const float4*>(p), b = *reinterpret_cast<const float4*>(p + 4);
            const float v[8] = {a.x * qscale, a.y * qscale, a.z * qscale, a.w * qscale, b.x * qscale, b.y * qscale, b.z * qscale, b.w * qscale};
            split8_frag_s(v, qh[t][s], ql[t][s]);
        }
    }
    f32x16 o[QT];
    float m[QT], l[QT];
#pragma unroll
    for (int t = 0; t < QT; ++t) {
#pragma unroll
        for (int r = 0; r < 16; ++r) o[t][r] = 0.f;
        m[t] = -INFINITY;
        l[t] = 0.f;
    }
    const char* src = img + ((size_t)n * heads + h) * ntile * KV_TILE_BYTES;
    s_u32x4 pk[NCH], pv[NCH];
    auto request = [&](const int tile) {
#pragma unroll
        for (int u = 0; u < NCH; ++u) {
            const char* p = src + (size_t)tile * KV_TILE_BYTES + (tid + u * NT) * 16;
            pk[u] = *reinterpret_cast<const s_u32x4*>(p);
            pv[u] = *reinterpret_cast<const s_u32x4*>(p + 8192);
        }
    };
    // per-lane part of the V^T fragment address (ds_read_b64_tr_b16; see ccdm_attention.hip)
    const unsigned vlane = (unsigned)(size_t)vt + ((lane >> 4) & 1) * SVPLANE + (4 * half + ((lane & 15) >> 2)) * 32 + (lane & 3) * 8;
    request(0);
    for (int tile = 0; tile < ntile; ++tile) {
        const int j0 = tile * SKT;
        __syncthreads();
#pragma unroll
        for (int u = 0; u < NCH; ++u) {
            const int g = tid + u * NT;
            *reinterpret_cast<s_u32x4*>(kt + (g >> 3) * SKROW + (g & 7) * 16) = pk[u];
            *reinterpret_cast<s_u32x4*>(vt + (g >> 7) * SVPLANE + (g & 127) * 16) = pv[u];
        }
        __syncthreads();
        if (tile + 1 < ntile) request(tile + 1);
        const int nsub = (T - j0) >= SKT ? 2 : ((T - j0) + 31) / 32;     // 32-key sub-tiles in this tile
        f32x16 sc[QT][2];
#pragma unroll
        for (int st = 0; st < 2; ++st) {
            if (st < nsub) {
                f16x8 kh[DS], kl[DS];
#pragma unroll
                for (int s = 0; s < DS; ++s) {     // A = K rows (key = 32*st + lane&31), k-slot (half, j) = d 16*s + 8*half + j
                    const char* p = kt + (32 * st + qi) * SKROW + 32 * s + 16 * half;
                    kh[s] = *reinterpret_cast<const f16x8*>(p);
                    kl[s] = *reinterpret_cast<const f16x8*>(p + 2 * D);
                }
#pragma unroll
                for (int t = 0; t < QT; ++t) {
                    f32x16 acc;
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
                    for (int s = 0; s < DS; ++s) {
                        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(kl[s], qh[t][s], acc, 0, 0, 0);
                        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(kh[s], ql[t][s], acc, 0, 0, 0);
                        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(kh[s], qh[t][s], acc, 0, 0, 0);
                    }
                    // acc[r] = score(query = lane&31, key = 32*st + (r&3) + 8*(r>>2) + 4*half); only the last tile can hold keys beyond T
                    if (j0 + SKT > T) {
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            const int key = j0 + 32 * st + (r & 3) + 8 * (r >> 2) + 4 * half;
                            acc[r] = key < T ? acc[r] : -INFINITY;
                        }
                    }
                    sc[t][st] = acc;
                }
            }
        }
        // ---- online softmax per query tile: p = 2^(s - max) in place ----
#pragma unroll
        for (int t = 0; t < QT; ++t) {
            float mx = m[t];
#pragma unroll
            for (int st = 0; st < 2; ++st) {
                if (st < nsub) {
#pragma unroll
                    for (int r = 0; r < 16; r += 2) mx = fmaxf(fmaxf(mx, sc[t][st][r]), sc[t][st][r + 1]);      // v_max3_f32
                }
            }
            mx = fmaxf(mx, __shfl_xor(mx, 32));
            if (__builtin_amdgcn_ballot_w64(mx > m[t]) != 0ull) {     // some query's running maximum grew: rescale (else the factor is exactly 1)
                const float corr = __builtin_amdgcn_exp2f(m[t] - mx);     // first tile: 2^(-inf) = 0
                m[t] = mx;
                l[t] *= corr;
#pragma unroll
                for (int r = 0; r < 16; ++r) o[t][r] *= corr;
            }
            const f32x2 mx2 = {mx, mx};
            f32x2 lsum = {0.f, 0.f};
#pragma unroll
            for (int st = 0; st < 2; ++st) {
                if (st < nsub) {
#pragma unroll
                    for (int r = 0; r < 16; r += 2) {
                        f32x2 e = {sc[t][st][r], sc[t][st][r + 1]};
                        e = e - mx2;                                                       // v_pk_add_f32
                        e[0] = __builtin_amdgcn_exp2f(e[0]);
                        e[1] = __builtin_amdgcn_exp2f(e[1]);
                        lsum = lsum + e;                                                   // v_pk_add_f32
                        sc[t][st][r] = e[0];
                        sc[t][st][r + 1] = e[1];
                    }
                }
            }
            l[t] += lsum[0] + lsum[1];
        }
        // ---- O^T += V^T * P^T: the V^T fragments of a k-step are read once and serve both query tiles ----
#pragma unroll
        for (int st = 0; st < 2; ++st) {
            if (st < nsub) {
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    // lane (d = lane&31, half): hi keys +0..3, +8..11 of k-step s, then the same of the lo image
                    f16x4 vh0, vh1, vl0, vl1;
                    const unsigned va = vlane + (32 * st + 16 * s) * 32;
                    asm volatile("ds_read_b64_tr_b16 %0, %4\n\t"
                                 "ds_read_b64_tr_b16 %1, %4 offset:256\n\t"
                                 "ds_read_b64_tr_b16 %2, %4 offset:%5\n\t"
                                 "ds_read_b64_tr_b16 %3, %4 offset:%6\n\t"
                                 "s_waitcnt lgkmcnt(0)"
                                 : "=&v"(vh0), "=&v"(vh1), "=&v"(vl0), "=&v"(vl1)
                                 : "v"(va), "i"(2 * SVPLANE), "i"(2 * SVPLANE + 256)
                                 : "memory");
                    f16x8 vh, vl;
#pragma unroll
                    for (int j = 0; j < 4; ++j) { vh[j] = vh0[j]; vh[4 + j] = vh1[j]; vl[j] = vl0[j]; vl[4 + j] = vl1[j]; }
#pragma unroll
                    for (int t = 0; t < QT; ++t) {
                        float p[8];
#pragma unroll
                        for (int j = 0; j < 8; ++j) p[j] = sc[t][st][8 * s + j];
                        f16x8 ph, pl;
                        split8_frag_s(p, ph, pl);
                        o[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vl, ph, o[t], 0, 0, 0);
                        o[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vh, pl, o[t], 0, 0, 0);
                        o[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vh, ph, o[t], 0, 0, 0);
                    }
                }
            }
        }
    }
#pragma unroll
    for (int t = 0; t < QT; ++t) {
        const float lt = l[t] + __shfl_xor(l[t], 32);
        if (q0 + 32 * t + qi < T) {
            const float inv = 1.0f / lt;
            // o[r] = O[query = lane&31][d = (r&3) + 8*(r>>2) + 4*half]: four float4 rows of 4 consecutive d each
            float* dst = out + ((size_t)n * Ta + q0 + 32 * t + qi) * C + h * D + 4 * half;
#pragma unroll
            for (int g = 0; g < 4; ++g)
                *reinterpret_cast<float4*>(dst + 8 * g) = make_float4(o[t][4 * g] * inv, o[t][4 * g + 1] * inv, o[t][4 * g + 2] * inv, o[t][4 * g + 3] * inv);
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------------
// The same key loop, software-pipelined over the two query tiles of a wave.  Measured (tools/ubench/mfma_valu_interleave.hip): plain
// vector instructions of the SAME wave hide in the matrix pipe's 32-cycle issue interval (~6 per MFMA), those of another wave do not —
// and the unpipelined loop above runs its three phases one after the other (QK^T MFMAs, softmax on the vector pipe, PV MFMAs), so its
// time is the SUM of the two pipes (T = 8192: 384 matrix + ~770 vector cycles per 1024 scores, 1220 measured).  Here the softmax of
// query tile 0 is issued between the QK^T MFMAs of tile 1, and the softmax of tile 1 between the PV MFMAs of tile 0:
//     S1: S^T(0)   |   S2: S^T(1) + softmax(0), split P(0)   |   S3: O^T(0) += V^T P^T(0) + softmax(1)   |   S4: split P(1), O^T(1) +=
// Full key tiles only; a ragged last tile takes the plain phase order (with the key mask).  K/V tiles are double-buffered in LDS: one
// barrier per key tile.
// ---------------------------------------------------------------------------------------------------------------------------
struct VFrag { f16x8 h[4], l[4]; };       // V^T operand fragments of the four 16-key k-steps of a tile (hi, lo)

__device__ __forceinline__ void read_vfrag(const unsigned vlane, VFrag& v) {
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {      // k-step ks = 2*st + s covers keys 32*st + 16*s + {0..3, 8..11} + 4*half
        f16x4 vh0, vh1, vl0, vl1;
        const unsigned va = vlane + 16 * ks * 32;
        asm volatile("ds_read_b64_tr_b16 %0, %4\n\t"
                     "ds_read_b64_tr_b16 %1, %4 offset:256\n\t"
                     "ds_read_b64_tr_b16 %2, %4 offset:%5\n\t"
                     "ds_read_b64_tr_b16 %3, %4 offset:%6\n\t"
                     "s_waitcnt lgkmcnt(0)"
                     : "=&v"(vh0), "=&v"(vh1), "=&v"(vl0), "=&v"(vl1)
                     : "v"(va), "i"(2 * SVPLANE), "i"(2 * SVPLANE + 256)
                     : "memory");
#pragma unroll
        for (int j = 0; j < 4; ++j) { v.h[ks][j] = vh0[j]; v.h[ks][4 + j] = vh1[j]; v.l[ks][j] = vl0[j]; v.l[ks][4 + j] = vl1[j]; }
    }
}

struct PFrag { f16x8 h[4], l[4]; };       // P^T operand fragments (B operand) of the four k-steps

// running maximum / rescale / exponentials / row sum of one query tile over a 64-key tile; sc = the two 32-key score accumulators
template <bool MASK, int ABL = 0>
__device__ __forceinline__ void softmax_tile(f32x16 (&sc)[2], float& m, float& l, f32x16& o, const int j0, const int T, const int half, const int nsub) {
    if (ABL & 1) { l += sc[0][0] + sc[1][0]; return; }
    if (MASK) {
#pragma unroll
        for (int st = 0; st < 2; ++st)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int key = j0 + 32 * st + (r & 3) + 8 * (r >> 2) + 4 * half;
                sc[st][r] = (key < T && st < nsub) ? sc[st][r] : -INFINITY;
            }
    }
    float mx = m;
#pragma unroll
    for (int st = 0; st < 2; ++st)
#pragma unroll
        for (int r = 0; r < 16; r += 2) mx = fmaxf(fmaxf(mx, sc[st][r]), sc[st][r + 1]);      // v_max3_f32
    mx = fmaxf(mx, __shfl_xor(mx, 32));
    const float corr = __builtin_amdgcn_exp2f(m - mx);     // first tile: 2^(-inf) = 0; maximum unchanged: exactly 1
    m = mx;
    const f32x2 c2 = {corr, corr};
#pragma unroll
    for (int r = 0; r < 16; r += 2) {                      // v_pk_mul_f32
        f32x2 v = {o[r], o[r + 1]};
        v = v * c2;
        o[r] = v[0]; o[r + 1] = v[1];
    }
    const f32x2 mx2 = {mx, mx};
    f32x2 lsum = {0.f, 0.f};
#pragma unroll
    for (int st = 0; st < 2; ++st)
#pragma unroll
        for (int r = 0; r < 16; r += 2) {
            f32x2 e = {sc[st][r], sc[st][r + 1]};
            e = e - mx2;                                   // v_pk_add_f32
            e[0] = __builtin_amdgcn_exp2f(e[0]);
            e[1] = __builtin_amdgcn_exp2f(e[1]);
            lsum = lsum + e;
            sc[st][r] = e[0]; sc[st][r + 1] = e[1];
        }
    l = l * corr + (lsum[0] + lsum[1]);
}

template <int ABL = 0>
__device__ __forceinline__ void split_p(const f32x16 (&sc)[2], PFrag& p) {
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
        if (ABL & 2) {
            const f32x4 a = {sc[ks >> 1][8 * (ks & 1)], sc[ks >> 1][8 * (ks & 1) + 1], sc[ks >> 1][8 * (ks & 1) + 2], sc[ks >> 1][8 * (ks & 1) + 3]};
            const f32x4 b = {sc[ks >> 1][8 * (ks & 1) + 4], sc[ks >> 1][8 * (ks & 1) + 5], sc[ks >> 1][8 * (ks & 1) + 6], sc[ks >> 1][8 * (ks & 1) + 7]};
            p.h[ks] = __builtin_bit_cast(f16x8, a); p.l[ks] = __builtin_bit_cast(f16x8, b);
            continue;
        }
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = sc[ks >> 1][8 * (ks & 1) + j];
        split8_frag_s(v, p.h[ks], p.l[ks]);
    }
}

// ABL (CCDM_EXPERIMENTS builds only; results are wrong by construction): 1 no exponentials / maxima, 2 no P split, 4 no PV MFMAs, 8 no QK MFMAs,
// 16 no tile copy / barrier (the first tile is reused), 32 no V fragment reads
template <int WAVES, int ABL = 0>
__global__ __launch_bounds__(WAVES * 64) void k_attention_split32p(const float* __restrict__ qkv, const char* __restrict__ img, float* __restrict__ out,
                                                                  int T, int Ta, int C, int order, int ntile) {
    constexpr int NT = WAVES * 64, D = 32, DS = 2, QT = 2;
    constexpr int NCH = 512 / NT;
    constexpr int KBUF = SKT * SKROW, VBUF = 4 * SVPLANE, BUF = KBUF + VBUF;
    __shared__ __attribute__((aligned(16))) char lds[2 * BUF];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int h = blockIdx.y, n = blockIdx.z, heads = gridDim.y;
    const int q0 = (blockIdx.x * WAVES + wave) * 32 * QT;
    const int C3 = 3 * C;
    const int qoff = order == 0 ? h * 96 : h * 32;
    const float qscale = (float)(1.0 / sqrt(sqrt(32.0))) * 1.4426950408889634f;
    const float* base = qkv + (size_t)n * Ta * C3;
    const int qi = lane & 31, half = lane >> 5;

    f16x8 qh[QT][DS], ql[QT][DS];
#pragma unroll
    for (int t = 0; t < QT; ++t) {
        const int tq = min(q0 + 32 * t + qi, T - 1);
#pragma unroll
        for (int s = 0; s < DS; ++s) {
            const float* p = base + (size_t)tq * C3 + qoff + 16 * s + 8 * half;
            const float4 a = *reinterpret_cast<const float4*>(p), b = *reinterpret_cast<const float4*>(p + 4);
            const float v[8] = {a.x * qscale, a.y * qscale, a.z * qscale, a.w * qscale, b.x * qscale, b.y * qscale, b.z * qscale, b.w * qscale};
            split8_frag_s(v, qh[t][s], ql[t][s]);
        }
    }
    f32x16 o0, o1;
#pragma unroll
    for (int r = 0; r < 16; ++r) { o0[r] = 0.f; o1[r] = 0.f; }
    float m0 = -INFINITY, m1 = -INFINITY, l0 = 0.f, l1 = 0.f;

    const char* src = img + ((size_t)n * heads + h) * ntile * KV_TILE_BYTES;
    s_u32x4 pk[NCH], pv[NCH];
    auto request = [&](const int tile) {
#pragma unroll
        for (int u = 0; u < NCH; ++u) {
            const char* p = src + (size_t)tile * KV_TILE_BYTES + (tid + u * NT) * 16;
            pk[u] = *reinterpret_cast<const s_u32x4*>(p);
            pv[u] = *reinterpret_cast<const s_u32x4*>(p + 8192);
        }
    };
    auto commit = [&](char* buf) {
#pragma unroll
        for (int u = 0; u < NCH; ++u) {
            const int g = tid + u * NT;
            *reinterpret_cast<s_u32x4*>(buf + (g >> 3) * SKROW + (g & 7) * 16) = pk[u];
            *reinterpret_cast<s_u32x4*>(buf + KBUF + (g >> 7) * SVPLANE + (g & 127) * 16) = pv[u];
        }
    };
    auto qk = [&](const char* kt, const int t, f32x16 (&sc)[2]) {
#pragma unroll
        for (int st = 0; st < 2; ++st) {
            f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
            for (int s = 0; s < DS; ++s) {
                const char* p = kt + (32 * st + qi) * SKROW + 32 * s + 16 * half;
                const f16x8 kh = *reinterpret_cast<const f16x8*>(p);
                const f16x8 kl = *reinterpret_cast<const f16x8*>(p + 2 * D);
                if (ABL & 8) {
#pragma unroll
                    for (int r = 0; r < 8; ++r) acc[r] += (float)kh[r] + (float)kl[r] * (float)qh[t][s][r];
                    continue;
                }
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(kl, qh[t][s], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(kh, ql[t][s], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(kh, qh[t][s], acc, 0, 0, 0);
            }
            sc[st] = acc;
        }
    };
    auto pvmul = [&](const VFrag& v, const PFrag& p, f32x16& o) {
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            if (ABL & 4) {
#pragma unroll
                for (int r = 0; r < 8; ++r) o[r] += (float)v.l[ks][r] * (float)p.h[ks][r] + (float)v.h[ks][r] * (float)p.l[ks][r];
                continue;
            }
            o = __builtin_amdgcn_mfma_f32_32x32x16_f16(v.l[ks], p.h[ks], o, 0, 0, 0);
            o = __builtin_amdgcn_mfma_f32_32x32x16_f16(v.h[ks], p.l[ks], o, 0, 0, 0);
            o = __builtin_amdgcn_mfma_f32_32x32x16_f16(v.h[ks], p.h[ks], o, 0, 0, 0);
        }
    };
    const unsigned vlane0 = (unsigned)(size_t)lds + KBUF + ((lane >> 4) & 1) * SVPLANE + (4 * half + ((lane & 15) >> 2)) * 32 + (lane & 3) * 8;

    request(0);
    commit(lds);
    if (ntile > 1) request(1);
    __syncthreads();
    VFrag vf;
    f32x16 sa[2], sb[2];
    PFrag pa, pb;
    const int nfull = T / SKT;                        // full key tiles: the rotated, interleaved schedule; a ragged last tile follows in plain order
    if (nfull > 0) qk(lds, 0, sa);                    // S1 of the first tile (later ones run inside S3 of their predecessor)
    // one full key tile; NEXT: the following tile is full too, its S^T(0) is issued here
    auto full_tile = [&](const int tile, auto next) {
        constexpr bool NEXT = decltype(next)::value;
        const int j0 = tile * SKT;
        char* const cur = lds + ((ABL & 16) ? 0 : (tile & 1) * BUF);
        char* const nxt = lds + ((ABL & 16) ? 0 : ((tile + 1) & 1) * BUF);
        if (tile + 1 < ntile && !(ABL & 16)) {        // that buffer's last readers (previous iteration) are behind the barrier below
            commit(nxt);
            if (tile + 2 < ntile) request(tile + 2);
        }
        const unsigned vlane = vlane0 + ((ABL & 16) ? 0 : (tile & 1) * BUF);
        // S2: S^T of query tile 1 on the matrix pipe, softmax of tile 0 on the vector pipe
        qk(cur, 1, sb);
        softmax_tile<false, ABL>(sa, m0, l0, o0, j0, T, half, 2);
#pragma unroll
        for (int i = 0; i < 12; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x002, 7, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
        if (!(ABL & 32) || tile == 0) read_vfrag(vlane, vf);
        if (!(ABL & 16)) __syncthreads();             // the next tile is visible; nobody reads this tile's LDS image after S2 / the fragment reads
        __builtin_amdgcn_sched_barrier(0);
        // S3: O^T(0) and the next tile's S^T(0) on the matrix pipe; split of P(0) and softmax of tile 1 on the vector pipe
        split_p<ABL>(sa, pa);
        pvmul(vf, pa, o0);
        if (NEXT) qk(nxt, 0, sa);
        softmax_tile<false, ABL>(sb, m1, l1, o1, j0, T, half, 2);
#pragma unroll
        for (int i = 0; i < (NEXT ? 24 : 12); ++i) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 1);
            __builtin_amdgcn_sched_group_barrier(0x002, NEXT ? 7 : 13, 1);
        }
        __builtin_amdgcn_sched_barrier(0);
        // S4: split of P(1), O^T(1)
        split_p<ABL>(sb, pb);
        pvmul(vf, pb, o1);
#pragma unroll
        for (int i = 0; i < 12; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 2);
            __builtin_amdgcn_sched_group_barrier(0x002, 6, 2);
        }
        __builtin_amdgcn_sched_barrier(0);
    };
    for (int tile = 0; tile + 1 < nfull; ++tile) full_tile(tile, std::true_type{});
    if (nfull > 0) full_tile(nfull - 1, std::false_type{});
    if (nfull < ntile) {                              // ragged last tile (committed and made visible by its predecessor's iteration, or above)
        const int j0 = nfull * SKT, nsub = ((T - j0) + 31) / 32;
        char* const cur = lds + (nfull & 1) * BUF;
        qk(cur, 0, sa);
        qk(cur, 1, sb);
        softmax_tile<true>(sa, m0, l0, o0, j0, T, half, nsub);
        softmax_tile<true>(sb, m1, l1, o1, j0, T, half, nsub);
        split_p(sa, pa);
        split_p(sb, pb);
        read_vfrag(vlane0 + (nfull & 1) * BUF, vf);
        pvmul(vf, pa, o0);
        pvmul(vf, pb, o1);
    }
    auto store = [&](const int t, const f32x16& o, float l) {
        l += __shfl_xor(l, 32);
        if (q0 + 32 * t + qi < T) {
            const float inv = 1.0f / l;
            float* dst = out + ((size_t)n * Ta + q0 + 32 * t + qi) * C + h * D + 4 * half;
#pragma unroll
            for (int g = 0; g < 4; ++g)
                *reinterpret_cast<float4*>(dst + 8 * g) = make_float4(o[4 * g] * inv, o[4 * g + 1] * inv, o[4 * g + 2] * inv, o[4 * g + 3] * inv);
        }
    };
    store(0, o0, l0);
    store(1, o1, l1);
}

// geometry this path is built for: head width 32 and enough tokens that the one-off split pass pays (and fills the chip)
int launch_attention(const float* qkv, float* out, int N, int T, int Ta, int C, int heads, int order, hipStream_t s);
bool attention_split_eligible(int T, int C, int heads) { return heads > 0 && C == 32 * heads && T >= 1024; }

size_t attention_split_workspace(int N, int T, int C, int heads) {
    if (!attention_split_eligible(T, C, heads)) return 0;
    return (size_t)N * heads * cdiv(T, SKT) * KV_TILE_BYTES;
}

int launch_attention_split(const float* qkv, float* out, int N, int T, int Ta, int C, int heads, int order, void* ws, size_t ws_bytes, hipStream_t s) {
    CCDM_REQUIRE(attention_split_eligible(T, C, heads), "attention_split: T=%d C=%d heads=%d is not a pre-split geometry (head width 32, T >= 1024)", T, C, heads);
    const size_t need = attention_split_workspace(N, T, C, heads);
    CCDM_REQUIRE(ws && ws_bytes >= need, "attention_split: workspace of %zu bytes, %zu needed (ccdm_attention_workspace_bytes)", ws_bytes, need);
    CCDM_REQUIRE(((size_t)ws & 15) == 0, "attention_split: workspace must be 16-byte aligned");
    const int ntile = cdiv(T, SKT);
    hipLaunchKernelGGL(k_kv_split32, dim3(ntile, heads, N), dim3(256), 0, s, qkv, (char*)ws, T, Ta, C, order);
    CCDM_CHECK_LAUNCH("kv_split32");
    // two query tiles per wave where that still leaves >= 2 blocks per CU's worth of work; else one
    const int mode = exp_env("CCDM_ATTN_SPLIT_MODE");        // A/B hook of CCDM_EXPERIMENTS builds: 1 = <4,1>, 2 = <4,2>, 3 = <8,2>, 4 = <8,1>, 5 / 6 = pipelined 4 / 8 waves
    const long long blocks_p8 = (long long)cdiv(T, 512) * heads * N;        // 8-wave pipelined blocks of 512 queries
    if (mode == 5 || mode == 6 || (mode == 0 && blocks_p8 >= 256)) {
        const int w = mode == 5 ? 4 : 8;
        dim3 grid(cdiv(T, 64 * w), heads, N);
#ifdef CCDM_EXPERIMENTS
        const int abl = exp_env("CCDM_ATTN_ABL");
#define ABL_CASE(A) case A: hipLaunchKernelGGL((k_attention_split32p<8, A>), grid, dim3(512), 0, s, qkv, (const char*)ws, out, T, Ta, C, order, ntile); return 0;
        switch (abl) { ABL_CASE(1) ABL_CASE(2) ABL_CASE(3) ABL_CASE(4) ABL_CASE(8) ABL_CASE(12) ABL_CASE(16) ABL_CASE(32) ABL_CASE(48) ABL_CASE(15) ABL_CASE(51) ABL_CASE(60) default: break; }
#undef ABL_CASE
#endif
        if (w == 4) hipLaunchKernelGGL((k_attention_split32p<4>), grid, dim3(256), 0, s, qkv, (const char*)ws, out, T, Ta, C, order, ntile);
        else hipLaunchKernelGGL((k_attention_split32p<8>), grid, dim3(512), 0, s, qkv, (const char*)ws, out, T, Ta, C, order, ntile);
        CCDM_CHECK_LAUNCH("attention_split32p");
        return 0;
    }
    const long long blocks42 = (long long)cdiv(T, 256) * heads * N;
    int w = 4, qt = blocks42 >= 512 ? 2 : 1;
    if (mode == 1) { w = 4; qt = 1; } else if (mode == 2) { w = 4; qt = 2; } else if (mode == 3) { w = 8; qt = 2; } else if (mode == 4) { w = 8; qt = 1; }
    dim3 grid(cdiv(T, 32 * w * qt), heads, N);
    if (w == 4 && qt == 2) hipLaunchKernelGGL((k_attention_split32<4, 2>), grid, dim3(256), 0, s, qkv, (const char*)ws, out, T, Ta, C, order, ntile);
    else if (w == 4) hipLaunchKernelGGL((k_attention_split32<4, 1>), grid, dim3(256), 0, s, qkv, (const char*)ws, out, T, Ta, C, order, ntile);
    else if (qt == 2) hipLaunchKernelGGL((k_attention_split32<8, 2>), grid, dim3(512), 0, s, qkv, (const char*)ws, out, T, Ta, C, order, ntile);
    else hipLaunchKernelGGL((k_attention_split32<8, 1>), grid, dim3(512), 0, s, qkv, (const char*)ws, out, T, Ta, C, order, ntile);
    CCDM_CHECK_LAUNCH("attention_split32");
    return 0;
}

}  // namespace ccdm

extern "C" size_t ccdm_attention_workspace_bytes(int N, int T, int C, int heads) { return ccdm::attention_split_workspace(N, T, C, heads); }

extern "C" int ccdm_attention_ws(const float* qkv, float* out, int N, int T, int T_alloc, int C, int heads, int order, void* workspace,
                                 size_t workspace_bytes, void* stream) {
    using namespace ccdm;
    CCDM_REQUIRE(qkv && out, "attention_ws: null pointer");
    CCDM_REQUIRE(heads > 0 && C % heads == 0, "attention_ws: C=%d not divisible by heads=%d", C, heads);
    CCDM_REQUIRE(T > 0 && T_alloc >= T, "attention_ws: T=%d, %d rows allocated per sample", T, T_alloc);
    CCDM_REQUIRE(order == 0 || order == 1, "attention_ws: order = %d", order);
    if (workspace && attention_split_eligible(T, C, heads))
        return launch_attention_split(qkv, out, N, T, T_alloc, C, heads, order, workspace, workspace_bytes, (hipStream_t)stream);
    return launch_attention(qkv, out, N, T, T_alloc, C, heads, order, (hipStream_t)stream);
}
