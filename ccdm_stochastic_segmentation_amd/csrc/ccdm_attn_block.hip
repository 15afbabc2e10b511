// First half of an AttentionBlock in ONE launch, for the low-resolution stages (T <= 256 tokens):
//
//     a = attention( qkv( GroupNorm32(x) ) )                                       unet.py:291-311 (_forward), :343-360 (legacy core)
//
// One workgroup per (sample, head); the 3C-wide qkv tensor of the three-launch form (qkv conv -> attention core -> proj conv)
// never exists in memory, and neither does a separate GroupNorm pass: the block reads x once, normalises it on load (affine from
// the producer's per-channel partial sums), multiplies it by its head's 96 rows of the qkv weight on the matrix cores, and runs
// softmax(q k^T) v for its head out of registers and LDS.  proj_out + residual (+ statistics) stay the 1x1 conv that follows.
// (A whole-block-per-sample variant that also folded proj_out in was built first and measured: 60 us at T=256 against 47 us for
//  the three launches — 64 workgroups serialise on 64 CUs what 192 head-blocks spread over the chip.)
//
// Wave mt owns the 32-token tile mt of the sample; T/32 waves per block.
//   1. qkv GEMM for this head.  The wave's x tile [32 tokens][C] is loaded once into registers, normalised, split into fp16 hi/lo and
//      used as the B operand (columns = tokens) for Q^T and K^T = W x^T, and as the A operand (rows = tokens) for V = x W^T: the two
//      fragment layouts hold the same numbers, and so do the packed weight fragments (ccdm_pack_conv_weight, lane = (cout, k-group)).
//      The head's weight fragments go through LDS in two chunks of C/32 k-steps (double-buffered, one barrier each).
//      Q^T leaves the GEMM in exactly the layout the score MFMA wants as its B operand (column = query, k-slots = the d values
//      the accumulator rows map to); K is written to LDS row-major [key][d], V^T as [d][key] from the token-major V tile
//      (8-byte writes of 4 consecutive keys).
//   2. attention, streaming over pairs of 32-key tiles, S^T = K Q^T and O^T = V^T P^T as in ccdm_attention.hip: the query is
//      the MFMA column = lane & 31, so softmax statistics are per-lane scalars and P never leaves its registers.
// Every product is the 3-term fp16 hi/lo split (lo*hi + hi*lo + hi*hi, fp32 accumulate) of the conv and attention kernels.
#include "ccdm_common.h"
#include "ccdm_conv_common.h"

namespace ccdm {

struct AttnBlockK {
    ccdm_attn_block_args a;
    const float* wsq;     // [3C] power-of-two un-scale of the packed qkv weights (incl. 1/ACT_PRESCALE)
};

__device__ __forceinline__ void split8(const float* v, f16x8& hi, f16x8& lo) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        hi[j] = (_Float16)v[j];
        lo[j] = (_Float16)(v[j] - (float)hi[j]);
    }
}
__device__ __forceinline__ void split4v(const float a, const float b, const float c, const float d, f16x4& hi, f16x4& lo) {
    hi[0] = (_Float16)a; hi[1] = (_Float16)b; hi[2] = (_Float16)c; hi[3] = (_Float16)d;
    lo[0] = (_Float16)(a - (float)hi[0]); lo[1] = (_Float16)(b - (float)hi[1]);
    lo[2] = (_Float16)(c - (float)hi[2]); lo[3] = (_Float16)(d - (float)hi[3]);
}
// three-term split product: acc += A * B with A = ah + al, B = bh + bl (lo*lo dropped)
__device__ __forceinline__ f32x16 mfma3(const f16x8 ah, const f16x8 al, const f16x8 bh, const f16x8 bl, f32x16 acc) {
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc, 0, 0, 0);
    return acc;
}
__device__ __forceinline__ f16x8 cat4(const f16x4 a, const f16x4 b) {
    f16x8 r;
#pragma unroll
    for (int j = 0; j < 4; ++j) { r[j] = a[j]; r[4 + j] = b[j]; }
    return r;
}


template <int T, int C>
struct QkvAttnGeo {
    static constexpr int MT = T / 32, NT = MT * 64, KS = C / 16, KSC = KS / 2;   // k-steps per weight chunk
    static constexpr int FRAG = 3 * 2048;                  // one k-step of this head's packed fragments: [q|k|v tile][hi|lo][64 lanes][16 B]
    static constexpr int CHUNK = KSC * FRAG;
    static constexpr int KROW = 136;                       // bytes per K row: 32 hi | 32 lo halfs | 8 pad (34 dwords: conflict-free b64 reads)
    static constexpr int VROW = 4 * T + 8;                 // bytes per V^T row: T hi | T lo halfs | 8 pad
    static constexpr int KV = T * KROW + 32 * VROW;
    static constexpr int TABLES = C * 8 + 2 * 96 * 4;      // GroupNorm (scale, shift) + this head's bias / un-scale
    static constexpr int LDS = TABLES + 2 * CHUNK + KV;
    static_assert(C % 32 == 0 && T % 64 == 0 && MT <= 16, "geometry");
    static_assert(TABLES % 16 == 0, "alignment");
};

template <int T, int C>
__global__ __launch_bounds__((T / 32) * 64) void k_qkv_attention(const AttnBlockK k) {
    using Geo = QkvAttnGeo<T, C>;
    constexpr int MT = Geo::MT, NT = Geo::NT, KS = Geo::KS, KSC = Geo::KSC, FRAG = Geo::FRAG, CHUNK = Geo::CHUNK;
    constexpr int KROW = Geo::KROW, VROW = Geo::VROW;
    const ccdm_attn_block_args& a = k.a;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float2* ab = reinterpret_cast<float2*>(smem);                 // [C]  GroupNorm (scale, shift), pre-multiplied by ACT_PRESCALE
    float* tb = reinterpret_cast<float*>(smem + C * 8);           // bias[96] | un-scale[96] of this head's q|k|v channels
    char* wst = smem + Geo::TABLES;                               // weight stage: 2 chunks
    char* kbase = wst + 2 * CHUNK;                                // K rows, then V^T rows
    char* vbase = kbase + T * KROW;

    const int tid = threadIdx.x, lane = tid & 63, qi = lane & 31, half = lane >> 5;
    const int mt = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n = blockIdx.x, hd = blockIdx.y;
    const int NQ = 3 * a.heads;                                   // n-tiles per k-step of the packed qkv weights
    const float* xn = a.x + (size_t)n * T * C;

    // ---- x tile of this wave: [32 tokens][C], lane (token, half) holds channels 16*ks + 8*half + j ----
    f32x4 xr[KS][2];
    {
        const float* p = xn + (size_t)(mt * 32 + qi) * C + 8 * half;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            xr[ks][0] = *reinterpret_cast<const f32x4*>(p + 16 * ks);
            xr[ks][1] = *reinterpret_cast<const f32x4*>(p + 16 * ks + 4);
        }
    }
    // ---- weight chunk -> registers -> LDS (the head's 3 tiles of each k-step are contiguous in the packed array) ----
    constexpr int WITEMS = (CHUNK / 16 + NT - 1) / NT;
    f32x4 wreg[WITEMS];
    auto wload = [&](const int chunk) {
#pragma unroll
        for (int i = 0; i < WITEMS; ++i) {
            const int it = min(tid + i * NT, CHUNK / 16 - 1);
            const int ksl = it / (FRAG / 16), rem = it % (FRAG / 16);
            const char* src = reinterpret_cast<const char*>(a.wqkv) + ((size_t)(chunk * KSC + ksl) * NQ + 3 * hd) * 2048;
            wreg[i] = *reinterpret_cast<const f32x4*>(src + (size_t)rem * 16);
        }
    };
    auto wstore = [&](const int buf) {
        f32x4* dst = reinterpret_cast<f32x4*>(wst + buf * CHUNK);
#pragma unroll
        for (int i = 0; i < WITEMS; ++i) {
            const int it = tid + i * NT;
            if ((i + 1) * NT <= CHUNK / 16 || it < CHUNK / 16) dst[it] = wreg[i];
        }
    };
    wload(0);
    // per-thread operands of the tables below, requested now: fetched where they are consumed (behind the statistics barrier) they were a
    // second full memory round trip in a kernel whose time is one dependent chain
    const int c_own = min(tid, C - 1), i_own = min(tid, 95);
    float pf_gamma = a.gamma[c_own], pf_beta = a.beta[c_own];
    float pf_bias = a.bqkv[96 * hd + i_own], pf_wsq = k.wsq[96 * hd + i_own];
    // ---- tables: GroupNorm affine from the producer's per-channel partial sums; this head's biases and weight un-scales ----
    {
        // per-channel sums over the slices (16 independent loads, not one round trip per partial), exchanged through LDS (the K-row
        // region is free until the qkv GEMM has finished), then added over the group's channels — all in ascending order
        constexpr int cpg = C / 32;
        f64x2* scratch = reinterpret_cast<f64x2*>(kbase);
        const int S = a.slices;
        for (int c = tid; c < C; c += NT) {
            const char* base = reinterpret_cast<const char*>(a.stats + ((size_t)n * S * C + c) * 2);
            const unsigned stride = (unsigned)C * 16u, last = (unsigned)(S - 1) * stride;
            f64x2 v[16];
            // (few-pixel producers leave 1-4 slices: 12 of 16 clamped requests would be duplicates — 1 KB per wave each through the vector
            //  memory front end, ahead of everything else in this kernel's chain; the slice count is uniform: one scalar branch)
#pragma unroll
            for (int u = 0; u < 4; ++u) v[u] = *reinterpret_cast<const f64x2*>(base + min((unsigned)u * stride, last));
            if (S > 4) {
#pragma unroll
                for (int u = 4; u < 16; ++u) v[u] = *reinterpret_cast<const f64x2*>(base + min((unsigned)u * stride, last));
            } else {
#pragma unroll
                for (int u = 4; u < 16; ++u) v[u] = f64x2{0.0, 0.0};
            }
            f64x2 own = {0.0, 0.0};
#pragma unroll
            for (int u = 0; u < 16; ++u) { own[0] += u < S ? v[u][0] : 0.0; own[1] += u < S ? v[u][1] : 0.0; }
            for (int s = 16; s < S; ++s) own += *reinterpret_cast<const f64x2*>(base + (unsigned)s * stride);
            scratch[c] = own;
        }
        __syncthreads();
        for (int c = tid; c < C; c += NT) {
            const int c_lo = (c / cpg) * cpg;
            f64x2 g = {0.0, 0.0};
#pragma unroll
            for (int j = 0; j < cpg; ++j) g += scratch[c_lo + j];
            float meanf, rstd;
            gn_mean_rstd(g[0], g[1], (double)cpg * (double)T, a.eps, meanf, rstd);
            const float sc = rstd * (c == tid ? pf_gamma : a.gamma[c]);
            const float sh = (c == tid ? pf_beta : a.beta[c]) - sc * meanf;
            ab[c] = make_float2(sc * ACT_PRESCALE, sh * ACT_PRESCALE);       // activation pre-scale 2^4, undone through wsq
        }
        for (int i = tid; i < 96; i += NT) { tb[i] = i == tid ? pf_bias : a.bqkv[96 * hd + i]; tb[96 + i] = i == tid ? pf_wsq : k.wsq[96 * hd + i]; }
    }
    wstore(0);
    wload(1);
    __syncthreads();

    // =========================================== 1. qkv GEMM ===========================================
    // acc[0] = Q^T tile (row = d, col = token); acc[1] = K^T (same); acc[2] = V (row = token, col = d)
    f32x16 acc[3];
#pragma unroll
    for (int j = 0; j < 3; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
        if (ks == KSC) { wstore(1); __syncthreads(); }
        f16x8 xh, xl;
        {
            float v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float2 t = ab[16 * ks + 8 * half + j];
                v[j] = fmaf(xr[ks][j >> 2][j & 3], t.x, t.y);
            }
            split8(v, xh, xl);
        }
        const f16x8* wb = reinterpret_cast<const f16x8*>(wst + (ks / KSC) * CHUNK + (ks % KSC) * FRAG) + lane;
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const f16x8 wh = wb[(j * 2) * 64], wl = wb[(j * 2 + 1) * 64];
            if (j < 2) acc[j] = mfma3(wh, wl, xh, xl, acc[j]);      // W (rows = cout) x X^T (cols = tokens)
            else acc[j] = mfma3(xh, xl, wh, wl, acc[j]);            // X (rows = tokens) x W^T (cols = cout)
        }
    }

    // ---- bias, un-scale, attention scale; K -> LDS rows, V^T -> LDS rows, Q^T -> B-operand fragments ----
    const float qscale = 0.42044820762685725f;          // 32^-1/4, applied to q and to k (unet.py:354-357)
    f16x8 qh[2], ql[2];
    {
        const float* bq = tb;                           // channel = {q,k,v}*32 + d of this head
        const float* ws = tb + 96;
        float qv[16];
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
            const int d0 = 8 * g4 + 4 * half;           // this lane's rows (r & 3) + 8*(r >> 2) + 4*half of the Q^T / K^T tiles
            const f32x4 b_q = *reinterpret_cast<const f32x4*>(bq + d0), s_q = *reinterpret_cast<const f32x4*>(ws + d0);
            const f32x4 b_k = *reinterpret_cast<const f32x4*>(bq + 32 + d0), s_k = *reinterpret_cast<const f32x4*>(ws + 32 + d0);
            float kv[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                qv[4 * g4 + e] = fmaf(acc[0][4 * g4 + e], s_q[e], b_q[e]) * qscale;
                kv[e] = fmaf(acc[1][4 * g4 + e], s_k[e], b_k[e]) * qscale;
            }
            f16x4 hi, lo;
            split4v(kv[0], kv[1], kv[2], kv[3], hi, lo);
            char* kr = kbase + (mt * 32 + qi) * KROW + 2 * d0;
            *reinterpret_cast<f16x4*>(kr) = hi;
            *reinterpret_cast<f16x4*>(kr + 64) = lo;
        }
        split8(qv, qh[0], ql[0]);
        split8(qv + 8, qh[1], ql[1]);
        // V tile: row = token (r & 3) + 8*(r >> 2) + 4*half, col = d = lane & 31
        const float b_v = bq[64 + qi], s_v = ws[64 + qi];
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
            f16x4 hi, lo;
            split4v(fmaf(acc[2][4 * g4], s_v, b_v), fmaf(acc[2][4 * g4 + 1], s_v, b_v),
                    fmaf(acc[2][4 * g4 + 2], s_v, b_v), fmaf(acc[2][4 * g4 + 3], s_v, b_v), hi, lo);
            char* vr = vbase + qi * VROW + 2 * (mt * 32 + 8 * g4 + 4 * half);
            *reinterpret_cast<f16x4*>(vr) = hi;
            *reinterpret_cast<f16x4*>(vr + 2 * T) = lo;
        }
    }
    __syncthreads();

    // =========================================== 2. attention ===========================================
    f32x16 oo;
#pragma unroll
    for (int r = 0; r < 16; ++r) oo[r] = 0.f;
    float m = -INFINITY, l = 0.f;
    for (int kt = 0; kt < MT; kt += 2) {
        // S^T tiles: row = key kt*32 + 32*u + (r&3) + 8*(r>>2) + 4*half, col = query.  A = K rows with the d-permutation of qh/ql:
        // k-step s, slot (half, j) = d 16*s + (j&3) + 8*(j>>2) + 4*half.  Two key tiles per round: their MFMA chains are independent.
        f32x16 sc[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
#pragma unroll
            for (int r = 0; r < 16; ++r) sc[u][r] = 0.f;
        }
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const char* kr = kbase + ((kt + u) * 32 + qi) * KROW + 8 * half + 32 * s;
                const f16x8 kh = cat4(*reinterpret_cast<const f16x4*>(kr), *reinterpret_cast<const f16x4*>(kr + 16));
                const f16x8 kl = cat4(*reinterpret_cast<const f16x4*>(kr + 64), *reinterpret_cast<const f16x4*>(kr + 64 + 16));
                sc[u] = mfma3(kh, kl, qh[s], ql[s], sc[u]);
            }
        float mx = m;
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int r = 0; r < 16; ++r) mx = fmaxf(mx, sc[u][r]);
        mx = fmaxf(mx, __shfl_xor(mx, 32));
        const float corr = __expf(m - mx);            // first round: exp(-inf) = 0
        m = mx;
        l *= corr;
#pragma unroll
        for (int r = 0; r < 16; ++r) oo[r] *= corr;
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            float p[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) { p[r] = __expf(sc[u][r] - mx); l += p[r]; }
            // O^T += V^T P^T: k-step s covers this lane's registers 8s..8s+7 = keys (kt+u)*32 + 16*s + {0..3, 8..11} + 4*half
            const char* vr = vbase + qi * VROW + 2 * ((kt + u) * 32 + 4 * half);
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                f16x8 ph, pl;
                split8(p + 8 * s, ph, pl);
                const f16x8 vh = cat4(*reinterpret_cast<const f16x4*>(vr + 32 * s), *reinterpret_cast<const f16x4*>(vr + 32 * s + 16));
                const f16x8 vl = cat4(*reinterpret_cast<const f16x4*>(vr + 2 * T + 32 * s), *reinterpret_cast<const f16x4*>(vr + 2 * T + 32 * s + 16));
                oo = mfma3(vh, vl, ph, pl, oo);
            }
        }
    }
    l += __shfl_xor(l, 32);
    const float inv = 1.0f / l;
    // oo[r] = O[query = lane&31][d = (r&3) + 8*(r>>2) + 4*half]: four float4 rows of 4 consecutive d each
    float* dst = a.out + ((size_t)n * T + mt * 32 + qi) * C + hd * 32 + 4 * half;
#pragma unroll
    for (int g4 = 0; g4 < 4; ++g4) {
        f32x4 v;
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = oo[4 * g4 + e] * inv;
        *reinterpret_cast<f32x4*>(dst + 8 * g4) = v;
    }
}

template <int T, int C>
static int launch_one(const AttnBlockK& k, hipStream_t s) {
    using Geo = QkvAttnGeo<T, C>;
    static_assert(Geo::LDS <= 160 * 1024, "LDS budget");
    auto kern = k_qkv_attention<T, C>;
    static bool configured = false;
    if (!configured) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, Geo::LDS) != hipSuccess)
            return fail("norm_qkv_attention: cannot reserve %d bytes of LDS", Geo::LDS);
        configured = true;
    }
    hipLaunchKernelGGL(kern, dim3(k.a.N, k.a.heads), dim3(Geo::NT), Geo::LDS, s, k);
    return 0;
}

bool attn_block_supported(int T, int C, int heads) {
    if (C != heads * 32) return false;
    return (T == 256 && (C == 96 || C == 128)) || (T == 64 && (C == 96 || C == 128)) || (T == 128 && (C == 128 || C == 256));
}

int launch_attn_block(const ccdm_attn_block_args& a, hipStream_t s) {
    CCDM_REQUIRE(a.x && a.out && a.stats && a.gamma && a.beta && a.wqkv && a.bqkv, "norm_qkv_attention: null pointer");
    CCDM_REQUIRE(a.N > 0 && a.slices >= 1, "norm_qkv_attention: N=%d slices=%d", a.N, a.slices);
    {
        int ntiles, NI;
        conv_ntiles(3 * a.C, &ntiles, &NI);
        CCDM_REQUIRE(ntiles == 3 * a.heads, "norm_qkv_attention: packed qkv weights have %d n-tiles, expected %d", ntiles, 3 * a.heads);
    }
    CCDM_REQUIRE(attn_block_supported(a.T, a.C, a.heads), "norm_qkv_attention: (T=%d, C=%d, heads=%d) is not built; use ccdm_conv2d (GN + qkv) + ccdm_attention",
                 a.T, a.C, a.heads);
    AttnBlockK k;
    k.a = a;
    // the scale table follows the packed fragments (ccdm_pack_conv_weight_ex): fragments = (Cin/16) * ntiles * 2 * 1024 bytes
    k.wsq = reinterpret_cast<const float*>(static_cast<const char*>(a.wqkv) + (size_t)(a.C / 16) * (3 * a.C / 32) * 2048);
    int rc;
    if (a.T == 256 && a.C == 96) rc = launch_one<256, 96>(k, s);
    else if (a.T == 256) rc = launch_one<256, 128>(k, s);
    else if (a.T == 64 && a.C == 96) rc = launch_one<64, 96>(k, s);
    else if (a.T == 64) rc = launch_one<64, 128>(k, s);
    else if (a.C == 128) rc = launch_one<128, 128>(k, s);
    else rc = launch_one<128, 256>(k, s);
    if (rc) return rc;
    CCDM_CHECK_LAUNCH("norm_qkv_attention");
    return 0;
}

}  // namespace ccdm

extern "C" int ccdm_norm_qkv_attention_supported(int T, int C, int heads) { return ccdm::attn_block_supported(T, C, heads) ? 1 : 0; }

extern "C" int ccdm_norm_qkv_attention(const ccdm_attn_block_args* a, void* stream) {
    if (!a) return ccdm::fail("ccdm_norm_qkv_attention: null args");
    return ccdm::launch_attn_block(*a, (hipStream_t)stream);
}
