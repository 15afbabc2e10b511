// Self-attention core on the matrix cores: head width 32 (every shipped reference config: num_head_channels: 32), 64 (the DINO
// ViT-S/8 feature encoder, whose token rows may be padded: Ta rows allocated per sample, T of them tokens) and, round 3, ANY head
// width that is a multiple of 4 up to 128 — `create_unet_openai`'s own defaults are num_heads=1, num_head_channels=-1
// (unet_openai/__init__.py:14-15, heads rule unet.py:283-289), i.e. one head as wide as the block (96 / 128 channels at the LIDC
// widths), and num_heads=4 gives 24.  The kernel is instantiated for the padded widths DP in {32, 64, 96, 128}; a narrower head
// (D < DP) stages zeros in the missing columns of K, Q and V (they add nothing to a score, and their output rows are not stored).
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
#include <cstdlib>
#include "ccdm_conv_common.h"

namespace ccdm {

static constexpr int KT = 64;              // keys per tile
template <int D> struct KRow { static constexpr int B = 4 * D + 16; };   // bytes per K-tile row: D hi | D lo halfs | 16 pad (36 / 68 dwords: conflict-free b128)
// V tile in LDS: row-major by key, in planes of 16 d-columns: [plane = d / 16][key 0..63][16 halfs = 32 B], + 128 B between planes so
// that the two planes a 32-lane read group touches sit on opposite halves of the 64 banks.  The PV product needs V^T fragments
// (row = d, 8 consecutive keys per lane): gfx950's ds_read_b64_tr_b16 returns exactly that from the row-major image — within each
// group of 16 lanes, lane l receives column l of the [4 keys][16 d] block whose rows lanes 4j..4j+3 point at (measured:
// tools/ubench/tr_b16_probe.hip) — so V is staged with 8-byte writes like K instead of being transposed with 2-byte writes.
static constexpr int VPLANE = KT * 32 + 128;

#ifndef CCDM_ATTN_STAGE_FLOAT4
#define CCDM_ATTN_STAGE_FLOAT4 0          // 1: stage K/V through HIP float4 structs instead of native vectors (tools/ubench/attn_float4_repro.sh)
#endif
#if CCDM_ATTN_STAGE_FLOAT4
typedef float4 stage_t;
__device__ __forceinline__ float sget(const stage_t& v, int i) { return i == 0 ? v.x : (i == 1 ? v.y : (i == 2 ? v.z : v.w)); }
#else
typedef f32x4 stage_t;
__device__ __forceinline__ float sget(const stage_t& v, int i) { return v[i]; }
#endif

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

// fp16 hi/lo split of 8 floats into MFMA operand fragments (split2_f16: ccdm_conv_common.h)
__device__ __forceinline__ void split8_frag(const float* v, f16x8& hi, f16x8& lo) {
    u32x4 h, l;
    unsigned a, b;
    split2_f16(v[0], v[1], a, b); h[0] = a; l[0] = b;
    split2_f16(v[2], v[3], a, b); h[1] = a; l[1] = b;
    split2_f16(v[4], v[5], a, b); h[2] = a; l[2] = b;
    split2_f16(v[6], v[7], a, b); h[3] = a; l[3] = b;
    hi = __builtin_bit_cast(f16x8, h);
    lo = __builtin_bit_cast(f16x8, l);
}

// D = the padded head width the instantiation is laid out for; Dr = the head's real width (EXACT: Dr == D, the tuned round-2 code path)
template <int WAVES, int D, bool EXACT = true>
__global__ __launch_bounds__(WAVES * 64) void k_attention_mfma(const float* __restrict__ qkv, float* __restrict__ out,
                                                              int T, int Ta, int C, int order, int Dr_) {
    constexpr int NT = WAVES * 64, KROW = KRow<D>::B, DS = D / 16 /* 16-wide k-steps over d */, DM = D / 32 /* 32-row tiles of d */;
    constexpr int NPL = D / 16, VLO = NPL * VPLANE;                // V planes; byte offset of the lo image
    extern __shared__ __attribute__((aligned(16))) char smem_attn[];       // dynamic: the 128-wide layout needs 68.6 KB
    char* const kt = smem_attn;                                             // [KT * KROW]
    char* const vt = smem_attn + KT * KROW;                                 // [2 * NPL * VPLANE]
    const int Dr = EXACT ? D : Dr_;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int h = blockIdx.y, n = blockIdx.z;
    const int q0 = (blockIdx.x * WAVES + wave) * 32;
    const int C3 = 3 * C;
    int qoff, koff, voff;
    if (order == 0) { qoff = h * 3 * Dr; koff = qoff + Dr; voff = qoff + 2 * Dr; }
    else { qoff = h * Dr; koff = C + h * Dr; voff = 2 * C + h * Dr; }
    const float scale = (float)(1.0 / sqrt(sqrt((double)Dr)));
    // scores are kept in units of log2(e): softmax(s) = 2^(s' - max s') / sum with s' = s * log2(e), so every exponential is one
    // v_exp_f32 with no multiply in front of it; the factor rides in the (already scaled) query
    const float qscale = scale * 1.4426950408889634f;
    const float* base = qkv + (size_t)n * Ta * C3;
    const int qi = lane & 31, half = lane >> 5;

    // ---- Q^T fragment (B operand): column = query, k-slot (half, j) = d 16*s + 8*half + j ----
    f16x8 qh[DS], ql[DS];
    {
        const int tq = min(q0 + qi, T - 1);
#pragma unroll
        for (int s = 0; s < DS; ++s) {
            const int d0 = 16 * s + 8 * half;
            // (a narrower head: columns beyond Dr are zeros; the clamped addresses stay inside the row)
            const float* p = base + (size_t)tq * C3 + qoff + (EXACT ? d0 : min(d0, Dr - 4));
            const float* p4 = base + (size_t)tq * C3 + qoff + (EXACT ? d0 + 4 : min(d0 + 4, Dr - 4));
            const float4 a = *reinterpret_cast<const float4*>(p), b = *reinterpret_cast<const float4*>(p4);
            const float sa = EXACT || d0 < Dr ? qscale : 0.f, sb = EXACT || d0 + 4 < Dr ? qscale : 0.f;
            const float v[8] = {a.x * sa, a.y * sa, a.z * sa, a.w * sa, b.x * sb, b.y * sb, b.z * sb, b.w * sb};
            split8_frag(v, qh[s], ql[s]);
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
    stage_t pk[NSTG], pv[NSTG];
    auto request = [&](const int j0) {
#pragma unroll
        for (int u = 0; u < NSTG; ++u) {
            const int item = tid + u * NT;
            const int key = min(j0 + item / (D / 4), T - 1), c4 = EXACT ? item % (D / 4) : min(item % (D / 4), (Dr >> 2) - 1);
            const float* p = base + (size_t)key * C3;
            pk[u] = *reinterpret_cast<const stage_t*>(p + koff + 4 * c4);
            pv[u] = *reinterpret_cast<const stage_t*>(p + voff + 4 * c4);
        }
    };
    // per-lane part of the V^T fragment address (ds_read_b64_tr_b16): group-local lane gl = lane & 15 points at row (gl >> 2) of the
    // group's [4 keys][16 d] block, 4 halfs from column 4*(gl & 3); the group's d-plane is (lane >> 4) & 1, its key offset 4*half
    const unsigned vlane = (unsigned)(size_t)vt + ((lane >> 4) & 1) * VPLANE + (4 * half + ((lane & 15) >> 2)) * 32 + (lane & 3) * 8;
    request(0);
    for (int j0 = 0; j0 < T; j0 += KT) {
        __syncthreads();
        // ---- stage K (scaled) and V row-major as fp16 hi/lo ----
#pragma unroll
        for (int u = 0; u < NSTG; ++u) {
            const int item = tid + u * NT;
            const int key = item / (D / 4), c4 = item % (D / 4);
            const bool in = j0 + key < T && (EXACT || 4 * c4 < Dr);
            const float ks = in ? scale : 0.f, vs = in ? 1.f : 0.f;        // rows beyond T (and columns beyond a narrower head) are staged as zeros
            u32x2 hi, lo;
            unsigned a, b;
            split2_f16(sget(pk[u], 0) * ks, sget(pk[u], 1) * ks, a, b); hi[0] = a; lo[0] = b;
            split2_f16(sget(pk[u], 2) * ks, sget(pk[u], 3) * ks, a, b); hi[1] = a; lo[1] = b;
            *reinterpret_cast<u32x2*>(kt + key * KROW + 8 * c4) = hi;
            *reinterpret_cast<u32x2*>(kt + key * KROW + 2 * D + 8 * c4) = lo;
            split2_f16(sget(pv[u], 0) * vs, sget(pv[u], 1) * vs, a, b); hi[0] = a; lo[0] = b;
            split2_f16(sget(pv[u], 2) * vs, sget(pv[u], 3) * vs, a, b); hi[1] = a; lo[1] = b;
            char* vd = vt + (c4 >> 2) * VPLANE + key * 32 + (c4 & 3) * 8;
            *reinterpret_cast<u32x2*>(vd) = hi;
            *reinterpret_cast<u32x2*>(vd + VLO) = lo;
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
                // acc[r] = score(query = lane&31, key = 32*st + (r&3) + 8*(r>>2) + 4*half); only the sample's last tile can hold keys beyond T
                if (j0 + KT > T) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int key = j0 + 32 * st + (r & 3) + 8 * (r >> 2) + 4 * half;
                        acc[r] = key < T ? acc[r] : -INFINITY;
                    }
                }
#pragma unroll
                for (int r = 0; r < 16; ++r) mx = fmaxf(mx, acc[r]);
                sc[st] = acc;
            }
        }
        mx = fmaxf(mx, __shfl_xor(mx, 32));
        if (__builtin_amdgcn_ballot_w64(mx > m) != 0ull) {     // some query's running maximum grew: rescale (else the factor is exactly 1)
            const float corr = __builtin_amdgcn_exp2f(m - mx);     // first tile: 2^(-inf) = 0
            m = mx;
            l *= corr;
#pragma unroll
            for (int mt = 0; mt < DM; ++mt)
#pragma unroll
                for (int r = 0; r < 16; ++r) o[mt][r] *= corr;
        }
#pragma unroll
        for (int st = 0; st < 2; ++st) {
            if (st < nsub) {
                float p[16];
#pragma unroll
                for (int r = 0; r < 16; ++r) { p[r] = __builtin_amdgcn_exp2f(sc[st][r] - mx); l += p[r]; }
                // O^T += V^T * P^T.  k-step s covers this lane's registers r = 8s..8s+7, i.e. keys
                // 32*st + 16*s + {0..3, 8..11} + 4*half — the same keys are read (transposed) from the V image for the A operand.
                // (Issuing a whole sub-tile's fragment reads ahead of the exponentials and waiting behind them was tried: +1 % at
                //  T = 8192 and a wrong result in the head-width-64 instantiation — a register the compiler moved between the two
                //  asm statements — so each group of reads keeps its wait in the same statement.)
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    f16x8 ph, pl;
                    split8_frag(p + 8 * s, ph, pl);
#pragma unroll
                    for (int mt = 0; mt < DM; ++mt) {
                        // lane (d = 32*mt + lane&31, half): hi keys +0..3, +8..11, then the same of the lo image
                        f16x4 vh0, vh1, vl0, vl1;
                        const unsigned va = vlane + (32 * st + 16 * s) * 32 + 2 * mt * VPLANE;
                        asm volatile("ds_read_b64_tr_b16 %0, %4\n\t"
                                     "ds_read_b64_tr_b16 %1, %4 offset:256\n\t"
                                     "ds_read_b64_tr_b16 %2, %4 offset:%5\n\t"
                                     "ds_read_b64_tr_b16 %3, %4 offset:%6\n\t"
                                     "s_waitcnt lgkmcnt(0)"
                                     : "=&v"(vh0), "=&v"(vh1), "=&v"(vl0), "=&v"(vl1)
                                     : "v"(va), "i"(VLO), "i"(VLO + 256)
                                     : "memory");
                        f16x8 vh, vl;
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
        float* dst = out + ((size_t)n * Ta + q0 + qi) * C + h * Dr + 4 * half;
#pragma unroll
        for (int mt = 0; mt < DM; ++mt)
#pragma unroll
            for (int g = 0; g < 4; ++g)
                if (EXACT || 32 * mt + 8 * g + 4 * half < Dr)
                    *reinterpret_cast<float4*>(dst + 32 * mt + 8 * g) =
                        make_float4(o[mt][4 * g] * inv, o[mt][4 * g + 1] * inv, o[mt][4 * g + 2] * inv, o[mt][4 * g + 3] * inv);
    }
}

template <int WAVES, int DP, bool EXACT>
static void launch_attn_inst(dim3 grid, hipStream_t s, const float* qkv, float* out, int T, int Ta, int C, int order, int D) {
    constexpr size_t lds = (size_t)KT * KRow<DP>::B + (size_t)2 * (DP / 16) * VPLANE;
    hipLaunchKernelGGL((k_attention_mfma<WAVES, DP, EXACT>), grid, dim3(WAVES * 64), lds, s, qkv, out, T, Ta, C, order, D);
}

// padded head width the MFMA kernel runs a head of width D on (0: none — D not a multiple of 4, or beyond 128)
int attention_mfma_width(int D) {
    if (D <= 0 || D % 4 || D > 128) return 0;
    return D <= 32 ? 32 : (D <= 64 ? 64 : (D <= 96 ? 96 : 128));
}

int launch_attention_mfma(const float* qkv, float* out, int N, int T, int Ta, int C, int heads, int order, hipStream_t s) {
    const int D = C / heads;
    const int DP = attention_mfma_width(D);
    CCDM_REQUIRE(DP > 0, "attention_mfma: head width %d (C=%d, heads=%d) is not a multiple of 4 up to 128", D, C, heads);
    // 8 waves (256 queries) per block for long sequences at head width 32: a key tile's staging (split to fp16 hi/lo, LDS writes) is
    // shared by twice as many queries — one item per thread instead of two
    // (T = 8192, N = 4, 4 heads: 570 -> 502 us = 29.0 -> 32.8 % of the fp16 matrix peak by instruction count; T = 2048: 90 -> 83 us)
    const int w8_env = exp_env("CCDM_ATTN_W8_MIN");      // A/B hook of CCDM_EXPERIMENTS builds (-1: never)
    const int w8_min = w8_env ? w8_env : 2048;
    // ... as long as the 8-wave grid still covers the chip: at T = 2048 with 4 samples x 4 heads it is 128 blocks on 256 CUs, and 256
    // 4-wave blocks run the same launch in 52 instead of 63 us.  (Every wave owns its queries and walks the same key tiles in the same
    // order: the block shape changes no result bit, so the rule may look at N.)
    const bool w8 = DP == 32 && w8_min > 0 && T >= w8_min && (long long)cdiv(T, 256) * heads * N >= 256;
    int waves = w8 ? 8 : (T >= 128 ? 4 : (T >= 64 ? 2 : 1));
    // wide heads: the staged K/V items and the query fragments of a wave grow with the width — keep them inside the register file by
    // spreading a key tile over more threads (96: at least 4 waves; 128: always 8)
    if (DP == 96 && waves < 4) waves = 4;
    if (DP == 128) waves = 8;
    dim3 grid(cdiv(T, 32 * waves), heads, N);
    if (DP == 128) {
        if (D == 128) launch_attn_inst<8, 128, true>(grid, s, qkv, out, T, Ta, C, order, D);
        else launch_attn_inst<8, 128, false>(grid, s, qkv, out, T, Ta, C, order, D);
    } else if (DP == 96) {
        if (waves == 8) { if (D == 96) launch_attn_inst<8, 96, true>(grid, s, qkv, out, T, Ta, C, order, D); else launch_attn_inst<8, 96, false>(grid, s, qkv, out, T, Ta, C, order, D); }
        else { if (D == 96) launch_attn_inst<4, 96, true>(grid, s, qkv, out, T, Ta, C, order, D); else launch_attn_inst<4, 96, false>(grid, s, qkv, out, T, Ta, C, order, D); }
    } else if (DP == 64) {
        if (D == 64) {
            if (waves == 4) launch_attn_inst<4, 64, true>(grid, s, qkv, out, T, Ta, C, order, D);
            else if (waves == 2) launch_attn_inst<2, 64, true>(grid, s, qkv, out, T, Ta, C, order, D);
            else launch_attn_inst<1, 64, true>(grid, s, qkv, out, T, Ta, C, order, D);
        } else {
            if (waves < 2) { waves = 2; grid = dim3(cdiv(T, 64), heads, N); }
            if (waves == 4) launch_attn_inst<4, 64, false>(grid, s, qkv, out, T, Ta, C, order, D);
            else launch_attn_inst<2, 64, false>(grid, s, qkv, out, T, Ta, C, order, D);
        }
    } else {
        if (D == 32) {
            if (waves == 8) launch_attn_inst<8, 32, true>(grid, s, qkv, out, T, Ta, C, order, D);
            else if (waves == 4) launch_attn_inst<4, 32, true>(grid, s, qkv, out, T, Ta, C, order, D);
            else if (waves == 2) launch_attn_inst<2, 32, true>(grid, s, qkv, out, T, Ta, C, order, D);
            else launch_attn_inst<1, 32, true>(grid, s, qkv, out, T, Ta, C, order, D);
        } else {
            if (waves > 4) waves = 4;
            grid = dim3(cdiv(T, 32 * waves), heads, N);
            if (waves == 4) launch_attn_inst<4, 32, false>(grid, s, qkv, out, T, Ta, C, order, D);
            else if (waves == 2) launch_attn_inst<2, 32, false>(grid, s, qkv, out, T, Ta, C, order, D);
            else launch_attn_inst<1, 32, false>(grid, s, qkv, out, T, Ta, C, order, D);
        }
    }
    CCDM_CHECK_LAUNCH("attention_mfma");
    return 0;
}

}  // namespace ccdm
