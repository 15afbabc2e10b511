// Upsample (nearest x2) + conv 3x3 in sub-pixel form, F16X3 — built for the low-resolution decoder levels, taken (upconv_eligible) by EVERY
// Upsample conv of 32 / 64 / 96 / 128 input channels whose input tiles exactly (whole 8x16 tiles, or 8x8 tiles for 8-pixel-wide inputs),
// whatever its size (tested up to 128x256 inputs):
//
//     out(2y+dy, 2x+dx) = sum_{a,b in {0,1}} W'[dy,dx][a,b] . in(y+dy-1+a, x+dx-1+b)            unet.py:106-116, include/ccdm_hip.h `up = 2`
//
// (weights: ccdm_pack_upconv_weight — four 2x2 convs whose kernels are fp64 sums of the 3x3 taps that fall on one input pixel, packed as
// a 2x2 conv with 4*Cout output channels, n-tile = 4 * channel tile + phase).
//
// The general kernel's form of this layer (k_conv<.., NI = 4, UP2>) walks 16-channel chunks, each with the chunk's four phases of weight
// fragments staged through LDS (32 KB per chunk, by LDS-DMA behind barrier A): its block timeline (tools/timeline_op.py, 64 -> 64 at
// 32x32 -> 64x64) is 4 chunks x (3 200-4 500 cycles of commit = waiting for that DMA, + two barriers) around 48 matrix instructions per
// wave, 57 000 cycles per block in 1.33 rounds: 52.6 us for a layer whose bytes take 13 us and whose matrix work takes 12.
// Here:
//   * WAVE = PHASE.  Wave p of the block's four computes phase (dy, dx) = (p >> 1, p & 1) for ALL four 32-pixel sub-tiles of the 8x16
//     input tile, so a weight fragment is needed by exactly one wave, which reads it straight from L2 into the registers the MFMA takes
//     it from (two requests ahead): no LDS staging of B, no barrier for it, 12 matrix instructions per 2 KB fragment pair.
//   * the halo tile (10 x 18 pixels) is staged ONCE per tile with ALL input channels (raw input: x 2^4, fp16 hi/lo split, pitch 4 C + 16
//     bytes: conflict-free 16-byte fragment reads for C in {32, 64, 96, 128}): one barrier in front of the matrix phase;
//   * a block may walk several 32-channel output tiles from one staged tile (ctb — chosen from the BLOCK COUNT, i.e. from N too: a
//     partitioning choice only, every output element is computed by the same instruction sequence either way);
//   * C = 128 without aliasing the epilogue buffer onto the tile needs ~113 KB of LDS: one block per CU there.
// Same products in the same order as the general kernel's form (k-step outer, window tap inner; lo*hi, hi*lo, hi*hi): identical OUTPUTS
// (tested bit for bit through CCDM_DIAG_GENERAL_KERNEL).  The output statistics are one partial per (sample, slice) like the general
// kernel's with the four phases folded in phase order, but accumulated per lane in another order: equal to fp32 rounding of the lane
// sums (tested on the slice-summed partials at rtol 1e-6), not bit for bit.
#include "ccdm_common.h"
#include "ccdm_conv_common.h"

#include <algorithm>

namespace ccdm {

struct UpK {
    const float* in;        // [N, Hin, Win, C]
    const void* w;          // ccdm_pack_upconv_weight
    const float* wscale;    // [ntiles * 32]
    const float* bias;      // [Cout] or NULL
    float* out;             // [N, 2 Hin, 2 Win, Cout]
    double* out_stats;      // [N, slices, Cout, 2] or NULL
    int N, Hin, Win, Cout, ntiles, slices, tiles_x, tiles_y, ctb;
};

constexpr int UP_TH = 8, UP_HH = UP_TH + 2;
constexpr int UP_EPS = 36;                                     // floats per pixel row of the transpose buffer
constexpr int UP_EPI_BYTES = 4 * 32 * UP_EPS * 4;              // four wave-private [32 pixels][36] buffers
constexpr int UP_CTB_MAX = 2;
#ifndef CCDM_UP_BDEPTH
#define CCDM_UP_BDEPTH 2
#endif
constexpr int UP_BD = CCDM_UP_BDEPTH;                         // weight fragments requested ahead of the one being multiplied

// TW = 16: 8x16 input tiles, four 32-pixel sub-tiles per wave; TW = 8 (8-pixel-wide inputs: LIDC's 8x8 -> 16x16 launch): 8x8 tiles, two
// sub-tiles per wave, and — like the general kernel's one-phase-per-block form there — one statistics partial per (slice, PHASE)
template <int C, int TW> struct UpGeo {
    static constexpr int PIXB = 4 * C + 16, QPP = C / 4, KS16 = C / 16, NF = KS16 * 4;
    static constexpr int HW = TW + 2, HP = UP_HH * HW, MT = UP_TH * TW / 32;
    static constexpr int A_BYTES = HP * PIXB;
    static constexpr int NITEMS = (HP * QPP + 255) / 256;
};

// ALIAS: the epilogue's transpose rows (and the statistics fold) alias the halo tile — one channel tile per block, one more barrier
template <int C, int TW, bool ALIAS>
__global__ __launch_bounds__(256, 2) void k_upconv(const UpK k) {
    using G = UpGeo<C, TW>;
    constexpr int PIXB = G::PIXB, QPP = G::QPP, KS16 = G::KS16, NF = G::NF, NITEMS = G::NITEMS;
    constexpr int UP_TW = TW, UP_HW = G::HW, UP_HP = G::HP, MT = G::MT;
    extern __shared__ __attribute__((aligned(16))) char smem_up[];
    char* const tile = smem_up;
    float* const epi_all = reinterpret_cast<float*>(ALIAS ? smem_up : smem_up + G::A_BYTES);

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int dy = wave >> 1, dx = wave & 1;
    int bid = blockIdx.x;
    if ((gridDim.x & 7) == 0) bid = (blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3);      // the slices of a sample meet in one XCD's L2
    const int n = bid / k.slices, slice = bid - n * k.slices;
    const int ct0 = blockIdx.y * k.ctb;
    const int Wout = 2 * k.Win, Cout = k.Cout;
    const char* const in_n = reinterpret_cast<const char*>(k.in + (size_t)n * k.Hin * k.Win * C);
    char* const out_n = reinterpret_cast<char*>(k.out + (size_t)n * 4 * k.Hin * k.Win * Cout);

    // this lane's A-fragment base of each 32-pixel sub-tile: pixel p = 32 mi + (lane & 31) of the 8x16 tile, window origin (dy, dx)
    int hpb[MT];
#pragma unroll
    for (int mi = 0; mi < MT; ++mi) {
        const int p = mi * 32 + (lane & 31);
        hpb[mi] = ((p / UP_TW + dy) * UP_HW + (p % UP_TW + dx)) * PIXB + (lane >> 5) * 16;
    }
    float* const epi = epi_all + wave * (32 * UP_EPS);
    const int cq = lane & 7, prow = lane >> 3;

    float s1[UP_CTB_MAX][4], s2[UP_CTB_MAX][4];
#pragma unroll
    for (int c = 0; c < UP_CTB_MAX; ++c)
#pragma unroll
        for (int e = 0; e < 4; ++e) { s1[c][e] = 0.f; s2[c][e] = 0.f; }

    const int ntile_sp = k.tiles_x * k.tiles_y;
#pragma unroll 1
    for (int t = slice; t < ntile_sp; t += k.slices) {
        const int ty = t / k.tiles_x, tx = t - ty * k.tiles_x;
        const int oy0 = ty * UP_TH, ox0 = tx * UP_TW;
        // ---- stage the halo tile: every request first, then the split ----
        f32x4 v[NITEMS];
        unsigned okm = 0;
        unsigned t_ = tid;
        asm volatile("" : "+v"(t_));       // item geometry recomputed where it is used: hoisted out of the tile loop it costs more registers than ALU
#pragma unroll
        for (int i = 0; i < NITEMS; ++i) {
            const int item = (int)t_ + i * 256;
            const int hp = item / QPP, q = item - hp * QPP;
            const int hy = hp / UP_HW, hx = hp - hy * UP_HW;
            const int iy = oy0 - 1 + hy, ix = ox0 - 1 + hx;
            const bool ok = item < UP_HP * QPP && (unsigned)iy < (unsigned)k.Hin && (unsigned)ix < (unsigned)k.Win;
            const int iyc = min(max(iy, 0), k.Hin - 1), ixc = min(max(ix, 0), k.Win - 1);
            v[i] = load16_global(in_n + (((size_t)(iyc * k.Win + ixc) * C + 4 * q) << 2));
            okm |= (ok ? 1u : 0u) << i;
        }
        __syncthreads();                                   // the previous tile's matrix phase (and epilogue rows, when they alias) are done
        asm volatile("" : "+v"(t_));
#pragma unroll
        for (int i = 0; i < NITEMS; ++i) {
            const int item = (int)t_ + i * 256;
            if ((i + 1) * 256 <= UP_HP * QPP || item < UP_HP * QPP) {
                const int hp = item / QPP, q = item - hp * QPP;
                const float lim = (okm >> i) & 1u ? __builtin_inff() : 0.f;            // padding -> 0 (one select per item)
                const float x0 = __builtin_amdgcn_fmed3f(v[i][0] * ACT_PRESCALE, -lim, lim), x1 = __builtin_amdgcn_fmed3f(v[i][1] * ACT_PRESCALE, -lim, lim);
                const float x2 = __builtin_amdgcn_fmed3f(v[i][2] * ACT_PRESCALE, -lim, lim), x3 = __builtin_amdgcn_fmed3f(v[i][3] * ACT_PRESCALE, -lim, lim);
                typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
                u32x2 hi, lo;
                unsigned h0, l0, h1, l1;
                split2_f16(x0, x1, h0, l0);
                split2_f16(x2, x3, h1, l1);
                hi[0] = h0; hi[1] = h1; lo[0] = l0; lo[1] = l1;
                char* d = tile + hp * PIXB + 8 * q;
                *reinterpret_cast<u32x2*>(d) = hi;
                *reinterpret_cast<u32x2*>(d + 2 * C) = lo;
            }
        }
        __syncthreads();

#pragma unroll 1
        for (int ct = 0; ct < k.ctb; ++ct) {
            const int nt = 4 * (ct0 + ct) + wave;                                      // this wave's n-tile of the packed 2x2 conv
            // fragment (k-step ks, window tap bt): slab ((bt * KS16 + ks) * ntiles + nt) of 2 KB, hi then lo
            const char* const wb = static_cast<const char*>(k.w) + ((size_t)nt << 11);
            const unsigned wstep = (unsigned)k.ntiles << 11;                            // bytes between consecutive (bt, ks) slabs
            // walk order j -> (k-step, window tap): k-step outer, tap inner for the 8x16 tiles; for the 8x8 tiles pairs of k-steps outer, tap, then
            // the pair's two k-steps — each the order in which the general kernel's form of the same geometry accumulates (16- / 32-channel chunks)
            auto ks_of = [&](const int j) { return TW == 8 ? 2 * (j >> 3) + (j & 1) : j >> 2; };
            auto bt_of = [&](const int j) { return TW == 8 ? (j & 7) >> 1 : j & 3; };
            auto frag_off = [&](const int j) { return (unsigned)(bt_of(j) * KS16 + ks_of(j)) * wstep; };
            f32x4 bq[UP_BD + 1][2];
            auto issue_b = [&](const int j) {
                bq[j % (UP_BD + 1)][0] = load16_uniform_base(wb + frag_off(j), (unsigned)lane << 4);
                bq[j % (UP_BD + 1)][1] = load16_uniform_base(wb + frag_off(j) + 1024, (unsigned)lane << 4);
            };
#pragma unroll
            for (int j = 0; j < UP_BD; ++j) issue_b(j);
            const float add = k.bias ? k.bias[min((ct0 + ct) * 32 + (lane & 31), Cout - 1)] : 0.f;
            const float wsc = k.wscale[nt * 32 + (lane & 31)];
            f32x16 acc[MT];
#pragma unroll
            for (int mi = 0; mi < MT; ++mi)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[mi][r] = 0.f;
            // software pipeline, one step deep: the A fragments of step j + 1 (8 LDS reads) and the B fragment of step j + 2 are requested in
            // front of the 12 MFMAs of step j (sched_barrier pins the order — the scheduler otherwise hoists reads until the registers are gone)
            f16x8 ah[2][MT], al[2][MT];
            auto load_a = [&](const int buf, const int j) {
                const int toff = ((bt_of(j) >> 1) * UP_HW + (bt_of(j) & 1)) * PIXB + 32 * ks_of(j);
#pragma unroll
                for (int mi = 0; mi < MT; ++mi) {
                    ah[buf][mi] = *reinterpret_cast<const f16x8*>(tile + hpb[mi] + toff);
                    al[buf][mi] = *reinterpret_cast<const f16x8*>(tile + hpb[mi] + toff + 2 * C);
                }
            };
            load_a(0, 0);
#pragma unroll
            for (int j = 0; j < NF; ++j) {
                if (j + UP_BD < NF) issue_b(j + UP_BD);
                if (j + 1 < NF) load_a((j + 1) & 1, j + 1);
                __builtin_amdgcn_sched_barrier(0);
                const f16x8 bh = __builtin_bit_cast(f16x8, bq[j % (UP_BD + 1)][0]), bl = __builtin_bit_cast(f16x8, bq[j % (UP_BD + 1)][1]);
#pragma unroll
                for (int mi = 0; mi < MT; ++mi) acc[mi] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[j & 1][mi], bh, acc[mi], 0, 0, 0);
#pragma unroll
                for (int mi = 0; mi < MT; ++mi) acc[mi] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[j & 1][mi], bl, acc[mi], 0, 0, 0);
#pragma unroll
                for (int mi = 0; mi < MT; ++mi) acc[mi] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[j & 1][mi], bh, acc[mi], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
            // ---- epilogue: (x 2^-e) + bias -> wave-private transpose -> float4 rows to pixel (2y + dy, 2x + dx), statistics ----
            if (ALIAS) __syncthreads();                    // every wave is done reading the halo tile
            float t1[4] = {0.f, 0.f, 0.f, 0.f}, t2[4] = {0.f, 0.f, 0.f, 0.f};
            // lane (prow, cq): tile column col0 + prow = 2 prow output pixels to the right of the row pass's first one, channel quad cq
            const unsigned lane_off = ((unsigned)(2 * prow) * (unsigned)Cout + (unsigned)((ct0 + ct) * 32 + 4 * cq)) << 2;
#pragma unroll
            for (int mi = 0; mi < MT; ++mi) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int pl = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                    epi[pl * UP_EPS + (lane & 31)] = fmaf(acc[mi][r], wsc, add);        // wsc is a power of two: the product is exact
                }
#pragma unroll
                for (int jr = 0; jr < 4; ++jr) {
                    // row pass jr: tile pixel p = 32 mi + 8 jr + prow -> tile row (32 mi + 8 jr) / TW, column (8 jr) % TW + prow
                    const int pl = jr * 8 + prow;
                    const f32x4 o = *reinterpret_cast<const f32x4*>(epi + pl * UP_EPS + 4 * cq);
                    const int row = (32 * mi + 8 * jr) / UP_TW, col0 = (8 * jr) % UP_TW;
                    const unsigned rb = (unsigned)(((2 * (oy0 + row) + dy) * Wout + 2 * (ox0 + col0) + dx) * Cout) << 2;      // uniform
                    store16_uniform_base(out_n + rb, lane_off, o);
#pragma unroll
                    for (int e = 0; e < 4; ++e) { t1[e] += o[e]; t2[e] = fmaf(o[e], o[e], t2[e]); }
                }
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                if (ct == 0) { s1[0][e] += t1[e]; s2[0][e] += t2[e]; }
                else { s1[UP_CTB_MAX - 1][e] += t1[e]; s2[UP_CTB_MAX - 1][e] += t2[e]; }
            }
        }
    }

    if (k.out_stats) {
        // lanes that hold the same channel quad, then the four waves = phases in order: one partial per (sample, slice, channel)
        __syncthreads();
        double* red = reinterpret_cast<double*>(epi_all);                               // [4 waves][ctb][32][2]
#pragma unroll
        for (int c = 0; c < UP_CTB_MAX; ++c)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                double v1 = (double)s1[c][e], v2 = (double)s2[c][e];
#pragma unroll
                for (int off = 8; off < 64; off <<= 1) { v1 += __shfl_xor(v1, off); v2 += __shfl_xor(v2, off); }
                if (lane < 8 && c < k.ctb) {
                    red[((wave * UP_CTB_MAX + c) * 32 + 4 * lane + e) * 2 + 0] = v1;
                    red[((wave * UP_CTB_MAX + c) * 32 + 4 * lane + e) * 2 + 1] = v2;
                }
            }
        __syncthreads();
        if constexpr (TW == 8) {
            // one partial per (slice, phase): slot = 4 * slice + phase, as the general kernel's one-phase-per-block form leaves them
            for (int i = tid; i < 4 * k.ctb * 32; i += 256) {
                const int w = i / (k.ctb * 32), c = (i / 32) % k.ctb, l = i & 31;
                double* o = k.out_stats + (((size_t)n * 4 * k.slices + 4 * slice + w) * Cout + (ct0 + c) * 32 + l) * 2;
                o[0] = red[((w * UP_CTB_MAX + c) * 32 + l) * 2 + 0];
                o[1] = red[((w * UP_CTB_MAX + c) * 32 + l) * 2 + 1];
            }
        } else {
            for (int i = tid; i < k.ctb * 32; i += 256) {
                const int c = i >> 5, l = i & 31;
                double a1 = 0.0, a2 = 0.0;
                for (int w = 0; w < 4; ++w) {
                    a1 += red[((w * UP_CTB_MAX + c) * 32 + l) * 2 + 0];
                    a2 += red[((w * UP_CTB_MAX + c) * 32 + l) * 2 + 1];
                }
                double* o = k.out_stats + (((size_t)n * k.slices + slice) * Cout + (ct0 + c) * 32 + l) * 2;
                o[0] = a1; o[1] = a2;
            }
        }
    }
}

bool upconv_eligible(const ccdm_conv_args& a) {
    if (a.prec != CCDM_PREC_F16X3 || a.up != 2) return false;                           // (a diagnostic bit in prec >> 8: the general kernel)
    if (a.C1 || a.in1 || a.stats0 || a.act != CCDM_ACT_NONE || a.film || a.emb_off >= 0 || a.resid || a.skip0 || a.fine_slices) return false;
    if (!(a.C0 == 32 || a.C0 == 64 || a.C0 == 96 || a.C0 == 128) || a.Cout % 32) return false;
    // a rule of the geometry, never of the batch: whole 8x16 tiles, or 8x8 tiles for 8-pixel-wide inputs (where the general rule tiles 8x8 too)
#ifdef CCDM_UP_MAXPX
    if (a.Hin * a.Win > CCDM_UP_MAXPX) return false;
#endif
    return a.Hin % UP_TH == 0 && (a.Win % 16 == 0 || a.Win == 8);
}

template <int C, int TW>
static int launch_upconv_ct(const UpK& k, dim3 grid, bool alias, hipStream_t s) {
    const size_t lds = alias ? (size_t)std::max(UpGeo<C, TW>::A_BYTES, UP_EPI_BYTES) : (size_t)UpGeo<C, TW>::A_BYTES + UP_EPI_BYTES;
    auto kern = alias ? k_upconv<C, TW, true> : k_upconv<C, TW, false>;
    static bool configured[2] = {false, false};          // (per instantiation and form: the attribute is set once, not on every enqueue)
    if (lds > 64 * 1024 && !configured[alias ? 1 : 0]) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
            return fail("upconv: cannot reserve %zu bytes of LDS", lds);
        configured[alias ? 1 : 0] = true;
    }
    hipLaunchKernelGGL(kern, grid, dim3(256), lds, s, k);
    return 0;
}
template <int C>
static int launch_upconv_c(const UpK& k, dim3 grid, bool alias, hipStream_t s) {
    return k.Win == 8 ? launch_upconv_ct<C, 8>(k, grid, alias, s) : launch_upconv_ct<C, 16>(k, grid, alias, s);
}

int launch_upconv(const ccdm_conv_args& a, int slices, int ntiles, const float* wscale, hipStream_t s) {
    UpK k;
    k.in = a.in0; k.w = a.w; k.wscale = wscale; k.bias = a.bias; k.out = a.out; k.out_stats = a.out_stats;
    k.N = a.N; k.Hin = a.Hin; k.Win = a.Win; k.Cout = a.Cout; k.ntiles = ntiles; k.slices = slices;
    k.tiles_x = a.Win == 8 ? 1 : a.Win / 16; k.tiles_y = a.Hin / UP_TH;
    // channel tiles per block: two from one staged tile while >= 512 blocks remain (a partitioning choice only: every output element
    // and every statistics partial is computed by the same instruction sequence either way)
    const int ctiles = a.Cout / 32;
#ifdef CCDM_UP_CTB1
    k.ctb = 1;
#else
    k.ctb = (ctiles % 2 == 0 && (long)a.N * slices * (ctiles / 2) >= 512) ? 2 : 1;
#endif
    const dim3 grid(a.N * slices, ctiles / k.ctb);
    const bool alias = k.ctb == 1;
    switch (a.C0) {
        case 32: return launch_upconv_c<32>(k, grid, alias, s);
        case 64: return launch_upconv_c<64>(k, grid, alias, s);
        case 96: return launch_upconv_c<96>(k, grid, alias, s);
        default: return launch_upconv_c<128>(k, grid, alias, s);
    }
}

}  // namespace ccdm
