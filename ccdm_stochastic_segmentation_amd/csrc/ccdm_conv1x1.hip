// 1x1 conv ([GroupNorm ->] conv +bias +residual, + per-channel output statistics) of a low-resolution tensor, F16X3 arithmetic:
// AttentionBlock.proj_out + the block's residual (reference unet.py:300,311) — 11 launches per LIDC denoise step — and, where the
// norm+qkv+attention kernel does not apply (T > 256: Cityscapes sizes), AttentionBlock.norm + qkv (unet.py:291-299,306).
//
// The general kernel (ccdm_conv.hip) stages every input through LDS — two barrier-separated round trips per channel chunk,
// a halo walk, a commit pass — which is what a 3x3 conv with GroupNorm on load needs and what a 64-pixel x 128-channel GEMM
// does not: such a launch is one round of blocks and its time is one block's chain (tools/timeline_op.py: 14 000 cycles, of
// which the arithmetic is under 2 000).  Here a wave's A operand never touches LDS: lane (pixel = lane & 31, k-group = lane >> 5)
// reads its 8 consecutive channels of every 16-channel k-step straight into registers (2 x 16 B), splits them (x = hi + lo) and
// feeds the MFMA; the B fragments are read in their packed layout (ccdm_pack_conv_weight: [k-step][n-tile][hi|lo][lane] x 16 B,
// one coalesced 1 KB request per fragment).  Every load of the block — A, B, residual, bias, weight scales — is issued before the
// first use: ONE memory round trip, no barrier before the statistics fold.
//
//   block = one statistics slice of one sample (HW / slices pixels, 32 per wave) x one 32-channel output tile
//   same products in the same order as ccdm_conv.hip's 1x1 path (k-steps ascending; lo*hi, hi*lo, hi*hi): identical outputs.
//   statistics: lane = channel, 16 pixels per lane in fp32, widened to fp64 before lanes and waves are combined (fixed order).
#include "ccdm_common.h"
#include "ccdm_conv_common.h"

#include <cstdlib>

namespace ccdm {

struct Conv1x1K {
    const float* in;        // [N, HW, C]
    const void* w;          // packed fragments, ksize 1
    const float* wscale;    // [ntiles*32]
    const float* bias;      // [Cout] or NULL
    const float* resid;     // [N, HW, Cout] or NULL
    float* out;             // [N, HW, Cout]
    double* out_stats;      // [N, slices, Cout, 2] or NULL
    int C, Cout, HW, slices, ntiles, px_per_block;   // slices = pixel blocks per sample (= statistics slices when out_stats is written)
    ccdm_conv_args a;       // GroupNorm operands (stats0, slices0, gamma, beta, eps, Hin, Win) for the GN variant
};

// KSB k-steps (16 channels each) are in flight at a time: C <= 16 * KSB runs with every load issued up front
// (MAXT: the block size bound the register budget is planned for — 4 waves with 8 k-steps in flight, 16 waves with 4)
// GN: the input is normalised on load — (scale, shift) table of this sample in LDS, built in the prologue from the producer's
// statistics partials exactly like ccdm_conv.hip does (gn_prefetch / gn_affine_block), one fma per element before the split.
template <int KSB, int MAXT, bool GN>
__global__ __launch_bounds__(MAXT) void k_conv1x1(const Conv1x1K k) {
    extern __shared__ __attribute__((aligned(16))) char smem1[];
    float2* ab = reinterpret_cast<float2*>(smem1);                             // [C] (GN only)
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nwaves = blockDim.x >> 6;
    const int n = blockIdx.x / k.slices, slice = blockIdx.x - n * k.slices;
    const int nt = blockIdx.y;
    const int row = lane & 31, kg = lane >> 5;
    const int p = slice * k.px_per_block + wave * 32 + row;                    // pixel within the sample
    const int nks = k.C >> 4;
    const char* ap = reinterpret_cast<const char*>(k.in + ((size_t)n * k.HW + p) * k.C + 8 * kg);
    const char* bp = static_cast<const char*>(k.w) + (((size_t)nt * 128 + lane) << 4);
    const size_t bks = (size_t)k.ntiles * 128 * 16;                            // bytes per k-step of the packed weights
    const int co = nt * 32 + row;                                              // this lane's output channel
    // accumulator register r holds pixel wave*32 + (r & 3) + 8 * (r >> 2) + 4 * kg of the block
    const size_t obase = ((size_t)n * k.HW + slice * k.px_per_block + wave * 32 + 4 * kg) * k.Cout + co;

    GnPrefetch gpf;
    if (GN) gn_prefetch(k.a, true, n, 0, tid, (int)blockDim.x, k.w, gpf);         // ahead of the operand loads, consumed behind them
    f32x4 a0[KSB], a1[KSB], bh[KSB], bl[KSB];
    auto issue = [&](const int ks0) {
#pragma unroll
        for (int i = 0; i < KSB; ++i) {
            const int ks = min(ks0 + i, nks - 1);                              // clamped: the loads stay unconditional
            a0[i] = load16_global(ap + ks * 64);
            a1[i] = load16_global(ap + ks * 64 + 16);
            bh[i] = load16_global(bp + ks * bks);
            bl[i] = load16_global(bp + ks * bks + 1024);
        }
    };
    issue(0);
    float rs[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) rs[r] = 0.f;
    if (k.resid) {
#pragma unroll
        for (int r = 0; r < 16; ++r) rs[r] = k.resid[obase + (size_t)((r & 3) + 8 * (r >> 2)) * k.Cout];
    }
    const float add = k.bias ? k.bias[co] : 0.f;
    const float wsc = k.wscale[co];
    if (GN) {
        gn_affine_block(k.a, n, 0, gpf, reinterpret_cast<f64x2*>(smem1 + (size_t)k.C * 8), ab);
        __syncthreads();
    }
    const float2* abl = ab + 8 * kg;

    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    for (int ks0 = 0; ks0 < nks; ks0 += KSB) {
#pragma unroll
        for (int i = 0; i < KSB; ++i) {
            if (ks0 + i < nks) {                                               // uniform
                unsigned h[4], l[4];
                float x[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float v = j < 4 ? a0[i][j] : a1[i][j - 4];
                    if (GN) {                                                  // same roundings as the general kernel: (scale, shift) * 2^4 exactly, one fma
                        const float2 t = abl[16 * (ks0 + i) + j];
                        x[j] = fmaf(v, t.x * ACT_PRESCALE, t.y * ACT_PRESCALE);
                    } else x[j] = v * ACT_PRESCALE;
                }
                split2_f16(x[0], x[1], h[0], l[0]);
                split2_f16(x[2], x[3], h[1], l[1]);
                split2_f16(x[4], x[5], h[2], l[2]);
                split2_f16(x[6], x[7], h[3], l[3]);
                typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
                const u32x4 hv = {h[0], h[1], h[2], h[3]}, lv = {l[0], l[1], l[2], l[3]};
                const f16x8 ah = __builtin_bit_cast(f16x8, hv), al = __builtin_bit_cast(f16x8, lv);
                const f16x8 wh = __builtin_bit_cast(f16x8, bh[i]), wl = __builtin_bit_cast(f16x8, bl[i]);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, wh, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, wl, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, wh, acc, 0, 0, 0);
            }
        }
        if (ks0 + KSB < nks) issue(ks0 + KSB);                                 // wide inputs (C > 16 * KSB): next batch
    }

    float t1 = 0.f, t2 = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        float v = fmaf(acc[r], wsc, add);                                      // wsc is a power of two: exact product
        if (k.resid) v += rs[r];
        k.out[obase + (size_t)((r & 3) + 8 * (r >> 2)) * k.Cout] = v;
        t1 += v;
        t2 = fmaf(v, v, t2);
    }
    if (k.out_stats) {
        __shared__ double red[(MAXT / 64) * 32 * 2];
        double v1 = (double)t1, v2 = (double)t2;
        v1 += __shfl_xor(v1, 32);
        v2 += __shfl_xor(v2, 32);
        if (lane < 32) { red[(wave * 32 + lane) * 2] = v1; red[(wave * 32 + lane) * 2 + 1] = v2; }
        __syncthreads();
        if (tid < 32) {
            double s1 = 0.0, s2 = 0.0;
            for (int w = 0; w < nwaves; ++w) { s1 += red[(w * 32 + tid) * 2]; s2 += red[(w * 32 + tid) * 2 + 1]; }
            double* o = k.out_stats + (((size_t)n * k.slices + slice) * k.Cout + nt * 32 + tid) * 2;
            o[0] = s1; o[1] = s2;
        }
    }
}

// Several output-channel tiles per block (wide outputs on few pixels: qkv has 3C channels): the block's A fragments — loaded,
// normalised and split ONCE — stay in registers while the block walks `ntb` consecutive n-tiles, fetching each tile's packed B
// fragments (the next tile's while the current one multiplies).  No output statistics (qkv has no consumer GroupNorm), C <= 128.
// One n-tile per block would redo the GroupNorm table, the 64 KB of A loads and the split 12 times for a 128 -> 384 conv.
template <bool GN>
__global__ __launch_bounds__(256) void k_conv1x1_multi(const Conv1x1K k, const int ntb) {
    constexpr int KSB = 8;
    extern __shared__ __attribute__((aligned(16))) char smem1[];
    float2* ab = reinterpret_cast<float2*>(smem1);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n = blockIdx.x / k.slices, slice = blockIdx.x - n * k.slices;
    const int row = lane & 31, kg = lane >> 5;
    const int p = slice * k.px_per_block + wave * 32 + row;
    const int nks = k.C >> 4;
    const char* ap = reinterpret_cast<const char*>(k.in + ((size_t)n * k.HW + p) * k.C + 8 * kg);
    const size_t bks = (size_t)k.ntiles * 128 * 16;
    const int nt_real = k.Cout >> 5;
    const int nt_first = blockIdx.y * ntb;

    GnPrefetch gpf;
    if (GN) gn_prefetch(k.a, true, n, 0, tid, (int)blockDim.x, k.w, gpf);
    f32x4 a0[KSB], a1[KSB];
#pragma unroll
    for (int i = 0; i < KSB; ++i) {
        const int ks = min(i, nks - 1);
        a0[i] = load16_global(ap + ks * 64);
        a1[i] = load16_global(ap + ks * 64 + 16);
    }
    f32x4 bh[2][KSB], bl[2][KSB];
    auto issueB = [&](const int buf, const int nt) {
        const char* bp = static_cast<const char*>(k.w) + (((size_t)min(nt, nt_real - 1) * 128 + lane) << 4);
#pragma unroll
        for (int i = 0; i < KSB; ++i) {
            const int ks = min(i, nks - 1);
            bh[buf][i] = load16_global(bp + ks * bks);
            bl[buf][i] = load16_global(bp + ks * bks + 1024);
        }
    };
    issueB(0, nt_first);
    if (GN) {
        gn_affine_block(k.a, n, 0, gpf, reinterpret_cast<f64x2*>(smem1 + (size_t)k.C * 8), ab);
        __syncthreads();
    }
    const float2* abl = ab + 8 * kg;
    f16x8 ah[KSB], al[KSB];
#pragma unroll
    for (int i = 0; i < KSB; ++i) {
        unsigned h[4], l[4];
        float x[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float v = j < 4 ? a0[i][j] : a1[i][j - 4];
            if (GN) {
                const float2 t = abl[16 * min(i, nks - 1) + j];
                x[j] = fmaf(v, t.x * ACT_PRESCALE, t.y * ACT_PRESCALE);
            } else x[j] = v * ACT_PRESCALE;
        }
        split2_f16(x[0], x[1], h[0], l[0]);
        split2_f16(x[2], x[3], h[1], l[1]);
        split2_f16(x[4], x[5], h[2], l[2]);
        split2_f16(x[6], x[7], h[3], l[3]);
        typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
        const u32x4 hv = {h[0], h[1], h[2], h[3]}, lv = {l[0], l[1], l[2], l[3]};
        ah[i] = __builtin_bit_cast(f16x8, hv);
        al[i] = __builtin_bit_cast(f16x8, lv);
    }
    auto tile = [&](const int buf, const int nt) {
        const int co = nt * 32 + row;
        const size_t obase = ((size_t)n * k.HW + slice * k.px_per_block + wave * 32 + 4 * kg) * k.Cout + co;
        float rs[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) rs[r] = 0.f;
        if (k.resid) {
#pragma unroll
            for (int r = 0; r < 16; ++r) rs[r] = k.resid[obase + (size_t)((r & 3) + 8 * (r >> 2)) * k.Cout];
        }
        const float add = k.bias ? k.bias[co] : 0.f;
        const float wsc = k.wscale[co];
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
        for (int i = 0; i < KSB; ++i) {
            if (i < nks) {
                const f16x8 wh = __builtin_bit_cast(f16x8, bh[buf][i]), wl = __builtin_bit_cast(f16x8, bl[buf][i]);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[i], wh, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], wl, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], wh, acc, 0, 0, 0);
            }
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            float v = fmaf(acc[r], wsc, add);
            if (k.resid) v += rs[r];
            k.out[obase + (size_t)((r & 3) + 8 * (r >> 2)) * k.Cout] = v;
        }
    };
    for (int t = 0; t < ntb; t += 2) {                                          // two n-tiles per trip: static buffer indices
        const int nt = nt_first + t;
        if (nt >= nt_real) break;
        issueB(1, nt + 1);
        tile(0, nt);
        if (t + 1 >= ntb || nt + 1 >= nt_real) break;
        issueB(0, nt + 2);
        tile(1, nt + 1);
    }
}

// pixels per block: one statistics slice when statistics are written (<= 512 pixels: 16 waves), else 128 / 64 / 32 (any that divides HW)
static int conv1x1_px_per_block(const ccdm_conv_args& a, int slices) {
    const int HW = a.Hout * a.Wout;
    if (a.out_stats) return (HW % (slices * 32) == 0 && HW / slices <= 512) ? HW / slices : 0;
    for (int p = 128; p >= 32; p >>= 1)
        if (HW % p == 0) return p;
    return 0;
}

// 1x1 conv of a low-resolution tensor: the whole geometry must tile exactly (no masks in the kernel)
bool conv1x1_eligible(const ccdm_conv_args& a, int slices) {
    if (a.prec != CCDM_PREC_F16X3) return false;                               // (any diagnostic bit in prec >> 8, e.g. CCDM_DIAG_GENERAL_KERNEL: general kernel)
    if (a.ksize != 1 || a.stride != 1 || a.up || a.act != CCDM_ACT_NONE || a.film || a.skip0 || a.emb_off >= 0 || a.in1) return false;
    const int HW = a.Hout * a.Wout;
    if (a.C0 % 16 || a.Cout % 32 || a.C0 > CCDM_MAX_CHANNELS) return false;
    const int ppb = conv1x1_px_per_block(a, slices);
    if (a.stats0 && ppb > 128) return false;                                   // the GroupNorm variant is planned for 4-wave blocks (registers)
    // low-resolution stages only — a rule of the geometry alone, never of the batch size: this kernel and the general one partition the
    // output statistics differently, so the choice must not change when a batch is sharded over ranks or sub-batches
    return ppb > 0 && HW <= 16384;
}

int launch_conv1x1(const ccdm_conv_args& a, int slices, int ntiles, const float* wscale, hipStream_t s) {
    Conv1x1K k;
    k.in = a.in0; k.w = a.w; k.wscale = wscale; k.bias = a.bias; k.resid = a.resid; k.out = a.out; k.out_stats = a.out_stats;
    k.C = a.C0; k.Cout = a.Cout; k.HW = a.Hout * a.Wout; k.ntiles = ntiles;
    k.px_per_block = conv1x1_px_per_block(a, slices);
    k.slices = k.HW / k.px_per_block;
    k.a = a;
    const dim3 grid(a.N * k.slices, a.Cout / 32), block(k.px_per_block / 32 * 64);
    const size_t lds = a.stats0 ? (size_t)a.C0 * 8 + (size_t)((unsigned)a.C0 > block.x ? (unsigned)a.C0 : block.x) * 16 : 0;   // (scale, shift) table + the statistics exchange
    // wide outputs without statistics (qkv): walk several n-tiles per block so that ~512 blocks remain
    const int nt_real = a.Cout / 32;
    const long xb = (long)a.N * k.slices;
    int ntb = (int)((xb * nt_real + 256) / 512);
    ntb = ntb < 1 ? 1 : (ntb > nt_real ? nt_real : ntb);
    if (!a.out_stats && a.C0 <= 128 && block.x <= 256 && ntb > 1) {
        const dim3 gridm((unsigned)xb, (unsigned)((nt_real + ntb - 1) / ntb));
        if (a.stats0) hipLaunchKernelGGL((k_conv1x1_multi<true>), gridm, block, lds, s, k, ntb);
        else hipLaunchKernelGGL((k_conv1x1_multi<false>), gridm, block, 0, s, k, ntb);
        return 0;
    }
    if (a.stats0) {
        if (block.x <= 256) hipLaunchKernelGGL((k_conv1x1<8, 256, true>), grid, block, lds, s, k);
        else hipLaunchKernelGGL((k_conv1x1<4, 1024, true>), grid, block, lds, s, k);
    } else {
        if (block.x <= 256) hipLaunchKernelGGL((k_conv1x1<8, 256, false>), grid, block, 0, s, k);
        else hipLaunchKernelGGL((k_conv1x1<4, 1024, false>), grid, block, 0, s, k);
    }
    return 0;
}

}  // namespace ccdm
