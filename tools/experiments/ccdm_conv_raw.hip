// 3x3 convs of a RAW input (no GroupNorm, no activation on load), F16X3 arithmetic, without staging the input through LDS:
//   * Downsample.op — conv 3x3, stride 2, padding 1                                   reference unet.py:137-146
// (+bias, + per-channel output statistics for the GroupNorm that follows).
//
// Why another kernel.  The general kernel (ccdm_conv.hip) stages a halo tile through LDS because its input needs an affine, a SiLU and a
// split per element and each staged element is used by 9 taps.  A Downsample conv has nothing to apply, reads every input pixel for ~2.25
// taps only, and its 17 x 33 halo tile (45 KB per 16 channels + 18 KB of fragments) admits two blocks per CU: 2048 one-tile blocks are
// four rounds of a 23 500-cycle chain (commit 8 900, matrix work 3 300) — 62 us at 128x128 -> 64x64 for 168 MB of traffic (21 us at
// 8 TB/s) and 14 000 cycles of matrix work per SIMD (6 us).  Here, as in ccdm_conv1x1.hip, the A operand never touches LDS:
//   lane (output pixel p, k-group g) reads the 8 channels of input pixel (2y + dy - 1, 2x + dx - 1) for tap (dy, dx) and k-step ks
//   straight into registers (2 x 16 B; the 128-byte line of a pixel serves both k-steps of a 32-channel input and stays in L1 for the
//   taps that share it), splits them (x = hi + lo; 16 vector instructions per 6 matrix instructions) and feeds the MFMA; requests run
//   RAW_DEPTH steps ahead of their use.  The layer's weight fragments for the block's n-tile (9 taps x C/16 k-steps x 2 KB) are copied
//   to LDS once per block: one barrier per block, none per tile.
//   block = 4 waves, one 8 x 32 output tile at a time (wave w: rows 2w, 2w + 1), tiles slice, slice + slices, ...; epilogue straight
//   from the accumulator layout (lane = channel), statistics partial per block in ccdm_conv.hip's slot layout.
// Same products as the general kernel (lo*hi + hi*lo + hi*hi per k-step), taps outer / k-steps inner instead of chunks outer: equal to
// fp32 rounding.  Built for C in {32, 64} (one source), Cout % 32 == 0, Hout % 8 == 0, Wout % 32 == 0, even input sizes.
#include "ccdm_common.h"
#include "ccdm_conv_common.h"

namespace ccdm {

struct ConvRawK {
    ccdm_conv_args a;
    const float* wscale;
    int ntiles, slices, tiles_x, tiles_y;
    int dbg;                 // experiments builds: 1 no stores, 2 every request to one line, 4 no matrix instructions
};

constexpr int RAW_DEPTH = 2;          // request sets in flight ahead of the one being multiplied

template <int NKS>
__global__ __launch_bounds__(256, 3) void k_conv_down(const ConvRawK k) {
    constexpr int NSTEP = 9 * NKS, C = 16 * NKS;
    const ccdm_conv_args& a = k.a;
    extern __shared__ __attribute__((aligned(16))) char smem_r[];
    f32x4* ldsB = reinterpret_cast<f32x4*>(smem_r);                  // [step = tap * NKS + ks][hi|lo][64 lanes] x 16 B
    __shared__ double red[4 * 32 * 2];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int bid = blockIdx.x;
    if ((gridDim.x & 7) == 0) bid = (blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3);
    const int n = __builtin_amdgcn_readfirstlane(bid / k.slices), slice = __builtin_amdgcn_readfirstlane(bid - n * k.slices);
    const int nt = blockIdx.y;
    const int Hin = a.Hin, Win = a.Win, Wout = a.Wout;

    // ---- the n-tile's weight fragments -> LDS, once ----
    {
        const char* wsrc = static_cast<const char*>(a.w);
        constexpr int ITEMS = NSTEP * 128;                            // 16-byte items
#pragma unroll
        for (int i = 0; i < (ITEMS + 255) / 256; ++i) {
            const int it = tid + 256 * i;
            if (it < ITEMS) {
                const int st = it >> 7, rem = it & 127;              // packed: [tap][ks][ntile][hi|lo][64] — step st = tap * NKS + ks
                ldsB[it] = load16_global(wsrc + (((size_t)st * k.ntiles + nt) << 11) + ((size_t)rem << 4));
            }
        }
    }
    const int co = nt * 32 + (lane & 31);
    float add = a.bias ? a.bias[co] : 0.f;
    float wsc = k.wscale[co];
    asm volatile("" : "+v"(add), "+v"(wsc));
    __syncthreads();

    const int g = lane >> 5, pc = lane & 31;
    const char* inb = reinterpret_cast<const char*>(a.in0 + (size_t)n * Hin * Win * C);
    const int ntile_sp = k.tiles_x * k.tiles_y;
    const int my_tiles = (ntile_sp - slice + k.slices - 1) / k.slices;
    float t1 = 0.f, t2 = 0.f;
    const f16x8* bq = reinterpret_cast<const f16x8*>(ldsB) + lane;

    for (int it = 0; it < my_tiles; ++it) {
        const int tile = slice + it * k.slices;
        const int ty = tile / k.tiles_x, tx = tile - ty * k.tiles_x;
        const int oyw = ty * 8 + 2 * wave, ox = tx * 32 + pc;          // this lane's output pixels: (oyw + mi, ox)
        // addressing: the input row of (mi, dy) is wave-uniform (scalar base), the column part is one 32-bit lane offset per dx; padding = 1:
        // only input row -1 (uniform) and column -1 (the lanes of output column 0) fall outside
        f32x4 buf[RAW_DEPTH + 1][2][2];                               // [set][mi][half of the 8 channels]
        unsigned coloff[3];
        bool colok[3];
#pragma unroll
        for (int dx = 0; dx < 3; ++dx) {
            const int ix = 2 * ox + dx - 1;
            colok[dx] = ix >= 0;
            coloff[dx] = (unsigned)(max(ix, 0) * C * 4 + 32 * g);
        }
        auto request = [&](const int st) {
            const int tap = st / NKS, ks = st - tap * NKS;
#pragma unroll
            for (int mi = 0; mi < 2; ++mi) {
                const int iy = 2 * (oyw + mi) + tap / 3 - 1;           // uniform
                const char* rowp = inb + (size_t)max(iy, 0) * Win * C * 4;
#ifdef CCDM_EXPERIMENTS
                if (k.dbg & 2) {
                    buf[st % (RAW_DEPTH + 1)][mi][0] = load16_uniform_base(inb, 64u * ks + 32u * g);
                    buf[st % (RAW_DEPTH + 1)][mi][1] = load16_uniform_base(inb, 64u * ks + 32u * g + 16u);
                    continue;
                }
#endif
                buf[st % (RAW_DEPTH + 1)][mi][0] = load16_uniform_base(rowp, coloff[tap % 3] + 64u * ks);
                buf[st % (RAW_DEPTH + 1)][mi][1] = load16_uniform_base(rowp, coloff[tap % 3] + 64u * ks + 16u);
            }
        };
        auto scale_of = [&](const int st, const int mi) {             // 2^4 pre-scale, or 0 for a padding pixel
            const int tap = st / NKS;
            const bool rok = 2 * (oyw + mi) + tap / 3 - 1 >= 0;
            return (rok && colok[tap % 3]) ? ACT_PRESCALE : 0.f;
        };
        f32x16 acc[2];
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mi][r] = 0.f;
#pragma unroll
        for (int st = 0; st < RAW_DEPTH; ++st) request(st);
#pragma unroll
        for (int st = 0; st < NSTEP; ++st) {
            if (st + RAW_DEPTH < NSTEP) request(st + RAW_DEPTH);
            __builtin_amdgcn_sched_barrier(0);       // requests stay RAW_DEPTH steps ahead, no further (the scheduler would hoist them all: registers)
            const f16x8 wh = bq[(st * 2) * 64], wl = bq[(st * 2 + 1) * 64];
            f16x8 ah[2], al[2];
#pragma unroll
            for (int mi = 0; mi < 2; ++mi) {
                const f32x4 v0 = buf[st % (RAW_DEPTH + 1)][mi][0], v1 = buf[st % (RAW_DEPTH + 1)][mi][1];
                const float m = scale_of(st, mi);                     // zero padding; exact power-of-two pre-scale
                unsigned h[4], l[4];
                split2_f16(v0[0] * m, v0[1] * m, h[0], l[0]);
                split2_f16(v0[2] * m, v0[3] * m, h[1], l[1]);
                split2_f16(v1[0] * m, v1[1] * m, h[2], l[2]);
                split2_f16(v1[2] * m, v1[3] * m, h[3], l[3]);
                typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
                const u32x4 hv = {h[0], h[1], h[2], h[3]}, lv = {l[0], l[1], l[2], l[3]};
                ah[mi] = __builtin_bit_cast(f16x8, hv);
                al[mi] = __builtin_bit_cast(f16x8, lv);
            }
#ifdef CCDM_EXPERIMENTS
            if (k.dbg & 4) { acc[0][0] += (float)ah[0][0] + (float)al[1][1]; continue; }
#endif
#pragma unroll
            for (int mi = 0; mi < 2; ++mi) acc[mi] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[mi], wh, acc[mi], 0, 0, 0);
#pragma unroll
            for (int mi = 0; mi < 2; ++mi) acc[mi] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[mi], wl, acc[mi], 0, 0, 0);
#pragma unroll
            for (int mi = 0; mi < 2; ++mi) acc[mi] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[mi], wh, acc[mi], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        // epilogue: accumulator register r of sub-tile mi = output pixel column tx*32 + (r & 3) + 8 (r >> 2) + 4 g of row oyw + mi, channel = lane & 31
#pragma unroll
        for (int mi = 0; mi < 2; ++mi) {
            float* orow = a.out + (((size_t)n * a.Hout + oyw + mi) * Wout + tx * 32 + 4 * g) * a.Cout + co;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float v = fmaf(acc[mi][r], wsc, add);                   // wsc is a power of two: exact product
#ifdef CCDM_EXPERIMENTS
                if (!(k.dbg & 1) || v == 123.456f)
#endif
                orow[(size_t)((r & 3) + 8 * (r >> 2)) * a.Cout] = v;
                t1 += v;
                t2 = fmaf(v, v, t2);
            }
        }
    }
    if (a.out_stats) {
        double v1 = (double)t1, v2 = (double)t2;
        v1 += __shfl_xor(v1, 32);
        v2 += __shfl_xor(v2, 32);
        if (lane < 32) { red[(wave * 32 + lane) * 2] = v1; red[(wave * 32 + lane) * 2 + 1] = v2; }
        __syncthreads();
        if (tid < 32) {
            double s1 = 0.0, s2 = 0.0;
            for (int w = 0; w < 4; ++w) { s1 += red[(w * 32 + tid) * 2]; s2 += red[(w * 32 + tid) * 2 + 1]; }
            double* o = a.out_stats + (((size_t)n * k.slices + slice) * a.Cout + nt * 32 + tid) * 2;
            o[0] = s1; o[1] = s2;
        }
    }
}

// statistics slices (= blocks per sample): one per 8x32 output tile up to 16, a function of the output size only
int conv_down_slices(const ccdm_conv_args& a) {
    const int tiles = (a.Hout / 8) * (a.Wout / 32);
    return tiles < 16 ? tiles : 16;
}

bool conv_down_eligible(const ccdm_conv_args& a) {
    if (a.prec != CCDM_PREC_F16X3) return false;                          // (a diagnostic bit in prec >> 8: the general kernel)
    if (a.ksize != 3 || a.stride != 2 || a.up || a.film || a.stats0 || a.act != CCDM_ACT_NONE || a.emb_off >= 0 || a.resid || a.skip0 || a.in1) return false;
    if (a.fine_slices) return false;
    if (a.C0 != 32 && a.C0 != 64) return false;
    if (a.Cout % 32 || a.Hout % 8 || a.Wout % 32) return false;
    if (a.Hin != 2 * a.Hout || a.Win != 2 * a.Wout) return false;
    return a.Hout * a.Wout > 256;                                         // (smaller outputs: the K-split few-pixel kernel)
}

int launch_conv_down(const ccdm_conv_args& a, int ntiles, const float* wscale, hipStream_t s) {
    ConvRawK k;
    k.a = a; k.wscale = wscale; k.ntiles = ntiles;
    k.slices = conv_down_slices(a);
    k.dbg = exp_env("CCDM_RAW_DBG");
    k.tiles_x = a.Wout / 32; k.tiles_y = a.Hout / 8;
    const dim3 grid(a.N * k.slices, a.Cout / 32), block(256);
    if (a.C0 == 32) {
        hipLaunchKernelGGL((k_conv_down<2>), grid, block, 9 * 2 * 2048, s, k);
    } else {
        static bool configured = false;
        if (!configured) {
            if (hipFuncSetAttribute(reinterpret_cast<const void*>(k_conv_down<4>), hipFuncAttributeMaxDynamicSharedMemorySize, 9 * 4 * 2048) != hipSuccess)
                return fail("conv_down: cannot reserve %d bytes of LDS", 9 * 4 * 2048);
            configured = true;
        }
        hipLaunchKernelGGL((k_conv_down<4>), grid, block, 9 * 4 * 2048, s, k);
    }
    return 0;
}

}  // namespace ccdm
