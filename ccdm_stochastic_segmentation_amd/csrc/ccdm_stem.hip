// The stem conv of the U-Net — conv3x3(concat[one_hot(x_t), image]) -> model_channels, reference unet.py:517 (input_blocks[0]) fed by
// unet.py:760 (torch.cat([x, condition], 1)) — for inputs of at most 4 channels (LIDC: 2 classes + 1 image channel), F16X3 arithmetic.
//
// Why its own kernel.  On the general kernel (ccdm_conv.hip) the 3-4 real input channels are one 16-channel k-step per tap: 9 k-steps,
// 54 matrix instructions per wave and tile of which 3/4 multiply zeros, a 27 KB halo tile staged per chunk — 43.6 us for a layer whose
// traffic (16 MB in, 134 MB out at batch 64) is 19 us at 8 TB/s.  And the one-hot half of its input is a function of a BYTE per pixel that
// the epilogue kernel expanded into fp32 channels of `xin` every step ("the kernel may carry x_t as a uint8 class index
// and only materialise one-hot fp32 at the boundary").  Here:
//   * K axis = (tap, channel): k-step j covers taps 4j .. 4j+3 x 4 channels, 3 k-steps for the 9 taps (the 3 slots past tap 8 have zero
//     weights) — 18 matrix instructions per wave and tile.  Lane (pixel, k-group g) takes taps 4j+2g and 4j+2g+1: its A fragment is
//     two 8-byte LDS reads (the 4 staged channels of two neighbouring halo pixels), hi and lo.
//   * the one-hot is built on load: halo pixel -> one byte of x_t + the pixel's 16 B of `xin` (image channels; its channels [0, K) are
//     ignored) -> 4 values (x 2^4, exact) -> fp16 hi | lo -> 16 B of LDS.  The halo tile is 5.4 KB, double-buffered: one barrier per tile.
//   * the weight fragments of the block's n-tile (3 k-steps x hi|lo) live in registers for the whole block.
//   * epilogue straight from the accumulator layout (lane = channel: 32 lanes store one pixel's 128 B) + per-channel statistics partials
//     in the slot layout of ccdm_conv.hip (one slice per block, ccdm_conv_slices of the output size; fp32 per lane, fp64 from there on).
// Same products as the general kernel (lo*hi + hi*lo + hi*hi per slot), another summation order over (tap, channel): equal to fp32
// rounding.  Built for H % 8 == 0, W % 32 == 0, Cout % 32 == 0; everything else (and the exact-fp32 mode) stays on the general kernel.
#include "ccdm_common.h"
#include "ccdm_conv_common.h"

#include <cmath>
#include <vector>

namespace ccdm {

int conv_slices(int Hout, int Wout, int stride, bool up2, int fine);

struct StemK {
    ccdm_stem_args a;
    const float* wscale;     // [Cout] powers of two undoing the weight / activation pre-scales
    int slices, tiles_x, tiles_y;
};

constexpr int ST_TH = 8, ST_TW = 32, ST_HW = ST_TW + 2, ST_HH = ST_TH + 2, ST_HP = ST_HH * ST_HW;     // 10 x 34 halo pixels
constexpr int ST_TILE_BYTES = (ST_HP * 16 + 63) / 64 * 64;

__global__ __launch_bounds__(256) void k_stem(const StemK k) {
    const ccdm_stem_args& a = k.a;
    __shared__ __attribute__((aligned(16))) char tileb[2][ST_TILE_BYTES];
    __shared__ double red[4 * 32 * 2];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int bid = blockIdx.x;
    if ((gridDim.x & 7) == 0) bid = (blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3);      // slices of a sample on one XCD (speed only)
    const int n = __builtin_amdgcn_readfirstlane(bid / k.slices), slice = __builtin_amdgcn_readfirstlane(bid - n * k.slices);
    const int nt = blockIdx.y;
    const int H = a.H, W = a.W, K = a.K;
    const size_t px_n = (size_t)n * H * W;

    // ---- this n-tile's weight fragments: [k-step][hi|lo][64 lanes] x 16 B, resident for the whole block ----
    f16x8 wh[3], wl[3];
    {
        const char* wp = static_cast<const char*>(a.w) + ((size_t)nt * 3 * 2048) + ((size_t)lane << 4);
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            wh[j] = __builtin_bit_cast(f16x8, load16_global(wp + j * 2048));
            wl[j] = __builtin_bit_cast(f16x8, load16_global(wp + j * 2048 + 1024));
        }
    }
    const int co = nt * 32 + (lane & 31);
    const float add = a.bias ? a.bias[co] : 0.f;
    const float wsc = k.wscale[co];

    // ---- staging: thread -> up to two halo pixels of a tile (340 pixels, 256 threads) ----
    const int ntile_sp = k.tiles_x * k.tiles_y;
    const int my_tiles = (ntile_sp - slice + k.slices - 1) / k.slices;
    unsigned xtv[2];
    f32x4 xv[2];
    bool okv[2];
    auto request = [&](const int tile) {
        const int ty = tile / k.tiles_x, tx = tile - ty * k.tiles_x;
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int hp = min(tid + 256 * u, ST_HP - 1);
            const int hy = hp / ST_HW, hx = hp - hy * ST_HW;
            const int iy = ty * ST_TH - 1 + hy, ix = tx * ST_TW - 1 + hx;
            okv[u] = ((unsigned)iy < (unsigned)H) & ((unsigned)ix < (unsigned)W);
            const size_t p = px_n + (size_t)min(max(iy, 0), H - 1) * W + min(max(ix, 0), W - 1);      // clamped: the loads stay unconditional
            xtv[u] = a.xt[p];
            xv[u] = load16_global(reinterpret_cast<const char*>(a.xin + p * 4));
        }
    };
    auto commit = [&](const int buf) {
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int hp = tid + 256 * u;
            if (hp < ST_HP) {
                float v[4];
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const float raw = c < K ? ((int)xtv[u] == c ? 1.0f : 0.0f) : xv[u][c];       // one-hot built here; image channels from xin
                    v[c] = okv[u] ? raw * ACT_PRESCALE : 0.f;
                }
                unsigned h0, l0, h1, l1;
                split2_f16(v[0], v[1], h0, l0);
                split2_f16(v[2], v[3], h1, l1);
                typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
                const u32x4 o = {h0, h1, l0, l1};                                                 // [hi0..hi3 | lo0..lo3]
                *reinterpret_cast<u32x4*>(tileb[buf] + hp * 16) = o;
            }
        }
    };

    // per-lane A geometry: sub-tile mi of this wave = tile row 2 wave + mi, pixel column lane & 31; k-group g = lane >> 5
    const int g = lane >> 5, pc = lane & 31;
    int aoff[3][2];                                     // byte offset of tap (4j + 2g + e) relative to the output pixel's halo origin
#pragma unroll
    for (int j = 0; j < 3; ++j)
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const int tap = min(4 * j + 2 * g + e, 8);  // slots past tap 8 carry zero weights: any finite value will do
            aoff[j][e] = ((tap / 3) * ST_HW + tap % 3) * 16;
        }
    float t1 = 0.f, t2 = 0.f;
    // name the block-resident operands here: inside the tile loop the compiler cannot tell these loads from the loop's own prefetch and
    // would wait for everything in flight (vmcnt(0): the next tile's request, just issued) at their first use of every tile
    float addv = add, wscv = wsc;
#pragma unroll
    for (int j = 0; j < 3; ++j) asm volatile("" : "+v"(wh[j]), "+v"(wl[j]));
    asm volatile("" : "+v"(addv), "+v"(wscv));

    if (my_tiles > 0) request(slice);
    for (int it = 0; it < my_tiles; ++it) {
        const int tile = slice + it * k.slices;
        const int buf = it & 1;
        commit(buf);
        if (it + 1 < my_tiles) request(tile + k.slices);
        __syncthreads();
        const int ty = tile / k.tiles_x, tx = tile - ty * k.tiles_x;
        f32x16 acc[2];
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mi][r] = 0.f;
#pragma unroll
        for (int mi = 0; mi < 2; ++mi) {
            const char* base = tileb[buf] + ((2 * wave + mi) * ST_HW + pc) * 16;
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
                typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
                const u32x2 h0 = *reinterpret_cast<const u32x2*>(base + aoff[j][0]), l0 = *reinterpret_cast<const u32x2*>(base + aoff[j][0] + 8);
                const u32x2 h1 = *reinterpret_cast<const u32x2*>(base + aoff[j][1]), l1 = *reinterpret_cast<const u32x2*>(base + aoff[j][1] + 8);
                const u32x4 hv = {h0[0], h0[1], h1[0], h1[1]}, lv = {l0[0], l0[1], l1[0], l1[1]};
                const f16x8 ah = __builtin_bit_cast(f16x8, hv), al = __builtin_bit_cast(f16x8, lv);
                acc[mi] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, wh[j], acc[mi], 0, 0, 0);
                acc[mi] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, wl[j], acc[mi], 0, 0, 0);
                acc[mi] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, wh[j], acc[mi], 0, 0, 0);
            }
        }
        // epilogue: accumulator register r of sub-tile mi = pixel column (r & 3) + 8 (r >> 2) + 4 g of tile row 2 wave + mi, channel = lane & 31
#pragma unroll
        for (int mi = 0; mi < 2; ++mi) {
            const int oy = ty * ST_TH + 2 * wave + mi;
            float* orow = a.out + (px_n + (size_t)oy * W + tx * ST_TW + 4 * g) * a.Cout + co;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float v = fmaf(acc[mi][r], wscv, addv);                 // wsc is a power of two: exact product
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

static bool stem_supported(int Cs, int Cout, int H, int W, int prec) {
    return prec == CCDM_PREC_F16X3 && Cs == 4 && Cout > 0 && Cout % 32 == 0 && H > 0 && W > 0 && H % ST_TH == 0 && W % ST_TW == 0;
}

int launch_stem(const ccdm_stem_args& a, hipStream_t s) {
    CCDM_REQUIRE(a.xt && a.xin && a.w && a.out, "stem_conv: null pointer");
    CCDM_REQUIRE(stem_supported(a.Cs, a.Cout, a.H, a.W, CCDM_PREC_F16X3), "stem_conv: Cs=%d Cout=%d %dx%d is not built (ccdm_stem_conv_supported)", a.Cs, a.Cout,
                 a.H, a.W);
    CCDM_REQUIRE(a.K >= 1 && a.K <= a.Cs && a.N > 0, "stem_conv: K=%d of Cs=%d channels, N=%d", a.K, a.Cs, a.N);
    StemK k;
    k.a = a;
    k.slices = conv_slices(a.H, a.W, 1, false, 0);
    k.tiles_x = a.W / ST_TW; k.tiles_y = a.H / ST_TH;
    if (a.out_stats) CCDM_REQUIRE(a.out_slices == k.slices, "stem_conv: out_slices %d != %d", a.out_slices, k.slices);
    k.wscale = reinterpret_cast<const float*>(static_cast<const char*>(a.w) + (size_t)(a.Cout / 32) * 3 * 2048);
    hipLaunchKernelGGL(k_stem, dim3(a.N * k.slices, a.Cout / 32), dim3(256), 0, s, k);
    CCDM_CHECK_LAUNCH("stem_conv");
    return 0;
}

}  // namespace ccdm

extern "C" int ccdm_stem_conv_supported(int Cs, int Cout, int H, int W, int prec) { return ccdm::stem_supported(Cs, Cout, H, W, prec) ? 1 : 0; }

extern "C" int ccdm_stem_conv(const ccdm_stem_args* a, void* stream) {
    if (!a) return ccdm::fail("ccdm_stem_conv: null args");
    return ccdm::launch_stem(*a, (hipStream_t)stream);
}

// Packed layout: [n-tile][k-step j = 0..2][hi|lo][64 lanes][8 halfs]; lane l (cout = nt*32 + (l & 31), k-group g = l >> 5), element e:
// W[cout][ch = e & 3][tap = 4j + 2g + (e >> 2)] * 2^e(cout), zero past tap 8 / beyond Cin, split into fp16 hi + lo; then [Cout] floats
// 2^-e(cout) / ACT_PRESCALE.  (The power of two puts max|W| of the channel in [2^9, 2^10), as ccdm_pack_conv_weight does.)
extern "C" size_t ccdm_pack_stem_weight(const float* oihw, int Cout, int Cin, void* out) {
    if (Cout <= 0 || Cout % 32 || Cin < 1 || Cin > 4) { ccdm::fail("pack_stem: Cout=%d Cin=%d (need Cout %% 32 == 0, Cin <= 4)", Cout, Cin); return 0; }
    const int ntiles = Cout / 32;
    const size_t frag = (size_t)ntiles * 3 * 2048, total = frag + (size_t)Cout * sizeof(float);
    if (!out) return total;
    _Float16* o = static_cast<_Float16*>(out);
    float* sc = reinterpret_cast<float*>(static_cast<char*>(out) + frag);
    std::vector<float> mul(Cout, 1.0f);
    for (int co = 0; co < Cout; ++co) {
        float mx = 0.f;
        for (int i = 0; i < Cin * 9; ++i) mx = fmaxf(mx, fabsf(oihw[(size_t)co * Cin * 9 + i]));
        int e = 0;
        if (mx > 0.f && std::isfinite(mx)) { int ex; frexpf(mx, &ex); e = 10 - ex; }
        if (e > 60) e = 60;
        if (e < -60) e = -60;
        mul[co] = ldexpf(1.0f, e);
        sc[co] = ldexpf(1.0f, -e) / ccdm::ACT_PRESCALE;
    }
    for (int nt = 0; nt < ntiles; ++nt)
        for (int j = 0; j < 3; ++j)
            for (int l = 0; l < 64; ++l)
                for (int e = 0; e < 8; ++e) {
                    const int co = nt * 32 + (l & 31), tap = 4 * j + 2 * (l >> 5) + (e >> 2), ch = e & 3;
                    float v = 0.f;
                    if (tap < 9 && ch < Cin) v = oihw[((size_t)co * Cin + ch) * 9 + tap] * mul[co];
                    const _Float16 hi = (_Float16)v;
                    const _Float16 lo = (_Float16)(v - (float)hi);
                    const size_t base = ((size_t)(nt * 3 + j) * 2) * 64 * 8;
                    o[base + (size_t)l * 8 + e] = hi;
                    o[base + 64 * 8 + (size_t)l * 8 + e] = lo;
                }
    return total;
}
