// The head of the U-Net and the step epilogue in ONE launch, for few classes (9 K <= 32: LIDC's K = 2) and 32 head channels:
//
//     logits = conv3x3( SiLU( GroupNorm32(h) ) ) + bias                                   unet.py:701-707 (self.out)
//     x_{t-1} | outputs = softmax -> theta_post_prob -> clamp -> normalise -> draw         unet.py:706, diffusion_denoising.py:99-128,204-212
//
// Why its own kernel.  On the general kernel (ccdm_conv.hip) the 2-4 real output channels ride a 32-wide matrix tile: 108 matrix
// instructions per wave and 8x32 tile of which 15/16 compute padding, the [N,H,W,4] logits make a round trip through HBM, and the
// epilogue kernel is one more launch (74 + 16 us of a 2.9 ms step).  Here (north_star: "a fused one-hot -> logits -> categorical-posterior
// -> sample epilogue"):
//   * the TAPS are the N dimension: Z[q, (tap, c)] = sum_ci W[c, ci, tap] a[q, ci] for every pixel q of the HALO tile — a 1x1 product with
//     9 K <= 32 outputs, 11 sub-tiles x 2 k-steps x 3 = 66 matrix instructions per tile for the whole block (432 before) — and
//     logit[p, c] = bias_c + 2^-e_c * sum_tap Z[p + tap, (tap, c)] is nine 8-byte LDS reads per pixel;
//   * the whole 32-channel halo tile (10 x 34 pixels) is normalised, activated and split ONCE per tile into LDS (48 KB; Z aliases it behind
//     a barrier): one commit pass, no channel chunks;
//   * thread = output pixel runs the epilogue's arithmetic (ccdm_sampler_common.h: the SAME function the stand-alone kernel calls) on its
//     K logits in registers: x_{t-1} (1 byte), and on the last step the probabilities / one-hot — the logits never reach memory.
// Same products as the general kernel (lo*hi + hi*lo + hi*hi per k-step), summed over channels first and taps second: equal to fp32
// rounding.  Built for C == 32, 9 K <= 32, H % 8 == 0, W % 32 == 0; everything else stays the general conv + ccdm_posterior_sample.
#include "ccdm_common.h"
#include "ccdm_conv_common.h"
#include "ccdm_sampler_common.h"

#include <cmath>
#include <vector>

namespace ccdm {

int conv_slices(int Hout, int Wout, int stride, bool up2, int fine);

struct HeadK {
    ccdm_head_args a;
    ccdm_post_args post;
    ccdm_conv_args gn;          // GroupNorm operands in the form gn_prefetch / gn_affine_block take
    const float* wscale;        // [K] powers of two undoing the weight / activation pre-scales
    int slices, tiles_x, tiles_y;
};

constexpr int HD_TH = 8, HD_TW = 32, HD_HW = HD_TW + 2, HD_HH = HD_TH + 2, HD_HP = HD_HH * HD_HW;      // 10 x 34 halo pixels
constexpr int HD_C = 32, HD_PIX = HD_C * 4 + 16;                     // 32 hi | 32 lo halfs | 16 B pad: 144 B per pixel (conflict-free b128 reads)
constexpr int HD_SUB = (HD_HP + 31) / 32;                            // 11 sub-tiles of 32 halo pixels (the last 12 rows are padding)
constexpr int HD_A_BYTES = HD_SUB * 32 * HD_PIX;
constexpr int HD_ZP = 20;                                            // floats per pixel of the Z tile (9 K <= 18 used at K = 2; up to 32 at K = 3: see HD_ZP3)
constexpr int HD_NITEM = (HD_HP * (HD_C / 4) + 255) / 256;           // 11 staging items per thread

template <int KP>
__global__ __launch_bounds__(256, 3) void k_head(const HeadK k) {
    constexpr int ZP = KP == 2 ? HD_ZP : 36;                         // Z pitch in floats (multiple of 4, not of 32)
    const ccdm_head_args& a = k.a;
    extern __shared__ __attribute__((aligned(16))) char smem_h[];
    float2* ab = reinterpret_cast<float2*>(smem_h);                   // [32] GroupNorm (scale, shift)
    char* tileb = smem_h + HD_C * 8;                                  // A tile; the Z tile aliases its head
    float* zt = reinterpret_cast<float*>(tileb);

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int bid = blockIdx.x;
    if ((gridDim.x & 7) == 0) bid = (blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3);
    const int n = __builtin_amdgcn_readfirstlane(bid / k.slices), slice = __builtin_amdgcn_readfirstlane(bid - n * k.slices);
    const int H = a.H, W = a.W, K = a.K;
    const size_t px_n = (size_t)n * H * W;
    typedef const __attribute__((address_space(4))) int32_t* cptr_t;
    const int step = k.post.step_ptr ? *(cptr_t)(k.post.step_ptr) : 0;
    const ccdm_post_args post = post_resolve_run(k.post);
    // the step's coefficients: fetched once per block, ahead of everything (three floats of one table row)
    float st_al, st_cu, st_mode;
    {
        const float* row = post.step_table + (size_t)step * 4;
        st_al = row[0]; st_cu = row[1]; st_mode = row[2];
    }

    // ---- small loads first: GroupNorm operands, then the weight fragments (2 k-steps x hi|lo, resident), bias / un-scale of this lane's class ----
    GnPrefetch gpf;
    gn_prefetch(k.gn, true, n, 0, tid, 256, a.w, gpf);
    f16x8 wh[2], wl[2];
    {
        const char* wp = static_cast<const char*>(a.w) + ((size_t)lane << 4);
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            wh[j] = __builtin_bit_cast(f16x8, load16_global(wp + j * 2048));
            wl[j] = __builtin_bit_cast(f16x8, load16_global(wp + j * 2048 + 1024));
        }
    }
    float cb[KP], cs[KP];
#pragma unroll
    for (int c = 0; c < KP; ++c) { cb[c] = c < K ? a.bias[c] : 0.f; cs[c] = c < K ? k.wscale[c] : 0.f; }

    // ---- staging: item i of thread t = halo pixel t / 8 + 32 i, channel quad t % 8 ----
    const int ntile_sp = k.tiles_x * k.tiles_y;
    const int my_tiles = (ntile_sp - slice + k.slices - 1) / k.slices;
    const int q4 = tid & 7;
    f32x4 reg[HD_NITEM];
    unsigned okmask = 0;
    auto request = [&](const int tile) {
        const int ty = tile / k.tiles_x, tx = tile - ty * k.tiles_x;
        int hp = tid >> 3;
        int hy = hp / HD_HW, hx = hp - hy * HD_HW;
        okmask = 0;
#pragma unroll
        for (int i = 0; i < HD_NITEM; ++i) {
            const int iy = ty * HD_TH - 1 + hy, ix = tx * HD_TW - 1 + hx;
            const bool ok = ((unsigned)iy < (unsigned)H) & ((unsigned)ix < (unsigned)W) & (hp < HD_HP);
            const size_t p = px_n + (size_t)min(max(iy, 0), H - 1) * W + min(max(ix, 0), W - 1);       // clamped: unconditional loads
            reg[i] = load16_global(reinterpret_cast<const char*>(a.x + p * HD_C + 4 * q4));
            okmask |= (ok ? 1u : 0u) << i;
            hp += 32; hx += 32;
            if (hx >= HD_HW) { hx -= HD_HW; hy += 1; }
        }
    };
    if (my_tiles > 0) request(slice);
    __builtin_amdgcn_sched_barrier(0);
    gn_affine_block(k.gn, n, 0, gpf, reinterpret_cast<f64x2*>(tileb), ab);
    __syncthreads();
    // name the block-resident operands here: inside the tile loop the compiler cannot tell these loads from the loop's own prefetch and
    // would wait for EVERYTHING in flight (vmcnt(0): the next tile's halo request, just issued) at their first use of every tile
#pragma unroll
    for (int j = 0; j < 2; ++j) asm volatile("" : "+v"(wh[j]), "+v"(wl[j]));
#pragma unroll
    for (int c = 0; c < KP; ++c) asm volatile("" : "+v"(cb[c]), "+v"(cs[c]));
    asm volatile("" : "+v"(st_al), "+v"(st_cu), "+v"(st_mode));
    const int smode = (int)st_mode;
    auto commit = [&]() {
        constexpr float PS = ACT_PRESCALE;
        auto silu = [&](const float x) {     // x * sigmoid(x) * PS with v_exp_f32 / v_rcp_f32, as in ccdm_conv.hip
            return x * __builtin_amdgcn_rcpf(1.0f / PS + __builtin_amdgcn_exp2f(fmaf(x, -1.4426950408889634f, -4.0f)));
        };
        static_assert(ACT_PRESCALE == 16.0f, "the exp2 bias above is log2(ACT_PRESCALE)");
        int hp = tid >> 3;
        const float2 t0 = ab[4 * q4], t1 = ab[4 * q4 + 1], t2 = ab[4 * q4 + 2], t3 = ab[4 * q4 + 3];     // (re-read per tile: 8 registers not held across the matrix phase)
#pragma unroll
        for (int i = 0; i < HD_NITEM; ++i) {
            if (hp < HD_HP) {
                const f32x4 r = reg[i];
                float4 v = make_float4(fmaf(r[0], t0.x, t0.y), fmaf(r[1], t1.x, t1.y), fmaf(r[2], t2.x, t2.y), fmaf(r[3], t3.x, t3.y));
                v.x = silu(v.x); v.y = silu(v.y); v.z = silu(v.z); v.w = silu(v.w);
                const float lim = ((okmask >> i) & 1u) ? __builtin_inff() : 0.f;     // zero padding; nothing is clipped inside the image
                v.x = __builtin_amdgcn_fmed3f(v.x, -lim, lim); v.y = __builtin_amdgcn_fmed3f(v.y, -lim, lim);
                v.z = __builtin_amdgcn_fmed3f(v.z, -lim, lim); v.w = __builtin_amdgcn_fmed3f(v.w, -lim, lim);
                typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
                u32x2 hi, lo;
                unsigned h0, l0, h1, l1;
                split2_f16(v.x, v.y, h0, l0);
                split2_f16(v.z, v.w, h1, l1);
                hi[0] = h0; hi[1] = h1; lo[0] = l0; lo[1] = l1;
                char* d = tileb + hp * HD_PIX + 8 * q4;
                *reinterpret_cast<u32x2*>(d) = hi;
                *reinterpret_cast<u32x2*>(d + 2 * HD_C) = lo;
            }
            hp += 32;
        }
    };

    // sub-tiles of this wave: s = wave, wave + 4, wave + 8 (< 11)
    const int g = lane >> 5, qi = lane & 31;
    for (int it = 0; it < my_tiles; ++it) {
        const int tile = slice + it * k.slices;
        if (it > 0) __syncthreads();                 // the previous tile's Z values have been gathered
        // x_t of this thread's output pixel: requested here, consumed behind the matrix phase
        const size_t opix = px_n + (size_t)((tile / k.tiles_x) * HD_TH + (tid >> 5)) * W + (tile % k.tiles_x) * HD_TW + (tid & 31);
        const unsigned xtb = post.xt[opix];
        commit();
        __syncthreads();
        f32x16 acc[3];
#pragma unroll
        for (int u = 0; u < 3; ++u) {
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[u][r] = 0.f;
            const int s = wave + 4 * u;
            if (s < HD_SUB) {                        // uniform
                const char* pa = tileb + (s * 32 + qi) * HD_PIX + g * 16;
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const f16x8 ah = *reinterpret_cast<const f16x8*>(pa + 32 * j), al = *reinterpret_cast<const f16x8*>(pa + 32 * j + 2 * HD_C);
                    acc[u] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, wh[j], acc[u], 0, 0, 0);
                    acc[u] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, wl[j], acc[u], 0, 0, 0);
                    acc[u] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, wh[j], acc[u], 0, 0, 0);
                }
            }
        }
        __syncthreads();                             // every wave is done reading the A tile: Z may overwrite it
        // Z[q][n]: accumulator register r of sub-tile s = halo pixel 32 s + (r & 3) + 8 (r >> 2) + 4 g, column n = lane & 31 = tap * K + c
        if (qi < 9 * K) {
#pragma unroll
            for (int u = 0; u < 3; ++u) {
                const int s = wave + 4 * u;
                if (s < HD_SUB) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) zt[(s * 32 + (r & 3) + 8 * (r >> 2) + 4 * g) * ZP + qi] = acc[u][r];
                }
            }
        }
        // the next tile's halo request goes out here: its 44 registers do not have to live beside the three accumulators, and it
        // flies under the gather, the epilogue arithmetic and the loop-top barrier (three blocks per CU cover the rest)
        if (it + 1 < my_tiles) request(tile + k.slices);
        __syncthreads();
        // gather + epilogue: thread = output pixel (row tid / 32, column tid % 32 of the tile)
        {
            const int ty = tile / k.tiles_x, tx = tile - ty * k.tiles_x;
            const int py = tid >> 5, pxx = tid & 31;
            const float* z0 = zt + (py * HD_HW + pxx) * ZP;
            float x0[KP];
#pragma unroll
            for (int c = 0; c < KP; ++c) x0[c] = 0.f;
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) {      // taps ascending
                const float* zp = z0 + ((tap / 3) * HD_HW + tap % 3) * ZP + tap * K;
#pragma unroll
                for (int c = 0; c < KP; ++c) if (c < K) x0[c] += zp[c];
            }
#pragma unroll
            for (int c = 0; c < KP; ++c) x0[c] = c < K ? fmaf(x0[c], cs[c], cb[c]) : -INFINITY;       // cs is a power of two: exact product
            const size_t i = opix;
            (void)ty; (void)tx;
            if (a.logits_out) {
#pragma unroll
                for (int c = 0; c < KP; ++c) if (c < K) a.logits_out[i * K + c] = x0[c];
            }
            posterior_pixel_core<KP>(post, i, x0, step, st_al, st_cu, smode, (int)xtb);
        }
    }
}

static bool head_supported(int C, int K, int H, int W, int prec) {
    return prec == CCDM_PREC_F16X3 && C == HD_C && K >= 2 && 9 * K <= 32 && H > 0 && W > 0 && H % HD_TH == 0 && W % HD_TW == 0;
}

int launch_head(const ccdm_head_args& a, const ccdm_post_args& post, hipStream_t s) {
    CCDM_REQUIRE(a.x && a.stats && a.gamma && a.beta && a.w && a.bias, "head_posterior: null pointer");
    CCDM_REQUIRE(head_supported(a.C, a.K, a.H, a.W, CCDM_PREC_F16X3), "head_posterior: C=%d K=%d %dx%d is not built (ccdm_head_posterior_supported)", a.C, a.K,
                 a.H, a.W);
    CCDM_REQUIRE(a.N > 0 && a.slices >= 1, "head_posterior: N=%d slices=%d", a.N, a.slices);
    CCDM_REQUIRE(post.xt && post.step_table && post.xt_next, "head_posterior: epilogue without xt / step_table / xt_next");
    CCDM_REQUIRE(post.N == a.N && post.HW == a.H * a.W && post.K == a.K, "head_posterior: epilogue geometry (N=%d, HW=%d, K=%d) does not match the head's", post.N,
                 post.HW, post.K);
    HeadK k;
    k.a = a;
    k.post = post;
    k.gn = ccdm_conv_args{};
    k.gn.C0 = a.C; k.gn.stats0 = a.stats; k.gn.slices0 = a.slices; k.gn.gamma = a.gamma; k.gn.beta = a.beta; k.gn.eps = a.eps;
    k.gn.Hin = a.H; k.gn.Win = a.W; k.gn.N = a.N;
    k.slices = conv_slices(a.H, a.W, 1, false, 0);
    k.tiles_x = a.W / HD_TW; k.tiles_y = a.H / HD_TH;
    k.wscale = reinterpret_cast<const float*>(static_cast<const char*>(a.w) + 2 * 2048);
    const size_t lds = (size_t)HD_C * 8 + HD_A_BYTES;
    if (a.K == 2) hipLaunchKernelGGL(k_head<2>, dim3(a.N * k.slices), dim3(256), lds, s, k);
    else hipLaunchKernelGGL(k_head<4>, dim3(a.N * k.slices), dim3(256), lds, s, k);
    CCDM_CHECK_LAUNCH("head_posterior");
    return 0;
}

}  // namespace ccdm

extern "C" int ccdm_head_posterior_supported(int C, int K, int H, int W, int prec) { return ccdm::head_supported(C, K, H, W, prec) ? 1 : 0; }

extern "C" int ccdm_head_posterior(const ccdm_head_args* a, const ccdm_post_args* post, void* stream) {
    if (!a || !post) return ccdm::fail("ccdm_head_posterior: null args");
    return ccdm::launch_head(*a, *post, (hipStream_t)stream);
}

// Packed layout: [k-step j = 0, 1][hi|lo][64 lanes][8 halfs]; lane l (column n = l & 31 = tap * K + c, k-group g = l >> 5), element e:
// W[c][ci = 16 j + 8 g + e][tap] * 2^e(c), zero for n >= 9 K, split into fp16 hi + lo; then [K] floats 2^-e(c) / ACT_PRESCALE
// (one power of two per class: max |W| of the class over all taps and channels in [2^9, 2^10)).
extern "C" size_t ccdm_pack_head_weight(const float* oihw, int K, int Cin, void* out) {
    if (K < 2 || 9 * K > 32 || Cin != ccdm::HD_C) { ccdm::fail("pack_head: K=%d Cin=%d (need 9 K <= 32, Cin == 32)", K, Cin); return 0; }
    const size_t frag = 2 * 2048, total = frag + (size_t)K * sizeof(float);
    if (!out) return total;
    _Float16* o = static_cast<_Float16*>(out);
    float* sc = reinterpret_cast<float*>(static_cast<char*>(out) + frag);
    std::vector<float> mul(K, 1.0f);
    for (int c = 0; c < K; ++c) {
        float mx = 0.f;
        for (int i = 0; i < Cin * 9; ++i) mx = fmaxf(mx, fabsf(oihw[(size_t)c * Cin * 9 + i]));
        int e = 0;
        if (mx > 0.f && std::isfinite(mx)) { int ex; frexpf(mx, &ex); e = 10 - ex; }
        if (e > 60) e = 60;
        if (e < -60) e = -60;
        mul[c] = ldexpf(1.0f, e);
        sc[c] = ldexpf(1.0f, -e) / ccdm::ACT_PRESCALE;
    }
    for (int j = 0; j < 2; ++j)
        for (int l = 0; l < 64; ++l)
            for (int e = 0; e < 8; ++e) {
                const int nn = l & 31, ci = 16 * j + 8 * (l >> 5) + e;
                float v = 0.f;
                if (nn < 9 * K) {
                    const int tap = nn / K, c = nn % K;
                    v = oihw[((size_t)c * Cin + ci) * 9 + tap] * mul[c];
                }
                const _Float16 hi = (_Float16)v;
                const _Float16 lo = (_Float16)(v - (float)hi);
                const size_t base = ((size_t)j * 2) * 64 * 8;
                o[base + (size_t)l * 8 + e] = hi;
                o[base + 64 * 8 + (size_t)l * 8 + e] = lo;
            }
    return total;
}
