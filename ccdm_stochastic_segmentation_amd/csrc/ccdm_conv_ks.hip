// 3x3 conv of a FEW-PIXEL image ([GroupNorm -> SiLU ->] conv 3x3, stride 1 or 2, [+bias +emb +residual], fused 1x1 skip, + per-channel
// output statistics), F16X3 arithmetic: the ResBlock convs and the Downsample conv of the deepest U-Net levels
// (reference unet.py:242-262, :137-146; LIDC: the 8x8 stage — 13 launches per denoise step).
//
// Why another kernel.  At 8x8 pixels per sample a conv launch is ONE round of single-tile blocks and its time is the length of one
// block's dependent chain (tools/timeline_op.py on the general kernel, 128 -> 128 @ 8x8: 26 000 cycles for 2 300 cycles of matrix
// work: two channel chunks, each a weight-fragment round trip global -> registers -> LDS -> registers with two barriers, a commit
// pass that is mostly copying those fragments, 6 waves on 4 SIMDs).  Here the chain is: every load of the block issued up front ->
// GroupNorm table -> ONE commit of the whole halo tile (all input channels) -> ONE barrier -> matrix phase -> cross-wave reduction.
//
//   block  = one 8x8 output tile of one sample x one 32-channel output tile, 8 waves (2 per SIMD)
//   K split: the GEMM's K axis — (tap, 16-channel k-step) pairs, then the k-steps of the fused 1x1 skip segment — is dealt round-robin
//            to the 8 waves; each wave multiplies BOTH 32-pixel sub-tiles by its steps.  A weight fragment is therefore needed by exactly
//            one wave, which reads it straight from global memory (L2) into the registers the MFMA takes it from: no LDS staging of
//            B, no barrier for it, and the requests are on their way before the GroupNorm table is built.
//   A:       the (7*stride + 3)^2 halo pixels x all input channels, normalised, activated and split (fp16 hi | lo) ONCE into LDS
//            (pixel pitch 4*C + 16 bytes); the fused skip segment's raw input as 64 core pixels behind it.
//   epilogue: the 8 partial accumulators meet in LDS (aliasing the dead A tile), wave w adds them in fixed order for output row w of
//            the tile, adds bias (+emb) (+residual), stores, and leaves the tile's statistics partial — no atomics, fixed order.
// Same products as the general kernel (lo*hi, hi*lo, hi*hi per k-step), another summation order over K: results agree to fp32 rounding.
//
// Round 4 — NI output-channel tiles per block (16x16 images: LIDC's second-deepest stage, 21 launches per denoise step).  There a
// launch of the general kernel is ONE round of 1.5 blocks per CU and its time is one block's chain — per 32-channel chunk a commit
// (3 900 cycles), two barriers and a matrix phase (2 950), with the same halo tile normalised, activated and split by each of the
// three n-tile blocks of a pixel tile (tools/timeline_op.py: 96 -> 96 35 800 cycles, 224 -> 96 74 800).  With all n-tiles of an
// 8x8 pixel tile in ONE block the grid is 64 samples x 4 tiles = 256 blocks = one per CU: the tile is staged once, every weight
// fragment of the layer is read by exactly one wave of the block, a step feeds 2 x NI accumulators (6 NI matrix instructions per
// A / B fragment set), and the partial accumulators meet in LDS one n-tile at a time (the exchange buffer holds one).
#include "ccdm_common.h"
#include "ccdm_conv_common.h"

namespace ccdm {

struct ConvKS {
    ccdm_conv_args a;
    const float* wscale;           // [ntiles*32] powers of two undoing the weight / activation pre-scales
    int ntiles;                    // n-tiles of the packed weights
    int tiles_x, tiles_y;          // 8x8 output tiles per sample
    int C, SC;                     // input channels of the main / fused-skip segment
    int nks_m, nks_s;              // their 16-channel k-steps
    int pitch_m, pitch_s;          // LDS bytes per staged pixel (64 * nks + 16)
    int lgq_m, lgq_s;              // log2 of the lanes one staged pixel takes (channel quads rounded up to 16 / 32 / 64)
    int nit_m, nit_s;              // staging items per thread
    int skip_off;                  // LDS offset of the skip pixels within the staging region
    int nstep;                     // 9 * nks_m + nks_s
    int red_off;                   // LDS offset (within the staging region) of the statistics fold buffer
};

// phase timeline of one block (CCDM_ABLATION builds, diagnostic bit 16 of prec >> 8; read back with ccdm_debug_read_timeline)
#ifdef CCDM_ABLATION
__device__ unsigned long long g_timeline_ks[64];
static bool g_ks_stamped = false;
#define KS_STAMP(slot) do { if (tl_on && tid == 0 && tl < 60) g_timeline_ks[tl++] = ((unsigned long long)(slot) << 56) | (__builtin_amdgcn_s_memtime() & 0x00ffffffffffffffull); } while (0)
#else
#define KS_STAMP(slot) do { } while (0)
#endif

constexpr int KS_NT = 512, KS_NW = 8, KS_PART_BYTES = 8 * 2 * 4 * 64 * 16;
// weight-fragment sets in flight per wave (one set = NI hi|lo pairs = 8 NI registers): ~80-96 registers of queue whatever NI
constexpr int ks_bq(int ni) { return ni == 1 ? 10 : (ni == 2 ? 6 : (ni == 3 ? 4 : 2)); }

// GNACT: the input is GroupNorm'ed and SiLU'ed on load (ResBlock convs) / staged raw (Downsample) — the only two combinations this
// kernel is built for.  NITM / NITS: staging items per thread of the halo tile / the skip pixels the instantiation has registers for.
template <int STRIDE, bool GNACT, int NITM, int NITS, int NI = 1>
__global__ __launch_bounds__(KS_NT, NI == 1 ? 2 : 1) void k_conv_ks(const ConvKS k) {
    constexpr int HWt = 7 * STRIDE + 3, HP = HWt * HWt;
    constexpr int NIT = NITM + NITS;
    constexpr int KS_BQ = ks_bq(NI);
    warm_kernargs<sizeof(ConvKS)>();
    const ccdm_conv_args& a = k.a;
    extern __shared__ __attribute__((aligned(16))) char smem_ks[];
    constexpr bool has_gn = GNACT;
    const int C = k.C;
    float2* ab = reinterpret_cast<float2*>(smem_ks);                                   // [C] (scale, shift), GroupNorm only
    char* const region = smem_ks + (has_gn ? (size_t)C * 8 : 0);                         // A tile (+ skip pixels) | statistics exchange | partials
    double* const red = reinterpret_cast<double*>(region + k.red_off);

    const int tid = threadIdx.x, lane = tid & 63;
    const int g = __builtin_amdgcn_readfirstlane(tid >> 6);                              // wave = K group
#ifdef CCDM_ABLATION
    const bool tl_on = ((a.prec >> 8) & 16) && blockIdx.x == gridDim.x / 2 && blockIdx.y == 0;
    int tl = 0;
#endif
    KS_STAMP(1);
    const int ntile_sp = k.tiles_x * k.tiles_y;
    // several tiles per sample: block b runs on XCD b % 8 — give each XCD a contiguous range of (sample, tile) pairs so that the tiles
    // of a sample (shared halo rows, the same statistics partials) meet in one L2 (speed only)
    int bid = blockIdx.x;
    if (ntile_sp > 1 && (gridDim.x & 7) == 0) bid = (blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3);
    const int n = __builtin_amdgcn_readfirstlane(bid / ntile_sp), tile = __builtin_amdgcn_readfirstlane(bid - n * ntile_sp);
    const int ty = tile / k.tiles_x, tx = tile - ty * k.tiles_x;
    // the step counter and this sample's table row: SCALAR loads, requested before any vector memory request.  (As vector loads behind
    // the halo and fragment requests their wait was a vmcnt(0): the whole halo round trip sat in front of the epilogue constants, and
    // the residual requests behind them cost a second full round trip before the GroupNorm table.)
    typedef const __attribute__((address_space(4))) int32_t* ks_cptr_t;
    int step = 0, row0 = 0;
    if (a.step_ptr) step = *(ks_cptr_t)(a.step_ptr);
    if (a.emb_row_of_sample) row0 = ((ks_cptr_t)(a.emb_row_of_sample))[n];
    const int emb_row = row0 + step;
    const int nt = blockIdx.y * NI;                                                     // first n-tile of this block
    const int oy0 = ty * 8, ox0 = tx * 8;
    const int Hin = a.Hin, Win = a.Win;

    // ---- 1. GroupNorm operands: thread c < C requests gamma, beta and the first 4 statistics partials of channel c (few-pixel images
    //         have 1-4 slices; a clamped 16-slice prefetch by all 512 threads, as the general kernel does, is 128 redundant 1 KB requests
    //         through a vector-memory front end that moves ~40 B/clk — 2 000 cycles of this block's chain) ----
    const bool gn_thread = has_gn && tid < C;                            // (waves beyond C / 64 skip the requests: wave-uniform)
    f64x2 gsl[4];
    float g_gamma = 1.f, g_beta = 0.f;
    int gS = 1;
    unsigned g_stride = 0;
    const char* g_base = nullptr;
    if (gn_thread) {
        g_base = gn_channel_row(a, n, tid, gS, g_stride);
        const unsigned last = (unsigned)(gS - 1) * g_stride;
#pragma unroll
        for (int u = 0; u < 4; ++u) gsl[u] = *reinterpret_cast<const f64x2*>(g_base + min((unsigned)u * g_stride, last));
        g_gamma = a.gamma[tid];
        g_beta = a.beta[tid];
    }

    KS_STAMP(20);
    // ---- 2. halo requests: item i of wave g covers staged pixels (g + 8 i) * PPW .. + PPW - 1, lane = (pixel within the item, channel quad) ----
    const int lgq_m = k.lgq_m, lgq_s = k.lgq_s;
    const int sub_m = lane >> lgq_m, q_m = lane & ((1 << lgq_m) - 1);
    const int sub_s = lane >> lgq_s, q_s = lane & ((1 << lgq_s) - 1);
    const int cm = 4 * q_m, cs = 4 * q_s;
    const bool cok_m = cm < C, cok_s = cs < k.SC;
    // per-lane source of this lane's channel quad (item-invariant): tensor base of sample n, channels of that tensor, first channel within it
    const size_t in_px = (size_t)Hin * Win, out_px = (size_t)a.Hout * a.Wout;
    const bool sec_m = a.C1 > 0 && cm >= a.C0 && cok_m;
    const int Cs_m = sec_m ? a.C1 : a.C0, cb_m = cok_m ? (sec_m ? cm - a.C0 : cm) : 0;
    const char* const src_m = reinterpret_cast<const char*>((sec_m ? a.in1 : a.in0) + (size_t)n * in_px * Cs_m + cb_m);
    const bool sec_s = a.SC1 > 0 && cs >= a.SC0 && cok_s;
    const int Cs_s = sec_s ? a.SC1 : a.SC0, cb_s = cok_s ? (sec_s ? cs - a.SC0 : cs) : 0;
    const char* const src_s = k.nks_s ? reinterpret_cast<const char*>((sec_s ? a.skip1 : a.skip0) + (size_t)n * out_px * Cs_s + cb_s) : src_m;
    const int nit_m = k.nit_m, nit_s = k.nit_s;
    f32x4 reg[NIT];                          // [0, NITM): halo items, [NITM, NIT): skip items
    unsigned okmask = 0;                     // bit i: halo item i lies inside the image (else zero padding)
    {
        // halo pixel of item i: hp = (g + 8 i) * PPW + sub, advancing by 8 PPW per item — (hy, hx) are carried incrementally (no
        // per-item division; 14 vector instructions per request, where the closed form took 85: at 8x8 the prologue IS the kernel)
        const int dhp = KS_NW << (6 - lgq_m);
        const int dhy = dhp / HWt, dhx = dhp - dhy * HWt;                // uniform
        const int hp0 = (g << (6 - lgq_m)) + sub_m;
        int hy = hp0 / HWt, hx = hp0 - hy * HWt;
        const int oyb = oy0 * STRIDE - 1, oxb = ox0 * STRIDE - 1;
        const unsigned cs4 = (unsigned)Cs_m << 2;
#pragma unroll
        for (int i = 0; i < NITM; ++i) {
            if (i < nit_m) {                 // uniform
                const int iy = oyb + hy, ix = oxb + hx;                  // (items beyond the tile request a clamped pixel and are never stored)
                const bool ok = ((unsigned)iy < (unsigned)Hin) & ((unsigned)ix < (unsigned)Win);
                const int iyc = min(max(iy, 0), Hin - 1), ixc = min(max(ix, 0), Win - 1);
                reg[i] = load16_global(src_m + (size_t)((unsigned)(iyc * Win + ixc) * cs4));
                okmask |= (ok ? 1u : 0u) << i;
                hx += dhx; hy += dhy;
                if (hx >= HWt) { hx -= HWt; hy += 1; }
            }
        }
        if constexpr (NITS > 0) {
            const unsigned ss4 = (unsigned)Cs_s << 2;
            const int dpx = KS_NW << (6 - lgq_s);
            int px = (g << (6 - lgq_s)) + sub_s;
#pragma unroll
            for (int j = 0; j < NITS; ++j) {
                if (j < nit_s) {
                    const int pxc = min(px, 63);
                    reg[NITM + j] = load16_global(src_s + (size_t)((unsigned)((oy0 + (pxc >> 3)) * a.Wout + ox0 + (pxc & 7)) * ss4));
                    px += dpx;
                }
            }
        }
    }

    KS_STAMP(2);
    // ---- 3. this wave's weight fragments: steps s = g, g + 8, ...; the first batch is requested now ----
    const int nstep = k.nstep, nmain = 9 * k.nks_m;
    const int nsw = (nstep - g + KS_NW - 1) / KS_NW;                                     // steps of this wave (>= 1: nstep >= 9)
    auto frag_ptr = [&](const int j) {       // fragment pair of this wave's step j (clamped), uniform
        const int s = min(g + KS_NW * j, nstep - 1);
        const bool sk = s >= nmain;
        const int ks = sk ? s - nmain : s / 9, tap = sk ? 0 : s - 9 * ks;
        const char* const base = static_cast<const char*>(sk ? a.skip_w : a.w);
        return base + (((size_t)(tap * (sk ? k.nks_s : k.nks_m) + ks) * k.ntiles + nt) << 11);
    };
    // The vector-memory front end of a CU takes ~40 B/clk from L2: the 147 KB of fragments a 128-channel block needs are ~3 700 cycles of
    // streaming.  Requested in one burst they stall every wave at issue for that long; so only the first pairs go out here, the rest
    // one pair behind each committed halo item (the front end streams them while the waves do the commit's arithmetic).
    f32x4 bh[KS_BQ][NI], bl[KS_BQ][NI];       // (the n-tiles of a step are adjacent 2 KB slabs: hi | lo)
    const unsigned lane16 = (unsigned)lane << 4;
    constexpr int KS_B0 = NI == 1 ? 2 : 1;
    auto load_set = [&](const int slot, const int j) {
        const char* p = frag_ptr(j);
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) {
            bh[slot][ni] = load16_uniform_base(p, lane16 + 2048u * ni);
            bl[slot][ni] = load16_uniform_base(p, lane16 + 2048u * ni + 1024u);
        }
    };
#pragma unroll
    for (int jj = 0; jj < KS_B0; ++jj) load_set(jj, jj);
    KS_STAMP(22);

    // ---- 4. epilogue constants and residual of the output row this wave will finish (row g of the tile).  One n-tile per block: requested
    //         here, early.  Several: behind the matrix phase (their 7 NI registers would not fit beside 2 NI accumulators and the fragment
    //         queue — a spilled load destination is a vmcnt(0) in the middle of the prologue), arriving under the first exchange pass ----
    const int co = nt * 32 + (lane & 31);                               // channel of n-tile 0; n-tile ni: + 32 ni
    const float* const dummyf = reinterpret_cast<const float*>(a.w);
    float raw_bias[NI], raw_emb[NI], wsc[NI];
    // reducer geometry: wave g finishes tile row g; lane = (half h = lane >> 5: pixels 4h .. 4h + 3 of the row, channel lane & 31)
    const size_t obase = (((size_t)n * a.Hout + (oy0 + g)) * a.Wout + ox0 + 4 * (lane >> 5)) * a.Cout + co;
    float rs[NI][4];
    auto request_epilogue_operands = [&]() {
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) {
            raw_bias[ni] = *(a.bias ? a.bias + co + 32 * ni : dummyf);
            raw_emb[ni] = *(a.emb_off >= 0 ? a.emb_table + (size_t)emb_row * a.emb_stride + a.emb_off + co + 32 * ni : dummyf);
            wsc[ni] = k.wscale[co + 32 * ni];
        }
#pragma unroll
        for (int ni = 0; ni < NI; ++ni)
#pragma unroll
            for (int j = 0; j < 4; ++j) rs[ni][j] = 0.f;
        if (a.resid) {
#pragma unroll
            for (int ni = 0; ni < NI; ++ni)
#pragma unroll
                for (int j = 0; j < 4; ++j) rs[ni][j] = a.resid[obase + (size_t)j * a.Cout + 32 * ni];
        }
    };
    if constexpr (NI == 1) request_epilogue_operands();

    KS_STAMP(3);
    // ---- 5. GroupNorm (scale, shift) table (the staging region is free until the commit) ----
    if (has_gn) {
        f64x2* scratch = reinterpret_cast<f64x2*>(region);
        if (gn_thread) {
            f64x2 acc2 = {0.0, 0.0};
#pragma unroll
            for (int u = 0; u < 4; ++u) {                                // ascending slices; selects, not branches
                acc2[0] += u < gS ? gsl[u][0] : 0.0;
                acc2[1] += u < gS ? gsl[u][1] : 0.0;
            }
            for (int sl = 4; sl < gS; ++sl) acc2 += *reinterpret_cast<const f64x2*>(g_base + (unsigned)sl * g_stride);
            scratch[tid] = acc2;
        }
        __syncthreads();
        if (gn_thread) {
            const int cpg = C / 32, c_lo = (tid / cpg) * cpg;
            f64x2 acc2 = {0.0, 0.0};
            for (int j = 0; j < cpg; ++j) acc2 += scratch[c_lo + j];     // ascending channels of the group
            GnParams gp;
            gp.gamma = g_gamma; gp.beta = g_beta; gp.film_scale = 0.f; gp.film_shift = 0.f;
            ab[tid] = gn_finalize(a, gp, acc2[0], acc2[1]);
        }
        __syncthreads();
    }

    KS_STAMP(4);
    // ---- 6. commit: registers -> affine -> SiLU -> fp16 hi | lo -> LDS; one weight-fragment pair requested behind every item ----
    {
        constexpr float PS = ACT_PRESCALE;
        float2 t0 = make_float2(1.f, 0.f), t1 = t0, t2 = t0, t3 = t0;
        if (GNACT && cok_m) { t0 = ab[cm]; t1 = ab[cm + 1]; t2 = ab[cm + 2]; t3 = ab[cm + 3]; }
        auto silu = [&](const float x) {     // x * sigmoid(x) * PS with v_exp_f32 / v_rcp_f32, as in ccdm_conv.hip
            return x * __builtin_amdgcn_rcpf(1.0f / PS + __builtin_amdgcn_exp2f(fmaf(x, -1.4426950408889634f, -4.0f)));
        };
        static_assert(ACT_PRESCALE == 16.0f, "the exp2 bias above is log2(ACT_PRESCALE)");
        typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
        auto store_split = [&](char* d, const int lo_off, float4 v, const bool ok) {
            const float lim = ok ? __builtin_inff() : 0.f;               // padding / padded channels -> 0 (nothing is clipped inside the image)
            v.x = __builtin_amdgcn_fmed3f(v.x, -lim, lim); v.y = __builtin_amdgcn_fmed3f(v.y, -lim, lim);
            v.z = __builtin_amdgcn_fmed3f(v.z, -lim, lim); v.w = __builtin_amdgcn_fmed3f(v.w, -lim, lim);
            u32x2 hi, lo;
            unsigned h0, l0, h1, l1;
            split2_f16(v.x, v.y, h0, l0);
            split2_f16(v.z, v.w, h1, l1);
            hi[0] = h0; hi[1] = h1; lo[0] = l0; lo[1] = l1;
            *reinterpret_cast<u32x2*>(d) = hi;
            *reinterpret_cast<u32x2*>(d + lo_off) = lo;
        };
        auto next_pair = [&](const int jj) {
            if (jj < KS_BQ) load_set(jj < KS_BQ ? jj : 0, jj);
        };
        const int Cm2 = 32 * k.nks_m, Cs2 = 32 * k.nks_s;               // bytes of a pixel's hi plane
        const bool wr_m = 4 * q_m < 16 * k.nks_m, wr_s = 4 * q_s < 16 * k.nks_s;      // lanes beyond the padded channel count write nothing
        const int dhp = KS_NW << (6 - lgq_m);
        int hp = (g << (6 - lgq_m)) + sub_m;
        char* dm = region + hp * k.pitch_m + 8 * q_m;
        const int ddm = dhp * k.pitch_m;
#pragma unroll
        for (int i = 0; i < NITM; ++i) {
            if (i < nit_m) {
                const f32x4 r = reg[i];
                float4 v = make_float4(r[0], r[1], r[2], r[3]);
                if (GNACT) {
                    v.x = fmaf(v.x, t0.x, t0.y); v.y = fmaf(v.y, t1.x, t1.y); v.z = fmaf(v.z, t2.x, t2.y); v.w = fmaf(v.w, t3.x, t3.y);
                    v.x = silu(v.x); v.y = silu(v.y); v.z = silu(v.z); v.w = silu(v.w);
                } else { v.x *= PS; v.y *= PS; v.z *= PS; v.w *= PS; }
                if (wr_m && hp < HP) store_split(dm, Cm2, v, cok_m && ((okmask >> i) & 1u));
                hp += dhp; dm += ddm;
            }
            next_pair(KS_B0 + i);
            KS_STAMP(30 + i);
        }
        if constexpr (NITS > 0) {
            const int dpx = KS_NW << (6 - lgq_s);
            int px = (g << (6 - lgq_s)) + sub_s;
#pragma unroll
            for (int j = 0; j < NITS; ++j) {
                if (j < nit_s) {
                    const f32x4 r = reg[NITM + j];
                    const float4 v = make_float4(r[0] * PS, r[1] * PS, r[2] * PS, r[3] * PS);
                    if (wr_s && px < 64) store_split(region + k.skip_off + px * k.pitch_s + 8 * q_s, Cs2, v, cok_s);
                    px += dpx;
                }
                next_pair(KS_B0 + NITM + j);
            }
        }
#pragma unroll
        for (int jj = KS_B0 + NIT; jj < KS_BQ; ++jj) next_pair(jj);     // (instantiations with fewer items than fragment pairs)
    }
    KS_STAMP(5);
    __syncthreads();
    KS_STAMP(6);

    // ---- 7. matrix phase: this wave's steps, both sub-tiles, fragments from registers (B) and LDS (A) ----
    f32x16 acc[2][NI];
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;
    int abase[2], sbase[2];
#pragma unroll
    for (int mi = 0; mi < 2; ++mi) {
        const int p = mi * 32 + (lane & 31);
        abase[mi] = (((p >> 3) * HWt + (p & 7)) * STRIDE) * k.pitch_m + (lane >> 5) * 16;
        sbase[mi] = k.skip_off + p * k.pitch_s + (lane >> 5) * 16;
    }
    const int lo_m = 32 * k.nks_m, lo_s = 32 * k.nks_s;
    static_assert(KS_BQ % 2 == 0, "static step parity");
    // step j -> LDS byte offset of its A fragments (relative to abase / sbase) and the hi -> lo distance; uniform
    auto a_off = [&](const int j, int& lo_off, bool& sk) {
        const int s = g + KS_NW * j;
        sk = s >= nmain;
        const int ks = sk ? s - nmain : s / 9, tap = sk ? 4 : s - 9 * ks;
        lo_off = sk ? lo_s : lo_m;
        return sk ? 32 * ks : ((tap / 3) * HWt + (tap % 3)) * k.pitch_m + 32 * ks;
    };
    f16x8 ah[2][2], al[2][2];                // two fragment sets: step j + 1 is requested before the MFMAs of step j
    auto frag_load_a = [&](const int buf, const int j) {
        int lo_off;
        bool sk;
        const int toff = a_off(min(j, nsw - 1), lo_off, sk);
#pragma unroll
        for (int mi = 0; mi < 2; ++mi) {
            const char* pa = region + (sk ? sbase[mi] : abase[mi]) + toff;
            ah[buf][mi] = *reinterpret_cast<const f16x8*>(pa);
            al[buf][mi] = *reinterpret_cast<const f16x8*>(pa + lo_off);
        }
    };
    frag_load_a(0, 0);
    // one batch of (up to) KS_BQ steps, straight-line; REFILL: each slot's fragments of the next batch are requested behind its MFMAs.
    // The first batch is peeled out of the loop over batches: inside a loop the compiler must assume that any fragment register may
    // hold a refill of the previous trip and waits for (nearly) everything in flight before the first MFMA; peeled, step j waits for
    // pair j only (the pairs were requested in step order).
    auto batch = [&](const int j0) {
#pragma unroll
        for (int jj = 0; jj < KS_BQ; ++jj) {
            const int j = j0 + jj;
            if (j < nsw) {                   // uniform
                const int cur = jj & 1;      // (KS_BQ is even: the parity of a step is static after unrolling)
                frag_load_a(cur ^ 1, j + 1);
                __builtin_amdgcn_sched_barrier(0);                       // keep the requests in front of the MFMAs
#pragma unroll
                for (int ni = 0; ni < NI; ++ni) {
                    const f16x8 wh = __builtin_bit_cast(f16x8, bh[jj][ni]), wl = __builtin_bit_cast(f16x8, bl[jj][ni]);
#pragma unroll
                    for (int mi = 0; mi < 2; ++mi) acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[cur][mi], wh, acc[mi][ni], 0, 0, 0);
#pragma unroll
                    for (int mi = 0; mi < 2; ++mi) acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[cur][mi], wl, acc[mi][ni], 0, 0, 0);
#pragma unroll
                    for (int mi = 0; mi < 2; ++mi) acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[cur][mi], wh, acc[mi][ni], 0, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
                if (j + KS_BQ < nsw) load_set(jj, j + KS_BQ);       // the slot's fragments of the next batch (wide inputs)
            }
        }
    };
    batch(0);
    for (int j0 = KS_BQ; j0 < nsw; j0 += KS_BQ) batch(j0);

    if constexpr (NI > 1) request_epilogue_operands();
    // ---- 8. cross-wave reduction, one n-tile at a time: partial [wave][sub-tile][row quad][lane] x 16 B, then wave g finishes tile row g ----
    KS_STAMP(7);
    __syncthreads();                         // every wave is done reading the A tile
    KS_STAMP(8);
    float t1[NI], t2[NI];
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) {
        if (ni > 0) __syncthreads();         // the previous n-tile's partials have been consumed
        {
            f32x4* part = reinterpret_cast<f32x4*>(region);
#pragma unroll
            for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                for (int rq = 0; rq < 4; ++rq) {
                    f32x4 v;
                    v[0] = acc[mi][ni][4 * rq]; v[1] = acc[mi][ni][4 * rq + 1]; v[2] = acc[mi][ni][4 * rq + 2]; v[3] = acc[mi][ni][4 * rq + 3];
                    part[((g * 2 + mi) * 4 + rq) * 64 + lane] = v;
                }
        }
        __syncthreads();
        if (ni == 0) KS_STAMP(9);
        t1[ni] = 0.f; t2[ni] = 0.f;
        {
            // accumulator register 4 rq + j of sub-tile mi holds pixel mi*32 + 8 rq + 4 (lane >> 5) + j, i.e. tile row 4 mi + rq, column 4 (lane >> 5) + j
            const f32x4* part = reinterpret_cast<const f32x4*>(region) + ((g >> 2) * 4 + (g & 3)) * 64 + lane;
            f32x4 v = part[0];
#pragma unroll
            for (int w = 1; w < KS_NW; ++w) v += part[w * 8 * 64];                            // fixed order: wave 0 + 1 + ... + 7
            float add = a.bias ? raw_bias[ni] : 0.f;
            if (a.emb_off >= 0) add += raw_emb[ni];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float o = fmaf(v[j], wsc[ni], add);                                           // wsc is a power of two: exact product
                if (a.resid) o += rs[ni][j];
                a.out[obase + (size_t)j * a.Cout + 32 * ni] = o;
                t1[ni] += o;
                t2[ni] = fmaf(o, o, t2[ni]);
            }
        }
    }
    KS_STAMP(10);
    if (a.out_stats) {
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) {
            double v1 = (double)t1[ni], v2 = (double)t2[ni];
            v1 += __shfl_xor(v1, 32);
            v2 += __shfl_xor(v2, 32);
            if (lane < 32) { red[((ni * KS_NW + g) * 32 + lane) * 2] = v1; red[((ni * KS_NW + g) * 32 + lane) * 2 + 1] = v2; }
        }
        __syncthreads();
        if (tid < 32 * NI) {
            const int ni = tid >> 5, l = tid & 31;
            double s1 = 0.0, s2 = 0.0;
            for (int w = 0; w < KS_NW; ++w) { s1 += red[((ni * KS_NW + w) * 32 + l) * 2]; s2 += red[((ni * KS_NW + w) * 32 + l) * 2 + 1]; }
            double* o = a.out_stats + (((size_t)n * ntile_sp + tile) * a.Cout + (nt + ni) * 32 + l) * 2;
            o[0] = s1; o[1] = s2;
        }
    }
    KS_STAMP(11);
#ifdef CCDM_ABLATION
    if (tl_on && tid == 0) g_timeline_ks[63] = tl;
#endif
}

static int pow2_lg_quads(int nks) {          // lanes per staged pixel: 4 nks channel quads rounded up to 16 / 32 / 64
    const int q = 4 * nks;
    return q <= 16 ? 4 : (q <= 32 ? 5 : 6);
}

// n-tiles per block: a rule of the image size and the channel count, never of N (the statistics partials follow the tiling).
//   <= 128 pixels (one or two 8x8 tiles per sample: LIDC's 8x8 stage, 64 blocks per n-tile at batch 64): 1 — the n-tiles are the grid
//   <= 256 pixels (16x16: 4 tiles per sample, 256 pixel tiles at batch 64 = one per CU): every n-tile of the layer, at most 4
static int conv_ks_ni(const ccdm_conv_args& a) {
    if (a.Hout * a.Wout <= 128) return 1;
    const int nt = a.Cout / 32;
    for (int ni = 4; ni > 1; --ni)
        if (nt % ni == 0) return ni;
    return 1;
}

// geometry and resources of a launch; false = not for this kernel
static bool conv_ks_plan(const ccdm_conv_args& a, ConvKS& k, size_t& lds, int& nit_max, int& NI) {
#ifdef CCDM_ABLATION
    if ((a.prec & ~(16 << 8)) != CCDM_PREC_F16X3) return false;           // (the timeline bit is this kernel's too)
#else
    if (a.prec != CCDM_PREC_F16X3) return false;                          // (a diagnostic bit in prec >> 8: the general kernel)
#endif
    if (a.ksize != 3 || a.up || a.film || (a.stride != 1 && a.stride != 2)) return false;
    if ((a.stats0 != nullptr) != (a.act == CCDM_ACT_SILU)) return false;  // built for GroupNorm + SiLU on load, or neither
    const int max_area = exp_env("CCDM_KS_MAX_AREA") ? exp_env("CCDM_KS_MAX_AREA") : 256;      // (A/B hook of experiments builds: 128 = round 3)
    if (a.Hout % 8 || a.Wout % 8 || a.Hout * a.Wout > max_area) return false;   // a rule of the geometry, never of N
    if (a.Cout % 32) return false;
    const int C = a.C0 + a.C1, SC = a.skip0 ? a.SC0 + a.SC1 : 0;
    if (C > 256 || SC > 256 || (a.stride == 2 && a.skip0)) return false;
    NI = conv_ks_ni(a);
    // instantiations with several n-tiles per block: the ResBlock convs (GroupNorm + SiLU on load, stride 1) and Downsample (raw, stride 2)
    if (NI > 1 && ((a.stride == 1) != (a.stats0 != nullptr))) return false;
    k.a = a;
    k.C = C; k.SC = SC;
    k.nks_m = cdiv(C, 16); k.nks_s = cdiv(SC, 16);
    k.pitch_m = 64 * k.nks_m + 16; k.pitch_s = 64 * k.nks_s + 16;
    k.lgq_m = pow2_lg_quads(k.nks_m); k.lgq_s = SC ? pow2_lg_quads(k.nks_s) : 6;
    const int HWt = 7 * a.stride + 3, HP = HWt * HWt;
    k.nit_m = cdiv(cdiv(HP, 64 >> k.lgq_m), KS_NW);
    k.nit_s = SC ? cdiv(64 >> (6 - k.lgq_s), KS_NW) : 0;
    // instantiations: stride 1: (13 halo items, no skip) or (8, 8); stride 2: (20, none)
    nit_max = a.stride == 1 ? (SC ? 8 : 13) : 20;
    if (k.nit_m > nit_max || k.nit_s > (a.stride == 1 && SC ? 8 : 0)) return false;
    k.skip_off = (HP * k.pitch_m + 15) / 16 * 16;
    k.nstep = 9 * k.nks_m + k.nks_s;
    k.tiles_x = a.Wout / 8; k.tiles_y = a.Hout / 8;
    size_t stage = (size_t)k.skip_off + (SC ? (size_t)64 * k.pitch_s : 0);
    stage = (stage + 15) / 16 * 16;
    if (stage < (size_t)KS_PART_BYTES) stage = KS_PART_BYTES;
    const size_t ex = (size_t)(C > KS_NT ? C : KS_NT) * 16;               // gn_affine_block's exchange
    if (a.stats0 && stage < ex) stage = ex;
    k.red_off = (int)stage;
    lds = (a.stats0 ? (size_t)C * 8 : 0) + stage + (size_t)NI * KS_NW * 32 * 16;
    return lds <= 160 * 1024;
}

bool conv_ks_eligible(const ccdm_conv_args& a) {
    ConvKS k;
    size_t lds;
    int nit, ni;
    return conv_ks_plan(a, k, lds, nit, ni);
}

// statistics slices a launch of this kernel leaves: one per 8x8 tile
int conv_ks_slices(const ccdm_conv_args& a) { return (a.Hout / 8) * (a.Wout / 8); }

template <int NI>
static void launch_conv_ks_ni(const ConvKS& k, dim3 grid, dim3 block, size_t lds, hipStream_t s) {
    const ccdm_conv_args& a = k.a;
    const bool gnact = a.stats0 != nullptr;
    if (a.stride == 1) {
        if (a.skip0) hipLaunchKernelGGL((k_conv_ks<1, true, 8, 8, NI>), grid, block, lds, s, k);
        else hipLaunchKernelGGL((k_conv_ks<1, true, 13, 0, NI>), grid, block, lds, s, k);
    } else {
        hipLaunchKernelGGL((k_conv_ks<2, false, 20, 0, NI>), grid, block, lds, s, k);
    }
    (void)gnact;
}

int launch_conv_ks(const ccdm_conv_args& a, int ntiles, const float* wscale, hipStream_t s) {
    ConvKS k;
    size_t lds;
    int nit, NI;
    if (!conv_ks_plan(a, k, lds, nit, NI)) return fail("conv_ks: geometry not built");
    k.wscale = wscale;
    k.ntiles = ntiles;
    const dim3 grid(a.N * k.tiles_x * k.tiles_y, a.Cout / (32 * NI)), block(KS_NT);
#ifdef CCDM_ABLATION
    if ((a.prec >> 8) & 16) g_ks_stamped = true;
#endif
    if (NI == 4) { launch_conv_ks_ni<4>(k, grid, block, lds, s); return 0; }
    if (NI == 3) { launch_conv_ks_ni<3>(k, grid, block, lds, s); return 0; }
    if (NI == 2) { launch_conv_ks_ni<2>(k, grid, block, lds, s); return 0; }
    const bool gnact = a.stats0 != nullptr;
    if (a.stride == 1) {
        if (a.skip0) {
            if (gnact) hipLaunchKernelGGL((k_conv_ks<1, true, 8, 8>), grid, block, lds, s, k);
            else hipLaunchKernelGGL((k_conv_ks<1, false, 8, 8>), grid, block, lds, s, k);
        } else {
            if (gnact) hipLaunchKernelGGL((k_conv_ks<1, true, 13, 0>), grid, block, lds, s, k);
            else hipLaunchKernelGGL((k_conv_ks<1, false, 13, 0>), grid, block, lds, s, k);
        }
    } else {
        if (gnact) hipLaunchKernelGGL((k_conv_ks<2, true, 20, 0>), grid, block, lds, s, k);
        else hipLaunchKernelGGL((k_conv_ks<2, false, 20, 0>), grid, block, lds, s, k);
    }
    return 0;
}

#ifdef CCDM_ABLATION
bool conv_ks_timeline_read(unsigned long long* host, int n) {
    if (!g_ks_stamped) return false;
    g_ks_stamped = false;
    for (int i = 0; i < n; ++i) host[i] = 0;
    unsigned long long tmp[64];
    if (hipMemcpyFromSymbol(tmp, HIP_SYMBOL(g_timeline_ks), sizeof(tmp), 0, hipMemcpyDeviceToHost) != hipSuccess) return false;
    for (int i = 0; i < 63 && i < n; ++i) host[i] = tmp[i];
    if (n >= 1024) host[1023] = tmp[63];
    return true;
}
#endif

}  // namespace ccdm
