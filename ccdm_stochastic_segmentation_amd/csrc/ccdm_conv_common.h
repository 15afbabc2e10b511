// Device helpers of the conv kernel (ccdm_conv.hip).
#pragma once
#include "ccdm_common.h"

namespace ccdm {

static constexpr float ACT_PRESCALE = 16.0f;       // F16X3 activation pre-scale (power of two)

// Load 16 bytes from global memory at (wave-uniform pointer + 32-bit per-lane byte offset).  The pointer is passed
// through readfirstlane so the compiler must keep it in scalar registers and select the saddr + voffset addressing
// form: no 64-bit vector adds, one VGPR of address per load.
typedef const __attribute__((address_space(1))) char* gptr_t;
__device__ __forceinline__ f32x4 load16_uniform_base(const char* base, unsigned voff) {
    const unsigned long long v = reinterpret_cast<unsigned long long>(base);
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
    gptr_t g = reinterpret_cast<gptr_t>(((unsigned long long)hi << 32) | lo);
    return *reinterpret_cast<const __attribute__((address_space(1))) f32x4*>(g + voff);
}

// Load 16 bytes of GLOBAL memory through a pointer whose address space the compiler cannot see (rebuilt from integers / register
// lanes).  Left generic it becomes flat_load, which also counts in LGKM_CNT: the wave's next LDS wait (s_waitcnt lgkmcnt) then waits
// for this memory round trip too, and with a flat load pending the waitcnt pass can no longer count vector loads in order (vmcnt(N)
// becomes vmcnt(0)).
__device__ __forceinline__ f32x4 load16_global(const char* p) {
    return *reinterpret_cast<const __attribute__((address_space(1))) f32x4*>(reinterpret_cast<unsigned long long>(p));
}

// NT: non-temporal (streaming) store — the line is marked evict-first in L2
// CCDM_SC1_STORES (build-time probe): write-through stores (sc1: the line leaves the XCD's L2 at once) — what a kernel leaves dirty in
// L2 is written back at the kernel boundary, B / 6 TB/s on the critical path of the next launch (MI355X_MICROARCH.md, "boundary")
template <bool NT = false>
__device__ __forceinline__ void store16_uniform_base(char* base, unsigned voff, const f32x4 v) {
    const unsigned long long u = reinterpret_cast<unsigned long long>(base);
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)u), hi = __builtin_amdgcn_readfirstlane((unsigned)(u >> 32));
#ifdef CCDM_SC1_STORES
    const unsigned long long sb = ((unsigned long long)hi << 32) | lo;
    asm volatile("global_store_dwordx4 %0, %1, %2 sc1" : : "v"(voff), "v"(v), "s"(sb) : "memory");
#else
    __attribute__((address_space(1))) char* g = reinterpret_cast<__attribute__((address_space(1))) char*>(((unsigned long long)hi << 32) | lo);
    if (NT) __builtin_nontemporal_store(v, reinterpret_cast<__attribute__((address_space(1))) f32x4*>(g + voff));
    else *reinterpret_cast<__attribute__((address_space(1))) f32x4*>(g + voff) = v;
#endif
}
// fp16 hi/lo split of two fp32 values: hi = RNE(x) packed, lo = RNE(x - hi) packed.  x - hi is one v_fma_mix_f32 (the fp16 half is
// read straight out of the packed register and widened by the instruction: fma(hi, -1, x), exactly the subtraction's single rounding),
// instead of an unpack (v_cvt_f32_f16) plus a subtract per element: 4 VALU instructions per pair instead of 7.
__device__ __forceinline__ void split2_f16(const float a, const float b, unsigned& hi, unsigned& lo) {
    float la, lb;
    asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(hi) : "v"(a), "v"(b));
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(la) : "v"(hi), "v"(a));
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(lb) : "v"(hi), "v"(b));
    asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(lo) : "v"(la), "v"(lb));
}

struct ConvK {   // kernel-side copy of ccdm_conv_args (+ derived)
    ccdm_conv_args a;
    int cin_pad, ntiles, slices, tiles_x, tiles_y;
    int cin_pad_skip;        // padded channels of the fused 1x1 skip segment (0: none)
    int skip_wide;           // the skip segment runs in 32-channel chunks staged core-pixels-only (wide-tile F16X3 variant, SC % 32 == 0)
    const float* wscale;     // F16X3: [ntiles*32] powers of two undoing the per-output-channel weight pre-scale
    int core_unmasked;       // every core column of every tile lies inside the image and every channel quad exists (W % TW == 0, C % CK == 0):
                             // core halo items need no per-lane padding mask, only the wave-uniform row test
    unsigned long long* timeline;   // diagnostics of the producer/consumer experiment (CCDM_EXPERIMENTS builds), else NULL
};

// plain 1x1 conv (+bias +residual +statistics) of a low-resolution tensor without LDS staging (ccdm_conv1x1.hip)
bool conv1x1_eligible(const ccdm_conv_args& a, int slices);
int launch_conv1x1(const ccdm_conv_args& a, int slices, int ntiles, const float* wscale, hipStream_t s);

// 3x3 conv of a few-pixel image, K split over the waves of a block, weight fragments straight from L2 (ccdm_conv_ks.hip)
bool conv_ks_eligible(const ccdm_conv_args& a);
int conv_ks_slices(const ccdm_conv_args& a);            // statistics slices that kernel leaves (one per 8x8 tile)
int launch_conv_ks(const ccdm_conv_args& a, int ntiles, const float* wscale, hipStream_t s);
// Upsample + conv 3x3 in sub-pixel form at the low-resolution decoder levels: wave = phase, weight fragments straight from L2 (ccdm_upconv.hip)
bool upconv_eligible(const ccdm_conv_args& a);
int launch_upconv(const ccdm_conv_args& a, int slices, int ntiles, const float* wscale, hipStream_t s);
#ifdef CCDM_ABLATION
bool conv_ks_timeline_read(unsigned long long* host, int n);
#endif

#ifdef CCDM_EXPERIMENTS
// producer/consumer form of the full-width 3x3 stages (tools/experiments/ccdm_conv_pc.hip: measured slower, never in the shipped library)
bool conv_pc_eligible(const ConvK& k, const ConvGeo& g, int NI);
int launch_conv_pc(const ConvK& k, hipStream_t s);
bool conv_pc_timeline_read(unsigned long long* host, int n);
#endif

// ---------------------------------------------------------------------------------------------------
// GroupNorm affine for sample n:  ab[c] = (scale, shift) such that  y = scale*x + shift
//
// Two halves so that a kernel can put its first HBM requests between them:
//   gn_group_sums  — the (sum, sum^2) of channel c's group: every (channel of the group, slice) partial, added in ascending
//                    (channel, slice) order.  The partials are fetched 16 at a time with independent loads — one L2 round trip per 16
//                    instead of one per partial (12 dependent round trips = 2.6 us per block at the 128x128 stage before).
//   gn_finalize    — mean / rstd in fp64, gamma / beta (and FiLM) folded into one (scale, shift) pair.
// ---------------------------------------------------------------------------------------------------
typedef double f64x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ void gn_group_sums(const ccdm_conv_args& a, int n, int c, double& sum, double& sq) {
    const int C = a.C0 + a.C1;
    const int cpg = C / 32;
    const int c_lo = (c / cpg) * cpg, c_hi = c_lo + cpg;
    sum = 0.0; sq = 0.0;
    int cc = c_lo, s = 0;
    while (cc < c_hi) {
        f64x2 v[16];
        bool ok[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            ok[u] = cc < c_hi;
            const int ccl = ok[u] ? cc : c_hi - 1;                       // clamped: the load stays unconditional
            const bool second = ccl >= a.C0;
            const double* st = second ? a.stats1 : a.stats0;
            const int ci = second ? ccl - a.C0 : ccl, Cs = second ? a.C1 : a.C0, S = second ? a.slices1 : a.slices0;
            const int sl = ok[u] ? s : 0;
            v[u] = *reinterpret_cast<const f64x2*>(st + (((size_t)n * S + sl) * Cs + ci) * 2);
            if (++s >= S) { s = 0; ++cc; }
        }
#pragma unroll
        for (int u = 0; u < 16; ++u) {                                   // fixed order: ascending (channel, slice)
            sum += ok[u] ? v[u][0] : 0.0;
            sq += ok[u] ? v[u][1] : 0.0;
        }
    }
}

// mean and 1/sqrt(var + eps) of a group from its (sum, sum^2) over cnt values.  fp64 throughout, but without the division /
// square-root expansions (three v_div sequences and a v_sqrt: ~800 cycles of a block prologue that small-spatial launches cannot
// hide): hardware reciprocal / reciprocal-square-root estimates refined by two Newton steps each — relative error < 2^-50,
// invisible after the rounding to fp32.
__device__ __forceinline__ void gn_mean_rstd(double sum, double sq, double cnt, float eps, float& meanf, float& rstd) {
    double ic = __builtin_amdgcn_rcp(cnt);
    ic = ic * (2.0 - cnt * ic);
    ic = ic * (2.0 - cnt * ic);
    const double mean = sum * ic;
    double var = sq * ic - mean * mean;
    if (var < 0.0) var = 0.0;
    const double ve = var + (double)eps;
    double rs = __builtin_amdgcn_rsq(ve);
    rs = rs * (1.5 - 0.5 * ve * rs * rs);
    rs = rs * (1.5 - 0.5 * ve * rs * rs);
    meanf = (float)mean;
    rstd = (float)rs;
}

// per-channel parameters of the affine, fetched ahead of the arithmetic (gn_params) so that a kernel can issue every small load
// before its first HBM request and do the fp64 finalisation (gn_finalize: registers only) while that request is in flight
struct GnParams { float gamma, beta, film_scale, film_shift; };
__device__ __forceinline__ GnParams gn_params(const ccdm_conv_args& a, int emb_row, int c) {
    GnParams p;
    p.gamma = a.gamma[c]; p.beta = a.beta[c]; p.film_scale = 0.f; p.film_shift = 0.f;
    if (a.film) {
        const float* row = a.emb_table + (size_t)emb_row * a.emb_stride + a.film_off;
        p.film_scale = row[c]; p.film_shift = row[a.C0 + a.C1 + c];
    }
    return p;
}
__device__ __forceinline__ float2 gn_finalize(const ccdm_conv_args& a, const GnParams& p, double sum, double sq) {
    const int C = a.C0 + a.C1;
    const int cpg = C / 32;
    float meanf, rstd;
    gn_mean_rstd(sum, sq, (double)cpg * (double)a.Hin * (double)a.Win, a.eps, meanf, rstd);
    float sc = rstd * p.gamma;
    float sh = p.beta - sc * meanf;
    if (a.film) {   // h = GN(h) * (1 + scale) + shift          unet.py:254-258
        const float one_plus = 1.0f + p.film_scale;
        sc = sc * one_plus;
        sh = sh * one_plus + p.film_shift;
    }
    return make_float2(sc, sh);
}

// Prologue form for k_conv.  gn_prefetch issues EVERY load the affine of channel c needs — gamma, beta, the FiLM row and the
// channel's own first 16 slice partials — unconditionally (addresses clamped; `dummy` = any readable device memory, read when there
// is no GroupNorm), with two or three instructions of address arithmetic per load and no branch, and waits for none of them.  The
// kernel then issues its first halo request; the small loads return first (vector memory returns in order), so gn_affine_block —
// per-channel sums over the slices (ascending), exchanged through LDS, added over the group's channels (ascending), finalised in
// fp64 — runs while the halo is in flight.  (A branch between the loads and their use would make the waitcnt pass drain the whole
// queue at the join; a flat-addressed load anywhere in flight does the same.)
struct GnPrefetch { GnParams p; f64x2 v[16]; };
__device__ __forceinline__ const char* gn_channel_row(const ccdm_conv_args& a, int n, int c, int& S, unsigned& stride) {
    const bool second = c >= a.C0;                                       // (only with a concatenated input)
    const double* st = second ? a.stats1 : a.stats0;
    const int ci = second ? c - a.C0 : c, Cs = second ? a.C1 : a.C0;
    S = second ? a.slices1 : a.slices0;
    stride = (unsigned)Cs * 16u;
    return reinterpret_cast<const char*>(st + ((size_t)n * S * Cs + ci) * 2);
}
// Thread -> (channel, slot group).  A block has more threads than channels wherever many slices occur (32-64 channels on 256
// threads at the full-resolution stages), so the first G * C threads each take 16 slices of one channel: up to 64 slices are summed
// from ONE prefetch round.  G = min(NT / C, 4); threads beyond G * C idle (their loads are clamped duplicates).
struct GnLane { int c, grp, G; };
__device__ __forceinline__ GnLane gn_lane(const ccdm_conv_args& a, int tid, int NT) {
    const int C = a.C0 + a.C1;
    const int smax = a.slices0 > a.slices1 ? a.slices0 : a.slices1;
    const int need = smax > 16 ? (smax + 15) >> 4 : 1;                   // slot groups the slice count asks for (1 wherever <= 16 slices)
    GnLane l;
    l.G = NT >= 4 * C ? 4 : (NT >= 3 * C ? 3 : (NT >= 2 * C ? 2 : 1));
    l.G = l.G < need ? l.G : need;
    l.grp = (tid >= C ? 1 : 0) + (tid >= 2 * C ? 1 : 0) + (tid >= 3 * C ? 1 : 0);
    l.c = tid - l.grp * C;
    if (l.grp >= l.G || l.c >= C) { l.grp = l.G; l.c = C - 1; }         // idle lane (grp == G marks it)
    return l;
}
__device__ __forceinline__ void gn_prefetch(const ccdm_conv_args& a, bool has_gn, int n, int emb_row, int tid, int NT, const void* dummy, GnPrefetch& g) {
    const int C = a.C0 + a.C1;
    const GnLane l = gn_lane(a, tid, NT);
    const int c = l.c, grp = l.grp < l.G ? l.grp : l.G - 1;
    const float* df = static_cast<const float*>(dummy);
    const bool film = has_gn && a.film;
    const float* gam = has_gn ? a.gamma + c : df;
    const float* bet = has_gn ? a.beta + c : df;
    const float* row = film ? a.emb_table + (size_t)emb_row * a.emb_stride + a.film_off + c : df;
    g.p.gamma = *gam; g.p.beta = *bet;
    g.p.film_scale = row[0]; g.p.film_shift = row[film ? C : 0];
    int S = 1;
    unsigned stride = 0;
    const char* base = static_cast<const char*>(dummy);
    if (has_gn) base = gn_channel_row(a, n, c, S, stride);               // uniform condition, selects only
    const unsigned last = (unsigned)(S - 1) * stride, first = (unsigned)(16 * grp) * stride;
    // Few-pixel images leave 1-4 slices: 12 of the 16 requests would be clamped duplicates — 1 KB per wave each through a vector-memory
    // front end that takes ~40-64 B/clk, ~800 cycles of every block's prologue ahead of its halo request.  The slice count is a kernel
    // argument (uniform), so the extra requests sit behind one scalar branch.
    const bool few = !has_gn || (a.slices0 <= 4 && a.slices1 <= 4);
#pragma unroll
    for (int u = 0; u < 4; ++u) g.v[u] = *reinterpret_cast<const f64x2*>(base + min(first + (unsigned)u * stride, last));
    if (!few) {
#pragma unroll
        for (int u = 4; u < 16; ++u) g.v[u] = *reinterpret_cast<const f64x2*>(base + min(first + (unsigned)u * stride, last));
    } else {
#pragma unroll
        for (int u = 4; u < 16; ++u) g.v[u] = f64x2{0.0, 0.0};
    }
}
// (sum, sum^2) of channel c over the slices [16 grp, 16 grp + 16) from the prefetched values g (NULL: fetched here), ascending; the lane
// of group 0 also takes the slices beyond 16 G (more slices than one prefetch round of the block covers: blocking loads)
__device__ __forceinline__ f64x2 gn_channel_sums(const ccdm_conv_args& a, int n, int c, int grp, int G, const GnPrefetch* g) {
    int S;
    unsigned stride;
    const char* base = gn_channel_row(a, n, c, S, stride);
    f64x2 acc = {0.0, 0.0};
    const int s_first = 16 * grp;
    if (g) {
#pragma unroll
        for (int u = 0; u < 16; ++u) {                                   // selects, not branches (x + 0.0 == x)
            acc[0] += s_first + u < S ? g->v[u][0] : 0.0;
            acc[1] += s_first + u < S ? g->v[u][1] : 0.0;
        }
    } else {
        for (int s = s_first; s < S && s < s_first + 16; ++s) acc += *reinterpret_cast<const f64x2*>(base + (unsigned)s * stride);
    }
    if (grp == 0)
        for (int s = 16 * G; s < S; ++s) acc += *reinterpret_cast<const f64x2*>(base + (unsigned)s * stride);
    return acc;
}
// the whole table ab[0..C): call with the block's threads converged; `scratch` = max(C, blockDim.x) x 16 B of LDS not otherwise in use
// until the caller's next barrier.  Order of additions per group: channels ascending, per channel the slot groups ascending, per slot
// group the slices ascending (for <= 16 slices: the plain ascending (channel, slice) order).
__device__ __forceinline__ void gn_affine_block(const ccdm_conv_args& a, int n, int emb_row, const GnPrefetch& g, f64x2* scratch, float2* ab) {
    const int C = a.C0 + a.C1, cpg = C / 32;
    const int tid = threadIdx.x, NT = blockDim.x;
    const GnLane l = gn_lane(a, tid, NT);
    const int G = l.G;
    auto group = [&](const int c) {
        const int c_lo = (c / cpg) * cpg;
        f64x2 acc = {0.0, 0.0};
        for (int j = 0; j < cpg; ++j)
            for (int q = 0; q < G; ++q) acc += scratch[q * C + c_lo + j];
        return acc;
    };
    // the lanes of the first G * C threads work from the prefetched values alone — no load here, which would be younger than the caller's
    // halo request and drag its round trip into this wait; channels beyond the block size (C > NT, then G = 1: rare) take blocking loads
    if (l.grp < G) scratch[l.grp * C + l.c] = gn_channel_sums(a, n, l.c, l.grp, G, &g);
    for (int c = tid + NT; c < C; c += NT) scratch[c] = gn_channel_sums(a, n, c, 0, 1, nullptr);
    __syncthreads();
    if (tid < C) {
        const f64x2 acc = group(tid);
        ab[tid] = gn_finalize(a, g.p, acc[0], acc[1]);
    }
    for (int c = tid + NT; c < C; c += NT) {
        const f64x2 acc = group(c);
        ab[c] = gn_finalize(a, gn_params(a, emb_row, c), acc[0], acc[1]);
    }
}

__device__ __forceinline__ void compute_gn_affine(const ccdm_conv_args& a, int n, int emb_row, float2* ab, int c_first = 0) {
    const int C = a.C0 + a.C1;
    for (int c = c_first + threadIdx.x; c < C; c += blockDim.x) {
        double sum, sq;
        const GnParams p = gn_params(a, emb_row, c);
        gn_group_sums(a, n, c, sum, sq);
        ab[c] = gn_finalize(a, p, sum, sq);
    }
}


}  // namespace ccdm
