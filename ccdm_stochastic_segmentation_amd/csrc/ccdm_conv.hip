// Fused [GroupNorm -> (FiLM) -> SiLU ->] conv KxK [+bias +emb +residual] (+ per-channel output statistics)
// on NHWC fp32, implicit GEMM on the gfx950 matrix cores.
//
//   GEMM view:  M = output pixels (32-pixel sub-tiles of a TH x TW tile), N = output channels (32-wide
//   tiles), K = taps x input channels, walked in chunks of CK input channels.  Per chunk the block stages
//   the (TH-1)*stride+k halo tile into LDS *already normalised and activated* (GroupNorm's affine is folded
//   to one fma per element from per-channel partial sums the producer left behind), then every tap re-reads
//   it from LDS: each input element is fetched from HBM once per tile, transformed once, used k*k*Cout times.
//
//   Pipeline per block (one (sample, slice) pair, looping over that slice's tiles x channel chunks):
//       issueB(i): weight fragments of chunk i, global (L2) -> registers
//       barrier ; commit(i): halo registers -> GN affine -> SiLU -> [fp16 hi|lo split] -> LDS, fragment registers -> LDS ; barrier
//       issue(i+1): next halo, global -> registers   (raw fp32, in flight during the MFMA phase of step i)
//       MFMA phase(i): A and B fragments from LDS only — no vector-memory op between issue and the next
//       commit, so the prefetch is never dragged in early by the in-order vmcnt queue; the LDS reads of tap t+1 are
//       issued ahead of the MFMAs of tap t.  (F16X3; the exact F32 path reads its pre-packed B fragments straight from L2.)
//       epilogue on a tile's last chunk: accumulators -> LDS transpose -> (+bias +emb +residual) -> float4 stores, statistics
//   Every geometry constant (halo width, item counts, tap offsets) is compile-time; the (tile, chunk) walk and each chunk's
//   operand pointers are scalar state (counters, register-lane descriptors), so the loop carries almost no address arithmetic.
//
//   A block leaves its per-channel (sum, sum^2) partials for the *next* GroupNorm in a fixed slot —
//   no atomics, fixed order, run-to-run deterministic, independent of the batch size.
//
// Replaces (reference, /root/reference/ddpm/models/unet_openai/unet.py): ResBlock in/out layers :186-219,
// :242-262; skip 1x1 :221-228; Downsample :137-146; Upsample :106-116; AttentionBlock norm+qkv / proj_out
// :291-300; stem :517; head GN-SiLU-conv :701-707.   GroupNorm32 = nn.py:17-19,93-100.
#include "ccdm_common.h"
#include "ccdm_conv_common.h"

#include <cmath>
#include <cstdlib>
#include <type_traits>
#include <vector>

namespace ccdm {

// Ablation / timeline switches (bits 8.. of `prec`, used by tools/bench_conv.py) exist only in a -DCCDM_ABLATION build
// (CCDM_ABLATION=1 python -c "from ccdm_stochastic_segmentation_amd import hip; hip.build()"): as run-time tests they put
// a branch around every store and every phase of the production kernel.
//   1 no MFMA | 2 no commit | 4 no loads | 8 no stores | 16 phase timeline | 32 no weight-fragment staging (wrong results: what
//   removing the global -> registers -> LDS round trip of the B chunk could buy) | 64 no halo (every halo request is pointed at the
//   nearest CORE pixel of its tile — wrong results: what the tile's halo rows / columns cost in fetched bytes and time) |
//   256 no barriers (wrong results)
//   512 / 1024: pad the LDS request so that at most 2 / 1 blocks fit a CU (host side, always available)
#ifdef CCDM_ABLATION
#define CCDM_DBG(bit) ((dbg & (bit)) != 0)
#else
#define CCDM_DBG(bit) false
#endif
// phase timeline of one block (ablation bit 16 of prec; read back with ccdm_debug_read_timeline)
__device__ unsigned long long g_timeline[1024];
#define CCDM_STAMP(slot) do { if (CCDM_DBG(16) && blockIdx.x == gridDim.x / 2 && blockIdx.y == 0 && tid == 0 && tl < 1020) \
        g_timeline[tl++] = ((unsigned long long)(slot) << 56) | (__builtin_amdgcn_s_memtime() & 0x00ffffffffffffffull); } while (0)

// F32  : CK = 32 channels per chunk; LDS pixel = 33 floats (odd stride: conflict-free column reads).
// F16X3: CK = 16 channels per chunk; LDS pixel = 16 hi halfs | 16 lo halfs | 16 B pad = 80 B = 20 dwords:
//        the 16 pixels of a ds_read_b128 lane group land on 16 disjoint 4-bank slots (20*p mod 64).
//        Small-spatial stages (few pixels, many channels) take CK = 32 per chunk instead (pixel = 32 hi | 32 lo | pad
//        = 144 B = 36 dwords, also conflict-free): half as many barrier/latency round trips per tile.
template <int PREC, int CKT> struct Lds {
    static constexpr int CK = CKT;
    static constexpr int PIXB = PREC == CCDM_PREC_F32 ? 33 * 4 : CKT * 4 + 16;
};

// (ACT_PRESCALE, ConvK, compute_gn_affine, load16/store16_uniform_base: ccdm_conv_common.h)

// register budget: >= 3 waves per SIMD (<= 168 VGPRs) when the accumulator tile is small — matches the 3 blocks
// per CU the LDS footprint (A tile 27 KB + B chunk 18 KB) admits
constexpr int min_waves(int mi, int ni, int prec, int ckt, int threads) {
    if (threads > 512) return 3;                           // 12-wave blocks (tap-split 8x16 tiles): 3 waves per SIMD
    if (prec != CCDM_PREC_F32 && ckt >= 32) return 2;      // small-spatial variants: few blocks per CU anyway, take the registers
    return mi * ni <= 2 ? 3 : (mi * ni <= 4 ? 2 : 1);
}

// KSP > 1 (small-spatial 3x3 convs): KSP wave groups share one staged tile and split the kernel rows between them
// (group r multiplies only taps (r, *)); their partial accumulators meet in the epilogue's LDS buffer.  A few-pixel
// stage has too few tiles to fill the chip with pixel parallelism alone — this triples the waves per tile and cuts
// the per-wave MFMA chain to a third.
//
// UP2 (nearest x2 upsample + 3x3 conv in sub-pixel form, include/ccdm_hip.h `up = 2`): tiles, halo and staging live in the
// LOW-resolution input space exactly as for a plain 3x3 conv; an output-channel tile of the launch is a (real channel tile,
// phase) pair (n-tile = 4 * tile + phase), phase (dy, dx) multiplies the 2x2 taps at halo offset (dy + a, dx + b) with its own
// (pre-summed) weights and the epilogue writes pixel (2y + dy, 2x + dx).  4 taps instead of 9 per output pixel.
// NI = 4: one block computes all four phases of a channel tile from ONE staged halo (a quarter of the staging work per output
// pixel): the tap walk visits the 9 halo positions once and feeds each A fragment to the 1, 2 or 4 phases whose window contains
// it; the phases' statistics fold into the block's one partial.  NI = 1 (8x8 inputs, where blocks are scarce): one phase per
// block, statistics slot = slice * 4 + phase.
template <int PREC, int CKT, int KS, int STRIDE, int TH, int TW, int WAVES, int MI, int NI, int KSP = 1, bool UP2 = false, bool SKWT = false>
__global__ __launch_bounds__(WAVES * KSP * 64, min_waves(MI, NI, PREC, CKT, WAVES * KSP * 64)) void k_conv(const ConvK k) {
    constexpr int NT = WAVES * KSP * 64;
    static_assert(KSP == 1 || KSP == KS, "tap split is by kernel row");
    static_assert(!UP2 || (KS == 3 && STRIDE == 1 && KSP == 1 && (NI == 1 || NI == 4) && PREC != CCDM_PREC_F32), "sub-pixel form: 3x3, stride 1, F16X3, one phase or all four");
    constexpr int NTAP = UP2 ? 4 : KS * KS;
    constexpr int CK = Lds<PREC, CKT>::CK, PIXB = Lds<PREC, CKT>::PIXB;
    constexpr int KST = CK / 16;                                  // F16X3: 16-channel MFMA k-steps per chunk
    constexpr int PAD = KS / 2;
    constexpr int HHt = (TH - 1) * STRIDE + KS, HWt = (TW - 1) * STRIDE + KS, HP = HHt * HWt;
    constexpr int QPP = CK / 4;                                   // float4 items per halo pixel
    static_assert(NT % QPP == 0, "channel quad must be item-invariant");
    // Stride-1 staging is row-structured: a pass of the block covers RPP whole rows of the TW core columns of the halo
    // (thread -> (row in pass, column, channel quad), all item-invariant), the 2*PAD edge columns are one extra item for
    // the first few threads.  Row index and row validity are wave-uniform (scalar ALU), the column part of the address
    // is computed once per tile-chunk, every LDS address is thread-constant + immediate: staging costs no per-item VALU
    // beyond the arithmetic on the data itself.  Stride 2 (round 3) walks the same way: the core is the CW = 2 TW columns the tile's
    // outputs read from halo column PAD on, the single left padding column is the edge item (58.7 -> 5x us at 128x128 -> 64x64 with
    // the generic item -> (hy, hx) walk before).
    constexpr bool ROWS = true;
    constexpr int CW = TW * STRIDE;                               // core columns of the halo (halo x = PAD + core column)
    constexpr int PXW = NT / QPP;                                 // halo pixels per pass
    static_assert(!ROWS || PXW % CW == 0, "a pass must cover whole core rows");
    constexpr int RPP = ROWS ? PXW / CW : 1;                      // core rows per pass
    constexpr int NCORE = ROWS ? (HHt + RPP - 1) / RPP : 0;
    constexpr int ECOLS = HWt - CW > 0 ? HWt - CW : 1;            // edge columns: 2 PAD at stride 1, PAD (left only) at stride 2 (1: placeholder when there are none)
    constexpr int EDGE_ITEMS = ROWS ? HHt * (HWt - CW) * QPP : 0;
    constexpr int NEDGE = (EDGE_ITEMS + NT - 1) / NT;
    constexpr bool ROW_UNIFORM = ROWS && (CW * QPP) % 64 == 0;    // a wave never straddles two core rows
    constexpr int NITEM = ROWS ? NCORE + NEDGE : (HP * QPP + NT - 1) / NT;   // staging items per thread
    static_assert(NITEM <= 32, "validity mask is 32 bits");
    // Wide skip chunks (SKW; run-time switch k.skip_wide).  A chunk of the fused 1x1 skip segment feeds the centre tap only: it needs no
    // halo, so the same LDS holds 32 channels of the TH x TW core pixels (144-byte pixels: 36.9 KB) + 2 k-steps of one tap's fragments
    // (4 KB).  Half as many skip chunks — each one is two barriers, an issue and a commit phase around 12 instead of 6 MFMAs per wave,
    // and the 16-channel skip chunks were 45 % of the tile time of a decoder ResBlock's second conv — and no staging of halo pixels.
    // (its own instantiation, SKWT: 159 registers against the plain one's 149 — both keep 3 waves per SIMD)
    constexpr bool SKW = SKWT && PREC != CCDM_PREC_F32 && KS == 3 && STRIDE == 1 && TW == 32 && NI == 1 && KSP == 1 && !UP2 && CKT == 16;
    constexpr int CKS = 32, PIXS = CKS * 4 + 16, QPS = CKS / 4;
    constexpr int NITEM_S = SKW ? TH * TW * QPS / NT : 0;          // core-only items per thread (8)
    constexpr int NITEM_R = NITEM_S > NITEM ? NITEM_S : NITEM;    // halo register set size
    static_assert(!SKW || (TH * TW * QPS) % NT == 0, "a wide skip chunk tiles the block");
    constexpr int A_BYTES = (HP * PIXB + 15) / 16 * 16;
    // F16X3: the chunk's B fragments, [tap][k-step] slabs of G = [ni][hi|lo][64 lanes] x 16 B, staged through registers
    // like the halo.  A pass covers MB whole slabs (or 1/DB of one); the slab index is wave-uniform.
    constexpr int G = NI * 128;
    constexpr int NB4 = PREC == CCDM_PREC_F32 ? 0 : NTAP * KST * G;
    constexpr int NITEM_B = (NB4 + NT - 1) / NT;
    constexpr bool B_MULTI = NT % G == 0;
    static_assert(PREC == CCDM_PREC_F32 || B_MULTI || G % NT == 0, "B slabs must tile the block");
    constexpr int MB = B_MULTI ? NT / G : 1, DB = B_MULTI ? 1 : G / NT;

    const ccdm_conv_args& a = k.a;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int C = a.C0 + a.C1;
    float2* ab = reinterpret_cast<float2*>(smem);                              // [C] (only if stats0)
    char* halo_b = smem + (a.stats0 ? (size_t)C * 8 : 0);
    float* halo = reinterpret_cast<float*>(halo_b);
    f32x4* ldsB = reinterpret_cast<f32x4*>(halo_b + A_BYTES);      // native vector type: HIP's float4 struct defeats SROA here

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave_all = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wave = wave_all % WAVES, krow = wave_all / WAVES;       // pixel sub-tile owner, kernel row (KSP > 1)
    // XCD-aware mapping: block b runs on XCD b % 8 — give each XCD a contiguous range of (sample, slice)
    // pairs so the slices of a sample (shared halo rows, shared statistics) meet in one L2.
    int bid = blockIdx.x;
    if ((gridDim.x & 7) == 0) bid = (blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3);
    const int n = bid / k.slices, slice = bid % k.slices;
    const int nt0 = blockIdx.y * NI;
    // sub-pixel form: n-tile = 4 * (real channel tile) + phase
    auto tile_of = [&](const int ni) { return UP2 ? (nt0 + ni) >> 2 : nt0 + ni; };      // real 32-channel tile of n-tile ni
    auto phase_of = [&](const int ni) { return UP2 ? (nt0 + ni) & 3 : 0; };
    const int Hc = (a.up && !UP2) ? a.Hin * 2 : a.Hin, Wc = (a.up && !UP2) ? a.Win * 2 : a.Win;    // conv-input space (= tile space)
    const int step = a.step_ptr ? *a.step_ptr : 0;
    const int emb_row = (a.emb_row_of_sample ? a.emb_row_of_sample[n] : 0) + step;
    const bool has_gn = a.stats0 != nullptr;
    const int dbg = a.prec >> 8;          // ablation switches for tools/bench_conv.py (CCDM_ABLATION builds only)
    (void)dbg;
    int tl = 0;
    CCDM_STAMP(12);

    const int aWout = a.Wout, aCout = a.Cout, aHout = a.Hout, aWin = a.Win;
    const size_t in_px = (size_t)a.Hin * a.Win;
    const size_t out_px = (size_t)a.Hout * a.Wout;
    float* outn = a.out + (size_t)n * out_px * aCout;
    const float* residn = a.resid + (size_t)n * out_px * aCout;   // guarded by a.resid at the uses

    // per-lane LDS base of each of this wave's 32-pixel sub-tiles (A operand: row = lane&31, k-group = lane>>5)
    int base[MI];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
        const int p = (wave * MI + mi) * 32 + (lane & 31);
        const int hpix = (p / TW) * STRIDE * HWt + (p % TW) * STRIDE;
        base[mi] = PREC == CCDM_PREC_F32 ? hpix * 33 + (lane >> 5) : hpix * PIXB + (lane >> 5) * 16;   // floats | bytes
        if (UP2 && NI == 1) base[mi] += ((phase_of(0) >> 1) * HWt + (phase_of(0) & 1)) * PIXB;   // this phase's 2x2 window starts at halo offset (dy, dx)
    }
    // output statistics: slow epilogue -> lane = channel (index 0 used); fast epilogue -> lane = (pixel, channel quad)
    // (fp32 per lane: <= a few hundred values each; widened to fp64 before lanes, waves and slices are combined)
    float s1[NI][4], s2[NI][4];
#pragma unroll
    for (int ni = 0; ni < NI; ++ni)
#pragma unroll
        for (int j = 0; j < 4; ++j) { s1[ni][j] = 0.f; s2[ni][j] = 0.f; }
    const bool fast_epi = (a.Cout & 3) == 0;       // uniform: float4 rows through an LDS transpose
    // per-lane epilogue constants of output channel (n-tile, lane & 31): bias (+ emb row) and the power of two that undoes
    // the weight / activation pre-scales.  Fetched once per block — inside the tile loop their L2 latency sat in every
    // epilogue's critical path.
    //   The block's small loads — these constants and everything GroupNorm's (scale, shift) table needs — are all ISSUED here, ahead
    //   of the first halo request, unconditionally (clamped / dummy addresses: no branch, hence no waitcnt drain at a join), and
    //   consumed only after that request is on its way: vector memory returns in order, so they come back first and the fp64
    //   finalisation runs while the halo is in flight.  (Consumed in place they cost the block two to three dependent round trips
    //   before its first commit: 3700 of the 7300 prologue cycles of an 8x8 block, whose chain is the launch time.)
    float epi_add[NI], epi_wsc[NI], raw_bias[NI], raw_emb[NI];
    const float* const dummyf = reinterpret_cast<const float*>(a.w);
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) {
        const int co = min(tile_of(ni) * 32 + (lane & 31), a.Cout - 1);              // real output channel (clamped; masked below)
        raw_bias[ni] = *(a.bias ? a.bias + co : dummyf);
        raw_emb[ni] = *(a.emb_off >= 0 ? a.emb_table + (size_t)emb_row * a.emb_stride + a.emb_off + co : dummyf);
        epi_wsc[ni] = PREC != CCDM_PREC_F32 ? k.wscale[min((nt0 + ni) * 32 + (lane & 31), k.ntiles * 32 - 1)] : 1.0f;   // exact power of two (per packed channel: a phase has its own)
    }
    GnPrefetch gpf;
    gn_prefetch(a, has_gn, n, emb_row, tid, NT, a.w, gpf);
    __builtin_amdgcn_sched_barrier(0);             // the requests above stay above the halo request
    constexpr int EPS = 36;                        // floats per pixel row of the transpose buffer (16-B aligned rows)
    float* epi = reinterpret_cast<float*>(halo_b) + wave_all * (MI * 32 * EPS);      // [krow][wave][MI*32][EPS]

    const int ntile_sp = k.tiles_x * k.tiles_y;
    const bool skw = SKW && k.skip_wide;                        // uniform
    const int CKSK = skw ? CKS : CK;                            // channels per skip chunk
    const int nchunk_main = k.cin_pad / CK;
    const int nchunk = nchunk_main + k.cin_pad_skip / CKSK;    // main segment, then the fused 1x1 skip segment
    const int my_tiles = (ntile_sp - slice + k.slices - 1) / k.slices;
    const int n_iter = my_tiles * nchunk;

    // Staging register sets: ONE halo set and one fragment set.  The halo of iteration i+1 is requested right after the commit of
    // iteration i; the weight fragments at the top of the iteration that consumes them (registers are the scarce resource of the
    // wide-tile variants: 3 waves per SIMD).  Two-deep variants (a second halo set, fragments two iterations ahead, next-chunk fragments
    // requested with the next halo) were built, parity-tested and measured neutral to slower in rounds 1-2; they are gone.
    constexpr int DEPTH = 1;
    f32x4 reg[DEPTH][NITEM_R];
#ifndef CCDM_BDMA
#define CCDM_BDMA 1
#endif
    // weight fragments by LDS-DMA instead of through registers — on the narrow-tile variants, where the B chunk (36-74 KB) outweighs the
    // halo tile: same-box A/B per stage 16x16 423 -> 402 us, 32x32 420 -> 414 us per denoise step; the wide-tile variants LOSE with it
    // (128x128 1358 -> 1378 us: the request sits behind barrier A instead of in front of it, and their chunk is only 18 KB)
    // (not the stride-2 variants: their commit is short — raw input — so the DMA's round trip sat exposed between the two barriers:
    //  Downsample 128x128 -> 64x64 61.2 -> 58.9 us, 64x64 -> 32x32 20.6 -> 19.8 us through registers, round 5)
    constexpr bool BDMA = CCDM_BDMA && PREC != CCDM_PREC_F32 && TW < 32 && STRIDE == 1;
    f32x4 regB[1][(NITEM_B > 0 && !BDMA) ? NITEM_B : 1];
    unsigned valid[DEPTH];         // generic walk: bit i = item i lies inside the image
    unsigned rowmask[DEPTH];       // row-structured: bit i = core row of pass i inside the image (wave-uniform)
    unsigned evalid[DEPTH];        //                 bit j = edge item j inside the image
    bool xok[DEPTH];               //                 this thread's core column (and channel quad) exists
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) { valid[d] = 0; rowmask[d] = 0; evalid[d] = 0; xok[d] = false; }

    // thread -> staging coordinates (row-structured walk): channel quad tq, core column px, row within a pass rip.
    // tq and px are re-derived from an opaque copy of tid inside issue/commit: kept live across the MFMA phase and the
    // epilogue they cost registers the kernel does not have (3 waves per SIMD = 168), re-deriving them is 3 VALU ops.
    const int rip = ROW_UNIFORM ? __builtin_amdgcn_readfirstlane((tid / QPP) / CW) : (tid / QPP) / CW;
    // B staging: slab within a pass (wave-uniform: G >= 128 lanes), item within the slab
    const unsigned tgB = B_MULTI ? (unsigned)__builtin_amdgcn_readfirstlane(tid / G) : 0u;

    // Per-chunk descriptors, one chunk per LANE of a few registers (lane c = chunk c; read back with v_readlane and a
    // scalar chunk index): the source tensor's per-sample base pointer, its channel count and the chunk's first channel
    // within it, the chunk's weight-fragment base.  Built once; the loop then selects a chunk's operands with five
    // readlanes instead of re-deriving them (selects over kernel-argument fields, 64-bit multiplies, scalar loads).
    unsigned T_lo = 0, T_hi = 0, T_cc = 0, T_wlo = 0, T_whi = 0;
    {
        const int ch = lane < nchunk ? lane : 0;
        const bool sk = ch >= nchunk_main;                       // this chunk belongs to the fused 1x1 skip segment
        const int c0 = sk ? (ch - nchunk_main) * CKSK : ch * CK;
        const int sC0 = sk ? a.SC0 : a.C0, sC1 = sk ? a.SC1 : a.C1;
        // a chunk never straddles the concat seam (launcher: C0 % CK == 0 when there is a second source)
        const bool second = sC1 > 0 && c0 >= sC0;
        const int Cs = second ? sC1 : sC0, cb = second ? c0 - sC0 : c0;
        const float* srcsel = sk ? (second ? a.skip1 : a.skip0) : (second ? a.in1 : a.in0);
        const unsigned long long sb = reinterpret_cast<unsigned long long>(srcsel + (size_t)n * (sk ? out_px : in_px) * Cs);
        T_lo = (unsigned)sb; T_hi = (unsigned)(sb >> 32);
        T_cc = (unsigned)Cs | ((unsigned)cb << 16);
        const unsigned long long wb = reinterpret_cast<unsigned long long>(sk ? a.skip_w : a.w) + (((size_t)(c0 >> 4) * k.ntiles + nt0) * 128 << 4);
        T_wlo = (unsigned)wb; T_whi = (unsigned)(wb >> 32);
    }

    // ---- issue: global -> registers for iteration `it` (tile, chunk) ----
    auto issue = [&](auto D_, const int ch, const int ty, const int tx) {
        constexpr int d = decltype(D_)::value;
        const unsigned cc = __builtin_amdgcn_readlane(T_cc, ch);
        const int Cs = cc & 0xffffu, cb = cc >> 16;
        const char* srcb = reinterpret_cast<const char*>(((unsigned long long)(unsigned)__builtin_amdgcn_readlane(T_hi, ch) << 32) |
                                                         (unsigned)__builtin_amdgcn_readlane(T_lo, ch));      // uniform per-sample base
        const int ups = UP2 ? 0 : __builtin_amdgcn_readfirstlane(a.up);   // scalar shift amount (in a vector register the row math below turns vector too)
        const int oy0 = ty * TH, ox0 = tx * TW;
        if constexpr (SKW) {
            if (skw && ch >= nchunk_main) {
                // wide skip chunk: the TH x TW core pixels, 8 channel quads per pixel, one row per pass (all of it wave-uniform row math)
                unsigned t_ = tid;
                asm volatile("" : "+v"(t_));
                const int tq = t_ % QPS, px = t_ / QPS;
                const unsigned c = (unsigned)cb + 4u * (unsigned)tq;
                const unsigned cq = min(c, (unsigned)Cs - 4u);
                const int ix = ox0 + px;
                xok[d] = (c < (unsigned)Cs) & (ix < Wc);
                const unsigned colb = ((unsigned)min(ix, Wc - 1) * (unsigned)Cs + cq) << 2;
                const unsigned rowb = (unsigned)aWin * (unsigned)Cs * 4u;
                rowmask[d] = 0;
#pragma unroll
                for (int i = 0; i < NITEM_S; ++i) {
                    const int iy = oy0 + i;
                    reg[d][i] = load16_uniform_base(srcb + (size_t)((unsigned)min(iy, Hc - 1) * rowb), colb);
                    rowmask[d] |= (iy < Hc ? 1u : 0u) << i;
                }
                return;
            }
        }
        const bool skseg = KS > 1 && (ch >= nchunk_main || CCDM_DBG(64));   // uniform: skip-segment chunk (1x1, no halo needed)
        const int ylo = skseg ? min(oy0, Hc - 1) : 0, yhi = skseg ? min(oy0 + TH - 1, Hc - 1) : Hc - 1;
        const int xlo = skseg ? min(ox0, Wc - 1) : 0, xhi = skseg ? min(ox0 + TW - 1, Wc - 1) : Wc - 1;
        // Every load is issued unconditionally with its address clamped into the tensor; padding is zeroed at commit.
        // (A conditional load would put a control-flow join between the prefetch and the MFMA phase, and the waitcnt
        //  pass then drains the whole prefetch (vmcnt(0)) at the join.)
        unsigned t_ = tid;
        asm volatile("" : "+v"(t_));     // recompute the item geometry each call: hoisting it costs more registers than ALU
        if constexpr (ROWS) {
            const int tq = t_ % QPP, px = (t_ / QPP) % CW;
            const unsigned c = (unsigned)cb + 4u * (unsigned)tq;
            const unsigned cq = min(c, (unsigned)Cs - 4u);
            const bool cok = c < (unsigned)Cs;
            const unsigned rowb = (unsigned)aWin * (unsigned)Cs * 4u;            // bytes per source row (uniform)
            {
                const int ix = ox0 * STRIDE + px;                                  // core columns: halo x = px + PAD
                xok[d] = cok & (ix < Wc);
                const int ixc = min(ix, Wc - 1);
                const unsigned colb = ((unsigned)(ixc >> ups) * (unsigned)Cs + cq) << 2;
                rowmask[d] = 0;
#pragma unroll
                for (int i = 0; i < NCORE; ++i) {
                    const int row = rip + i * RPP;
                    const int iy = oy0 * STRIDE - PAD + row;
                    const bool rok = ((unsigned)iy < (unsigned)Hc) & ((i + 1) * RPP <= HHt || row < HHt);
                    // (skip-segment chunks feed the centre tap only: their halo rows/columns are never read, so those requests
                    //  are pointed at the nearest core row/column — a cache hit instead of 25 % more HBM lines)
                    const int iyc = min(max(iy, ylo), yhi);
                    const unsigned sy = (unsigned)(iyc >> ups);
                    if constexpr (ROW_UNIFORM) reg[d][i] = load16_uniform_base(srcb + (size_t)(sy * rowb), colb);
                    else reg[d][i] = load16_global(srcb + (sy * rowb + colb));
                    rowmask[d] |= (rok ? 1u : 0u) << i;
                }
            }
            evalid[d] = 0;
#pragma unroll
            for (int j = 0; j < NEDGE; ++j) {
                const unsigned e = t_ + j * NT;
                const unsigned side = (e / QPP) % ECOLS, row = e / (ECOLS * QPP);
                const int hx = side < (unsigned)PAD ? (int)side : CW + (int)side;
                const int iy = oy0 * STRIDE - PAD + (int)row, ix = ox0 * STRIDE - PAD + hx;
                const bool ok = cok & (e < (unsigned)EDGE_ITEMS) & ((unsigned)iy < (unsigned)Hc) & ((unsigned)ix < (unsigned)Wc);
                const int iyc = min(max(iy, ylo), yhi), ixc = min(max(ix, xlo), xhi);
                const unsigned sy = (unsigned)(iyc >> ups), sx = (unsigned)(ixc >> ups);
                reg[d][NCORE + j] = load16_global(srcb + (size_t)(sy * rowb + ((sx * (unsigned)Cs + cq) << 2)));
                evalid[d] |= (ok ? 1u : 0u) << j;
            }
        } else {
            valid[d] = 0;
            // item = tid + i*NT  ->  halo pixel hp = item / QPP (hy = hp / HWt, hx = hp % HWt), channel quad q = item % QPP.
            // NT % QPP == 0, so q is the same for every i and hp advances by NT/QPP: (hy, hx) are carried incrementally
            // (unsigned, no per-item division).
            constexpr unsigned DHP = NT / QPP, DHY = DHP / HWt, DHX = DHP % HWt;
            const unsigned q = t_ % QPP, hp0 = t_ / QPP;
            unsigned hy = hp0 / HWt, hx = hp0 % HWt;
            const unsigned c = (unsigned)cb + 4u * q;
            const unsigned cq = min(c, (unsigned)Cs - 4u);
            const bool cok = c < (unsigned)Cs;
#pragma unroll
            for (int i = 0; i < NITEM; ++i) {
                const int iy = oy0 * STRIDE - PAD + (int)hy, ix = ox0 * STRIDE - PAD + (int)hx;
                // padding test: unsigned compare folds the < 0 and >= extent checks; no short-circuit branches
                const bool ok = cok & ((unsigned)iy < (unsigned)Hc) & ((unsigned)ix < (unsigned)Wc) & (hy < (unsigned)HHt);
                const int iyc = min(max(iy, 0), Hc - 1), ixc = min(max(ix, 0), Wc - 1);
                const int sy = iyc >> ups, sx = ixc >> ups;
                const unsigned off = ((unsigned)(sy * aWin + sx) * (unsigned)Cs + cq) << 2;      // bytes within the sample
                reg[d][i] = load16_global(srcb + off);
                valid[d] |= (ok ? 1u : 0u) << i;
                hy += DHY; hx += DHX;
                if (hx >= (unsigned)HWt) { hx -= HWt; hy += 1; }
            }
        }
    };
    // ---- issueB: this chunk's weight fragments, global (L2-resident) -> registers.  Requested at the top of the
    //      iteration that consumes them — their (short) latency hides behind the commit's arithmetic — so that they
    //      do not occupy 4*NITEM_B registers across the MFMA phase and the epilogue like the halo prefetch does.
    auto issueB = [&](auto D_, const int ch) {
        constexpr int d = decltype(D_)::value;
        const bool sk = ch >= nchunk_main;
        unsigned t_ = tid;
        asm volatile("" : "+v"(t_));
        if (PREC != CCDM_PREC_F32 && !BDMA) {
            // B chunk: [tap][k-step] slabs; skip chunks carry one tap (1x1): only their first KST slabs are meaningful,
            // the passes beyond re-read slab 0 (the load stays unconditional: regB[] stays in registers)
            const char* wq = reinterpret_cast<const char*>(((unsigned long long)(unsigned)__builtin_amdgcn_readlane(T_whi, ch) << 32) |
                                                           (unsigned)__builtin_amdgcn_readlane(T_wlo, ch));
            const unsigned wtap = (unsigned)((sk ? k.cin_pad_skip : k.cin_pad) >> 4) * k.ntiles * 128;
            const unsigned wks = (unsigned)k.ntiles * 128;
            const bool skwc = SKW && skw && sk;                          // wide skip chunk: 2 k-steps of the one tap
            const unsigned nslab = skwc ? (unsigned)(CKS / 16) : (sk ? KST : NTAP * KST);
            const unsigned remb = (B_MULTI ? t_ % G : t_) << 4;      // lane offset, 32-bit (hoisted as a 64-bit pair it defeats the saddr form)
#pragma unroll
            for (int i = 0; i < NITEM_B; ++i) {
                unsigned ts = B_MULTI ? i * MB + tgB : (unsigned)(i / DB);          // wave-uniform
                const unsigned rem = B_MULTI ? remb : remb + (unsigned)(NT * (i % DB) * 16);
                ts = ts < nslab ? ts : 0u;
                const unsigned slab = skwc ? ts * wks : (ts / KST) * wtap + (ts % KST) * wks;
                regB[0][i] = load16_uniform_base(wq + ((size_t)slab << 4), rem);
            }
        }
    };
    // ---- issueB_dma (BDMA): the same fragments by LDS-DMA (global_load_lds, 1 KB per wave instruction): the chunk's B image in LDS is
    //      the packed layout itself — [tap][k-step][n-tile][hi|lo][64 lanes] x 16 B, lane-linear — so a fragment slab needs no
    //      register, no ds_write and no vector instruction besides its request.  Issued behind barrier A (the previous matrix phase
    //      has released the buffer), it lands while the block commits the halo tile; the barrier in front of the matrix phase waits
    //      for it (an LDS-DMA counts on vmcnt).
    auto issueB_dma = [&](const int ch) {
        const bool sk = ch >= nchunk_main;
        const char* wq = reinterpret_cast<const char*>(((unsigned long long)(unsigned)__builtin_amdgcn_readlane(T_whi, ch) << 32) |
                                                       (unsigned)__builtin_amdgcn_readlane(T_wlo, ch));
        const unsigned wtap = (unsigned)((sk ? k.cin_pad_skip : k.cin_pad) >> 4) * k.ntiles * 128;
        const unsigned wks = (unsigned)k.ntiles * 128;
        const bool skwc = SKW && skw && sk;
        const unsigned nslab = skwc ? (unsigned)(CKS / 16) : (sk ? KST : NTAP * KST);
        constexpr int UPS = G / 64;                                   // 1 KB units per slab
        constexpr int NWV = WAVES * KSP;
        constexpr int MAXU = (NTAP * KST * UPS + NWV - 1) / NWV;
        const unsigned nunit = nslab * UPS;
        const unsigned dst0 = (unsigned)(size_t)(skwc ? halo_b + TH * TW * PIXS : reinterpret_cast<char*>(ldsB));
        unsigned lane_ = lane;
        asm volatile("" : "+v"(lane_));
#pragma unroll
        for (int i = 0; i < MAXU; ++i) {
            const unsigned u = (unsigned)wave_all + (unsigned)(i * NWV);        // wave-uniform
            if (u < nunit) {
                const unsigned ts = u / UPS, part = u % UPS;
                const unsigned slab = skwc ? ts * wks : (ts / KST) * wtap + (ts % KST) * wks;
                const char* src = wq + (((size_t)slab + part * 64) << 4) + (lane_ << 4);
                __builtin_amdgcn_global_load_lds(reinterpret_cast<const __attribute__((address_space(1))) void*>(reinterpret_cast<unsigned long long>(src)),
                                                 reinterpret_cast<__attribute__((address_space(3))) void*>(dst0 + u * 1024u), 16, 0, 0);
            }
        }
    };
    // ---- commit: registers -> affine -> SiLU -> [fp16 hi|lo split] -> LDS (zero where padded) ----
    // The chunk's transform is uniform (GroupNorm / SiLU apply to the main segment only): one specialised, branch-free
    // body per combination; padding is a select, not a branch.
    auto commit_body = [&](auto D_, auto GN_, auto ACT_, int c0) {
        constexpr int d = decltype(D_)::value;
        constexpr bool GN = decltype(GN_)::value, ACT = decltype(ACT_)::value;
        unsigned t_ = tid;
        asm volatile("" : "+v"(t_));
        const int tq = t_ % QPP, px = (t_ / QPP) % CW;
        float2 t0 = make_float2(1.f, 0.f), t1 = t0, t2 = t0, t3 = t0;
        if (GN) { const int c = c0 + 4 * tq; t0 = ab[c]; t1 = ab[c + 1]; t2 = ab[c + 2]; t3 = ab[c + 3]; }
        // F16X3: activations are pre-scaled by 2^4 (exact; undone through the weight-scale table) so that the lo half of values
        // down to ~2e-3 stays in fp16's normal range.  The factor rides along for free: folded into the GroupNorm affine
        // when there is no activation, into the sigmoid's reciprocal when there is (16/(1+e) = 1/(2^-4 + e*2^-4), exact
        // rescaling of the same roundings).
        constexpr bool F16 = PREC != CCDM_PREC_F32;
        constexpr float PS = F16 ? ACT_PRESCALE : 1.0f;
        if (GN && !ACT && F16) { t0.x *= PS; t0.y *= PS; t1.x *= PS; t1.y *= PS; t2.x *= PS; t2.y *= PS; t3.x *= PS; t3.y *= PS; }
        // (packed fp32 pairs — v_pk_fma_f32 / v_pk_add_f32 / v_pk_mul_f32 for the affine, the sigmoid's argument and denominator and
        //  the final product, 6 instructions instead of 12 per item — were measured 0.7 % SLOWER on the whole step: kept scalar)
        auto act = [&](const float x) {
            if (!ACT) return (GN || !F16) ? x : x * PS;
            // x * sigmoid(x) * PS with v_exp_f32 / v_rcp_f32 (1 ulp each); limits: x -> -inf gives -0, x -> +inf gives PS*x
            return x * __builtin_amdgcn_rcpf(1.0f / PS + __builtin_amdgcn_exp2f(fmaf(x, -1.4426950408889634f, F16 ? -4.0f : 0.0f)));
        };
        static_assert(ACT_PRESCALE == 16.0f, "the exp2 bias above is log2(ACT_PRESCALE)");
        // MASKED = false: the item is known to lie inside the image (core columns of an image whose width is a multiple of the
        // tile, rows tested by the caller): no padding select at all.
        auto put = [&](auto MASKED_, const f32x4 r, const bool ok, const int hp) {
            constexpr bool MASKED = decltype(MASKED_)::value;
            float4 v = make_float4(r[0], r[1], r[2], r[3]);
            if (GN) { v.x = fmaf(v.x, t0.x, t0.y); v.y = fmaf(v.y, t1.x, t1.y); v.z = fmaf(v.z, t2.x, t2.y); v.w = fmaf(v.w, t3.x, t3.y); }
            v.x = act(v.x); v.y = act(v.y); v.z = act(v.z); v.w = act(v.w);
            if (PREC == CCDM_PREC_F32) {
                float* d = halo + hp * 33 + 4 * tq;
                d[0] = ok ? v.x : 0.f; d[1] = ok ? v.y : 0.f; d[2] = ok ? v.z : 0.f; d[3] = ok ? v.w : 0.f;
            } else {
                // fp16 hi/lo split: x = hi + lo + O(2^-22 |x|); both halves round-to-nearest; full split precision holds for
                // 2e-3 <= |x| <= 4094 (below: absolute error <= 2^-29).  A value beyond fp16's range is NOT clipped: its hi half
                // becomes inf, lo = x - inf = -inf, and every output it reaches is NaN — the step epilogue raises the engine's
                // sticky range flag on it and the host falls back to the exact-fp32 kernels (include/ccdm_hip.h).  The median
                // below only zeroes padding: its bound is 0 there and infinite elsewhere (one select per item; lo = 0 - 0 follows).
                if (MASKED) {
                    const float lim = ok ? __builtin_inff() : 0.f;
                    v.x = __builtin_amdgcn_fmed3f(v.x, -lim, lim); v.y = __builtin_amdgcn_fmed3f(v.y, -lim, lim);
                    v.z = __builtin_amdgcn_fmed3f(v.z, -lim, lim); v.w = __builtin_amdgcn_fmed3f(v.w, -lim, lim);
                }
                typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
                u32x2 hi, lo;
                unsigned h0, l0, h1, l1;
                char* d = halo_b + hp * PIXB + 8 * tq;
                if constexpr (PREC == CCDM_PREC_F16) {          // single-pass mode: the hi halves only (the lo plane of the tile is never read)
                    asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(h0) : "v"(v.x), "v"(v.y));
                    asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(h1) : "v"(v.z), "v"(v.w));
                    hi[0] = h0; hi[1] = h1;
                    *reinterpret_cast<u32x2*>(d) = hi;
                } else {
                    split2_f16(v.x, v.y, h0, l0);
                    split2_f16(v.z, v.w, h1, l1);
                    hi[0] = h0; hi[1] = h1; lo[0] = l0; lo[1] = l1;
                    *reinterpret_cast<u32x2*>(d) = hi;
                    *reinterpret_cast<u32x2*>(d + 2 * CK) = lo;
                }
            }
        };
        auto put_zero = [&](const int hp) {          // a halo row outside the image (wave-uniform): zeros, no arithmetic
            typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
            u32x2 z;
            z[0] = 0u; z[1] = 0u;
            char* d = halo_b + hp * PIXB + 8 * tq;
            *reinterpret_cast<u32x2*>(d) = z;
            *reinterpret_cast<u32x2*>(d + 2 * CK) = z;
        };
        if constexpr (ROWS) {
            const int hp0 = rip * HWt + PAD + px;
            if (ROW_UNIFORM && PREC != CCDM_PREC_F32 && k.core_unmasked) {
                // the row test is wave-uniform (scalar branch); inside the image nothing is masked
#pragma unroll
                for (int i = 0; i < NCORE; ++i)
                    if ((i + 1) * RPP <= HHt || rip + i * RPP < HHt) {
                        if (__builtin_amdgcn_readfirstlane((rowmask[d] >> i) & 1u)) put(std::false_type{}, reg[d][i], true, hp0 + i * RPP * HWt);
                        else put_zero(hp0 + i * RPP * HWt);
                    }
            } else {
#pragma unroll
                for (int i = 0; i < NCORE; ++i)
                    if ((i + 1) * RPP <= HHt || rip + i * RPP < HHt)
                        put(std::true_type{}, reg[d][i], xok[d] & (((rowmask[d] >> i) & 1u) != 0u), hp0 + i * RPP * HWt);
            }
#pragma unroll
            for (int j = 0; j < NEDGE; ++j) {
                const unsigned e = t_ + j * NT;
                if (e < (unsigned)EDGE_ITEMS) {
                    const unsigned side = (e / QPP) % ECOLS, row = e / (ECOLS * QPP);
                    const int hx = side < (unsigned)PAD ? (int)side : CW + (int)side;
                    put(std::true_type{}, reg[d][NCORE + j], ((evalid[d] >> j) & 1u) != 0u, (int)row * HWt + hx);
                }
            }
        } else {
#pragma unroll
            for (int i = 0; i < NITEM; ++i) {
                const unsigned item = t_ + i * NT;
                if (item < (unsigned)(HP * QPP)) put(std::true_type{}, reg[d][i], ((valid[d] >> i) & 1u) != 0u, (int)(item / QPP));
            }
        }
        if (PREC != CCDM_PREC_F32 && !BDMA && !CCDM_DBG(32)) {
#pragma unroll
            for (int i = 0; i < NITEM_B; ++i) {
                const int j = (int)t_ + i * NT;
                if ((i + 1) * NT <= NB4 || j < NB4) ldsB[j] = regB[0][i];
            }
        }
    };
    // wide skip chunk: raw values (x 2^4), split, core pixel p = row * TW + px at 144-byte pitch; its two fragment slabs behind them
    auto commit_skipw = [&](auto D_) {
        constexpr int d = decltype(D_)::value;
        if constexpr (SKW) {
            unsigned t_ = tid;
            asm volatile("" : "+v"(t_));
            const int tq = t_ % QPS, px = t_ / QPS;
            typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
#pragma unroll
            for (int i = 0; i < NITEM_S; ++i) {
                const bool ok = xok[d] & (((rowmask[d] >> i) & 1u) != 0u);
                const f32x4 r = reg[d][i];
                const float lim = ok ? __builtin_inff() : 0.f;         // padding -> 0 (one select per item, as in the main path)
                const float v0 = __builtin_amdgcn_fmed3f(r[0] * ACT_PRESCALE, -lim, lim), v1 = __builtin_amdgcn_fmed3f(r[1] * ACT_PRESCALE, -lim, lim);
                const float v2 = __builtin_amdgcn_fmed3f(r[2] * ACT_PRESCALE, -lim, lim), v3 = __builtin_amdgcn_fmed3f(r[3] * ACT_PRESCALE, -lim, lim);
                u32x2 hi, lo;
                unsigned h0, l0, h1, l1;
                split2_f16(v0, v1, h0, l0);
                split2_f16(v2, v3, h1, l1);
                hi[0] = h0; hi[1] = h1; lo[0] = l0; lo[1] = l1;
                char* dd = halo_b + (i * TW + px) * PIXS + 8 * tq;
                *reinterpret_cast<u32x2*>(dd) = hi;
                *reinterpret_cast<u32x2*>(dd + 2 * CKS) = lo;
            }
            if (!BDMA) reinterpret_cast<f32x4*>(halo_b + TH * TW * PIXS)[t_] = regB[0][0];      // [k-step][hi|lo][64 lanes] x 16 B: NT = 2 * 128 items
        }
    };
    auto commit = [&](auto D_, const int ch) {
        const bool sk = ch >= nchunk_main;
        if (SKW && skw && sk) { commit_skipw(D_); return; }
        const int c0 = (sk ? ch - nchunk_main : ch) * CK;
        const bool gn = has_gn && !sk, act = a.act == CCDM_ACT_SILU && !sk;
        if (gn && act) commit_body(D_, std::true_type{}, std::true_type{}, c0);
        else if (gn) commit_body(D_, std::true_type{}, std::false_type{}, c0);
        else if (act) commit_body(D_, std::false_type{}, std::true_type{}, c0);
        else commit_body(D_, std::false_type{}, std::false_type{}, c0);
    };

    f32x16 acc[MI][NI];
    CCDM_STAMP(1);
    // (tile, chunk) walk of this block — tile = slice, slice + slices, ... — carried as scalar counters: the current
    // iteration's (chunk, ty, tx) and, one step ahead, the prefetch's (no divisions in the loop)
    const int adv_y = k.slices / k.tiles_x, adv_x = k.slices % k.tiles_x;
    auto advance = [&](int& ch, int& ty, int& tx) {
        if (++ch == nchunk) {
            ch = 0; tx += adv_x; ty += adv_y;
            if (tx >= k.tiles_x) { tx -= k.tiles_x; ++ty; }
        }
    };
    int chunk = 0, cur_ty = slice / k.tiles_x, cur_tx = slice % k.tiles_x;
    int pf_ch = 0, pf_ty = cur_ty, pf_tx = cur_tx;
    // prologue: the first halo request
    // (a launch always has n_iter >= 1: slices <= tiles)
    issue(std::integral_constant<int, 0>{}, pf_ch, pf_ty, pf_tx);
    advance(pf_ch, pf_ty, pf_tx);
    __builtin_amdgcn_sched_barrier(0);
    CCDM_STAMP(13);
    // the small loads issued at the top are consumed here, behind the first halo request
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) {
        const bool cv = tile_of(ni) * 32 + (lane & 31) < a.Cout;
        float add = a.bias ? raw_bias[ni] : 0.f;
        if (a.emb_off >= 0) add += raw_emb[ni];             // (conv + bias) + emb == conv + (bias + emb) up to 1 ulp
        epi_add[ni] = (cv && krow == 0) ? add : 0.f;        // bias (+emb) enters once, through row group 0's partial
        if (!cv) epi_wsc[ni] = 1.0f;
    }
    // GroupNorm's (scale, shift) table for this sample, from the prefetched partials (the halo region of LDS is free until the
    // first commit, which sits behind the loop-top barrier)
    if (has_gn) gn_affine_block(a, n, emb_row, gpf, reinterpret_cast<f64x2*>(halo_b), ab);
    CCDM_STAMP(14);
    // one iteration = one (tile, chunk); D_ = the register set it consumes (static: the loop below is unrolled by DEPTH)
    auto iterate = [&](auto D_) {
        if (chunk == 0) {
#pragma unroll
            for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                for (int ni = 0; ni < NI; ++ni)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.0f;
        }
        CCDM_STAMP(2);
        if (!CCDM_DBG(4) && !CCDM_DBG(32)) issueB(D_, chunk);
        if (!CCDM_DBG(256)) __syncthreads();          // previous MFMA phase has finished reading LDS (and ab[] is visible)
        CCDM_STAMP(3);
        if constexpr (BDMA) {
            // (the halo registers were requested an iteration ago; naming them here makes the compiler place its wait for them in
            //  front of the DMA requests — with a DMA in flight it would otherwise wait for EVERYTHING at the commit's first use)
#pragma unroll
            for (int i = 0; i < NITEM_R; ++i) asm volatile("" : "+v"(reg[0][i]));
            if (!CCDM_DBG(4) && !CCDM_DBG(32)) issueB_dma(chunk);
        }
        if (!CCDM_DBG(2)) commit(D_, chunk);
        CCDM_STAMP(4);
        if (!CCDM_DBG(256)) __syncthreads();
        CCDM_STAMP(5);
        // next tile-chunk's HBM reads fly during the MFMA phase (after the last iteration this requests a tile past the
        // slice's last one: addresses are clamped into the tensor, the data is never committed — harmless, branch-free)
        if (!CCDM_DBG(4)) {       // refill the set just committed: iteration it + DEPTH
            issue(D_, pf_ch, pf_ty, pf_tx);
            advance(pf_ch, pf_ty, pf_tx);
        }

        CCDM_STAMP(6);
        const bool skc = chunk >= nchunk_main;                   // uniform
        const int c0 = (skc ? chunk - nchunk_main : chunk) * CK;
        if (CCDM_DBG(1)) {
        } else if (PREC == CCDM_PREC_F32) {
            // taps x 16 k-steps of v_mfma_f32_32x32x2_f32; B: [tap][cin_pad/2][ntiles][64] floats
            const float* wc = reinterpret_cast<const float*>(skc ? a.skip_w : a.w) + ((size_t)(c0 >> 1) * k.ntiles + nt0) * 64 + lane;
            const size_t wtap = (size_t)(k.cin_pad >> 1) * k.ntiles * 64;
#pragma unroll
            for (int tap = 0; tap < KS * KS; ++tap) {
                if (skc && tap != (KS * KS) / 2) continue;       // skip segment: centre tap only, its weights are tap 0
                const int toff = ((tap / KS) * HWt + (tap % KS)) * 33;
                const float* wt = wc + (skc ? 0 : tap * wtap);
#pragma unroll 4
                for (int kk = 0; kk < CK / 2; ++kk) {
                    float av[MI];
#pragma unroll
                    for (int mi = 0; mi < MI; ++mi) av[mi] = halo[base[mi] + toff + 2 * kk];
#pragma unroll
                    for (int ni = 0; ni < NI; ++ni) {
                        const float bv = wt[((size_t)kk * k.ntiles + ni) * 64];
#pragma unroll
                        for (int mi = 0; mi < MI; ++mi)
                            acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[mi], bv, acc[mi][ni], 0, 0, 0);
                    }
                }
            }
        } else {
            // taps x {lo*hi, hi*lo, hi*hi} of v_mfma_f32_32x32x16_f16 (KST 16-channel k-steps per chunk);
            // A: halo tile, B: [tap][k-step][ni][hi|lo][lane] fragments, both in LDS.  The tap walk is straight-line code
            // (the skip segment's single centre tap is its own copy), so the LDS reads of the next tap are scheduled
            // under the MFMAs of the current one.
            const f16x8* bq = reinterpret_cast<const f16x8*>(ldsB) + lane;
            // One step = one (tap, 16-channel k-step): 2*MI A fragments + 2*NI B fragments from LDS, 3*MI*NI MFMAs.
            // The fragments of step s+1 are requested before the MFMAs of step s are issued (two register sets, static
            // indices after unrolling): the LDS round trip — ~130+ cycles that an in-order wave otherwise spends idle
            // in front of every tap — runs under the previous step's matrix work.
            constexpr bool PF = MI * NI <= 2;  // the second fragment set fits the register budget (not beside the wide skip chunks' 8-item set)
            f16x8 ah[2][MI], al[2][MI], bh[2][NI], bl[2][NI];
            auto frag_load_a = [&](const int buf, const int toff, const int ks) {
#pragma unroll
                for (int mi = 0; mi < MI; ++mi) {
                    const char* p = halo_b + base[mi] + toff + 32 * ks;
                    ah[buf][mi] = *reinterpret_cast<const f16x8*>(p);
                    al[buf][mi] = *reinterpret_cast<const f16x8*>(p + 2 * CK);
                }
            };
            auto frag_load = [&](const int buf, const int toff, const int bt, const int ks) {
                frag_load_a(buf, toff, ks);
#pragma unroll
                for (int ni = 0; ni < NI; ++ni) {
                    bh[buf][ni] = bq[((bt * KST + ks) * NI + ni) * 128];
                    bl[buf][ni] = bq[((bt * KST + ks) * NI + ni) * 128 + 64];
                }
            };
            auto frag_mfma = [&](const int buf) {
#pragma unroll
                for (int ni = 0; ni < NI; ++ni) {
                    if constexpr (PREC != CCDM_PREC_F16) {          // (the opt-in single-pass mode multiplies the hi halves only)
#pragma unroll
                        for (int mi = 0; mi < MI; ++mi) acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[buf][mi], bh[buf][ni], acc[mi][ni], 0, 0, 0);
#pragma unroll
                        for (int mi = 0; mi < MI; ++mi) acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[buf][mi], bl[buf][ni], acc[mi][ni], 0, 0, 0);
                    }
#pragma unroll
                    for (int mi = 0; mi < MI; ++mi) acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[buf][mi], bh[buf][ni], acc[mi][ni], 0, 0, 0);
                }
            };
            // walk(NTAP, tap -> (toff, bt)): NTAP*KST steps, software-pipelined one step deep
            auto walk = [&](auto NTAP_, auto&& toff_of, auto&& bt_of) {
                constexpr int NSTEP = decltype(NTAP_)::value * KST;
                if constexpr (PF) {
                    frag_load(0, toff_of(0), bt_of(0), 0);
#pragma unroll
                    for (int st = 0; st < NSTEP; ++st) {
                        if (st + 1 < NSTEP) frag_load((st + 1) & 1, toff_of((st + 1) / KST), bt_of((st + 1) / KST), (st + 1) % KST);
                        __builtin_amdgcn_sched_barrier(0);      // keep the requests in front of the MFMAs (the scheduler sinks them to save registers)
                        frag_mfma(st & 1);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                } else {
#pragma unroll
                    for (int st = 0; st < NSTEP; ++st) {
                        frag_load(0, toff_of(st / KST), bt_of(st / KST), st % KST);
                        frag_mfma(0);
                    }
                }
            };
            if (skc) {
                // skip segment: centre tap only, its weights staged as B slot 0 (tap split: the centre row's group).  One code path for the
                // halo-tile form (CK channels, pixel pitch PIXB, origin at the tile's first core pixel) and the core-only form (SKW: 32
                // channels, pitch PIXS, origin 0, its fragments behind the pixels): pitch, offsets and k-step count are run-time uniform.
                if (KSP == 1 || krow == KS / 2) {
                    const bool w_ = SKW && skw;
                    const int nks = w_ ? CKS / 16 : KST, lo_off = w_ ? 2 * CKS : 2 * CK;
                    const f16x8* bsk = w_ ? reinterpret_cast<const f16x8*>(halo_b + TH * TW * PIXS) + lane : bq;
                    int pbs[MI];
#pragma unroll
                    for (int mi = 0; mi < MI; ++mi)
                        pbs[mi] = w_ ? ((wave * MI + mi) * 32 + (lane & 31)) * PIXS + (lane >> 5) * 16 : base[mi] + (PAD * HWt + PAD) * PIXB;
#pragma unroll 1
                    for (int ks = 0; ks < nks; ++ks) {
#pragma unroll
                        for (int mi = 0; mi < MI; ++mi) {
                            const char* pa = halo_b + pbs[mi] + 32 * ks;
                            ah[0][mi] = *reinterpret_cast<const f16x8*>(pa);
                            al[0][mi] = *reinterpret_cast<const f16x8*>(pa + lo_off);
                        }
#pragma unroll
                        for (int ni = 0; ni < NI; ++ni) {
                            bh[0][ni] = bsk[(ks * NI + ni) * 128];
                            bl[0][ni] = bsk[(ks * NI + ni) * 128 + 64];
                        }
                        frag_mfma(0);
                    }
                }
            } else if (UP2 && NI == 4) {
                // all four phases: halo position (r, c) of the 3x3 neighbourhood is tap (r - dy, c - dx) of phase (dy, dx) when that
                // lies in its 2x2 window — 16 (position, phase) products per k-step, each A fragment fetched once
#pragma unroll
                for (int pos = 0; pos < 9; ++pos)
#pragma unroll
                    for (int ks = 0; ks < KST; ++ks) {
                        const int r = pos / 3, c = pos % 3;
                        frag_load_a(0, (r * HWt + c) * PIXB, ks);
#pragma unroll
                        for (int ni = 0; ni < NI; ++ni) {
                            const int ta = r - (ni >> 1), tb = c - (ni & 1);
                            if (ta < 0 || ta > 1 || tb < 0 || tb > 1) continue;
                            const int bt = ta * 2 + tb;
                            const f16x8 wh = bq[((bt * KST + ks) * NI + ni) * 128], wl = bq[((bt * KST + ks) * NI + ni) * 128 + 64];
#pragma unroll
                            for (int mi = 0; mi < MI; ++mi) {
                                acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[0][mi], wh, acc[mi][ni], 0, 0, 0);
                                acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[0][mi], wl, acc[mi][ni], 0, 0, 0);
                                acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[0][mi], wh, acc[mi][ni], 0, 0, 0);
                            }
                        }
                    }
            } else if (UP2) {
                walk(std::integral_constant<int, 4>{}, [&](int t) { return ((t >> 1) * HWt + (t & 1)) * PIXB; }, [&](int t) { return t; });
            } else if (KSP > 1) {
                // tap split: this wave group owns kernel row `krow`
                walk(std::integral_constant<int, KS>{}, [&](int u) { return (krow * HWt + u) * PIXB; }, [&](int u) { return krow * KS + u; });
            } else {
                walk(std::integral_constant<int, KS * KS>{}, [&](int tap) { return ((tap / KS) * HWt + (tap % KS)) * PIXB; }, [&](int tap) { return tap; });
            }
        }

        CCDM_STAMP(7);
        if (chunk == nchunk - 1) {
            // ---- epilogue: (x 2^-e) + bias (+ emb) (+ residual), store NHWC, accumulate output statistics ----
            const int oy0 = cur_ty * TH, ox0 = cur_tx * TW;
            int lane_ = lane;
            asm volatile("" : "+v"(lane_));
            if (fast_epi) {
                // ---- fast path: accumulators -> wave-private LDS rows [pixel][32 ch] -> float4 per lane
                //      (8 lanes cover one pixel's 128-byte row: residual loads and stores move 16 B per lane) ----
                if (!CCDM_DBG(256)) __syncthreads();                 // every wave is done reading the A/B tiles
                CCDM_STAMP(9);
                const int cq = lane_ & 7, prow = lane_ >> 3;
#pragma unroll
                for (int ni = 0; ni < NI; ++ni) {
                    if (KSP > 1 && ni > 0) __syncthreads();        // the partials of the previous n-tile have been consumed
                    const int co4 = tile_of(ni) * 32 + 4 * cq;
                    const int dy = phase_of(ni) >> 1, dx = phase_of(ni) & 1;
                    const bool cv4 = co4 < a.Cout;
                    // Row j of this wave's float4 pass is output pixel (oy, ox0 + cx + prow): oy and cx are wave-uniform /
                    // compile-time (8 | TW), so for a tile that lies fully inside the image and the channel range the
                    // residual loads and the stores are (scalar row base) + (one thread-constant offset): no per-row VALU.
                    // Ragged tiles take the checked copy.
                    const int eH = UP2 ? Hc : aHout, eW = UP2 ? Wc : aWout;      // extent of the tile space
                    const bool full = oy0 + TH <= eH && ox0 + TW <= eW && tile_of(ni) * 32 + 32 <= aCout;   // uniform
                    constexpr int RS_FIRST = MI * 4 > 4 ? 4 : MI * 4;
                    f32x4 rs[MI * 4];
                    unsigned lane_off = ((unsigned)(UP2 ? 2 * prow : prow) * (unsigned)aCout + (unsigned)co4) << 2;
                    asm volatile("" : "+v"(lane_off));      // stays a 32-bit offset (hoisted out of the tile loop it becomes a 64-bit pair and the saddr form is lost)
                    auto row_of = [&](const int j) { return oy0 + wave * (MI * 32 / TW) + (j * 8) / TW; };       // uniform
                    auto row_base = [&](const int j) {      // byte offset of pixel (oy, ox0 + cx), channel 0, within the sample
                        if (UP2) return (unsigned)(((2 * row_of(j) + dy) * aWout + 2 * (ox0 + (j * 8) % TW) + dx) * aCout) << 2;
                        return (unsigned)((row_of(j) * aWout + ox0 + (j * 8) % TW) * aCout) << 2;
                    };
                    auto pix_of = [&](const int oy, const int ox) {      // checked copy: pixel index within the sample
                        return UP2 ? (unsigned)((2 * oy + dy) * aWout + 2 * ox + dx) : (unsigned)(oy * aWout + ox);
                    };
                    // residual rows: the first half is requested before the transpose (its latency hides behind the LDS
                    // writes), the second half behind the first half's stores — all of them live at once together with the
                    // accumulators and the next tile's prefetch would not fit the registers
                    float t1[4] = {0.f, 0.f, 0.f, 0.f}, t2[4] = {0.f, 0.f, 0.f, 0.f};
                    auto epi_ni = [&](auto FULL_, auto RESID_) {
                    constexpr bool FULL = decltype(FULL_)::value, RESID = decltype(RESID_)::value;
                    auto load_resid = [&](const int j0, const int j1) {
#pragma unroll
                        for (int j = j0; j < j1; ++j) {
                            if (KSP > 1 && j % KSP != krow) continue;
                            if (FULL) rs[j] = load16_uniform_base(reinterpret_cast<const char*>(residn) + row_base(j), lane_off);
                            else {
                                const int oy = min(row_of(j), eH - 1), ox = min(ox0 + (j * 8) % TW + prow, eW - 1);
                                const int cc = min(co4, aCout - 4);
                                rs[j] = *reinterpret_cast<const f32x4*>(reinterpret_cast<const char*>(residn) +
                                                                        ((pix_of(oy, ox) * (unsigned)aCout + (unsigned)cc) << 2));
                            }
                        }
                    };
                    if (RESID) load_resid(0, RS_FIRST);
                    {
                        const float add = epi_add[ni], wsc = epi_wsc[ni];
#pragma unroll
                        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                            for (int r = 0; r < 16; ++r) {
                                const int pl = mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane_ >> 5);
                                epi[pl * EPS + (lane_ & 31)] = PREC == CCDM_PREC_F32 ? acc[mi][ni][r] + add : fmaf(acc[mi][ni][r], wsc, add);   // wsc is a power of two: the product is exact, fma == mul + add
                            }
                    }
                    if (KSP > 1) __syncthreads();                  // all row groups' partials are in LDS
                    CCDM_STAMP(10);
                    const float* epi0 = reinterpret_cast<const float*>(halo_b) + wave * (MI * 32 * EPS);   // row group 0 of this sub-tile
                    {
#pragma unroll
                        for (int j = 0; j < MI * 4; ++j) {
                            if (j == RS_FIRST && RESID) load_resid(RS_FIRST, MI * 4);   // behind the first half's stores
                            if (KSP > 1 && j % KSP != krow) continue;   // the row groups share the final pass
                            const int pl = j * 8 + prow;
                            f32x4 v = *reinterpret_cast<const f32x4*>(epi0 + pl * EPS + 4 * cq);
#pragma unroll
                            for (int g = 1; g < KSP; ++g)               // fixed order: row 0 + row 1 + row 2
                                v += *reinterpret_cast<const f32x4*>(epi0 + g * (WAVES * MI * 32 * EPS) + pl * EPS + 4 * cq);
                            if (RESID) v += rs[j];
                            if (FULL) {
                                if (!CCDM_DBG(8)) store16_uniform_base(reinterpret_cast<char*>(outn) + row_base(j), lane_off, v);
#pragma unroll
                                for (int e = 0; e < 4; ++e) { t1[e] += v[e]; t2[e] = fmaf(v[e], v[e], t2[e]); }
                            } else {
                                const int oy = row_of(j), ox = ox0 + (j * 8) % TW + prow;
                                if (cv4 && oy < eH && ox < eW) {
                                    if (!CCDM_DBG(8))
                                        *reinterpret_cast<f32x4*>(reinterpret_cast<char*>(outn) +
                                                                  ((pix_of(oy, ox) * (unsigned)aCout + (unsigned)co4) << 2)) = v;
#pragma unroll
                                    for (int e = 0; e < 4; ++e) { t1[e] += v[e]; t2[e] = fmaf(v[e], v[e], t2[e]); }
                                }
                            }
                        }
                    }
                    };
                    // one straight-line copy per (full tile, residual) combination: uniform branches taken once
                    if (a.resid) { if (full) epi_ni(std::true_type{}, std::true_type{}); else epi_ni(std::false_type{}, std::true_type{}); }
                    else { if (full) epi_ni(std::true_type{}, std::false_type{}); else epi_ni(std::false_type{}, std::false_type{}); }
#pragma unroll
                    for (int e = 0; e < 4; ++e) { s1[ni][e] += t1[e]; s2[ni][e] += t2[e]; }
                    CCDM_STAMP(11);
                }
            } else {
#pragma unroll
            for (int ni = 0; ni < NI; ++ni) {
                const int co = (nt0 + ni) * 32 + (lane_ & 31);
                const bool cv = co < a.Cout;
                const float add = epi_add[ni], wsc = epi_wsc[ni];
                // per-tile partial statistics in fp32 (<= 32 values per lane), folded into the fp64 running sums once per tile
                float t1 = 0.f, t2 = 0.f;
#pragma unroll
                for (int mi = 0; mi < MI; ++mi) {
                    // pixel of accumulator register r: p = msub*32 + (r&3) + 8*(r>>2) + 4*(lane>>5)
                    const int pb = (wave * MI + mi) * 32 + 4 * (lane_ >> 5);
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int p = pb + (r & 3) + 8 * (r >> 2);
                        const int oy = oy0 + p / TW, ox = ox0 + p % TW;
                        if (cv && oy < a.Hout && ox < a.Wout) {
                            const size_t idx = ((size_t)(n * a.Hout + oy) * a.Wout + ox) * a.Cout + co;
                            float v = (PREC == CCDM_PREC_F32 ? acc[mi][ni][r] : acc[mi][ni][r] * wsc) + add;
                            if (a.resid) v += a.resid[idx];
                            if (!CCDM_DBG(8)) a.out[idx] = v;
                            t1 += v;
                            t2 = fmaf(v, v, t2);
                        }
                    }
                }
                s1[ni][0] += t1;
                s2[ni][0] += t2;
            }
            }
        }
        advance(chunk, cur_ty, cur_tx);
    };
    for (int it = 0; it < n_iter; ++it) iterate(std::integral_constant<int, 0>{});

    CCDM_STAMP(8);
    if (CCDM_DBG(16) && blockIdx.x == gridDim.x / 2 && blockIdx.y == 0 && tid == 0) g_timeline[1023] = tl;
    if (a.out_stats) {
        // fold the lanes that hold the same channel, then the block's waves; fixed order everywhere
        __syncthreads();
        double* red = reinterpret_cast<double*>(halo_b);     // [WAVES*KSP][NI][32][2]
        if (fast_epi) {
#pragma unroll
            for (int ni = 0; ni < NI; ++ni)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    double v1 = (double)s1[ni][e], v2 = (double)s2[ni][e];
#pragma unroll
                    for (int off = 8; off < 64; off <<= 1) { v1 += __shfl_xor(v1, off); v2 += __shfl_xor(v2, off); }
                    if (lane < 8) {
                        red[((wave_all * NI + ni) * 32 + 4 * lane + e) * 2 + 0] = v1;
                        red[((wave_all * NI + ni) * 32 + 4 * lane + e) * 2 + 1] = v2;
                    }
                }
        } else {
#pragma unroll
            for (int ni = 0; ni < NI; ++ni) {
                const double m1 = (double)s1[ni][0], m2 = (double)s2[ni][0];
                const double o1 = __shfl_xor(m1, 32), o2 = __shfl_xor(m2, 32);
                if (lane < 32) {
                    red[((wave_all * NI + ni) * 32 + lane) * 2 + 0] = m1 + o1;
                    red[((wave_all * NI + ni) * 32 + lane) * 2 + 1] = m2 + o2;
                }
            }
        }
        __syncthreads();
        // (sub-pixel form with all four phases in the block: they are n-tiles of ONE 32-channel tile and fold into one partial,
        //  phase 0..3 in order; with one phase per block each phase leaves its own slot)
        constexpr int NFOLD = UP2 && NI == 4 ? 4 : 1;
        for (int i = tid; i < NI / NFOLD * 32; i += NT) {
            const int ni = (i >> 5) * NFOLD, l = i & 31;
            double t1 = 0.0, t2 = 0.0;
            for (int f = 0; f < NFOLD; ++f)
                for (int w = 0; w < WAVES * KSP; ++w) {
                    t1 += red[((w * NI + ni + f) * 32 + l) * 2 + 0];
                    t2 += red[((w * NI + ni + f) * 32 + l) * 2 + 1];
                }
            const int co = tile_of(ni) * 32 + l;
            if (co < a.Cout) {
                const int nslot = UP2 && NI == 1 ? 4 * k.slices : k.slices, slot = UP2 && NI == 1 ? 4 * slice + phase_of(ni) : slice;
                double* o = a.out_stats + (((size_t)n * nslot + slot) * a.Cout + co) * 2;
                o[0] = t1; o[1] = t2;
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------
// chunk width of the F16X3 path: 32 channels for the small-spatial geometries when the channel counts allow it
static bool tap_split(const ccdm_conv_args& a, const ConvGeo& g);
static int chunk_ck(const ccdm_conv_args& a, const ConvGeo& g) {
    if ((a.prec & 255) == CCDM_PREC_F32) return 32;
    if (a.up == 2 && g.TW == 16) return 16;      // four phases of fragments per chunk: 32 KB of B at 16 channels
    const int C = a.C0 + a.C1, SC = a.SC0 + a.SC1;
    const bool ok32 = g.TW < 32 && a.stride == 1 && C % 32 == 0 && (a.C1 == 0 || a.C0 % 32 == 0) &&
                      (!a.skip0 || (SC % 32 == 0 && (a.SC1 == 0 || a.SC0 % 32 == 0)));
    // 8x8 images: 64 channels per chunk — the kernel there is one round of single-tile blocks whose time is the length of
    // their chain, and every chunk is a barrier-separated round trip; the whole chunk still fits (A 27 KB + B 74 KB for 3x3)
    const bool ok64 = ok32 && g.TW == 8 && C % 64 == 0 && (a.C1 == 0 || a.C0 % 64 == 0) &&
                      (!a.skip0 || (SC % 64 == 0 && (a.SC1 == 0 || a.SC0 % 64 == 0))) &&
                      (a.ksize == 1 || tap_split(a, g));
    return ok64 && !exp_env("CCDM_NO_CK64") ? 64 : (ok32 ? 32 : 16);
}

static bool tap_split(const ccdm_conv_args& a, const ConvGeo& g) {
    // measured: pays at 8x8 images (27 -> 18 us for 128->128), loses at 16x16 (25 -> 33 us with 12-wave blocks)
    return (a.prec & 255) != CCDM_PREC_F32 && a.ksize == 3 && a.stride == 1 && a.up != 2 && g.TW == 8 && (a.Cout & 3) == 0;
}

template <int PREC, int CKT, int KS, int STRIDE, int TH, int TW, int WAVES, int MI>
static int launch_ni(const ConvK& k, int NI, dim3 grid, size_t lds, hipStream_t s) {
    dim3 block(WAVES * 64);
    if constexpr (TW < 32 || (CKT >= 32 && PREC != CCDM_PREC_F32)) {       // narrow tiles always run one n-tile per block (launch_conv)
        hipLaunchKernelGGL((k_conv<PREC, CKT, KS, STRIDE, TH, TW, WAVES, MI, 1>), grid, block, lds, s, k);
    } else {
        switch (NI) {
            case 1: hipLaunchKernelGGL((k_conv<PREC, CKT, KS, STRIDE, TH, TW, WAVES, MI, 1>), grid, block, lds, s, k); break;
            case 2: hipLaunchKernelGGL((k_conv<PREC, CKT, KS, STRIDE, TH, TW, WAVES, MI, 2>), grid, block, lds, s, k); break;
            default: return fail("conv: bad NI %d", NI);
        }
    }
    return 0;
}

template <int PREC, int KS>
static int launch_geo(const ConvK& k, const ConvGeo& g, int NI, int ck, dim3 grid, size_t lds, hipStream_t s) {
    constexpr int CK0 = PREC == CCDM_PREC_F32 ? 32 : 16;
    if (k.a.stride == 2) {
        if constexpr (KS == 3) { if (g.TW == 16) return launch_ni<PREC, CK0, KS, 2, 8, 16, 4, 1>(k, NI, grid, lds, s); }
        return launch_ni<PREC, CK0, KS, 2, 8, 8, 2, 1>(k, NI, grid, lds, s);
    }
    if constexpr (PREC == CCDM_PREC_F16X3 && KS == 3) {
        if (k.a.up == 2) {          // sub-pixel upsample conv: one n-tile (= phase x channel tile) per block
            if (g.TW == 16) hipLaunchKernelGGL((k_conv<PREC, CK0, 3, 1, 8, 16, 4, 1, 4, 1, true>), grid, dim3(256), lds, s, k);      // four phases per block
            else if (ck == 32) hipLaunchKernelGGL((k_conv<PREC, 32, 3, 1, 8, 8, 2, 1, 1, 1, true>), grid, dim3(128), lds, s, k);
            else hipLaunchKernelGGL((k_conv<PREC, CK0, 3, 1, 8, 8, 2, 1, 1, 1, true>), grid, dim3(128), lds, s, k);
            return 0;
        }
    }
    if constexpr (PREC == CCDM_PREC_F16X3 && KS == 3) {
        if (g.TW == 32 && k.skip_wide) {          // fused 1x1 skip in 32-channel core-only chunks
            hipLaunchKernelGGL((k_conv<PREC, CK0, 3, 1, 8, 32, 4, 2, 1, 1, false, true>), grid, dim3(256), lds, s, k);
            return 0;
        }
    }
    if (g.TW == 32) return launch_ni<PREC, CK0, KS, 1, 8, 32, 4, 2>(k, NI, grid, lds, s);
    if (PREC != CCDM_PREC_F32 && KS == 3 && tap_split(k.a, g)) {       // small-spatial 3x3: kernel rows split over 3 wave groups
        constexpr int KSPL = KS == 3 ? 3 : 1;
        if (ck == 64) hipLaunchKernelGGL((k_conv<PREC, 64, KS, 1, 8, 8, 2, 1, 1, KSPL>), grid, dim3(2 * KSPL * 64), lds, s, k);
        else if (ck == 32) hipLaunchKernelGGL((k_conv<PREC, 32, KS, 1, 8, 8, 2, 1, 1, KSPL>), grid, dim3(2 * KSPL * 64), lds, s, k);
        else hipLaunchKernelGGL((k_conv<PREC, CK0, KS, 1, 8, 8, 2, 1, 1, KSPL>), grid, dim3(2 * KSPL * 64), lds, s, k);
        return 0;
    }
    if (PREC != CCDM_PREC_F32 && ck == 64) {
        if constexpr (KS == 1) return launch_ni<PREC, 64, KS, 1, 8, 8, 2, 1>(k, NI, grid, lds, s);
    }
    if (PREC != CCDM_PREC_F32 && ck == 32) {
        if (g.TW == 16) return launch_ni<PREC, 32, KS, 1, 8, 16, 4, 1>(k, NI, grid, lds, s);
        return launch_ni<PREC, 32, KS, 1, 8, 8, 2, 1>(k, NI, grid, lds, s);
    }
    if (g.TW == 16) return launch_ni<PREC, CK0, KS, 1, 8, 16, 4, 1>(k, NI, grid, lds, s);
    return launch_ni<PREC, CK0, KS, 1, 8, 8, 2, 1>(k, NI, grid, lds, s);
}

template <int PREC>
static int launch_prec(const ConvK& k, const ConvGeo& g, int NI, int ck, dim3 grid, size_t lds, hipStream_t s) {
#ifdef CCDM_EXPERIMENT   // compile one instantiation only (register/ISA experiments)
#ifndef CCDM_EXPERIMENT_THREADS
#define CCDM_EXPERIMENT_THREADS 256
#endif
    hipLaunchKernelGGL((k_conv<CCDM_PREC_F16X3, CCDM_EXPERIMENT_CK, 3, 1, CCDM_EXPERIMENT_GEO>), grid, dim3(CCDM_EXPERIMENT_THREADS), lds, s, k);
    return 0;
#else
    if (k.a.ksize == 3) return launch_geo<PREC, 3>(k, g, NI, ck, grid, lds, s);
    return launch_geo<PREC, 1>(k, g, NI, ck, grid, lds, s);
#endif
}

static int conv_slices_default(int tiles, bool up2, int stride);
int conv_slices(int Hout, int Wout, int stride, bool up2 = false, int fine = 0) {
    const ConvGeo g = conv_geo(Hout, Wout, stride, up2, fine != 0);
    const int tiles = cdiv(Hout, g.TH) * cdiv(Wout, g.TW);
    const int dflt = conv_slices_default(tiles, up2, stride);
    // latency slicing (ccdm_conv_args.fine_slices): up to CCDM_STATS_MAX_SLICES one- or two-tile workgroups per sample.  Measured on
    // the LIDC step: batch 8 2.06 -> 1.73 ms per denoise step, batch 64 3.32 -> 3.49 (every block prologue is paid per 2 tiles
    // instead of per 5.3) — hence a mode, not the rule.
    // level 1: up to 32 slices; level 2 (batches of <= 8): up to CCDM_STATS_MAX_SLICES = 64, i.e. one tile per block at 128x128
    // (batch 8 1.47 -> 1.39 ms, batch 4 1.41 -> 1.28 ms per LIDC denoise step; batch 16 loses 4 % with it)
    if (fine) { const int cap = fine >= 2 ? CCDM_STATS_MAX_SLICES : 32; const int f = tiles < cap ? tiles : cap; return f > dflt ? f : dflt; }
    return dflt;
}
static int conv_slices_default(int tiles, bool up2, int stride) {
    // 12 slices for 128x128: with 3 resident blocks per CU, 64 samples x 12 slices = 768 blocks fill the 256 CUs
    // in exactly one round (5.3 tiles per block).  Larger images keep that work per block — one slice per 5.3 tiles (256x512:
    // 96, 512x1024: 384) — so a Cityscapes-sized batch of 4-16 samples still fills the chip (at 12 slices, 4 samples were
    // 48 blocks on 256 CUs).  More than CCDM_STATS_MAX_SLICES partials are folded to 16 by ccdm_stats_fold before a GroupNorm
    // reads them.  A function of the spatial size only (never of N): sharding the batch must not change the order in which
    // statistics partials are added.
    const int ovr = exp_env("CCDM_SLICES");
    if (ovr > 0 && ovr <= CCDM_STATS_MAX_SLICES && tiles >= ovr) return ovr;
    // Tile counts that only Cityscapes-sized images produce (64x128 and 32x64 with 8x16 tiles: 32 tiles; 128x256: 128 tiles) come with
    // batches of 4-16 samples: one slice per tile / per four tiles there (C5 shard 14.26 -> 13.24 ms, C4 7.13 -> 6.84 ms per step)
    // (not for the sub-pixel upsample form, whose 8x16 tiling gives LIDC's 64x64 input the same 32 tiles at batch 64)
    if (!up2 && tiles >= 128 && tiles < 256) return 32;
#ifndef CCDM_NO_S2_SLICES
    // stride-2 convs of a 32-tile output (LIDC's Downsample 128x128 -> 64x64 on 8x16 tiles): one slice per FOUR tiles.  The rule below
    // was made for Cityscapes batches of 4-16; at LIDC's batch its 2048 one-tile blocks were four rounds of 512 resident blocks, each
    // paying the block prologue for one tile (round 6: 59 -> 5x us).  Still a function of the layer's shape only.
    if (!up2 && stride == 2 && tiles == 32) return 8;
#endif
    if (!up2 && tiles >= 24 && tiles < 48) return tiles < 32 ? tiles : 32;
    if (tiles >= 128) return tiles / 16 * 3;
    if (tiles >= 48) return 12;
    // 64x64: 8 slices x 2 tiles (512 blocks, all resident) 28.8 us vs 16 x 1 (1024 blocks, a thin second round) 30.7.  Round 6, same-box
    // A/B of 12 / 16 slices here and 16 / 24 at 128x128 (profiles/r06_slices_ab.txt): the two-stream mode gains 0-1.7 % from more, shorter
    // blocks (its half batches leave slots empty), the single-stream mode loses 0.3-2.5 % — inside the box spread either way; not adopted.
    if (tiles >= 16) return 8;
    return tiles < 16 ? tiles : 16;
}

static int cin_pad_for(int Cin, int prec) { return prec == CCDM_PREC_F32 ? cdiv(Cin, 32) * 32 : cdiv(Cin, 16) * 16; }

// bytes of the packed B fragments (the F16X3 per-channel scale table follows them)
static size_t packed_frag_bytes(int Cout, int Cin, int ksize, int prec) {
    int ntiles, NI;
    conv_ntiles(Cout, &ntiles, &NI);
    const size_t cin_pad = cin_pad_for(Cin, prec), taps = (size_t)ksize * ksize;
    if (prec == CCDM_PREC_F32) return taps * (cin_pad / 2) * ntiles * 64 * sizeof(float);
    return taps * (cin_pad / 16) * ntiles * 2 * 64 * 16;
}

// does this conv run the K-split few-pixel kernel?  (a function of the layer's shape and operands only — the host sizes the
// statistics buffer by it through ccdm_conv_out_slices)
static bool conv_takes_ks(const ccdm_conv_args& a) {
    return a.up != 2 && !a.fine_slices && !exp_env("CCDM_NO_KS") && conv_ks_eligible(a);
}

int conv_out_slices(const ccdm_conv_args& a) {
    if (a.up == 2) return (conv_geo(a.Hin, a.Win, 1, true).TW == 16 ? 1 : 4) * conv_slices(a.Hin, a.Win, 1, true, a.fine_slices);
    if (conv_takes_ks(a)) return conv_ks_slices(a);
    return conv_slices(a.Hout, a.Wout, a.stride, false, a.fine_slices);
}

int launch_conv(const ccdm_conv_args& a, hipStream_t s) {
    const int C = a.C0 + a.C1;
    CCDM_REQUIRE(a.in0 && a.out && a.w, "conv: null in0/out/w");
    CCDM_REQUIRE(a.ksize == 1 || a.ksize == 3, "conv: ksize %d (need 1 or 3)", a.ksize);
    CCDM_REQUIRE(a.stride == 1 || a.stride == 2, "conv: stride %d", a.stride);
    CCDM_REQUIRE(a.stride == 1 || a.ksize == 3, "conv: stride 2 is built for 3x3 only");
    CCDM_REQUIRE(a.C0 % 4 == 0 && a.C1 % 4 == 0 && C > 0, "conv: C0=%d C1=%d must be multiples of 4", a.C0, a.C1);
    CCDM_REQUIRE((a.C1 == 0) == (a.in1 == nullptr), "conv: in1/C1 mismatch");
    CCDM_REQUIRE((a.prec & 255) == CCDM_PREC_F32 || (a.prec & 255) == CCDM_PREC_F16X3 || (a.prec & 255) == CCDM_PREC_F16, "conv: precision %d not built", a.prec);
    if (a.stats0) {
        CCDM_REQUIRE(C % 32 == 0, "conv: GroupNorm(32, %d) needs C %% 32 == 0", C);
        CCDM_REQUIRE(C <= CCDM_MAX_CHANNELS, "conv: %d input channels > CCDM_MAX_CHANNELS", C);
        CCDM_REQUIRE(a.gamma && a.beta, "conv: GroupNorm without gamma/beta");
        CCDM_REQUIRE(a.C1 == 0 || a.stats1, "conv: stats1 missing for concatenated input");
        CCDM_REQUIRE(a.slices0 >= 1 && (a.C1 == 0 || a.slices1 >= 1), "conv: bad stats slices");
    }
    CCDM_REQUIRE(!a.film || (a.stats0 && a.emb_table), "conv: FiLM needs GroupNorm and an emb table");
    CCDM_REQUIRE(a.up >= 0 && a.up <= 2, "conv: up = %d", a.up);
    const bool up2 = a.up == 2;
    if (up2)
        CCDM_REQUIRE((a.prec & 255) == CCDM_PREC_F16X3 && a.ksize == 3 && a.stride == 1 && a.Cout % 32 == 0 && !a.resid && !a.skip0 &&
                     a.Hout == 2 * a.Hin && a.Wout == 2 * a.Win,
                     "conv: the sub-pixel upsample form needs F16X3, 3x3, stride 1, Cout %% 32 == 0, no residual / fused skip (ccdm_upconv_supported)");
    if (a.skip0) {
        CCDM_REQUIRE(a.stride == 1 && !a.up && a.skip_w, "conv: fused skip needs stride 1, no upsample, packed skip_w");
        CCDM_REQUIRE(a.SC0 % 4 == 0 && a.SC1 % 4 == 0 && a.SC0 > 0, "conv: skip channels %d/%d must be multiples of 4", a.SC0, a.SC1);
        CCDM_REQUIRE((a.SC1 == 0) == (a.skip1 == nullptr), "conv: skip1/SC1 mismatch");
        CCDM_REQUIRE(a.Hin == a.Hout && a.Win == a.Wout, "conv: fused skip needs equal input and output size");
    }
    const int Hc = a.up ? 2 * a.Hin : a.Hin, Wc = a.up ? 2 * a.Win : a.Win;
    const int pad = a.ksize / 2;
    CCDM_REQUIRE(a.Hout == (Hc + 2 * pad - a.ksize) / a.stride + 1 && a.Wout == (Wc + 2 * pad - a.ksize) / a.stride + 1,
                 "conv: output %dx%d inconsistent with input %dx%d k%d s%d up%d", a.Hout, a.Wout, a.Hin, a.Win, a.ksize, a.stride, a.up);

    ConvK k;
    k.a = a;
    k.timeline = nullptr;
    k.core_unmasked = 0;
    const int prec = a.prec & 255;
    k.cin_pad = cin_pad_for(C, prec);
    k.cin_pad_skip = a.skip0 ? cin_pad_for(a.SC0 + a.SC1, prec) : 0;
    k.skip_wide = 0;
    int NI;
    conv_ntiles(up2 ? 4 * a.Cout : a.Cout, &k.ntiles, &NI);
    // the sub-pixel form tiles the low-resolution input space
    const int tH = up2 ? a.Hin : a.Hout, tW = up2 ? a.Win : a.Wout;
    const ConvGeo g = conv_geo(tH, tW, a.stride, up2, a.fine_slices != 0);
    // small spatial stages have few pixel tiles: spread the output-channel tiles over blocks instead;
    // wide tiles take at most 2 n-tiles per block (register budget of the staged B chunk)
    if (up2) NI = g.TW == 16 ? 4 : 1;
    else if (g.TW < 32) NI = 1;
    else {
        // two n-tiles per block share one staged halo (half the staging work) but need 236 VGPRs and 64 KB of LDS (two blocks
        // per CU): worth it only when there are blocks to spare.  Measured at the 32x32 stage (64 samples x 4 tiles): one n-tile
        // per block 25.6 us vs 28.0; at 64x64 outputs (16 tiles per sample) two n-tiles win, 67.8 vs 70.6.  (A partitioning
        // choice only: every output element is computed by the same instruction sequence either way.)
        const long blocks2 = (long)a.N * conv_slices(a.Hout, a.Wout, a.stride, false, a.fine_slices) * (k.ntiles / 2);
        NI = (k.ntiles % 2 == 0 && blocks2 >= 512) ? 2 : 1;
    }
    k.tiles_x = cdiv(tW, g.TW);
    k.tiles_y = cdiv(tH, g.TH);
    k.slices = conv_slices(tH, tW, a.stride, up2, a.fine_slices);
    // few-pixel images: K split over the waves, weight fragments straight from L2 (ccdm_conv_ks.hip) — one statistics slice per 8x8 tile
    const bool ks = conv_takes_ks(a);
    if (ks) k.slices = conv_ks_slices(a);
    k.wscale = reinterpret_cast<const float*>(static_cast<const char*>(a.w) +
                                              (up2 ? packed_frag_bytes(4 * a.Cout, C, 2, prec) : packed_frag_bytes(a.Cout, C, a.ksize, prec)));
    const int want_slices = (up2 && NI == 1 ? 4 : 1) * k.slices;      // one phase per block: every (slice, phase) pair leaves a partial
    if (a.out_stats) CCDM_REQUIRE(a.out_slices == want_slices, "conv: out_slices %d != %d", a.out_slices, want_slices);
    if (up2 && upconv_eligible(a)) {               // low-resolution Upsample convs: wave = phase, weight fragments straight from L2
        const int rcu = launch_upconv(a, k.slices, k.ntiles, k.wscale, s);
        if (rcu) return rcu;
        CCDM_CHECK_LAUNCH("upconv");
        return 0;
    }
    if (conv1x1_eligible(a, k.slices)) {           // AttentionBlock.proj_out + residual at the low-resolution stages: no LDS staging at all
        const int rc1 = launch_conv1x1(a, k.slices, k.ntiles, k.wscale, s);
        if (rc1) return rc1;
        CCDM_CHECK_LAUNCH("conv1x1");
        return 0;
    }
    if (ks) {
        const int rck = launch_conv_ks(a, k.ntiles, k.wscale, s);
        if (rck) return rck;
        CCDM_CHECK_LAUNCH("conv_ks");
        return 0;
    }
    const int HP = ((g.TH - 1) * a.stride + a.ksize) * ((g.TW - 1) * a.stride + a.ksize);
    const int ck = chunk_ck(a, g);
    {   // wide skip chunks (k_conv, SKW): the wide-tile F16X3 one-n-tile variant, skip sources in multiples of 32 channels
        k.skip_wide = (!exp_env("CCDM_NO_SKIP_WIDE") && a.skip0 && prec == CCDM_PREC_F16X3 && a.ksize == 3 && a.stride == 1 && !a.up && g.TW == 32 && NI == 1 && ck == 16 &&
                       a.SC0 % 32 == 0 && a.SC1 % 32 == 0 && !(a.prec >> 8)) ? 1 : 0;
    }
    {   // core halo items need no per-lane padding mask when every tile column and every channel quad exists (see ConvK)
        const int SC = a.SC0 + a.SC1;
        const bool chan_ok = a.C0 % ck == 0 && a.C1 % ck == 0 && (!a.skip0 || (a.SC0 % ck == 0 && a.SC1 % ck == 0 && SC > 0));
        k.core_unmasked = ((up2 ? a.Win : Wc) % (g.TW * a.stride) == 0 && chan_ok) ? 1 : 0;
    }
    CCDM_REQUIRE(a.C1 == 0 || a.C0 % ck == 0, "conv: first source has %d channels; a concatenated input must split at a multiple of the %d-channel chunk", a.C0, ck);
    CCDM_REQUIRE(a.SC1 == 0 || a.SC0 % ck == 0, "conv: first skip source has %d channels; a concatenated input must split at a multiple of the %d-channel chunk", a.SC0, ck);
    CCDM_REQUIRE((k.cin_pad + k.cin_pad_skip) / ck <= 64, "conv: %d input (+%d skip) channels make more than 64 chunks of %d (chunk descriptors live in the 64 lanes of a register)",
                 k.cin_pad, k.cin_pad_skip, ck);
    size_t lds = (size_t)HP * (prec == CCDM_PREC_F32 ? 33 * 4 : ck * 4 + 16);
    lds = (lds + 15) / 16 * 16;
    if (prec != CCDM_PREC_F32) lds += (size_t)(up2 ? 4 : a.ksize * a.ksize) * (ck / 16) * NI * 128 * 16;     // staged B chunk
    const int ksp = tap_split(a, g) ? 3 : 1;
    const size_t red = (size_t)g.waves * ksp * NI * 32 * 16;
    if (lds < red) lds = red;
    const size_t epi = (size_t)g.waves * ksp * g.MI * 32 * 36 * 4;        // epilogue transpose buffer (wave-private rows)
    if (lds < epi) lds = epi;
    if (a.stats0) {
        const size_t nthreads = (size_t)g.waves * ksp * 64;
        const size_t ex = (size_t)C > nthreads ? (size_t)C : nthreads;          // the prologue's statistics exchange (gn_affine_block) aliases the tiles
        if (lds < ex * 16) lds = ex * 16;
        lds += (size_t)C * 8;
    }
    if ((a.prec >> 8) & 512) lds = 60 * 1024;      // diagnostics (tools/bench_conv.py): at most 2 blocks per CU
    if ((a.prec >> 8) & 1024) lds = 100 * 1024;    //                                   1 block per CU
    CCDM_REQUIRE(lds <= 160 * 1024, "conv: LDS %zu too large", lds);
#ifdef CCDM_EXPERIMENTS
    if (conv_pc_eligible(k, g, NI)) {          // full-width 3x3 stages: producer/consumer form (tools/experiments/ccdm_conv_pc.hip, CCDM_PC=1)
        const int rc_pc = launch_conv_pc(k, s);
        if (rc_pc) return rc_pc;
        CCDM_CHECK_LAUNCH("conv(pc)");
        return 0;
    }
#endif
    dim3 grid(a.N * k.slices, k.ntiles / NI);
    const int rc = prec == CCDM_PREC_F32 ? launch_prec<CCDM_PREC_F32>(k, g, NI, ck, grid, lds, s)
                   : prec == CCDM_PREC_F16 ? launch_prec<CCDM_PREC_F16>(k, g, NI, ck, grid, lds, s)
                                           : launch_prec<CCDM_PREC_F16X3>(k, g, NI, ck, grid, lds, s);
    if (rc) return rc;
    CCDM_CHECK_LAUNCH("conv");
    return 0;
}

}  // namespace ccdm

extern "C" int ccdm_conv_slices(int Hout, int Wout, int stride, int ksize) {
    (void)ksize;
    return ccdm::conv_slices(Hout, Wout, stride);
}

extern "C" int ccdm_upconv_supported(int Cin, int Cout, int prec) {
    return prec == CCDM_PREC_F16X3 && Cin > 0 && Cin % 4 == 0 && Cout > 0 && Cout % 32 == 0;
}

extern "C" int ccdm_upconv_slices(int Hin, int Win) {
    return (ccdm::conv_geo(Hin, Win, 1, true).TW == 16 ? 1 : 4) * ccdm::conv_slices(Hin, Win, 1, true);
}

extern "C" int ccdm_conv_out_slices(const ccdm_conv_args* a) {
    if (!a) return ccdm::fail("ccdm_conv_out_slices: null args");
    return ccdm::conv_out_slices(*a);
}

extern "C" int ccdm_conv_slices_ex(int Hin, int Win, int ksize, int stride, int up, int fine) {
    if (up == 2) return (ccdm::conv_geo(Hin, Win, 1, true).TW == 16 ? 1 : 4) * ccdm::conv_slices(Hin, Win, 1, true, fine);
    const int Hc = up ? 2 * Hin : Hin, Wc = up ? 2 * Win : Win, pad = ksize / 2;
    return ccdm::conv_slices((Hc + 2 * pad - ksize) / stride + 1, (Wc + 2 * pad - ksize) / stride + 1, stride, false, fine);
}

extern "C" int ccdm_debug_read_timeline(unsigned long long* host, int n) {
    if (!host || n <= 0 || n > 1024) return ccdm::fail("debug_read_timeline: bad args");
#ifdef CCDM_ABLATION
    if (ccdm::conv_ks_timeline_read(host, n)) return 0;
#endif
#ifdef CCDM_EXPERIMENTS
    if (ccdm::conv_pc_timeline_read(host, n)) return 0;
#endif
    if (hipMemcpyFromSymbol(host, HIP_SYMBOL(ccdm::g_timeline), (size_t)n * 8, 0, hipMemcpyDeviceToHost) != hipSuccess)
        return ccdm::fail("debug_read_timeline: copy failed");
    return 0;
}

extern "C" int ccdm_conv2d(const ccdm_conv_args* a, void* stream) {
    if (!a) return ccdm::fail("ccdm_conv2d: null args");
    return ccdm::launch_conv(*a, (hipStream_t)stream);
}

// Packed layouts — the B operand exactly as the MFMA wants it, so a wave fetches a fragment with one
// coalesced load:
//  CCDM_PREC_F32:   [tap][cin_pad/2][ntiles][64] floats (cin_pad = Cin rounded up to 32); lane l holds
//                   W[cout = nt*32 + (l&31)][cin = 2*kp + (l>>5)][tap]      (v_mfma_f32_32x32x2_f32).
//  CCDM_PREC_F16X3: [tap][cin_pad/16][ntiles][hi|lo][64][8] halfs (cin_pad = Cin rounded up to 16); lane l,
//                   element j holds W[cout = nt*32 + (l&31)][cin = 16*ks + 8*(l>>5) + j][tap] * 2^e(cout), split
//                   into fp16 hi + lo (v_mfma_f32_32x32x16_f16), followed by [ntiles*32] floats 2^-e(cout).
//                   The per-output-channel power of two puts max|W| of the channel in [2^9, 2^10) so hi and lo
//                   both sit in fp16's normal range; it is exact and undone exactly in the epilogue (the table also
//                   carries 2^-4 for the kernel's activation pre-scale).
extern "C" size_t ccdm_pack_conv_weight_ex(const float* oihw, int Cout, int Cin, int ksize, int prec, const float* cout_absmax, void* out);

extern "C" size_t ccdm_pack_conv_weight(const float* oihw, int Cout, int Cin, int ksize, int prec, void* out) {
    return ccdm_pack_conv_weight_ex(oihw, Cout, Cin, ksize, prec, nullptr, out);
}

extern "C" size_t ccdm_pack_conv_weight_ex(const float* oihw, int Cout, int Cin, int ksize, int prec, const float* cout_absmax, void* out) {
    if (prec != CCDM_PREC_F32 && prec != CCDM_PREC_F16X3) { ccdm::fail("pack: precision %d not built", prec); return 0; }
    int ntiles, NI;
    ccdm::conv_ntiles(Cout, &ntiles, &NI);
    const int cin_pad = ccdm::cin_pad_for(Cin, prec), taps = ksize * ksize;
    const size_t frag = ccdm::packed_frag_bytes(Cout, Cin, ksize, prec);
    const size_t total = frag + (prec == CCDM_PREC_F16X3 ? (size_t)ntiles * 32 * sizeof(float) : 0);
    if (!out) return total;
    if (prec == CCDM_PREC_F32) {
        float* o = static_cast<float*>(out);
        for (int tap = 0; tap < taps; ++tap)
            for (int kp = 0; kp < cin_pad / 2; ++kp)
                for (int nt = 0; nt < ntiles; ++nt)
                    for (int l = 0; l < 64; ++l) {
                        const int co = nt * 32 + (l & 31), ci = 2 * kp + (l >> 5);
                        float v = 0.f;
                        if (co < Cout && ci < Cin) v = oihw[((size_t)co * Cin + ci) * taps + tap];
                        o[(((size_t)tap * (cin_pad / 2) + kp) * ntiles + nt) * 64 + l] = v;
                    }
        return total;
    }
    _Float16* o = static_cast<_Float16*>(out);
    float* sc = reinterpret_cast<float*>(static_cast<char*>(out) + frag);
    std::vector<float> mul(ntiles * 32, 1.0f);
    for (int co = 0; co < ntiles * 32; ++co) {
        float mx = 0.f;
        if (co < Cout) {
            if (cout_absmax) mx = cout_absmax[co];
            else for (size_t i = 0; i < (size_t)Cin * taps; ++i) mx = fmaxf(mx, fabsf(oihw[(size_t)co * Cin * taps + i]));
        }
        int e = 0;
        if (mx > 0.f && std::isfinite(mx)) { int ex; frexpf(mx, &ex); e = 10 - ex; }     // mx*2^e in [2^9, 2^10)
        if (e > 60) e = 60;
        if (e < -60) e = -60;
        mul[co] = ldexpf(1.0f, e);
        sc[co] = ldexpf(1.0f, -e) / ccdm::ACT_PRESCALE;     // also undoes the kernel's activation pre-scale
    }
    for (int tap = 0; tap < taps; ++tap)
        for (int ks = 0; ks < cin_pad / 16; ++ks)
            for (int nt = 0; nt < ntiles; ++nt)
                for (int l = 0; l < 64; ++l)
                    for (int j = 0; j < 8; ++j) {
                        const int co = nt * 32 + (l & 31), ci = 16 * ks + 8 * (l >> 5) + j;
                        float v = 0.f;
                        if (co < Cout && ci < Cin) v = oihw[((size_t)co * Cin + ci) * taps + tap] * mul[co];
                        const _Float16 hi = (_Float16)v;
                        const _Float16 lo = (_Float16)(v - (float)hi);
                        const size_t base = ((((size_t)tap * (cin_pad / 16) + ks) * ntiles + nt) * 2) * 64 * 8;
                        o[base + (size_t)l * 8 + j] = hi;
                        o[base + 64 * 8 + (size_t)l * 8 + j] = lo;
                    }
    return total;
}


// Sub-pixel form of Upsample(nearest x2) + conv 3x3 (include/ccdm_hip.h, `up = 2`): phase (dy, dx) sees the low-resolution 2x2
// window rows {y+dy-1, y+dy}; kernel row u of the 3x3 lands on window row a = (dy + u + 1) / 2 - dy  (dy=0: u=0 -> 0, u=1,2 -> 1;
// dy=1: u=0,1 -> 0, u=2 -> 1), columns alike.  Taps that share a window cell are added in fp64 and rounded once to fp32; the result
// is packed as a 2x2 conv with 4*Cout output channels, channel = (4 * (co / 32) + phase) * 32 + co % 32 (the four phases of a
// 32-channel tile are adjacent n-tiles).
extern "C" size_t ccdm_pack_upconv_weight(const float* oihw, int Cout, int Cin, int prec, void* out) {
    if (!ccdm_upconv_supported(Cin, Cout, prec)) { ccdm::fail("pack_upconv: Cin=%d Cout=%d prec=%d not supported", Cin, Cout, prec); return 0; }
    if (!out) return ccdm_pack_conv_weight_ex(nullptr, 4 * Cout, Cin, 2, prec, nullptr, nullptr);
    std::vector<float> w2((size_t)4 * Cout * Cin * 4);
    for (int ph = 0; ph < 4; ++ph) {
        const int dy = ph >> 1, dx = ph & 1;
        for (int co = 0; co < Cout; ++co)
            for (int ci = 0; ci < Cin; ++ci) {
                double acc[2][2] = {{0.0, 0.0}, {0.0, 0.0}};
                for (int u = 0; u < 3; ++u)
                    for (int v = 0; v < 3; ++v)
                        acc[(dy + u + 1) / 2 - dy][(dx + v + 1) / 2 - dx] += (double)oihw[(((size_t)co * Cin + ci) * 3 + u) * 3 + v];
                float* d = &w2[(((size_t)(4 * (co / 32) + ph) * 32 + co % 32) * Cin + ci) * 4];
                d[0] = (float)acc[0][0]; d[1] = (float)acc[0][1]; d[2] = (float)acc[1][0]; d[3] = (float)acc[1][1];
            }
    }
    return ccdm_pack_conv_weight_ex(w2.data(), 4 * Cout, Cin, 2, prec, nullptr, out);
}
