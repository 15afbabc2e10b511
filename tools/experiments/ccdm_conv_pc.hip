// Producer/consumer form of the fused [GroupNorm -> (FiLM) -> SiLU ->] conv3x3 [+bias +emb +residual] (+ statistics) kernel for the
// full-width stages (output rows >= 32 pixels wide, stride 1, one 32-wide output-channel tile per block, CCDM_PREC_F16X3):
// the geometry that carries 70 % of a LIDC denoise step.  Same arithmetic, same tiles, same statistics slots as k_conv
// (ccdm_conv.hip) — every output element is produced by the same instruction sequence — but the work of a block is split by ROLE:
//
//   loader waves (0-7)                                    matrix waves (8-11)
//   ----------------------------------------------------  ------------------------------------------------------
//   halo + weight fragments: global -> registers          A/B fragments: LDS -> registers, 54 MFMAs per chunk
//   GroupNorm affine, SiLU, fp16 hi/lo split -> LDS        accumulators -> (x 2^-e + bias + emb) -> LDS transpose buffer
//   epilogue rows: transpose buffer + residual -> HBM,
//   per-channel statistics
//
// One block per CU, two loader waves and one matrix wave per SIMD.  The halo tile and the weight chunk are double-buffered in LDS: in
// phase i the matrix waves multiply chunk i while the loaders stage chunk i+1 and finish the previous tile's epilogue; one
// barrier per phase.  k_conv alternates these roles inside every wave, so a SIMD's matrix pipe idles while its three co-resident
// waves wait on memory, barriers or each other's vector work (measured: ~50 % of SIMD time idle); here the vector work of one
// wave runs beside the MFMAs of the other, and the loaders keep two chunks of HBM requests in flight (two register sets).
#include "ccdm_common.h"
#include "ccdm_conv_common.h"

#include <cstdlib>
#include <type_traits>

namespace ccdm {

static constexpr int PC_TH = 8, PC_TW = 32, PC_CK = 16, PC_KS = 3, PC_MI = 2;
static constexpr int PC_PIXB = PC_CK * 4 + 16;                          // 80 B per halo pixel: 16 hi | 16 lo halfs | 16 pad
static constexpr int PC_HHt = PC_TH + 2, PC_HWt = PC_TW + 2, PC_HP = PC_HHt * PC_HWt;
static constexpr int PC_A_BYTES = (PC_HP * PC_PIXB + 15) / 16 * 16;
static constexpr int PC_B_BYTES = PC_KS * PC_KS * 128 * 16;              // [tap][hi|lo][64 lanes] x 16 B, one n-tile
static constexpr int PC_EPS = 36;                                        // floats per pixel row of the transpose buffer
static constexpr int PC_EPI_BYTES = 4 * PC_MI * 32 * PC_EPS * 4;

size_t conv_pc_lds_bytes(int C) { return (size_t)C * 8 + 2 * PC_A_BYTES + 2 * PC_B_BYTES + PC_EPI_BYTES; }

static constexpr int PC_NL = 8;                                          // loader waves (matrix waves: 4)

__global__ __launch_bounds__((PC_NL + 4) * 64, 3) void k_conv_pc(const ConvK k) {
    constexpr int TH = PC_TH, TW = PC_TW, CK = PC_CK, KS = PC_KS, MI = PC_MI, PIXB = PC_PIXB;
    constexpr int NL = PC_NL, NTL = NL * 64;                      // loader waves / threads = staging threads
    constexpr int RJ = MI * 4 * 4 / NL;                           // epilogue row-iterations per loader wave (8 per matrix wave's sub-tiles)
    constexpr int PAD = 1, HHt = PC_HHt, HWt = PC_HWt;
    constexpr int QPP = CK / 4;                                   // float4 items per halo pixel
    constexpr int PXW = NTL / QPP, RPP = PXW / TW, NCORE = (HHt + RPP - 1) / RPP;
    constexpr int ECOLS = 2, EDGE_ITEMS = HHt * 2 * PAD * QPP, NEDGE = (EDGE_ITEMS + NTL - 1) / NTL;
    constexpr int NITEM = NCORE + NEDGE;
    constexpr int G = 128, NB4 = KS * KS * G, NITEM_B = (NB4 + NTL - 1) / NTL, MB = NTL / G;
    constexpr int EPS = PC_EPS;
    static_assert((TW * QPP) % 64 == 0 && PXW % TW == 0 && NTL % G == 0, "geometry");

    const ccdm_conv_args& a = k.a;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int C = a.C0 + a.C1;
    // diagnostics: phase stamps of one mid-grid block — loader wave 0 -> slots [0, 512), matrix wave 0 -> [512, 1024)
    int tlp = 0;
    const bool tl_on = k.timeline != nullptr && blockIdx.x == gridDim.x / 2 && blockIdx.y == 0 && (threadIdx.x == 0 || threadIdx.x == PC_NL * 64);
#define PC_STAMP(id) do { if (tl_on && tlp < 510) k.timeline[(threadIdx.x ? 512 : 0) + 1 + tlp++] = ((unsigned long long)(id) << 56) | (__builtin_amdgcn_s_memtime() & 0x00ffffffffffffffull); } while (0)
    float2* ab = reinterpret_cast<float2*>(smem);                              // [C] (only if stats0)
    char* lds0 = smem + (a.stats0 ? (size_t)C * 8 : 0);
    // [A0 | A1 | B0 | B1 | EPI]
    char* epi_b = lds0 + 2 * PC_A_BYTES + 2 * PC_B_BYTES;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave_all = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool is_loader = wave_all < NL;
    const int wave = wave_all & 3;                                  // pixel sub-tile pair: matrix wave NL + w, loader waves w, w + 4, ...
    const int jpart = (wave_all % NL) >> 2;                         // which share of that pair's epilogue rows this loader wave takes
    const int tl = tid % NTL;                                       // staging thread index of a loader thread
    int bid = blockIdx.x;
    if ((gridDim.x & 7) == 0) bid = (blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3);
    const int n = bid / k.slices, slice = bid % k.slices;
    const int nt0 = blockIdx.y;
    const int Hc = a.up ? a.Hin * 2 : a.Hin, Wc = a.up ? a.Win * 2 : a.Win;    // conv-input space
    const int step = a.step_ptr ? *a.step_ptr : 0;
    const int emb_row = (a.emb_row_of_sample ? a.emb_row_of_sample[n] : 0) + step;
    const bool has_gn = a.stats0 != nullptr;

    const int aWout = a.Wout, aCout = a.Cout, aHout = a.Hout, aWin = a.Win;
    const size_t in_px = (size_t)a.Hin * a.Win;
    const size_t out_px = (size_t)a.Hout * a.Wout;
    float* outn = a.out + (size_t)n * out_px * aCout;
    const float* residn = a.resid + (size_t)n * out_px * aCout;   // guarded by a.resid at the uses

    // matrix waves: per-lane LDS base of the wave's two 32-pixel sub-tiles (A operand: row = lane&31, k-group = lane>>5)
    int base[MI];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
        const int p = (wave * MI + mi) * 32 + (lane & 31);
        base[mi] = ((p / TW) * HWt + (p % TW)) * PIXB + (lane >> 5) * 16;
    }
    // loader waves: statistics partials, lane = (pixel, channel quad) of the row pass
    float s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f};
    // matrix waves: per-lane epilogue constants of output channel (n-tile, lane & 31)
    float epi_add = 0.f, epi_wsc = 1.0f;
    {
        const int co = nt0 * 32 + (lane & 31);
        if (co < a.Cout) {
            epi_add = a.bias ? a.bias[co] : 0.f;
            if (a.emb_off >= 0) epi_add += a.emb_table[(size_t)emb_row * a.emb_stride + a.emb_off + co];
            epi_wsc = k.wscale[co];
        }
    }

    const int ntile_sp = k.tiles_x * k.tiles_y;
    const int nchunk_main = k.cin_pad / CK;
    const int nchunk = nchunk_main + k.cin_pad_skip / CK;      // main segment, then the fused 1x1 skip segment
    const int my_tiles = (ntile_sp - slice + k.slices - 1) / k.slices;
    const int n_iter = my_tiles * nchunk;

    f32x4 reg[2][NITEM];
    f32x4 regB[NITEM_B];
    unsigned rowmask[2] = {0, 0}, evalid[2] = {0, 0};
    bool xok[2] = {false, false};
    const int rip = __builtin_amdgcn_readfirstlane((tl / QPP) / TW);
    const unsigned tgB = (unsigned)__builtin_amdgcn_readfirstlane(tl / G);

    // per-chunk descriptors, one chunk per lane (see k_conv)
    unsigned T_lo = 0, T_hi = 0, T_cc = 0, T_wlo = 0, T_whi = 0;
    {
        const int ch = lane < nchunk ? lane : 0;
        const bool sk = ch >= nchunk_main;
        const int c0 = (sk ? ch - nchunk_main : ch) * CK;
        const int sC0 = sk ? a.SC0 : a.C0, sC1 = sk ? a.SC1 : a.C1;
        const bool second = sC1 > 0 && c0 >= sC0;
        const int Cs = second ? sC1 : sC0, cb = second ? c0 - sC0 : c0;
        const float* srcsel = sk ? (second ? a.skip1 : a.skip0) : (second ? a.in1 : a.in0);
        const unsigned long long sb = reinterpret_cast<unsigned long long>(srcsel + (size_t)n * (sk ? out_px : in_px) * Cs);
        T_lo = (unsigned)sb; T_hi = (unsigned)(sb >> 32);
        T_cc = (unsigned)Cs | ((unsigned)cb << 16);
        const unsigned long long wb = reinterpret_cast<unsigned long long>(sk ? a.skip_w : a.w) + (((size_t)(c0 >> 4) * k.ntiles + nt0) * 128 << 4);
        T_wlo = (unsigned)wb; T_whi = (unsigned)(wb >> 32);
    }

    // ---- issue: global -> registers for iteration (ch, ty, tx) ----
    auto issue = [&](auto D_, const int ch, const int ty, const int tx) __attribute__((always_inline)) {
        constexpr int d = decltype(D_)::value;
        const unsigned cc = __builtin_amdgcn_readlane(T_cc, ch);
        const int Cs = cc & 0xffffu, cb = cc >> 16;
        const char* srcb = reinterpret_cast<const char*>(((unsigned long long)(unsigned)__builtin_amdgcn_readlane(T_hi, ch) << 32) |
                                                         (unsigned)__builtin_amdgcn_readlane(T_lo, ch));
        const int ups = __builtin_amdgcn_readfirstlane(a.up);
        const int oy0 = ty * TH, ox0 = tx * TW;
        const bool skseg = ch >= nchunk_main;
        const int ylo = skseg ? min(oy0, Hc - 1) : 0, yhi = skseg ? min(oy0 + TH - 1, Hc - 1) : Hc - 1;
        const int xlo = skseg ? min(ox0, Wc - 1) : 0, xhi = skseg ? min(ox0 + TW - 1, Wc - 1) : Wc - 1;
        unsigned t_ = tl;
        asm volatile("" : "+v"(t_));
        const int tq = t_ % QPP, px = (t_ / QPP) % TW;
        const unsigned c = (unsigned)cb + 4u * (unsigned)tq;
        const unsigned cq = min(c, (unsigned)Cs - 4u);
        const bool cok = c < (unsigned)Cs;
        const unsigned rowb = (unsigned)aWin * (unsigned)Cs * 4u;
        {
            const int ix = ox0 + px;
            xok[d] = cok & (ix < Wc);
            const int ixc = min(ix, Wc - 1);
            const unsigned colb = ((unsigned)(ixc >> ups) * (unsigned)Cs + cq) << 2;
            rowmask[d] = 0;
#pragma unroll
            for (int i = 0; i < NCORE; ++i) {
                const int row = rip + i * RPP;
                const int iy = oy0 - PAD + row;
                const bool rok = ((unsigned)iy < (unsigned)Hc) & ((i + 1) * RPP <= HHt || row < HHt);
                const int iyc = min(max(iy, ylo), yhi);
                const unsigned sy = (unsigned)(iyc >> ups);
                reg[d][i] = load16_uniform_base(srcb + (size_t)(sy * rowb), colb);
                rowmask[d] |= (rok ? 1u : 0u) << i;
            }
        }
        evalid[d] = 0;
#pragma unroll
        for (int j = 0; j < NEDGE; ++j) {
            const unsigned e = t_ + j * NTL;
            const unsigned side = (e / QPP) % ECOLS, row = e / (ECOLS * QPP);
            const int hx = side < (unsigned)PAD ? (int)side : TW + (int)side;
            const int iy = oy0 - PAD + (int)row, ix = ox0 - PAD + hx;
            const bool ok = cok & (e < (unsigned)EDGE_ITEMS) & ((unsigned)iy < (unsigned)Hc) & ((unsigned)ix < (unsigned)Wc);
            const int iyc = min(max(iy, ylo), yhi), ixc = min(max(ix, xlo), xhi);
            const unsigned sy = (unsigned)(iyc >> ups), sx = (unsigned)(ixc >> ups);
            reg[d][NCORE + j] = load16_global(srcb + (size_t)(sy * rowb + ((sx * (unsigned)Cs + cq) << 2)));
            evalid[d] |= (ok ? 1u : 0u) << j;
        }
    };
    auto issueB = [&](const int ch) __attribute__((always_inline)) {
        const bool sk = ch >= nchunk_main;
        unsigned t_ = tl;
        asm volatile("" : "+v"(t_));
        const char* wq = reinterpret_cast<const char*>(((unsigned long long)(unsigned)__builtin_amdgcn_readlane(T_whi, ch) << 32) |
                                                       (unsigned)__builtin_amdgcn_readlane(T_wlo, ch));
        const unsigned wtap = (unsigned)((sk ? k.cin_pad_skip : k.cin_pad) >> 4) * k.ntiles * 128;
        const unsigned nslab = sk ? 1 : KS * KS;
        const unsigned remb = (t_ % G) << 4;
#pragma unroll
        for (int i = 0; i < NITEM_B; ++i) {
            unsigned ts = i * MB + tgB;
            ts = ts < nslab ? ts : 0u;
            regB[i] = load16_uniform_base(wq + ((size_t)(ts * wtap) << 4), remb);
        }
    };
    // ---- commit: registers -> affine -> SiLU -> fp16 hi|lo split -> LDS buffer `buf` (zero where padded) ----
    auto commit_body = [&](auto D_, auto GN_, auto ACT_, int c0, char* halo_b, f32x4* ldsB) __attribute__((always_inline)) {
        constexpr int d = decltype(D_)::value;
        constexpr bool GN = decltype(GN_)::value, ACT = decltype(ACT_)::value;
        unsigned t_ = tl;
        asm volatile("" : "+v"(t_));
        const int tq = t_ % QPP, px = (t_ / QPP) % TW;
        float2 t0 = make_float2(1.f, 0.f), t1 = t0, t2 = t0, t3 = t0;
        if (GN) { const int c = c0 + 4 * tq; t0 = ab[c]; t1 = ab[c + 1]; t2 = ab[c + 2]; t3 = ab[c + 3]; }
        constexpr float PS = ACT_PRESCALE;
        if (GN && !ACT) { t0.x *= PS; t0.y *= PS; t1.x *= PS; t1.y *= PS; t2.x *= PS; t2.y *= PS; t3.x *= PS; t3.y *= PS; }
        auto act = [&](const float x) __attribute__((always_inline)) {
            if (!ACT) return GN ? x : x * PS;
            return x * __builtin_amdgcn_rcpf(1.0f / PS + __builtin_amdgcn_exp2f(fmaf(x, -1.4426950408889634f, -4.0f)));
        };
        auto put = [&](const f32x4 r, const bool ok, const int hp) __attribute__((always_inline)) {
            float4 v = make_float4(r[0], r[1], r[2], r[3]);
            if (GN) { v.x = fmaf(v.x, t0.x, t0.y); v.y = fmaf(v.y, t1.x, t1.y); v.z = fmaf(v.z, t2.x, t2.y); v.w = fmaf(v.w, t3.x, t3.y); }
            v.x = act(v.x); v.y = act(v.y); v.z = act(v.z); v.w = act(v.w);
            const float lim = ok ? __builtin_inff() : 0.f;          // zeroes padding; beyond fp16's range the value is NOT clipped (k_conv)
            v.x = __builtin_amdgcn_fmed3f(v.x, -lim, lim); v.y = __builtin_amdgcn_fmed3f(v.y, -lim, lim);
            v.z = __builtin_amdgcn_fmed3f(v.z, -lim, lim); v.w = __builtin_amdgcn_fmed3f(v.w, -lim, lim);
            f16x4 hi, lo;
            hi[0] = (_Float16)v.x; hi[1] = (_Float16)v.y; hi[2] = (_Float16)v.z; hi[3] = (_Float16)v.w;
            lo[0] = (_Float16)(v.x - (float)hi[0]); lo[1] = (_Float16)(v.y - (float)hi[1]);
            lo[2] = (_Float16)(v.z - (float)hi[2]); lo[3] = (_Float16)(v.w - (float)hi[3]);
            char* dp = halo_b + hp * PIXB + 8 * tq;
            *reinterpret_cast<f16x4*>(dp) = hi;
            *reinterpret_cast<f16x4*>(dp + 2 * CK) = lo;
        };
        const int hp0 = rip * HWt + PAD + px;
#pragma unroll
        for (int i = 0; i < NCORE; ++i)
            if ((i + 1) * RPP <= HHt || rip + i * RPP < HHt)
                put(reg[d][i], xok[d] & (((rowmask[d] >> i) & 1u) != 0u), hp0 + i * RPP * HWt);
#pragma unroll
        for (int j = 0; j < NEDGE; ++j) {
            const unsigned e = t_ + j * NTL;
            if (e < (unsigned)EDGE_ITEMS) {
                const unsigned side = (e / QPP) % ECOLS, row = e / (ECOLS * QPP);
                const int hx = side < (unsigned)PAD ? (int)side : TW + (int)side;
                put(reg[d][NCORE + j], ((evalid[d] >> j) & 1u) != 0u, (int)row * HWt + hx);
            }
        }
#pragma unroll
        for (int i = 0; i < NITEM_B; ++i) {
            const int j = (int)t_ + i * NTL;
            if ((i + 1) * NTL <= NB4 || j < NB4) ldsB[j] = regB[i];
        }
    };
    auto commit = [&](auto D_, const int ch, const int buf) __attribute__((always_inline)) {
        const bool sk = ch >= nchunk_main;
        const int c0 = (sk ? ch - nchunk_main : ch) * CK;
        const bool gn = has_gn && !sk, act = a.act == CCDM_ACT_SILU && !sk;
        char* halo_b = lds0 + buf * PC_A_BYTES;
        f32x4* ldsB = reinterpret_cast<f32x4*>(lds0 + 2 * PC_A_BYTES + buf * PC_B_BYTES);
        if (gn && act) commit_body(D_, std::true_type{}, std::true_type{}, c0, halo_b, ldsB);
        else if (gn) commit_body(D_, std::true_type{}, std::false_type{}, c0, halo_b, ldsB);
        else if (act) commit_body(D_, std::false_type{}, std::true_type{}, c0, halo_b, ldsB);
        else commit_body(D_, std::false_type{}, std::false_type{}, c0, halo_b, ldsB);
    };

    // ---- epilogue rows (loader waves): transpose buffer (+ residual) -> float4 stores, statistics ----
    const int cq = lane & 7, prow = lane >> 3;
    const int co4 = nt0 * 32 + 4 * cq;
    const bool cv4 = co4 < a.Cout;
    f32x4 rs[RJ];                                  // residual rows, requested one phase ahead of the row pass that adds them
    auto row_geo = [&](const int ty, const int tx, const int j, unsigned& rbase, int& oy, int& ox) __attribute__((always_inline)) {
        oy = ty * TH + wave * (MI * 32 / TW) + (j * 8) / TW;
        ox = tx * TW + (j * 8) % TW;
        rbase = (unsigned)((oy * aWout + ox) * aCout) << 2;
    };
    auto tile_full = [&](const int ty, const int tx) __attribute__((always_inline)) {
        return ty * TH + TH <= aHout && tx * TW + TW <= aWout && nt0 * 32 + 32 <= aCout;   // uniform
    };
    auto resid_issue = [&](const int ty, const int tx) __attribute__((always_inline)) {
        const bool full = tile_full(ty, tx);
        unsigned lane_off = ((unsigned)prow * (unsigned)aCout + (unsigned)co4) << 2;
        asm volatile("" : "+v"(lane_off));
#pragma unroll
        for (int jj = 0; jj < RJ; ++jj) {
            const int j = jpart * RJ + jj;
            unsigned rb; int oy, ox;
            row_geo(ty, tx, j, rb, oy, ox);
            if (full) rs[jj] = load16_uniform_base(reinterpret_cast<const char*>(residn) + rb, lane_off);
            else {
                const int oyc = min(oy, aHout - 1), oxc = min(ox + prow, aWout - 1);
                const int cc = min(co4, aCout - 4);
                rs[jj] = *reinterpret_cast<const f32x4*>(reinterpret_cast<const char*>(residn) +
                                                         (((unsigned)(oyc * aWout + oxc) * (unsigned)aCout + (unsigned)cc) << 2));
            }
        }
    };
    auto rows_pass = [&](const int ty, const int tx) __attribute__((always_inline)) {
        const bool full = tile_full(ty, tx);
        unsigned lane_off = ((unsigned)prow * (unsigned)aCout + (unsigned)co4) << 2;
        asm volatile("" : "+v"(lane_off));
        const float* epi0 = reinterpret_cast<const float*>(epi_b) + wave * (MI * 32 * EPS);
        float t1[4] = {0.f, 0.f, 0.f, 0.f}, t2[4] = {0.f, 0.f, 0.f, 0.f};
        auto body = [&](auto FULL_, auto RESID_) __attribute__((always_inline)) {
            constexpr bool FULL = decltype(FULL_)::value, RESID = decltype(RESID_)::value;
#pragma unroll
            for (int jj = 0; jj < RJ; ++jj) {
                const int j = jpart * RJ + jj;
                unsigned rb; int oy, ox;
                row_geo(ty, tx, j, rb, oy, ox);
                const int pl = j * 8 + prow;
                f32x4 v = *reinterpret_cast<const f32x4*>(epi0 + pl * EPS + 4 * cq);
                if (RESID) v += rs[jj];
                if (FULL) {
                    store16_uniform_base<false>(reinterpret_cast<char*>(outn) + rb, lane_off, v);
#pragma unroll
                    for (int e = 0; e < 4; ++e) { t1[e] += v[e]; t2[e] = fmaf(v[e], v[e], t2[e]); }
                } else {
                    if (cv4 && oy < aHout && ox + prow < aWout) {
                        *reinterpret_cast<f32x4*>(reinterpret_cast<char*>(outn) + (((unsigned)(oy * aWout + ox + prow) * (unsigned)aCout + (unsigned)co4) << 2)) = v;
#pragma unroll
                        for (int e = 0; e < 4; ++e) { t1[e] += v[e]; t2[e] = fmaf(v[e], v[e], t2[e]); }
                    }
                }
            }
        };
        if (a.resid) { if (full) body(std::true_type{}, std::true_type{}); else body(std::false_type{}, std::true_type{}); }
        else { if (full) body(std::true_type{}, std::false_type{}); else body(std::false_type{}, std::false_type{}); }
#pragma unroll
        for (int e = 0; e < 4; ++e) { s1[e] += t1[e]; s2[e] += t2[e]; }
    };

    // ---- MFMA phase (matrix waves): A and B fragments from LDS buffer `buf` ----
    f32x16 acc[MI];
    auto mfma_phase = [&](const int chunk, const int buf) __attribute__((always_inline)) {
        const bool skc = chunk >= nchunk_main;
        const char* halo_b = lds0 + buf * PC_A_BYTES;
        const f16x8* bq = reinterpret_cast<const f16x8*>(lds0 + 2 * PC_A_BYTES + buf * PC_B_BYTES) + lane;
        f16x8 ah[2][MI], al[2][MI], bh[2], bl[2];
        auto frag_load = [&](const int fb, const int toff, const int bt) __attribute__((always_inline)) {
#pragma unroll
            for (int mi = 0; mi < MI; ++mi) {
                const char* p = halo_b + base[mi] + toff;
                ah[fb][mi] = *reinterpret_cast<const f16x8*>(p);
                al[fb][mi] = *reinterpret_cast<const f16x8*>(p + 2 * CK);
            }
            bh[fb] = bq[bt * 128];
            bl[fb] = bq[bt * 128 + 64];
        };
        auto frag_mfma = [&](const int fb) __attribute__((always_inline)) {
#pragma unroll
            for (int mi = 0; mi < MI; ++mi) acc[mi] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[fb][mi], bh[fb], acc[mi], 0, 0, 0);
#pragma unroll
            for (int mi = 0; mi < MI; ++mi) acc[mi] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[fb][mi], bl[fb], acc[mi], 0, 0, 0);
#pragma unroll
            for (int mi = 0; mi < MI; ++mi) acc[mi] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[fb][mi], bh[fb], acc[mi], 0, 0, 0);
        };
        if (skc) {
            frag_load(0, (PAD * HWt + PAD) * PIXB, 0);               // centre tap only; its weights are slab 0
            frag_mfma(0);
        } else {
            frag_load(0, 0, 0);
#pragma unroll
            for (int tap = 0; tap < KS * KS; ++tap) {
                if (tap + 1 < KS * KS) frag_load((tap + 1) & 1, (((tap + 1) / KS) * HWt + ((tap + 1) % KS)) * PIXB, tap + 1);
                __builtin_amdgcn_sched_barrier(0);
                frag_mfma(tap & 1);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    };
    auto acc_to_epi = [&]() __attribute__((always_inline)) {
        float* epi = reinterpret_cast<float*>(epi_b) + wave * (MI * 32 * EPS);
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int pl = mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                epi[pl * EPS + (lane & 31)] = fmaf(acc[mi][r], epi_wsc, epi_add);
            }
    };

    // (tile, chunk) walks: `cur` = the iteration the matrix waves multiply, `nx` = the one the loaders commit (cur + 1),
    // `pf` = the one the loaders request (cur + 3).  The two roles run separate programs (disjoint register live ranges: the
    // kernel's register count is the larger of the two, not their sum) that meet at one barrier per phase.
    const int adv_y = k.slices / k.tiles_x, adv_x = k.slices % k.tiles_x;
    auto advance = [&](int& ch, int& ty, int& tx) __attribute__((always_inline)) {
        if (++ch == nchunk) {
            ch = 0; tx += adv_x; ty += adv_y;
            if (tx >= k.tiles_x) { tx -= k.tiles_x; ++ty; }
        }
    };
    using D0 = std::integral_constant<int, 0>;
    using D1 = std::integral_constant<int, 1>;
    if (has_gn) compute_gn_affine(a, n, emb_row, ab);

    if (is_loader) {
        // ================================= loader program =================================
        int chunk = 0, cur_ty = slice / k.tiles_x, cur_tx = slice % k.tiles_x;
        int nx_ch = 0, nx_ty = cur_ty, nx_tx = cur_tx;
        int pf_ch = 0, pf_ty = cur_ty, pf_tx = cur_tx;
        int done_ty = 0, done_tx = 0;
        bool rows_pending = false;
        PC_STAMP(1);
        issue(D0{}, pf_ch, pf_ty, pf_tx);         // iteration 0
        advance(pf_ch, pf_ty, pf_tx);
        issue(D1{}, pf_ch, pf_ty, pf_tx);         // iteration 1
        advance(pf_ch, pf_ty, pf_tx);
        issueB(0);
        PC_STAMP(2);
        __syncthreads();                          // (A) GroupNorm table visible
        PC_STAMP(3);
        advance(nx_ch, nx_ty, nx_tx);             // nx = iteration 1
        // Request order matters: vector-memory results return in order, so the weight fragments of the NEXT commit are requested
        // BEFORE the halo prefetch that follows it — the commit then waits for "all but the newest loads" and the two chunks
        // of halo requests stay in flight across it.
        commit(D0{}, 0, 0);                       // iteration 0 -> buffer 0
        issueB(nx_ch);                            // fragments of iteration 1
        issue(D0{}, pf_ch, pf_ty, pf_tx);         // halo of iteration 2
        advance(pf_ch, pf_ty, pf_tx);
        PC_STAMP(4);
        __syncthreads();                          // (B) buffer 0 staged
        PC_STAMP(5);
        auto lphase = [&](auto DN_, const int it) __attribute__((always_inline)) {    // DN_ = register set holding iteration it + 1
            const int buf = it & 1;
            if (it + 1 < n_iter) {
                commit(DN_, nx_ch, buf ^ 1);      // iteration it + 1 (its fragments were requested in the previous phase)
                PC_STAMP(10);
                issueB(nx_ch + 1 == nchunk ? 0 : nx_ch + 1);     // fragments of iteration it + 2
                issue(DN_, pf_ch, pf_ty, pf_tx);  // halo of iteration it + 3 (clamped addresses beyond the last tile: harmless, never committed)
                advance(pf_ch, pf_ty, pf_tx);
                PC_STAMP(11);
            }
            if (rows_pending) { rows_pass(done_ty, done_tx); PC_STAMP(12); }       // the tile the matrix waves finished in the previous phase
            if (a.resid && chunk == nchunk - 1) resid_issue(cur_ty, cur_tx);     // for the row pass of the next phase
            rows_pending = chunk == nchunk - 1;
            done_ty = cur_ty; done_tx = cur_tx;
            advance(chunk, cur_ty, cur_tx);
            advance(nx_ch, nx_ty, nx_tx);
            __syncthreads();
            PC_STAMP(14);
        };
        for (int it = 0; it < n_iter; it += 2) {
            lphase(D1{}, it);
            if (it + 1 < n_iter) lphase(D0{}, it + 1);
        }
        if (rows_pending) rows_pass(done_ty, done_tx);
        PC_STAMP(15);
    } else {
        // ================================= matrix program =================================
        int chunk = 0;
        PC_STAMP(1);
        __syncthreads();                          // (A)
        __syncthreads();                          // (B)
        PC_STAMP(5);
        for (int it = 0; it < n_iter; ++it) {
            if (chunk == 0) {
#pragma unroll
                for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[mi][r] = 0.0f;
            }
            mfma_phase(chunk, it & 1);
            PC_STAMP(20);
            if (chunk == nchunk - 1) { acc_to_epi(); PC_STAMP(21); }
            if (++chunk == nchunk) chunk = 0;
            __syncthreads();
            PC_STAMP(22);
        }
    }

    if (tl_on) k.timeline[threadIdx.x ? 512 : 0] = tlp;
    if (a.out_stats) {
        // fold the lanes that hold the same channel, then the loader waves; fixed order everywhere
        __syncthreads();
        double* red = reinterpret_cast<double*>(lds0);     // [NL][32][2]
        if (is_loader) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                double v1 = (double)s1[e], v2 = (double)s2[e];
#pragma unroll
                for (int off = 8; off < 64; off <<= 1) { v1 += __shfl_xor(v1, off); v2 += __shfl_xor(v2, off); }
                if (lane < 8) {
                    red[(wave_all * 32 + 4 * lane + e) * 2 + 0] = v1;
                    red[(wave_all * 32 + 4 * lane + e) * 2 + 1] = v2;
                }
            }
        }
        __syncthreads();
        if (tid < 32) {
            double t1 = 0.0, t2 = 0.0;
            for (int w = 0; w < NL; ++w) { t1 += red[(w * 32 + tid) * 2 + 0]; t2 += red[(w * 32 + tid) * 2 + 1]; }
            const int co = nt0 * 32 + tid;
            if (co < a.Cout) {
                double* o = a.out_stats + (((size_t)n * k.slices + slice) * a.Cout + co) * 2;
                o[0] = t1; o[1] = t2;
            }
        }
    }
}

// Geometries the producer/consumer kernel takes over from k_conv (everything else about the launch is identical).
bool conv_pc_eligible(const ConvK& k, const ConvGeo& g, int NI) {
    // Off by default — a measured negative result kept as an A/B experiment (CCDM_PC=1): 112.8 us against k_conv's 86.9 us on the
    // 32->32 @128x128 layer (same box).  Its phase timeline (CCDM_PC_TIMELINE=1, tools/bench_conv.py) shows why: a SIMD's MFMAs and
    // the plain VALU work of ANOTHER wave on that SIMD do not overlap — the loaders' commit takes 3400 cycles beside 54 MFMAs
    // (1728 cycles) and those MFMAs stretch to 3000 — so splitting roles between waves buys nothing, while one block per CU
    // exposes every block's prologue and tail that k_conv's three co-resident blocks hide.
    static const int on = getenv("CCDM_PC") ? atoi(getenv("CCDM_PC")) : 0;
    const ccdm_conv_args& a = k.a;
    if (!on || (a.prec & ~255) || a.up == 2) return false;
    if ((a.prec & 255) != CCDM_PREC_F16X3 || a.ksize != 3 || a.stride != 1 || g.TW != 32) return false;
    if ((a.Cout & 3) != 0) return false;
    const int nchunk = (k.cin_pad + k.cin_pad_skip) / PC_CK;
    if (nchunk < 2 || nchunk > 64) return false;
    (void)NI;
    return conv_pc_lds_bytes(a.C0 + a.C1) <= 160 * 1024;
}

static unsigned long long* g_pc_timeline = nullptr;      // device buffer, allocated only under CCDM_PC_TIMELINE=1 (diagnostics)

bool conv_pc_timeline_read(unsigned long long* host, int n) {
    if (!g_pc_timeline) return false;
    return hipMemcpy(host, g_pc_timeline, (size_t)n * 8, hipMemcpyDeviceToHost) == hipSuccess;
}

int launch_conv_pc(const ConvK& k_in, hipStream_t s) {
    ConvK k = k_in;
    static const bool want_tl = getenv("CCDM_PC_TIMELINE") != nullptr;
    if (want_tl && !g_pc_timeline) {
        if (hipMalloc(&g_pc_timeline, 1024 * 8) != hipSuccess) return fail("conv: timeline buffer");
        (void)hipMemset(g_pc_timeline, 0, 1024 * 8);
    }
    k.timeline = g_pc_timeline;
    const size_t lds = conv_pc_lds_bytes(k.a.C0 + k.a.C1);
    static bool configured = false;
    if (!configured) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(k_conv_pc), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)
            return fail("conv: cannot reserve LDS for the producer/consumer kernel");
        configured = true;
    }
    dim3 grid(k.a.N * k.slices, k.ntiles);
    hipLaunchKernelGGL(k_conv_pc, grid, dim3((PC_NL + 4) * 64), lds, s, k);
    return 0;
}

}  // namespace ccdm
