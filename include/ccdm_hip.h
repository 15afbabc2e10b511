/*
 * ccdm_hip.h — C ABI of libccdm_hip.so: the MI355X (gfx950) kernels of the categorical reverse-diffusion
 * sampler, plus the step executor ("engine") that replays them for T denoise steps.
 *
 * Plain pointers and sizes only; no torch types.  Every pointer marked `dev` is HIP device memory owned by
 * the caller (the Python host allocates it through torch); the library never allocates device memory and
 * keeps no global state besides a thread-local error string.  Every launch goes on the `stream` the caller
 * passes (a hipStream_t cast to void*; NULL = default stream).  Return value: 0 = ok, negative = error
 * (text via ccdm_last_error_string()).
 *
 * The reference (/root/reference, pure Python on torch) has no FFI; each entry point cites the reference
 * function whose arithmetic it replaces.  Activations are NHWC fp32; the reference's BCHW tensors are
 * re-laid-out once at the boundary (ccdm_nchw_to_nhwc / the Python host).
 */
#ifndef CCDM_HIP_H
#define CCDM_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CCDM_ABI_VERSION 10
#define CCDM_MAX_CHANNELS 1024      /* max C0+C1 of a GroupNorm'ed conv input */
#define CCDM_STATS_MAX_SLICES 64    /* partial-statistics slices per sample a GroupNorm consumer reads (more: ccdm_stats_fold) */
#define CCDM_STATS_FOLD_SLICES 16   /* what ccdm_stats_fold reduces a larger slice count to */

int ccdm_version(void);
const char* ccdm_last_error_string(void);

/* ---------------------------------------------------------------------------------------------------
 * Per-channel statistics of an NHWC tensor, the form every GroupNorm consumer reads:
 *   stats[n][s][c][0] = sum_x, stats[n][s][c][1] = sum_x^2 over the pixels slice s covered (fp64),
 *   s in [0, slices).  The consumer adds the slices in ascending s — a fixed order, so results are
 *   run-to-run deterministic (no floating-point atomics anywhere).
 * Replaces the statistics half of GroupNorm32 (unet_openai/nn.py:17-19 -> torch group_norm).
 * ------------------------------------------------------------------------------------------------- */
int ccdm_gn_stats(const float* x /*dev [N,HW,C]*/, int N, int HW, int C, int slices,
                  double* stats /*dev [N,slices,C,2]*/, void* stream);

/* ---------------------------------------------------------------------------------------------------
 * Fused  [GroupNorm(32 groups) -> (FiLM) -> SiLU ->]  conv KxK  [+bias +emb | +residual]  (+output stats)
 * on NHWC fp32, implicit GEMM on the matrix cores.
 * Replaces: ResBlock.in_layers / out_layers (unet.py:186-219, 242-262), skip_connection 1x1 (:221-228),
 * Downsample.op (:137-146), Upsample nearest-x2 + conv (:106-116), AttentionBlock.norm+qkv and proj_out
 * (:291-300, conv1d k=1 == 1x1 conv over tokens), the stem conv (:517) and the head GN-SiLU-conv (:701-707).
 * ------------------------------------------------------------------------------------------------- */
enum { CCDM_ACT_NONE = 0, CCDM_ACT_SILU = 1 };
enum { CCDM_PREC_F32 = 0,      /* v_mfma_f32_32x32x2_f32: exact fp32 products and accumulation          */
       CCDM_PREC_F16X3 = 1,    /* fp16 hi/lo split, 3 x v_mfma_f32_32x32x16_f16, fp32 accumulate (~2^-22) */
       CCDM_PREC_F16 = 2 };    /* OPT-IN fast mode (ccdm_conv2d only; weights packed as for CCDM_PREC_F16X3): ONE fp16 MFMA per product — operands
                                  rounded to fp16 (~2^-11 each), fp32 accumulate.  Narrower arithmetic than the reference's: outside the parity
                                  contract, never a default, never the benchmarked configuration (tools/fast_mode_report.py prints its error) */
/* Range of CCDM_PREC_F16X3: a staged activation a (after GroupNorm/SiLU, or the raw input where there is none) must satisfy
 * |a| < 4094.  Beyond that its fp16 hi half is infinite and every output the element reaches is NaN/Inf — never a silently
 * clipped number; the step epilogue (ccdm_post_args.range_flag) turns that into a sticky device flag the host checks.
 * Below |a| ~ 2e-3 the split keeps an ABSOLUTE error <= 2^-29 (fp16 subnormal spacing after the 2^4 pre-scale). */

/* Diagnostic bit of ccdm_conv_args.prec (bits 8 and up are diagnostics; the arithmetic is prec & 255): run the general staging kernel
 * even where a specialised one applies (the LDS-free 1x1 kernel).  The parity tests use it to require identical bits from both. */
#define CCDM_DIAG_GENERAL_KERNEL (2048 << 8)

typedef struct ccdm_conv_args {
    /* input: virtual channel concat [in0 | in1] (in1 may be NULL); both [N,Hin,Win,C*] */
    const float* in0; const float* in1; int32_t C0; int32_t C1;
    /* GroupNorm over the concatenated channels: partial stats of each source (NULL = no normalisation) */
    const double* stats0; const double* stats1; int32_t slices0; int32_t slices1;
    const float* gamma; const float* beta;          /* dev [C0+C1] */
    float eps; int32_t act;                         /* CCDM_ACT_* applied after the affine */
    /* FiLM (use_scale_shift_norm): h = GN(h)*(1+scale)+shift, scale|shift = emb row [film_off, +2*C) */
    int32_t film; int32_t film_off;
    /* geometry */
    int32_t N, Hin, Win, Hout, Wout;
    int32_t ksize;                                  /* 1 or 3 (pad = ksize/2) */
    int32_t stride;                                 /* 1 or 2 */
    int32_t up;                                     /* 1: nearest x2 upsample of the input on load (weights: ccdm_pack_conv_weight);
                                                     * 2: the same operator in sub-pixel form (weights: ccdm_pack_upconv_weight,
                                                     *    out_slices = ccdm_upconv_slices) — see below */
    /* weights, packed by ccdm_pack_conv_weight for `prec` */
    const void* w; const float* bias; int32_t Cout; int32_t prec;
    /* epilogue */
    const float* emb_table; int32_t emb_stride; int32_t emb_off;   /* += emb_table[row*emb_stride + emb_off + c]; emb_off<0: none */
    const int32_t* emb_row_of_sample;               /* dev [N] or NULL (row = 0) ; row += *step_ptr */
    const int32_t* step_ptr;                        /* dev scalar or NULL (0) */
    const float* resid;                             /* dev [N,Hout,Wout,Cout] added last, or NULL */
    float* out;                                     /* dev [N,Hout,Wout,Cout] */
    double* out_stats; int32_t out_slices;          /* dev [N,out_slices,Cout,2] or NULL; out_slices = ccdm_conv_slices() */
    /* fused 1x1 skip connection (ResBlock.skip_connection, unet.py:221-228,262): out += W_s * [skip0 | skip1] as extra
     * K-segments of the same GEMM (raw input, centre tap).  skip_w is packed with ksize 1 and the SAME per-channel
     * exponents as w (ccdm_pack_conv_weight_ex with a shared absmax); its bias is pre-added into `bias` by the host.
     * Requires stride 1, up 0, skip tensors [N,Hout,Wout,SC*].  skip0 == NULL: none. */
    const float* skip0; const float* skip1; int32_t SC0; int32_t SC1;
    const void* skip_w;
    /* 1 / 2: latency slicing — more, shorter workgroups per sample (ccdm_conv_slices_ex(..., fine): up to 32 / 64 slices where the
     * default rule gives fewer, e.g. 32 instead of 12 at 128x128; 32x32 images on 8x16 tiles, 16x16 on 8x8) for batches too small to fill the chip.  The
     * statistics partials — and with them the last bit of a GroupNorm — depend on the slice count, so a run is bit-reproducible
     * across batch shardings only within one slicing mode; 0 (default) is the batch-size-independent rule. */
    int32_t fine_slices;
} ccdm_conv_args;

/* number of statistics slices the conv kernel produces for an Hout x Wout output (depends only on the
 * spatial size, never on N, so sharding the batch does not change any rounding).  Large images yield more than
 * CCDM_STATS_MAX_SLICES (one slice = one workgroup per sample: 256x512 -> 96, 512x1024 -> 384): fold them with
 * ccdm_stats_fold before handing them to a GroupNorm consumer. */
int ccdm_conv_slices(int Hout, int Wout, int stride, int ksize);
/* out_slices of any conv: geometry of the INPUT, `up` as in ccdm_conv_args (2: sub-pixel form), fine as ccdm_conv_args.fine_slices */
int ccdm_conv_slices_ex(int Hin, int Win, int ksize, int stride, int up, int fine);
/* out_slices of THIS conv (every field but out / out_stats / out_slices filled in): the rule above, except where the layer runs a
 * kernel with its own tiling — 3x3 F16X3 convs of images of at most 256 pixels leave one slice per 8x8 tile (16x16: 4, where the
 * rule says 2).  A function of the layer's shape and operands, never of N.  What a caller should size out_stats by. */
int ccdm_conv_out_slices(const ccdm_conv_args* a);
/* Upsample (nearest x2) + conv 3x3 in sub-pixel form (unet.py:106-116), `up = 2`:
 *   out(2y+dy, 2x+dx) = sum over a,b in {0,1} of W'[dy,dx][a,b] . in(y+dy-1+a, x+dx-1+b),
 *   W'[dy][..][a] = the 3x3 kernel rows that land on low-resolution row y+dy-1+a  (dy=0: {r0}, {r1+r2}; dy=1: {r0+r1}, {r2}; columns alike):
 * four 2x2 convs of the LOW-resolution input — 4 instead of 9 taps per output pixel and a quarter of the staged halo.  The four
 * phases run as adjacent output-channel tiles of one launch (one block computes all four from one staged halo).  Needs CCDM_PREC_F16X3, ksize 3, stride 1, Cout % 32 == 0, no residual / fused
 * skip (ccdm_upconv_supported).  Sums of kernel taps are formed in fp64 and rounded once to fp32 before the fp16 split: results
 * differ from `up = 1` by fp32 rounding only.  out_slices = ccdm_upconv_slices(Hin, Win): the slices of the low-resolution
 * tiling (x 4 for inputs narrower than 16 pixels, where every phase runs in its own block); a function of the spatial size only. */
int ccdm_upconv_supported(int Cin, int Cout, int prec);
int ccdm_upconv_slices(int Hin, int Win);
size_t ccdm_pack_upconv_weight(const float* oihw /*[Cout,Cin,3,3]*/, int Cout, int Cin, int prec, void* out);
/* out[n][j] = sum of in[n][i] over i in [j*S_in/S_out, (j+1)*S_in/S_out), ascending (fixed order); S_out <= CCDM_STATS_MAX_SLICES (the engine folds to CCDM_STATS_FOLD_SLICES) */
int ccdm_stats_fold(const double* in /*dev [N,S_in,C,2]*/, int N, int S_in, int C, int S_out, double* out /*dev [N,S_out,C,2]*/, void* stream);
int ccdm_conv2d(const ccdm_conv_args* a, void* stream);

/* ---------------------------------------------------------------------------------------------------
 * The resamplers of the `resblock_updown=True` topology (ResBlock(up=True / down=True), unet.py:202-219 and :242-250, built from
 * Downsample(ch, use_conv=False) = AvgPool2d(2), unet.py:137-141, and Upsample(ch, use_conv=False) = nearest x2, unet.py:106-114):
 *   out_act = R(act(GroupNorm(in)))    — the `h` branch of a down block, between in_layers' SiLU and its conv (NULL: not wanted)
 *   out_raw = R(in)                    — the `x` branch that becomes the block's residual                     (NULL: not wanted)
 * R = AvgPool2d(2) ([N,H,W,C] -> [N,H/2,W/2,C], floor; sum order ((x00 + x01) + x10) + x11 like torch's CPU kernel, so out_raw is
 * bit-exact) or nearest x2 ([N,H,W,C] -> [N,2H,2W,C]).  GroupNorm as in ccdm_conv_args: `stats` are the producer's partial-statistics
 * slices of `in` (NULL: no normalisation; then gamma/beta are unused); C % 4 == 0, with GroupNorm C % 32 == 0.
 * ------------------------------------------------------------------------------------------------- */
#define CCDM_RESAMPLE_AVGPOOL2 0
#define CCDM_RESAMPLE_NEAREST_UP2 1
typedef struct ccdm_resample_args {
    const float* in;            /* dev NHWC fp32 [N,Hin,Win,C] */
    int32_t C;
    const double* stats;        /* dev [N,slices,C,2] or NULL */
    int32_t slices;
    const float* gamma;         /* dev [C] */
    const float* beta;          /* dev [C] */
    float eps;
    int32_t act;                /* CCDM_ACT_NONE / CCDM_ACT_SILU, applied to out_act only */
    int32_t N, Hin, Win;
    int32_t mode;               /* CCDM_RESAMPLE_* */
    float* out_act;
    float* out_raw;
} ccdm_resample_args;
int ccdm_resample(const ccdm_resample_args* a, void* stream);

/* ---------------------------------------------------------------------------------------------------
 * The stem conv for inputs of at most 4 channels (LIDC: 2 classes + 1 image channel), CCDM_PREC_F16X3:
 *     h = conv3x3( cat([one_hot(x_t), image], 1) ) + bias                          unet.py:517 fed by unet.py:760
 * x_t arrives as the uint8 class index the epilogue leaves and the one-hot is built while staging; the image is
 * read from channels [K, Cs) of `xin` (its channels [0, K) are ignored — the epilogue need not write a one-hot there:
 * ccdm_post_args.xin = NULL).  The K axis of the GEMM is (tap, channel): 3 k-steps instead of the general kernel's 9.
 * Built for Cs == 4, H % 8 == 0, W % 32 == 0, Cout % 32 == 0 (ccdm_stem_conv_supported); out_slices = ccdm_conv_slices(H, W, 1, 3).
 * Weights: ccdm_pack_stem_weight(oihw [Cout, Cin <= 4, 3, 3]).
 * ------------------------------------------------------------------------------------------------- */
typedef struct ccdm_stem_args {
    const uint8_t* xt;          /* dev [N,H*W] class index of x_t */
    const float* xin;           /* dev NHWC fp32 [N,H,W,Cs]; channels [K, Cs): the conditioning image (zero beyond it) */
    int32_t Cs, K;
    const void* w; const float* bias;
    int32_t N, H, W, Cout;
    float* out;                 /* dev [N,H,W,Cout] */
    double* out_stats; int32_t out_slices;
} ccdm_stem_args;
int ccdm_stem_conv_supported(int Cs, int Cout, int H, int W, int prec);
size_t ccdm_pack_stem_weight(const float* oihw, int Cout, int Cin, void* out);      /* out == NULL: returns the byte count */
int ccdm_stem_conv(const ccdm_stem_args* a, void* stream);

/* F16X3 range diagnostics: max |a| over everything this conv stages — the main input after GroupNorm (+ SiLU) where it normalises on
 * load, raw otherwise, and the raw input of the fused 1x1 skip segment — before the kernel's 2^4 pre-scale; Inf if any value is not
 * finite.  The result is max'ed INTO *out (dev float, >= 0: zero it first); the split is exact for values below CCDM_F16X3_LIMIT.
 * Not on the sampling path: tools/range_report.py and the host's per-layer fp32 fallback call it. */
#define CCDM_F16X3_LIMIT 4094.0f
int ccdm_conv_input_absmax(const ccdm_conv_args* a, float* out /*dev [1]*/, void* stream);

/* host-side weight packing.  `oihw` = reference layout [Cout,Cin,k,k] (conv2d) / [Cout,Cin,1] (conv1d).
 * Returns the packed size in bytes (call with out=NULL to query). */
size_t ccdm_pack_conv_weight(const float* oihw, int Cout, int Cin, int ksize, int prec, void* out);
/* same, with the per-output-channel max|W| given by the caller (dev-independent host array [Cout], or NULL): two weight
 * sets that accumulate into one GEMM (conv + fused skip) must share their F16X3 power-of-two pre-scale. */
size_t ccdm_pack_conv_weight_ex(const float* oihw, int Cout, int Cin, int ksize, int prec, const float* cout_absmax, void* out);

/* ---------------------------------------------------------------------------------------------------
 * Self-attention core over tokens, softmax(q k^T * ch^-1/2) v per head, streaming (score matrix never in HBM).
 * qkv: [N,T,3C] (the qkv 1x1 conv output, NHWC), out: [N,T,C].
 * order 0 = QKVAttentionLegacy (channel = head*3ch + {q,k,v}*ch + c, unet.py:343-360),
 * order 1 = QKVAttention        (channel = {q,k,v}*C + head*ch + c,   unet.py:376-395).
 * order | CCDM_ATTENTION_FORCE_VALU: the vector-pipe kernel (plain fp32 FMAs, head widths 4, 8, 12, 16, 24, 32, 48, 64) instead of the
 * matrix-core one, whose fp16 hi/lo split of q, k, v has the range of CCDM_PREC_F16X3 — what the host pins an attention core to
 * whose operands left that range ("<block>.attention" in DenoisingModel.f32_layers).
 * ------------------------------------------------------------------------------------------------- */
#define CCDM_ATTENTION_FORCE_VALU 256
int ccdm_attention(const float* qkv, float* out, int N, int T, int C, int heads, int order, void* stream);

/* ---------------------------------------------------------------------------------------------------
 * First half of an AttentionBlock in one launch, for the low-resolution stages:
 *     a = attention(qkv(GroupNorm32(x)))                                          unet.py:291-311, core :343-360 / :376-395
 * One workgroup per (sample, head): GroupNorm on load, the head's 96 rows of the qkv 1x1 conv and softmax(q k^T) v run out of
 * registers and LDS; the 3C-wide qkv tensor never reaches memory.  proj_out + residual remain a ccdm_conv2d (1x1, resid = x).
 * Built for head width 32 and (T, C) in {64,256} x {96,128} and 128 x {128,256} — ccdm_norm_qkv_attention_supported(); every
 * other geometry runs as ccdm_conv2d (GN + qkv) -> ccdm_attention.  CCDM_PREC_F16X3 arithmetic.
 *   wqkv : qkv.weight [3C,C,1] with rows in LEGACY order (channel = head*96 + {q,k,v}*32 + d; the host permutes the rows of a
 *          `use_new_attention_order` model), packed by ccdm_pack_conv_weight(…, Cout=3C, Cin=C, ksize=1, CCDM_PREC_F16X3); bqkv alike.
 * ------------------------------------------------------------------------------------------------- */
typedef struct ccdm_attn_block_args {
    const float* x;                                  /* dev [N,T,C] block input (NHWC, T = h*w) */
    const double* stats; int32_t slices;             /* dev [N,slices,C,2] partial statistics of x */
    const float* gamma; const float* beta; float eps;/* AttentionBlock.norm */
    const void* wqkv; const float* bqkv;             /* packed qkv weights (see above), dev [3C] bias in legacy order */
    float* out;                                      /* dev [N,T,C] attention output, channel = head*32 + d */
    int32_t N, T, C, heads;
} ccdm_attn_block_args;
int ccdm_norm_qkv_attention_supported(int T, int C, int heads);
int ccdm_norm_qkv_attention(const ccdm_attn_block_args* a, void* stream);

/* ---------------------------------------------------------------------------------------------------
 * Time conditioning for a list of steps (depends only on t, so computed once per run):
 *   emb = Linear(SiLU(Linear(sinusoid(t))))                  unet.py:506-510,:758
 *   out[s] = Wcat * SiLU(emb) + bcat                          all ResBlock.emb_layers at once, unet.py:205-211,:250
 * `sinus` is timestep_embedding(t, model_channels) (nn.py:103-121), evaluated by the host with the same torch
 * ops as the reference: t*freq reaches 1e3 rad, so a 1-ulp difference in exp() would already move cos/sin
 * by 1e-5 — the table is [S, mc] floats, not worth a second libm.
 * ------------------------------------------------------------------------------------------------- */
int ccdm_time_table(const float* sinus /*dev [S,mc]*/, int S, int model_channels,
                    const float* w0, const float* b0, const float* w2, const float* b2,  /* dev, reference [out,in] layout */
                    const float* wcat /*dev [E,4mc]*/, const float* bcat /*dev [E]*/, int E,
                    float* emb_out /*dev [S,4mc] or NULL*/, float* out /*dev [S,E]*/, void* stream);

/* ---------------------------------------------------------------------------------------------------
 * Fused epilogue of one denoise step, one thread per pixel:
 *   x0 = softmax_K(head) (or head itself)                                     unet.py:706
 *   P  = theta_post_prob(x_t, x0, t)   (O(K) closed form, in registers)       diffusion_denoising.py:99-128
 *   P  = max(P, 1e-12); P^ = P / sum_K P                                      :204 ; torch Categorical
 *   t>1 : x_{t-1} = argmax_k P^_k / E_k (first index wins)                    one_hot_categorical.py:30-32
 *   t==1: "confidence" -> P^ (fp32) ; "majority" -> one_hot(argmax P^) int64  :46-54 ; diffusion_denoising.py:208-212
 * E is read from `noise` (host-drawn Exp(1), order ((n*H+h)*W+w)*K+k) or generated with Philox4x32-10.
 * ------------------------------------------------------------------------------------------------- */
enum { CCDM_STEP_SAMPLE = 0, CCDM_STEP_LAST_CONFIDENCE = 1, CCDM_STEP_LAST_MAJORITY = 2, CCDM_STEP_LAST_KEEP = 3,
       CCDM_STEP_SOFTMAX_ONLY = 4 /* out_probs = x0 (the U-Net output itself): forward_step, diffusion_denoising.py:161-162 */ };

/* The per-run fields of the epilogue as a DEVICE-resident block (ABI 7).  With ccdm_post_args.run set the kernel reads these eight
 * values from the block instead of the argument struct, like it reads the step row through step_ptr: a captured HIP graph of the
 * denoise step then survives a new Philox key, another noise buffer or another output pointer (the host rewrites the block with
 * a stream-ordered one-thread launch; nothing is re-captured, nothing is destroyed while earlier launches are in flight). */
typedef struct ccdm_post_run {
    const float* noise; int64_t noise_step_stride;
    uint64_t philox_seed; uint32_t sample_offset; int32_t noise_row0;
    float* out_probs; int64_t* out_onehot; float* posterior_out;
} ccdm_post_run;

#define CCDM_MAX_CLASSES 255        /* x_t is a uint8 class index; K <= 32 keeps a pixel's classes in registers, more go through LDS rows */
#define CCDM_POST_DIAG_MANY 256     /* diagnostic bit of ccdm_post_args.softmax: run the many-class (LDS-row) kernel at any K (parity tests) */
typedef struct ccdm_post_args {
    const float* head;           /* dev [N,HW,head_stride] head conv output (logits, or probabilities if !softmax), first K channels used */
    int32_t softmax;             /* bit 0: apply softmax over K first (| CCDM_POST_DIAG_MANY) */
    int32_t head_stride;         /* floats per pixel of `head` (>= K; the head conv pads K up to a multiple of 4) */
    const uint8_t* xt;           /* dev [N,HW] class index of x_t */
    int32_t N, HW, K;
    /* per-step coefficients: row = *step_ptr (0 if NULL) of step_table = {alpha_t, cumalpha_tm1, mode, 0} */
    const float* step_table; const int32_t* step_ptr;
    /* noise */
    const float* noise; int64_t noise_step_stride;        /* dev or NULL -> Philox; row r of the buffer is step row noise_row0 + r */
    uint64_t philox_seed; uint32_t sample_offset;         /* global index of sample 0 (batch sharding) */
    /* outputs */
    uint8_t* xt_next;            /* dev [N,HW] (may alias xt) */
    float* xin; int32_t xin_stride;  /* dev [N,HW,xin_stride]: one-hot written to channels [0,K) ; or NULL */
    float* out_probs;            /* dev [N,HW,K] fp32   (confidence) or NULL */
    int64_t* out_onehot;         /* dev [N,HW,K] int64  (majority)   or NULL */
    float* posterior_out;        /* dev [N,HW,K] optional debug/teacher-forcing tap of P^ , or NULL */
    int32_t noise_row0;          /* step row the first row of `noise` belongs to (host noise uploaded in blocks of steps) */
    int32_t* range_flag;         /* dev scalar or NULL: set to 1 (sticky, never cleared by the kernel) when the head output of
                                    any pixel is not finite — the signature of an F16X3 range overflow upstream (see above) */
    const ccdm_post_run* run;    /* dev block or NULL: when set, noise / noise_step_stride / noise_row0 / philox_seed / sample_offset /
                                    out_probs / out_onehot / posterior_out are read from it and the fields above are ignored */
} ccdm_post_args;

int ccdm_posterior_sample(const ccdm_post_args* a, void* stream);

/* ---------------------------------------------------------------------------------------------------
 * The head of the U-Net and the epilogue above in ONE launch, for few classes (9 K <= 32) and 32 head channels, CCDM_PREC_F16X3:
 *     logits = conv3x3(SiLU(GroupNorm32(x))) + bias   (self.out, unet.py:701-707)   ->   ccdm_posterior_sample's arithmetic on them
 * (the same device function: identical bits from identical logits); the logits never reach memory.  The conv's taps are the N
 * dimension of a 1x1 product over the halo tile (9 K columns), summed per pixel afterwards: results equal the general kernel's to fp32
 * rounding.  `post` is a ccdm_post_args whose head / head_stride / xin fields are unused (x_t travels as the uint8 index).
 * Built for C == 32, 2 <= K <= 3, H % 8 == 0, W % 32 == 0 (ccdm_head_posterior_supported).  Weights: ccdm_pack_head_weight.
 * ------------------------------------------------------------------------------------------------- */
typedef struct ccdm_head_args {
    const float* x;             /* dev NHWC fp32 [N,H,W,C]: the last ResBlock's output */
    const double* stats; int32_t slices;        /* its partial statistics [N,slices,C,2] */
    const float* gamma; const float* beta; float eps;       /* out.0 (GroupNorm32) */
    const void* w; const float* bias;           /* out.2: ccdm_pack_head_weight(oihw [K,C,3,3]); dev [K] */
    int32_t N, H, W, C, K;
    float* logits_out;          /* dev [N,H*W,K] optional tap of the logits (tests), or NULL */
} ccdm_head_args;
int ccdm_head_posterior_supported(int C, int K, int H, int W, int prec);
size_t ccdm_pack_head_weight(const float* oihw, int K, int Cin, void* out);         /* out == NULL: returns the byte count */
int ccdm_head_posterior(const ccdm_head_args* a, const ccdm_post_args* post, void* stream);

/* ---------------------------------------------------------------------------------------------------
 * LIDC metrics, device part: for every image and every pair (i, j) of class-index maps
 * a[img][i], b[img][j], the per-class pixel counts out[img][i][j][k] = {|a==k & b==k|, |a==k | b==k|}.
 * Replaces the [B,S,S',HW,K] boolean broadcast of `batched_distance` / `iou`
 * (evaluation/evaluate_lidc_uncertainty.py:27-39); GED and Hungarian-matched IoU follow on the host from the
 * exact integer counts.
 * ------------------------------------------------------------------------------------------------- */
int ccdm_pairwise_class_counts(const uint8_t* a /*dev [B,S,HW]*/, const uint8_t* b /*dev [B,L,HW]*/, int B, int S, int L,
                               int HW, int K, int32_t* out /*dev [B,S,L,K,2]*/, void* stream);

/* ---------------------------------------------------------------------------------------------------
 * Training-time forward pieces of the categorical diffusion, BCHW fp32 like the reference's
 * tensors, per-sample coefficients (the host resolves t -> alpha_t / cumalpha_{t-1} incl. the t == 1 override).
 *   ccdm_mix_uniform : out = s[n]*x + (1-s[n])/K.   With s = 1-beta_t it is the probability table of
 *                      DiffusionModel.q_xt_given_xtm1 (diffusion_denoising.py:72-78), with s = cumalpha_t of
 *                      q_xt_given_x0 (:80-86).
 *   ccdm_theta_post  : prob_mode 0: DiffusionModel.theta_post (:88-97)  q(x_{t-1} | x_t, x_0), both inputs any float
 *                      tensors (one-hot in the reference's use); prob_mode 1: theta_post_prob (:99-129), the second
 *                      input a distribution over x_0 — O(K) closed form of the reference's [B,K,K,H,W] product.
 *   ccdm_kl_clamped  : p*(log p - log max(q, floor)), 0 where p == 0: the diffusion term of Trainer.train_step
 *                      (trainer.py:266-270, kl_div(log(clamp(q, 1e-12)), p, reduction='none')).
 * K in [2, 32].  a, c, s: dev [N] fp32.
 * ------------------------------------------------------------------------------------------------- */
int ccdm_mix_uniform(const float* x /*dev [N,K,HW]*/, const float* s, int N, int K, int HW, float* out, void* stream);
int ccdm_theta_post(const float* xt /*dev [N,K,HW]*/, const float* x0 /*dev [N,K,HW]*/, const float* a, const float* c,
                    int N, int K, int HW, int prob_mode, float* out /*dev [N,K,HW]*/, void* stream);
int ccdm_kl_clamped(const float* p, const float* q, size_t n, float floor, float* out, void* stream);

/* ---------------------------------------------------------------------------------------------------
 * DINO ViT-S/8 key-feature extractor (reference call sites ddpm/models/dino.py:211-229,279-305 and
 * condition_encoder.py:26-46; the network itself is facebookresearch/dino's VisionTransformer, fetched by the reference
 * with torch.hub (dino.py:58-82) and therefore not part of /root/reference: restated from its published form).
 * The linear layers run on ccdm_conv2d as 1x1 convs over a [N, T_alloc/16, 16, C] token image; these cover the rest:
 *   ccdm_attention_ex : ccdm_attention with T_alloc >= T token rows allocated per sample (only the first T are tokens)
 *   ccdm_layernorm    : nn.LayerNorm(C, eps) over the last axis of [rows, C]
 *   ccdm_gelu         : nn.GELU(), exact erf form
 * ------------------------------------------------------------------------------------------------- */
int ccdm_attention_ex(const float* qkv /*dev [N,T_alloc,3C]*/, float* out /*dev [N,T_alloc,C]*/, int N, int T, int T_alloc, int C,
                      int heads, int order, void* stream);
int ccdm_layernorm(const float* x, const float* gamma, const float* beta, float eps, long rows, int C, float* out, void* stream);
int ccdm_gelu(const float* x, size_t n, float* out, void* stream);

/* debugging aid: phase timestamps (s_memtime) one block of the last conv launched with ablation bit 16 recorded */
int ccdm_debug_read_timeline(unsigned long long* host, int n);

/* boundary re-layout helpers */
int ccdm_nchw_to_nhwc(const float* src, float* dst, int N, int C, int HW, int dst_stride, int dst_off, void* stream);
int ccdm_onehot_to_xin(const uint8_t* idx, float* xin, int N, int HW, int K, int xin_stride, void* stream);

/* ---------------------------------------------------------------------------------------------------
 * Step executor.  The host describes one denoise step as a list of ops (fully resolved device pointers);
 * ccdm_engine_run replays it n_steps times with a device-resident step counter, optionally through a
 * HIP graph captured on first use.  One engine per (model, N, H, W); single-threaded like the
 * reference's DenoisingModel.forward_denoising loop (diffusion_denoising.py:189-212).
 * ------------------------------------------------------------------------------------------------- */
typedef struct ccdm_engine ccdm_engine;

ccdm_engine* ccdm_engine_create(int32_t* step_counter /*dev scalar*/);
void ccdm_engine_destroy(ccdm_engine* e);
int ccdm_engine_add_conv(ccdm_engine* e, const ccdm_conv_args* a);          /* step_ptr is overridden with the engine's counter */
int ccdm_engine_add_attention(ccdm_engine* e, const float* qkv, float* out, int N, int T, int C, int heads, int order);
int ccdm_engine_add_norm_qkv_attention(ccdm_engine* e, const ccdm_attn_block_args* a);
int ccdm_engine_add_stats_fold(ccdm_engine* e, const double* in, int N, int S_in, int C, int S_out, double* out);
int ccdm_engine_add_resample(ccdm_engine* e, const ccdm_resample_args* a);
int ccdm_engine_add_stem(ccdm_engine* e, const ccdm_stem_args* a);
/* the LAST op of the step: head conv + the epilogue set by ccdm_engine_set_epilogue in one launch (no separate epilogue launch then;
 * ccdm_engine_run needs with_epilogue = 1) */
int ccdm_engine_add_head_posterior(ccdm_engine* e, const ccdm_head_args* a);
int ccdm_engine_set_epilogue(ccdm_engine* e, const ccdm_post_args* a);      /* run after the ops of each step */
int ccdm_engine_num_ops(const ccdm_engine* e);
int ccdm_engine_num_captures(const ccdm_engine* e);   /* how often ccdm_engine_run has captured + instantiated the step's HIP graph so far */
/* per-run mutable fields of the epilogue (everything else is fixed at build time).  With a run block (ccdm_engine_set_run_block:
 * caller-owned device memory of at least sizeof(ccdm_post_run) bytes, set once before the first run) the values travel to the
 * device by a one-thread launch at the head of the next ccdm_engine_run, on its stream, and the captured graph of the step stays
 * valid; without one they are kernel arguments and a change re-captures the graph. */
int ccdm_engine_set_run_block(ccdm_engine* e, void* dev_block);
int ccdm_engine_set_run(ccdm_engine* e, const float* noise, int64_t noise_step_stride, int32_t noise_row0,
                        uint64_t philox_seed, uint32_t sample_offset,
                        float* out_probs, int64_t* out_onehot, float* posterior_out);
/* run `n_steps` denoise steps starting at table row `first_row`; use_graph: 0 eager launches, 1 HIP graph of one step */
int ccdm_engine_run(ccdm_engine* e, int first_row, int n_steps, int with_epilogue, int use_graph, void* stream);
/* timing taps: record HIP events around every launch of op `op_index` during the following runs (at most `capacity` launches
 * per series; a run that starts at table row 0 starts a new series; tapped runs launch eagerly).  Several ops may be tapped;
 * op_index < 0 removes every tap.  ccdm_engine_profile_read returns the number of samples of one tapped op and their
 * mean/min/max in ms. */
int ccdm_engine_profile_op(ccdm_engine* e, int op_index, int capacity);
int ccdm_engine_profile_read(ccdm_engine* e, int op_index, double* mean_ms, double* min_ms, double* max_ms);
/* ccdm_conv_input_absmax of every conv op of the step on the tensors the last run left behind: out[i] = max(out[i], ...) for conv op i;
 * for an attention-core op the largest |q|, |k|, |v| of its qkv tensor (what the core's own fp16 split stages; the fused
 * norm+qkv+attention op keeps qkv on chip and is not covered: probe an engine that runs the two launches).  Other ops: untouched.
 * out: dev float [ccdm_engine_num_ops], zeroed by the caller (calls accumulate: max over several steps of a run).
 * `row`: the step-table row the probed activations were produced with — GroupNorm's FiLM scale / shift are rebuilt from
 * emb_table[emb_row_of_sample[n] + row]; row < 0 = the last row the last ccdm_engine_run executed (the device counter itself stands
 * one past it).  The step counter is set to `row` for the probe and put back to where the run left it. */
int ccdm_engine_input_absmax(ccdm_engine* e, float* out, int row, void* stream);
/* describe op i: writes a short text ("conv3x3 32->32 @128x128 gn silu ...") */
int ccdm_engine_describe_op(const ccdm_engine* e, int i, char* buf, int buflen);

#ifdef __cplusplus
}
#endif
#endif /* CCDM_HIP_H */
