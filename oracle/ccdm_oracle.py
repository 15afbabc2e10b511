"""ORACLE — TEST INFRASTRUCTURE ONLY.  Not part of the product path.

CPU restatement (torch-CPU fp32, functional style) of the reference's categorical reverse-diffusion
sampler.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module;
the product (ccdm_stochastic_segmentation_amd/) never does and fails loudly without its HIP library.

Where the arithmetic lives: the reference is pure Python on top of third-party torch (requirements.txt
pins torch==1.7.0; this image has torch 2.10.0 CPU) — conv2d / group_norm / silu / softmax / einsum /
multinomial are torch ops, so the restatement calls the same torch-CPU ops in the same order.

PARITY PIN: the reference has no tests and no golden vectors.  This oracle is pinned against
outputs of the reference itself, captured in this container by tools/gen_goldens.py (which imports
/root/reference/ddpm/models) and committed as tests/golden/*.npz; tests/test_oracle_golden.py checks every
function below against them.

The U-Net topology is inferred from the state_dict keys alone (independent of the product's
unet_spec.py, so the two cross-check each other).

Reference citations are relative to /root/reference/.
"""
from __future__ import annotations

import math
import re
from typing import Dict, List, Optional, Sequence

import numpy as np
import torch
import torch.nn.functional as F

Tensor = torch.Tensor


# ----------------------------------------------------------------------------------------------
# A1  noise schedules                         ddpm/models/diffusion_denoising.py:18-39
# ----------------------------------------------------------------------------------------------
def linear_schedule(time_steps: int, start: float = 1e-2, end: float = 0.2):
    betas = torch.linspace(start, end, time_steps)
    alphas = 1 - betas
    cumalphas = torch.cumprod(alphas, dim=0)
    return betas, alphas, cumalphas


def cosine_schedule(time_steps: int, s: float = 8e-3):
    # `s` is overwritten with 0.008 whatever is passed (:27); cumalphas[i] = f(i/T) directly in fp32 torch,
    # NOT cumprod(alphas) (:26,28); betas in python doubles then cast to fp32 (:32-37).
    t = torch.arange(0, time_steps)
    s = 0.008
    cumalphas = torch.cos(((t / time_steps + s) / (1 + s)) * (math.pi / 2)) ** 2

    def f(u):
        return math.cos((u + s) / (1.0 + s) * math.pi / 2) ** 2

    betas = torch.tensor([min(1 - f((i + 1) / time_steps) / f(i / time_steps), 0.999) for i in range(time_steps)])
    alphas = 1 - betas
    return betas, alphas, cumalphas


def make_schedule(schedule: str, time_steps: int, schedule_params: Optional[dict] = None):
    fn = {"linear": linear_schedule, "cosine": cosine_schedule}[schedule]          # :50-59
    return fn(time_steps, **schedule_params) if schedule_params is not None else fn(time_steps)


# ----------------------------------------------------------------------------------------------
# A5  step list                               ddpm/models/diffusion_denoising.py:178-187
# ----------------------------------------------------------------------------------------------
def step_values(time_steps: int, init_t: Optional[int]) -> List[int]:
    if init_t is None:
        init_t = time_steps
    if init_t > 10000:
        k = init_t % 10000
        assert 0 < k <= time_steps
        if k == time_steps:
            return list(range(k, 0, -1))
        return [round(v) for v in np.linspace(time_steps, 1, k)]     # python round on numpy float64: half-to-even
    return list(range(init_t, 0, -1))


# ----------------------------------------------------------------------------------------------
# A7  timestep embedding                      ddpm/models/unet_openai/nn.py:103-121
# ----------------------------------------------------------------------------------------------
def timestep_embedding(timesteps: Tensor, dim: int, max_period: int = 10000) -> Tensor:
    half = dim // 2
    freqs = torch.exp(-math.log(max_period) * torch.arange(start=0, end=half, dtype=torch.float32) / half)
    args = timesteps[:, None].float() * freqs[None]
    emb = torch.cat([torch.cos(args), torch.sin(args)], dim=-1)
    if dim % 2:
        emb = torch.cat([emb, torch.zeros_like(emb[:, :1])], dim=-1)
    return emb


def time_embed(sd: Dict[str, Tensor], timesteps: Tensor) -> Tensor:
    """time_embed = Linear -> SiLU -> Linear     unet.py:506-510, used :758."""
    mc = sd["time_embed.0.weight"].shape[1]
    e = timestep_embedding(timesteps, mc)
    e = F.linear(e, sd["time_embed.0.weight"], sd["time_embed.0.bias"])
    e = F.silu(e)
    return F.linear(e, sd["time_embed.2.weight"], sd["time_embed.2.bias"])


# ----------------------------------------------------------------------------------------------
# A10-A12  blocks
# ----------------------------------------------------------------------------------------------
def group_norm32(x: Tensor, w: Tensor, b: Tensor) -> Tensor:
    """GroupNorm32(32, C): fp32, 32 groups, eps 1e-5, biased variance, affine.   nn.py:17-19,93-100."""
    return F.group_norm(x.float(), 32, w, b, 1e-5).type(x.dtype)


def res_block(sd: Dict[str, Tensor], p: str, x: Tensor, emb: Tensor, updown: Optional[str] = None) -> Tensor:
    """ResBlock._forward.   unet.py:242-262.

    updown = "down" / "up": the `resblock_updown=True` variants (:202-208, :243-248) — GroupNorm and SiLU at the input resolution, then
    h and x both through AvgPool2d(2) (Downsample(ch, False), :137-141) or a nearest x2 upsample (Upsample(ch, False), :106-114),
    then in_conv."""
    h = F.silu(group_norm32(x, sd[p + "in_layers.0.weight"], sd[p + "in_layers.0.bias"]))
    if updown == "down":
        h, x = F.avg_pool2d(h, 2, 2), F.avg_pool2d(x, 2, 2)
    elif updown == "up":
        h, x = F.interpolate(h, scale_factor=2, mode="nearest"), F.interpolate(x, scale_factor=2, mode="nearest")
    else:
        assert updown is None, updown
    h = F.conv2d(h, sd[p + "in_layers.2.weight"], sd[p + "in_layers.2.bias"], padding=1)
    cout = sd[p + "in_layers.2.weight"].shape[0]
    emb_out = F.linear(F.silu(emb), sd[p + "emb_layers.1.weight"], sd[p + "emb_layers.1.bias"])[..., None, None]
    if emb_out.shape[1] == 2 * cout:                                   # use_scale_shift_norm (FiLM) :254-258
        scale, shift = torch.chunk(emb_out, 2, dim=1)
        h = group_norm32(h, sd[p + "out_layers.0.weight"], sd[p + "out_layers.0.bias"]) * (1 + scale) + shift
        h = F.conv2d(F.silu(h), sd[p + "out_layers.3.weight"], sd[p + "out_layers.3.bias"], padding=1)
    else:
        h = h + emb_out                                                # :260
        h = F.conv2d(F.silu(group_norm32(h, sd[p + "out_layers.0.weight"], sd[p + "out_layers.0.bias"])),
                     sd[p + "out_layers.3.weight"], sd[p + "out_layers.3.bias"], padding=1)
    if (p + "skip_connection.weight") in sd:                           # 1x1 conv when Cin != Cout :221-228
        x = F.conv2d(x, sd[p + "skip_connection.weight"], sd[p + "skip_connection.bias"])
    return x + h                                                       # :262


def qkv_attention_legacy(qkv: Tensor, n_heads: int) -> Tensor:
    """QKVAttentionLegacy.forward.   unet.py:343-360.  Channel index = head*3ch + {q,k,v}*ch + c."""
    bs, width, length = qkv.shape
    assert width % (3 * n_heads) == 0
    ch = width // (3 * n_heads)
    q, k, v = qkv.reshape(bs * n_heads, ch * 3, length).split(ch, dim=1)
    scale = 1 / math.sqrt(math.sqrt(ch))
    weight = torch.einsum("bct,bcs->bts", q * scale, k * scale)
    weight = torch.softmax(weight.float(), dim=-1).type(weight.dtype)
    a = torch.einsum("bts,bcs->bct", weight, v)
    return a.reshape(bs, -1, length)


def qkv_attention_new(qkv: Tensor, n_heads: int) -> Tensor:
    """QKVAttention.forward (use_new_attention_order).   unet.py:376-395."""
    bs, width, length = qkv.shape
    assert width % (3 * n_heads) == 0
    ch = width // (3 * n_heads)
    q, k, v = qkv.chunk(3, dim=1)
    scale = 1 / math.sqrt(math.sqrt(ch))
    weight = torch.einsum("bct,bcs->bts", (q * scale).view(bs * n_heads, ch, length),
                          (k * scale).view(bs * n_heads, ch, length))
    weight = torch.softmax(weight.float(), dim=-1).type(weight.dtype)
    a = torch.einsum("bts,bcs->bct", weight, v.reshape(bs * n_heads, ch, length))
    return a.reshape(bs, -1, length)


def attention_block(sd: Dict[str, Tensor], p: str, x: Tensor, n_heads: int, new_order: bool = False) -> Tensor:
    """AttentionBlock._forward.   unet.py:305-311."""
    b, c, *spatial = x.shape
    x = x.reshape(b, c, -1)
    qkv = F.conv1d(group_norm32(x, sd[p + "norm.weight"], sd[p + "norm.bias"]), sd[p + "qkv.weight"], sd[p + "qkv.bias"])
    h = (qkv_attention_new if new_order else qkv_attention_legacy)(qkv, n_heads)
    h = F.conv1d(h, sd[p + "proj_out.weight"], sd[p + "proj_out.bias"])
    return (x + h).reshape(b, c, *spatial)


def downsample(sd: Dict[str, Tensor], p: str, x: Tensor) -> Tensor:
    """Downsample with conv_resample=True: conv3x3 stride 2 pad 1.   unet.py:137-146."""
    return F.conv2d(x, sd[p + "op.weight"], sd[p + "op.bias"], stride=2, padding=1)


def upsample(sd: Dict[str, Tensor], p: str, x: Tensor) -> Tensor:
    """Upsample: nearest x2 then conv3x3.   unet.py:106-116."""
    x = F.interpolate(x, scale_factor=2, mode="nearest")
    return F.conv2d(x, sd[p + "conv.weight"], sd[p + "conv.bias"], padding=1)


def _heads(ch: int, num_heads: int, num_head_channels: int) -> int:
    return num_heads if num_head_channels == -1 else ch // num_head_channels      # unet.py:283-289


def _updown_of(cfg: dict, prefix: str, j: int) -> Optional[str]:
    """Which ResBlocks of a `resblock_updown=True` network resample (they hold no parameter that says so): the single ResBlock of
    every encoder block that closes a level (unet.py:586-600: after each `num_res_blocks` blocks, where Downsample would sit) and any
    ResBlock that is not the first layer of its decoder block (:683-697, where Upsample would sit)."""
    if not cfg.get("resblock_updown", False):
        return None
    stem, idx = prefix.rsplit(".", 1) if prefix[-1].isdigit() else (prefix, "")
    if stem == "input_blocks":
        i = int(idx)
        return "down" if i > 0 and i % (int(cfg["num_res_blocks"]) + 1) == 0 else None
    if stem == "output_blocks":
        return "up" if j > 0 else None
    return None


def _run_sequential(sd, prefix: str, h: Tensor, emb: Tensor, cfg: dict, decoder: bool = False) -> Tensor:
    """TimestepEmbedSequential.forward, dispatching on which parameters exist.   unet.py:70-84."""
    j = 0
    while True:
        p = f"{prefix}.{j}."
        if (p + "in_layers.0.weight") in sd:
            h = res_block(sd, p, h, emb, _updown_of(cfg, prefix, j))
        elif (p + "qkv.weight") in sd:
            nh = cfg.get("num_heads_upsample", cfg["num_heads"]) if decoder else cfg["num_heads"]
            h = attention_block(sd, p, h, _heads(h.shape[1], nh, cfg["num_head_channels"]),
                                cfg.get("use_new_attention_order", False))
        elif (p + "op.weight") in sd:
            h = downsample(sd, p, h)
        elif (p + "conv.weight") in sd:
            h = upsample(sd, p, h)
        elif (p + "weight") in sd and sd[p + "weight"].ndim == 4:
            h = F.conv2d(h, sd[p + "weight"], sd[p + "bias"], padding=1)          # stem, unet.py:517
        else:
            break
        j += 1
    assert j > 0, prefix
    return h


def _count(sd, stem: str) -> int:
    idx = {int(m.group(1)) for k in sd for m in [re.match(rf"{stem}\.(\d+)\.", k)] if m}
    return max(idx) + 1 if idx else 0


def unet_forward(sd: Dict[str, Tensor], cfg: dict, x: Tensor, input_condition: Tensor,
                 feature_condition: Optional[Tensor], timesteps: Tensor, taps: Optional[dict] = None) -> Dict[str, Optional[Tensor]]:
    """UNetModel.forward.   unet.py:744-808.

    cfg keys: num_heads, num_head_channels, [num_heads_upsample], [use_new_attention_order],
              [softmax_output=True], [feature_condition_idx=[]], [resblock_updown=False -> needs num_res_blocks].
    """
    emb = time_embed(sd, timesteps)                                                # :758
    h = torch.cat([x, input_condition], dim=1).float()                             # :760,:767
    hs = []
    fidx = cfg.get("feature_condition_idx", [])
    for i in range(_count(sd, "input_blocks")):                                    # :768
        if feature_condition is not None and i in fidx:
            h = torch.cat([h, feature_condition], dim=1)                           # :783-786
        h = _run_sequential(sd, f"input_blocks.{i}", h, emb, cfg)
        if taps is not None:
            taps[f"input_blocks.{i}"] = h
        hs.append(h)
    h = _run_sequential(sd, "middle_block", h, emb, cfg)                           # :794
    if taps is not None:
        taps["middle_block"] = h
    for i in range(_count(sd, "output_blocks")):                                   # :796-798
        h = torch.cat([h, hs.pop()], dim=1)
        h = _run_sequential(sd, f"output_blocks.{i}", h, emb, cfg, decoder=True)
        if taps is not None:
            taps[f"output_blocks.{i}"] = h
    g = F.silu(group_norm32(h, sd["out.0.weight"], sd["out.0.bias"]))              # :701-707
    logits = F.conv2d(g, sd["out.2.weight"], sd["out.2.bias"], padding=1)
    out = torch.softmax(logits, dim=1) if cfg.get("softmax_output", True) else logits
    ret = {"diffusion_out": out, "logits": None, "pre_softmax": logits}
    if "out_ce.2.weight" in sd:                                                    # :716-726,:805-807
        gc = F.silu(group_norm32(h, sd["out_ce.0.weight"], sd["out_ce.0.bias"]))
        ret["logits"] = F.conv2d(gc, sd["out_ce.2.weight"], sd["out_ce.2.bias"], padding=1)
    return ret


# ----------------------------------------------------------------------------------------------
# A3  posterior                               ddpm/models/diffusion_denoising.py:99-128
# ----------------------------------------------------------------------------------------------
def posterior_coeffs(alphas: Tensor, cumalphas: Tensor, t: int):
    """(alpha_t, cumalpha_{t-1}) with the t==1 override a:=0, c:=1 (:112-113)."""
    i = t - 1
    if i == 0:
        return 0.0, 1.0
    return float(alphas[i]), float(cumalphas[i - 1])


def theta_post_prob_ref(xt: Tensor, theta_x0: Tensor, a: float, c: float) -> Tensor:
    """O(K^2) form with the reference's op order (materialises [B,K,K,H,W])."""
    K = xt.shape[1]
    a_t = torch.full((xt.shape[0], 1, 1, 1), a, dtype=torch.float32)
    c_t = torch.full((xt.shape[0], 1, 1, 1, 1), c, dtype=torch.float32)
    x0 = torch.eye(K)[None, :, :, None, None]
    theta_xt_xtm1 = a_t * xt + (1 - a_t) / K
    theta_xtm1_x0 = c_t * x0 + (1 - c_t) / K
    aux = theta_xt_xtm1[:, :, None] * theta_xtm1_x0
    post = aux / aux.sum(dim=1, keepdim=True)
    return torch.einsum("bcdhw,bdhw->bchw", post, theta_x0)


def theta_post_prob(xt: Tensor, theta_x0: Tensor, a: float, c: float) -> Tensor:
    """O(K) closed form of the same posterior, fp32, fixed op order — this is the order
    the HIP epilogue implements:
        A_k = a*xt_k + (1-a)/K ;  b = (1-c)/K ;  S = sum_k A_k  (k ascending)
        r_d = x0_d / (c*A_d + b*S) ;  R = sum_d r_d (d ascending) ;  out_k = A_k * (c*r_k + b*R)
    """
    K = xt.shape[1]
    a32, c32 = np.float32(a), np.float32(c)
    u = np.float32(np.float32(1) - a32) / np.float32(K)
    b = np.float32(np.float32(1) - c32) / np.float32(K)
    A = float(a32) * xt + float(u)
    S = A[:, 0].clone()
    for k in range(1, K):
        S = S + A[:, k]
    r = theta_x0 / (float(c32) * A + float(b) * S[:, None])
    R = r[:, 0].clone()
    for k in range(1, K):
        R = R + r[:, k]
    return A * (float(c32) * r + float(b) * R[:, None])


# ----------------------------------------------------------------------------------------------
# Training-time forward pieces: per-sample t, any float xt / x0
#   diffusion_denoising.py:72-129, trainer.py:257-270
# ----------------------------------------------------------------------------------------------
def per_sample_coeffs(alphas: Tensor, cumalphas: Tensor, t: Tensor):
    """(alpha_t, cumalpha_{t-1}) per sample with the t == 1 override (:91-94, :111-113), as fp32 [N] tensors."""
    i = t.long() - 1
    a = alphas[i].clone().float()
    c = cumalphas[i - 1].clone().float()        # i == 0 wraps to the last entry, overwritten below — as in the reference
    a[i == 0] = 0.0
    c[i == 0] = 1.0
    return a, c


def q_probs(x: Tensor, s: Tensor) -> Tensor:
    """s*x + (1-s)/K per sample: q(x_t | x_{t-1}) with s = 1 - beta_t (:72-78), q(x_t | x_0) with s = cumalpha_t (:80-86)."""
    K = x.shape[1]
    s = s.float()[:, None, None, None]
    return s * x + (1 - s) / K


def theta_post_t(xt: Tensor, x0: Tensor, a: Tensor, c: Tensor) -> Tensor:
    """theta_post (:88-97): ((a*xt + (1-a)/K) * (c*x0 + (1-c)/K)) normalised over the class axis."""
    K = xt.shape[1]
    a = a[:, None, None, None]
    c = c[:, None, None, None]
    theta = (a * xt + (1 - a) / K) * (c * x0 + (1 - c) / K)
    return theta / theta.sum(dim=1, keepdim=True)


def theta_post_prob_t(xt: Tensor, theta_x0: Tensor, a: Tensor, c: Tensor) -> Tensor:
    """theta_post_prob (:99-129) with per-sample coefficients, O(K) closed form (same algebra as theta_post_prob above):
        A_k = a*xt_k + (1-a)/K ; b = (1-c)/K ; S = sum_k A_k ; r_d = theta_d / (c*A_d + b*S) ; out_k = A_k * (c*r_k + b*sum_d r_d)"""
    K = xt.shape[1]
    a = a[:, None, None, None]
    c = c[:, None, None, None]
    A = a * xt + (1 - a) / K
    b = (1 - c) / K
    S = A.sum(dim=1, keepdim=True)
    r = theta_x0 / (c * A + b * S)
    return A * (c * r + b * r.sum(dim=1, keepdim=True))


def kl_clamped(p_true: Tensor, q_pred: Tensor, floor: float = 1e-12) -> Tensor:
    """The diffusion loss term of Trainer.train_step (trainer.py:266-270): kl_div(log(clamp(q, floor)), p, 'none')
    = p * (log p - log max(q, floor)), 0 where p == 0."""
    return torch.nn.functional.kl_div(torch.log(torch.clamp(q_pred, min=floor)), p_true, reduction="none")


# ----------------------------------------------------------------------------------------------
# A6 / T2  categorical draw                   ddpm/models/one_hot_categorical.py:10-54
# ----------------------------------------------------------------------------------------------
def ordered_sum_lastdim(p: Tensor) -> Tensor:
    """Sum over the last dim in the order torch's CPU outer-reduction uses for a channels-last *view*
    (multi_row_sum cascade, ATen/native/cpu/SumKernel.cpp): blocks of 16 accumulate sequentially from 0,
    the tail accumulates separately, then tail + blocks."""
    K = p.shape[-1]
    full = (K // 16) * 16
    acc_hi = None
    for s in range(0, full, 16):
        blk = p[..., s].clone()
        for k in range(s + 1, s + 16):
            blk = blk + p[..., k]
        acc_hi = blk if acc_hi is None else acc_hi + blk
    if full == K:
        return acc_hi
    tail = p[..., full].clone()
    for k in range(full + 1, K):
        tail = tail + p[..., k]
    return tail if acc_hi is None else tail + acc_hi


def row_sum_order_lastdim(p: Tensor) -> Tensor:
    """The other order torch's CPU sum uses (row_sum, ilp_factor 4: four interleaved partial sums, the
    remainder added to partial 0, then p0+p1+p2+p3) — taken for the last < 4*Vec::size() pixels of a row
    chunk.  Identical to the sequential order for K <= 4."""
    K = p.shape[-1]
    q = K // 4
    parts = []
    for j in range(4):
        acc = torch.zeros_like(p[..., 0])
        for i in range(q):
            acc = acc + p[..., 4 * i + j]
        parts.append(acc)
    for i in range(4 * q, K):
        parts[0] = parts[0] + p[..., i]
    return ((parts[0] + parts[1]) + parts[2]) + parts[3]


def normalise_probs(probs_bchw: Tensor, order: str = "torch") -> Tensor:
    """clamp is applied by the caller; torch.distributions.Categorical divides by the sum over the
    channels-last view (one_hot_categorical.py:25-28 -> torch/distributions/categorical.py).
    Returns channels-last [B,H,W,K] contiguous.

    order="torch":   p.sum(-1) as torch computes it on this host — what the reference does.  For K > 4 that
                     order is NOT one order: ATen's vectorized_outer_sum uses the cascade order for the first
                     multiple-of-(4*Vec) pixels of each chunk and the row_sum order for the rest, so the
                     reference's own last-ulp result depends on pixel position, ISA and thread split.
    order="cascade": the explicit cascade order (== sequential for K <= 16) — what the HIP epilogue
                     implements; bit-identical to "torch" for K <= 4 (LIDC: K = 2)."""
    p = probs_bchw.permute(0, 2, 3, 1)
    if order == "torch":
        return (p / p.sum(-1, keepdim=True)).contiguous()
    pc = p.contiguous()
    return pc / ordered_sum_lastdim(pc)[..., None]


def sample_index(p_hat_nhwk: Tensor, noise_nhwk: Tensor) -> Tensor:
    """multinomial(n=1, replacement=True) == argmax_k p_k / E_k, first index on ties."""
    return torch.argmax(p_hat_nhwk / noise_nhwk, dim=-1)


def draw_exponential(shape, generator: Optional[torch.Generator] = None) -> Tensor:
    """The noise torch.multinomial draws internally: empty(shape).exponential_(1) on the CPU generator."""
    return torch.empty(shape, dtype=torch.float32).exponential_(1, generator=generator)


def draw_x_T(n: int, k: int, h: int, w: int, generator: Optional[torch.Generator] = None):
    """Uniform one-hot x_T as the callers draw it (evaluate_lidc_uncertainty.py:100): logits = 0 ->
    p = 1/K -> argmax (1/K)/E.  Returns (class index [N,H,W] int64, noise used)."""
    e = draw_exponential((n * h * w, k), generator)
    p = torch.full((n * h * w, k), 1.0 / k, dtype=torch.float32)
    p = p / p.sum(-1, keepdim=True)
    idx = torch.argmax(p / e, dim=-1).reshape(n, h, w)
    return idx, e


def one_hot_bchw(idx: Tensor, k: int, dtype=torch.float32) -> Tensor:
    return F.one_hot(idx, k).permute(0, 3, 1, 2).to(dtype)


# ----------------------------------------------------------------------------------------------
# Philox4x32-10 (throughput-mode device RNG) — numpy restatement of the published algorithm
# (Salmon et al., "Parallel random numbers: as easy as 1, 2, 3", SC'11; constants as in Random123).
# ----------------------------------------------------------------------------------------------
PHILOX_M0, PHILOX_M1 = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57)
PHILOX_W0, PHILOX_W1 = np.uint32(0x9E3779B9), np.uint32(0xBB67AE85)


def philox4x32_10(ctr: np.ndarray, key: np.ndarray) -> np.ndarray:
    """ctr [...,4] uint32, key [...,2] uint32 -> [...,4] uint32."""
    c = [ctr[..., i].astype(np.uint32) for i in range(4)]
    k0 = np.broadcast_to(key[..., 0].astype(np.uint32), c[0].shape).copy()
    k1 = np.broadcast_to(key[..., 1].astype(np.uint32), c[0].shape).copy()
    with np.errstate(over="ignore"):
        for _ in range(10):
            p0 = c[0].astype(np.uint64) * PHILOX_M0
            p1 = c[2].astype(np.uint64) * PHILOX_M1
            hi0, lo0 = (p0 >> np.uint64(32)).astype(np.uint32), p0.astype(np.uint32)
            hi1, lo1 = (p1 >> np.uint64(32)).astype(np.uint32), p1.astype(np.uint32)
            c = [hi1 ^ c[1] ^ k0, lo1, hi0 ^ c[3] ^ k1, lo0]
            k0 = k0 + PHILOX_W0
            k1 = k1 + PHILOX_W1
    return np.stack(c, axis=-1)


def philox_exponential(seed: int, step: int, sample0: int, n: int, hw: int, k: int) -> np.ndarray:
    """Exp(1) noise [n, hw, k] fp32 exactly as the device sampler generates it:
    counter = (pixel, global_sample, step, k // 4), key = (seed_lo, seed_hi); word k % 4 of the block;
    U = (bits >> 8 + 0.5) * 2^-24 in (0,1);  E = -log(U) evaluated in float32."""
    pix = np.arange(hw, dtype=np.uint32)[None, :, None]
    smp = (np.arange(n, dtype=np.uint32) + np.uint32(sample0))[:, None, None]
    kq = (np.arange(k, dtype=np.uint32) // 4)[None, None, :]
    ctr = np.stack(np.broadcast_arrays(pix, smp, np.uint32(step), kq), axis=-1).astype(np.uint32)
    key = np.array([seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF], dtype=np.uint32)
    blk = philox4x32_10(ctr, key)
    word = np.take_along_axis(blk, (np.arange(k) % 4)[None, None, :, None].astype(np.int64)
                              * np.ones((n, hw, k, 1), dtype=np.int64), axis=-1)[..., 0]
    u = ((word >> np.uint32(8)).astype(np.float32) + np.float32(0.5)) * np.float32(2.0 ** -24)
    return (-np.log(u.astype(np.float32))).astype(np.float32)


# ----------------------------------------------------------------------------------------------
# A5  the T-step loop                         ddpm/models/diffusion_denoising.py:164-215
# ----------------------------------------------------------------------------------------------
def forward_denoising(sd: Dict[str, Tensor], cfg: dict, schedule, x: Tensor, condition: Tensor,
                      feature_condition: Optional[Tensor] = None, init_t: Optional[int] = None,
                      step_T_sample: Optional[str] = "majority",
                      generator: Optional[torch.Generator] = None,
                      noise: Optional[Sequence[Tensor]] = None,
                      teacher: Optional[Sequence[Tensor]] = None,
                      trace: Optional[list] = None) -> Dict[str, Tensor]:
    """x: one-hot [N,K,H,W] fp32.  `noise[j]` ([N*H*W,K] fp32) overrides the generator draw of step j;
    `teacher[j]` (class index [N,H,W]) overrides the x_t fed to step j (teacher forcing).
    `trace` collects per-step dicts (t, x0pred, probs, idx)."""
    betas, alphas, cumalphas = schedule
    T = len(betas)
    xt = x
    N, K, H, W = x.shape
    for j, t in enumerate(step_values(T, init_t)):
        if teacher is not None:
            xt = one_hot_bchw(teacher[j], K)
        t_ = torch.full((N,), t)
        x0pred = unet_forward(sd, cfg, xt, condition, feature_condition, t_.float())["diffusion_out"]   # :194
        a, c = posterior_coeffs(alphas, cumalphas, t)
        probs = theta_post_prob_ref(xt, x0pred, a, c)                                                     # :197
        probs = torch.clamp(probs, min=1e-12)                                                             # :204
        p_hat = normalise_probs(probs)
        rec = {"t": t, "x0pred": x0pred, "p_hat": p_hat}
        if t > 1:                                                                                         # :206-207
            e = noise[j].reshape(N, H, W, K) if noise is not None else \
                draw_exponential((N * H * W, K), generator).reshape(N, H, W, K)
            idx = sample_index(p_hat, e)
            xt = one_hot_bchw(idx, K)
            rec["idx"] = idx
            rec["noise"] = e
        else:                                                                                             # :208-212
            if step_T_sample is None or step_T_sample == "majority":
                xt = F.one_hot(p_hat.argmax(dim=-1), K).permute(0, 3, 1, 2)     # int64 one-hot
            elif step_T_sample == "confidence":
                xt = p_hat.permute(0, 3, 1, 2)
        if trace is not None:
            trace.append(rec)
    return {"diffusion_out": xt}


# ----------------------------------------------------------------------------------------------
# N1  LIDC metrics (numpy restatement)        evaluation/evaluate_lidc_uncertainty.py:27-73
# ----------------------------------------------------------------------------------------------
def metrics_iou(x, y, axis=-1):
    with np.errstate(divide="ignore", invalid="ignore"):
        iou_ = (x & y).sum(axis) / (x | y).sum(axis)
    iou_[np.isnan(iou_)] = 1.
    return iou_


def metrics_batched_distance(x, y):
    per_class_iou = metrics_iou(x[:, :, None], y[:, None, :], axis=-2)          # exclude background (class 0)
    return 1 - per_class_iou[..., 1:].mean(-1)


def metrics_ged(samples_dist_0, samples_dist_1, num_classes):
    a = samples_dist_0.reshape(*samples_dist_0.shape[:2], -1)
    b = samples_dist_1.reshape(*samples_dist_1.shape[:2], -1)
    eye = np.eye(num_classes)
    a, b = eye[a].astype(bool), eye[b].astype(bool)
    cross = np.mean(metrics_batched_distance(a, b), axis=(1, 2))
    d0 = np.mean(metrics_batched_distance(a, a), axis=(1, 2))
    d1 = np.mean(metrics_batched_distance(b, b), axis=(1, 2))
    return 2 * cross - d0 - d1, d0, d1


def metrics_hungarian_iou(samples_dist_0, samples_dist_1, num_classes):
    from scipy.optimize import linear_sum_assignment
    a = samples_dist_0.reshape(*samples_dist_0.shape[:2], -1)
    b = samples_dist_1.reshape(*samples_dist_1.shape[:2], -1)
    eye = np.eye(num_classes)
    cost = metrics_batched_distance(eye[a].astype(bool), eye[b].astype(bool))
    return [(1 - cost[i])[linear_sum_assignment(cost[i])].mean() for i in range(a.shape[0])]
