"""Topology of the denoiser U-Net, as a flat, framework-free description.

This is host logic only: it turns the ``unet_openai:`` YAML sub-dict into the list of
blocks the HIP engine executes and into the exact ``state_dict`` key/shape list of the
reference network, so pretrained checkpoints load unchanged.

Follows (behaviour, not code) the reference constructors:
  * channel_mult defaults by image size      /root/reference/ddpm/models/unet_openai/__init__.py:28-38
  * block layout, attention placement, DINO  /root/reference/ddpm/models/unet_openai/unet.py:433-726
  * builder argument mapping                 /root/reference/ddpm/models/builder.py:14-51
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np

GN_GROUPS = 32          # GroupNorm32(32, C)     unet_openai/nn.py:93-100
GN_EPS = 1e-5           # nn.GroupNorm default


@dataclass
class ConvLayer:            # plain 3x3 conv (stem)          unet.py:517
    name: str
    cin: int
    cout: int
    kind: str = "conv"


@dataclass
class ResLayer:             # ResBlock                         unet.py:149-262
    name: str
    cin: int
    cout: int
    film: bool = False      # use_scale_shift_norm
    kind: str = "res"
    updown: str = ""        # "down" / "up": ResBlock(down=True / up=True) of a resblock_updown network   unet.py:202-208,243-248

    @property
    def has_skip_conv(self) -> bool:
        return self.cin != self.cout


@dataclass
class AttnLayer:            # AttentionBlock                   unet.py:265-311
    name: str
    ch: int
    heads: int
    new_order: bool = False  # QKVAttention instead of QKVAttentionLegacy
    kind: str = "attn"


@dataclass
class DownLayer:            # Downsample (conv3x3 stride 2)    unet.py:119-146
    name: str
    ch: int
    cout: int
    kind: str = "down"


@dataclass
class UpLayer:              # Upsample (nearest x2 + conv3x3)  unet.py:87-116
    name: str
    ch: int
    cout: int
    kind: str = "up"


@dataclass
class UNetSpec:
    in_channels: int
    model_channels: int
    out_channels: int
    channel_mult: Tuple[float, ...]
    input_blocks: List[List[object]] = field(default_factory=list)
    middle_block: List[object] = field(default_factory=list)
    output_blocks: List[List[object]] = field(default_factory=list)
    feature_condition_idx: List[int] = field(default_factory=list)   # input-block indices that get `cat([h, feat])`
    feature_channels: int = 0
    # configured injection points whose block was NOT widened (target_layer / output_stride do not line up): the reference
    # builds such a model but its forward fails on the channel mismatch as soon as a feature tensor is passed
    feature_condition_unwired: List[int] = field(default_factory=list)
    softmax_output: bool = True
    ce_head: bool = False
    time_embed_dim: int = 0
    head_in: int = 0          # channels entering `out`

    # ------------------------------------------------------------------ parameters
    def param_shapes(self) -> "Dict[str, Tuple[int, ...]]":
        """state_dict keys -> shapes, in the reference's registration order."""
        out: Dict[str, Tuple[int, ...]] = {}
        mc, ted = self.model_channels, self.time_embed_dim
        out["time_embed.0.weight"] = (ted, mc)
        out["time_embed.0.bias"] = (ted,)
        out["time_embed.2.weight"] = (ted, ted)
        out["time_embed.2.bias"] = (ted,)

        def layer(prefix: str, l) -> None:
            if l.kind == "conv":
                out[f"{prefix}.weight"] = (l.cout, l.cin, 3, 3)
                out[f"{prefix}.bias"] = (l.cout,)
            elif l.kind == "res":
                out[f"{prefix}.in_layers.0.weight"] = (l.cin,)
                out[f"{prefix}.in_layers.0.bias"] = (l.cin,)
                out[f"{prefix}.in_layers.2.weight"] = (l.cout, l.cin, 3, 3)
                out[f"{prefix}.in_layers.2.bias"] = (l.cout,)
                e = 2 * l.cout if l.film else l.cout
                out[f"{prefix}.emb_layers.1.weight"] = (e, ted)
                out[f"{prefix}.emb_layers.1.bias"] = (e,)
                out[f"{prefix}.out_layers.0.weight"] = (l.cout,)
                out[f"{prefix}.out_layers.0.bias"] = (l.cout,)
                out[f"{prefix}.out_layers.3.weight"] = (l.cout, l.cout, 3, 3)
                out[f"{prefix}.out_layers.3.bias"] = (l.cout,)
                if l.has_skip_conv:
                    out[f"{prefix}.skip_connection.weight"] = (l.cout, l.cin, 1, 1)
                    out[f"{prefix}.skip_connection.bias"] = (l.cout,)
            elif l.kind == "attn":
                out[f"{prefix}.norm.weight"] = (l.ch,)
                out[f"{prefix}.norm.bias"] = (l.ch,)
                out[f"{prefix}.qkv.weight"] = (3 * l.ch, l.ch, 1)
                out[f"{prefix}.qkv.bias"] = (3 * l.ch,)
                out[f"{prefix}.proj_out.weight"] = (l.ch, l.ch, 1)
                out[f"{prefix}.proj_out.bias"] = (l.ch,)
            elif l.kind == "down":
                out[f"{prefix}.op.weight"] = (l.cout, l.ch, 3, 3)
                out[f"{prefix}.op.bias"] = (l.cout,)
            elif l.kind == "up":
                out[f"{prefix}.conv.weight"] = (l.cout, l.ch, 3, 3)
                out[f"{prefix}.conv.bias"] = (l.cout,)
            else:  # pragma: no cover
                raise AssertionError(l.kind)

        for i, blk in enumerate(self.input_blocks):
            for j, l in enumerate(blk):
                layer(f"input_blocks.{i}.{j}", l)
        for j, l in enumerate(self.middle_block):
            layer(f"middle_block.{j}", l)
        for i, blk in enumerate(self.output_blocks):
            for j, l in enumerate(blk):
                layer(f"output_blocks.{i}.{j}", l)
        out["out.0.weight"] = (self.head_in,)
        out["out.0.bias"] = (self.head_in,)
        # NB the reference builds the head conv with `input_ch` inputs (unet.py:705), which equals
        # the channel count leaving the decoder (channel_mult[0] * model_channels).
        out["out.2.weight"] = (self.out_channels, int(self.channel_mult[0] * self.model_channels), 3, 3)
        out["out.2.bias"] = (self.out_channels,)
        if self.ce_head:
            out["out_ce.0.weight"] = (self.head_in,)
            out["out_ce.0.bias"] = (self.head_in,)
            out["out_ce.2.weight"] = (self.out_channels - 1, int(self.channel_mult[0] * self.model_channels), 3, 3)
            out["out_ce.2.bias"] = (self.out_channels - 1,)
        return out

    def num_params(self) -> int:
        return int(sum(int(np.prod(s)) for s in self.param_shapes().values()))

    def all_layers(self):
        for i, blk in enumerate(self.input_blocks):
            for j, l in enumerate(blk):
                yield f"input_blocks.{i}.{j}", l
        for j, l in enumerate(self.middle_block):
            yield f"middle_block.{j}", l
        for i, blk in enumerate(self.output_blocks):
            for j, l in enumerate(blk):
                yield f"output_blocks.{i}.{j}", l


def default_channel_mult(image_size: int) -> Tuple[float, ...]:
    """unet_openai/__init__.py:28-38."""
    table = {512: (0.5, 1, 1, 2, 2, 4, 4), 256: (1, 1, 2, 2, 4, 4), 128: (1, 1, 2, 3, 4), 64: (1, 2, 3, 4)}
    if image_size not in table:
        raise ValueError(f"unsupported image size: {image_size}")
    return table[image_size]


def make_unet_spec(
    image_size: int,
    base_channels: int,
    in_channels: int,
    out_channels: int,
    num_res_blocks: int = 2,
    cond_encoded_shape=None,
    channel_mult: Optional[Sequence[float]] = None,
    use_checkpoint: bool = False,
    attention_resolutions: Sequence[int] = (32, 16, 8),
    num_heads: int = 1,
    num_head_channels: int = -1,
    num_heads_upsample: int = -1,
    use_scale_shift_norm: bool = False,
    dropout: float = 0,
    resblock_updown: bool = False,
    use_fp16: bool = False,
    use_new_attention_order: bool = False,
    softmax_output: bool = True,
    ce_head: bool = False,
    feature_cond_encoder: Optional[dict] = None,
) -> UNetSpec:
    """Same keyword surface as the reference's ``create_unet_openai`` (unet_openai/__init__.py:5-61)."""
    if channel_mult is None:
        channel_mult = default_channel_mult(image_size)
    channel_mult = tuple(channel_mult)
    if use_fp16:
        raise NotImplementedError("use_fp16 is unused by the reference configs (fp16_util.py)")
    mc = base_channels
    ted = mc * 4
    if num_heads_upsample == -1:
        num_heads_upsample = num_heads

    # --- feature-condition (DINO) bookkeeping, unet.py:476-494 / :545-550
    fidx: List[int] = []
    fch = 0
    f_stride = None
    if feature_cond_encoder is not None:
        ftype = feature_cond_encoder["type"]
        if ftype == "dino":
            if feature_cond_encoder["scale"] != "single":
                raise NotImplementedError(f"feature_cond_encoder dino with scale {feature_cond_encoder['scale']}")
            tl = feature_cond_encoder["target_layer"]
            fidx = [tl] if tl is not None else []
            fch = int(feature_cond_encoder["channels"])
            f_stride = feature_cond_encoder["output_stride"]
        else:
            raise NotImplementedError(f"feature_cond_encoder type {ftype!r}")

    def heads_for(ch: int, nh: int) -> int:
        if num_head_channels == -1:
            return nh
        if ch % num_head_channels != 0:
            raise AssertionError(
                f"q,k,v channels {ch} is not divisible by num_head_channels {num_head_channels}")
        return ch // num_head_channels

    spec = UNetSpec(in_channels=in_channels, model_channels=mc, out_channels=out_channels,
                    channel_mult=channel_mult, softmax_output=bool(softmax_output), ce_head=bool(ce_head),
                    time_embed_dim=ted)
    ch = input_ch = int(channel_mult[0] * mc)
    spec.input_blocks.append([ConvLayer("input_blocks.0.0", in_channels, ch)])
    chans = [ch]
    ds = 1
    cnt = 1
    widened: List[int] = []
    for level, mult in enumerate(channel_mult):
        for _ in range(num_res_blocks):
            if fidx and cnt in fidx and f_stride == ds:
                ch = ch + fch
                widened.append(cnt)
            cout = int(mult * mc)
            layers: List[object] = [ResLayer(f"input_blocks.{cnt}.0", ch, cout, film=use_scale_shift_norm)]
            ch = cout
            if ds in attention_resolutions:
                # the reference re-binds `num_heads` here, so the value leaks to later blocks
                # (unet.py:567-572); reproduce that.
                if num_head_channels != -1:
                    num_heads = ch // num_head_channels
                layers.append(AttnLayer(f"input_blocks.{cnt}.1", ch, heads_for(ch, num_heads),
                                        new_order=use_new_attention_order))
            spec.input_blocks.append(layers)
            cnt += 1
            chans.append(ch)
        if level != len(channel_mult) - 1:
            # unet.py:586-600: a ResBlock(down=True) with out_channels == channels (identity skip) stands in for Downsample
            spec.input_blocks.append([ResLayer(f"input_blocks.{cnt}.0", ch, ch, film=use_scale_shift_norm, updown="down")
                                      if resblock_updown else DownLayer(f"input_blocks.{cnt}.0", ch, ch)])
            cnt += 1
            chans.append(ch)
            ds *= 2
    if num_head_channels != -1:
        num_heads = ch // num_head_channels
    spec.middle_block = [
        ResLayer("middle_block.0", ch, ch, film=use_scale_shift_norm),
        AttnLayer("middle_block.1", ch, heads_for(ch, num_heads), new_order=use_new_attention_order),
        ResLayer("middle_block.2", ch, ch, film=use_scale_shift_norm),
    ]
    ob = 0
    for level, mult in list(enumerate(channel_mult))[::-1]:
        for i in range(num_res_blocks + 1):
            ich = chans.pop()
            cout = int(mc * mult)
            layers = [ResLayer(f"output_blocks.{ob}.0", ch + ich, cout, film=use_scale_shift_norm)]
            ch = cout
            if ds in attention_resolutions:
                if num_head_channels != -1:
                    num_heads = ch // num_head_channels
                # decoder attention is constructed with num_heads_upsample (unet.py:676-684); with
                # num_head_channels != -1 the block recomputes heads = ch // num_head_channels itself.
                layers.append(AttnLayer(f"output_blocks.{ob}.{len(layers)}", ch, heads_for(ch, num_heads_upsample),
                                        new_order=use_new_attention_order))
            if level and i == num_res_blocks:
                name = f"output_blocks.{ob}.{len(layers)}"
                layers.append(ResLayer(name, ch, ch, film=use_scale_shift_norm, updown="up") if resblock_updown     # unet.py:683-697
                              else UpLayer(name, ch, ch))
                ds //= 2
            spec.output_blocks.append(layers)
            ob += 1
    spec.head_in = ch
    spec.feature_condition_idx = [i for i in fidx if i in widened]
    spec.feature_condition_unwired = [i for i in fidx if i not in widened]
    spec.feature_channels = fch if spec.feature_condition_idx else 0
    for _, l in spec.all_layers():
        c = l.cin if l.kind == "res" else (l.ch if l.kind == "attn" else None)
        if c is not None and c % GN_GROUPS != 0:
            raise ValueError(f"GroupNorm(32, {c}) is invalid: num_channels must be divisible by num_groups")
        if l.kind == "res" and l.cout % GN_GROUPS != 0:
            raise ValueError(f"GroupNorm(32, {l.cout}) is invalid: num_channels must be divisible by num_groups")
    if input_ch != ch:
        # reference would fail at run time in `out` (conv expects input_ch channels); keep the error early
        raise ValueError(f"decoder leaves {ch} channels but the head conv expects {input_ch}")
    return spec


# ---------------------------------------------------------------------------------------------
# Build-owned synthetic weights (no checkpoint is available offline, SURVEY §7 hard part 10).
# numpy-seeded so the GPU box regenerates identical tensors with no torch-RNG coupling.
# ---------------------------------------------------------------------------------------------
def make_synthetic_state_dict(spec: UNetSpec, seed: int = 0, head_gain: float = 2.0) -> "Dict[str, np.ndarray]":
    """Deterministic fp32 weights keyed like the reference ``unet.state_dict()``.

    Convs / linears: N(0, 1/fan_in) so activations stay O(1) through the net (a fresh reference
    model has zeroed `out_layers.3`, `proj_out`, `out.2` and would emit exactly uniform probabilities,
    which makes a useless parity input); biases 0.1*N(0,1); GroupNorm gamma = 1 + 0.1*N, beta = 0.1*N.
    """
    rng = np.random.default_rng(seed)
    sd: Dict[str, np.ndarray] = {}
    for key, shape in spec.param_shapes().items():
        leaf = key.rsplit(".", 1)[1]
        is_norm = len(shape) == 1 and (
            ".in_layers.0." in key or ".out_layers.0." in key or ".norm." in key
            or key.startswith("out.0.") or key.startswith("out_ce.0."))
        if is_norm:
            v = (1.0 + 0.1 * rng.standard_normal(shape)) if leaf == "weight" else 0.1 * rng.standard_normal(shape)
        elif leaf == "bias":
            v = 0.1 * rng.standard_normal(shape)
        else:
            fan_in = int(np.prod(shape[1:]))
            gain = head_gain if key.startswith("out.2.") else 1.0
            v = gain * rng.standard_normal(shape) / np.sqrt(fan_in)
        sd[key] = np.ascontiguousarray(v, dtype=np.float32)
    return sd
