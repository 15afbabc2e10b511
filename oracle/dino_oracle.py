"""CPU restatement of the DINO ViT-S/8 key-feature extractor.  TEST INFRASTRUCTURE ONLY.

PARITY UNPINNED.  The reference obtains the network with `torch.hub.load('facebookresearch/dino:main', 'dino_vits8')`
(/root/reference/ddpm/models/dino.py:58-82): neither that repository's vision_transformer.py nor the pretrained weights are
part of /root/reference, and there is no network here.  What follows restates the published algorithm of that file
(VisionTransformer: conv patch embedding, class token, bicubically interpolated position embedding, pre-norm blocks with
LayerNorm(eps=1e-6), fused-qkv multi-head attention, GELU MLP) and the reference's own wrapper around it — the key hook
(dino.py:176-183), descriptor re-layout and resize (dino.py:297-305), called from condition_encoder.py:41-44.  No golden
vectors of the reference exist for it; the HIP path is checked against this file on synthetic weights only.
"""
import math
from typing import Dict

import torch
import torch.nn.functional as F

Tensor = torch.Tensor


def interpolate_pos_encoding(pos_embed: Tensor, H: int, W: int, patch: int) -> Tensor:
    """vision_transformer.py VisionTransformer.interpolate_pos_encoding: class row kept, the sqrt(N) x sqrt(N) patch grid resized
    bicubically to (H // patch, W // patch) with the '+0.1' scale-factor trick of the original."""
    n0 = pos_embed.shape[1] - 1
    hp, wp = H // patch, W // patch
    if hp * wp == n0 and H == W:
        return pos_embed
    dim = pos_embed.shape[-1]
    g = int(math.sqrt(n0))
    grid = pos_embed[:, 1:].reshape(1, g, g, dim).permute(0, 3, 1, 2)
    grid = F.interpolate(grid, scale_factor=((hp + 0.1) / g, (wp + 0.1) / g), mode="bicubic")
    assert grid.shape[-2] == hp and grid.shape[-1] == wp
    return torch.cat([pos_embed[:, :1], grid.permute(0, 2, 3, 1).reshape(1, hp * wp, dim)], 1)


def vit_tokens(sd: Dict[str, Tensor], x: Tensor, patch: int) -> Tensor:
    """prepare_tokens: conv patch embedding, class token, position embedding."""
    B, _, H, W = x.shape
    t = F.conv2d(x, sd["patch_embed.proj.weight"], sd["patch_embed.proj.bias"], stride=patch).flatten(2).transpose(1, 2)
    t = torch.cat([sd["cls_token"].expand(B, -1, -1), t], 1)
    return t + interpolate_pos_encoding(sd["pos_embed"], H, W, patch)


def vit_block_qkv(sd: Dict[str, Tensor], i: int, x: Tensor) -> Tensor:
    p = f"blocks.{i}."
    y = F.layer_norm(x, (x.shape[-1],), sd[p + "norm1.weight"], sd[p + "norm1.bias"], 1e-6)
    return F.linear(y, sd[p + "attn.qkv.weight"], sd[p + "attn.qkv.bias"])


def vit_block(sd: Dict[str, Tensor], i: int, x: Tensor, heads: int) -> Tensor:
    """Block.forward: x + attn(norm1(x)); x + mlp(norm2(x))."""
    p = f"blocks.{i}."
    B, T, C = x.shape
    d = C // heads
    qkv = vit_block_qkv(sd, i, x).reshape(B, T, 3, heads, d).permute(2, 0, 3, 1, 4)
    q, k, v = qkv[0], qkv[1], qkv[2]
    attn = ((q @ k.transpose(-2, -1)) * d ** -0.5).softmax(-1)
    a = (attn @ v).transpose(1, 2).reshape(B, T, C)
    x = x + F.linear(a, sd[p + "attn.proj.weight"], sd[p + "attn.proj.bias"])
    y = F.layer_norm(x, (C,), sd[p + "norm2.weight"], sd[p + "norm2.bias"], 1e-6)
    h = F.gelu(F.linear(y, sd[p + "mlp.fc1.weight"], sd[p + "mlp.fc1.bias"]))
    return x + F.linear(h, sd[p + "mlp.fc2.weight"], sd[p + "mlp.fc2.bias"])


def extract_key_descriptors(sd: Dict[str, Tensor], x: Tensor, layer: int = 11, heads: int = 6, patch: int = 8) -> Tensor:
    """ViTExtractor.extract_descriptors(batch, layers=layer, facet='key', include_cls=False) with stride == patch size
    (dino.py:279-305): keys of block `layer` as [B, d*heads (d-major), H/patch, W/patch], then a bilinear resize to
    (H // stride, W // stride) — the identity at this stride."""
    B, _, H, W = x.shape
    t = vit_tokens(sd, x, patch)
    for i in range(layer):
        t = vit_block(sd, i, t, heads)
    C = t.shape[-1]
    k = vit_block_qkv(sd, layer, t).reshape(B, -1, 3, heads, C // heads).permute(2, 0, 3, 1, 4)[1]     # B x h x t x d   (dino.py:181-182)
    k = k[:, :, 1:, :]                                                                                 # drop the class token
    k = k.permute(0, 2, 3, 1).flatten(-2, -1)                                                          # B x t x (d*h)
    k = k.reshape(B, H // patch, W // patch, -1).permute(0, 3, 1, 2)
    return F.interpolate(k, (H // patch, W // patch), mode="bilinear")
