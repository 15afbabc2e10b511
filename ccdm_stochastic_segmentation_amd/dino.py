"""DINO ViT-S/8 key-feature extractor on the MI355X.

Mirrors the reference's `ViTExtractor.extract_descriptors` / `DinoViT` (ddpm/models/dino.py:211-229,279-305,
condition_encoder.py:24-46) for what the Cityscapes configs use: `dino_vits8`, stride = patch size = 8, layer-11 *keys*,
no class token, resize to (H // 8, W // 8).  The output is the `feature_condition` tensor of `DenoisingModel.forward`.

Every linear layer runs on the fused conv kernel as a 1x1 conv over a [N, T_alloc/16, 16, C] token image (16 wide: 32-channel chunks,
so the 1536-channel fc2 input stays within the kernel's 64 chunks) (split-fp16 x3 MFMA,
~2^-22 per product); LayerNorm, GELU and the D=64 attention are HIP kernels of their own (ccdm_layernorm, ccdm_gelu,
ccdm_attention_ex).  torch only moves memory: patch unfolding, token padding, the final re-layout of the keys.

PARITY UNPINNED: the network definition and weights come from `torch.hub.load('facebookresearch/dino:main', ...)` in the
reference (dino.py:58-82) — third-party, not in /root/reference, no network here.  The state_dict layout below is that
repository's published VisionTransformer; the test tree carries a CPU restatement of its forward, and the tests compare against that on
synthetic weights.  There is no hub download: pass `state_dict=` (e.g. `torch.load('dino_deitsmall8_pretrain.pth')`).
"""
import ctypes as C
import math
from typing import Dict, Optional, Union

import numpy as np
import torch

from . import hip

VIT_CONFIGS = {"dino_vits8": dict(dim=384, depth=12, heads=6, mlp_ratio=4, patch=8, pretrain_size=224)}
LN_EPS = 1e-6
TOKW = 16          # width of the token image the linear layers see


def vit_param_shapes(model_type: str = "dino_vits8") -> Dict[str, tuple]:
    """state_dict keys and shapes of facebookresearch/dino's VisionTransformer (vision_transformer.py), in order."""
    c = VIT_CONFIGS[model_type]
    d, p = c["dim"], c["patch"]
    n0 = (c["pretrain_size"] // p) ** 2
    out = {"cls_token": (1, 1, d), "pos_embed": (1, n0 + 1, d), "patch_embed.proj.weight": (d, 3, p, p), "patch_embed.proj.bias": (d,)}
    for i in range(c["depth"]):
        b = f"blocks.{i}."
        out.update({b + "norm1.weight": (d,), b + "norm1.bias": (d,), b + "attn.qkv.weight": (3 * d, d), b + "attn.qkv.bias": (3 * d,),
                    b + "attn.proj.weight": (d, d), b + "attn.proj.bias": (d,), b + "norm2.weight": (d,), b + "norm2.bias": (d,),
                    b + "mlp.fc1.weight": (c["mlp_ratio"] * d, d), b + "mlp.fc1.bias": (c["mlp_ratio"] * d,),
                    b + "mlp.fc2.weight": (d, c["mlp_ratio"] * d), b + "mlp.fc2.bias": (d,)})
    out.update({"norm.weight": (d,), "norm.bias": (d,)})
    return out


def make_synthetic_vit_state_dict(model_type: str = "dino_vits8", seed: int = 0) -> Dict[str, np.ndarray]:
    """Random weights of the right shapes (tests, benchmarks): N(0, 1/fan_in) matrices, small biases, LayerNorm gains near 1."""
    r = np.random.default_rng(seed)
    sd = {}
    for k, shp in vit_param_shapes(model_type).items():
        if k.endswith("norm1.weight") or k.endswith("norm2.weight") or k == "norm.weight":
            v = 1.0 + 0.1 * r.standard_normal(shp)
        elif k.endswith(".bias"):
            v = 0.05 * r.standard_normal(shp)
        elif k in ("cls_token", "pos_embed"):
            v = 0.2 * r.standard_normal(shp)
        else:
            v = r.standard_normal(shp) / math.sqrt(int(np.prod(shp[1:])))
        sd[k] = v.astype(np.float32)
    return sd


class ViTExtractor:
    """`extract_descriptors(batch)` of the reference's class of this name, for facet 'key' at stride = patch size."""

    def __init__(self, model_type: str = "dino_vits8", stride: int = 8, model=None, device: str = "cuda",
                 state_dict: Optional[Dict[str, Union[np.ndarray, torch.Tensor]]] = None):
        if model_type not in VIT_CONFIGS:
            raise NotImplementedError(f"ViT type {model_type!r} is not built (the reference's configs use 'dino_vits8')")
        self.cfg = VIT_CONFIGS[model_type]
        self.p = self.cfg["patch"]
        if stride != self.p:
            raise NotImplementedError("only stride == patch size (8) is built: the overlapping-patch variant is not used by the reference's configs")
        self.stride = (stride, stride)
        self.device = torch.device(device if device != "cuda" else "cuda:0")
        if self.device.type != "cuda":
            raise hip.CcdmHipError("ViTExtractor needs a GPU device (the HIP kernels are the only implementation)")
        if model is not None and state_dict is None:
            state_dict = model.state_dict()
        if state_dict is None:
            raise hip.CcdmHipError("no weights: torch.hub is not reachable from here — pass state_dict= (the dino_vits8 checkpoint's tensors)")
        self.lib = hip.load()
        self._pos_cache = {}
        self.load_state_dict(state_dict)
        self.load_size = None
        self.num_patches = None

    # ------------------------------------------------------------------ weights
    def load_state_dict(self, sd) -> None:
        want = vit_param_shapes("dino_vits8")
        sd = {k: (v.detach().cpu().numpy() if isinstance(v, torch.Tensor) else np.asarray(v)).astype(np.float32) for k, v in sd.items() if k in want}
        missing = [k for k in want if k not in sd]
        if missing:
            raise KeyError(f"ViT state_dict misses {missing[:4]}{'...' if len(missing) > 4 else ''}")
        for k, shp in want.items():
            if tuple(sd[k].shape) != tuple(shp):
                raise ValueError(f"{k}: shape {tuple(sd[k].shape)} != {tuple(shp)}")
        dev = self.device
        d = self.cfg["dim"]

        def lin(w, b):   # [out, in] -> packed 1x1 conv weight + bias on the device
            pk = hip.pack_conv_weight(np.ascontiguousarray(w.reshape(w.shape[0], -1, 1, 1)), 1, hip.PREC_F16X3)
            return torch.from_numpy(pk).to(dev), torch.from_numpy(np.ascontiguousarray(b)).to(dev), int(w.shape[0]), int(np.prod(w.shape[1:]))
        self.w_patch = lin(sd["patch_embed.proj.weight"].reshape(d, -1), sd["patch_embed.proj.bias"])
        self.blocks = []
        for i in range(self.cfg["depth"]):
            b = f"blocks.{i}."
            g = lambda k: torch.from_numpy(np.ascontiguousarray(sd[b + k])).to(dev)
            self.blocks.append(dict(n1=(g("norm1.weight"), g("norm1.bias")), qkv=lin(sd[b + "attn.qkv.weight"], sd[b + "attn.qkv.bias"]),
                                    proj=lin(sd[b + "attn.proj.weight"], sd[b + "attn.proj.bias"]), n2=(g("norm2.weight"), g("norm2.bias")),
                                    fc1=lin(sd[b + "mlp.fc1.weight"], sd[b + "mlp.fc1.bias"]), fc2=lin(sd[b + "mlp.fc2.weight"], sd[b + "mlp.fc2.bias"])))
        self.cls_token = torch.from_numpy(sd["cls_token"]).reshape(1, d)
        self.pos_embed = torch.from_numpy(sd["pos_embed"])
        self.patch_bias = torch.from_numpy(sd["patch_embed.proj.bias"])
        self._pos_cache = {}

    def _pos_for(self, H: int, W: int) -> torch.Tensor:
        """Position embedding for an H x W image: class row + the pretrain grid resized bicubically (weight preprocessing, once per
        image size, like the weight packing; vision_transformer.py interpolate_pos_encoding incl. its '+0.1' scale factors)."""
        key = (H, W)
        if key not in self._pos_cache:
            n0 = self.pos_embed.shape[1] - 1
            hp, wp = H // self.p, W // self.p
            if hp * wp == n0 and H == W:
                pe = self.pos_embed
            else:
                g = int(math.sqrt(n0))
                grid = self.pos_embed[:, 1:].reshape(1, g, g, -1).permute(0, 3, 1, 2)
                grid = torch.nn.functional.interpolate(grid, scale_factor=((hp + 0.1) / g, (wp + 0.1) / g), mode="bicubic")
                if grid.shape[-2] != hp or grid.shape[-1] != wp:
                    raise ValueError(f"position grid {tuple(grid.shape[-2:])} != {(hp, wp)}")
                pe = torch.cat([self.pos_embed[:, :1], grid.permute(0, 2, 3, 1).reshape(1, hp * wp, -1)], 1)
            self._pos_cache[key] = pe[0].contiguous()
        return self._pos_cache[key]

    # ------------------------------------------------------------------ kernels
    def _stream(self):
        return torch.cuda.current_stream(self.device).cuda_stream

    def _linear(self, x: torch.Tensor, w, resid: Optional[torch.Tensor] = None, out: Optional[torch.Tensor] = None) -> torch.Tensor:
        """x: [N, R, TOKW, Cin] token image -> [N, R, TOKW, Cout] = x @ W^T + b (+ resid), one ccdm_conv2d launch."""
        wdev, bdev, cout, cin = w
        N, R, Wd, Cin = x.shape
        assert Cin == cin, (Cin, cin)
        if out is None:
            out = torch.empty((N, R, Wd, cout), device=self.device, dtype=torch.float32)
        a = hip.ConvArgs()
        a.in0, a.C0 = x.data_ptr(), Cin
        a.eps, a.act = 1e-5, hip.ACT_NONE
        a.N, a.Hin, a.Win, a.Hout, a.Wout = N, R, Wd, R, Wd
        a.ksize, a.stride, a.up = 1, 1, 0
        a.w, a.bias, a.Cout, a.prec = wdev.data_ptr(), bdev.data_ptr(), cout, hip.PREC_F16X3
        a.emb_off = -1
        if resid is not None:
            a.resid = resid.data_ptr()
        a.out = out.data_ptr()
        hip.check(self.lib.ccdm_conv2d(C.byref(a), self._stream()), "vit linear")
        return out

    def _layernorm(self, x: torch.Tensor, gb, out: Optional[torch.Tensor] = None) -> torch.Tensor:
        out = torch.empty_like(x) if out is None else out
        Cc = x.shape[-1]
        hip.check(self.lib.ccdm_layernorm(x.data_ptr(), gb[0].data_ptr(), gb[1].data_ptr(), LN_EPS, x.numel() // Cc, Cc, out.data_ptr(), self._stream()), "layernorm")
        return out

    def _gelu(self, x: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
        out = torch.empty_like(x) if out is None else out
        hip.check(self.lib.ccdm_gelu(x.data_ptr(), x.numel(), out.data_ptr(), self._stream()), "gelu")
        return out

    def _attention(self, qkv: torch.Tensor, T: int, out: Optional[torch.Tensor] = None) -> torch.Tensor:
        N, R, Wd, C3 = qkv.shape
        Cc = C3 // 3
        if out is None:
            out = torch.zeros((N, R, Wd, Cc), device=self.device, dtype=torch.float32)     # padding rows stay 0
        hip.check(self.lib.ccdm_attention_ex(qkv.data_ptr(), out.data_ptr(), N, T, R * Wd, Cc, self.cfg["heads"], 1, self._stream()), "vit attention")
        return out

    def _workspace(self, N: int, Ta: int, H: int, W: int) -> Dict[str, torch.Tensor]:
        key = (N, H, W)
        if getattr(self, "_ws_key", None) != key:
            d, R, p = self.cfg["dim"], Ta // TOKW, self.p
            T = 1 + (H // p) * (W // p)
            mk = lambda c: torch.zeros((N, R, TOKW, c), device=self.device, dtype=torch.float32)
            # what is added after the patch projection: position embedding; the class row gets cls + pos[0] - bias (its "projection" is the bias)
            pos = self._pos_for(H, W)
            add = torch.zeros((Ta, d), dtype=torch.float32)
            add[:T] = pos
            add[0] = self.cls_token[0] + pos[0] - self.patch_bias
            self._ws = dict(xa=mk(d), xb=mk(d), ln=mk(d), qkv=mk(3 * d), att=mk(d), h=mk(self.cfg["mlp_ratio"] * d), g=mk(self.cfg["mlp_ratio"] * d),
                            tok=torch.zeros((N, Ta, 3 * p * p), device=self.device, dtype=torch.float32),
                            add=add.to(self.device).unsqueeze(0).expand(N, Ta, d).contiguous())
            self._ws_key = key
        return self._ws

    # ------------------------------------------------------------------ forward
    def _qkv_of_layer(self, batch: torch.Tensor, layer: int) -> torch.Tensor:
        """qkv(norm1(x)) of block `layer` as [N, T_alloc, 3*dim] (token 0 = class token, rows >= T are padding)."""
        if not batch.is_cuda:
            raise hip.CcdmHipError("ViTExtractor: the batch must live on the GPU")
        N, Cimg, H, W = batch.shape
        p, d = self.p, self.cfg["dim"]
        if Cimg != 3 or H % p or W % p:
            raise ValueError(f"ViTExtractor: need [N,3,H,W] with H, W multiples of {p}, got {tuple(batch.shape)}")
        hp, wp = H // p, W // p
        T = 1 + hp * wp
        Ta = (T + 31) // 32 * 32
        dev = self.device
        ws = self._workspace(N, Ta, H, W)
        # patches in (c, ky, kx) order = the flattened conv weight's, gathered straight into the token buffer; row 0 (class token)
        # and the padding rows stay zero
        tok_in = ws["tok"]
        tok_in[:, 1:T].view(N, hp, wp, 3, p, p).copy_(batch.float().unfold(2, p, p).unfold(3, p, p).permute(0, 2, 3, 1, 4, 5))
        add = ws["add"]
        # (every buffer of this (batch, image size) is allocated once and reused: per-call device allocations of this size make the
        #  caching allocator free and re-map memory, which cost 50-70 ms per call against 12 ms of kernels)
        x = self._linear(tok_in.view(N, Ta // TOKW, TOKW, -1), self.w_patch, resid=add.view(N, Ta // TOKW, TOKW, d), out=ws["xa"])
        spare = ws["xb"]
        for i in range(layer + 1):
            blk = self.blocks[i]
            qkv = self._linear(self._layernorm(x, blk["n1"], out=ws["ln"]), blk["qkv"], out=ws["qkv"])
            if i == layer:
                return qkv.view(N, Ta, 3 * d), T, hp, wp
            x2 = self._linear(self._attention(qkv, T, out=ws["att"]), blk["proj"], resid=x, out=spare)
            h = self._gelu(self._linear(self._layernorm(x2, blk["n2"], out=ws["ln"]), blk["fc1"], out=ws["h"]), out=ws["g"])
            x, spare = self._linear(h, blk["fc2"], resid=x2, out=x), x2          # the residual stream alternates between the two buffers
        raise AssertionError

    def extract_descriptors(self, batch: torch.Tensor, layers: Union[int, list] = 11, facet: str = "key", include_cls: bool = False,
                            resize_shape: Union[tuple, None] = None) -> torch.Tensor:
        """[B, d*heads (d-major), H/8, W/8] keys of block `layers` (dino.py:279-305)."""
        if facet != "key" or include_cls or not isinstance(layers, int):
            raise NotImplementedError("built for facet='key', include_cls=False, one integer layer (what condition_encoder.py:41-44 asks for)")
        if not 0 <= layers < self.cfg["depth"]:
            raise ValueError(f"layer {layers} outside [0, {self.cfg['depth']})")
        qkv, T, hp, wp = self._qkv_of_layer(batch, layers)
        N, d, hd = batch.shape[0], self.cfg["dim"], self.cfg["dim"] // self.cfg["heads"]
        H, W = batch.shape[2:]
        self.load_size = (H, W)
        self.num_patches = (hp, wp)
        target = (H // self.stride[0], W // self.stride[1]) if resize_shape is None else tuple(resize_shape)
        if target != (hp, wp):
            raise NotImplementedError(f"resize to {target} != token grid {(hp, wp)}: only the identity resize of stride 8 is built")
        k = qkv[:, 1:T, d:2 * d].reshape(N, hp, wp, self.cfg["heads"], hd)
        return k.permute(0, 4, 3, 1, 2).reshape(N, d, hp, wp).contiguous()       # channel = d_index * heads + head (dino.py:299)


class DinoViT:
    """`DinoViT(name, train_encoder, conditioning, stride, resize_shape, layers)(x)` of condition_encoder.py:24-46 (inference only)."""

    def __init__(self, name: str, train_encoder: bool = False, conditioning: str = "concat_pixels_concat_features", stride: int = 8,
                 resize_shape: Union[tuple, None] = None, layers: Union[list, int] = 11, state_dict=None, device: str = "cuda"):
        if train_encoder:
            raise NotImplementedError("training the feature encoder is out of scope")
        self.extractor = ViTExtractor(name, stride, state_dict=state_dict, device=device)
        self.stride, self.conditioning, self.layers, self.resize_shape = stride, conditioning, layers, resize_shape

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        return self.extractor.extract_descriptors(x, self.layers, resize_shape=self.resize_shape)

    __call__ = forward
