"""Host side of the HIP step executor: turns (UNetSpec, state_dict, batch geometry) into device buffers,
packed weights and the op list of one denoise step, then drives ccdm_engine_run.

torch is used for device memory, streams and H2D copies only; every arithmetic op of the hot path is a
kernel of libccdm_hip.so (no eager/PyTorch fallback exists — `hip.load()` raises without the library).
"""
from __future__ import annotations

import ctypes as C
import math
import os
from dataclasses import dataclass
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch

from . import hip
from .unet_spec import GN_EPS, UNetSpec

_TIMELINE_OP = int(os.environ.get("CCDM_TIMELINE_OP", "-1"))          # tools/timeline_op.py
_NO_SUBPIXEL = bool(int(os.environ.get("CCDM_NO_SUBPIXEL", "0")))     # same-box A/B hook: Upsample convs in the direct (9-tap) form


@dataclass
class DevTensor:
    """An NHWC fp32 activation [N,h,w,C] plus the per-channel partial statistics its producer left."""
    buf: torch.Tensor
    C: int
    h: int
    w: int
    stats: Optional[torch.Tensor] = None
    slices: int = 0

    @property
    def ptr(self) -> int:
        return self.buf.data_ptr()

    @property
    def stats_ptr(self) -> int:
        return self.stats.data_ptr() if self.stats is not None else 0


def timestep_embedding_host(timesteps: torch.Tensor, dim: int, max_period: int = 10000) -> torch.Tensor:
    """Sinusoidal embedding, evaluated with the same torch-CPU ops as the reference
    (/root/reference/ddpm/models/unet_openai/nn.py:103-121) so the table is bit-identical; see
    include/ccdm_hip.h (ccdm_time_table) for why this tiny table is not recomputed on the device."""
    half = dim // 2
    freqs = torch.exp(-math.log(max_period) * torch.arange(start=0, end=half, dtype=torch.float32) / half)
    args = timesteps[:, None].float() * freqs[None]
    emb = torch.cat([torch.cos(args), torch.sin(args)], dim=-1)
    if dim % 2:
        emb = torch.cat([emb, torch.zeros_like(emb[:, :1])], dim=-1)
    return emb


class SamplerEngine:
    """One engine per (weights, N, H, W).  Not thread-safe (like the reference's module)."""

    def __init__(self, spec: UNetSpec, state_dict: Dict[str, torch.Tensor], N: int, H: int, W: int,
                 num_classes: int, img_channels: int, device: torch.device, max_steps: int,
                 feature_shape: Optional[Tuple[int, int, int]] = None, prec: int = hip.PREC_F32, fine_slices: int = 0,
                 f32_layers=frozenset()):
        self.lib = hip.load()
        # conv layers (state_dict prefixes, e.g. "input_blocks.3.0.op") pinned to the exact-fp32 kernels inside an F16X3 engine: the
        # host's per-layer answer to a range overflow (DenoisingModel.on_range_error = "layers")
        self.f32_layers = frozenset(f32_layers)
        self.fine_slices = int(fine_slices)        # latency slicing level (ccdm_conv_args.fine_slices): more, shorter workgroups per sample
        if device.type != "cuda":
            raise hip.CcdmHipError("SamplerEngine needs a HIP device (torch device 'cuda'); there is no CPU path")
        self.spec, self.N, self.H, self.W = spec, int(N), int(H), int(W)
        self.K, self.C_img = int(num_classes), int(img_channels)
        self.device, self.prec = device, prec
        self.max_steps = max(int(max_steps), self.N, 1)
        if spec.in_channels != self.K + self.C_img or spec.out_channels != self.K:
            raise ValueError("spec channels do not match num_classes/img_channels")
        self._keep: List[torch.Tensor] = []          # every device buffer the op list points into
        self._sd = {k: v.detach().to("cpu", torch.float32).contiguous() for k, v in state_dict.items()}
        self.feature_shape = feature_shape
        self._handle = None
        with torch.cuda.device(device):
            # a private non-default stream: HIP graph capture is not allowed on the legacy default stream
            self.stream = torch.cuda.Stream(device=device)
            with torch.cuda.stream(self.stream):
                self._build()
            self.stream.synchronize()

    # ------------------------------------------------------------------ buffers
    def _dev(self, shape, dtype=torch.float32, zero=False) -> torch.Tensor:
        t = (torch.zeros if zero else torch.empty)(shape, dtype=dtype, device=self.device)
        self._keep.append(t)
        return t

    def _upload(self, arr) -> torch.Tensor:
        t = torch.as_tensor(np.ascontiguousarray(arr)) if not isinstance(arr, torch.Tensor) else arr.contiguous()
        d = t.to(self.device)
        self._keep.append(d)
        return d

    def _act(self, C_: int, h: int, w: int, stats: bool, slices: int = 0) -> DevTensor:
        buf = self._dev((self.N, h, w, C_))
        t = DevTensor(buf, C_, h, w)
        if stats:
            t.slices = slices
            t.stats = self._dev((self.N, t.slices, C_, 2), torch.float64)
        return t

    def _fold_stats(self, t: DevTensor) -> None:
        """Large images: the conv leaves one statistics slice per workgroup (up to 384 per sample); a GroupNorm consumer reads at
        most STATS_MAX_SLICES, so fold them right behind the producer (fixed order, independent of N)."""
        if t.stats is None or t.slices <= hip.STATS_MAX_SLICES:
            return
        folded = self._dev((self.N, hip.STATS_FOLD_SLICES, t.C, 2), torch.float64)
        hip.check(self.lib.ccdm_engine_add_stats_fold(self._handle, t.stats.data_ptr(), self.N, t.slices, t.C, hip.STATS_FOLD_SLICES,
                                                      folded.data_ptr()), "engine_add_stats_fold")
        self.op_names.append("stats_fold")
        self.op_info.append(dict(kind="stats_fold", name="stats_fold", io_bytes=0, gn_read_bytes=0, weight_bytes=0, flop=0))
        t.stats, t.slices = folded, hip.STATS_FOLD_SLICES

    # ------------------------------------------------------------------ op emission
    def _conv(self, src: Sequence[DevTensor], wkey: str, cout: int, ksize: int, *, gn: Optional[str] = None,
              act: int = hip.ACT_NONE, stride: int = 1, up: bool = False, emb_off: int = -1,
              film_off: int = -1, resid: Optional[DevTensor] = None, stats: bool = True,
              skip_src: Optional[Sequence[DevTensor]] = None, skip_key: Optional[str] = None) -> DevTensor:
        sd = self._sd
        prec = hip.PREC_F32 if wkey in self.f32_layers else self.prec
        pack_prec = hip.PREC_F16X3 if prec == hip.PREC_F16 else prec        # the single-pass mode reads the hi halves of the F16X3 packing
        a, b = src[0], (src[1] if len(src) > 1 else None)
        cin = a.C + (b.C if b else 0)
        w = sd[wkey + ".weight"].numpy()
        assert w.shape[0] == cout and w.shape[1] <= cin, (wkey, w.shape, cin, cout)
        if w.shape[1] < cin:      # stem: xin is channel-padded to a multiple of 4 with zero channels
            wp = np.zeros((cout, cin) + tuple(w.shape[2:]), np.float32)
            wp[:, : w.shape[1]] = w
            w = wp
        bias_np = sd[wkey + ".bias"].numpy()
        absmax = None
        if skip_src is not None:
            # the 1x1 skip connection rides in the same GEMM: shared F16X3 exponents, biases pre-added
            ws = sd[skip_key + ".weight"].numpy().reshape(cout, -1)
            absmax = np.maximum(np.abs(w.reshape(cout, -1)).max(1), np.abs(ws).max(1)).astype(np.float32)
            skip_w = self._upload(hip.pack_conv_weight(ws.reshape(cout, -1, 1, 1), 1, pack_prec, absmax))
            bias_np = bias_np + sd[skip_key + ".bias"].numpy()
        # Upsample + conv 3x3 runs in sub-pixel form (4 taps of the low-resolution input per output pixel instead of 9) where built
        subpixel = bool(up) and ksize == 3 and stride == 1 and resid is None and skip_src is None and gn is None and emb_off < 0 and \
            not _NO_SUBPIXEL and \
            bool(self.lib.ccdm_upconv_supported(cin, cout, prec))
        wdev = self._upload(hip.pack_upconv_weight(w, prec) if subpixel else hip.pack_conv_weight(w, ksize, pack_prec, absmax))
        bias = self._upload(bias_np)
        hin, win = a.h, a.w
        hc, wc = (2 * hin, 2 * win) if up else (hin, win)
        pad = ksize // 2
        hout = (hc + 2 * pad - ksize) // stride + 1
        wout = (wc + 2 * pad - ksize) // stride + 1
        up_mode = 2 if subpixel else int(bool(up))
        args = hip.ConvArgs()
        args.in0, args.C0 = a.ptr, a.C
        args.in1, args.C1 = (b.ptr, b.C) if b else (0, 0)
        if gn is not None:
            assert a.stats is not None and (b is None or b.stats is not None), f"{wkey}: input has no statistics"
            args.stats0, args.slices0 = a.stats_ptr, a.slices
            args.stats1, args.slices1 = (b.stats_ptr, b.slices) if b else (0, 0)
            args.gamma = self._upload(sd[gn + ".weight"].numpy()).data_ptr()
            args.beta = self._upload(sd[gn + ".bias"].numpy()).data_ptr()
        args.eps, args.act = GN_EPS, act
        args.film, args.film_off = (1, film_off) if film_off >= 0 else (0, 0)
        args.N, args.Hin, args.Win, args.Hout, args.Wout = self.N, hin, win, hout, wout
        args.ksize, args.stride, args.up, args.fine_slices = ksize, stride, up_mode, int(self.fine_slices)
        args.w, args.bias, args.Cout, args.prec = wdev.data_ptr(), bias.data_ptr(), cout, prec
        args.emb_table, args.emb_stride, args.emb_off = self.emb_table.data_ptr(), self.E, emb_off
        args.emb_row_of_sample = self.rowmap.data_ptr()
        args.step_ptr = 0
        args.resid = resid.ptr if resid is not None else 0
        if resid is not None:
            assert (resid.C, resid.h, resid.w) == (cout, hout, wout), wkey
        if skip_src is not None:
            sa, sb = skip_src[0], (skip_src[1] if len(skip_src) > 1 else None)
            assert (sa.h, sa.w) == (hout, wout), wkey
            args.skip0, args.SC0 = sa.ptr, sa.C
            args.skip1, args.SC1 = (sb.ptr, sb.C) if sb else (0, 0)
            args.skip_w = skip_w.data_ptr()
        if len(self.op_names) == _TIMELINE_OP:      # diagnostics: phase stamps of one block of this op (-DCCDM_ABLATION library only);
            args.prec |= 16 << 8                    # set BEFORE the slice query: the bit may change the kernel the layer selects
        # the statistics slices this launch will leave: the library's answer for the fully described layer (the kernel it selects owns the tiling)
        out = self._act(cout, hout, wout, stats, hip.check(self.lib.ccdm_conv_out_slices(C.byref(args)), "conv_out_slices " + wkey))
        args.out = out.ptr
        args.out_stats, args.out_slices = out.stats_ptr, out.slices
        hip.check(self.lib.ccdm_engine_add_conv(self._handle, C.byref(args)), "engine_add_conv " + wkey)
        self.op_names.append(wkey)
        # algorithmic work of this launch per SAMPLE (SURVEY 8d accounting: conv io + one GroupNorm statistics read + weights;
        # a fused 1x1 skip counts as the separate conv it replaces: its input read and its output write + re-read as residual)
        io = 4 * (cin * hin * win + cout * hout * wout)
        flop = 2 * cin * cout * ksize * ksize * hout * wout
        wbytes = 4 * cout * cin * ksize * ksize
        if skip_src is not None:
            sc = sum(t.C for t in skip_src)
            io += 4 * (sc * hout * wout + cout * hout * wout)
            flop += 2 * sc * cout * hout * wout
            wbytes += 4 * cout * sc
        self.op_info.append(dict(kind="conv", name=wkey, cin=cin, cout=cout, k=ksize, hin=hin, win=win, hout=hout, wout=wout,
                                 stride=stride, up=bool(up), subpixel=subpixel, gn=gn is not None, skip=skip_src is not None,
                                 # (the launcher's rule for the core-only skip-chunk instantiation k_conv<...,SKWT>: another kernel symbol)
                                 prec=prec,
                                 skip_wide=bool(skip_src is not None and prec == hip.PREC_F16X3 and ksize == 3 and stride == 1 and not up
                                                and wout >= 32 and hout * wout > 512 and cout <= 32
                                                and all(t.C % 32 == 0 for t in skip_src)),
                                 io_bytes=io, gn_read_bytes=(4 * cin * hin * win if gn is not None else 0), weight_bytes=wbytes, flop=flop,
                                 # bytes the launch really has to move per sample beyond SURVEY 8d's conv-io figure: the identity-residual
                                 # read of a ResBlock's second conv (8d has no term for it; bench.py adds it to must_move, not to `achieved`)
                                 resid_bytes=(4 * cout * hout * wout if resid is not None else 0)))
        self._fold_stats(out)
        return out

    def _stem(self, wkey: str, cout: int) -> DevTensor:
        """input_blocks[0] on the stem kernel (ccdm_stem.hip): the one-hot of x_t is built from the uint8 index while staging, the image
        comes from xin's channels [K, Cs); (tap, channel)-major K axis.  The epilogue then has no one-hot to write (post.xin = NULL)."""
        sd = self._sd
        w = sd[wkey + ".weight"].numpy()
        assert w.shape[0] == cout and w.shape[1] == self.K + self.C_img <= self.Cs == 4, (wkey, w.shape)
        wdev = self._upload(hip.pack_stem_weight(w))
        bias = self._upload(sd[wkey + ".bias"].numpy())
        a = hip.StemArgs()
        a.xt, a.xin, a.Cs, a.K = self.xt.data_ptr(), self.xin.ptr, self.Cs, self.K
        a.w, a.bias = wdev.data_ptr(), bias.data_ptr()
        a.N, a.H, a.W, a.Cout = self.N, self.H, self.W, cout
        out = self._act(cout, self.H, self.W, True, self.lib.ccdm_conv_slices(self.H, self.W, 1, 3))
        a.out, a.out_stats, a.out_slices = out.ptr, out.stats_ptr, out.slices
        hip.check(self.lib.ccdm_engine_add_stem(self._handle, C.byref(a)), "engine_add_stem " + wkey)
        self.op_names.append(wkey)
        cin = self.Cs
        self.op_info.append(dict(kind="conv", name=wkey, cin=cin, cout=cout, k=3, hin=self.H, win=self.W, hout=self.H, wout=self.W,
                                 stride=1, up=False, subpixel=False, gn=False, skip=False, prec=hip.PREC_F16X3, skip_wide=False, stem=True,
                                 io_bytes=4 * (cin * self.H * self.W + cout * self.H * self.W), gn_read_bytes=0,
                                 weight_bytes=4 * cout * cin * 9, flop=2 * cin * cout * 9 * self.H * self.W))
        self.stem_onehot_on_load = True
        self._fold_stats(out)          # (like every other producer: > STATS_MAX_SLICES partials are folded right behind it)
        return out

    def _resample(self, x: DevTensor, mode: int, *, gn: Optional[str] = None, act: int = hip.ACT_NONE, want_act: bool, want_raw: bool,
                  name: str) -> Tuple[Optional[DevTensor], Optional[DevTensor]]:
        """AvgPool2d(2) / nearest x2 of `x`: (R(act(GroupNorm(x))), R(x)) — the two branches of an updown ResBlock (unet.py:243-248)."""
        ho, wo = (x.h // 2, x.w // 2) if mode == hip.RESAMPLE_AVGPOOL2 else (2 * x.h, 2 * x.w)
        oa = self._act(x.C, ho, wo, False) if want_act else None
        orw = self._act(x.C, ho, wo, False) if want_raw else None
        a = hip.ResampleArgs()
        a.in_, a.C = x.ptr, x.C
        if gn is not None:
            assert x.stats is not None, f"{name}: input has no statistics"
            a.stats, a.slices = x.stats_ptr, x.slices
            a.gamma = self._upload(self._sd[gn + ".weight"].numpy()).data_ptr()
            a.beta = self._upload(self._sd[gn + ".bias"].numpy()).data_ptr()
        a.eps, a.act = GN_EPS, act
        a.N, a.Hin, a.Win, a.mode = self.N, x.h, x.w, mode
        a.out_act, a.out_raw = (oa.ptr if oa else 0), (orw.ptr if orw else 0)
        hip.check(self.lib.ccdm_engine_add_resample(self._handle, C.byref(a)), "engine_add_resample " + name)
        self.op_names.append(name)
        n_out = int(want_act) + int(want_raw)
        self.op_info.append(dict(kind="resample", name=name, C=x.C, hin=x.h, win=x.w, hout=ho, wout=wo,
                                 io_bytes=4 * x.C * (x.h * x.w + n_out * ho * wo), gn_read_bytes=0, weight_bytes=0, flop=0))
        return oa, orw

    def _res_updown(self, p: str, l, x: DevTensor) -> DevTensor:
        """ResBlock(down=True / up=True), unet.py:243-248: h = in_conv(R(SiLU(GN(x)))), x = R(x), then as every ResBlock (Cout == Cin, so
        the skip is the identity).  Down: one elementwise pass leaves both pooled tensors and in_conv reads the activated one as is.  Up:
        in_conv upsamples on load behind its fused GroupNorm + SiLU (nearest commutes with both), only the raw branch is materialised."""
        off = self.emb_offsets[p]
        assert l.cin == l.cout, p
        emb_off = -1 if l.film else off
        if l.updown == "down":
            hp, xr = self._resample(x, hip.RESAMPLE_AVGPOOL2, gn=p + ".in_layers.0", act=hip.ACT_SILU, want_act=True, want_raw=True,
                                    name=p + ".avgpool2")
            h = self._conv([hp], p + ".in_layers.2", l.cout, 3, emb_off=emb_off)
        else:
            _, xr = self._resample(x, hip.RESAMPLE_NEAREST_UP2, want_act=False, want_raw=True, name=p + ".nearest_up2")
            h = self._conv([x], p + ".in_layers.2", l.cout, 3, gn=p + ".in_layers.0", act=hip.ACT_SILU, up=True, emb_off=emb_off)
        return self._conv([h], p + ".out_layers.3", l.cout, 3, gn=p + ".out_layers.0", act=hip.ACT_SILU,
                          film_off=off if l.film else -1, resid=xr)

    def _res(self, p: str, l, src: Sequence[DevTensor]) -> DevTensor:
        if l.updown:
            if len(src) != 1:
                raise NotImplementedError(f"{p}: updown ResBlock on a concatenated input")
            return self._res_updown(p, l, src[0])
        off = self.emb_offsets[p]
        if l.film:
            h = self._conv(src, p + ".in_layers.2", l.cout, 3, gn=p + ".in_layers.0", act=hip.ACT_SILU)
        else:
            h = self._conv(src, p + ".in_layers.2", l.cout, 3, gn=p + ".in_layers.0", act=hip.ACT_SILU, emb_off=off)
        if l.has_skip_conv:
            # skip_connection (1x1 conv of the block input) is fused into the second conv as extra K-segments
            return self._conv([h], p + ".out_layers.3", l.cout, 3, gn=p + ".out_layers.0", act=hip.ACT_SILU,
                              film_off=off if l.film else -1, skip_src=src, skip_key=p + ".skip_connection")
        if len(src) != 1:
            raise NotImplementedError(f"{p}: identity skip on a concatenated input")
        return self._conv([h], p + ".out_layers.3", l.cout, 3, gn=p + ".out_layers.0", act=hip.ACT_SILU,
                          film_off=off if l.film else -1, resid=src[0])

    def _attn(self, p: str, l, x: DevTensor) -> DevTensor:
        T_ = x.h * x.w
        # pinned to exact fp32 by the range fallback: "<block>.qkv" = the qkv conv, "<block>.attention" = the core (its own fp16 split
        # stages q, k, v: the vector-pipe kernel computes the same softmax(q k^T) v in plain fp32 FMAs for head widths it is built for)
        # An exact-fp32 engine (validation mode, the range fallback's diagnosing re-run) takes that kernel too, up to 2048 tokens — its
        # cost grows with T^2 on the vector pipe; beyond, the core keeps the matrix kernel and its operand range.
        core_f32 = (p + ".attention") in self.f32_layers or (self.prec == hip.PREC_F32 and T_ <= 2048)
        fused = (self.prec == hip.PREC_F16X3 and (p + ".qkv") not in self.f32_layers and not core_f32 and x.stats is not None
                 and not os.environ.get("CCDM_NO_ATTN_BLOCK")
                 and self.lib.ccdm_norm_qkv_attention_supported(T_, l.ch, l.heads))
        if fused:
            # GroupNorm + qkv + attention core in one launch (low-resolution stages): the 3C-wide qkv tensor stays on chip
            sd = self._sd
            wq, bq = hip.pack_qkv_weights(sd[p + ".qkv.weight"].numpy(), sd[p + ".qkv.bias"].numpy(), l.heads, bool(l.new_order))
            att = self._act(l.ch, x.h, x.w, False)
            a = hip.AttnBlockArgs()
            a.x, a.stats, a.slices = x.ptr, x.stats_ptr, x.slices
            a.gamma = self._upload(sd[p + ".norm.weight"].numpy()).data_ptr()
            a.beta = self._upload(sd[p + ".norm.bias"].numpy()).data_ptr()
            a.eps = GN_EPS
            a.wqkv, a.bqkv = self._upload(wq).data_ptr(), self._upload(bq).data_ptr()
            a.out = att.ptr
            a.N, a.T, a.C, a.heads = self.N, T_, l.ch, l.heads
            hip.check(self.lib.ccdm_engine_add_norm_qkv_attention(self._handle, C.byref(a)), "engine_add_norm_qkv_attention " + p)
            self.op_names.append(p + ".norm_qkv_attention")
            C_ = l.ch
            self.op_info.append(dict(kind="norm_qkv_attention", name=p + ".norm_qkv_attention", T=T_, C=C_, heads=l.heads,
                                     io_bytes=4 * 8 * C_ * T_, gn_read_bytes=4 * C_ * T_, weight_bytes=4 * 3 * C_ * C_,
                                     flop=2 * T_ * C_ * 3 * C_ + 4 * T_ * T_ * C_))
            return self._conv([att], p + ".proj_out", l.ch, 1, resid=x)
        qkv = self._conv([x], p + ".qkv", 3 * l.ch, 1, gn=p + ".norm", act=hip.ACT_NONE, stats=False)
        a = self._act(l.ch, x.h, x.w, False)
        order = 1 if l.new_order else 0
        if core_f32 and (l.ch // l.heads) in hip.ATTENTION_VALU_WIDTHS:
            order |= hip.ATTENTION_FORCE_VALU
        hip.check(self.lib.ccdm_engine_add_attention(self._handle, qkv.ptr, a.ptr, self.N, x.h * x.w, l.ch, l.heads, order),
                  "engine_add_attention")
        self.op_names.append(p + ".attention")
        self.op_info.append(dict(kind="attention", name=p + ".attention", T=T_, C=l.ch, heads=l.heads, io_bytes=4 * 4 * l.ch * T_,
                                 gn_read_bytes=0, weight_bytes=0, flop=4 * T_ * T_ * l.ch))
        return self._conv([a], p + ".proj_out", l.ch, 1, resid=x)

    def _layers(self, layers, src: Sequence[DevTensor]) -> DevTensor:
        h: Optional[DevTensor] = None
        for l in layers:
            cur = src if h is None else [h]
            if l.kind == "conv":
                if (cur[0] is self.xin and len(cur) == 1 and self.prec == hip.PREC_F16X3 and l.name not in self.f32_layers and not self.fine_slices
                        and self.lib.ccdm_stem_conv_supported(self.Cs, l.cout, self.H, self.W, hip.PREC_F16X3)):
                    h = self._stem(l.name, l.cout)
                else:
                    h = self._conv(cur, l.name, l.cout, 3)
            elif l.kind == "res":
                h = self._res(l.name, l, cur)
            elif l.kind == "attn":
                h = self._attn(l.name, l, cur[0])
            elif l.kind == "down":
                h = self._conv(cur, l.name + ".op", l.cout, 3, stride=2)
            elif l.kind == "up":
                h = self._conv(cur, l.name + ".conv", l.cout, 3, up=True)
            else:  # pragma: no cover
                raise AssertionError(l.kind)
        return h

    # ------------------------------------------------------------------ build
    def _build(self) -> None:
        spec, sd, N, H, W, K = self.spec, self._sd, self.N, self.H, self.W, self.K
        lib = self.lib
        missing = [k for k in spec.param_shapes() if k not in sd]
        if missing:
            raise KeyError(f"state_dict is missing {len(missing)} keys, e.g. {missing[:3]}")
        self.step = self._dev((1,), torch.int32, zero=True)
        self.rowmap = self._dev((N,), torch.int32, zero=True)
        self._handle = lib.ccdm_engine_create(self.step.data_ptr())
        if not self._handle:
            raise hip.CcdmHipError("engine_create: " + hip.last_error())
        self.op_names: List[str] = []
        self.op_info: List[dict] = []
        self.stem_onehot_on_load = False

        # --- time-conditioning parameters: every ResBlock's emb_layers.1 concatenated -----------------
        ted, mc = spec.time_embed_dim, spec.model_channels
        self.emb_offsets: Dict[str, int] = {}
        ws, bs, off = [], [], 0
        for name, l in spec.all_layers():
            if l.kind == "res":
                self.emb_offsets[name] = off
                ws.append(sd[name + ".emb_layers.1.weight"].numpy())
                bs.append(sd[name + ".emb_layers.1.bias"].numpy())
                off += ws[-1].shape[0]
        self.E = off
        self.wcat = self._upload(np.concatenate(ws, 0))
        self.bcat = self._upload(np.concatenate(bs, 0))
        self.te = [self._upload(sd[k].numpy()) for k in
                   ("time_embed.0.weight", "time_embed.0.bias", "time_embed.2.weight", "time_embed.2.bias")]
        S = self.max_steps
        self.sinus = self._dev((S, mc))
        self.emb_table = self._dev((S, self.E))
        self.step_table = self._dev((S, 4), zero=True)

        # --- boundary buffers -------------------------------------------------------------------------
        self.Cs = (K + self.C_img + 3) // 4 * 4
        self.xt = self._dev((N, H * W), torch.uint8, zero=True)
        self.xin = DevTensor(self._dev((N, H, W, self.Cs), zero=True), self.Cs, H, W)
        self.feat: Optional[DevTensor] = None
        if spec.feature_condition_idx:
            if self.feature_shape is None:
                raise ValueError("this model concatenates feature conditioning; feature_shape=(C,h,w) is required")
            fc, fh, fw = self.feature_shape
            if fc != spec.feature_channels:
                raise ValueError(f"feature_condition has {fc} channels, model expects {spec.feature_channels}")
            self.feat = DevTensor(self._dev((N, fh, fw, fc)), fc, fh, fw,
                                  self._dev((N, 1, fc, 2), torch.float64), 1)
        self.out_probs = self._dev((N, H, W, K))
        self.out_onehot = self._dev((N, H, W, K), torch.int64)

        # --- the op list of one denoise step (UNetModel.forward, unet.py:744-808) ----------------------
        hs: List[DevTensor] = []
        h = self.xin
        for i, blk in enumerate(spec.input_blocks):
            src = [h]
            if i in spec.feature_condition_idx:
                if (self.feat.h, self.feat.w) != (h.h, h.w):
                    raise ValueError(f"feature_condition is {self.feat.h}x{self.feat.w}, U-Net stage is {h.h}x{h.w}")
                src = [h, self.feat]
            h = self._layers(blk, src)
            hs.append(h)
        h = self._layers(spec.middle_block, [h])
        for blk in spec.output_blocks:
            h = self._layers(blk, [h, hs.pop()])
        # head conv: K output channels padded to a multiple of 4 (zero weights) so it takes the float4 epilogue
        Kp = (K + 3) // 4 * 4
        if Kp != K:
            wk, bk = self._sd["out.2.weight"], self._sd["out.2.bias"]
            self._sd["out.2.weight"] = torch.cat([wk, torch.zeros((Kp - K,) + tuple(wk.shape[1:]))], 0)
            self._sd["out.2.bias"] = torch.cat([bk, torch.zeros(Kp - K)], 0)
        # few classes on a 32-channel head (LIDC): head conv + step epilogue in ONE launch (ccdm_head.hip) — the logits never reach memory
        self.head_fused = bool(self.prec == hip.PREC_F16X3 and "out.2" not in self.f32_layers and not self.fine_slices and h.stats is not None
                               and lib.ccdm_head_posterior_supported(h.C, K, H, W, hip.PREC_F16X3))
        self._head_src = h
        self.head = None if self.head_fused else self._conv([h], "out.2", Kp, 3, gn="out.0", act=hip.ACT_SILU, stats=False)
        # optional parallel head (unet.py:716-726,805-807): GN -> SiLU -> conv to K-1 logits, no softmax; evaluated on every
        # U-Net call like the reference's forward does
        self.head_ce: Optional[DevTensor] = None
        if spec.ce_head:
            Kc = K - 1
            Kcp = (Kc + 3) // 4 * 4
            if Kcp != Kc:
                wk, bk = self._sd["out_ce.2.weight"], self._sd["out_ce.2.bias"]
                self._sd["out_ce.2.weight"] = torch.cat([wk, torch.zeros((Kcp - Kc,) + tuple(wk.shape[1:]))], 0)
                self._sd["out_ce.2.bias"] = torch.cat([bk, torch.zeros(Kcp - Kc)], 0)
            self.head_ce = self._conv([h], "out_ce.2", Kcp, 3, gn="out_ce.0", act=hip.ACT_SILU, stats=False)
        if self.head_fused:          # (after the optional ce head: the fused op must be the last of the step)
            sdw = self._sd
            a = hip.HeadArgs()
            a.x, a.stats, a.slices = h.ptr, h.stats_ptr, h.slices
            a.gamma = self._upload(sdw["out.0.weight"].numpy()).data_ptr()
            a.beta = self._upload(sdw["out.0.bias"].numpy()).data_ptr()
            a.eps = GN_EPS
            a.w = self._upload(hip.pack_head_weight(sdw["out.2.weight"].numpy()[:K])).data_ptr()
            a.bias = self._upload(sdw["out.2.bias"].numpy()[:K]).data_ptr()
            a.N, a.H, a.W, a.C, a.K = N, H, W, h.C, K
            a.logits_out = 0
            hip.check(lib.ccdm_engine_add_head_posterior(self._handle, C.byref(a)), "engine_add_head_posterior")
            self.op_names.append("out.2")
            self.op_info.append(dict(kind="conv", name="out.2", cin=h.C, cout=Kp, k=3, hin=H, win=W, hout=H, wout=W, stride=1, up=False,
                                     subpixel=False, gn=True, skip=False, prec=hip.PREC_F16X3, skip_wide=False, head_fused=True,
                                     io_bytes=4 * (h.C * H * W + Kp * H * W), gn_read_bytes=4 * h.C * H * W, weight_bytes=4 * Kp * h.C * 9,
                                     flop=2 * h.C * Kp * 9 * H * W))
        self.n_unet_ops = lib.ccdm_engine_num_ops(self._handle)

        post = hip.PostArgs()
        post.head, post.softmax, post.head_stride = (0 if self.head_fused else self.head.ptr), int(spec.softmax_output), (Kp if self.head_fused else self.head.C)
        post.xt, post.N, post.HW, post.K = self.xt.data_ptr(), N, H * W, K
        post.step_table, post.step_ptr = self.step_table.data_ptr(), 0
        post.noise, post.noise_step_stride = 0, 0
        post.philox_seed, post.sample_offset = 0, 0
        post.xt_next = self.xt.data_ptr()
        # (the stem kernel builds the one-hot from xt while staging: nothing reads xin's class channels, the epilogue does not write them)
        post.xin, post.xin_stride = (0 if self.stem_onehot_on_load else self.xin.ptr), self.Cs
        post.out_probs, post.out_onehot, post.posterior_out = self.out_probs.data_ptr(), self.out_onehot.data_ptr(), 0
        self.flag = self._dev((1,), torch.int32, zero=True)      # sticky: a head output was not finite (F16X3 range overflow upstream)
        post.noise_row0, post.range_flag = 0, self.flag.data_ptr()
        self._post = post
        hip.check(lib.ccdm_engine_set_epilogue(self._handle, C.byref(post)), "engine_set_epilogue")
        # the per-run epilogue fields (Philox key, noise block, output pointers) live in a device block the kernel reads, like the step
        # counter: a new key per sampling call or a new host-noise block does not re-capture the step's graph
        self.run_block = self._dev((hip.POST_RUN_BYTES // 8,), torch.int64, zero=True)
        hip.check(lib.ccdm_engine_set_run_block(self._handle, self.run_block.data_ptr()), "engine_set_run_block")
        self._sd = None   # host copies no longer needed
        self._tables_key = None
        self._per_sample_rows = False

    def __del__(self):
        try:
            if self._handle:
                self.lib.ccdm_engine_destroy(self._handle)
                self._handle = None
        except Exception:
            pass

    # ------------------------------------------------------------------ run
    def _stream(self) -> int:
        return self.stream.cuda_stream

    def enter(self):
        """Order the engine's stream after the caller's current stream and make it current."""
        self.stream.wait_stream(torch.cuda.current_stream(self.device))
        return torch.cuda.stream(self.stream)

    def leave(self) -> None:
        """Order the caller's current stream after everything launched on the engine's stream."""
        torch.cuda.current_stream(self.device).wait_stream(self.stream)

    def graph_captures(self) -> int:
        """How often the step's HIP graph has been captured so far (a change of the Philox key, the noise block or an output pointer must
        not add to it: those travel through the device-resident run block)."""
        return int(self.lib.ccdm_engine_num_captures(self._handle))

    def describe_ops(self) -> List[str]:
        out = []
        buf = C.create_string_buffer(256)
        for i in range(self.n_unet_ops):
            self.lib.ccdm_engine_describe_op(self._handle, i, buf, 256)
            out.append(f"{i:3d} {self.op_names[i]}: {buf.value.decode()}")
        return out

    def set_inputs(self, xt_idx: torch.Tensor, cond: torch.Tensor, feat: Optional[torch.Tensor] = None) -> None:
        """xt_idx uint8 [N,H,W]; cond fp32 [N,C_img,H,W] (reference BCHW); feat fp32 [N,Cf,h,w] or None."""
        lib, s, N, HW = self.lib, self._stream(), self.N, self.H * self.W
        assert tuple(xt_idx.shape) == (N, self.H, self.W) and xt_idx.dtype == torch.uint8 and xt_idx.is_cuda
        assert tuple(cond.shape) == (N, self.C_img, self.H, self.W), (tuple(cond.shape), (N, self.C_img, self.H, self.W))
        cond = cond.to(self.device, torch.float32).contiguous()
        self.xt.copy_(xt_idx.reshape(N, HW))
        hip.check(lib.ccdm_nchw_to_nhwc(cond.data_ptr(), self.xin.ptr, N, self.C_img, HW, self.Cs, self.K, s), "nchw_to_nhwc")
        hip.check(lib.ccdm_onehot_to_xin(self.xt.data_ptr(), self.xin.ptr, N, HW, self.K, self.Cs, s), "onehot_to_xin")
        if self.feat is not None:
            if feat is None:
                raise ValueError("feature_condition is required by this model")
            f = self.feat
            assert tuple(feat.shape) == (N, f.C, f.h, f.w), tuple(feat.shape)
            feat = feat.to(self.device, torch.float32).contiguous()
            hip.check(lib.ccdm_nchw_to_nhwc(feat.data_ptr(), f.ptr, N, f.C, f.h * f.w, f.C, 0, s), "nchw_to_nhwc(feat)")
            hip.check(lib.ccdm_gn_stats(f.ptr, N, f.h * f.w, f.C, 1, f.stats_ptr, s), "gn_stats(feat)")
        self._cond_keepalive = (cond, feat)

    def set_tables(self, t_rows: Sequence[float], coeffs: Sequence[Tuple[float, float, int]], per_sample: bool = False) -> None:
        """t_rows[i] = timestep of table row i; coeffs[i] = (alpha_t, cumalpha_tm1, mode).
        per_sample=True maps sample n to row n (forward_step with a per-sample t)."""
        S = len(t_rows)
        if S > self.max_steps:
            raise ValueError(f"{S} table rows > max_steps {self.max_steps}")
        key = (tuple(float(t) for t in t_rows), tuple(coeffs), per_sample)
        if key == self._tables_key:
            return
        sin = timestep_embedding_host(torch.tensor(list(t_rows), dtype=torch.float32), self.spec.model_channels)
        self.sinus[:S].copy_(sin)
        tab = torch.zeros((S, 4), dtype=torch.float32)
        for i, (a, c, mode) in enumerate(coeffs):
            tab[i, 0], tab[i, 1], tab[i, 2] = a, c, float(mode)
        self.step_table[:S].copy_(tab)
        self.rowmap.copy_(torch.arange(self.N, dtype=torch.int32) if per_sample else torch.zeros(self.N, dtype=torch.int32))
        self._per_sample_rows = bool(per_sample)
        w0, b0, w2, b2 = (t.data_ptr() for t in self.te)
        hip.check(self.lib.ccdm_time_table(self.sinus.data_ptr(), S, self.spec.model_channels, w0, b0, w2, b2,
                                           self.wcat.data_ptr(), self.bcat.data_ptr(), self.E, 0,
                                           self.emb_table.data_ptr(), self._stream()), "time_table")
        self._tables_key = key

    def run(self, n_steps: int, *, noise: Optional[torch.Tensor] = None, philox_seed: int = 0, sample_offset: int = 0,
            with_epilogue: bool = True, use_graph: bool = False, first_row: int = 0, noise_row0: int = 0,
            posterior_out: Optional[torch.Tensor] = None) -> None:
        """Launch n_steps denoise steps starting at table row `first_row` (asynchronous on the engine's stream).
        noise: [rows, N*H*W*K] host-drawn Exp(1), row r belonging to step row noise_row0 + r."""
        npn = self.N * self.H * self.W * self.K
        if noise is not None:
            assert noise.is_cuda and noise.dtype == torch.float32 and noise.is_contiguous()
            assert first_row >= noise_row0 and noise.numel() >= max(first_row - noise_row0 + n_steps - 1, 1) * npn, "noise tensor too small"
            self._noise_keepalive = noise
        hip.check(self.lib.ccdm_engine_set_run(
            self._handle, noise.data_ptr() if noise is not None else 0, npn, int(noise_row0), int(philox_seed) & (2 ** 64 - 1),
            int(sample_offset), self.out_probs.data_ptr(), self.out_onehot.data_ptr(),
            posterior_out.data_ptr() if posterior_out is not None else 0), "engine_set_run")
        hip.check(self.lib.ccdm_engine_run(self._handle, first_row, n_steps, int(with_epilogue), int(use_graph),
                                           self._stream()), "engine_run")

    def ce_logits(self) -> Optional[torch.Tensor]:
        """[N,K-1,H,W] logits of the optional ce head after the last run (BCHW view of channels-last memory), else None."""
        if self.head_ce is None:
            return None
        return self.head_ce.buf[..., : self.K - 1].clone().permute(0, 3, 1, 2)

    def probe_buffer(self) -> torch.Tensor:
        """A zeroed device buffer for `probe_ranges` (one float per op), created on the engine's stream."""
        with self.enter():
            buf = torch.zeros((self.n_unet_ops,), dtype=torch.float32, device=self.device)
        self.leave()
        return buf

    def probe_ranges(self, buf: torch.Tensor, row: int = -1) -> None:
        """Enqueue the F16X3 range diagnostics of the tensors the last run left behind (no synchronisation): buf[i] = max(buf[i], largest
        |a| op i stages).  `row` = the step-table row those activations were produced with (-1: the last row the last run executed —
        the device counter itself stands one past it, and FiLM's scale / shift are read from that row)."""
        rows = self.N if self._per_sample_rows else 1
        r = int(row)
        if r >= 0 and r + rows > self.max_steps:
            raise ValueError(f"probe_ranges: row {r} (+{rows - 1} per-sample rows) lies beyond the {self.max_steps}-row tables")
        with self.enter():
            hip.check(self.lib.ccdm_engine_input_absmax(self._handle, buf.data_ptr(), r, self._stream()), "engine_input_absmax")
        self.leave()

    def read_ranges(self, buf: torch.Tensor) -> Dict[str, float]:
        """Synchronises; op name -> largest staged |a| (Inf if a value was not finite) for the conv ops and the attention cores."""
        with self.enter():
            vals = buf.cpu().tolist()
        self.leave()
        return {self.op_names[i]: float(vals[i]) for i in range(self.n_unet_ops) if self.op_info[i]["kind"] in ("conv", "attention")}

    def input_absmax(self, row: int = -1) -> Dict[str, float]:
        """F16X3 range diagnostics on the tensors the last run left behind (synchronises): for every conv op its state_dict prefix ->
        the largest |a| it stages (include/ccdm_hip.h: ccdm_conv_input_absmax; Inf if a value is not finite); for an attention core
        ("<block>.attention") the largest |q|, |k|, |v|.  Meaningful on an engine whose values are trustworthy — the exact-fp32 one, or an
        F16X3 one that did not overflow."""
        buf = self.probe_buffer()
        self.probe_ranges(buf, row)
        return self.read_ranges(buf)

    def check_and_clear_flag(self) -> bool:
        """Synchronises with the engine's stream; True (and the flag cleared) if any head output of the runs since the last check was
        not finite.  Read and reset on the engine's own stream: the epilogue that raises the flag runs there."""
        with self.enter():
            hit = int(self.flag.item()) != 0
            if hit:
                self.flag.zero_()
        self.leave()
        return hit

    def raise_range_error(self) -> None:
        raise hip.CcdmRangeError(
            "the network output is not finite" + (": a staged activation left the range of the fp16 split (|a| >= 4094, "
            "include/ccdm_hip.h); re-run with prec=PREC_F32" if self.prec != hip.PREC_F32 else " (exact-fp32 kernels: check the weights and inputs)"))

    def raise_if_flagged(self) -> None:
        """Synchronises with the engine's stream.  Raises CcdmRangeError (and clears the flag) if any head output of the runs
        since the last check was not finite."""
        if self.check_and_clear_flag():
            self.raise_range_error()

    # timing taps for bench.py: HIP events on the engine's stream around every launch of the tapped ops
    def profile_op(self, op_index: int, capacity: int = 4096) -> None:
        """Tap one more op (op_index < 0: remove every tap)."""
        hip.check(self.lib.ccdm_engine_profile_op(self._handle, op_index, capacity), "engine_profile_op")

    def profile_read(self, op_index: int) -> Tuple[int, float, float, float]:
        """(launches timed, mean ms, min ms, max ms) of a tapped op over the series recorded since the last run from row 0."""
        m, lo, hi = C.c_double(), C.c_double(), C.c_double()
        n = hip.check(self.lib.ccdm_engine_profile_read(self._handle, op_index, C.byref(m), C.byref(lo), C.byref(hi)), "profile_read")
        return n, m.value, lo.value, hi.value
