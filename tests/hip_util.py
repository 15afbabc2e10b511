"""Thin test-side wrappers that call the C ABI (include/ccdm_hip.h) on torch CUDA tensors."""
import ctypes as C

import numpy as np
import torch

from ccdm_stochastic_segmentation_amd import hip

DEV = torch.device("cuda:0")


def nhwc(x_bchw: torch.Tensor) -> torch.Tensor:
    return x_bchw.permute(0, 2, 3, 1).contiguous().to(DEV)


def bchw(x_nhwc: torch.Tensor) -> torch.Tensor:
    return x_nhwc.permute(0, 3, 1, 2).contiguous().cpu()


def sync():
    torch.cuda.synchronize()


def gn_stats(x_nhwc: torch.Tensor, slices: int = 1) -> torch.Tensor:
    lib = hip.load()
    N, H, W, Cc = x_nhwc.shape
    st = torch.empty((N, slices, Cc, 2), dtype=torch.float64, device=DEV)
    hip.check(lib.ccdm_gn_stats(x_nhwc.data_ptr(), N, H * W, Cc, slices, st.data_ptr(), 0), "gn_stats")
    return st


def conv2d(srcs, weight, bias, ksize, *, stats=None, gamma=None, beta=None, act=hip.ACT_NONE, stride=1, up=False,
           emb=None, emb_rows=None, film=None, resid=None, want_stats=True, prec=hip.PREC_F32, skip=None, bench=None, fine=False, diag=0):
    """srcs: list of 1-2 NHWC cuda tensors; weight OIHW numpy/torch cpu; stats: list of stats tensors or None.
    emb: [rows, E] cpu tensor added per output channel (row per sample via emb_rows) ; film: (table cpu [rows, 2C]).
    Returns (out NHWC cuda, out_stats or None)."""
    lib = hip.load()
    a = srcs[0]
    b = srcs[1] if len(srcs) > 1 else None
    N, Hin, Win, C0 = a.shape
    C1 = b.shape[3] if b is not None else 0
    w = np.ascontiguousarray(weight, dtype=np.float32)
    cout = w.shape[0]
    absmax = None
    bias = np.asarray(bias, dtype=np.float32)
    conv_prec = prec
    if prec == hip.PREC_F16:        # the opt-in single-pass mode reads the hi halves of the F16X3 packing
        prec = hip.PREC_F16X3
    if skip is not None:            # (list of NHWC tensors, weight [Cout,Cs,1,1], bias): fused 1x1 skip connection
        ssrc, sw, sb = skip
        sw = np.ascontiguousarray(sw, dtype=np.float32).reshape(cout, -1)
        absmax = np.maximum(np.abs(w.reshape(cout, -1)).max(1), np.abs(sw).max(1)).astype(np.float32)
        swdev = torch.from_numpy(hip.pack_conv_weight(sw.reshape(cout, -1, 1, 1), 1, prec, absmax)).to(DEV)
        bias = bias + np.asarray(sb, dtype=np.float32)
    # up: False / True (upsample on load) / 2 (the same operator in sub-pixel form, include/ccdm_hip.h)
    wdev = torch.from_numpy(hip.pack_upconv_weight(w, prec) if up == 2 else hip.pack_conv_weight(w, ksize, prec, absmax)).to(DEV)
    bdev = torch.as_tensor(bias).to(DEV)
    Hc, Wc = (2 * Hin, 2 * Win) if up else (Hin, Win)
    pad = ksize // 2
    Hout, Wout = (Hc + 2 * pad - ksize) // stride + 1, (Wc + 2 * pad - ksize) // stride + 1
    out = torch.empty((N, Hout, Wout, cout), device=DEV)
    keep = [wdev, bdev]
    args = hip.ConvArgs()
    args.in0, args.C0 = a.data_ptr(), C0
    args.in1, args.C1 = (b.data_ptr(), C1) if b is not None else (0, 0)
    if stats is not None:
        args.stats0, args.slices0 = stats[0].data_ptr(), stats[0].shape[1]
        if b is not None:
            args.stats1, args.slices1 = stats[1].data_ptr(), stats[1].shape[1]
        g = torch.as_tensor(np.asarray(gamma, dtype=np.float32)).to(DEV)
        bt = torch.as_tensor(np.asarray(beta, dtype=np.float32)).to(DEV)
        keep += [g, bt]
        args.gamma, args.beta = g.data_ptr(), bt.data_ptr()
    args.eps, args.act = 1e-5, act
    args.N, args.Hin, args.Win, args.Hout, args.Wout = N, Hin, Win, Hout, Wout
    args.ksize, args.stride, args.up, args.fine_slices = ksize, stride, int(up), int(fine)      # fine: False / True (level 1) / 2
    args.w, args.bias, args.Cout, args.prec = wdev.data_ptr(), bdev.data_ptr(), cout, conv_prec | int(diag)      # diag: hip.DIAG_* bits
    args.emb_off = -1
    table = emb if emb is not None else film
    if table is not None:
        t = torch.as_tensor(np.asarray(table, dtype=np.float32)).contiguous().to(DEV)
        keep.append(t)
        args.emb_table, args.emb_stride = t.data_ptr(), t.shape[1]
        rows = torch.as_tensor(np.asarray(emb_rows if emb_rows is not None else np.zeros(N), dtype=np.int32)).to(DEV)
        keep.append(rows)
        args.emb_row_of_sample = rows.data_ptr()
        if emb is not None:
            args.emb_off = 0
        if film is not None:
            args.film, args.film_off = 1, 0
    if resid is not None:
        args.resid = resid.data_ptr()
    if skip is not None:
        keep.append(swdev)
        args.skip0, args.SC0 = ssrc[0].data_ptr(), ssrc[0].shape[3]
        if len(ssrc) > 1:
            args.skip1, args.SC1 = ssrc[1].data_ptr(), ssrc[1].shape[3]
        args.skip_w = swdev.data_ptr()
    args.out = out.data_ptr()
    ost = None
    if want_stats:
        S = hip.check(lib.ccdm_conv_out_slices(C.byref(args)), "conv_out_slices")     # the kernel the library selects owns the tiling
        if up == 2 and not fine:
            assert S == lib.ccdm_upconv_slices(Hin, Win)
        ost = torch.empty((N, S, cout, 2), dtype=torch.float64, device=DEV)
        args.out_stats, args.out_slices = ost.data_ptr(), S
    hip.check(lib.ccdm_conv2d(C.byref(args), 0), "conv2d")
    sync()
    if bench is not None:       # tools/bench_conv.py: {"iters": n} in, {"ms": mean launch time} out
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(bench.get("iters", 20)):
            lib.ccdm_conv2d(C.byref(args), 0)
        e1.record()
        sync()
        bench["ms"] = e0.elapsed_time(e1) / bench.get("iters", 20)
    del keep
    return out, ost


def resample(x_nhwc, mode, *, stats=None, gamma=None, beta=None, act=hip.ACT_NONE, want_act=True, want_raw=True):
    """ccdm_resample: returns (R(act(GN(x))) or None, R(x) or None), NHWC cuda."""
    lib = hip.load()
    N, H, W, C_ = x_nhwc.shape
    ho, wo = (H // 2, W // 2) if mode == hip.RESAMPLE_AVGPOOL2 else (2 * H, 2 * W)
    # (an empty output — AvgPool2d(2) of a 1-pixel-high image — still gets a real pointer: the library refuses it by geometry)
    oa = torch.full((N, max(ho, 1), max(wo, 1), C_), float("nan"), device=DEV) if want_act else None
    orw = torch.full((N, max(ho, 1), max(wo, 1), C_), float("nan"), device=DEV) if want_raw else None
    a = hip.ResampleArgs()
    a.in_, a.C = x_nhwc.data_ptr(), C_
    keep = []
    if stats is not None:
        g = torch.as_tensor(np.asarray(gamma, dtype=np.float32)).to(DEV)
        bt = torch.as_tensor(np.asarray(beta, dtype=np.float32)).to(DEV)
        keep += [g, bt]
        a.stats, a.slices, a.gamma, a.beta = stats.data_ptr(), stats.shape[1], g.data_ptr(), bt.data_ptr()
    a.eps, a.act = 1e-5, act
    a.N, a.Hin, a.Win, a.mode = N, H, W, mode
    a.out_act, a.out_raw = (oa.data_ptr() if want_act else 0), (orw.data_ptr() if want_raw else 0)
    hip.check(lib.ccdm_resample(C.byref(a), 0), "resample")
    sync()
    return oa, orw


def attention(qkv_ntc: torch.Tensor, heads: int, order: int) -> torch.Tensor:
    lib = hip.load()
    N, T, C3 = qkv_ntc.shape
    out = torch.empty((N, T, C3 // 3), device=DEV)
    hip.check(lib.ccdm_attention(qkv_ntc.data_ptr(), out.data_ptr(), N, T, C3 // 3, heads, order, 0), "attention")
    sync()
    return out


def posterior_sample(head_nhwk, xt_idx, a, c, mode, *, softmax=True, noise=None, philox_seed=0, sample_offset=0, step=0,
                     head_stride=None, xin_stride=None, xin_fill=0.0, many=False):
    """head [N,HW,K] cuda fp32; xt_idx uint8 [N,HW] cuda.  Returns dict of outputs (cpu).  `head_stride` > K: the K values sit in rows of
    that many floats (the rest is junk the kernel must not read into its result); `xin_stride` / `xin_fill`: pitch and initial content
    of the stem input the one-hot is written into (channels >= K must come back untouched)."""
    lib = hip.load()
    N, HW, K = head_nhwk.shape
    if head_stride is not None and head_stride != K:
        padded = torch.full((N, HW, head_stride), 1.0e3, device=DEV)
        padded[..., :K] = head_nhwk
        head_nhwk = padded
    table = torch.zeros((step + 1, 4), dtype=torch.float32)
    table[step, 0], table[step, 1], table[step, 2] = a, c, float(mode)
    table = table.to(DEV)
    stepbuf = torch.tensor([step], dtype=torch.int32, device=DEV)
    xt_next = torch.full((N, HW), 255, dtype=torch.uint8, device=DEV)
    xin = torch.full((N, HW, xin_stride if xin_stride is not None else (K + 4) // 4 * 4), float(xin_fill), device=DEV)
    probs = torch.zeros((N, HW, K), device=DEV)
    onehot = torch.zeros((N, HW, K), dtype=torch.int64, device=DEV)
    post = torch.zeros((N, HW, K), device=DEV)
    p = hip.PostArgs()
    p.head, p.softmax, p.xt = head_nhwk.data_ptr(), int(softmax) | (hip.POST_DIAG_MANY if many else 0), xt_idx.data_ptr()       # many: the LDS-row kernel at any K
    p.head_stride = head_nhwk.shape[2]
    p.N, p.HW, p.K = N, HW, K
    p.step_table, p.step_ptr = table.data_ptr(), stepbuf.data_ptr()
    if noise is not None:
        p.noise, p.noise_step_stride = noise.data_ptr(), N * HW * K
    p.philox_seed, p.sample_offset = philox_seed, sample_offset
    p.xt_next, p.xin, p.xin_stride = xt_next.data_ptr(), xin.data_ptr(), xin.shape[2]
    p.out_probs, p.out_onehot, p.posterior_out = probs.data_ptr(), onehot.data_ptr(), post.data_ptr()
    hip.check(lib.ccdm_posterior_sample(C.byref(p), 0), "posterior_sample")
    sync()
    return dict(xt_next=xt_next.cpu(), xin=xin.cpu(), probs=probs.cpu(), onehot=onehot.cpu(), posterior=post.cpu())


def norm_qkv_attention(x_nhwc, norm_w, norm_b, qkv_w, qkv_b, heads: int, new_order: bool):
    """ccdm_norm_qkv_attention: GroupNorm + qkv + attention core in one launch.  x [N,h,w,C] cuda; parameters as numpy in the
    reference's layout.  Returns the attention output [N,h,w,C] (channel = head*32 + d)."""
    lib = hip.load()
    N, h, w, Cc = x_nhwc.shape
    st = gn_stats(x_nhwc, 1)
    wq, bq = hip.pack_qkv_weights(qkv_w, qkv_b, heads, new_order)
    dev = [torch.from_numpy(np.ascontiguousarray(v)).to(DEV) for v in (wq, bq, np.asarray(norm_w, np.float32), np.asarray(norm_b, np.float32))]
    out = torch.full_like(x_nhwc, float("nan"))
    a = hip.AttnBlockArgs()
    a.x, a.stats, a.slices = x_nhwc.data_ptr(), st.data_ptr(), 1
    a.gamma, a.beta, a.eps = dev[2].data_ptr(), dev[3].data_ptr(), 1e-5
    a.wqkv, a.bqkv = dev[0].data_ptr(), dev[1].data_ptr()
    a.out = out.data_ptr()
    a.N, a.T, a.C, a.heads = N, h * w, Cc, heads
    hip.check(lib.ccdm_norm_qkv_attention(C.byref(a), 0), "norm_qkv_attention")
    sync()
    return out


def stem_conv(xt_idx, xin_nhwc, K, weight, bias):
    """ccdm_stem_conv: xt_idx uint8 cuda [N,H,W]; xin NHWC cuda [N,H,W,4] (image in channels [K, 4)); weight [Cout, Cin<=4, 3, 3].
    Returns (out NHWC cuda, stats [N, slices, Cout, 2])."""
    lib = hip.load()
    N, H, W, Cs = xin_nhwc.shape
    w = np.ascontiguousarray(weight, dtype=np.float32)
    cout = w.shape[0]
    wdev = torch.from_numpy(hip.pack_stem_weight(w)).to(DEV)
    bdev = torch.as_tensor(np.asarray(bias, dtype=np.float32)).to(DEV)
    S = lib.ccdm_conv_slices(H, W, 1, 3)
    out = torch.empty((N, H, W, cout), device=DEV)
    st = torch.empty((N, S, cout, 2), dtype=torch.float64, device=DEV)
    a = hip.StemArgs()
    a.xt, a.xin, a.Cs, a.K = xt_idx.data_ptr(), xin_nhwc.data_ptr(), Cs, K
    a.w, a.bias = wdev.data_ptr(), bdev.data_ptr()
    a.N, a.H, a.W, a.Cout = N, H, W, cout
    a.out, a.out_stats, a.out_slices = out.data_ptr(), st.data_ptr(), S
    hip.check(lib.ccdm_stem_conv(C.byref(a), 0), "stem_conv")
    sync()
    return out, st


def head_posterior(x_nhwc, gamma, beta, weight, bias, xt_idx, a, c, mode, *, softmax=True, noise=None, philox_seed=0, sample_offset=0, step=0):
    """ccdm_head_posterior: x [N,H,W,32] cuda; weight [K,32,3,3]; xt_idx uint8 [N,H*W] cuda.  Returns dict of outputs (cpu) incl. the
    logits tap."""
    lib = hip.load()
    N, H, W, Cc = x_nhwc.shape
    w = np.ascontiguousarray(weight, dtype=np.float32)
    K = w.shape[0]
    HW = H * W
    st = gn_stats(x_nhwc, 4)
    dev = [torch.from_numpy(np.ascontiguousarray(v)).to(DEV) for v in (hip.pack_head_weight(w), np.asarray(bias, np.float32), np.asarray(gamma, np.float32),
                                                                       np.asarray(beta, np.float32))]
    table = torch.zeros((step + 1, 4), dtype=torch.float32)
    table[step, 0], table[step, 1], table[step, 2] = a, c, float(mode)
    table = table.to(DEV)
    stepbuf = torch.tensor([step], dtype=torch.int32, device=DEV)
    xt_next = torch.full((N, HW), 255, dtype=torch.uint8, device=DEV)
    probs = torch.zeros((N, HW, K), device=DEV)
    onehot = torch.zeros((N, HW, K), dtype=torch.int64, device=DEV)
    post = torch.zeros((N, HW, K), device=DEV)
    logits = torch.zeros((N, HW, K), device=DEV)
    flag = torch.zeros((1,), dtype=torch.int32, device=DEV)
    h = hip.HeadArgs()
    h.x, h.stats, h.slices = x_nhwc.data_ptr(), st.data_ptr(), 4
    h.gamma, h.beta, h.eps = dev[2].data_ptr(), dev[3].data_ptr(), 1e-5
    h.w, h.bias = dev[0].data_ptr(), dev[1].data_ptr()
    h.N, h.H, h.W, h.C, h.K = N, H, W, Cc, K
    h.logits_out = logits.data_ptr()
    p = hip.PostArgs()
    p.softmax, p.xt = int(softmax), xt_idx.data_ptr()
    p.N, p.HW, p.K = N, HW, K
    p.step_table, p.step_ptr = table.data_ptr(), stepbuf.data_ptr()
    if noise is not None:
        p.noise, p.noise_step_stride = noise.data_ptr(), N * HW * K
    p.philox_seed, p.sample_offset = philox_seed, sample_offset
    p.xt_next = xt_next.data_ptr()
    p.out_probs, p.out_onehot, p.posterior_out = probs.data_ptr(), onehot.data_ptr(), post.data_ptr()
    p.range_flag = flag.data_ptr()
    hip.check(lib.ccdm_head_posterior(C.byref(h), C.byref(p), 0), "head_posterior")
    sync()
    return dict(xt_next=xt_next.cpu(), probs=probs.cpu(), onehot=onehot.cpu(), posterior=post.cpu(), logits=logits.cpu(), flag=int(flag.item()))
