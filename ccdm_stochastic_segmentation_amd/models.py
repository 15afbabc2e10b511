"""Drop-in host interface of the sampler: same names, arguments and error behaviour as the reference's
``ddpm.models`` package for the hot path, backed by the HIP engine.

    build_model(...)                         /root/reference/ddpm/models/builder.py:14-51
    DenoisingModel.forward(x, condition, feature_condition=None, t=None, label_ref_logits=None, validation=False)
                                             /root/reference/ddpm/models/diffusion_denoising.py:144-159
    DiffusionModel (schedule buffers)        /root/reference/ddpm/models/diffusion_denoising.py:42-70
    UNetModel (parameter container with the reference's state_dict key layout)
    OneHotCategoricalBCHW                    /root/reference/ddpm/models/one_hot_categorical.py:10-54

Inference only: there is no autograd through the HIP kernels.
"""
from __future__ import annotations

import logging
import math
from typing import Any, Dict, List, Optional, Tuple, Union, cast

import numpy as np
import torch
from torch import Tensor, nn

from . import hip
from .engine import SamplerEngine
from .unet_spec import UNetSpec, make_unet_spec

LOGGER = logging.getLogger(__name__)

__all__ = ["build_model", "DiffusionModel", "DenoisingModel", "UNetModel", "OneHotCategoricalBCHW",
           "linear_schedule", "cosine_schedule", "step_values"]


# --------------------------------------------------------------------------------------------------
# schedules (host, once per model) — same expressions, same dtypes as the reference so the three
# buffers are bit-identical (diffusion_denoising.py:18-39)
# --------------------------------------------------------------------------------------------------
def linear_schedule(time_steps: int, start=1e-2, end=0.2) -> Tuple[Tensor, Tensor, Tensor]:
    betas = torch.linspace(start, end, time_steps)
    alphas = 1 - betas
    return betas, alphas, torch.cumprod(alphas, dim=0)


def cosine_schedule(time_steps: int, s: float = 8e-3) -> Tuple[Tensor, Tensor, Tensor]:
    s = 0.008                                     # the reference ignores the argument (:27)
    t = torch.arange(0, time_steps)
    cumalphas = torch.cos(((t / time_steps + s) / (1 + s)) * (math.pi / 2)) ** 2

    def f(u):
        return math.cos((u + s) / (1.0 + s) * math.pi / 2) ** 2

    betas = torch.tensor([min(1 - f((i + 1) / time_steps) / f(i / time_steps), 0.999) for i in range(time_steps)])
    return betas, 1 - betas, cumalphas


def step_values(time_steps: int, init_t: Optional[int]) -> List[int]:
    """Step list of forward_denoising (:178-187): full range, a shortened range, or (init_t > 10000) a
    strided walk of K = init_t % 10000 steps from T to 1 (python round = half-to-even)."""
    if init_t is None:
        init_t = time_steps
    if init_t > 10000:
        k = init_t % 10000
        assert 0 < k <= time_steps
        if k == time_steps:
            return list(range(k, 0, -1))
        vals = [round(v) for v in np.linspace(time_steps, 1, k)]
        LOGGER.warning(f"Override default {time_steps} time steps with {len(vals)}.")
        return vals
    return list(range(init_t, 0, -1))


class OneHotCategoricalBCHW:
    """Categorical over dim=1 of a BCHW tensor, sampled as torch.multinomial does: argmax_k p_k / E_k with
    E ~ Exp(1) drawn as one [B*H*W, K] block from the generator of the tensor's device.  Host-side helper
    for callers (x_T is drawn with it on the CPU, evaluate_lidc_uncertainty.py:100); the per-step draws
    inside the loop run in the HIP epilogue."""

    def __init__(self, probs: Optional[Tensor] = None, logits: Optional[Tensor] = None, validate_args=None):
        if (probs is None) == (logits is None):
            raise ValueError("Either `probs` or `logits` must be specified, but not both.")
        if probs is not None and probs.ndim < 2:
            raise ValueError("`probs.ndim` should be at least 2")
        if logits is not None and logits.ndim < 2:
            raise ValueError("`logits.ndim` should be at least 2")
        if probs is not None:
            p = self.channels_last(probs)
            self.probs = p / p.sum(-1, keepdim=True)
        else:
            lg = self.channels_last(logits)
            self.probs = torch.softmax(lg - lg.logsumexp(dim=-1, keepdim=True), dim=-1)

    @staticmethod
    def channels_last(arr: Tensor) -> Tensor:
        return arr.permute((0,) + tuple(range(2, arr.ndim)) + (1,))

    @staticmethod
    def channels_second(arr: Tensor) -> Tensor:
        return arr.permute((0, arr.ndim - 1) + tuple(range(1, arr.ndim - 1)))

    def sample(self, sample_shape=torch.Size(), generator: Optional[torch.Generator] = None) -> Tensor:
        if len(sample_shape) != 0:
            raise NotImplementedError("sample_shape other than () is not used on this path")
        k = self.probs.shape[-1]
        p2d = self.probs.reshape(-1, k)
        q = torch.empty_like(p2d).exponential_(1, generator=generator)
        idx = torch.argmax(p2d / q, dim=-1)
        res = torch.nn.functional.one_hot(idx, k).to(self.probs.dtype).reshape(self.probs.shape)
        return self.channels_second(res)

    def max_prob_sample(self) -> Tensor:
        k = self.probs.shape[-1]
        return self.channels_second(torch.nn.functional.one_hot(self.probs.argmax(dim=-1), k))

    def prob_sample(self) -> Tensor:
        return self.channels_second(self.probs)


class DiffusionModel(nn.Module):
    betas: Tensor
    alphas: Tensor
    cumalphas: Tensor

    def __init__(self, schedule: str, time_steps: int, num_classes: int, schedule_params=None):
        super().__init__()
        fn = {"linear": linear_schedule, "cosine": cosine_schedule}[schedule]
        betas, alphas, cumalphas = fn(time_steps, **schedule_params) if schedule_params is not None else fn(time_steps)
        self.register_buffer("betas", betas)
        self.register_buffer("alphas", alphas)
        self.register_buffer("cumalphas", cumalphas)
        self.num_classes = num_classes

    @property
    def time_steps(self) -> int:
        return len(self.betas)

    def posterior_coeffs(self, t: int) -> Tuple[float, float]:
        """(alpha_t, cumalpha_{t-1}) as theta_post_prob uses them; t == 1 -> (0, 1) (:112-113)."""
        i = t - 1
        if i == 0:
            return 0.0, 1.0
        return float(self.alphas[i]), float(self.cumalphas[i - 1])

    # ------------------------------------------------------------------ training-time forward pieces
    # Same names, arguments and results as the reference's methods (diffusion_denoising.py:72-129); the arithmetic runs in
    # the HIP kernels behind ccdm_mix_uniform / ccdm_theta_post (device tensors only: there is no CPU path).
    @staticmethod
    def _dev(x: Tensor, what: str) -> Tensor:
        if not x.is_cuda:
            raise hip.CcdmHipError(f"{what}: tensors must live on the GPU (the HIP kernels are the only implementation)")
        return x.contiguous().float()

    def _per_sample(self, table: Tensor, t: Tensor, N: int, device) -> Tensor:
        v = table.to(device)[(t.to(device).long() - 1).reshape(-1)].float()
        return (v.expand(N) if v.numel() == 1 else v).contiguous()

    def _coeffs(self, t: Tensor, N: int, device) -> Tuple[Tensor, Tensor]:
        """(alpha_t, cumalpha_{t-1}) per sample with the t == 1 override (:91-94, :110-113)."""
        i = (t.to(device).long() - 1).reshape(-1)
        a = self.alphas.to(device)[i].clone().float()
        c = self.cumalphas.to(device)[i - 1].clone().float()
        a[i == 0] = 0.0
        c[i == 0] = 1.0
        if a.numel() == 1:
            a, c = a.expand(N), c.expand(N)
        return a.contiguous(), c.contiguous()

    def _mix(self, x: Tensor, s: Tensor) -> Tensor:
        x = self._dev(x, "q_xt")
        N, K, H, W = x.shape
        out = torch.empty_like(x)
        lib = hip.load()
        hip.check(lib.ccdm_mix_uniform(x.data_ptr(), s.data_ptr(), N, K, H * W, out.data_ptr(),
                                       torch.cuda.current_stream(x.device).cuda_stream), "mix_uniform")
        return out

    def q_xt_given_xtm1(self, xtm1: Tensor, t: Tensor) -> OneHotCategoricalBCHW:
        """(1 - beta_t) * x_{t-1} + beta_t / K   (:72-78)"""
        s = 1.0 - self._per_sample(self.betas, t, xtm1.shape[0], xtm1.device)
        return OneHotCategoricalBCHW(self._mix(xtm1, s.contiguous()))

    def q_xt_given_x0(self, x0: Tensor, t: Tensor) -> OneHotCategoricalBCHW:
        """cumalpha_t * x_0 + (1 - cumalpha_t) / K   (:80-86)"""
        return OneHotCategoricalBCHW(self._mix(x0, self._per_sample(self.cumalphas, t, x0.shape[0], x0.device)))

    def _theta(self, xt: Tensor, other: Tensor, t: Tensor, prob_mode: int) -> Tensor:
        xt, other = self._dev(xt, "theta_post"), self._dev(other, "theta_post")
        N, K, H, W = xt.shape
        if other.shape != xt.shape:
            raise ValueError(f"theta_post: shapes differ {tuple(xt.shape)} vs {tuple(other.shape)}")
        a, c = self._coeffs(t, N, xt.device)
        out = torch.empty_like(xt)
        lib = hip.load()
        hip.check(lib.ccdm_theta_post(xt.data_ptr(), other.data_ptr(), a.data_ptr(), c.data_ptr(), N, K, H * W, prob_mode,
                                      out.data_ptr(), torch.cuda.current_stream(xt.device).cuda_stream), "theta_post")
        return out

    def theta_post(self, xt: Tensor, x0: Tensor, t: Tensor) -> Tensor:
        """q(x_{t-1} | x_t, x_0)   (:88-97)"""
        return self._theta(xt, x0, t, 0)

    def theta_post_prob(self, xt: Tensor, theta_x0: Tensor, t: Tensor) -> Tensor:
        """sum over x_0 of q(x_{t-1} | x_t, x_0) * theta_x0   (:99-129), O(K) per pixel"""
        return self._theta(xt, theta_x0, t, 1)

    def kl_clamped(self, prob_true: Tensor, prob_pred: Tensor, floor: float = 1e-12) -> Tensor:
        """The diffusion term of the reference's train_step (trainer.py:266-270):
        kl_div(log(clamp(prob_pred, min=floor)), prob_true, reduction='none')."""
        p, q = self._dev(prob_true, "kl_clamped"), self._dev(prob_pred, "kl_clamped")
        if p.shape != q.shape:
            raise ValueError(f"kl_clamped: shapes differ {tuple(p.shape)} vs {tuple(q.shape)}")
        out = torch.empty_like(p)
        hip.check(hip.load().ccdm_kl_clamped(p.data_ptr(), q.data_ptr(), p.numel(), float(floor), out.data_ptr(),
                                             torch.cuda.current_stream(p.device).cuda_stream), "kl_clamped")
        return out


class UNetModel(nn.Module):
    """Parameter container with the reference network's exact state_dict layout (398 tensors for the LIDC
    config), so `load_state_dict(strict=True)` and ignite's `load_objects` work unchanged.  It has no
    torch forward: the forward pass is the HIP op list built by SamplerEngine from these parameters."""

    def __init__(self, spec: UNetSpec):
        super().__init__()
        self.spec = spec
        self.in_channels, self.model_channels, self.out_channels = spec.in_channels, spec.model_channels, spec.out_channels
        self._version_seen = 0
        self._weights_version = 0

        def container_for(path: List[str]) -> nn.Module:
            mod: nn.Module = self
            for part in path:
                if not hasattr(mod, part):
                    mod.add_module(part, nn.Module())
                mod = getattr(mod, part)
            return mod

        for key, shape in spec.param_shapes().items():
            *path, leaf = key.split(".")
            container_for(path).register_parameter(leaf, nn.Parameter(torch.zeros(shape), requires_grad=False))
        self.register_load_state_dict_post_hook(lambda module, incompatible: module.mark_weights_changed())

    def mark_weights_changed(self) -> None:
        self._weights_version += 1

    def _apply(self, fn, *a, **k):   # .to()/.cuda(): parameters are re-created
        r = super()._apply(fn, *a, **k)
        self._weights_version += 1
        return r

    def forward(self, *args, **kwargs):
        raise RuntimeError("UNetModel has no eager forward; call it through DenoisingModel (HIP engine)")


# host-noise staging budget of rng="torch_cpu": the Exp(1) draws are made and uploaded in blocks of whole steps no larger
# than this many bytes (the reference never holds more than one step of noise; one block of a few steps keeps the copies
# large without O(T) memory on the host and the device)
HOST_NOISE_BLOCK_BYTES = 256 << 20


def auto_substreams(N: int, H: int, W: int) -> int:
    """`substreams = 0`: how many concurrent sub-batch streams a batch of N samples of H x W is sampled on."""
    return 2 if (N >= 2 and N * H * W >= 32 * 128 * 128) else 1


class DenoisingModel(nn.Module):
    """The sampler.  `rng` selects where the Exp(1) noise of the categorical draws comes from:
    "philox" (default) — Philox4x32-10 inside the epilogue kernel, keyed by (pixel, global sample index, step) under a 64-bit key
      derived from (`philox_seed`, `philox_call`): the throughput mode, statistically identical to the reference's draws.
      `philox_call` counts the sampling calls made on this model and advances by one after every `forward_denoising`, so successive
      calls — the batches of an evaluation loop, S calls at batch 1 to draw S samples of one image — see independent noise, like
      successive draws from the reference's generator do.  The key does not depend on how the batch is split over ranks or
      sub-batches (every rank makes the same sequence of calls), so sharding stays invariant; set `philox_call` back (or
      `philox_advance = False`) to replay a call bit for bit.  torch.manual_seed does not reach this stream: seed it with
      `philox_seed` (params file key of the same name);
    "torch_cpu" — drawn on the host from torch's global CPU generator in exactly the order the reference's CPU path
      consumes it (parity mode: seeded runs reproduce the reference's class indices; the host RNG is the bottleneck).
    `prec` selects the conv arithmetic: hip.PREC_F16X3 (default; fp16 hi/lo split x3 on the matrix cores, ~2^-22 per
    product, with the exact-fp32 kernel taking over any layer whose raw input leaves the split's range) or
    hip.PREC_F32 (exact fp32 MFMA everywhere, the validation mode).  hip.PREC_F16 is the OPT-IN single-pass fast mode (every conv on the
    general kernel with one fp16 MFMA per product; attention cores keep the split): narrower arithmetic than the reference's, outside the
    parity contract, never a default."""

    def __init__(self, diffusion: DiffusionModel, unet: UNetModel, dataset_file: str, step_T_sample: str = "majority"):
        super().__init__()
        self.diffusion = diffusion
        self.unet = unet
        self.dataset_file = dataset_file
        self.step_T_sample = step_T_sample
        self.rng = "philox"
        self.philox_seed = 0
        self.philox_call = 0            # index of the next sampling call's noise stream (see the class docstring)
        self.philox_advance = True
        self.sample_offset = 0          # global index of sample 0 when the batch is sharded over ranks
        self.noise_slice: Optional[Tuple[int, int]] = None   # (global_batch, first_sample) for torch_cpu sharding
        # each denoise step is replayed as one captured HIP graph (identical results to eager launches, tested)
        self.use_graph = True
        # the batch is sampled as this many contiguous sub-batches on concurrent HIP streams (bit-identical samples).  0 (default) =
        # automatic: two once the batch holds as many pixels as 32 samples of 128x128 (+7-10 % at N = 64: the low-resolution kernels of one half run beside the full-width
        # kernels of the other), one below.  bench.py times this default and collects its per-kernel taps in a separate
        # single-stream pass (under concurrency a launch's duration no longer describes the kernel).
        self.substreams = 0
        # Round 6: with substreams = 0 the choice between the bit-identical execution modes — one stream or two sub-batch streams, HIP-graph
        # replay or eager launches — is MEASURED once per (batch geometry, weights) on the box at hand (`_calibrate_mode`: a dozen denoise
        # steps of the call's own workload in each mode, results discarded), because the ranking moves with the box by a few per cent
        # (round 5: two graph streams 89.7 against 85.9 samples/s on one box, 89.4 against 91.9 on another).  The samples do not depend on it.
        # False = the static rule (auto_substreams) and `use_graph` as set.  `mode_choice` records what was measured and picked.
        self.calibrate_mode = True
        self.mode_choice: Dict[Any, dict] = {}
        self.last_mode: Optional[Tuple[int, bool]] = None           # (sub-batch streams, graph replay) of the most recent sampling call
        self.prec = hip.PREC_F16X3
        # what to do when a PREC_F16X3 run reports a range overflow (hip.CcdmRangeError):
        # "layers" (default) = repeat the call with the exact-fp32 kernels (same seeds, so its samples are the ones an all-fp32 run
        #   would have drawn), measure on that run what every conv stages (ccdm_engine_input_absmax) and pin the layers whose staged
        #   values come within RANGE_MARGIN of the fp16 split's limit to the exact-fp32 kernels from then on (`f32_layers`): later
        #   calls run the F16X3 engine with those few layers in fp32 instead of paying 4x for everything;
        # "f32" = only repeat the call in fp32 (round-2 behaviour); "raise" = propagate the error
        self.on_range_error = "layers"
        self.f32_layers: set = set()
        # what the range fallback changed by itself, and for which weights: automatic pins and an automatic switch to fp32 are undone
        # when the U-Net's parameters change (they describe THOSE weights); `range_events` counts them so a caller can see the mode change
        self._auto_pins: set = set()
        self._auto_f32_from: Optional[int] = None           # the precision an unattributable overflow switched away from
        self._auto_weights_key: Optional[Tuple[int, int]] = None
        self.range_events = {"overflows": 0, "layers_pinned": 0, "switched_to_f32": 0, "reset_on_new_weights": 0}
        self._range_probe: Optional[Dict[str, float]] = None        # filled while a diagnosing fp32 re-run is under way
        # workgroup slicing of the conv kernels: "throughput" (default) = the batch-size-independent rule — samples do not depend on how
        # a batch is sharded over ranks or sub-batches, bit for bit; "latency" = up to 32 one- or two-tile workgroups per sample
        # (ccdm_conv_args.fine_slices) for batches too small to fill the chip (LIDC batch 8: 2.05 -> 1.39 ms per denoise step; batch
        # 64 loses 5 %): GroupNorm's partial sums are then added in another order, so its samples may differ from the default mode's in
        # the last bit of a probability (never between two runs of the same mode and batch split)
        self.slicing = "throughput"
        self._engines: Dict[Any, Tuple[int, SamplerEngine]] = {}

    @property
    def time_steps(self) -> int:
        return self.diffusion.time_steps

    def _fine_slices(self, N: int) -> int:
        """ccdm_conv_args.fine_slices for a batch of N: 0 = the batch-size-independent rule; latency mode: 2 (up to 64 slices: one tile per
        workgroup at 128x128) for N <= 8, else 1 (up to 32)."""
        if self.slicing not in ("throughput", "latency"):
            raise ValueError(f"slicing: {self.slicing!r} (expected 'throughput' or 'latency')")
        return 0 if self.slicing != "latency" else (2 if N <= 8 else 1)

    # ------------------------------------------------------------------ reference API
    def forward(self, x: Tensor, condition: Tensor, feature_condition: Tensor = None, t: Optional[Tensor] = None,
                label_ref_logits: Optional[Tensor] = None, validation: bool = False) -> Union[Tensor, dict]:
        if self.training:
            if not isinstance(t, Tensor):
                raise ValueError("'t' needs to be a Tensor at training time")
            if not isinstance(x, Tensor):
                raise ValueError("'x' needs to be a Tensor at training time")
            return self.forward_step(x, condition, feature_condition, t)
        if validation:
            return self.forward_step(x, condition, feature_condition, t)
        if t is None:
            return self.forward_denoising(x, condition, feature_condition, label_ref_logits=label_ref_logits)
        return self.forward_denoising(x, condition, feature_condition, cast(int, t.item()), label_ref_logits)

    # ------------------------------------------------------------------ engine plumbing
    def _weights_key(self) -> Tuple[int, int]:
        """Changes whenever the U-Net's parameters do: load_state_dict / .to() bump the explicit counter, in-place updates
        (optimizer steps, Polyak averaging, p.data.copy_) bump the tensors' own version counters."""
        return self.unet._weights_version, sum(int(p._version) for p in self.unet.parameters())

    def _engine(self, x: Tensor, condition: Tensor, feature_condition: Optional[Tensor], slot: int = 0) -> SamplerEngine:
        """The step executor for this input geometry (cached).  `slot` tells apart the engines of concurrent sub-batches."""
        N, K, H, W = x.shape
        spec = self.unet.spec
        if feature_condition is not None and spec.feature_condition_unwired:
            # the reference concatenates at these blocks without having widened them and fails in the conv (unet.py:770-788)
            raise RuntimeError(f"feature_condition given, but input block(s) {spec.feature_condition_unwired} were not widened for it "
                               "(feature_cond_encoder target_layer / output_stride do not line up): channel mismatch")
        fshape = tuple(feature_condition.shape[1:]) if feature_condition is not None else None
        if not spec.feature_condition_idx:
            fshape = None                        # no injection point configured: the reference ignores the tensor too
        dev = next(self.unet.parameters()).device
        f32_layers = frozenset(self.f32_layers) if self.prec == hip.PREC_F16X3 else frozenset()
        key = (N, H, W, int(condition.shape[1]), fshape, str(dev), self.prec, slot, self._fine_slices(N), f32_layers)
        wkey = self._weights_key()
        hit = self._engines.get(key)
        if hit is not None and hit[0] == wkey:
            return hit[1]
        eng = SamplerEngine(spec, self.unet.state_dict(), N, H, W, K, int(condition.shape[1]), dev,
                            max_steps=self.time_steps, feature_shape=fshape, prec=self.prec, fine_slices=self._fine_slices(N),
                            f32_layers=f32_layers)
        self._engines = {k: v for k, v in self._engines.items() if v[0] == wkey}
        self._engines[key] = (wkey, eng)
        return eng

    def _philox_key(self) -> int:
        """64-bit Philox key of the current sampling call: splitmix64 of (philox_seed, philox_call)."""
        if int(self.philox_call) == 0:
            return int(self.philox_seed) & 0xFFFFFFFFFFFFFFFF        # call 0 uses the seed itself (the key the kernel tests pin)
        z = (int(self.philox_seed) + 0x9E3779B97F4A7C15 * int(self.philox_call)) & 0xFFFFFFFFFFFFFFFF
        z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & 0xFFFFFFFFFFFFFFFF
        z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & 0xFFFFFFFFFFFFFFFF
        return z ^ (z >> 31)

    @staticmethod
    def _to_index(x: Tensor, device) -> Tensor:
        return x.argmax(dim=1).to(device=device, dtype=torch.uint8).contiguous()

    # staged values within this factor of hip.F16X3_LIMIT pin a layer to fp32 (the probe sees the steps of one call, not every input)
    RANGE_MARGIN = 0.5

    def _probe_ranges(self, engines) -> None:
        """diagnosing fp32 re-run: fold what every conv / attention core of these (exact-fp32) engines staged in the step they just ran
        into the probe — enqueued on the engine's stream into a device buffer per engine (a running maximum over all steps of the call),
        read once at the end (`_collect_probe`)"""
        if self._range_probe is None:
            return
        for eng in engines:
            hit = self._range_probe.get(id(eng))
            if hit is None:
                hit = self._range_probe[id(eng)] = (eng, eng.probe_buffer())
            eng.probe_ranges(hit[1])

    def _collect_probe(self) -> Optional[Dict[str, float]]:
        if self._range_probe is None:
            return None
        out: Dict[str, float] = {}
        for eng, buf in self._range_probe.values():
            for name, v in eng.read_ranges(buf).items():
                out[name] = max(out.get(name, 0.0), v)
        return out

    def _with_range_fallback(self, fn):
        if self.on_range_error not in ("layers", "f32", "raise"):
            raise ValueError(f"on_range_error: {self.on_range_error!r} (expected 'layers', 'f32' or 'raise')")
        state = torch.get_rng_state() if self.rng == "torch_cpu" else None
        if self._auto_weights_key is not None and self._auto_weights_key != self._weights_key():
            # new weights: what the fallback learned about the old ones (pins, the switch to fp32) no longer applies
            self.f32_layers -= self._auto_pins
            if self._auto_f32_from is not None and self.prec == hip.PREC_F32:
                self.prec = self._auto_f32_from
            self._auto_pins, self._auto_f32_from, self._auto_weights_key = set(), None, None
            self.range_events["reset_on_new_weights"] += 1
        try:
            return fn()
        except hip.CcdmRangeError as e:
            if self.prec == hip.PREC_F32 or self.on_range_error == "raise":
                raise
            self.range_events["overflows"] += 1
            LOGGER.warning("%s -- repeating this call with the exact-fp32 kernels", e)
            if state is not None:
                torch.set_rng_state(state)
            prec, self.prec = self.prec, hip.PREC_F32
            self._range_probe = {} if self.on_range_error == "layers" else None
            probe = None
            try:
                out = fn()
                probe = self._collect_probe()
            finally:
                self.prec = prec
                self._range_probe = None
            if probe is not None:
                limit = hip.F16X3_LIMIT * self.RANGE_MARGIN
                hot = sorted(k for k, v in probe.items() if not (v < limit))
                new = [k for k in hot if k not in self.f32_layers]
                if new:
                    self.f32_layers.update(new)
                    self._auto_pins.update(new)
                    self._auto_weights_key = self._weights_key()
                    self.range_events["layers_pinned"] += len(new)
                    LOGGER.warning("F16X3 range: %d layer(s) stage values beyond %.0f (%s ...); they run the exact-fp32 kernels from now on "
                                   "(DenoisingModel.f32_layers)", len(new), limit, ", ".join(new[:4]))
                    # the engines built for the old pin set and the diagnosing all-fp32 ones are dead weight now — at Cityscapes-sized
                    # batches three activation sets per sub-batch slot is what runs a GPU out of memory exactly when the fallback triggers
                    self._engines.clear()
                else:
                    # an overflow the probe cannot attribute to a pinnable layer (every step of this call was looked at): without a
                    # new pin every later call would run F16X3, overflow and be repeated in fp32 — about 5x the cost, forever
                    LOGGER.warning("F16X3 range: the overflow could not be attributed to a layer (largest staged value %.3g); this model "
                                   "runs the exact-fp32 kernels until its weights change (DenoisingModel.prec = PREC_F32; range_events counts it)",
                                   max(probe.values()) if probe else float("nan"))
                    self._auto_f32_from, self._auto_weights_key = prec, self._weights_key()
                    self.range_events["switched_to_f32"] += 1
                    self.prec = hip.PREC_F32
                    self._engines = {k: v for k, v in self._engines.items() if k[6] == hip.PREC_F32}
            return out

    # ------------------------------------------------------------------ measured choice of the execution mode (substreams = 0)
    CALIBRATION_MIN_STEPS = 100         # shorter walks are not worth a calibration (~200 denoise steps of probing): the static rule decides
    CALIBRATION_STEPS = (4, 16)         # (untimed, timed) denoise steps per candidate mode and round
    CALIBRATION_ROUNDS = 3              # interleaved timing rounds; the minimum counts (C2: ~0.6 s once per geometry and set of weights)

    def _calibrate_mode(self, x: Tensor, condition: Tensor, feature_condition: Optional[Tensor], prepare, run_steps,
                        nsub_rule: int, graph_allowed: bool, n_rows: int) -> Tuple[int, bool]:
        """Time the bit-identical execution modes on this call's own workload and return the fastest (sub-batch streams, graph replay).
        Candidates: one stream or `nsub_rule` streams; graph replay (if `use_graph` allows it) or eager launches.  Each runs
        CALIBRATION_ROUNDS x CALIBRATION_STEPS denoise steps from x_T with the call's tables (the draws are discarded: every engine's inputs are set again
        by the caller, Philox counters depend on the step row only, the host generator is not touched — this path is never taken with
        rng = "torch_cpu").  Once per (geometry, precision, slicing, weights); `mode_choice` keeps the timings."""
        import time
        N, K, H, W = x.shape
        ck = (N, K, H, W, int(condition.shape[1]), None if feature_condition is None else tuple(feature_condition.shape[1:]), self.prec,
              self._fine_slices(N), graph_allowed, nsub_rule, self._weights_key())
        hit = self.mode_choice.get(ck)
        if hit is not None:
            return hit["nsub"], hit["use_graph"]
        warm, timed = (max(1, min(int(v), n_rows)) for v in self.CALIBRATION_STEPS)       # (rows of the call's own tables)
        cands = [(ns, g) for ns in (nsub_rule, 1) for g in ((True, False) if graph_allowed else (False,))]
        try:
            parts_of = {ns: prepare(ns) for ns in (nsub_rule, 1)}       # (both engine sets exist from here on: twice the activation memory)
        except torch.cuda.OutOfMemoryError:
            LOGGER.warning("execution-mode measurement skipped for N=%d %dx%d: no memory for both engine sets; static rule (%d streams)", N, H, W, nsub_rule)
            self._engines = {}
            torch.cuda.empty_cache()
            self.mode_choice[ck] = {"nsub": nsub_rule, "use_graph": graph_allowed, "ms_per_denoise_step": {}}
            return nsub_rule, graph_allowed
        dev = parts_of[1][0][0].device
        for ns, g in cands:                                                   # untimed: captures each engine's graph, warms weights and code
            run_steps(parts_of[ns], 0, warm, [None] * ns, 0, g)
        torch.cuda.synchronize(dev)
        times = {c: float("inf") for c in cands}
        for _ in range(self.CALIBRATION_ROUNDS):                              # interleaved rounds, minimum per candidate: the modes differ by 1-4 %
            for ns, g in cands:
                t0 = time.perf_counter()
                run_steps(parts_of[ns], 0, timed, [None] * ns, 0, g)
                torch.cuda.synchronize(dev)
                times[(ns, g)] = min(times[(ns, g)], (time.perf_counter() - t0) / timed * 1e3)
        for parts_ in parts_of.values():
            for eng, lo, hi in parts_:
                eng.check_and_clear_flag()                                   # (the real call reports overflows; a probe run must not leave a flag behind)
        best = min(times, key=times.get)
        self.mode_choice = {k_: v for k_, v in self.mode_choice.items() if k_[-1] == ck[-1]}      # (entries of older weights go)
        self.mode_choice[ck] = {"nsub": best[0], "use_graph": best[1],
                                "ms_per_denoise_step": {f"{ns} stream{'s' if ns > 1 else ''}, {'graph' if g else 'eager'}": round(v, 4) for (ns, g), v in times.items()}}
        LOGGER.info("execution mode for N=%d %dx%d: %d stream(s), %s (ms per denoise step: %s)", N, H, W, best[0], "graph" if best[1] else "eager",
                    self.mode_choice[ck]["ms_per_denoise_step"])
        return best

    def forward_step(self, x: Tensor, condition: Tensor, feature_condition: Tensor, t: Tensor) -> dict:
        return self._with_range_fallback(lambda: self._forward_step(x, condition, feature_condition, t))

    def forward_denoising(self, x: Optional[Tensor], condition: Tensor, feature_condition: Tensor,
                          init_t: Optional[int] = None, label_ref_logits: Optional[Tensor] = None) -> dict:
        out = self._with_range_fallback(lambda: self._forward_denoising(x, condition, feature_condition, init_t, label_ref_logits))
        if self.philox_advance:
            self.philox_call += 1           # the next call draws from a fresh stream (a range-error re-run above replayed this one)
        return out

    def _forward_step(self, x: Tensor, condition: Tensor, feature_condition: Tensor, t: Tensor) -> dict:
        """One U-Net evaluation at per-sample timesteps `t` (diffusion_denoising.py:161-162 -> unet.py:744-808): the
        network's own output (probabilities, or logits with `softmax_output: no`) and, with `ce_head`, the extra head's
        logits (unet.py:716-726,805-807).  `x` must be one-hot (it is everywhere on this path)."""
        eng = self._engine(x, condition, feature_condition)
        N = x.shape[0]
        tt = t.detach().float().reshape(-1).cpu()
        if tt.numel() == 1:
            tt = tt.expand(N)
        with eng.enter():
            eng.set_inputs(self._to_index(x, eng.device), condition.to(eng.device), feature_condition)
            eng.set_tables([float(v) for v in tt], [(0.0, 1.0, hip.STEP_SOFTMAX_ONLY)] * N, per_sample=True)
            eng.run(1, with_epilogue=True, use_graph=False)
            out = eng.out_probs.clone().permute(0, 3, 1, 2)
            logits = eng.ce_logits()
        eng.leave()
        eng.raise_if_flagged()
        self._probe_ranges([eng])
        return {"diffusion_out": out, "logits": logits}

    def _forward_denoising(self, x: Optional[Tensor], condition: Tensor, feature_condition: Tensor,
                           init_t: Optional[int] = None, label_ref_logits: Optional[Tensor] = None) -> dict:
        if label_ref_logits is not None:
            # the reference's guidance branch reads attributes that do not exist (guidance_scale_weights,
            # diffusion_denoising.py:172-174): it raises AttributeError there too.
            raise AttributeError("'DenoisingModel' object has no attribute 'guidance_scale_weights'")
        T = self.time_steps
        t_values = step_values(T, init_t)
        N, K, H, W = x.shape
        S = len(t_values)
        vote = self.step_T_sample
        last_mode = (hip.STEP_LAST_MAJORITY if vote is None or vote == "majority"
                     else hip.STEP_LAST_CONFIDENCE if vote == "confidence" else hip.STEP_LAST_KEEP)
        coeffs = []
        for t in t_values:
            a, c = self.diffusion.posterior_coeffs(t)
            coeffs.append((a, c, hip.STEP_SAMPLE if t > 1 else last_mode))
        if self.rng not in ("philox", "torch_cpu"):
            raise ValueError(f"unknown rng mode {self.rng!r}")
        host_rng = self.rng == "torch_cpu"
        key = self._philox_key()
        gN, first = self.noise_slice if (host_rng and self.noise_slice is not None) else (N, 0)
        # Samples are independent through all T steps: the batch may be walked as several contiguous
        # sub-batches, each with its own step executor on its own HIP stream.  The kernels of the low-resolution stages
        # fill a fraction of the GPU; two sub-batches half a step apart fill each other's gaps.  Nothing a sample sees
        # depends on the split (statistics are per sample, slices are a function of the spatial size only, noise is keyed
        # by the global sample index or sliced from the full-batch host draw): the results are bit-identical.
        # automatic: two sub-batches once the batch holds as many pixels as 32 LIDC samples (N >= 32 at 128x128; the Cityscapes-shaped
        # batches of 16 x 256x512 and 4 x 512x1024 qualify: +2.3 % / +2.0 % measured), one below
        def prepare(nsub_: int):
            bounds = [(N * j) // nsub_ for j in range(nsub_ + 1)]
            parts_ = []
            for j in range(nsub_):
                lo, hi = bounds[j], bounds[j + 1]
                fc = feature_condition[lo:hi] if feature_condition is not None else None
                eng = self._engine(x[lo:hi], condition[lo:hi], fc, slot=j)
                with eng.enter():
                    eng.set_inputs(self._to_index(x[lo:hi], eng.device), condition[lo:hi].to(eng.device), fc)
                    eng.set_tables([float(t) for t in t_values], coeffs)
                parts_.append((eng, lo, hi))
            return parts_

        def run_steps(parts_, s0_: int, s1_: int, noises_, noise_row0_: int, graph_: bool):
            if len(parts_) == 1:
                parts_[0][0].run(s1_ - s0_, first_row=s0_, noise=noises_[0], noise_row0=noise_row0_, philox_seed=key,
                                 sample_offset=self.sample_offset, use_graph=graph_)
            else:
                for s in range(s0_, s1_):              # one step of every sub-batch in turn: the streams advance side by side
                    for j, (eng, lo, hi) in enumerate(parts_):
                        eng.run(1, first_row=s, noise=noises_[j], noise_row0=noise_row0_, philox_seed=key,
                                sample_offset=self.sample_offset + lo, use_graph=graph_)

        nsub = int(self.substreams) if int(self.substreams) > 0 else auto_substreams(N, H, W)
        nsub = max(1, min(nsub, N))
        use_graph = bool(self.use_graph)
        if (int(self.substreams) <= 0 and self.calibrate_mode and not host_rng and self._range_probe is None and nsub > 1
                and S >= self.CALIBRATION_MIN_STEPS):
            nsub, use_graph = self._calibrate_mode(x, condition, feature_condition, prepare, run_steps, nsub, use_graph, S)
        self.last_mode = (nsub, use_graph)
        parts = prepare(nsub)
        # Step blocks.  Device RNG: one block.  Host RNG (parity mode): the Exp(1) noise is drawn from torch's CPU generator
        # one [gN*H*W, K] block per sampling step — exactly the reference's consumption order, whatever the blocking — and
        # uploaded a bounded number of steps at a time, so host and device hold O(block), not O(T), noise.
        if host_rng:
            per_step = gN * H * W * K * 4
            blk = max(1, HOST_NOISE_BLOCK_BYTES // max(per_step, 1))
            blocks = [(s0, min(s0 + blk, S)) for s0 in range(0, S, blk)]
        else:
            blocks = [(0, S)]
        if self._range_probe is not None:      # diagnosing fp32 re-run: one step per block, every step's activations are probed
            blocks = [(s, s + 1) for s in range(S)]
        for s0, s1 in blocks:
            noises: List[Optional[Tensor]] = [None] * nsub
            if host_rng:
                host = torch.empty((s1 - s0, gN, H * W * K), dtype=torch.float32)
                for j in range(s0, s1):             # one draw per step with t > 1, like torch.multinomial; the last step draws nothing
                    if t_values[j] > 1:
                        host[j - s0].view(-1).exponential_(1)
                for j, (eng, lo, hi) in enumerate(parts):
                    with torch.cuda.stream(eng.stream):
                        noises[j] = host[:, first + lo:first + hi].contiguous().to(eng.device)
            run_steps(parts, s0, s1, noises, s0, use_graph)
            self._probe_ranges([p_[0] for p_ in parts])
        outs = []
        for eng, lo, hi in parts:
            with eng.enter():
                if t_values[-1] > 1 or last_mode == hip.STEP_LAST_KEEP:
                    idx = eng.xt.reshape(hi - lo, H, W).long()
                    out = torch.nn.functional.one_hot(idx, K).permute(0, 3, 1, 2).to(torch.float32)
                elif last_mode == hip.STEP_LAST_CONFIDENCE:
                    out = eng.out_probs.clone().permute(0, 3, 1, 2)          # BCHW view of channels-last memory, like the reference
                else:
                    out = eng.out_onehot.clone().permute(0, 3, 1, 2)
            eng.leave()
            outs.append(out)
        # read AND clear the sticky range flag of every sub-batch engine before raising: a flag left set on a cached engine would
        # fail the next, unrelated call on it
        flagged = [eng.check_and_clear_flag() for eng, lo, hi in parts]
        if any(flagged):
            parts[0][0].raise_range_error()
        out = outs[0] if nsub == 1 else torch.cat(outs, 0)
        if out.device != x.device:
            out = out.to(x.device)
        return {"diffusion_out": out}


def build_model(
        time_steps: int,
        schedule: str,
        schedule_params: Union[dict, None],
        input_shapes: List[Tuple[int, int, int]],
        cond_encoded_shape,
        backbone: str,
        backbone_params: Dict[str, Any],
        dataset_file: str,
        step_T_sample: str = None,
        feature_cond_encoder: dict = None
) -> DenoisingModel:
    """Same signature and dispatch as the reference's backbone registry (builder.py:14-51)."""
    img_shape, label_shape = input_shapes
    img_channels = img_shape[0]
    num_classes = label_shape[0]
    if not 2 <= num_classes <= hip.MAX_CLASSES:
        # x_t travels as a uint8 class index; up to 32 classes a pixel's K values stay in registers, more go through LDS rows
        # (ccdm_sampler.hip: k_posterior_many).  The reference's datasets have 2 (LIDC) and 20 (Cityscapes) classes.
        raise ValueError(f"label shape {tuple(label_shape)}: {num_classes} classes, the HIP sampler is built for 2..{hip.MAX_CLASSES} (DESIGN.md section 8)")
    diffusion = DiffusionModel(schedule, time_steps, num_classes, schedule_params=schedule_params)
    if backbone == "unet_openai":
        spec = make_unet_spec(
            image_size=min(img_shape[1], img_shape[2]),
            in_channels=num_classes + img_channels,
            out_channels=num_classes,
            num_res_blocks=2,
            cond_encoded_shape=cond_encoded_shape,
            feature_cond_encoder=feature_cond_encoder,
            **backbone_params
        )
        model = UNetModel(spec)
    else:
        raise NotImplementedError(f"backbone {backbone}")
    LOGGER.info("%s trainable params: %d", backbone, spec.num_params())
    return DenoisingModel(diffusion, model, dataset_file, step_T_sample)
