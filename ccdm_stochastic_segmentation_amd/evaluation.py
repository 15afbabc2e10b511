"""Evaluation harness around the sampler: the callers of the hot path, with the reference's
parameter-file keys.

    eval_lidc_uncertainty(params)      /root/reference/evaluation/evaluate_lidc_uncertainty.py:164-216 (Tester.test_step :89-136)
    eval_lidc_sampling_speed(params)   /root/reference/evaluation/evaluate_lidc_sampling_speed.py:165-223 (t = 10000 + K sweep)

No ignite: the loop is a plain `for batch in loader`.  The dataset is `datasets.lidc`'s Test_LIDC re-stated on h5py
when `data_lidc.hdf5` is available, or a deterministic synthetic stand-in (`dataset_file: synthetic.lidc`) so the
entry point runs end to end without the (non-redistributable) data.
"""
from __future__ import annotations

import logging
import os
from typing import Dict, List, Optional, Sequence

import numpy as np
import torch

from . import metrics as M
from .models import OneHotCategoricalBCHW, build_model

LOGGER = logging.getLogger(__name__)


def expanduservars(path: str) -> str:
    return os.path.expanduser(os.path.expandvars(path))


# ------------------------------------------------------------------------------------------------ data
class SyntheticLIDC(torch.utils.data.Dataset):
    """LIDC-shaped stand-in: image [1,128,128] in [-1,1], four one-hot annotations [4,2,128,128] (discs of jittered radius)."""
    NUM_CLASSES, RESOLUTION = 2, 128

    def __init__(self, size: int = 8, seed: int = 0):
        self.size, self.seed = size, seed

    def __len__(self):
        return self.size

    def __getitem__(self, i):
        rng = np.random.default_rng(self.seed * 100003 + i)
        r = self.RESOLUTION
        yy, xx = np.mgrid[0:r, 0:r]
        cy, cx, rad = rng.uniform(40, 88), rng.uniform(40, 88), rng.uniform(6, 20)
        image = np.exp(-((yy - cy) ** 2 + (xx - cx) ** 2) / (2 * rad ** 2)) + 0.1 * rng.standard_normal((r, r))
        image = torch.from_numpy(((image - 0.5) * 2).astype(np.float32))[None].clamp(-1, 1)
        labs = []
        for a in range(4):
            ra = rad * rng.uniform(0.6, 1.3) if rng.random() > 0.2 else 0.0       # an annotator may mark nothing
            m = ((yy - cy) ** 2 + (xx - cx) ** 2 <= ra ** 2).astype(np.int64)
            labs.append(torch.nn.functional.one_hot(torch.from_numpy(m), 2).permute(2, 0, 1).float())
        return image, torch.stack(labs), np.array([0.25, 0.25, 0.25, 0.25])


class TestLIDC(torch.utils.data.Dataset):
    """`Test_LIDC` + `batch_transform` of the reference (datasets/lidc.py:164-198): image*2 -> [-1,1]; labels
    [4,2,128,128] one-hot; uniform annotator weights.  `source` is the path of `data_lidc.hdf5` (read with h5py) or any
    mapping with the file's layout: source[split]["images"][i] -> [128,128] float, source[split]["labels"][i] -> [4,128,128]
    integer.  A path ending in `.npz` is read with numpy as a mirror of the HDF5 file (arrays named "<split>/images" and
    "<split>/labels": `tools/lidc_hdf5_to_npz.py` writes one where h5py exists) — the file form that runs on hosts without h5py.
    `max_size` follows `test_dataset(max_size)` (datasets/lidc.py:201-210): None = the whole split, else the
    first max_size items (the reference's Subset(range(max_size)) fails when the split is smaller; here it is capped)."""

    def __init__(self, source, split: str = "test", max_size: Optional[int] = None):
        if isinstance(source, (str, os.PathLike)) and str(source).endswith(".npz"):
            arrays = np.load(source)
            missing = [k for k in (f"{split}/images", f"{split}/labels") if k not in arrays.files]
            if missing:
                raise KeyError(f"{source}: no array named {missing[0]!r} (an .npz mirror of data_lidc.hdf5 holds '<split>/images' and '<split>/labels')")
            source = {split: {"images": arrays[f"{split}/images"], "labels": arrays[f"{split}/labels"]}}
        elif isinstance(source, (str, os.PathLike)):
            try:
                import h5py
            except ImportError as e:
                raise ImportError("reading data_lidc.hdf5 needs h5py (README.md, requirements), which is not installed here; "
                                  "tools/lidc_hdf5_to_npz.py writes an .npz mirror this reader takes without it") from e
            source = h5py.File(source, "r")
        self.ds = source[split]
        self.n = len(self.ds["images"]) if max_size is None else min(int(max_size), len(self.ds["images"]))

    def __len__(self):
        return self.n

    def __getitem__(self, i):
        image = torch.from_numpy(np.asarray(self.ds["images"][i], dtype=np.float32))[None] * 2
        labs = [torch.nn.functional.one_hot(torch.from_numpy(np.asarray(self.ds["labels"][i][a]).astype(np.int64)), 2).permute(2, 0, 1).float()
                for a in range(4)]
        return image, torch.stack(labs), np.array([0.25, 0.25, 0.25, 0.25])


def make_dataset(params: dict):
    name = params["dataset_file"]
    # the reference passes params["dataset_val_max_size"] straight to test_dataset() (evaluate_lidc_uncertainty.py:174):
    # null in the YAML = the whole test split; only when the key were absent would test_dataset's own default of 500 apply
    max_size = params.get("dataset_val_max_size", 500)
    if "synthetic" in name:
        return SyntheticLIDC(size=max_size or 8)
    if "lidc" in name:
        path = expanduservars(params.get("dataset_path", os.environ.get("LIDC_HDF5", "data_lidc.hdf5")))
        return TestLIDC(path, "test", max_size)
    raise ValueError("Unknown dataset")


# ------------------------------------------------------------------------------------------------ model
def build_from_params(params: dict, input_shapes, device) :
    """`_build_model` (ddpm/trainer.py:589-601) key mapping."""
    fce = params.get("feature_cond_encoder", {"type": "none"})
    model = build_model(
        time_steps=params["time_steps"], schedule=params["beta_schedule"],
        schedule_params=params.get("beta_schedule_params", None), cond_encoded_shape=input_shapes[0],
        input_shapes=input_shapes, backbone=params["backbone"], backbone_params=params[params["backbone"]],
        dataset_file=params["dataset_file"], step_T_sample=params.get("evaluation_vote_strategy", None),
        feature_cond_encoder=fce if fce.get("type", "none") != "none" else None).to(device)
    return model.eval()


def load_checkpoint(model, filename: str, key: str = "average_model", allow_pickle: Optional[bool] = None) -> None:
    """ignite ModelCheckpoint files are plain dicts of state_dicts (trainer.py:357-376); LIDC evaluation reads
    `average_model` (evaluate_lidc_uncertainty.py:138-143,157-161).  The file is read with torch's restricted unpickler
    (tensors, containers, numbers).  A checkpoint whose optimizer / engine entries need arbitrary classes is only read with
    the full unpickler — which executes code from the file — when the caller opts in: allow_pickle=True, or
    CCDM_ALLOW_UNSAFE_PICKLE=1 in the environment."""
    import pickle
    LOGGER.info("Loading state from %s...", filename)
    try:
        state = torch.load(filename, map_location="cpu", weights_only=True)
    except pickle.UnpicklingError as e:
        if allow_pickle is None:
            allow_pickle = os.environ.get("CCDM_ALLOW_UNSAFE_PICKLE", "") == "1"
        if not allow_pickle:
            raise RuntimeError(f"{filename} holds objects torch's safe loader rejects ({e}); if you trust the file, pass "
                               "allow_pickle=True or set CCDM_ALLOW_UNSAFE_PICKLE=1 to read it with the full unpickler") from e
        LOGGER.warning("reading %s with the full (unsafe) unpickler", filename)
        state = torch.load(filename, map_location="cpu", weights_only=False)
    sd = state[key] if key in state else state
    model.unet.load_state_dict(sd, strict=True)


def apply_sampler_options(model, params: dict) -> None:
    """Build-owned keys of the params file (absent in the reference's YAML, so an unchanged file runs the defaults):
         rng:  "philox" (default, device RNG) | "torch_cpu" (host generator in the reference's consumption order, parity mode)
         prec: "f16x3" (default) | "f32" (exact-fp32 kernels) | "f16" (OPT-IN single-pass fast mode: operands rounded to fp16, outside
               the 1e-4 parity contract — tools/fast_mode_report.py prints its error; logged as a warning)
         philox_seed, use_graph, substreams, calibrate_mode, on_range_error (layers | f32 | raise), f32_layers, slicing: DenoisingModel
         attributes of the same names."""
    from . import hip
    model.rng = str(params.get("rng", "philox"))
    prec = str(params.get("prec", "f16x3")).lower()
    if prec not in ("f16x3", "f32", "f16"):
        raise ValueError(f"prec: {prec!r} (expected 'f16x3', 'f32' or 'f16')")
    if prec == "f16":
        import logging
        logging.getLogger(__name__).warning("prec: f16 — single-pass fp16 products (~2^-11 per operand): faster, NOT the reference's arithmetic "
                                            "(outside the 1e-4 parity contract; see tools/fast_mode_report.py)")
    model.prec = {"f32": hip.PREC_F32, "f16x3": hip.PREC_F16X3, "f16": hip.PREC_F16}[prec]
    model.philox_seed = int(params.get("philox_seed", 0))
    model.use_graph = bool(params.get("use_graph", True))
    model.substreams = int(params.get("substreams", 0))          # 0 = automatic (DenoisingModel): the execution mode is measured once per geometry
    model.calibrate_mode = bool(params.get("calibrate_mode", True))    # False: the static rule (two streams from 32 x 128x128 pixels) and use_graph as set
    model.on_range_error = str(params.get("on_range_error", "layers"))
    model.f32_layers = set(params.get("f32_layers", []) or [])       # conv layers pinned to fp32 up front (tools/range_report.py --pin)
    model.slicing = str(params.get("slicing", "throughput"))
    model._fine_slices(1)                                         # validates the value


def _as_list(v) -> List[int]:
    return [int(x) for x in v] if isinstance(v, (list, tuple)) else [int(v)]


# ------------------------------------------------------------------------------------------------ evaluation
@torch.no_grad()
def eval_lidc_uncertainty(params: dict, dataset=None, device=None, init_t: Optional[int] = None, synthetic_weights_seed: Optional[int] = None,
                          model=None) -> Dict[str, object]:
    """`eval_lidc_uncertainty` + `Tester.test_step` of the reference (evaluate_lidc_uncertainty.py:89-136,164-216).

    Launched under torchrun (WORLD_SIZE > 1, BASELINE config C3) every rank walks the same batches and samples its contiguous
    shard of the B_img*S flattened batch (distributed.sample_sharded: no collective inside the T loop, one gather of the
    predictions per batch); the metrics are then computed identically on every rank.
    `model`: a ready DenoisingModel-like callable (tests inject fixed predictions); default: built from `params`."""
    from . import distributed as D
    rank, local_rank, world = D.init_from_env()
    if device is None:
        device = torch.device("cuda", local_rank) if torch.cuda.is_available() else torch.device("cuda")
    device = torch.device(device)
    dataset = dataset if dataset is not None else make_dataset(params)
    LOGGER.info("%d images in test dataset '%s'", len(dataset), params["dataset_file"])
    loader = torch.utils.data.DataLoader(dataset, batch_size=params["batch_size"], shuffle=False, num_workers=params.get("mp_loaders", 0))
    image0, labels0, _ = dataset[0]
    input_shapes = [tuple(image0.shape), tuple(labels0.shape[1:])]
    num_classes = input_shapes[1][0]
    if model is None:
        model = build_from_params(params, input_shapes, device)
        if params.get("load_from"):
            load_checkpoint(model, expanduservars(params["load_from"]))
        elif synthetic_weights_seed is not None:
            from .unet_spec import make_synthetic_state_dict
            model.unet.load_state_dict({k: torch.from_numpy(v) for k, v in make_synthetic_state_dict(model.unet.spec, synthetic_weights_seed).items()})
        apply_sampler_options(model, params)
    evaluations = _as_list(params["evaluations"])
    S = max(evaluations)
    geds, div_s, div_e, hm = (np.zeros(len(evaluations)) for _ in range(4))
    conf = torch.zeros((num_classes, num_classes), dtype=torch.int64)
    nonzero_total, n_img = 0, 0
    majority = getattr(model, "step_T_sample", None) in (None, "majority")
    for image, labels, _ in loader:                                              # Tester.test_step, :89-136
        image = image.to(device).repeat_interleave(S, dim=0)
        # x_T: uniform one-hot from the CPU generator, full batch on every rank (same seed => same draw)
        x = OneHotCategoricalBCHW(logits=torch.zeros(labels[:, 0].repeat_interleave(S, dim=0).shape)).sample().to(device)
        t = None if init_t is None else torch.as_tensor(init_t)
        if world > 1:
            # one-hot ("majority") predictions travel as uint8 class maps, probabilities as fp32
            prediction = D.sample_sharded(model, x, image, t=t, gather="index" if majority else True)
        else:
            prediction = model(x, image, **({} if t is None else {"t": t}))["diffusion_out"]
        prediction = prediction.reshape(labels.shape[0], -1, *labels.shape[2:])
        lab_idx = labels.to(device).argmax(dim=2)
        pred_idx = prediction.argmax(dim=2)
        for i, s in enumerate(evaluations):
            ged, sim_e, sim_s = M.calc_batched_generalised_energy_distance(lab_idx, pred_idx[:, :s], num_classes)
            geds[i] += ged.sum(); div_e[i] += sim_e.sum(); div_s[i] += sim_s.sum()
            lcm = int(np.lcm(s, lab_idx.shape[1]))
            hm[i] += np.sum(M.batched_hungarian_matching(lab_idx.repeat_interleave(lcm // lab_idx.shape[1], dim=1),
                                                          pred_idx[:, :s].repeat_interleave(lcm // s, dim=1), num_classes))
        # log-mean vote exactly as the reference takes it (:125): log(0) = -inf stays -inf (one-hot "majority" predictions:
        # a class any sample rejects is out; where every class is rejected by someone argmax falls to class 0)
        mean_pred = torch.log(prediction).mean(dim=1).argmax(dim=1)
        nz = torch.count_nonzero(lab_idx, dim=(2, 3)) > 0
        nonzero_total += int(nz.sum())
        for b in range(lab_idx.shape[0]):
            for a in range(lab_idx.shape[1]):
                if nz[b, a]:                                                     # y = labels[nonzero], y_pred repeated per annotation (:127-136)
                    idx = (lab_idx[b, a].reshape(-1) * num_classes + mean_pred[b].reshape(-1)).cpu()
                    conf += torch.bincount(idx, minlength=num_classes ** 2).reshape(num_classes, num_classes)
        n_img += lab_idx.shape[0]
    # ignite.metrics IoU / mIoU / DiceCoefficient on the accumulated confusion matrix (rows = y, columns = y_pred)
    cm = conf.double()
    iou = cm.diag() / (cm.sum(dim=1) + cm.sum(dim=0) - cm.diag() + 1e-15)
    dice = 2.0 * cm.diag() / (cm.sum(dim=1) + cm.sum(dim=0) + 1e-15)
    res = {"evaluations": evaluations, "GED": (geds / n_img).tolist(), "diversity_samples": (div_s / n_img).tolist(),
           "diversity_experts": float(div_e[0] / n_img), "HM_IoU": (hm / n_img).tolist(), "IoU": iou.tolist(), "mIoU": float(iou.mean()),
           "Dice": dice.tolist(), "nonzero": nonzero_total / (n_img * 4), "images": n_img, "world_size": world}
    for i, s in enumerate(evaluations):
        LOGGER.info("GED (%d): %.4g  diversity samples: %.4g  HM IoU: %.4g", s, res["GED"][i], res["diversity_samples"][i], res["HM_IoU"][i])
    return res


def eval_lidc_sampling_speed(params: dict, timesteps: Sequence[int] = (250, 200, 150, 100, 50, 25, 10), **kw) -> Dict[int, dict]:
    """Strided-step sweep: the sampler is called with t = 10000 + K (evaluate_lidc_sampling_speed.py:195-199,:103)."""
    out = {}
    for k in timesteps:
        if k > params["time_steps"]:
            continue
        LOGGER.info("Evaluate model with sampling step %d...", k)
        out[k] = eval_lidc_uncertainty(params, init_t=10000 + k, **kw)
    return out
