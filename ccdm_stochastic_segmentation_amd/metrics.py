"""LIDC uncertainty metrics: generalised energy distance, sample/expert diversity and
Hungarian-matched IoU, as the reference computes them
(/root/reference/evaluation/evaluate_lidc_uncertainty.py:27-73; duplicates in ddpm/utils.py:129-174).

The O(B*S*S'*HW*K) part — per-class intersection/union counts of every pair of label maps — runs in a HIP
kernel (ccdm_pairwise_class_counts); the host only divides integers and solves the <= 100x100 assignment
problems with scipy, exactly as the reference does, so results are bit-identical to the reference's numpy."""
from __future__ import annotations

from typing import List, Tuple

import numpy as np
import torch

from . import hip


def pairwise_class_counts(a_idx: torch.Tensor, b_idx: torch.Tensor, num_classes: int) -> np.ndarray:
    """a_idx [B,S,...] / b_idx [B,L,...] integer class maps on the GPU -> int32 [B,S,L,K,2] (intersection, union)."""
    lib = hip.load()
    if a_idx.device.type != "cuda":
        raise hip.CcdmHipError("pairwise_class_counts needs GPU tensors (no CPU path)")
    B, S = a_idx.shape[:2]
    L = b_idx.shape[1]
    a8 = a_idx.reshape(B, S, -1).to(torch.uint8).contiguous()
    b8 = b_idx.reshape(B, L, -1).to(device=a8.device, dtype=torch.uint8).contiguous()
    HW = a8.shape[2]
    assert b8.shape[2] == HW and b8.shape[0] == B
    out = torch.empty((B, S, L, num_classes, 2), dtype=torch.int32, device=a8.device)
    hip.check(lib.ccdm_pairwise_class_counts(a8.data_ptr(), b8.data_ptr(), B, S, L, HW, num_classes, out.data_ptr(),
                                             torch.cuda.current_stream(a8.device).cuda_stream), "pairwise_class_counts")
    return out.cpu().numpy()


def _distance_from_counts(counts: np.ndarray) -> np.ndarray:
    """1 - mean IoU over the non-background classes; an empty union counts as IoU 1 (iou(): nan -> 1)."""
    inter, uni = counts[..., 0].astype(np.int64), counts[..., 1].astype(np.int64)
    with np.errstate(divide="ignore", invalid="ignore"):
        iou = inter / uni
    iou[np.isnan(iou)] = 1.0
    return 1 - iou[..., 1:].mean(-1)


def batched_distance(x_idx: torch.Tensor, y_idx: torch.Tensor, num_classes: int) -> np.ndarray:
    """[B,S,...], [B,L,...] -> [B,S,L] distances (reference `batched_distance`, :33-39)."""
    return _distance_from_counts(pairwise_class_counts(x_idx, y_idx, num_classes))


def calc_batched_generalised_energy_distance(samples_dist_0: torch.Tensor, samples_dist_1: torch.Tensor,
                                             num_classes: int) -> Tuple[np.ndarray, np.ndarray, np.ndarray]:
    """GED, diversity of dist_0, diversity of dist_1 — per image (reference :42-54)."""
    cross = np.mean(batched_distance(samples_dist_0, samples_dist_1, num_classes), axis=(1, 2))
    diversity_0 = np.mean(batched_distance(samples_dist_0, samples_dist_0, num_classes), axis=(1, 2))
    diversity_1 = np.mean(batched_distance(samples_dist_1, samples_dist_1, num_classes), axis=(1, 2))
    return 2 * cross - diversity_0 - diversity_1, diversity_0, diversity_1


def batched_hungarian_matching(samples_dist_0: torch.Tensor, samples_dist_1: torch.Tensor, num_classes: int) -> List[float]:
    """Hungarian-matched IoU per image (reference :57-73)."""
    from scipy.optimize import linear_sum_assignment
    cost = batched_distance(samples_dist_0, samples_dist_1, num_classes)
    return [float((1 - cost[i])[linear_sum_assignment(cost[i])].mean()) for i in range(cost.shape[0])]
