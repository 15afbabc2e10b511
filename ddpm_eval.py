#!/usr/bin/env python3
"""Evaluation entry point, same command line as the reference's script of this name (/root/reference/ddpm_eval.py:28-47):

    python ddpm_eval.py [params_<anything>.yml]

`dataset_file` in the YAML selects the evaluator: "...lidc_sampling_speed" -> the T-sweep timing run,
"...lidc" -> GED / HM-IoU over the LIDC test split.  Everything runs through the MI355X sampler; the result
dictionary is also printed as one JSON line so scripts can pick it up."""
import json
import logging
import os
import random
import sys

import numpy as np
import torch
import yaml

from ccdm_stochastic_segmentation_amd import evaluation

DEFAULT_PARAMS = "params_eval.yml"
SEED = 0


def _seed_everything(seed):
    # the reference seeds python, numpy and torch (cpu + every gpu) with 0 before anything else runs
    os.environ["PYTHONHASHSEED"] = str(seed)
    for seeder in (random.seed, lambda s: np.random.seed(s % 2 ** 32), torch.manual_seed, torch.cuda.manual_seed_all):
        seeder(seed)


def _params_path(argv):
    # only an argument that looks like a params file overrides the default, as in the reference
    if len(argv) == 2 and "params_" in argv[1]:
        print(f"Overriding params file with {argv[1]}...")
        return argv[1]
    return DEFAULT_PARAMS


def _pick_evaluator(dataset_file):
    """-> (evaluator, dataset_file to hand it).  Order matters: the speed sweep's name contains 'lidc'."""
    if "lidc_sampling_speed" in dataset_file:
        return evaluation.eval_lidc_sampling_speed, dataset_file.replace("lidc_sampling_speed", "lidc")
    if "lidc" in dataset_file:
        return evaluation.eval_lidc_uncertainty, dataset_file
    if "cityscapes" in dataset_file:
        raise NotImplementedError("the Cityscapes evaluator is broken on the reference's main branch "
                                  "(SURVEY §2 #22) and is out of scope")
    raise ValueError("Unknown dataset")


def main(argv):
    logging.basicConfig(level=logging.INFO, format="%(asctime)s [%(name)s] %(message)s")
    _seed_everything(SEED)
    with open(_params_path(argv)) as fh:
        params = yaml.safe_load(fh)
    evaluator, params["dataset_file"] = _pick_evaluator(params["dataset_file"])
    synthetic = "synthetic" in params["dataset_file"]
    result = evaluator(params, synthetic_weights_seed=SEED if synthetic else None)
    print(json.dumps(result))


if __name__ == "__main__":
    main(sys.argv)
