#!/usr/bin/env python3
"""Entry point with the reference's name, arguments and dispatch (/root/reference/ddpm_eval.py:28-47):

    python ddpm_eval.py [params_eval.yml]

runs the LIDC uncertainty evaluation (or the sampling-speed sweep when dataset_file contains
'lidc_sampling_speed') through the MI355X sampler."""
import json
import logging
import os
import random
import sys

import numpy as np
import torch
import yaml

from ccdm_stochastic_segmentation_amd.evaluation import eval_lidc_sampling_speed, eval_lidc_uncertainty


def set_seeds(seed: int):
    random.seed(seed)
    os.environ["PYTHONHASHSEED"] = str(seed)
    np.random.seed(seed % 2 ** 32)
    torch.manual_seed(seed)
    torch.cuda.manual_seed_all(seed)


def main(argv):
    logging.basicConfig(level=logging.INFO, format="%(asctime)s [%(name)s] %(message)s")
    set_seeds(0)
    params_file = "params_eval.yml"
    if len(argv) == 2 and "params_" in argv[1]:
        params_file = argv[1]
        print(f"Overriding params file with {params_file}...")
    with open(params_file, "r") as f:
        params = yaml.safe_load(f)
    if "lidc_sampling_speed" in params["dataset_file"]:
        params["dataset_file"] = params["dataset_file"].replace("lidc_sampling_speed", "lidc")
        res = eval_lidc_sampling_speed(params, synthetic_weights_seed=0 if "synthetic" in params["dataset_file"] else None)
    elif "lidc" in params["dataset_file"]:
        res = eval_lidc_uncertainty(params, synthetic_weights_seed=0 if "synthetic" in params["dataset_file"] else None)
    elif "cityscapes" in params["dataset_file"]:
        raise NotImplementedError("the Cityscapes evaluator is broken on the reference's main branch (SURVEY §2 #22) and is out of scope")
    else:
        raise ValueError("Unknown dataset")
    print(json.dumps(res))


if __name__ == "__main__":
    main(sys.argv)
