#!/usr/bin/env python3
"""G16: the reference's AttentionBlock at head widths other than 32, and one whole U-Net step with create_unet_openai's OWN defaults
(num_heads=1, num_head_channels=-1: one head as wide as the block — 96 and 128 channels at the LIDC widths).  Imports the reference
like tools/gen_goldens.py; weights and inputs are seeded (tests/golden_util.py), the fixture holds the reference's outputs.

    python tools/gen_goldens_heads.py        # rewrites tests/golden/g16_head_widths.npz   (build container only)
"""
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import gen_goldens as G  # noqa: E402
from tests.golden_util import HEAD_CASES, block_tensors  # noqa: E402


@torch.no_grad()
def main():
    g, shapes_meta = {}, {}
    for tag, (ch, nh, nhc, new, xs, seed) in HEAD_CASES.items():
        blk = G.ref_unet.AttentionBlock(ch, num_heads=nh, num_head_channels=nhc, use_new_attention_order=new).eval()
        shapes = {k: list(v.shape) for k, v in blk.state_dict().items()}
        shapes_meta[tag] = shapes
        w, x, _ = block_tensors(seed, shapes, xs)
        blk.load_state_dict({k: torch.from_numpy(v) for k, v in w.items()}, strict=True)
        g[tag + ".y"] = blk(torch.from_numpy(x)).numpy()
        g[tag + ".heads"] = np.array(blk.num_heads)
    # whole U-Net step, LIDC shape, the factory's default heads
    bp = dict(G.LIDC_BP, num_heads=1, num_head_channels=-1)
    m, _ = G.build((1, 128, 128), (2, 128, 128), bp, seed=16)
    rng = np.random.default_rng(1616)
    image = torch.from_numpy(rng.uniform(-1, 1, (1, 1, 128, 128)).astype(np.float32))
    idx = torch.from_numpy(rng.integers(0, 2, (1, 128, 128)))
    xt = torch.nn.functional.one_hot(idx, 2).permute(0, 3, 1, 2).float()
    out = m.unet(xt, image, None, torch.full((1,), 61.0))["diffusion_out"]
    g["unet_default_heads.out_c0"] = out[:, 0].numpy()          # (two classes: channel 1 is its complement)
    g["unet_default_heads.sum_err"] = np.array((out.sum(1) - 1).abs().max().item())
    g["unet_default_heads.t"] = np.array(61)
    G.save("g16_head_widths", **g)
    with open(os.path.join(G.OUT, "meta_heads.json"), "w") as f:
        json.dump({"block_shapes": shapes_meta}, f)


if __name__ == "__main__":
    main()
