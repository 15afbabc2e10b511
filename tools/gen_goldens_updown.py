#!/usr/bin/env python3
"""G17: the `resblock_updown=True` topology — the reference's ResBlock(down=True) / ResBlock(up=True) (unet.py:202-208, :243-248) as
single blocks, one whole U-Net step of the LIDC-shaped network built with resblock_updown=True, and a short seeded sampling walk
through it.  Imports the reference like tools/gen_goldens.py; weights and inputs are seeded (tests/golden_util.py), the fixture holds
the reference's outputs.

    python tools/gen_goldens_updown.py        # rewrites tests/golden/g17_resblock_updown.npz   (build container only)
"""
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import gen_goldens as G  # noqa: E402
from tests.golden_util import UPDOWN_BP, UPDOWN_CASES, block_tensors  # noqa: E402


@torch.no_grad()
def main():
    g, shapes_meta = {}, {}
    for tag, (ch, mode, film, xs, seed) in UPDOWN_CASES.items():
        blk = G.ref_unet.ResBlock(ch, 128, 0, out_channels=ch, use_scale_shift_norm=film, down=mode == "down", up=mode == "up").eval()
        shapes = {k: list(v.shape) for k, v in blk.state_dict().items()}
        shapes_meta[tag] = shapes
        w, x, emb = block_tensors(seed, shapes, xs)
        blk.load_state_dict({k: torch.from_numpy(v) for k, v in w.items()}, strict=True)
        g[tag + ".y"] = blk(torch.from_numpy(x), torch.from_numpy(emb)).numpy()
    # whole U-Net step, LIDC shape
    m, spec = G.build((1, 128, 128), (2, 128, 128), dict(UPDOWN_BP), seed=17)
    keys = [[k, list(v.shape)] for k, v in m.unet.state_dict().items()]
    rng = np.random.default_rng(1717)
    image = torch.from_numpy(rng.uniform(-1, 1, (1, 1, 128, 128)).astype(np.float32))
    idx = torch.from_numpy(rng.integers(0, 2, (1, 128, 128)))
    xt = torch.nn.functional.one_hot(idx, 2).permute(0, 3, 1, 2).float()
    taps = {}
    hooks = [m.unet.input_blocks[3].register_forward_hook(lambda _m, _i, o: taps.__setitem__("input_blocks.3", o)),
             m.unet.output_blocks[2].register_forward_hook(lambda _m, _i, o: taps.__setitem__("output_blocks.2", o))]
    out = m.unet(xt, image, None, torch.full((1,), 61.0))["diffusion_out"]
    for h in hooks:
        h.remove()
    g["unet.out_c0"] = out[:, 0].numpy()          # (two classes: channel 1 is its complement)
    g["unet.t"] = np.array(61)
    # the first down block's and the first up block's outputs: per-channel means (position of a mismatch without megabytes)
    for k, v in taps.items():
        g[f"unet.tap.{k}.mean_hw"] = v.mean((2, 3)).numpy()
        g[f"unet.tap.{k}.shape"] = np.array(v.shape)
    # a seeded 6-step strided walk from x_T (t = 10006: diffusion_denoising.py:178-187), like G7: every x_t the reference draws, the
    # network output on a lattice, the final probabilities
    torch.manual_seed(42)
    x = G.OneHotCategoricalBCHW(logits=torch.zeros(1, 2, 128, 128)).sample()
    rec = []
    orig = m.diffusion.theta_post_prob

    def spy(xt_, x0_, t_):
        rec.append((int(t_[0]), xt_.argmax(1).numpy().copy(), x0_.numpy().copy()))
        return orig(xt_, x0_, t_)
    m.diffusion.theta_post_prob = spy
    out = m(x, image, t=torch.as_tensor(10006))["diffusion_out"]
    m.diffusion.theta_post_prob = orig
    g["walk.t_values"] = np.array([r[0] for r in rec])
    g["walk.xT"] = G.packbits(x.argmax(1).numpy())
    for j, r in enumerate(rec):
        g[f"walk.xt_{j}"] = G.packbits(r[1])
        g[f"walk.x0pred0_{j}"] = r[2][:, 0, ::16, ::16]
    g["walk.out_c0"] = out[:, 0].numpy()
    G.save("g17_resblock_updown", **g)
    with open(os.path.join(G.OUT, "meta_updown.json"), "w") as f:
        json.dump({"block_shapes": shapes_meta, "unet_keys": keys, "unet_params": int(sum(p.numel() for p in m.unet.parameters()))}, f)


if __name__ == "__main__":
    main()
