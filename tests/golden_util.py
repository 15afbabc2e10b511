"""Shared by tools/gen_goldens.py (fixture writer) and the tests (fixture readers): seeded block weights
and inputs, so fixtures hold only seeds + the reference's outputs, not megabytes of weights."""
import numpy as np

BLOCK_CASES = {
    # tag: (kind, args, input shape [N,C,H,W], seed)
    "res32": ("res", dict(cin=32, cout=32, film=False), (2, 32, 16, 16), 301),
    "res64_32": ("res", dict(cin=64, cout=32, film=False), (2, 64, 16, 16), 302),
    "res224_96": ("res", dict(cin=224, cout=96, film=False), (2, 224, 16, 16), 303),   # 7 ch/group
    "resfilm": ("res", dict(cin=64, cout=64, film=True), (2, 64, 16, 16), 304),
    "attn96": ("attn", dict(ch=96, new=False), (2, 96, 16, 16), 305),
    "attn128": ("attn", dict(ch=128, new=False), (2, 128, 8, 8), 306),
    "attn64new": ("attn", dict(ch=64, new=True), (2, 64, 8, 8), 307),
    "down": ("down", dict(ch=32), (2, 32, 16, 16), 308),
    "up": ("up", dict(ch=64), (2, 64, 8, 8), 309),
}
EMB_DIM = 128

# AttentionBlock at the head widths the reference's factory accepts beyond the shipped num_head_channels: 32 (G16,
# tools/gen_goldens_heads.py): create_unet_openai's own defaults num_heads=1, num_head_channels=-1 make ONE head as wide as the block
# (unet_openai/__init__.py:14-15, heads rule unet.py:283-289); num_heads=4 at 96 channels gives width 24.
HEAD_CASES = {
    # tag: (channels, num_heads, num_head_channels, new attention order, input shape [N,C,H,W], seed)
    "attn96_h1": (96, 1, -1, False, (2, 96, 16, 16), 601),        # width 96, T = 256
    "attn128_h1": (128, 1, -1, False, (2, 128, 8, 8), 602),       # width 128, T = 64
    "attn96_h4": (96, 4, -1, False, (1, 96, 16, 16), 603),        # width 24
    "attn128_h1_new": (128, 1, -1, True, (1, 128, 8, 16), 604),   # width 128, new attention order, T = 128
    "attn64_h1": (64, 1, -1, False, (1, 64, 16, 16), 605),        # width 64 on the U-Net path, T = 256
    "attn160_h2": (160, 2, -1, False, (1, 160, 8, 8), 606),       # width 80 (padded to 96 inside), T = 64
}

# ResBlock(down=True / up=True) of the `resblock_updown=True` topology (G17, tools/gen_goldens_updown.py; unet.py:202-208,243-248)
UPDOWN_CASES = {
    # tag: (channels, "down" / "up", use_scale_shift_norm, input shape [N,C,H,W], seed)
    "down64": (64, "down", False, (2, 64, 16, 16), 701),
    "down32_ragged": (32, "down", False, (2, 32, 20, 12), 702),      # 10x6 output: ragged 8x8 tiles
    "down96_film": (96, "down", True, (1, 96, 16, 16), 703),
    "up64": (64, "up", False, (2, 64, 8, 8), 704),
    "up32_ragged": (32, "up", False, (2, 32, 10, 6), 705),
    "up96_film": (96, "up", True, (1, 96, 8, 8), 706),
}
UPDOWN_BP = dict(base_channels=32, channel_mult=None, attention_resolutions=[32, 16, 8], num_heads=1, num_head_channels=32,
                 softmax_output=True, resblock_updown=True)


def block_tensors(seed, shapes, x_shape):
    """Deterministic (weights dict, x, emb) for one block.  `shapes` is an ordered {key: shape}."""
    rng = np.random.default_rng(seed)
    w = {}
    for k, shp in shapes.items():
        shp = tuple(shp)
        if len(shp) == 1:
            is_gamma = k.endswith("weight") and (".0." in k or k.startswith("norm") or "norm." in k) and "emb" not in k
            v = 0.1 * rng.standard_normal(shp) + (1.0 if is_gamma else 0.0)
        else:
            v = rng.standard_normal(shp) / np.sqrt(np.prod(shp[1:]))
        w[k] = v.astype(np.float32)
    x = rng.standard_normal(x_shape).astype(np.float32)
    emb = rng.standard_normal((x_shape[0], EMB_DIM)).astype(np.float32)
    return w, x, emb


def harness_case(vote: str):
    """Fixed inputs of the LIDC-harness golden (G14): (batches, evaluations, K, predict).  batches = [(image [B,1,H,W],
    labels [B,4,K,H,W] one-hot, weights)], some annotations all-background; predict(call_index, n) -> the stand-in model's
    output for the n = B*S samples of that batch: fp32 probabilities ("confidence") or int64 one-hot ("majority")."""
    import torch
    K, H, W, evaluations = 2, 32, 32, [1, 4, 8]
    rng = np.random.default_rng(1400)
    batches = []
    for B in (2, 2, 1):
        image = torch.from_numpy(rng.uniform(-1, 1, (B, 1, H, W)).astype(np.float32))
        lab = rng.integers(0, K, (B, 4, H, W)) * (rng.random((B, 4, 1, 1)) > 0.3)          # ~30 % of the annotations are empty
        lab[:, :, : H // 2] = 0
        labels = torch.nn.functional.one_hot(torch.from_numpy(lab), K).permute(0, 1, 4, 2, 3).float()
        batches.append((image, labels, np.tile(np.array([0.25] * 4), (B, 1))))

    def predict(call, n):
        g = np.random.default_rng(9000 + call)
        logits = torch.from_numpy((2.0 * g.standard_normal((n, K, H, W))).astype(np.float32))
        logits[:, 0, : H // 2] += 3.0
        if vote == "confidence":
            return torch.softmax(logits, dim=1)
        return torch.nn.functional.one_hot(logits.argmax(1), K).permute(0, 3, 1, 2)          # int64, like max_prob_sample
    return batches, evaluations, K, predict
