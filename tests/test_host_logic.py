"""CPU-only tests: host logic of the product (spec, schedules, boundary classes), the C-ABI library's
symbol table and host-side packer, and the multi-process sharding path on gloo."""
import ctypes
import json
import os
import re
import subprocess
import sys

import numpy as np
import pytest
import torch

import ccdm_stochastic_segmentation_amd as P
from ccdm_stochastic_segmentation_amd import hip, models
from ccdm_stochastic_segmentation_amd.distributed import shard_range

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIDC_BP = dict(base_channels=32, channel_mult=None, attention_resolutions=[32, 16, 8], num_heads=1,
               num_head_channels=32, softmax_output=True)


def lidc_model(vote="confidence"):
    return P.build_model(250, "cosine", {"s": 0.008}, [(1, 128, 128), (2, 128, 128)], (1, 128, 128), "unet_openai", LIDC_BP,
                         "datasets.lidc", vote, None)


def test_library_exports_every_declared_symbol():
    """The C-ABI library loads here (no GPU) and exports exactly what include/ccdm_hip.h declares."""
    import __graft_entry__ as g
    g.build()
    lib = hip.load()
    hdr = open(os.path.join(ROOT, "include", "ccdm_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(ccdm_[a-z0-9_]+)\s*\(", hdr))
    assert declared == set(hip.SIGNATURES), declared ^ set(hip.SIGNATURES)
    for name in declared:
        assert hasattr(lib, name), name
    assert lib.ccdm_version() == hip.ABI_VERSION == 2
    assert ctypes.sizeof(hip.ConvArgs) % 8 == 0 and ctypes.sizeof(hip.PostArgs) % 8 == 0


def test_struct_layout_matches_header():
    """ctypes mirrors of the two argument structs agree with the C compiler's layout."""
    src = r'''
    #include <stdio.h>
    #include <stddef.h>
    #include "ccdm_hip.h"
    int main(){
      printf("%zu %zu %zu %zu %zu %zu %zu %zu\n", sizeof(ccdm_conv_args), offsetof(ccdm_conv_args, gamma), offsetof(ccdm_conv_args, w),
             offsetof(ccdm_conv_args, emb_row_of_sample), offsetof(ccdm_conv_args, out), offsetof(ccdm_conv_args, out_slices),
             offsetof(ccdm_conv_args, SC1), offsetof(ccdm_conv_args, skip_w));
      printf("%zu %zu %zu %zu %zu %zu %zu\n", sizeof(ccdm_post_args), offsetof(ccdm_post_args, step_table), offsetof(ccdm_post_args, philox_seed),
             offsetof(ccdm_post_args, xin_stride), offsetof(ccdm_post_args, posterior_out), offsetof(ccdm_post_args, noise_row0),
             offsetof(ccdm_post_args, range_flag));
      return 0; }'''
    import tempfile
    with tempfile.TemporaryDirectory() as d:
        open(os.path.join(d, "t.c"), "w").write(src)
        subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), os.path.join(d, "t.c"), "-o", os.path.join(d, "t")])
        out = subprocess.check_output([os.path.join(d, "t")]).decode().split()
    A, B = hip.ConvArgs, hip.PostArgs
    mine = [ctypes.sizeof(A), A.gamma.offset, A.w.offset, A.emb_row_of_sample.offset, A.out.offset, A.out_slices.offset,
            A.SC1.offset, A.skip_w.offset,
            ctypes.sizeof(B), B.step_table.offset, B.philox_seed.offset, B.xin_stride.offset, B.posterior_out.offset,
            B.noise_row0.offset, B.range_flag.offset]
    assert [int(v) for v in out] == mine


def test_pack_conv_weight_layout():
    rng = np.random.default_rng(0)
    w = rng.standard_normal((40, 6, 3, 3)).astype(np.float32)       # Cout 40 -> 2 n-tiles, Cin 6 -> padded to 32
    buf = hip.pack_conv_weight(w, 3).view(np.float32).reshape(9, 16, 2, 64)
    for tap, kp, nt, l in [(0, 0, 0, 0), (4, 2, 1, 7), (8, 1, 0, 45), (3, 2, 1, 63), (5, 10, 0, 3)]:
        co, ci = nt * 32 + (l & 31), 2 * kp + (l >> 5)
        exp = w[co, ci, tap // 3, tap % 3] if (co < 40 and ci < 6) else 0.0
        assert buf[tap, kp, nt, l] == exp
    assert hip.load().ccdm_conv_slices(128, 128, 1, 3) == 12
    assert hip.load().ccdm_conv_slices(16, 16, 1, 3) == 2
    assert hip.load().ccdm_conv_slices(8, 8, 1, 3) == 1


def test_missing_library_is_a_hard_error(monkeypatch, tmp_path):
    monkeypatch.setattr(hip, "_lib", None)
    monkeypatch.setattr(hip, "LIB_PATH", str(tmp_path / "nope.so"))
    monkeypatch.setenv("CCDM_NO_AUTOBUILD", "1")
    with pytest.raises(hip.CcdmHipError, match="no CPU fallback"):
        hip.load()


def test_no_cpu_path():
    m = lidc_model().eval()
    x = torch.zeros(1, 2, 128, 128); x[:, 0] = 1
    with pytest.raises(hip.CcdmHipError, match="no CPU path"):
        m(x, torch.zeros(1, 1, 128, 128))


def test_product_does_not_import_oracle():
    pkg = os.path.join(ROOT, "ccdm_stochastic_segmentation_amd")
    for fn in os.listdir(pkg):
        if fn.endswith(".py"):
            assert "oracle" not in open(os.path.join(pkg, fn)).read().replace("# oracle", ""), fn


def test_state_dict_layout_and_strict_load(golden):
    m = lidc_model()
    keys = [[k, list(v.shape)] for k, v in m.unet.state_dict().items()]
    assert keys == golden.meta["lidc_keys"] and len(keys) == 398
    assert m.unet.spec.num_params() == 5699138
    assert list(m.state_dict())[:3] == ["diffusion.betas", "diffusion.alphas", "diffusion.cumalphas"]
    sd = {k: torch.from_numpy(v) for k, v in P.make_synthetic_state_dict(m.unet.spec, 0).items()}
    v0 = m.unet._weights_version
    m.unet.load_state_dict(sd, strict=True)
    assert m.unet._weights_version > v0
    assert torch.equal(m.unet.state_dict()["out.2.weight"], sd["out.2.weight"])
    bad = dict(sd); bad.pop("out.2.bias")
    with pytest.raises(RuntimeError, match="Missing key"):
        m.unet.load_state_dict(bad, strict=True)
    bad = dict(sd); bad["extra.weight"] = torch.zeros(1)
    with pytest.raises(RuntimeError, match="Unexpected key"):
        m.unet.load_state_dict(bad, strict=True)
    # an ignite-style checkpoint dict {"model":..., "average_model":...} loads the same way
    ckpt = {"model": sd, "average_model": sd, "optimizer": {}, "engine": {}}
    m.unet.load_state_dict(ckpt["average_model"], strict=True)
    assert m.time_steps == 250 and m.diffusion.num_classes == 2


def test_builder_contract():
    with pytest.raises(NotImplementedError, match="backbone resnet50"):
        P.build_model(250, "cosine", None, [(1, 128, 128), (2, 128, 128)], None, "resnet50", {}, "x")
    with pytest.raises(ValueError, match="unsupported image size"):
        P.build_model(250, "cosine", None, [(1, 100, 100), (2, 100, 100)], None, "unet_openai", LIDC_BP, "x")
    with pytest.raises(ValueError, match="GroupNorm"):          # C5 at base 32: GroupNorm(32, 16)
        P.build_model(250, "cosine", None, [(3, 512, 1024), (20, 512, 1024)], None, "unet_openai", LIDC_BP, "x")
    m = lidc_model()
    m.train()
    with pytest.raises(ValueError, match="'t' needs to be a Tensor"):
        m(torch.zeros(1, 2, 128, 128), torch.zeros(1, 1, 128, 128))
    m.eval()
    with pytest.raises(AttributeError, match="guidance_scale_weights"):
        m(torch.zeros(1, 2, 128, 128), torch.zeros(1, 1, 128, 128), label_ref_logits=torch.zeros(1, 2, 128, 128))


def test_schedules_and_steps_match_reference_golden(golden):
    g = golden["g1_schedules"]
    for name, sched, T, sp in [("cosine250", "cosine", 250, {"s": 0.008}), ("cosine1000", "cosine", 1000, None), ("linear250", "linear", 250, None)]:
        d = P.DiffusionModel(sched, T, 2, sp)
        assert np.array_equal(d.betas.numpy(), g[name + "_betas"])
        assert np.array_equal(d.alphas.numpy(), g[name + "_alphas"])
        assert np.array_equal(d.cumalphas.numpy(), g[name + "_cumalphas"])
    for T, K in [(250, 10), (250, 25), (250, 200), (250, 150), (250, 100), (250, 50), (1000, 16)]:
        assert P.step_values(T, 10000 + K) == list(g[f"steps_T{T}_K{K}"])
    assert P.step_values(250, None) == list(range(250, 0, -1))
    assert P.step_values(250, 3) == [3, 2, 1]
    with pytest.raises(AssertionError):
        P.step_values(250, 10000 + 251)
    d = P.DiffusionModel("cosine", 250, 2)
    assert d.posterior_coeffs(1) == (0.0, 1.0)
    assert d.posterior_coeffs(250) == (float(d.alphas[249]), float(d.cumalphas[248]))


def test_one_hot_categorical_bchw_matches_reference_golden(golden):
    g = golden["g6_sampler"]
    for K in (2, 20):
        probs = torch.from_numpy(g[f"K{K}_probs"])
        torch.manual_seed(6)
        d = P.OneHotCategoricalBCHW(probs=probs)
        s = d.sample()
        assert np.array_equal(s.argmax(1).numpy(), g[f"K{K}_idx"])
        assert s.dtype == torch.float32 and s.shape == probs.shape
        assert np.array_equal(d.max_prob_sample().numpy(), g[f"K{K}_maxprob"])
        assert np.array_equal(d.prob_sample().numpy(), g[f"K{K}_probsample"])
    torch.manual_seed(42)
    x = P.OneHotCategoricalBCHW(logits=torch.zeros(3, 2, 8, 8)).sample()
    assert np.array_equal(x.argmax(1).numpy(), g["xT_seed42_K2"])
    assert tuple(x.stride()) == (128, 1, 16, 2)                       # BCHW view of channels-last memory
    torch.manual_seed(42)
    x = P.OneHotCategoricalBCHW(logits=torch.zeros(2, 20, 8, 8)).sample()
    assert np.array_equal(x.argmax(1).numpy(), g["xT_seed42_K20"])
    with pytest.raises(ValueError):
        P.OneHotCategoricalBCHW(probs=torch.ones(3))
    with pytest.raises(ValueError):
        P.OneHotCategoricalBCHW()


def test_shard_range_partitions():
    for n in (0, 1, 7, 64, 65):
        for w in (1, 2, 3, 8):
            rs = [shard_range(n, r, w) for r in range(w)]
            assert rs[0][0] == 0 and rs[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(rs, rs[1:]))
            sizes = [b - a for a, b in rs]
            assert max(sizes) - min(sizes) <= 1


WORKER = r'''
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, os.environ["CCDM_ROOT"])
from ccdm_stochastic_segmentation_amd.distributed import init_from_env, sample_sharded, shard_range
rank, local, world = init_from_env("gloo")
class Fake:                      # stands in for DenoisingModel: output encodes (global sample index, offset seen)
    sample_offset = 0; noise_slice = None
    def __call__(self, x, cond, feat=None, **kw):
        n = x.shape[0]
        assert self.noise_slice == (7, self.sample_offset)
        out = x.clone()
        out[:, 0, 0, 0] = torch.arange(n, dtype=x.dtype) + self.sample_offset
        out[:, 1, 0, 0] = cond[:, 0, 0, 0]
        return {"diffusion_out": out}
N = 7
x = torch.zeros(N, 2, 4, 4); cond = torch.arange(N, dtype=torch.float32).reshape(N, 1, 1, 1).expand(N, 1, 4, 4).contiguous() * 10
out = sample_sharded(Fake(), x, cond)
assert out.shape == (N, 2, 4, 4)
assert torch.equal(out[:, 0, 0, 0], torch.arange(N, dtype=torch.float32)), out[:, 0, 0, 0]
assert torch.equal(out[:, 1, 0, 0], torch.arange(N, dtype=torch.float32) * 10)
lo, hi = shard_range(N, rank, world)
loc = sample_sharded(Fake(), x, cond, gather=False)
assert loc.shape[0] == hi - lo
dist.barrier(); dist.destroy_process_group()
sys.stdout.write(f"rank{rank}ok\n"); sys.stdout.flush()
'''


def test_sharded_sampling_world_size_2_gloo(tmp_path):
    """N > 1 path on CPU: two processes, gloo, ragged shard sizes (7 samples over 2 ranks)."""
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    env = dict(os.environ, CCDM_ROOT=ROOT, MASTER_ADDR="127.0.0.1", MASTER_PORT="29611")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
                        "--master-port", "29611", str(script)], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "rank0ok" in r.stdout and "rank1ok" in r.stdout


def test_evaluation_harness_host_side():
    from ccdm_stochastic_segmentation_amd import evaluation as E
    ds = E.SyntheticLIDC(size=3)
    img, labs, w = ds[1]
    assert img.shape == (1, 128, 128) and labs.shape == (4, 2, 128, 128) and img.abs().max() <= 1
    assert torch.equal(labs.sum(1), torch.ones(4, 128, 128)) and list(w) == [0.25] * 4
    img2, labs2, _ = ds[1]
    assert torch.equal(img, img2) and torch.equal(labs, labs2)            # deterministic
    assert E._as_list(8) == [8] and E._as_list([1, 4]) == [1, 4]         # `evaluations: 8` (shipped yml) or a list
    assert isinstance(E.make_dataset({"dataset_file": "synthetic.lidc", "dataset_val_max_size": 5}), E.SyntheticLIDC)
    with pytest.raises(ValueError, match="Unknown dataset"):
        E.make_dataset({"dataset_file": "ade20k"})
    import yaml
    p = yaml.safe_load(open(os.path.join(ROOT, "params_eval.yml")))
    for key in ("time_steps", "beta_schedule", "beta_schedule_params", "backbone", "unet_openai", "feature_cond_encoder",
                "evaluation_vote_strategy", "evaluations", "batch_size", "load_from", "dataset_file"):
        assert key in p, key
    m = E.build_from_params(p, [(1, 128, 128), (2, 128, 128)], "cpu")
    assert m.step_T_sample == "confidence" and m.time_steps == 250


def test_ddpm_eval_dispatch(tmp_path, monkeypatch):
    import yaml
    import ddpm_eval
    p = yaml.safe_load(open(os.path.join(ROOT, "params_eval.yml")))
    p["dataset_file"] = "datasets.ade20k"
    f = tmp_path / "params_x.yml"
    yaml.safe_dump(p, open(f, "w"))
    with pytest.raises(ValueError, match="Unknown dataset"):
        ddpm_eval.main(["ddpm_eval.py", str(f)])
    p["dataset_file"] = "datasets.cityscapes"
    yaml.safe_dump(p, open(f, "w"))
    with pytest.raises(NotImplementedError):
        ddpm_eval.main(["ddpm_eval.py", str(f)])
