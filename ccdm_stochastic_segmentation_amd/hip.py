"""ctypes binding of libccdm_hip.so (the C ABI declared in include/ccdm_hip.h).

There is no CPU fallback: every compute entry point goes to the HIP library, and a missing library is a
hard error (``load()`` raises).  Importing this module does not load anything, so host-only logic
(spec, schedules, packing through the library's host functions) works on a box without a GPU.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from typing import Optional

_HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(_HERE)
LIB_PATH = os.environ.get("CCDM_LIB") or os.path.join(_HERE, "libccdm_hip.so")      # CCDM_LIB: A/B two builds on one GPU box
CSRC = os.path.join(_HERE, "csrc")
SOURCES = ["ccdm_conv.hip", "ccdm_conv_ks.hip", "ccdm_upconv.hip", "ccdm_stem.hip", "ccdm_head.hip", "ccdm_conv1x1.hip", "ccdm_misc.hip", "ccdm_attention.hip", "ccdm_attn_block.hip", "ccdm_sampler.hip", "ccdm_metrics.hip", "ccdm_range.hip", "ccdm_resample.hip", "ccdm_engine.hip"]
# CCDM_EXPERIMENTS=1 builds add the measured-and-rejected kernels of tools/experiments/ and the environment switches the A/B tools use
# (exp_env in ccdm_common.h); the shipped library contains neither
EXPERIMENT_SOURCES = [os.path.join(ROOT, "tools", "experiments", "ccdm_conv_pc.hip"), os.path.join(ROOT, "tools", "experiments", "ccdm_attention_split.hip")]
# -amdgpu-mfma-vgpr-form: MFMA accumulators stay in the (unified) VGPR file.  The default heuristic parks them in AccVGPRs and pays a
# v_accvgpr_read/_write for every vector op that touches a score or an output accumulator: 240 extra instructions per key tile in
# the attention kernels (2066 in ccdm_attention.hip, 576 in ccdm_attn_block.hip; the conv kernels have none either way).
HIPCC_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-mllvm", "-amdgpu-mfma-vgpr-form=1", "-fPIC", "-shared"]

ACT_NONE, ACT_SILU = 0, 1
RESAMPLE_AVGPOOL2, RESAMPLE_NEAREST_UP2 = 0, 1      # CCDM_RESAMPLE_*
ATTENTION_FORCE_VALU = 256          # bit 8 of ccdm_attention's `order`: the vector-pipe kernel (plain fp32 FMAs) instead of the matrix cores
ATTENTION_VALU_WIDTHS = (4, 8, 12, 16, 24, 32, 48, 64)      # head widths that kernel is instantiated for
DIAG_GENERAL_KERNEL = 2048 << 8     # CCDM_DIAG_GENERAL_KERNEL: OR into ConvArgs.prec to bypass the specialised conv kernels (parity tests)
PREC_F32, PREC_F16X3 = 0, 1
PREC_F16 = 2             # CCDM_PREC_F16: the OPT-IN single-pass fast mode (one fp16 MFMA per product, ~2^-11 per operand) — outside the parity contract
STEP_SAMPLE, STEP_LAST_CONFIDENCE, STEP_LAST_MAJORITY, STEP_LAST_KEEP, STEP_SOFTMAX_ONLY = 0, 1, 2, 3, 4
STATS_MAX_SLICES = 64       # CCDM_STATS_MAX_SLICES: what a GroupNorm consumer reads
F16X3_LIMIT = 4094.0        # CCDM_F16X3_LIMIT: the fp16 split is exact for staged |a| below this
STATS_FOLD_SLICES = 16      # CCDM_STATS_FOLD_SLICES: what the engine folds a larger slice count to
ABI_VERSION = 10         # CCDM_ABI_VERSION of include/ccdm_hip.h
MAX_CLASSES = 255        # CCDM_MAX_CLASSES: x_t is a uint8 class index (K <= 32 in registers, more through LDS rows)
POST_DIAG_MANY = 256     # CCDM_POST_DIAG_MANY: OR into PostArgs.softmax to run the many-class epilogue kernel at any K (parity tests)


class ConvArgs(C.Structure):
    _fields_ = [
        ("in0", C.c_void_p), ("in1", C.c_void_p), ("C0", C.c_int32), ("C1", C.c_int32),
        ("stats0", C.c_void_p), ("stats1", C.c_void_p), ("slices0", C.c_int32), ("slices1", C.c_int32),
        ("gamma", C.c_void_p), ("beta", C.c_void_p),
        ("eps", C.c_float), ("act", C.c_int32),
        ("film", C.c_int32), ("film_off", C.c_int32),
        ("N", C.c_int32), ("Hin", C.c_int32), ("Win", C.c_int32), ("Hout", C.c_int32), ("Wout", C.c_int32),
        ("ksize", C.c_int32), ("stride", C.c_int32), ("up", C.c_int32),
        ("w", C.c_void_p), ("bias", C.c_void_p), ("Cout", C.c_int32), ("prec", C.c_int32),
        ("emb_table", C.c_void_p), ("emb_stride", C.c_int32), ("emb_off", C.c_int32),
        ("emb_row_of_sample", C.c_void_p),
        ("step_ptr", C.c_void_p),
        ("resid", C.c_void_p),
        ("out", C.c_void_p),
        ("out_stats", C.c_void_p), ("out_slices", C.c_int32),
        ("skip0", C.c_void_p), ("skip1", C.c_void_p), ("SC0", C.c_int32), ("SC1", C.c_int32),
        ("skip_w", C.c_void_p),
        ("fine_slices", C.c_int32),
    ]


class PostArgs(C.Structure):
    _fields_ = [
        ("head", C.c_void_p), ("softmax", C.c_int32), ("head_stride", C.c_int32),
        ("xt", C.c_void_p),
        ("N", C.c_int32), ("HW", C.c_int32), ("K", C.c_int32),
        ("step_table", C.c_void_p), ("step_ptr", C.c_void_p),
        ("noise", C.c_void_p), ("noise_step_stride", C.c_int64),
        ("philox_seed", C.c_uint64), ("sample_offset", C.c_uint32),
        ("xt_next", C.c_void_p),
        ("xin", C.c_void_p), ("xin_stride", C.c_int32),
        ("out_probs", C.c_void_p),
        ("out_onehot", C.c_void_p),
        ("posterior_out", C.c_void_p),
        ("noise_row0", C.c_int32),
        ("range_flag", C.c_void_p),
        ("run", C.c_void_p),
    ]


POST_RUN_BYTES = 56      # sizeof(ccdm_post_run): the device-resident per-run fields of the epilogue


class AttnBlockArgs(C.Structure):
    _fields_ = [
        ("x", C.c_void_p),
        ("stats", C.c_void_p), ("slices", C.c_int32),
        ("gamma", C.c_void_p), ("beta", C.c_void_p), ("eps", C.c_float),
        ("wqkv", C.c_void_p), ("bqkv", C.c_void_p),
        ("out", C.c_void_p),
        ("N", C.c_int32), ("T", C.c_int32), ("C", C.c_int32), ("heads", C.c_int32),
    ]


class ResampleArgs(C.Structure):
    _fields_ = [
        ("in_", C.c_void_p), ("C", C.c_int32),
        ("stats", C.c_void_p), ("slices", C.c_int32),
        ("gamma", C.c_void_p), ("beta", C.c_void_p), ("eps", C.c_float), ("act", C.c_int32),
        ("N", C.c_int32), ("Hin", C.c_int32), ("Win", C.c_int32), ("mode", C.c_int32),
        ("out_act", C.c_void_p), ("out_raw", C.c_void_p),
    ]


class StemArgs(C.Structure):
    _fields_ = [
        ("xt", C.c_void_p), ("xin", C.c_void_p), ("Cs", C.c_int32), ("K", C.c_int32),
        ("w", C.c_void_p), ("bias", C.c_void_p),
        ("N", C.c_int32), ("H", C.c_int32), ("W", C.c_int32), ("Cout", C.c_int32),
        ("out", C.c_void_p), ("out_stats", C.c_void_p), ("out_slices", C.c_int32),
    ]


class HeadArgs(C.Structure):
    _fields_ = [
        ("x", C.c_void_p), ("stats", C.c_void_p), ("slices", C.c_int32),
        ("gamma", C.c_void_p), ("beta", C.c_void_p), ("eps", C.c_float),
        ("w", C.c_void_p), ("bias", C.c_void_p),
        ("N", C.c_int32), ("H", C.c_int32), ("W", C.c_int32), ("C", C.c_int32), ("K", C.c_int32),
        ("logits_out", C.c_void_p),
    ]


# name -> (restype, argtypes); every symbol include/ccdm_hip.h declares
SIGNATURES = {
    "ccdm_version": (C.c_int, []),
    "ccdm_last_error_string": (C.c_char_p, []),
    "ccdm_gn_stats": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "ccdm_conv_slices": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int]),
    "ccdm_conv_slices_ex": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]),
    "ccdm_conv_out_slices": (C.c_int, [C.POINTER(ConvArgs)]),
    "ccdm_upconv_supported": (C.c_int, [C.c_int, C.c_int, C.c_int]),
    "ccdm_upconv_slices": (C.c_int, [C.c_int, C.c_int]),
    "ccdm_pack_upconv_weight": (C.c_size_t, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "ccdm_conv2d": (C.c_int, [C.POINTER(ConvArgs), C.c_void_p]),
    "ccdm_conv_input_absmax": (C.c_int, [C.POINTER(ConvArgs), C.c_void_p, C.c_void_p]),
    "ccdm_engine_input_absmax": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]),
    "ccdm_stats_fold": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "ccdm_norm_qkv_attention_supported": (C.c_int, [C.c_int, C.c_int, C.c_int]),
    "ccdm_norm_qkv_attention": (C.c_int, [C.POINTER(AttnBlockArgs), C.c_void_p]),
    "ccdm_engine_add_norm_qkv_attention": (C.c_int, [C.c_void_p, C.POINTER(AttnBlockArgs)]),
    "ccdm_engine_add_stats_fold": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "ccdm_engine_add_resample": (C.c_int, [C.c_void_p, C.POINTER(ResampleArgs)]),
    "ccdm_engine_add_stem": (C.c_int, [C.c_void_p, C.POINTER(StemArgs)]),
    "ccdm_engine_add_head_posterior": (C.c_int, [C.c_void_p, C.POINTER(HeadArgs)]),
    "ccdm_head_posterior_supported": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]),
    "ccdm_pack_head_weight": (C.c_size_t, [C.c_void_p, C.c_int, C.c_int, C.c_void_p]),
    "ccdm_head_posterior": (C.c_int, [C.POINTER(HeadArgs), C.POINTER(PostArgs), C.c_void_p]),
    "ccdm_stem_conv_supported": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]),
    "ccdm_pack_stem_weight": (C.c_size_t, [C.c_void_p, C.c_int, C.c_int, C.c_void_p]),
    "ccdm_stem_conv": (C.c_int, [C.POINTER(StemArgs), C.c_void_p]),
    "ccdm_resample": (C.c_int, [C.POINTER(ResampleArgs), C.c_void_p]),
    "ccdm_pack_conv_weight": (C.c_size_t, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "ccdm_pack_conv_weight_ex": (C.c_size_t, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "ccdm_attention": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "ccdm_time_table": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                  C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]),
    "ccdm_posterior_sample": (C.c_int, [C.POINTER(PostArgs), C.c_void_p]),
    "ccdm_pairwise_class_counts": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "ccdm_attention_ex": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "ccdm_layernorm": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_float, C.c_long, C.c_int, C.c_void_p, C.c_void_p]),
    "ccdm_gelu": (C.c_int, [C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p]),
    "ccdm_mix_uniform": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "ccdm_theta_post": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "ccdm_kl_clamped": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_float, C.c_void_p, C.c_void_p]),
    "ccdm_debug_read_timeline": (C.c_int, [C.c_void_p, C.c_int]),
    "ccdm_nchw_to_nhwc": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "ccdm_onehot_to_xin": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "ccdm_engine_create": (C.c_void_p, [C.c_void_p]),
    "ccdm_engine_destroy": (None, [C.c_void_p]),
    "ccdm_engine_add_conv": (C.c_int, [C.c_void_p, C.POINTER(ConvArgs)]),
    "ccdm_engine_add_attention": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]),
    "ccdm_engine_set_epilogue": (C.c_int, [C.c_void_p, C.POINTER(PostArgs)]),
    "ccdm_engine_num_ops": (C.c_int, [C.c_void_p]),
    "ccdm_engine_num_captures": (C.c_int, [C.c_void_p]),
    "ccdm_engine_set_run_block": (C.c_int, [C.c_void_p, C.c_void_p]),
    "ccdm_engine_set_run": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_uint64, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p]),
    "ccdm_engine_run": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "ccdm_engine_profile_op": (C.c_int, [C.c_void_p, C.c_int, C.c_int]),
    "ccdm_engine_profile_read": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    "ccdm_engine_describe_op": (C.c_int, [C.c_void_p, C.c_int, C.c_char_p, C.c_int]),
}

_lib: Optional[C.CDLL] = None


class CcdmHipError(RuntimeError):
    pass


class CcdmRangeError(CcdmHipError):
    """The network output of a run was not finite: with PREC_F16X3 the signature of a staged activation beyond the fp16
    split's range (|a| >= 4094, include/ccdm_hip.h).  Outputs of that run are invalid."""


def clean_build_tree(keep: Optional[str] = None) -> None:
    """Remove the object directories of every build flavour except `keep` and the default one (build/obj).  Never called implicitly
    by an incremental build: another process may be compiling into one of them."""
    import shutil
    root = os.path.join(_HERE, "build")
    if not os.path.isdir(root):
        return
    for d in os.listdir(root):
        p = os.path.join(root, d)
        if d.startswith("obj_") and p != keep and os.path.isdir(p):
            shutil.rmtree(p, ignore_errors=True)


def build(force: bool = False, verbose: bool = False) -> str:
    """Compile libccdm_hip.so in-tree with hipcc for gfx950 (cross-compiles without a GPU).  One object per source,
    compiled in parallel into <package>/build/ (only the sources that changed), then linked."""
    from concurrent.futures import ThreadPoolExecutor
    import hashlib
    srcs = [os.path.join(CSRC, s) for s in SOURCES]
    hdrs = [os.path.join(CSRC, h) for h in sorted(os.listdir(CSRC)) if h.endswith(".h")] + [os.path.join(ROOT, "include", "ccdm_hip.h")]
    extra = ["-DCCDM_ABLATION"] if os.environ.get("CCDM_ABLATION") else []      # tools/bench_conv.py ABLATE / TIMELINE modes
    if os.environ.get("CCDM_EXPERIMENTS"):
        extra += ["-DCCDM_EXPERIMENTS", "-I" + CSRC]
        srcs += EXPERIMENT_SOURCES
    extra += os.environ.get("CCDM_HIPCC_EXTRA", "").split()                     # one-off experiments
    # one object directory per build flavour (stable name), and a stamp next to the library saying which flavour it was linked from:
    # a plain build after an ablation / experiments build must relink, not return the other flavour's library
    flavour = hashlib.sha1(" ".join(extra).encode()).hexdigest()[:10] if extra else "default"
    objdir = os.path.join(_HERE, "build", "obj" + ("" if flavour == "default" else "_" + flavour))
    os.makedirs(objdir, exist_ok=True)
    # prune: objects whose source left the list; the object directories of OTHER flavours only on force=True (clean_build_tree() does the
    # same explicitly) — alternating ablation / experiment / default builds keep their incremental caches, and a concurrent build of
    # another flavour (the A/B tools) never loses its directory mid-compile
    wanted = {os.path.basename(s) + ".o" for s in srcs}
    for f in os.listdir(objdir):
        if f.endswith(".o") and f not in wanted:
            os.remove(os.path.join(objdir, f))
    if force:
        clean_build_tree(keep=objdir)
    stamp = LIB_PATH + ".flavour"
    try:
        linked_flavour = open(stamp).read().strip()
    except OSError:
        linked_flavour = None
    newest_hdr = max(os.path.getmtime(h) for h in hdrs)
    flags = [f for f in HIPCC_FLAGS if f != "-shared"]

    def compile_one(src: str):
        obj = os.path.join(objdir, os.path.basename(src) + ".o")
        if not force and os.path.exists(obj) and os.path.getmtime(obj) >= max(os.path.getmtime(src), newest_hdr):
            return obj, False
        cmd = ["hipcc", *flags, *extra, "-I" + os.path.join(ROOT, "include"), "-c", src, "-o", obj]
        if verbose:
            print(" ".join(cmd))
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise CcdmHipError("hipcc failed:\n" + r.stdout + r.stderr)
        return obj, True

    with ThreadPoolExecutor(max_workers=min(len(srcs), os.cpu_count() or 1)) as ex:
        res = list(ex.map(compile_one, srcs))
    objs = [o for o, _ in res]
    if not force and not any(c for _, c in res) and os.path.exists(LIB_PATH) and linked_flavour == flavour and \
            all(os.path.getmtime(LIB_PATH) >= os.path.getmtime(o) for o in objs):
        return LIB_PATH
    cmd = ["hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", *objs, "-o", LIB_PATH]
    if verbose:
        print(" ".join(cmd))
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise CcdmHipError("hipcc link failed:\n" + r.stdout + r.stderr)
    with open(stamp, "w") as fh:
        fh.write(flavour + "\n")
    return LIB_PATH


def load() -> C.CDLL:
    """Load the HIP library; raises (never falls back) if it is missing or lacks a declared symbol."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        # not built yet: compile it now if a hipcc is around (the GPU box has the same ROCm image); otherwise fail loudly
        import shutil
        if shutil.which("hipcc") and os.path.isdir(CSRC) and not os.environ.get("CCDM_NO_AUTOBUILD"):
            build(force=True)
    if not os.path.exists(LIB_PATH):
        raise CcdmHipError(
            f"{LIB_PATH} not found: the HIP extension is required (no CPU fallback). "
            "Build it with `python -c 'import __graft_entry__ as g; g.build()'`.")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:
            raise CcdmHipError(f"{LIB_PATH} does not export {name}") from e
        fn.restype = res
        fn.argtypes = args
    if lib.ccdm_version() != ABI_VERSION:
        raise CcdmHipError(f"{LIB_PATH} implements ABI version {lib.ccdm_version()}, this host code needs {ABI_VERSION}: rebuild it")
    _lib = lib
    return lib


def last_error() -> str:
    return load().ccdm_last_error_string().decode("utf-8", "replace")


def check(rc: int, what: str = "") -> int:
    if rc is None or rc < 0:
        raise CcdmHipError(f"{what}: {last_error()}")
    return rc


def pack_conv_weight(w, ksize: int, prec: int = PREC_F32, cout_absmax=None):
    """OIHW / OIK numpy fp32 -> packed numpy uint8 buffer (host-side, no GPU needed).
    cout_absmax: optional [Cout] fp32 max|W| per output channel shared by several weight sets of one GEMM."""
    import numpy as np
    lib = load()
    w = np.ascontiguousarray(w, dtype=np.float32)
    cout, cin = int(w.shape[0]), int(w.shape[1])
    assert w.size == cout * cin * ksize * ksize, (w.shape, ksize)
    am = None
    if cout_absmax is not None:
        am = np.ascontiguousarray(cout_absmax, dtype=np.float32)
        assert am.shape == (cout,)
    amp = am.ctypes.data if am is not None else None
    nbytes = lib.ccdm_pack_conv_weight_ex(None, cout, cin, ksize, prec, amp, None)
    if nbytes == 0:
        raise CcdmHipError("pack_conv_weight: " + last_error())
    out = np.empty(nbytes, dtype=np.uint8)
    lib.ccdm_pack_conv_weight_ex(w.ctypes.data, cout, cin, ksize, prec, amp, out.ctypes.data)
    return out


def pack_head_weight(w):
    """Head conv weight out.2 [K, 32, 3, 3] -> the taps-as-columns fragments of ccdm_head_posterior (host-side, no GPU needed)."""
    import numpy as np
    lib = load()
    w = np.ascontiguousarray(w, dtype=np.float32)
    k, cin = int(w.shape[0]), int(w.shape[1])
    assert w.shape == (k, cin, 3, 3), w.shape
    nbytes = lib.ccdm_pack_head_weight(None, k, cin, None)
    if nbytes == 0:
        raise CcdmHipError("pack_head_weight: " + last_error())
    out = np.empty(nbytes, dtype=np.uint8)
    lib.ccdm_pack_head_weight(w.ctypes.data, k, cin, out.ctypes.data)
    return out


def pack_stem_weight(w):
    """Stem conv weight [Cout, Cin <= 4, 3, 3] -> the packed (tap, channel)-major fragments of ccdm_stem_conv (host-side, no GPU needed)."""
    import numpy as np
    lib = load()
    w = np.ascontiguousarray(w, dtype=np.float32)
    cout, cin = int(w.shape[0]), int(w.shape[1])
    assert w.shape == (cout, cin, 3, 3), w.shape
    nbytes = lib.ccdm_pack_stem_weight(None, cout, cin, None)
    if nbytes == 0:
        raise CcdmHipError("pack_stem_weight: " + last_error())
    out = np.empty(nbytes, dtype=np.uint8)
    lib.ccdm_pack_stem_weight(w.ctypes.data, cout, cin, out.ctypes.data)
    return out


def pack_upconv_weight(w, prec: int = PREC_F16X3):
    """Upsample's 3x3 conv weight [Cout,Cin,3,3] -> packed sub-pixel form for ccdm_conv_args.up = 2 (host-side, no GPU needed)."""
    import numpy as np
    lib = load()
    w = np.ascontiguousarray(w, dtype=np.float32)
    cout, cin = int(w.shape[0]), int(w.shape[1])
    assert w.shape == (cout, cin, 3, 3), w.shape
    nbytes = lib.ccdm_pack_upconv_weight(None, cout, cin, prec, None)
    if nbytes == 0:
        raise CcdmHipError("pack_upconv_weight: " + last_error())
    out = np.empty(nbytes, dtype=np.uint8)
    lib.ccdm_pack_upconv_weight(w.ctypes.data, cout, cin, prec, out.ctypes.data)
    return out


def pack_qkv_weights(qkv_w, qkv_b, heads: int, new_order: bool):
    """AttentionBlock.qkv parameters -> the operands of ccdm_norm_qkv_attention (include/ccdm_hip.h): rows in legacy order
    (head*96 + {q,k,v}*32 + d), packed as a 1x1 conv.  Returns (wqkv uint8, bqkv fp32) numpy arrays (host-side, no GPU needed)."""
    import numpy as np
    qkv_w = np.ascontiguousarray(qkv_w, dtype=np.float32).reshape(qkv_w.shape[0], -1)
    C3, Cc = qkv_w.shape
    assert C3 == 3 * Cc and Cc == heads * 32, (qkv_w.shape, heads)
    qkv_b = np.ascontiguousarray(qkv_b, dtype=np.float32)
    if new_order:          # channel = {q,k,v}*C + head*32 + d  ->  head*96 + {q,k,v}*32 + d
        idx = np.array([j * Cc + h * 32 + d for h in range(heads) for j in range(3) for d in range(32)])
        qkv_w, qkv_b = qkv_w[idx], qkv_b[idx]
    return pack_conv_weight(qkv_w.reshape(C3, Cc, 1, 1), 1, PREC_F16X3), np.ascontiguousarray(qkv_b)
