"""ctypes binding of the C ABI in include/rca.h (librca_hip.so, built in-tree by hipcc).

There is NO CPU fallback: if the library is missing or cannot be loaded, `lib()` raises
RuntimeError and every product entry point that needs it fails loudly.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from typing import Dict, Iterable, List, Optional, Tuple

import numpy as np

_PKG_DIR = os.path.dirname(os.path.abspath(__file__))
CSRC_DIR = os.path.join(_PKG_DIR, "csrc")
# RCA_LIB_PATH: load / build another copy of the library (kernel A/B experiments: RCA_EXTRA_HIPCC_FLAGS=-D... RCA_LIB_PATH=...)
LIB_PATH = os.environ.get("RCA_LIB_PATH") or os.path.join(_PKG_DIR, "librca_hip.so")
HIP_SOURCES = ["rca_codec.hip", "rca_lm.hip"]
# -amdgpu-kernarg-preload-count=14: the command processor hands the first 14 argument dwords to a wave in SGPRs at dispatch instead
# of the wave fetching them from the kernarg segment first (a cold ~0.5 us round trip in front of every first load of every one of
# the 85 dependent launches of a decode step: LM step 0.857 -> 0.843 ms at 6.6 k context, 0.828 -> 0.807 ms at 1 k, same bits)
HIPCC_FLAGS = ["--offload-arch=gfx950", "-O3", "-ffp-contract=off", "-std=c++17", "-fPIC", "-shared",
               "-Wno-unused-result", "-mllvm", "-amdgpu-kernarg-preload-count=14"]

MAX_STAGES = 8
RCA_F32, RCA_BF16, RCA_Q8_0, RCA_F16, RCA_Q4_K, RCA_Q6_K = 0, 1, 2, 3, 4, 5


class RcaError(RuntimeError):
    pass


class Q8Blocks:
    """A GGUF Q8_0 tensor kept as its raw 34-byte blocks (fp16 scale + 32 int8): raw uint8 [rows, cols / 32 * 34], logical shape (rows, cols)."""

    def __init__(self, raw: np.ndarray, shape):
        self.shape = tuple(int(x) for x in shape)
        self.raw = raw.reshape(self.shape[0], self.shape[1] // 32 * 34)
        self.dtype = np.dtype(np.uint8)

    def dequantize(self) -> np.ndarray:
        blk = self.raw.reshape(-1, 34)
        d = blk[:, :2].copy().view(np.float16).astype(np.float32)
        q = blk[:, 2:].view(np.int8).astype(np.float32)
        return (q * d).reshape(self.shape)

    def take_rows(self, index) -> "Q8Blocks":
        r = self.raw[index]
        return Q8Blocks(np.ascontiguousarray(r), (r.shape[0], self.shape[1]))


class Q4KBlocks:
    """A GGUF Q4_K tensor kept as its raw 144-byte super-blocks (fp16 d, fp16 dmin, 12 bytes of 6-bit scales / minima, 128 bytes of
    nibbles per 256 values): raw uint8 [rows, cols / 256 * 144], logical shape (rows, cols)."""

    def __init__(self, raw: np.ndarray, shape):
        self.shape = tuple(int(x) for x in shape)
        self.raw = raw.reshape(self.shape[0], self.shape[1] // 256 * 144)
        self.dtype = np.dtype(np.uint8)

    def dequantize(self) -> np.ndarray:
        """llama.cpp's dequantize_row_q4_K: (d * sc) * q - (dmin * m), f32."""
        blk = self.raw.reshape(-1, 144)
        d = blk[:, 0:2].copy().view(np.float16).astype(np.float32)
        dmin = blk[:, 2:4].copy().view(np.float16).astype(np.float32)
        s = blk[:, 4:16]
        sc = np.empty((blk.shape[0], 8), np.uint8)
        m = np.empty_like(sc)
        sc[:, 0:4], m[:, 0:4] = s[:, 0:4] & 63, s[:, 4:8] & 63
        sc[:, 4:8] = (s[:, 8:12] & 0xF) | ((s[:, 0:4] >> 6) << 4)
        m[:, 4:8] = (s[:, 8:12] >> 4) | ((s[:, 4:8] >> 6) << 4)
        qs = blk[:, 16:144].reshape(-1, 4, 32)
        q = np.stack([qs & 0xF, qs >> 4], axis=2).astype(np.float32)
        d1 = (d * sc.astype(np.float32)).astype(np.float32).reshape(-1, 4, 2, 1)
        m1 = (dmin * m.astype(np.float32)).astype(np.float32).reshape(-1, 4, 2, 1)
        return ((d1 * q).astype(np.float32) - m1).astype(np.float32).reshape(self.shape)

    def take_rows(self, index) -> "Q4KBlocks":
        r = self.raw[index]
        return Q4KBlocks(np.ascontiguousarray(r), (r.shape[0], self.shape[1]))


class Q6KBlocks:
    """A GGUF Q6_K tensor kept as its raw 210-byte super-blocks (128 bytes of low nibbles, 64 bytes of high bit pairs, 16 int8 scales,
    fp16 d per 256 values): raw uint8 [rows, cols / 256 * 210], logical shape (rows, cols).  llama-quantize Q4_K_M writes
    output.weight and some attn_v / ffn_down tensors in this format."""

    def __init__(self, raw: np.ndarray, shape):
        self.shape = tuple(int(x) for x in shape)
        self.raw = raw.reshape(self.shape[0], self.shape[1] // 256 * 210)
        self.dtype = np.dtype(np.uint8)

    def dequantize(self) -> np.ndarray:
        """llama.cpp's dequantize_row_q6_K: d * scale(group of 16) * (6-bit value - 32), f32 (exact)."""
        blk = self.raw.reshape(-1, 210)
        nb = blk.shape[0]
        ql = blk[:, 0:128].reshape(nb, 2, 64)
        qh = blk[:, 128:192].reshape(nb, 2, 32)
        sc = blk[:, 192:208].copy().view(np.int8).astype(np.float32)                 # [nb, 16], natural group order
        d = blk[:, 208:210].copy().view(np.float16).astype(np.float32)                # [nb, 1]
        q = np.empty((nb, 2, 4, 32), np.int16)
        q[:, :, 0] = (ql[:, :, 0:32] & 0xF) | (((qh >> 0) & 3) << 4)
        q[:, :, 1] = (ql[:, :, 32:64] & 0xF) | (((qh >> 2) & 3) << 4)
        q[:, :, 2] = (ql[:, :, 0:32] >> 4) | (((qh >> 4) & 3) << 4)
        q[:, :, 3] = (ql[:, :, 32:64] >> 4) | (((qh >> 6) & 3) << 4)
        q = (q.reshape(nb, 256) - 32).astype(np.float32)
        return ((d * sc).astype(np.float32).repeat(16, axis=1) * q).astype(np.float32).reshape(self.shape)

    def take_rows(self, index) -> "Q6KBlocks":
        r = self.raw[index]
        return Q6KBlocks(np.ascontiguousarray(r), (r.shape[0], self.shape[1]))


class Tensor(C.Structure):
    _fields_ = [("name", C.c_char_p), ("data", C.c_void_p), ("numel", C.c_int64), ("dtype", C.c_int32)]


class CodecConfigC(C.Structure):
    _fields_ = [
        ("sample_rate", C.c_int32),
        ("n_stages", C.c_int32),
        ("strides", C.c_int32 * MAX_STAGES),
        ("channels", C.c_int32 * (MAX_STAGES + 1)),
        ("k_in", C.c_int32),
        ("k_latent", C.c_int32),
        ("latent_dim", C.c_int32),
        ("codebook_size", C.c_int32),
        ("codebook_raw_dim", C.c_int32),
        ("codebook_dim", C.c_int32),
        ("leaky_slope", C.c_float),
    ]


class LMConfigC(C.Structure):
    _fields_ = [
        ("vocab_size", C.c_int32),
        ("hidden", C.c_int32),
        ("n_layers", C.c_int32),
        ("n_heads", C.c_int32),
        ("n_kv_heads", C.c_int32),
        ("head_dim", C.c_int32),
        ("ffn", C.c_int32),
        ("n_ctx", C.c_int32),
        ("rms_eps", C.c_float),
        ("rope_theta", C.c_float),
        ("rope_scaling", C.c_int32),
        ("rope_factor", C.c_float),
        ("rope_low_freq_factor", C.c_float),
        ("rope_high_freq_factor", C.c_float),
        ("rope_orig_ctx", C.c_int32),
        ("logits_all", C.c_int32),
        ("decode_weights", C.c_int32),
    ]


class SamplerParamsC(C.Structure):
    _fields_ = [
        ("top_k", C.c_int32),
        ("top_p", C.c_float),
        ("min_p", C.c_float),
        ("temp", C.c_float),
        ("seed", C.c_uint32),
        ("n_bias", C.c_int32),
        ("bias_ids", C.POINTER(C.c_int32)),
        ("bias_vals", C.POINTER(C.c_float)),
        ("repeat_penalty", C.c_float),
        ("freq_penalty", C.c_float),
        ("presence_penalty", C.c_float),
        ("penalty_last_n", C.c_int32),
    ]


class DuplexFrameArgsC(C.Structure):
    _fields_ = [
        ("pcm_window", C.c_void_p),
        ("code_ctx", C.c_void_p),
        ("T", C.c_int32),
        ("F_ctx", C.c_int32),
        ("n_steps", C.c_int32),
        ("n_samples", C.c_int32),
        ("code_token_base", C.c_int32),
        ("audio_id_floor", C.c_int32),
        ("probe_id", C.c_int32),
        ("first_pair", C.c_int32 * 2),
    ]


class DuplexFrameOutC(C.Structure):
    _fields_ = [
        ("user_codes", C.c_int64 * 8),
        ("tokens", C.c_int32 * 8),
        ("n_done", C.c_int32),
        ("flags", C.c_int32),
        ("probe_prob", C.c_float),
    ]


def sources() -> List[str]:
    return [os.path.join(CSRC_DIR, s) for s in HIP_SOURCES if os.path.exists(os.path.join(CSRC_DIR, s))]


def _build_key() -> Tuple[str, List[str], List[str]]:
    """(key of compiler + flags, hipcc, compile flags): the key names the objects under csrc/.obj and is stamped next to the library"""
    import hashlib
    extra = os.environ.get("RCA_EXTRA_HIPCC_FLAGS", "").split()   # e.g. -DRCA_CONV_TIMELINE (scripts/conv_timeline.py)
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    if not os.path.exists(hipcc):
        hipcc = "hipcc"
    cflags = [f for f in HIPCC_FLAGS if f != "-shared"] + extra
    return hashlib.sha1(" ".join([hipcc] + cflags).encode()).hexdigest()[:10], hipcc, cflags


def _stamp_path() -> str:
    return LIB_PATH + ".flags"


def needs_build() -> bool:
    """True when the library is missing, older than a source / header, or was built with other flags than this process asks for
    (the stamp file next to it holds the flags key: a diagnostic build left in place is never mistaken for the product build)."""
    if not os.path.exists(LIB_PATH):
        return True
    try:
        with open(_stamp_path()) as f:
            if f.read().strip() != _build_key()[0]:
                return True
    except OSError:
        return True
    deps = sources() + [os.path.join(CSRC_DIR, f) for f in os.listdir(CSRC_DIR) if f.endswith(".h")]
    deps.append(os.path.join(os.path.dirname(_PKG_DIR), "include", "rca.h"))
    t = os.path.getmtime(LIB_PATH)
    return any(os.path.getmtime(d) > t for d in deps if os.path.exists(d))


def build(force: bool = False, verbose: bool = False) -> str:
    """hipcc cross-compiles every HIP source for gfx950 into ONE in-tree shared library.

    Safe against concurrent builders (torchrun ranks, spawned session workers on a fresh tree): an flock on csrc/.obj/.lock
    serialises compile + link, the second process then finds the library current; temporaries carry the pid and are removed
    when a compile fails.  A build with RCA_EXTRA_HIPCC_FLAGS (diagnostic -D switches) refuses to overwrite the in-tree library:
    it needs RCA_LIB_PATH pointing somewhere else."""
    key, hipcc, cflags = _build_key()
    extra = os.environ.get("RCA_EXTRA_HIPCC_FLAGS", "").split()
    if extra and not os.environ.get("RCA_LIB_PATH"):
        raise RcaError(f"RCA_EXTRA_HIPCC_FLAGS={' '.join(extra)!r} without RCA_LIB_PATH: a diagnostic build must not replace "
                       f"{LIB_PATH}; set RCA_LIB_PATH=/tmp/<name>.so")
    if not force and not needs_build():
        return LIB_PATH
    import fcntl
    # one object per source, compiled side by side and kept under csrc/.obj (keyed by the flags): editing one kernel file
    # recompiles that file only; `force` recompiles everything
    obj_dir = os.path.join(CSRC_DIR, ".obj")
    os.makedirs(obj_dir, exist_ok=True)
    with open(os.path.join(obj_dir, ".lock"), "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        if not force and not needs_build():      # another process built it while this one waited
            return LIB_PATH
        headers = [os.path.join(CSRC_DIR, f) for f in os.listdir(CSRC_DIR) if f.endswith(".h")]
        headers.append(os.path.join(os.path.dirname(_PKG_DIR), "include", "rca.h"))
        hdr_t = max([os.path.getmtime(p) for p in headers if os.path.exists(p)] + [0.0])
        sfx = f".{os.getpid()}.tmp"
        objs, jobs = [], []
        for src in sources():
            obj = os.path.join(obj_dir, f"{os.path.splitext(os.path.basename(src))[0]}.{key}.o")
            objs.append(obj)
            if force or not os.path.exists(obj) or os.path.getmtime(obj) < max(os.path.getmtime(src), hdr_t):
                cmd = [hipcc] + cflags + ["-c", src, "-o", obj + sfx]
                if verbose:
                    print(" ".join(cmd))
                jobs.append((subprocess.Popen(cmd), cmd, obj))
        tmps = [obj + sfx for _, _, obj in jobs] + [LIB_PATH + sfx]
        try:
            failed = [cmd for p, cmd, _ in jobs if p.wait() != 0]
            if failed:
                raise subprocess.CalledProcessError(1, failed[0])
            for _, _, obj in jobs:
                os.replace(obj + sfx, obj)
            cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC"] + objs + ["-o", LIB_PATH + sfx]
            if verbose:
                print(" ".join(cmd))
            subprocess.check_call(cmd)
            os.replace(LIB_PATH + sfx, LIB_PATH)
            with open(_stamp_path() + sfx, "w") as f:
                f.write(key + "\n")
            os.replace(_stamp_path() + sfx, _stamp_path())
        finally:
            for t in tmps + [_stamp_path() + sfx]:
                if os.path.exists(t):
                    os.remove(t)
    return LIB_PATH


_lib = None

# every symbol include/rca.h declares (tests check the library exports all of them)
ABI_SYMBOLS = [
    "rca_last_error", "rca_device_count", "rca_device_sync", "rca_version",
    "rca_codec_create", "rca_codec_destroy", "rca_codec_hop", "rca_codec_num_frames",
    "rca_codec_encode", "rca_codec_encode_dev", "rca_codec_encode_windows_dev", "rca_codec_encode_chunk_range_dev", "rca_codec_encode_rows_dev",
    "rca_codec_decode", "rca_codec_decode_dev",
    "rca_codec_encode_tail_dev", "rca_codec_decode_tail_dev", "rca_codec_encode_tail", "rca_codec_decode_tail", "rca_codec_set_stream_graphs", "rca_codec_receptive_field", "rca_codec_set_window_trim",
    "rca_codec_encoder_dev", "rca_codec_quantize_dev", "rca_codec_decoder_dev", "rca_codec_codebook_dev",
    "rca_codec_codebook", "rca_codec_encode_tap", "rca_codec_set_variant", "rca_codec_sync",
    "rca_codec_profile", "rca_codec_profile_read",
    "rca_lm_create", "rca_lm_create_random", "rca_lm_destroy", "rca_lm_reset", "rca_lm_eval",
    "rca_lm_get_n_tokens", "rca_lm_set_n_tokens", "rca_lm_get_logits", "rca_lm_get_logits_row",
    "rca_lm_logits_dev", "rca_lm_sampler_init", "rca_lm_sample", "rca_lm_step", "rca_lm_token_probs",
    "rca_lm_sync", "rca_lm_set_graphs", "rca_lm_mask_head_rows", "rca_lm_set_mfma_prefill", "rca_lm_set_logits_all",
    "rca_lm_persist_codec_embeddings", "rca_lm_create_shared", "rca_lm_eval_async", "rca_lm_copy_kv", "rca_lm_swap_kv",
    "rca_lm_set_low_priority", "rca_lm_frame", "rca_lm_weight_format", "rca_lm_set_attn_fuse",
    "rca_duplex_frame", "rca_codec_workspace_sig", "rca_codec_stream_handoff", "rca_codec_codebook_size", "rca_codec_set_mfma_mode", "rca_lm_step_probe", "rca_duplex_prepare", "rca_duplex_precapture",
]


def lib():
    """Load librca_hip.so; raise (never fall back) when it is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RcaError(
            f"{LIB_PATH} not found: build the HIP extension first (python -c 'import __graft_entry__ as g; g.build()'). "
            "There is no CPU fallback for the product path."
        )
    # torch bundles its own libamdhip64.so.7; whichever copy of that soname is loaded first serves the
    # whole process.  Load torch's first so its allocator/streams and our kernels share ONE runtime
    # (loading the system copy first leaves torch unable to see the GPU).
    import torch  # noqa: F401
    try:
        _lib = C.CDLL(LIB_PATH)
    except OSError as e:  # pragma: no cover
        raise RcaError(f"cannot load {LIB_PATH}: {e}") from e
    _lib.rca_last_error.restype = C.c_char_p
    _lib.rca_version.restype = C.c_char_p
    return _lib


def check(rc: int, what: str = "") -> None:
    if rc != 0:
        msg = lib().rca_last_error()
        raise RcaError(f"{what} failed (rc={rc}): {msg.decode(errors='replace') if msg else ''}")


def make_tensors(weights: Dict[str, np.ndarray]) -> Tuple[C.Array, list]:
    """dict name -> ndarray (float32, float16, uint16 holding bf16 bits, or Q8Blocks) -> rca_tensor_t[]; returns keep-alive list."""
    keep = []
    arr = (Tensor * len(weights))()
    for i, (name, a) in enumerate(weights.items()):
        if isinstance(a, (Q8Blocks, Q4KBlocks, Q6KBlocks)):      # GGUF q8_0 / Q4_K / Q6_K blocks, handed over as they sit in the file
            raw = np.ascontiguousarray(a.raw)
            nb = name.encode()
            keep += [raw, nb]
            arr[i] = Tensor(nb, raw.ctypes.data, int(np.prod(a.shape)), RCA_Q8_0 if isinstance(a, Q8Blocks) else (RCA_Q4_K if isinstance(a, Q4KBlocks) else RCA_Q6_K))
            continue
        if a.dtype == np.uint16:
            dt = RCA_BF16
        elif a.dtype == np.float16:
            dt = RCA_F16
        else:
            a = np.ascontiguousarray(a, dtype=np.float32)
            dt = RCA_F32
        a = np.ascontiguousarray(a)
        nb = name.encode()
        keep += [a, nb]
        arr[i] = Tensor(nb, a.ctypes.data, a.size, dt)
    return arr, keep


def codec_config_c(cfg) -> CodecConfigC:
    c = CodecConfigC()
    c.sample_rate = cfg.sample_rate
    c.n_stages = len(cfg.strides)
    for i, s in enumerate(cfg.strides):
        c.strides[i] = s
    for i, ch in enumerate(cfg.channels):
        c.channels[i] = ch
    c.k_in, c.k_latent, c.latent_dim = cfg.k_in, cfg.k_latent, cfg.latent_dim
    c.codebook_size, c.codebook_raw_dim, c.codebook_dim = cfg.codebook_size, cfg.codebook_raw_dim, cfg.codebook_dim
    c.leaky_slope = cfg.leaky_slope
    return c
