"""LlamaForAlternatingCodeChannels -- the LM step object of the duplex loop.

Same control surface as the reference subclass of llama_cpp.Llama
(realtime_codec_agent/utils/llamacpp_utils.py:26-181) for every member the agent uses
(SURVEY.md 8b-3): reset, eval, generate(tokens, reset=False), sample, read/WRITE n_tokens,
init_sampler_for_generate, set_seed, get_logprobs, logits_to_logprobs, _ctx.get_logits(),
_n_vocab, _scores, model_path, n_ctx().  The forward pass, KV cache and sampler live behind the
C ABI (include/rca.h, rca_lm_*) as HIP kernels; there is no CPU fallback.

The model is a vanilla Llama (codec_llama.py after persist_codec_embeddings, codec_llama.py:178-206).
`model_path` may name a directory holding config.json + *.safetensors, an .npz written by
`save_npz`, or the literal "random:<name>" for seeded random-init weights generated on the device
(bench configs 3/4: no checkpoint exists offline).
"""
from __future__ import annotations

import ctypes as C
import json
import math
import os
from dataclasses import asdict, dataclass
from typing import Dict, Generator, List, Optional, Sequence

import numpy as np

from . import _native as N


@dataclass
class LMConfig:
    vocab_size: int = 259344
    hidden: int = 2048
    n_layers: int = 16
    n_heads: int = 32
    n_kv_heads: int = 8
    head_dim: int = 64
    ffn: int = 8192
    rms_eps: float = 1e-5
    rope_theta: float = 500000.0
    rope_scaling: Optional[str] = "llama3"  # None | "llama3"
    rope_factor: float = 32.0
    rope_low_freq_factor: float = 1.0
    rope_high_freq_factor: float = 4.0
    rope_orig_ctx: int = 8192

    @staticmethod
    def llama_3_2_1b(vocab_size: int = 259344) -> "LMConfig":
        """Llama-3.2-1B dims with the codec vocabulary."""
        return LMConfig(vocab_size=vocab_size)

    @staticmethod
    def from_hf(d: dict) -> "LMConfig":
        rp = d.get("rope_parameters") or d.get("rope_scaling") or {}
        theta = rp.get("rope_theta", d.get("rope_theta", 10000.0))
        rtype = rp.get("rope_type", rp.get("type", "default"))
        nh = d["num_attention_heads"]
        return LMConfig(
            vocab_size=d["vocab_size"], hidden=d["hidden_size"], n_layers=d["num_hidden_layers"], n_heads=nh,
            n_kv_heads=d.get("num_key_value_heads", nh), head_dim=d.get("head_dim") or d["hidden_size"] // nh,
            ffn=d["intermediate_size"], rms_eps=d.get("rms_norm_eps", 1e-5), rope_theta=float(theta),
            rope_scaling="llama3" if rtype == "llama3" else None, rope_factor=float(rp.get("factor", 32.0)),
            rope_low_freq_factor=float(rp.get("low_freq_factor", 1.0)), rope_high_freq_factor=float(rp.get("high_freq_factor", 4.0)),
            rope_orig_ctx=int(rp.get("original_max_position_embeddings", 8192)),
        )

    def n_params(self) -> int:
        per_layer = self.hidden * (self.n_heads + 2 * self.n_kv_heads) * self.head_dim + self.n_heads * self.head_dim * self.hidden \
            + 3 * self.hidden * self.ffn
        return self.n_layers * per_layer + 2 * self.vocab_size * self.hidden

    def weight_bytes_per_step(self) -> int:
        """bf16 bytes streamed by one decode step: all layer weights + lm_head (embedding rows are gathered)."""
        return 2 * (self.n_params() - self.vocab_size * self.hidden)


def rope_inv_freq(cfg: LMConfig) -> np.ndarray:
    """inv_freq exactly as transformers computes it in float32 (default and llama3 scaling)."""
    import torch
    dim = cfg.head_dim
    inv = 1.0 / (cfg.rope_theta ** (torch.arange(0, dim, 2, dtype=torch.int64).to(dtype=torch.float) / dim))
    if cfg.rope_scaling == "llama3":
        factor, low, high, old = cfg.rope_factor, cfg.rope_low_freq_factor, cfg.rope_high_freq_factor, cfg.rope_orig_ctx
        low_wl, high_wl = old / low, old / high
        wl = 2 * math.pi / inv
        inv_l = torch.where(wl > low_wl, inv / factor, inv)
        smooth = (old / wl - low) / (high - low)
        smoothed = (1 - smooth) * inv_l / factor + smooth * inv_l
        medium = ~(wl < high_wl) * ~(wl > low_wl)
        inv = torch.where(medium, smoothed, inv_l)
    return inv.numpy().astype(np.float32)


def f32_to_bf16_bits(a: np.ndarray) -> np.ndarray:
    """float32 -> bf16 bit pattern (round to nearest even) as uint16."""
    u = np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)
    return ((u + 0x7FFF + ((u >> 16) & 1)) >> 16).astype(np.uint16)


def bf16_bits_to_f32(b: np.ndarray) -> np.ndarray:
    return (b.astype(np.uint32) << 16).view(np.float32)


CODEC_PREFIX = "model.embed_codec_tokens."   # CodecLlamaCodecEmbedding's tensors (codec_llama.py:46-69)


def lm_tensor_names(cfg: LMConfig) -> List[str]:
    names = ["model.embed_tokens.weight", "model.norm.weight", "lm_head.weight"]
    for l in range(cfg.n_layers):
        p = f"model.layers.{l}."
        names += [p + s for s in ("self_attn.q_proj.weight", "self_attn.k_proj.weight", "self_attn.v_proj.weight",
                                  "self_attn.o_proj.weight", "mlp.gate_proj.weight", "mlp.up_proj.weight", "mlp.down_proj.weight",
                                  "input_layernorm.weight", "post_attention_layernorm.weight")]
    return names


def save_npz(path: str, cfg: LMConfig, weights: Dict[str, np.ndarray]) -> None:
    np.savez(path, __config__=np.array(json.dumps(asdict(cfg))), **{k: v for k, v in weights.items()})


def load_weights(model_path: str):
    """-> (LMConfig, {name: ndarray}) from an .npz (save_npz), a llama-architecture .gguf (the reference's format,
    realtime_agent_resources.py:12,19-25) or an HF directory with safetensors."""
    if model_path.endswith(".gguf"):
        from .gguf import load_llama_gguf
        cfg, weights, _meta = load_llama_gguf(model_path)
        return cfg, weights
    if model_path.endswith(".npz"):
        z = np.load(model_path)
        cfg = LMConfig(**json.loads(str(z["__config__"])))
        return cfg, {k: z[k] for k in z.files if k != "__config__"}
    d = model_path if os.path.isdir(model_path) else os.path.dirname(model_path)
    with open(os.path.join(d, "config.json")) as f:
        cfg = LMConfig.from_hf(json.load(f))
    from safetensors import safe_open
    import torch
    weights = {}
    for fn in sorted(os.listdir(d)):
        if fn.endswith(".safetensors"):
            with safe_open(os.path.join(d, fn), framework="pt") as sf:
                for k in sf.keys():
                    t = sf.get_tensor(k)
                    if t.dtype == torch.bfloat16:
                        weights[k] = t.view(torch.int16).numpy().view(np.uint16)
                    elif t.dtype == torch.float16 and t.dim() == 2:   # fp16 matrices stay fp16 on the device (RCA_F16)
                        weights[k] = t.numpy()
                    else:
                        weights[k] = t.float().numpy()
    if "lm_head.weight" not in weights:  # tied checkpoints
        weights["lm_head.weight"] = weights["model.embed_tokens.weight"]
    if CODEC_PREFIX + "codec_embed.weight" in weights:
        # a CodecLlamaForCausalLM checkpoint saved BEFORE persist_codec_embeddings (codec_llama.py:178-206): the constructor
        # bakes the projected codec embeddings into the table on the device
        with open(os.path.join(d, "config.json")) as f:
            hf = json.load(f)
        if hf.get("projector_hidden_act", "gelu") != "gelu":
            raise NotImplementedError(f"projector_hidden_act={hf['projector_hidden_act']!r}: only the reference default 'gelu' is built")
        weights["codec.vocab_start"] = np.array(int(hf.get("codec_vocab_start", 0)))
        weights["codec.codebook_size"] = np.array(int(hf.get("codebook_size", 131072)))
    return cfg, weights


class _Ctx:
    """llm._ctx.get_logits(): a C float* over the last logits (realtime_agent_v2.py:449,461)."""

    def __init__(self, llm: "LlamaForAlternatingCodeChannels"):
        self._llm = llm

    def get_logits(self):
        buf = self._llm._fetch_logits()
        return buf.ctypes.data_as(C.POINTER(C.c_float))

    def kv_cache_seq_rm(self, seq_id: int, p0: int, p1: int) -> None:
        # stale slots are simply overwritten by the next eval
        return None


class LlamaForAlternatingCodeChannels:
    def __init__(
        self,
        model_path: Optional[str] = None,
        n_ctx: int = 16384,
        n_gpu_layers: int = -1,
        verbose: bool = False,
        flash_attn: bool = True,
        logits_all: bool = False,
        seed: int = 0xFFFFFFFF,
        *,
        config: Optional[LMConfig] = None,
        weights: Optional[Dict[str, np.ndarray]] = None,
        device: Optional[int] = None,
        random_seed: int = 0,
        init_std: float = 0.02,
        share_weights_with: Optional["LlamaForAlternatingCodeChannels"] = None,
        weight_format: Optional[str] = None,
        **_ignored,
    ):
        """weight_format: the format every projection matrix and lm_head is KEPT in (one copy: the decode step streams it, the prefill
        tiles de-quantise it while staging).  None = as the checkpoint supplies it (bf16 / fp16 / Q8_0 tensors keep their format, f32 is
        rounded to bf16); "q8_0" = quantised at load like the Q8_0 file the reference deploys (prep_test_model.sh:29); "f16" = bf16
        values converted to fp16 (the F16 file of prep_test_model.sh:28, the reference's default model); "q4_k" = GGUF Q4_K blocks
        (the bulk of the Q4_K_M file of prep_test_model.sh:31), quantised at load with this build's own min / max rule."""
        if weight_format not in (None, "bf16", "q8_0", "f16", "q4_k"):
            raise ValueError(f"weight_format {weight_format!r}: None / 'bf16' (as supplied), 'q8_0', 'f16' or 'q4_k'")
        self._lib = N.lib()
        self.model_path = model_path
        self.verbose = verbose
        self.draft_model = None
        self._logits_all = bool(logits_all)
        if device is None:
            import torch
            if not torch.cuda.is_available():
                raise N.RcaError("LlamaForAlternatingCodeChannels needs a GPU; there is no CPU fallback")
            device = torch.cuda.current_device()
        self._device = device
        if share_weights_with is not None:
            # a second instance over the same device weights (the reference loads its GGUF twice: llm and the logits_all twin
            # aux_llm, realtime_agent_resources.py:19-33): own KV cache / sampler / stream, n_ctx <= the parent's
            parent = share_weights_with
            self.config = parent.config
            self._n_vocab = parent._n_vocab
            self._n_ctx = int(n_ctx)
            self._h = C.c_void_p()
            N.check(self._lib.rca_lm_create_shared(parent._h, self._n_ctx, 1 if logits_all else 0, C.byref(self._h)), "rca_lm_create_shared")
            self._weights_parent = parent   # (the library also copes with the parent being closed first)
            self._finish_init(seed)
            self.weight_format = parent.weight_format
            return
        random_init = weights is None and (model_path is None or str(model_path).startswith("random:"))
        if weights is None and not random_init:
            config, weights = load_weights(model_path)
        if config is None:
            config = LMConfig.llama_3_2_1b()
        self.config = config
        self._n_vocab = config.vocab_size
        self._n_ctx = int(n_ctx)
        c = N.LMConfigC(
            vocab_size=config.vocab_size, hidden=config.hidden, n_layers=config.n_layers, n_heads=config.n_heads,
            n_kv_heads=config.n_kv_heads, head_dim=config.head_dim, ffn=config.ffn, n_ctx=self._n_ctx, rms_eps=config.rms_eps,
            rope_theta=config.rope_theta, rope_scaling=1 if config.rope_scaling == "llama3" else 0, rope_factor=config.rope_factor,
            rope_low_freq_factor=config.rope_low_freq_factor, rope_high_freq_factor=config.rope_high_freq_factor,
            rope_orig_ctx=config.rope_orig_ctx, logits_all=1 if logits_all else 0, decode_weights={"q8_0": 1, "f16": 2, "q4_k": 3}.get(weight_format, 0),
        )
        self._h = C.c_void_p()
        if random_init:
            N.check(self._lib.rca_lm_create_random(C.byref(c), C.c_uint64(random_seed), C.c_float(init_std), device, C.byref(self._h)),
                    "rca_lm_create_random")
        else:
            w = {k: v for k, v in weights.items() if not k.startswith((CODEC_PREFIX, "codec."))}
            w.setdefault("rope.inv_freq", rope_inv_freq(config))   # a GGUF brings the file's own frequencies
            tensors, keep = N.make_tensors(w)
            N.check(self._lib.rca_lm_create(C.byref(c), tensors, len(w), device, C.byref(self._h)), "rca_lm_create")
            del keep
            if CODEC_PREFIX + "codec_embed.weight" in weights:   # un-persisted checkpoint: bake the projector output now
                self.persist_codec_embeddings({k[len(CODEC_PREFIX):]: v for k, v in weights.items() if k.startswith(CODEC_PREFIX)},
                                              int(weights["codec.vocab_start"]), int(weights.get("codec.codebook_size", 0)) or None)
        self._finish_init(seed)
        self.weight_format = self._query_format()[0]

    def _query_format(self):
        fmt, nbytes = C.c_int32(), C.c_int64()
        N.check(self._lib.rca_lm_weight_format(self._h, C.byref(fmt), C.byref(nbytes)), "rca_lm_weight_format")
        names = {0: "bf16", 1: "q8_0", 2: "f16", 3: "q4_k", 4: "q6_k"}    # 4: a file whose gate / up tensors are Q6_K (llama-quantize Q6_K)
        if fmt.value not in names:
            raise N.RcaError(f"rca_lm_weight_format reported an unknown format id {fmt.value}")
        return names[fmt.value], int(nbytes.value)

    def _finish_init(self, seed: int) -> None:
        self._ctx = _Ctx(self)
        self._input_ids = np.zeros(self._n_ctx, dtype=np.intc)
        self.input_ids = self._input_ids
        self._logits_host = np.zeros(self._n_vocab, dtype=np.float32)
        self._logits_valid = False
        self._sampler = None
        self._seed = seed
        self._sampler_params = None
        self._mfma_prefill = True
        self.last_n_tokens_size = 64      # llama_cpp.Llama's default penalty window (the reference never overrides it)

    # ------------------------------------------------------------------ lifetime
    def close(self):
        if getattr(self, "_h", None) is not None and self._h:
            self._lib.rca_lm_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ------------------------------------------------------------------ llama_cpp.Llama surface
    def n_ctx(self) -> int:
        return self._n_ctx

    def n_vocab(self) -> int:
        return self._n_vocab

    @property
    def n_tokens(self) -> int:
        n = C.c_int32()
        N.check(self._lib.rca_lm_get_n_tokens(self._h, C.byref(n)), "rca_lm_get_n_tokens")
        return n.value

    @n_tokens.setter
    def n_tokens(self, value: int) -> None:
        # the agent rolls the KV cache back by assigning n_tokens (realtime_agent_v2.py:208,219,261,465,730)
        N.check(self._lib.rca_lm_set_n_tokens(self._h, int(value)), "rca_lm_set_n_tokens")

    def reset(self) -> None:
        N.check(self._lib.rca_lm_reset(self._h), "rca_lm_reset")
        self._logits_valid = False

    def set_seed(self, seed: int) -> None:
        self._seed = seed & 0xFFFFFFFF

    def eval(self, tokens: Sequence[int]) -> None:
        tokens = list(tokens)
        if not tokens:
            return
        n0 = self.n_tokens
        arr = (C.c_int32 * len(tokens))(*tokens)
        N.check(self._lib.rca_lm_eval(self._h, arr, len(tokens)), "rca_lm_eval")
        self._input_ids[n0:n0 + len(tokens)] = tokens
        self._logits_valid = False

    def _fetch_logits(self) -> np.ndarray:
        if not self._logits_valid:
            N.check(self._lib.rca_lm_get_logits(self._h, C.c_void_p(self._logits_host.ctypes.data)), "rca_lm_get_logits")
            self._logits_valid = True
        return self._logits_host

    @property
    def _scores(self) -> np.ndarray:
        """Rows of logits of the last eval call (all of them for logits_all handles, else the last)."""
        rows = []
        r = 0
        while True:
            buf = np.empty(self._n_vocab, np.float32)
            rc = self._lib.rca_lm_get_logits_row(self._h, r, C.c_void_p(buf.ctypes.data))
            if rc != 0:
                break
            rows.append(buf)
            r += 1
        return np.stack(rows) if rows else np.zeros((0, self._n_vocab), np.float32)

    @property
    def scores(self) -> np.ndarray:
        return self._scores

    @staticmethod
    def logits_to_logprobs(logits: np.ndarray, axis: int = -1) -> np.ndarray:
        logits = np.asarray(logits, dtype=np.float32)
        mx = np.max(logits, axis=axis, keepdims=True)
        shifted = logits - mx
        return shifted - np.log(np.sum(np.exp(shifted), axis=axis, keepdims=True))

    def get_logprobs(self, ctx_input_ids, input_ids):
        """llamacpp_utils.py:30-37: log p(input_ids[i] | ctx, input_ids[:i]) on a logits_all handle."""
        if not self._logits_all:
            raise N.RcaError("get_logprobs needs a handle created with logits_all=True")
        self.reset()
        # only the LAST context position is scored: evaluate the context as a plain prefill (no per-position head
        # GEMV, no [n_ctx_tokens, vocab] logits buffer), then the scored tokens with every position kept
        N.check(self._lib.rca_lm_set_logits_all(self._h, 0), "rca_lm_set_logits_all")
        self._logits_all = False
        try:
            self.eval(ctx_input_ids)
            last_ctx = self._scores[-1].copy()
        finally:
            N.check(self._lib.rca_lm_set_logits_all(self._h, 1), "rca_lm_set_logits_all")
            self._logits_all = True
        self.eval(input_ids)
        logits = np.concatenate([last_ctx[None, :], self._scores], axis=0)[-len(input_ids) - 1:-1]
        logprobs = self.logits_to_logprobs(logits)
        return logprobs[range(len(input_ids)), list(input_ids)]

    def init_sampler_for_generate(
        self,
        top_k: int = 40,
        top_p: float = 0.95,
        min_p: float = 0.05,
        typical_p: float = 1.0,
        temp: float = 0.80,
        repeat_penalty: float = 1.0,
        frequency_penalty: float = 0.0,
        presence_penalty: float = 0.0,
        tfs_z: float = 1.0,
        mirostat_mode: int = 0,
        mirostat_tau: float = 5.0,
        mirostat_eta: float = 0.1,
        penalize_nl: bool = True,
        logits_processor=None,
        grammar=None,
        seed: Optional[int] = None,
        logit_bias: Optional[Dict[int, float]] = None,
    ) -> None:
        """llamacpp_utils.py:39-77.  The chain llama-cpp-python builds for these arguments, on the device: logit bias (the logits
        processor) -> penalties over the last 64 accepted tokens (repeat / frequency / presence) -> top_k -> top_p -> min_p -> temp ->
        dist -- every member the agent's config forwards (realtime_agent_v2.py:172-185, realtime_agent_config.py:11-20), top_k anywhere
        from 1 to the vocabulary or <= 0.  typical_p, mirostat and grammars are not part of the duplex path and are refused."""
        if typical_p != 1.0 or mirostat_mode != 0 or grammar is not None:
            raise NotImplementedError("typical_p / mirostat / grammar samplers are not implemented (the duplex agent never sets them)")
        self.set_seed(seed if seed is not None else -1)
        bias = dict(logit_bias or {})
        if logits_processor is not None:
            lb = getattr(logits_processor, "logit_bias_map", None)
            if lb is None:
                raise NotImplementedError("logits_processor must come from get_logits_bias_processor")
            bias.update(lb)
        ids = (C.c_int32 * max(1, len(bias)))(*bias.keys())
        vals = (C.c_float * max(1, len(bias)))(*bias.values())
        p = N.SamplerParamsC(top_k=int(top_k), top_p=float(top_p), min_p=float(min_p), temp=float(temp), seed=self._seed,
                             n_bias=len(bias), bias_ids=ids, bias_vals=vals, repeat_penalty=float(repeat_penalty),
                             freq_penalty=float(frequency_penalty), presence_penalty=float(presence_penalty),
                             penalty_last_n=int(getattr(self, "last_n_tokens_size", 64)))
        N.check(self._lib.rca_lm_sampler_init(self._h, C.byref(p)), "rca_lm_sampler_init")
        self._sampler = True
        self._sampler_params = dict(top_k=top_k, top_p=top_p, min_p=min_p, temp=temp, seed=self._seed, logit_bias=bias,
                                    repeat_penalty=repeat_penalty, frequency_penalty=frequency_penalty, presence_penalty=presence_penalty)

    def sample(self, idx: Optional[int] = None) -> int:
        assert self.n_tokens > 0
        assert self._sampler is not None
        tok = C.c_int32()
        N.check(self._lib.rca_lm_sample(self._h, C.byref(tok)), "rca_lm_sample")
        return tok.value

    def step(self, tokens: Sequence[int]) -> int:
        """eval(tokens) + sample() in one C-ABI call (hipGraph replay for 1-2 tokens)."""
        tokens = list(tokens)
        n0 = self.n_tokens
        arr = (C.c_int32 * len(tokens))(*tokens)
        tok = C.c_int32()
        N.check(self._lib.rca_lm_step(self._h, arr, len(tokens), C.byref(tok)), "rca_lm_step")
        self._input_ids[n0:n0 + len(tokens)] = tokens
        self._logits_valid = False
        return tok.value

    def step_probe(self, tokens: Sequence[int], probe_ids: Sequence[int]):
        """step(tokens) and token_probs(probe_ids) of the position it evaluated in ONE C-ABI call (one replay, one synchronisation):
        the agent's speculative <|end_audio|> step (realtime_agent_v2.py:455-466).  Returns (token, probabilities)."""
        tokens, probe_ids = list(tokens), list(probe_ids)
        n0 = self.n_tokens
        arr = (C.c_int32 * len(tokens))(*tokens)
        pid = (C.c_int32 * len(probe_ids))(*probe_ids)
        out = (C.c_float * len(probe_ids))()
        tok = C.c_int32()
        N.check(self._lib.rca_lm_step_probe(self._h, arr, len(tokens), pid, len(probe_ids), C.byref(tok), out), "rca_lm_step_probe")
        self._input_ids[n0:n0 + len(tokens)] = tokens
        self._logits_valid = False
        return tok.value, np.array(out[:], dtype=np.float32)

    def frame(self, first_pair: Sequence[int], user_ids: Sequence[int], audio_id_floor: int) -> List[int]:
        """One chunk of process_audio_input_ids (realtime_agent_v2.py:332-372) as ONE graph replay: len(user_ids) S=2 steps with
        the sampled agent token fed back on the device.  Returns the sampled tokens; the list is shorter than user_ids when a
        step sampled a token <= audio_id_floor (it is the last element): n_tokens and the sampler state are then what the same
        number of single steps would have left, and the caller continues step by step."""
        first_pair, user_ids = [int(t) for t in first_pair], [int(t) for t in user_ids]
        if len(first_pair) != 2:
            raise ValueError("frame() starts from the last [agent, user] pair")
        n, n0 = len(user_ids), self.n_tokens
        fp = (C.c_int32 * 2)(*first_pair)
        us = (C.c_int32 * n)(*user_ids)
        out = (C.c_int32 * n)()
        done = C.c_int32()
        N.check(self._lib.rca_lm_frame(self._h, fp, us, n, int(audio_id_floor), out, C.byref(done)), "rca_lm_frame")
        toks = list(out[:done.value])
        evaluated = first_pair + [t for pair in zip(toks[:-1], user_ids) for t in pair]
        self._input_ids[n0:n0 + len(evaluated)] = evaluated
        self._logits_valid = False
        return toks

    def duplex_prepare(self, T: int, F_ctx: int, n_steps: int, n_samples: int) -> None:
        """duplex_frame's one-time allocations for a call shape, ahead of the first frame (rca_duplex_prepare)."""
        N.check(self._lib.rca_duplex_prepare(self._h, int(T), int(F_ctx), int(n_steps), int(n_samples)), "rca_duplex_prepare")

    def duplex_precapture(self, codec_handle, T: int, F_ctx: int, n_steps: int, n_samples: int, code_token_base: int, with_probe: bool,
                          twin: Optional["LlamaForAlternatingCodeChannels"] = None, n_step_probe: int = 0) -> None:
        """Every graph the session's frames can replay, captured now (rca_duplex_precapture): the one-replay frame of this call
        shape for every context bucket on this handle's KV cache and on `twin`'s (the shadow cache the two trade at a trim), and the
        speculative step with n_step_probe probabilities.  Needs the sampler to be initialised."""
        a = N.DuplexFrameArgsC(pcm_window=None, code_ctx=None, T=int(T), F_ctx=int(F_ctx), n_steps=int(n_steps), n_samples=int(n_samples),
                               code_token_base=int(code_token_base), audio_id_floor=-1, probe_id=0 if with_probe else -1)
        N.check(self._lib.rca_duplex_precapture(self._h, twin._h if twin is not None else None, codec_handle._h, C.byref(a), int(n_step_probe)),
                "rca_duplex_precapture")

    def duplex_frame(self, codec_handle, pcm_window: np.ndarray, code_ctx: np.ndarray, n_steps: int, n_samples: int,
                     code_token_base: int, audio_id_floor: int, probe_id: int, first_pair: Sequence[int]) -> dict:
        """One whole duplex frame (encode tail -> the chunk's LM steps -> decode tail -> P(probe)) as ONE graph replay
        (rca_duplex_frame; RealtimeAgent.process_audio, realtime_agent_v2.py:504-554).  `codec_handle` is the HipCodec whose tail
        calls are captured.  Returns user_codes [n_steps], tokens (as frame(): shorter when a step left audio mode), pcm
        [n_samples] or None when the decode of this replay must not be used, probe_prob or None."""
        first_pair = [int(t) for t in first_pair]
        if len(first_pair) != 2:
            raise ValueError("duplex_frame() starts from the last [agent, user] pair")
        pcm_window = np.ascontiguousarray(pcm_window, dtype=np.float32).reshape(-1)
        code_ctx = np.ascontiguousarray(code_ctx, dtype=np.int64).reshape(-1)
        n0 = self.n_tokens
        a = N.DuplexFrameArgsC(pcm_window=pcm_window.ctypes.data, code_ctx=code_ctx.ctypes.data if code_ctx.size else None,
                               T=pcm_window.size, F_ctx=code_ctx.size, n_steps=int(n_steps), n_samples=int(n_samples),
                               code_token_base=int(code_token_base), audio_id_floor=int(audio_id_floor), probe_id=int(probe_id))
        a.first_pair[0], a.first_pair[1] = first_pair
        out = N.DuplexFrameOutC()
        pcm = np.empty(int(n_samples), dtype=np.float32)
        N.check(self._lib.rca_duplex_frame(self._h, codec_handle._h, C.byref(a), C.byref(out), C.c_void_p(pcm.ctypes.data)), "rca_duplex_frame")
        codes = [int(c) for c in out.user_codes[:n_steps]]
        toks = [int(t) for t in out.tokens[:out.n_done]]
        user_ids = [int(code_token_base) + c for c in codes]
        evaluated = first_pair + [t for pair in zip(toks[:-1], user_ids) for t in pair]
        self._input_ids[n0:n0 + len(evaluated)] = evaluated
        self._logits_valid = False
        return {"user_codes": codes, "tokens": toks, "pcm": pcm if out.flags == 0 else None,
                "probe_prob": float(out.probe_prob) if out.probe_prob >= 0.0 else None}

    def generate(self, tokens: Sequence[int], reset: bool = True, stopping_criteria=None) -> Generator[int, Optional[Sequence[int]], None]:
        """llamacpp_utils.py:97-181.  The agent always calls next(generate(ids, reset=False)) and drops the
        generator, so the first yield is the fused step; continuing the generator keeps sampling."""
        tokens = list(tokens)
        if reset and self.n_tokens > 0:
            longest_prefix = 0
            for a, b in zip(self._input_ids[: self.n_tokens], tokens[:-1]):
                if a == b:
                    longest_prefix += 1
                else:
                    break
            if longest_prefix > 0:
                reset = False
                tokens = tokens[longest_prefix:]
                self.n_tokens = longest_prefix
        if reset:
            self.reset()
        while True:
            token = self.step(tokens)
            if stopping_criteria is not None and stopping_criteria(self._input_ids[: self.n_tokens], self._fetch_logits()):
                return
            tokens_or_none = yield token
            tokens = [token]
            if tokens_or_none is not None:
                tokens.extend(tokens_or_none)

    def token_probs(self, token_ids: Sequence[int]) -> np.ndarray:
        """softmax(last logits)[ids], reduced on the device (measure_event_prob, realtime_agent_v2.py:448-452)."""
        ids = (C.c_int32 * len(token_ids))(*token_ids)
        out = (C.c_float * len(token_ids))()
        N.check(self._lib.rca_lm_token_probs(self._h, ids, len(token_ids), out), "rca_lm_token_probs")
        return np.array(out[:], dtype=np.float32)

    def sync(self) -> None:
        N.check(self._lib.rca_lm_sync(self._h), "rca_lm_sync")

    # ------------------------------------------------------------------ shadow KV cache (sliding-window trim without a prefill spike)
    def make_kv_shadow(self, low_priority: bool = True) -> "LlamaForAlternatingCodeChannels":
        """A twin over the same device weights with its own KV cache, workspace and (low-priority) stream: the place where the
        post-trim cache is built while this handle keeps stepping (kv_shadow.py; reference behaviour it replaces:
        recompute_kv_cache inside the trimming frame, realtime_agent_v2.py:187-190,725-733)."""
        twin = LlamaForAlternatingCodeChannels(model_path=self.model_path, n_ctx=self._n_ctx, share_weights_with=self, device=self._device)
        if low_priority:
            N.check(self._lib.rca_lm_set_low_priority(twin._h, 1), "rca_lm_set_low_priority")
        return twin

    def eval_async(self, tokens: Sequence[int]) -> None:
        """eval() that returns once its last tile / pass is enqueued; the next call on this handle waits for it."""
        tokens = list(tokens)
        if not tokens:
            return
        n0 = self.n_tokens
        arr = (C.c_int32 * len(tokens))(*tokens)
        N.check(self._lib.rca_lm_eval_async(self._h, arr, len(tokens)), "rca_lm_eval_async")
        self._input_ids[n0:n0 + len(tokens)] = tokens
        self._logits_valid = False

    def copy_kv_from(self, other: "LlamaForAlternatingCodeChannels", n_positions: int) -> None:
        """KV entries [0, n_positions) of every layer, device to device, from `other`'s cache into this one's."""
        N.check(self._lib.rca_lm_copy_kv(self._h, other._h, int(n_positions)), "rca_lm_copy_kv")
        self._input_ids[:n_positions] = other._input_ids[:n_positions]

    def swap_kv(self, other: "LlamaForAlternatingCodeChannels") -> None:
        """Exchange the KV caches of the two handles (O(1); n_tokens stays with each handle and is set by the caller)."""
        N.check(self._lib.rca_lm_swap_kv(self._h, other._h), "rca_lm_swap_kv")
        n = max(self._n_ctx, other._n_ctx)
        tmp = self._input_ids[:n].copy()
        self._input_ids[:n] = other._input_ids[:n]
        other._input_ids[:n] = tmp
        self._logits_valid = other._logits_valid = False

    def mask_head_rows(self, row_begin: int, row_end: int) -> None:
        """Zero lm_head rows (random-init models: keep sampling on codec tokens like a trained model in audio mode)."""
        N.check(self._lib.rca_lm_mask_head_rows(self._h, int(row_begin), int(row_end)), "rca_lm_mask_head_rows")

    def persist_codec_embeddings(self, codec_state: Dict[str, np.ndarray], codec_vocab_start: int, codebook_size: Optional[int] = None,
                                 return_f32: bool = False) -> Optional[np.ndarray]:
        """CodecLlamaForCausalLM.persist_codec_embeddings (codec_llama.py:178-206) on the device: table row
        codec_vocab_start + i <- linear_2(gelu(linear_1(codec_embed[i]))), one projector per codebook (codec_llama.py:62-67).
        `codec_state` holds CodecLlamaCodecEmbedding's tensors under their state-dict names ("codec_embed.weight",
        "codebook_projectors.<i>.linear_{1,2}.{weight,bias}").  With return_f32 the rows are also returned before the
        16-bit rounding (the reference's own check compares the table with the projector output, :206)."""
        f32 = lambda a: np.ascontiguousarray(bf16_bits_to_f32(a) if a.dtype == np.uint16 else a, dtype=np.float32)
        embed = f32(codec_state["codec_embed.weight"])
        n_books = 1 + max(int(k.split(".")[1]) for k in codec_state if k.startswith("codebook_projectors."))
        size = codebook_size or embed.shape[0] // n_books
        if size * n_books != embed.shape[0]:
            raise ValueError(f"codec_embed has {embed.shape[0]} rows, expected {n_books} codebooks of {size}")
        out = np.empty((embed.shape[0], self.config.hidden), dtype=np.float32) if return_f32 else None
        fp = lambda a: a.ctypes.data_as(C.POINTER(C.c_float))
        for i in range(n_books):
            p = f"codebook_projectors.{i}."
            w1, b1, w2, b2 = (f32(codec_state[p + s]) for s in ("linear_1.weight", "linear_1.bias", "linear_2.weight", "linear_2.bias"))
            if w1.shape != (self.config.hidden, embed.shape[1]) or w2.shape != (self.config.hidden, self.config.hidden):
                raise ValueError(f"projector {i}: linear_1 {w1.shape} / linear_2 {w2.shape} do not match hidden={self.config.hidden}, dim={embed.shape[1]}")
            rows = np.ascontiguousarray(embed[i * size:(i + 1) * size])
            dst = out[i * size:(i + 1) * size] if return_f32 else None
            N.check(self._lib.rca_lm_persist_codec_embeddings(self._h, fp(rows), size, embed.shape[1], fp(w1), fp(b1), fp(w2), fp(b2),
                                                              codec_vocab_start + i * size, fp(dst) if return_f32 else None),
                    "rca_lm_persist_codec_embeddings")
        self._logits_valid = False
        return out

    def set_mfma_prefill(self, enable: bool) -> None:
        """Long evals on bf16 MFMA tiles (default) or on the exact path (the decode GEMV kernels, two tokens per pass)."""
        N.check(self._lib.rca_lm_set_mfma_prefill(self._h, 1 if enable else 0), "rca_lm_set_mfma_prefill")
        self._mfma_prefill = bool(enable)

    def set_attn_fuse(self, enable: bool) -> None:
        """Merge the attention splits inside the attention launch (default) or in a launch of its own."""
        N.check(self._lib.rca_lm_set_attn_fuse(self._h, 1 if enable else 0), "rca_lm_set_attn_fuse")

    def weight_bytes_per_step(self) -> int:
        """bytes of weights one decode step streams, in the format this handle keeps them"""
        return self._query_format()[1]

    def set_graphs(self, enable: bool) -> None:
        N.check(self._lib.rca_lm_set_graphs(self._h, 1 if enable else 0), "rca_lm_set_graphs")


class _LogitBiasProcessorList(list):
    """Return type of get_logits_bias_processor: a one-element processor list that also exposes the
    bias map so the device sampler can apply it (the llama.cpp path copies the whole logits array on the
    host per step, llamacpp_utils.py:13-22)."""
    logit_bias_map: Dict[int, float] = {}


def get_logits_bias_processor(logit_bias: Dict[int, float]):
    """llamacpp_utils.py:8-24."""
    logit_bias_map = {int(k): float(v) for k, v in logit_bias.items()}

    def logit_bias_processor(input_ids, scores):
        new_scores = np.copy(scores)
        for input_id, score in logit_bias_map.items():
            new_scores[input_id] = score + scores[input_id]
        return new_scores

    out = _LogitBiasProcessorList([logit_bias_processor])
    out.logit_bias_map = logit_bias_map
    return out
