"""Deterministic stand-ins for the LM object, shared by tests/golden/make_agent_golden.py (which drives
the REFERENCE RealtimeAgent with them) and tests/test_agent_cpu.py (which drives this repo's agent)."""
import ctypes
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


class _FakeCtx:
    def __init__(self, llm):
        self.llm = llm

    def get_logits(self):
        self.llm._logits = self.llm.logits_for_state()
        return self.llm._logits.ctypes.data_as(ctypes.POINTER(ctypes.c_float))


class FakeLLM:
    """llama_cpp.Llama look-alike.  The next token is a pure function of the evaluated context, so any two
    agents that evaluate the same token sequence get the same stream.  `script` maps the ordinal of a
    sample() call to a forced token (to steer the text branches)."""

    def __init__(self, n_vocab, codec_start, n_codec, script=None):
        self._n_vocab = n_vocab
        self.codec_start, self.n_codec = codec_start, n_codec
        self.script = dict(script or {})
        self._ctx = _FakeCtx(self)
        self.model_path = "fake"
        self.reset()
        self.log = []  # (op, n_tokens_before, tokens)
        self.n_samples = 0
        self.sampler_calls = []

    def n_ctx(self):
        return 1 << 20

    def reset(self):
        self.kv = []
        self.n_tokens = 0

    def eval(self, tokens):
        tokens = [int(t) for t in tokens]
        self.log.append(("eval", self.n_tokens, tokens))
        self.kv = self.kv[: self.n_tokens] + tokens
        self.n_tokens = len(self.kv)

    def _state_hash(self):
        kv = self.kv[: self.n_tokens]
        return (sum((i + 1) * t for i, t in enumerate(kv[-6:])) * 2654435761 + len(kv) * 97) & 0x7FFFFFFF

    def sample(self, idx=None):
        k = self.n_samples
        self.n_samples += 1
        if k in self.script:
            return self.script[k]
        return self.codec_start + self._state_hash() % self.n_codec

    def generate(self, tokens, reset=False):
        assert reset is False
        self.eval(tokens)
        while True:
            tok = self.sample()
            more = yield tok
            self.eval([tok] + list(more or []))

    def init_sampler_for_generate(self, **kw):
        self.sampler_calls.append({k: v for k, v in kw.items() if k != "logits_processor"} | {"bias": kw.get("logits_processor") is not None})

    def logits_for_state(self):
        rng = np.random.default_rng(self._state_hash())
        return rng.normal(0, 2, self._n_vocab).astype(np.float32)


class OracleCodecModel:
    """codec_model object backed by the C oracle (CPU)."""

    def __init__(self, oc):
        self.oc = oc
        self.codebook_size = oc.cfg.codebook_size
        self.sample_rate = oc.cfg.sample_rate

    def eval(self):
        return self

    def to(self, device):
        return self

    def encode_codes(self, x):
        return torch.from_numpy(self.oc.encode(x.cpu().numpy()))

    def decode_codes(self, codes):
        return torch.from_numpy(self.oc.decode(codes.cpu().numpy())).unsqueeze(1)


class FakeAuxLLM:
    """aux_llm.get_logprobs stand-in (finalize_last_response): a fixed function of the arguments."""

    def __init__(self):
        self.calls = 0

    def get_logprobs(self, ctx_input_ids, input_ids):
        self.calls += 1
        n = len(input_ids)
        return np.array([-1.0 - 0.4 * (((i * 7) + len(ctx_input_ids)) % 3) for i in range(n)], dtype=np.float64)


class FakeResources:
    def __init__(self, llm, tokenizer, audio_tokenizer):
        self.llm = llm
        self.aux_llm = FakeAuxLLM()
        self.tokenizer = tokenizer
        self.audio_tokenizer = audio_tokenizer
        self.whisper_model = None
        self.llm_model_dir = ""


def build_fakes(script=None):
    from oracle.codec import OracleCodec
    from realtime_codec_agent_amd.audio_tokenizer import AudioTokenizer
    from realtime_codec_agent_amd.codec_model import init_codec_weights, tiny_codec_config
    from realtime_codec_agent_amd.tokenizer import CodecTokenizer
    cfg = tiny_codec_config()
    oc = OracleCodec(cfg, init_codec_weights(cfg, seed=0))
    at = AudioTokenizer(codec_model=OracleCodecModel(oc), device="cpu")
    tok = CodecTokenizer(base_vocab_size=512, codebook_size=cfg.codebook_size)
    llm = FakeLLM(tok.vocab_size, tok.codec_vocab_start, cfg.codebook_size, script)
    return FakeResources(llm, tok, at), tok


def user_audio(n, seed=3):
    rng = np.random.default_rng(seed)
    knots = np.arange(0, n + 1600, 1600)
    env = np.abs(np.interp(np.arange(n), knots, rng.normal(0, 0.2, len(knots))))
    return np.clip(rng.normal(0, 1, n) * env, -1, 1).astype(np.float32)


def scenarios(tok):
    """name -> (config kwargs, script, seconds).  Scripts steer the LM into the text branches."""
    end_audio = tok.convert_tokens_to_ids("<|end_audio|>")
    start_audio = tok.convert_tokens_to_ids("<|audio|>")
    sp_a = tok.encode(" A", add_special_tokens=False)[0]
    sp_b = tok.encode(" B", add_special_tokens=False)[0]
    hi = tok.encode(": hi there", add_special_tokens=False)
    ok = tok.encode(": ok", add_special_tokens=False)
    base = dict(use_whisper=False, force_trans_after_inactivity_secs=0.0, force_response_after_inactivity_secs=0.0)
    return {
        "plain_100ms": (dict(chunk_size_secs=0.1, **base), {}, 3.0),
        "plain_80ms": (dict(chunk_size_secs=0.08, **base), {}, 2.4),
        "trim": (dict(chunk_size_secs=0.1, max_context_secs=1.0, trim_by_secs=0.4, **base), {}, 3.0),
        # sample #12: <|end_audio|>, then user label + transcription text, back to audio;
        # later the agent label + response text
        "text_branches": (dict(chunk_size_secs=0.1, **base),
                          {12: end_audio, 13: sp_b, **{14 + i: t for i, t in enumerate(hi)}, 14 + len(hi): start_audio,
                           40: end_audio, 41: sp_a, **{42 + i: t for i, t in enumerate(ok)}, 42 + len(ok): start_audio,
                           # an empty transcription is rolled back and <|end_audio|> suppressed once
                           70: end_audio, 71: sp_b, 72: start_audio},
                          3.0),
        "forced": (dict(chunk_size_secs=0.1, use_whisper=False, force_trans_after_inactivity_secs=0.0, force_response_after_inactivity_secs=0.5,
                        finalize_response_after_inactivity_secs=0.3),
                   {}, 4.0),
    }


class _OracleCtx:
    def __init__(self, llm):
        self.llm = llm

    def get_logits(self):
        return self.llm._logits.ctypes.data_as(ctypes.POINTER(ctypes.c_float))


class OracleLLM:
    """The llama_cpp.Llama surface the agent uses over the torch oracle oracle/lm_ref.py::LMRef and the C
    sampler restatement: the CPU counterpart of realtime_codec_agent_amd.llm.LlamaForAlternatingCodeChannels, for running
    the SAME agent over oracle objects and over HIP objects (tests/test_agent_gpu.py)."""

    def __init__(self, cfg, weights, n_ctx=4096):
        from oracle import lm_ref
        self._lm_ref = lm_ref
        self.ref = lm_ref.LMRef(cfg, weights, kv_dtype=torch.float16)
        self._n_vocab = cfg.vocab_size
        self._n_ctx = n_ctx
        self._ctx = _OracleCtx(self)
        self._logits = np.zeros(cfg.vocab_size, np.float32)
        self.model_path = "oracle"
        self._sampler = None
        self._counter = 0
        self.n_evals = 0

    def n_ctx(self):
        return self._n_ctx

    @property
    def n_tokens(self):
        return self.ref.n_tokens

    @n_tokens.setter
    def n_tokens(self, n):
        self.ref.set_n_tokens(int(n))

    def reset(self):
        self.ref.reset()

    def eval(self, tokens):
        tokens = [int(t) for t in tokens]
        if tokens:
            self._logits = np.ascontiguousarray(self.ref.eval(tokens, last_only=True, chunk=512)[-1].numpy(), dtype=np.float32)
            self.n_evals += 1

    def init_sampler_for_generate(self, top_k=40, top_p=0.95, min_p=0.05, temp=0.8, seed=None, logits_processor=None,
                                  repeat_penalty=1.0, frequency_penalty=0.0, presence_penalty=0.0, **_):
        bias = dict(getattr(logits_processor, "logit_bias_map", {}) or {})
        self._sampler = dict(top_k=top_k, top_p=top_p, min_p=min_p, temp=temp, seed=(seed if seed is not None else -1) & 0xFFFFFFFF, bias=bias,
                             pen=dict(repeat_penalty=repeat_penalty, frequency_penalty=frequency_penalty, presence_penalty=presence_penalty))
        self._counter = 0
        self._accepted = []          # a fresh llama.cpp sampler starts with an empty penalty window

    def sample(self, idx=None):
        s = self._sampler
        tok = self._lm_ref.sample(self._logits, s["top_k"], s["top_p"], s["min_p"], s["temp"], s["seed"], self._counter, s["bias"],
                                  prev_tokens=self._accepted, **s["pen"])
        self._counter += 1
        self._accepted.append(tok)
        return tok

    def generate(self, tokens, reset=False):
        assert reset is False
        self.eval(tokens)
        while True:
            tok = self.sample()
            more = yield tok
            self.eval([tok] + list(more or []))
