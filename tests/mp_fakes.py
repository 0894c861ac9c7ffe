"""Picklable resource factories for the RealtimeAgentMultiprocessing tests (imported inside the spawned worker)."""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
for p in (HERE, os.path.dirname(HERE)):
    if p not in sys.path:
        sys.path.insert(0, p)


def fake_resources(script=None):
    from agent_fakes import build_fakes
    return build_fakes(script)[0]


def broken_resources():
    raise FileNotFoundError("no such model file: /nonexistent/model.gguf")


class _FailsOnSecondFrame:
    """Wraps the fake LM: its 9th generate call (inside the second process_audio) blows up inside the worker."""

    def __init__(self, inner):
        self._inner, self._evals = inner, 0

    def __getattr__(self, name):
        return getattr(self._inner, name)

    def generate(self, tokens, reset=False):
        self._evals += 1
        if self._evals == 9:
            raise RuntimeError("synthetic LM failure")
        return self._inner.generate(tokens, reset=reset)


def flaky_resources():
    res = fake_resources()
    res.llm = _FailsOnSecondFrame(res.llm)
    return res


def tiny_gpu_resources():
    """HIP model objects at test size: tiny conv codec + a 2-layer LM, random-init on the visible GPU."""
    from types import SimpleNamespace
    from realtime_codec_agent_amd.audio_tokenizer import AudioTokenizer
    from realtime_codec_agent_amd.codec import MagiCodecHIP
    from realtime_codec_agent_amd.codec_model import init_codec_weights, tiny_codec_config
    from realtime_codec_agent_amd.llm import LlamaForAlternatingCodeChannels, LMConfig
    from realtime_codec_agent_amd.tokenizer import CodecTokenizer
    ccfg = tiny_codec_config()
    tok = CodecTokenizer(base_vocab_size=512, codebook_size=ccfg.codebook_size)
    lcfg = LMConfig(vocab_size=tok.vocab_size, hidden=256, n_layers=2, n_heads=4, n_kv_heads=2, head_dim=64, ffn=512)
    llm = LlamaForAlternatingCodeChannels(model_path="random:mp", config=lcfg, n_ctx=2048, random_seed=3, init_std=0.05)
    llm.mask_head_rows(0, tok.codec_vocab_start)
    llm.mask_head_rows(len(tok), llm.n_vocab())
    return SimpleNamespace(llm=llm, aux_llm=None, tokenizer=tok, audio_tokenizer=AudioTokenizer(codec_model=MagiCodecHIP(ccfg, init_codec_weights(ccfg, 0))),
                           whisper_model=None, llm_model_dir="", visible=os.environ.get("HIP_VISIBLE_DEVICES"))
