"""Generates tests/golden/tokenizer_ref.npz by running the REFERENCE AudioTokenizer
(/root/reference/realtime_codec_agent/audio_tokenizer.py, loaded by path, no bytecode written, nothing copied)
over the oracle-backed model object of oracle/codec_staged.py through the call script of tests/tokenizer_script.py.

What this pins ("wrapper semantics"): window trim (:74), keep-last codes (:99-101), the 100-code decode
context (:113), kept samples + preroll bookkeeping (:141-145), the hanging-code quirk (:161-168), stereo interleave
(:96,116), chunked_tokenize_audio (:52-65), int16 / (sr, array) / down-mix inputs (:203-215), the framerate probe
(:181-187) and the b-1 call sequence pad_audio -> encoder -> quantizer.inference / embedding -> decoder (:189-201).

Third-party modules the reference imports are absent offline and replaced for the import only:
  librosa      to_mono = channel mean (its documented behaviour); resample is never reached (all audio is fed at 16 kHz)
  codec_bpe    codes_to_chars / chars_to_codes / UNICODE_OFFSET_LARGE come from this repo's restatement
               (realtime_codec_agent_amd/codec_chars.py) -- the char mapping itself therefore stays "unpinned"
  load_magicodec_model  never called: the model OBJECT is passed in, as the reference allows (:26-28)
    python tests/golden/make_tokenizer_golden.py
"""
import importlib.util
import os
import sys
import types

sys.dont_write_bytecode = True
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from tokenizer_script import replay, scenarios  # noqa: E402

REF_FILE = "/root/reference/realtime_codec_agent/audio_tokenizer.py"


def load_reference_tokenizer():
    from realtime_codec_agent_amd import codec_chars
    librosa = types.ModuleType("librosa")
    librosa.to_mono = lambda y: np.mean(y, axis=0) if y.ndim > 1 else y

    def _no_resample(*a, **k):
        raise AssertionError("the golden script feeds audio at the codec rate; librosa.resample must not be reached")
    librosa.resample = _no_resample
    cb = types.ModuleType("codec_bpe")
    cb.codes_to_chars, cb.chars_to_codes, cb.UNICODE_OFFSET_LARGE = codec_chars.codes_to_chars, codec_chars.chars_to_codes, codec_chars.UNICODE_OFFSET_LARGE
    tools = types.ModuleType("codec_bpe.tools")
    cu = types.ModuleType("codec_bpe.tools.codec_utils")
    cu.load_magicodec_model = None
    for name, mod in (("librosa", librosa), ("codec_bpe", cb), ("codec_bpe.tools", tools), ("codec_bpe.tools.codec_utils", cu)):
        sys.modules[name] = mod
    spec = importlib.util.spec_from_file_location("_ref_audio_tokenizer", REF_FILE)
    m = importlib.util.module_from_spec(spec)
    sys.modules["_ref_audio_tokenizer"] = m
    spec.loader.exec_module(m)
    return m.AudioTokenizer


def main():
    from oracle.codec import OracleCodec
    from oracle.codec_staged import OracleStagedCodecModel
    from realtime_codec_agent_amd.codec_model import init_codec_weights, tiny_codec_config
    RefTok = load_reference_tokenizer()
    cfg = tiny_codec_config()
    oc = OracleCodec(cfg, init_codec_weights(cfg, seed=0))
    out = {}
    for name in scenarios():
        model = OracleStagedCodecModel(oc)
        rec = replay(lambda **kw: RefTok(codec_model=model, device="cpu", **kw), name)
        rec["model_calls"] = np.array([model.calls[k] for k in ("pad_audio", "encoder", "inference", "decoder")], np.int64)
        for k, v in rec.items():
            out[f"{name}/{k}"] = v
        print(f"{name}: {len(rec['tok_lens'])} strings, {int(rec['tok_lens'].sum())} chars, {int(rec['pcm_shapes'].max(axis=1).sum())}+ samples, framerate {float(rec['framerate'])}")
    path = os.path.join(HERE, "tokenizer_ref.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
