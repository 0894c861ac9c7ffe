"""This repo's AudioTokenizer replays the call script of tests/tokenizer_script.py and must reproduce, character for
character and bit for bit, what the REFERENCE AudioTokenizer recorded over the same (oracle) codec
(tests/golden/tokenizer_ref.npz, written by tests/golden/make_tokenizer_golden.py from
/root/reference/realtime_codec_agent/audio_tokenizer.py:67-187).

CPU: over the oracle model object, through the reference's own b-1 call sequence (pad_audio -> encoder -> quantizer.inference,
embedding -> decoder) and through the fused encode_codes / decode_codes calls.
GPU: over MagiCodecHIP (the HIP kernels through the C ABI), streaming tail on and off, and through the b-1 staged calls.
"""
import numpy as np
import pytest

from conftest import GOLDEN
from tokenizer_script import replay, scenarios

NAMES = list(scenarios())


@pytest.fixture(scope="module")
def golden():
    return np.load(f"{GOLDEN}/tokenizer_ref.npz")


def _check(rec, golden, name, skip=()):
    keys = [k.split("/", 1)[1] for k in golden.files if k.startswith(name + "/")]
    assert keys, name
    for k in keys:
        if k in skip or k == "model_calls":
            continue
        want, got = golden[f"{name}/{k}"], rec[k]
        if k == "pcm_sha256" and not np.array_equal(want, got):
            d = np.abs(rec["pcm_dec"] - golden[f"{name}/pcm_dec"])
            raise AssertionError(f"{name}: PCM differs from the reference recording (max |d| on the decimated samples {d.max():.3e})")
        assert np.array_equal(np.asarray(want), np.asarray(got)), f"{name}/{k}: {want!r} != {got!r}"


@pytest.fixture(scope="module")
def oracle_tiny():
    from oracle.codec import OracleCodec
    from realtime_codec_agent_amd.codec_model import init_codec_weights, tiny_codec_config
    cfg = tiny_codec_config()
    return OracleCodec(cfg, init_codec_weights(cfg, seed=0))


@pytest.mark.parametrize("name", NAMES)
def test_cpu_staged_calls_match_reference_recording(name, golden, oracle_tiny):
    from oracle.codec_staged import OracleStagedCodecModel
    from realtime_codec_agent_amd.audio_tokenizer import AudioTokenizer
    model = OracleStagedCodecModel(oracle_tiny)
    rec = replay(lambda **kw: AudioTokenizer(codec_model=model, device="cpu", **kw), name)
    _check(rec, golden, name)
    # same number of model-object calls as the reference made, except that every channel of a stereo window goes out as
    # one batched call here (the reference loops over channels, audio_tokenizer.py:84,137) and the codebook is projected
    # through the same calls
    ref_calls = golden[f"{name}/model_calls"]
    mine = np.array([model.calls[k] for k in ("pad_audio", "encoder", "inference", "decoder")])
    ch = scenarios()[name][0].get("num_channels", 1)
    assert np.all(mine <= ref_calls) and np.all(mine * ch >= ref_calls - ch), (mine, ref_calls)


@pytest.mark.parametrize("name", NAMES)
def test_cpu_fused_calls_match_reference_recording(name, golden, oracle_tiny):
    from agent_fakes import OracleCodecModel
    from oracle.codec_staged import OracleStagedCodecModel
    from realtime_codec_agent_amd.audio_tokenizer import AudioTokenizer
    model = OracleCodecModel(oracle_tiny)
    model.quantizer = OracleStagedCodecModel(oracle_tiny).quantizer     # get_codec_embeddings only
    rec = replay(lambda **kw: AudioTokenizer(codec_model=model, device="cpu", **kw), name)
    _check(rec, golden, name)


def _hip_model():
    from realtime_codec_agent_amd.codec import MagiCodecHIP
    from realtime_codec_agent_amd.codec_model import init_codec_weights, tiny_codec_config
    cfg = tiny_codec_config()
    return MagiCodecHIP(cfg, init_codec_weights(cfg, seed=0))


@pytest.mark.gpu
@pytest.mark.parametrize("tail", [True, False])
@pytest.mark.parametrize("name", NAMES)
def test_gpu_audio_tokenizer_matches_reference_recording(name, tail, golden):
    from realtime_codec_agent_amd.audio_tokenizer import AudioTokenizer
    model = _hip_model()

    def make(**kw):
        at = AudioTokenizer(codec_model=model, **kw)
        at.streaming_tail = tail
        return at
    _check(replay(make, name), golden, name)


class _StagedOnly:
    """Hides the fused / tail entry points of MagiCodecHIP so AudioTokenizer drives it exactly as the reference drives
    MagiCodec: pad_audio -> encoder -> quantizer.inference and embedding -> decoder (audio_tokenizer.py:189-201)."""

    def __init__(self, m):
        self._m = m
        self.codebook_size, self.sample_rate, self.quantizer = m.codebook_size, m.sample_rate, m.quantizer
        self.pad_audio, self.encoder, self.decoder = m.pad_audio, m.encoder, m.decoder

    def eval(self):
        return self

    def to(self, device):
        return self


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["mono_80ms", "stereo_hanging", "int16_and_long_chunk"])
def test_gpu_staged_model_calls_match_reference_recording(name, golden):
    from realtime_codec_agent_amd.audio_tokenizer import AudioTokenizer
    model = _StagedOnly(_hip_model())
    _check(replay(lambda **kw: AudioTokenizer(codec_model=model, **kw), name), golden, name)
