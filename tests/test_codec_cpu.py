"""CPU tests: the oracle against its golden vectors and against a plain torch fp32 twin, and the
AudioTokenizer wrapper semantics (reference audio_tokenizer.py:67-149) driven with a CPU model object.
No GPU, no HIP calls."""
import numpy as np
import pytest
import torch

from conftest import GOLDEN, bench_signal, rich_signal
from oracle.codec_ref import MagiCodecStyleRef
from realtime_codec_agent_amd.audio_tokenizer import AudioTokenizer
from realtime_codec_agent_amd.codec_chars import UNICODE_OFFSET_LARGE, chars_to_codes, codes_to_chars
from realtime_codec_agent_amd.utils.audio_utils import create_crossfade_ramps, pad_or_trim, smooth_join


@pytest.mark.parametrize("tag", ["tiny", "full"])
def test_oracle_matches_golden(tag, tiny_oracle, full_oracle):
    oc = tiny_oracle if tag == "tiny" else full_oracle
    g = np.load(f"{GOLDEN}/codec_{tag}.npz")
    pcm = np.stack([bench_signal(32000, 0), rich_signal(32000, 5)])
    codes = oc.encode(pcm)
    assert np.array_equal(codes, g["codes"])
    assert np.array_equal(oc.encode(rich_signal(4000, 7)[None, :]), g["codes_ragged"])
    rec = oc.decode(codes)
    assert np.array_equal(rec[:, :3200], g["pcm_head"]) and np.array_equal(rec[:, -3200:], g["pcm_tail"])
    assert np.array_equal(oc.codebook()[:64], g["codebook_head"])


@pytest.mark.parametrize("tag", ["tiny", "full"])
def test_oracle_vs_torch_twin(tag, tiny_codec, full_codec, tiny_oracle, full_oracle):
    cfg, w = tiny_codec if tag == "tiny" else full_codec
    oc = tiny_oracle if tag == "tiny" else full_oracle
    ref = MagiCodecStyleRef(cfg, w).eval()
    x = np.stack([bench_signal(9600, 1), rich_signal(9600, 2)])
    codes, z = oc.encode(x, tap_layer=cfg.n_stages + 2)
    with torch.no_grad():
        ze = ref.encoder(ref.pad_audio(torch.from_numpy(x)))
        zr = ref.quantizer.in_proj(ze).reshape(-1, cfg.codebook_dim).numpy()
        _, idx = ref.quantizer.inference(ze)
        cb = ref.quantizer.codebook_proj(ref.quantizer.codebook.weight)
        rec_ref = ref.decoder(torch.nn.functional.embedding(torch.from_numpy(codes), cb)).numpy()[:, 0]
    # same network, different accumulation order: fp32 tolerance
    assert np.abs(zr - z).max() < 5e-5
    assert (idx.numpy() == codes).mean() >= 0.95
    assert np.abs(oc.decode(codes) - rec_ref).max() < 5e-5
    assert np.abs(cb.numpy() - oc.codebook()).max() < 1e-5


def test_oracle_pad_audio_equals_explicit_zero_pad(tiny_oracle):
    x = rich_signal(4000, 3)[None, :]
    xp = np.pad(x, ((0, 0), (0, 4160 - 4000)))
    assert np.array_equal(tiny_oracle.encode(x), tiny_oracle.encode(xp))
    assert tiny_oracle.encode(x).shape == (1, 13)


def test_codes_chars_roundtrip():
    codes = np.array([[0, 5, 131071, 77]])
    s = codes_to_chars(codes, 131072)
    assert len(s) == 4 and ord(s[0]) == UNICODE_OFFSET_LARGE and ord(s[2]) == UNICODE_OFFSET_LARGE + 131071
    back = chars_to_codes(s, 1, 131072)
    assert np.array_equal(back, codes)
    with pytest.raises(ValueError):
        codes_to_chars(np.array([[131072]]), 131072)
    with pytest.raises(ValueError):
        chars_to_codes("a", 1, 131072)
    # two codebooks interleave per frame
    s2 = codes_to_chars(np.array([[1, 2], [3, 4]]), 10, unicode_offset=0x4E00)
    assert [ord(c) - 0x4E00 for c in s2] == [1, 13, 2, 14]
    assert np.array_equal(chars_to_codes(s2, 2, 10, unicode_offset=0x4E00), np.array([[1, 2], [3, 4]]))
    assert chars_to_codes("", 1, 10).shape == (1, 0)


def test_crossfade_helpers():
    L, fi, fo = create_crossfade_ramps(16000, 0.02)
    assert L == 320 and fi.dtype == np.float32 and fi[0] == 0.0 and np.array_equal(fo, fi[::-1])
    assert np.allclose(fi ** 2 + fo[::-1][::-1] ** 2 + 0 * fi, fi ** 2 + fo ** 2)
    a, b = np.ones(1000, np.float32), np.ones(1000, np.float32) * 2
    j = smooth_join(a, b, L, fi, fo)
    assert j.shape[-1] == 1680 and j[0] == 1 and j[-1] == 2
    assert np.array_equal(smooth_join(np.zeros((0,), np.float32), b, L, fi, fo), b)
    assert smooth_join(a, b, 0, fi, fo).shape[-1] == 2000
    assert pad_or_trim(a, 1200).shape[-1] == 1200 and pad_or_trim(a, 1200, "left")[0] == 0
    assert pad_or_trim(a, 10).shape[-1] == 10
    with pytest.raises(ValueError):
        pad_or_trim(np.zeros((2, 3)), 4)


class _OracleModel:
    """codec_model object backed by the C oracle (bit-exact reference for wrapper tests)."""

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


@pytest.mark.parametrize("chunk", [1600, 1280])
def test_tokenizer_roundtrip_config1(chunk, tiny_codec, tiny_oracle):
    """BASELINE config 1: 10 s mono, chunked tokenize -> detokenize(preroll=320) -> crossfade assembly.
    Expect 500 codes and 160000 samples; codes equal the windowed batch semantics."""
    cfg, w = tiny_codec
    tok = AudioTokenizer(codec_model=_OracleModel(tiny_oracle), device="cpu")
    assert tok.framerate == 50.0 and tok.context_samples == 32000 and tok.context_frames == 100
    sig = bench_signal(160000, 0)
    L, fi, fo = create_crossfade_ramps(16000, 0.02)
    fpc = chunk // 320
    out = np.zeros((0,), np.float32)
    codes_str = ""
    for s in range(0, 160000, chunk):
        cs = tok.tokenize_audio(sig[s:s + chunk])
        assert len(cs) == fpc
        codes_str += cs
        (sr, pcm), hanging, pre = tok.detokenize_audio(cs, preroll_samples=L)
        assert sr == 16000 and hanging == ""
        out = smooth_join(out, pcm, L, fi, fo)
    assert len(codes_str) == 500
    # first chunk yields chunk samples (no preroll available), later ones chunk+320 that overlap by 320
    assert out.shape[-1] == 160000
    want = tiny_oracle.encode_windows(sig[None, :], chunk, 32000)[0]
    got = chars_to_codes(codes_str, 1, cfg.codebook_size)[0]
    assert np.array_equal(got, want)


def test_tokenizer_via_reference_call_sequence(tiny_codec, tiny_oracle):
    """A model object WITHOUT the fused entry points is driven through pad_audio/encoder/
    quantizer.inference/decoder exactly as the reference does (audio_tokenizer.py:189-201)."""
    cfg, w = tiny_codec
    ref = MagiCodecStyleRef(cfg, w).eval()
    tok = AudioTokenizer(codec_model=ref, num_channels=2, device="cpu")
    x = np.stack([bench_signal(3200, 0), rich_signal(3200, 1)])
    s = tok.tokenize_audio(x)
    assert len(s) == 20  # 10 frames x 2 channels, interleaved ch0,ch1,ch0,...
    want = tiny_oracle.encode(x)
    got = chars_to_codes(s, 1, cfg.codebook_size)[0].reshape(10, 2).T
    assert (got == want).mean() >= 0.9  # torch conv accumulates in a different order: near-ties may flip
    (sr, pcm), hanging, pre = tok.detokenize_audio(s[:-1], preroll_samples=320)
    # 19 chars / 2 channels: one is dropped; the reference returns the tail of the KEPT string
    # (audio_tokenizer.py:161-168) and only 2880 of the 2880+320 requested samples exist -> preroll 0
    assert pcm.shape == (2, 9 * 320) and pre == 0 and hanging == s[17]
    emb = tok.get_codec_embeddings()
    assert tuple(emb.shape) == (cfg.codebook_size, 16)


def test_tokenizer_int16_and_stereo_downmix(tiny_oracle):
    tok = AudioTokenizer(codec_model=_OracleModel(tiny_oracle), device="cpu")
    x = (bench_signal(1600, 0) * 32768).astype(np.int16)
    a = tok.tokenize_audio(x)
    tok.reset_context()
    b = tok.tokenize_audio(x.astype(np.float32) / 32768.0)
    assert a == b
    tok.reset_context()
    c = tok.tokenize_audio(np.stack([x, x]).astype(np.float32) / 32768.0)
    assert c == a
    tok.reset_context()
    d = tok.tokenize_audio((8000, bench_signal(800, 0, sr=8000)))  # resampled 8k -> 16k
    assert len(d) == 5
    assert abs(tok.get_audio_codes_str_secs(a) - 0.1) < 1e-12
    assert len(tok.chunked_tokenize_audio(np.zeros(4800, np.float32), 0.1)) == 15


# --------------------------------------------------------------------- receptive field
@pytest.mark.parametrize("tag", ["tiny", "full"])
def test_receptive_field_margins_hold_on_the_oracle(tag, tiny_oracle, full_oracle):
    """The streaming tail rests on one claim about the codec DEFINITION: the last n codes of a window, and the last
    n samples of a decode, do not depend on anything further left than CodecConfig.receptive_field() frames.
    Checked on the CPU oracle itself: a window cut down to kept + margin frames gives the same codes / PCM bit for
    bit, and one frame less of margin does not (the margin is tight for the decoder, whose output is not quantised)."""
    oc = tiny_oracle if tag == "tiny" else full_oracle
    enc_left, dec_left = oc.cfg.receptive_field()
    hop = oc.cfg.hop
    T = 6400 if tag == "full" else 16000
    x = np.stack([rich_signal(T, 31), bench_signal(T, 32)])
    full = oc.encode(x)
    F = full.shape[1]
    for keep in (1, 4, 5, 7):
        j = max(0, F - keep - enc_left)
        assert np.array_equal(oc.encode(x[:, j * hop:])[:, -keep:], full[:, -keep:]), keep
    # ragged window (not a hop multiple: frames are counted from the window start, so whole frames are dropped)
    xr = x[:, : T - 100]
    fr = oc.encode(xr)
    j = max(0, fr.shape[1] - 5 - enc_left)
    assert np.array_equal(oc.encode(xr[:, j * hop:])[:, -5:], fr[:, -5:])
    rng = np.random.default_rng(5)
    codes = rng.integers(0, oc.cfg.codebook_size, (2, 20 if tag == "full" else 40))
    pcm = oc.decode(codes)
    Fd = codes.shape[1]
    for n in (320, 1920, 1600, 1000):
        f0 = (Fd * hop - n) // hop
        j = max(0, f0 - dec_left)
        assert np.array_equal(oc.decode(codes[:, j:])[:, -n:], pcm[:, -n:]), n
        if j > 0 and n % hop == 0:   # tight when the first kept sample sits on a frame boundary
            assert not np.array_equal(oc.decode(codes[:, j + 1:])[:, -n:], pcm[:, -n:]), f"decoder margin not tight for n={n}"
