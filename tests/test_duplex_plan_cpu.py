"""Host side of the one-replay duplex frame (AudioTokenizer.duplex_plan / duplex_commit_*; rca_duplex_frame on the GPU): what the plan
hands to the fused call, and the state the commits leave, must be exactly what tokenize_audio + detokenize_audio hand to the codec and
leave behind (reference audio_tokenizer.py:67-149).  Checked here on the CPU with the oracle codec standing in for the HIP handle."""
from types import SimpleNamespace

import numpy as np

from conftest import rich_signal


class _TailModel:
    """codec_model object over the C oracle that offers what AudioTokenizer's streaming-tail and fused paths ask of MagiCodecHIP."""

    def __init__(self, oc):
        import torch
        self._torch = torch
        self.oc = oc
        self.codebook_size = oc.cfg.codebook_size
        self.sample_rate = oc.cfg.sample_rate
        self.hip = SimpleNamespace(hop=oc.cfg.hop)
        self.tail_calls = []

    def eval(self):
        return self

    def to(self, device):
        return self

    def encode_codes(self, x):
        return self._torch.from_numpy(self.oc.encode(x.cpu().numpy()))

    def decode_codes(self, codes):
        return self._torch.from_numpy(self.oc.decode(codes.cpu().numpy())).unsqueeze(1)

    def encode_tail_np(self, x, n_keep):
        self.tail_calls.append(("enc", x.copy(), n_keep))
        return self.oc.encode(x)[:, -n_keep:]

    def decode_tail_np(self, codes, n_samples):
        self.tail_calls.append(("dec", codes.copy(), n_samples))
        return self.oc.decode(codes)[:, -n_samples:]


def test_duplex_plan_and_commits_equal_the_separate_calls():
    from oracle.codec import OracleCodec
    from realtime_codec_agent_amd.audio_tokenizer import AudioTokenizer
    from realtime_codec_agent_amd.codec_chars import chars_to_codes, codes_to_chars
    from realtime_codec_agent_amd.codec_model import init_codec_weights, tiny_codec_config
    cfg = tiny_codec_config()
    oc = OracleCodec(cfg, init_codec_weights(cfg, seed=0))
    ma, mb = _TailModel(oc), _TailModel(oc)
    a = AudioTokenizer(codec_model=ma, device="cpu", context_secs=0.5)     # 8000 samples / 25 codes of context: full after 7 frames
    b = AudioTokenizer(codec_model=mb, device="cpu", context_secs=0.5)
    sig = rich_signal(1280 * 16, 3)
    L = 320
    fused = 0

    def agent_codes(user_str, i):   # a deterministic stand-in for the LM: the agent's codes of this frame
        c = chars_to_codes(user_str, 1, cfg.codebook_size)[0]
        return codes_to_chars((c * 7 + i) % cfg.codebook_size, cfg.codebook_size)

    for i in range(16):
        chunk = sig[1280 * i:1280 * (i + 1)]
        # the separate calls
        ua = a.tokenize_audio(chunk)
        (sra, pa), ha, prea = a.detokenize_audio(agent_codes(ua, i), preroll_samples=L)
        # the fused frame, emulated: plan -> (what rca_duplex_frame computes) -> commits
        plan = b.duplex_plan(chunk, preroll_samples=L)
        if plan is None:
            ub = b.tokenize_audio(chunk)
            (srb, pb), hb, preb = b.detokenize_audio(agent_codes(ub, i), preroll_samples=L)
        else:
            fused += 1
            n0 = len(mb.tail_calls)
            codes = mb.encode_tail_np(plan["window"], plan["n_codes"])[0]
            ub = codes_to_chars(codes, cfg.codebook_size)
            b.duplex_commit_pcm(plan)
            out_str = agent_codes(ub, i)
            new = chars_to_codes(out_str, 1, cfg.codebook_size)[0]
            pcm = mb.decode_tail_np(np.concatenate([plan["code_ctx"], new])[None], plan["n_samples"])[0]
            (srb, pb), hb, preb = b.duplex_commit_codes(plan, out_str, pcm)
            # the codec saw exactly the arguments the separate calls pass
            ea, da = ma.tail_calls[-2], ma.tail_calls[-1]
            eb, db = mb.tail_calls[n0], mb.tail_calls[n0 + 1]
            assert ea[0] == "enc" and np.array_equal(ea[1], eb[1]) and ea[2] == eb[2]
            assert da[0] == "dec" and np.array_equal(da[1], db[1]) and da[2] == db[2]
        assert ua == ub and sra == srb and ha == hb and prea == preb and np.array_equal(pa, pb)
        assert a.detokenize_context == b.detokenize_context and np.array_equal(a.tokenize_context, b.tokenize_context)
    # both windows are full from the 7th frame on: every later frame plans
    assert fused == 10, fused
    # a stereo tokenizer, a codec object without a HIP handle and a chunk of another size never plan
    assert AudioTokenizer(codec_model=mb, device="cpu", num_channels=2).duplex_plan(np.zeros((2, 1280), np.float32)) is None
    mc = _TailModel(oc); del mc.hip
    c = AudioTokenizer(codec_model=mc, device="cpu", context_secs=0.5)
    for i in range(8):
        c.tokenize_audio(sig[1280 * i:1280 * (i + 1)]); c.detokenize_audio("" * 4)
    assert c.duplex_plan(sig[:1280]) is None
    assert b.duplex_plan(sig[:12800]) is None        # 40 codes: more than a frame graph holds
