"""This repo's RealtimeAgent against fixtures recorded from the REFERENCE RealtimeAgent driven with the
same deterministic fakes (tests/golden/make_agent_golden.py): identical evaluated token stream, KV
positions, emitted audio and transcript.  CPU only."""
import numpy as np
import pytest

from agent_fakes import build_fakes, scenarios, user_audio
from conftest import GOLDEN
from realtime_codec_agent_amd.realtime_agent_config import RealtimeAgentConfig
from realtime_codec_agent_amd.realtime_agent_v2 import RealtimeAgent, RealtimeAgentV2
from realtime_codec_agent_amd.realtime_agent_stats import RealtimeAgentStats
from realtime_codec_agent_amd.tokenizer import CodecTokenizer

_, _TOK = build_fakes()
SCEN = scenarios(_TOK)


def run_agent(cfg_kw, script, secs):
    resources, tok = build_fakes(script)
    agent = RealtimeAgent(resources=resources, config=RealtimeAgentConfig(**cfg_kw))
    n = int(secs * 16000)
    audio = user_audio(n)
    cs = agent.chunk_size_samples
    outs = [agent.process_audio(audio[s:s + cs]) for s in range(0, n - cs + 1, cs)]
    return agent, resources, outs


@pytest.mark.parametrize("name", sorted(SCEN))
def test_agent_matches_reference_recording(name):
    cfg_kw, script, secs = SCEN[name]
    g = np.load(f"{GOLDEN}/agent_{name}.npz")
    agent, res, outs = run_agent(cfg_kw, script, secs)
    llm = res.llm
    assert np.array_equal(np.array(agent.input_ids), g["input_ids"])
    assert np.array_equal(np.array(agent.audio_tokens_idx), g["audio_tokens_idx"])
    evals = [(a, t) for op, a, t in llm.log]
    assert np.array_equal(np.array([a for a, _ in evals]), g["eval_pos"])
    assert np.array_equal(np.array([len(t) for _, t in evals]), g["eval_len"])
    assert np.array_equal(np.concatenate([np.array(t) for _, t in evals]), g["eval_tokens"])
    assert llm.n_tokens == int(g["final_n_tokens"]) and llm.n_samples == int(g["n_samples"])
    assert len(llm.sampler_calls) == int(g["n_sampler_calls"]) and res.aux_llm.calls == int(g["aux_calls"])
    out = np.concatenate(outs)
    assert out.shape[0] == int(g["n_out"])
    assert np.abs(out[::7] - g["out_audio_dec"]).max() < 1e-6
    assert np.abs(agent.get_audio_history()[:, ::11] - g["history_dec"]).max() < 1e-6
    assert agent.format_transcript() == str(g["transcript"])
    assert agent.get_sequence_str()[-200:] == str(g["sequence_tail"])
    assert np.allclose([v[0] for v in agent.stats.event_prob.values], g["event_prob"], rtol=1e-5, atol=1e-9)
    assert agent.total_secs == float(g["total_secs"])


def test_sequence_grammar_and_shapes():
    agent, res, outs = run_agent(*SCEN["plain_80ms"])
    assert agent.chunk_size_samples == 1280 and agent.chunk_size_frames_per_channel == 4
    assert all(o.shape == (1280,) and o.dtype == np.float32 for o in outs)
    ids = agent.input_ids
    body = ids[agent.context_start_pos:]
    start = body.index(agent.start_audio_token_id) + 1
    pairs = body[start:]
    assert len(pairs) == 2 * 4 * len(outs) and all(t > agent.end_header_token_id for t in pairs)
    assert agent.audio_tokens_idx == list(range(len(ids) - len(pairs), len(ids)))
    # user channel (odd slots) is exactly the tokenised user audio
    assert agent.get_audio_history().shape == (2, 1280 * len(outs))
    assert RealtimeAgentV2 is RealtimeAgent and RealtimeAgent.step is RealtimeAgent.process_audio
    with pytest.raises(AssertionError):
        agent.process_audio(np.zeros(1000, np.float32))


def test_self_play_mode_returns_ids():
    resources, tok = build_fakes()
    agent = RealtimeAgent(resources=resources, config=RealtimeAgentConfig(use_whisper=False), self_play_mode=True)
    chunk, ids = agent.process_audio(np.zeros(1600, np.float32))
    assert chunk.shape == (1600,) and len(ids) == 5
    # a second agent can be fed the first one's ids instead of audio (inference_client_self_play.py:148-159)
    r2, _ = build_fakes()
    b = RealtimeAgent(resources=r2, config=RealtimeAgentConfig(use_whisper=False), self_play_mode=True)
    chunk2, ids2 = b.process_audio(chunk, ids)
    assert len(ids2) == 5 and b.input_ids[-1] == ids[-1]


def test_config_validation_and_out_of_scope_switches():
    with pytest.raises(ValueError):
        RealtimeAgentConfig(chunk_size_secs=0.05)
    with pytest.raises(ValueError):
        RealtimeAgentConfig(chunk_size_secs=0.02, chunk_fade_secs=0.04)
    RealtimeAgentConfig(chunk_size_secs=0.08)
    resources, tok = build_fakes()
    with pytest.raises(NotImplementedError):
        RealtimeAgent(resources=resources, config=RealtimeAgentConfig(use_external_llm=True))
    with pytest.raises(NotImplementedError):
        RealtimeAgent(resources=resources, config=RealtimeAgentConfig(use_external_tts=True))


def test_stats_zscore_schedule():
    s = RealtimeAgentStats(RealtimeAgentConfig(chunk_size_secs=0.1), value_size=2)
    assert s.last_zscore == (0.0, 0.0) and s.window_chunks == 200 and s.update_interval_chunks == 50
    for i in range(60):
        s.add_value((float(i), float(2 * i)))
    # mean/std frozen after the 50th value until the 100th
    first50 = np.array([[i, 2 * i] for i in range(50)], float)
    assert np.isclose(s.mean, first50.mean()) and np.isclose(s.std, first50.std())
    z = s.last_zscore
    assert np.isclose(z[0], (59 - s.mean) / s.std)


def test_tokenizer_layout():
    tok = CodecTokenizer()
    assert tok.vocab_size == 259344 and tok.codec_vocab_start == 128266 and tok.bos_token_id == 128000
    assert tok.convert_tokens_to_ids("<|end_header|>") == 128265 == tok.codec_vocab_start - 1
    ids = tok.encode("<|agent|><|speaker|> A<|speaker|> B<|agent_voice|>" + chr(0xE000) + chr(0xE000 + 131071) + "<|end_header|> A: hello?<|audio|>")
    assert ids[0] == 128000 and ids[3] == tok.encode(" A", add_special_tokens=False)[0] and len(tok.encode(" A", add_special_tokens=False)) == 1
    assert ids[7] == 128266 and ids[8] == 128266 + 131071
    assert tok.decode(ids[1:]) == "<|agent|><|speaker|> A<|speaker|> B<|agent_voice|>" + chr(0xE000) + chr(0xE000 + 131071) + "<|end_header|> A: hello?<|audio|>"
    assert tok.decode(tok.encode("héllo †", add_special_tokens=False)) == "héllo †"
    assert tok.convert_tokens_to_ids(chr(0xE000)) == 128266
    assert all(i > 128265 for i in tok.encode(chr(0xE005) * 4, add_special_tokens=False))


def test_shadow_cache_host_logic_on_fakes():
    """kv_shadow.py + the agent's use of it, on a fake LM that offers make_kv_shadow (no GPU): the session trims through the shadow
    cache -- tiles fed to a twin ahead of time, caches traded at the trim -- and ends in the state the reference's recompute leaves
    (same evaluated stream on the live handle up to the recompute evals, same sampled tokens, same final KV); a background tile is
    enqueued at the END of a frame and the next frame waits for it before it launches anything of its own."""
    from agent_fakes import FakeLLM, build_fakes, scenarios, user_audio
    events = []

    class Twin:
        def __init__(self):
            self.kv, self.n_tokens = [], 0

        def set_mfma_prefill(self, on):
            pass

        def copy_kv_from(self, other, n):
            self.kv = list(other.kv[:n])

        def eval_async(self, tokens):
            events.append(("tile", len(tokens)))
            self.kv = self.kv[:self.n_tokens] + [int(t) for t in tokens]
            self.n_tokens = len(self.kv)

        def sync(self):
            events.append(("settle",))

    class ShadowLLM(FakeLLM):
        def make_kv_shadow(self):
            self.twin = Twin()
            return self.twin

        def swap_kv(self, other):
            events.append(("swap",))
            self.kv, other.kv = other.kv, self.kv

        def eval(self, tokens):
            events.append(("eval", len(list(tokens))))
            super().eval(tokens)

    cfg_kw, script, secs = scenarios(_TOK)["trim"]

    def run(llm_cls, tile):
        resources, tok = build_fakes(script)
        resources.llm = llm_cls(tok.vocab_size, tok.codec_vocab_start, resources.audio_tokenizer.codebook_size, script)
        agent = RealtimeAgent.__new__(RealtimeAgent)
        RealtimeAgent.__init__(agent, resources=resources, config=RealtimeAgentConfig(**cfg_kw))
        agent.kv_shadow_tile = tile
        audio = user_audio(int(secs * 16000))
        cs = agent.chunk_size_samples
        marks = []
        for s in range(0, len(audio) - cs + 1, cs):
            marks.append(len(events))
            agent.process_audio(audio[s:s + cs])
        return agent, resources.llm, marks

    ref_agent, ref_llm, _ = run(FakeLLM, 8)
    events.clear()
    agent, llm, marks = run(ShadowLLM, 8)
    sh = agent._kv_shadow
    assert sh is not None and sh.stats["swaps"] >= 2 and sh.stats["tiles"] >= 4 and sh.stats["fallbacks"] == 0
    # same session: sequence, sampled stream, and the live cache after the swaps == after the reference's recomputes
    assert agent.input_ids == ref_agent.input_ids and llm.n_samples == ref_llm.n_samples
    assert llm.n_tokens == ref_llm.n_tokens and llm.kv[:llm.n_tokens] == ref_llm.kv[:ref_llm.n_tokens]
    # the live handle evaluated no long suffix (the reference's recompute_kv_cache(0) evaluates hundreds of tokens at a trim)
    long_ref = [n for op, _, t in ref_llm.log for n in [len(t)] if n > 16]
    long_shadow = [n for op, _, t in llm.log for n in [len(t)] if n > 16]
    assert long_ref and len(long_shadow) < len(long_ref)
    # per frame: a tile is the LAST thing a frame does; the frame after it settles the twin before its own first eval
    marks.append(len(events))
    checked = 0
    for a, b, c in zip(marks, marks[1:], marks[2:]):
        frame, nxt = events[a:b], events[b:c]
        if any(e[0] == "tile" for e in frame) and not any(e[0] == "swap" for e in frame):
            assert frame[-1][0] == "tile"
            first_eval = next(i for i, e in enumerate(nxt) if e[0] == "eval")
            assert ("settle",) in nxt[:first_eval]
            checked += 1
    assert checked >= 4
