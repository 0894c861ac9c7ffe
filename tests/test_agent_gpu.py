"""End-to-end duplex loop on the GPU: the real HIP AudioTokenizer + HIP LM under RealtimeAgent."""
import numpy as np
import pytest

from conftest import rich_signal

pytestmark = pytest.mark.gpu


def make_agent(chunk=0.08, **cfg_kw):
    from realtime_codec_agent_amd.llm import LMConfig
    from realtime_codec_agent_amd.realtime_agent_config import RealtimeAgentConfig
    from realtime_codec_agent_amd.realtime_agent_resources import RealtimeAgentResources
    from realtime_codec_agent_amd.realtime_agent_v2 import RealtimeAgent
    lm = LMConfig(vocab_size=259344, hidden=256, n_layers=2, n_heads=4, n_kv_heads=2, head_dim=64, ffn=512)
    res = RealtimeAgentResources(llm_model_path="random:small", llm_n_ctx=4096, llm_config=lm, with_aux_llm=False, llm_random_seed=3)
    cfg = RealtimeAgentConfig(chunk_size_secs=chunk, use_whisper=False, force_trans_after_inactivity_secs=0.0,
                              force_response_after_inactivity_secs=0.0, **cfg_kw)
    return RealtimeAgent(resources=res, config=cfg), res


def test_duplex_loop_end_to_end_and_deterministic():
    sig = rich_signal(16000 * 2, 4)
    runs = []
    for _ in range(2):
        agent, res = make_agent()
        outs = [agent.process_audio(sig[s:s + 1280]) for s in range(0, len(sig) - 1279, 1280)]
        assert all(o.shape == (1280,) and np.isfinite(o).all() for o in outs)
        runs.append((list(agent.input_ids), np.concatenate(outs)))
        summ = agent.profilers.summary()
        assert summ["total"]["p50"] is not None
    # seeded sampler + bit-stable kernels: the whole session is reproducible
    assert runs[0][0] == runs[1][0] and np.array_equal(runs[0][1], runs[1][1])
    ids = runs[0][0]
    assert all(t > agent.end_header_token_id for t in ids[-50:])
    # user channel tokens in the sequence equal the tokenizer's codes for the user audio
    from realtime_codec_agent_amd.audio_tokenizer import AudioTokenizer
    tok2 = AudioTokenizer(codec_model=res.audio_tokenizer.codec_model, device=res.audio_tokenizer.device)
    tok2.chunked_tokenize_audio(np.zeros(48000, np.float32), 0.08)  # the enrollment audio precedes the stream (reset())
    want = res.tokenizer.encode(tok2.chunked_tokenize_audio(sig[: 1280 * 25], 0.08), add_special_tokens=False)
    got = [agent.input_ids[i] for i in agent.audio_tokens_idx[1::2]]
    assert got == want


def test_trim_recompute_matches_fresh_prefill():
    """After the sliding-window eviction (realtime_agent_v2.py:187-190,725-733) the LM state equals a
    fresh prefill of header + surviving suffix: next-step logits are identical."""
    agent, res = make_agent(chunk=0.1, max_context_secs=1.0, trim_by_secs=0.4)
    res.llm.set_mfma_prefill(False)   # exact mode: recompute and fresh prefill then agree bit for bit
    agent.reset()                     # re-prefill the header in exact mode too
    sig = rich_signal(16000 * 2, 9)
    for s in range(0, len(sig) - 1599, 1600):
        agent.process_audio(sig[s:s + 1600])
    assert agent.trim_to_secs > 0
    llm = res.llm
    trim_pos = agent.audio_tokens_idx[agent.frames_from_secs(agent.trim_to_secs)]
    seq = agent.input_ids[: agent.context_start_pos] + agent.input_ids[trim_pos:]
    assert llm.n_tokens == len(seq) - 2
    llm.eval(agent.input_ids[-2:])
    a = llm._scores[-1].copy()
    llm.reset()
    llm.eval(seq)
    assert np.array_equal(llm._scores[-1], a)


def test_self_play_clone_shares_weights_and_talks_to_the_original():
    """clone_for_self_play (realtime_agent_resources.py:41-49): the second agent's LM is a fresh model state over the same
    device weights, with its own KV cache; two agents in self-play mode feed each other's ids
    (inference_client_self_play.py:148-159) and the clone behaves exactly like an agent built from scratch."""
    from realtime_codec_agent_amd.realtime_agent_config import RealtimeAgentConfig
    from realtime_codec_agent_amd.realtime_agent_v2 import RealtimeAgent
    a, res = make_agent()
    a = RealtimeAgent(resources=res, config=a.config, self_play_mode=True)
    r2 = res.clone_for_self_play()
    assert r2.llm is not res.llm and r2.audio_tokenizer.codec_model is res.audio_tokenizer.codec_model
    b = RealtimeAgent(resources=r2, config=RealtimeAgentConfig(chunk_size_secs=0.08, use_whisper=False, force_trans_after_inactivity_secs=0.0,
                                                               force_response_after_inactivity_secs=0.0), self_play_mode=True)
    fresh, _ = make_agent()          # same seeds, loaded separately: the reference's way
    sig = rich_signal(16000, 12)
    for s in range(0, len(sig) - 1279, 1280):
        chunk = sig[s:s + 1280]
        out_a, ids_a = a.process_audio(chunk)
        out_b, ids_b = b.process_audio(chunk)
        out_f = fresh.process_audio(chunk)
        assert np.array_equal(out_b, out_f)              # the clone is indistinguishable from a separately loaded agent
        assert len(ids_a) == len(ids_b) == 4
    assert a.input_ids == b.input_ids                    # same seeds, same audio, same weights: same sessions
    assert res.llm.n_tokens == r2.llm.n_tokens


def _paired_resources(seed=3):
    """Two resource sets over the SAME weights: HIP objects (MagiCodecHIP + the HIP LM through the C ABI) and oracle objects
    (C codec oracle + torch LMRef + C sampler restatement)."""
    from types import SimpleNamespace
    from agent_fakes import OracleCodecModel, OracleLLM
    from oracle import lm_ref
    from oracle.codec import OracleCodec
    from realtime_codec_agent_amd.audio_tokenizer import AudioTokenizer
    from realtime_codec_agent_amd.codec import MagiCodecHIP
    from realtime_codec_agent_amd.codec_model import init_codec_weights, tiny_codec_config
    from realtime_codec_agent_amd.llm import LlamaForAlternatingCodeChannels, LMConfig
    from realtime_codec_agent_amd.tokenizer import CodecTokenizer
    ccfg = tiny_codec_config()
    cw = init_codec_weights(ccfg, seed=0)
    tok = CodecTokenizer(base_vocab_size=512, codebook_size=ccfg.codebook_size)
    lcfg = LMConfig(vocab_size=tok.vocab_size, hidden=256, n_layers=2, n_heads=4, n_kv_heads=2, head_dim=64, ffn=512)
    w = lm_ref.random_weights(lcfg, seed, 0.05)
    head = w["lm_head.weight"].copy()
    head[: tok.codec_vocab_start] = 0          # like a trained codec LM in audio mode: the text rows never win
    head[len(tok):] = 0
    w["lm_head.weight"] = head
    hip = SimpleNamespace(llm=LlamaForAlternatingCodeChannels(config=lcfg, weights=w, n_ctx=2048, device=0), aux_llm=None, tokenizer=tok,
                          audio_tokenizer=AudioTokenizer(codec_model=MagiCodecHIP(ccfg, cw)), whisper_model=None, llm_model_dir="")
    ora = SimpleNamespace(llm=OracleLLM(lcfg, w, n_ctx=2048), aux_llm=None, tokenizer=tok,
                          audio_tokenizer=AudioTokenizer(codec_model=OracleCodecModel(OracleCodec(ccfg, cw)), device="cpu"),
                          whisper_model=None, llm_model_dir="")
    return hip, ora


@pytest.mark.parametrize("chunk,exact_prefill", [(0.08, True), (0.1, True), (0.08, False)])
def test_agent_over_hip_objects_equals_agent_over_oracle_objects(chunk, exact_prefill):
    """a12 / a16 against the oracle, not against itself: the SAME RealtimeAgent code runs once over the HIP model objects and once
    over the CPU oracle objects (greedy sampling, the reference's process_audio loop realtime_agent_v2.py:504-554 with the
    sliding-window trim + KV recompute :187-190,725-733 firing several times).  Token stream, KV position, emitted PCM and the
    audio history must be equal -- ids exactly, PCM bit for bit (the codec kernels are bit-exact and the ids are equal)."""
    from realtime_codec_agent_amd.realtime_agent_config import RealtimeAgentConfig
    from realtime_codec_agent_amd.realtime_agent_v2 import RealtimeAgent
    hip, ora = _paired_resources()
    hip.llm.set_mfma_prefill(not exact_prefill)
    cfg = dict(chunk_size_secs=chunk, use_whisper=False, force_trans_after_inactivity_secs=0.0, force_response_after_inactivity_secs=0.0,
               temperature=0.0, max_context_secs=1.0, trim_by_secs=0.4)
    a_hip = RealtimeAgent(resources=hip, config=RealtimeAgentConfig(**cfg))
    a_ora = RealtimeAgent(resources=ora, config=RealtimeAgentConfig(**cfg))
    assert a_hip.input_ids == a_ora.input_ids                      # header: enrollment codes from both tokenizers
    n = int(chunk * 16000)
    sig = rich_signal(n * 36, 21)
    for s in range(0, len(sig), n):
        o_hip = a_hip.process_audio(sig[s:s + n])
        o_ora = a_ora.process_audio(sig[s:s + n])
        assert a_hip.input_ids == a_ora.input_ids, f"token streams diverge in chunk {s // n}"
        assert np.array_equal(o_hip, o_ora), f"emitted PCM differs in chunk {s // n}"
        assert hip.llm.n_tokens == ora.llm.n_tokens
    assert a_hip.trim_to_secs == a_ora.trim_to_secs and a_hip.trim_to_secs >= 0.8      # several trims happened
    assert np.array_equal(a_hip.get_audio_history(), a_ora.get_audio_history())
    assert a_hip.audio_tokens_idx == a_ora.audio_tokens_idx
    d = np.abs(hip.llm._scores[-1] - ora.llm._logits).max()
    print(f"last-step logits HIP vs oracle after {len(a_hip.input_ids)} tokens: max|d| = {d:.2e}")
    assert d < (2e-4 if exact_prefill else 6e-4)
    assert len(set(a_hip.input_ids[a_hip.context_start_pos + 8::2])) > 10           # the agent channel is not stuck on one code


def test_long_session_over_hip_objects_equals_oracle_session_past_1k_tokens_with_shadow_trims():
    """The paired session above at the context lengths of a real dialogue: a 10 s window (header ~150 + 100 tokens per second) that
    grows past 1 100 tokens before the first sliding-window trim and is trimmed twice (realtime_agent_v2.py:187-190,725-733;
    window knobs realtime_agent_config.py:23-24).  The HIP agent runs its default fast paths -- one graph replay per frame
    (rca_duplex_frame), flash prefill tiles, the post-trim cache built ahead on the shadow twin and swapped in -- the oracle agent
    the reference's own call sequence over CPU objects (C codec oracle, LMRef, C sampler).  Ids exact, PCM bit for bit, final logits
    within the MFMA-prefill tolerance of the short paired test."""
    from realtime_codec_agent_amd.realtime_agent_config import RealtimeAgentConfig
    from realtime_codec_agent_amd.realtime_agent_v2 import RealtimeAgent
    hip, ora = _paired_resources()
    cfg = dict(chunk_size_secs=0.08, use_whisper=False, force_trans_after_inactivity_secs=0.0, force_response_after_inactivity_secs=0.0,
               temperature=0.0, max_context_secs=10.0, trim_by_secs=2.5)
    a_hip = RealtimeAgent(resources=hip, config=RealtimeAgentConfig(**cfg))
    a_ora = RealtimeAgent(resources=ora, config=RealtimeAgentConfig(**cfg))
    assert a_hip.use_kv_shadow and a_hip.use_frame_graph
    assert a_hip.input_ids == a_ora.input_ids
    n = 1280
    sig = rich_signal(n * 170, 29)            # 13.6 s: trims at 10 s and 12.5 s
    peak = 0
    for s in range(0, len(sig), n):
        o_hip = a_hip.process_audio(sig[s:s + n])
        o_ora = a_ora.process_audio(sig[s:s + n])
        assert a_hip.input_ids == a_ora.input_ids, f"token streams diverge in chunk {s // n}"
        assert np.array_equal(o_hip, o_ora), f"emitted PCM differs in chunk {s // n}"
        assert hip.llm.n_tokens == ora.llm.n_tokens
        peak = max(peak, hip.llm.n_tokens)
    assert peak > 1100, peak
    assert a_hip.trim_to_secs == a_ora.trim_to_secs == 5.0                           # two trims
    st = a_hip._kv_shadow.stats
    print("peak context", peak, "tokens; shadow stats", st)
    assert st["swaps"] == 2 and st["fallbacks"] == 0 and st["tiles"] >= 8
    assert np.array_equal(a_hip.get_audio_history(), a_ora.get_audio_history())
    d = np.abs(hip.llm._scores[-1] - ora.llm._logits).max()
    print(f"last-step logits HIP vs oracle after {len(a_hip.input_ids)} tokens: max|d| = {d:.2e}")
    assert d < 6e-4


@pytest.mark.parametrize("mfma_prefill", [True, False])
def test_trim_through_shadow_cache_equals_reference_recompute(mfma_prefill):
    """a16 / SURVEY 8f-1: the same sampled session (seeded top-k sampling, several sliding-window trims) with the post-trim cache
    built ahead of time on the twin LM and swapped in, and with the reference's in-frame recompute (realtime_agent_v2.py:187-190,
    725-733).  Token streams, emitted audio and the final logits must be identical; the shadow run must really have swapped."""
    runs = []
    for shadow in (True, False):
        agent, res = make_agent(chunk=0.08, max_context_secs=1.2, trim_by_secs=0.4)
        res.llm.set_mfma_prefill(mfma_prefill)
        agent.use_kv_shadow = shadow
        agent.kv_shadow_tile = 16
        agent.reset()
        sig = rich_signal(1280 * 60, 33)
        outs = [agent.process_audio(sig[s:s + 1280]) for s in range(0, len(sig), 1280)]
        res.llm.eval(agent.input_ids[-2:])
        runs.append((list(agent.input_ids), np.concatenate(outs), res.llm._scores[-1].copy(), agent.trim_to_secs))
        if shadow:
            st = agent._kv_shadow.stats
            print("shadow stats:", st)
            assert st["swaps"] >= 5 and st["tiles"] >= 5 and st["fallbacks"] == 0
        else:
            assert agent._kv_shadow is None or agent._kv_shadow.stats["swaps"] == agent._kv_shadow.stats["planned"] == 0   # (the twin may exist since reset(): never used)
    assert runs[0][3] == runs[1][3] >= 2.0
    assert runs[0][0] == runs[1][0]
    assert np.array_equal(runs[0][1], runs[1][1])
    assert np.array_equal(runs[0][2], runs[1][2])


def test_frame_graph_session_equals_step_by_step_session():
    """N1 (north_star: the per-frame loop is hipGraph-captured): the same sampled session (a) with the WHOLE frame -- encode tail, code ->
    id, the chunk's LM steps, id -> code, decode tail, P(<|end_audio|>) -- as one graph replay (rca_duplex_frame), (b) with one replay
    per LM chunk (llm.frame) between the separate codec calls, (c) with one replay per step -- identical token streams, audio, event
    statistics and final logits, trims included."""
    runs = []
    for duplex_graph, frame_graph in ((True, True), (False, True), (False, False)):
        agent, res = make_agent(chunk=0.08, max_context_secs=1.2, trim_by_secs=0.4)
        agent.use_frame_graph = frame_graph
        agent.use_duplex_graph = duplex_graph
        sig = rich_signal(1280 * 70, 35)
        outs = [agent.process_audio(sig[s:s + 1280]) for s in range(0, len(sig), 1280)]
        res.llm.eval(agent.input_ids[-2:])
        runs.append((list(agent.input_ids), np.concatenate(outs), res.llm._scores[-1].copy(), list(agent.stats.event_prob.values),
                     res.audio_tokenizer.detokenize_context, res.audio_tokenizer.tokenize_context.copy()))
        assert agent.frame_graph_active == frame_graph
        # both codec windows are full after 2 s (25 frames): every later frame that stays on the steady-state path is one replay
        print("one-replay frames:", agent.duplex_graph_frames)
        assert (agent.duplex_graph_frames >= 30) if duplex_graph else (agent.duplex_graph_frames == 0)
    for other in runs[1:]:
        assert runs[0][0] == other[0]
        assert np.array_equal(runs[0][1], other[1]) and np.array_equal(runs[0][2], other[2])
        assert runs[0][3] == other[3] and runs[0][4] == other[4] and np.array_equal(runs[0][5], other[5])


def test_duplex_frame_call_equals_the_separate_calls():
    """rca_duplex_frame against the calls it fuses, on two handles over the same weights: codes of the user's window, sampled tokens,
    decode tail and the probe probability are the ones encode_tail -> frame -> decode_tail -> token_probs give, eagerly (first call
    of a shape) and replayed (later calls), and also when a step leaves audio mode inside the frame."""
    agent, res = make_agent(chunk=0.08)
    llm, at = res.llm, res.audio_tokenizer
    hip = at.codec_model.hip
    twin = llm.make_kv_shadow(low_priority=False)
    for m in (llm, twin):          # the same sampler, both draw counters at 0
        m.init_sampler_for_generate(top_k=100, top_p=1.0, min_p=0.0, temp=1.0, seed=11)
    base, floor = agent._code_token_base, agent.end_header_token_id
    rng = np.random.default_rng(5)
    prompt = [int(t) for t in rng.integers(base, base + at.codebook_size, size=40)]
    for m in (llm, twin):
        m.reset()
        m.eval(prompt[:-2])
    sig = rich_signal(32000 + 1280 * 6, 17)
    ctx = rng.integers(0, at.codebook_size, size=96).astype(np.int64)
    pair = prompt[-2:]
    for k in range(6):
        window = sig[1280 * (k + 1):1280 * (k + 1) + 32000]
        out = llm.duplex_frame(hip, window, ctx, 4, 1280 + 160, base, floor, agent.end_audio_token_id, pair)
        codes = hip.encode_tail(window[None], 4)[0]
        toks = twin.frame(pair, [base + int(c) for c in codes], floor)
        assert out["user_codes"] == [int(c) for c in codes] and out["tokens"] == toks
        assert len(toks) == 4
        new_codes = np.array([t - base for t in toks], dtype=np.int64)
        pcm = hip.decode_tail(np.concatenate([ctx, new_codes])[None], 1280 + 160)[0]
        assert np.array_equal(out["pcm"], pcm)
        assert out["probe_prob"] == float(twin.token_probs([agent.end_audio_token_id])[0])
        assert llm.n_tokens == twin.n_tokens
        ctx = np.concatenate([ctx, new_codes])[-96:]
        pair = [toks[-1], base + int(codes[-1])]
    # a floor above every token id: the first step "leaves audio mode" -> cut short, no PCM, state as after one step
    out = llm.duplex_frame(hip, sig[:32000], ctx, 4, 1280 + 160, base, llm.n_vocab(), -1, pair)
    toks = twin.frame(pair, [base + int(c) for c in hip.encode_tail(sig[None, :32000], 4)[0]], twin.n_vocab())
    assert out["tokens"] == toks and len(toks) == 1 and out["pcm"] is None and out["probe_prob"] is None
    assert llm.n_tokens == twin.n_tokens
    assert llm.step(pair) == twin.step(pair)       # the draw counters were put back alike
    twin.close()


def test_full_size_duplex_session_equals_the_oracle_session():
    """BASELINE config [3] at full size against the oracle, not only inside bench.py: the default codec (131072 x 16 codebook, hop 320) and
    the Llama-3.2-1B-dims LM (V = 259 344, hash-generated weights, text rows of lm_head zeroed like a trained codec LM in audio mode) run
    2.7 s of duplex audio through the SAME RealtimeAgent twice -- over the HIP objects (frame graph, streaming codec tails, graph steps) and
    over the CPU oracle objects (C codec oracle + LMRef + C sampler), greedy sampling.  Token stream, KV position and emitted PCM must
    be equal (ids exactly, PCM bit for bit); the last logits within the 1B decode tolerance."""
    from types import SimpleNamespace
    from agent_fakes import OracleCodecModel, OracleLLM
    from oracle import lm_ref
    from oracle.codec import OracleCodec
    from realtime_codec_agent_amd.audio_tokenizer import AudioTokenizer
    from realtime_codec_agent_amd.codec import MagiCodecHIP
    from realtime_codec_agent_amd.codec_model import CodecConfig, init_codec_weights
    from realtime_codec_agent_amd.llm import LlamaForAlternatingCodeChannels, LMConfig
    from realtime_codec_agent_amd.realtime_agent_config import RealtimeAgentConfig
    from realtime_codec_agent_amd.realtime_agent_v2 import RealtimeAgent
    from realtime_codec_agent_amd.tokenizer import CodecTokenizer
    ccfg = CodecConfig()
    cw = init_codec_weights(ccfg, seed=0)
    tok = CodecTokenizer(codebook_size=ccfg.codebook_size)
    lcfg = LMConfig.llama_3_2_1b()
    assert len(tok) <= lcfg.vocab_size
    w = lm_ref.random_weights(lcfg, 0, 0.02)                   # what rca_lm_create_random(seed 0) generates on the device
    head = w["lm_head.weight"].copy()
    head[: tok.codec_vocab_start] = 0
    head[len(tok):] = 0
    w["lm_head.weight"] = head
    hip_llm = LlamaForAlternatingCodeChannels(model_path="random:1b", config=lcfg, n_ctx=4096, random_seed=0, init_std=0.02, device=0)
    hip_llm.mask_head_rows(0, tok.codec_vocab_start)
    hip_llm.mask_head_rows(len(tok), lcfg.vocab_size)
    hip = SimpleNamespace(llm=hip_llm, aux_llm=None, tokenizer=tok, audio_tokenizer=AudioTokenizer(codec_model=MagiCodecHIP(ccfg, cw)),
                          whisper_model=None, llm_model_dir="")
    ora = SimpleNamespace(llm=OracleLLM(lcfg, w, n_ctx=4096), aux_llm=None, tokenizer=tok,
                          audio_tokenizer=AudioTokenizer(codec_model=OracleCodecModel(OracleCodec(ccfg, cw)), device="cpu"),
                          whisper_model=None, llm_model_dir="")
    cfg = dict(chunk_size_secs=0.08, use_whisper=False, force_trans_after_inactivity_secs=0.0, force_response_after_inactivity_secs=0.0, temperature=0.0)
    a_hip = RealtimeAgent(resources=hip, config=RealtimeAgentConfig(**cfg))
    a_ora = RealtimeAgent(resources=ora, config=RealtimeAgentConfig(**cfg))
    assert a_hip.input_ids == a_ora.input_ids                  # header incl. the 3 s enrollment encoded by both codecs
    sig = rich_signal(1280 * 34, 33)      # 34 frames: from the 25th on both codec windows are full and the HIP side runs ONE replay per frame
    for s in range(0, len(sig), 1280):
        o_hip = a_hip.process_audio(sig[s:s + 1280])
        o_ora = a_ora.process_audio(sig[s:s + 1280])
        assert a_hip.input_ids == a_ora.input_ids, f"token streams diverge in frame {s // 1280}"
        assert np.array_equal(o_hip, o_ora), f"emitted PCM differs in frame {s // 1280}"
        assert hip.llm.n_tokens == ora.llm.n_tokens
    d = np.abs(hip.llm._scores[-1] - ora.llm._logits).max()
    print(f"full-size session: {len(a_hip.input_ids)} tokens, last-step logits HIP vs oracle max|d| = {d:.2e}")
    assert d < 1.5e-3
    assert a_hip.frame_graph_active and not getattr(a_ora, "frame_graph_active", False)   # one graph replay per chunk on the HIP side
    print("one-replay frames at full size:", a_hip.duplex_graph_frames)
    assert a_hip.duplex_graph_frames >= 8 and a_ora.duplex_graph_frames == 0               # ... and per FRAME once the windows are full
    assert len(set(a_hip.input_ids[a_hip.context_start_pos + 8::2])) > 10
