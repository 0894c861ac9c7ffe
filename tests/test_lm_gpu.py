"""GPU parity tests (MI355X): the HIP LM step through the C ABI against (a) logits produced by the
reference's own codec_llama classes (golden fixture), (b) the torch fp32 oracle with the same fp16 KV
rounding, and (c) the sampler restatement.  Tolerances are stated per test."""
import numpy as np
import pytest
import torch

from conftest import GOLDEN
from oracle import lm_ref
from test_lm_cpu import tiny_cfg, tiny_weights

pytestmark = pytest.mark.gpu

# bf16 weights are exact in both paths; the HIP path keeps K/V in fp16 and accumulates dot products in
# a different order -> |dlogit| <= TOL_ORACLE vs the oracle with the same fp16 KV, and <= TOL_REF vs the
# reference's fp32-KV logits (tiny model, |logit| ~ 1).
# Evals longer than 8 tokens (prefill) run on bf16 MFMA with activations split into bf16 hi + lo (16 mantissa bits
# instead of 24): TOL_MFMA.  For scale: the reference's llama.cpp GEMM path rounds activations to fp16 (11 bits).
TOL_ORACLE = 2e-4
TOL_MFMA = 6e-4
TOL_REF = 2.5e-3


def make_llm(rope, n_ctx=512, logits_all=False):
    from realtime_codec_agent_amd.llm import LlamaForAlternatingCodeChannels
    w, ids = tiny_weights()
    return LlamaForAlternatingCodeChannels(config=tiny_cfg(rope), weights=w, n_ctx=n_ctx, logits_all=logits_all, device=0), w, ids


@pytest.mark.parametrize("rope", ["default", "llama3"])
@pytest.mark.parametrize("mfma_prefill", [False, True])
def test_logits_match_reference_fixture(rope, mfma_prefill):
    g = np.load(f"{GOLDEN}/lm_tiny.npz")
    llm, w, ids = make_llm(rope)
    llm.set_mfma_prefill(mfma_prefill)
    ref = lm_ref.LMRef(tiny_cfg(rope), w, kv_dtype=torch.float16)
    # the agent's pattern: prefill, then 2-token evals (realtime_agent_v2.py:100,355)
    llm.eval(ids[:9].tolist())
    got = [llm._scores[-1].copy()]
    want = [ref.eval(ids[:9])[-1].numpy()]
    for i in range(9, 29, 2):
        llm.eval(ids[i:i + 2].tolist())
        got.append(llm._scores[-1].copy())
        want.append(ref.eval(ids[i:i + 2])[-1].numpy())
    got, want = np.stack(got), np.stack(want)
    assert llm.n_tokens == 29
    d_oracle = np.abs(got - want).max()
    d_ref = np.abs(got - g[f"logits_steps_{rope}"]).max()
    print(f"max|dlogit| vs oracle(fp16 KV) {d_oracle:.3e}, vs reference fixture {d_ref:.3e}")
    assert d_oracle < (TOL_MFMA if mfma_prefill else TOL_ORACLE) and d_ref < TOL_REF
    assert (got.argmax(-1) == g[f"logits_steps_{rope}"].argmax(-1)).all()


def test_prefill_equals_incremental_bit_exact():
    """Size-independent property: one 29-token eval (chunks of 8) and the step-by-step evals leave the
    same KV cache and produce the same last logits bit for bit (accumulation order does not depend on
    how many tokens share a pass).  This is what makes recompute_kv_cache (realtime_agent_v2.py:725-733)
    invisible to the sampler."""
    llm, w, ids = make_llm("llama3")
    llm.set_mfma_prefill(False)   # exact mode: long evals walk the same GEMV kernels in 8-token chunks
    llm.eval(ids.tolist())
    a = llm._scores[-1].copy()
    llm.reset()
    llm.eval(ids[:9].tolist())
    for i in range(9, 29, 2):
        llm.eval(ids[i:i + 2].tolist())
    b = llm._scores[-1].copy()
    assert np.array_equal(a, b)
    # rollback by writing n_tokens (realtime_agent_v2.py:465,730), then re-eval: identical logits
    llm.n_tokens = 27
    llm.eval(ids[27:29].tolist())
    assert np.array_equal(llm._scores[-1], a)
    # one-token evals too
    llm.n_tokens = 20
    for i in range(20, 29):
        llm.eval([int(ids[i])])
    assert np.array_equal(llm._scores[-1], a)


def test_mfma_prefill_close_to_exact_path_and_oracle():
    """Evals longer than 8 tokens run as 32-token tiles on bf16 MFMA with hi/lo-split activations: same logits as
    the exact GEMV path within TOL_MFMA (tiny model), independent of how the tokens are cut into evals, and the KV
    cache it leaves supports bit-stable decode afterwards."""
    llm, w, ids = make_llm("llama3")
    ref = lm_ref.LMRef(tiny_cfg("llama3"), w, kv_dtype=torch.float16)
    want = ref.eval(ids)[-1].numpy()
    llm.set_mfma_prefill(False)
    llm.eval(ids.tolist())
    exact = llm._scores[-1].copy()
    llm.set_mfma_prefill(True)
    llm.reset()
    llm.eval(ids.tolist())                      # one 29-token tile
    a = llm._scores[-1].copy()
    llm.reset()
    llm.eval(ids[:13].tolist()); llm.eval(ids[13:].tolist())   # 13 + 16 tokens: two tiles
    b = llm._scores[-1].copy()
    assert np.array_equal(a, b)
    d1, d2 = np.abs(a - exact).max(), np.abs(a - want).max()
    print(f"mfma prefill vs exact path {d1:.3e}, vs oracle {d2:.3e}")
    assert d1 < TOL_MFMA and d2 < TOL_MFMA
    # decode on top of the MFMA-built cache: rollback + re-eval is bit-stable
    llm.eval([5, 120]); c1 = llm._scores[-1].copy()
    llm.n_tokens = 29
    llm.eval([5, 120])
    assert np.array_equal(llm._scores[-1], c1)


def test_get_logits_pointer_and_token_probs():
    llm, w, ids = make_llm("llama3")
    llm.eval(ids[:11].tolist())
    ptr = llm._ctx.get_logits()
    logits = np.ctypeslib.as_array(ptr, shape=(llm._n_vocab,))  # exactly what the agent does (realtime_agent_v2.py:449)
    assert np.array_equal(logits, llm._scores[-1])
    p = np.exp(logits - logits.max()); p /= p.sum()
    got = llm.token_probs([3, 120, 163])
    assert np.abs(got - p[[3, 120, 163]]).max() < 1e-6


def test_sampler_matches_restatement_and_graph_equals_eager():
    llm, w, ids = make_llm("llama3")
    params = dict(top_k=20, top_p=1.0, min_p=0.0, temp=1.0, seed=42)
    seq_graph, seq_eager = [], []
    for use_graph, out in ((True, seq_graph), (False, seq_eager)):
        llm.set_graphs(use_graph)
        llm.reset()
        llm.init_sampler_for_generate(**params)
        llm.eval(ids[:9].tolist())
        toks = ids[9:11].tolist()
        for step in range(24):
            t = llm.step(toks)
            logits = llm._scores[-1]
            want = lm_ref.sample(logits, 20, 1.0, 0.0, 1.0, 42, step)
            assert t == want, (step, t, want)
            out.append(t)
            toks = [t, int(ids[9 + (step % 20)])]
    assert seq_graph == seq_eager
    assert len(set(seq_graph)) > 3
    # generate() as the agent drives it: next(generate(ids, reset=False)) then drop the generator
    llm.set_graphs(True)
    llm.reset(); llm.init_sampler_for_generate(**params); llm.eval(ids[:9].tolist())
    t0 = next(llm.generate(ids[9:11].tolist(), reset=False))
    assert t0 == seq_graph[0] and llm.n_tokens == 11
    # greedy + logit bias (set_sampler(suppress_end_audio=True), realtime_agent_v2.py:172-185)
    from realtime_codec_agent_amd.llm import get_logits_bias_processor
    llm.init_sampler_for_generate(top_k=100, top_p=1.0, min_p=0.0, temp=0.0, seed=42)
    llm.n_tokens = 9
    g = llm.step(ids[9:11].tolist())
    assert g == int(llm._scores[-1].argmax())
    llm.init_sampler_for_generate(top_k=100, top_p=1.0, min_p=0.0, temp=0.0, seed=42, logits_processor=get_logits_bias_processor({g: -100}))
    llm.n_tokens = 9
    g2 = llm.step(ids[9:11].tolist())
    assert g2 != g and g2 == int(np.argsort(-llm._scores[-1], kind="stable")[1])


@pytest.mark.parametrize("min_p,temp", [(0.0, 1.0), (0.02, 0.7)])
def test_whole_vocabulary_sampler_matches_restatement(min_p, temp):
    """llama.cpp's top_k <= 0 = whole vocabulary (llamacpp_utils.py:39-77 passes top_k straight through): the device draws by the
    Gumbel-max rule over every token that passes min_p -- token for token the C restatement's choice, eagerly, as graph replays and
    inside a frame graph; switching between the top-k and the whole-vocabulary sampler on one handle keeps both right."""
    llm, w, ids = make_llm("llama3")
    params = dict(top_k=0, top_p=1.0, min_p=min_p, temp=temp, seed=23)
    seqs = []
    for use_graph in (True, False):
        llm.set_graphs(use_graph)
        llm.reset()
        llm.init_sampler_for_generate(**params)
        llm.eval(ids[:9].tolist())
        toks, out = ids[9:11].tolist(), []
        for step in range(24):
            t = llm.step(toks)
            want = lm_ref.sample(llm._scores[-1], 0, 1.0, min_p, temp, 23, step)
            assert t == want, (step, t, want)
            out.append(t)
            toks = [t, int(ids[9 + (step % 20)])]
        seqs.append(out)
    assert seqs[0] == seqs[1] and len(set(seqs[0])) > 6
    # the frame graph: 4 steps with the sampled token fed back on the device == the same 4 single steps
    llm.set_graphs(True)
    llm.reset(); llm.init_sampler_for_generate(**params); llm.eval(ids[:9].tolist())
    user = [int(ids[9 + i]) for i in range(4)]
    got = llm.frame(ids[9:11].tolist(), user, -1)
    assert got == [seqs[0][0]] + got[1:] and len(got) == 4
    llm.reset(); llm.init_sampler_for_generate(**params); llm.eval(ids[:9].tolist())
    toks, single = ids[9:11].tolist(), []
    for i in range(4):
        t = llm.step(toks)
        single.append(t)
        toks = [t, user[i]]
    assert got == single
    # back to top-k on the same handle (the graphs of the other sampler were dropped), then whole vocabulary again
    llm.reset(); llm.init_sampler_for_generate(top_k=20, top_p=1.0, min_p=0.0, temp=1.0, seed=42); llm.eval(ids[:9].tolist())
    t = llm.step(ids[9:11].tolist())
    assert t == lm_ref.sample(llm._scores[-1], 20, 1.0, 0.0, 1.0, 42, 0)
    llm.reset(); llm.init_sampler_for_generate(**params); llm.eval(ids[:9].tolist())
    assert llm.step(ids[9:11].tolist()) == seqs[0][0]


def test_sampler_penalties_match_restatement_and_leave_the_logits_raw():
    """repeat / frequency / presence penalties (realtime_agent_v2.py:172-185 forwards realtime_agent_config.py:18-20 to llama.cpp's
    penalties sampler, window = the last 64 tokens the sampler accepted): applied on the device in place, undone by the sampler's
    last kernel.  Token for token the C restatement's choice given the device's logits and the history of sampled tokens -- 150 steps
    (the 128-slot ring wraps, the 64-token window slides), eager and graph replays -- together with a logit bias; the logits the caller
    reads afterwards are the RAW ones (llama.cpp penalises its candidate copy, never the context's logits); a frame graph equals
    single steps, also when it is cut short (the speculative steps' tokens must leave the window again)."""
    from realtime_codec_agent_amd.llm import get_logits_bias_processor
    llm, w, ids = make_llm("llama3")
    ids = ids.tolist()
    pen = dict(repeat_penalty=1.3, frequency_penalty=0.25, presence_penalty=0.15)
    params = dict(top_k=20, top_p=1.0, min_p=0.0, temp=1.0, seed=77, **pen)
    seqs = []
    for use_graph in (True, False):
        llm.set_graphs(use_graph)
        llm.reset()
        llm.init_sampler_for_generate(logits_processor=get_logits_bias_processor({5: 2.0, 120: -3.0}), **params)
        llm.eval(ids[:9])
        toks, hist = ids[9:11], []
        for step in range(150):
            if step == 60:
                llm.n_tokens = 11          # keep the context inside n_ctx: roll the cache back, the sampler's history stays (as in llama.cpp)
            t = llm.step(toks)
            want = lm_ref.sample(llm._scores[-1], 20, 1.0, 0.0, 1.0, 77, step, {5: 2.0, 120: -3.0}, prev_tokens=hist, **pen)
            assert t == want, (use_graph, step, t, want)
            hist.append(t)
            toks = [t, ids[9 + (step % 20)]]
            if step == 100:
                llm.n_tokens = 11
        seqs.append(hist)
    assert seqs[0] == seqs[1]
    plain = []
    llm.reset(); llm.init_sampler_for_generate(top_k=20, top_p=1.0, min_p=0.0, temp=1.0, seed=77); llm.eval(ids[:9])
    toks = ids[9:11]
    for step in range(40):
        t = llm.step(toks); plain.append(t); toks = [t, ids[9 + (step % 20)]]
    assert plain != seqs[0][:40]                      # the penalties really change the stream
    # raw logits after a penalised draw: eval -> logits -> sample -> the same logits
    llm.reset(); llm.init_sampler_for_generate(**params); llm.eval(ids[:9]); llm.eval(ids[9:11])
    for _ in range(3):
        llm.sample()
    before = llm._scores[-1].copy()
    llm.sample()
    assert np.array_equal(llm._scores[-1], before)
    llm.n_tokens = 9; llm.eval(ids[9:11])
    assert np.array_equal(llm._scores[-1], before)
    # frame graph == single steps, whole and cut short
    llm.set_graphs(True)

    def run(use_frame, fl):
        llm.reset(); llm.init_sampler_for_generate(**params); llm.eval(ids[:9])
        for _ in range(5):                         # a history before the frame
            llm.sample()
        if use_frame:
            toks = llm.frame(ids[9:11], ids[11:15], fl)
        else:
            toks, cur = [], ids[9:11]
            for u in ids[11:15]:
                t = llm.step(cur); toks.append(t)
                if t <= fl:
                    break
                cur = [t, u]
        follow = [llm.step([toks[-1], ids[20]])]
        follow.append(llm.step([follow[0], ids[21]]))
        return toks, follow, llm.n_tokens
    whole = run(True, -1)
    assert whole == run(False, -1) and len(whole[0]) == 4
    fl = whole[0][1]                               # the second sampled token counts as "not audio": the frame is cut there (or at the first)
    cut = run(True, fl)
    assert cut == run(False, fl) and len(cut[0]) <= 2


@pytest.mark.parametrize("top_k,top_p,min_p,temp", [(1000, 1.0, 0.0, 1.0), (0, 0.9, 0.0, 1.0), (3000, 0.7, 0.01, 0.8), (-1, 0.35, 0.0, 1.2)])
def test_big_path_sampler_matches_restatement(top_k, top_p, min_p, temp):
    """top_k > 256 (a rank cut up to the vocabulary) and top_p < 1 without a small top_k -- values llamacpp_utils.py:39-77 passes
    straight to llama.cpp, refused here until round 4: the thresholds come from radix selects over the whole vocabulary (counts /
    fixed-point masses), the draw is the whole-vocabulary Gumbel-max pick above them.  Token for token the C restatement (which sorts
    the vocabulary and walks it) on an 8192-token model, eager, graph replays and a frame graph; with penalties on top; and the
    histograms are left clean for the next draw (a second sampler configuration on the same handle stays right)."""
    from realtime_codec_agent_amd.llm import LlamaForAlternatingCodeChannels, LMConfig
    cfg = LMConfig(vocab_size=8192, hidden=512, n_layers=2, n_heads=8, n_kv_heads=2, head_dim=64, ffn=1024)
    llm = LlamaForAlternatingCodeChannels(model_path="random:mid", config=cfg, n_ctx=512, random_seed=3, init_std=0.05, device=0)
    ids = np.random.default_rng(4).integers(0, 8192, 80).tolist()
    params = dict(top_k=top_k, top_p=top_p, min_p=min_p, temp=temp, seed=31)
    seqs = []
    for use_graph in (True, False):
        llm.set_graphs(use_graph)
        llm.reset(); llm.init_sampler_for_generate(**params); llm.eval(ids[:30])
        toks, out = ids[30:32], []
        for step in range(20):
            t = llm.step(toks)
            want = lm_ref.sample(llm._scores[-1], top_k, top_p, min_p, temp, 31, step)
            assert t == want, (use_graph, step, t, want)
            out.append(t)
            toks = [t, ids[32 + step]]
        seqs.append(out)
    assert seqs[0] == seqs[1] and len(set(seqs[0])) > 8
    llm.set_graphs(True)
    llm.reset(); llm.init_sampler_for_generate(**params); llm.eval(ids[:30])
    assert llm.frame(ids[30:32], ids[32:36], -1) == seqs[0][:4]
    # penalties on top of the big path
    pen = dict(repeat_penalty=1.2, frequency_penalty=0.3, presence_penalty=0.0)
    llm.reset(); llm.init_sampler_for_generate(**params, **pen); llm.eval(ids[:30])
    toks, hist = ids[30:32], []
    for step in range(12):
        t = llm.step(toks)
        assert t == lm_ref.sample(llm._scores[-1], top_k, top_p, min_p, temp, 31, step, prev_tokens=hist, **pen), step
        hist.append(t); toks = [t, ids[32 + step]]
    # another configuration on the same handle: the small-k chain, then the plain whole-vocabulary sampler
    llm.reset(); llm.init_sampler_for_generate(top_k=50, top_p=0.9, min_p=0.0, temp=1.0, seed=5); llm.eval(ids[:30])
    assert llm.step(ids[30:32]) == lm_ref.sample(llm._scores[-1], 50, 0.9, 0.0, 1.0, 5, 0)
    llm.reset(); llm.init_sampler_for_generate(top_k=0, top_p=1.0, min_p=0.0, temp=1.0, seed=5); llm.eval(ids[:30])
    assert llm.step(ids[30:32]) == lm_ref.sample(llm._scores[-1], 0, 1.0, 0.0, 1.0, 5, 0)
    llm.close()


def test_sampler_random_configurations_and_ties_match_restatement():
    """A sweep over the sampler surface the reference's config can reach (llamacpp_utils.py:39-77, realtime_agent_v2.py:172-185): 24
    seeded random configurations -- top_k anywhere from 1 to beyond the vocabulary or <= 0, top_p, min_p, temperature, the three
    penalties, a logit bias -- six graph steps each, token for token the C restatement given the device's logits and the history of
    accepted tokens.  Then the case a radix select can get wrong: 4096 of the 8192 logits EXACTLY equal (masked head rows) with the rank
    cut inside the tie group (ties rank by lowest index) and the mass cut behind it."""
    from realtime_codec_agent_amd.llm import LlamaForAlternatingCodeChannels, LMConfig, get_logits_bias_processor
    cfg = LMConfig(vocab_size=8192, hidden=512, n_layers=2, n_heads=8, n_kv_heads=2, head_dim=64, ffn=1024)
    llm = LlamaForAlternatingCodeChannels(model_path="random:mid", config=cfg, n_ctx=512, random_seed=3, init_std=0.05, device=0)
    ids = np.random.default_rng(4).integers(0, 8192, 80).tolist()
    rng = np.random.default_rng(2024)

    def run(p, bias, steps=6):
        llm.reset()
        llm.init_sampler_for_generate(logits_processor=get_logits_bias_processor(bias) if bias else None, **p)
        llm.eval(ids[:30])
        toks, hist = ids[30:32], []
        for step in range(steps):
            t = llm.step(toks)
            want = lm_ref.sample(llm._scores[-1], p["top_k"], p["top_p"], p["min_p"], p["temp"], p["seed"], step, bias or None, prev_tokens=hist,
                                 repeat_penalty=p["repeat_penalty"], frequency_penalty=p["frequency_penalty"], presence_penalty=p["presence_penalty"])
            assert t == want, (p, bias, step, t, want)
            hist.append(t)
            toks = [t, ids[32 + step]]
        return hist
    kinds = set()
    for i in range(24):
        top_k = int(rng.choice([1, 7, 100, 256, 257, 300, 1000, 4000, 8191, 8192, 20000, 0, -1]))
        p = dict(top_k=top_k, top_p=float(rng.choice([1.0, 1.0, 0.95, 0.6, 0.2, 0.02])), min_p=float(rng.choice([0.0, 0.0, 0.01, 0.2])),
                 temp=float(rng.choice([1.0, 0.7, 1.5])), seed=int(rng.integers(0, 1 << 31)),
                 repeat_penalty=float(rng.choice([1.0, 1.0, 1.15, 1.6])), frequency_penalty=float(rng.choice([0.0, 0.0, 0.3])),
                 presence_penalty=float(rng.choice([0.0, 0.0, 0.5])))
        bias = {int(rng.integers(0, 8192)): float(rng.normal(0, 3)) for _ in range(int(rng.integers(0, 3)))}
        run(p, bias)
        kinds.add(("serial" if 1 <= top_k <= 256 else "big" if (p["top_p"] < 1.0 or 256 < top_k < 8192) else "whole", p["repeat_penalty"] != 1.0 or p["frequency_penalty"] != 0.0 or p["presence_penalty"] != 0.0))
    assert {k for k, _ in kinds} == {"serial", "big", "whole"} and {b for _, b in kinds} == {True, False}, kinds
    # ties: 4096 logits exactly 0.0; about half of the others are positive, so a rank cut at 5000 ends INSIDE the tie group
    llm.mask_head_rows(0, 4096)
    for p in (dict(top_k=5000, top_p=1.0), dict(top_k=5000, top_p=0.97), dict(top_k=0, top_p=0.9), dict(top_k=7000, top_p=0.999)):
        full = dict(min_p=0.0, temp=2.0, seed=5, repeat_penalty=1.0, frequency_penalty=0.0, presence_penalty=0.0, **p)
        run(full, {}, steps=8)
    lg = llm._scores[-1]
    assert (lg[:4096] == 0).all() and 1000 < int((lg[4096:] > 0).sum()) < 3500
    llm.close()


def test_step_probe_equals_step_plus_token_probs():
    """rca_lm_step_probe (the agent's speculative <|end_audio|> step as one replay): same token, same probabilities, same state as
    step() followed by token_probs(), replayed and eager, across a graph-bucket boundary, and the rollback afterwards works."""
    llm, w, ids = make_llm("llama3")
    twin, _, _ = make_llm("llama3")
    params = dict(top_k=20, top_p=1.0, min_p=0.0, temp=1.0, seed=3)
    for use_graph in (True, False):
        for m in (llm, twin):
            m.set_graphs(use_graph)
            m.reset()
            m.init_sampler_for_generate(**params)
            m.eval(ids[:9].tolist())
        toks = ids[9:11].tolist()
        for step in range(12):
            probe = [int(ids[step]), int(ids[step + 3]), 5]
            t, p = llm.step_probe(toks[:1 + step % 2], probe)
            t2 = twin.step(toks[:1 + step % 2])
            p2 = twin.token_probs(probe)
            assert t == t2 and np.array_equal(p, p2) and llm.n_tokens == twin.n_tokens
            if step % 3 == 0:                       # the agent rolls the speculative token back
                llm.n_tokens -= 1
                twin.n_tokens -= 1
            toks = [t, int(ids[9 + step])]
        assert np.array_equal(llm._scores[-1], twin._scores[-1])


def test_logits_all_and_get_logprobs():
    llm, w, ids = make_llm("llama3", logits_all=True)
    ref = lm_ref.LMRef(tiny_cfg("llama3"), w, kv_dtype=torch.float16)
    lp = llm.get_logprobs(ids[:12].tolist(), ids[12:20].tolist())
    full = ref.eval(ids[:20]).numpy()
    want = torch.log_softmax(torch.from_numpy(full[11:19]), -1).numpy()[range(8), ids[12:20]]
    assert lp.shape == (8,) and np.abs(lp - want).max() < 1e-3


def test_step_graphs_survive_a_logits_all_excursion():
    """A captured step graph has the logits buffer's address baked into its head-GEMV and sampler nodes.  Switching the handle to
    logits_all and evaluating more rows than the buffer holds reallocates it; stepping afterwards must re-capture (not replay a
    graph over freed memory) and give exactly the eager result."""
    llm, w, ids = make_llm("llama3")
    params = dict(top_k=20, top_p=1.0, min_p=0.0, temp=1.0, seed=7)

    def run(excursion, graphs):
        llm.set_graphs(graphs)
        llm.reset()
        llm.init_sampler_for_generate(**params)
        llm.eval(ids[:9].tolist())
        out = [llm.step(ids[9:11].tolist())]
        if excursion:
            keep = llm.n_tokens
            llm._lib.rca_lm_set_logits_all(llm._h, 1)
            llm._logits_all = True
            llm.eval(ids[11:24].tolist())                      # 13 rows > the single row allocated: the buffer moves
            assert llm._scores.shape[0] == 13
            llm._lib.rca_lm_set_logits_all(llm._h, 0)
            llm._logits_all = False
            llm.n_tokens = keep
        out.append(llm.step([out[0], int(ids[11])]))
        out.append(llm.step([out[1], int(ids[12])]))
        return out, llm._scores[-1].copy(), llm.token_probs([3, 120])

    base = run(False, False)
    for excursion, graphs in ((True, True), (True, False), (False, True)):
        got = run(excursion, graphs)
        assert got[0] == base[0] and np.array_equal(got[1], base[1]) and np.array_equal(got[2], base[2]), (excursion, graphs)


def test_sampler_and_token_ids_are_validated():
    """top_k outside what the device sampler implements is refused, never clamped: more than 256 ranked candidates, and llama.cpp's
    'whole vocabulary' (top_k <= 0) combined with top_p < 1 (top_k <= 0 with top_p >= 1 is implemented, next test).  Token ids outside the
    vocabulary are an error, not a clamped embedding row."""
    from realtime_codec_agent_amd._native import RcaError
    llm, w, ids = make_llm("default")
    for ok in (dict(top_k=257, top_p=1.0), dict(top_k=0, top_p=0.9), dict(top_k=-1, top_p=0.5), dict(top_k=10 ** 6, top_p=0.5)):
        llm.init_sampler_for_generate(min_p=0.0, temp=1.0, seed=1, **ok)          # refused until round 4; llama.cpp honours them all
    with pytest.raises(RcaError):
        llm.init_sampler_for_generate(top_k=20, top_p=1.0, min_p=0.0, temp=1.0, seed=1, repeat_penalty=-1.0)
    with pytest.raises(NotImplementedError):
        llm.init_sampler_for_generate(top_k=20, top_p=1.0, min_p=0.0, temp=1.0, seed=1, typical_p=0.5)
    llm.init_sampler_for_generate(top_k=0, top_p=1.0, min_p=0.0, temp=0.0, seed=1)      # greedy ignores top_k
    llm.init_sampler_for_generate(top_k=256, top_p=1.0, min_p=0.0, temp=1.0, seed=1)
    llm.eval(ids[:4].tolist())
    for bad in ([llm._n_vocab], [-1], [3, 1 << 20]):
        with pytest.raises(RcaError):
            llm.eval(bad)
        with pytest.raises(RcaError):
            llm.step(bad)
    assert llm.n_tokens == 4


@pytest.mark.parametrize("mfma_prefill", [False, True])
def test_shadow_cache_swap_equals_fresh_recompute(mfma_prefill):
    """The sliding-window trim through a shadow cache (rca_lm_copy_kv / _eval_async / _swap_kv; kv_shadow.py) leaves the live handle in
    exactly the state of the reference's recompute (realtime_agent_v2.py:725-733: n_tokens = header, eval(surviving suffix)):
    next-step logits bit for bit equal, whatever way the suffix was cut into asynchronous evals, also after a rollback of the twin,
    and graph steps keep working on both caches."""
    llm, w, ids = make_llm("llama3")
    llm.set_mfma_prefill(mfma_prefill)
    ids = ids.tolist()
    hdr, suffix, nxt = ids[:7], ids[12:27], ids[27:29]
    llm.eval(ids[:27])                                      # the session so far: header + 20 tokens of dialogue
    twin = llm.make_kv_shadow()
    twin.set_mfma_prefill(mfma_prefill)
    twin.copy_kv_from(llm, len(hdr))
    twin.n_tokens = len(hdr)
    twin.eval_async(suffix[:9])                             # fed ahead of time, in two pieces ...
    llm.init_sampler_for_generate(top_k=20, top_p=1.0, min_p=0.0, temp=1.0, seed=5)
    t_live = llm.step(ids[27:29])                           # ... while the live handle keeps stepping on its own cache
    twin.eval_async(suffix[9:13])
    twin.eval_async([3, 4])                                 # two tokens that turn out wrong (the sequence was edited):
    twin.n_tokens = len(hdr) + 13                           # ... rolled back
    twin.eval_async(suffix[13:])                            # (pieces of a long eval keep the prefill arithmetic whatever their size)
    llm.swap_kv(twin)
    llm.n_tokens = len(hdr) + len(suffix)
    llm.eval(nxt)
    got = llm._scores[-1].copy()
    # the reference way on a fresh handle
    ref, _, _ = make_llm("llama3")
    ref.set_mfma_prefill(mfma_prefill)
    ref.eval(ids[:27])
    ref.n_tokens = len(hdr)
    ref.eval(suffix)
    ref.eval(nxt)
    assert np.array_equal(got, ref._scores[-1])
    # graph steps on the swapped-in cache, then swap back: the set captured over the first cache is still valid
    llm.init_sampler_for_generate(top_k=20, top_p=1.0, min_p=0.0, temp=1.0, seed=5)
    ref.init_sampler_for_generate(top_k=20, top_p=1.0, min_p=0.0, temp=1.0, seed=5)
    llm.n_tokens = ref.n_tokens = len(hdr) + len(suffix)
    assert llm.step(nxt) == ref.step(nxt)
    llm.swap_kv(twin)                                       # back on the original cache (27 tokens + the live step above)
    llm.n_tokens = 27
    llm.init_sampler_for_generate(top_k=20, top_p=1.0, min_p=0.0, temp=1.0, seed=5)
    assert llm.step(ids[27:29]) == t_live
    with pytest.raises(Exception):
        llm.swap_kv(make_llm("llama3", n_ctx=256)[0])       # different cache shape


def _frame_vs_steps(llm, ctx, pair, users, floor, params):
    """tokens of frame() vs the same steps taken one at a time (fresh sampler each), plus the logits / next token afterwards"""
    res = []
    for use_frame in (True, False):
        llm.reset()
        llm.init_sampler_for_generate(**params)
        llm.eval(ctx)
        if use_frame:
            toks = llm.frame(pair, users, floor)
        else:
            toks, cur = [], list(pair)
            for u in users:
                t = llm.step(cur)
                toks.append(t)
                if t <= floor:
                    break
                cur = [t, u]
        n_after = llm.n_tokens
        follow = [llm.step([toks[-1], users[len(toks) - 1]]) for _ in range(1)]       # the draw counter must be in the same place
        follow.append(llm.step([follow[0], users[0]]))
        res.append((toks, n_after, follow, llm._scores[-1].copy()))
    return res


@pytest.mark.parametrize("graphs", [True, False])
def test_frame_graph_equals_single_steps(graphs):
    """rca_lm_frame: 4 or 5 S=2 steps with the sampled agent token fed back on the device == the same steps one at a time
    (process_audio_input_ids, realtime_agent_v2.py:332-372): same tokens, KV position, draw counter and logits -- also when a step
    leaves audio mode in the middle of the frame (the frame is cut there and the caller continues step by step)."""
    llm, w, ids = make_llm("llama3")
    llm.set_graphs(graphs)
    ids = ids.tolist()
    params = dict(top_k=20, top_p=1.0, min_p=0.0, temp=1.0, seed=11)
    for n in (4, 5):
        a, b = _frame_vs_steps(llm, ids[:9], ids[9:11], ids[11:11 + n], -1, params)
        assert a[0] == b[0] and len(a[0]) == n and a[1] == b[1] == 9 + 2 * n
        assert a[2] == b[2] and np.array_equal(a[3], b[3])
    # cut short: choose the floor so that the SECOND sampled token is "not audio"
    full = _frame_vs_steps(llm, ids[:9], ids[9:11], ids[11:15], -1, params)[0][0]
    floor = full[1] if full[1] < full[0] else None
    if floor is None:
        floor = full[1]        # then the first one is cut too: still a valid case
    a, b = _frame_vs_steps(llm, ids[:9], ids[9:11], ids[11:15], floor, params)
    assert a[0] == b[0] and len(a[0]) < 4 and a[0][-1] <= floor
    assert a[1] == b[1] == 9 + 2 * len(a[0])
    assert a[2] == b[2] and np.array_equal(a[3], b[3])
    from realtime_codec_agent_amd._native import RcaError
    with pytest.raises(RcaError):
        llm.frame(ids[9:11], [1] * 9, -1)            # more steps than a frame graph holds
    with pytest.raises(RcaError):
        llm.frame(ids[9:11], [llm._n_vocab], -1)


def test_fused_attention_merge_equals_separate_merge_launch():
    """Decode steps merge the attention splits inside the attention launch: every workgroup publishes its partial with write-through
    stores and the one that arrives last merges them (sc1 hand-off).  Must equal the separate merge launch bit for bit: the 1B model,
    contexts on both sides of the 256-key split and the graph-bucket boundaries, 1- and 2-token steps, graph and eager, and a few
    hundred consecutive steps at 3 k context (a stale or torn partial would show up as a different token or logit)."""
    from realtime_codec_agent_amd.llm import LlamaForAlternatingCodeChannels, LMConfig
    cfg = LMConfig.llama_3_2_1b()
    llm = LlamaForAlternatingCodeChannels(model_path="random:1b", config=cfg, n_ctx=8192, random_seed=0, device=0)
    rng = np.random.default_rng(5)
    ids = rng.integers(128266, 259338, 3400).tolist()
    llm.eval(ids[:3000])
    params = dict(top_k=100, top_p=1.0, min_p=0.0, temp=1.0, seed=9)

    def run(fuse, graphs, start, steps, n):
        llm.set_attn_fuse(fuse)
        llm.set_graphs(graphs)
        llm.n_tokens = start
        llm.init_sampler_for_generate(**params)
        toks, cur = [], ids[start:start + n]
        for s in range(steps):
            t = llm.step(cur)
            toks.append(t)
            cur = ([t] + [ids[start + 2 + s]])[:n] if n == 2 else [t]
        return toks, llm._scores[-1].copy()
    for start in (250, 255, 256, 511, 1023, 1024, 2040, 2047):        # split and bucket boundaries
        for n in (1, 2):
            a = run(True, True, start, 3, n)
            b = run(False, True, start, 3, n)
            assert a[0] == b[0] and np.array_equal(a[1], b[1]), (start, n)
    a = run(True, True, 3000, 300, 2)
    b = run(False, False, 3000, 300, 2)
    assert a[0] == b[0] and np.array_equal(a[1], b[1])
    c = run(True, False, 3000, 40, 2)
    assert c[0] == a[0][:40]


def test_context_overflow_and_bad_args():
    from realtime_codec_agent_amd._native import RcaError
    llm, w, ids = make_llm("default", n_ctx=16)
    llm.eval(ids[:12].tolist())
    with pytest.raises(RcaError):
        llm.eval(ids[:8].tolist())
    with pytest.raises(RcaError):
        llm.n_tokens = 17
    llm.n_tokens = 0
    llm.eval(ids[:16].tolist())
    assert llm.n_tokens == 16


def test_mid_size_random_init_matches_oracle():
    """H=512, 4 layers, GQA 8/2, ffn 4096 (two K slices), V=8192: device-generated weights are
    regenerated on the CPU from the same hash; logits agree with the torch oracle."""
    from realtime_codec_agent_amd.llm import LlamaForAlternatingCodeChannels, LMConfig
    cfg = LMConfig(vocab_size=8192, hidden=512, n_layers=4, n_heads=8, n_kv_heads=2, head_dim=64, ffn=4096)
    llm = LlamaForAlternatingCodeChannels(model_path="random:mid", config=cfg, n_ctx=1024, random_seed=11, init_std=0.05, device=0)
    ref = lm_ref.LMRef(cfg, lm_ref.random_weights(cfg, 11, 0.05), kv_dtype=torch.float16)
    rng = np.random.default_rng(0)
    ids = rng.integers(0, 8192, 300)
    llm.eval(ids[:298].tolist())
    ref.eval(ids[:298])
    llm.eval(ids[298:300].tolist())
    got = llm._scores[-1]
    want = ref.eval(ids[298:300])[-1].numpy()
    d = np.abs(got - want).max()
    print(f"mid config max|dlogit| = {d:.3e} (|logit| max {np.abs(want).max():.2f})")
    assert d < 2e-3 * max(1.0, np.abs(want).max())
    assert got.argmax() == want.argmax()


def test_q8_0_decode_and_prefill_match_oracle_over_the_dequantised_model():
    """packed q8_0 weights, ONE copy.  weight_format='q8_0' quantises every projection and lm_head on the device with
    llama.cpp's rule; the decode GEMVs stream the packed blocks (8.5 bits / weight) and the prefill tiles de-quantise the same
    blocks while staging (d * q split into bf16 hi + lo).  Against LMRef over the SAME blocks de-quantised on the host
    (oracle/q8_ref.py): decode within the f32 summation-order tolerance, prefill tiles within the tolerance the bf16 model's tiles
    get (the activation split, not the weights, sets it); argmax equal; graph == eager; masked head rows read exactly zero."""
    from oracle import q8_ref
    from realtime_codec_agent_amd.llm import LlamaForAlternatingCodeChannels, LMConfig
    cfg = LMConfig(vocab_size=8192, hidden=512, n_layers=4, n_heads=8, n_kv_heads=2, head_dim=64, ffn=4096)
    llm = LlamaForAlternatingCodeChannels(model_path="random:mid", config=cfg, n_ctx=1024, random_seed=11, init_std=0.05, device=0, weight_format="q8_0")
    assert llm.weight_format == "q8_0" and llm.weight_bytes_per_step() < 0.54 * cfg.weight_bytes_per_step()
    llm.set_mfma_prefill(False)
    ref = lm_ref.LMRef(cfg, q8_ref.quantized_model(lm_ref.random_weights(cfg, 11, 0.05)), kv_dtype=torch.float16)
    ids = np.random.default_rng(0).integers(0, 8192, 300)
    llm.eval(ids[:39].tolist())
    ref.eval(ids[:39])
    llm.eval(ids[39:41].tolist())
    got = llm._scores[-1].copy()
    want = ref.eval(ids[39:41])[-1].numpy()
    d = np.abs(got - want).max()
    print(f"q8_0 decode vs LMRef over the de-quantised blocks: max|dlogit| = {d:.3e} (|logit| max {np.abs(want).max():.2f})")
    assert d < 1e-3 * max(1.0, np.abs(want).max()) and got.argmax() == want.argmax()
    # graph step == eager step on the packed weights
    llm.init_sampler_for_generate(top_k=50, top_p=1.0, min_p=0.0, temp=1.0, seed=3)
    llm.n_tokens = 39
    t_graph = llm.step(ids[39:41].tolist())
    assert np.array_equal(llm._scores[-1], got) and t_graph == lm_ref.sample(got, 50, 1.0, 0.0, 1.0, 3, 0)
    # the prefill tiles read the SAME blocks: 298 tokens through the MFMA tiles, then a decode pass on top of their cache
    llm.set_mfma_prefill(True)
    llm.reset(); ref.reset()
    llm.eval(ids[:298].tolist())
    ref.eval(ids[:298])
    llm.eval(ids[298:300].tolist())
    got2 = llm._scores[-1].copy()
    want2 = ref.eval(ids[298:300])[-1].numpy()
    d2 = np.abs(got2 - want2).max()
    print(f"q8_0 MFMA prefill (staged de-quantisation) + decode vs LMRef: max|dlogit| = {d2:.3e} (|logit| max {np.abs(want2).max():.2f})")
    assert d2 < 2e-3 * max(1.0, np.abs(want2).max()) and got2.argmax() == want2.argmax()
    llm.mask_head_rows(0, 100)
    llm.n_tokens = 298
    llm.eval(ids[298:300].tolist())
    assert np.all(llm._scores[-1][:100] == 0) and np.array_equal(llm._scores[-1][100:], got2[100:])


def test_f16_weights_stay_fp16_decode_and_prefill():
    """The reference's default model file is the F16 GGUF (realtime_agent_resources.py:12).  weight_format='f16' keeps every matrix
    as fp16 on the device (one copy): the decode GEMVs widen with v_cvt_f32_f16, the prefill tiles split each fp16 value exactly
    into bf16 hi + lo while staging.  Oracle: LMRef over the fp16 values."""
    from realtime_codec_agent_amd.llm import LlamaForAlternatingCodeChannels, LMConfig, bf16_bits_to_f32
    cfg = LMConfig(vocab_size=8192, hidden=512, n_layers=4, n_heads=8, n_kv_heads=2, head_dim=64, ffn=4096)
    llm = LlamaForAlternatingCodeChannels(model_path="random:mid", config=cfg, n_ctx=1024, random_seed=11, init_std=0.05, device=0, weight_format="f16")
    assert llm.weight_format == "f16" and llm.weight_bytes_per_step() == cfg.weight_bytes_per_step()
    w = lm_ref.random_weights(cfg, 11, 0.05)
    wf = {k: (bf16_bits_to_f32(v).astype(np.float16).astype(np.float32) if (k.endswith("_proj.weight") or k == "lm_head.weight") else v) for k, v in w.items()}
    changed = sum(int((bf16_bits_to_f32(w[k]) != wf[k]).sum()) for k in wf if k.endswith("_proj.weight"))
    assert changed > 0          # some bf16 values are not fp16 values (subnormal range): the conversion is a real one
    ref = lm_ref.LMRef(cfg, wf, kv_dtype=torch.float16)
    ids = np.random.default_rng(0).integers(0, 8192, 300)
    for mfma, n, tol in ((False, 41, 1e-3), (True, 300, 2e-3)):
        llm.set_mfma_prefill(mfma)
        llm.reset(); ref.reset()
        llm.eval(ids[:n - 2].tolist())
        ref.eval(ids[:n - 2])
        llm.eval(ids[n - 2:n].tolist())
        got = llm._scores[-1].copy()
        want = ref.eval(ids[n - 2:n])[-1].numpy()
        d = np.abs(got - want).max()
        print(f"f16 weights, mfma_prefill={mfma}: max|dlogit| = {d:.3e} (|logit| max {np.abs(want).max():.2f})")
        assert d < tol * max(1.0, np.abs(want).max()) and got.argmax() == want.argmax()


def test_q8_0_gguf_blocks_stay_packed_and_match_their_dequantisation(tmp_path):
    """A Q8_0 GGUF (what prep_test_model.sh:29 produces) through model_path=: the 34-byte blocks of the projection matrices go to the
    device as they are (RCA_Q8_0), are re-laid-out there and streamed packed; the embedding table is de-quantised exactly (f32 rows,
    as llama.cpp's get_rows does).  Logits equal LMRef over the file's own blocks de-quantised on the host -- table included --
    within the f32 summation-order tolerance."""
    import gguf_writer as gw
    from realtime_codec_agent_amd.gguf import load_llama_gguf
    from realtime_codec_agent_amd.llm import LlamaForAlternatingCodeChannels, bf16_bits_to_f32
    cfg = tiny_cfg("default")
    w, ids = tiny_weights()
    wf = {k: (bf16_bits_to_f32(v) if v.dtype == np.uint16 else v.astype(np.float32)) for k, v in w.items()}
    path = str(tmp_path / "tiny-q8.gguf")
    gw.write_llama_gguf(path, cfg, wf, matrix_type=gw.Q8_0)
    g = LlamaForAlternatingCodeChannels(model_path=path, n_ctx=512, device=0)
    assert g.weight_format == "q8_0"
    g.set_mfma_prefill(False)
    _, file_w, _ = load_llama_gguf(path)
    deq = {k: (v.dequantize() if hasattr(v, "dequantize") else v) for k, v in file_w.items() if k != "rope.inv_freq"}
    ref = lm_ref.LMRef(cfg, deq, kv_dtype=torch.float16)
    g.eval(ids.tolist())
    want = ref.eval(ids)[-1].numpy()
    d = np.abs(g._scores[-1] - want).max()
    print(f"Q8_0 GGUF, packed decode vs LMRef over the file's blocks: max|dlogit| = {d:.3e}")
    assert d < TOL_ORACLE and g._scores[-1].argmax() == want.argmax()


def test_f16_gguf_keeps_fp16_weights_and_an_exact_table(tmp_path):
    """An F16 GGUF (prep_test_model.sh:28; the reference's default, realtime_agent_resources.py:12) whose values are NOT bf16
    values: matrices stay fp16 on the device, the embedding table is widened exactly.  Logits equal LMRef over the file's fp16
    values within the f32 summation-order tolerance (a bf16 copy would be off by 2^-9 per weight)."""
    import gguf_writer as gw
    from realtime_codec_agent_amd.gguf import load_llama_gguf
    from realtime_codec_agent_amd.llm import LlamaForAlternatingCodeChannels, bf16_bits_to_f32
    cfg = tiny_cfg("default")
    w, ids = tiny_weights()
    rng = np.random.default_rng(5)
    wf = {k: (bf16_bits_to_f32(v) if v.dtype == np.uint16 else v.astype(np.float32)) for k, v in w.items()}
    wf = {k: (v * (1.0 + 0.003 * rng.standard_normal(v.shape)).astype(np.float32) if v.ndim == 2 else v) for k, v in wf.items()}   # off the bf16 grid
    path = str(tmp_path / "tiny-f16.gguf")
    gw.write_llama_gguf(path, cfg, wf, matrix_type=gw.F16)
    g = LlamaForAlternatingCodeChannels(model_path=path, n_ctx=512, device=0)
    assert g.weight_format == "f16"
    g.set_mfma_prefill(False)
    _, file_w, _ = load_llama_gguf(path)
    assert file_w["model.layers.0.mlp.up_proj.weight"].dtype == np.float16
    deq = {k: np.asarray(v, np.float32) for k, v in file_w.items() if k != "rope.inv_freq"}
    from realtime_codec_agent_amd.llm import f32_to_bf16_bits
    assert (bf16_bits_to_f32(f32_to_bf16_bits(deq["model.embed_tokens.weight"])) != deq["model.embed_tokens.weight"]).any()
    ref = lm_ref.LMRef(cfg, deq, kv_dtype=torch.float16)
    g.eval(ids.tolist())
    want = ref.eval(ids)[-1].numpy()
    d = np.abs(g._scores[-1] - want).max()
    print(f"F16 GGUF vs LMRef over the file's fp16 values: max|dlogit| = {d:.3e}")
    assert d < TOL_ORACLE and g._scores[-1].argmax() == want.argmax()


def test_q4_k_decode_and_prefill_match_oracle_over_the_dequantised_model():
    """SURVEY 8f-4, the reference's third deployed format (llama-quantize Q4_K_M, prep_test_model.sh:31): weight_format='q4_k' turns
    every projection and lm_head into GGUF Q4_K super-blocks on the device (this build's min / max rule; oracle/q4k_ref.py does the
    same on the host, block for block); the decode GEMVs stream the quad-interleaved nibbles (4.6 bits / weight), the prefill tiles
    de-quantise the same blocks while staging with llama.cpp's dequantize_row_q4_K arithmetic.  Against LMRef over the host's blocks
    de-quantised by that rule: decode and prefill within the tolerances the other formats get; argmax equal; graph == eager."""
    from oracle import q4k_ref
    from realtime_codec_agent_amd.llm import LlamaForAlternatingCodeChannels, LMConfig
    cfg = LMConfig(vocab_size=8192, hidden=512, n_layers=4, n_heads=8, n_kv_heads=2, head_dim=64, ffn=4096)
    llm = LlamaForAlternatingCodeChannels(model_path="random:mid", config=cfg, n_ctx=1024, random_seed=11, init_std=0.05, device=0, weight_format="q4_k")
    assert llm.weight_format == "q4_k" and llm.weight_bytes_per_step() < 0.30 * cfg.weight_bytes_per_step()
    llm.set_mfma_prefill(False)
    ref = lm_ref.LMRef(cfg, q4k_ref.quantized_model(lm_ref.random_weights(cfg, 11, 0.05)), kv_dtype=torch.float16)
    ids = np.random.default_rng(0).integers(0, 8192, 300)
    llm.eval(ids[:39].tolist())
    ref.eval(ids[:39])
    llm.eval(ids[39:41].tolist())
    got = llm._scores[-1].copy()
    want = ref.eval(ids[39:41])[-1].numpy()
    d = np.abs(got - want).max()
    print(f"q4_k decode vs LMRef over the de-quantised blocks: max|dlogit| = {d:.3e} (|logit| max {np.abs(want).max():.2f})")
    assert d < 1e-3 * max(1.0, np.abs(want).max()) and got.argmax() == want.argmax()
    llm.init_sampler_for_generate(top_k=50, top_p=1.0, min_p=0.0, temp=1.0, seed=3)
    llm.n_tokens = 39
    t_graph = llm.step(ids[39:41].tolist())
    assert np.array_equal(llm._scores[-1], got) and t_graph == lm_ref.sample(got, 50, 1.0, 0.0, 1.0, 3, 0)
    llm.set_mfma_prefill(True)
    llm.reset(); ref.reset()
    llm.eval(ids[:298].tolist())
    ref.eval(ids[:298])
    llm.eval(ids[298:300].tolist())
    got2 = llm._scores[-1].copy()
    want2 = ref.eval(ids[298:300])[-1].numpy()
    d2 = np.abs(got2 - want2).max()
    print(f"q4_k MFMA prefill (staged de-quantisation) + decode vs LMRef: max|dlogit| = {d2:.3e} (|logit| max {np.abs(want2).max():.2f})")
    assert d2 < 2e-3 * max(1.0, np.abs(want2).max()) and got2.argmax() == want2.argmax()
    llm.mask_head_rows(0, 100)
    llm.n_tokens = 298
    llm.eval(ids[298:300].tolist())
    assert np.all(llm._scores[-1][:100] == 0) and np.array_equal(llm._scores[-1][100:], got2[100:])


def test_q4_k_gguf_blocks_stay_packed_and_match_their_dequantisation(tmp_path):
    """A GGUF whose matrices are Q4_K (type 12, 144-byte super-blocks as llama-quantize lays them out) through model_path=: the blocks
    go to the device as they are (RCA_Q4_K), are re-laid-out there and streamed packed; Q/K rows un-permuted on the raw blocks; the
    embedding table is de-quantised exactly.  Logits equal LMRef over the file's own blocks de-quantised on the host."""
    import gguf_writer as gw
    from realtime_codec_agent_amd.gguf import load_llama_gguf
    from realtime_codec_agent_amd.llm import LlamaForAlternatingCodeChannels, LMConfig, bf16_bits_to_f32
    cfg = LMConfig(vocab_size=1024, hidden=256, n_layers=2, n_heads=4, n_kv_heads=2, head_dim=64, ffn=512, rope_scaling=None, rope_theta=10000.0)
    w = lm_ref.random_weights(cfg, 5, 0.05)
    wf = {k: (bf16_bits_to_f32(v) if v.dtype == np.uint16 else v.astype(np.float32)) for k, v in w.items()}
    path = str(tmp_path / "small-q4k.gguf")
    gw.write_llama_gguf(path, cfg, wf, matrix_type=gw.Q4_K)
    g = LlamaForAlternatingCodeChannels(model_path=path, n_ctx=512, device=0)
    assert g.weight_format == "q4_k"
    _, file_w, _ = load_llama_gguf(path)
    deq = {k: (v.dequantize() if hasattr(v, "dequantize") else v) for k, v in file_w.items() if k != "rope.inv_freq"}
    ref = lm_ref.LMRef(cfg, deq, kv_dtype=torch.float16)
    ids = np.random.default_rng(1).integers(0, 1024, 40)
    for mfma in (False, True):
        g.set_mfma_prefill(mfma)
        g.reset(); ref.reset()
        g.eval(ids.tolist())
        want = ref.eval(ids)[-1].numpy()
        d = np.abs(g._scores[-1] - want).max()
        print(f"Q4_K GGUF (mfma_prefill={mfma}) vs LMRef over the file's blocks: max|dlogit| = {d:.3e} (|logit| max {np.abs(want).max():.2f})")
        assert d < (2e-3 if mfma else 1e-3) * max(1.0, np.abs(want).max()) and g._scores[-1].argmax() == want.argmax()


def test_q4_k_m_gguf_mix_of_q4_k_and_q6_k_loads_and_matches_its_dequantisation(tmp_path):
    """The reference's third model file is `llama-quantize ... Q4_K_M` (prep_test_model.sh:31): a MIX -- output.weight and the attn_v /
    ffn_down tensors of the layers use_more_bits() picks are Q6_K (210-byte super-blocks), everything else Q4_K.  Such a file loads as
    it is: Q6_K tensors are re-encoded losslessly on the device (int8 values + f32 scale per 16, streamed by the q8_0 GEMV body), a
    layer whose attn_v differs in format from attn_q / attn_k runs its V projection as a matrix of its own (row base in the RoPE /
    KV-write epilogue).  Logits equal LMRef over the file's own blocks de-quantised by llama.cpp's rules, decode and prefill tiles;
    masked Q6_K head rows read exactly zero."""
    import gguf_writer as gw
    from realtime_codec_agent_amd._native import Q4KBlocks, Q6KBlocks
    from realtime_codec_agent_amd.gguf import load_llama_gguf
    from realtime_codec_agent_amd.llm import LlamaForAlternatingCodeChannels, LMConfig, bf16_bits_to_f32
    cfg = LMConfig(vocab_size=1024, hidden=256, n_layers=8, n_heads=4, n_kv_heads=2, head_dim=64, ffn=512, rope_scaling=None, rope_theta=10000.0)
    w = lm_ref.random_weights(cfg, 5, 0.05)
    wf = {k: (bf16_bits_to_f32(v) if v.dtype == np.uint16 else v.astype(np.float32)) for k, v in w.items()}
    path = str(tmp_path / "small-q4_k_m.gguf")
    gw.write_llama_gguf(path, cfg, wf, matrix_type="Q4_K_M")
    _, file_w, _ = load_llama_gguf(path)
    kinds = {k: type(v).__name__ for k, v in file_w.items()}
    assert kinds["lm_head.weight"] == "Q6KBlocks" and kinds["model.layers.0.self_attn.v_proj.weight"] == "Q6KBlocks"
    assert kinds["model.layers.1.self_attn.v_proj.weight"] == "Q4KBlocks" and kinds["model.layers.0.self_attn.q_proj.weight"] == "Q4KBlocks"
    assert sum(v == "Q6KBlocks" for v in kinds.values()) >= 1 + 2 * 3        # head + attn_v / ffn_down of at least three layers
    g = LlamaForAlternatingCodeChannels(model_path=path, n_ctx=512, device=0)
    assert g.weight_format == "q4_k"
    deq = {k: (v.dequantize() if hasattr(v, "dequantize") else v) for k, v in file_w.items() if k != "rope.inv_freq"}
    ref = lm_ref.LMRef(cfg, deq, kv_dtype=torch.float16)
    ids = np.random.default_rng(1).integers(0, 1024, 40)
    for mfma in (False, True):
        g.set_mfma_prefill(mfma)
        g.reset(); ref.reset()
        g.eval(ids.tolist())
        want = ref.eval(ids)[-1].numpy()
        d = np.abs(g._scores[-1] - want).max()
        print(f"Q4_K_M GGUF (mfma_prefill={mfma}) vs LMRef over the file's blocks: max|dlogit| = {d:.3e} (|logit| max {np.abs(want).max():.2f})")
        assert d < (2e-3 if mfma else 1e-3) * max(1.0, np.abs(want).max()) and g._scores[-1].argmax() == want.argmax()
    g.init_sampler_for_generate(top_k=50, top_p=1.0, min_p=0.0, temp=0.0, seed=1)
    g.n_tokens = 38
    assert g.step(ids[38:40].tolist()) == int(np.argmax(want))                 # graph step over the split V projection
    got = g._scores[-1].copy()
    g.mask_head_rows(0, 64)
    g.n_tokens = 38
    g.eval(ids[38:40].tolist())
    assert np.all(g._scores[-1][:64] == 0) and np.array_equal(g._scores[-1][64:], got[64:])


def test_q6_k_only_gguf_loads_reports_its_format_and_matches_its_dequantisation(tmp_path):
    """A `llama-quantize ... Q6_K` file: EVERY matrix is Q6_K, so the format the handle reports is that of its gate / up tensors
    (rca_lm_weight_format -> "q6_k"; a bare KeyError at the end of the constructor before round 4).  Logits equal LMRef over the
    file's own blocks de-quantised by llama.cpp's rule, decode and prefill tiles; weight_bytes_per_step() answers."""
    import gguf_writer as gw
    from realtime_codec_agent_amd.gguf import load_llama_gguf
    from realtime_codec_agent_amd.llm import LlamaForAlternatingCodeChannels, LMConfig, bf16_bits_to_f32
    cfg = LMConfig(vocab_size=1024, hidden=256, n_layers=2, n_heads=4, n_kv_heads=2, head_dim=64, ffn=512, rope_scaling=None, rope_theta=10000.0)
    w = lm_ref.random_weights(cfg, 9, 0.05)
    wf = {k: (bf16_bits_to_f32(v) if v.dtype == np.uint16 else v.astype(np.float32)) for k, v in w.items()}
    path = str(tmp_path / "small-q6_k.gguf")
    gw.write_llama_gguf(path, cfg, wf, matrix_type=gw.Q6_K)
    _, file_w, _ = load_llama_gguf(path)
    assert type(file_w["model.layers.0.mlp.gate_proj.weight"]).__name__ == "Q6KBlocks"
    g = LlamaForAlternatingCodeChannels(model_path=path, n_ctx=512, device=0)
    assert g.weight_format == "q6_k" and g.weight_bytes_per_step() > 0
    deq = {k: (v.dequantize() if hasattr(v, "dequantize") else v) for k, v in file_w.items() if k != "rope.inv_freq"}
    ref = lm_ref.LMRef(cfg, deq, kv_dtype=torch.float16)
    ids = np.random.default_rng(2).integers(0, 1024, 40)
    for mfma in (False, True):
        g.set_mfma_prefill(mfma)
        g.reset(); ref.reset()
        g.eval(ids.tolist())
        want = ref.eval(ids)[-1].numpy()
        d = np.abs(g._scores[-1] - want).max()
        print(f"Q6_K GGUF (mfma_prefill={mfma}) vs LMRef over the file's blocks: max|dlogit| = {d:.3e}")
        assert d < (2e-3 if mfma else 1e-3) * max(1.0, np.abs(want).max()) and g._scores[-1].argmax() == want.argmax()


def test_full_size_1b_properties():
    """BASELINE config 3 dims (Llama-3.2-1B, V=259344) with random-init weights: checks that do not
    need a CPU forward -- graph replay == eager, prefill == incremental, rollback, determinism."""
    from realtime_codec_agent_amd.llm import LlamaForAlternatingCodeChannels, LMConfig
    cfg = LMConfig.llama_3_2_1b()
    llm = LlamaForAlternatingCodeChannels(model_path="random:1b", config=cfg, n_ctx=2048, random_seed=0, device=0)
    rng = np.random.default_rng(1)
    ids = rng.integers(128266, 259338, 600).tolist()
    llm.init_sampler_for_generate(top_k=100, top_p=1.0, min_p=0.0, temp=1.0, seed=42)
    llm.set_mfma_prefill(False)
    llm.eval(ids[:520])
    a = llm._scores[-1].copy()
    assert np.isfinite(a).all() and a.std() > 0
    llm.reset()
    llm.eval(ids[:260]); llm.eval(ids[260:518]); llm.eval(ids[518:520])
    assert np.array_equal(llm._scores[-1], a)
    # bf16-MFMA prefill tiles: same logits up to the hi/lo split (|logit| ~ 1), independent of the tiling
    llm.set_mfma_prefill(True)
    llm.reset(); llm.eval(ids[:520]); m1 = llm._scores[-1].copy()
    llm.reset(); llm.eval(ids[:300]); llm.eval(ids[300:520]); m2 = llm._scores[-1].copy()
    assert np.array_equal(m1, m2)
    # a pass holds up to 1024 tokens (8 token blocks per GEMM launch): 1500 tokens as 1024 + 476, as 128-token pieces (the shadow
    # cache's feeding pattern) and as ragged pieces give the same bits
    long_ids = rng.integers(128266, 259338, 1500).tolist()

    def next_logits(pieces):
        llm.reset()
        off = 0
        for n in pieces:
            llm.eval_async(long_ids[off:off + n])
            off += n
        assert off == 1499
        llm.eval(long_ids[1499:])          # one decode pass over the cache the pieces built
        return llm._scores[-1].copy()
    one = next_logits([1499])
    assert np.array_equal(one, next_logits([128] * 11 + [91]))
    assert np.array_equal(one, next_logits([1030, 7, 1, 461]))
    llm.reset(); llm.eval(long_ids[:1499]); llm.eval(long_ids[1499:])
    assert np.array_equal(one, llm._scores[-1])
    d = np.abs(m1 - a).max()
    print(f"1B: mfma prefill vs exact GEMV path max|dlogit| = {d:.3e} (std {a.std():.3f})")
    assert d < 5e-3 * max(1.0, np.abs(a).max())
    llm.eval(ids[520:522])
    outs = []
    for use_graph in (True, False):
        llm.set_graphs(use_graph)
        llm.n_tokens = 520
        llm.init_sampler_for_generate(top_k=100, top_p=1.0, min_p=0.0, temp=1.0, seed=42)
        toks = ids[520:522]
        seq = []
        for s in range(6):
            t = llm.step(toks)
            seq.append(t)
            toks = [t, ids[530 + s]]
        outs.append(seq)
    assert outs[0] == outs[1]
    want = lm_ref.sample(llm._scores[-1], 100, 1.0, 0.0, 1.0, 42, 5)
    assert outs[1][-1] == want
    # the same six steps as one 4-step frame graph + two single steps
    llm.set_graphs(True)
    llm.n_tokens = 520
    llm.init_sampler_for_generate(top_k=100, top_p=1.0, min_p=0.0, temp=1.0, seed=42)
    seq = llm.frame(ids[520:522], ids[530:534], -1)       # (no lm_head rows are masked in this model: every id counts as audio here)
    assert seq == outs[0][:4] and llm.n_tokens == 528
    seq.append(llm.step([seq[-1], ids[533]]))
    seq.append(llm.step([seq[-1], ids[534]]))
    assert seq == outs[0]


# ~1B dims (H=2048, ffn 8192, 16 layers, V=259 344): |logit| <= ~4.5, std 0.9.  Exact mode sums 2048- and 8192-long dot
# products in a different order than torch (measured 3e-4); the MFMA prefill adds the bf16 hi/lo split (measured 8e-4).
TOL_1B = {False: 1.5e-3, True: 3e-3}


def _check_1b_point(got, want_full, fix, i, tol, tag, moment_tol=1e-4):
    import lm_1b_case as case
    got = np.asarray(got, np.float32)
    s = case.summarize(got)
    top_ids, top_vals = fix[f"p{i}/top_ids"], fix[f"p{i}/top_vals"]
    d_fix = max(np.abs(got[top_ids] - top_vals).max(), np.abs(s["strided"] - fix[f"p{i}/strided"]).max())
    d_live = np.abs(got - want_full).max()
    print(f"1B {tag} point {i}: max|dlogit| vs fixture slice {d_fix:.3e}, vs live oracle (all {got.size} logits) {d_live:.3e}")
    assert d_fix < tol and d_live < tol
    assert int(s["top_ids"][0]) == int(top_ids[0]) == int(np.argmax(want_full))
    # top-100 membership may differ only where the 100th / 101st logits are closer than the tolerance
    missing = set(top_ids.tolist()) - set(s["top_ids"].tolist())
    assert all(got[j] > s["top_vals"][-1] - 2 * tol for j in missing), missing
    assert len(missing) <= 3
    assert abs(s["std"] - float(fix[f"p{i}/std"])) < moment_tol and abs(s["mean"] - float(fix[f"p{i}/mean"])) < moment_tol


@pytest.mark.parametrize("mfma_prefill", [False, True])
def test_1b_logits_match_oracle_and_committed_slice(mfma_prefill):
    """BASELINE config 3's model at full size against oracle.lm_ref.LMRef on the same hash-generated weights: the last-token
    logits after a 96-token context (one eval: GEMV chunks in exact mode, one bf16-MFMA tile otherwise) and after two S=2
    decode steps -- all 259 344 of them against the oracle run live on the host, and the committed top-100 / strided
    slice (tests/golden/lm_1b_topk.npz, made by tests/golden/make_lm_1b_golden.py).  Also through the graph step."""
    import lm_1b_case as case
    from realtime_codec_agent_amd.llm import LlamaForAlternatingCodeChannels
    fix = np.load(f"{GOLDEN}/lm_1b_topk.npz")
    ctx, steps = case.token_ids()
    assert np.array_equal(ctx, fix["ctx_ids"]) and np.array_equal(np.stack(steps), fix["step_ids"])
    want = case.oracle_points()
    llm = LlamaForAlternatingCodeChannels(model_path="random:1b", config=case.config(), n_ctx=1024, random_seed=case.SEED,
                                          init_std=case.INIT_STD, device=0)
    llm.set_mfma_prefill(mfma_prefill)
    tol = TOL_1B[mfma_prefill]
    tag = "mfma-prefill" if mfma_prefill else "exact"
    llm.eval(ctx.tolist())
    _check_1b_point(llm._scores[-1], want[0], fix, 0, tol, tag)
    for i, s in enumerate(steps):
        llm.eval(s.tolist())
        _check_1b_point(llm._scores[-1], want[i + 1], fix, i + 1, tol, tag)
    # the same two steps as captured-graph steps (greedy): same logits bit for bit, token = argmax
    llm.n_tokens = len(ctx)
    llm.init_sampler_for_generate(top_k=100, top_p=1.0, min_p=0.0, temp=0.0, seed=42)
    eager = llm._scores[-1].copy()
    for i, s in enumerate(steps):
        tok = llm.step(s.tolist())
        assert tok == int(fix[f"p{i + 1}/top_ids"][0])
    assert np.array_equal(llm._scores[-1], eager)
    llm.close()


@pytest.mark.parametrize("mfma_prefill", [False, True])
@pytest.mark.parametrize("fmt", ["q8_0", "q4_k"])
def test_1b_quantised_logits_match_oracle(mfma_prefill, fmt):
    """The same case with weight_format='q8_0' / 'q4_k' at LMConfig.llama_3_2_1b(): the kernels the quantised duplex legs of bench.py run
    (lm_gemv_kernel<..., Q = q8_0> at K = 2048 / 8192, the packed head at V = 259 344, the prefill tiles de-quantising the blocks
    while staging) against LMRef over the same hash weights put through llama.cpp's quantize_row_q8_0 rule on the host
    (oracle/q8_ref.py; for q4_k: through oracle/q4k_ref.py's quantiser and llama.cpp's dequantize_row_q4_K) -- all 259 344 logits
    live, and the committed slices tests/golden/lm_1b_q8_topk.npz / lm_1b_q4k_topk.npz (make_lm_1b_golden.py q8_0 / q4_k); graph
    step == eager."""
    import lm_1b_case as case
    from realtime_codec_agent_amd.llm import LlamaForAlternatingCodeChannels
    fix = np.load(f"{GOLDEN}/" + {"q8_0": "lm_1b_q8_topk.npz", "q4_k": "lm_1b_q4k_topk.npz"}[fmt])
    ctx, steps = case.token_ids()
    assert np.array_equal(ctx, fix["ctx_ids"]) and np.array_equal(np.stack(steps), fix["step_ids"])
    want = case.oracle_points(fmt)
    llm = LlamaForAlternatingCodeChannels(model_path="random:1b", config=case.config(), n_ctx=1024, random_seed=case.SEED,
                                          init_std=case.INIT_STD, device=0, weight_format=fmt)
    assert llm.weight_format == fmt
    llm.set_mfma_prefill(mfma_prefill)
    tol = TOL_1B[mfma_prefill]
    tag = f"{fmt} mfma-prefill" if mfma_prefill else f"{fmt} exact"
    llm.eval(ctx.tolist())
    _check_1b_point(llm._scores[-1], want[0], fix, 0, tol, tag)
    for i, s in enumerate(steps):
        llm.eval(s.tolist())
        _check_1b_point(llm._scores[-1], want[i + 1], fix, i + 1, tol, tag)
    llm.n_tokens = len(ctx)
    llm.init_sampler_for_generate(top_k=100, top_p=1.0, min_p=0.0, temp=0.0, seed=42)
    eager = llm._scores[-1].copy()
    for i, s in enumerate(steps):
        tok = llm.step(s.tolist())
        assert tok == int(fix[f"p{i + 1}/top_ids"][0])
    assert np.array_equal(llm._scores[-1], eager)
    llm.close()


@pytest.mark.parametrize("rope", ["default", "llama3"])
def test_gguf_model_path_gives_the_same_logits(tmp_path, rope):
    """A GGUF written the way convert_hf_to_gguf.py writes a Llama (permuted Q/K rows, rope_freqs for llama3 scaling)
    and loaded through model_path= runs to the same logits, bit for bit, as the Hugging Face tensors it was made from."""
    import gguf_writer as gw
    from realtime_codec_agent_amd.llm import LMConfig, LlamaForAlternatingCodeChannels, bf16_bits_to_f32, rope_inv_freq
    llm, w, ids = make_llm(rope)
    llm.set_mfma_prefill(False)
    cfg = tiny_cfg(rope)
    wf = {k: (bf16_bits_to_f32(v) if v.dtype == np.uint16 else v.astype(np.float32)) for k, v in w.items()}
    factors = None
    if rope == "llama3":
        factors = rope_inv_freq(LMConfig(**{**cfg.__dict__, "rope_scaling": None})) / rope_inv_freq(cfg)
    path = str(tmp_path / "tiny.gguf")
    gw.write_llama_gguf(path, cfg, wf, matrix_type=gw.BF16, rope_freqs=factors)
    g = LlamaForAlternatingCodeChannels(model_path=path, n_ctx=512, device=0)
    g.set_mfma_prefill(False)
    llm.eval(ids.tolist()); g.eval(ids.tolist())
    a, b = llm._scores[-1], g._scores[-1]
    if rope == "default":
        assert np.array_equal(a, b)
    else:   # inv_freq goes through one extra divide/multiply: last-bit differences in the RoPE table
        assert np.abs(a - b).max() < 1e-4


def _codec_state():
    g = np.load(f"{GOLDEN}/lm_tiny.npz")
    pre = "c:model.embed_codec_tokens."
    return {k[len(pre):]: g[k] for k in g.files if k.startswith(pre)}, g


def test_persist_codec_embeddings_on_device():
    """codec_llama.py:178-206 on the device: start from a table whose codec rows are zero, bake the projector output and
    compare (a) the fp32 rows with the rows the reference's own persist_codec_embeddings produced (different summation
    order, erf implementation: 2e-6 on values of magnitude ~1), (b) the table with the bf16 rounding of those rows -- the
    reference's torch.equal check at :206 -- through bit-identical logits, (c) the logits with the reference fixture."""
    from realtime_codec_agent_amd.llm import LlamaForAlternatingCodeChannels, f32_to_bf16_bits
    state, g = _codec_state()
    w, ids = tiny_weights()
    blank = dict(w)
    tbl = w["model.embed_tokens.weight"].copy()
    tbl[100:164] = 0
    blank["model.embed_tokens.weight"] = tbl
    llm = LlamaForAlternatingCodeChannels(config=tiny_cfg("default"), weights=blank, n_ctx=512, device=0)
    llm.set_mfma_prefill(False)
    rows = llm.persist_codec_embeddings(state, codec_vocab_start=100, return_f32=True)
    d = np.abs(rows - g["persist_rows_f32"]).max()
    print(f"max|d| of the fp32 rows vs the reference's persist_codec_embeddings: {d:.2e} (max |row| {np.abs(rows).max():.2f})")
    assert d < 2e-6
    # the table now holds bf16(rows): same logits, bit for bit, as a model created with that table
    baked = dict(w)
    tbl2 = w["model.embed_tokens.weight"].copy()
    tbl2[100:164] = f32_to_bf16_bits(rows)
    baked["model.embed_tokens.weight"] = tbl2
    ref = LlamaForAlternatingCodeChannels(config=tiny_cfg("default"), weights=baked, n_ctx=512, device=0)
    ref.set_mfma_prefill(False)
    llm.eval(ids.tolist()); ref.eval(ids.tolist())
    assert np.array_equal(llm._scores[-1], ref._scores[-1])
    assert np.abs(llm._scores[-1] - g["logits_full_default"][-1]).max() < TOL_REF
    # bad arguments come back as errors, not faults
    from realtime_codec_agent_amd._native import RcaError
    with pytest.raises(RcaError):
        llm.persist_codec_embeddings(state, codec_vocab_start=120)   # 120 + 64 > vocabulary 164
    with pytest.raises(ValueError):
        llm.persist_codec_embeddings({**state, "codebook_projectors.0.linear_2.weight": np.zeros((64, 64), np.float32)}, 100)


def test_unpersisted_checkpoint_directory_loads(tmp_path):
    """A CodecLlamaForCausalLM checkpoint saved before persist_codec_embeddings (config.json with codec_vocab_start +
    safetensors carrying model.embed_codec_tokens.*) loads through model_path= and is baked on the way in."""
    import json
    from safetensors.torch import save_file
    from realtime_codec_agent_amd.llm import LlamaForAlternatingCodeChannels, bf16_bits_to_f32
    state, g = _codec_state()
    w, ids = tiny_weights()
    cfg = tiny_cfg("default")
    # a bf16 checkpoint, as the fixture's model is: persist writes the projector output into a bf16 table (the reference assigns
    # into embed_tokens.weight.data, codec_llama.py:199-204, i.e. in the table's dtype); an f32 checkpoint would keep f32 rows
    tensors = {k: np.ascontiguousarray(bf16_bits_to_f32(v) if v.dtype == np.uint16 else v, dtype=np.float32) for k, v in w.items()}
    tensors["model.embed_tokens.weight"][100:164] = 0
    tensors.update({"model.embed_codec_tokens." + k: np.ascontiguousarray(v, dtype=np.float32) for k, v in state.items()})
    tt = {k: (torch.from_numpy(v).to(torch.bfloat16) if (v.ndim == 2 and not k.startswith("model.embed_codec_tokens.")) else torch.from_numpy(v)) for k, v in tensors.items()}
    save_file(tt, str(tmp_path / "model.safetensors"))
    (tmp_path / "config.json").write_text(json.dumps(dict(
        vocab_size=cfg.vocab_size, hidden_size=cfg.hidden, num_hidden_layers=cfg.n_layers, num_attention_heads=cfg.n_heads,
        num_key_value_heads=cfg.n_kv_heads, head_dim=cfg.head_dim, intermediate_size=cfg.ffn, rms_norm_eps=cfg.rms_eps,
        rope_theta=cfg.rope_theta, codec_vocab_start=100, codebook_size=64, codebook_dim=16, num_codebooks=1, projector_hidden_act="gelu")))
    llm = LlamaForAlternatingCodeChannels(model_path=str(tmp_path), n_ctx=512, device=0)
    llm.eval(ids.tolist())
    assert np.abs(llm._scores[-1] - g["logits_full_default"][-1]).max() < TOL_REF
    assert llm._scores[-1].argmax() == g["logits_full_default"][-1].argmax()


def test_shared_weights_instance_is_an_independent_model_over_the_same_weights():
    """rca_lm_create_shared: the logits_all twin (aux_llm, realtime_agent_resources.py:26-33) borrows the device weights of
    `llm`.  Same logits bit for bit as a separately loaded instance, independent KV caches, and either handle may be closed first."""
    from realtime_codec_agent_amd.llm import LlamaForAlternatingCodeChannels
    llm, w, ids = make_llm("llama3")
    twin = LlamaForAlternatingCodeChannels(logits_all=True, share_weights_with=llm, n_ctx=512, device=0)
    solo = LlamaForAlternatingCodeChannels(config=tiny_cfg("llama3"), weights=w, n_ctx=512, logits_all=True, device=0)
    for m in (llm, twin, solo):
        m.set_mfma_prefill(False)
    llm.eval(ids[:9].tolist())                       # the parent is mid-sequence while the twin scores another one
    a = twin.get_logprobs(ids[:20].tolist(), ids[20:29].tolist())
    b = solo.get_logprobs(ids[:20].tolist(), ids[20:29].tolist())
    assert np.array_equal(np.asarray(a), np.asarray(b))
    llm.eval(ids[9:11].tolist())
    assert llm.n_tokens == 11 and twin.n_tokens != 11
    want = llm._scores[-1].copy()
    llm.close()                                       # owner first: the weights must outlive it
    twin.reset()
    twin.eval(ids[:11].tolist())
    assert np.array_equal(twin._scores[-1], want)
    twin.close()
    solo.close()
    with pytest.raises(Exception):
        LlamaForAlternatingCodeChannels(logits_all=True, share_weights_with=make_llm("default", n_ctx=256)[0], n_ctx=4096, device=0)


def test_fuzz_lm_short():
    """Half a minute of scripts/fuzz_lm.py on the ~1B model: random splits of random sequences into evals / graph steps /
    eager steps (exact mode: bit-identical), random MFMA tilings (bit-identical to each other, within tolerance of exact),
    rollback + re-eval, contexts across the 256-key split and graph-bucket boundaries."""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "scripts", "fuzz_lm.py"), "20", "11"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "fuzz ok" in r.stdout
