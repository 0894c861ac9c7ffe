"""GPU parity tests of the LM step at the context lengths the duplex loop actually runs (MI355X).

The other LM tests compare with the oracle at <= 300 tokens (mid model), 96 + 4 tokens (1B dims) and ~440 tokens (full-size
session).  The bench's decode step runs at 2.2 k - 8.2 k tokens and the reference allocates n_ctx = 16384
(realtime_agent_resources.py:13; window of realtime_agent_config.py:23-24): the 32-split limit of the in-launch merge, the
separate combine launch behind it, the flash prefill over a multi-thousand-token prefix, the > 64-split tail of the merge and
RoPE at large positions were only ever compared with themselves.  Here they meet oracle.lm_ref.LMRef (fp32 torch, fp16 KV like
the device cache; pinned to the reference's codec_llama.py classes by tests/golden/lm_tiny.npz).

Tolerances are the existing ones of test_lm_gpu.py (TOL_1B: 1.5e-3 exact path, 3e-3 with bf16-MFMA prefill tiles), absolute, at
|logit| <= ~4.5 -- nothing loosened for the longer sums."""
import numpy as np
import pytest
import torch

from conftest import GOLDEN
from oracle import lm_ref

import lm_long_case as lc

pytestmark = pytest.mark.gpu

TOL = {False: 1.5e-3, True: 3e-3}      # == test_lm_gpu.TOL_1B


def _maxdiff(got, want):
    got, want = np.asarray(got, np.float32), np.asarray(want, np.float32)
    assert np.isfinite(got).all()
    return float(np.abs(got - want).max())


def test_wide_model_matches_oracle_from_1k_to_16k_tokens():
    """The 1B model's layer shape (its GEMV / attention instantiations and launch geometry), 2 layers, V = 8192, n_ctx = 20480.
    Two handles -- exact prefill (decode passes of two tokens) and bf16-MFMA prefill tiles with flash attention -- follow ONE
    oracle through checkpoints at 1 000, 2 100, 4 200, 8 100, 8 300, 10 000 and 16 500 tokens; at each checkpoint:
      * the last-token logits of the prefill that reached it (a long eval on top of a non-empty prefix),
      * an eager S = 2 eval, a graph-replayed S = 2 step (sampled token == the C sampler on the device's logits),
      * one 4-step frame() whose sampled tokens are fed back on the device; the oracle then evaluates the very sequence the
        device evaluated and the final logits are compared,
    each against the oracle's logits.  Buckets crossed: 4, 8 (1 000 -> 2 100), 16, 32 (4 200 -> 8 100: 8 kv heads x 32 splits = the
    last grid the in-launch merge takes), 64 (8 300, 10 000: separate combine launch), 80 of 80 (16 500: the merge's > 64-split tail).
    The oracle also shows that the tolerance has teeth: masking one 256-key split, or the partial split past 8 192, moves the
    logits by > 100x the tolerance."""
    from realtime_codec_agent_amd.llm import LlamaForAlternatingCodeChannels
    cfg = lc.wide_config()
    ids = lc.wide_ids().tolist()
    llms = {}
    for mfma in (False, True):
        llm = LlamaForAlternatingCodeChannels(model_path="random:wide", config=cfg, n_ctx=lc.WIDE_N_CTX, random_seed=lc.WIDE_SEED,
                                              init_std=lc.WIDE_STD, device=0)
        llm.set_mfma_prefill(mfma)
        llms[mfma] = llm
    ref = lm_ref.LMRef(cfg, lm_ref.random_weights(cfg, lc.WIDE_SEED, lc.WIDE_STD), kv_dtype=torch.float16)
    params = dict(top_k=50, top_p=1.0, min_p=0.0, temp=1.0)
    worst = {False: 0.0, True: 0.0}
    pos = 0
    for c in lc.WIDE_CHECKPOINTS:
        rows = []

        def check(tag, want):
            for mfma, llm in llms.items():
                d = _maxdiff(llm._scores[-1], want)
                rows.append(f"{tag} {'mfma ' if mfma else 'exact'} {d:.2e}")
                worst[mfma] = max(worst[mfma], d)
                assert d < TOL[mfma], (c, tag, mfma, d)
                assert int(np.argmax(llm._scores[-1])) == int(np.argmax(want)), (c, tag, mfma)

        # (1) prefill up to the checkpoint, on top of the prefix the previous checkpoint left
        want = ref.eval(ids[pos:c], last_only=True, chunk=1024)[-1].numpy()
        for llm in llms.values():
            llm.eval(ids[pos:c])
            assert llm.n_tokens == c
        check("prefill", want)
        # (2) eager S = 2 eval
        if c == 8300:   # teeth: a lost split would be seen (then rolled back: the dropped keys also change the later layers' K / V)
            full = ref.eval(ids[c:c + 2])[-1].numpy()
            for a, b in ((4096, 4352), (8192, 8300)):
                ref.set_n_tokens(c)
                lost = ref.eval(ids[c:c + 2], drop_keys=(a, b))[-1].numpy()
                assert _maxdiff(lost, full) > 100 * TOL[True], (a, b)
            ref.set_n_tokens(c)
        want = ref.eval(ids[c:c + 2])[-1].numpy()
        for llm in llms.values():
            llm.set_graphs(False)
            llm.eval(ids[c:c + 2])
        check("eager", want)
        # (3) graph-replayed S = 2 step with the sampler behind it
        want = ref.eval(ids[c + 2:c + 4])[-1].numpy()
        for llm in llms.values():
            llm.set_graphs(True)
            llm.init_sampler_for_generate(seed=c, **params)
            tok = llm.step(ids[c + 2:c + 4])
            assert tok == lm_ref.sample(llm._scores[-1], params["top_k"], 1.0, 0.0, 1.0, c, 0)
        check("graph step", want)
        # (4) one frame graph: 4 steps, sampled agent tokens fed back on the device
        users = ids[c + 6:c + 10]
        for mfma, llm in llms.items():
            llm.init_sampler_for_generate(seed=c + 1, **params)
            toks = llm.frame(ids[c + 4:c + 6], users, -1)
            assert len(toks) == 4 and llm.n_tokens == c + 12
            evaluated = ids[c + 4:c + 6] + [t for pair in zip(toks[:-1], users) for t in pair]
            ref.set_n_tokens(c + 4)
            want = ref.eval(evaluated)[-1].numpy()
            d = _maxdiff(llm._scores[-1], want)
            rows.append(f"frame {'mfma ' if mfma else 'exact'} {d:.2e}")
            worst[mfma] = max(worst[mfma], d)
            assert d < TOL[mfma], (c, "frame", mfma, d)
            assert toks[-1] == lm_ref.sample(llm._scores[-1], params["top_k"], 1.0, 0.0, 1.0, c + 1, 3)
            llm.n_tokens = c + 4
        ref.set_n_tokens(c + 4)
        pos = c + 4
        print(f"context {c:6d}: max|dlogit| " + "; ".join(rows))
    print(f"worst: exact {worst[False]:.3e} (tol {TOL[False]}), mfma prefill {worst[True]:.3e} (tol {TOL[True]})")
    for llm in llms.values():
        llm.close()


_ORACLE_CACHE = {}


def _long_1b_oracle(fmt):
    if fmt not in _ORACLE_CACHE:
        _ORACLE_CACHE.clear()           # one set of 259 344-wide points at a time
        _ORACLE_CACHE[fmt] = lc.long_1b_oracle_points(fmt)
    return _ORACLE_CACHE[fmt]


@pytest.mark.parametrize("mfma_prefill", [False, True])
@pytest.mark.parametrize("fmt", [None, "q4_k"])
def test_1b_logits_match_oracle_behind_a_2200_token_context(fmt, mfma_prefill):
    """BASELINE config 3's model (Llama-3.2-1B dims, V = 259 344, hash-generated weights) behind a 2 200-token context in the
    duplex grammar: the prefill's last-token logits and two S = 2 steps, ALL 259 344 logits against LMRef run live on the host,
    and against the committed top-100 / strided slices tests/golden/lm_1b_long_topk.npz / lm_1b_long_q4k_topk.npz
    (tests/golden/make_lm_1b_long_golden.py).  bf16 and Q4_K weights (the formats of the bench's duplex legs), both prefill modes,
    then the same steps through the captured graph."""
    import lm_1b_case as case
    from test_lm_gpu import _check_1b_point
    from realtime_codec_agent_amd.llm import LlamaForAlternatingCodeChannels
    fix = np.load(f"{GOLDEN}/" + ("lm_1b_long_q4k_topk.npz" if fmt else "lm_1b_long_topk.npz"))
    ctx, steps = lc.long_1b_ids()
    assert np.array_equal(ctx, fix["ctx_ids"]) and np.array_equal(np.stack(steps), fix["step_ids"])
    want = _long_1b_oracle(fmt)
    llm = LlamaForAlternatingCodeChannels(model_path="random:1b", config=case.config(), n_ctx=4096, random_seed=case.SEED,
                                          init_std=case.INIT_STD, device=0, weight_format=fmt)
    llm.set_mfma_prefill(mfma_prefill)
    tol = TOL[mfma_prefill]
    tag = f"{fmt or 'bf16'} 2.2k {'mfma-prefill' if mfma_prefill else 'exact'}"
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
