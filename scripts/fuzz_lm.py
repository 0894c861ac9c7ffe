"""Fuzz the LM's consistency properties on the ~1B random-init model (no oracle needed at this size):
  exact mode  : any split of a token sequence into evals / graph steps / eager steps gives bit-identical logits;
  MFMA prefill: any split into evals > 8 tokens gives bit-identical logits, and they stay within tolerance of exact;
  rollback    : n_tokens -= k followed by re-eval reproduces the logits.
Contexts are drawn across the 256-key split and graph-bucket boundaries.  usage: fuzz_lm.py [seconds] [seed]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from realtime_codec_agent_amd.llm import LlamaForAlternatingCodeChannels, LMConfig

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
rng = np.random.default_rng(seed)
cfg = LMConfig.llama_3_2_1b()
llm = LlamaForAlternatingCodeChannels(model_path="random:fuzz", config=cfg, n_ctx=8192, device=0)
llm.init_sampler_for_generate(top_k=100, top_p=1.0, min_p=0.0, temp=1.0, seed=42)


def run(ids, cuts, mfma, graphs):
    """eval ids[:cuts[0]], ids[cuts[0]:cuts[1]], ...; 1-2 token pieces go through step() (graph or eager)."""
    llm.set_mfma_prefill(mfma); llm.set_graphs(graphs); llm.reset()
    pos = 0
    for c in list(cuts) + [len(ids)]:
        piece = ids[pos:c]
        if not piece:
            continue
        if len(piece) <= 2 and pos > 0:
            llm.step(piece)
        else:
            llm.eval(piece)
        pos = c
    return llm._scores[-1].copy() if len(ids) - (cuts[-1] if cuts else 0) > 2 or not cuts else np.array(llm._fetch_logits())


t0 = time.time()
cases = 0
worst = 0.0
while time.time() - t0 < budget:
    n = int(rng.choice([rng.integers(20, 300), rng.integers(250, 270), rng.integers(1020, 1030), rng.integers(2040, 2060), rng.integers(300, 3000)]))
    ids = rng.integers(128266, 259338, n).tolist()
    # exact mode: two random splittings, one ending in 2-token graph steps, one in eager steps
    k = int(rng.integers(1, 6))
    cuts_a = sorted(set(int(v) for v in rng.integers(1, n, k)))
    tail = max(1, n - 2 * int(rng.integers(1, 5)))
    cuts_b = sorted(set([int(v) for v in rng.integers(1, tail, k)] + list(range(tail, n, 2))))
    a = run(ids, cuts_a, False, True)
    # rollback on the exact cache: drop the last k tokens, evaluate them again
    kback = int(rng.integers(1, min(6, n - 1) + 1))
    llm.n_tokens = n - kback
    llm.eval(ids[n - kback:])
    assert np.array_equal(llm._scores[-1], a), ("rollback", n, kback)
    b = run(ids, cuts_b, False, bool(rng.integers(0, 2)))
    assert np.array_equal(a, b), ("exact", n, cuts_a, cuts_b)
    # MFMA prefill: two splittings whose pieces are all > 8 tokens (so every piece is tiled), same final piece rule
    def long_cuts():
        pts, p = [], 0
        while n - p > 40:
            p += int(rng.integers(9, 300)); 
            if p < n - 9: pts.append(p)
        return pts
    m1 = run(ids, long_cuts(), True, True)
    m2 = run(ids, long_cuts(), True, True)
    assert np.array_equal(m1, m2), ("mfma tiling", n)
    d = float(np.abs(m1 - a).max())
    worst = max(worst, d)
    assert d < 5e-3 * max(1.0, float(np.abs(a).max())), ("mfma vs exact", n, d)
    cases += 1
print(f"fuzz ok: {cases} sequences in {time.time() - t0:.0f} s, worst |mfma - exact| logit {worst:.2e} (seed {seed})")
