"""Bit-identity A/B of two builds of the library: prints one SHA-256 per scenario of the LM step (1B dims, random-init weights).
Run once per library and compare the outputs:
    RCA_LIB_PATH=scripts/dbg/librca_hip_r3.so python scripts/ab_logits.py > a.txt; python scripts/ab_logits.py > b.txt; diff a.txt b.txt
Scenarios: last-token logits after eager and graph S=2 / S=1 steps at contexts on both sides of the split / bucket limits, fused and
separate merge, a 4-step frame; sampled tokens included."""
import hashlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from realtime_codec_agent_amd.llm import LlamaForAlternatingCodeChannels, LMConfig

fmt = sys.argv[1] if len(sys.argv) > 1 else None
cfg = LMConfig.llama_3_2_1b()
llm = LlamaForAlternatingCodeChannels(model_path="random:1b", config=cfg, n_ctx=16384, random_seed=0, device=0, weight_format=fmt)
rng = np.random.default_rng(5)
ids = rng.integers(128266, 259338, 10300).tolist()
h = lambda a: hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()[:16]
pos = 0
for ctx in (250, 1000, 2047, 3000, 6600, 8190, 8300, 10000):
    llm.eval(ids[pos:ctx]); pos = ctx
    print(f"ctx {ctx} prefill {h(llm._scores[-1])}")
    for fuse in (True, False):
        for graphs in (True, False):
            for n in (2, 1):
                llm.set_attn_fuse(fuse); llm.set_graphs(graphs)
                llm.n_tokens = ctx
                llm.init_sampler_for_generate(top_k=100, top_p=1.0, min_p=0.0, temp=1.0, seed=9)
                toks, cur = [], ids[ctx:ctx + n]
                for s in range(3):
                    t = llm.step(cur); toks.append(t)
                    cur = [t, ids[ctx + 2 + s]][:n]
                print(f"ctx {ctx} fuse {int(fuse)} graphs {int(graphs)} n {n}: {toks} {h(llm._scores[-1])}")
    llm.set_attn_fuse(True); llm.set_graphs(True)
    llm.n_tokens = ctx
    llm.init_sampler_for_generate(top_k=100, top_p=1.0, min_p=0.0, temp=1.0, seed=11)
    toks = llm.frame(ids[ctx:ctx + 2], ids[ctx + 2:ctx + 6], -1)
    print(f"ctx {ctx} frame: {toks} {h(llm._scores[-1])}")
    llm.n_tokens = ctx
