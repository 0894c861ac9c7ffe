import sys, time, numpy as np
sys.path.insert(0, ".")
from realtime_codec_agent_amd.llm import LlamaForAlternatingCodeChannels, LMConfig
llm = LlamaForAlternatingCodeChannels(model_path="random:1b", n_ctx=512, device=0)
rng = np.random.default_rng(0)
H = llm.config.hidden
state = {"codec_embed.weight": rng.standard_normal((131072, 16), dtype=np.float32),
         "codebook_projectors.0.linear_1.weight": (0.1 * rng.standard_normal((H, 16))).astype(np.float32),
         "codebook_projectors.0.linear_1.bias": np.zeros(H, np.float32),
         "codebook_projectors.0.linear_2.weight": (0.02 * rng.standard_normal((H, H))).astype(np.float32),
         "codebook_projectors.0.linear_2.bias": np.zeros(H, np.float32)}
start = llm.config.vocab_size - 131072 - 8
t = time.perf_counter(); llm.persist_codec_embeddings(state, start); dt = time.perf_counter() - t
print(f"persist 131072 x 16 -> {H} -> {H}: {dt*1e3:.1f} ms (incl. 17 MB upload)")
t = time.perf_counter(); rows = llm.persist_codec_embeddings(state, start, return_f32=True); dt = time.perf_counter() - t
import torch
e = torch.from_numpy(state["codec_embed.weight"][:64]); h = torch.nn.functional.gelu(e @ torch.from_numpy(state["codebook_projectors.0.linear_1.weight"]).T)
want = (h @ torch.from_numpy(state["codebook_projectors.0.linear_2.weight"]).T).numpy()
print(f"with fp32 rows returned: {dt*1e3:.1f} ms; max|d| vs torch on 64 rows {np.abs(rows[:64]-want).max():.2e}")
