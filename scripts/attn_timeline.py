"""Per-phase timeline of the decode attention launch (diagnostic build: RCA_EXTRA_HIPCC_FLAGS=-DRCA_ATTN_TIMELINE RCA_LIB_PATH=/tmp/rca_tl.so).
usage (GPU box): RCA_EXTRA_HIPCC_FLAGS=-DRCA_ATTN_TIMELINE RCA_LIB_PATH=/tmp/rca_tl.so python scripts/attn_timeline.py [ctx]"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from realtime_codec_agent_amd import _native
assert os.environ.get('RCA_LIB_PATH'), 'set RCA_LIB_PATH=/tmp/<name>.so: the diagnostic build must not replace the in-tree library'
_native.build()
from realtime_codec_agent_amd.llm import LlamaForAlternatingCodeChannels, LMConfig
ctx = int(sys.argv[1]) if len(sys.argv) > 1 else 6600
cfg = LMConfig.llama_3_2_1b()
llm = LlamaForAlternatingCodeChannels(model_path="random:1b", config=cfg, n_ctx=16384, device=0)
llm.init_sampler_for_generate(top_k=100, top_p=1.0, min_p=0.0, temp=1.0, seed=42)
ids = np.random.default_rng(0).integers(128266, 259338, ctx + 2).tolist()
llm.eval(ids[:ctx]); llm.sync()
for _ in range(5):
    llm.n_tokens = ctx; llm.step(ids[ctx:ctx + 2])
lib = _native.lib()
assert lib.rca_debug_attn_timeline(None, 1) == 0
llm.n_tokens = ctx; llm.step(ids[ctx:ctx + 2]); llm.sync()     # the LAST layer's launch leaves its stamps
buf = np.zeros((1024, 16), np.int64)
assert lib.rca_debug_attn_timeline(buf.ctypes.data_as(C.c_void_p), 0) == 0
t = buf[buf[:, 0] > 0]
t0 = t[:, 0].min()
us = (t - t0) * 0.01
us[t == 0] = np.nan
names = ["entry", "V in regs (wave 0)", "S/softmax/PV issued (wave 0)", "wave merge + stores issued", "stores drained + barrier", "ticket returned", "merged (last arriver)",
         "barrier 1 passed (all waves' PV issued)", "wo written + barrier 2", "V in regs (wave 7)", "S/softmax/PV issued (wave 7)",
         "entry (wave 7)", "step state read (wave 0)", "K in regs, image requested (wave 0)"]
order = [0, 11, 12, 13, 1, 9, 2, 10, 7, 8, 3, 4, 5, 6]
print(f"{len(t)} workgroups (kv heads x splits) of the last layer, ctx {ctx}; microseconds after the first workgroup's entry")
for k in order:
    n = names[k]
    col = us[:, k][~np.isnan(us[:, k])]
    col = col[col >= 0]
    if len(col):
        print(f"  {n:42s} n={len(col):4d}  min {col.min():6.2f}  median {np.median(col):6.2f}  max {col.max():6.2f}")
