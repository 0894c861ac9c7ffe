"""Per-wave timeline of the flash prefill attention launch (diagnostic build, as scripts/attn_timeline.py).
usage (GPU box): RCA_EXTRA_HIPCC_FLAGS=-DRCA_ATTN_TIMELINE RCA_LIB_PATH=/tmp/rca_tl.so python scripts/flash_timeline.py [prefix] [tile]"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from realtime_codec_agent_amd import _native
assert os.environ.get('RCA_LIB_PATH'), 'set RCA_LIB_PATH: the diagnostic build must not replace the in-tree library'
_native.build()
from realtime_codec_agent_amd.llm import LlamaForAlternatingCodeChannels, LMConfig
prefix = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
tile = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
cfg = LMConfig.llama_3_2_1b()
llm = LlamaForAlternatingCodeChannels(model_path="random:1b", config=cfg, n_ctx=16384, device=0)
ids = np.random.default_rng(0).integers(128266, 259338, prefix + tile).tolist()
llm.eval(ids); llm.sync()          # warm
llm.n_tokens = 0
llm.eval(ids[:prefix]); llm.sync()
lib = _native.lib()
G = cfg.n_heads // cfg.n_kv_heads
n_wg = cfg.n_kv_heads * ((tile * G + 31) // 32)
NW = 3
lib.rca_debug_flash_timeline.argtypes = [C.c_void_p, C.c_int, C.c_int]
assert lib.rca_debug_flash_timeline(None, 1, n_wg * NW) == 0
llm.eval(ids[prefix:prefix + tile]); llm.sync()     # the LAST layer's launch leaves its stamps
buf = np.zeros((n_wg * NW, 16), np.int64)
assert lib.rca_debug_flash_timeline(buf.ctypes.data_as(C.c_void_p), 0, n_wg * NW) == 0
t = buf[buf[:, 0] > 0]
t0 = t[:, 0].min()
us = (t[:, :4] - t0) * 0.01
print(f"{len(t)} waves ({n_wg} workgroups x {NW}) of the last layer; prefix {prefix}, pass of {tile} tokens; microseconds after the first entry")
for k, n in enumerate(["entry", "loop start", "loop end", "exit (after merge barrier)"]):
    c = us[:, k]
    print(f"  {n:28s} min {c.min():7.2f}  p10 {np.percentile(c, 10):7.2f}  median {np.median(c):7.2f}  p90 {np.percentile(c, 90):7.2f}  max {c.max():7.2f}")
nb = t[:, 8].astype(float)
loop_us = us[:, 2] - us[:, 1]
print(f"  blocks per wave: min {nb.min():.0f} median {np.median(nb):.0f} max {nb.max():.0f};  loop time per block: median {np.median(loop_us / np.maximum(nb, 1)) * 1e3:.0f} ns")
tot = t[:, 4:8].sum(axis=1).astype(float)
print(f"  shader cycles per block (s_memtime): median {np.median(tot / np.maximum(nb, 1)):.0f}  -> clock ~ {np.median(tot / np.maximum(loop_us, 1e-3)) / 1e3:.2f} GHz")
for k, n in enumerate(["QK^T MFMAs + mask / max", "exp + sum", "tr reads issued + P split", "PV MFMAs + loop back"]):
    c = t[:, 4 + k] / np.maximum(nb, 1)
    print(f"    {n:28s} median {np.median(c):7.0f} cycles / block   p10 {np.percentile(c, 10):7.0f}  p90 {np.percentile(c, 90):7.0f}")
hw = t[:, 9]
cu = ((t[:, 11] & 15) << 8) | (((hw >> 13) & 3) << 4) | ((hw >> 8) & 15)
simd = (cu << 2) | ((hw >> 4) & 3)
u, cnt = np.unique(cu, return_counts=True)
print(f"  CUs used: {len(u)}; waves per CU min {cnt.min()} median {int(np.median(cnt))} max {cnt.max()}")
u2, cnt2 = np.unique(simd, return_counts=True)
print(f"  SIMDs used: {len(u2)}; waves per SIMD min {cnt2.min()} median {int(np.median(cnt2))} max {cnt2.max()}")
# concurrency: how many waves alive on a SIMD over time
ends = us[:, 3]
per_cu_end = np.array([ends[cu == c].max() for c in u])
print(f"  last exit per CU: min {per_cu_end.min():.1f} median {np.median(per_cu_end):.1f} max {per_cu_end.max():.1f} us")
blocks_cu = np.array([nb[cu == c].sum() for c in u])
print(f"  blocks per CU: min {blocks_cu.min():.0f} median {np.median(blocks_cu):.0f} max {blocks_cu.max():.0f}")
