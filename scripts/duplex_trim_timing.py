"""What a trim frame spends where: wraps KVShadow.finish / plan and the twin's eval_async / the swap with host timers (each followed by
a stream sync, so the numbers are the pieces' own durations, not their overlap).  usage: duplex_trim_timing.py [secs]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from realtime_codec_agent_amd.duplex_bench import synth_signal, session_resources_kwargs
from realtime_codec_agent_amd.realtime_agent_config import RealtimeAgentConfig
from realtime_codec_agent_amd.realtime_agent_resources import RealtimeAgentResources
from realtime_codec_agent_amd.realtime_agent_v2 import RealtimeAgent
from realtime_codec_agent_amd import kv_shadow

secs = float(sys.argv[1]) if len(sys.argv) > 1 else 125.0
res = RealtimeAgentResources(**session_resources_kwargs(0, 16384, None))
config = RealtimeAgentConfig(chunk_size_secs=0.08, use_whisper=False, top_k=100, temperature=1.0, seed=42, max_context_secs=80.0,
                             trim_by_secs=20.0, force_trans_after_inactivity_secs=0.0, force_response_after_inactivity_secs=0.0)
agent = RealtimeAgent(resources=res, config=config)
log = []
orig_finish, orig_plan = kv_shadow.KVShadow.finish, kv_shadow.KVShadow.plan
def timed_finish(self, input_ids, src_pos, prefix_len, end):
    twin, llm = self.twin, self.llm
    ev, sw = twin.eval_async, llm.swap_kv
    t = {}
    def ev2(toks):
        t0 = time.perf_counter(); ev(toks); t["enqueue_ms"] = (time.perf_counter() - t0) * 1e3; t["rest"] = len(toks)
        t0 = time.perf_counter(); twin.sync(); t["rest_gpu_ms"] = (time.perf_counter() - t0) * 1e3
    def sw2(other):
        t0 = time.perf_counter(); sw(other); t["swap_ms"] = (time.perf_counter() - t0) * 1e3
    twin.eval_async, llm.swap_kv = ev2, sw2
    t0 = time.perf_counter()
    try:
        r = orig_finish(self, input_ids, src_pos, prefix_len, end)
    finally:
        twin.eval_async, llm.swap_kv = ev, sw
    t["finish_ms"] = (time.perf_counter() - t0) * 1e3
    log.append(("finish", t))
    return r
def timed_plan(self, prefix_len, src_pos):
    t0 = time.perf_counter(); orig_plan(self, prefix_len, src_pos); self.twin.sync()
    log.append(("plan", {"plan_ms": (time.perf_counter() - t0) * 1e3, "prefix": prefix_len}))
kv_shadow.KVShadow.finish, kv_shadow.KVShadow.plan = timed_finish, timed_plan
sig = synth_signal(int(secs * 16000), 0)
cs = agent.chunk_size_samples
last_trim = agent.trim_to_secs
for i, s in enumerate(range(0, len(sig) - cs + 1, cs)):
    n0 = len(log)
    t0 = time.perf_counter()
    agent.process_audio(sig[s:s + cs])
    dt = (time.perf_counter() - t0) * 1e3
    if len(log) > n0:
        print(f"frame {i} ({i * 0.08:.2f} s) {dt:.2f} ms: " + "; ".join(f"{k} {({a: round(b, 2) for a, b in v.items()})}" for k, v in log[n0:]))
