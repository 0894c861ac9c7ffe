"""Duplex session long enough to trim: reports the frame-latency tail with KV recomputes in it.
usage: duplex_trim_probe.py [secs] [max_context_secs] [trim_by_secs]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from realtime_codec_agent_amd.duplex_bench import synth_signal
from realtime_codec_agent_amd.llm import LMConfig
from realtime_codec_agent_amd.realtime_agent_config import RealtimeAgentConfig
from realtime_codec_agent_amd.realtime_agent_resources import RealtimeAgentResources
from realtime_codec_agent_amd.realtime_agent_v2 import RealtimeAgent

secs = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
max_ctx = float(sys.argv[2]) if len(sys.argv) > 2 else 80.0
trim_by = float(sys.argv[3]) if len(sys.argv) > 3 else 20.0
res = RealtimeAgentResources(llm_model_path="random:Llama-3.2-1B-codec", llm_n_ctx=16384, llm_config=LMConfig.llama_3_2_1b(), with_aux_llm=False)
config = RealtimeAgentConfig(chunk_size_secs=0.08, use_whisper=False, top_k=100, temperature=1.0, seed=42, max_context_secs=max_ctx,
                             trim_by_secs=trim_by, force_trans_after_inactivity_secs=0.0, force_response_after_inactivity_secs=0.0)
agent = RealtimeAgent(resources=res, config=config)
sig = synth_signal(int(secs * 16000), 0)
cs = agent.chunk_size_samples
lat, trims = [], []
last_trim = agent.trim_to_secs
for s in range(0, len(sig) - cs + 1, cs):
    t0 = time.perf_counter()
    agent.process_audio(sig[s:s + cs])
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) * 1e3
    lat.append(dt)
    if agent.trim_to_secs != last_trim:
        trims.append((s / 16000.0, dt, res.llm.n_tokens))
        last_trim = agent.trim_to_secs
lat = np.array(lat[10:])
print(f"frames {len(lat)}  p50 {np.percentile(lat, 50):.2f} ms  p95 {np.percentile(lat, 95):.2f}  p99 {np.percentile(lat, 99):.2f}  max {lat.max():.1f} ms")
for t, dt, n in trims:
    print(f"  trim at {t:6.1f} s: frame took {dt:6.1f} ms, context after {n} tokens")
