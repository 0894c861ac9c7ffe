"""Which frames of the FIRST duplex session of a process are slow, and what they did: per frame the latency, whether the shadow cache fed
a prefill tile, whether the speculative <|end_audio|> step ran, the context length.  usage: duplex_tail_probe.py [secs] [sessions]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from realtime_codec_agent_amd.duplex_bench import synth_signal, session_resources_kwargs
from realtime_codec_agent_amd.realtime_agent_config import RealtimeAgentConfig
from realtime_codec_agent_amd.realtime_agent_resources import RealtimeAgentResources
from realtime_codec_agent_amd.realtime_agent_v2 import RealtimeAgent

secs = float(sys.argv[1]) if len(sys.argv) > 1 else 125.0
for sess in range(int(sys.argv[2]) if len(sys.argv) > 2 else 2):
    res = RealtimeAgentResources(**session_resources_kwargs(0, 16384, None))
    config = RealtimeAgentConfig(chunk_size_secs=0.08, use_whisper=False, top_k=100, temperature=1.0, seed=42, max_context_secs=80.0,
                                 trim_by_secs=20.0, force_trans_after_inactivity_secs=0.0, force_response_after_inactivity_secs=0.0)
    agent = RealtimeAgent(resources=res, config=config)
    sig = synth_signal(int(secs * 16000), 0)
    cs = agent.chunk_size_samples
    rows = []
    probes0 = 0
    orig = res.llm.step_probe
    def counted(*a, **k):
        global probes0
        probes0 += 1
        return orig(*a, **k)
    res.llm.step_probe = counted
    for i, s in enumerate(range(0, len(sig) - cs + 1, cs)):
        sh = getattr(agent, "_kv_shadow", None)
        tiles0 = sh.stats["tiles"] if sh is not None else 0
        p0 = probes0
        t0 = time.perf_counter()
        agent.process_audio(sig[s:s + cs])
        dt = (time.perf_counter() - t0) * 1e3
        sh = getattr(agent, "_kv_shadow", None)
        rows.append((i, dt, (sh.stats["tiles"] if sh is not None else 0) - tiles0, probes0 - p0, res.llm.n_tokens))
    lat = np.array([r[1] for r in rows[10:]])
    print(f"session {sess}: p50 {np.percentile(lat, 50):.2f} p99 {np.percentile(lat, 99):.2f} max {lat.max():.2f} ms")
    tile_f = np.array([r[1] for r in rows[10:] if r[2]]); plain = np.array([r[1] for r in rows[10:] if not r[2] and not r[3]]); pr = np.array([r[1] for r in rows[10:] if r[3] and not r[2]])
    print(f"   frames with a shadow tile: {len(tile_f)}, median {np.median(tile_f) if len(tile_f) else 0:.2f} max {tile_f.max() if len(tile_f) else 0:.2f};  with the speculative step: {len(pr)}, median {np.median(pr):.2f};  plain: {len(plain)}, median {np.median(plain):.2f} max {plain.max():.2f}")
    for r in sorted(rows[10:], key=lambda r: -r[1])[:8]:
        print(f"   frame {r[0]:5d} ({r[0] * 0.08:6.2f} s) {r[1]:6.2f} ms  tiles fed {r[2]}  speculative steps {r[3]}  context {r[4]}")
    del agent, res
