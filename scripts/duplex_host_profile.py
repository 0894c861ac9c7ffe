"""Where the host spends a duplex frame: cProfile over the bench's duplex loop (GPU box).  usage: duplex_host_profile.py [frames]"""
import cProfile, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from realtime_codec_agent_amd.duplex_bench import synth_signal
from realtime_codec_agent_amd.llm import LMConfig
from realtime_codec_agent_amd.realtime_agent_config import RealtimeAgentConfig
from realtime_codec_agent_amd.realtime_agent_resources import RealtimeAgentResources
from realtime_codec_agent_amd.realtime_agent_v2 import RealtimeAgent

frames = int(sys.argv[1]) if len(sys.argv) > 1 else 600
res = RealtimeAgentResources(llm_model_path="random:Llama-3.2-1B-codec", llm_n_ctx=16384, llm_config=LMConfig.llama_3_2_1b(), with_aux_llm=False)
config = RealtimeAgentConfig(chunk_size_secs=0.08, use_whisper=False, top_k=100, temperature=1.0, seed=42, max_context_secs=80.0, trim_by_secs=20.0,
                             force_trans_after_inactivity_secs=0.0, force_response_after_inactivity_secs=0.0)
agent = RealtimeAgent(resources=res, config=config)
cs = agent.chunk_size_samples
sig = synth_signal((frames + 60) * cs, 0)
for s in range(0, 50 * cs, cs):
    agent.process_audio(sig[s:s + cs])
pr = cProfile.Profile()
t0 = time.perf_counter()
pr.enable()
for s in range(50 * cs, (50 + frames) * cs, cs):
    agent.process_audio(sig[s:s + cs])
pr.disable()
dt = time.perf_counter() - t0
print(f"{frames} frames, {dt / frames * 1e3:.3f} ms per frame under the profiler")
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(40)
