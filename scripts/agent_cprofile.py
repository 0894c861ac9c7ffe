"""cProfile of the duplex loop's host side (which Python work sits between the GPU calls)."""
import cProfile, os, pstats, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from realtime_codec_agent_amd.duplex_bench import synth_signal
from realtime_codec_agent_amd.llm import LMConfig
from realtime_codec_agent_amd.realtime_agent_config import RealtimeAgentConfig
from realtime_codec_agent_amd.realtime_agent_resources import RealtimeAgentResources
from realtime_codec_agent_amd.realtime_agent_v2 import RealtimeAgent

res = RealtimeAgentResources(llm_model_path="random:Llama-3.2-1B-codec", llm_n_ctx=16384, llm_config=LMConfig.llama_3_2_1b(), with_aux_llm=False)
config = RealtimeAgentConfig(chunk_size_secs=0.08, use_whisper=False, top_k=100, temperature=1.0, seed=42,
                             force_trans_after_inactivity_secs=0.0, force_response_after_inactivity_secs=0.0)
agent = RealtimeAgent(resources=res, config=config)
sig = synth_signal(int(30 * 16000), 0)
cs = agent.chunk_size_samples
for s in range(0, 20 * cs, cs):
    agent.process_audio(sig[s:s + cs])
pr = cProfile.Profile()
pr.enable()
n = 0
for s in range(20 * cs, len(sig) - cs + 1, cs):
    agent.process_audio(sig[s:s + cs]); n += 1
pr.disable()
st = pstats.Stats(pr)
st.sort_stats("tottime")
print("frames", n)
st.print_stats(22)
