"""Generates tests/golden/agent_*.npz by running the REFERENCE RealtimeAgent (loaded from
/root/reference by path, no bytecode written, nothing copied) with the deterministic fakes of
tests/agent_fakes.py.  The fixture pins this repo's control loop to the reference's own behaviour:
evaluated token sequence, KV positions, emitted audio, transcript.

Third-party modules the reference imports but never reaches with these fakes (librosa, codec_bpe,
llama_cpp, transformers.AutoTokenizer) are replaced by empty stand-in modules for the import only.
    python tests/golden/make_agent_golden.py
"""
import importlib.util
import os
import sys
import types

sys.dont_write_bytecode = True
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
from agent_fakes import build_fakes, scenarios, user_audio  # noqa: E402

REF = "/root/reference/realtime_codec_agent"


def load_reference_agent():
    for name in ("librosa", "codec_bpe", "codec_bpe.tools", "codec_bpe.tools.codec_utils", "llama_cpp"):
        sys.modules.setdefault(name, types.ModuleType(name))
    sys.modules["codec_bpe"].codes_to_chars = sys.modules["codec_bpe"].chars_to_codes = None
    sys.modules["codec_bpe"].UNICODE_OFFSET_LARGE = 0xE000
    sys.modules["codec_bpe.tools.codec_utils"].load_magicodec_model = None
    lc = sys.modules["llama_cpp"]
    lc.Llama = object
    lc.LogitsProcessorList = list
    lc.StoppingCriteriaList = list
    lc.LlamaGrammar = object
    pkg = types.ModuleType("_refpkg")
    pkg.__path__ = [REF]
    sys.modules["_refpkg"] = pkg
    sub = types.ModuleType("_refpkg.utils")
    sub.__path__ = [os.path.join(REF, "utils")]
    sys.modules["_refpkg.utils"] = sub

    def load(modname, relpath):
        path = os.path.join(REF, relpath)
        spec = importlib.util.spec_from_file_location(modname, path)
        m = importlib.util.module_from_spec(spec)
        sys.modules[modname] = m
        src = open(path, encoding="utf-8").read()
        # realtime_agent_v2.py:771 nests double quotes inside an f-string (Python >= 3.12 syntax); this
        # container runs 3.10, so that one expression is re-quoted IN MEMORY before compiling.
        src = src.replace('{entry["speaker"]}', "{entry['speaker']}")
        exec(compile(src, path, "exec"), m.__dict__)
        return m

    load("_refpkg.utils.audio_utils", "utils/audio_utils.py")
    load("_refpkg.utils.llamacpp_utils", "utils/llamacpp_utils.py")
    load("_refpkg.realtime_agent_config", "realtime_agent_config.py")
    load("_refpkg.realtime_agent_stats", "realtime_agent_stats.py")
    load("_refpkg.realtime_agent_profiler", "realtime_agent_profiler.py")
    res = types.ModuleType("_refpkg.realtime_agent_resources")
    res.RealtimeAgentResources = object
    sys.modules["_refpkg.realtime_agent_resources"] = res
    agent = load("_refpkg.realtime_agent_v2", "realtime_agent_v2.py")
    return agent, sys.modules["_refpkg.realtime_agent_config"]


def run(agent_cls, config_cls, name, cfg_kw, script, secs):
    resources, tok = build_fakes(script)
    agent = agent_cls(resources=resources, config=config_cls(**cfg_kw))
    n = int(secs * 16000)
    audio = user_audio(n)
    outs = []
    cs = agent.chunk_size_samples
    for s in range(0, n - cs + 1, cs):
        outs.append(agent.process_audio(audio[s:s + cs]))
    llm = resources.llm
    evals = [(a, np.array(t, np.int64)) for op, a, t in llm.log]
    return dict(
        input_ids=np.array(agent.input_ids, np.int64),
        audio_tokens_idx=np.array(agent.audio_tokens_idx, np.int64),
        out_audio_dec=np.concatenate(outs).astype(np.float32)[::7],
        history_dec=agent.get_audio_history().astype(np.float32)[:, ::11],
        n_out=np.int64(sum(len(o) for o in outs)),
        aux_calls=np.int64(resources.aux_llm.calls),
        eval_pos=np.array([a for a, _ in evals], np.int64),
        eval_len=np.array([len(t) for _, t in evals], np.int64),
        eval_tokens=np.concatenate([t for _, t in evals]),
        final_n_tokens=np.int64(llm.n_tokens),
        n_samples=np.int64(llm.n_samples),
        n_sampler_calls=np.int64(len(llm.sampler_calls)),
        transcript=np.array(agent.format_transcript()),
        sequence_tail=np.array(agent.get_sequence_str()[-200:]),
        event_prob=np.array([v[0] for v in agent.stats.event_prob.values], np.float64),
        total_secs=np.float64(agent.total_secs),
    )


def main():
    agent_mod, cfg_mod = load_reference_agent()
    _, tok = build_fakes()
    for name, (cfg_kw, script, secs) in scenarios(tok).items():
        out = run(agent_mod.RealtimeAgent, cfg_mod.RealtimeAgentConfig, name, cfg_kw, script, secs)
        np.savez_compressed(os.path.join(HERE, f"agent_{name}.npz"), **out)
        print(name, "ids", len(out["input_ids"]), "evals", len(out["eval_pos"]), "samples", int(out["n_samples"]),
              "transcript:", str(out["transcript"])[:120].replace("\n", " | "))


if __name__ == "__main__":
    main()
