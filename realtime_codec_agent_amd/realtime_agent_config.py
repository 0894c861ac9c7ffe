"""RealtimeAgentConfig -- the knobs of the duplex loop, as a documented table.

The reference declares these knobs as a plain dataclass (realtime_codec_agent/realtime_agent_config.py:5-59); drivers
construct it with keyword arguments (cli_benchmark.py:60-66, run_demo.py).  Here every knob is one row of `KNOBS` --
name, type, default, which part of this build consumes it, what it does -- and the dataclass is generated from the
table, so the keyword surface (names, defaults, positional order, validation errors) is the reference's while the
table doubles as documentation (`describe()`) and as the source for command-line flags.

consumer codes: loop = RealtimeAgent's frame loop, sampler = the device sampler (rca_lm_sampler_init), header = the
sequence header built at reset(), grammar = sequence grammar tokens, ext = external services (accepted for
compatibility; switching them on raises: network services are out of scope),
text = constrained text generation, prof = xRT profilers.
"""
from dataclasses import field, make_dataclass
from typing import Any, List, NamedTuple, Optional, Tuple

import numpy as np


class Knob(NamedTuple):
    name: str
    type: Any
    default: Any
    consumer: str
    doc: str


KNOBS: List[Knob] = [
    Knob("agent_opening_text", Optional[str], "hello?", "header", "what the agent says first (None: stay silent)"),
    Knob("agent_voice_enrollment", Optional[Tuple[int, np.ndarray]], None, "header", "(sr, pcm) voice sample encoded into the header"),
    Knob("agent_identity", str, "A", "header", "speaker label of the agent channel"),
    Knob("user_identity", str, "B", "header", "speaker label of the user channel"),
    Knob("temperature", float, 1.0, "sampler", "audio-token sampling temperature"),
    Knob("trans_temperature", float, 0.0, "sampler", "temperature while generating transcript text (0 = greedy)"),
    Knob("force_trans_after_inactivity_secs", float, 0.5, "loop", "force a user transcription after this much user silence (0: never)"),
    Knob("use_whisper", bool, True, "loop", "transcribe the user with an injected whisper object instead of the LM"),
    Knob("top_k", int, 100, "sampler", "keep the k most likely tokens"),
    Knob("top_p", float, 1.0, "sampler", "nucleus mass"),
    Knob("min_p", float, 0.0, "sampler", "drop tokens below min_p * p_max"),
    Knob("repeat_penalty", float, 1.0, "sampler", "accepted; penalties are off in every reference driver"),
    Knob("presence_penalty", float, 0.0, "sampler", "accepted; off"),
    Knob("frequency_penalty", float, 0.0, "sampler", "accepted; off"),
    Knob("chunk_size_secs", float, 0.1, "loop", "audio per process_audio call; whole [agent, user] frame pairs at 50 Hz"),
    Knob("chunk_fade_secs", float, 0.02, "loop", "equal-power crossfade between consecutive output chunks"),
    Knob("max_context_secs", float, 80.0, "loop", "audio context that triggers the sliding-window trim"),
    Knob("trim_by_secs", float, 20.0, "loop", "audio dropped from the front at each trim (followed by a KV recompute)"),
    Knob("target_volume_rms", float, 0.0, "loop", "normalise user audio to this RMS (0: leave as is)"),
    Knob("force_response_after_inactivity_secs", float, 3.0, "loop", "force an agent response after this much silence (0: never)"),
    Knob("finalize_response_after_inactivity_secs", float, 3.0, "loop", "close an open response after this much agent silence"),
    Knob("finalize_response_improbable_token_tolerance", int, 3, "loop", "improbable tokens tolerated when re-scoring a response"),
    Knob("seed", Optional[int], 42, "sampler", "RNG seed of the sampler"),
    Knob("header_audio_first_token", str, "<|audio_first|>", "grammar", ""),
    Knob("header_text_only_token", str, "<|text_only|>", "grammar", ""),
    Knob("header_agent_token", str, "<|agent|>", "grammar", ""),
    Knob("header_agent_voice_token", str, "<|agent_voice|>", "grammar", ""),
    Knob("header_speaker_token", str, "<|speaker|>", "grammar", ""),
    Knob("end_header_token", str, "<|end_header|>", "grammar", "every id above this one is an audio token"),
    Knob("start_audio_token", str, "<|audio|>", "grammar", ""),
    Knob("end_audio_token", str, "<|end_audio|>", "grammar", "its probability is the loop's turn-taking signal"),
    Knob("external_marker_token", str, "†", "grammar", "marks text produced by an external LLM"),
    Knob("use_external_llm", bool, False, "ext", ""),
    Knob("external_llm_api_key", Optional[str], "empty", "ext", ""),
    Knob("external_llm_base_url", Optional[str], "http://localhost:8080/v1", "ext", ""),
    Knob("external_llm_model", Optional[str], None, "ext", ""),
    Knob("external_llm_top_p", float, 0.95, "ext", ""),
    Knob("external_llm_instructions", Optional[str], None, "ext", ""),
    Knob("use_external_tts", bool, False, "ext", ""),
    Knob("external_tts_server_url", str, "http://localhost:8001", "ext", ""),
    Knob("external_tts_prompt_text", Optional[str], None, "ext", ""),
    Knob("external_tts_allow_fallback", bool, False, "ext", ""),
    Knob("constrain_allow_noise", bool, False, "text", "allow [noise] tags in generated text"),
    Knob("constrain_allow_breathing", bool, False, "text", "allow [breathing] tags"),
    Knob("constrain_allow_laughter", bool, True, "text", "allow [laughter] tags"),
    Knob("run_profilers", bool, True, "prof", "collect per-stage timings and the real-time factor"),
    Knob("profiler_report_interval_secs", float, 2.0, "prof", "window of the real-time-factor estimate"),
]


def _validate(self) -> None:
    frames_x2 = int(self.chunk_size_secs * 100)   # 50 Hz frames, two tokens (agent, user) per frame
    if frames_x2 % 2:
        raise ValueError("Chunk size must be a multiple of 0.02 seconds.")
    if self.chunk_size_secs < self.chunk_fade_secs:
        raise ValueError("Chunk fade length cannot be longer than the chunk size.")


def _describe(cls) -> str:
    """One line per knob: name, default, consumer, meaning."""
    return "\n".join(f"{k.name:46s} {k.default!r:28} [{k.consumer}] {k.doc}" for k in KNOBS)


RealtimeAgentConfig = make_dataclass(
    "RealtimeAgentConfig",
    [(k.name, k.type, field(default=k.default)) for k in KNOBS],
    namespace={"__post_init__": _validate, "describe": classmethod(_describe), "__doc__": __doc__},
)
RealtimeAgentConfig.__module__ = __name__   # picklable across RealtimeAgentMultiprocessing's process boundary
