"""RealtimeAgentConfig -- the knobs of the duplex loop.

Field names, defaults and validation follow the reference dataclass
(realtime_codec_agent/realtime_agent_config.py:5-59) so existing drivers can pass the same
keyword arguments.  Only the groups marked HOT PATH are consumed by this build's
RealtimeAgent; the external-service groups are accepted and ignored unless switched on, in
which case the agent raises (those services are out of scope).
"""
from dataclasses import dataclass
from typing import Optional, Tuple

import numpy as np


@dataclass
class RealtimeAgentConfig:
    # --- identities / opening (sequence header, realtime_agent_v2.py:84-99)
    agent_opening_text: Optional[str] = "hello?"
    agent_voice_enrollment: Optional[Tuple[int, np.ndarray]] = None
    agent_identity: str = "A"
    user_identity: str = "B"
    # --- HOT PATH: sampler (llamacpp_utils.py:39-77)
    temperature: float = 1.0
    trans_temperature: float = 0.0
    force_trans_after_inactivity_secs: float = 0.5
    use_whisper: bool = True
    top_k: int = 100
    top_p: float = 1.0
    min_p: float = 0.0
    repeat_penalty: float = 1.0
    presence_penalty: float = 0.0
    frequency_penalty: float = 0.0
    # --- HOT PATH: framing / context window
    chunk_size_secs: float = 0.1
    chunk_fade_secs: float = 0.02
    max_context_secs: float = 80.0
    trim_by_secs: float = 20.0
    target_volume_rms: float = 0.0
    force_response_after_inactivity_secs: float = 3.0
    finalize_response_after_inactivity_secs: float = 3.0
    finalize_response_improbable_token_tolerance: int = 3
    seed: Optional[int] = 42
    # --- sequence grammar tokens (lm_dataset_builder.py:195-230)
    header_audio_first_token: str = "<|audio_first|>"
    header_text_only_token: str = "<|text_only|>"
    header_agent_token: str = "<|agent|>"
    header_agent_voice_token: str = "<|agent_voice|>"
    header_speaker_token: str = "<|speaker|>"
    end_header_token: str = "<|end_header|>"
    start_audio_token: str = "<|audio|>"
    end_audio_token: str = "<|end_audio|>"
    external_marker_token: str = "†"
    # --- external LLM (out of scope: network service)
    use_external_llm: bool = False
    external_llm_api_key: Optional[str] = "empty"
    external_llm_base_url: Optional[str] = "http://localhost:8080/v1"
    external_llm_model: Optional[str] = None
    external_llm_top_p: float = 0.95
    external_llm_instructions: Optional[str] = None
    # --- external TTS (out of scope: separate model + HTTP)
    use_external_tts: bool = False
    external_tts_server_url: str = "http://localhost:8001"
    external_tts_prompt_text: Optional[str] = None
    external_tts_allow_fallback: bool = False
    # --- constrained text generation
    constrain_allow_noise: bool = False
    constrain_allow_breathing: bool = False
    constrain_allow_laughter: bool = True
    # --- profiling (realtime_agent_profiler.py)
    run_profilers: bool = True
    profiler_report_interval_secs: float = 2.0

    def __post_init__(self):
        # a chunk must hold a whole number of [agent, user] frame pairs at 50 Hz
        if int(self.chunk_size_secs * 100) % 2 != 0:
            raise ValueError("Chunk size must be a multiple of 0.02 seconds.")
        if self.chunk_fade_secs > self.chunk_size_secs:
            raise ValueError("Chunk fade length cannot be longer than the chunk size.")
