"""MI355X-native duplex codec-LM hot path (drop-in surface of realtime_codec_agent).

Exports mirror realtime_codec_agent/__init__.py:1-5.  Heavy members are imported lazily so that
`import realtime_codec_agent_amd` works on a machine without a GPU (building, CPU tests).
"""
from importlib import import_module

_LAZY = {
    "AudioTokenizer": ".audio_tokenizer",
    "RealtimeAgentResources": ".realtime_agent_resources",
    "RealtimeAgentConfig": ".realtime_agent_config",
    "RealtimeAgent": ".realtime_agent_v2",
    "RealtimeAgentV2": ".realtime_agent_v2",
    "RealtimeAgentMultiprocessing": ".realtime_agent_v2",
    "LlamaForAlternatingCodeChannels": ".llm",
    "MagiCodecHIP": ".codec",
    "HipCodec": ".codec",
    "CodecConfig": ".codec_model",
    "add_common_inference_args": ".utils.cli_utils",
}


def __getattr__(name):
    if name in _LAZY:
        return getattr(import_module(_LAZY[name], __name__), name)
    raise AttributeError(name)


__all__ = sorted(_LAZY)
