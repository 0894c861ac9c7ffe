"""RealtimeAgentResources -- loads the model objects one duplex session needs.

Same constructor and attributes as realtime_codec_agent/realtime_agent_resources.py:9-49:
llm, aux_llm (logits_all twin used by finalize_last_response), tokenizer, audio_tokenizer,
whisper_model, llm_model_dir, clone_for_self_play().  Everything heavy is a HIP object on the
current GPU; whisper.cpp is out of scope, so `whisper_model` defaults to None
and an already constructed object may be passed in.
"""
import os
from typing import Any, Optional, Union

import torch

from .audio_tokenizer import AudioTokenizer
from .llm import LlamaForAlternatingCodeChannels, LMConfig
from .tokenizer import CodecTokenizer


class RealtimeAgentResources:
    def __init__(
        self,
        llm_model_path: str = "random:Llama-3.2-1B-magicodec-no-bpe-multi-131k-stereo",
        llm_n_ctx: int = 16384,
        codec_model: Union[str, Any] = "MagiCodec-50Hz-Base",
        codec_device: Optional[Union[str, torch.device]] = None,
        whisper_model: Optional[Any] = None,
        llm_config: Optional[LMConfig] = None,
        tokenizer: Optional[Any] = None,
        with_aux_llm: bool = True,
        llm_random_seed: int = 0,
        share_llm_weights_with: Optional[LlamaForAlternatingCodeChannels] = None,
        llm_weight_format: Optional[str] = None,
    ):
        self.llm_model_dir = os.path.dirname(llm_model_path) if not llm_model_path.startswith("random:") else ""
        kw = dict(model_path=llm_model_path, n_ctx=llm_n_ctx, n_gpu_layers=-1, verbose=False, flash_attn=True,
                  config=llm_config, random_seed=llm_random_seed, weight_format=llm_weight_format)
        if share_llm_weights_with is not None:   # clone_for_self_play: a fresh model state over the weights already on the device
            self.llm = LlamaForAlternatingCodeChannels(model_path=llm_model_path, n_ctx=llm_n_ctx, share_weights_with=share_llm_weights_with)
        else:
            self.llm = LlamaForAlternatingCodeChannels(**kw)
        # second instance with every position's logits (realtime_agent_resources.py:26-33): it shares the device weights of
        # `llm` (own KV cache and workspace only); only finalize_last_response uses it
        self.aux_llm = LlamaForAlternatingCodeChannels(logits_all=True, share_weights_with=self.llm, n_ctx=llm_n_ctx,
                                                       model_path=llm_model_path) if with_aux_llm else None
        if isinstance(whisper_model, str):
            raise NotImplementedError("whisper.cpp transcription is out of scope; pass a model object or None")
        self.whisper_model = whisper_model
        self.audio_tokenizer = AudioTokenizer(codec_model=codec_model, device=codec_device)
        if tokenizer is None and self.llm_model_dir and os.path.exists(os.path.join(self.llm_model_dir, "tokenizer.json")):
            # the reference keeps the trained HF fast tokenizer next to the model file (realtime_agent_resources.py:34)
            from transformers import AutoTokenizer
            tokenizer = AutoTokenizer.from_pretrained(self.llm_model_dir, local_files_only=True)
        if tokenizer is None:
            tokenizer = CodecTokenizer(codebook_size=self.audio_tokenizer.codebook_size, unicode_offset=self.audio_tokenizer.unicode_offset)
        self.tokenizer = tokenizer
        if len(self.tokenizer) > self.llm.n_vocab():
            raise ValueError(f"tokenizer has {len(self.tokenizer)} ids but the LM has {self.llm.n_vocab()} logits")
        if llm_model_path.startswith("random:"):
            # A trained codec LM in audio mode puts its probability mass on codec tokens.  Random-init weights
            # do not, so the text / padding rows of lm_head are zeroed: the top-k then holds codec tokens only
            # and the loop stays on its steady-state path.  Bytes streamed per step are unchanged.
            for m in (self.llm, self.aux_llm):
                if m is not None and hasattr(self.tokenizer, "codec_vocab_start"):
                    m.mask_head_rows(0, self.tokenizer.codec_vocab_start)
                    m.mask_head_rows(len(self.tokenizer), m.n_vocab())
        self._llm_config = self.llm.config
        self._llm_random_seed = llm_random_seed

    def clone_for_self_play(self) -> "RealtimeAgentResources":
        """Copy sharing everything except the LLM, which gets a fresh instance (reference :41-49) -- fresh KV cache, sampler
        and position, over the same device weights (the reference loads the file again)."""
        return RealtimeAgentResources(
            llm_model_path=self.llm.model_path or "random:clone",
            llm_n_ctx=self.llm.n_ctx(),
            codec_model=self.audio_tokenizer.codec_model,
            codec_device=self.audio_tokenizer.device,
            whisper_model=self.whisper_model,
            llm_config=self._llm_config,
            tokenizer=self.tokenizer,
            with_aux_llm=self.aux_llm is not None,
            llm_random_seed=self._llm_random_seed,
            share_llm_weights_with=self.llm,
        )
