"""Generates tests/golden/lm_tiny.npz by running the REFERENCE's own LM classes here.

The reference's realtime LM is llama.cpp on a GGUF produced from codec_llama.py after
persist_codec_embeddings; llama.cpp is absent offline, but
/root/reference/realtime_codec_agent/codec_llama.py is importable in this container.  This
script loads that file by path (no bytecode written, nothing copied), builds a tiny seeded
CodecLlamaForCausalLM, runs `persist_codec_embeddings` (codec_llama.py:178-206, unmodified) and
records logits.  CodecLlamaModel.forward itself cannot run under the installed transformers 5.x
(kwarg renames), so the embed-merge of codec_llama.py:107-112 is restated in six
lines with the reference's own modules and the installed LlamaModel.forward is driven with
inputs_embeds -- the identical layer stack.

Only data (weights, ids, logits) is written to the fixture.  Run from the repo root:
    python tests/golden/make_lm_golden.py
"""
import importlib.util
import os
import sys

sys.dont_write_bytecode = True

import numpy as np
import torch

OUT = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference/realtime_codec_agent/codec_llama.py"

ROPE_LLAMA3 = dict(rope_type="llama3", rope_theta=500000.0, factor=32.0, low_freq_factor=1.0,
                   high_freq_factor=4.0, original_max_position_embeddings=8192)


def load_ref():
    spec = importlib.util.spec_from_file_location("_ref_codec_llama", REF)
    m = importlib.util.module_from_spec(spec)
    sys.modules["_ref_codec_llama"] = m
    spec.loader.exec_module(m)
    return m


def to_bf16_representable(t: torch.Tensor) -> torch.Tensor:
    return t.to(torch.bfloat16).to(torch.float32)


def embed_merge(model, input_ids):
    """codec_llama.py:107-112 restated with the reference's own embedding modules."""
    mm = model.model
    cfg = model.config
    inputs_embeds = torch.empty(input_ids.shape + (cfg.hidden_size,), dtype=torch.float32)
    vocab_tokens = input_ids < cfg.codec_vocab_start
    codec_tokens = input_ids >= cfg.codec_vocab_start
    inputs_embeds[vocab_tokens] = mm.embed_tokens(input_ids[vocab_tokens])
    inputs_embeds[codec_tokens] = mm.embed_codec_tokens(input_ids[codec_tokens])
    return inputs_embeds


def run(model, input_ids=None, inputs_embeds=None, past=None):
    from transformers.models.llama import LlamaModel
    if inputs_embeds is None:
        inputs_embeds = model.model.embed_tokens(input_ids)
    out = LlamaModel.forward(model.model, inputs_embeds=inputs_embeds, past_key_values=past, use_cache=True)
    return model.lm_head(out.last_hidden_state), out.past_key_values


def main():
    ref = load_ref()
    torch.manual_seed(1234)
    base = dict(vocab_size=164, hidden_size=128, intermediate_size=256, num_hidden_layers=2, num_attention_heads=4,
                num_key_value_heads=2, head_dim=64, codebook_size=64, codebook_dim=16, codec_vocab_start=100,
                rms_norm_eps=1e-5, max_position_embeddings=16384, tie_word_embeddings=False, attention_bias=False,
                mlp_bias=False, pad_token_id=None)
    saved = {}
    state = None
    for tag, rope in (("default", dict(rope_type="default", rope_theta=500000.0)), ("llama3", ROPE_LLAMA3)):
        cfg = ref.CodecLlamaConfig(rope_parameters=rope, **base)
        model = ref.CodecLlamaForCausalLM(cfg).eval()
        if state is None:
            with torch.no_grad():
                for n, p in model.named_parameters():
                    if p.dim() >= 2:
                        p.copy_(to_bf16_representable(torch.randn_like(p) * 0.08))
                    elif "norm" in n:
                        p.copy_(to_bf16_representable(1.0 + 0.1 * torch.randn_like(p)))
                    else:
                        p.copy_(to_bf16_representable(0.02 * torch.randn_like(p)))
                model.set_codec_embeddings(torch.randn(64, 16))
            state = {k: v.clone() for k, v in model.state_dict().items()}
        else:
            model.load_state_dict(state)

        g = torch.Generator().manual_seed(7)
        header = torch.randint(0, 100, (9,), generator=g)
        pairs = torch.randint(100, 164, (20,), generator=g)  # [agent_t, user_t] codec-id pairs
        ids = torch.cat([header, pairs]).unsqueeze(0)

        with torch.no_grad():
            logits_codec, _ = run(model, inputs_embeds=embed_merge(model, ids))
            if tag == "default":   # the un-persisted pieces, for the on-device bake (rca_lm_persist_codec_embeddings)
                for k, v in model.state_dict().items():
                    if "embed_codec_tokens" in k:
                        saved["c:" + k] = v.float().numpy().copy()
            # deployment step of the reference: bake projected codec embeddings into the table
            model.persist_codec_embeddings(batch_size=16, show_progress=False)
            if tag == "default":   # the baked rows as the reference computed them (fp32, before any 16-bit storage)
                saved["persist_rows_f32"] = model.model.embed_tokens.weight.data[100:164].float().numpy().copy()
            # the baked rows are fp32 projector outputs; the deployed GGUF stores them as 16-bit floats.
            # Round them to bf16 so the HIP LM (bf16 weights) holds exactly the table used here.
            model.model.embed_tokens.weight.data = to_bf16_representable(model.model.embed_tokens.weight.data)
            logits_full, _ = run(model, input_ids=ids)
            # incremental: prefill 9 header ids (minus nothing), then S=2 evals as the agent does
            # (realtime_agent_v2.py:355: generate(input_ids[-2:]))
            lg, past = run(model, input_ids=ids[:, :9])
            steps = [lg[0, -1]]
            for i in range(9, 29, 2):
                lg, past = run(model, input_ids=ids[:, i:i + 2], past=past)
                steps.append(lg[0, -1])
            logits_steps = torch.stack(steps)
        inc_vs_full = (logits_steps[1:] - logits_full[0, 10::2]).abs().max().item()
        persist_delta = (logits_codec - logits_full).abs().max().item()
        print(tag, "incremental-vs-full max|d| =", inc_vs_full, " codec-vs-persisted(bf16 table) max|d| =", persist_delta)
        saved[f"logits_full_{tag}"] = logits_full[0].numpy()
        saved[f"logits_steps_{tag}"] = logits_steps.numpy()
        if tag == "default":
            sd = model.state_dict()
            for k, v in sd.items():
                if "embed_codec_tokens" in k or "rotary" in k:
                    continue
                v = v.float()
                assert torch.equal(v, to_bf16_representable(v)), k
                saved["w:" + k] = (v.contiguous().view(torch.int32) >> 16).to(torch.int16).numpy().view(np.uint16)
            saved["ids"] = ids[0].numpy().astype(np.int32)
    saved["config"] = np.array(repr(base))
    np.savez_compressed(os.path.join(OUT, "lm_tiny.npz"), **saved)
    print("wrote", os.path.join(OUT, "lm_tiny.npz"), os.path.getsize(os.path.join(OUT, "lm_tiny.npz")))


if __name__ == "__main__":
    main()
