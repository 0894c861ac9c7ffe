"""Duplex-stream leg of bench.py (BASELINE configs[2]/[3]): RealtimeAgent over the HIP AudioTokenizer and a
~1B random-init codec LM, fed the SURVEY.md 8d signal in 80 ms frames -- the cli_benchmark.py loop
(cli_benchmark.py:67-71) with device-synchronised timestamps.  Reports xRT by the reference's definition
(median of 2 s window means, realtime_agent_profiler.py:30-38,75) and p50/p95 process_audio latency."""
from __future__ import annotations

import time

import numpy as np


def synth_signal(n: int, seed: int = 0) -> np.ndarray:
    rng = np.random.default_rng(seed)
    t = np.arange(n) / 16000.0
    x = sum(0.1 * np.sin(2 * np.pi * f * t) for f in (220.0, 440.0, 1330.0)) + rng.normal(0.0, 0.01, n)
    return np.clip(x, -1.0, 1.0).astype(np.float32)


def run_duplex_bench(dev=None, secs: float = 20.0, chunk_size_secs: float = 0.08, n_ctx: int = 16384, lm_steps_probe: int = 64) -> dict:
    import torch
    from .llm import LMConfig
    from .realtime_agent_config import RealtimeAgentConfig
    from .realtime_agent_resources import RealtimeAgentResources
    from .realtime_agent_v2 import RealtimeAgent

    cfg = LMConfig.llama_3_2_1b()
    t0 = time.perf_counter()
    res = RealtimeAgentResources(llm_model_path="random:Llama-3.2-1B-codec", llm_n_ctx=n_ctx, llm_config=cfg, with_aux_llm=False)
    config = RealtimeAgentConfig(chunk_size_secs=chunk_size_secs, use_whisper=False, top_k=100, temperature=1.0, seed=42,
                                 force_trans_after_inactivity_secs=0.0, force_response_after_inactivity_secs=0.0)
    agent = RealtimeAgent(resources=res, config=config)
    load_s = time.perf_counter() - t0
    n = int(secs * 16000)
    sig = synth_signal(n, 0)
    cs = agent.chunk_size_samples
    # warm-up (graph capture, allocator) then restart the profilers
    for s in range(0, 10 * cs, cs):
        agent.process_audio(sig[s:s + cs])
    agent.profilers.reset()
    t1 = time.perf_counter()
    nchunks = 0
    for s in range(10 * cs, n - cs + 1, cs):
        out = agent.process_audio(sig[s:s + cs])
        nchunks += 1
    torch.cuda.synchronize()
    wall = time.perf_counter() - t1
    summ = agent.profilers.summary()
    # raw LM step rate at the current context length (S=2 per step, graph replay)
    llm = res.llm
    n0 = llm.n_tokens
    toks = [agent.input_ids[-2], agent.input_ids[-1]]
    llm.sync()
    t2 = time.perf_counter()
    for _ in range(lm_steps_probe):
        llm.n_tokens = n0 - 2
        llm.step(toks)
    llm.sync()
    lm_ms = (time.perf_counter() - t2) * 1e3 / lm_steps_probe
    wbytes = cfg.weight_bytes_per_step()
    kv_bytes = 2 * 2 * cfg.n_layers * cfg.n_kv_heads * cfg.head_dim * n0
    out_ids_ok = all(t > agent.end_header_token_id for t in agent.input_ids[-8:])
    return {
        "workload": f"1 duplex stream, Llama-3.2-1B dims (V={cfg.vocab_size}) random-init bf16, {int(chunk_size_secs * 1000)} ms frames, "
                    f"top_k=100 T=1.0 seed=42, {secs:.0f} s of audio",
        "xRT": summ["total"]["xrt_median"],
        "xRT_wall": nchunks * chunk_size_secs / wall,
        "p50_frame_step_ms": summ["total"].get("p50"),
        "p95_frame_step_ms": summ["total"].get("p95"),
        "p99_frame_step_ms": summ["total"].get("p99"),
        "stage_p50_ms": {k: v.get("p50") for k, v in summ.items()},
        "frames": nchunks,
        "lm_step_ms": lm_ms,
        "lm_ctx_tokens": n0,
        "lm_hbm_gbs": (wbytes + kv_bytes) / (lm_ms * 1e-3) / 1e9,
        "lm_weight_gb_per_step": wbytes / 1e9,
        "model_load_s": load_s,
        "audio_tokens_ok": bool(out_ids_ok),
    }
