"""Duplex-stream leg of bench.py (BASELINE configs[2]/[3]): RealtimeAgent over the HIP AudioTokenizer and a
~1B random-init codec LM, fed the SURVEY.md 8d signal in 80 ms frames -- the cli_benchmark.py loop
(cli_benchmark.py:67-71) with device-synchronised timestamps.  Reports xRT by the reference's definition
(median of 2 s window means, realtime_agent_profiler.py:30-38,75), the process_audio latency distribution INCLUDING its
tail -- the run is long enough for the sliding-window eviction (realtime_agent_v2.py:187-190,725-733: 80 s of context,
trimmed by 20 s) to fire inside the timed window -- and the LM decode step against the HBM roof (`roofline_lm`)."""
from __future__ import annotations

import time

import numpy as np

HBM_PEAK_GBS = 8000.0   # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.3 TB/s is what a plain copy reaches)


def synth_signal(n: int, seed: int = 0) -> np.ndarray:
    rng = np.random.default_rng(seed)
    t = np.arange(n) / 16000.0
    x = sum(0.1 * np.sin(2 * np.pi * f * t) for f in (220.0, 440.0, 1330.0)) + rng.normal(0.0, 0.01, n)
    return np.clip(x, -1.0, 1.0).astype(np.float32)


def session_resources_kwargs(seed: int = 0, n_ctx: int = 16384, weight_format=None) -> dict:
    """RealtimeAgentResources keyword arguments of the bench session (picklable: RealtimeAgentMultiprocessing builds the resources in
    its worker process from them, realtime_agent_v2.py:832-836)."""
    from .llm import LMConfig
    return dict(llm_model_path="random:Llama-3.2-1B-codec", llm_n_ctx=n_ctx, llm_config=LMConfig.llama_3_2_1b(), with_aux_llm=False,
                llm_random_seed=seed, llm_weight_format=weight_format)


def run_duplex_bench(dev=None, secs: float = 125.0, chunk_size_secs: float = 0.08, n_ctx: int = 16384, lm_steps_probe: int = 256,
                     max_context_secs: float = 80.0, trim_by_secs: float = 20.0, weight_format=None, duplex_graph: bool = True) -> dict:
    import torch
    from .llm import LMConfig
    from .realtime_agent_config import RealtimeAgentConfig
    from .realtime_agent_resources import RealtimeAgentResources
    from .realtime_agent_v2 import RealtimeAgent

    import gc
    gc.collect()          # an earlier leg's handles (a 3 GB model, its KV caches) must not be torn down inside this leg's timed frames
    cfg = LMConfig.llama_3_2_1b()
    t0 = time.perf_counter()
    res = RealtimeAgentResources(**session_resources_kwargs(0, n_ctx, weight_format))
    config = RealtimeAgentConfig(chunk_size_secs=chunk_size_secs, use_whisper=False, top_k=100, temperature=1.0, seed=42,
                                 max_context_secs=max_context_secs, trim_by_secs=trim_by_secs,
                                 force_trans_after_inactivity_secs=0.0, force_response_after_inactivity_secs=0.0)
    agent = RealtimeAgent(resources=res, config=config)
    agent.use_duplex_graph = duplex_graph       # False: one replay per LM chunk between the separate codec calls (the round-2 path)
    load_s = time.perf_counter() - t0
    n = int(secs * 16000)
    sig = synth_signal(n, 0)
    cs = agent.chunk_size_samples
    # warm-up (graph capture, allocator) then restart the profilers
    for s in range(0, 10 * cs, cs):
        agent.process_audio(sig[s:s + cs])
    agent.profilers.reset()
    # What a serving process does once its model is loaded: everything allocated so far (the imported modules, the model objects --
    # about a million Python objects) moves to the permanent generation, so a full collection inside a frame scans the session's own
    # objects only.  Collections that still happen in the timed frames are recorded (generation, pause, frame) and reported: a frame
    # that is tens of milliseconds long once in a few runs is attributable instead of "not reproducible".
    gc.collect()
    gc.freeze()
    gc_log, gc_t = [], [0.0]
    def _gc_cb(phase, info):
        if phase == "start":
            gc_t[0] = time.perf_counter()
        else:
            gc_log.append((int(info.get("generation", -1)), (time.perf_counter() - gc_t[0]) * 1e3, nchunks))
    gc.callbacks.append(_gc_cb)
    one_replay0 = agent.duplex_graph_frames
    t1 = time.perf_counter()
    nchunks = 0
    trims = []                                  # (audio second, frame latency ms, context tokens after the trim)
    last_trim = agent.trim_to_secs
    lat = agent.profilers.total_profiler.latencies_secs
    try:
        for s in range(10 * cs, n - cs + 1, cs):
            agent.process_audio(sig[s:s + cs])
            nchunks += 1
            if agent.trim_to_secs != last_trim:
                last_trim = agent.trim_to_secs
                trims.append(dict(at_audio_secs=round(s / 16000.0, 2), frame_ms=lat[-1] * 1e3 if lat else None, context_tokens_after=res.llm.n_tokens))
        torch.cuda.synchronize()
    finally:
        gc.callbacks.remove(_gc_cb)
        gc.unfreeze()
    wall = time.perf_counter() - t1
    summ = agent.profilers.summary()
    lat_ms = np.asarray(lat) * 1e3
    budget_ms = chunk_size_secs * 1e3
    # raw LM step rate at the current context length (S=2 per step, graph replay)
    llm = res.llm
    n0 = llm.n_tokens
    toks = [agent.input_ids[-2], agent.input_ids[-1]]
    for _ in range(8):                      # untimed: the step graph of this bucket last ran before the session's final frames
        llm.n_tokens = n0 - 2
        llm.step(toks)
    llm.sync()
    t2 = time.perf_counter()
    for _ in range(lm_steps_probe):
        llm.n_tokens = n0 - 2
        llm.step(toks)
    llm.sync()
    lm_ms = (time.perf_counter() - t2) * 1e3 / lm_steps_probe
    wbytes = llm.weight_bytes_per_step()
    kv_bytes = 2 * 2 * cfg.n_layers * cfg.n_kv_heads * cfg.head_dim * n0
    gbs = (wbytes + kv_bytes) / (lm_ms * 1e-3) / 1e9
    out_ids_ok = all(t > agent.end_header_token_id for t in agent.input_ids[-8:])
    return {
        "workload": f"1 duplex stream, Llama-3.2-1B dims (V={cfg.vocab_size}) random-init {llm.weight_format}, {int(chunk_size_secs * 1000)} ms frames, "
                    f"top_k=100 T=1.0 seed=42, {secs:.0f} s of audio, context {max_context_secs:.0f} s trimmed by {trim_by_secs:.0f} s",
        "xRT": summ["total"]["xrt_median"],
        "xRT_wall": nchunks * chunk_size_secs / wall,
        "p50_frame_step_ms": summ["total"].get("p50"),
        "p95_frame_step_ms": summ["total"].get("p95"),
        "p99_frame_step_ms": summ["total"].get("p99"),
        "max_frame_step_ms": float(lat_ms.max()) if lat_ms.size else None,
        "frame_budget_ms": budget_ms,
        "frames_over_budget": int((lat_ms > budget_ms).sum()),
        "slowest_frames": [dict(frame=int(i), at_audio_secs=round((10 + int(i)) * chunk_size_secs, 2), ms=float(lat_ms[i])) for i in np.argsort(-lat_ms)[:5]] if lat_ms.size else [],
        "trims_in_timed_window": trims,
        "gc": {"frozen_after_warmup": True, "collections_in_timed_frames": len(gc_log),
               "max_pause_ms": max([g[1] for g in gc_log], default=0.0),
               "pauses_over_1ms": [dict(generation=g[0], ms=round(g[1], 2), frame=g[2]) for g in gc_log if g[1] > 1.0][:8]},
        "kv_shadow": bool(getattr(agent, "kv_shadow_active", False)),
        "frame_graph": bool(getattr(agent, "frame_graph_active", False)),
        "one_replay_frames": agent.duplex_graph_frames - one_replay0,   # frames that ran as ONE graph replay (rca_duplex_frame)
        "stage_p50_ms": {k: v.get("p50") for k, v in summ.items()},
        "frames": nchunks,
        "lm_step_ms": lm_ms,
        "lm_ctx_tokens": n0,
        "lm_hbm_gbs": gbs,
        "lm_weight_gb_per_step": wbytes / 1e9,
        "model_load_s": load_s,
        "audio_tokens_ok": bool(out_ids_ok),
        "roofline_lm": {
            "kernel": "one S=2 decode step (16 layers of GEMV + attention, head GEMV, sampler) replayed as one hipGraph",
            "bound": "hbm", "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS,
            "bytes_per_step": wbytes + kv_bytes, "weight_bytes_per_step": wbytes, "kv_bytes_per_step": kv_bytes, "ms_per_step": lm_ms,
            "context_tokens": n0, "floor_ms_at_peak": (wbytes + kv_bytes) / (HBM_PEAK_GBS * 1e9) * 1e3,
            "weight_format": llm.weight_format,
            "note": "algorithmic bytes = every weight once in the format the step streams (bf16: 2 B, q8_0: 34 B per 32) + the fp16 KV of the live context; timed as wall time over "
                    f"{lm_steps_probe} replays incl. the host round trip of the sampled token",
        },
    }
