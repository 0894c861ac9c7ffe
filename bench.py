#!/usr/bin/env python3
"""bench.py -- headline benchmark of the duplex codec-LM hot path on MI355X.

Contract (driver): python bench.py --gpus N --steps K --warmup W  -> ONE JSON line on rank 0.

Workload at N=1 = BASELINE.json configs[1]: batch MagiCodec-style encode of synthetic 16 kHz
stereo, 0.1 s chunks with 2.0 s left context, 256 windows per pass (encode_audio_gpu_1.sh:2-8).
One STEP = one pass = 256 windows (128 chunk positions x 2 channels = 12.8 s of new stereo audio);
the default K=282 steps is ~1 h of stereo.  Inputs are resident in HBM before the timed region.
`value` = hours of (stereo) audio encoded per wall-clock hour, summed over all ranks (the path
shards by chunk range with no collective: weak scaling, "replicas only" for RCCL).

The same line carries the duplex-stream leg (BASELINE configs[2]/[3]: ~1B random-init
codec-LM, 80 ms frames) as `duplex`: xRT and p50 frame-step latency, plus `roofline` for the
dominant kernel of the headline workload and `cpu_baseline` (the CPU oracle timed on the host
cores on a bounded sample, rank 0, N=1 only).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

F32_MFMA_PEAK_TFLOPS = 157.3  # MI355X_MICROARCH.md: peak FP32 (matrix), dense
HBM_PEAK_GBS = 8000.0         # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def synth_audio(n: int, seed: int, device) -> torch.Tensor:
    """SURVEY.md 8d generator (3 sines @ 0.1 + N(0, 0.01), clipped), built on the device."""
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    t = torch.arange(n, device=device, dtype=torch.float64) / 16000.0
    x = torch.zeros(n, device=device, dtype=torch.float64)
    for f in (220.0, 440.0, 1330.0):
        x += 0.1 * torch.sin(2 * np.pi * f * t)
    x = x.float() + 0.01 * torch.randn(n, device=device, generator=g)
    return x.clamp_(-1.0, 1.0)


def cpu_baseline_encode(cfg, weights, chunk, ctx, n_chunks, channels):
    """The CPU oracle (kind 'port': the reference's codec arithmetic is a third-party package that
    is absent offline) on a bounded sample of the same workload, all host cores."""
    from oracle.codec import OracleCodec
    oc = OracleCodec(cfg, weights)
    rng = np.random.default_rng(0)
    n = ctx + n_chunks * chunk
    audio = np.clip(rng.normal(0, 0.1, (channels, n)), -1, 1).astype(np.float32)
    W = max(chunk, ctx)
    wins = np.stack([audio[c, (i + 1) * chunk + ctx - W:(i + 1) * chunk + ctx] for i in range(n_chunks) for c in range(channels)])
    oc.encode(wins[:2])  # warm
    t0 = time.perf_counter()
    oc.encode(wins)
    dt = time.perf_counter() - t0
    return dict(value=(n_chunks * chunk / 16000.0) / dt, unit="audio-hours/hour", cores=os.cpu_count(), kind="port",
                sample=f"{len(wins)} windows of {W} samples ({n_chunks} chunk positions x {channels} ch) in {dt:.2f} s, OpenMP x{os.cpu_count()}")


def cpu_baseline_lm_step(cfg, ctx_tokens=64, steps=3):
    """The LM oracle (oracle/lm_ref.py::LMRef, fp32 torch on the host cores, kind 'port': the reference's realtime LM is
    llama.cpp, absent offline) at the SAME dims and hash-generated weights as the GPU model: S=2 decode steps on top of a
    short context.  Bounded: weight generation ~10 s, a step streams 6 GB of fp32 weights."""
    import torch as _t
    from oracle import lm_ref
    threads = min(os.cpu_count(), 32)     # M=2 GEMVs: more threads than memory channels only add contention (256 threads: 42 s per step)
    _t.set_num_threads(threads)
    rng = np.random.default_rng(7)
    ids = rng.integers(128266, 128266 + 131072, ctx_tokens + 2 * (steps + 1))
    t0 = time.perf_counter()
    ref = lm_ref.LMRef(cfg, lm_ref.random_weights(cfg, 0, 0.02, embed_rows=ids), kv_dtype=_t.float16)
    gen_s = time.perf_counter() - t0
    ref.eval(ids[:ctx_tokens])
    ref.eval(ids[ctx_tokens:ctx_tokens + 2])    # warm
    t0 = time.perf_counter()
    for i in range(1, steps + 1):
        ref.eval(ids[ctx_tokens + 2 * i:ctx_tokens + 2 * i + 2])
    ms = (time.perf_counter() - t0) * 1e3 / steps
    return dict(value=ms, unit="ms per S=2 LM step", cores=threads, kind="port",
                sample=f"{steps} S=2 steps of LMRef (fp32 torch, {threads} threads) at Llama-3.2-1B dims, V={cfg.vocab_size}, "
                       f"context {ctx_tokens}+ tokens; weights regenerated from the device hash in {gen_s:.1f} s")


def batch_cli_leg(hours=2.0, seed=0):
    """The SAME encode through the batch CLI (`audio_to_codes`, the drop-in for codec_bpe.audio_to_codes in encode_audio_gpu_*.sh) on
    a synthetic corpus of 10-60 s stereo .wav utterances written to a temporary directory: file reading, cross-file window batching,
    H2D, encode, D2H and .npy writing included -- the deployment-level number next to the kernel-level `value`.  2 h of audio (~2.3 s
    of encoding: several super-batches, so that start-up and steady state can be told apart): `value` is end to end;
    `gpu_busy_fraction` = encoder time on the GPU / wall time of the pipeline; `steady_state` leaves out the time before the first
    super-batch reached the GPU."""
    import shutil
    import tempfile
    import wave
    from realtime_codec_agent_amd import audio_to_codes
    rng = np.random.default_rng(seed)
    root = tempfile.mkdtemp(prefix="rca_bench_corpus_")
    try:
        raw = os.path.join(root, "raw")
        total, i = 0.0, 0
        while total < hours * 3600:
            secs = float(rng.uniform(10, 60))
            n = int(secs * 16000)
            t = np.arange(n, dtype=np.float32) / np.float32(16000.0)
            noise = rng.standard_normal((2, n), dtype=np.float32) * np.float32(0.02)
            sig = np.stack([np.float32(0.1) * np.sin(np.float32(2 * np.pi * f) * t) for f in (220.0 + i, 330.0 + i)]) + noise
            d = os.path.join(raw, f"spk{i % 7:02d}")
            os.makedirs(d, exist_ok=True)
            with wave.open(os.path.join(d, f"utt{i:04d}.wav"), "wb") as w:
                w.setnchannels(2); w.setsampwidth(2); w.setframerate(16000)
                w.writeframes((np.clip(sig.T, -1, 1) * 32767).astype("<i2").tobytes())
            total += secs
            i += 1
        # the CLI prints its own JSON summary: keep it off stdout, which carries exactly ONE line (the bench contract)
        import contextlib
        with contextlib.redirect_stdout(sys.stderr):
            s = audio_to_codes.main(["--audio_path", raw, "--codes_path", os.path.join(root, "codes"), "--stereo"])
        st = s.get("stages") or {}
        out = dict(value=s["audio_hours_per_hour"], unit="audio-hours/hour", files=i, audio_hours=total / 3600.0, elapsed_s=s["elapsed_s"],
                   note="audio_to_codes CLI end to end (read .wav, batch windows across files, encode, write .npy), one process")
        if st.get("total_s"):
            gpu_s = st.get("encode_many_gpu_ms", 0.0) * 1e-3
            out["gpu_busy_fraction"] = gpu_s / st["total_s"]
            out["startup_s"] = st.get("first_batch_ready_s")
            out["steady_state"] = (total / 3600.0) / ((st["total_s"] - (st.get("first_batch_ready_s") or 0.0)) / 3600.0) if st["total_s"] > 0 else None
            out["gpu_level"] = (total / 3600.0) / (gpu_s / 3600.0) if gpu_s > 0 else None
            out["stages"] = {k: (round(v, 4) if isinstance(v, float) else v) for k, v in st.items()}
        return out
    finally:
        shutil.rmtree(root, ignore_errors=True)


def latest_profile_traffic():
    """HBM bytes per launch of the dominant kernel from the newest committed PMC passes (profiles/rNN/traffic.json, written by
    scripts/collect_profiles.py from separate rocprofv3 --pmc runs: FETCH_SIZE doubled per the gfx950 note + WRITE_SIZE).  None when
    no such file is committed -- the number is a property of the profiled build, never of this run."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*", "traffic.json")))
    if not files:
        return None, None
    try:
        with open(files[-1]) as f:
            d = json.load(f)
        return float(d["conv1d_mfma_kernel"]["hbm_bytes_per_launch"]), os.path.relpath(files[-1], ROOT)
    except Exception:
        return None, None


def launch_ranks(args) -> int:
    """`python bench.py --gpus N` with N > 1 outside a launcher: start the N ranks as CHILD processes (one per GPU, the reference's
    own form of parallelism: encode_audio_gpu_{1..4}.sh, realtime_agent_v2.py:832-836) before anything here has touched the GPU,
    relay rank 0's JSON line and the exit code.  Never an exec of a process that holds the GPU."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, text=True)
    line = next((ln for ln in reversed(r.stdout.splitlines()) if ln.startswith("{") and '"metric"' in ln), None)
    if line is not None:
        print(line)
    else:
        sys.stdout.write(r.stdout)
    return r.returncode if line is not None or r.returncode else 1


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=282)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch-windows", type=int, default=256)
    ap.add_argument("--chunk-secs", type=float, default=0.1)
    ap.add_argument("--context-secs", type=float, default=2.0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-duplex", action="store_true")
    ap.add_argument("--no-cli-leg", action="store_true", help="skip the batch-CLI leg (audio_to_codes on a synthetic corpus of short files)")
    ap.add_argument("--no-trim-leg", action="store_true", help="skip the receptive-field-trimmed batch leg (profiling runs: keeps per-kernel averages to the headline path)")
    ap.add_argument("--no-bf16-leg", action="store_true", help="skip the opt-in bf16-MFMA encoder legs (rca_codec_set_mfma_mode)")
    ap.add_argument("--duplex-secs", type=float, default=125.0,
                    help="audio seconds of the duplex leg; >= 110 puts two sliding-window trims (80 s context, trim by 20 s) inside the timed window")
    ap.add_argument("--variant", type=int, default=1)
    args = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(launch_ranks(args))          # nothing above this line has touched the GPU
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        sys.exit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: start one rank per GPU (python bench.py --gpus N does it itself)")
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # rehearsal hooks (not used by the driver): several ranks on ONE card exercise the N > 1 code path on a 1-GPU box --
    # RCA_BENCH_DEVICE pins every rank to that device, RCA_BENCH_BACKEND=gloo skips the RCCL attempt
    if os.environ.get("RCA_BENCH_DEVICE") is not None:
        local_rank = int(os.environ["RCA_BENCH_DEVICE"])
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    # The data path has no collective: the control plane (barrier, max of one float, gather of the duplex summaries) rides on RCCL
    # when every rank can bring it up and on gloo over CPU tensors otherwise -- same process, no re-exec; the line says which.
    from realtime_codec_agent_amd.dist_utils import ControlPlane
    cp = ControlPlane(prefer=os.environ.get("RCA_BENCH_BACKEND", "nccl"), device_index=local_rank)

    from realtime_codec_agent_amd import _native
    if _native.needs_build() and rank == 0:
        _native.build()
    cp.barrier()
    from realtime_codec_agent_amd.codec import HipCodec
    from realtime_codec_agent_amd.codec_model import CodecConfig, init_codec_weights

    cfg = CodecConfig()
    weights = init_codec_weights(cfg, seed=0)
    hip = HipCodec(cfg, weights, device=local_rank)
    hip.set_variant(args.variant)

    C = 2
    chunk = int(args.chunk_secs * cfg.sample_rate)
    ctx = int(args.context_secs * cfg.sample_rate)
    chunks_per_step = args.batch_windows // C
    fpc = hip.frames_per_chunk(chunk)
    first = (ctx + chunk - 1) // chunk  # first chunk whose window is full (no warm-up passes in the timed region)
    total_steps = args.warmup + args.steps
    n_chunks = first + total_steps * chunks_per_step
    N = n_chunks * chunk
    audio = torch.stack([synth_audio(N, 1000 * rank + 1 + c, dev) for c in range(C)]).contiguous()
    codes = torch.empty((C, chunks_per_step * fpc * total_steps), dtype=torch.int64, device=dev)
    stream = torch.cuda.current_stream(dev).cuda_stream

    def step(i):
        c0 = first + i * chunks_per_step
        hip.encode_chunk_range_dev(audio.data_ptr(), C, N, chunk, ctx, args.batch_windows, c0, c0 + chunks_per_step,
                                   codes.data_ptr() + 8 * i * chunks_per_step * fpc, codes.shape[1], stream)

    for i in range(args.warmup):
        step(i)
    torch.cuda.synchronize(dev)
    cp.barrier()
    torch.cuda.synchronize(dev)
    hip.profile(True)
    t0 = time.perf_counter()
    for i in range(args.warmup, total_steps):
        step(i)
    torch.cuda.synchronize(dev)
    cp.barrier()
    torch.cuda.synchronize(dev)
    elapsed = time.perf_counter() - t0
    hip.profile(False)
    prof = {k: hip.profile_read(k) for k in (0, 1, 2, 3)}       # the headline's launches (later legs profile on their own)
    elapsed = cp.max(elapsed)

    # SURVEY 8f-1 leg (reported beside the headline, never as `value`): the same steps with every window cut down
    # to the receptive field of its kept frames -- identical codes (checked), ~10x less encoder work
    elapsed_trim, trim_identical = None, None
    if not args.no_trim_leg:
        codes_full = codes.clone()
        hip.set_window_trim(True)
        for i in range(args.warmup):
            step(i)
        torch.cuda.synchronize(dev)
        cp.barrier()
        t0 = time.perf_counter()
        for i in range(args.warmup, total_steps):
            step(i)
        torch.cuda.synchronize(dev)
        cp.barrier()
        elapsed_trim = time.perf_counter() - t0
        hip.set_window_trim(False)
        elapsed_trim = cp.max(elapsed_trim)
        trim_identical = bool(torch.equal(codes, codes_full))
        assert trim_identical, "window-trimmed batch encode produced different codes"

    # Opt-in arithmetic legs (reported beside the headline, never as `value`): the same steps with the encoder's conv layers on the
    # bf16 matrix instruction -- operands split into bf16 hi + lo (mode 3) or rounded to bf16 like the reference's bf16 autocast
    # (mode 1, audio_tokenizer.py:24,78-82).  Not bit-exact: the fraction of code ids equal to the f32 path is measured here.
    bf16_legs, roofline_bf16 = None, None
    conv_bytes_f32 = prof[0]["bytes"]
    if not args.no_bf16_leg:
        bf16_legs = {}
        codes_f32 = codes.clone()
        for mode, name in ((3, "bf16_hi_lo_split"), (1, "bf16_rounded")):
            hip.set_mfma_mode(mode)
            for i in range(args.warmup):
                step(i)
            torch.cuda.synchronize(dev)
            cp.barrier()
            hip.profile(True)
            t0 = time.perf_counter()
            for i in range(args.warmup, total_steps):
                step(i)
            torch.cuda.synchronize(dev)
            cp.barrier()
            el = time.perf_counter() - t0
            hip.profile(False)
            el = cp.max(el)
            eq = float((codes == codes_f32).double().mean().item())
            bf16_legs[name] = {"value": world * args.steps * chunks_per_step * chunk / cfg.sample_rate / el, "unit": "audio-hours/hour",
                               "ms_per_step": 1e3 * el / args.steps, "code_ids_equal_to_f32_path": eq, "mfma_mode": mode}
            leg_prof = {k: hip.profile_read(k) for k in (0, 1, 2, 3)}       # (reading empties the records of this leg)
            if mode == 1:
                # the reference's own arithmetic class (bf16 autocast, audio_tokenizer.py:24,78-82) is HBM-bound: conv launches of the
                # blocked bf16 pipeline (conv_in_blk + conv_bf16_blk), algorithmic bytes = every activation once in and once out as
                # bf16 (PCM f32 in, the last layer f32 out) + the weights, over their HIP-event time
                pc, pi = leg_prof[0], leg_prof[2]
                ms, by = pc["ms"] + pi["ms"], pc["bytes"] + pi["bytes"]
                if ms > 0:
                    roofline_bf16 = {"kernel": "conv_in_blk_kernel + conv_bf16_blk_kernel (rca_codec_set_mfma_mode(1): bf16 MFMA, channel-blocked bf16 activations)",
                                     "bound": "hbm", "achieved": by / (ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                     "frac": by / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "algorithmic_bytes_per_step": by / args.steps,
                                     "conv_ms_per_step": ms / args.steps, "launches": pc["launches"] + pi["launches"],
                                     "f32_path_bytes_per_step": conv_bytes_f32 / args.steps,
                                     "frac_if_counted_in_f32_bytes": conv_bytes_f32 / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                     "traffic": None, "opt_in": True}
        hip.set_mfma_mode(0)
        codes.copy_(codes_f32)

    audio_secs = args.steps * chunks_per_step * chunk / cfg.sample_rate  # per rank, stereo seconds
    value = world * audio_secs / elapsed
    conv = prof[0]
    achieved = conv["flops"] / (conv["ms"] * 1e-3) / 1e12 if conv["ms"] > 0 else 0.0
    # sanity: emitted ids are in range and not constant
    sample = codes[:, : 64 * fpc].cpu().numpy()
    assert sample.min() >= 0 and sample.max() < cfg.codebook_size

    traffic, traffic_src = latest_profile_traffic()
    out = {
        "metric": "batch-encode audio-hours/hour (+ xRT and p50 frame-step latency of 1 duplex stream in `duplex`)",
        "value": value,
        "unit": "audio-hours/hour",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": 1e3 * elapsed / args.steps,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f32",
        "data": "synthetic",
        "config": {
            "workload": "batch MagiCodec-style encode, 16 kHz stereo, 0.1 s chunks / 2.0 s context, 256 windows per step (BASELINE configs[1])",
            "windows_per_step": args.batch_windows,
            # the codebook search runs on the frames a window KEEPS (its chunk's own codes), not on all 100 frames of the window: the
            # quantiser is per-frame, so the discarded 95 never influence an output (SURVEY 8d counts 25 600 rows per step)
            "vq_rows_per_step": args.batch_windows * fpc,
            "vq_rows_per_step_if_every_window_frame_were_searched": args.batch_windows * (max(chunk, ctx) // cfg.hop),
            "audio_hours_timed_per_gpu": audio_secs / 3600.0,
            "channel_hours_per_hour": value * C,
            "codebook": f"{cfg.codebook_size}x{cfg.codebook_dim}",
            "encoder_gflop_per_window": cfg.encoder_flops_per_sample() * ctx / 1e9,
            "sharding": "chunk ranges per rank, no collective (replicas only)",
            "control_plane": cp.describe(),
            "bf16_mfma_opt_in": bf16_legs,
            "receptive_field_trimmed": None if args.no_trim_leg else {
                "value": world * audio_secs / elapsed_trim, "unit": "audio-hours/hour", "ms_per_step": 1e3 * elapsed_trim / args.steps,
                "codes_identical_to_full_windows": trim_identical,
                "note": "same steps with rca_codec_set_window_trim(1): each window cut to the kept frames + their receptive "
                        "field (SURVEY 8f-1); valid for this build's conv codec only, so it is not the headline value",
            },
        },
        "roofline": {
            "kernel": "conv1d_mfma_kernel (implicit-GEMM conv, v_mfma_f32_32x32x2_f32)",
            "bound": "mfma",
            "achieved": achieved,
            "peak": F32_MFMA_PEAK_TFLOPS,
            "unit": "TFLOP/s",
            "frac": achieved / F32_MFMA_PEAK_TFLOPS,
            # HBM bytes per launch of this kernel from the committed PMC passes (profiles/r01: FETCH_SIZE x2 per
            # the gfx950 note + WRITE_SIZE, separate rocprofv3 --pmc runs); algorithmic bytes for comparison
            "traffic": traffic,
            "traffic_source": traffic_src,
            "algorithmic_bytes_per_launch": conv["bytes"] / max(1, conv["launches"]),
            "avg_launch_ms": conv["ms"] / max(1, conv["launches"]),
            "launches": conv["launches"],
            "flop_per_launch": conv["flops"] / max(1, conv["launches"]),
            "share_of_step_time": conv["ms"] * 1e-3 / elapsed if elapsed > 0 else None,
            "other_kernels_ms": {"vq_search": prof[1]["ms"], "conv_in": prof[2]["ms"], "other": prof[3]["ms"]},
            "conv_in_hbm_gbs": prof[2]["bytes"] / (prof[2]["ms"] * 1e-3) / 1e9 if prof[2]["ms"] > 0 else None,
            # the same launches against the HBM roof (north_star asks for it): the f32 conv stack is compute-bound,
            # algorithmic bytes / launch time is ~1/9 of 8 TB/s
            "hbm": {"achieved_gbs": conv["bytes"] / (conv["ms"] * 1e-3) / 1e9 if conv["ms"] > 0 else None, "peak_gbs": HBM_PEAK_GBS,
                    "frac": conv["bytes"] / (conv["ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS if conv["ms"] > 0 else None},
        },
    }
    if roofline_bf16 is not None:
        out["roofline_bf16"] = roofline_bf16
    # The duplex legs run BEFORE the batch-CLI and CPU-baseline legs: a session that started behind them (256 OpenMP threads, 32 torch
    # threads, a 5 GB host model, the CLI's reader / writer threads and pinned ring, all just released) showed one 27-35 ms frame in
    # three of five default runs and in none of 17 runs without those legs in front of it (scripts/duplex_outlier_ab.sh).
    if not args.no_duplex:
        # one independent duplex session per GPU (BASELINE configs[3]/[4]; no exchange between sessions)
        from realtime_codec_agent_amd.duplex_bench import run_duplex_bench
        mine = run_duplex_bench(dev, secs=args.duplex_secs if world == 1 else min(args.duplex_secs, 10.0))
        if cp.active:
            allr = cp.all_gather_object({k: mine[k] for k in ("xRT", "p50_frame_step_ms", "p95_frame_step_ms", "p99_frame_step_ms", "max_frame_step_ms",
                                                              "frames_over_budget", "lm_step_ms")})
            mine = dict(mine, sessions=world, per_gpu=allr, xRT_min=min(r["xRT"] for r in allr),
                        p50_frame_step_ms_max=max(r["p50_frame_step_ms"] for r in allr))
        out["duplex"] = mine
        out["roofline_lm"] = mine.pop("roofline_lm")
        if world == 1:
            # the same stream with the decode step streaming the quantised formats the reference deploys besides F16 (q8_0 and the
            # Q4_K blocks of its Q4_K_M file, prep_test_model.sh:29,31): reported beside the bf16 leg, never instead of it
            for fmt, secs in (("q8_0", 30.0), ("q4_k", 20.0)):
                q = run_duplex_bench(dev, secs=secs, weight_format=fmt)
                rq = q.pop("roofline_lm")
                out[f"duplex_{fmt}"] = {k: q[k] for k in ("workload", "xRT", "p50_frame_step_ms", "p99_frame_step_ms", "max_frame_step_ms", "frames", "lm_step_ms",
                                                           "lm_ctx_tokens", "lm_hbm_gbs", "lm_weight_gb_per_step")}
                out[f"duplex_{fmt}"]["roofline_lm"] = rq
    if rank == 0 and world == 1 and not args.no_cli_leg:
        out["config"]["batch_cli"] = batch_cli_leg()
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline_encode(cfg, weights, chunk, ctx, 64, C)
        if not args.no_duplex:
            from realtime_codec_agent_amd.llm import LMConfig
            out["cpu_baseline"]["lm_step"] = cpu_baseline_lm_step(LMConfig.llama_3_2_1b())
    if rank == 0:
        print(json.dumps(out))
    cp.close()


if __name__ == "__main__":
    main()
