"""The multi-process paths executed for real on ONE MI355X (no mock): every rank / worker is its own OS process.
  1. bench.py --gpus 2 through its own launcher (parent touches no GPU, children via torch.distributed.run), ranks pinned to card 0,
     gloo for the barrier / all_reduce / all_gather_object of the duplex leg
  2. audio_to_codes with 2 ranks (torch.distributed.run) on card 0: sharded by duration, tree byte-identical to the 1-rank run
  3. two RealtimeAgentMultiprocessing(gpu_id=0) duplex sessions side by side (1B random-init LM + full codec each)
Reference forms: encode_audio_gpu_{1..4}.sh, inference_client_self_play.py:148-159, realtime_agent_v2.py:832-836.
usage (GPU box): python scripts/multirank_rehearsal.py > gpurun_out/multirank.txt"""
import json, os, shutil, subprocess, sys, tempfile, time, wave
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np



def tree(p):
    return {os.path.relpath(os.path.join(r, f), p): open(os.path.join(r, f), "rb").read() for r, _, fs in os.walk(p) for f in fs}


def main():
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    out = {}

    # ---- 1. bench.py --gpus 2 (self-launch)
    # RCCL is ASKED for (the driver's 8-GPU run does too); two ranks on one card make it refuse, so this also runs the control plane's
    # fall-back to gloo inside the rank processes for real (dist_utils.ControlPlane; RCA_REHEARSE_BACKEND=gloo skips the attempt)
    e1 = dict(env, RCA_BENCH_DEVICE="0", RCA_BENCH_BACKEND=os.environ.get("RCA_REHEARSE_BACKEND", "nccl"))
    t0 = time.perf_counter()
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "5", "--warmup", "2", "--duplex-secs", "10"], env=e1,
                       capture_output=True, text=True, timeout=900)
    line = next((ln for ln in reversed(r.stdout.splitlines()) if ln.startswith("{")), None)
    assert r.returncode == 0 and line, (r.returncode, r.stdout[-2000:], r.stderr[-3000:])
    b = json.loads(line)
    assert b["n_gpus"] == 2 and b["duplex"]["sessions"] == 2 and len(b["duplex"]["per_gpu"]) == 2
    cp = b["config"]["control_plane"]
    assert cp["backend"] in ("gloo", "nccl") and (cp["backend"] == "nccl" or e1["RCA_BENCH_BACKEND"] == "gloo" or cp["fallback_reason"]), cp
    out["bench_gpus2"] = dict(wall_s=round(time.perf_counter() - t0, 1), n_gpus=b["n_gpus"], value=b["value"], ms_per_step=b["ms_per_step"], scaling=b["scaling"],
                              control_plane=cp,
                              duplex_sessions=b["duplex"]["sessions"], duplex_per_rank=b["duplex"]["per_gpu"])
    print("1. bench.py --gpus 2:", json.dumps(out["bench_gpus2"]), flush=True)

    # ---- 2. audio_to_codes, 1 rank vs 2 ranks
    root = tempfile.mkdtemp(prefix="rca_mr_")
    raw = os.path.join(root, "raw")
    rng = np.random.default_rng(0)
    for i in range(24):
        n = int(rng.uniform(8, 40) * 16000)
        t = np.arange(n) / 16000.0
        sig = np.stack([0.1 * np.sin(2 * np.pi * f * t) + rng.normal(0, 0.02, n) for f in (220.0 + i, 330.0 + i)])
        d = os.path.join(raw, f"spk{i % 5}")
        os.makedirs(d, exist_ok=True)
        with wave.open(os.path.join(d, f"utt{i:03d}.wav"), "wb") as w:
            w.setnchannels(2); w.setsampwidth(2); w.setframerate(16000)
            w.writeframes((np.clip(sig.T, -1, 1) * 32767).astype("<i2").tobytes())
    res = {}
    for world in (1, 2):
        codes = os.path.join(root, f"codes{world}")
        argv = ["-m", "realtime_codec_agent_amd.audio_to_codes", "--audio_path", raw, "--codes_path", codes, "--stereo"]
        if world == 1:
            cmd = [sys.executable] + argv
        else:
            cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1", "--master-port", "29611"] + argv
        r = subprocess.run(cmd, env=dict(env, RCA_DEVICE="0", RCA_DIST_BACKEND=os.environ.get("RCA_REHEARSE_BACKEND", "nccl")), capture_output=True, text=True, timeout=900, cwd=ROOT)
        line = next((ln for ln in reversed(r.stdout.splitlines()) if ln.startswith("{")), None)
        assert r.returncode == 0 and line, (world, r.returncode, r.stdout[-2000:], r.stderr[-3000:])
        res[world] = json.loads(line)

    t1, t2 = tree(os.path.join(root, "codes1")), tree(os.path.join(root, "codes2"))
    assert t1.keys() == t2.keys() and all(t1[k] == t2[k] for k in t1) and res[1]["codes"] == res[2]["codes"] and res[2]["world_size"] == 2
    out["audio_to_codes"] = dict(files=res[1]["files"], trees_identical=True, codes=res[1]["codes"], control_plane_2_ranks=res[2].get("control_plane"),
                                 audio_hours_per_hour={"1 rank": res[1]["audio_hours_per_hour"], "2 ranks on one card": res[2]["audio_hours_per_hour"]})
    print("2. audio_to_codes 1 vs 2 ranks:", json.dumps(out["audio_to_codes"]), flush=True)
    shutil.rmtree(root, ignore_errors=True)

    # ---- 3. two duplex sessions in two worker processes on card 0
    from realtime_codec_agent_amd.realtime_agent_config import RealtimeAgentConfig
    from realtime_codec_agent_amd.realtime_agent_mp import RealtimeAgentMultiprocessing
    from realtime_codec_agent_amd.duplex_bench import session_resources_kwargs
    cfg = RealtimeAgentConfig(chunk_size_secs=0.08, use_whisper=False, top_k=100, temperature=1.0, seed=42,
                              force_trans_after_inactivity_secs=0.0, force_response_after_inactivity_secs=0.0)   # the bench session's settings
    t0 = time.perf_counter()
    sessions = [RealtimeAgentMultiprocessing(wait_until_running=False, config=cfg, gpu_id=0, **session_resources_kwargs(seed=s)) for s in (0, 1)]
    for s in sessions:
        s.wait_until_running()
    load_s = time.perf_counter() - t0
    n = 250                                           # 20 s of audio per session
    t = np.arange(n * 1280) / 16000.0
    audio = (0.1 * np.sin(2 * np.pi * 220 * t) + 0.1 * np.sin(2 * np.pi * 440 * t) + np.random.default_rng(0).normal(0, 0.01, t.size)).astype(np.float32)
    t0 = time.perf_counter()
    rts = [[], []]
    for i in range(n):
        for s in sessions:
            s.queue_input(audio[i * 1280:(i + 1) * 1280])
        for k, s in enumerate(sessions):
            chunk, rt = s.next_output(block=True)
            assert chunk.shape == (1280,)
            if rt is not None:
                rts[k].append(rt)
    wall = time.perf_counter() - t0
    pids = [s.worker_pid for s in sessions]
    for s in sessions:
        s.close()
    assert pids[0] != pids[1]
    out["two_sessions_one_card"] = dict(load_s=round(load_s, 1), frames_per_session=n, audio_s=n * 0.08, wall_s=round(wall, 2), both_real_time=bool(wall < n * 0.08),
                                        xRT_per_session=[float(np.median(r)) if r else None for r in rts], worker_pids=pids)
    print("3. two RealtimeAgentMultiprocessing(gpu_id=0) sessions:", json.dumps(out["two_sessions_one_card"]), flush=True)
    print(json.dumps(out))


if __name__ == "__main__":      # the session workers are spawned processes: they import this file and must not run it again
    main()
