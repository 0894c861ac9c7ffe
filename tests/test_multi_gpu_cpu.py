"""The N>1 path on CPU: two gloo ranks run the sharded batch-encode CLI (no data-path collective) with a
CPU encoder injected; outputs must equal the single-process run.  Also the ABI surface check."""
import ctypes
import json
import os
import re
import subprocess
import sys
import wave

import numpy as np
import pytest

from conftest import ROOT, bench_signal, rich_signal
from realtime_codec_agent_amd.dist_utils import shard_by_duration, shard_chunk_range

WORKER = r'''
import os, sys, json
sys.path.insert(0, {root!r}); sys.path.insert(0, os.path.join({root!r}, "tests"))
import numpy as np
from oracle.codec import OracleCodec
from realtime_codec_agent_amd.codec_model import init_codec_weights, tiny_codec_config
from realtime_codec_agent_amd import audio_to_codes

class CpuEncoder:  # test-only stand-in for HipWindowEncoder
    def __init__(self):
        self.cfg = tiny_codec_config()
        self.oc = OracleCodec(self.cfg, init_codec_weights(self.cfg, seed=0))
    def encode(self, audio, chunk, ctx, batch):
        return self.oc.encode_windows(audio, chunk, ctx)

out = audio_to_codes.main({argv!r}, encoder=CpuEncoder(), backend="gloo")
'''


def _make_corpus(tmp):
    raw = os.path.join(tmp, "raw")
    os.makedirs(os.path.join(raw, "CallHome", "a"))
    os.makedirs(os.path.join(raw, "Other"))
    lens = {"CallHome/a/x1.wav": 9600, "CallHome/a/x2.wav": 25600, "CallHome/x3.npy": 16000, "Other/y.wav": 8000, "CallHome/x4.wav": 12800}
    for i, (rel, n) in enumerate(lens.items()):
        sig = np.stack([bench_signal(n, i), rich_signal(n, i + 10)])
        p = os.path.join(raw, rel)
        if rel.endswith(".npy"):
            np.save(p, sig)
        else:
            with wave.open(p, "wb") as w:
                w.setnchannels(2); w.setsampwidth(2); w.setframerate(16000)
                w.writeframes((np.clip(sig.T, -1, 1) * 32767).astype("<i2").tobytes())
    return raw


def _run(tmp, raw, world, tag):
    codes = os.path.join(tmp, f"codes_{tag}")
    argv = ["--audio_path", raw, "--codes_path", codes, "--stereo", "--audio_filter", "CallHome"]
    script = os.path.join(tmp, f"worker_{tag}.py")
    with open(script, "w") as f:
        f.write(WORKER.format(root=ROOT, argv=argv))
    if world == 1:
        env = dict(os.environ, RANK="0", WORLD_SIZE="1")
        out = subprocess.run([sys.executable, script], env=env, capture_output=True, text=True, timeout=300)
    else:
        out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
                              "--master-port", "29541", script], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    summary = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    return codes, summary


def test_sharded_batch_encode_two_gloo_ranks(tmp_path):
    tmp = str(tmp_path)
    raw = _make_corpus(tmp)
    c1, s1 = _run(tmp, raw, 1, "w1")
    c2, s2 = _run(tmp, raw, 2, "w2")
    assert s1["files"] == s2["files"] == 4 and s2["world_size"] == 2
    assert s1["codes"] == s2["codes"] == 2 * 5 * (6 + 16 + 10 + 8) and abs(s1["audio_hours"] - s2["audio_hours"]) < 1e-12
    files1 = sorted(os.path.relpath(os.path.join(r, f), c1) for r, _, fs in os.walk(c1) for f in fs)
    files2 = sorted(os.path.relpath(os.path.join(r, f), c2) for r, _, fs in os.walk(c2) for f in fs)
    assert files1 == files2 and len(files1) == 1 + 2 * 4
    for f in files1:
        if f.endswith(".npy"):
            a, b = np.load(os.path.join(c1, f)), np.load(os.path.join(c2, f))
            assert a.shape == b.shape and a.ndim == 2 and a.shape[0] == 1 and a.dtype == np.int64 and np.array_equal(a, b)
            assert re.match(r"(.+)_c(\d+)[_.]", os.path.basename(f))  # lm_dataset_builder.py:79
    info = json.load(open(os.path.join(c1, "MagiCodec-50Hz-Base", "0.1s_2.0s", "stereo", "codec_info.json")))
    assert info == {"num_codebooks": 1, "codebook_size": 1024, "framerate": 50.0}
    assert os.path.isdir(os.path.join(c1, "MagiCodec-50Hz-Base", "0.1s_2.0s", "stereo", "CallHome", "a"))


CP_WORKER = """
import json, os, sys
sys.path.insert(0, {root!r})
from realtime_codec_agent_amd.dist_utils import ControlPlane
cp = ControlPlane(prefer="nccl")          # no GPU in this container: RCCL cannot come up on any rank
rank = cp.rank
cp.barrier()
out = dict(backend=cp.backend, reason=cp.fallback_reason, mx=cp.max(1.5 + rank), sm=cp.sum(float(rank + 1)),
           gathered=cp.all_gather_object(dict(r=rank)), active=cp.active)
cp.barrier()
cp.close()
if rank == 0:
    print(json.dumps(out))
"""


def test_control_plane_falls_back_to_gloo_when_rccl_cannot_come_up(tmp_path):
    """bench.py / audio_to_codes ask for RCCL; where it cannot be brought up (here: no GPU; on a node: IPC handles, topology, two
    ranks on one card) the SAME processes carry barrier / max / sum / gather over gloo on CPU tensors and say so -- the data path
    has no collective, so the control plane must never be the thing that takes an 8-GPU run down."""
    script = os.path.join(str(tmp_path), "cp_worker.py")
    with open(script, "w") as f:
        f.write(CP_WORKER.format(root=ROOT))
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
                          "--master-port", "29547", script], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    r = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert r["backend"] == "gloo" and r["active"] and r["reason"] and r["reason"].startswith("rank 0:")
    assert r["mx"] == 2.5 and r["sm"] == 3.0 and r["gathered"] == [{"r": 0}, {"r": 1}]
    assert "[control plane] 2 ranks" in out.stderr and "carried by gloo" in out.stderr
    # a single process has no control plane at all
    from realtime_codec_agent_amd.dist_utils import ControlPlane
    env = {k: os.environ.pop(k) for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK") if k in os.environ}
    try:
        cp = ControlPlane(prefer="nccl")
        assert cp.backend == "none" and not cp.active and cp.max(3.0) == 3.0 and cp.all_gather_object(1) == [1]
        cp.barrier(); cp.close()
    finally:
        os.environ.update(env)


def test_shard_plans():
    d = [5.0, 1.0, 3.0, 3.0, 8.0, 0.5, 2.0]
    for world in (1, 2, 3, 8):
        shards = shard_by_duration(d, world)
        assert sorted(i for s in shards for i in s) == list(range(len(d))) and len(shards) == world
        loads = [sum(d[i] for i in s) for s in shards]
        assert max(loads) - min(loads) <= max(d)
    assert shard_by_duration(d, 2) == shard_by_duration(d, 2)
    for n, w in ((10, 3), (7, 8), (36000, 8)):
        rs = [shard_chunk_range(n, w, r) for r in range(w)]
        assert rs[0][0] == 0 and rs[-1][1] == n and all(rs[i][1] == rs[i + 1][0] for i in range(w - 1))
        assert max(b - a for a, b in rs) - min(b - a for a, b in rs) <= 1


def test_c_abi_library_exports_every_declared_symbol():
    """The in-tree HIP library loads without a GPU and exports exactly the entry points include/rca.h
    declares (no compute call is made here)."""
    from realtime_codec_agent_amd import _native
    header = open(os.path.join(ROOT, "include", "rca.h")).read()
    declared = sorted(set(re.findall(r"\b(rca_[a-z0-9_]+)\s*\(", header)))
    assert declared == sorted(_native.ABI_SYMBOLS)
    if _native.needs_build():
        _native.build()
    lib = _native.lib()
    missing = [s for s in declared if not hasattr(lib, s)]
    assert not missing, missing
    assert b"gfx950" in lib.rca_version()
    # error behaviour without a device: bad arguments are rejected before any HIP call
    lib.rca_codec_hop.restype = ctypes.c_int
    assert lib.rca_codec_hop(None, None) == -1 and b"null" in lib.rca_last_error()


def _load_bench():
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_under_test", os.path.join(ROOT, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_bench_gpus_flag_starts_one_rank_per_gpu(monkeypatch, capsys):
    """`python bench.py --gpus N` as the driver invokes it (no launcher, WORLD_SIZE unset) must start N ranks itself -- as CHILD
    processes through torch.distributed.run, before anything touches the GPU -- relay rank 0's JSON line and exit with the ranks'
    status; under a launcher a --gpus / WORLD_SIZE mismatch is refused (the reference's parallelism is N pinned processes:
    encode_audio_gpu_{1..4}.sh, realtime_agent_v2.py:832-836)."""
    import subprocess
    import types
    bench = _load_bench()
    seen = {}

    def fake_run(cmd, env=None, stdout=None, text=None):
        seen["cmd"], seen["env"] = cmd, env
        return types.SimpleNamespace(returncode=0, stdout='noise\n{"metric": "m", "value": 1.0, "n_gpus": 4}\n')
    monkeypatch.setattr(subprocess, "run", fake_run)
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "7", "--warmup", "2"])
    called = {"cuda": False}
    import torch
    monkeypatch.setattr(torch.cuda, "set_device", lambda *a, **k: called.__setitem__("cuda", True))
    with pytest.raises(SystemExit) as e:
        bench.main()
    assert e.value.code == 0 and not called["cuda"]
    cmd = seen["cmd"]
    assert cmd[1:3] == ["-m", "torch.distributed.run"] and "--nproc-per-node=4" in cmd and "127.0.0.1" in cmd
    assert cmd[-6:] == ["--gpus", "4", "--steps", "7", "--warmup", "2"] and cmd[-7].endswith("bench.py")
    assert seen["env"].get("HSA_ENABLE_IPC_MODE_LEGACY") == "0"
    assert capsys.readouterr().out.strip() == '{"metric": "m", "value": 1.0, "n_gpus": 4}'
    # a failing rank: non-zero exit even though nothing was printed
    monkeypatch.setattr(subprocess, "run", lambda *a, **k: types.SimpleNamespace(returncode=0, stdout="no json here\n"))
    with pytest.raises(SystemExit) as e:
        bench.main()
    assert e.value.code != 0
    # under a launcher with the wrong world size
    monkeypatch.setenv("WORLD_SIZE", "2")
    with pytest.raises(SystemExit) as e:
        bench.main()
    assert e.value.code not in (0, None) and not called["cuda"]


class _OracleRowsEncoder:
    """CPU stand-in for HipWindowEncoder with the SAME two entry points: encode (one file) and encode_many (a super-batch through
    the window table, every window encoded on its own by the C oracle, exactly what rca_codec_encode_rows_dev does per row)."""

    def __init__(self):
        from oracle.codec import OracleCodec
        from realtime_codec_agent_amd.codec_model import init_codec_weights, tiny_codec_config
        self.cfg = tiny_codec_config()
        self.oc = OracleCodec(self.cfg, init_codec_weights(self.cfg, seed=0))
        self.passes = []

    def encode(self, audio, chunk, ctx, batch):
        return self.oc.encode_windows(audio, chunk, ctx)

    def encode_many(self, audios, chunk, ctx, batch_windows):
        from realtime_codec_agent_amd.audio_to_codes import window_table
        W, fpc = max(chunk, ctx), int((chunk / self.cfg.sample_rate) * self.cfg.framerate)
        lengths = [a.shape[-1] for a in audios for _ in range(a.shape[0])]
        src_base = np.cumsum([0] + lengths)[:-1]
        n_codes = [(n // chunk) * fpc for n in lengths]
        dst_base = np.cumsum([0] + n_codes)[:-1]
        flat = np.concatenate([a[c] for a in audios for c in range(a.shape[0])])
        T, src, dst = window_table(lengths, chunk, W, fpc, src_base, dst_base)
        out = np.full(int(sum(n_codes)), -1, np.int64)
        i = 0
        while i < len(T):
            j = i
            while j < len(T) and T[j] == T[i]:
                j += 1
            for k in range(i, j, batch_windows):
                rows = np.stack([flat[s:s + T[i]] for s in src[k:min(j, k + batch_windows)]])
                self.passes.append(rows.shape)
                codes = self.oc.encode(rows)[:, -fpc:]
                for d, c in zip(dst[k:k + len(rows)], codes):
                    out[d:d + fpc] = c
            i = j
        assert (out >= 0).all()
        return out, [(int(b), int(b + n)) for b, n in zip(dst_base, n_codes)], (lambda: None)


def _tree(root):
    out = {}
    for r, _, fs in os.walk(root):
        for f in fs:
            with open(os.path.join(r, f), "rb") as fh:
                out[os.path.relpath(os.path.join(r, f), root)] = fh.read()
    return out


def test_cross_file_batching_writes_the_same_tree(tmp_path):
    """audio_to_codes with windows batched ACROSS files (window table, warm-up windows grouped by length, reader / writer threads)
    writes byte for byte the tree of the one-file-at-a-time loop; passes are filled from several files."""
    from realtime_codec_agent_amd import audio_to_codes
    tmp = str(tmp_path)
    raw = _make_corpus(tmp)
    enc = _OracleRowsEncoder()
    base = ["--audio_path", raw, "--stereo", "--batch_size", "16", "--context_secs", "0.5"]
    a = audio_to_codes.main(base + ["--codes_path", os.path.join(tmp, "simple"), "--one_file_at_a_time"], encoder=enc, backend="gloo")
    assert not enc.passes
    b = audio_to_codes.main(base + ["--codes_path", os.path.join(tmp, "piped"), "--super_batch_samples", "60000", "--reader_threads", "2"],
                            encoder=enc, backend="gloo")
    ta, tb = _tree(os.path.join(tmp, "simple")), _tree(os.path.join(tmp, "piped"))
    assert ta.keys() == tb.keys() and len(ta) == 1 + 2 * 5
    assert all(ta[k] == tb[k] for k in ta)
    assert a["codes"] == b["codes"] and abs(a["audio_hours"] - b["audio_hours"]) < 1e-12
    full = [p for p in enc.passes if p[1] == 8000]
    assert full and max(p[0] for p in full) == 16                      # full passes of 16 windows ...
    assert any(p[1] == 1600 and p[0] > 2 for p in enc.passes)          # ... and warm-up windows of several files in one pass
    # mono down-mix and a window longer than any file's tail
    c = audio_to_codes.main(["--audio_path", raw, "--codes_path", os.path.join(tmp, "m1"), "--one_file_at_a_time"], encoder=enc, backend="gloo")
    d = audio_to_codes.main(["--audio_path", raw, "--codes_path", os.path.join(tmp, "m2")], encoder=enc, backend="gloo")
    tc, td = _tree(os.path.join(tmp, "m1")), _tree(os.path.join(tmp, "m2"))
    assert tc.keys() == td.keys() and all(tc[k] == td[k] for k in tc) and c["codes"] == d["codes"]
    # a --stereo corpus that mixes mono and stereo files (either kind first in a super-batch): every file keeps its own channel
    # count, and every code row lands under its own file's name
    mixed = os.path.join(tmp, "mixed")
    os.makedirs(mixed)
    for name, ch, n in (("a_mono.wav", 1, 11200), ("b_stereo.wav", 2, 9600), ("c_mono.wav", 1, 14400), ("d_stereo.wav", 2, 8000)):
        sig = np.stack([bench_signal(n, 70 + i) for i in range(ch)])
        with wave.open(os.path.join(mixed, name), "wb") as w:
            w.setnchannels(ch); w.setsampwidth(2); w.setframerate(16000)
            w.writeframes((np.clip(sig.T, -1, 1) * 32767).astype("<i2").tobytes())
    mbase = ["--audio_path", mixed, "--stereo", "--batch_size", "16", "--context_secs", "0.5"]
    e = audio_to_codes.main(mbase + ["--codes_path", os.path.join(tmp, "x1"), "--one_file_at_a_time"], encoder=enc, backend="gloo")
    f = audio_to_codes.main(mbase + ["--codes_path", os.path.join(tmp, "x2")], encoder=enc, backend="gloo")
    te, tf = _tree(os.path.join(tmp, "x1")), _tree(os.path.join(tmp, "x2"))
    assert te.keys() == tf.keys() and len(te) == 1 + 6 and all(te[k] == tf[k] for k in te) and e["codes"] == f["codes"]
