"""CLI-level throughput of the batch encoder on a synthetic corpus of short utterances (10-60 s stereo .wav, like the reference's
corpora): `audio_to_codes` one file at a time vs. windows batched across files with read / encode / write overlapped.
usage: cli_corpus_bench.py [hours=1.5] [seed=0]   (writes the corpus under $TMPDIR or /tmp)"""
import json, os, shutil, sys, tempfile, time, wave
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from realtime_codec_agent_amd import audio_to_codes

hours = float(sys.argv[1]) if len(sys.argv) > 1 else 1.5
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
root = tempfile.mkdtemp(prefix="rca_corpus_")
raw = os.path.join(root, "raw")
os.makedirs(raw)
total, i = 0.0, 0
t0 = time.perf_counter()
while total < hours * 3600:
    secs = float(rng.uniform(10, 60))
    n = int(secs * 16000)
    t = np.arange(n) / 16000.0
    sig = np.stack([0.1 * np.sin(2 * np.pi * f * t) + rng.normal(0, 0.02, n) for f in (220.0 + i, 330.0 + i)])
    d = os.path.join(raw, f"spk{i % 17:02d}")
    os.makedirs(d, exist_ok=True)
    with wave.open(os.path.join(d, f"utt{i:05d}.wav"), "wb") as w:
        w.setnchannels(2); w.setsampwidth(2); w.setframerate(16000)
        w.writeframes((np.clip(sig.T, -1, 1) * 32767).astype("<i2").tobytes())
    total += secs
    i += 1
print(f"corpus: {i} files, {total / 3600:.2f} h stereo, written in {time.perf_counter() - t0:.1f} s", flush=True)
res = {}
sweep = [int(x) for x in os.environ.get("RCA_CLI_SWEEP", "").split(",") if x]      # super-batch sizes (log2) to try, e.g. 22,23,24,25,26
runs = [("one_file_at_a_time", ["--one_file_at_a_time"]), ("cross_file_pipelined", []), ("cross_file_pipelined+rf_trim", ["--receptive_field_trim"])]
runs += [(f"super_batch_2^{k}", ["--super_batch_samples", str(1 << k)]) for k in sweep]
for name, extra in runs:
    out = os.path.join(root, name.replace("+", "_").replace("^", ""))
    s = audio_to_codes.main(["--audio_path", raw, "--codes_path", out, "--stereo"] + extra)
    res[name] = dict(audio_hours_per_hour=s["audio_hours_per_hour"], elapsed_s=s["elapsed_s"])
    st = s.get("stages")
    if st:
        res[name]["stages"] = {k: (round(v, 3) if isinstance(v, float) else v) for k, v in st.items()}
    print(name, json.dumps(res[name]), flush=True)
same = True
a, b = os.path.join(root, "one_file_at_a_time"), os.path.join(root, "cross_file_pipelined")
for r, _, fs in os.walk(a):
    for f in fs:
        same &= open(os.path.join(r, f), "rb").read() == open(os.path.join(b, os.path.relpath(os.path.join(r, f), a)), "rb").read()
print(json.dumps(dict(files=i, audio_hours=total / 3600, trees_identical=bool(same), **res)))
shutil.rmtree(root, ignore_errors=True)
