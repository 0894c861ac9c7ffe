"""The call script shared by tests/golden/make_tokenizer_golden.py (which replays it on the REFERENCE AudioTokenizer,
loaded from /root/reference by path) and tests/test_tokenizer_parity.py (which replays it on this repo's
AudioTokenizer over the oracle on the CPU and over MagiCodecHIP on the GPU).

A scenario is a constructor kwargs dict plus a list of calls; `replay` drives any object with the reference's
AudioTokenizer surface (audio_tokenizer.py:10-215) and returns a flat {key: ndarray} record.  Strings are stored as
code-point arrays, PCM as float32 -- data only.
"""
import hashlib

import numpy as np


def _signal(n, seed, channels=None):
    rng = np.random.default_rng(seed)
    shape = (n,) if channels is None else (channels, n)
    knots = np.arange(0, n + 800, 800)
    env = np.abs(np.interp(np.arange(n), knots, rng.normal(0, 0.3, len(knots))))
    return np.clip(rng.normal(0, 1, shape) * env, -1, 1).astype(np.float32)


def _cp(s):
    return np.array([ord(c) for c in s], dtype=np.int64)


def scenarios():
    """name -> (ctor kwargs, [(op, args...)])."""
    sc = {}
    # mono, 100 ms chunks, 2.6 s: crosses the 2.0 s window (trim rule :74, keep-last :99-101, 100-code context :113)
    x = _signal(41600, 11)
    sc["mono_100ms"] = (dict(num_channels=1), [("tok", x[i:i + 1600]) for i in range(0, 41600, 1600)], dict(detok_preroll=320))
    # mono, 80 ms chunks (BASELINE's frame), 2.4 s
    x = _signal(38400, 12)
    sc["mono_80ms"] = (dict(num_channels=1), [("tok", x[i:i + 1280]) for i in range(0, 38400, 1280)], dict(detok_preroll=320))
    # stereo interleave (:96,116), window shorter than the run, no preroll
    x = _signal(19200, 13, channels=2)
    sc["stereo_100ms"] = (dict(num_channels=2, context_secs=0.5), [("tok", x[:, i:i + 1600]) for i in range(0, 19200, 1600)],
                          dict(detok_preroll=0))
    # short context, preroll larger than one chunk's samples
    x = _signal(16000, 14)
    sc["mono_ctx0.3_preroll2000"] = (dict(num_channels=1, context_secs=0.3), [("tok", x[i:i + 1600]) for i in range(0, 16000, 1600)],
                                     dict(detok_preroll=2000))
    # chunked_tokenize_audio with a ragged last chunk (:52-65), (sr, array) tuples, int16 input, a chunk longer than the window
    x = _signal(16800, 15)
    sc["chunked_ragged"] = (dict(num_channels=1, context_secs=0.5), [("chunked", x, 0.1)], dict(detok_preroll=320, detok_step=7))
    sc["chunked_tuple_80ms"] = (dict(num_channels=1), [("chunked", (16000, _signal(9000, 16)), 0.08)], dict(detok_preroll=160, detok_step=4))
    xi = (np.clip(_signal(6400, 17), -1, 1) * 32767).astype(np.int16)
    sc["int16_and_long_chunk"] = (dict(num_channels=1, context_secs=0.2),
                                  [("tok", xi[:1600]), ("tok", (16000, xi[1600:3200])), ("tok", _signal(8000, 18)), ("tok", xi[3200:4800])],
                                  dict(detok_preroll=320))
    # a [2, T] array into a mono tokenizer is down-mixed (librosa.to_mono = channel mean, :209-210)
    x = _signal(4800, 19, channels=2)
    sc["downmix"] = (dict(num_channels=1, context_secs=0.25), [("tok", x[:, i:i + 1600]) for i in range(0, 4800, 1600)],
                     dict(detok_preroll=0, empty_at=1))   # "" with no preroll returns the whole decoded window ([-0:], :143-144)
    # stereo strings with a hanging (odd) character (:161-168) and an empty string
    x = _signal(9600, 20, channels=2)
    sc["stereo_hanging"] = (dict(num_channels=2, context_secs=0.4), [("tok", x[:, i:i + 1600]) for i in range(0, 9600, 1600)],
                            dict(detok_preroll=320, hanging=True))
    return sc


def replay(make_tokenizer, name):
    """Runs scenario `name` on make_tokenizer(**ctor_kwargs) and returns the record."""
    ctor, calls, opt = scenarios()[name]
    at = make_tokenizer(**ctor)
    rec = {"framerate": np.float64(at.framerate), "context_samples": np.int64(at.context_samples), "context_frames": np.int64(at.context_frames),
           "sampling_rate": np.int64(at.sampling_rate), "codebook_size": np.int64(at.codebook_size), "num_codebooks": np.int64(at.num_codebooks)}
    strings = []
    for call in calls:
        if call[0] == "tok":
            strings.append(at.tokenize_audio(call[1]))
        else:
            whole = at.chunked_tokenize_audio(call[1], call[2])
            step = opt.get("detok_step", 5)
            strings += [whole[i:i + step] for i in range(0, len(whole), step)]
    rec["tok_lens"] = np.array([len(s) for s in strings], np.int64)
    rec["tok_chars"] = _cp("".join(strings))
    rec["tok_ctx_len"] = np.int64(at.tokenize_context.shape[-1])
    rec["secs_of_first"] = np.float64(at.get_audio_codes_str_secs(strings[0]))
    # feed the strings back through detokenize_audio (:105-149), carrying the preroll the way the agent does
    # (realtime_agent_v2.py:560: a fixed preroll every call) and, for the hanging case, re-attaching dropped characters
    pcm, hang, pre = [], [], []
    carry = ""
    for i, s in enumerate(strings):
        if opt.get("hanging"):
            s = carry + s
            if i % 2 == 0 and len(s) > 1:
                s, carry = s[:-1], s[-1:]      # odd length for a 2-channel tokenizer
            else:
                carry = ""
            if i == 3:
                s = ""
        if opt.get("empty_at") == i:
            s = ""
        (sr, audio), end_hanging, preroll_left = at.detokenize_audio(s, preroll_samples=opt["detok_preroll"])
        assert sr == at.sampling_rate
        pcm.append(np.asarray(audio, dtype=np.float32))
        hang.append(end_hanging)
        pre.append(preroll_left)
    rec["pcm_shapes"] = np.array([list(p.shape) + [0] * (2 - p.ndim) for p in pcm], np.int64)
    flat = np.concatenate([p.reshape(-1) for p in pcm]) if pcm else np.zeros(0, np.float32)
    rec["pcm_dec"] = flat[::6].copy()      # every 6th sample for diagnosis, the SHA-256 of all of them for exactness
    rec["pcm_sha256"] = np.frombuffer(hashlib.sha256(np.ascontiguousarray(flat).tobytes()).digest(), dtype=np.uint8).copy()
    rec["hang_lens"] = np.array([len(h) for h in hang], np.int64)
    rec["hang_chars"] = _cp("".join(hang))
    rec["preroll_left"] = np.array(pre, np.int64)
    rec["detok_ctx"] = _cp(at.detokenize_context)
    sil = at._encode_silence(0.5)
    rec["silence_codes"] = np.asarray(sil.cpu() if hasattr(sil, "cpu") else sil).astype(np.int64).reshape(-1)
    emb = at.get_codec_embeddings()
    emb = np.asarray(emb.cpu() if hasattr(emb, "cpu") else emb, dtype=np.float32)
    rec["emb_shape"] = np.array(emb.shape, np.int64)
    rec["emb_rows"] = emb[::97].copy()
    at.reset_context()
    rec["reset_ok"] = np.int64(at.tokenize_context.shape == (ctor.get("num_channels", 1), 0) and at.detokenize_context == "")
    return rec
