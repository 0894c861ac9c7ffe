"""Batch encoder CLI -- the job `encode_audio_gpu_{1..4}.sh` / `encode_audio_stereo.sh` run through the
third-party `python -m codec_bpe.audio_to_codes` (encode_audio_gpu_1.sh:1-8):

    python -m realtime_codec_agent_amd.audio_to_codes --audio_path data/audio/raw --codes_path data/audio/codes \
        --chunk_size_secs 0.1 --context_secs 2.0 --batch_size 256 --codec_model MagiCodec-50Hz-Base [--stereo] \
        [--audio_filter CallFriend CallHome ...]

Same flags; output layout as the reference's consumers expect it:
<codes_path>/<codec_model>/<chunk>s_<ctx>s/{mono,stereo}/<relative path>_c<channel>.npy holding an int64 array
(num_codebooks=1, T), plus codec_info.json {num_codebooks, codebook_size, framerate} in that leaf directory
(prep_lm_dataset.py:47-52; lm_dataset_builder.py:79-83,397-408).

Every 0.1 s chunk is encoded with 2.0 s of left context and only its own codes are kept, so the offline codes
equal what streaming AudioTokenizer.tokenize_audio emits.  Multi-GPU: one process per GPU
(`torchrun --nproc-per-node N` or RANK/WORLD_SIZE env); files are dealt to ranks by duration; no collective on
the data path.  Inputs: .wav (PCM16/float32) and .npy ([N] or [C,N]); compressed formats need a decoder that is
not part of this build.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time
import wave
from typing import List, Optional, Sequence, Tuple

import numpy as np

from .dist_utils import env_rank_world, init_dist, max_over_ranks, shard_by_duration, sum_over_ranks

AUDIO_EXTS = (".wav", ".npy")


def read_audio(path: str) -> Tuple[int, np.ndarray]:
    """-> (sample_rate, float32 [C,N])."""
    if path.endswith(".npy"):
        a = np.load(path)
        sr = 16000
        if a.dtype == np.int16:
            a = a.astype(np.float32) / 32768.0
        a = np.atleast_2d(a.astype(np.float32, copy=False))
        return sr, a
    with wave.open(path, "rb") as w:
        sr, ch, sw, n = w.getframerate(), w.getnchannels(), w.getsampwidth(), w.getnframes()
        raw = w.readframes(n)
    if sw == 2:
        a = np.frombuffer(raw, dtype="<i2").astype(np.float32) / 32768.0
    elif sw == 4:
        a = np.frombuffer(raw, dtype="<f4").astype(np.float32)
    else:
        raise ValueError(f"{path}: unsupported sample width {sw}")
    return sr, np.ascontiguousarray(a.reshape(-1, ch).T)


def probe_duration(path: str) -> float:
    if path.endswith(".npy"):
        a = np.load(path, mmap_mode="r")
        return a.shape[-1] / 16000.0
    with wave.open(path, "rb") as w:
        return w.getnframes() / float(w.getframerate())


def list_audio_files(audio_path: str, audio_filter: Optional[Sequence[str]]) -> List[str]:
    out = []
    for root, _, files in os.walk(audio_path):
        for f in sorted(files):
            p = os.path.join(root, f)
            if f.lower().endswith(AUDIO_EXTS) and (not audio_filter or any(s in p for s in audio_filter)):
                out.append(p)
    return sorted(out)


class HipWindowEncoder:
    """audio [C,N] f32 -> codes [C, n_chunks*fpc] int64 through rca_codec_encode_windows_dev."""

    def __init__(self, codec_model: str, device_index: int):
        import torch
        from .codec import load_magicodec_model
        self.torch = torch
        self.device = torch.device("cuda", device_index)
        self.model, _, _ = load_magicodec_model(codec_model, self.device)
        self.cfg = self.model.cfg

    def encode(self, audio: np.ndarray, chunk: int, ctx: int, batch_windows: int) -> np.ndarray:
        torch = self.torch
        hip = self.model.hip
        C, N = audio.shape
        n_chunks = N // chunk
        fpc = hip.frames_per_chunk(chunk)
        if n_chunks == 0:
            return np.zeros((C, 0), np.int64)
        dev = torch.from_numpy(np.ascontiguousarray(audio)).to(self.device)
        out = torch.empty((C, n_chunks * fpc), dtype=torch.int64, device=self.device)
        hip.encode_windows_dev(dev.data_ptr(), C, N, chunk, ctx, batch_windows, out.data_ptr(), n_chunks * fpc,
                               torch.cuda.current_stream(self.device).cuda_stream)
        return out.cpu().numpy()


def encode_files(files: Sequence[str], encoder, args, rank: int = 0) -> Tuple[float, int]:
    """Encode `files` with `encoder` (anything with .cfg and .encode(audio, chunk, ctx, batch)); returns
    (audio seconds, codes written)."""
    cfg = encoder.cfg
    sr = cfg.sample_rate
    chunk = int(args.chunk_size_secs * sr)
    ctx = int(args.context_secs * sr)
    sub = "stereo" if args.stereo else "mono"
    out_root = os.path.join(args.codes_path, args.codec_model, f"{args.chunk_size_secs}s_{args.context_secs}s", sub)
    os.makedirs(out_root, exist_ok=True)
    # consumers call get_codec_info(<codes_path>/<model>/<chunk>s_<ctx>s/{mono,stereo}) (tools/total_duration_codes.py:6-7)
    info = os.path.join(out_root, "codec_info.json")
    if rank == 0 and not os.path.exists(info):
        with open(info, "w") as f:
            json.dump({"num_codebooks": 1, "codebook_size": cfg.codebook_size, "framerate": cfg.framerate}, f)
    total_secs, total_codes = 0.0, 0
    for path in files:
        fsr, audio = read_audio(path)
        if fsr != sr:
            from .audio_tokenizer import _resample
            audio = _resample(audio, fsr, sr)
        if not args.stereo and audio.shape[0] > 1:
            audio = audio.mean(axis=0, keepdims=True)
        codes = encoder.encode(audio, chunk, ctx, args.batch_size)
        rel = os.path.splitext(os.path.relpath(path, args.audio_path))[0]
        for c in range(codes.shape[0]):
            dst = os.path.join(out_root, f"{rel}_c{c}.npy")
            os.makedirs(os.path.dirname(dst), exist_ok=True)
            np.save(dst, codes[c][None, :])  # (num_codebooks, T)
        total_secs += audio.shape[-1] / sr
        total_codes += int(codes.size)
    return total_secs, total_codes


def build_parser() -> argparse.ArgumentParser:
    ap = argparse.ArgumentParser(description="Encode audio files to codec codes (sharded batch encode).")
    ap.add_argument("--audio_path", required=True)
    ap.add_argument("--codes_path", required=True)
    ap.add_argument("--chunk_size_secs", type=float, default=0.1)
    ap.add_argument("--context_secs", type=float, default=2.0)
    ap.add_argument("--batch_size", type=int, default=256)
    ap.add_argument("--codec_model", default="MagiCodec-50Hz-Base")
    ap.add_argument("--stereo", action="store_true")
    ap.add_argument("--audio_filter", nargs="+")
    ap.add_argument("--receptive_field_trim", action="store_true",
                    help="encode only what each chunk's kept frames can see instead of the whole context window: identical "
                         "codes for this build's conv codec, ~8x faster (rca_codec_set_window_trim)")
    return ap


def main(argv=None, encoder=None, backend: Optional[str] = None) -> dict:
    args = build_parser().parse_args(argv)
    rank, world, local = env_rank_world()
    if encoder is None:
        encoder = HipWindowEncoder(args.codec_model, local)
        encoder.model.hip.set_window_trim(args.receptive_field_trim)
        backend = backend or "nccl"
    dist = init_dist(backend or "gloo", local) if world > 1 else None
    files = list_audio_files(args.audio_path, args.audio_filter)
    shards = shard_by_duration([probe_duration(f) for f in files], world)
    mine = [files[i] for i in shards[rank]]
    dev = getattr(encoder, "device", None) if backend == "nccl" else None
    if dist is not None:
        dist.barrier()
    t0 = time.perf_counter()
    secs, ncodes = encode_files(mine, encoder, args, rank)
    if dev is not None:
        encoder.torch.cuda.synchronize(dev)
    if dist is not None:
        dist.barrier()
    elapsed = max_over_ranks(time.perf_counter() - t0, dist, dev)
    total_secs = sum_over_ranks(secs, dist, dev)
    total_codes = sum_over_ranks(float(ncodes), dist, dev)
    summary = dict(files=len(files), world_size=world, audio_hours=total_secs / 3600.0, codes=int(total_codes), elapsed_s=elapsed,
                   audio_hours_per_hour=(total_secs / elapsed) if elapsed > 0 else None)
    if rank == 0:
        print(json.dumps(summary))
    return summary


if __name__ == "__main__":
    main()
