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

Throughput path (HipWindowEncoder.encode_many, used whenever the encoder offers it): the corpora of the reference are tens
of thousands of 10-60 s utterances, so a 256-window pass must not stop at a file boundary.  Files are gathered into
"super-batches" (a few hundred MB of PCM), uploaded once, and EVERY window of every file in the super-batch -- the full
2.0 s windows and the shorter warm-up windows at the start of each file, grouped by length -- goes through
rca_codec_encode_rows_dev in full passes.  Three stages overlap: reader threads (disk -> float32), the main thread
(pack, H2D from pinned memory, enqueue), a writer thread (D2H'd codes -> .npy).  The output tree is byte-identical to the
one-file-at-a-time path (tests compare the two).
"""
from __future__ import annotations

import argparse
import json
import os
import queue
import sys
import threading
import time
import wave
from concurrent.futures import ThreadPoolExecutor
from typing import Iterator, List, Optional, Sequence, Tuple

import numpy as np

from .dist_utils import ControlPlane, env_rank_world, shard_by_duration

AUDIO_EXTS = (".wav", ".npy")
# PCM samples (all channels) per super-batch: 2^25 = 35 min of mono audio = ~0.28 s of encoder time on an MI355X.  Measured on a
# 2.0 h corpus of 10-60 s stereo utterances (profiles/r04/cli_corpus_bench.txt): 2^22 2 850, 2^23 3 110, 2^24 2 950-3 100, 2^25 3 170,
# 2^26 3 110 audio-hours/hour -- small super-batches leave the window-length groups of the files' warm-up windows with a few dozen
# rows per pass, a 2^26 one spends 0.5 s enqueueing before the GPU has anything (and swallowed a whole 0.5 h corpus: nothing
# overlapped anything).  From the second super-batch on reading / packing the next one and writing the previous one hide behind the GPU.
DEFAULT_SUPER_BATCH = 1 << 25
# The reference's corpora are mp3 (tools/sph_to_mp3.py:28-41) and codec_bpe reads them through librosa / soundfile.  Neither is part
# of this image; when one of them is importable these extensions are picked up too, otherwise such files are reported and skipped.
COMPRESSED_EXTS = (".mp3", ".flac", ".ogg", ".m4a")


def _compressed_reader():
    try:
        import soundfile as sf

        def rd(path):
            a, sr = sf.read(path, dtype="float32", always_2d=True)
            return sr, np.ascontiguousarray(a.T)
        return rd
    except Exception:
        pass
    try:
        import librosa

        def rd(path):
            a, sr = librosa.load(path, sr=None, mono=False)
            return sr, np.atleast_2d(a).astype(np.float32)
        return rd
    except Exception:
        return None


def read_audio(path: str) -> Tuple[int, np.ndarray]:
    """-> (sample_rate, float32 [C,N])."""
    if path.lower().endswith(COMPRESSED_EXTS):
        rd = _compressed_reader()
        if rd is None:
            raise RuntimeError(f"{path}: decoding compressed audio needs soundfile or librosa, neither is installed")
        return rd(path)
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
    if path.lower().endswith(COMPRESSED_EXTS):
        sr, a = read_audio(path)
        return a.shape[-1] / float(sr)
    if path.endswith(".npy"):
        a = np.load(path, mmap_mode="r")
        return a.shape[-1] / 16000.0
    with wave.open(path, "rb") as w:
        return w.getnframes() / float(w.getframerate())


def list_audio_files(audio_path: str, audio_filter: Optional[Sequence[str]]) -> List[str]:
    out, skipped = [], 0
    exts = AUDIO_EXTS + (COMPRESSED_EXTS if _compressed_reader() is not None else ())
    for root, _, files in os.walk(audio_path):
        for f in sorted(files):
            p = os.path.join(root, f)
            if audio_filter and not any(s in p for s in audio_filter):
                continue
            if f.lower().endswith(exts):
                out.append(p)
            elif f.lower().endswith(COMPRESSED_EXTS):
                skipped += 1
    if skipped:
        print(f"audio_to_codes: {skipped} compressed file(s) skipped (no soundfile / librosa in this environment)", file=sys.stderr)
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

    RING = 3      # super-batches in flight: one being packed, one on the GPU, one being written

    def _slot(self, k: int, n_samples: int, n_codes: int, n_windows: int):
        """Ring slot k: pinned staging + device buffers, grown on demand and reused (a pinned allocation of a few hundred MB costs
        tens of milliseconds: once per slot, not once per super-batch)."""
        torch = self.torch
        if not hasattr(self, "_ring"):
            self._ring = [dict() for _ in range(self.RING)]
        sl = self._ring[k]
        if sl.get("ev") is not None:
            sl["ev"].synchronize()                              # the slot's previous super-batch has left the GPU (its codes are on the host)
        if sl.get("free") is not None:
            sl["free"].wait()                                   # ... and the writer is done with its host codes
        sl["free"] = threading.Event()
        def grow(name, n, dtype, pinned):
            cur = sl.get(name)
            if cur is None or cur.numel() < n:
                cap = max(int(n * 1.25), 1)
                sl[name] = torch.empty(cap, dtype=dtype).pin_memory() if pinned else torch.empty(cap, dtype=dtype, device=self.device)
            return sl[name]
        return (grow("stage", n_samples, torch.float32, True), grow("dev_audio", n_samples, torch.float32, False),
                grow("host_codes", n_codes, torch.int64, True), grow("dev_codes", n_codes, torch.int64, False),
                grow("tab_host", 2 * n_windows, torch.int64, True), grow("tab_dev", 2 * n_windows, torch.int64, False), sl)

    def encode_many(self, audios: Sequence[np.ndarray], chunk: int, ctx: int, batch_windows: int):
        """audios: float32 [C_f, N_f] per file (a --stereo corpus may mix mono and stereo files: every file brings its own
        channel count).  -> (pinned host int64 codes [total], [(a, b)] slice of every (file, channel) row, wait()) -- wait()
        blocks until the codes have landed in the host buffer.  Everything up to the D2H copy is enqueued asynchronously, so the
        caller can prepare the next super-batch while this one runs.  Staging and device buffers come from a ring of RING slots
        (the returned host buffer stays valid until RING - 1 further calls have been made)."""
        torch, hip = self.torch, self.model.hip
        t_in = time.perf_counter()
        W = max(chunk, ctx)
        fpc = hip.frames_per_chunk(chunk)
        lengths = [a.shape[-1] for a in audios for _ in range(a.shape[0])]      # one entry per (file, channel) row
        # one pinned staging buffer, rows back to back (file-major, channel-minor), each start aligned to 4 samples
        row_len = [((n + 3) // 4) * 4 for n in lengths]
        src_base = np.concatenate([[0], np.cumsum(row_len)[:-1]]).astype(np.int64)
        total = int(sum(row_len))
        n_codes = [(n // chunk) * fpc for n in lengths]
        dst_base = np.concatenate([[0], np.cumsum(n_codes)[:-1]]).astype(np.int64)
        total_codes = int(sum(n_codes))
        slices = [(int(b), int(b + n)) for b, n in zip(dst_base, n_codes)]
        T, src, dst = window_table(lengths, chunk, W, fpc, src_base, dst_base)
        self._calls = getattr(self, "_calls", 0) + 1
        stage, dev_audio, host_codes, dev_codes, tab_host, tab_dev, sl = self._slot(self._calls % self.RING, total, total_codes, len(T))
        t_slot = time.perf_counter()
        sv = stage.numpy()
        r = 0
        for a in audios:
            for c in range(a.shape[0]):
                sv[src_base[r]:src_base[r] + a.shape[-1]] = a[c]
                r += 1
        st_ = getattr(self, "stage_times", None)
        if st_ is None:
            st_ = self.stage_times = dict(slot_wait_s=0.0, pack_s=0.0, enqueue_s=0.0, gpu_ms=0.0, super_batches=0, passes=0)
        if len(T) == 0:
            sl["free"].set()
            return host_codes.numpy()[:0], slices, (lambda: None)
        tab_host.numpy()[:len(T)] = src
        tab_host.numpy()[len(T):2 * len(T)] = dst
        t_pack = time.perf_counter()
        with torch.cuda.device(self.device):
            st = torch.cuda.current_stream(self.device)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            dev_audio[:total].copy_(stage[:total], non_blocking=True)
            tab_dev[:2 * len(T)].copy_(tab_host[:2 * len(T)], non_blocking=True)
            e0.record(st)
            src_ptr, dst_ptr = tab_dev.data_ptr(), tab_dev.data_ptr() + 8 * len(T)
            i, n = 0, len(T)
            while i < n:
                t = int(T[i])
                j = i
                while j < n and T[j] == t:
                    j += 1                                        # rows [i, j): one window length
                for k in range(i, j, batch_windows):
                    b = min(batch_windows, j - k)
                    hip.encode_rows_dev(dev_audio.data_ptr(), src_ptr + 8 * k, b, t, fpc, dev_codes.data_ptr(),
                                        dst_ptr + 8 * k, total, st.cuda_stream)
                    st_["passes"] += 1
                i = j
            e1.record(st)
            host_codes[:max(total_codes, 1)].copy_(dev_codes[:max(total_codes, 1)], non_blocking=True)
            ev = torch.cuda.Event()
            ev.record(st)
            sl["ev"] = ev
        hc = host_codes.numpy()[:total_codes]
        t_out = time.perf_counter()
        st_["slot_wait_s"] += t_slot - t_in; st_["pack_s"] += t_pack - t_slot; st_["enqueue_s"] += t_out - t_pack; st_["super_batches"] += 1

        def wait(_e0=e0, _e1=e1):
            ev.synchronize()
            st_["gpu_ms"] += _e0.elapsed_time(_e1)
        wait.release = sl["free"].set                           # the consumer calls it once it no longer reads `hc`
        return hc, slices, wait

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


def window_table(lengths: Sequence[int], chunk: int, W: int, fpc: int, src_base: Sequence[int], dst_base: Sequence[int]):
    """Every window of every (file, channel) row of a super-batch.  Row r (file-major, channel-minor) holds lengths[r]
    samples at element offset src_base[r]; its codes start at dst_base[r].  Chunk i of a row is encoded from the window of
    T_i = min((i + 1) * chunk, W) samples ending at (i + 1) * chunk (the rolling context of audio_tokenizer.py:72-74) and keeps its
    last fpc codes.  -> (T [n], src_off [n], dst_off [n]) int64, sorted by T descending (full windows first), stable."""
    Ts, srcs, dsts = [], [], []
    for r, (sb, db) in enumerate(zip(src_base, dst_base)):
        n_chunks = lengths[r] // chunk
        if n_chunks == 0:
            continue
        end = (np.arange(n_chunks, dtype=np.int64) + 1) * chunk
        T = np.minimum(end, W)
        Ts.append(T)
        srcs.append(sb + end - T)
        dsts.append(db + np.arange(n_chunks, dtype=np.int64) * fpc)
    if not Ts:
        z = np.zeros(0, np.int64)
        return z, z, z
    T, src, dst = np.concatenate(Ts), np.concatenate(srcs), np.concatenate(dsts)
    order = np.argsort(-T, kind="stable")
    return T[order], src[order], dst[order]


def encode_files(files: Sequence[str], encoder, args, rank: int = 0) -> Tuple[float, int]:
    """Encode `files` with `encoder` (anything with .cfg and .encode(audio, chunk, ctx, batch)); returns
    (audio seconds, codes written)."""
    if hasattr(encoder, "encode_many") and not getattr(args, "one_file_at_a_time", False):
        return encode_files_pipelined(files, encoder, args, rank)
    cfg = encoder.cfg
    sr = cfg.sample_rate
    chunk = int(args.chunk_size_secs * sr)
    ctx = int(args.context_secs * sr)
    out_root = _out_root(args, cfg, rank)
    total_secs, total_codes = 0.0, 0
    for path in files:
        audio = _prepare(path, sr, args.stereo)
        codes = encoder.encode(audio, chunk, ctx, args.batch_size)
        rel = os.path.splitext(os.path.relpath(path, args.audio_path))[0]
        for c in range(codes.shape[0]):
            dst = os.path.join(out_root, f"{rel}_c{c}.npy")
            os.makedirs(os.path.dirname(dst), exist_ok=True)
            np.save(dst, codes[c][None, :])  # (num_codebooks, T)
        total_secs += audio.shape[-1] / sr
        total_codes += int(codes.size)
    return total_secs, total_codes


def _prepare(path: str, sr: int, stereo: bool) -> np.ndarray:
    fsr, audio = read_audio(path)
    if fsr != sr:
        from .audio_tokenizer import _resample
        audio = _resample(audio, fsr, sr)
    if not stereo and audio.shape[0] > 1:
        audio = audio.mean(axis=0, keepdims=True)
    return np.ascontiguousarray(audio, dtype=np.float32)


def _super_batches(files: Sequence[str], sr: int, stereo: bool, budget_samples: int, readers: int) -> Iterator[List[Tuple[str, np.ndarray]]]:
    """Reader stage: files are decoded by a small thread pool (disk and int16 -> float32 conversion release the GIL) a bounded
    distance ahead of the consumer and handed over in groups of about budget_samples samples (all channels counted)."""
    with ThreadPoolExecutor(max_workers=readers) as pool:
        pending: "queue.Queue" = queue.Queue()
        it = iter(files)
        inflight = 0

        def submit_more():
            nonlocal inflight
            while inflight < 4 * readers:
                p = next(it, None)
                if p is None:
                    return
                pending.put((p, pool.submit(_prepare, p, sr, stereo)))
                inflight += 1
        submit_more()
        group, size, n_groups = [], 0, 0
        while inflight:
            p, fut = pending.get()
            audio = fut.result()
            inflight -= 1
            submit_more()
            # ramp-up: the first super-batches are small (1/8, 1/2 of the budget), so the GPU has work a few milliseconds after the
            # start instead of after a whole budget has been read and packed; from the third on nothing is exposed any more
            budget = budget_samples >> 3 if n_groups == 0 else budget_samples >> 1 if n_groups == 1 else budget_samples
            if group and size + audio.size > budget:
                yield group
                group, size, n_groups = [], 0, n_groups + 1
            group.append((p, audio))
            size += audio.size
        if group:
            yield group


def encode_files_pipelined(files: Sequence[str], encoder, args, rank: int = 0) -> Tuple[float, int]:
    """The same output tree as the one-file-at-a-time loop below, with windows batched across files and the three stages
    (read, encode, write) overlapped.  Per-stage wall times are left in `encoder.pipeline_times` (what the main thread and the
    writer spent where): the steady state is GPU-bound when read_wait_s + pack + enqueue < the GPU's busy time."""
    cfg = encoder.cfg
    sr = cfg.sample_rate
    chunk = int(args.chunk_size_secs * sr)
    ctx = int(args.context_secs * sr)
    out_root = _out_root(args, cfg, rank)
    done: "queue.Queue" = queue.Queue(maxsize=1)     # the ring of the encoder bounds what is in flight
    totals = [0.0, 0]
    errors: List[BaseException] = []
    times = dict(read_wait_s=0.0, encode_many_s=0.0, queue_wait_s=0.0, writer_wait_s=0.0, writer_save_s=0.0, first_batch_ready_s=None)
    made_dirs = set()

    def writer():
        while True:
            item = done.get()
            if item is None:
                return
            try:
                names, wait, host_codes, slices = item
                t0 = time.perf_counter()
                wait()                                        # the D2H copy of this super-batch has landed
                t1 = time.perf_counter()
                for (rel, c), (a, b) in zip(names, slices):
                    dst = os.path.join(out_root, f"{rel}_c{c}.npy")
                    d = os.path.dirname(dst)
                    if d not in made_dirs:
                        os.makedirs(d, exist_ok=True)
                        made_dirs.add(d)
                    np.save(dst, host_codes[a:b][None, :])    # (num_codebooks, T)
                times["writer_wait_s"] += t1 - t0
                times["writer_save_s"] += time.perf_counter() - t1
            except BaseException as e:                        # surfaced by the main thread after the join
                errors.append(e)
            finally:
                rel_fn = getattr(item[1], "release", None) if item is not None else None
                if rel_fn is not None:
                    rel_fn()
    wt = threading.Thread(target=writer, daemon=True)
    wt.start()
    t_start = time.perf_counter()
    try:
        gen = _super_batches(files, sr, args.stereo, getattr(args, "super_batch_samples", DEFAULT_SUPER_BATCH), getattr(args, "reader_threads", 4))
        while True:
            t0 = time.perf_counter()
            group = next(gen, None)
            t1 = time.perf_counter()
            times["read_wait_s"] += t1 - t0
            if group is None:
                break
            if times["first_batch_ready_s"] is None:
                times["first_batch_ready_s"] = t1 - t_start
            audios = [a for _, a in group]
            host_codes, slices, wait = encoder.encode_many(audios, chunk, ctx, args.batch_size)
            t2 = time.perf_counter()
            times["encode_many_s"] += t2 - t1
            names = [(os.path.splitext(os.path.relpath(p, args.audio_path))[0], c) for p, a in group for c in range(a.shape[0])]
            if len(names) != len(slices):
                raise RuntimeError(f"encode_many returned {len(slices)} code rows for {len(names)} (file, channel) rows")
            done.put((names, wait, host_codes, slices))
            times["queue_wait_s"] += time.perf_counter() - t2
            totals[0] += sum(a.shape[-1] for a in audios) / sr
            totals[1] += int(sum(b - a for a, b in slices))
    finally:
        done.put(None)
        wt.join()
    if errors:
        raise errors[0]
    times["total_s"] = time.perf_counter() - t_start
    if hasattr(encoder, "stage_times"):
        times.update({f"encode_many_{k}": v for k, v in encoder.stage_times.items()})
    encoder.pipeline_times = times
    return totals[0], totals[1]


def _out_root(args, cfg, rank: int) -> str:
    sub = "stereo" if args.stereo else "mono"
    out_root = os.path.join(args.codes_path, args.codec_model, f"{args.chunk_size_secs}s_{args.context_secs}s", sub)
    os.makedirs(out_root, exist_ok=True)
    # consumers call get_codec_info(<codes_path>/<model>/<chunk>s_<ctx>s/{mono,stereo}) (tools/total_duration_codes.py:6-7)
    info = os.path.join(out_root, "codec_info.json")
    if rank == 0 and not os.path.exists(info):
        with open(info, "w") as f:
            json.dump({"num_codebooks": 1, "codebook_size": cfg.codebook_size, "framerate": cfg.framerate}, f)
    return out_root


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
    ap.add_argument("--one_file_at_a_time", action="store_true", help="the simple loop (a pass never spans files, no overlap of read / encode / write)")
    ap.add_argument("--super_batch_samples", type=int, default=DEFAULT_SUPER_BATCH, help="PCM samples (all channels) uploaded and encoded per super-batch")
    ap.add_argument("--reader_threads", type=int, default=4)
    ap.add_argument("--receptive_field_trim", action="store_true",
                    help="encode only what each chunk's kept frames can see instead of the whole context window: identical "
                         "codes for this build's conv codec, ~8x faster (rca_codec_set_window_trim)")
    return ap


def main(argv=None, encoder=None, backend: Optional[str] = None) -> dict:
    args = build_parser().parse_args(argv)
    rank, world, local = env_rank_world()
    if encoder is None:
        # rehearsal hooks (several ranks on ONE card): RCA_DEVICE pins every rank to that device, RCA_DIST_BACKEND picks the backend
        # of the start / stop barrier (RCCL refuses two ranks on one GPU; the data path has no collective either way)
        if os.environ.get("RCA_DEVICE") is not None:
            local = int(os.environ["RCA_DEVICE"])
        encoder = HipWindowEncoder(args.codec_model, local)
        encoder.model.hip.set_window_trim(args.receptive_field_trim)
        backend = backend or os.environ.get("RCA_DIST_BACKEND", "nccl")
    cp = ControlPlane(prefer=backend or "gloo", device_index=local)     # falls back to gloo by itself when RCCL cannot come up
    files = list_audio_files(args.audio_path, args.audio_filter)
    shards = shard_by_duration([probe_duration(f) for f in files], world)
    mine = [files[i] for i in shards[rank]]
    dev = getattr(encoder, "device", None)
    cp.barrier()
    t0 = time.perf_counter()
    secs, ncodes = encode_files(mine, encoder, args, rank)
    if dev is not None and hasattr(encoder, "torch"):
        encoder.torch.cuda.synchronize(dev)
    cp.barrier()
    elapsed = cp.max(time.perf_counter() - t0)
    total_secs = cp.sum(secs)
    total_codes = cp.sum(float(ncodes))
    summary = dict(files=len(files), world_size=world, audio_hours=total_secs / 3600.0, codes=int(total_codes), elapsed_s=elapsed,
                   audio_hours_per_hour=(total_secs / elapsed) if elapsed > 0 else None, control_plane=cp.backend,
                   stages=getattr(encoder, "pipeline_times", None))
    if rank == 0:
        print(json.dumps(summary))
    return summary


if __name__ == "__main__":
    main()
