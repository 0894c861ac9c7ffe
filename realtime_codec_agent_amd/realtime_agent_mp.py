"""RealtimeAgentMultiprocessing -- one duplex session in its own process, pinned to one GPU.

This is the reference's only form of parallelism (realtime_codec_agent/realtime_agent_v2.py:791-928; self-play and the
8-session configuration run N of these, one per GPU, inference_client_self_play.py:148-159): no collective, no shared
device state.  Only the PUBLIC surface is kept -- wait_until_running, is_running, queue_input, next_output, reset,
set_config_and_reset, get_info, and the item formats of SURVEY.md 8b-5 (input: chunk or (chunk, ids | None); output:
(chunk | (chunk, ids), realtime_factor | None)).  The machinery is this build's own:

  * ONE ordered command channel parent -> worker (audio frames and control requests travel in the order they were
    issued) and one result channel back; the worker BLOCKS on the channel when idle -- no polled flags, no sleeps
    (the reference marks its own polls "#TODO: use an Event", :825,907,914).
  * reset() bumps a shared epoch counter before it enqueues the request: frames queued before the reset are skipped
    by the worker the moment it sees them and results of the old epoch never reach the caller, which is what the
    reference's queue flush (:866-870) is for.
  * failures are surfaced: a worker that dies while loading its models (no GPU, missing library, bad path) makes
    wait_until_running() raise with the worker's traceback instead of spinning forever; an exception inside
    process_audio comes back through next_output() as RealtimeAgentWorkerError (the reference prints and carries on,
    :891-894).
  * the worker entry point is a module-level function: nothing but the constructor arguments is pickled.
  * the worker sets HIP_VISIBLE_DEVICES (and CUDA_VISIBLE_DEVICES, which torch-ROCm honours too) before anything touches
    the GPU (reference :832-836 sets CUDA_VISIBLE_DEVICES).
"""
from __future__ import annotations

import multiprocessing as mp
import os
import queue as _queue
import time
import traceback
from dataclasses import dataclass
from typing import Any, Callable, Dict, List, Optional

import numpy as np

from .realtime_agent_config import RealtimeAgentConfig


class RealtimeAgentWorkerError(RuntimeError):
    """Raised in the parent when the worker process failed; carries the worker-side traceback text."""


@dataclass
class RealtimeAgentMultiprocessingInfo:
    config: RealtimeAgentConfig
    sampling_rate: int
    chunk_size_samples: int
    total_secs: float
    transcript: str
    sequence: str
    audio_history: np.ndarray
    external_llm_messages: Optional[List[Dict[str, str]]]


# message tags (parent -> worker)
_AUDIO, _RESET, _CONFIG, _INFO, _STOP = "audio", "reset", "config", "info", "stop"
# (worker -> parent)
_READY, _FAILED, _RESULT, _ERROR, _REPLY = "ready", "failed", "result", "error", "reply"


def _snapshot(agent) -> RealtimeAgentMultiprocessingInfo:
    return RealtimeAgentMultiprocessingInfo(
        config=agent.config, sampling_rate=agent.resources.audio_tokenizer.sampling_rate, chunk_size_samples=agent.chunk_size_samples,
        total_secs=agent.total_secs, transcript=agent.format_transcript(), sequence=agent.get_sequence_str(),
        audio_history=agent.get_audio_history(), external_llm_messages=agent.get_external_llm_messages())


def _worker_main(commands, results, epoch, config, self_play_mode, gpu_id, resources_factory, resources_kwargs):
    """Worker process: build the session, then serve the command channel until told to stop."""
    try:
        if gpu_id is not None:
            os.environ["HIP_VISIBLE_DEVICES"] = str(gpu_id)
            os.environ["CUDA_VISIBLE_DEVICES"] = str(gpu_id)
        from .realtime_agent_v2 import RealtimeAgent
        if resources_factory is not None:
            resources = resources_factory(**resources_kwargs)
        else:
            from .realtime_agent_resources import RealtimeAgentResources
            resources = RealtimeAgentResources(**resources_kwargs)
        agent = RealtimeAgent(resources=resources, config=config, self_play_mode=self_play_mode)
    except BaseException:
        results.put((_FAILED, traceback.format_exc()))
        return
    results.put((_READY, os.getpid()))
    parent = os.getppid()
    while True:
        try:
            msg = commands.get(timeout=2.0)       # blocks while the session is idle ...
        except _queue.Empty:
            if os.getppid() != parent:            # ... but an orphan (parent killed outright) must not keep its GPU
                os._exit(3)
            continue
        tag = msg[0]
        if tag == _STOP:
            return
        try:
            if tag == _AUDIO:
                _, msg_epoch, seq, chunk, ids = msg
                if msg_epoch != epoch.value:      # queued before a reset: dropped unprocessed
                    continue
                out = agent.process_audio(chunk, ids)
                vals = agent.profilers.total_profiler.realtime_factor_values
                results.put((_RESULT, msg_epoch, seq, out, vals[-1] if vals else None))
            elif tag == _RESET:
                agent.reset()
                results.put((_REPLY, msg[1], None))
            elif tag == _CONFIG:
                agent.set_config(msg[2])
                agent.reset()
                results.put((_REPLY, msg[1], None))
            elif tag == _INFO:
                results.put((_REPLY, msg[1], _snapshot(agent)))
        except Exception:
            err = traceback.format_exc()
            if tag == _AUDIO:
                results.put((_ERROR, msg[1], msg[2], err))
            else:
                results.put((_REPLY, msg[1], RealtimeAgentWorkerError(err)))


class RealtimeAgentMultiprocessing:
    def __init__(self, wait_until_running: bool = True, config: RealtimeAgentConfig = None, self_play_mode: bool = False,
                 gpu_id: Optional[int] = None, idle_tol_secs: float = 1.0, resources_factory: Optional[Callable[..., Any]] = None,
                 start_timeout_secs: float = 600.0, **resources_kwargs):
        """Same arguments as the reference (:795-803); `idle_tol_secs` is accepted and unused (the worker blocks instead of polling).
        resources_factory: optional picklable callable building the resources object inside the worker (tests use fakes);
        default RealtimeAgentResources(**resources_kwargs)."""
        ctx = mp.get_context("spawn")           # a forked child of a process that has touched the GPU is not usable
        self._commands = ctx.Queue()
        self._results = ctx.Queue()
        self._epoch = ctx.Value("q", 0)
        self._start_timeout = start_timeout_secs
        self._running = False
        self._closed = False
        self._next_req = 0
        self._next_seq = 0
        self._pending: List[tuple] = []          # results read while waiting for a control reply
        self._process = ctx.Process(target=_worker_main, daemon=True,
                                    args=(self._commands, self._results, self._epoch, config, self_play_mode, gpu_id, resources_factory,
                                          resources_kwargs))
        self._process.start()
        if wait_until_running:
            self.wait_until_running()

    # ------------------------------------------------------------------ lifecycle
    def _recv(self, timeout: Optional[float]):
        """One message from the worker; raises if the worker is gone."""
        deadline = None if timeout is None else time.monotonic() + timeout
        while True:
            try:
                if timeout is not None and timeout <= 0.0:
                    return self._results.get_nowait()          # a poll (is_running): never block
                return self._results.get(timeout=0.2)
            except _queue.Empty:
                if not self._process.is_alive():
                    try:                          # a last message may have been flushed while it exited
                        return self._results.get(timeout=0.2)
                    except _queue.Empty:
                        raise RealtimeAgentWorkerError(f"agent worker exited (exit code {self._process.exitcode}) without answering") from None
                if deadline is not None and time.monotonic() > deadline:
                    raise TimeoutError("agent worker did not answer in time")

    def wait_until_running(self, timeout: Optional[float] = None):
        if self._running:
            return
        msg = self._recv(self._start_timeout if timeout is None else timeout)
        if msg[0] == _FAILED:
            self._process.join(5)
            raise RealtimeAgentWorkerError("agent worker failed to start:\n" + msg[1])
        assert msg[0] == _READY, msg[0]
        self.worker_pid = msg[1]
        self._running = True

    def is_running(self) -> bool:
        if not self._running and self._process.is_alive():
            try:
                self.wait_until_running(timeout=0.0)
            except TimeoutError:
                pass
        return self._running and self._process.is_alive()

    def close(self, timeout: float = 10.0):
        if self._closed:
            return
        self._closed = True
        try:
            if self._process.is_alive():          # never write to a channel nobody reads any more
                self._commands.put((_STOP,))
                self._process.join(timeout)
        finally:
            if self._process.is_alive():
                self._process.terminate()
                self._process.join(5)
            for q in (self._commands, self._results):   # release the pipes without waiting on a reader that is gone
                q.cancel_join_thread()
                q.close()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def __del__(self):
        try:
            self.close(timeout=2.0)
        except Exception:
            pass

    # ------------------------------------------------------------------ control requests
    def _request(self, tag: str, *payload):
        self.wait_until_running()
        req = self._next_req
        self._next_req += 1
        self._commands.put((tag, req) + payload)
        while True:
            msg = self._recv(None)
            if msg[0] == _REPLY and msg[1] == req:
                if isinstance(msg[2], Exception):
                    raise msg[2]
                return msg[2]
            self._pending.append(msg)             # an audio result that arrived first: kept for next_output

    def reset(self):
        with self._epoch.get_lock():
            self._epoch.value += 1                # frames already queued belong to the old epoch: the worker skips them
        self._pending = []
        self._request(_RESET)

    def set_config_and_reset(self, config: RealtimeAgentConfig):
        with self._epoch.get_lock():
            self._epoch.value += 1
        self._pending = []
        self._request(_CONFIG, config)

    def get_info(self) -> RealtimeAgentMultiprocessingInfo:
        return self._request(_INFO)

    # ------------------------------------------------------------------ the audio path
    def queue_input(self, input):
        """input: np.ndarray chunk, or (chunk, audio_chunk_input_ids | None) as produced by a self-play partner."""
        chunk, ids = (input, None) if isinstance(input, np.ndarray) else input
        self._commands.put((_AUDIO, self._epoch.value, self._next_seq, chunk, ids))
        self._next_seq += 1

    def next_output(self, block: bool = False):
        """(out_chunk | (out_chunk, out_ids), realtime_factor | None), or None when nothing is ready and block is False.
        Raises RealtimeAgentWorkerError if the frame failed inside the worker."""
        while True:
            if self._pending:
                msg = self._pending.pop(0)
            else:
                if not block:
                    try:
                        msg = self._results.get_nowait()
                    except _queue.Empty:
                        if not self._process.is_alive():
                            raise RealtimeAgentWorkerError(f"agent worker exited (exit code {self._process.exitcode})") from None
                        return None
                else:
                    msg = self._recv(None)
            if msg[0] == _READY:
                self._running = True
                continue
            if msg[0] == _FAILED:
                raise RealtimeAgentWorkerError("agent worker failed to start:\n" + msg[1])
            if msg[0] in (_RESULT, _ERROR) and msg[1] != self._epoch.value:
                continue                           # produced before the last reset
            if msg[0] == _ERROR:
                raise RealtimeAgentWorkerError(f"process_audio failed on frame {msg[2]}:\n{msg[3]}")
            if msg[0] == _RESULT:
                return msg[3], msg[4]
