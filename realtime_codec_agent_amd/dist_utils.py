"""Multi-GPU plumbing.  The hot path shards by file / chunk range with NO data-path collective
(the reference runs one OS process per GPU, encode_audio_gpu_{1..4}.sh and
realtime_agent_v2.py:833-836) -- "replicas only".  torch.distributed is used solely for the start/stop barrier and the
max-over-ranks time that bench.py and the batch CLI report (ControlPlane: RCCL when every rank can bring it up, gloo otherwise)."""
from __future__ import annotations

import os
from typing import List, Sequence, Tuple


def env_rank_world() -> Tuple[int, int, int]:
    return int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("LOCAL_RANK", "0"))


class ControlPlane:
    """The start / stop protocol of the multi-process runs: barrier, max / sum of one float over the ranks, gather of one small
    object.  The DATA path has no collective, so nothing here may be able to take a run down:

      * the default process group is ALWAYS gloo over CPU tensors -- it cannot fail for GPU-side reasons (IPC handles, xGMI
        topology, a second rank on one card);
      * with prefer="nccl" (= RCCL on ROCm) an RCCL group is created on top and one all-reduce is pushed through it; every rank
        reports over gloo whether that worked, and RCCL carries the protocol only if it worked EVERYWHERE.  Otherwise the
        protocol stays on gloo -- same process, no re-exec -- and `fallback_reason` says why.

    `backend` names what carries the protocol ("none" for a single process); both runs log it."""

    def __init__(self, prefer: str = "nccl", device_index: int | None = None, timeout_s: float = 180.0, log=None):
        import sys
        self.rank, self.world, local = env_rank_world()
        self.backend, self.fallback_reason, self._dist, self._group, self._dev = "none", None, None, None, None
        self._log = log or (lambda m: print(m, file=sys.stderr, flush=True))
        if self.world <= 1:
            return
        import datetime
        import torch
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        if not dist.is_initialized():
            dist.init_process_group("gloo", timeout=datetime.timedelta(seconds=timeout_s))
        self._dist, self.backend = dist, "gloo"
        if prefer == "nccl":
            ok, why, group = 1, "", None
            try:
                if not torch.cuda.is_available():
                    raise RuntimeError("no GPU visible to this rank")
                idx = local if device_index is None else device_index
                dev = torch.device("cuda", idx)
                group = dist.new_group(backend="nccl", timeout=datetime.timedelta(seconds=timeout_s), device_id=dev)
                t = torch.ones(1, device=dev)
                dist.all_reduce(t, group=group)
                torch.cuda.synchronize(dev)
                if int(t.item()) != self.world:
                    raise RuntimeError(f"RCCL all-reduce probe returned {t.item()} for {self.world} ranks")
            except Exception as e:   # noqa: BLE001 -- whatever RCCL / the driver raised: the protocol does not need it
                ok, why = 0, f"{type(e).__name__}: {str(e).splitlines()[0] if str(e) else ''}"[:300]
            flags = [None] * self.world
            dist.all_gather_object(flags, (ok, why))            # over gloo
            if all(f[0] for f in flags):
                self._group, self._dev, self.backend = group, dev, "nccl"
            else:
                self.fallback_reason = next(f"rank {r}: {f[1]}" for r, f in enumerate(flags) if not f[0])
        if self.rank == 0:
            self._log(f"[control plane] {self.world} ranks, barrier / max-over-ranks carried by {self.backend}"
                      + (f" (RCCL not used: {self.fallback_reason})" if self.fallback_reason else ""))

    @property
    def active(self) -> bool:
        return self._dist is not None

    def describe(self) -> dict:
        return {"backend": self.backend, "fallback_reason": self.fallback_reason}

    def barrier(self) -> None:
        if self._dist is None:
            return
        if self._group is not None:
            self._dist.barrier(group=self._group)
        else:
            self._dist.barrier()

    def _reduce(self, value: float, op) -> float:
        if self._dist is None:
            return value
        import torch
        t = torch.tensor([value], dtype=torch.float64, device=self._dev)
        self._dist.all_reduce(t, op=op, group=self._group)
        return float(t.item())

    def max(self, value: float) -> float:
        return value if self._dist is None else self._reduce(value, self._dist.ReduceOp.MAX)

    def sum(self, value: float) -> float:
        return value if self._dist is None else self._reduce(value, self._dist.ReduceOp.SUM)

    def all_gather_object(self, obj) -> list:
        if self._dist is None:
            return [obj]
        out = [None] * self.world
        self._dist.all_gather_object(out, obj)      # pickles travel over gloo either way
        return out

    def close(self) -> None:
        if self._dist is not None and self._dist.is_initialized():
            self._dist.destroy_process_group()
        self._dist = None


def shard_by_duration(durations: Sequence[float], world: int) -> List[List[int]]:
    """Longest-processing-time-first partition of item indices over `world` ranks: deterministic, disjoint,
    complete.  The reference partitions by corpus name by hand, one script per GPU."""
    order = sorted(range(len(durations)), key=lambda i: (-durations[i], i))
    loads = [0.0] * world
    shards: List[List[int]] = [[] for _ in range(world)]
    for i in order:
        r = min(range(world), key=lambda k: (loads[k], k))
        shards[r].append(i)
        loads[r] += durations[i]
    for s in shards:
        s.sort()
    return shards


def shard_chunk_range(n_chunks: int, world: int, rank: int) -> Tuple[int, int]:
    """Contiguous chunk range of one long signal for a rank (left context is read from the signal itself)."""
    base, rem = divmod(n_chunks, world)
    begin = rank * base + min(rank, rem)
    return begin, begin + base + (1 if rank < rem else 0)
