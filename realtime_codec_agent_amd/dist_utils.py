"""Multi-GPU plumbing.  The hot path shards by file / chunk range with NO data-path collective
(the reference runs one OS process per GPU, encode_audio_gpu_{1..4}.sh and
realtime_agent_v2.py:833-836) -- "replicas only".  torch.distributed (RCCL on GPUs, gloo on CPU) is used
solely for the start/stop barrier and the max-over-ranks time that bench.py and the batch CLI report."""
from __future__ import annotations

import os
from typing import List, Sequence, Tuple


def env_rank_world() -> Tuple[int, int, int]:
    return int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("LOCAL_RANK", "0"))


def init_dist(backend: str, device_index: int | None = None):
    """Returns torch.distributed when WORLD_SIZE > 1 (initialised), else None."""
    rank, world, local = env_rank_world()
    if world <= 1:
        return None
    import torch
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29533")
    if not dist.is_initialized():
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local if device_index is None else device_index))
        else:
            dist.init_process_group(backend)
    return dist


def max_over_ranks(value: float, dist, device=None) -> float:
    if dist is None:
        return value
    import torch
    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def sum_over_ranks(value: float, dist, device=None) -> float:
    if dist is None:
        return value
    import torch
    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return float(t.item())


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
