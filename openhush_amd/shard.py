"""Data-parallel sharding of independent 30 s windows over ranks.

The path has no exchange step: windows are independent, so every rank transcribes its own windows and
the only communication is collecting token ids on rank 0.  Used by bench.py (RCCL on GPUs) and tested
with the gloo backend on CPU (tests/test_shard_gloo.py).
"""
from __future__ import annotations

from typing import List, Optional, Sequence

import torch


def assign_windows(n_windows: int, world: int, rank: int) -> List[int]:
    """Static round-robin: window i -> rank i % world (keeps every rank's batches equally full)."""
    return list(range(rank, n_windows, world))


def pack_tokens(token_lists: Sequence[Sequence[int]], width: int) -> torch.Tensor:
    """[n][width + 1] int32: column 0 = length, then the tokens (zero padded)."""
    out = torch.zeros(len(token_lists), width + 1, dtype=torch.int32)
    for i, t in enumerate(token_lists):
        n = min(len(t), width)
        out[i, 0] = n
        if n:
            out[i, 1:1 + n] = torch.tensor(list(t[:n]), dtype=torch.int32)
    return out


def unpack_tokens(packed: torch.Tensor) -> List[List[int]]:
    p = packed.cpu()
    return [[int(x) for x in row[1:1 + int(row[0])]] for row in p]


def gather_tokens(packed: torch.Tensor, dist, world: int, rank: int, device: Optional[torch.device] = None, force: bool = False):
    """All ranks send their packed tokens to rank 0; returns the list of per-rank tensors there.
    force=True runs the collective even for a single rank (rehearsal of the N > 1 path on one GPU)."""
    if world == 1 and not force:
        return [packed]
    t = packed.to(device) if device is not None else packed
    bufs = [torch.zeros_like(t) for _ in range(world)] if rank == 0 else None
    dist.gather(t, bufs, dst=0)
    return bufs


def interleave(per_rank: Sequence[List[List[int]]], n_windows: int, world: int) -> List[List[int]]:
    """Undo assign_windows: per_rank[r][j] is window r + j * world."""
    out: List[List[int]] = [[] for _ in range(n_windows)]
    for r, lst in enumerate(per_rank):
        for j, t in enumerate(lst):
            w = r + j * world
            if w < n_windows:
                out[w] = t
    return out
