"""The N > 1 path on CPU: world-size-2 gloo ranks shard windows and gather token ids to rank 0."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from openhush_amd import shard


def _fake_tokens(window: int):
    return [1000 + window, 7 * window % 50257] + [window] * (window % 5)


def _worker(rank, world, port, n_windows, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    mine = shard.assign_windows(n_windows, world, rank)
    # every rank packs the same number of rows (pad with empty windows) so gather shapes agree
    rows = (n_windows + world - 1) // world
    toks = [_fake_tokens(w) for w in mine] + [[] for _ in range(rows - len(mine))]
    packed = shard.pack_tokens(toks, 16)
    got = shard.gather_tokens(packed, dist, world, rank)
    dist.barrier()
    if rank == 0:
        per_rank = [shard.unpack_tokens(g) for g in got]
        q.put(shard.interleave(per_rank, n_windows, world))
    dist.destroy_process_group()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_two_ranks_cover_every_window_once_and_in_order():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    n_windows, world = 7, 2
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_windows, q)) for r in range(world)]
    for p in procs:
        p.start()
    result = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert result == [_fake_tokens(w) for w in range(n_windows)]


def test_assignment_is_a_partition():
    for n, world in ((120, 8), (7, 2), (3, 4), (0, 2)):
        seen = sorted(w for r in range(world) for w in shard.assign_windows(n, world, r))
        assert seen == list(range(n))
        sizes = [len(shard.assign_windows(n, world, r)) for r in range(world)]
        assert max(sizes) - min(sizes) <= 1


def _worker_long(rank, world, port, n_windows, q):
    import numpy as np
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    pcm = np.arange(n_windows * 480000 - 1234, dtype=np.float32)          # the last window is short
    seen = []

    def fake_engine(windows):          # "tokens" that identify the window by its first sample and its length
        seen.extend(len(w) for w in windows)
        return [[int(w[0]) // 480000, len(w) % 50000] for w in windows]

    res = shard.transcribe_sharded(fake_engine, pcm, n_windows, 8, dist, world, rank)
    dist.barrier()
    if rank == 0:
        q.put(res)
    else:
        assert res is None
    dist.destroy_process_group()


def test_one_recording_over_two_ranks_returns_windows_in_order():
    """config #4 in miniature: fixed 30 s cuts of one recording, round-robin over ranks, gathered on rank 0."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    n_windows, world = 5, 2
    port = _free_port()
    procs = [ctx.Process(target=_worker_long, args=(r, world, port, n_windows, q)) for r in range(world)]
    for p in procs:
        p.start()
    result = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert result == [[w, 480000 % 50000] for w in range(4)] + [[4, (480000 - 1234) % 50000]]

