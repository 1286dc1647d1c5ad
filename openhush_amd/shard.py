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


def transcribe_sharded(transcribe_windows, pcm, n_windows: int, width: int, dist, world: int, rank: int,
                       device: Optional[torch.device] = None, by_index: bool = False):
    """One long recording over `world` ranks (BASELINE.json config #4: 1 h = 120 x 30 s windows over 8 GPUs).

    pcm: the whole recording (1-D float32, 16 kHz), present on every rank; fixed 30 s cuts, window i -> rank i % world.
    transcribe_windows(list of 1-D sample arrays) -> list of token lists runs this rank's windows through the engine
    (it chooses its own batching).  No collective in the data path; the one gather brings the token ids to rank 0,
    which returns the per-window token lists in recording order (other ranks return None)."""
    chunk = 480000
    mine = assign_windows(n_windows, world, rank)
    # by_index: the runner gets window numbers (recording_window_runner cuts them from the recording it holds)
    toks = transcribe_windows(list(mine) if by_index else [pcm[w * chunk:(w + 1) * chunk] for w in mine]) if mine else []
    if len(toks) != len(mine):
        raise RuntimeError("transcribe_windows must return one token list per window")
    rows = (n_windows + world - 1) // world             # equal shapes for the gather: pad with empty windows
    packed = pack_tokens(list(toks) + [[] for _ in range(rows - len(mine))], width)
    got = gather_tokens(packed, dist, world, rank, device)
    if rank != 0:
        return None
    return interleave([unpack_tokens(g) for g in got], n_windows, world)


def engine_window_runner(ctx, max_batch: int, params=None, mel_mode: Optional[int] = None):
    """transcribe_windows for transcribe_sharded on top of the staged C ABI: batches of up to max_batch windows through
    mel -> encode -> greedy on this rank's device (the fixed-window path of ohw_engine_transcribe)."""
    import numpy as np
    from . import engine as E
    st = E.State(ctx, max_batch)
    p = params or ctx.default_params()
    mode = E.OHW_MEL_ZERO_TAIL if mel_mode is None else mel_mode

    def run(windows):
        out = []
        for i in range(0, len(windows), max_batch):
            group = windows[i:i + max_batch]
            buf = np.zeros((len(group), E.CHUNK_SAMPLES), np.float32)
            ns = []
            for b, w in enumerate(group):
                buf[b, :len(w)] = w
                ns.append(len(w))
            st.mel(buf, ns, mode, want=False)
            st.encode(len(group))
            toks, _ = st.greedy(len(group), p)
            out += toks
        return out

    return run


def recording_window_runner(ctx, max_batch: int, pcm, params=None):
    """transcribe_windows for transcribe_sharded(.., by_index=True): this rank's windows are cut from the spectrogram of the
    WHOLE recording (ohw_recording_set / ohw_mel_seek: one clamp maximum for all windows, real samples across the 30 s
    marks - what whisper.cpp computes for a recording handed over in one call; the engine's OHW_WINDOW_FIXED_RECORDING_MEL).
    Every rank holds the recording and finds the same maximum (a pass of the mel kernel over all frames)."""
    from . import engine as E
    st = E.State(ctx, max_batch)
    p = params or ctx.default_params()
    st.recording_set(pcm)

    def run(window_ids):
        out = []
        for i in range(0, len(window_ids), max_batch):
            group = list(window_ids[i:i + max_batch])
            st.mel_seek([w * E.CHUNK_FRAMES for w in group], want=False)
            st.encode(len(group))
            toks, _ = st.greedy(len(group), p)
            out += toks
        return out

    return run


def load_model_broadcast(model_path: str, dist, world: int, rank: int, device_index: int, dtype: int, via_host: bool = False):
    """Rank 0 reads and repacks the ggml file; its resident weight blob (3.1 GB at large-v3) goes to the other ranks in ONE
    broadcast (RCCL over xGMI with the nccl backend), which import it into a shell context - instead of `world` file reads
    (SURVEY.md 8e).  The hyper-parameters travel first (11 ints).  via_host=True stages through host memory (gloo)."""
    from . import engine as E
    dev = torch.device("cuda", device_index)
    # word 0 = status of rank 0's load (0 ok, else the library's error code), then the 11 hyper-parameters: every rank
    # learns of a failed load in the same broadcast and raises, instead of waiting in the blob broadcast for the timeout
    hp_t = torch.zeros(13, dtype=torch.int32)        # status, 11 hyper-parameters, the dtype rank 0 resolved (OHW_DTYPE_AUTO)
    ctx = None
    err = None
    if rank == 0:
        try:
            ctx = E.Context.from_file(model_path, device_index, dtype)
            hp_t[1:12] = torch.tensor(ctx.hp.as_list(), dtype=torch.int32)
            hp_t[12] = ctx.dtype
        except E.WhisperError as ex:
            err = ex
            hp_t[0] = int(ex.code) if ex.code else -1
    if world == 1:
        if err is not None:
            raise err
        return ctx
    hp_t = hp_t if via_host else hp_t.to(dev)
    dist.broadcast(hp_t, src=0)
    status = int(hp_t[0])
    if status != 0:
        if err is not None:
            raise err
        raise E.LoadFailed(status, f"rank 0 could not load {model_path} (code {status})")
    if rank != 0:
        ctx = E.Context.shell([int(v) for v in hp_t.cpu()[1:12]], device_index, int(hp_t.cpu()[12]))
    n = ctx.blob_size()
    blob = torch.empty(n, dtype=torch.uint8, device=dev)
    if rank == 0:
        ctx.export_blob(blob.data_ptr(), n)
    if via_host:
        host = blob.cpu()
        dist.broadcast(host, src=0)
        if rank != 0:
            blob.copy_(host)
    else:
        dist.broadcast(blob, src=0)
    torch.cuda.synchronize(dev)
    if rank != 0:
        ctx.import_blob(blob.data_ptr(), n)
    del blob
    return ctx

