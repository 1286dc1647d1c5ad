"""BASELINE.json config #4 in miniature on ONE GPU: two gloo ranks (two processes sharing cuda:0) split the fixed 30 s
windows of one recording, run them through the HIP engine and gather the token ids on rank 0 - the result equals the
single-process engine's.  (The 8-GPU run is the driver's; RCCL needs one device per rank.)"""
import os
import socket

import numpy as np
import pytest
import torch.multiprocessing as mp

from openhush_amd import synth

pytestmark = pytest.mark.gpu


def _recording(n_windows, tail):
    # the second window 26 dB quieter: its own clamp maximum differs from the recording's
    parts = [synth.synth_audio(50 + w) * (0.05 if w == 1 else 1.0) for w in range(n_windows - 1)] + [synth.synth_audio(50 + n_windows - 1, tail)]
    return np.concatenate(parts).astype(np.float32)


def _worker(rank, world, port, model_path, n_windows, tail, q, recording_mel=False):
    import torch.distributed as dist
    from openhush_amd import engine as E, shard
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    pcm = _recording(n_windows, tail)
    ctx = E.Context.from_file(model_path, 0, E.OHW_DTYPE_BF16)
    run = shard.recording_window_runner(ctx, 2, pcm) if recording_mel else shard.engine_window_runner(ctx, 2)
    res = shard.transcribe_sharded(run, pcm, n_windows, ctx.hp.n_text_ctx, dist, world, rank, by_index=recording_mel)
    dist.barrier()
    if rank == 0:
        q.put(res)
    dist.destroy_process_group()


@pytest.mark.parametrize("recording_mel", [False, True])
def test_two_ranks_on_one_gpu_match_the_single_process_engine(tmp_models, recording_mel):
    """recording_mel: the windows of both ranks are cut from the spectrogram of the whole recording (shard.recording_window_runner)
    and equal the engine's OHW_WINDOW_FIXED_RECORDING_MEL mode."""
    from openhush_amd import engine as E
    path = tmp_models("micro")
    n_windows, tail = 5, 200000
    pcm = _recording(n_windows, tail)
    eng = E.WhisperEngine.new(path, "auto", False, True, 0, E.OHW_DTYPE_BF16, 2)
    eng.set_decode_policy(temperature_inc=0.0)      # the sharded runner below is the staged T = 0 path
    if recording_mel:
        eng.set_window_mode(E.OHW_WINDOW_FIXED_RECORDING_MEL)
    eng.transcribe(E.AudioBuffer(pcm, 16000))
    ref, lens = eng.last_tokens(), [q[0] for q in eng.last_quality()]
    eng.close()
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, path, n_windows, tail, q, recording_mel)) for r in range(2)]
    for p in procs:
        p.start()
    got = q.get(timeout=300)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert [len(t) for t in got] == lens
    assert [t for w in got for t in w] == ref


def _worker_bcast(rank, world, port, model_path, q):
    import torch.distributed as dist
    from openhush_amd import engine as E, shard
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    ctx = shard.load_model_broadcast(model_path, dist, world, rank, 0, E.OHW_DTYPE_BF16, via_host=True)
    st = E.State(ctx, 1)
    st.mel(synth.synth_audio(90)[None, :], [480000], E.OHW_MEL_ZERO_TAIL, want=False)
    st.encode(1)
    p = ctx.default_params()
    p.n_max = 16
    toks, _ = st.greedy(1, p)
    q.put((rank, ctx.weight_digests(), toks[0]))
    dist.barrier()
    dist.destroy_process_group()


def test_weights_broadcast_from_rank0_equal_the_file(tmp_models):
    """Only rank 0 reads the model file; rank 1 imports the broadcast blob into a shell context: same resident weights
    (every buffer's digest), same tokens.  (gloo through host memory here; the nccl path hands RCCL the device blob.)"""
    path = tmp_models("micro")
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker_bcast, args=(r, 2, port, path, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = dict()
    for _ in range(2):
        r, dig, toks = q.get(timeout=300)
        got[r] = (dig, toks)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert got[0][0] == got[1][0] and len(got[0][0]) > 20
    assert got[0][1] == got[1][1] and len(got[0][1]) > 0

