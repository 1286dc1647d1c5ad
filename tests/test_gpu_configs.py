"""BASELINE.json configs #2 and #4 at their own sizes, and the engine's threading contract.

  #2  small dims, greedy, batch 1, one 30 s window, bf16 - every stage against the oracle (seconds on the host cores)
  #4  large-v3 dims, one 1 h recording = 120 x 30 s windows through ohw_engine_transcribe(max_batch = 32) on ONE GPU:
      the oracle is far too slow for 120 windows, so the checks are size-independent properties (two batches in flight ==
      one batch after the other; sampled windows == the same window run alone through the staged API) plus the
      two-rank sharded run (two processes sharing the GPU, gloo) against the single-process engine.
Tolerances as in test_gpu_parity: bf16 activations 6e-2 (x2 after the last block), logits 0.25 abs on sigma ~ 4.
"""
import os
import socket
import threading

import numpy as np
import pytest

from openhush_amd import synth

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")


@pytest.fixture(scope="module")
def E():
    if not torch.cuda.is_available():
        pytest.fail("GPU test selected but no GPU is visible")
    from openhush_amd import engine
    engine.lib()
    return engine


@pytest.fixture(scope="module")
def oracle():
    from oracle import oracle as o
    return o


class _Ctx:
    """borrowed (non-owning) view of an engine's context"""
    def __init__(self, E, eng):
        self.h = eng.ctx_h
        self.hp, self.tok = E.HParams(), E.SpecialTokens()
        E.lib().ohw_ctx_info(self.h, self.hp, self.tok)

    def default_params(self):
        import ctypes as C
        from openhush_amd import engine as E
        p = E.SampleParams()
        E.lib().ohw_default_sample_params(self.h, C.byref(p))
        return p


def test_config2_small_batch1_bf16_against_oracle(E, oracle):
    """BASELINE config #2: `small` dimensions (768 / 12 heads / 12 + 12 layers), one 30 s window, batch 1, bf16."""
    hp = synth.PRESETS["small"]
    oracle.set_num_threads(min(16, len(os.sched_getaffinity(0))))
    om = oracle.Model.synth(hp.as_list(), 1234)
    ctx = E.Context.synthetic(hp.as_list(), 1234, 0, E.OHW_DTYPE_BF16)
    st = E.State(ctx, 1)
    pcm = synth.synth_audio(2)[None]
    mel = st.mel(pcm, None, E.OHW_MEL_ZERO_TAIL)
    ref_mel = om.log_mel(pcm[0], 1)
    assert np.abs(mel[0] - ref_mel).max() < 2e-4
    st.encode(1)
    ref_enc = om.encode(ref_mel)
    enc = st.fetch("enc", 1)[0]
    err = np.abs(enc - ref_enc)
    print(f"config#2 encoder: max abs err {err.max():.4f}, mean {err.mean():.5f}")
    # observed on MI355X (round 2): max 0.0273, mean 0.0039 -> tolerances at 2x observed
    assert err.max() < 0.055 and err.mean() < 0.008, (err.max(), err.mean())
    s = oracle.State(om)
    s.set_encoder_output(ref_enc)
    # teacher-forced logits: the prompt in one call, then 6 single-token steps on the oracle's own picks
    prompt = [ctx.tok.sot, ctx.tok.sot + 1, ctx.tok.transcribe]
    ref = s.decode(prompt, 0)
    got = st.decode(np.asarray([prompt], np.int32), [0])[0]
    sig = float(ref.std())
    worst = float(np.abs(got - ref).max())
    tok = int(ref.argmax())
    for i in range(6):
        ref = s.decode([tok], 3 + i)
        got = st.decode(np.asarray([[tok]], np.int32), [3 + i])[0]
        worst = max(worst, float(np.abs(got - ref).max()))
        tok = int(ref.argmax())
    print(f"config#2 logits: worst abs err {worst:.4f} at sigma {sig:.2f}")
    assert worst < 0.23, (worst, sig)          # observed 0.113 - 0.116 abs at sigma 4.01
    # greedy through the device loop: the oracle walks the GPU's path and agrees at every step outside near-ties
    p = ctx.default_params(); p.n_max = 32
    g = st.greedy_ex(1, p)[0]
    op = om.default_params(); op.n_max = 32
    forced = g["tokens"] + ([om.tok_eot] if g["ended_by_eot"] else [])
    r = s.greedy_ex(op, None, forced)
    same = sum(1 for i, t in enumerate(forced) if r["choice"][i] == t)
    for i, t in enumerate(forced):
        assert r["choice"][i] == t or r["margins"][i] < 0.5, (i, t, r["choice"][i], float(r["margins"][i]))
    print(f"config#2 greedy: {same} / {len(forced)} steps identical")
    assert same >= 0.9 * len(forced) and len(forced) >= 16
    om.close()


def test_engine_built_on_one_thread_used_and_freed_on_another(E, tmp_models):
    """SURVEY.md A8 / reference src/queue/worker.rs:22,100-110: the engine is constructed on a runtime thread, moved to
    the worker thread inside WorkerCommand::LoadEngine, used there and dropped there.  Handles carry no thread affinity."""
    path = tmp_models("micro")
    pcm = np.concatenate([synth.synth_audio(61), synth.synth_audio(62, 100000)])
    box = {}

    def builder():
        box["eng"] = E.WhisperEngine.new(path, "auto", False, True, 0, E.OHW_DTYPE_F16, 2)
        box["eng"].set_decode_policy(temperature_inc=0.0)       # T = 0 only (the ladder: test_gpu_policy.py)
        box["first"] = box["eng"].transcribe(E.AudioBuffer(pcm, 16000)).text      # also used once where it was built

    def worker():
        try:
            eng = box["eng"]
            box["second"] = eng.transcribe(E.AudioBuffer(pcm, 16000)).text
            box["tokens"] = eng.last_tokens()
            bm = eng.benchmark(0.2)
            box["bench"] = bm.overhead_secs
            eng.close()                                                             # dropped on the worker thread
            # lazy re-load after an idle unload (reference src/daemon.rs:2242-2283): a new engine on this thread
            e2 = E.WhisperEngine.new(path, "auto", False, True, 0, E.OHW_DTYPE_F16, 2)
            e2.set_decode_policy(temperature_inc=0.0)
            box["third"] = e2.transcribe(E.AudioBuffer(pcm, 16000)).text
            e2.close()
        except Exception as ex:       # surfaced in the main thread below
            box["error"] = ex

    t = threading.Thread(target=builder); t.start(); t.join()
    t = threading.Thread(target=worker); t.start(); t.join()
    assert "error" not in box, box.get("error")
    assert box["first"] == box["second"] == box["third"] and len(box["tokens"]) > 0 and box["bench"] >= 0.001
    # two engines alive at once, each driven by its own thread (the CLI and `record` build their own engines)
    outs = [None, None]

    def both(i):
        e = E.WhisperEngine.new(path, "auto", False, True, 0, E.OHW_DTYPE_F16, 1)
        e.set_decode_policy(temperature_inc=0.0)
        outs[i] = e.transcribe(E.AudioBuffer(pcm, 16000)).text
        e.close()

    th = [threading.Thread(target=both, args=(i,)) for i in range(2)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert outs[0] == outs[1] == box["first"]


@pytest.fixture(scope="module")
def large_v3_file(tmp_models):
    return tmp_models("large-v3")


def _recording(n_windows, tail=None):
    parts = [synth.synth_audio(1000 + w) for w in range(n_windows - 1)]
    parts.append(synth.synth_audio(1000 + n_windows - 1, tail) if tail else synth.synth_audio(1000 + n_windows - 1))
    return np.concatenate(parts)


def test_config4_one_hour_120_windows_on_one_gpu(E, large_v3_file, monkeypatch):
    """BASELINE config #4 at full size on one GPU: 1 h = 120 windows, large-v3 dims, bf16, max_batch 32 (4 batches).
    The default schedule (4 decodes side by side after 4 front ends) and round 1's two-batch pipeline == one batch after
    the other, token for token; four sampled windows == the same window alone through the staged API (other kernel
    variants: single-m-tile GEMMs, split cross-attention)."""
    n_win = 120
    pcm = _recording(n_win, tail=300000)
    out = {}
    import time
    for cus, sched in (("lanes", E.OHW_SCHEDULE_LANES), ("96", E.OHW_SCHEDULE_PIPELINE), ("0", E.OHW_SCHEDULE_SEQUENTIAL)):
        eng = E.WhisperEngine.new(large_v3_file, "auto", False, True, 0, E.OHW_DTYPE_BF16, 32)
        eng.set_decode_policy(temperature_inc=0.0)           # T = 0 only: the fallback ladder is a per-window host path
        eng.set_schedule(sched)                              # LANES: the defaults (4 lanes; 4 batches = one per lane)
        eng.transcribe(E.AudioBuffer(pcm, 16000))                    # warm-up: states, streams, graph captures
        t0 = time.perf_counter()
        res = eng.transcribe(E.AudioBuffer(pcm, 16000))
        print(f"config#4 schedule {cus}: 1 h of audio in {time.perf_counter() - t0:.2f} s")
        q = eng.last_quality()
        out[cus] = (res.text, eng.last_tokens(), [x[0] for x in q])
        if cus == "0":
            assert len(q) == n_win and all(n > 0 for n in out[cus][2])
            # the staged single-window path on the engine's own resident weights
            ctx = _Ctx(E, eng)
            st1 = E.State(ctx, 1)
            p = ctx.default_params()
            starts = np.concatenate([[0], np.cumsum(out[cus][2])])
            agree = total = 0
            for w in (0, 37, 95, 119):
                chunk = pcm[w * 480000:(w + 1) * 480000]
                st1.mel(chunk[None, :], [len(chunk)], E.OHW_MEL_ZERO_TAIL, want=False)
                st1.encode(1)
                toks, _ = st1.greedy(1, p)
                ref = out[cus][1][starts[w]:starts[w + 1]]
                n = min(len(ref), len(toks[0]))
                agree += sum(1 for i in range(n) if ref[i] == toks[0][i])
                total += max(len(ref), len(toks[0]))
            print(f"config#4: sampled windows agree on {agree} / {total} tokens with the single-window path")
            assert agree >= 0.98 * total
            st1.close()
        eng.close()
    assert out["96"] == out["0"] and out["lanes"] == out["0"]
    # merged decode batches: 2 lanes x 2 front-end batches = decodes of 64 and 56 rows (the sequential schedule's last batch has 24)
    eng = E.WhisperEngine.new(large_v3_file, "auto", False, True, 0, E.OHW_DTYPE_BF16, 32)
    eng.set_decode_policy(temperature_inc=0.0)
    eng.set_schedule(E.OHW_SCHEDULE_LANES, 2, 2)
    res = eng.transcribe(E.AudioBuffer(pcm, 16000))
    assert (res.text, eng.last_tokens(), [x[0] for x in eng.last_quality()]) == out["0"]
    eng.close()
    assert len(out["0"][0]) > 0


@pytest.mark.parametrize("n", [40, 96])
def test_large_v3_window_result_does_not_depend_on_its_batch(E, large_v3_file, n):
    """Size-independent property at BASELINE dimensions: n large-v3 windows decoded as one n-row batch on a 64-CU stream (a
    LANES lane: four n-tiles per workgroup in the decoder GEMMs; n = 96 is the bench's own lane shape, three front-end batches
    merged into one decode) against the same windows as batches of at most 32 on the whole chip - logits of the prompt pass and
    of a single-token step, greedy tokens and their log-probabilities, bit for bit (ohw_state_set_batch_invariant; what makes
    every schedule of ohw_engine_transcribe give the same tokens)."""
    ctx = E.Context.from_file(large_v3_file, 0, E.OHW_DTYPE_BF16)
    pcm = np.stack([synth.synth_audio(3000 + w) for w in range(n)])
    tok = ctx.tok
    prompt = np.asarray([tok.sot, tok.sot + 1, tok.transcribe, tok.no_timestamps], np.int32)
    p = ctx.default_params(); p.force_len = 24
    lane = E.Stream(0, 0, 64)
    big = E.State(ctx, n)
    big.set_batch_invariant(True)
    big.set_stream(lane.ptr)
    big.mel(pcm, None, E.OHW_MEL_ZERO_TAIL, want=False); big.encode(n)
    L0 = big.decode(np.tile(prompt, (n, 1)), [0] * n)
    L1 = big.decode(L0.argmax(axis=1).astype(np.int32)[:, None], [4] * n)
    G = big.greedy_ex(n, p)
    small = E.State(ctx, 32)
    small.set_batch_invariant(True)
    for f, m in [(f, min(32, n - f)) for f in range(0, n, 32)]:
        small.mel(pcm[f:f + m], None, E.OHW_MEL_ZERO_TAIL, want=False); small.encode(m)
        l0 = small.decode(np.tile(prompt, (m, 1)), [0] * m)
        l1 = small.decode(l0.argmax(axis=1).astype(np.int32)[:, None], [4] * m)
        g = small.greedy_ex(m, p)
        assert np.array_equal(l0, L0[f:f + m]) and np.array_equal(l1, L1[f:f + m]), f
        for a, b in zip(g, G[f:f + m]):
            assert a["tokens"] == b["tokens"] and np.array_equal(a["logprobs"], b["logprobs"]), f
    big.close(); small.close(); lane.close()


def _worker_large(rank, world, port, model_path, n_windows, q):
    import torch.distributed as dist
    from openhush_amd import engine as E, shard
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    pcm = _recording(n_windows)
    ctx = shard.load_model_broadcast(model_path, dist, world, rank, 0, E.OHW_DTYPE_BF16, via_host=True)
    p = ctx.default_params()
    p.n_max = 48
    run = shard.engine_window_runner(ctx, 4, p)
    res = shard.transcribe_sharded(run, pcm, n_windows, ctx.hp.n_text_ctx, dist, world, rank)
    dist.barrier()
    if rank == 0:
        q.put(res)
    dist.destroy_process_group()


def test_config4_two_ranks_share_the_gpu_at_large_v3_dims(E, large_v3_file):
    """The sharded path of config #4 at large-v3 dims: two gloo ranks on cuda:0, rank 0 reads the 3.1 GB file and
    broadcasts the resident blob, 8 windows dealt round-robin, tokens gathered on rank 0 == one process, 4 at a time."""
    import torch.multiprocessing as mp
    n_windows = 8
    pcm = _recording(n_windows)
    ctx = E.Context.from_file(large_v3_file, 0, E.OHW_DTYPE_BF16)
    from openhush_amd import shard
    p = ctx.default_params()
    p.n_max = 48
    run = shard.engine_window_runner(ctx, 4, p)
    ref = run([pcm[w * 480000:(w + 1) * 480000] for w in range(n_windows)])
    del run
    ctx.close()
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    mpc = mp.get_context("spawn")
    q = mpc.Queue()
    procs = [mpc.Process(target=_worker_large, args=(r, 2, port, large_v3_file, n_windows, q)) for r in range(2)]
    for pr in procs:
        pr.start()
    got = q.get(timeout=600)
    for pr in procs:
        pr.join(timeout=120)
        assert pr.exitcode == 0
    assert got == ref and all(len(t) == 48 for t in got)


# ---- BASELINE config #5 at its own size: large-v3 dims, one 5 s chunk, beam = 5, the hipGraph-captured beam step -----------
# logit biases picked on the oracle (CPU, round 3): with (4, 9) the five beams finish after 3 and 6 tokens and the winner is a
# 6-token sequence (finished candidates of two lengths ranked by log-probability per token); with (5, 8) two beams finish
# early and the others run to n_max = 16 (the live beams join the finished pool)
CFG5_BIAS, CFG5_BIAS_LONG = (4.0, 9.0), (5.0, 8.0)


def _cfg5_bias(n_vocab, tok_beg, tok_eot, ts_eot=CFG5_BIAS):
    b = np.zeros(n_vocab, np.float32)
    b[tok_beg:] = ts_eot[0]
    b[tok_eot] = ts_eot[1]
    return b


def test_config5_large_v3_beam5_chunk_against_oracle_and_its_properties(E, oracle, monkeypatch):
    """large-v3 dimensions (procedural weights, bf16), one 5 s chunk zero-padded to a window, beam = 5, n_max 16, a logit bias
    that lets timestamps and end-of-text occur.  (a) against oracle.beam_search: the oracle's exact winner, or a sequence
    that scores as well under the ORACLE's own scoring (beam search ranks 30 candidates by cumulative log-probability: a bf16
    near-tie may legitimately pick another one; tests/test_gpu_beam.py holds the same rule at micro dims).  (b) properties
    at this size: two consecutive calls identical (graph replay); graphs on == graphs off (OHW_GRAPHS=0: every iteration
    launched kernel by kernel); a window's beams do not depend on another window in the batch."""
    hp = synth.PRESETS["large-v3"]
    oracle.set_num_threads(min(16, len(os.sched_getaffinity(0))))
    om = oracle.Model.synth(hp.as_list(), 1234)
    ctx = E.Context.synthetic(hp.as_list(), 1234, 0, E.OHW_DTYPE_BF16)
    K, n_max = 5, 16
    bias = _cfg5_bias(om.n_vocab, om.tok_beg, om.tok_eot)
    chunk = synth.synth_audio(21)[:80000]                      # 5 s
    other = synth.synth_audio(22)[:80000]
    p = ctx.default_params(); p.n_max = n_max
    st = E.State(ctx, 2 * K)
    st.set_logit_bias(bias)
    # (b3) is a bit-for-bit claim: it holds with ohw_state_set_batch_invariant (kernel variants picked from the number of new
    # tokens alone); without the switch few rows cut a (row, head)'s keys over up to 8 workgroups - another summation order,
    # observed here as the same tokens with sum_logprob -6.6440 against -6.6523
    st.set_batch_invariant(True)
    mel = st.mel(np.stack([chunk, other]), [80000, 80000], E.OHW_MEL_ZERO_TAIL)
    st.encode(2)
    both = st.beam_search(2, K, p)
    caps = st.counter("beam_captures")
    assert st.beam_search(2, K, p) == both and st.counter("beam_captures") == caps == 1          # (b1) replay, no re-capture
    # (b3) window 0 alone, in a state of its own
    st1 = E.State(ctx, K)
    st1.set_logit_bias(bias)
    st1.set_batch_invariant(True)
    st1.mel(chunk[None, :], [80000], E.OHW_MEL_ZERO_TAIL, want=False)
    st1.encode(1)
    alone = st1.beam_search(1, K, p)[0]
    assert alone == both[0], (alone, both[0])
    # (b2) graphs off: the state reads OHW_GRAPHS when it is made
    monkeypatch.setenv("OHW_GRAPHS", "0")
    st0 = E.State(ctx, K)
    monkeypatch.delenv("OHW_GRAPHS")
    st0.set_logit_bias(bias)
    st0.set_batch_invariant(True)
    st0.mel(chunk[None, :], [80000], E.OHW_MEL_ZERO_TAIL, want=False)
    st0.encode(1)
    assert st0.beam_search(1, K, p)[0] == alone and st0.counter("beam_captures") == 0
    # the same three properties where the search runs to n_max with finished and live beams mixed (GPU only: the oracle
    # needs 90 s of 8 cores for this one)
    long_bias = _cfg5_bias(om.n_vocab, om.tok_beg, om.tok_eot, CFG5_BIAS_LONG)
    for s_ in (st, st1, st0):
        s_.set_logit_bias(long_bias)
    both_l = st.beam_search(2, K, p)
    assert st.beam_search(2, K, p) == both_l and st.counter("beam_captures") == 1
    alone_l = st1.beam_search(1, K, p)[0]
    assert alone_l == both_l[0] and st0.beam_search(1, K, p)[0] == alone_l
    assert max(len(x["tokens"]) for x in both_l) > len(alone["tokens"])
    # without the switch (the default for a single streaming window: lower latency) the tokens still agree here
    st1.set_batch_invariant(False)
    st1.set_logit_bias(bias)
    fast = st1.beam_search(1, K, p)[0]
    assert fast["tokens"] == alone["tokens"] and abs(fast["sum_logprob"] - alone["sum_logprob"]) < 0.05
    # (a) the oracle on the same window (its own mel and encoder: fp32)
    ref_mel = om.log_mel(chunk, 1)
    assert np.abs(mel[0] - ref_mel).max() < 2e-4
    enc = om.encode(ref_mel)
    op = om.default_params(); op.n_max = n_max
    ref = oracle.beam_search(om, enc, op, K, bias)
    g = both[0]
    print(f"config#5 beam=5: gpu {g['tokens']} (finished {g['n_finished']}, sum {g['sum_logprob']:.3f}); oracle {ref['tokens']} "
          f"(finished {ref['n_finished']}, sum {ref['sum_logprob']:.3f})")
    assert len(g["tokens"]) > 0 and ref["n_finished"] > 0
    if g["tokens"] == ref["tokens"]:
        assert abs(g["sum_logprob"] - ref["sum_logprob"]) < 0.5 * max(1, len(g["tokens"])) ** 0.5
    else:
        s = oracle.State(om); s.set_encoder_output(enc)
        best = max(c[1] / max(1, len(c[0])) for c in ref["candidates"])
        mine = max(s.score_sequence(op, g["tokens"], e, bias) / max(1, len(g["tokens"])) for e in (True, False))
        assert mine > best - 0.1, (g, ref["tokens"], mine, best)
    ts = [t for t in g["tokens"] if t >= om.tok_beg]
    assert ts == sorted(ts)
    st.close(); st1.close(); st0.close(); om.close()


def test_config5_streaming_session_at_large_v3_runs_the_whole_stage_order(E):
    """BASELINE config #5 end to end at its own size: large-v3 dims, 5 s chunks through StreamingSession - chunk timer, noise
    reduction (the denoise hook with a frame-halving stand-in for the RNNoise network) -> normalise / compress / limit -> VAD hook
    -> mel + encoder + cross K/V -> beam = 5 on the two alternating step graphs -> tracker.  Every chunk must equal the same
    preprocessed samples taken through a fresh state as one window; a silent chunk is dropped by the VAD; the second session over
    the same audio (graphs cached) returns the same results."""
    from openhush_amd import streaming as S
    from openhush_amd.tracker import ChunkResult, TranscriptionTracker
    hp = synth.PRESETS["large-v3"]
    ctx = E.Context.synthetic(hp.as_list(), 1234, 0, E.OHW_DTYPE_BF16)
    tok = ctx.tok
    bias = np.zeros(hp.n_vocab, np.float32); bias[tok.timestamp_begin:] = 4.0; bias[tok.eot] = 9.0
    p = ctx.default_params(); p.n_max = 12
    cfg = E.default_preprocess_config(); cfg.preprocessing = 1
    half = lambda f: (f * np.float32(0.5)).astype(np.float32)           # noqa: E731
    n = 16000 * 5
    rec = np.concatenate([synth.synth_audio(70)[:n], np.zeros(n, np.float32), synth.synth_audio(71)[:n], synth.synth_audio(72)[:n]])
    vad = E.EnergyVad(-40.0)

    def run():
        ses = S.StreamingSession(ctx, beam_size=5, vad=vad, sequence_id=9, params=p, audio_config=cfg, noise_reduction=True,
                                 noise_reduction_strength=1.0, denoiser=E.Denoiser(half))
        ses.state.set_logit_bias(bias)               # end-of-text and timestamps win now and then: beams finish at different steps
        out = []
        for i in range(4):
            out += ses.tick(rec, (i + 1) * n, is_final=(i == 3))
        return ses, out
    ses, out = run()
    assert [r.chunk_id for r in out] == [0, 1, 2, 3] and ses.skipped_silent == 1 and out[1].text == ""
    assert ses.windows_decoded == 3 and any(r.text for r in out)
    _, again = run()
    assert [r.text for r in again] == [r.text for r in out]
    st = E.State(ctx, 5)
    st.set_logit_bias(bias)
    tr = TranscriptionTracker()
    for i in (0, 2, 3):
        buf = E.AudioBuffer(rec[i * n:(i + 1) * n].copy(), 16000)
        buf.preprocess(cfg, True, 1.0, E.Denoiser(half))
        st.mel(buf.samples[None, :], [len(buf.samples)], E.OHW_MEL_ZERO_TAIL, want=False); st.encode(1)
        toks = st.beam_search(1, 5, p)[0]["tokens"]
        text = b"".join(ctx.token_text(t) for t in toks if t < tok.eot).decode("utf-8", "replace").strip()
        tr.add_result(ChunkResult(text, 9, i))
    tr.add_result(ChunkResult("", 9, 1))                                # the chunk the VAD dropped
    want = tr.take_ready()                                              # the tracker's overlap de-duplication applied
    assert [r.text for r in want] == [r.text for r in out]
    st.close()
