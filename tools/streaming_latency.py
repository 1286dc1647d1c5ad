"""Config #5 on one MI355X: large-v3, streaming chunks, beam = 5 (two alternating hipGraph-captured decoder steps), energy VAD
in front (the reference's RNNoise and Silero models are not available offline).  A chunk timer of --chunk
seconds over a synthetic recording; prints the per-chunk latency of StreamingSession.transcribe_job (mel + encoder +
cross-K/V + beam search of --tokens forced steps, detokenise) and the real-time factor of the stream.

    gpurun -- python tools/streaming_latency.py [--chunk 5 --chunks 12 --beam 5 --tokens 48]
"""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from openhush_amd import engine as E, streaming as S, synth   # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="large-v3")
    ap.add_argument("--chunk", type=float, default=5.0)
    ap.add_argument("--chunks", type=int, default=12)
    ap.add_argument("--beam", type=int, default=5)
    ap.add_argument("--tokens", type=int, default=48, help="decoder steps per chunk (synthetic weights never emit end-of-text by themselves)")
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "f16"])
    a = ap.parse_args()
    ctx = E.Context.synthetic(synth.PRESETS[a.model].as_list(), 1234, 0, E.OHW_DTYPE_BF16 if a.dtype == "bf16" else E.OHW_DTYPE_F16)
    p = ctx.default_params()
    if a.beam >= 2:
        p.n_max = a.tokens          # beams end at n_max (force_len is the greedy loop's knob)
    else:
        p.force_len = a.tokens
    n = int(a.chunk * 16000)
    rec = np.concatenate([synth.synth_audio(40 + i)[:n] for i in range(a.chunks)])
    vad = E.EnergyVad(-40.0)
    ses = S.StreamingSession(ctx, beam_size=a.beam, vad=vad, params=p)
    lat = []
    for i in range(a.chunks):
        job = ses.scheduler.tick(rec, (i + 1) * n)
        t0 = time.perf_counter()
        r = ses.transcribe_job(job)
        lat.append(time.perf_counter() - t0)
        ses.tracker.add_result(r)
        ses.tracker.take_ready()
    warm = lat[2:] or lat[-1:]        # the first chunks capture the two beam-step graphs (a one-chunk run - profiling - reports that chunk)
    print(json.dumps({"workload": f"{a.model} streaming, {a.chunk:g} s chunks, beam {a.beam}, {a.tokens} decoder steps per chunk, {a.dtype}",
                      "first_chunk_ms": round(1e3 * lat[0], 2), "chunk_latency_ms": {"mean": round(1e3 * float(np.mean(warm)), 2),
                      "min": round(1e3 * min(warm), 2), "max": round(1e3 * max(warm), 2)},
                      "stream_real_time_factor": round(a.chunk / float(np.mean(warm)), 1), "silent_chunks_skipped": ses.skipped_silent}))


if __name__ == "__main__":
    main()
