"""Do two decodes overlap?  (development probe)

The greedy decode of 32 large-v3 windows alternates an HBM-bound kernel (cross-attention, 38 us per layer) with a chain of
latency-bound ones (six weight-streaming GEMMs + self-attention, ~44 us per layer).  Two decodes in flight could hide one's
latency chain under the other's K/V stream.  This probe times one decode alone and two decodes (two states, two host
threads, two streams) side by side: unrestricted streams, CU-masked halves, and a few split points.
    gpurun -- python tools/decode_overlap_probe.py [--tokens 60]
"""
import argparse, os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from openhush_amd import engine as E, synth

ap = argparse.ArgumentParser()
ap.add_argument("--tokens", type=int, default=60)
ap.add_argument("--batch", type=int, default=32)
a = ap.parse_args()
hp = synth.PRESETS["large-v3"]
ctx = E.Context.synthetic(hp.as_list(), 1234, 0, E.OHW_DTYPE_BF16)
B = a.batch
NMAX = 8
sts = [E.State(ctx, B) for _ in range(NMAX)]
pcm = torch.from_numpy(np.stack([synth.synth_audio(b) for b in range(B)])).cuda()
p = ctx.default_params(); p.force_len = a.tokens
for s in sts:
    s.mel_device(pcm.data_ptr(), pcm.shape[1], [synth.CHUNK_SAMPLES] * B, E.OHW_MEL_ZERO_TAIL)
    s.encode(B)
torch.cuda.synchronize()


def one(s):
    s.greedy(B, p)


def timed(fn, reps=3):
    fn()
    best = 1e9
    for _ in range(reps):
        torch.cuda.synchronize()
        t0 = time.perf_counter(); fn(); torch.cuda.synchronize()
        best = min(best, 1e3 * (time.perf_counter() - t0))
    return best


def side_by_side(n):
    th = [threading.Thread(target=one, args=(s,)) for s in sts[:n]]
    for t in th:
        t.start()
    for t in th:
        t.join()


t1 = timed(lambda: one(sts[0]))
print(f"one decode of {B} windows, {a.tokens} tokens, all CUs:      {t1:7.1f} ms", flush=True)
for n, splits in ((4, [(0, 64), (64, 64), (128, 64), (192, 64)]), (4, [(0, 128), (43, 128), (85, 128), (128, 128)]),
                  (4, [(0, 256), (0, 256), (0, 256), (0, 256)]), (6, [(i * 42, 42) for i in range(6)]), (8, [(i * 32, 32) for i in range(8)]),
                  (8, [(i * 32 - (32 if i else 0), 64) for i in range(8)])):
    ms = [E.Stream(0, f, c) for f, c in splits]
    for s, f in zip(sts, ms):
        s.set_stream(f.ptr)
    t3 = timed(lambda: side_by_side(n))
    print(f"{n} decodes side by side, CU ranges {splits}: {t3:7.1f} ms  ({n * t1 / t3:.2f}x of {n} in a row)", flush=True)
    for s in sts[:n]:
        s.set_stream(None)
    for f in ms:
        f.close()
