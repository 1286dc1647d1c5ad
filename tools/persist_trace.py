#!/usr/bin/env python3
"""In-kernel timeline of the persistent small-batch decoder step (instrumented build, development tool).

    OHW_BUILD_VARIANT=trace python -m openhush_amd.build
    gpurun -- python tools/persist_trace.py [--model large-v3 --rows 1 --group 1 --tokens 40]

decode_persist.hip stamps the 100 MHz clock for the middle layer of the LAST step: per workgroup and phase, the first IO
wave (phase entered, inputs gathered, partial tiles ready, published) and the first MFMA wave (weights requested, inputs
ready, weights landed, products done).  Printed: per phase, when the workgroups entered / published it relative to the
layer's first stamp, and the medians of the segments between stamps.
"""
import argparse
import os
import sys

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
os.environ.setdefault("OHW_LIB", os.path.join(R, "openhush_amd", "libohw_trace.so"))

import ctypes as C  # noqa: E402

import numpy as np  # noqa: E402

from openhush_amd import engine as E, synth  # noqa: E402

PH = "A:LN1+QKV B:self-attn C:out-proj D:LNx+xq E:x-attn F:merge G:x-out H:LN2+fc1 I:fc2 J:x+=".split()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="large-v3")
    ap.add_argument("--rows", type=int, default=1)
    ap.add_argument("--tokens", type=int, default=40)
    a = ap.parse_args()
    hp = synth.PRESETS[a.model]
    ctx = E.Context.synthetic(hp.as_list(), 1234, 0, E.OHW_DTYPE_BF16)
    st = E.State(ctx, a.rows)
    pcm = np.stack([synth.synth_audio(b) for b in range(a.rows)])
    st.mel(pcm, None, E.OHW_MEL_ZERO_TAIL, want=False)
    st.encode(a.rows)
    p = ctx.default_params()
    p.force_len = a.tokens
    import time
    st.greedy(a.rows, p)
    t0 = time.perf_counter()
    st.greedy(a.rows, p)
    dt = time.perf_counter() - t0
    print(f"{a.model} rows {a.rows}: {dt * 1e3 / a.tokens:.3f} ms per token (host wall, {a.tokens} tokens, persistent launches {st.counter('persist_launches')})")
    L = E.lib()
    L.ohw_dbg_persist_trace_read.argtypes = [C.c_void_p, C.c_int]
    buf = np.zeros(256 * 10 * 8, np.uint64)
    n = L.ohw_dbg_persist_trace_read(buf.ctypes.data, buf.size)
    assert n == buf.size
    t = buf.reshape(256, 10, 8).astype(np.float64) / 100.0       # us
    t[t == 0] = np.nan
    base = np.nanmin(t)
    t -= base
    print("times in us relative to the layer's first stamp; med = median over the workgroups that had a task")
    print(f"{'phase':14s} {'n':>4s} | {'enter min/med/max':>22s} | {'publish min/med/max':>22s} | gather  wait#2  epilog | w-req->in-ready  w-landed-after-ready  mfma")
    prev_pub = 0.0
    for ph in range(10):
        e, g, m2, pub = t[:, ph, 0], t[:, ph, 1], t[:, ph, 2], t[:, ph, 3]
        wq, wr, wl, wd = t[:, ph, 4], t[:, ph, 5], t[:, ph, 6], t[:, ph, 7]
        has = ~np.isnan(e)
        if not has.any():
            continue
        f = lambda x: f"{np.nanmin(x):6.2f}/{np.nanmedian(x):6.2f}/{np.nanmax(x):6.2f}"
        md = lambda x: f"{np.nanmedian(x):6.2f}" if (~np.isnan(x)).any() else "     -"
        print(f"{PH[ph]:14s} {int(has.sum()):4d} | {f(e):>22s} | {f(pub):>22s} | {md(g - e)} {md(m2 - g)} {md(pub - m2 if (~np.isnan(m2)).any() else pub - g)} | {md(wr - wq)} {md(wl - wr)} {md(wd - wl)}")
    print(f"layer span {np.nanmax(t):.2f} us")


if __name__ == "__main__":
    main()
