#!/usr/bin/env python3
"""In-kernel timeline of one greedy decode step (instrumented build, development tool).

    OHW_BUILD_VARIANT=trace python -m openhush_amd.build        # here (hipcc cross-compiles)
    gpurun -- python tools/dec_trace.py [--model large-v3 --batch 32 --tokens 12]

Every decoder kernel of libohw_trace.so stamps the 100 MHz wall clock at entry (stage 0), after its LayerNorm
prologue (1), after the last MFMA / key block (2) and at exit (3), for its first, middle and last workgroup.  The
tool prints, for the last complete token step, one line per launch: the gap since the previous launch's latest
exit mark and the time between stages - where a 6 us kernel spends its 6 us.
"""
import argparse
import collections
import os
import sys

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
os.environ.setdefault("OHW_LIB", os.path.join(R, "openhush_amd", "libohw_trace.so"))

import ctypes as C  # noqa: E402

import numpy as np  # noqa: E402
import torch  # noqa: E402

from openhush_amd import engine as E, synth  # noqa: E402

NAMES = {2: "self_attn", 3: "cross_attn", 4: "sampler"}
EPI = {0: "QKV", 1: "BIAS_T", 2: "GELU", 3: "RESID", 4: "LOGITS"}


def kname(kid):
    if kid in NAMES:
        return NAMES[kid]
    e, ln = (kid - 16) // 2, (kid - 16) % 2
    return f"gemm<{EPI.get(e, e)}{',LN' if ln else ''}>"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="large-v3")
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--tokens", type=int, default=12)
    ap.add_argument("--dtype", default="bf16")
    ap.add_argument("--layers", type=int, default=2, help="how many layers of the step to print")
    a = ap.parse_args()
    # "small:1" = the preset with ONE decoder layer (its 16.5 MB of weights stay in the L2s from token to token: what a launch costs
    # when its weights need not come from beyond L2)
    name, _, nl = a.model.partition(":")
    hp = synth.PRESETS[name]
    if nl:
        import dataclasses
        hp = dataclasses.replace(hp, n_text_layer=int(nl))
    ctx = E.Context.synthetic(hp.as_list(), 1234, 0, E.OHW_DTYPE_BF16 if a.dtype == "bf16" else E.OHW_DTYPE_F16)
    st = E.State(ctx, a.batch)
    pcm = torch.from_numpy(np.stack([synth.synth_audio(b) for b in range(a.batch)])).cuda()
    p = ctx.default_params()
    p.force_len = a.tokens
    L = E.lib()
    L.ohw_dbg_trace_read.argtypes = [C.c_void_p, C.c_int]
    buf = np.zeros(1 << 18, np.uint64)
    for it in range(2):
        st.mel_device(pcm.data_ptr(), pcm.shape[1], [synth.CHUNK_SAMPLES] * a.batch, E.OHW_MEL_ZERO_TAIL)
        st.encode(a.batch)
        L.ohw_dbg_trace_read(buf.ctypes.data, len(buf))          # drop what the prompt pass recorded
        st.greedy(a.batch, p)
        n = L.ohw_dbg_trace_read(buf.ctypes.data, len(buf))
    rec = sorted((int(v) >> 16, (int(v) >> 8) & 255, (int(v) >> 2) & 63, int(v) & 3) for v in buf[:n])
    # launches in order of their earliest entry mark: group consecutive records of one kernel id
    launches = []
    cur_of = {}
    for t, kid, stage, which in rec:
        if stage == 0 and which == 0:          # workgroup 0 enters: a new launch of this kernel
            cur_of[kid] = {"id": kid, "t": collections.defaultdict(list)}
            launches.append(cur_of[kid])
        if kid in cur_of:
            cur_of[kid]["t"][stage].append(t)
    # one token step = the launches between two LOGITS gemms
    names = [kname(l["id"]) for l in launches]
    idx = [i for i, k in enumerate(names) if k.startswith("gemm<LOGITS") and (i + 1 == len(names) or not names[i + 1].startswith("gemm<LOGITS"))]
    if len(idx) < 3:
        print("too few steps recorded", len(launches))
        return
    lo, hi = idx[-2] + 1, idx[-1] + 1
    step = launches[lo:hi]
    print(f"# {a.model} B={a.batch}: {len(step)} traced launches in the last token step, "
          f"{(max(step[-1]['t'][3]) - min(step[0]['t'][0])) / 100:.1f} us entry-to-exit")
    print("# launch                first-entry  gap_prev_exit  entry_spread  ->LN   ->mfma_done  ->exit(first wg)  last_exit-first_entry")
    prev_exit = None
    per = collections.defaultdict(lambda: [0, 0.0, 0.0])
    t0 = min(step[0]["t"][0])
    shown = 0
    for l in step:
        t = l["t"]
        e0, e1 = min(t[0]), max(t[0])
        x1 = max(t[3]) if t[3] else e1
        gap = (e0 - prev_exit) / 100 if prev_exit is not None else 0.0
        ln = (min(t[1]) - e0) / 100 if t[1] else float("nan")
        mf = (min(t[2]) - e0) / 100 if t[2] else float("nan")
        fx = (min(t[3]) - e0) / 100 if t[3] else float("nan")
        k = kname(l["id"])
        per[k][0] += 1
        per[k][1] += gap
        per[k][2] += (x1 - e0) / 100
        if shown < a.layers * 8 + 1:
            print(f"{k:22s} {(e0 - t0) / 100:10.2f} {gap:12.2f} {(e1 - e0) / 100:12.2f} {ln:8.2f} {mf:10.2f} {fx:12.2f} {(x1 - e0) / 100:18.2f}")
            shown += 1
        prev_exit = x1
    print("# per kernel over the step: launches, mean gap before entry (us), mean entry->last exit (us)")
    for k, (c, g, d) in per.items():
        print(f"{k:22s} {c:4d} {g / c:8.2f} {d / c:8.2f}")


if __name__ == "__main__":
    main()
