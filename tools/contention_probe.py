"""What slows the decoder down beside the encoder?  (development probe)

Decode of 32 large-v3 windows on 160 CUs, alone and beside (a) the encoder's own GEMM on the other 96 CUs, (b) a
device-to-device copy loop (HBM traffic, no MFMA), both driven from a second host thread.
    gpurun -- python tools/contention_probe.py
"""
import os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from openhush_amd import engine as E, synth

hp = synth.PRESETS["large-v3"]
ctx = E.Context.synthetic(hp.as_list(), 1234, 0, E.OHW_DTYPE_BF16)
B = 32
st = E.State(ctx, B)
pcm = torch.from_numpy(np.stack([synth.synth_audio(b) for b in range(B)])).cuda()
p = ctx.default_params(); p.force_len = 60
es, ds = E.Stream(0, 0, 96), E.Stream(0, 96, 160)
st.mel_device(pcm.data_ptr(), pcm.shape[1], [synth.CHUNK_SAMPLES] * B, E.OHW_MEL_ZERO_TAIL)
st.encode(B)
torch.cuda.synchronize()
st.set_stream(ds.ptr)
L = E.lib()
M, N, K = 48000, 3840, 1280
A = (torch.randn(M, K, device="cuda") * 0.5).to(torch.bfloat16)
W = (torch.randn(N, K, device="cuda") / K ** 0.5).to(torch.bfloat16)
bias = torch.randn(N, device="cuda")
out = torch.zeros(M, N, device="cuda", dtype=torch.bfloat16)
src = torch.empty(1 << 30, dtype=torch.uint8, device="cuda"); dst = torch.empty_like(src)
stop = False

def burn_gemm():
    while not stop:
        for _ in range(8):
            L.ohw_dbg_gemm(0, A.data_ptr(), W.data_ptr(), bias.data_ptr(), out.data_ptr(), M, N, K, E.EPI_BIAS_T, es.h)
        es.sync()

def burn_copy():
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        while not stop:
            for _ in range(4):
                dst.copy_(src, non_blocking=True)
            s.synchronize()

def decode_ms():
    st.greedy(B, p)
    t0 = time.perf_counter(); st.greedy(B, p); return 1e3 * (time.perf_counter() - t0)

print(f"decode alone (160 CUs):          {decode_ms():.1f} ms")
for name, fn in (("encoder GEMM on 96 CUs", burn_gemm), ("device-to-device copy loop", burn_copy)):
    stop = False
    th = threading.Thread(target=fn); th.start(); time.sleep(0.3)
    print(f"decode beside {name:27s}: {decode_ms():.1f} ms")
    stop = True; th.join(); torch.cuda.synchronize()
