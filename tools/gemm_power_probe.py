#!/usr/bin/env python3
"""Does gemm256_kernel run power-limited?  One shape in a loop for some seconds; every 0.5 s the rate of the last launches and what
rocm-smi reports (clocks, power).   gpurun -- python tools/gemm_power_probe.py [--seconds 6] [--decode: a 32-window greedy decode in a
loop instead - what the HBM-bound phase draws]"""
import argparse, ctypes as C, os, subprocess, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from openhush_amd import engine as E   # noqa: E402


def smi():
    try:
        o = subprocess.run(["rocm-smi", "--showclocks", "--showpower", "--showtemp", "--json"], capture_output=True, text=True, timeout=10).stdout
        import json
        d = json.loads(o)
        c = d.get("card0", {})
        keep = {k: v for k, v in c.items() if any(t in k.lower() for t in ("sclk", "mclk", "power", "junction", "hotspot", "edge"))}
        return keep
    except Exception as e:  # noqa
        return {"err": str(e)[:80]}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=6.0)
    ap.add_argument("--windows", type=int, default=96)
    ap.add_argument("--decode", action="store_true")
    a = ap.parse_args()
    if a.decode:
        import numpy as np
        from openhush_amd import synth
        hp = synth.PRESETS["large-v3"]
        ctx = E.Context.synthetic(hp.as_list(), 1234, 0, E.OHW_DTYPE_BF16)
        st = E.State(ctx, 32)
        pcm = np.stack([synth.synth_audio(b) for b in range(32)])
        st.mel(pcm, None, E.OHW_MEL_ZERO_TAIL, want=False); st.encode(32)
        p = ctx.default_params(); p.force_len = 100
        print("idle:", smi(), flush=True)
        import threading
        stop = [False]
        samples = []

        def sampler():
            while not stop[0]:
                samples.append(smi()); time.sleep(0.3)
        th = threading.Thread(target=sampler); th.start()
        t_end = time.time() + a.seconds
        n = 0
        t0 = time.time()
        while time.time() < t_end:
            st.greedy(32, p); n += 1
        dt = time.time() - t0
        stop[0] = True; th.join()
        print(f"{n} decodes of 32 windows x 100 tokens: {1e3 * dt / n:.1f} ms each")
        for smp in samples:
            print(smp, flush=True)
        return
    L = E.lib()
    L.ohw_dbg_gemm.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_int64, C.c_int, C.c_void_p]
    M, N, K = a.windows * 1500, 3840, 1280
    A = (torch.rand(M, K, device="cuda") - 0.5).to(torch.bfloat16)
    W = (torch.rand(N, K, device="cuda") - 0.5).to(torch.bfloat16)
    bias = torch.zeros(N, device="cuda")
    out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    s = torch.cuda.current_stream()
    run = lambda: E._check(L.ohw_dbg_gemm(E.OHW_DTYPE_BF16, A.data_ptr(), W.data_ptr(), bias.data_ptr(), out.data_ptr(), M, N, K, E.EPI_BIAS_T, s.cuda_stream))
    print("idle:", smi(), flush=True)
    t_end = time.time() + a.seconds
    k = 0
    while time.time() < t_end:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(s)
        n = 200
        for _ in range(n):
            run()
        e1.record(s)
        info = smi()                       # sampled while the launches are still queued / running
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / n
        print(f"t={k:2d}: {ms * 1e3:7.1f} us per launch = {2.0 * M * N * K / ms / 1e9:6.0f} TFLOP/s  {info}", flush=True)
        k += 1


if __name__ == "__main__":
    main()
