"""What a 256 x 256 tile of gemm256_kernel costs beyond its main loop: the same M x N at K = 1280, 2560 and 5120 through
ohw_dbg_gemm (bias + 16-bit store epilogue); time(2K) - time(K) is pure main loop, the rest of time(K) is fixed per tile
round (first-load latency, pipeline fill, epilogue, the launch's ramp and tail).

    gpurun -- python tools/gemm_probe.py [--windows 32]
"""
import argparse
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from openhush_amd import engine as E   # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--windows", type=int, default=32)
    ap.add_argument("--reps", type=int, default=20)
    a = ap.parse_args()
    L = E.lib()
    L.ohw_dbg_gemm.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_int64, C.c_int, C.c_void_p]
    M = a.windows * 1500
    s = torch.cuda.current_stream()
    for N in (1280, 3840, 5120):
        times = {}
        for K in (1280, 2560, 5120):
            A = (torch.rand(M, K, device="cuda") - 0.5).to(torch.bfloat16)
            W = (torch.rand(N, K, device="cuda") - 0.5).to(torch.bfloat16)
            bias = torch.zeros(N, device="cuda")
            out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
            run = lambda: E._check(L.ohw_dbg_gemm(E.OHW_DTYPE_BF16, A.data_ptr(), W.data_ptr(), bias.data_ptr(), out.data_ptr(), M, N, K, E.EPI_BIAS_T, s.cuda_stream))
            for _ in range(3):
                run()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(s)
            for _ in range(a.reps):
                run()
            e1.record(s)
            torch.cuda.synchronize()
            times[K] = e0.elapsed_time(e1) / a.reps * 1e3
            del A, W, out
        tiles = ((M + 255) // 256) * (N // 256)
        rounds = -(-tiles // 256)
        loop_1280 = times[2560] - times[1280]
        print(f"M {M} N {N}: {tiles} tiles = {tiles / 256:.2f} rounds;  K=1280 {times[1280]:8.1f} us ({2 * M * N * 1280 / times[1280] / 1e6:6.0f} TFLOP/s)  "
              f"K=2560 {times[2560]:8.1f} us ({2 * M * N * 2560 / times[2560] / 1e6:6.0f})  K=5120 {times[5120]:8.1f} us ({2 * M * N * 5120 / times[5120] / 1e6:6.0f});  "
              f"main loop of 20 K-tiles {loop_1280:7.1f} us = {2 * M * N * 1280 / loop_1280 / 1e6:6.0f} TFLOP/s, fixed per launch {times[1280] - loop_1280:6.1f} us "
              f"= {(times[1280] - loop_1280) / rounds:5.2f} us per tile round")


if __name__ == "__main__":
    main()
