#!/usr/bin/env python3
"""Where a 256 x 256 tile of gemm256_kernel spends its time (instrumented build, development tool).

    OHW_BUILD_VARIANT=trace python -m openhush_amd.build
    gpurun -- python tools/gemm_trace.py [--windows 96]

Every workgroup stamps the 100 MHz clock at entry, when its first K-tile has landed, after the main loop and at exit (all
stores drained), with its XCC / CU id.  Printed per shape and epilogue: medians of the three segments, and the gap between a
workgroup's exit and the entry of the next workgroup on the SAME CU (dispatch latency)."""
import argparse
import ctypes as C
import os
import sys

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
os.environ.setdefault("OHW_LIB", os.path.join(R, "openhush_amd", "libohw_trace.so"))

import numpy as np  # noqa: E402
import torch  # noqa: E402

from openhush_amd import engine as E  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--windows", type=int, default=96)
    a = ap.parse_args()
    L = E.lib()
    L.ohw_dbg_gemm.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_int64, C.c_int, C.c_void_p]
    L.ohw_dbg_gemm_trace_read.argtypes = [C.c_void_p, C.c_int]
    M = a.windows * 1500
    s = torch.cuda.current_stream()
    buf = np.zeros((1 << 16, 12), np.uint64)
    EPI = {E.EPI_BIAS_T: "BIAS_T", E.EPI_BIAS_GELU_T: "GELU", E.EPI_BIAS_RESID_F32: "RESID"}
    print(f"M = {M} ({a.windows} windows); us; med [p10 .. p90]")
    ap_shapes = ((3840, 1280, -E.EPI_BIAS_T - 100), (3840, 1280, E.EPI_BIAS_T), (3840, 1280, -E.EPI_BIAS_T - 100), (3840, 1280, E.EPI_BIAS_T)) if os.environ.get("GT_BIAS_AB") else None
    for (N, K, epi) in ap_shapes or ((3840, 1280, E.EPI_BIAS_T), (5120, 1280, E.EPI_BIAS_GELU_T), (1280, 1280, E.EPI_BIAS_T), (1280, 1280, E.EPI_BIAS_RESID_F32),
                        (1280, 5120, E.EPI_BIAS_RESID_F32)):
        A = (torch.rand(M, K, device="cuda") - 0.5).to(torch.bfloat16)
        W = (torch.rand(N, K, device="cuda") - 0.5).to(torch.bfloat16)
        no_bias = epi < 0
        if no_bias:
            epi = -(epi + 100)
        bias = torch.zeros(N, device="cuda")
        out = torch.zeros(M, N, device="cuda", dtype=torch.float32 if epi == E.EPI_BIAS_RESID_F32 else torch.bfloat16)
        run = lambda: E._check(L.ohw_dbg_gemm(E.OHW_DTYPE_BF16, A.data_ptr(), W.data_ptr(), (None if no_bias else bias.data_ptr()), out.data_ptr(), M, N, K, epi, s.cuda_stream))
        run(); run()
        torch.cuda.synchronize()
        L.ohw_dbg_gemm_trace_read(buf.ctypes.data, buf.shape[0])
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(s); run(); e1.record(s)
        torch.cuda.synchronize()
        n = L.ohw_dbg_gemm_trace_read(buf.ctypes.data, buf.shape[0])
        r = buf[:n].astype(np.int64)
        t = r[:, 2:6].astype(np.float64) / 100.0
        t -= t[:, 0].min()
        q = lambda x: f"{np.median(x):6.2f} [{np.percentile(x, 10):6.2f} .. {np.percentile(x, 90):6.2f}]"
        cu = r[:, 1] & ((0xf << 32) | 0xff00)          # XCC id | HW_ID's se / sh / cu fields
        gaps = []
        for c in np.unique(cu):
            idx = np.where(cu == c)[0]
            o = idx[np.argsort(t[idx, 0])]
            gaps += list(t[o[1:], 0] - t[o[:-1], 3])
        ms = e0.elapsed_time(e1)
        print(f"N {N:5d} K {K:5d} {EPI[epi] + ('-nob' if no_bias else ''):6s}: {n:6d} tiles on {len(np.unique(cu))} CUs, launch {ms * 1e3:8.1f} us = {2.0 * M * N * K / ms / 1e9:6.0f} TFLOP/s | "
              f"entry->first data {q(t[:, 1] - t[:, 0])} | main loop {q(t[:, 2] - t[:, 1])} | epilogue {q(t[:, 3] - t[:, 2])} | exit->next entry on the CU {q(np.array(gaps))} | "
              f"last exit {t[:, 3].max():8.1f}")
        if epi != E.EPI_BIAS_RESID_F32 and r[:, 9].any():
            print("        LDS-staged store: first barrier passed %.2f | tile converted and in LDS (this wave) %.2f | second barrier passed %.2f (us after the main loop, med)" % tuple(np.median((r[:, [9, 7, 8]] - r[:, [4, 4, 4]]) / 100.0, axis=0)))
        if epi != E.EPI_BIAS_RESID_F32:
            print("        stores issued %.2f us after the main loop (the rest of the epilogue figure is the wait for their acknowledgement, which only this build makes)" % np.median((r[:, 6] - r[:, 4]) / 100.0))
        if epi == E.EPI_BIAS_RESID_F32:
            e = r[:, 6:12].astype(np.float64) / 100.0 - (r[:, 4].astype(np.float64) / 100.0)[:, None]
            print("        RESID epilogue stamps after the main loop (med): round 0 loads issued %.2f | round 0 in LDS %.2f | round 0 stored %.2f | round 1 in LDS %.2f | round 1 stored %.2f" % tuple(np.median(e, axis=0)[:5]))
        del A, W, out


if __name__ == "__main__":
    main()
