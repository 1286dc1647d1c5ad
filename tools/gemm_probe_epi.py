"""GPU probe: the encoder's N = 1280 GEMM shapes with the bias -> T epilogue against the fp32 residual read-modify-write."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from openhush_amd import engine as E

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
L = E.lib()
for (M, N, K) in [(48000, 1280, 1280), (48000, 1280, 5120), (48000, 3840, 1280)]:
    A = (torch.randn(M, K, device="cuda") * 0.5).to(torch.bfloat16)
    W = (torch.randn(N, K, device="cuda") / K ** 0.5).to(torch.bfloat16)
    bias = torch.randn(N, device="cuda")
    s = torch.cuda.current_stream().cuda_stream
    for name, epi, out in (("bias->T", E.EPI_BIAS_T, torch.zeros(M, N, device="cuda", dtype=torch.bfloat16)),
                           ("resid f32", E.EPI_BIAS_RESID_F32, torch.zeros(M, N, device="cuda", dtype=torch.float32))):
        for _ in range(2):
            assert L.ohw_dbg_gemm(0, A.data_ptr(), W.data_ptr(), bias.data_ptr(), out.data_ptr(), M, N, K, epi, s) == 0
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            L.ohw_dbg_gemm(0, A.data_ptr(), W.data_ptr(), bias.data_ptr(), out.data_ptr(), M, N, K, epi, s)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / reps
        print(f"M={M} N={N} K={K} {name:10s}: {ms*1e3:7.1f} us  {2*M*N*K/ms/1e9:7.1f} TFLOP/s", flush=True)
